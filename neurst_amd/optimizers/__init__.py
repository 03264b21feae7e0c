from neurst_amd.optimizers.registries import (build_lr_schedule, build_optimizer,  # noqa: F401
                                              register_lr_schedule, register_optimizer)
from neurst_amd.optimizers import adam  # noqa: F401
from neurst_amd.optimizers.schedules import inverse_sqrt_schedule, noam_schedule, piecewise_schedule  # noqa: F401
