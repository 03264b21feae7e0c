from neurst_amd.criterions.criterion import Criterion, build_criterion, register_criterion  # noqa: F401
from neurst_amd.criterions import label_smoothed_cross_entropy  # noqa: F401
