"""`inverse_sqrt` learning-rate schedule (reference: neurst/optimizers/schedules/inverse_sqrt_schedule.py:23-79).

    step s = global_step + registered initial step + 1
    s <  warmup_steps :  init_lr + s * (peak_lr - init_lr) / warmup_steps        (linear warm-up)
    s >= warmup_steps :  peak_lr * sqrt(warmup_steps / s)                        (reciprocal square-root decay)

Plain host arithmetic (the value is a scalar argument of the fused Adam kernel); pinned on the reference class by
tests/golden/lr_schedules.json.
"""
from neurst_amd.optimizers.registries import register_lr_schedule
from neurst_amd.utils import compat
from neurst_amd.utils.flags_core import Flag


@register_lr_schedule("inverse_sqrt")
class InverseSquareRootSchedule(object):
    def __init__(self, args):
        self._offset = float(compat.get_registered_initial_step()) + 1.0
        self._peak, self._init = float(args["peak_lr"]), float(args["init_lr"])
        self._warmup = float(args["warmup_steps"])

    @staticmethod
    def class_or_method_args():
        F = Flag
        return [F("peak_lr", dtype=F.TYPE.FLOAT, default=5e-4, help="Rate reached at the end of the warm-up."),
                F("init_lr", dtype=F.TYPE.FLOAT, default=0., help="Rate the warm-up starts from."),
                F("warmup_steps", dtype=F.TYPE.INTEGER, default=4000, help="Length of the linear warm-up in steps.")]

    def __call__(self, global_step):
        s = float(global_step) + self._offset
        if s < self._warmup:
            return self._init + s * (self._peak - self._init) / self._warmup
        return self._peak * (self._warmup / s) ** 0.5

    def get_config(self):
        return {"peak_lr": self._peak, "init_lr": self._init, "warmup_steps": int(self._warmup)}
