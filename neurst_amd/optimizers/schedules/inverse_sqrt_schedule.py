"""InverseSquareRootSchedule (neurst/optimizers/schedules/inverse_sqrt_schedule.py:23-79): linear warm-up from `init_lr`
to `peak_lr` over `warmup_steps`, then peak_lr * sqrt(warmup_steps / step).  Host arithmetic like NoamSchedule."""
from neurst_amd.optimizers.registries import register_lr_schedule
from neurst_amd.utils import compat
from neurst_amd.utils.flags_core import Flag


@register_lr_schedule("inverse_sqrt")
class InverseSquareRootSchedule(object):
    def __init__(self, args):
        self._initial_step = float(compat.get_registered_initial_step())
        self._lr, self._init_lr = float(args["peak_lr"]), float(args["init_lr"])
        self._warmup_steps = float(args["warmup_steps"])
        self._lr_step = (self._lr - self._init_lr) / self._warmup_steps
        self._decay_factor = self._lr * self._warmup_steps ** 0.5

    @staticmethod
    def class_or_method_args():
        return [Flag("peak_lr", dtype=Flag.TYPE.FLOAT, default=5e-4, help="The configured lr."),
                Flag("init_lr", dtype=Flag.TYPE.FLOAT, default=0., help="The initial lr."),
                Flag("warmup_steps", dtype=Flag.TYPE.INTEGER, default=4000,
                     help="The number of steps required for linear warmup.")]

    def __call__(self, global_step):
        s = float(global_step) + self._initial_step + 1.
        if s < self._warmup_steps:
            return self._init_lr + s * self._lr_step
        return self._decay_factor * s ** -0.5

    def get_config(self):
        return {"peak_lr": self._lr, "init_lr": self._init_lr, "warmup_steps": int(self._warmup_steps)}
