"""PiecewiseSchedule (neurst/optimizers/schedules/piecewise_schedule.py:23-90): linear warm-up to schedule_lrs[0] over
schedule_steps[0] steps, then piecewise-constant rates.

Bug-compatible on purpose: the reference builds its `tf.case` branches in a loop as `lambda: tf.constant(lr)` (:78-80), so
every middle branch returns the LAST middle rate (Python closures bind late).  With steps [s0, s1, s2] and rates
[a, b, c, d] the reference yields warm-up to a, then c on [s0, s2), then d -- `b` is never used.  The golden values in
tests/golden/lr_schedules.json come from the reference class itself; `strict=True` gives the documented intent instead."""
import yaml

from neurst_amd.optimizers.registries import register_lr_schedule
from neurst_amd.utils import compat
from neurst_amd.utils.flags_core import Flag


def _as_list(v):
    return yaml.safe_load(v) if isinstance(v, str) else list(v)


@register_lr_schedule("piecewise")
class PiecewiseSchedule(object):
    def __init__(self, args, strict=False):
        self._strict = bool(strict or args.get("strict", False))
        self._schedule_steps, self._schedule_lrs = _as_list(args["schedule_steps"]), _as_list(args["schedule_lrs"])
        assert len(self._schedule_steps) + 1 == len(self._schedule_lrs)
        self._initial_step = float(compat.get_registered_initial_step())

    @staticmethod
    def class_or_method_args():
        return [Flag("schedule_steps", dtype=Flag.TYPE.STRING, default=None, help="A list of triggered steps."),
                Flag("schedule_lrs", dtype=Flag.TYPE.STRING, default=None, help="A list of learning rates.")]

    def __call__(self, global_step):
        s = float(global_step) + self._initial_step + 1.
        if s < self._schedule_steps[0]:
            return self._schedule_lrs[0] / float(self._schedule_steps[0]) * s
        middle = self._schedule_lrs[1:-1]
        for step, lr in zip(self._schedule_steps[1:], middle):
            if s < step:
                return float(lr if self._strict else middle[-1])
        return float(self._schedule_lrs[-1])

    def get_config(self):
        return {"schedule_steps": self._schedule_steps, "schedule_lrs": self._schedule_lrs}
