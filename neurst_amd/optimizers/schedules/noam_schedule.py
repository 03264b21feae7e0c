"""`noam` learning-rate schedule (reference: neurst/optimizers/schedules/noam_schedule.py:22-108).

    s      = global_step + registered initial step + 1
    factor = end + (initial - end) * (1 - clamp(s - start_decay_at, 0, decay_steps) / decay_steps)
    lr     = factor * d_model^-0.5 * min(1, s / warmup) / sqrt(max(s, warmup))

i.e. linear warm-up, reciprocal-square-root decay, and a multiplier that moves linearly from `initial_factor` to
`end_factor` over `decay_steps` steps starting at `start_decay_at` (no move unless all three are given).  Host
arithmetic in double precision -- the value is a scalar argument of the fused Adam kernel; pinned on the reference class
(float32) by tests/golden/lr_schedules.json.
"""
import math

from neurst_amd.optimizers.registries import register_lr_schedule
from neurst_amd.utils import compat
from neurst_amd.utils.flags_core import Flag


@register_lr_schedule("noam")
class NoamSchedule(object):
    def __init__(self, args):
        self._dmodel = args["dmodel"]
        self._warmup = float(args["warmup_steps"])
        self._offset = float(compat.get_registered_initial_step()) + 1.0
        self._f0 = float(args["initial_factor"])
        moving = all(args.get(k, None) is not None for k in ("end_factor", "start_decay_at", "decay_steps"))
        self._f1 = float(args["end_factor"]) if moving else self._f0
        self._t0 = float(args["start_decay_at"]) if moving else 0.0
        self._span = float(args["decay_steps"]) if moving else 1.0

    @staticmethod
    def class_or_method_args():
        F = Flag
        return [
            F("dmodel", dtype=F.TYPE.INTEGER, default=None, help="d_model: the rate is scaled by d_model ** -0.5."),
            F("warmup_steps", dtype=F.TYPE.INTEGER, default=4000, help="Steps of linear warm-up before the 1/sqrt(step) decay."),
            F("initial_factor", dtype=F.TYPE.FLOAT, default=1., help="Multiplier of the whole schedule at the start."),
            F("end_factor", dtype=F.TYPE.FLOAT, default=None, help="Multiplier after the linear factor decay (default: no decay)."),
            F("start_decay_at", dtype=F.TYPE.INTEGER, default=0, help="Step at which the multiplier starts to move towards end_factor."),
            F("decay_steps", dtype=F.TYPE.INTEGER, default=None, help="Length of that move in steps."),
        ]

    def __call__(self, global_step):
        s = float(global_step) + self._offset
        progress = min(max(s - self._t0, 0.0), self._span) / self._span
        factor = self._f1 + (self._f0 - self._f1) * (1.0 - progress)
        return factor * self._dmodel ** -0.5 * min(1.0, s / self._warmup) / math.sqrt(max(s, self._warmup))

    def get_config(self):
        return {"initial_factor": self._f0, "dmodel": self._dmodel, "warmup_steps": int(self._warmup), "end_factor": self._f1,
                "start_decay_at": int(self._t0), "decay_steps": int(self._span)}
