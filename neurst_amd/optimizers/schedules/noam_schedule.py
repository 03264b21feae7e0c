"""NoamSchedule (neurst/optimizers/schedules/noam_schedule.py:22-108): linear warm-up, rsqrt decay, with a scaling
factor that decays linearly from initial_factor to end_factor.  Pure host arithmetic (float32 semantics are not
needed: the value is a scalar handed to the fused Adam kernel)."""
import math

from neurst_amd.optimizers.registries import register_lr_schedule
from neurst_amd.utils import compat
from neurst_amd.utils.flags_core import Flag


@register_lr_schedule("noam")
class NoamSchedule(object):
    def __init__(self, args):
        self._dmodel = args["dmodel"]
        self._warmup_steps = float(args["warmup_steps"])
        self._initial_step = float(compat.get_registered_initial_step())
        self._initial_learning_rate = float(args["initial_factor"])
        end = args.get("end_factor", None)
        if end is not None and args.get("start_decay_at", None) is not None and args.get("decay_steps", None) is not None:
            start_decay_at, decay_steps = args["start_decay_at"], args["decay_steps"]
        else:
            end, start_decay_at, decay_steps = self._initial_learning_rate, 0, 1
        self._end_learning_rate, self._start_decay_at, self._decay_steps = float(end), float(start_decay_at), float(decay_steps)

    @staticmethod
    def class_or_method_args():
        return [
            Flag("dmodel", dtype=Flag.TYPE.INTEGER, default=None, help="d_model: the rate is scaled by d_model ** -0.5."),
            Flag("warmup_steps", dtype=Flag.TYPE.INTEGER, default=4000, help="Steps of linear warm-up before the 1/sqrt(step) decay."),
            Flag("initial_factor", dtype=Flag.TYPE.FLOAT, default=1., help="Multiplier of the whole schedule at the start."),
            Flag("end_factor", dtype=Flag.TYPE.FLOAT, default=None, help="Multiplier after the linear factor decay (default: no decay)."),
            Flag("start_decay_at", dtype=Flag.TYPE.INTEGER, default=0, help="Step at which the multiplier starts to move towards end_factor."),
            Flag("decay_steps", dtype=Flag.TYPE.INTEGER, default=None, help="Length of that move in steps."),
        ]

    def __call__(self, global_step):
        """noam_schedule.py:76-97."""
        s = float(global_step) + self._initial_step + 1.
        step_factor = max(min(s - self._start_decay_at, self._decay_steps), 0.)
        lr = self._end_learning_rate + (self._initial_learning_rate - self._end_learning_rate) * (
            1. - step_factor / self._decay_steps)
        lr *= self._dmodel ** -0.5
        lr *= min(1.0, s / self._warmup_steps)
        lr /= math.sqrt(max(s, self._warmup_steps))
        return lr

    def get_config(self):
        return {"initial_factor": self._initial_learning_rate, "dmodel": self._dmodel,
                "warmup_steps": int(self._warmup_steps), "end_factor": self._end_learning_rate,
                "start_decay_at": int(self._start_decay_at), "decay_steps": int(self._decay_steps)}
