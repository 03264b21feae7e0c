"""CriterionValidator (neurst/training/validator.py:25-60, neurst/training/criterion_validator.py:27-160): every
`eval_steps` optimizer steps the criterion is evaluated on a held-out dataset (forward only, no dropout), the metrics
are logged with the best value so far, and the checkpoints with the best metric are kept under `<model_dir>/best`
(TrainingStatusRecorder + KeepBestCheckpointSaver, training_utils.py:274-360, checkpoints.py:186-237).
"""
import logging
import os
import time

import torch

from neurst_amd.criterions import Criterion, build_criterion
from neurst_amd.data.datasets import Dataset, build_dataset
from neurst_amd.utils import compat
from neurst_amd.utils.checkpoints import KeepBestCheckpointSaver
from neurst_amd.utils.flags_core import Flag, ModuleFlag
from neurst_amd.utils.registry import setup_registry


class Validator(object):
    REGISTRY_NAME = "validator"

    def __init__(self, args):
        self._eval_steps = args.get("eval_steps", None) or 1000
        self._eval_start_at = args.get("eval_start_at", None) or 0
        self._eval_on_begin = bool(args.get("eval_on_begin", False))

    @staticmethod
    def class_or_method_args():
        return [Flag("eval_steps", dtype=Flag.TYPE.INTEGER, default=1000, help="The steps between two validation steps."),
                Flag("eval_start_at", dtype=Flag.TYPE.INTEGER, default=0, help="The step to start validation process."),
                Flag("eval_on_begin", dtype=Flag.TYPE.BOOLEAN, default=False, help="Whether to trigger evaluation on the beginning.")]

    def due(self, step):
        return step >= self._eval_start_at and step % self._eval_steps == 0

    def build(self, task, model, model_dir):
        raise NotImplementedError

    def validate(self, step):
        raise NotImplementedError


build_validator, register_validator = setup_registry(Validator.REGISTRY_NAME, base_class=Validator, backend="pt")


class _AsMetric(object):
    """The comparison interface the checkpoint savers need (flag value of a result dict, direction) from a criterion's
    as_metric() descriptor."""

    def __init__(self, metric):
        self.flag, self.greater_is_better = metric.flag, metric.greater_is_better

    def get_value(self, result):
        return float(result if isinstance(result, (int, float)) else result[self.flag])

    def greater_or_eq(self, a, b):
        a, b = self.get_value(a), self.get_value(b)
        return a >= b if self.greater_is_better else a <= b


@register_validator(["criterion", "CriterionValidator"])
class CriterionValidator(Validator):
    def __init__(self, args):
        super().__init__(args)
        self.args = args
        self._eval_task_args = dict(args.get("eval_task_args", None) or {})
        self._eval_task_args["batch_size"] = args.get("eval_batch_size", None) or 32
        self._top_keep = args.get("eval_top_checkpoints_to_keep", None) or 0
        self._criterion = self._dataset = self._task = self._model = self._saver = None
        self.best, self.history = None, []
        self._start = None

    @staticmethod
    def class_or_method_args():
        return Validator.class_or_method_args() + [
            ModuleFlag("eval_criterion", Criterion.REGISTRY_NAME, help="The criterion for validation."),
            ModuleFlag("eval_dataset", Dataset.REGISTRY_NAME, help="The dataset for validation."),
            Flag("eval_batch_size", dtype=Flag.TYPE.INTEGER, default=32, help="The batch size for validation process."),
            Flag("eval_task_args", dtype=Flag.TYPE.STRING, default=None, help="Other parameters for building validation dataset."),
            Flag("eval_top_checkpoints_to_keep", dtype=Flag.TYPE.INTEGER, default=0,
                 help="The number of checkpoints with the best validation metric kept under <model_dir>/best."),
        ]

    def build(self, task, model, model_dir):
        self._task, self._model = task, model
        crit_cls = self.args.get("eval_criterion.class", None) or "label_smoothed_cross_entropy"
        self._criterion = build_criterion({"criterion.class": crit_cls, "criterion.params": self.args.get("eval_criterion.params", None) or {}})
        self._start = time.time()
        if self.args.get("eval_dataset.class", None) is None:
            logging.info("WARNING: no validation dataset is provided in CriterionValidator for validation process.")
            return self
        self._dataset = build_dataset({"dataset.class": self.args["eval_dataset.class"],
                                       "dataset.params": self.args.get("eval_dataset.params", None) or {}})
        if model_dir and self._top_keep > 0:
            self._saver = KeepBestCheckpointSaver(model, os.path.join(model_dir, "best"), _AsMetric(self._criterion.as_metric()),
                                                  max_to_keep=self._top_keep)
        self._start = time.time()
        return self

    def _batches(self):
        dev = self._model.rt.device
        if getattr(self._dataset, "batched", True):
            yield from self._dataset.build_iterator(map_func=lambda b: self._task.example_to_input(b, compat.ModeKeys.EVAL), device=dev)
            return
        for b in self._task.create_and_batch(self._dataset, compat.ModeKeys.EVAL, args=self._eval_task_args):
            yield self._task.example_to_input({k: torch.from_numpy(v).to(dev) for k, v in b.items()}, compat.ModeKeys.EVAL)

    def validate(self, step):
        """criterion_validator.py:103-160: one pass over the validation set -> {"NLL", "PPL"} (criterion.reduce_metrics)."""
        if self._dataset is None:
            return None
        t0 = time.time()
        results = []
        for inputs in self._batches():
            logits = self._model(inputs, is_training=False)
            nll_sum, n_samples, n_tokens = self._criterion(inputs, logits)
            results.append((nll_sum.sum().reshape(1), n_samples, n_tokens.sum().reshape(1)))
        if not results:
            return None
        res = self._criterion.reduce_metrics([tuple(x.cpu() for x in r) for r in results])
        metric = self._criterion.as_metric()
        value = res[metric.flag]
        better = self.best is None or (value >= self.best[metric.flag] if metric.greater_is_better else value <= self.best[metric.flag])
        if better:
            self.best = dict(res)
        if self._saver is not None:     # keep-best rule of checkpoints.py:186-237: saved if fewer than K kept or >= the worst kept
            self._saver.save(step, res)
        self.history.append((step, dict(res)))
        for k, v in res.items():
            logging.info("Evaluating (%s) validation set: %s=%.2f (Best %.2f)  step=%d\tElapsed %.2fs  FromSTART %.2fs",
                         metric.flag, k, v, self.best[k], step, time.time() - t0, time.time() - self._start)
        return res
