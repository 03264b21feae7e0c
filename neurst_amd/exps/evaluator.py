"""Evaluator, entry "evaluation" / "eval" (neurst/exps/evaluator.py): restores the latest checkpoint of model_dir and
reports the criterion's metrics (NLL, PPL) over the dataset -- the one-shot form of the CriterionValidator."""
import logging

from neurst_amd.criterions import Criterion
from neurst_amd.exps.base_experiment import BaseExperiment, register_exp
from neurst_amd.training.criterion_validator import CriterionValidator
from neurst_amd.utils.checkpoints import restore_checkpoint_if_possible
from neurst_amd.utils.flags_core import Flag, ModuleFlag


@register_exp(["evaluation", "eval"])
class Evaluator(BaseExperiment):
    def __init__(self, args, **kwargs):
        super().__init__(**kwargs)
        self._criterion_args = {"eval_criterion.class": args.get("criterion.class", None) or "label_smoothed_cross_entropy",
                                "eval_criterion.params": args.get("criterion.params", None) or {},
                                "eval_batch_size": args.get("batch_size", None) or 32}

    @staticmethod
    def class_or_method_args():
        return [ModuleFlag(Criterion.REGISTRY_NAME, default="label_smoothed_cross_entropy", help="The evaluation criterion."),
                Flag("batch_size", dtype=Flag.TYPE.INTEGER, default=32, help="Utterances / sentences per evaluation batch.")]

    def run(self):
        if self.model_dir:
            got = restore_checkpoint_if_possible(self.model, self.model_dir)
            logging.info("checkpoint: %s", got or "none restored (random weights)")
        v = CriterionValidator(self._criterion_args)
        v.build(self.task, self.model, None)
        v._dataset = self.custom_dataset
        res = v.validate(0)
        for k, val in (res or {}).items():
            logging.info("Evaluation Result: %s=%.4f", k, val)
        return res
