"""SeqGenerationValidator (neurst/training/seq_generation_validator.py:30-290 with TrainingStatusRecorder,
training_utils.py:274-360): on top of the criterion validation, every `eval_steps` the validation set is DECODED with the
configured search method, scored with the configured metric (BLEU / WER ...), and
  * the checkpoints with the best scores are kept under `eval_best_checkpoint_path` (default `<model_dir>_best`, at most
    `eval_top_checkpoints_to_keep`),
  * their average is refreshed under `eval_best_avg_checkpoint_path` (default `<best path>_avg`) when
    `eval_auto_average_checkpoints` is on,
  * `should_stop` turns true after `eval_estop_patience` validations without a better score (the trainer stops).
"""
import logging
import os
import time

import torch

from neurst_amd.layers.search import SequenceSearch, build_search_layer
from neurst_amd.metrics import Metric, build_metric
from neurst_amd.training.criterion_validator import CriterionValidator, register_validator
from neurst_amd.utils import compat
from neurst_amd.utils.checkpoints import AverageCheckpointSaver, KeepBestCheckpointSaver
from neurst_amd.utils.flags_core import Flag, ModuleFlag


@register_validator(["seq_generation", "SeqGenerationValidator"])
class SeqGenerationValidator(CriterionValidator):
    def __init__(self, args):
        super().__init__(args)
        self._gen_metric = self._search = self._gen_saver = self._references = None
        self._patience = args.get("eval_estop_patience", None) or 0
        self._gen_keep = args.get("eval_top_checkpoints_to_keep", None)
        self._gen_keep = 10 if self._gen_keep is None else self._gen_keep
        self._top_keep = 0                      # the generation metric, not the NLL, selects the checkpoints to keep
        self.gen_best, self.gen_history, self._bad_count, self.should_stop = None, [], 0, False
        self._best_path = self._avg_path = None

    @staticmethod
    def class_or_method_args():
        return [f for f in CriterionValidator.class_or_method_args() if f.name != "eval_top_checkpoints_to_keep"] + [
            ModuleFlag("eval_metric", Metric.REGISTRY_NAME, help="The metric for evaluating generation results."),
            ModuleFlag("eval_search_method", SequenceSearch.REGISTRY_NAME, help="The search layer for sequence generation."),
            Flag("eval_estop_patience", dtype=Flag.TYPE.INTEGER, default=0,
                 help="Stop training after this many validations without a better score (0 = never)."),
            Flag("eval_best_checkpoint_path", dtype=Flag.TYPE.STRING, default=None,
                 help="The path for checkpoints with best metric scores (default `model_dir`_best)."),
            Flag("eval_auto_average_checkpoints", dtype=Flag.TYPE.BOOLEAN, default=True,
                 help="Whether to average the kept best checkpoints into an extra directory."),
            Flag("eval_best_avg_checkpoint_path", dtype=Flag.TYPE.STRING, default=None,
                 help="The path to saving the averaged checkpoints (default `eval_best_checkpoint_path`_avg)."),
            Flag("eval_top_checkpoints_to_keep", dtype=Flag.TYPE.INTEGER, default=10,
                 help="The maximum number of best checkpoints kept (and averaged)."),
        ]

    def build(self, task, model, model_dir):
        super().build(task, model, model_dir)
        if self._dataset is None:
            return self
        if self.args.get("eval_metric.class", None) is None:
            logging.info("WARNING: no metric is provided in SeqGenerationValidator for validation process.")
            return self
        params = dict(self.args.get("eval_metric.params", None) or {})
        params.setdefault("language", (getattr(task, "trg_meta", None) or {}).get("language", "en"))
        self._gen_metric = build_metric({"metric.class": self.args["eval_metric.class"], "metric.params": params})
        self._gen_metric.flag = self.args["eval_metric.class"]
        self._search = build_search_layer({"search_method.class": self.args.get("eval_search_method.class", None) or "beam_search",
                                           "search_method.params": self.args.get("eval_search_method.params", None) or {}})
        refs = getattr(self._dataset, "raw_targets", None) or getattr(self._dataset, "targets", None)
        if refs is None:
            logging.info("WARNING: no ground truth found for validation dataset and no validation will be applied.")
            self._gen_metric = None
            return self
        self._references = [r if isinstance(r, str) else self._ids_to_text(r) for r in refs]
        self._gen_metric.set_groundtruth(self._references)
        if model_dir and self._gen_keep > 0:
            self._best_path = self.args.get("eval_best_checkpoint_path", None) or (model_dir.rstrip("/") + "_best")
            self._gen_saver = KeepBestCheckpointSaver(model, self._best_path, self._gen_metric, max_to_keep=self._gen_keep)
            if self.args.get("eval_auto_average_checkpoints", True):
                self._avg_path = self.args.get("eval_best_avg_checkpoint_path", None) or (self._best_path.rstrip("/") + "_avg")
                self._avg_saver = AverageCheckpointSaver(model, self._avg_path, self._gen_metric, max_to_keep=self._gen_keep)
        return self

    def _ids_to_text(self, ids):
        dp = getattr(self._task, "_trg_data_pipeline", None)
        ids = [int(x) for x in ids]
        if dp is not None:
            return dp.decode(ids)
        eos = self._task.trg_meta["eos_id"]
        return " ".join(str(x) for x in (ids[:ids.index(eos)] if eos in ids else ids))

    def generate(self):
        """The hypotheses of the whole validation set, in dataset order (SequenceGenerator's loop)."""
        dev, hyps = self._model.rt.device, []
        if getattr(self._dataset, "batched", True):
            batches = self._dataset.build_iterator(shard_id=0, total_shards=1, device=dev)
        else:
            batches = ({k: torch.from_numpy(v).to(dev) for k, v in b.items()}
                       for b in self._task.create_and_batch(self._dataset, compat.ModeKeys.INFER, args=self._eval_task_args))
        for batch in batches:
            inputs = self._task.example_to_input(batch, compat.ModeKeys.INFER)
            hyp, _ = self._search(self._model, inputs)
            k = self._search.top_k
            hyps.extend(self._ids_to_text(row) for row in hyp.view(-1, k, hyp.shape[-1])[:, 0].cpu().tolist())
        return hyps

    def validate(self, step):
        res = super().validate(step)
        if self._gen_metric is None:
            return res
        t0 = time.time()
        hyps = self.generate()
        if len(hyps) != len(self._references):
            raise RuntimeError(f"{len(hyps)} hypotheses for {len(self._references)} references")
        score = self._gen_metric(hyps)
        better = self.gen_best is None or self._gen_metric.greater_or_eq(score, self.gen_best)
        # TrainingStatusRecorder.record (training_utils.py:335-372): both savers see EVERY validation and apply their own
        # keep-best rule (checkpoints.py:186-312); the best result only drives the patience counter
        if self._gen_saver is not None:
            self._gen_saver.save(step, score)
        if getattr(self, "_avg_saver", None) is not None:
            self._avg_saver.save(step, score)
        if better:
            self.gen_best, self._bad_count = dict(score), 0
        else:
            self._bad_count += 1
            if self._patience > 0 and self._bad_count >= self._patience:
                logging.info("No better %s for %d validations: stop training.", self._gen_metric.flag, self._bad_count)
                self.should_stop = True
        self.gen_history.append((step, dict(score)))
        flag = self._gen_metric.flag
        logging.info("Evaluating (%s) validation set: %s=%.2f (Best %.2f)  step=%d\tElapsed %.2fs", flag, flag,
                     self._gen_metric.get_value(score), self._gen_metric.get_value(self.gen_best), step, time.time() - t0)
        for i in range(min(3, len(hyps))):
            logging.info("  Reference: %s\n  Hypothesis: %s", self._references[i], hyps[i])
        return dict(res or {}, **score)
