"""Trainer (neurst/exps/trainer.py:38-315), registered as entry.class=trainer.

run(): criterion / optimizer / lr schedule construction, rank-0 weight broadcast, the training loop with the
reference's logging (steps/sec, src_real_tokens_per_sec == audio frames/sec summed over workers,
training/callbacks.py:209-245), checkpoint save of model weights + model_configs.yml on rank 0.
"""
import logging
import math
import os
import time

import torch

from neurst_amd.criterions import Criterion, build_criterion
from neurst_amd.data.prefetch import Prefetcher
from neurst_amd.exps.base_experiment import BaseExperiment, register_exp
from neurst_amd.optimizers import build_lr_schedule, build_optimizer
from neurst_amd.training.distributed import GradientReducer
from neurst_amd.training import seq_generation_validator  # noqa: F401  (registers SeqGenerationValidator)
from neurst_amd.training.criterion_validator import Validator, build_validator
from neurst_amd.training.train_step import TrainStep
from neurst_amd.utils import compat
from neurst_amd.utils.checkpoints import NameBasedCheckpointManager, restore_checkpoint_if_possible
from neurst_amd.utils.configurable import ModelConfigs
from neurst_amd.utils.flags_core import Flag, ModuleFlag


@register_exp(["train", "training"])
class Trainer(BaseExperiment):
    def __init__(self, args, **kwargs):
        super().__init__(**kwargs)
        self._args = args
        self._criterion = build_criterion(args)
        self._train_steps = args["train_steps"]
        self._summary_steps = args["summary_steps"]
        self._save_checkpoint_steps = args["save_checkpoint_steps"]
        self._update_cycle = args["update_cycle"] or 1
        self._optimizer_args = {"optimizer.class": args["optimizer.class"], "optimizer.params": args["optimizer.params"]}
        self._lr_args = {"lr_schedule.class": args["lr_schedule.class"], "lr_schedule.params": args["lr_schedule.params"]}
        self._bucket_mb = args.get("allreduce_bucket_mb", None) or None   # None: the reducer's default (one message per report)
        self._clip_value, self._clip_norm = args.get("clip_value", None), args.get("clip_norm", None)
        self._validator = build_validator(args)
        self._max_to_keep = args.get("checkpoints_max_to_keep", 8) or 8
        self._pretrain_model = args.get("pretrain_model", None)
        self._ckpt_manager = None

    @staticmethod
    def class_or_method_args():
        return [
            ModuleFlag(Criterion.REGISTRY_NAME, default="label_smoothed_cross_entropy", help="The training criterion."),
            ModuleFlag("optimizer", default="Adam", help="The optimizer for training."),
            ModuleFlag("lr_schedule", default=None, help="The learning schedule for training."),
            ModuleFlag(Validator.REGISTRY_NAME, default=None, help="The validation process while training."),
            Flag("train_steps", dtype=Flag.TYPE.INTEGER, default=10000000, help="The maximum steps for training."),
            Flag("summary_steps", dtype=Flag.TYPE.INTEGER, default=200, help="Doing summary (logging) every N steps."),
            Flag("save_checkpoint_steps", dtype=Flag.TYPE.INTEGER, default=1000, help="Saving checkpoints every N steps."),
            Flag("checkpoints_max_to_keep", dtype=Flag.TYPE.INTEGER, default=8, help="Number of checkpoints to keep."),
            Flag("update_cycle", dtype=Flag.TYPE.INTEGER, default=1, help="Gradient accumulation micro steps."),
            Flag("clip_value", dtype=Flag.TYPE.FLOAT, default=None, help="Gradient clipping by value."),
            Flag("clip_norm", dtype=Flag.TYPE.FLOAT, default=None, help="Gradient clipping by norm."),
            Flag("pretrain_model", dtype=Flag.TYPE.STRING, default=None,
                 help="A checkpoint (directory or prefix, TensorFlow bundle format) to initialise the weights from."),
            Flag("allreduce_bucket_mb", dtype=Flag.TYPE.INTEGER, default=None,
                 help="Upper bound of one RCCL all-reduce message of the flat gradient buffer in MiB (default: the "
                      "reducer's own, NST_DIST_BUCKET_MB or 256 -- the configuration bench.py measures)."),
        ]

    def _save(self, step):
        if self._ckpt_manager is None:
            return
        try:
            self._ckpt_manager.save(step)
        except Exception as e:  # the reference also only warns (callbacks.py:88-92)
            logging.warning("fail to save checkpoint: %s", e)

    def run(self):
        rank, world, _ = compat.get_distributed_worker_setting()
        model, rt = self.model, self.model.rt
        lr = build_lr_schedule(self._lr_args) if self._lr_args["lr_schedule.class"] else None
        opt_params = dict(self._optimizer_args["optimizer.params"] or {})
        optimizer = build_optimizer({"optimizer.class": self._optimizer_args["optimizer.class"],
                                     "optimizer.params": opt_params})
        optimizer.bind(model.store)
        if lr is not None:
            optimizer.learning_rate = lr
        # weights: --pretrain_model first, then the latest checkpoint of model_dir (trainer.py:207-236); rank 0 reads,
        # everyone receives the broadcast
        start_step = 0
        if rank == 0:
            if self._pretrain_model:
                got = restore_checkpoint_if_possible(model, self._pretrain_model)
                logging.info("pretrain_model %s: %s", self._pretrain_model, "restored" if got else "nothing restored")
            if self.model_dir:
                self._ckpt_manager = NameBasedCheckpointManager(model, self.model_dir, self._max_to_keep, optimizer=optimizer)
                if self._ckpt_manager.restore() is not None:
                    start_step = optimizer.iterations
                    logging.info("resuming %s at step %d", self.model_dir, start_step)
        reducer = GradientReducer(model.store, bucket_bytes=(self._bucket_mb << 20) if self._bucket_mb else None)
        reducer.broadcast_parameters(0)
        if world > 1:
            start_step = int(reducer.reduce_metrics({"start_step": float(start_step)})["start_step"])
            if start_step:
                optimizer.iterations = start_step
                reducer.broadcast_tensors([optimizer.m, optimizer.v], 0)
        step_fn = TrainStep(model, self._criterion, optimizer, reducer, self._update_cycle, clip_value=self._clip_value,
                            clip_norm=self._clip_norm)
        if rank == 0 and self.model_dir:
            ModelConfigs.dump({"model.class": model.__class__.__name__, "model.params": model.args,
                               "task.class": self.task.__class__.__name__, "task.params": self.task.get_config()},
                              self.model_dir)
        if getattr(self.custom_dataset, "batched", True):
            it = self.custom_dataset.build_iterator(
                map_func=lambda b: self.task.example_to_input(b, compat.ModeKeys.TRAIN), shard_id=rank,
                total_shards=world, device=rt.device)
        else:  # single examples (TFRecords): the task buckets and pads them (tasks/*.create_and_batch).  One process per
            # GPU is the reference's Horovod arrangement: every worker batches its own file shard with
            # num_replicas_in_sync = 1 (training_utils.py:146-151), i.e. `batch_size` is per worker.
            def _feed():
                host_batches = Prefetcher(self.task.create_and_batch(
                    self.custom_dataset, compat.ModeKeys.TRAIN, num_replicas_in_sync=1, shard_id=rank, total_shards=world,
                    seed=self._args.get("seed", None) or 1234), depth=4)   # parsing / bucketing / padding run ahead in a thread
                for b in host_batches:
                    yield self.task.example_to_input({k: torch.from_numpy(v).to(rt.device, non_blocking=True)
                                                      for k, v in b.items()}, compat.ModeKeys.TRAIN)

            it = _feed()
        if self._validator is not None and rank == 0:
            self._validator.build(self.task, model, self.model_dir)
            if self._validator._eval_on_begin:
                self._validator.validate(start_step)
        # resume: the dropout masks are keyed by (seed, rt.step, site) -- continue the step count so a restarted run does not
        # replay the masks of steps 0.. (the data iterator restarts with its epoch, like the reference's tf.data pipeline)
        rt.step = start_step
        t0, last_loss = time.time(), None
        frames_acc, steps_acc = None, 0   # summed ON THE DEVICE between summaries: no host sync per step (callbacks.py:209-245)
        check_numerics = bool(getattr(self, "enable_check_numerics", False) or self._args.get("enable_check_numerics", False))
        for step in range(start_step + 1, self._train_steps + 1):
            try:
                batches = [next(it) for _ in range(self._update_cycle)]
            except StopIteration:
                break
            frames_dev = sum(b["src_length"].sum() for b in batches)
            frames_acc = frames_dev if frames_acc is None else frames_acc + frames_dev
            steps_acc += 1
            last_loss = step_fn(batches)
            if step % self._summary_steps == 0 or step == self._train_steps:
                if rt.device.type == "cuda":
                    torch.cuda.synchronize()
                dt = time.time() - t0
                m = reducer.reduce_metrics({"loss": float(last_loss) / world, "src_real_tokens": float(frames_acc)})
                if rank == 0:
                    logging.info("step %d: loss=%.4f  %.3f steps/sec  %.1f src_real_tokens_per_sec  lr=%.3e", step,
                                 m["loss"], steps_acc / dt, m["src_real_tokens"] / dt, optimizer.current_lr())
                self.last_summary = {"step": step, "loss": m["loss"], "steps_per_sec": steps_acc / dt,
                                     "src_real_tokens_per_sec": m["src_real_tokens"] / dt}
                if check_numerics and not math.isfinite(m["loss"]):   # the reduced loss: every rank raises at the same step
                    raise FloatingPointError(f"--enable_check_numerics: loss is {m['loss']} at step {step}")
                t0, frames_acc, steps_acc = time.time(), None, 0
            if rank == 0 and self._save_checkpoint_steps and step % self._save_checkpoint_steps == 0:
                self._save(step)
            if self._validator is not None and self._validator.due(step):
                if rank == 0:
                    self._validator.validate(step)
                # early stopping (eval_estop_patience) is decided on rank 0 and agreed on by every rank at the same step
                stop = reducer.reduce_metrics({"stop": 1.0 if (rank == 0 and getattr(self._validator, "should_stop", False)) else 0.0})
                if stop["stop"] > 0:
                    logging.info("early stop at step %d", step)
                    break
        reducer.close()      # (explicit: the library's communicator is not left to the garbage collector)
        return last_loss
