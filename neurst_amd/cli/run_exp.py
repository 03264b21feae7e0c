"""`neurst-run` for the MI355X path (neurst/cli/run_exp.py:27-123): same flag list and flow --
build_task -> build_dataset -> task.build_model -> build_exp -> entry.run().

  python -m neurst_amd.cli.run_exp --entry trainer --task speech2text --hparams_set speech_transformer_s \
      --dataset synthetic_speech --dtype bfloat16 --distribution_strategy rccl --train_steps 100
Multi GPU: one process per GPU,  python -m torch.distributed.run --nproc-per-node 8 -m neurst_amd.cli.run_exp ...
"""
import logging
import sys

import neurst_amd.utils.flags_core as flags_core
from neurst_amd.data.datasets import Dataset, build_dataset
from neurst_amd.exps import BaseExperiment, build_exp
from neurst_amd.models import BaseModel
from neurst_amd.tasks import Task, build_task
from neurst_amd.training.distributed import init_distributed
from neurst_amd.utils.configurable import ModelConfigs, deep_merge_dict, load_from_config_path, yaml_load_checking
from neurst_amd.utils.hparams_sets import get_hyper_parameters

_F, _M = flags_core.Flag, flags_core.ModuleFlag
FLAG_LIST = [
    _F("distribution_strategy", dtype=_F.TYPE.STRING, default="rccl",
       help="rccl = one process per GPU (horovod / byteps / mirrored are accepted as aliases), or none."),
    _F("dtype", dtype=_F.TYPE.STRING, default="bfloat16",
       help="Compute dtype of activations and GEMM weights: bfloat16 (default) or float32."),
    _F("enable_check_numerics", dtype=_F.TYPE.BOOLEAN, default=None, help="Stop when the logged loss is NaN or Inf."),
    _F("enable_xla", dtype=_F.TYPE.BOOLEAN, default=None, help="Ignored (no XLA here)."),
    _F("hparams_set", dtype=_F.TYPE.STRING,
       help="Name of a registered hyper-parameter set (speech_transformer_s, transformer_big, ...)."),
    _F("model_dir", dtype=_F.TYPE.STRING,
       help="Directory of checkpoints and model_configs.yml (written by training, read by everything else)."),
    _F("seed", dtype=_F.TYPE.INTEGER, default=1234, help="Base seed of dropout masks and synthetic data (+ rank)."),
    _M(BaseExperiment.REGISTRY_NAME, help="What to run: trainer, evaluation, predict."),
    _M(Task.REGISTRY_NAME, help="Task class (speech2text, translation, waitk_translation)."),
    _M(BaseModel.REGISTRY_NAME, help="Model class (normally given by --hparams_set)."),
    _M(Dataset.REGISTRY_NAME, help="Dataset class."),
]


def _pre_load_args(args):
    """Defaults below the command line, weakest first (run_exp.py:53-76): model_dir/model_configs.yml, then the named
    hyper-parameter set (its model.* keys at top level, everything else as entry.params), then --config_paths files."""
    from_files = yaml_load_checking(load_from_config_path(flags_core._flatten_string_list(getattr(args, "config_paths", None))))
    named = dict(get_hyper_parameters(args.hparams_set or from_files.get("hparams_set", None)))
    layered = {k: named.pop(k) for k in ("model.class", "model", "model.params") if k in named}
    if named:
        layered["entry.params"] = named
    try:
        stored = ModelConfigs.load(args.model_dir or from_files.get("model_dir", None))
    except Exception:
        return deep_merge_dict(layered, from_files)
    return deep_merge_dict(deep_merge_dict(stored, layered), from_files)


def _rank_device(local_rank):
    """The device init_distributed() pinned for this rank (local_rank modulo the visible devices: a rehearsal of more ranks
    than GPUs shares devices); CPU hosts keep the literal name so that the error is the usual "no ROCm device"."""
    import torch
    return f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else f"cuda:{local_rank}"


def run_experiment(args, remaining_argv, device=None):
    """`device` is for the host-logic tests only (they install emulated kernels and pass "cpu"); the product always runs
    on this rank's GPU -- neurst_amd.kernels refuses anything else."""
    rank, local_rank, world = init_distributed()
    flags_core.verbose_flags(FLAG_LIST, args, remaining_argv)
    task = build_task(args)
    custom_dataset = build_dataset(args)
    model = task.build_model(args, device=device or _rank_device(local_rank), dtype=args["dtype"], seed=args["seed"] + rank)
    entry = build_exp(args, strategy=args["distribution_strategy"], model=model, task=task,
                      model_dir=args["model_dir"], custom_dataset=custom_dataset)
    if args.get("enable_check_numerics", None):
        entry.enable_check_numerics = True   # the trainer raises on a non-finite logged loss (all ranks, same step)
    return entry.run()


def _main(argv=None, device=None):
    arg_parser = flags_core.define_flags(FLAG_LIST, argv=argv)
    args, remaining_argv = flags_core.intelligent_parse_flags(FLAG_LIST, arg_parser, _pre_load_args, argv=argv)
    if args["entry.class"] is None:
        raise ValueError("Must provide entry/entry.class.")
    return run_experiment(args, remaining_argv, device=device)


def cli_main():
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    _main(sys.argv[1:])


if __name__ == "__main__":
    cli_main()
