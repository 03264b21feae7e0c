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

FLAG_LIST = [
    flags_core.Flag("distribution_strategy", dtype=flags_core.Flag.TYPE.STRING, default="rccl",
                    help="The distribution strategy: rccl (one process per GPU; horovod/byteps/mirrored are accepted "
                         "as aliases) or none."),
    flags_core.Flag("dtype", dtype=flags_core.Flag.TYPE.STRING, default="bfloat16",
                    help="The computation type of the whole model: float32 or bfloat16."),
    flags_core.Flag("enable_check_numerics", dtype=flags_core.Flag.TYPE.BOOLEAN, default=None,
                    help="Check the loss for NaN/Inf at every summary step."),
    flags_core.Flag("enable_xla", dtype=flags_core.Flag.TYPE.BOOLEAN, default=None, help="Ignored (no XLA here)."),
    flags_core.Flag("hparams_set", dtype=flags_core.Flag.TYPE.STRING,
                    help="A pre-defined hyper-parameter set, e.g. speech_transformer_s."),
    flags_core.Flag("model_dir", dtype=flags_core.Flag.TYPE.STRING, help="The path for saving and loading checkpoints."),
    flags_core.Flag("seed", dtype=flags_core.Flag.TYPE.INTEGER, default=1234, help="Dropout / data seed."),
    flags_core.ModuleFlag(BaseExperiment.REGISTRY_NAME, help="The program."),
    flags_core.ModuleFlag(Task.REGISTRY_NAME, help="The binding task."),
    flags_core.ModuleFlag(BaseModel.REGISTRY_NAME, help="The model."),
    flags_core.ModuleFlag(Dataset.REGISTRY_NAME, help="The dataset."),
]


def _pre_load_args(args):
    """run_exp.py:53-76: model_dir/model_configs.yml < hparams_set < --config_paths."""
    paths = flags_core._flatten_string_list(getattr(args, "config_paths", None))
    cfg_file_args = yaml_load_checking(load_from_config_path(paths))
    model_dir = args.model_dir or cfg_file_args.get("model_dir", None)
    hparams_set = args.hparams_set or cfg_file_args.get("hparams_set", None)
    predefined = dict(get_hyper_parameters(hparams_set))
    formatted = {}
    for k in ("model.class", "model", "model.params"):
        if k in predefined:
            formatted[k] = predefined.pop(k)
    if predefined:
        formatted["entry.params"] = predefined
    try:
        return deep_merge_dict(deep_merge_dict(ModelConfigs.load(model_dir), formatted), cfg_file_args)
    except Exception:
        return deep_merge_dict(formatted, cfg_file_args)


def run_experiment(args, remaining_argv):
    rank, local_rank, world = init_distributed()
    flags_core.verbose_flags(FLAG_LIST, args, remaining_argv)
    task = build_task(args)
    custom_dataset = build_dataset(args)
    model = task.build_model(args, device=f"cuda:{local_rank}", dtype=args["dtype"], seed=args["seed"] + rank)
    entry = build_exp(args, strategy=args["distribution_strategy"], model=model, task=task,
                      model_dir=args["model_dir"], custom_dataset=custom_dataset)
    return entry.run()


def _main(argv=None):
    arg_parser = flags_core.define_flags(FLAG_LIST, argv=argv)
    args, remaining_argv = flags_core.intelligent_parse_flags(FLAG_LIST, arg_parser, _pre_load_args, argv=argv)
    if args["entry.class"] is None:
        raise ValueError("Must provide entry/entry.class.")
    return run_experiment(args, remaining_argv)


def cli_main():
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    _main(sys.argv[1:])


if __name__ == "__main__":
    cli_main()
