"""Flag / ModuleFlag definitions and hierarchical argument parsing with NeurST's conventions
(neurst/utils/flags_core.py:32-489), TF/absl-free.

Conventions kept from the reference:
  * every registered class advertises its flags through ``class_or_method_args()``;
  * a ``ModuleFlag("model")`` expands to ``--model.class`` (alias ``--model``) and ``--model.params`` (a
    yaml/json string); the selected class's own flags may also be given flat on the command line or flat in a
    config file and are folded into ``model.params``;
  * precedence: command line > ``--config_paths`` yaml(s) > ``--hparams_set`` > ``model_dir/model_configs.yml``
    (the last three are merged by the caller-supplied ``args_preload_func``, neurst/cli/run_exp.py:53-76);
  * values are yaml-parsed (``"{a: 1}"`` -> dict).
"""
import argparse
import copy
import importlib
import logging
import os
from collections import namedtuple

from neurst_amd.utils.configurable import deep_merge_dict, load_from_config_path, yaml_load_checking
from neurst_amd.utils.registry import REGISTRIES

_DEFINED_FLAGS = dict()
BACKEND = "pt"


class Flag(object):
    TYPE = namedtuple("FLAG_ARG_TYPES", "INTEGER BOOLEAN FLOAT STRING")(int, bool, float, str)

    def __init__(self, name, dtype, required=False, choices=None, help="", default=None, multiple=False, alias=None):
        if name in ["class", "params"]:
            raise ValueError("Invalid flag name: {}".format(name))
        if "-" in name:
            raise ValueError("Flag name with '-' is not supported.")
        self._name, self._dtype, self._default, self._help = name.strip(), dtype, default, help
        self._choices, self._multiple, self._required, self._alias = choices, multiple, required, alias

    name = property(lambda self: self._name)
    dtype = property(lambda self: self._dtype)
    default = property(lambda self: self._default)
    multiple = property(lambda self: self._multiple)
    help = property(lambda self: self._help)
    alias = property(lambda self: self._alias)
    choices = property(lambda self: self._choices)

    def define(self, arg_parser, default_is_none=True):
        names = ["--" + self.name] + (["--" + self.alias] if self.alias else [])
        kwargs = {"dest": self.name, "help": self.help}
        if self.dtype is bool:
            kwargs.update(action="store_true", default=None)
        else:
            kwargs["type"] = str if self.dtype is str else self.dtype
        if self.multiple:
            kwargs["nargs"] = "+"
        if self.choices:
            kwargs["choices"] = self.choices
        if self.default and not default_is_none and self.dtype is not bool:
            kwargs["default"] = self.default
        if self._required:
            kwargs["required"] = True
        try:
            arg_parser.add_argument(*names, **kwargs)
        except argparse.ArgumentError:
            raise ValueError(f"Defined duplicate arg key: {self.name}")
        _DEFINED_FLAGS[self.name] = self
        return arg_parser


class ModuleFlag(object):
    def __init__(self, name, module_name=None, default=None, help=""):
        self._name, self._module_name, self._help, self._default = name, module_name or name, help, default

    name = property(lambda self: self._name)
    module_name = property(lambda self: self._module_name)
    help = property(lambda self: self._help)
    default = property(lambda self: self._default)
    cls_key = property(lambda self: self._name + ".class")
    params_key = property(lambda self: self._name + ".params")

    def define(self, arg_parser):
        _DEFINED_FLAGS[self.name] = self
        Flag(self.cls_key, dtype=Flag.TYPE.STRING, alias=self.name, default=self.default,
             help=f"The class name of {self.module_name} for '{self.help}'").define(arg_parser)
        Flag(self.params_key, dtype=Flag.TYPE.STRING,
             help=f"The json/yaml-like parameter string for {self.module_name}").define(arg_parser)
        return arg_parser


DEFAULT_CONFIG_FLAG = Flag("config_paths", dtype=Flag.TYPE.STRING, multiple=True,
                           help="Path to json/yaml configuration files defining FLAG values; merged recursively.")
EXTRA_IMPORT_LIB = Flag("include", dtype=Flag.TYPE.STRING, multiple=True,
                        help="The extra python path (module, or directory of plugin files) to be imported.")


def add_extra_includes(argv=None):
    """--include <module | dir>: imports user plugins so their @register_* decorators run (flags_core.py:207-247).
    Directory plugins are imported in place (the reference copies them into neurst/utils/userdef first)."""
    parser = argparse.ArgumentParser(add_help=False)
    parser.add_argument("--include", nargs="+", default=None)
    parsed, _ = parser.parse_known_args(argv)
    for path in parsed.include or []:
        if not os.path.isdir(path):
            try:
                importlib.import_module(path)
            except ImportError as e:
                logging.warning("fail to import %s: %s", path, e)
            continue
        for file in sorted(os.listdir(path)):
            if file.startswith(("_", ".")) or not file.endswith(".py"):
                continue
            src = os.path.join(path, file)
            with open(src) as fp:
                if not any(line.strip().startswith("@register") for line in fp):
                    continue
            spec = importlib.util.spec_from_file_location("neurst_amd_userdef_" + file[:-3], src)
            mod = importlib.util.module_from_spec(spec)
            try:
                spec.loader.exec_module(mod)
            except Exception as e:  # noqa
                logging.warning("fail to import %s: %s", src, e)


def define_flags(flag_list, arg_parser=None, with_config_file=True, argv=None):
    add_extra_includes(argv)
    if arg_parser is None:
        arg_parser = argparse.ArgumentParser()
    if with_config_file:
        DEFAULT_CONFIG_FLAG.define(arg_parser)
    EXTRA_IMPORT_LIB.define(arg_parser)
    for f in flag_list:
        f.define(arg_parser)
    return arg_parser


def _registered(module_name, cls_name):
    table = REGISTRIES.get(BACKEND, {}).get(module_name, {})
    if cls_name not in table:
        raise ValueError("Not registered class name: {}.".format(cls_name))
    return table[cls_name]


def _class_flags(module_name, cls_name):
    cls_ = _registered(module_name, cls_name)
    return cls_.class_or_method_args() if hasattr(cls_, "class_or_method_args") else []


def _flatten_string_list(x):
    if x is None:
        return None
    if isinstance(x, str):
        x = [x]
    out = []
    for e in x:
        out.extend([p for p in str(e).split(",") if p])
    return out


def _args_preload_from_config_files(args):
    return yaml_load_checking(load_from_config_path(_flatten_string_list(getattr(args, "config_paths", None))))


def _resolve_module(f, cli_args, cfg_args, remaining_argv, depth=0):
    """Resolves one ModuleFlag: returns (cls_name, params dict, remaining argv).  cfg_args is consumed
    (flat keys that belong to the class are moved into its params)."""
    cls_name = cli_args.get(f.cls_key) or cfg_args.get(f.cls_key) or cfg_args.get(f.name) or f.default
    cfg_args.pop(f.cls_key, None)
    cfg_args.pop(f.name, None)
    cli_params = cli_args.get(f.params_key) or {}
    if isinstance(cli_params, str):
        cli_params = yaml_load_checking({"x": cli_params})["x"]
    cfg_params = cfg_args.pop(f.params_key, None) or {}
    params = deep_merge_dict(copy.deepcopy(cfg_params), cli_params)
    if cls_name is None or (isinstance(cls_name, str) and cls_name.lower() == "none"):
        return None, params, remaining_argv
    flags = _class_flags(f.module_name, cls_name)
    if flags:
        sub = argparse.ArgumentParser(add_help=False)
        for ff in flags:
            try:
                ff.define(sub)
            except ValueError:
                pass
        sub_parsed, remaining_argv = sub.parse_known_args(remaining_argv)
        sub_parsed = yaml_load_checking({k: v for k, v in sub_parsed.__dict__.items() if v is not None})
        for ff in flags:
            if isinstance(ff, ModuleFlag):
                inner_cli = {k: sub_parsed[k] for k in (ff.cls_key, ff.params_key) if k in sub_parsed}
                inner_cfg = {k: params.pop(k) for k in (ff.cls_key, ff.name, ff.params_key) if k in params}
                for k in (ff.cls_key, ff.name, ff.params_key):
                    if k in cfg_args and k not in inner_cfg:
                        inner_cfg[k] = cfg_args.pop(k)
                icls, iparams, remaining_argv = _resolve_module(ff, inner_cli, inner_cfg, remaining_argv, depth + 1)
                params[ff.cls_key] = icls
                params[ff.params_key] = iparams
            else:
                if ff.name in sub_parsed:
                    params[ff.name] = sub_parsed[ff.name]
                elif ff.name in params:
                    pass
                elif ff.name in cfg_args:
                    params[ff.name] = cfg_args.pop(ff.name)
                else:
                    params[ff.name] = ff.default
    return cls_name, params, remaining_argv


def intelligent_parse_flags(flag_list, arg_parser, args_preload_func=_args_preload_from_config_files, argv=None):
    """Parses the program flags and folds config files / hparams sets in (flags_core.py:367-440)."""
    parsed, remaining_argv = arg_parser.parse_known_args(argv)
    cfg_args = args_preload_func(parsed) if args_preload_func is not None else {}
    cfg_args = copy.deepcopy(cfg_args or {})
    cli_args = yaml_load_checking({k: v for k, v in parsed.__dict__.items() if v is not None})
    out = {}
    for f in flag_list:
        if isinstance(f, Flag):
            if f.name in cli_args:
                out[f.name] = cli_args[f.name]
            elif f.name in cfg_args:
                out[f.name] = cfg_args.pop(f.name)
            else:
                out[f.name] = f.default
            cfg_args.pop(f.name, None)
    for f in flag_list:
        if isinstance(f, ModuleFlag):
            cls_name, params, remaining_argv = _resolve_module(f, cli_args, cfg_args, remaining_argv)
            out[f.cls_key] = cls_name
            out[f.params_key] = params
    # whatever is left in the config files is kept (the reference keeps unknown keys too)
    out = deep_merge_dict(cfg_args, out)
    return out, remaining_argv


def verbose_flags(flag_list, args, remaining_argv=None, log=logging.info):
    log("==========================================================================")
    log("Parsed all matched flags: ")
    for f in flag_list:
        if isinstance(f, Flag):
            log(f"  {f.name}: {args.get(f.name)}     # {f.help}")
        else:
            log(f"  {f.cls_key}: {args.get(f.cls_key)}")
            for k, v in (args.get(f.params_key) or {}).items():
                log(f"    {k}: {v if not (isinstance(v, list) and len(v) > 10) else v[:10] + ['...']}")
    if remaining_argv:
        log(f"  unparsed: {remaining_argv}")
    log("==========================================================================")
