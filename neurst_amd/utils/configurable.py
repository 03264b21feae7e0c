"""Config plumbing of neurst/utils/configurable.py used by the CLI: recursive dict merge, yaml loading and
value checking, and the model_configs.yml dump that sits beside checkpoints."""
import copy
import os

import yaml


def deep_merge_dict(dict_x, dict_y, path=None, merge_only_exist=False):
    """Recursively merges dict_y into a copy-less dict_x (dict_y wins), configurable.py semantics."""
    if dict_y is None:
        return dict_x
    if dict_x is None:
        return dict_y
    if path is None:
        path = []
    for key in dict_y:
        if key in dict_x:
            if isinstance(dict_x[key], dict) and isinstance(dict_y[key], dict):
                deep_merge_dict(dict_x[key], dict_y[key], path + [str(key)], merge_only_exist=merge_only_exist)
            elif dict_y[key] is not None or dict_x[key] is None:
                dict_x[key] = dict_y[key]
        elif not merge_only_exist:
            dict_x[key] = dict_y[key]
    return dict_x


def yaml_load_checking(args):
    """Values given as strings on the command line are yaml-parsed ("{a: 1}" -> dict, "0.1" -> float)."""
    if args is None:
        return {}
    out = {}
    for k, v in args.items():
        if isinstance(v, str):
            try:
                parsed = yaml.load(v, Loader=yaml.FullLoader)
                v = parsed if not isinstance(parsed, str) or parsed == v else v
            except Exception:
                pass
        elif isinstance(v, dict):
            v = yaml_load_checking(v)
        out[k] = v
    return out


def load_from_config_path(config_paths):
    """Loads and recursively merges a list of yaml/json config files (later files win)."""
    if config_paths is None:
        return {}
    if isinstance(config_paths, str):
        config_paths = [p for p in config_paths.split(",") if p]
    merged = {}
    for p in config_paths:
        with open(p) as fp:
            merged = deep_merge_dict(merged, yaml.load(fp, Loader=yaml.FullLoader) or {})
    return merged


class ModelConfigs(object):
    """model_configs.yml beside the checkpoints (configurable.py:277-318)."""
    MODEL_CONFIG_YAML_FILE = "model_configs.yml"

    @staticmethod
    def dump(model_config, output_dir):
        os.makedirs(output_dir, exist_ok=True)
        with open(os.path.join(output_dir, ModelConfigs.MODEL_CONFIG_YAML_FILE), "w") as fp:
            yaml.dump(copy.deepcopy(model_config), fp, default_flow_style=False)

    @staticmethod
    def load(model_dir):
        path = os.path.join(model_dir, ModelConfigs.MODEL_CONFIG_YAML_FILE)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        with open(path) as fp:
            return yaml.load(fp, Loader=yaml.FullLoader)
