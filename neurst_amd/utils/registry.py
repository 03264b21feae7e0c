"""Class registries with NeurST's semantics (neurst/utils/registry.py:24-151), TF/absl-free.

    build_x, register_x = setup_registry("model", base_class=BaseModel, create_fn="new", backend="pt")

  * register_x: bare decorator, decorator with one alias, or with a list of aliases; a class is reachable under
    its name, its lower-cased name and its snake_case name; registering a different class under a taken name
    raises ValueError("Cannot register duplicate ...").
  * build_x(args, *extra, **kwargs): args is a dict holding "<name>.class"/"class"/"<name>" and
    "<name>.params"/"params", or the class (name) itself; unknown names raise
    ValueError("Not registered class name: ..."); flags advertised by class_or_method_args() are filled with
    their defaults; with create_fn the class method of that name is the constructor.
"""
import logging
import re

from neurst_amd.utils.configurable import deep_merge_dict

REGISTRIES = {}
REGISTRIED_CLS2ALIAS = {}


def _ensure(backend, registry_name):
    REGISTRIES.setdefault(backend, {}).setdefault(registry_name, {})
    REGISTRIED_CLS2ALIAS.setdefault(backend, {}).setdefault(registry_name, {})


def _snake(name):
    return "_".join(re.sub("([A-Z])", r" \1", name).lower().strip().split())


def _resolve(table, registry_name, args):
    """(class or None, params dict) named by `args`: a dict with "<name>.class" / "class" / "<name>" and "<name>.params" /
    "params", or the class (name) itself."""
    params = {}
    if isinstance(args, dict):
        wanted = args.get("class", None) or args.get(f"{registry_name}.class", None) or args.get(registry_name, None)
        params = (args.get("params", None) or args.get(f"{registry_name}.params", {})) or {}
    else:
        wanted = args
    if wanted is None or (isinstance(wanted, str) and wanted.lower() == "none"):
        return None, params
    if isinstance(wanted, str):
        if wanted not in table:
            raise ValueError("Not registered class name: {}.".format(wanted))
        return table[wanted], params
    if not callable(wanted):
        raise ValueError("Not supported type: {} for builder.".format(type(wanted)))
    return wanted, params


def setup_registry(registry_name, base_class=None, create_fn=None, verbose_creation=False, backend="pt"):
    _ensure(backend, registry_name)
    table = REGISTRIES[backend][registry_name]

    def build_x(args, *extra_args, **kwargs):
        from neurst_amd.utils.flags_core import Flag, ModuleFlag
        target, params = _resolve(table, registry_name, args)
        if target is None:
            return None
        assert isinstance(params, dict), f"Not supported type: {type(params)} for params"
        params = dict(params)
        make = target
        if create_fn is not None:
            assert hasattr(target, create_fn), "{} has no {} for creation.".format(target, create_fn)
            make = getattr(target, create_fn)
        if verbose_creation:
            logging.info("Creating %s: %s", registry_name, target)
        if not hasattr(target, "class_or_method_args"):       # plain classes: parameters are keyword arguments
            return make(*extra_args, **deep_merge_dict(params, kwargs, merge_only_exist=False))
        for flag in target.class_or_method_args():            # flagged classes: one dict with every flag filled in
            if isinstance(flag, ModuleFlag):
                params.setdefault(flag.cls_key, flag.default)
                params.setdefault(flag.params_key, {})
            elif isinstance(flag, Flag):
                if flag.name in kwargs:
                    params[flag.name] = kwargs.pop(flag.name)
                else:
                    params.setdefault(flag.name, flag.default)
        return make(params, *extra_args, **kwargs)

    def _add(cls_, aliases):
        if base_class is not None and not issubclass(cls_, base_class):
            raise ValueError("{} must extend {}".format(cls_.__name__, base_class.__name__))
        names = set(aliases) | {cls_.__name__, cls_.__name__.lower(), _snake(cls_.__name__)}
        for n in names:
            if table.get(n, cls_) is not cls_:
                raise ValueError("Cannot register duplicate {} (under {})".format(n, registry_name))
            table[n] = cls_
        REGISTRIED_CLS2ALIAS[backend][registry_name][cls_.__name__] = names
        return cls_

    def register_x(name):
        if isinstance(name, str):
            return lambda c: _add(c, [name])
        if isinstance(name, list):
            return lambda c: _add(c, name)
        if callable(name):
            return _add(name, [])
        raise ValueError("Not supported type: {}".format(type(name)))

    return build_x, register_x


def get_registered_class(cls_, registry_name, backend="pt"):
    if cls_ is None:
        return None
    if isinstance(cls_, str):
        if cls_.lower() == "none":
            return None
        if cls_ not in REGISTRIES[backend][registry_name]:
            raise ValueError("Not registered class name: {}.".format(cls_))
        return REGISTRIES[backend][registry_name][cls_]
    return cls_ if callable(cls_) else None
