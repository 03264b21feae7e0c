"""Named hyper-parameter sets (neurst/utils/hparams_sets.py:19-57)."""
from neurst_amd.utils.registry import REGISTRIES


def register_hparams_set(name, backend="pt"):
    REGISTRIES.setdefault(backend, {}).setdefault("hparams_set", {})
    table = REGISTRIES[backend]["hparams_set"]

    def register_x_fn(fn_, short_name=None):
        names = set(n.lower() for n in (short_name or []))
        names.add(fn_.__name__)
        for n in names:
            if n in table and table[n] != fn_:
                raise ValueError("Cannot register duplicate {} (under hparams_set)".format(n))
            table[n] = fn_
        return fn_

    if isinstance(name, str):
        return lambda fn: register_x_fn(fn, [name])
    if isinstance(name, list):
        return lambda fn: register_x_fn(fn, name)
    raise ValueError("Not supported type: {}".format(type(name)))


def get_hyper_parameters(name, backend="pt"):
    """hparams_sets.py:45-57: a registered set, else the first model whose build_model_args_by_name knows it."""
    if name is None:
        return {}
    table = REGISTRIES.get(backend, {}).get("hparams_set", {})
    if name in table:
        return table[name]()
    for mc in set(REGISTRIES.get(backend, {}).get("model", {}).values()):
        if hasattr(mc, "build_model_args_by_name"):
            p = mc.build_model_args_by_name(name)
            if p is not None:
                return p
    return {}
