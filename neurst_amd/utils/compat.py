"""Global settings of neurst/utils/compat.py that the training path consults (TF-free)."""
FLOAT_MIN = -1.0e9          # compat.py:24
CUSTOM_GLOBAL_FLOATX = "float32"


class ModeKeys(object):
    TRAIN = "train"
    EVAL = "eval"
    INFER = "infer"


class DataStatus(object):
    RAW = "raw"
    PROCESSED = "processed"
    PROJECTED = "projected"


class PaddingMode(object):
    DEFAULT = 0
    EOS_AS_PADDING = 1


_GLOBAL = {"worker_id": 0, "num_workers": 1, "strategy": None, "initial_step": 0}


def register_distributed_worker_setting(worker_id, num_workers, strategy):
    """compat.py:93-99."""
    _GLOBAL.update(worker_id=worker_id, num_workers=num_workers, strategy=strategy)


def get_distributed_worker_setting():
    """compat.py:102-105 -> (worker_id, num_workers, strategy)."""
    return _GLOBAL["worker_id"], _GLOBAL["num_workers"], _GLOBAL["strategy"]


def register_initial_step(step):
    _GLOBAL["initial_step"] = int(step)


def get_registered_initial_step():
    return _GLOBAL["initial_step"]


def register_computation_dtype(floatx):
    global CUSTOM_GLOBAL_FLOATX
    CUSTOM_GLOBAL_FLOATX = floatx
