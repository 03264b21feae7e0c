from neurst_amd.metrics.metric import Metric  # noqa: F401
from neurst_amd.utils.registry import setup_registry

build_metric, register_metric = setup_registry(Metric.REGISTRY_NAME, base_class=Metric, backend="pt")

from neurst_amd.metrics import bleu, wer  # noqa: E402,F401
