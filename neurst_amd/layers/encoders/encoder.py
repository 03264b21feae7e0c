"""Encoder base class + registry (neurst/layers/encoders/encoder.py, __init__.py)."""
from neurst_amd.utils.registry import setup_registry


class Encoder(object):
    REGISTRY_NAME = "encoder"

    def __init__(self, **kwargs):
        self._params = kwargs

    def get_config(self):
        return dict(self._params)


build_encoder, register_encoder = setup_registry(Encoder.REGISTRY_NAME, base_class=Encoder, backend="pt")
