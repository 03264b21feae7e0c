"""Decoder base class + registry (neurst/layers/decoders/decoder.py, __init__.py)."""
from neurst_amd.utils.registry import setup_registry


class Decoder(object):
    REGISTRY_NAME = "decoder"

    def __init__(self, **kwargs):
        self._params = kwargs

    def get_config(self):
        return dict(self._params)


build_decoder, register_decoder = setup_registry(Decoder.REGISTRY_NAME, base_class=Decoder, backend="pt")
