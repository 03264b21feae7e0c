from neurst_amd.layers.decoders.decoder import Decoder, build_decoder, register_decoder  # noqa: F401
from neurst_amd.layers.decoders import transformer_decoder  # noqa: F401  (registers TransformerDecoder)
