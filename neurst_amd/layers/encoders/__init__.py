from neurst_amd.layers.encoders.encoder import Encoder, build_encoder, register_encoder  # noqa: F401
from neurst_amd.layers.encoders import transformer_encoder  # noqa: F401  (registers TransformerEncoder)
