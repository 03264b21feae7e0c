from neurst_amd.models.model import BaseModel, build_model, register_model  # noqa: F401
from neurst_amd.models import encoder_decoder_model, speech_transformer, transformer, waitk_transformer  # noqa: F401
