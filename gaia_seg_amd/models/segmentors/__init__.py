from .encoder_decoder import EncoderDecoder  # noqa: F401
from .dynamic_encoder_decoder import DynamicEncoderDecoder  # noqa: F401
