"""Host-side mirror of ``gaiaseg.models`` (same registry names, constructors and methods)."""
from .builder import (BACKBONES, HEADS, LOSSES, NECKS, PIXEL_SAMPLERS, SEGMENTORS,  # noqa: F401
                      build_backbone, build_head, build_loss, build_neck, build_pixel_sampler,
                      build_segmentor)
from .backbones import *  # noqa: F401,F403
from .decode_heads import *  # noqa: F401,F403
from .losses import *  # noqa: F401,F403
from .pixel_samplers import OHEMPixelSampler  # noqa: F401
from .segmentors import *  # noqa: F401,F403
from .utils import *  # noqa: F401,F403
