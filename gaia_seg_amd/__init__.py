"""MI355X-native implementation of the GAIA-seg supernet forward/backward hot path.

Package layout
  csrc/      hand-written HIP kernels for gfx950 + the C-ABI (include/gaiaseg_hip.h)
  hip/       ctypes binding, NHWC activation runtime, tape-aware operators
  core/      host-side mirror of the gaiavision / mmcv pieces the path needs
             (Registry, Config, DynamicMixin, dynamic bricks, model samplers, hooks, arena, DDP)
  models/    host-side mirror of gaiaseg.models (same registered names and constructors)
  apis/      train_segmentor (gaiaseg/apis/train.py)
"""
__version__ = "0.1.0"
