from .decode_head import DynamicBaseDecodeHead  # noqa: F401
from .dynamic_fcn_head import DynamicFCNHead  # noqa: F401
from .dynamic_psp_head import DynamicPPM, DynamicPSPHead  # noqa: F401
from .dynamic_uper_head import DynamicUPerHead  # noqa: F401
