from .registry import Registry, build_from_cfg  # noqa: F401
from .config import Config, ConfigDict, DictAction  # noqa: F401
from .dynamic import DynamicMixin, fold_dict, unfold_dict  # noqa: F401
