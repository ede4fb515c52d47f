from .dynamic_res_layer import DynamicResLayer  # noqa: F401
