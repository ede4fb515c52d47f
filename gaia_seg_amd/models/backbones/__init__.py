from .dynamic_resnet import DynamicResNet  # noqa: F401
