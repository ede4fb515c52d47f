from .train import set_random_seed, train_segmentor  # noqa: F401
