from .cross_entropy_loss import CrossEntropyLoss, accuracy, seg_loss_and_accuracy  # noqa: F401
