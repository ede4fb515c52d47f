from .gpu_pipeline import GpuTrainPipeline, draw_train_params  # noqa: F401
