from .custom import (DATASETS, CityscapesDataset, CustomDataset, build_dataset,  # noqa: F401
                     eval_pipeline_kwargs, train_pipeline_kwargs)
from .gpu_pipeline import GpuTrainPipeline, draw_train_params  # noqa: F401
from .loader import FileBatchLoader, FileEvalLoader, epoch_indices  # noqa: F401
