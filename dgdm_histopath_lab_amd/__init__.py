"""dgdm_histopath_lab_amd -- MI355X-native (gfx950) implementation of the DGDM hot path of
danieleschmidt/dgdm-histopath-lab: DGDMModel forward/backward over a batch of tissue graphs,
on hand-written HIP kernels behind a C ABI (include/dgdm_hip.h).  GPU only: there is no CPU
fallback, ops raise ``DGDMKernelError`` when libdgdm_hip.so or a GPU is missing."""
from ._lib import DGDMKernelError  # noqa: F401
from .graph import GraphBatch, GraphData, GraphStructure  # noqa: F401

__version__ = "0.1.0"
from .models.dgdm_model import DGDMModel, ModelConfigurationError, ModelInferenceError, ValidationError  # noqa: F401,E402
from .optim import DGDMAdamW  # noqa: F401,E402
from .training import BatchLayoutError, DGDMTrainer, GraphedPretrainStep, predict_graph  # noqa: F401,E402
