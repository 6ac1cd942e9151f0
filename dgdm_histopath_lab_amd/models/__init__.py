"""Model layer of the DGDM hot path (mirror of the reference's ``dgdm_histopath.models``)."""
from .decoders import ClassificationHead, RegressionHead  # noqa: F401
from .dgdm_model import DGDMModel, ModelConfigurationError, ModelInferenceError, ValidationError  # noqa: F401
from .encoders import FeatureEncoder, GraphEncoder  # noqa: F401
