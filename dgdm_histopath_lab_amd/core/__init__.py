"""Core ops of the DGDM hot path (mirror of the reference's ``dgdm_histopath.core``)."""
from .attention import MultiHeadAttention, SpatialAttention  # noqa: F401
from .diffusion import DiffusionLayer, DiffusionScheduler  # noqa: F401
from .graph_layers import AdaptiveGraphPooling, DynamicGraphLayer, GraphContext, GraphConvolution, GraphUNet  # noqa: F401
