"""Encoders of the DGDM hot path (mirror of the reference's ``models/encoders.py``:
``FeatureEncoder`` and ``GraphEncoder`` with identical parameter names)."""
from __future__ import annotations

from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from ..core.graph_layers import DynamicGraphLayer, GraphContext, GraphConvolution, _context

EDGE_DIM = 32  # the reference hard-codes the edge feature width (encoders.py:183)


def _activation(name: str) -> nn.Module:
    return {"relu": nn.ReLU, "gelu": nn.GELU, "elu": nn.ELU}[name]()


def _norm(kind: str, dim: int) -> nn.Module:
    if kind == "layer":
        return nn.LayerNorm(dim)
    if kind == "batch":
        return nn.BatchNorm1d(dim)
    if kind == "instance":
        return nn.InstanceNorm1d(dim)
    return nn.Identity()


def _norm_act_dropout(h: Tensor, norm: nn.Module, act: nn.Module, drop: nn.Dropout, training: bool) -> Tensor:
    """dropout(act(norm(h))).  LayerNorm and InstanceNorm1d: ONE fused HIP kernel (csrc/rownorm.hip) with any of the reference's
    activations (ReLU / GELU / ELU, encoders.py:57-62).  ``nn.InstanceNorm1d(dim)`` on a 2-D ``[N, dim]`` input (encoders.py:95-100
    builds it with the defaults affine=False, track_running_stats=False) treats the rows as the channels of ONE unbatched sample and
    normalises each row over its ``dim`` entries: a LayerNorm without affine parameters -- the row kernel with gamma = 1, beta = 0.
    ``nn.BatchNorm1d`` (statistics over the NODES of the batch, running averages in eval mode): the column-norm kernels of
    csrc/colnorm.hip, activation and dropout included (``ops.batch_norm``: the module keeps its parameters, running statistics and
    ``state_dict`` keys); Identity: the activation + dropout kernel alone."""
    aid = ops.act_id(act)
    if aid is not None and ops.row_norm_supported(h.size(1), 1):
        if isinstance(norm, nn.LayerNorm):
            return ops.row_norm(h, norm.weight, norm.bias, eps=norm.eps, act=aid, drop_p=drop.p, training=training)
        if isinstance(norm, nn.InstanceNorm1d) and not norm.affine and not norm.track_running_stats:
            one = ops.device_constant([1.0] * h.size(1), torch.float32, h.device)
            zero = ops.device_constant([0.0] * h.size(1), torch.float32, h.device)
            return ops.row_norm(h, one, zero, eps=norm.eps, act=aid, drop_p=drop.p, training=training)
    if aid is not None and isinstance(norm, nn.BatchNorm1d) and ops.batch_norm_supported(norm, h):
        return ops.batch_norm(h, norm, aid, drop.p, training)
    h = norm(h)
    if aid is not None and h.numel() % 4 == 0:
        return ops.act_dropout(h, aid, drop.p, training)
    return drop(act(h))


class FeatureEncoder(nn.Module):
    """[Linear -> norm -> act -> dropout] x num_layers, + residual projection of the input
    (reference: encoders.py:19-124; ``encoder.{0,1,4,5}`` / ``residual_proj`` keys)."""

    def __init__(self, input_dim: int, hidden_dim: int, num_layers: int = 2, dropout: float = 0.1, activation: str = "gelu",
                 normalization: str = "layer", use_residual: bool = True):
        super().__init__()
        if activation not in ("relu", "gelu", "elu"):
            raise ValueError(f"Unknown activation: {activation}")
        self.input_dim, self.hidden_dim, self.num_layers, self.use_residual = input_dim, hidden_dim, num_layers, use_residual
        self.activation = _activation(activation)
        layers: List[nn.Module] = []
        for i in range(num_layers):
            layers += [nn.Linear(input_dim if i == 0 else hidden_dim, hidden_dim), _norm(normalization, hidden_dim),
                       self.activation, nn.Dropout(dropout)]
        self.encoder = nn.Sequential(*layers)
        self.residual_proj = nn.Linear(input_dim, hidden_dim) if (use_residual and input_dim != hidden_dim) else None

    def forward(self, x: Tensor) -> Tensor:
        h = x
        mods = list(self.encoder)
        for i in range(0, len(mods), 4):  # (Linear, norm, act, dropout) quadruples
            lin, norm, act, drop = mods[i:i + 4]
            h = ops.lin(lin, h)
            h = _norm_act_dropout(h, norm, act, drop, self.training)
        if self.use_residual:
            if self.residual_proj is not None:   # the projection's GEMM accumulates into h: no separate add
                return ops.linear_add_into(h, x, self.residual_proj.weight, self.residual_proj.bias)
            h = h + x
        return h


class GraphEncoder(nn.Module):
    """Stack of graph layers, each followed by norm -> act -> dropout, then a Linear
    (reference: encoders.py:127-280).

    Repair R2 (SURVEY.md 8(a') D3): a ``DynamicGraphLayer(node_dim=in, hidden=out)`` returns an
    ``in``-wide tensor, so where in != out the reference's ``LayerNorm(out)`` cannot run; a
    ``dim_proj[i] = Linear(in, out)`` is applied between the layer and its norm (extra keys
    ``graph_encoder.dim_proj.{i}.*``; all reference keys are kept)."""

    def __init__(self, input_dim: int, hidden_dims: List[int], num_layers: int = 4, attention_heads: int = 8,
                 dropout: float = 0.1, activation: str = "gelu", normalization: str = "layer", use_edge_features: bool = True,
                 aggregation: str = "add"):
        super().__init__()
        self.input_dim, self.hidden_dims, self.num_layers, self.use_edge_features = input_dim, hidden_dims, num_layers, use_edge_features
        dims = [input_dim] + list(hidden_dims)
        if num_layers > len(dims):
            # the reference dies with a bare IndexError here (encoders.py:176, D2)
            raise ValueError(f"graph_layers={num_layers} needs at least {num_layers - 1} hidden_dims, got {len(hidden_dims)}")
        self.graph_layers, self.norm_layers, self.dim_proj = nn.ModuleList(), nn.ModuleList(), nn.ModuleDict()
        for i in range(num_layers):
            din, dout = dims[i], dims[min(i + 1, len(dims) - 1)]
            if use_edge_features:
                self.graph_layers.append(DynamicGraphLayer(node_dim=din, edge_dim=EDGE_DIM, hidden_dim=dout,
                                                           num_heads=attention_heads, dropout=dropout))
                if din != dout:
                    self.dim_proj[str(i)] = nn.Linear(din, dout)
            else:
                self.graph_layers.append(GraphConvolution(in_channels=din, out_channels=dout, edge_dim=None))
            self.norm_layers.append(_norm(normalization, dout))
        self.activation = _activation(activation) if activation in ("relu", "gelu", "elu") else nn.ReLU()
        self.dropout = nn.Dropout(dropout)
        self.output_proj = nn.Linear(dims[-1], dims[-1])

    def forward(self, x: Tensor, edge_index: Union[Tensor, GraphContext], edge_attr: Optional[Tensor] = None,
                batch: Optional[Tensor] = None) -> Dict[str, Tensor]:
        ctx = _context(edge_index, x, edge_attr if self.use_edge_features else None)
        h, outs = x, []
        for i, (layer, norm) in enumerate(zip(self.graph_layers, self.norm_layers)):
            h = layer(h, ctx)
            aid = ops.act_id(self.activation)
            if str(i) in self.dim_proj and isinstance(norm, nn.LayerNorm) and aid is not None and ops.row_norm_supported(norm.normalized_shape[0], 1):
                dp = self.dim_proj[str(i)]      # Linear -> LayerNorm -> act -> dropout: the norm as the GEMM's epilogue where it fits
                h = ops.linear_norm(h, dp.weight, dp.bias, norm.weight, norm.bias, eps=norm.eps, act=aid, drop_p=self.dropout.p,
                                    training=self.training)
            else:
                if str(i) in self.dim_proj:
                    h = ops.lin(self.dim_proj[str(i)], h)
                h = _norm_act_dropout(h, norm, self.activation, self.dropout, self.training)
            outs.append(h)
        return {"embeddings": ops.lin(self.output_proj, h), "layer_outputs": outs, "num_nodes": x.size(0)}
