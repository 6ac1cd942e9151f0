"""Node attention of the DGDM hot path on the fused HIP kernels (K4/K5).

Host-side mirror of the reference's ``core/attention.py`` (class names, constructor arguments,
parameter names).  ``SpatialAttention`` runs the whole batch in one variable-length launch: the
reference loops over graphs in Python and materialises [1,H,N,N] scores, the distance matrix and
the dropout mask per graph (attention.py:261-283,135-157; dgdm_model.py:346-357)."""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import ops

KERNEL_HEAD_DIM = 16  # the MFMA tiling of csrc/attn_*.hip (DGDMModel's defaults: hidden_dims[-1] / attention_heads = 16)


class MultiHeadAttention(nn.Module):
    """Parameter container + projection helpers (reference: attention.py:16-181).  The dense
    general-purpose forward of the reference is not on the DGDM path; the two uses that are --
    spatial self-attention and the single-query pooling attention -- have fused paths
    (``SpatialAttention.forward_batch``, ``models.dgdm_model.GlobalAttentionPool``)."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.1, bias: bool = True, kdv_bias: bool = True,
                 batch_first: bool = True, add_zero_attn: bool = False):
        super().__init__()
        assert (embed_dim // num_heads) * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.batch_first, self.add_zero_attn = batch_first, add_zero_attn
        self.q_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.k_proj = nn.Linear(embed_dim, embed_dim, bias=kdv_bias)
        self.v_proj = nn.Linear(embed_dim, embed_dim, bias=kdv_bias)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        self.attn_dropout = nn.Dropout(dropout)
        self.resid_dropout = nn.Dropout(dropout)
        for lin in (self.q_proj, self.k_proj, self.v_proj, self.out_proj):
            nn.init.xavier_uniform_(lin.weight)
            if lin.bias is not None:
                nn.init.constant_(lin.bias, 0.0)

    @property
    def kernel_head_dim(self) -> int:
        """The head width the attention kernels run this module at: the next of ops.ATTN_HEAD_DIMS (16 on the MFMA kernels, 32 / 64
        on csrc/attn_gen.hip); narrower heads are zero-padded (scores and outputs are unchanged by zero columns)."""
        for D in ops.ATTN_HEAD_DIMS:
            if self.head_dim <= D:
                return D
        raise ops._lib.DGDMKernelError(f"attention kernels support head_dim <= {ops.ATTN_HEAD_DIMS[-1]}, got {self.head_dim}")

    def fused_qkv(self, x: Tensor) -> Tensor:
        """[N, 3*H*D] projection with every head zero-padded to the kernels' head dim D = ``kernel_head_dim``."""
        H, d, C, D = self.num_heads, self.head_dim, self.embed_dim, self.kernel_head_dim
        w = torch.cat([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight], dim=0)
        b = torch.cat([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias], dim=0) if self.q_proj.bias is not None else None
        if d != D:
            w = F.pad(w.view(3 * H, d, C), (0, 0, 0, D - d)).reshape(3 * H * D, C)
            if b is not None:
                b = F.pad(b.view(3 * H, d), (0, D - d)).reshape(-1)
        return ops.linear(x, w, b)

    def unpad_heads(self, o: Tensor) -> Tensor:
        D = self.kernel_head_dim
        if self.head_dim == D:
            return o
        return o.view(-1, self.num_heads, D)[:, :, : self.head_dim].reshape(-1, self.embed_dim)

    def forward(self, *args, **kwargs):
        raise NotImplementedError("dense MultiHeadAttention.forward is not on the DGDM hot path; use SpatialAttention / "
                                  "GlobalAttentionPool, which run the fused HIP kernels")


class SpatialAttention(nn.Module):
    """LN(x + out_proj(softmax(QK^T/sqrt(d) - |p_i-p_j|/tau) V)) with Q,K,V from x + sinusoid(pos)
    (reference: attention.py:184-327).  ``pos_encoding`` and ``spatial_proj`` are never used by the
    reference forward (attention.py:211,216); they exist for checkpoint compatibility."""

    def __init__(self, embed_dim: int, num_heads: int = 8, max_positions: int = 10000, dropout: float = 0.1,
                 temperature: float = 1.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.max_positions, self.temperature = max_positions, temperature
        self.attention = MultiHeadAttention(embed_dim, num_heads, dropout=dropout)
        self.pos_encoding = nn.Parameter(torch.randn(max_positions, embed_dim) * 0.02)  # dead
        self.spatial_proj = nn.Sequential(nn.Linear(2, embed_dim // 2), nn.ReLU(), nn.Linear(embed_dim // 2, embed_dim))  # dead
        self.norm = nn.LayerNorm(embed_dim)

    def forward_batch(self, x: Tensor, pos: Tensor, plan: ops.AttnPlan) -> Tensor:
        """x [N_tot, C], pos [N_tot, 2]; attention is restricted to each graph of ``plan``."""
        att = self.attention
        xp = ops.add_posenc(x, pos, plan)
        qkv = att.fused_qkv(xp)
        o = ops.spatial_attention(qkv, pos, plan, att.num_heads, 1.0 / math.sqrt(att.head_dim), 1.0 / self.temperature,
                                  att.attn_dropout.p, att.training)
        if ops.row_norm_supported(self.embed_dim, 1):
            # LN(x + dropout(out_proj(o))): projection, residual dropout, residual add and norm in one launch where the row fits
            return ops.linear_norm(att.unpad_heads(o), att.out_proj.weight, att.out_proj.bias, self.norm.weight, self.norm.bias, eps=self.norm.eps,
                                   res=x, pre_drop_p=att.resid_dropout.p, training=att.training)
        o = ops.act_dropout(ops.lin(att.out_proj, att.unpad_heads(o)), ops.ACT_NONE, att.resid_dropout.p, att.training)
        return self.norm(x + o)

    def attention_weights(self, x: Tensor, pos: Tensor, plan: ops.AttnPlan) -> List[Tensor]:
        """Head-mean attention weights per graph, [N_g, N_g] each (attention.py:171-173); only built
        when the caller asks for ``return_attention`` (O(N^2) memory by definition)."""
        att = self.attention
        with torch.no_grad():
            qkv = att.fused_qkv(ops.add_posenc(x, pos, plan))
            return ops.spatial_attention_mean_weights(qkv, pos, plan, att.num_heads, 1.0 / math.sqrt(att.head_dim),
                                                      1.0 / self.temperature)

    def forward(self, x: Tensor, positions: Tensor, mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """Reference signature: x [B, N, C], positions [B, N, 2] (all graphs the same size)."""
        if mask is not None:
            raise NotImplementedError("additive masks are not on the DGDM path")
        Bsz, n, C = x.shape
        plan = ops.AttnPlan([i * n for i in range(Bsz + 1)], x.device)
        xf, pf = x.reshape(Bsz * n, C), positions.reshape(Bsz * n, 2)
        out = self.forward_batch(xf, pf, plan).view(Bsz, n, C)
        w = torch.stack(self.attention_weights(xf, pf, plan))
        return out, w
