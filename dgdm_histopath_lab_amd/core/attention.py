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
    """reference: attention.py:16-181.  The two uses of the class ON the DGDM path -- spatial self-attention over the graphs of a
    batch and the single-query pooling attention -- run on fused variable-length kernels (``SpatialAttention.forward_batch``,
    ``models.dgdm_model.GlobalAttentionPool``); the class's own dense ``forward`` (query / key / value [B, L, C], ``attn_mask``,
    ``key_padding_mask``, ``need_weights``) runs on csrc/attn_dense.hip."""

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

    @property
    def dense_head_dim(self) -> int:
        """Head width of the dense kernels (csrc/attn_dense.hip) for this module: the next of ops.ATTN_DENSE_HEAD_DIMS."""
        for D in ops.ATTN_DENSE_HEAD_DIMS:
            if self.head_dim <= D:
                return D
        raise ops._lib.DGDMKernelError(f"dense attention kernels support head_dim <= {ops.ATTN_DENSE_HEAD_DIMS[-1]}, got {self.head_dim}")

    def _padded(self, lins, D: int):
        """Weights / biases of the projections ``lins`` stacked, every head zero-padded from head_dim to D rows."""
        H, d, C, n = self.num_heads, self.head_dim, self.embed_dim, len(lins)
        w = lins[0].weight if n == 1 else torch.cat([m.weight for m in lins], dim=0)
        b = None
        if all(m.bias is not None for m in lins):
            b = lins[0].bias if n == 1 else torch.cat([m.bias for m in lins], dim=0)
        elif any(m.bias is not None for m in lins):        # bias=True with kdv_bias=False (or the reverse): zeros where there is none
            b = torch.cat([m.bias if m.bias is not None else lins[0].weight.new_zeros(C) for m in lins], dim=0)
        if d != D:
            w = F.pad(w.view(n * H, d, C), (0, 0, 0, D - d)).reshape(n * H * D, C)
            if b is not None:
                b = F.pad(b.view(n * H, d), (0, D - d)).reshape(-1)
        return w, b

    def fused_qkv(self, x: Tensor, D: Optional[int] = None) -> Tensor:
        """[N, 3*H*D] projection with every head zero-padded to the kernels' head dim D (default ``kernel_head_dim``)."""
        return ops.linear(x, *self._padded((self.q_proj, self.k_proj, self.v_proj), self.kernel_head_dim if D is None else D))

    def unpad_heads(self, o: Tensor, D: Optional[int] = None) -> Tensor:
        D = self.kernel_head_dim if D is None else D
        if self.head_dim == D:
            return o
        return o.view(-1, self.num_heads, D)[:, :, : self.head_dim].reshape(-1, self.embed_dim)

    def forward(self, query: Tensor, key: Optional[Tensor] = None, value: Optional[Tensor] = None,
                key_padding_mask: Optional[Tensor] = None, need_weights: bool = True, attn_mask: Optional[Tensor] = None,
                average_attn_weights: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
        """The reference's forward (attention.py:73-181), same arguments and returns: query [B, L, C] (or [L, B, C] with
        ``batch_first=False``), key / value default to query / key; ``attn_mask`` float (added to the scaled scores) or bool (-inf where
        True), any shape that broadcasts against [B, H, L, S] as the reference's in-place ops do (a 3-D [X, L, S] mask therefore meets
        the HEAD axis, attention.py:131-135); ``key_padding_mask`` [B, S] bool; ``add_zero_attn`` appends one zero key / value.
        Returns (output, weights): weights AFTER dropout (attention.py:145-146), head mean [B, L, S] or [B * H, L, S]; they carry no
        gradient here (the reference's do; nothing on the DGDM path differentiates through them), and neither does a float mask.
        A row whose keys are all masked is NaN, as in the reference."""
        if query.dim() != 3:
            raise ValueError(f"query must be [batch, seq, embed_dim] (or [seq, batch, embed_dim] with batch_first=False), got {tuple(query.shape)}")
        if not self.batch_first:
            query = query.transpose(0, 1)
            key = None if key is None else key.transpose(0, 1)
            value = None if value is None else value.transpose(0, 1)
        B, L, C = query.shape
        self_attn = key is None and value is None
        if key is None:
            key = query
        if value is None:
            value = key
        S = key.shape[1]
        H, D = self.num_heads, self.dense_head_dim
        HD = H * D
        if self_attn:
            qkv = self.fused_qkv(query.reshape(B * L, C), D)
            q, k, v = qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
        else:
            q = ops.linear(query.reshape(B * L, C), *self._padded((self.q_proj,), D))
            k = ops.linear(key.reshape(B * S, C), *self._padded((self.k_proj,), D))
            v = ops.linear(value.reshape(-1, C), *self._padded((self.v_proj,), D))
        if self.add_zero_attn:          # attention.py:118-126: one all-zero key / value per sequence, never masked
            zero = q.new_zeros(B, 1, HD)
            k = torch.cat([k.reshape(B, S, HD), zero], dim=1).reshape(B * (S + 1), HD)
            v = torch.cat([v.reshape(B, S, HD), zero], dim=1).reshape(B * (S + 1), HD)
            if attn_mask is not None:
                attn_mask = F.pad(attn_mask, (0, 1))
            if key_padding_mask is not None:
                key_padding_mask = F.pad(key_padding_mask, (0, 1))
            S += 1
        mask = ops.DenseMask(attn_mask, key_padding_mask, B, H, L, S, query.device)
        scale = 1.0 / math.sqrt(self.head_dim)
        o, lse, seed = ops.attn_dense(q, k, v, B, L, S, H, scale, mask, self.attn_dropout.p, self.training)
        out = ops.act_dropout(ops.lin(self.out_proj, self.unpad_heads(o, D)), ops.ACT_NONE, self.resid_dropout.p, self.training).view(B, L, C)
        if not self.batch_first:
            out = out.transpose(0, 1)
        weights = None
        if need_weights:
            p = self.attn_dropout.p if self.training else 0.0
            weights = ops.attn_dense_weights(q, k, lse, B, L, S, H, scale, mask, p, seed, per_head=not average_attn_weights)
        return out, weights


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

    def forward_batch(self, x: Tensor, pos: Tensor, plan: ops.AttnPlan, pos_extent: Optional[float] = None) -> Tensor:
        """x [N_tot, C], pos [N_tot, 2]; attention is restricted to each graph of ``plan``.  ``pos_extent``: host-side upper bound of
        the coordinate range inside any graph (GraphData.host_pos_extent), None when unknown -- see ops.attn_zero_blocks_possible."""
        att = self.attention
        xp = ops.add_posenc(x, pos, plan)
        qkv = att.fused_qkv(xp)
        o = ops.spatial_attention(qkv, pos, plan, att.num_heads, 1.0 / math.sqrt(att.head_dim), 1.0 / self.temperature,
                                  att.attn_dropout.p, att.training, pos_extent=pos_extent)
        return self._project_and_norm(x, att.unpad_heads(o))

    def _project_and_norm(self, x: Tensor, o: Tensor) -> Tensor:
        """LN(x + resid_dropout(out_proj(o))) (attention.py:176-181,325)."""
        att = self.attention
        if ops.row_norm_supported(self.embed_dim, 1):
            # LN(x + dropout(out_proj(o))): projection, residual dropout, residual add and norm in one launch where the row fits
            return ops.linear_norm(o, att.out_proj.weight, att.out_proj.bias, self.norm.weight, self.norm.bias, eps=self.norm.eps,
                                   res=x, pre_drop_p=att.resid_dropout.p, training=att.training)
        o = ops.act_dropout(ops.lin(att.out_proj, o), ops.ACT_NONE, att.resid_dropout.p, att.training)
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
        """Reference signature (attention.py:285-327): x [B, N, C], positions [B, N, 2] (all graphs the same size), optional ``mask``
        ADDED to the spatial bias (``attn_mask = mask + spatial_bias``, :311-314; a bool mask therefore counts as 0 / 1, not -inf).
        Returns (output, head-mean weights [B, N, N]).

        The reference hands its [B, N, N] bias to an in-place add against [B, H, N, N] scores (attention.py:135), which only
        broadcasts for B = 1 -- the way DGDMModel calls it, one graph at a time.  For B = 1 this forward follows that chain to the
        letter (a mask of shape [H, N, N] meets the head axis).  B > 1 is an extension: every sequence gets its own bias, positional
        encodings are normalised per sequence, and ``mask`` may be [N, N], [B, N, N] (per sequence) or [B, 1 | H, N, N]."""
        Bsz, n, C = x.shape
        plan = ops.AttnPlan([i * n for i in range(Bsz + 1)], x.device)
        xf, pf = x.reshape(Bsz * n, C), positions.reshape(Bsz * n, 2)
        if mask is None:
            out = self.forward_batch(xf, pf, plan).view(Bsz, n, C)
            w = torch.stack(self.attention_weights(xf, pf, plan))
            return out, w
        att = self.attention
        H, D = att.num_heads, att.dense_head_dim
        m = mask.float() if mask.dtype != torch.float32 else mask
        if Bsz == 1:
            m = m.expand(torch.broadcast_shapes(tuple(m.shape), (1, n, n)))      # mask + spatial_bias, then against [1, H, N, N]
        elif m.dim() == 3:
            m = m.unsqueeze(1)                                                  # [B, N, N]: one mask per sequence
        pf32 = pf.float()
        dm = ops.DenseMask(m, None, Bsz, H, n, n, x.device, posq=pf32, posk=pf32, inv_tau=1.0 / self.temperature)
        qkv = att.fused_qkv(ops.add_posenc(xf, pf, plan), D)
        HD = H * D
        scale = 1.0 / math.sqrt(att.head_dim)
        o, lse, seed = ops.attn_dense(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], Bsz, n, n, H, scale, dm, att.attn_dropout.p, att.training)
        out = self._project_and_norm(xf, att.unpad_heads(o, D)).view(Bsz, n, C)
        p = att.attn_dropout.p if att.training else 0.0
        w = ops.attn_dense_weights(qkv[:, :HD], qkv[:, HD:2 * HD], lse, Bsz, n, n, H, scale, dm, p, seed)
        return out, w
