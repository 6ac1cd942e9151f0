"""Message-passing layers of the DGDM hot path on the HIP kernels (K1/K2).

Host-side mirror of the reference's ``core/graph_layers.py`` (same class names, constructor
arguments, parameter names -> identical ``state_dict`` keys); the arithmetic runs through
``libdgdm_hip.so``:

* the edge list is turned into CSR/CSC + GCN weights ONCE per (edge list, node count) by
  ``GraphContext`` (K1) instead of once per convolution call (graph_layers.py:76-84);
* ``GraphConvolution`` aggregates first and projects second,
  ``out = [A_hat x | EA_hat] @ [W | W_e]^T + b``: A_hat(xW^T) == (A_hat x)W^T, and because
  ``edge_lin`` has no bias (graph_layers.py:49) the per-edge term ``sum_e norm_e W_e a_e``
  equals ``(sum_e norm_e a_e) W_e^T`` where the 32-wide aggregate EA_hat depends only on the
  graph -- it is computed once per GraphContext (K2) and shared by every convolution;
* self-loop edges carry a zero attribute row (repair R1, SURVEY.md 8(a')).
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import ops
from .._lib import DGDMKernelError
from ..graph import GraphStructure


class GraphContext:
    """Everything the convolutions need from one edge list over ``num_nodes`` nodes."""

    __slots__ = ("gs", "ea_hat", "num_nodes", "edge_index", "edge_attr", "max_degree")

    def __init__(self, edge_index: Tensor, num_nodes: int, edge_attr: Optional[Tensor] = None, add_loops: bool = True,
                 normalize: bool = True, max_degree: Optional[int] = None):
        """``max_degree``: host-side upper bound of the edge list's largest degree (GraphData.host_max_degree) or None; a pooled level
        of the U-Net keeps a subset of its parent's edges, so the parent's bound holds for it."""
        self.max_degree = max_degree
        self.gs = GraphStructure(edge_index, num_nodes, add_loops=add_loops, normalize=normalize, max_degree=max_degree)
        if edge_attr is not None and edge_attr.dim() == 2 and edge_attr.size(1) % 4:
            # the gather kernels move 16-byte pieces: attribute widths that are not multiples of 4 get zero columns (the matching
            # zero columns of edge_lin.weight are added by GraphConvolution.forward; the products are unchanged)
            edge_attr_k = F.pad(edge_attr, (0, -edge_attr.size(1) % 4))
        else:
            edge_attr_k = edge_attr
        self.ea_hat = ops.aggregate_edge_attr(edge_attr_k, self.gs)  # None when there are no attributes
        self.num_nodes = num_nodes
        self.edge_index, self.edge_attr = edge_index, edge_attr


    def extended(self, num_nodes: int) -> "GraphContext":
        """The context of the same edge list over more nodes (GraphStructure.extended): one launch instead of a CSR build and
        an edge-attribute aggregation."""
        c = GraphContext.__new__(GraphContext)
        c.gs, c.ea_hat = self.gs.extended(num_nodes, self.ea_hat)
        c.num_nodes, c.edge_index, c.edge_attr, c.max_degree = num_nodes, self.edge_index, self.edge_attr, self.max_degree
        return c


def _context(edge_index: Union[Tensor, GraphContext], x: Tensor, edge_attr: Optional[Tensor], add_loops=True) -> GraphContext:
    if isinstance(edge_index, GraphContext):
        return edge_index
    return GraphContext(edge_index, x.size(0), edge_attr, add_loops)


class GraphConvolution(nn.Module):
    """GCN-style convolution with an edge-feature term (reference: graph_layers.py:19-110)."""

    def __init__(self, in_channels: int, out_channels: int, edge_dim: Optional[int] = None, bias: bool = True,
                 add_self_loops: bool = True, normalize: bool = True, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.edge_dim = in_channels, out_channels, edge_dim
        self.add_self_loops, self.normalize = add_self_loops, normalize
        self.node_lin = nn.Linear(in_channels, out_channels, bias=False)
        self.edge_lin = nn.Linear(edge_dim, out_channels, bias=False) if edge_dim is not None else None
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.node_lin.weight)
        if self.edge_lin is not None:
            nn.init.xavier_uniform_(self.edge_lin.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, x: Tensor, edge_index: Union[Tensor, GraphContext], edge_attr: Optional[Tensor] = None, size=None,
                return_skip: bool = False):
        """``return_skip=True`` returns ``(out, x_skip)``: ``x_skip`` is ``x`` for a residual connection of the caller; on the
        fused path it is an alias routed through this convolution's autograd node, so the residual's gradient is added inside
        the convolution's backward kernel instead of by a separate element-wise pass."""
        if not self.normalize:
            # graph_layers.py:76-86: no norm and -- the loops are added inside the ``if self.normalize`` branch -- no self loops:
            # out[d] = sum over the incoming edges of (x W^T)[src] + W_e a_e.  An index set of its own (unit weights).
            shared = isinstance(edge_index, GraphContext)
            ctx = GraphContext(edge_index.edge_index if shared else edge_index, x.size(0), edge_index.edge_attr if shared else edge_attr,
                               add_loops=False, normalize=False, max_degree=edge_index.max_degree if shared else None)
        else:
            ctx = _context(edge_index, x, edge_attr, self.add_self_loops)
        if self.edge_lin is not None and ctx.ea_hat is not None:
            # one contraction over K = C_in + edge_dim:  [A_hat x | EA_hat] . [W | W_e]^T + b
            we = self.edge_lin.weight
            if ctx.ea_hat.size(1) != we.size(1):      # attribute width padded to a multiple of 4 by GraphContext
                we = F.pad(we, (0, ctx.ea_hat.size(1) - we.size(1)))
            if (x.size(0) >= ops.GEMM_MIN_ROWS and x.size(1) % 4 == 0 and self.out_channels % 4 == 0 and ctx.ea_hat.size(1) % 4 == 0
                    and x.dtype == torch.float32):
                return ops.graph_conv_linear(x, ctx.ea_hat, ctx.gs, self.node_lin.weight, we, self.bias, skip=return_skip)
            buf = ops.aggregate_concat(x, ctx.ea_hat, ctx.gs)
            out = ops.linear(buf, torch.cat([self.node_lin.weight, we], dim=1), self.bias)
        else:
            out = ops.linear(ops.aggregate(x, ctx.gs), self.node_lin.weight, self.bias)
        return (out, x) if return_skip else out


class DynamicGraphLayer(nn.Module):
    """norm1(output_proj(GELU(conv2(GELU(conv1(x))))) + x)  (reference: graph_layers.py:113-247).

    ``compute_dynamic_edges`` (:160-205) is evaluated and thrown away by the reference (:227-230):
    it changes neither outputs nor gradients and is not executed here.  Its parameters
    (``node_to_qkv``, ``edge_to_key``) and the unused ``norm2`` still exist so that reference
    checkpoints load."""

    def __init__(self, node_dim: int, edge_dim: int, hidden_dim: int, num_heads: int = 8, dropout: float = 0.1,
                 use_layer_norm: bool = True):
        super().__init__()
        assert hidden_dim % num_heads == 0, "hidden_dim must be divisible by num_heads"
        self.node_dim, self.edge_dim, self.hidden_dim, self.num_heads = node_dim, edge_dim, hidden_dim, num_heads
        self.head_dim = hidden_dim // num_heads
        self.node_to_qkv = nn.Linear(node_dim, hidden_dim * 3)   # dead in the reference forward
        self.edge_to_key = nn.Linear(edge_dim, hidden_dim)       # dead in the reference forward
        self.graph_conv1 = GraphConvolution(node_dim, hidden_dim, edge_dim)
        self.graph_conv2 = GraphConvolution(hidden_dim, hidden_dim, edge_dim)
        self.output_proj = nn.Linear(hidden_dim, node_dim)
        self.dropout = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(node_dim) if use_layer_norm else nn.Identity()
        self.norm2 = nn.LayerNorm(node_dim) if use_layer_norm else nn.Identity()  # unused (graph_layers.py:152)
        self.activation = nn.GELU()

    def forward(self, x: Tensor, edge_index: Union[Tensor, GraphContext], edge_attr: Optional[Tensor] = None) -> Tensor:
        ctx = _context(edge_index, x, edge_attr)
        p, tr = self.dropout.p, self.training
        c1, c2 = self.graph_conv1, self.graph_conv2
        if (isinstance(self.norm1, nn.LayerNorm) and c1.edge_lin is not None and c2.edge_lin is not None and c1.bias is not None
                and c2.bias is not None and c1.add_self_loops and c2.add_self_loops
                and ops.graph_layer_supported(x, ctx.ea_hat, self.node_dim, self.hidden_dim, self.edge_dim)):
            # the whole layer as one autograd node: 5 launches forward, 6 backward, activations and the norm as GEMM epilogues
            return ops.graph_layer(x, ctx.ea_hat, ctx.gs, c1, c2, self.output_proj, self.norm1, p, tr)
        h, xs = self.graph_conv1(x, ctx, return_skip=True)      # xs: x for the residual below (see GraphConvolution.forward)
        h = ops.act_dropout(h, ops.ACT_GELU, p, tr)
        h = ops.act_dropout(self.graph_conv2(h, ctx), ops.ACT_GELU, p, tr)
        out = ops.lin(self.output_proj, h)
        if isinstance(self.norm1, nn.LayerNorm) and ops.row_norm_supported(self.node_dim, 1):
            return ops.row_norm(out, self.norm1.weight, self.norm1.bias, res=xs, eps=self.norm1.eps)
        return self.norm1(out + xs)


class AdaptiveGraphPooling(nn.Module):
    """Top-k node pooling (reference: graph_layers.py:250-329).  k = max(1, int(ratio*N)) over ALL
    nodes of the batch (the reference ignores ``batch``); kept nodes in ascending id order.

    Sync-free variant of the edge filter: instead of compacting the edge list (data-dependent
    size) the pooled ``edge_index`` keeps its E columns and dropped edges are marked with the
    out-of-range id ``-1``; the CSR builder ignores out-of-range edges, and ``eid`` still indexes
    the un-compacted ``edge_attr``.  ``compact=True`` reproduces the reference's compacted
    tensors (one host sync) for callers that need them."""

    def __init__(self, in_channels: int, ratio: float = 0.5, min_score: Optional[float] = None, multiplier: float = 1.0,
                 nonlinearity: str = "tanh"):
        super().__init__()
        self.in_channels, self.ratio, self.min_score, self.multiplier = in_channels, ratio, min_score, multiplier
        self.score_net = nn.Sequential(nn.Linear(in_channels, in_channels // 2), nn.ReLU(), nn.Linear(in_channels // 2, 1))
        self._nl = nonlinearity if nonlinearity in ("tanh", "softmax") else "sigmoid"      # graph_layers.py:277-283: anything else is sigmoid

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Optional[Tensor] = None, batch: Optional[Tensor] = None,
                compact: bool = False, return_node_map: bool = False, *, relu_decisions: Optional[Tensor] = None,
                perm_decision: Optional[Tensor] = None, trace: Optional[dict] = None, trace_tag: str = ""):
        """``relu_decisions`` / ``perm_decision`` (parity tests only): the side of the score MLP's ReLU kink per element and the
        kept node ids, taken from the reference run instead of being decided here (see ``GraphUNet.forward``).  ``trace``
        receives this level's own pre-activation and scores so that the test can hold every decision that differs to the
        rounding margin."""
        n = x.size(0)
        k = max(1, int(self.ratio * n))
        h = ops.lin(self.score_net[0], x)
        w2, b2 = self.score_net[2].weight, self.score_net[2].bias
        if not (x.is_cuda and x.dtype == torch.float32 and ops.pool_supported(x.size(1), h.size(1)) and n < 2 ** 31):
            # no torch branch: a silent change of arithmetic (and, before round 4, of the min_score semantics) is worse than an error
            raise DGDMKernelError(f"AdaptiveGraphPooling: the K9 kernels take fp32 CUDA tensors whose node width and score width are "
                                  f"multiples of 4 with a score width <= 256 that ops.pool_supported accepts; got {x.dtype} on {x.device}, "
                                  f"node width {x.size(1)}, score width {h.size(1)}")
        # K9 kernels: no data-dependent shapes; no host sync unless min_score is set
        s = ops.pool_score(h, w2, b2, decide=relu_decisions, nonlinearity=self._nl)
        if trace is not None:
            trace[f"pre.pool{trace_tag}"], trace[f"score{trace_tag}"] = h.detach(), s.detach()
        if self.min_score is not None:
            # the reference keeps scores >= min_score (graph_layers.py:302-303): a data-dependent count, read back once; the
            # nodes with a score >= the threshold ARE the `count` largest, so the exact top-k selection below yields that set
            k = ops.count_ge(s, self.min_score)
            if perm_decision is not None and perm_decision.numel() != k:
                raise ValueError(f"injected perm has {perm_decision.numel()} entries, min_score keeps {k}")
            if k == 0:
                # as the reference (mask.nonzero() is empty, graph_layers.py:302-310): an empty pooled graph, every edge dropped
                perm = torch.empty(0, dtype=torch.int64, device=x.device)
                node_map = torch.full((n,), -1, dtype=torch.int32, device=x.device)
                pooled_x = x[:0] * s[:0].unsqueeze(-1)       # [0, C], attached to the graph of x and s
                if compact:
                    out = (pooled_x, edge_index[:, :0], (edge_attr[:0] if edge_attr is not None else None), perm)
                else:
                    out = (pooled_x, torch.full_like(edge_index, -1), edge_attr, perm)
                return out + (node_map,) if return_node_map else out
        perm, node_map = ops.topk_perm(s, k)
        if trace is not None:
            trace[f"own_perm{trace_tag}"] = perm
        if perm_decision is not None:
            perm = perm_decision.to(device=x.device, dtype=torch.int64)
            if perm.numel() != k:
                raise ValueError(f"injected perm has {perm.numel()} entries, this level keeps {k}")
            node_map = torch.full((n,), -1, dtype=torch.int32, device=x.device)
            node_map[perm] = torch.arange(k, dtype=torch.int32, device=x.device)
        pooled_x = ops.pool_gather(x, s, perm, node_map, self.multiplier)
        mapped_keep = ops.edge_relabel(edge_index, node_map)     # dropped edges (now or earlier) are (-1, -1)
        if compact:  # reference layout (graph_layers.py:322-327); boolean indexing syncs
            keep = mapped_keep[0] >= 0
            out = (pooled_x, mapped_keep[:, keep], (edge_attr[keep] if edge_attr is not None else None), perm)
        else:
            out = (pooled_x, mapped_keep, edge_attr, perm)
        return out + (node_map,) if return_node_map else out


class GraphUNet(nn.Module):
    """Graph U-Net (reference: graph_layers.py:332-458) with the repairs that make it runnable:
    R5a layers use the data's edge dim (32) instead of ``hidden_channels``; R5b ``up_convs`` take
    ``hidden_channels`` inputs (sum skip).  ``strict_reference=True`` replicates D10: decoder level
    j convolves with ``edge_indices[j+1]`` (the coarser graph's ids, graph_layers.py:453)."""

    def __init__(self, in_channels: int, hidden_channels: int, out_channels: int, depth: int = 3, pool_ratios=None,
                 sum_res: bool = True, act: str = "relu", edge_dim: int = 32, strict_reference: bool = True):
        super().__init__()
        if not sum_res:
            raise NotImplementedError("sum_res=False cannot run in the reference either (D7)")
        self.in_channels, self.hidden_channels, self.out_channels = in_channels, hidden_channels, out_channels
        self.depth, self.sum_res, self.strict_reference = depth, sum_res, strict_reference
        pool_ratios = pool_ratios or [0.5] * depth
        self.act = {"relu": F.relu, "gelu": F.gelu}.get(act, F.elu)
        mk = lambda cin: DynamicGraphLayer(cin, edge_dim, hidden_channels)
        self.down_convs = nn.ModuleList([mk(in_channels)] + [mk(hidden_channels) for _ in range(depth)])
        self.pools = nn.ModuleList([AdaptiveGraphPooling(hidden_channels, ratio=pool_ratios[i]) for i in range(depth)])
        self.bottom_conv = mk(hidden_channels)
        self.up_convs = nn.ModuleList([mk(hidden_channels) for _ in range(depth)])
        self.final_conv = nn.Linear(hidden_channels, out_channels)

    def _relu(self, x: Tensor, decide: Optional[Tensor] = None) -> Tensor:
        if self.act is F.relu:
            return ops.act_dropout(x, ops.ACT_RELU, decide=decide)      # raises DGDMKernelError for shapes / devices without a kernel
        if decide is not None:
            raise NotImplementedError("decision injection is wired into the ReLU kernels only")
        return self.act(x)

    def forward(self, x: Tensor, edge_index: Union[Tensor, GraphContext], edge_attr: Optional[Tensor] = None,
                batch: Optional[Tensor] = None, trace: Optional[dict] = None, decisions: Optional[dict] = None) -> Tensor:
        """``decisions`` (parity tests only; ``None`` on every product call): the discrete choices of a reference run of the same
        input -- ``relu.down{i}`` / ``relu.pool{i}`` / ``relu.bottom`` / ``relu.up{i}`` (bool, shape of the activation: which
        elements pass the ReLU) and ``perm{i}`` (kept node ids of pooling level i).  The kernels then take the side of every kink
        from them, forward and backward, so that this network and the reference differentiate the same piecewise-linear function;
        a pre-activation within rounding of zero (2 of 512 000 at 2 x 2 000 nodes) would otherwise make the gradients
        incomparable.  ``trace`` additionally receives the pre-activations (``pre.*``), this run's own scores and top-k choice, so
        that the caller can check that every decision that differs from the injected one lies within the rounding margin."""
        dec = decisions or {}
        ctx0 = _context(edge_index, x, edge_attr)
        eis, eas = [ctx0.edge_index], [ctx0.edge_attr]
        ctxs = {(0, x.size(0)): ctx0}

        sizes = {0: x.size(0)}       # nodes of level k (the relabelled edge list of level k has ids < sizes[k])

        def level(k: int, n: int) -> GraphContext:
            if (k, n) not in ctxs:
                base = ctxs.get((k, sizes.get(k, -1)))
                if ops.CSR_EXTEND and base is not None and n > base.num_nodes > 0:
                    # the reference's decoder (D10): the edge list of the coarser level over the finer level's nodes -- the
                    # coarser level's index set plus one self loop per extra node, copied instead of built
                    ctxs[(k, n)] = base.extended(n)
                else:
                    ctxs[(k, n)] = GraphContext(eis[k], n, eas[k], max_degree=ctx0.max_degree)
            return ctxs[(k, n)]

        x = self.down_convs[0](x, ctx0)
        xs, perms, nmaps = [x], [], []
        for i in range(self.depth):
            if trace is not None:
                trace[f"pre.down{i}"] = x.detach()
            xr = self._relu(x, dec.get(f"relu.down{i}"))
            x = self.down_convs[i + 1](xr, level(i, x.size(0)))
            xs.append(x)
            if trace is not None:
                trace[f"relu.down{i}"] = xr
            x, ei, ea, perm, nmap = self.pools[i](x, eis[-1], eas[-1], batch, return_node_map=True,
                                                  relu_decisions=dec.get(f"relu.pool{i}"), perm_decision=dec.get(f"perm{i}"),
                                                  trace=trace, trace_tag=str(i))
            eis.append(ei); eas.append(ea); perms.append(perm); nmaps.append(nmap)
            sizes[i + 1] = x.size(0)
            if trace is not None:
                trace[f"relu.pool{i}"] = F.relu(trace[f"pre.pool{i}"])
                trace[f"perm{i}"] = perm
        if trace is not None:
            trace["pre.bottom"] = x.detach()
        xr = self._relu(x, dec.get("relu.bottom"))
        if trace is not None:
            trace["relu.bottom"] = xr
        x = self.bottom_conv(xr, level(self.depth, x.size(0)))
        if trace is not None:
            trace["unet.bottom"] = x
            for k, t in enumerate(xs):
                trace[f"unet.xs{k}"] = t
        for i in range(self.depth):
            j = self.depth - 1 - i
            if trace is not None:   # the fused kernel below never materialises its pre-activation: rebuilt here for the margin check
                trace[f"pre.up{i}"] = (torch.zeros_like(xs[j + 1]).index_copy(0, perms[j], x.detach()) + xs[j + 1].detach())
            if self.act is F.relu:
                x = ops.unpool_add_relu(x, xs[j + 1], nmaps[j], decide=dec.get(f"relu.up{i}"))  # K9: gather by node_map, no zero fill
            elif dec.get(f"relu.up{i}") is not None:
                raise NotImplementedError("decision injection is wired into the K9 kernels only")
            else:
                up = torch.zeros(xs[j + 1].size(0), x.size(1), device=x.device, dtype=x.dtype).index_copy(0, perms[j], x)
                x = self.act(up + xs[j + 1])
            if trace is not None:
                trace[f"unet.up{i}.in"] = x
                trace[f"relu.up{i}"] = x
            lvl = j + 1 if self.strict_reference else j
            x = self.up_convs[i](x, level(lvl, x.size(0)))
            if trace is not None:
                trace[f"unet.up{i}.out"] = x
        return ops.lin(self.final_conv, x)
