"""TEST INFRASTRUCTURE ONLY -- CPU oracle (float half) for the DGDM hot path.

Nothing in the shipped package may import this module: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the
checker ("Oracle").

This is a from-scratch CPU *restatement* (plain torch on the CPU, no torch-geometric, no
nn.Module tree) of the algorithm the reference runs on the path
``DGDMModel.pretrain_step -> forward`` (reference = /root/reference/dgdm_histopath, read as
text).  It is functional: every function takes ``P`` -- a flat ``{name: tensor}`` dict that
uses the reference's own ``state_dict`` key names -- so the same weights can be fed to the
reference classes (golden capture), to this oracle and to the HIP product.

Parity pin: pinned against the reference classes executed in the dev
container -- ``core/attention.py`` and ``core/diffusion.py`` as-is, ``core/graph_layers.py``
/ ``models/encoders.py`` through the build-owned torch-geometric stand-in
(``oracle/pyg_standin.py``; PyG itself is not installed, so the third-party scatter
arithmetic is pinned by the stand-in's documented semantics + hand-computed cases) -- via
the fixtures in ``tests/golden`` written by ``oracle/capture_golden.py``.

Repairs the oracle freezes (SURVEY.md section 8(a'), the reference path does not run as
written): R1 self-loop edges carry a zero edge-attribute row; R2 ``dim_proj`` Linear
between a dim-changing graph layer and its LayerNorm; R3 diffusion layer is called on 2-D
``[N_g, C]``; R4 ``batch=None`` is the all-zero batch vector; R5a U-Net layers use the
data's edge dim (32); R5b ``up_convs`` take ``hidden`` inputs (sum skip).  D8 (loss target
is fresh noise) and D10 (decoder uses the coarser level's edge list) are replicated under
``strict_reference=True``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import csr_oracle

Tensor = torch.Tensor
EDGE_DIM = 32  # models/encoders.py:183 (hard-coded)


# --------------------------------------------------------------------------------------
# configuration / parameter construction
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    """Mirror of the DGDMModel constructor arguments (models/dgdm_model.py:45-61)."""
    node_features: int = 768
    hidden_dims: List[int] = field(default_factory=lambda: [512, 256, 128])
    num_diffusion_steps: int = 10
    attention_heads: int = 8
    dropout: float = 0.1
    graph_layers: int = 4
    use_spatial_attention: bool = True
    use_hierarchical: bool = True
    diffusion_schedule: str = "cosine"
    pooling: str = "attention"
    strict_reference: bool = True
    unet_depth: int = 3  # models/dgdm_model.py:154
    activation: str = "gelu"              # FeatureEncoder / GraphEncoder / heads: "relu" | "gelu" | "elu" (models/encoders.py:57-62,202-209)
    normalization: str = "layer"          # FeatureEncoder / GraphEncoder norms: "layer" | "batch" | "instance" (models/encoders.py:95-100,211-219)
    num_classes: Optional[int] = None     # models/dgdm_model.py:168-175 (heads)
    regression_targets: int = 0           # models/dgdm_model.py:177-184

    def encoder_dims(self):
        dims = [self.hidden_dims[0]] + list(self.hidden_dims)  # encoders.py:173
        out = []
        for i in range(self.graph_layers):
            out.append((dims[i], dims[min(i + 1, len(dims) - 1)]))  # encoders.py:176-177
        return out


def _dyn_layer_shapes(prefix, node_dim, hidden, edge_dim=EDGE_DIM):
    """Parameter shapes of one DynamicGraphLayer (core/graph_layers.py:138-152)."""
    return {
        f"{prefix}.node_to_qkv.weight": (3 * hidden, node_dim), f"{prefix}.node_to_qkv.bias": (3 * hidden,),
        f"{prefix}.edge_to_key.weight": (hidden, edge_dim), f"{prefix}.edge_to_key.bias": (hidden,),
        f"{prefix}.graph_conv1.node_lin.weight": (hidden, node_dim),
        f"{prefix}.graph_conv1.edge_lin.weight": (hidden, edge_dim),
        f"{prefix}.graph_conv1.bias": (hidden,),
        f"{prefix}.graph_conv2.node_lin.weight": (hidden, hidden),
        f"{prefix}.graph_conv2.edge_lin.weight": (hidden, edge_dim),
        f"{prefix}.graph_conv2.bias": (hidden,),
        f"{prefix}.output_proj.weight": (node_dim, hidden), f"{prefix}.output_proj.bias": (node_dim,),
        f"{prefix}.norm1.weight": (node_dim,), f"{prefix}.norm1.bias": (node_dim,),
        f"{prefix}.norm2.weight": (node_dim,), f"{prefix}.norm2.bias": (node_dim,),
    }


def param_shapes(cfg: OracleConfig) -> Dict[str, tuple]:
    """All parameters of DGDMModel (+R2 ``dim_proj``), keyed like the reference state_dict."""
    F0, H = cfg.node_features, cfg.hidden_dims
    C = H[-1]
    s: Dict[str, tuple] = {}
    # FeatureEncoder (models/encoders.py:73-91)
    affine = cfg.normalization in ("layer", "batch")     # nn.InstanceNorm1d(dim) has no parameters (affine=False), nn.Identity none
    s["feature_encoder.encoder.0.weight"] = (H[0], F0); s["feature_encoder.encoder.0.bias"] = (H[0],)
    s["feature_encoder.encoder.4.weight"] = (H[0], H[0]); s["feature_encoder.encoder.4.bias"] = (H[0],)
    if affine:
        s["feature_encoder.encoder.1.weight"] = (H[0],); s["feature_encoder.encoder.1.bias"] = (H[0],)
        s["feature_encoder.encoder.5.weight"] = (H[0],); s["feature_encoder.encoder.5.bias"] = (H[0],)
    if F0 != H[0]:
        s["feature_encoder.residual_proj.weight"] = (H[0], F0); s["feature_encoder.residual_proj.bias"] = (H[0],)
    # GraphEncoder (models/encoders.py:173-215) + R2
    for i, (din, dout) in enumerate(cfg.encoder_dims()):
        s.update(_dyn_layer_shapes(f"graph_encoder.graph_layers.{i}", din, dout))
        if affine:
            s[f"graph_encoder.norm_layers.{i}.weight"] = (dout,); s[f"graph_encoder.norm_layers.{i}.bias"] = (dout,)
        if din != dout:
            s[f"graph_encoder.dim_proj.{i}.weight"] = (dout, din); s[f"graph_encoder.dim_proj.{i}.bias"] = (dout,)
    s["graph_encoder.output_proj.weight"] = (C, C); s["graph_encoder.output_proj.bias"] = (C,)
    # DiffusionLayer (core/diffusion.py:87-104), hidden = 2*C (dgdm_model.py:133)
    Hd = 2 * C
    s["diffusion_layer.time_embed.0.weight"] = (Hd, 128); s["diffusion_layer.time_embed.0.bias"] = (Hd,)
    s["diffusion_layer.time_embed.2.weight"] = (Hd, Hd); s["diffusion_layer.time_embed.2.bias"] = (Hd,)
    s["diffusion_layer.denoise_net.0.weight"] = (2 * Hd, C + Hd); s["diffusion_layer.denoise_net.0.bias"] = (2 * Hd,)
    s["diffusion_layer.denoise_net.1.weight"] = (2 * Hd,); s["diffusion_layer.denoise_net.1.bias"] = (2 * Hd,)
    s["diffusion_layer.denoise_net.4.weight"] = (Hd, 2 * Hd); s["diffusion_layer.denoise_net.4.bias"] = (Hd,)
    s["diffusion_layer.denoise_net.5.weight"] = (Hd,); s["diffusion_layer.denoise_net.5.bias"] = (Hd,)
    s["diffusion_layer.denoise_net.8.weight"] = (C, Hd); s["diffusion_layer.denoise_net.8.bias"] = (C,)
    # SpatialAttention (core/attention.py:205-223)
    if cfg.use_spatial_attention:
        p = "spatial_attention"
        s[f"{p}.pos_encoding"] = (10000, C)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[f"{p}.attention.{n}.weight"] = (C, C); s[f"{p}.attention.{n}.bias"] = (C,)
        s[f"{p}.spatial_proj.0.weight"] = (C // 2, 2); s[f"{p}.spatial_proj.0.bias"] = (C // 2,)
        s[f"{p}.spatial_proj.2.weight"] = (C, C // 2); s[f"{p}.spatial_proj.2.bias"] = (C,)
        s[f"{p}.norm.weight"] = (C,); s[f"{p}.norm.bias"] = (C,)
    # GraphUNet (core/graph_layers.py:366-398) with R5a/R5b
    if cfg.use_hierarchical:
        p = "hierarchical_processor"
        for i in range(cfg.unet_depth + 1):
            s.update(_dyn_layer_shapes(f"{p}.down_convs.{i}", C, C))
        for i in range(cfg.unet_depth):
            s[f"{p}.pools.{i}.score_net.0.weight"] = (C // 2, C); s[f"{p}.pools.{i}.score_net.0.bias"] = (C // 2,)
            s[f"{p}.pools.{i}.score_net.2.weight"] = (1, C // 2); s[f"{p}.pools.{i}.score_net.2.bias"] = (1,)
            s.update(_dyn_layer_shapes(f"{p}.up_convs.{i}", C, C))
        s.update(_dyn_layer_shapes(f"{p}.bottom_conv", C, C))
        s[f"{p}.final_conv.weight"] = (C, C); s[f"{p}.final_conv.bias"] = (C,)
    # GlobalAttentionPool (models/dgdm_model.py:591-594)
    if cfg.pooling == "attention":
        s["global_pool.global_token"] = (1, 1, C)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[f"global_pool.attention.{n}.weight"] = (C, C); s[f"global_pool.attention.{n}.bias"] = (C,)
    # task heads (models/decoders.py:54-75,213-231 as DGDMModel builds them, dgdm_model.py:168-184: hidden_dims=[C // 2],
    # BatchNorm1d after the hidden Linear).  Sequential indices: 0 Linear, 1 BatchNorm1d, 2 act, 3 Dropout, 4 Linear.
    if cfg.num_classes is not None:
        p = "classification_head.classifier"
        s[f"{p}.0.weight"] = (C // 2, C); s[f"{p}.0.bias"] = (C // 2,)
        s[f"{p}.1.weight"] = (C // 2,); s[f"{p}.1.bias"] = (C // 2,)
        s[f"{p}.4.weight"] = (cfg.num_classes, C // 2); s[f"{p}.4.bias"] = (cfg.num_classes,)
    if cfg.regression_targets > 0:
        p = "regression_head"
        s[f"{p}.feature_layers.0.weight"] = (C // 2, C); s[f"{p}.feature_layers.0.bias"] = (C // 2,)
        s[f"{p}.feature_layers.1.weight"] = (C // 2,); s[f"{p}.feature_layers.1.bias"] = (C // 2,)
        s[f"{p}.mean_head.weight"] = (cfg.regression_targets, C // 2); s[f"{p}.mean_head.bias"] = (cfg.regression_targets,)
    return s


def batchnorm_buffers(cfg: OracleConfig, seed: int = 0, trained: bool = False) -> Dict[str, Tensor]:
    """running_mean / running_var / num_batches_tracked of the heads' BatchNorm1d layers (state_dict buffers).  Fresh modules
    hold (0, 1, 0); ``trained=True`` draws non-trivial statistics so that eval-mode parity exercises them."""
    C2 = cfg.hidden_dims[-1] // 2
    out: Dict[str, Tensor] = {}
    pres = (["classification_head.classifier.1"] if cfg.num_classes is not None else []) + \
           (["regression_head.feature_layers.1"] if cfg.regression_targets > 0 else [])
    for i, pre in enumerate(pres):
        g = torch.Generator().manual_seed(seed * 7919 + 31 + i)
        out[f"{pre}.running_mean"] = 0.3 * torch.randn(C2, generator=g) if trained else torch.zeros(C2)
        out[f"{pre}.running_var"] = 0.5 + torch.rand(C2, generator=g) if trained else torch.ones(C2)
        out[f"{pre}.num_batches_tracked"] = torch.tensor(7 if trained else 0)
    return out


def init_params(cfg: OracleConfig, seed: int = 0, dtype=torch.float32, perturb: float = 0.0) -> Dict[str, Tensor]:
    """Seed-reproducible parameters following the reference init *rules*
    (dgdm_model.py:259-269: xavier-uniform Linear weights, zero biases, norms (1,0);
    global_token ~ N(0,1) dgdm_model.py:594; pos_encoding ~ 0.02 N(0,1) attention.py:211).

    Each tensor is drawn from its own generator seeded by (seed, key index) so the result
    does not depend on construction order.  ``perturb`` > 0 adds N(0, perturb^2) noise to
    biases / norm affine terms so that tests also exercise non-trivial values there.
    """
    shapes = param_shapes(cfg)
    P: Dict[str, Tensor] = {}
    for idx, name in enumerate(sorted(shapes)):
        shp = shapes[name]
        g = torch.Generator().manual_seed(seed * 100003 + idx)
        if name.endswith("global_token"):
            t = torch.randn(shp, generator=g, dtype=torch.float64)
        elif name.endswith("pos_encoding"):
            t = torch.randn(shp, generator=g, dtype=torch.float64) * 0.02
        elif len(shp) == 2:  # Linear weight [out, in]
            bound = math.sqrt(6.0 / (shp[0] + shp[1]))
            t = (torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * bound
        else:
            is_norm_w = name.endswith(".weight")  # 1-D weight = norm gain
            t = torch.ones(shp, dtype=torch.float64) if is_norm_w else torch.zeros(shp, dtype=torch.float64)
            if perturb > 0:
                t = t + perturb * torch.randn(shp, generator=g, dtype=torch.float64)
        P[name] = t.to(dtype)
    return P


# --------------------------------------------------------------------------------------
# graph structure
# --------------------------------------------------------------------------------------
class OracleGraph:
    """Loop-extended COO + weights of one (possibly batched) graph, torch tensors on CPU."""

    def __init__(self, edge_index: Tensor, num_nodes: int, dtype=torch.float32, add_loops: bool = True, normalize: bool = True):
        """``normalize=False``: GraphConvolution(normalize=False) (core/graph_layers.py:76-86) -- no self loops (they are added inside
        the ``if self.normalize`` branch) and no norm: every entry weighs 1."""
        ei = edge_index.detach().cpu().numpy()
        g = csr_oracle.gcn_csr(ei, num_nodes, add_loops=add_loops and normalize)
        self.num_nodes = num_nodes
        self.num_input_edges = ei.shape[1]
        self.src = torch.from_numpy(g["src"])
        self.dst = torch.from_numpy(g["dst"])
        self.norm = torch.from_numpy(g["norm_coo"]).to(dtype) if normalize else torch.ones(len(g["src"]), dtype=dtype)
        self.csr = g


def _ext_edge_attr(edge_attr: Optional[Tensor], graph: OracleGraph, dtype) -> Tensor:
    """R1: the N appended self-loop edges carry a zero attribute row."""
    E, N = graph.num_input_edges, graph.num_nodes
    if edge_attr is None:  # encoders.py:258-261 -> zeros(E, 32)
        return torch.zeros(E + N, EDGE_DIM, dtype=dtype)
    return torch.cat([edge_attr.to(dtype), torch.zeros(N, edge_attr.shape[1], dtype=dtype)], dim=0)


# --------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------
# Test infrastructure for training-mode parity: when set, every dropout site asks the hook for its multiplier (0 or
# 1/(1-p) per element) instead of drawing one, so that a checker can hand this restatement the very masks a device kernel
# generated.  hook(site, x, p, graph) -> tensor like x; `site` names the call site, `graph` is the index of the graph the
# per-graph loops (attention, diffusion loss, pooling) are working on, None where the site sees the whole batch.
DROPOUT_HOOK = None
_DROP_GRAPH = None
# Optional injection of the graph U-Net's discrete decisions (tests / tools only; None = decide here, as the reference does): a dict
# with the keys of conftest.decisions_from_trace -- ``relu.down{i}`` / ``relu.pool{i}`` / ``relu.bottom`` / ``relu.up{i}`` (bool
# masks: which elements pass the ReLU) and ``perm{i}`` (kept node ids).  Lets a float32 run of this restatement differentiate the
# SAME piecewise-linear function as the float64 run (tools/arithmetic_error_report.py compares their gradients).
DECISIONS = None


def _relu(x: Tensor, key: str) -> Tensor:
    if DECISIONS is not None and key in DECISIONS:
        return torch.where(DECISIONS[key].reshape(x.shape), x, torch.zeros_like(x))
    return F.relu(x)


def _drop(x: Tensor, p: float, training: bool, site: Optional[str] = None) -> Tensor:
    if not (training and p > 0):
        return x
    if DROPOUT_HOOK is not None:
        return x * DROPOUT_HOOK(site, x, p, _DROP_GRAPH)
    return F.dropout(x, p, True)


def _ln(P, pre, x):
    return F.layer_norm(x, (x.shape[-1],), P[f"{pre}.weight"], P[f"{pre}.bias"], 1e-5)


def _enc_norm(P, pre, x, kind: str, training: bool):
    """The norm modules models/encoders.py:95-100 / :211-219 build, applied to a 2-D ``[N, C]`` tensor:
    "layer"    nn.LayerNorm(C);
    "batch"    nn.BatchNorm1d(C): statistics over the N NODES of the batch in training mode (biased variance), the running
               averages in eval mode -- this functional restatement keeps no running state: eval mode reads ``{pre}.running_mean`` /
               ``.running_var`` from P when present, else the module's initial values (0, 1);
    "instance" nn.InstanceNorm1d(C) with its defaults (affine=False, track_running_stats=False): a 2-D input is taken as ONE
               unbatched sample (channels = rows, length = C), i.e. every row is normalised over its C entries (biased variance,
               eps 1e-5) -- LayerNorm without affine parameters;
    anything else nn.Identity()."""
    if kind == "layer":
        return _ln(P, pre, x)
    if kind == "instance":
        return F.layer_norm(x, (x.shape[-1],), None, None, 1e-5)
    if kind == "batch":
        rm = P.get(f"{pre}.running_mean", torch.zeros(x.shape[-1], dtype=x.dtype)).detach()
        rv = P.get(f"{pre}.running_var", torch.ones(x.shape[-1], dtype=x.dtype)).detach()
        return F.batch_norm(x, rm.clone(), rv.clone(), P[f"{pre}.weight"], P[f"{pre}.bias"], training, 0.1, 1e-5)
    return x


def _enc_act(name: str):
    return {"relu": F.relu, "gelu": F.gelu, "elu": F.elu}[name]      # encoders.py:57-64 (FeatureEncoder raises on anything else)


def _lin(P, pre, x):
    return F.linear(x, P[f"{pre}.weight"], P.get(f"{pre}.bias"))


def graph_conv(P, pre: str, x: Tensor, graph: OracleGraph, ea_ext: Optional[Tensor]) -> Tensor:
    """GraphConvolution.forward/message (core/graph_layers.py:68-110), R1.

    out[d] = sum_{e: dst_e = d} norm_e * ((x W^T)[src_e] + W_e a_e) + b, edges in
    ascending edge-id order inside each destination (index_add_ on CPU is sequential).
    """
    h = F.linear(x, P[f"{pre}.node_lin.weight"])
    msg = h[graph.src]
    if ea_ext is not None and f"{pre}.edge_lin.weight" in P:
        msg = msg + F.linear(ea_ext, P[f"{pre}.edge_lin.weight"])
    msg = graph.norm.view(-1, 1) * msg
    out = torch.zeros(graph.num_nodes, h.shape[1], dtype=h.dtype).index_add_(0, graph.dst, msg)
    if f"{pre}.bias" in P:
        out = out + P[f"{pre}.bias"]
    return out


def dynamic_graph_layer(P, pre, x, graph, ea_ext, p_drop=0.0, training=False) -> Tensor:
    """DynamicGraphLayer.forward (core/graph_layers.py:207-247).

    ``compute_dynamic_edges`` (:160-205, :227-230) is evaluated and discarded by the
    reference -- it influences neither outputs nor gradients, so it is not restated.
    """
    h = F.gelu(graph_conv(P, f"{pre}.graph_conv1", x, graph, ea_ext))
    h = _drop(h, p_drop, training, f"{pre}.drop1")
    h = F.gelu(graph_conv(P, f"{pre}.graph_conv2", h, graph, ea_ext))
    h = _drop(h, p_drop, training, f"{pre}.drop2")
    out = _lin(P, f"{pre}.output_proj", h)
    return _ln(P, f"{pre}.norm1", out + x)


def feature_encoder(P, x, p_drop=0.0, training=False, activation: str = "gelu", normalization: str = "layer") -> Tensor:
    """FeatureEncoder.forward (models/encoders.py:104-124); activation / normalization: its constructor arguments (:57-64, :95-100)."""
    pre = "feature_encoder"
    act = _enc_act(activation)
    h = _drop(act(_enc_norm(P, f"{pre}.encoder.1", _lin(P, f"{pre}.encoder.0", x), normalization, training)), p_drop, training, f"{pre}.encoder.3")
    h = _drop(act(_enc_norm(P, f"{pre}.encoder.5", _lin(P, f"{pre}.encoder.4", h), normalization, training)), p_drop, training, f"{pre}.encoder.7")
    res = _lin(P, f"{pre}.residual_proj", x) if f"{pre}.residual_proj.weight" in P else x
    return h + res


def graph_encoder(P, cfg: OracleConfig, x, graph, ea_ext, training=False):
    """GraphEncoder.forward (models/encoders.py:228-280) with R2."""
    outs = []
    h = x
    for i, (din, dout) in enumerate(cfg.encoder_dims()):
        h = dynamic_graph_layer(P, f"graph_encoder.graph_layers.{i}", h, graph, ea_ext, cfg.dropout, training)
        if din != dout:
            h = _lin(P, f"graph_encoder.dim_proj.{i}", h)
        act = _enc_act(cfg.activation) if cfg.activation in ("relu", "gelu", "elu") else F.relu      # encoders.py:202-209
        h = _drop(act(_enc_norm(P, f"graph_encoder.norm_layers.{i}", h, cfg.normalization, training)), cfg.dropout, training,
                  f"graph_encoder.dropout.{i}")
        outs.append(h)
    return _lin(P, "graph_encoder.output_proj", h), outs


def plain_conv_encoder(P, dims, x, graph):
    """As-is ``GraphEncoder(use_edge_features=False)`` (encoders.py:188-193,262-271):
    a stack of GraphConvolution(in->out) + LayerNorm + GELU, then Linear."""
    outs, h = [], x
    for i in range(len(dims)):
        h = graph_conv(P, f"graph_layers.{i}", h, graph, None)
        h = F.gelu(_ln(P, f"norm_layers.{i}", h))
        outs.append(h)
    return _lin(P, "output_proj", h), outs


def sinusoid_pos_encoding(pos: Tensor, C: int) -> Tensor:
    """SpatialAttention.get_positional_encoding (core/attention.py:225-259) for one graph:
    one global min/max over both coordinates, C/4 frequencies, [sin x, cos x, sin y, cos y]
    interleaved with stride 4."""
    pn = pos.to(torch.float32) if pos.dtype not in (torch.float32, torch.float64) else pos
    if pn.numel() > 0:
        pn = (pn - pn.min()) / (pn.max() - pn.min() + 1e-8)
    div = torch.exp(torch.arange(0, C // 2, 2, dtype=pn.dtype) * -(math.log(10000.0) / (C // 2)))
    pe = torch.zeros(pos.shape[0], C, dtype=pn.dtype)
    pe[:, 0::4] = torch.sin(pn[:, 0:1] * div)
    pe[:, 1::4] = torch.cos(pn[:, 0:1] * div)
    pe[:, 2::4] = torch.sin(pn[:, 1:2] * div)
    pe[:, 3::4] = torch.cos(pn[:, 1:2] * div)
    return pe


def mha(P, pre, query, key, value, H, bias=None, p_drop=0.0, training=False):
    """MultiHeadAttention.forward (core/attention.py:73-181) for one un-batched sequence.
    query [Lq,C], key/value [Lk,C], bias [Lq,Lk] float (added after the 1/sqrt(d) scale).
    Returns (out [Lq,C], head-mean post-dropout weights [Lq,Lk])."""
    Lq, C = query.shape
    d = C // H
    q = _lin(P, f"{pre}.q_proj", query).view(Lq, H, d).transpose(0, 1)
    k = _lin(P, f"{pre}.k_proj", key).view(-1, H, d).transpose(0, 1)
    v = _lin(P, f"{pre}.v_proj", value).view(-1, H, d).transpose(0, 1)
    s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(d)
    if bias is not None:
        s = s + bias
    w = _drop(F.softmax(s, dim=-1), p_drop, training, f"{pre}.attn_dropout")
    o = torch.matmul(w, v).transpose(0, 1).reshape(Lq, C)
    o = _drop(_lin(P, f"{pre}.out_proj", o), p_drop, training, f"{pre}.resid_dropout")
    return o, w.mean(dim=0)


def mha_dense(P, pre, query, key=None, value=None, H=8, attn_mask=None, key_padding_mask=None, add_zero_attn=False, drop_mask=None):
    """MultiHeadAttention.forward for a BATCH with every argument of the reference (core/attention.py:73-181): query [B, L, C],
    key / value [B, S, C] (default: query / key), ``attn_mask`` float (added after the 1/sqrt(d) scale, :134-135) or bool (-inf where
    True, :132-133) broadcast against the [B, H, L, S] scores exactly as the reference's in-place ops do, ``key_padding_mask``
    [B, S] bool (:137-142), ``add_zero_attn`` (:118-126).  ``drop_mask`` [B, H, L, S]: keep-scale factors standing in for
    ``attn_dropout`` (:146; dropout cannot be matched draw for draw, the kernels' own mask is handed in).
    Returns (out [B, L, C] BEFORE resid_dropout, per-head weights after dropout [B, H, L, S])."""
    B, L, C = query.shape
    key = query if key is None else key
    value = key if value is None else value
    d = C // H
    q = _lin(P, f"{pre}.q_proj", query).view(B, L, H, d).transpose(1, 2)
    k = _lin(P, f"{pre}.k_proj", key).view(B, -1, H, d).transpose(1, 2)
    v = _lin(P, f"{pre}.v_proj", value).view(B, -1, H, d).transpose(1, 2)
    if add_zero_attn:
        z = torch.zeros(B, H, 1, d, dtype=k.dtype)
        k, v = torch.cat([k, z], dim=2), torch.cat([v, z], dim=2)
        if attn_mask is not None:
            attn_mask = F.pad(attn_mask, (0, 1))
        if key_padding_mask is not None:
            key_padding_mask = F.pad(key_padding_mask, (0, 1))
    s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(d)
    if attn_mask is not None:
        full = attn_mask.expand(s.shape)          # what masked_fill_ / += broadcast to (in place: the mask cannot enlarge the scores)
        s = s.masked_fill(full, float("-inf")) if attn_mask.dtype == torch.bool else s + full.to(s.dtype)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask.unsqueeze(1).unsqueeze(2), float("-inf"))
    w = F.softmax(s, dim=-1)
    if drop_mask is not None:
        w = w * drop_mask
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, L, C)
    return _lin(P, f"{pre}.out_proj", o), w


def spatial_attention_dense(P, x, pos, H, mask=None, temperature=1.0, pre="spatial_attention"):
    """SpatialAttention.forward (core/attention.py:285-327) for x [B, N, C], pos [B, N, 2] WITH the optional ``mask`` (:311-314:
    ``attn_mask = mask + spatial_bias``; runs in the reference for B = 1 only, see the note in the product's module).  Positional
    encodings normalised per sequence (B = 1: the reference's global min / max)."""
    B, N, C = x.shape
    pe = torch.stack([sinusoid_pos_encoding(pos[b], C).to(x.dtype) for b in range(B)])
    p = pos.to(x.dtype)
    bias = -torch.norm(p.unsqueeze(2) - p.unsqueeze(1), dim=-1) / temperature       # [B, N, N]  (:274-281)
    am = bias if mask is None else mask.to(x.dtype) + bias
    if B > 1 and am.dim() == 3:
        am = am.unsqueeze(1)                                                          # the product's B > 1 extension: one bias per sequence
    o, w = mha_dense(P, f"{pre}.attention", x + pe, H=H, attn_mask=am)
    return _ln(P, f"{pre}.norm", x + o), w.mean(dim=1)


def spatial_attention_graph(P, x, pos, H, temperature=1.0, p_drop=0.0, training=False, pre="spatial_attention"):
    """SpatialAttention.forward (core/attention.py:285-327) on one graph."""
    C = x.shape[1]
    pe = sinusoid_pos_encoding(pos, C).to(x.dtype)
    p = pos.to(x.dtype)
    bias = -torch.norm(p.unsqueeze(1) - p.unsqueeze(0), dim=-1) / temperature  # :274-281, raw positions
    xp = x + pe
    o, w = mha(P, f"{pre}.attention", xp, xp, xp, H, bias, p_drop, training)
    return _ln(P, f"{pre}.norm", x + o), w


def adaptive_pool(P, pre, x, edge_index, edge_attr, ratio=0.5, level: Optional[int] = None):
    """AdaptiveGraphPooling.forward (core/graph_layers.py:285-329); top-k over ALL nodes."""
    s = _lin(P, f"{pre}.score_net.2", _relu(_lin(P, f"{pre}.score_net.0", x), f"relu.pool{level}")).squeeze(-1)
    s = torch.tanh(s)
    forced = None if DECISIONS is None else DECISIONS.get(f"perm{level}")
    idx = csr_oracle.topk_pool_indices(s.detach().cpu().numpy(), edge_index.cpu().numpy(), ratio,
                                       **({} if forced is None else {"perm": forced.cpu().numpy()}))
    perm = torch.from_numpy(idx["perm"])
    keep = torch.from_numpy(idx["edge_keep"])
    px = x[perm] * s[perm].unsqueeze(-1)
    pea = edge_attr[keep] if edge_attr is not None else None
    return px, torch.from_numpy(idx["edge_index"]), pea, perm, s


def graph_unet(P, cfg: OracleConfig, x, edge_index, edge_attr, training=False, pre="hierarchical_processor",
               trace: Optional[dict] = None):
    """GraphUNet.forward (core/graph_layers.py:400-458), R5a/R5b, D10 under strict."""
    depth, p_drop = cfg.unet_depth, 0.1  # DynamicGraphLayer default dropout (graph_layers.py:125)
    dtype = x.dtype
    eis, eas = [edge_index], [edge_attr]

    def level(k, n):
        """Structure of edge list k applied to n nodes (+ R1-extended attributes)."""
        gk = OracleGraph(eis[k], n, dtype)
        return gk, _ext_edge_attr(eas[k], gk, dtype)

    g, e = level(0, x.shape[0])
    x = dynamic_graph_layer(P, f"{pre}.down_convs.0", x, g, e, p_drop, training)
    xs, perms = [x], []
    for i in range(depth):
        g, e = level(i, x.shape[0])  # graph_layers.py:420: edge_indices[-1] == level i here
        xr = _relu(x, f"relu.down{i}")
        if trace is not None: trace[f"relu.down{i}"] = xr
        x = dynamic_graph_layer(P, f"{pre}.down_convs.{i+1}", xr, g, e, p_drop, training)
        xs.append(x)
        if trace is not None:
            trace[f"relu.pool{i}"] = F.relu(_lin(P, f"{pre}.pools.{i}.score_net.0", x))
        x, ei2, ea2, perm, score = adaptive_pool(P, f"{pre}.pools.{i}", x, eis[-1], eas[-1], level=i)
        eis.append(ei2); eas.append(ea2); perms.append(perm)
        if trace is not None:
            trace[f"perm{i}"] = perm; trace[f"score{i}"] = score; trace[f"edge_index{i+1}"] = ei2
    g, e = level(depth, x.shape[0])
    xr = _relu(x, "relu.bottom")
    if trace is not None: trace["relu.bottom"] = xr
    x = dynamic_graph_layer(P, f"{pre}.bottom_conv", xr, g, e, p_drop, training)
    if trace is not None:
        trace["unet.bottom"] = x
        for k, t in enumerate(xs): trace[f"unet.xs{k}"] = t
    for i in range(depth):
        j = depth - 1 - i
        up = torch.zeros(xs[j + 1].shape[0], x.shape[1], dtype=dtype).index_copy(0, perms[j], x)
        x = _relu(up + xs[j + 1], f"relu.up{i}")
        if trace is not None: trace[f"unet.up{i}.in"] = x; trace[f"relu.up{i}"] = x
        lvl = j + 1 if cfg.strict_reference else j  # D10: graph_layers.py:453 uses edge_indices[j+1]
        gg, ee = level(lvl, x.shape[0])  # level-lvl edge list applied to x.shape[0] nodes
        x = dynamic_graph_layer(P, f"{pre}.up_convs.{i}", x, gg, ee, p_drop, training)
        if trace is not None: trace[f"unet.up{i}.out"] = x
    return _lin(P, f"{pre}.final_conv", x)


# -- diffusion -------------------------------------------------------------------------
def diffusion_schedule(T: int, schedule: str = "cosine", beta_start=1e-4, beta_end=0.02) -> Dict[str, Tensor]:
    """DiffusionScheduler (core/diffusion.py:16-61); float32 arithmetic like the reference."""
    if schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, T)
    elif schedule == "cosine":
        s = 0.008
        x = torch.linspace(0, T, T + 1)
        ac = torch.cos(((x / T) + s) / (1 + s) * math.pi * 0.5) ** 2
        ac = ac / ac[0]
        betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
    elif schedule == "sigmoid":
        # :56-61 -- called as (timesteps, beta_start, beta_end) => start=1e-4, end=0.02
        betas = torch.sigmoid(torch.linspace(-6, 6, T)) * (beta_end - beta_start) + beta_start
    else:
        raise ValueError(f"Unknown schedule: {schedule}")
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = torch.cat([torch.ones(1), ac[:-1]])
    post = betas * (1.0 - ac_prev) / (1.0 - ac)
    return dict(betas=betas, alphas=alphas, alphas_cumprod=ac, alphas_cumprod_prev=ac_prev, posterior_variance=post)


def timestep_embedding(t: Tensor, dim: int = 128) -> Tensor:
    """DiffusionLayer.get_timestep_embedding (core/diffusion.py:112-121)."""
    half = dim // 2
    f = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    e = t.float()[:, None] * f[None, :]
    return torch.cat([torch.sin(e), torch.cos(e)], dim=1)


def predict_noise(P, x_noisy: Tensor, t: Tensor, p_drop=0.1, training=False, pre="diffusion_layer", condition: Optional[Tensor] = None) -> Tensor:
    """DiffusionLayer.predict_noise (core/diffusion.py:147-172) on 2-D [N_g, C], t [1] (R3); ``condition`` [1, cd] or [N_g, cd]:
    t_emb + condition_net(condition) (:158-161), only with a ``condition_net`` in P (conditioning_dim given, :106-110)."""
    te = timestep_embedding(t).to(x_noisy.dtype)
    te = _lin(P, f"{pre}.time_embed.2", F.silu(_lin(P, f"{pre}.time_embed.0", te)))
    if condition is not None and f"{pre}.condition_net.weight" in P:
        te = te + _lin(P, f"{pre}.condition_net", condition)
    inp = torch.cat([x_noisy, te.expand(x_noisy.shape[0], -1)], dim=-1)
    h = _lin(P, f"{pre}.denoise_net.0", inp)
    h = F.group_norm(h, 8, P[f"{pre}.denoise_net.1.weight"], P[f"{pre}.denoise_net.1.bias"], 1e-5)
    h = _drop(F.silu(h), p_drop, training, f"{pre}.denoise_net.3")
    h = _lin(P, f"{pre}.denoise_net.4", h)
    h = F.group_norm(h, 8, P[f"{pre}.denoise_net.5.weight"], P[f"{pre}.denoise_net.5.bias"], 1e-5)
    h = _drop(F.silu(h), p_drop, training, f"{pre}.denoise_net.7")
    return _lin(P, f"{pre}.denoise_net.8", h)


def add_noise(sched, x0: Tensor, noise: Tensor, t: Tensor) -> Tensor:
    """DiffusionLayer.add_noise (core/diffusion.py:123-145); t indexes alphas_cumprod."""
    ac = sched["alphas_cumprod"][t]
    a, b = torch.sqrt(ac).to(x0.dtype), torch.sqrt(1.0 - ac).to(x0.dtype)
    while a.dim() < x0.dim():
        a, b = a.unsqueeze(-1), b.unsqueeze(-1)
    return a * x0 + b * noise


def ddpm_sample(P, sched, T: int, x_init: Tensor, noises: List[Tensor], num_inference_steps: int = 50, condition: Optional[Tensor] = None) -> Tensor:
    """DiffusionLayer.sample (core/diffusion.py:214-275) with the random draws injected:
    ``x_init`` replaces the initial randn, ``noises[i]`` the i-th per-step randn_like."""
    x = x_init
    ts = torch.linspace(T - 1, 0, num_inference_steps, dtype=torch.long)
    for i, t in enumerate(ts):
        eps = predict_noise(P, x, t.view(1), training=False, condition=condition)
        alpha, ac = sched["alphas"][t].to(x.dtype), sched["alphas_cumprod"][t].to(x.dtype)
        x0 = (x - torch.sqrt(1 - ac) * eps) / torch.sqrt(ac)
        if i < len(ts) - 1:
            var = sched["posterior_variance"][t].to(x.dtype)
            x = torch.sqrt(alpha) * x0 + torch.sqrt(var) * noises[i]
        else:
            x = x0
    return x


def attention_pool(P, x, ptr, H, p_drop=0.1, training=False, pre="global_pool"):
    """GlobalAttentionPool.forward (models/dgdm_model.py:596-615): one learned query per graph.
    MultiHeadAttention's own default dropout=0.1 applies (dgdm_model.py:593)."""
    out = []
    tok = P[f"{pre}.global_token"].view(1, -1)
    global _DROP_GRAPH
    for g in range(len(ptr) - 1):
        xg = x[ptr[g]:ptr[g + 1]]
        _DROP_GRAPH = g
        o, _ = mha(P, f"{pre}.attention", tok, xg, xg, H, None, p_drop, training)
        _DROP_GRAPH = None
        out.append(o)
    return torch.cat(out, dim=0)


def _head_act(name: str):
    return {"relu": F.relu, "gelu": F.gelu, "elu": F.elu}.get(name, F.relu)   # decoders.py:57-64: unknown names fall back to ReLU


def _head_features(P, lin: str, bn: str, x, act: str, p_drop: float, training: bool):
    """Linear -> BatchNorm1d -> act -> Dropout (decoders.py:69-78 / 228-237).  eval: running statistics; training: batch
    statistics (biased variance), as nn.BatchNorm1d."""
    h = _lin(P, lin, x)
    rm, rv = P.get(f"{bn}.running_mean"), P.get(f"{bn}.running_var")
    if rm is None:
        rm, rv = torch.zeros(h.shape[1], dtype=h.dtype), torch.ones(h.shape[1], dtype=h.dtype)
    h = F.batch_norm(h, rm.detach().to(h.dtype).clone(), rv.detach().to(h.dtype).clone(), P[f"{bn}.weight"], P[f"{bn}.bias"], training, 0.1, 1e-5)
    return _drop(_head_act(act)(h), p_drop, training, f"{lin}.dropout")


def classification_head(P, x, act="gelu", p_drop=0.1, training=False, pre="classification_head") -> Tensor:
    """ClassificationHead.forward (models/decoders.py:89-99)."""
    h = _head_features(P, f"{pre}.classifier.0", f"{pre}.classifier.1", x, act, p_drop, training)
    return _lin(P, f"{pre}.classifier.4", h)


def classification_loss(logits, targets, class_weights=None, label_smoothing: float = 0.0) -> Tensor:
    """ClassificationHead.compute_loss (models/decoders.py:101-128)."""
    if label_smoothing > 0:
        logp = F.log_softmax(logits, dim=-1)
        soft = torch.zeros_like(logp).scatter_(1, targets.unsqueeze(1), 1 - label_smoothing) + label_smoothing / logits.shape[1]
        return -(soft * logp).sum(dim=-1).mean()
    return F.cross_entropy(logits, targets, weight=class_weights)


def regression_head(P, x, act="gelu", p_drop=0.1, training=False, pre="regression_head") -> Tensor:
    """RegressionHead.forward (models/decoders.py:246-273), default construction (identity output activation, no variance head)."""
    h = _head_features(P, f"{pre}.feature_layers.0", f"{pre}.feature_layers.1", x, act, p_drop, training)
    return _lin(P, f"{pre}.mean_head", h)


def regression_loss(pred, targets, loss_type="mse") -> Tensor:
    """RegressionHead.compute_loss (models/decoders.py:275-315)."""
    return {"mse": F.mse_loss, "mae": F.l1_loss, "huber": F.huber_loss}[loss_type](pred, targets)


# --------------------------------------------------------------------------------------
# whole model
# --------------------------------------------------------------------------------------
def _ptr_from_batch(batch: Optional[Tensor], n: int) -> List[int]:
    if batch is None:  # R4
        return [0, n]
    counts = torch.bincount(batch)
    return [0] + torch.cumsum(counts, 0).tolist()


def forward(P, cfg: OracleConfig, data, mode: str = "inference", *, training: bool = False,
            timesteps: Optional[Tensor] = None, noise: Optional[Tensor] = None,
            noise_target: Optional[Tensor] = None, return_attention=False, return_embeddings=False,
            trace: Optional[dict] = None, stages: Optional[dict] = None) -> Dict[str, Tensor]:
    """DGDMModel.forward + _forward_continue + _compute_diffusion_loss
    (models/dgdm_model.py:271-445) with R1-R5.  ``timesteps`` [B], ``noise`` and
    ``noise_target`` [N_tot, C] are the injected random draws (pretrain mode).  ``stages``: dict that
    receives the forward wall time per stage in seconds (bench.py's CPU baseline split)."""
    import time
    _t0 = [time.perf_counter()]

    def _stage(name):
        if stages is not None:
            now = time.perf_counter()
            stages[name] = stages.get(name, 0.0) + now - _t0[0]
            _t0[0] = now
    x = data.x
    dtype = x.dtype
    n = x.shape[0]
    ptr = _ptr_from_batch(getattr(data, "batch", None), n)
    B = len(ptr) - 1
    graph = OracleGraph(data.edge_index, n, dtype)
    edge_attr = getattr(data, "edge_attr", None)
    ea_ext = _ext_edge_attr(edge_attr, graph, dtype)
    H = cfg.attention_heads
    out: Dict[str, Tensor] = {}

    _stage("graph_structure")
    h = feature_encoder(P, x, cfg.dropout, training, cfg.activation, cfg.normalization)
    _stage("feature_encoder")
    if trace is not None: trace["feature_encoder"] = h
    h, layer_outs = graph_encoder(P, cfg, h, graph, ea_ext, training)
    _stage("graph_encoder")
    if trace is not None:
        trace["graph_encoder"] = h
        for i, lo in enumerate(layer_outs): trace[f"graph_encoder.layer{i}"] = lo

    attn_w = None
    pos = getattr(data, "pos", None)
    if cfg.use_spatial_attention and pos is not None:
        outs, attn_w = [], []
        global _DROP_GRAPH
        for g in range(B):
            _DROP_GRAPH = g
            o, w = spatial_attention_graph(P, h[ptr[g]:ptr[g + 1]], pos[ptr[g]:ptr[g + 1]], H, 1.0, cfg.dropout, training)
            outs.append(o); attn_w.append(w)
        _DROP_GRAPH = None
        h = torch.cat(outs, dim=0)
        _stage("spatial_attention")
        if trace is not None: trace["spatial_attention"] = h

    if cfg.use_hierarchical:
        ea_for_unet = edge_attr if edge_attr is not None else None
        h = graph_unet(P, cfg, h, data.edge_index, ea_for_unet, training, trace=trace)
        _stage("graph_unet")
        if trace is not None: trace["graph_unet"] = h

    if mode == "pretrain":
        sched = diffusion_schedule(cfg.num_diffusion_steps, cfg.diffusion_schedule)
        if timesteps is None:
            timesteps = torch.randint(0, cfg.num_diffusion_steps, (B,))
        if noise is None:
            noise = torch.randn_like(h)
        if noise_target is None:
            noise_target = torch.randn_like(h)
        losses, noisy = [], None
        for g in range(B):
            sl = slice(ptr[g], ptr[g + 1])
            t = timesteps[g:g + 1]
            noisy = add_noise(sched, h[sl], noise[sl].to(dtype), t)
            _DROP_GRAPH = g
            pred = predict_noise(P, noisy, t, 0.1, training)
            _DROP_GRAPH = None
            target = noise_target[sl] if cfg.strict_reference else noise[sl]  # D8 (dgdm_model.py:429-430)
            losses.append(F.mse_loss(pred, target.to(dtype)))
        out["diffusion_loss"] = torch.stack(losses).mean()
        out["noisy_embeddings"] = noisy.unsqueeze(0)  # last graph's, [1,N_g,C] (dgdm_model.py:442-445)
        _stage("diffusion")

    if cfg.pooling == "attention":
        out["graph_embedding"] = attention_pool(P, h, ptr, H, 0.1, training)
    elif cfg.pooling in ("mean", "set2set"):  # dgdm_model.py:552-567, 627-642 (set2set == mean)
        out["graph_embedding"] = torch.stack([h[ptr[g]:ptr[g + 1]].mean(0) for g in range(B)])
    elif cfg.pooling == "max":
        out["graph_embedding"] = torch.stack([h[ptr[g]:ptr[g + 1]].max(0)[0] for g in range(B)])
    else:
        raise ValueError(cfg.pooling)
    _stage("pool")
    g = out["graph_embedding"]
    if cfg.num_classes is not None and mode in ("inference", "finetune"):      # dgdm_model.py:383-391
        out["classification_logits"] = classification_head(P, g, cfg.activation, cfg.dropout, training)
        out["classification_probs"] = F.softmax(out["classification_logits"], dim=-1)
    if cfg.regression_targets > 0 and mode in ("inference", "finetune"):
        out["regression_outputs"] = regression_head(P, g, cfg.activation, cfg.dropout, training)
    if return_embeddings:
        out["node_embeddings"] = h
    if return_attention and attn_w is not None:
        out["attention_weights"] = attn_w
    return out


def apply_entity_masking(x: Tensor, mask_indices: Tensor, mask_token: Tensor) -> Tensor:
    """_apply_entity_masking (models/dgdm_model.py:482-506) with the draws injected."""
    xm = x.clone()
    xm[mask_indices] = mask_token.to(x.dtype)
    return xm


def pretrain_step(P, cfg: OracleConfig, data, *, mask_indices=None, mask_token=None, training=False, **rng):
    """DGDMModel.pretrain_step (models/dgdm_model.py:447-480)."""
    class _D: pass
    d = _D()
    for k in ("x", "edge_index", "edge_attr", "pos", "batch"):
        setattr(d, k, getattr(data, k, None))
    if mask_indices is not None and mask_indices.numel() > 0:
        d.x = apply_entity_masking(data.x, mask_indices, mask_token)
    out = forward(P, cfg, d, "pretrain", training=training, **rng)
    out["total_pretrain_loss"] = out["diffusion_loss"]
    return out


def loss_and_grads(P, cfg, data, **kw):
    """Helper for tests/bench: run pretrain_step with grads on every parameter (and on every
    traced activation when a ``trace`` dict is passed)."""
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    out = pretrain_step(Pg, cfg, data, **kw)
    if kw.get("trace") is not None:
        for t in kw["trace"].values():
            if isinstance(t, torch.Tensor) and t.requires_grad:
                t.retain_grad()
    out["total_pretrain_loss"].backward()
    grads = {k: v.grad for k, v in Pg.items() if v.grad is not None}
    return out, grads
