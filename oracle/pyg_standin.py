"""TEST INFRASTRUCTURE ONLY -- build-owned stand-in for the seven torch-geometric symbols the
reference's hot-path files import (torch-geometric >= 2.3 is a declared dependency of the
reference, ``requirements.txt:4``, and is NOT installed in this image).

Used only by ``oracle/capture_golden.py`` (dev container, where /root/reference exists) so
that ``core/graph_layers.py``, ``models/encoders.py`` and ``models/dgdm_model.py`` can be
imported and executed *as they are*.  The semantics below restate PyG's published
behaviour for the call sites on the path (``graph_layers.py:12-14,78,81,92,203``):

* ``MessagePassing.propagate(edge_index, **kw)``: for every argument name of ``message``
  ending in ``_j`` gather ``kw[name[:-2]]`` by ``edge_index[0]`` (source), ``_i`` by
  ``edge_index[1]`` (target); other names pass through; ``aggr='add'`` scatter-adds the
  messages over ``edge_index[1]``.
* ``add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None)``: appends
  ``(i, i)`` for every node after the existing edges; attributes get ``fill_value``
  (default 1.0) rows.
* ``degree(index, num_nodes, dtype)``: bincount.   ``softmax(src, index, num_nodes)``:
  per-segment softmax.   ``Data``/``Batch``: attribute bags with PyG's collation rule.

R1 lives here (SURVEY.md 8(a') D1): the reference extends ``edge_index`` with self loops
but passes the un-extended ``edge_attr`` to ``propagate``; when the attribute tensor is
exactly ``num_nodes`` rows short, ``propagate`` pads it with zero rows (== what
``add_self_loops(edge_index, edge_attr, fill_value=0.)`` would have produced).
"""
from __future__ import annotations

import inspect
import sys
import types
from typing import Optional, Tuple, Union

import torch
from torch import Tensor


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr: str = "add", **kwargs):
        super().__init__()
        self.aggr = aggr

    def propagate(self, edge_index: Tensor, size=None, **kwargs):
        src, dst = edge_index[0], edge_index[1]
        if isinstance(kwargs.get("x"), Tensor):
            n = kwargs["x"].size(0)
        elif size is not None and size[1] is not None:
            n = size[1]
        else:
            n = int(dst.max()) + 1
        args = {}
        for name in inspect.signature(self.message).parameters:
            if name.endswith("_j"):
                args[name] = kwargs[name[:-2]][src]
            elif name.endswith("_i"):
                args[name] = kwargs[name[:-2]][dst]
            else:
                v = kwargs.get(name)
                if name == "edge_attr" and isinstance(v, Tensor) and v.size(0) == edge_index.size(1) - n:
                    v = torch.cat([v, v.new_zeros(n, v.size(1))], dim=0)  # R1
                args[name] = v
        msg = self.message(**args)
        if self.aggr != "add":
            raise NotImplementedError(self.aggr)
        out = msg.new_zeros(n, msg.size(1))
        return out.index_add(0, dst, msg)

    def message(self, x_j):  # pragma: no cover
        return x_j


def add_self_loops(edge_index: Tensor, edge_attr: Optional[Tensor] = None, fill_value=None,
                   num_nodes: Optional[int] = None) -> Tuple[Tensor, Optional[Tensor]]:
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    loop = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    ei = torch.cat([edge_index, torch.stack([loop, loop])], dim=1)
    if edge_attr is not None:
        fv = 1.0 if fill_value is None else fill_value
        edge_attr = torch.cat([edge_attr, edge_attr.new_full((n,) + tuple(edge_attr.shape[1:]), fv)], dim=0)
    return ei, edge_attr


def degree(index: Tensor, num_nodes: Optional[int] = None, dtype=None) -> Tensor:
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros(n, dtype=dtype or torch.float32)
    return out.index_add_(0, index, torch.ones(index.numel(), dtype=out.dtype))


def softmax(src: Tensor, index: Tensor, ptr=None, num_nodes: Optional[int] = None, dim: int = 0) -> Tensor:
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    shape = (n,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    mx = torch.full(shape, float("-inf"), dtype=src.dtype).scatter_reduce(0, idx, src, "amax")
    e = (src - mx[index]).exp()
    den = torch.zeros(shape, dtype=src.dtype).index_add_(0, index, e)
    return e / (den[index] + 1e-16)


class Data:
    """Attribute bag; unset standard fields read as None (PyG 2.x behaviour, D4)."""
    _std = ("x", "edge_index", "edge_attr", "y", "pos", "batch")

    def __init__(self, **kw):
        for k in self._std:
            object.__setattr__(self, k, None)
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_graphs(self):
        return 1 if self.batch is None else int(self.batch.max()) + 1

    def clone(self):
        out = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            object.__setattr__(out, k, v.clone() if isinstance(v, Tensor) else v)
        return out


class Batch(Data):
    @classmethod
    def from_data_list(cls, data_list):
        xs, eis, eas, poss, bs, off = [], [], [], [], [], 0
        for g, d in enumerate(data_list):
            n = d.x.size(0)
            xs.append(d.x); eis.append(d.edge_index + off)
            if d.edge_attr is not None: eas.append(d.edge_attr)
            if d.pos is not None: poss.append(d.pos)
            bs.append(torch.full((n,), g, dtype=torch.long)); off += n
        return cls(x=torch.cat(xs), edge_index=torch.cat(eis, dim=1),
                   edge_attr=torch.cat(eas) if eas else None, pos=torch.cat(poss) if poss else None,
                   batch=torch.cat(bs))


def install() -> None:
    """Register the stand-in as ``torch_geometric`` (only if the real one is absent)."""
    if "torch_geometric" in sys.modules:
        return
    root = types.ModuleType("torch_geometric"); root.__path__ = []
    nn = types.ModuleType("torch_geometric.nn"); nn.MessagePassing = MessagePassing
    utils = types.ModuleType("torch_geometric.utils")
    utils.add_self_loops, utils.degree, utils.softmax = add_self_loops, degree, softmax
    typing_m = types.ModuleType("torch_geometric.typing")
    typing_m.Adj, typing_m.OptTensor, typing_m.PairTensor = Tensor, Optional[Tensor], Tuple[Tensor, Tensor]
    data = types.ModuleType("torch_geometric.data"); data.Data, data.Batch = Data, Batch
    root.nn, root.utils, root.typing, root.data = nn, utils, typing_m, data
    root.__version__ = "0.0-standin"
    for m in (root, nn, utils, typing_m, data):
        sys.modules[m.__name__] = m
