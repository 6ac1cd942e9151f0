"""Synthetic tissue graphs of stated (nodes, edges, features) -- the inputs of BASELINE.json's
configs: graph g draws from ``torch.Generator().manual_seed(1000+g)``:
x ~ N(0,1) [N,F]; pos ~ U[0,1)^2; E/2 undirected pairs (u != v, uniform) emitted in both
directions as consecutive columns (the layout preprocessing/tissue_graph_builder.py:399-401
produces); edge_attr ~ N(0,1) [E/2,32] shared by the two directions."""
from __future__ import annotations

from typing import List

import torch

from .graph import GraphBatch, GraphData

EDGE_DIM = 32


def synthetic_graph(g: int, num_nodes: int, num_edges: int, node_features: int = 768, edge_attr: bool = True) -> GraphData:
    gen = torch.Generator().manual_seed(1000 + g)
    x = torch.randn(num_nodes, node_features, generator=gen)
    pos = torch.rand(num_nodes, 2, generator=gen)
    half = num_edges // 2
    u = torch.randint(0, num_nodes, (half,), generator=gen)
    v = torch.randint(0, num_nodes - 1, (half,), generator=gen)
    v = v + (v >= u).long()  # uniform over v != u
    ei = torch.stack([torch.stack([u, v]), torch.stack([v, u])], dim=2).reshape(2, -1)
    ea = torch.randn(half, EDGE_DIM, generator=gen).repeat_interleave(2, dim=0) if edge_attr else None
    return GraphData(x=x, edge_index=ei, edge_attr=ea, pos=pos, pos_extent=1.0)      # U[0,1)^2: known without looking


def synthetic_batch(first_graph: int, batch_size: int, num_nodes: int, num_edges: int, node_features: int = 768) -> GraphBatch:
    return GraphBatch.from_data_list([synthetic_graph(first_graph + i, num_nodes, num_edges, node_features)
                                      for i in range(batch_size)])
