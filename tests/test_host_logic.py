"""Host-side logic that needs no GPU: the positions' extent hint (graph.py, ops.attn_zero_blocks_possible) and the loader-side plumbing
that carries it into a recording's signature."""
import torch


def test_pos_extent_travels_from_the_host_tensor_to_the_batch_and_decides_the_zero_block_map():
    from dgdm_histopath_lab_amd import ops
    from dgdm_histopath_lab_amd.graph import GraphBatch, GraphData
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    from dgdm_histopath_lab_amd.training import GraphedPretrainStep
    b = synthetic_batch(0, 2, 50, 100, 16)
    assert b.pos_extent == 1.0                                            # U[0,1)^2, known without looking
    assert not ops.attn_zero_blocks_possible(b.pos_extent, 1.0)           # BASELINE's positions: no zero pair, no map
    g = torch.Generator().manual_seed(0)
    ei = torch.zeros(2, 0, dtype=torch.long)
    pix = GraphData(x=torch.randn(9, 16, generator=g), edge_index=ei, pos=torch.tensor([[0.0, 0.0], [22400.0, 100.0], [5.0, 9000.0]]).repeat(3, 1))
    assert pix.host_pos_extent() == 32768.0                               # 22400 rounded up to a power of two
    assert ops.attn_zero_blocks_possible(pix.host_pos_extent(), 1.0)
    both = GraphBatch.from_data_list([pix, pix])
    assert both.pos_extent == 32768.0 and both.to("cpu").pos_extent == 32768.0 and both.clone().pos_extent == 32768.0
    unknown = GraphData(x=pix.x, edge_index=ei, pos=pix.pos)
    unknown.pos = None
    assert unknown.host_pos_extent() is None and ops.attn_zero_blocks_possible(None, 1.0)
    # a recording made for one extent class must not replay a batch of another: the extent is part of the layout signature
    s1 = GraphedPretrainStep._sig(both)
    both2 = both.clone()
    both2.pos_extent = 1.0
    assert s1 != GraphedPretrainStep._sig(both2) and ("pos_extent", 32768.0) in s1
    # zero extent / tiny graphs
    one = GraphData(x=torch.randn(1, 16), edge_index=ei, pos=torch.zeros(1, 2))
    assert one.host_pos_extent() == 0.0 and not ops.attn_zero_blocks_possible(0.0, 1.0)


def test_max_degree_hint_is_taken_on_the_host_and_is_part_of_a_recordings_signature():
    from dgdm_histopath_lab_amd.graph import GraphBatch, GraphData
    from dgdm_histopath_lab_amd.synthetic import synthetic_batch
    from dgdm_histopath_lab_amd.training import GraphedPretrainStep
    ei = torch.tensor([[0, 0, 0, 1, 2, 3], [1, 2, 3, 0, 0, 0]])
    g = GraphData(x=torch.randn(4, 8), edge_index=ei)
    assert g.host_max_degree() == 3
    hub = GraphData(x=torch.randn(300, 8), edge_index=torch.stack([torch.arange(1, 300), torch.zeros(299, dtype=torch.long)]))
    assert hub.host_max_degree() == 299
    b = GraphBatch.from_data_list([g, hub])
    assert b.max_degree == 299 and b.to("cpu").max_degree == 299
    s = synthetic_batch(0, 2, 200, 1000, 16)
    assert 5 <= s.max_degree <= 128
    assert ("long_rows", False) in GraphedPretrainStep._sig(s) and ("long_rows", True) in GraphedPretrainStep._sig(b)
    s2 = s.clone()
    s2.max_degree = None                       # unknown: the worst case is assumed
    assert ("long_rows", True) in GraphedPretrainStep._sig(s2)
    empty = GraphData(x=torch.randn(3, 8), edge_index=torch.zeros(2, 0, dtype=torch.long))
    assert empty.host_max_degree() == 0
