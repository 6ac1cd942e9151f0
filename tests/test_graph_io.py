"""Graph files of the reference's data pipeline (data/dataset.py:186-240: .h5 schema, .pt, .pkl) -> GraphData.  CPU only."""
import pickle

import pytest
import torch

from dgdm_histopath_lab_amd import graph_io
from dgdm_histopath_lab_amd.graph import GraphBatch, GraphData
from dgdm_histopath_lab_amd.synthetic import synthetic_graph

needs_hdf5 = pytest.mark.skipif(graph_io.hdf5_backend() is None, reason="neither h5py nor libhdf5 >= 1.10 on this machine")


@needs_hdf5
def test_h5_schema_round_trip(tmp_path):
    g = synthetic_graph(3, 57, 200)
    g.y = torch.tensor([2])
    p = tmp_path / "slide_0003.h5"
    graph_io.save_graph_h5(p, g, metadata={"slide_id": "TCGA-XX-0003", "magnification": 20, "mpp": 0.5})
    assert open(p, "rb").read(8) == b"\x89HDF\r\n\x1a\n"                  # a real HDF5 container
    r = graph_io.load_graph(p)
    assert r.x.dtype == torch.float32 and r.edge_index.dtype == torch.int64 and r.y.dtype == torch.int64
    for a, b in ((r.x, g.x), (r.edge_index, g.edge_index), (r.edge_attr, g.edge_attr), (r.pos, g.pos), (r.y, g.y)):
        assert torch.equal(a, b)
    assert r.slide_id == "TCGA-XX-0003" and r.magnification == 20 and r.mpp == 0.5
    # what comes out batches like any other graph (the layout the model takes)
    b = GraphBatch.from_data_list([r, synthetic_graph(4, 10, 30)])
    assert b.x.shape == (67, 768) and b.ptr == [0, 57, 67] and int(b.edge_index.max()) < 67


@needs_hdf5
def test_h5_optional_datasets_and_type_conversion(tmp_path):
    """Only node_features and edge_index are mandatory (data/dataset.py:212-226); stored types are converted on the way in, as
    the reference's torch.tensor(..., dtype=float / long) does -- here float64 features and int32 indices written with h5py's
    defaults would read the same way (the HDF5 library converts to the memory type)."""
    g = GraphData(x=torch.randn(5, 8), edge_index=torch.tensor([[0, 1, 2], [1, 2, 3]]))
    p = tmp_path / "g.hdf5"
    graph_io.save_graph_h5(p, g)
    r = graph_io.load_graph(p)
    assert torch.equal(r.x, g.x) and torch.equal(r.edge_index, g.edge_index)
    assert r.edge_attr is None and r.pos is None and r.y is None
    bad = tmp_path / "bad.h5"
    graph_io.save_graph_h5(bad, GraphData(x=torch.randn(2, 2), edge_index=torch.zeros(2, 1, dtype=torch.long)))
    lib = graph_io._libhdf5()
    if lib is not None:      # a file without the mandatory datasets is rejected with the schema named
        f = lib.H5Fcreate(str(tmp_path / "empty.h5").encode(), 2, 0, 0)
        lib.H5Fclose(f)
        with pytest.raises(graph_io.GraphFormatError, match="node_features"):
            graph_io.load_graph(tmp_path / "empty.h5")
    with pytest.raises(graph_io.GraphFormatError):
        (tmp_path / "junk.h5").write_bytes(b"not an hdf5 file")
        graph_io.load_graph(tmp_path / "junk.h5")


def test_pt_pkl_and_unknown_suffix(tmp_path):
    g = synthetic_graph(1, 12, 40)
    torch.save({"x": g.x, "edge_index": g.edge_index, "edge_attr": g.edge_attr, "pos": g.pos, "y": torch.tensor([1])}, tmp_path / "g.pt")
    r = graph_io.load_graph(tmp_path / "g.pt")
    assert torch.equal(r.x, g.x) and torch.equal(r.edge_index, g.edge_index) and int(r.y) == 1
    with open(tmp_path / "g.pkl", "wb") as fh:
        pickle.dump({"node_features": g.x.numpy(), "edge_index": g.edge_index.numpy(), "node_pos": g.pos.numpy()}, fh)
    r2 = graph_io.load_graph(tmp_path / "g.pkl")
    assert torch.equal(r2.x, g.x) and torch.equal(r2.pos, g.pos) and r2.edge_attr is None
    with pytest.raises(graph_io.GraphFormatError, match="Unsupported graph format"):
        graph_io.load_graph(tmp_path / "g.json")
