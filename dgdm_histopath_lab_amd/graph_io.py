"""Graph files of the reference's data pipeline -> ``GraphData`` (the step upstream of the hot path).

The reference stores one tissue graph per file and dispatches on the suffix (data/dataset.py:186-200): ``.h5`` (HDF5 with the
datasets ``node_features`` float [N,F], ``edge_index`` int [2,E], and optionally ``edge_attr`` float [E,D], ``node_pos`` float
[N,2], ``labels`` int, plus scalar attributes on a ``metadata`` group, data/dataset.py:202-240), ``.pt`` (``torch.load``) and
``.pkl``.  ``load_graph`` reads all three; ``save_graph_h5`` writes the same HDF5 schema (test fixtures, export).

HDF5 access: ``h5py`` when it is installed, otherwise the HDF5 C library itself through ctypes (``libhdf5.so``; found via the
loader path, ``$CONDA_PREFIX/lib`` or ``/opt/conda/lib``).  Without either, ``load_graph`` of an ``.h5`` file raises
``GraphFormatError`` naming what is missing -- nothing is guessed.  Host code only: no kernel is involved.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
import pickle
from typing import Optional

import numpy as np
import torch

from .graph import GraphData


class GraphFormatError(ValueError):
    """Unsupported suffix, missing HDF5 support, or a file that does not follow the schema of data/dataset.py:202-240."""


# ----------------------------------------------------------------------------------------------- libhdf5 through ctypes
_hid = C.c_int64           # hid_t of HDF5 >= 1.10
_H5 = None


def _libhdf5():
    global _H5
    if _H5 is not None:
        return _H5
    cands = []
    found = ctypes.util.find_library("hdf5")
    if found:
        cands.append(found)
    for root in (os.environ.get("CONDA_PREFIX"), "/opt/conda", "/usr", "/usr/local"):
        if root:
            cands += [os.path.join(root, "lib", "libhdf5.so"), os.path.join(root, "lib", "x86_64-linux-gnu", "hdf5", "serial", "libhdf5.so")]
    for cand in cands:
        try:
            lib = C.CDLL(cand)
        except OSError:
            continue
        if lib.H5open() < 0:
            continue
        maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
        lib.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
        if (maj.value, mnr.value) < (1, 10):        # 32-bit hid_t before 1.10: not what the signatures below assume
            continue
        sig = {
            "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]), "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]), "H5Fclose": (C.c_int, [_hid]),
            "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]),
            "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]), "H5Dclose": (C.c_int, [_hid]), "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]),
            "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]), "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
            "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]),
            "H5Sget_simple_extent_ndims": (C.c_int, [_hid]), "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            "H5Screate_simple": (_hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]), "H5Screate": (_hid, [C.c_int]), "H5Sclose": (C.c_int, [_hid]),
            "H5Tget_class": (C.c_int, [_hid]), "H5Tget_size": (C.c_size_t, [_hid]), "H5Tclose": (C.c_int, [_hid]), "H5Tcopy": (_hid, [_hid]),
            "H5Tset_size": (C.c_int, [_hid, C.c_size_t]), "H5Tis_variable_str": (C.c_int, [_hid]),
            "H5Gopen2": (_hid, [_hid, C.c_char_p, _hid]), "H5Gcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid]), "H5Gclose": (C.c_int, [_hid]),
            "H5Aget_num_attrs": (C.c_int, [_hid]), "H5Aopen_by_idx": (_hid, [_hid, C.c_char_p, C.c_int, C.c_int, C.c_uint64, _hid, _hid]),
            "H5Aget_name": (C.c_ssize_t, [_hid, C.c_size_t, C.c_char_p]), "H5Aget_type": (_hid, [_hid]), "H5Aget_space": (_hid, [_hid]),
            "H5Aread": (C.c_int, [_hid, _hid, C.c_void_p]), "H5Aclose": (C.c_int, [_hid]),
            "H5Acreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid]), "H5Awrite": (C.c_int, [_hid, _hid, C.c_void_p]),
            "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        lib.H5Eset_auto2(0, None, None)          # errors come back as return codes, not as text on stderr
        _H5 = lib
        return lib
    return None


def hdf5_backend() -> Optional[str]:
    """'h5py', 'libhdf5' (ctypes) or None."""
    try:
        import h5py  # noqa: F401
        return "h5py"
    except Exception:
        return "libhdf5" if _libhdf5() is not None else None


def _native(lib, name: str) -> int:
    return _hid.in_dll(lib, name).value


_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING = 0, 1, 3


def _read_dataset(lib, loc: int, name: bytes, kind: str) -> np.ndarray:
    """Dataset ``name`` under ``loc`` as float32 (kind 'f') or int64 (kind 'i'); the library converts from the stored type."""
    d = lib.H5Dopen2(loc, name, 0)
    if d < 0:
        raise GraphFormatError(f"cannot open dataset {name.decode()!r}")
    try:
        sp = lib.H5Dget_space(d)
        nd = lib.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_uint64 * max(nd, 1))()
        if nd > 0:
            lib.H5Sget_simple_extent_dims(sp, dims, None)
        lib.H5Sclose(sp)
        shape = tuple(int(dims[i]) for i in range(nd))
        ft = lib.H5Dget_type(d)
        cls = lib.H5Tget_class(ft)
        lib.H5Tclose(ft)
        if cls not in (_H5T_INTEGER, _H5T_FLOAT):
            raise GraphFormatError(f"dataset {name.decode()!r} is neither integer nor floating point (HDF5 class {cls})")
        out = np.empty(shape, dtype=np.float32 if kind == "f" else np.int64)
        mem = _native(lib, "H5T_NATIVE_FLOAT_g" if kind == "f" else "H5T_NATIVE_INT64_g")
        if out.size and lib.H5Dread(d, mem, 0, 0, 0, out.ctypes.data_as(C.c_void_p)) < 0:
            raise GraphFormatError(f"reading dataset {name.decode()!r} failed")
        return out
    finally:
        lib.H5Dclose(d)


def _read_attrs(lib, obj: int) -> dict:
    out = {}
    for i in range(max(lib.H5Aget_num_attrs(obj), 0)):
        a = lib.H5Aopen_by_idx(obj, b".", 0, 0, i, 0, 0)      # H5_INDEX_NAME, H5_ITER_INC
        if a < 0:
            continue
        try:
            n = lib.H5Aget_name(a, 0, None)
            buf = C.create_string_buffer(n + 1)
            lib.H5Aget_name(a, n + 1, buf)
            key = buf.value.decode()
            ft, sp = lib.H5Aget_type(a), lib.H5Aget_space(a)
            cls, size, nd = lib.H5Tget_class(ft), lib.H5Tget_size(ft), lib.H5Sget_simple_extent_ndims(sp)
            if nd == 0 and cls == _H5T_INTEGER:
                v = C.c_int64()
                if lib.H5Aread(a, _native(lib, "H5T_NATIVE_INT64_g"), C.byref(v)) >= 0:
                    out[key] = int(v.value)
            elif nd == 0 and cls == _H5T_FLOAT:
                v = C.c_double()
                if lib.H5Aread(a, _native(lib, "H5T_NATIVE_DOUBLE_g"), C.byref(v)) >= 0:
                    out[key] = float(v.value)
            elif nd == 0 and cls == _H5T_STRING:
                if lib.H5Tis_variable_str(ft) > 0:
                    p = C.c_char_p()
                    mt = lib.H5Tcopy(_native(lib, "H5T_C_S1_g"))
                    lib.H5Tset_size(mt, C.c_size_t(-1).value)          # H5T_VARIABLE
                    if lib.H5Aread(a, mt, C.byref(p)) >= 0 and p.value is not None:
                        out[key] = p.value.decode(errors="replace")
                    lib.H5Tclose(mt)
                else:
                    sbuf = C.create_string_buffer(size + 1)
                    if lib.H5Aread(a, ft, sbuf) >= 0:
                        out[key] = sbuf.value.decode(errors="replace")
            lib.H5Tclose(ft)
            lib.H5Sclose(sp)
        finally:
            lib.H5Aclose(a)
    return out


def _load_h5_ctypes(path: str) -> dict:
    lib = _libhdf5()
    if lib is None:
        raise GraphFormatError("reading .h5 graphs needs h5py or the HDF5 C library (libhdf5.so >= 1.10); neither was found")
    f = lib.H5Fopen(os.fsencode(path), 0, 0)               # H5F_ACC_RDONLY
    if f < 0:
        raise GraphFormatError(f"{path}: not an HDF5 file (or unreadable)")
    try:
        has = lambda n: lib.H5Lexists(f, n, 0) > 0
        if not (has(b"node_features") and has(b"edge_index")):
            raise GraphFormatError(f"{path}: the graph schema needs the datasets 'node_features' and 'edge_index' (data/dataset.py:206-209)")
        rec = {"node_features": _read_dataset(lib, f, b"node_features", "f"), "edge_index": _read_dataset(lib, f, b"edge_index", "i")}
        for name, kind in ((b"edge_attr", "f"), (b"node_pos", "f"), (b"labels", "i")):
            if has(name):
                rec[name.decode()] = _read_dataset(lib, f, name, kind)
        if has(b"metadata"):
            g = lib.H5Gopen2(f, b"metadata", 0)
            if g >= 0:
                rec["metadata"] = _read_attrs(lib, g)
                lib.H5Gclose(g)
        return rec
    finally:
        lib.H5Fclose(f)


def _load_h5_h5py(path: str) -> dict:
    import h5py
    with h5py.File(path, "r") as f:
        if "node_features" not in f or "edge_index" not in f:
            raise GraphFormatError(f"{path}: the graph schema needs the datasets 'node_features' and 'edge_index' (data/dataset.py:206-209)")
        rec = {"node_features": np.asarray(f["node_features"][:], dtype=np.float32), "edge_index": np.asarray(f["edge_index"][:], dtype=np.int64)}
        for name, dt in (("edge_attr", np.float32), ("node_pos", np.float32), ("labels", np.int64)):
            if name in f:
                rec[name] = np.asarray(f[name][:], dtype=dt)
        if "metadata" in f:
            rec["metadata"] = {k: (v.decode() if isinstance(v, bytes) else (v.item() if hasattr(v, "item") else v)) for k, v in f["metadata"].attrs.items()}
        return rec


def _to_graph(rec: dict) -> GraphData:
    """data/dataset.py:204-238: float node features / edge attributes / positions, int64 edge_index and labels; the attributes of
    the ``metadata`` group become attributes of the graph object."""
    ei = torch.from_numpy(np.ascontiguousarray(rec["edge_index"]))
    if ei.dim() != 2 or ei.size(0) != 2:
        raise GraphFormatError(f"edge_index must be [2, E], got {tuple(ei.shape)}")
    opt = lambda k: torch.from_numpy(np.ascontiguousarray(rec[k])) if k in rec else None
    g = GraphData(x=torch.from_numpy(np.ascontiguousarray(rec["node_features"])), edge_index=ei, edge_attr=opt("edge_attr"), pos=opt("node_pos"),
                  y=opt("labels"))
    for k, v in rec.get("metadata", {}).items():
        if not hasattr(g, k):
            setattr(g, k, v)
    return g


def load_graph(path) -> GraphData:
    """The reference's ``_load_graph`` dispatch (data/dataset.py:186-200) without torch_geometric: ``.h5`` / ``.hdf5`` by the schema
    above, ``.pt`` via ``torch.load``, ``.pkl`` via ``pickle``.  ``.pt`` / ``.pkl`` payloads may be a ``GraphData``, any object
    with ``x`` / ``edge_index`` attributes (a PyG ``Data`` if PyG is installed) or a dict of tensors / arrays."""
    path = os.fspath(path)
    suffix = os.path.splitext(path)[1].lower()
    if suffix in (".h5", ".hdf5"):
        try:
            import h5py  # noqa: F401
            rec = _load_h5_h5py(path)
        except ImportError:
            rec = _load_h5_ctypes(path)
        return _to_graph(rec)
    if suffix == ".pt":
        obj = torch.load(path, map_location="cpu", weights_only=False)
    elif suffix == ".pkl":
        with open(path, "rb") as fh:
            obj = pickle.load(fh)
    else:
        raise GraphFormatError(f"Unsupported graph format: {suffix}")
    if isinstance(obj, GraphData):
        return obj
    get = (lambda k: obj.get(k)) if isinstance(obj, dict) else (lambda k: getattr(obj, k, None))
    x = get("x") if get("x") is not None else get("node_features")
    ei = get("edge_index")
    if x is None or ei is None:
        raise GraphFormatError(f"{path}: payload has no node features / edge_index")
    as_t = lambda v, dt: None if v is None else torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(dt)
    pos = get("pos") if get("pos") is not None else get("node_pos")
    y = get("y") if get("y") is not None else get("labels")
    return GraphData(x=as_t(x, torch.float32), edge_index=as_t(ei, torch.int64), edge_attr=as_t(get("edge_attr"), torch.float32),
                     pos=as_t(pos, torch.float32), y=as_t(y, torch.int64))


def save_graph_h5(path, graph, metadata: Optional[dict] = None) -> None:
    """Write ``graph`` with the schema ``load_graph`` reads (contiguous datasets, float32 / int64; ``metadata``: scalar int / float /
    str attributes of a ``metadata`` group)."""
    path = os.fspath(path)
    arrays = {"node_features": (graph.x, np.float32), "edge_index": (graph.edge_index, np.int64), "edge_attr": (graph.edge_attr, np.float32),
              "node_pos": (graph.pos, np.float32), "labels": (graph.y, np.int64)}
    arrays = {k: np.ascontiguousarray(v.detach().cpu().numpy().astype(dt)) for k, (v, dt) in arrays.items() if v is not None}
    try:
        import h5py
        with h5py.File(path, "w") as f:
            for k, a in arrays.items():
                f.create_dataset(k, data=a)
            if metadata:
                g = f.create_group("metadata")
                for k, v in metadata.items():
                    g.attrs[k] = v
        return
    except ImportError:
        pass
    lib = _libhdf5()
    if lib is None:
        raise GraphFormatError("writing .h5 graphs needs h5py or the HDF5 C library (libhdf5.so >= 1.10); neither was found")
    f = lib.H5Fcreate(os.fsencode(path), 2, 0, 0)            # H5F_ACC_TRUNC
    if f < 0:
        raise GraphFormatError(f"cannot create {path}")
    try:
        for k, a in arrays.items():
            dims = (C.c_uint64 * max(a.ndim, 1))(*a.shape)
            sp = lib.H5Screate_simple(a.ndim, dims, None)
            ftype = _native(lib, "H5T_IEEE_F32LE_g" if a.dtype == np.float32 else "H5T_STD_I64LE_g")
            mtype = _native(lib, "H5T_NATIVE_FLOAT_g" if a.dtype == np.float32 else "H5T_NATIVE_INT64_g")
            d = lib.H5Dcreate2(f, k.encode(), ftype, sp, 0, 0, 0)
            ok = d >= 0 and (a.size == 0 or lib.H5Dwrite(d, mtype, 0, 0, 0, a.ctypes.data_as(C.c_void_p)) >= 0)
            if d >= 0:
                lib.H5Dclose(d)
            lib.H5Sclose(sp)
            if not ok:
                raise GraphFormatError(f"writing dataset {k!r} failed")
        if metadata:
            g = lib.H5Gcreate2(f, b"metadata", 0, 0, 0)
            sp = lib.H5Screate(0)                              # H5S_SCALAR
            for k, v in metadata.items():
                if isinstance(v, bool) or isinstance(v, int):
                    val = C.c_int64(int(v))
                    a = lib.H5Acreate2(g, k.encode(), _native(lib, "H5T_STD_I64LE_g"), sp, 0, 0)
                    lib.H5Awrite(a, _native(lib, "H5T_NATIVE_INT64_g"), C.byref(val))
                elif isinstance(v, float):
                    val = C.c_double(v)
                    a = lib.H5Acreate2(g, k.encode(), _native(lib, "H5T_IEEE_F64LE_g"), sp, 0, 0)
                    lib.H5Awrite(a, _native(lib, "H5T_NATIVE_DOUBLE_g"), C.byref(val))
                else:
                    raw = str(v).encode()
                    st = lib.H5Tcopy(_native(lib, "H5T_C_S1_g"))
                    lib.H5Tset_size(st, max(len(raw), 1))
                    a = lib.H5Acreate2(g, k.encode(), st, sp, 0, 0)
                    lib.H5Awrite(a, st, C.create_string_buffer(raw, max(len(raw), 1)))
                    lib.H5Tclose(st)
                if a >= 0:
                    lib.H5Aclose(a)
            lib.H5Sclose(sp)
            lib.H5Gclose(g)
    finally:
        lib.H5Fclose(f)
