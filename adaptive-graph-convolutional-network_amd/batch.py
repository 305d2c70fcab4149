"""Ragged batches of graphs on the device.

Replaces the reference's feed construction (`models/tf_modules/graph_topology.py:84-135`:
every graph zero-padded to max_atom, the CSR Laplacian densified every step, int32
data_slice / lap_slice tensors) by one packed layout that the kernels index directly:

    rows   graph b owns rows [row_off[b], row_off[b+1]) of every node-feature matrix
    tiles  graph b owns n_b x n_b floats at tile_off[b] of every per-graph square output
    L0     the original Laplacians as one CSR over the row space (f32 values, local columns)

The integer bookkeeping (the *plan*) is built by the C library on the host
(`agcn_plan_build`) and uploaded once per batch.
"""
import ctypes as C
from collections.abc import Sequence

import numpy as np
import torch

from . import _ffi


# The reference's padded feed is gathered into packed rows at the first layer (GraphBatch.from_padded); set to
# False to keep the rows padded end to end (slower; the layout the reference itself computes on).
PACK_REFERENCE_FEED = True


def _csr_is_symmetric(n, indptr, indices, data):
    """Exact (bitwise, after the float32 cast the device sees) symmetry of one CSR matrix."""
    n = int(n)
    off = np.array([0, n], dtype=np.int64)
    return bool(csr_symmetry_flags(off, np.asarray(indptr, dtype=np.int64), indices, np.asarray(data).astype(np.float32))[0])


def csr_symmetry_flags(row_off, rowptr, col, val):
    """Per-graph exact symmetry of a block-diagonal CSR (graph g owns rows row_off[g]..row_off[g+1], `col` is
    the column inside the graph, `rowptr` spans all rows): flag[g] = 1 iff L_g == L_g^T bit for bit (explicit
    zeros count as entries).  One sort of the entries by (row, col) and one by (col, row): no per-graph loop."""
    row_off = np.asarray(row_off, dtype=np.int64)
    rowptr = np.asarray(rowptr, dtype=np.int64)
    G, R = row_off.size - 1, int(row_off[-1])
    flags = np.ones(G, dtype=np.uint8)
    nnz = int(rowptr[-1])
    if nnz == 0:
        return flags
    r = np.repeat(np.arange(R, dtype=np.int64), np.diff(rowptr))
    g = np.searchsorted(row_off, r, side="right") - 1
    c = row_off[g] + np.asarray(col, dtype=np.int64)
    v = np.asarray(val)
    k1, k2 = r * R + c, c * R + r
    o1, o2 = np.argsort(k1, kind="stable"), np.argsort(k2, kind="stable")
    bad = (k1[o1] != k2[o2]) | (v[o1] != v[o2])
    if bad.any():
        flags[np.unique(g[o1][bad])] = 0
        flags[np.unique(g[o2][bad])] = 0
    return flags


class Plan(object):
    """Host + device copies of the plan blob, plus numpy views of its arrays."""

    def __init__(self, n_nodes, row_stride=0, device=None):
        lib = _ffi.lib()
        n = np.ascontiguousarray(np.asarray(n_nodes, dtype=np.int32))
        if n.ndim != 1 or n.size < 1:
            raise ValueError("n_nodes must be a non-empty 1-D array")
        B = int(n.size)
        nbytes = lib.agcn_plan_bytes_for(n.ctypes.data, B, int(row_stride))
        if nbytes == 0:
            raise _ffi.AgcnError('invalid node counts')
        self.host = np.zeros(nbytes, dtype=np.uint8)
        _ffi.check(lib.agcn_plan_build(n.ctypes.data, B, int(row_stride), self.host.ctypes.data, nbytes))
        info = _ffi.PlanInfo()
        _ffi.check(lib.agcn_plan_info(self.host.ctypes.data, C.byref(info)))
        self._adopt(info, row_stride)
        if device is not None:
            self.to(device)

    def _adopt(self, info, row_stride):
        B = info.B
        self.info = info
        self.B, self.N1, self.rows, self.max_n = info.B, info.N1, info.rows, info.max_n
        self.N2, self.tile_elems = info.N2, info.tile_elems
        self.row_stride = int(row_stride)
        host = self.host

        def view(off, dtype, count):
            return host[off:off + count * np.dtype(dtype).itemsize].view(dtype)

        self.n = view(info.off_n, np.int32, B)
        self.row_off = view(info.off_row_off, np.int32, B + 1)
        self.tile_off = view(info.off_tile_off, np.int64, B + 1)
        self.order = view(info.off_order, np.int32, B)
        self.n_slabs = info.n_slabs
        self.slab_graph = view(info.off_slab_graph, np.int32, info.n_slabs)
        self.slab_row = view(info.off_slab_row, np.int32, info.n_slabs)
        self.dev = None
        self.device = None

    @classmethod
    def from_blob(cls, host, info, dev, device, host_ptr=None):
        """A packed plan that already exists as a host blob (numpy uint8 view, e.g. of pinned memory) with its
        device copy `dev` (what agcn_store_stage_batch leaves behind); `info` is its AgcnPlanInfo."""
        self = cls.__new__(cls)
        self.host = host
        self._adopt(info, 0)
        self.dev, self.device = dev, device
        if host_ptr is not None:
            self.__dict__["_host_ptr"] = host_ptr
        return self

    def to(self, device):
        device = torch.device(device)
        if self.dev is None or self.device != device:
            self.dev = torch.from_numpy(self.host).to(device)
            self.device = device
        return self

    @property
    def host_ptr(self):
        p = self.__dict__.get("_host_ptr")
        if p is None:                    # (numpy's .ctypes accessor costs ~2 us per use: looked up once)
            p = self.__dict__["_host_ptr"] = self.host.ctypes.data
        return p

    @property
    def dev_ptr(self):
        return self.dev.data_ptr()


class TileViews(Sequence):
    """Lazy list of per-graph [n_b, n_b] views into a packed tile tensor (what the reference
    returns as a Python list of B tensors: graphconv.py:252)."""

    def __init__(self, data, plan, symmetric=False):
        self.data, self.plan = data, plan
        # True: every tile equals its transpose (set by the layers for their own Laplacian outputs when their inputs were
        # symmetric; a caller handing in res_lap tiles of its own may vouch for them) -- AgcnGraphBatch.L_prev_symmetric
        self.symmetric = bool(symmetric)
        self._verified = False      # True: the flag was derived by this package, not vouched for (layers.py: _verify_vouches)

    def __len__(self):
        return self.plan.B

    def __getitem__(self, i):
        if isinstance(i, slice):
            idx = range(*i.indices(len(self)))
            if len(idx) == len(self) and i.step in (None, 1):
                return self
            return [self[j] for j in idx]
        if i < 0:
            i += len(self)
        n, off = int(self.plan.n[i]), int(self.plan.tile_off[i])
        return self.data[off:off + n * n].view(n, n)


class RowViews(Sequence):
    """Lazy list of per-graph node-feature matrices (what the reference passes between layers as a Python list of
    B [max_atom, F] tensors).  `data` is the row matrix the kernels work on:
      * packed rows (the fast layout): item b is [n_b, F] -- unless the batch came from the reference's padded
        feed (`batch.pad_stride` = max_atom): then item b is the reference's [max_atom, F] tensor, cut out of a
        padded copy that is produced (one `agcn_rows_unpack` launch, differentiable) only when an item is first
        looked at; layers of this package take `data` and never trigger it;
      * padded rows (row_stride > 0, `GraphBatch.from_padded(..., pack=False)`): item b is a view."""

    def __init__(self, data, plan, batch=None, pad_fill=None):
        self.data, self.plan, self.batch, self.pad_fill = data, plan, batch, pad_fill
        self._padded = None

    def __len__(self):
        return self.plan.B

    def padded(self):
        """[B * max_atom, F] padded copy (reference layout) of packed rows."""
        if self._padded is None:
            from .functional import UnpackRowsFunction
            self._padded = UnpackRowsFunction.apply(self.data, self.batch, self.batch.pad_stride, self.pad_fill)
        return self._padded

    def __getitem__(self, i):
        if isinstance(i, slice):
            idx = range(*i.indices(len(self)))
            if len(idx) == len(self) and i.step in (None, 1):
                return self
            return [self[j] for j in idx]
        if i < 0:
            i += len(self)
        stride = getattr(self.batch, "pad_stride", 0) if self.plan.row_stride == 0 else 0
        if stride:
            return self.padded()[i * stride:(i + 1) * stride]
        return self.data[int(self.plan.row_off[i]):int(self.plan.row_off[i + 1])]


class GraphBatch(object):
    """A batch of graphs resident on the device in the packed layout."""

    def __init__(self, plan, X, l0_rowptr=None, l0_col=None, l0_val=None, l0_dense=None, l0_ld=0,
                 l0_symmetric=False):
        self.plan = plan
        self.X = X
        self.l0_rowptr, self.l0_col, self.l0_val = l0_rowptr, l0_col, l0_val
        self.l0_dense, self.l0_ld = l0_dense, int(l0_ld)
        # True only when every original Laplacian equals its transpose (checked on the host at
        # pack time); the slab-parallel kernels rely on it
        self.l0_symmetric = bool(l0_symmetric)
        # the flag was COMPUTED by this package (from_arrays / from_padded) rather than vouched for by the caller of this
        # constructor: the layers verify a vouch on the device once (layers.py: _verify_vouches)
        self._l0_verified = False
        # > 0: the batch came from the reference's padded feed with this max_atom; per-graph views of layer outputs
        # are then padded to it on demand (RowViews)
        self.pad_stride = 0

    @property
    def device(self):
        return self.X.device

    @classmethod
    def from_graphs(cls, graphs, device, dtype=torch.float32):
        """Packed batch from `Graph`/`MolGraph`-like objects (node_features [n,F] array and a
        scipy CSR `Laplacian`).  This is the ingest that replaces pad_data2sparse /
        pad_Lap2sparse / batch_to_feed_dict (graph_topology.py:84-135)."""
        graphs = list(graphs)
        n = np.array([g.node_features.shape[0] for g in graphs], dtype=np.int32)
        for g in graphs:
            if getattr(g, "Laplacian", None) is None:
                raise ValueError("graph without a Laplacian (n <= 3 graphs are dropped by the reference loader)")
        X = np.concatenate([np.asarray(g.node_features, dtype=np.float32) for g in graphs], axis=0)
        csr = [(g.Laplacian.indptr, g.Laplacian.indices, g.Laplacian.data) for g in graphs]
        return cls.from_arrays(n, X, csr, device)

    @classmethod
    def from_arrays(cls, n, X, csr_list, device):
        """n int32 [B]; X float32 [N1, F] packed; csr_list = [(indptr, indices, data)] per graph."""
        plan = Plan(n, 0, device)
        nnz = np.array([len(c[1]) for c in csr_list], dtype=np.int64)
        rowptr = np.zeros(plan.rows + 1, dtype=np.int32)
        base = np.concatenate([[0], np.cumsum(nnz)])
        for b, (ip, _, _) in enumerate(csr_list):
            r0 = int(plan.row_off[b])
            rowptr[r0 + 1:r0 + int(n[b]) + 1] = np.asarray(ip[1:], dtype=np.int64) + base[b]
        col = np.concatenate([np.asarray(c[1], dtype=np.int32) for c in csr_list]) if len(csr_list) else np.zeros(0, np.int32)
        val = np.concatenate([np.asarray(c[2]).astype(np.float32) for c in csr_list]) if len(csr_list) else np.zeros(0, np.float32)
        dev = torch.device(device)
        sym = bool(csr_symmetry_flags(plan.row_off, rowptr, col, val).all())
        if col.size == 0:       # all-zero Laplacians: keep the device pointers non-null (never dereferenced)
            col, val = np.zeros(1, np.int32), np.zeros(1, np.float32)
        gb = cls(plan, torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(dev),
                 torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev),
                 l0_symmetric=sym)
        gb._l0_verified = True
        return gb

    @classmethod
    def from_padded(cls, node_features, laplacians, data_slice, device=None, pack=None):
        """Batch from the REFERENCE's feed layout: node_features / laplacians are lists of B [max_atom, F] /
        [max_atom, max_atom] tensors (or one 3-D tensor each; laplacians may be None for layers that do not need
        them), data_slice is the int32 [B, 2] array with rows [n, -1] (graph_topology.py:90).
        pack=True (default on the GPU): the real rows are gathered once into PACKED rows (`agcn_rows_pack`,
        differentiable), the dense Laplacians are used in place, and everything downstream runs on the fast packed
        paths; per-graph outputs are padded back to [max_atom, F] only when looked at (RowViews).
        pack=False: rows stay padded (row_stride = max_atom), outputs alias the reference's per-graph tensors."""
        X3 = node_features if torch.is_tensor(node_features) else torch.stack(list(node_features))
        ds = data_slice.detach().cpu().numpy() if torch.is_tensor(data_slice) else np.asarray(data_slice)
        n = np.ascontiguousarray(ds[:, 0], dtype=np.int32)
        B, max_atom, F = X3.shape
        device = X3.device if device is None else torch.device(device)
        L = None
        if laplacians is not None:
            L3 = laplacians if torch.is_tensor(laplacians) else torch.stack(list(laplacians))
            if L3.shape != (B, max_atom, max_atom):
                raise ValueError("inconsistent padded batch shapes")
            L = L3.to(device=device, dtype=torch.float32).contiguous()
        if n.shape[0] != B:
            raise ValueError("inconsistent padded batch shapes")
        sym = bool(torch.equal(L, L.transpose(1, 2))) if L is not None else False
        X = X3.to(device=device, dtype=torch.float32).reshape(B * max_atom, F)
        if pack is None:
            pack = PACK_REFERENCE_FEED and device.type == "cuda"
        if not pack:
            plan = Plan(n, max_atom, device)
            gb = cls(plan, X, l0_dense=L, l0_ld=max_atom if L is not None else 0, l0_symmetric=sym)
            gb._l0_verified = True
            return gb
        from .functional import PackRowsFunction
        gb = cls(Plan(n, 0, device), None, l0_dense=L, l0_ld=max_atom if L is not None else 0, l0_symmetric=sym)
        gb.pad_stride = int(max_atom)
        gb._l0_verified = True
        gb.X = PackRowsFunction.apply(X, gb, max_atom)
        return gb

    def rows(self, data):
        return RowViews(data, self.plan)

    def tiles(self, data):
        return TileViews(data, self.plan)
