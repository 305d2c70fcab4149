"""Dataset-level graph store and batch iterator: the ingest in front of the layer path.

The reference hands lists of `MolGraph` objects out of `DiskDataset.iterbatches`
(utils/dataset/diskdataset.py:239-295) and rebuilds the padded / densified feed from them in Python
every step (models/tf_modules/graph_topology.py:84-135).  Here the dataset is laid out ONCE
(concatenated node features, one CSR over all rows with float32 values, per-graph symmetry flags) and
a batch is a native gather of contiguous segments (`agcn_store_pack_batch`) into pinned staging
buffers + one asynchronous upload + `agcn_plan_build`.
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi
from .batch import GraphBatch, Plan, csr_symmetry_flags


class GraphStore(object):
    def __init__(self, n, X, rowptr, col, val, symmetric=None, y=None, w=None, ids=None):
        self.n = np.ascontiguousarray(n, dtype=np.int32)
        G = int(self.n.size)
        self.row_off = np.zeros(G + 1, dtype=np.int64)
        np.cumsum(self.n, out=self.row_off[1:])
        self.X = np.ascontiguousarray(X, dtype=np.float32)
        if self.X.ndim != 2 or self.X.shape[0] != int(self.row_off[-1]):
            raise ValueError("X must be [sum(n), F]")
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float32)
        if self.rowptr.size != self.X.shape[0] + 1 or self.col.size != int(self.rowptr[-1]) or \
                self.val.size != self.col.size:
            raise ValueError("inconsistent CSR arrays")
        self.symmetric = None if symmetric is None else np.ascontiguousarray(symmetric, dtype=np.uint8)
        self.y = None if y is None else np.asarray(y)
        self.w = None if w is None else np.asarray(w)
        self.ids = np.arange(G) if ids is None else np.asarray(ids)
        self.F = int(self.X.shape[1])
        self._c = _ffi.GraphStoreC(G, self.F, 0, self.n.ctypes.data, self.row_off.ctypes.data, self.X.ctypes.data,
                                   self.rowptr.ctypes.data, self.col.ctypes.data, self.val.ctypes.data,
                                   None if self.symmetric is None else self.symmetric.ctypes.data)
        self._stage = {}

    def __len__(self):
        return int(self.n.size)

    # ---- construction ------------------------------------------------------------------------------
    @classmethod
    def from_graphs(cls, graphs, y=None, w=None, ids=None):
        """`Graph` / `MolGraph`-like objects (node_features [n, F], scipy CSR `Laplacian`); graphs without a
        Laplacian (n <= 3, graph_structure.py:51-55) are rejected like the reference loader drops them."""
        graphs = list(graphs)
        for g in graphs:
            if getattr(g, "Laplacian", None) is None:
                raise ValueError("graph without a Laplacian (n <= 3 graphs are dropped by the reference loader)")
        X = [np.asarray(g.node_features, dtype=np.float32) for g in graphs]
        csr = [(g.Laplacian.indptr, g.Laplacian.indices, g.Laplacian.data) for g in graphs]
        return cls.from_arrays([x.shape[0] for x in X], X, csr, y=y, w=w, ids=ids)

    @classmethod
    def from_arrays(cls, n, X_list, csr_list, y=None, w=None, ids=None):
        n = np.asarray(n, dtype=np.int32)
        X = np.concatenate([np.asarray(x, dtype=np.float32) for x in X_list], 0) if not isinstance(X_list, np.ndarray) \
            else X_list
        nnz = np.array([len(c[1]) for c in csr_list], dtype=np.int64)
        base = np.concatenate([[0], np.cumsum(nnz)])
        rowptr = np.zeros(int(n.sum()) + 1, dtype=np.int64)
        r = 0
        for g, (ip, _, _) in enumerate(csr_list):
            rowptr[r + 1:r + int(n[g]) + 1] = np.asarray(ip[1:], dtype=np.int64) + base[g]
            r += int(n[g])
        col = np.concatenate([np.asarray(c[1], dtype=np.int32) for c in csr_list])
        val = np.concatenate([np.asarray(c[2]).astype(np.float32) for c in csr_list])
        row_off = np.concatenate([[0], np.cumsum(n, dtype=np.int64)])
        sym = csr_symmetry_flags(row_off, rowptr, col, val)
        return cls(n, X, rowptr, col, val, sym, y=y, w=w, ids=ids)

    # ---- batches -----------------------------------------------------------------------------------
    def _buffers(self, slot, rows, nnz, B, pin):
        key = (slot, pin)
        st = self._stage.get(key)
        if st is None or st["rows"] < rows or st["nnz"] < nnz or st["B"] < B:
            cap_r, cap_z, cap_b = int(rows * 1.25) + 64, int(nnz * 1.25) + 64, max(B, 1)
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=pin)  # noqa: E731
            st = {"rows": cap_r, "nnz": cap_z, "B": cap_b, "X": mk((cap_r, self.F), torch.float32),
                  "rowptr": mk((cap_r + 1,), torch.int32), "col": mk((cap_z,), torch.int32),
                  "val": mk((cap_z,), torch.float32), "n": np.zeros(cap_b, dtype=np.int32)}
            self._stage[key] = st
        return st

    def _pack(self, graph_ids, slot, pin, n_threads):
        """Host half of a batch: native gather into staging slot `slot` + plan build (no device work)."""
        lib = _ffi.lib()
        ids = np.ascontiguousarray(graph_ids, dtype=np.int64)
        B = int(ids.size)
        rows, nnz, mx = C.c_int64(), C.c_int64(), C.c_int32()
        _ffi.check(lib.agcn_store_batch_sizes(C.byref(self._c), ids.ctypes.data, B, C.byref(rows), C.byref(nnz),
                                              C.byref(mx)))
        rows, nnz = rows.value, nnz.value
        st = self._buffers(slot, rows, nnz, B, pin)
        if st.get("done") is not None:      # the uploads that last read this staging slot must have completed:
            st["done"].synchronize()        # a fast consumer runs many batches ahead of the GPU
            st["done"] = None
        sym = C.c_int32()
        _ffi.check(lib.agcn_store_pack_batch(C.byref(self._c), ids.ctypes.data, B, st["n"].ctypes.data,
                                             st["X"].data_ptr(), self.F, st["rowptr"].data_ptr(), st["col"].data_ptr(),
                                             st["val"].data_ptr(), C.byref(sym), int(n_threads)))
        plan = Plan(st["n"][:B].copy(), 0, None)
        if pin:      # the plan blob travels from pinned memory too, so that no upload blocks on the compute stream
            if st.get("plan") is None or st["plan"].numel() < plan.host.size:
                st["plan"] = torch.empty(int(plan.host.size * 1.25) + 256, dtype=torch.uint8, pin_memory=True)
                st["plan_np"] = st["plan"].numpy()
            # plain host memcpy through the numpy view (Tensor.copy_ into pinned memory synchronises the device)
            st["plan_np"][:plan.host.size] = plan.host
        return {"st": st, "plan": plan, "rows": rows, "nnz": nnz, "sym": bool(sym.value), "pin": pin}

    @staticmethod
    def _upload(pk, dev):
        st, plan, rows, nnz, pin = pk["st"], pk["plan"], pk["rows"], pk["nnz"], pk["pin"]
        up = lambda t, k: t[:k].to(dev, non_blocking=True) if pin else t[:k].clone()  # noqa: E731
        if pin:
            plan.dev, plan.device = st["plan"][:plan.host.size].to(dev, non_blocking=True), dev
        else:
            plan.to(dev)
        gb = GraphBatch(plan, up(st["X"], rows), up(st["rowptr"], rows + 1), up(st["col"], nnz), up(st["val"], nnz),
                        l0_symmetric=pk["sym"])
        if pin:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            st["done"] = ev
        return gb

    def batch(self, graph_ids, device, slot=0, n_threads=1):
        """Packed `GraphBatch` of the graphs `graph_ids` on `device`.  Uploads are asynchronous on the current
        stream from pinned staging buffers; `slot` selects one of several staging sets so that a batch can be
        packed while the previous one is still being copied."""
        dev = torch.device(device)
        return self._upload(self._pack(graph_ids, slot, dev.type == "cuda", n_threads), dev)

    def to_device(self, device, batch_cap, slots=4, lanes=1, dtype=torch.float32, stage_dtype=torch.float32):
        """The store resident in HBM (`ResidentStore`): arrays uploaded once, batches gathered on the device.
        dtype=torch.bfloat16 keeps the node features in bf16 at rest (rounded once, nearest-even); stage_dtype=torch.bfloat16
        also hands them to the layer in bf16 (its kernels read and write bf16 rows themselves)."""
        return ResidentStore(self, device, batch_cap, slots, lanes, dtype, stage_dtype)

    def _selections(self, batch_size, deterministic, rng):
        N = len(self)
        perm = np.arange(N) if deterministic else (rng if rng is not None else np.random).permutation(N)
        pts = np.linspace(0, N, int(np.ceil(float(N) / batch_size)) + 1, dtype=int)
        return [perm[pts[j]:pts[j + 1]] for j in range(len(pts) - 1)]

    def iterbatches(self, batch_size, device, deterministic=False, rng=None, n_threads=1, prefetch=0):
        """Same batching rule as DiskDataset.iterbatches (diskdataset.py:263-283): a permutation of the samples
        (identity when `deterministic`), cut at np.linspace(0, N, ceil(N / batch_size) + 1, dtype=int).
        Yields (GraphBatch, y_batch, w_batch, ids_batch).  With `prefetch` > 0 a worker thread packs the next
        batches (the native gather releases the GIL) while the caller's kernels run; a staging slot is reused
        only after the uploads that read it have completed.  (Measured on MI355X, 256-graph HIV-shaped batches:
        ingest alone 2.2 M graphs/s without and 2.6 M graphs/s with prefetch; with an eager autograd layer step
        per batch the Python side is the bottleneck either way, hence off by default.)"""
        dev = torch.device(device)
        pin = dev.type == "cuda"
        sels = self._selections(batch_size, deterministic, rng)
        extra = lambda sel: (None if self.y is None else self.y[sel], None if self.w is None else self.w[sel],  # noqa: E731
                             self.ids[sel])
        if prefetch <= 0:
            for j, sel in enumerate(sels):
                yield (self.batch(sel, dev, slot=j & 1, n_threads=n_threads),) + extra(sel)
            return
        import queue
        import threading
        nslots = prefetch + 2
        q = queue.Queue(maxsize=prefetch)
        stop = threading.Event()

        def worker():
            try:
                for j, sel in enumerate(sels):
                    if stop.is_set():
                        break
                    k = j % nslots
                    q.put((j, sel, self._pack(sel, ("it", k), pin, n_threads)))   # (_pack waits for the slot's uploads)
                q.put(None)
            except BaseException as e:          # surfaced in the consumer
                q.put(e)

        th = threading.Thread(target=worker, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                j, sel, pk = item
                yield (self._upload(pk, dev),) + extra(sel)
        finally:
            stop.set()
            while th.is_alive():            # unblock a worker waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.01)


class ResidentStore(object):
    """A `GraphStore` whose arrays live in HBM (288 GB hold whole datasets: the HIV-sized store is 0.3 GB).

    The reference reads its shards from disk and rebuilds the padded feed on the host every step
    (diskdataset.py:239-295 -> graph_topology.py:84-135).  Here X / CSR of ALL graphs are uploaded once; a batch is
    ONE C-ABI call (`agcn_store_stage_batch`): plan + gather table built into a pinned blob (tens of KB), one
    asynchronous copy of that blob, one gather launch into capacity-sized device buffers.  No per-batch feature
    bytes cross PCIe and no per-batch device memory is allocated.

    `stage(ids)` returns a `GraphBatch` whose tensors are VIEWS of the capacity buffers and whose plan lives in one
    of `slots` pinned blobs: it is valid until the next `stage()` on the same lane on the device side (stream
    order) and until `slots` further batches on the host side.  `lanes` > 1 keeps that many independent sets of
    capacity buffers, so that batches staged on different streams can be in flight together (`SgcLLEpochLanes`)."""

    def __init__(self, store, device, batch_cap, slots=4, lanes=1, dtype=torch.float32, stage_dtype=torch.float32):
        """`dtype`: the features at rest in HBM; `stage_dtype`: what a staged batch holds -- float32 (bf16 at rest is widened by
        the gather, the layer runs with fp32 features) or bfloat16 (the rows stay bf16, half the gather's bytes in and out: for
        layer calls with bf16 features, `SgcLLEpochStep(..., feat_dtype=torch.bfloat16)` / an SGC_LL called on a bf16 tensor)."""
        dev = torch.device(device)
        if dtype not in (torch.float32, torch.bfloat16) or stage_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("node features are kept in float32 or bfloat16")
        if dev.type != "cuda":
            raise _ffi.AgcnError("a resident store lives on the GPU: there is no CPU fallback")
        self.store, self.device = store, dev
        self.batch_cap = int(batch_cap)
        self.F = store.F
        up = lambda a: torch.from_numpy(a).to(dev)   # noqa: E731
        # bf16 at rest ("bf16 features + fp32 Laplacian"): rounded once here, widened by the gather;
        # every batch then carries float32(bf16(x)) and the layer computes in fp32
        self.X, self.rowptr = up(store.X).to(dtype), up(store.rowptr)
        self.dtype, self.stage_dtype = dtype, stage_dtype
        self.col = up(store.col if store.col.size else np.zeros(1, np.int32))
        self.val = up(store.val if store.val.size else np.zeros(1, np.float32))
        # capacity of any batch of <= batch_cap graphs: the batch_cap largest graphs / Laplacians of the store
        k = min(self.batch_cap, len(store))
        nnz_g = store.rowptr[store.row_off[1:]] - store.rowptr[store.row_off[:-1]]
        self.rows_cap = int(np.sort(store.n)[-k:].sum())
        self.nnz_cap = max(1, int(np.sort(nnz_g)[-k:].sum()))
        self.tiles_cap = int(((np.sort(store.n.astype(np.int64))[-k:] ** 2 + 3) // 4 * 4).sum())
        lib = _ffi.lib()
        self.stage_cap = int(lib.agcn_store_stage_bytes(k, max(self.rows_cap, k)))
        i32 = dict(dtype=torch.int32, device=dev)
        self.lanes = []
        for _ in range(max(1, int(lanes))):
            ln = {"X": torch.zeros((self.rows_cap, self.F), dtype=stage_dtype, device=dev),
                  "rowptr": torch.zeros(self.rows_cap + 1, **i32), "col": torch.zeros(self.nnz_cap, **i32),
                  "val": torch.zeros(self.nnz_cap, dtype=torch.float32, device=dev),
                  "stage": torch.zeros(self.stage_cap, dtype=torch.uint8, device=dev)}
            ln["args"] = (ln["stage"].data_ptr(), self.stage_cap, ln["X"].data_ptr(), self.F, self.rows_cap,
                          ln["rowptr"].data_ptr(), ln["col"].data_ptr(), ln["val"].data_ptr(), self.nnz_cap)
            self.lanes.append(ln)
        self._slots = []
        for _ in range(max(2, int(slots)) * len(self.lanes)):
            t = torch.zeros(self.stage_cap, dtype=torch.uint8, pin_memory=True)
            self._slots.append({"t": t, "np": t.numpy(), "ptr": t.data_ptr(), "done": None})
        self._k = 0
        self._lane_k = [0] * len(self.lanes)      # (engine.SgcLLEpochStep.step_ids: every lane rotates through its own blobs)
        h = store._c
        self._c = _ffi.ResidentStoreC(h, self.X.data_ptr(), self.rowptr.data_ptr(), self.col.data_ptr(),
                                      self.val.data_ptr(), 1 if dtype == torch.bfloat16 else 0,
                                      1 if stage_dtype == torch.bfloat16 else 0)
        self._info = _ffi.StagedBatch()

    def __len__(self):
        return len(self.store)

    def stage(self, graph_ids, lane=0, stream=None):
        """Batch `graph_ids` gathered into lane `lane`'s buffers on `stream` (a torch stream; default: the current
        one)."""
        ids = np.ascontiguousarray(graph_ids, dtype=np.int64)
        B = int(ids.size)
        if B > self.batch_cap:
            raise ValueError("batch of %d graphs exceeds the capacity %d" % (B, self.batch_cap))
        sl = self._slots[self._k]
        self._k = (self._k + 1) % len(self._slots)
        if sl["done"] is not None:              # the copy that last read this pinned blob must have executed
            sl["done"].synchronize()
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        info = self._info
        ln = self.lanes[lane]
        a = ln["args"]
        _ffi.check(_ffi.lib().agcn_store_stage_batch(C.byref(self._c), ids.ctypes.data, B, sl["ptr"], a[0], a[1], a[2],
                                                     a[3], a[4], a[5], a[6], a[7], a[8], C.byref(info),
                                                     stream.cuda_stream))
        if sl["done"] is None:
            sl["done"] = torch.cuda.Event()
        sl["done"].record(stream)
        pinfo = _ffi.PlanInfo.from_buffer_copy(info.plan)
        plan = Plan.from_blob(sl["np"][:pinfo.plan_bytes], pinfo, ln["stage"], self.device, host_ptr=sl["ptr"])
        rows, nnz = info.rows, info.nnz
        return GraphBatch(plan, ln["X"][:rows], ln["rowptr"][:rows + 1], ln["col"][:max(nnz, 1)],
                          ln["val"][:max(nnz, 1)], l0_symmetric=bool(info.symmetric))

    def iterbatches(self, batch_size, deterministic=False, rng=None):
        """Same batching rule as `GraphStore.iterbatches` / DiskDataset.iterbatches (diskdataset.py:263-283).
        Yields (GraphBatch, y_batch, w_batch, ids_batch); the batch is valid until the next one is staged."""
        st = self.store
        for sel in st._selections(batch_size, deterministic, rng):
            yield (self.stage(sel), None if st.y is None else st.y[sel], None if st.w is None else st.w[sel],
                   st.ids[sel])
