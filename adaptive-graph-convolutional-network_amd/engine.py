"""Static-buffer execution of one layer's forward + backward for a fixed batch: every
device buffer and every ctypes argument block is created once, so a step is nothing but the
C-ABI launches (no allocation, no marshalling) and can be captured into a HIP graph.

All parameter gradients of the layer live in ONE flat fp32 bucket
([dweight | dbias | dM_L | dalpha | dbeta]) so that data-parallel training needs a single
RCCL all-reduce per step (see dist.py)."""
import ctypes as C

import numpy as np

import torch

from . import _ffi
from . import functional as _fn
from .functional import _Cfg, _desc, _gbatch, _ptr


def _reserve_timeouts(lib, reserve, nbytes, stream, device):
    """(count, message) of agcn_sgc_ll_reserve_status: waits for `stream`; count = cooperative waits that ran out."""
    n = C.c_int32(0)
    with torch.cuda.device(device):
        rc = lib.agcn_sgc_ll_reserve_status(reserve.data_ptr(), int(nbytes), stream, C.byref(n))
    if rc != 0 and n.value == 0:
        _ffi.check(rc)
    return n.value, (lib.agcn_last_error().decode() if rc != 0 else "")


def _no_member_desc(d):
    return _ffi.LayerDesc(d.Fin, d.Fout, d.K, d.mode, d.variant, d.activation, d.ldx, d.ldy, d.feat_dtype,
                          _fn.no_member_flags(d.flags))


class SgcLLStep(object):
    def __init__(self, layer, batch, X=None, L_prev=None, dY=None, dL_sum=None, L_prev_symmetric=False, flags=None):
        """`flags`: per-call dispatch flags of every call of this step (default: the constructing thread's current ones,
        `agcn_amd.dispatch`)."""
        lib = _ffi.lib()
        self.lib, self.layer, self.batch = lib, layer, batch
        plan = batch.plan
        X = batch.X if X is None else X
        if not X.is_cuda:
            raise _ffi.AgcnError("node features must live on the GPU: there is no CPU fallback")
        dev = X.device
        layer.build(dev)
        self.device = dev
        cfg = _Cfg(layer.n_atom_feature, layer.nb_filter, layer.K, layer.laplacian_mode, layer._variant,
                   "relu" if layer.activation == "relu" else "linear")
        self.cfg = cfg
        reslap = cfg.variant == "SGC_LL_Reslap"
        paper = cfg.mode == "paper"
        self.X = X.detach().contiguous()
        self.L_prev = L_prev if reslap else None
        v = layer.vars
        ldx, ldy = self.X.shape[1], cfg.Fout
        f32 = dict(dtype=torch.float32, device=dev)
        # Every buffer of the step is carved out of ONE allocation (256-byte aligned pieces): the kernels then touch
        # a handful of contiguous 2 MB pages instead of a dozen scattered ones (address-translation misses at the
        # start of each kernel were a measurable part of its first load phase)
        self.desc = _desc(cfg, ldx, ldy, flags)
        self.recovered = 0
        self.rbytes = lib.agcn_sgc_ll_reserve_bytes(C.byref(self.desc), plan.host_ptr)
        self.wf = lib.agcn_sgc_ll_forward_workspace_bytes(C.byref(self.desc), plan.host_ptr)
        self.wb = lib.agcn_sgc_ll_backward_workspace_bytes(C.byref(self.desc), plan.host_ptr)
        has_prev = self.L_prev is not None
        nb = cfg.Fin * cfg.K * cfg.Fout + cfg.Fout + (cfg.Fin * cfg.Fin if paper else 0) + 1 + (1 if has_prev else 0)
        pieces = [4 * plan.rows * ldy, 4 * plan.tile_elems, 4 * plan.tile_elems, 4 * plan.tile_elems if reslap else 0,
                  4 * plan.rows * ldy, 4 * self.X.numel(), 4 * self.X.numel(), 4 * plan.tile_elems if has_prev else 0,
                  4 * nb, self.rbytes, max(self.wf, self.wb)]
        al = lambda v: (int(v) + 255) // 256 * 256   # noqa: E731
        arena = torch.zeros(sum(al(v) for v in pieces) + 256, dtype=torch.uint8, device=dev)
        self._arena, self._arena_off = arena, 0

        def take(nbytes, shape=None, dtype=torch.float32):
            o = self._arena_off
            self._arena_off += al(nbytes)
            t = arena[o:o + int(nbytes)]
            return t if dtype == torch.uint8 else t.view(dtype).view(shape if shape is not None else -1)

        self.Y = take(4 * plan.rows * ldy, (plan.rows, ldy))
        self.res_L = take(4 * plan.tile_elems)
        self.res_W = take(4 * plan.tile_elems)
        self.L_sum = take(4 * plan.tile_elems) if reslap else None   # SGC_LL has three outputs
        if dY is not None:
            self.dY = take(4 * plan.rows * ldy, (plan.rows, ldy))
            self.dY.copy_(dY)
        else:
            self.dY = take(4 * plan.rows * ldy, (plan.rows, ldy))
        self.dL_sum = dL_sum
        Xa = take(4 * self.X.numel(), tuple(self.X.shape))
        Xa.copy_(self.X)
        self.X = Xa
        self.dX = take(4 * self.X.numel(), tuple(self.X.shape))
        self.dL_prev = take(4 * plan.tile_elems) if has_prev else None
        # flat gradient bucket
        sizes = [("weight", cfg.Fin * cfg.K * cfg.Fout), ("bias", cfg.Fout)]
        if paper:
            sizes.append(("M_L", cfg.Fin * cfg.Fin))
        sizes.append(("alpha", 1))
        if has_prev:
            sizes.append(("beta", 1))
        assert sum(s for _, s in sizes) == nb
        self.bucket = take(4 * nb)
        self.grads, o = {}, 0
        for k, s in sizes:
            self.grads[k] = self.bucket[o:o + s].view(v[k].shape)
            o += s
        self.reserve = take(self.rbytes, dtype=torch.uint8)
        self.ws = take(max(self.wf, self.wb), dtype=torch.uint8)
        self.gb = _gbatch(batch, self.X, self.L_prev, L_prev_symmetric)
        self.pr = _ffi.Params(v["weight"].data_ptr(), v["bias"].data_ptr(), v["M_L"].data_ptr(),
                              v["alpha"].data_ptr(), _ptr(v.get("beta")))
        self.out = _ffi.Outputs(self.Y.data_ptr(), self.res_L.data_ptr(), self.res_W.data_ptr(), _ptr(self.L_sum))
        g = self.grads
        self.gr = _ffi.Grads(self.dX.data_ptr(), g["weight"].data_ptr(), g["bias"].data_ptr(),
                             _ptr(g.get("M_L")), g["alpha"].data_ptr(), _ptr(g.get("beta")), _ptr(self.dL_prev))
        self._graph = None
        self._bwd = {3: lib.agcn_sgc_ll_backward, 1: lib.agcn_sgc_ll_backward_nodes,
                     2: lib.agcn_sgc_ll_backward_params}

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def forward(self):
        _ffi.check(self.lib.agcn_sgc_ll_forward(
            C.byref(self.desc), C.byref(self.gb), C.byref(self.pr), C.byref(self.out),
            self.reserve.data_ptr(), self.rbytes, self.ws.data_ptr(), self.ws.numel(), self._stream()))

    def backward(self, stages=3):
        _ffi.check(self._bwd[stages](
            C.byref(self.desc), C.byref(self.gb), C.byref(self.pr), C.byref(self.out),
            self.dY.data_ptr(), _ptr(self.dL_sum), C.byref(self.gr), self.reserve.data_ptr(), self.rbytes,
            self.ws.data_ptr(), self.ws.numel(), self._stream()))

    def step(self):
        self.forward()
        self.backward()

    def capture(self, extra=None, steps=1):
        """Capture forward + backward into a HIP graph (torch.cuda.CUDAGraph on ROCm).  `extra`, if given, is
        called behind the step inside the capture (e.g. the RCCL all-reduce of `self.bucket`, so that a
        data-parallel step is still one graph replay).  `steps` > 1 captures that many consecutive steps into the ONE
        graph: kernels inside a graph follow each other without a gap, while two graph launches are ~5 us apart on
        MI355X (end-of-graph release + start-of-graph acquire), 9 % of a 256-graph step."""
        self.step()                      # warm-up outside capture (sets function attributes)
        if extra is not None:
            extra()
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(max(1, int(steps))):
                self.step()
                if extra is not None:
                    extra()
        self._graph = g
        self.steps_per_replay = max(1, int(steps))
        return g

    def kernel_names(self):
        """Names of the kernels one forward / per-graph backward / parameter-gradient call launches on this
        batch (three strings), found by running each once under the library's launch profiler."""
        names = []
        # (the parameter-gradient kernel of a WHOLE backward call: a call of that stage alone may take another kernel -- the
        #  tagged hand-off needs the per-graph kernel of the same call in front of it)
        for fn in (self.forward, lambda: self.backward(1), lambda: self.backward(3)):
            torch.cuda.synchronize(self.device)
            self.lib.agcn_profile_enable(1)
            try:
                fn()
                prof = _ffi.profile_collect()
            finally:
                self.lib.agcn_profile_enable(0)
            names.append([k for k, (ms, c) in prof.items() if c > 0])
        return ["+".join(names[0]), "+".join(names[1]), "+".join(k for k in names[2] if k not in names[1])]

    def replay(self):
        self._graph.replay()

    def check(self, recover=False):
        """Waits for the current stream.  If a wait between the workgroups that share a large graph ran out in any step of this
        object since the last check (include/agcn.h: agcn_sgc_ll_reserve_status; the step's results are NaN): raises AgcnError,
        or -- `recover=True` -- RE-ISSUES the step on the non-member dispatch (fresh launches of the kernels in which no
        workgroup waits for another, per-call flags; same inputs, same output buffers), counts it in `self.recovered` /
        `agcn_amd.recovered_calls` and returns the number of waits that had run out.  Returns 0 when nothing happened.
        Call it where the host synchronises anyway."""
        n, msg = _reserve_timeouts(self.lib, self.reserve, self.rbytes, self._stream(), self.device)
        if n == 0:
            return 0
        self.reserve[:_ffi.RESERVE_HEADER_BYTES].zero_()     # re-arms the header (exchange words + the sticky count)
        if not recover:
            raise _ffi.AgcnError("agcn error -3: %s" % msg)
        self._rerun_without_members()
        return n

    def _rerun_without_members(self):
        lib, plan = self.lib, self.batch.plan
        desc = _no_member_desc(self.desc)
        rb = lib.agcn_sgc_ll_reserve_bytes(C.byref(desc), plan.host_ptr)
        wb = max(lib.agcn_sgc_ll_forward_workspace_bytes(C.byref(desc), plan.host_ptr),
                 lib.agcn_sgc_ll_backward_workspace_bytes(C.byref(desc), plan.host_ptr))
        with torch.cuda.device(self.device):
            reserve = torch.zeros(int(rb), dtype=torch.uint8, device=self.device)
            ws = torch.empty(int(wb), dtype=torch.uint8, device=self.device)
            st = self._stream()
            _ffi.check(lib.agcn_sgc_ll_forward(C.byref(desc), C.byref(self.gb), C.byref(self.pr), C.byref(self.out),
                                               reserve.data_ptr(), rb, ws.data_ptr(), wb, st))
            _ffi.check(lib.agcn_sgc_ll_backward(C.byref(desc), C.byref(self.gb), C.byref(self.pr), C.byref(self.out),
                                                self.dY.data_ptr(), _ptr(self.dL_sum), C.byref(self.gr), reserve.data_ptr(), rb,
                                                ws.data_ptr(), wb, st))
            n, msg = _reserve_timeouts(lib, reserve, rb, st, self.device)
        if n:
            raise _ffi.AgcnError("a cooperative wait ran out on the non-member dispatch (which has none): %s" % msg)
        self.recovered += 1
        _fn.recovered_calls[0] += 1


class SgcLLEpochStep(object):
    """Forward + backward of one SGC_LL layer over a STREAM of batches (an epoch): every device buffer is allocated
    once at a capacity and reused, so a batch costs two C-ABI calls and a handful of ctypes fields -- no tensor
    allocation, no autograd graph (the eager autograd path spends ~200 us of Python per batch).
    Batches may have any shape; the library picks the kernel path per batch from its plan as usual.  Results of the
    current batch are views into the capacity buffers (`Y`, `res_L`, `res_W`, `dX`, `bucket` / `grads`), valid until
    the next `step()`.

        ep = SgcLLEpochStep(layer, rows_cap, tiles_cap, batch_cap)
        for gb, _, _, _ in store.iterbatches(256, dev, prefetch=2):
            ep.step(gb, dY=None)           # dY: [rows, Fout] upstream gradient (default: the static ones-filled one)
    """

    def __init__(self, layer, rows_cap, tiles_cap, batch_cap, device=None, flags=None, feat_dtype=torch.float32):
        """`feat_dtype=torch.bfloat16`: X (the staged rows), Y, dY and dX are bf16 tensors read and written by the kernels
        themselves (AgcnLayerDesc.feat_dtype; tiles, Laplacian, products and parameter gradients fp32) -- BASELINE.json's
        "bf16 features + fp32 Laplacian"; the resident store must stage bf16 (`to_device(..., stage_dtype=torch.bfloat16)`)."""
        lib = _ffi.lib()
        self.lib, self.layer = lib, layer
        dev = torch.device(device if device is not None else layer._device)
        layer.build(dev)
        self.device = dev
        cfg = _Cfg(layer.n_atom_feature, layer.nb_filter, layer.K, layer.laplacian_mode, layer._variant,
                   "relu" if layer.activation == "relu" else "linear", bf16=(feat_dtype == torch.bfloat16))
        if feat_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("features are float32 or bfloat16")
        self.feat_dtype = feat_dtype
        if cfg.variant != "SGC_LL":
            raise ValueError("SgcLLEpochStep drives SGC_LL layers")
        self.cfg = cfg
        self.paper = cfg.mode == "paper"
        self.rows_cap, self.tiles_cap, self.batch_cap = int(rows_cap), int(tiles_cap), int(batch_cap)
        f32 = dict(dtype=torch.float32, device=dev)
        fdt = dict(dtype=feat_dtype, device=dev)
        self._Y = torch.zeros((self.rows_cap, cfg.Fout), **fdt)
        self._res_L = torch.zeros(self.tiles_cap, **f32)
        self._res_W = torch.zeros(self.tiles_cap, **f32)
        self._dX = torch.zeros((self.rows_cap, cfg.Fin), **fdt)
        self._dY = torch.ones((self.rows_cap, cfg.Fout), **fdt)
        v = layer.vars
        sizes = [("weight", cfg.Fin * cfg.K * cfg.Fout), ("bias", cfg.Fout)]
        if self.paper:
            sizes.append(("M_L", cfg.Fin * cfg.Fin))
        sizes.append(("alpha", 1))
        self.bucket = torch.zeros(sum(n for _, n in sizes), **f32)
        self.grads, o = {}, 0
        for k, n in sizes:
            self.grads[k] = self.bucket[o:o + n].view(v[k].shape)
            o += n
        self.desc = _desc(cfg, cfg.Fin, cfg.Fout, flags)
        self.recovered = 0
        self._last_ids = None
        self.pr = _ffi.Params(v["weight"].data_ptr(), v["bias"].data_ptr(), v["M_L"].data_ptr(),
                              v["alpha"].data_ptr(), None)
        self.out = _ffi.Outputs(self._Y.data_ptr(), self._res_L.data_ptr(), self._res_W.data_ptr(), None)
        g = self.grads
        self.gr = _ffi.Grads(self._dX.data_ptr(), g["weight"].data_ptr(), g["bias"].data_ptr(), _ptr(g.get("M_L")),
                             g["alpha"].data_ptr(), None, None)
        self._reserve = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
        self._ws = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
        self.plan = None
        self._rows = self._tile_elems = 0
        self._ebufs, self._einfo, self._eneeds = {}, _ffi.StagedBatch(), _ffi.EpochNeeds()     # (step_ids)

    def _fit(self, plan):
        if plan.rows > self.rows_cap or plan.tile_elems > self.tiles_cap or plan.B > self.batch_cap:
            raise ValueError("batch (rows %d, tile floats %d, graphs %d) exceeds the capacity (%d, %d, %d)"
                             % (plan.rows, plan.tile_elems, plan.B, self.rows_cap, self.tiles_cap, self.batch_cap))
        lib, d, hp = self.lib, C.byref(self.desc), plan.host_ptr
        rb = lib.agcn_sgc_ll_reserve_bytes(d, hp)
        wb = max(lib.agcn_sgc_ll_forward_workspace_bytes(d, hp), lib.agcn_sgc_ll_backward_workspace_bytes(d, hp))
        if rb > self._reserve.numel() or wb > self._ws.numel():
            # grows geometrically: a handful of times per epoch at most.  The step may be running on a stream other
            # than the allocator's: nothing may still be using the old buffers when they go back to the pool, and the
            # new ones must be zero-filled before a kernel on that stream touches them
            torch.cuda.synchronize(self.device)
            if rb > self._reserve.numel():
                self._reserve = torch.zeros(int(rb * 1.5), dtype=torch.uint8, device=self.device)
            if wb > self._ws.numel():
                self._ws = torch.zeros(int(wb * 1.5), dtype=torch.uint8, device=self.device)
            torch.cuda.synchronize(self.device)
            self._ebufs.clear()
        return rb

    def step(self, batch, dY=None, stream=None):
        """Forward + backward of `batch` on `stream` (a torch stream; default: the current one)."""
        plan = batch.plan
        X = batch.X
        if (X.dtype != self.feat_dtype or not X.is_contiguous() or X.dim() != 2 or X.shape[1] != self.cfg.Fin
                or X.shape[0] < plan.rows):
            raise ValueError("node features must be contiguous %s [>= %d rows, Fin]" % (self.feat_dtype, plan.rows))
        if self.feat_dtype == torch.bfloat16 and not (self._dispatch(batch)[0] & 256):
            raise _ffi.AgcnError("this batch is not served with bf16 features by the specialised kernels (a molecule beyond the "
                                 "members' reach): run it through step_ids (which converts inside the call) or in float32")
        rb = self._fit(plan)
        if dY is not None and (dY.dtype != self.feat_dtype or not dY.is_contiguous() or dY.dim() != 2
                               or dY.shape[1] != self.cfg.Fout or dY.shape[0] < plan.rows):
            raise ValueError("dY must be contiguous fp32 [>= %d rows, Fout]" % plan.rows)
        gb = _gbatch(batch, X, None)
        stream = (torch.cuda.current_stream(self.device) if stream is None else stream).cuda_stream
        d, p, o = C.byref(self.desc), C.byref(self.pr), C.byref(self.out)
        _ffi.check(self.lib.agcn_sgc_ll_forward(d, C.byref(gb), p, o, self._reserve.data_ptr(), rb, self._ws.data_ptr(),
                                                self._ws.numel(), stream))
        _ffi.check(self.lib.agcn_sgc_ll_backward(d, C.byref(gb), p, o, (self._dY if dY is None else dY).data_ptr(), None,
                                                 C.byref(self.gr), self._reserve.data_ptr(), rb, self._ws.data_ptr(),
                                                 self._ws.numel(), stream))
        self.plan, self.batch = plan, batch       # (keeps the batch's device tensors alive until the next step)
        self._rows, self._tile_elems = plan.rows, plan.tile_elems
        self._stream = stream
        self._last = ("batch", batch, dY)
        return self

    def _dispatch(self, batch):
        out = (C.c_int32 * 8)()
        _ffi.check(self.lib.agcn_sgc_ll_dispatch_info(C.byref(self.desc), C.byref(_gbatch(batch, batch.X, None)), out, 8))
        return list(out)

    def step_ids(self, rstore, graph_ids, lane=0, dY=None, stream=None):
        """Stage the batch `graph_ids` out of the resident store `rstore` (lane `lane`'s buffers) and run forward + backward on
        it in ONE C-ABI call (include/agcn.h: agcn_epoch_step) -- no plan object, no size queries, no per-batch structs on
        the Python side.  Same kernels, same bits as `step(rstore.stage(graph_ids, lane))`."""
        ids = np.ascontiguousarray(graph_ids, dtype=np.int64)
        B = int(ids.size)
        if B > min(self.batch_cap, rstore.batch_cap):
            raise ValueError("batch of %d graphs exceeds the capacity" % B)
        if rstore.stage_dtype != self.feat_dtype:
            raise ValueError("the resident store stages %s rows, this step runs %s features" % (rstore.stage_dtype, self.feat_dtype))
        # (the pinned blobs of lane k: slots k, k + lanes, ... -- every lane rotates through its own, so that lanes driven from
        #  different host threads share nothing)
        nl = len(rstore.lanes)
        kk = rstore._lane_k[lane]
        rstore._lane_k[lane] = kk + 1
        sl = rstore._slots[lane + nl * (kk % (len(rstore._slots) // nl))]
        tstream = torch.cuda.current_stream(self.device) if stream is None else stream
        st = tstream.cuda_stream
        ev = sl.get("ev")
        if ev is None:                       # one HIP event per pinned blob: waited for and recorded inside the call
            if sl["done"] is not None:
                sl["done"].synchronize()
            e = torch.cuda.Event()
            e.record(tstream)
            sl["done"], ev = e, int(e.cuda_event)
            sl["ev"] = ev
        key = (id(rstore), lane, sl["ptr"], None if dY is None else dY.data_ptr())
        buf = self._ebufs.get(key)
        if buf is None:
            if rstore.F != self.cfg.Fin or rstore.rows_cap > self.rows_cap or rstore.tiles_cap > self.tiles_cap:
                raise ValueError("the resident store's capacities exceed this step's")
            ln = rstore.lanes[lane]
            g = self.grads
            buf = self._ebufs[key] = _ffi.EpochBuffers(
                sl["ptr"], ln["stage"].data_ptr(), rstore.stage_cap, ln["X"].data_ptr(), ln["rowptr"].data_ptr(),
                ln["col"].data_ptr(), ln["val"].data_ptr(), rstore.rows_cap, rstore.nnz_cap, self.tiles_cap,
                self._Y.data_ptr(), self._res_L.data_ptr(), self._res_W.data_ptr(), (self._dY if dY is None else dY).data_ptr(),
                self._dX.data_ptr(), g["weight"].data_ptr(), g["bias"].data_ptr(), _ptr(g.get("M_L")), g["alpha"].data_ptr(),
                self._reserve.data_ptr(), self._reserve.numel(), self._ws.data_ptr(), self._ws.numel())
        info, needs = self._einfo, self._eneeds
        rc = self.lib.agcn_epoch_step(C.byref(rstore._c), ids.ctypes.data, B, C.byref(self.desc), C.byref(self.pr), C.byref(buf), 1,
                                      C.byref(info), C.byref(needs), ev, st)
        if rc == -2 and needs.reserve_bytes:                # AGCN_ENOSPACE: grow (as _fit does) and run the staged batch
            torch.cuda.synchronize(self.device)
            if needs.reserve_bytes > self._reserve.numel():
                self._reserve = torch.zeros(int(needs.reserve_bytes * 1.5), dtype=torch.uint8, device=self.device)
            if needs.workspace_bytes > self._ws.numel():
                self._ws = torch.zeros(int(needs.workspace_bytes * 1.5), dtype=torch.uint8, device=self.device)
            torch.cuda.synchronize(self.device)
            self._ebufs.clear()
            return self._retry_ids(rstore, ids, lane, dY, sl, ev, st)
        _ffi.check(rc)
        self._rows, self._tile_elems = int(info.plan.rows), int(info.plan.tile_elems)
        self.plan = self.batch = None
        self._stream = st
        self._last = ("ids", rstore, ids, lane, dY, sl, ev)
        return self

    def _retry_ids(self, rstore, ids, lane, dY, sl, ev, st):
        ln, g = rstore.lanes[lane], self.grads
        buf = _ffi.EpochBuffers(
            sl["ptr"], ln["stage"].data_ptr(), rstore.stage_cap, ln["X"].data_ptr(), ln["rowptr"].data_ptr(),
            ln["col"].data_ptr(), ln["val"].data_ptr(), rstore.rows_cap, rstore.nnz_cap, self.tiles_cap,
            self._Y.data_ptr(), self._res_L.data_ptr(), self._res_W.data_ptr(), (self._dY if dY is None else dY).data_ptr(),
            self._dX.data_ptr(), g["weight"].data_ptr(), g["bias"].data_ptr(), _ptr(g.get("M_L")), g["alpha"].data_ptr(),
            self._reserve.data_ptr(), self._reserve.numel(), self._ws.data_ptr(), self._ws.numel())
        _ffi.check(self.lib.agcn_epoch_step(C.byref(rstore._c), ids.ctypes.data, int(ids.size), C.byref(self.desc), C.byref(self.pr),
                                            C.byref(buf), 0, C.byref(self._einfo), C.byref(self._eneeds), ev, st))
        self._rows, self._tile_elems = int(self._einfo.plan.rows), int(self._einfo.plan.tile_elems)
        self.plan = self.batch = None
        self._stream = st
        self._last = ("ids", rstore, ids, lane, dY, sl, ev)
        return self

    def check(self, recover=False):
        """Waits for the stream of the last step.  If a cooperative wait ran out in any step since the last check (include/agcn.h:
        agcn_sgc_ll_reserve_status): raises AgcnError, or -- `recover=True` -- re-issues THE LAST batch on the non-member dispatch
        (its inputs are still staged; fresh launches, per-call flags), counts it (`self.recovered`, `agcn_amd.recovered_calls`)
        and returns the number of waits that had run out.  Batches before the last one cannot be re-run from here: check after
        every batch whose results matter.  Returns 0 when nothing happened."""
        st = getattr(self, "_stream", None) or torch.cuda.current_stream(self.device).cuda_stream
        n, msg = _reserve_timeouts(self.lib, self._reserve, self._reserve.numel(), st, self.device)
        if n == 0:
            return 0
        self._reserve[:_ffi.RESERVE_HEADER_BYTES].zero_()     # re-arms the header (exchange words + the sticky count)
        if not recover or getattr(self, "_last", None) is None:
            raise _ffi.AgcnError("agcn error -3: %s" % msg)
        self._rerun_without_members(st)
        return n

    def _rerun_without_members(self, st):
        lib, desc = self.lib, _no_member_desc(self.desc)
        last = self._last
        with torch.cuda.device(self.device):
            if last[0] == "batch":
                _, batch, dY = last
                hp = batch.plan.host_ptr
                rb = lib.agcn_sgc_ll_reserve_bytes(C.byref(desc), hp)
                wb = max(lib.agcn_sgc_ll_forward_workspace_bytes(C.byref(desc), hp),
                         lib.agcn_sgc_ll_backward_workspace_bytes(C.byref(desc), hp))
                reserve = torch.zeros(int(rb), dtype=torch.uint8, device=self.device)
                ws = torch.empty(int(wb), dtype=torch.uint8, device=self.device)
                gb = _gbatch(batch, batch.X, None)
                d, p, o = C.byref(desc), C.byref(self.pr), C.byref(self.out)
                _ffi.check(lib.agcn_sgc_ll_forward(d, C.byref(gb), p, o, reserve.data_ptr(), rb, ws.data_ptr(), wb, st))
                _ffi.check(lib.agcn_sgc_ll_backward(d, C.byref(gb), p, o, (self._dY if dY is None else dY).data_ptr(), None,
                                                    C.byref(self.gr), reserve.data_ptr(), rb, ws.data_ptr(), wb, st))
            else:
                _, rstore, ids, lane, dY, sl, ev = last
                ln, g = rstore.lanes[lane], self.grads
                reserve, ws = self._reserve, self._ws
                for _ in range(2):       # (second round: the non-member kernels asked for more room)
                    buf = _ffi.EpochBuffers(
                        sl["ptr"], ln["stage"].data_ptr(), rstore.stage_cap, ln["X"].data_ptr(), ln["rowptr"].data_ptr(),
                        ln["col"].data_ptr(), ln["val"].data_ptr(), rstore.rows_cap, rstore.nnz_cap, self.tiles_cap,
                        self._Y.data_ptr(), self._res_L.data_ptr(), self._res_W.data_ptr(),
                        (self._dY if dY is None else dY).data_ptr(), self._dX.data_ptr(), g["weight"].data_ptr(),
                        g["bias"].data_ptr(), _ptr(g.get("M_L")), g["alpha"].data_ptr(),
                        reserve.data_ptr(), reserve.numel(), ws.data_ptr(), ws.numel())
                    rc = lib.agcn_epoch_step(C.byref(rstore._c), ids.ctypes.data, int(ids.size), C.byref(desc), C.byref(self.pr),
                                             C.byref(buf), 0, C.byref(self._einfo), C.byref(self._eneeds), ev, st)
                    if rc == -2 and self._eneeds.reserve_bytes:
                        reserve = torch.zeros(int(self._eneeds.reserve_bytes), dtype=torch.uint8, device=self.device)
                        ws = torch.zeros(max(int(self._eneeds.workspace_bytes), 256), dtype=torch.uint8, device=self.device)
                        continue
                    _ffi.check(rc)
                    break
            n, msg = _reserve_timeouts(lib, reserve, reserve.numel(), st, self.device)
        if n:
            raise _ffi.AgcnError("a cooperative wait ran out on the non-member dispatch (which has none): %s" % msg)
        self.recovered += 1
        _fn.recovered_calls[0] += 1

    # views of the current batch's results
    @property
    def Y(self):
        return self._Y[:self._rows]

    @property
    def dX(self):
        return self._dX[:self._rows]

    @property
    def res_L(self):
        return self._res_L[:self._tile_elems]

    @property
    def res_W(self):
        return self._res_W[:self._tile_elems]


def _overlapping_streams(device, k, candidates=12, spin=400000):
    """`k` streams of `device` whose work really runs side by side.  ROCm multiplexes streams onto a few hardware queues
    and which queue a (pooled) torch stream lands on is not under the caller's control: two streams on one queue run
    their kernels strictly one after the other (measured: 2 lanes at 1.9 instead of 3.1 M graphs/s).  So candidates are
    TESTED: a single-thread spin kernel on each of two streams takes one spin if they overlap, two if they do not.
    (The runtime may still re-map streams to queues later; a process that keeps more than ~3 streams busy should raise
    the number of hardware queues: GPU_MAX_HW_QUEUES=8 in the environment, before the first HIP call.)"""
    if k <= 1:
        return [torch.cuda.current_stream(device)]
    sleep = getattr(torch.cuda, "_sleep", None)
    pool = [torch.cuda.Stream(device) for _ in range(candidates)]
    if sleep is None:                                   # cannot test: take what the pool hands out
        return pool[:k]

    def side_by_side(a, b):
        torch.cuda.synchronize(device)
        t = []
        for pair in ((a,), (a, b)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cur = torch.cuda.current_stream(device)
            e0.record(cur)
            for s in pair:
                s.wait_event(e0)
                with torch.cuda.stream(s):
                    sleep(spin)
            for s in pair:
                cur.wait_stream(s)
            e1.record(cur)
            torch.cuda.synchronize(device)
            t.append(e0.elapsed_time(e1))
        return t[1] < 1.5 * t[0]

    chosen = []
    for s in pool:
        if all(side_by_side(c, s) for c in chosen):
            chosen.append(s)
            if len(chosen) == k:
                return chosen
    return (chosen + [s for s in pool if s not in chosen])[:k]     # fewer independent queues than lanes


class SgcLLEpochLanes(object):
    """`lanes` independent `SgcLLEpochStep`s, each on its own stream with its own buffers, fed round-robin from a
    `ResidentStore`: consecutive batches overlap on the GPU.  A step of one 256-graph batch is as long as the dependent
    chain of its largest graph (one workgroup per graph) and leaves most CUs idle behind it; a second batch in flight
    fills them.  Each batch is computed exactly as by a lone `SgcLLEpochStep` (same kernels, same bits).

    Valid wherever consecutive batches do not depend on each other: evaluation / prediction passes and gradient
    accumulation over `lanes` micro-batches (sum the lanes' `bucket`s; SUM semantics as in dist.py).  Plain SGD with a
    parameter update after every batch has that dependency: use one lane.

        rs = store.to_device(dev, 256, lanes=2)
        ep = SgcLLEpochLanes(layer, rs, lanes=2)
        for j, sel in enumerate(selections):
            step = ep.step(sel)            # the SgcLLEpochStep that holds this batch's results (on ep.streams[j % 2])
        ep.synchronize()
    """

    def __init__(self, layer, rstore, lanes=2):
        if len(rstore.lanes) < lanes:
            raise ValueError("the resident store was built with %d lane(s)" % len(rstore.lanes))
        self.rstore, self.device = rstore, rstore.device
        # All member workgroups of the launches in flight must be resident together (csrc/spec_backward_coop.h): the
        # library's budget of 120 per launch is sized for two concurrent launches; with more lanes it is divided -- a PER-CALL
        # flag of these lanes' calls (AGCN_FLAG_COOP_MEMBERS): nothing process-wide changes, step objects built elsewhere
        # keep the full budget
        flags = _fn.current_flags()
        if lanes > 2:
            have = ((flags >> 24) & 255) - 1 if (flags >> 24) & 255 else _ffi.lib().agcn_get_coop_members()
            flags = _ffi.dispatch_flags(flags, coop_members=min(have, 240 // lanes))
        self.member_budget = ((flags >> 24) & 255) - 1 if (flags >> 24) & 255 else None
        self.steps = [SgcLLEpochStep(layer, rstore.rows_cap, rstore.tiles_cap, rstore.batch_cap, device=self.device, flags=flags,
                                     feat_dtype=rstore.stage_dtype) for _ in range(lanes)]
        self.streams = _overlapping_streams(self.device, lanes)
        self._j = 0
        torch.cuda.synchronize(self.device)      # buffers, parameters and the store were filled on other streams

    def step(self, graph_ids, dY=None):
        k = self._j % len(self.steps)
        self._j += 1
        s = self.streams[k]
        return self.steps[k].step_ids(self.rstore, graph_ids, lane=k, dY=dY, stream=s)      # (one C-ABI call per batch)

    def run(self, selections, dY=None, on_batch=None):
        """A whole list of batches, lane k taking batches k, k + lanes, ... from a HOST THREAD of its own.  A batch is one
        C-ABI call (`agcn_epoch_step`: ~50 us of host time, almost all of it kernel-launch API calls) during which ctypes has
        released the GIL, so the lanes' host work can overlap.  (Measured on one MI355X with two lanes: 59.9 us per batch against
        55.0 us for `step()` from one thread -- the runtime serialises the launches and two batches in flight are within 10 %
        of the sum of their kernels' durations anyway; useful where `on_batch` has host work of its own.)
        `on_batch(j, step)` runs in the lane's thread right after batch j has been enqueued (enqueue further work on
        `self.streams[j % lanes]` there: the step's buffers are reused by that lane's next batch).  Returns when everything is
        enqueued; `synchronize()` waits for the GPU."""
        import threading
        sels = list(selections)
        nl = len(self.steps)
        errs = []

        def lane(k):
            try:
                with torch.cuda.device(self.device):
                    for j in range(k, len(sels), nl):
                        st = self.steps[k].step_ids(self.rstore, sels[j], lane=k, dY=dY, stream=self.streams[k])
                        if on_batch is not None:
                            on_batch(j, st)
            except BaseException as e:          # noqa: B902  (re-raised in the caller's thread)
                errs.append(e)

        threads = [threading.Thread(target=lane, args=(k,)) for k in range(nl)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        self._j += len(sels)
        if errs:
            raise errs[0]

    def close(self):
        """Waits for the lanes (kept for `with` blocks; there is no process-wide state to give back any more)."""
        for s in self.streams:
            s.synchronize()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self, check=True, recover=False):
        """Waits for every lane and (check=True) raises AgcnError if a cooperative wait ran out on any of them (recover=True:
        each lane re-issues its LAST batch on the non-member dispatch instead, SgcLLEpochStep.check)."""
        for s in self.streams:
            s.synchronize()
        if check:
            for st in self.steps:
                st.check(recover=recover)


class SgcLLEpochPipeline(object):
    """ONE lane of `SgcLLEpochStep` (batches strictly one after the other on the compute stream: plain SGD order) fed by a
    `ResidentStore` whose staging runs ONE BATCH AHEAD on a side stream: while batch k computes, the plan / gather table of
    batch k + 1 is copied and its rows are gathered into the other of two buffer sets.  A batch's data does not depend on
    the parameters, so this is prefetching, not batch overlap: results are those of the plain loop, bit for bit.

        rs = store.to_device(dev, 256, lanes=2)          # two buffer sets
        pipe = SgcLLEpochPipeline(layer, rs)
        for step in pipe.run(selections):                # step = the SgcLLEpochStep holding that batch's results
            ...                                          # (valid until the next iteration; enqueue an optimiser step here)
        pipe.synchronize()
    """

    def __init__(self, layer, rstore):
        if len(rstore.lanes) < 2:
            raise ValueError("the pipeline needs a resident store with lanes >= 2 (two buffer sets)")
        self.rstore, self.device = rstore, rstore.device
        self.step_obj = SgcLLEpochStep(layer, rstore.rows_cap, rstore.tiles_cap, rstore.batch_cap, device=self.device,
                                       feat_dtype=rstore.stage_dtype)
        self.side = _overlapping_streams(self.device, 2)[1]
        self._staged = [torch.cuda.Event(), torch.cuda.Event()]     # staging of buffer set b has executed
        self._freed = [None, None]                                  # last compute that read buffer set b has executed
        torch.cuda.synchronize(self.device)

    def _stage(self, sel, b, main):
        if self._freed[b] is not None:
            self.side.wait_event(self._freed[b])                    # the step that last used these buffers is done
        gb = self.rstore.stage(sel, lane=b, stream=self.side)
        self._staged[b].record(self.side)
        return gb

    def run(self, selections, dY=None):
        main = torch.cuda.current_stream(self.device)
        sels = list(selections)
        if not sels:
            return
        nxt = self._stage(sels[0], 0, main)
        for k in range(len(sels)):
            b = k & 1
            gb = nxt
            if k + 1 < len(sels):
                nxt = self._stage(sels[k + 1], b ^ 1, main)          # one batch ahead, on the side stream
            main.wait_event(self._staged[b])
            st = self.step_obj.step(gb, dY, stream=main)
            if self._freed[b] is None:
                self._freed[b] = torch.cuda.Event()
            self._freed[b].record(main)
            yield st

    def synchronize(self, check=True):
        torch.cuda.current_stream(self.device).synchronize()
        self.side.synchronize()
        if check:
            self.step_obj.check()


class GraphedStep(object):
    """A whole training step of layers of this package -- autograd forward + backward over a fixed batch --
    captured into ONE HIP graph (torch.cuda.CUDAGraph on ROCm).

    `fn()` must run the step on static tensors (same batch object, same parameter / input tensors every call) and
    leave its results in tensors it returns or in `.grad` fields; it is run `warmup` times on a side stream (code
    objects loaded, function attributes set, allocator pools primed), then captured.  `replay()` re-runs the
    captured launches; outputs and `.grad` tensors are the ones of the capture pass, overwritten in place.
    At 256 Tox21-shaped graphs the 4-layer SimpleAGCN-shaped step goes from 830 us (eager, Python-bound) to
    417 us."""

    def __init__(self, fn, device, warmup=3):
        dev = torch.device(device)
        self.fn, self.recovered = fn, 0
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # captured on the stream the warm-up ran on: autograd's AccumulateGrad nodes remember the stream they were created
        # on, and one that outlives an iteration (a container or layer attribute that still references the last step's
        # graph) would otherwise meet the capture on a different stream -- a synchronisation inside the capture
        # pooled reserve buffers the capture pins (functional._lease_reserve) belong to THIS object: they are freed
        # with the graph (reset() / __del__), not to a process-wide list
        from . import functional as _fn
        self._scope = _fn.capture_scope()
        with self._scope, torch.cuda.graph(self.graph, stream=side):
            self.outputs = fn()

    def replay(self):
        self.graph.replay()
        return self.outputs

    def replay_checked(self):
        """A replay that cannot be lost: replays, synchronises, and if a wait between the workgroups that share a large graph ran
        out anywhere in it (results NaN; include/agcn.h: agcn_device_status) RE-ISSUES the step -- `fn()` eagerly, fresh launches
        on the dispatch in which no workgroup waits for another (per-call flags) -- and returns THAT run's outputs (new tensors;
        parameter gradients in the `.grad` fields as `fn` leaves them).  Counted in `self.recovered` / `agcn_amd.recovered_calls`."""
        _fn.device_timeouts()
        out = self.replay()
        if _fn.device_timeouts() == 0:
            return out
        nm = _fn.no_member_flags()
        with _fn.dispatch(spec_mode=(nm & 15) - 1, pg_onepass=((nm >> 7) & 3) - 1):
            out = self.fn()
        if _fn.device_timeouts() != 0:
            raise _ffi.AgcnError("a cooperative wait ran out on the non-member dispatch (which has none): device fault")
        self.recovered += 1
        _fn.recovered_calls[0] += 1
        return out

    @property
    def pinned_reserve_bytes(self):
        return self._scope.nbytes

    def reset(self):
        """Drops the captured graph and its outputs and frees the reserve buffers it pinned."""
        if self.graph is not None:
            torch.cuda.synchronize()
            self.outputs = self.fn = None
            self.graph.reset()
            self.graph = None
            self._scope.release()

    def __del__(self):
        try:
            self.reset()
        except Exception:          # (interpreter shutdown)
            pass
