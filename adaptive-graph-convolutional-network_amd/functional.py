"""torch.autograd bridge to the C ABI: one forward / one backward call per layer for the
whole batch.  PyTorch is used for device memory, streams and autograd bookkeeping only --
all arithmetic happens in the HIP library (there is no eager fallback)."""
import ctypes as C
import functools
import threading

import torch

from . import _ffi


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _on_tensor_device(fn):
    """Runs an autograd Function's forward / backward with the device of its first GPU tensor made current: the
    library resolves per-device state (dynamic-LDS caps, the mixed-dispatch side stream) through hipGetDevice(), and
    a launch on a stream of another device than the current one would fail."""
    @functools.wraps(fn)
    def wrapped(ctx, *args):
        dev = next((a.device for a in args if torch.is_tensor(a) and a.is_cuda), None)
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(ctx, *args)
        with torch.cuda.device(dev):
            return fn(ctx, *args)
    return wrapped


def _require_gpu(t, what):
    if not t.is_cuda:
        raise _ffi.AgcnError("%s must live on the GPU: the AGCN layer path has no CPU fallback" % what)


def _f32c(t, what):
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32" % what)
    return t if t.is_contiguous() else t.contiguous()


_DT = {torch.float32: 0, torch.bfloat16: 1}


class CastRowsFunction(torch.autograd.Function):
    """Storage-dtype conversion of a feature matrix (bf16 <-> f32) through the C ABI; the gradient is cast back."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, X, dtype):
        _require_gpu(X, "features")
        if X.dtype not in _DT or dtype not in _DT:
            raise TypeError("features must be float32 or bfloat16")
        ctx.src = X.dtype
        if X.dtype == dtype:
            return X
        X = X if X.is_contiguous() else X.contiguous()
        out = torch.empty(X.shape, dtype=dtype, device=X.device)
        cols = X.shape[-1]
        _ffi.check(_ffi.lib().agcn_cast_rows(X.data_ptr(), _DT[X.dtype], out.data_ptr(), _DT[dtype],
                                             X.numel() // cols, cols, cols, cols, _stream(X.device)))
        return out

    @staticmethod
    @_on_tensor_device
    def backward(ctx, g):
        return CastRowsFunction.apply(g, ctx.src), None


class PackRowsFunction(torch.autograd.Function):
    """Reference feed -> packed rows: [B * stride, F] (graph b at rows b*stride .. b*stride + n_b) -> [N1, F].
    `batch.plan` is the PACKED plan.  Backward unpacks the gradient (pad rows get 0)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, Xpad, batch, stride):
        _require_gpu(Xpad, "node features")
        Xpad = _f32c(Xpad, "node features")
        plan, F = batch.plan, Xpad.shape[1]
        if Xpad.shape[0] != plan.B * stride:
            raise ValueError("padded node features must be [B * max_atom, F]")
        out = torch.empty((plan.N1, F), dtype=torch.float32, device=Xpad.device)
        _ffi.check(_ffi.lib().agcn_rows_pack(plan.host_ptr, plan.dev_ptr, Xpad.data_ptr(), int(stride), F, F,
                                             out.data_ptr(), F, _stream(Xpad.device)))
        ctx.batch, ctx.stride = batch, int(stride)
        return out

    @staticmethod
    @_on_tensor_device
    def backward(ctx, g):
        return UnpackRowsFunction.apply(g, ctx.batch, ctx.stride, None), None, None


class UnpackRowsFunction(torch.autograd.Function):
    """Packed rows -> the reference's padded layout [B * stride, F]; pad rows are 0, or `fill` [F] (DenseMol's
    bias lands on pad rows too, dense_layer.py:42-50).  Backward packs the gradient."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, Xp, batch, stride, fill):
        _require_gpu(Xp, "node features")
        Xp = _f32c(Xp, "node features")
        plan, F = batch.plan, Xp.shape[1]
        out = torch.empty((plan.B * int(stride), F), dtype=torch.float32, device=Xp.device)
        if fill is not None:
            fill = _f32c(fill.detach(), "fill")
        _ffi.check(_ffi.lib().agcn_rows_unpack(plan.host_ptr, plan.dev_ptr, Xp.data_ptr(), F, F, out.data_ptr(),
                                               int(stride), F, _ptr(fill), _stream(Xp.device)))
        ctx.batch, ctx.stride = batch, int(stride)
        return out

    @staticmethod
    @_on_tensor_device
    def backward(ctx, g):
        return PackRowsFunction.apply(g, ctx.batch, ctx.stride), None, None, None


class _Cfg(object):
    __slots__ = ("Fin", "Fout", "K", "mode", "variant", "activation", "lsum", "prev_sym", "bf16")

    def __init__(self, Fin, Fout, K, mode, variant, activation, lsum=True, prev_sym=False, bf16=False):
        # bf16: X / Y / dY / dX are bf16 tensors read and written by the kernels themselves (AgcnLayerDesc.feat_dtype)
        self.bf16 = bool(bf16)
        # prev_sym: the caller vouches that L_prev is symmetric (AgcnGraphBatch.L_prev_symmetric)
        self.prev_sym = bool(prev_sym)
        self.Fin, self.Fout, self.K = int(Fin), int(Fout), int(K)
        self.mode, self.variant, self.activation = mode, variant, activation
        # lsum=False: an SGC_LL call whose combined Laplacian nobody reads (the reference's SGC_LL returns three
        # outputs, graphconv.py:252) -- the L_sum tile is then neither allocated nor written
        self.lsum = bool(lsum) or variant == "SGC_LL_Reslap"


# Per-call dispatch flags (include/agcn.h: AgcnLayerDesc.flags): which of the equivalent kernel paths serves the calls THIS
# thread issues inside a `with dispatch(...)` block.  Nothing process-wide is touched; other threads keep their own flags.
_tls = threading.local()


def current_flags():
    return getattr(_tls, "flags", 0)


class dispatch(object):
    """`with agcn_amd.dispatch(spec_mode=3):` -- the layer calls issued by this thread inside the block carry these per-call
    dispatch overrides (names and values as the agcn_set_* functions of include/agcn.h: spec_mode, fused_mode, pg_onepass,
    spec_small, rows_split, slab_mode, fused_gb, spec_tail, coop_members).  The backward of an autograd call uses the flags
    its forward ran with, whichever thread runs it.  Step objects (SgcLLStep, ...) take the flags current at construction."""

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        self.prev = current_flags()
        _tls.flags = _ffi.dispatch_flags(self.prev, **self.knobs)
        return self

    def __exit__(self, *exc):
        _tls.flags = self.prev
        return False


def no_member_flags(base=None):
    """`base` (default: this thread's current flags) with everything switched off in which one workgroup WAITS for another
    inside a launch: the member workgroups that share a large graph (every graph then runs in one workgroup, or on the slab
    kernels) and the tagged hand-off of the parameter-gradient kernel (the arrival-counter kernel instead: its last arriver
    adds the partials, nobody polls) -- the dispatch a call is re-issued on when a bounded wait ran out."""
    base = current_flags() if base is None else int(base)
    sm = (base & 15) - 1 if (base & 15) else _ffi.lib().agcn_get_spec_mode()
    pg = ((base >> 7) & 3) - 1 if ((base >> 7) & 3) else _ffi.lib().agcn_get_param_grad_onepass()
    return _ffi.dispatch_flags(base, spec_mode=sm & 3, pg_onepass=min(pg, 1))


# Parameter gradients beside the backward chain (include/agcn.h: agcn_sgc_ll_backward_forked).  In a multi-layer step the
# parameter-gradient kernel of layer l depends on that layer's per-graph backward kernel only; nothing downstream of it is needed by
# the backward of layer l - 1.  Inside `with param_grad_fork(device):` every SGC layer's backward enqueues its parameter gradients
# on ONE side stream (an event edge per layer), the main stream goes straight on to the previous layer, and the two streams are
# joined once when the block ends.  Process-wide per device (autograd runs backward() on its own thread), so one training step at
# a time per device.  Parameters' .grad must be None (or untouched) until the block ends: the gradient tensors are filled on the
# side stream -- autograd only STORES them (no kernel) when .grad is None.
_forks = {}


class param_grad_fork(object):
    def __init__(self, device):
        dev = torch.device(device)
        self.idx = dev.index if dev.index is not None else torch.cuda.current_device()
        st = _forks.get(self.idx)
        if st is None:
            st = _forks[self.idx] = {"stream": torch.cuda.Stream(torch.device("cuda", self.idx)), "keep": [], "active": False}
        self.st = st

    def __enter__(self):
        self.st["active"] = True
        self.st["used"] = False
        return self

    def __exit__(self, *exc):
        st = self.st
        st["active"] = False
        if st["used"]:
            torch.cuda.current_stream(torch.device("cuda", self.idx)).wait_stream(st["stream"])      # the ONE join of the step
        if torch.cuda.is_current_stream_capturing():
            # a captured graph replays on these buffers: they belong to the capturing object (engine.GraphedStep's capture_scope)
            # for as long as its graph lives, like the pooled reserves a capture pins
            (_capture_scopes[-1].held if _capture_scopes else _unscoped_captures).extend((None, k) for k in st["keep"])
        del st["keep"][:]
        return False


def _active_fork(dev):
    st = _forks.get(dev.index if dev.index is not None else torch.cuda.current_device())
    return st if (st is not None and st["active"]) else None


# how many calls were re-issued on the non-member dispatch after a cooperative wait ran out (engine.*.check(recover=True),
# functional.run_checked)
recovered_calls = [0]


def _desc(cfg, ldx, ldy, flags=None):
    return _ffi.LayerDesc(cfg.Fin, cfg.Fout, cfg.K, _ffi.MODE[cfg.mode], _ffi.VARIANT[cfg.variant],
                          _ffi.ACT[cfg.activation], int(ldx), int(ldy), 1 if getattr(cfg, "bf16", False) else 0,
                          current_flags() if flags is None else int(flags))


def _gbatch(batch, X, L_prev, prev_symmetric=False):
    p = batch.plan
    return _ffi.GraphBatch(p.host_ptr, p.dev_ptr, X.data_ptr(), _ptr(batch.l0_dense), batch.l0_ld,
                           1 if batch.l0_symmetric else 0,
                           _ptr(batch.l0_rowptr), _ptr(batch.l0_col), _ptr(batch.l0_val), _ptr(L_prev),
                           1 if (prev_symmetric and L_prev is not None) else 0, 0)


def _sizes(plan, cfg, ldx, ldy, flags=None):
    """(desc, reserve bytes, forward workspace bytes, backward workspace bytes) of a layer on a plan: three host-side
    FFI calls, cached on the plan (a plan is immutable)."""
    lib = _ffi.lib()
    flags = current_flags() if flags is None else int(flags)
    # the per-call flags and, where they leave a field alone, the process defaults change which kernels (and so which
    # workspaces) a call takes
    knobs = (flags, lib.agcn_get_lds_limit(), lib.agcn_get_fused_mode(), lib.agcn_get_slab_mode(),
             lib.agcn_get_fused_global_b(), lib.agcn_get_param_grad_onepass(), lib.agcn_get_spec_mode(),
             lib.agcn_get_spec_small(), lib.agcn_get_spec_tail(), lib.agcn_get_rows_split(), lib.agcn_get_coop_members())
    key = (cfg.Fin, cfg.Fout, cfg.K, cfg.mode, cfg.variant, cfg.activation, int(ldx), int(ldy), bool(cfg.bf16)) + knobs
    cache = plan.__dict__.setdefault("_size_cache", {})
    hit = cache.get(key)
    if hit is None:
        lib = _ffi.lib()
        desc = _desc(cfg, ldx, ldy, flags)
        hit = (desc, lib.agcn_sgc_ll_reserve_bytes(C.byref(desc), plan.host_ptr),
               lib.agcn_sgc_ll_forward_workspace_bytes(C.byref(desc), plan.host_ptr),
               lib.agcn_sgc_ll_backward_workspace_bytes(C.byref(desc), plan.host_ptr))
        cache[key] = hit
    return hit


# One persistent status word per device, registered with the library (include/agcn.h: agcn_set_device_status_word): a wait
# between the workgroups that share a large graph that runs out counts itself there.  The reserves of the autograd path die
# with their autograd graph -- right after backward() --, so the count has to live somewhere that outlives them (the weak
# references this module kept until round 3 were all dead by the time a training loop looked at them).
_status_words = {}


def _status_word(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    w = _status_words.get(idx)
    if w is None:
        w = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
        _ffi.check(_ffi.lib().agcn_set_device_status_word(idx, w.data_ptr()))
        _status_words[idx] = w
    return w


def device_timeouts(clear=True):
    """Synchronises the devices that ran SGC_LL autograd calls; returns how many cooperative waits ran out since the words
    were last cleared (0: every result is good) and, by default, clears them."""
    total = 0
    for idx, w in list(_status_words.items()):
        with torch.cuda.device(idx):
            t = C.c_int32(0)
            _ffi.lib().agcn_device_status(idx, torch.cuda.current_stream(idx).cuda_stream, C.byref(t))
            if t.value and clear:
                w.zero_()
            total += int(t.value)
    return total


def run_checked(fn):
    """A step that cannot be lost: runs `fn()` (the calls of one training / evaluation step on this thread: forward, loss,
    backward -- it must be re-runnable: same inputs, gradients zeroed or overwritten inside), synchronises, and if a wait between
    the workgroups that share a large graph ran out anywhere in it -- results NaN, include/agcn.h: agcn_device_status --
    RE-ISSUES the same calls on the non-member dispatch (fresh launches, per-call flags: no workgroup of those kernels waits
    for another) and returns that run's result.  `recovered_calls[0]` counts such re-runs."""
    device_timeouts()           # (whatever an earlier, unchecked step left behind is not this step's)
    out = fn()
    if device_timeouts() == 0:
        return out
    nm = no_member_flags()
    with dispatch(spec_mode=(nm & 15) - 1, pg_onepass=((nm >> 7) & 3) - 1):
        out = fn()
    recovered_calls[0] += 1
    if device_timeouts() != 0:
        raise _ffi.AgcnError("a cooperative wait ran out on the non-member dispatch (which has none): device fault")
    return out


def check_device_status():
    """Synchronises the devices that ran SGC_LL autograd calls and raises AgcnError if a wait between the workgroups that share
    a large graph ran out in any call since the last check -- forward OR backward: the count is kept in a persistent word per
    device, not in the reserves (which autograd frees once backward has run).  The gradients of such a step are NaN.  Call it
    where the training loop synchronises anyway (reading the loss).  Returns the number of devices checked; the word is
    cleared when an error is raised, so that one bad step is reported once."""
    n = 0
    for idx, w in list(_status_words.items()):
        with torch.cuda.device(idx):
            t = C.c_int32(0)
            rc = _ffi.lib().agcn_device_status(idx, torch.cuda.current_stream(idx).cuda_stream, C.byref(t))
            if rc != 0:
                msg = _ffi.lib().agcn_last_error().decode()
                w.zero_()
                raise _ffi.AgcnError("agcn error %d: %s" % (rc, msg))
        n += 1
    return n


# Reserve buffers of the autograd path are LEASED from a pool per (device, stream, size class) instead of allocated -- and their
# 64 KB exchange header zero-filled, one fill kernel per layer call: 3.7 us, 5 % of a molecule network's step -- on every forward.
# A reserve may serve any batch after any other (the header is self-maintaining: include/agcn.h), so a buffer that comes back
# from the pool needs nothing.  A lease goes back when the autograd node that holds it dies (after backward(), or when the outputs
# of a forward-only call are dropped).  Under stream capture: a pooled buffer (allocated eagerly, e.g. by the warm-up steps
# GraphedStep runs on its capture stream) is used at its fixed address by every replay, so it belongs to the CAPTURING OBJECT for
# as long as its graph lives (capture_scope(): engine.GraphedStep owns the list and frees the buffers when it is
# reset or deleted; a capture outside any scope pins its buffers in _unscoped_captures until release_reserve_pool(force=True));
# a buffer allocated INSIDE a capture belongs to the graph's own memory pool and never enters this one.
_reserve_pool = {}
_reserve_pool_bytes = [0]
_RESERVE_POOL_CAP = 2 << 30
_unscoped_captures = []
_capture_scopes = []


def _size_class(rbytes):
    """256 KB classes below 64 MB, 1/8-octave classes (at most +12.5 %) above: a point-cloud reserve of ~256 MB that grows by a
    few KB with the batch finds the buffer of the previous batch instead of pinning a new one."""
    if rbytes < (64 << 20):
        return (rbytes + 262143) // 262144 * 262144
    step = 1 << (int(rbytes - 1).bit_length() - 4)
    return (rbytes + step - 1) // step * step


def _pool_put(key, tensor):
    if _reserve_pool_bytes[0] + tensor.numel() <= _RESERVE_POOL_CAP:
        _reserve_pool.setdefault(key, []).append(tensor)
        _reserve_pool_bytes[0] += tensor.numel()


class capture_scope(object):
    """Owner of the pooled reserves a stream capture pins.  `with capture_scope() as scope:` around the capture; the buffers
    stay out of the pool until `scope.release()` (call it once the captured graph is gone)."""

    def __init__(self):
        self.held = []

    def __enter__(self):
        _capture_scopes.append(self)
        return self

    def __exit__(self, *exc):
        _capture_scopes.remove(self)
        return False

    def release(self):
        # (back to the torch caching allocator, not to the lease pool: the pool is keyed by stream, and the side stream of a
        # GraphedStep that is gone may never ask again)
        self.held = []

    @property
    def nbytes(self):
        return sum(t.numel() for _, t in self.held if torch.is_tensor(t))


def reserve_pool_bytes():
    """(bytes idle in the pool, bytes pinned by captures outside any capture_scope)."""
    return _reserve_pool_bytes[0], sum(t.numel() for _, t in _unscoped_captures if torch.is_tensor(t))


def release_reserve_pool(force=False):
    """Drops every idle reserve buffer of the pool (the torch caching allocator gets them back; torch.cuda.empty_cache() then
    returns them to the device).  `force=True` also drops the buffers pinned by captures that ran outside a capture_scope --
    only once those graphs are gone.  Returns the number of bytes dropped."""
    n = _reserve_pool_bytes[0]
    _reserve_pool.clear()
    _reserve_pool_bytes[0] = 0
    if force:
        n += sum(t.numel() for _, t in _unscoped_captures if torch.is_tensor(t))
        del _unscoped_captures[:]
    return n


class _ReserveLease(object):
    __slots__ = ("tensor", "key", "pooled")

    def __del__(self):
        try:
            if self.pooled:
                _pool_put(self.key, self.tensor)
        except Exception:          # (interpreter shutdown)
            pass


def _lease_reserve(dev, rbytes):
    size = _size_class(rbytes)
    capturing = torch.cuda.is_current_stream_capturing()
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(dev).cuda_stream, size)
    lease = _ReserveLease()
    lease.key = key
    lst = _reserve_pool.get(key)
    if lst:
        lease.tensor = lst.pop()
        _reserve_pool_bytes[0] -= lease.tensor.numel()
        lease.pooled = not capturing
        if capturing:
            (_capture_scopes[-1].held if _capture_scopes else _unscoped_captures).append((key, lease.tensor))
    else:
        lease.tensor = torch.empty(size, dtype=torch.uint8, device=dev)
        lease.tensor[:min(size, _ffi.RESERVE_HEADER_BYTES)].zero_()      # (must be zero when a reserve is first used)
        lease.pooled = not capturing
    return lease


class SgcLLFunction(torch.autograd.Function):
    """(Y, res_L, res_W, L_sum) = SGC_LL[_Reslap](X; weight, bias, M_L, alpha, beta, L_prev)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, X, weight, bias, M_L, alpha, beta, L_prev, batch, cfg):
        lib = _ffi.lib()
        _require_gpu(X, "node features")
        fdt = torch.bfloat16 if cfg.bf16 else torch.float32
        if cfg.bf16:
            if X.dtype != torch.bfloat16:
                raise TypeError("a bf16 layer call needs bf16 node features")
            X = X if X.is_contiguous() else X.contiguous()
        else:
            X = _f32c(X, "node features")
        weight, bias, M_L, alpha = (_f32c(weight, "weight"), _f32c(bias, "bias"), _f32c(M_L, "M_L"),
                                    _f32c(alpha, "alpha"))
        plan = batch.plan
        if X.dim() != 2 or X.shape[0] != plan.rows or X.shape[1] < cfg.Fin:
            raise ValueError("node features must be [rows=%d, >=%d], got %s" % (plan.rows, cfg.Fin, tuple(X.shape)))
        if tuple(weight.shape) != (cfg.Fin * cfg.K, cfg.Fout) or tuple(M_L.shape) != (cfg.Fin, cfg.Fin):
            raise ValueError("parameter shapes do not match the layer")
        reslap = cfg.variant == "SGC_LL_Reslap"
        if L_prev is not None:
            L_prev = _f32c(L_prev, "L_prev")
            if L_prev.numel() != plan.tile_elems:
                raise ValueError("L_prev must be a packed tile tensor of %d floats" % plan.tile_elems)
            if beta is None:
                raise ValueError("L_prev given without beta")
            beta = _f32c(beta, "beta")
        dev = X.device
        _status_word(dev)
        ldx, ldy = X.shape[1], cfg.Fout
        Y = torch.empty((plan.rows, ldy), dtype=fdt, device=dev)
        res_L = torch.empty(plan.tile_elems, dtype=torch.float32, device=dev)
        res_W = torch.empty(plan.tile_elems, dtype=torch.float32, device=dev)
        L_sum = torch.empty(plan.tile_elems if cfg.lsum else 0, dtype=torch.float32, device=dev)
        flags = current_flags()
        desc, rbytes, wbytes, _ = _sizes(plan, cfg, ldx, ldy, flags)
        # (the reserve starts with the exchange header of the cooperative kernels, which must be zero when a reserve is FIRST
        # used -- include/agcn.h: AGCN_RESERVE_HEADER_BYTES; only that: at point-cloud sizes the reserve holds n x n tiles, and
        # zero-filling all of it was 75 us per layer call -- and only then: see _lease_reserve)
        lease = _lease_reserve(dev, rbytes)
        reserve = lease.tensor
        ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
        gb = _gbatch(batch, X, L_prev if reslap else None, cfg.prev_sym)
        pr = _ffi.Params(weight.data_ptr(), bias.data_ptr(), M_L.data_ptr(), alpha.data_ptr(), _ptr(beta))
        out = _ffi.Outputs(Y.data_ptr(), res_L.data_ptr(), res_W.data_ptr(), L_sum.data_ptr() if cfg.lsum else None)
        _ffi.check(lib.agcn_sgc_ll_forward(C.byref(desc), C.byref(gb), C.byref(pr), C.byref(out),
                                           reserve.data_ptr(), reserve.numel(), ws.data_ptr(), wbytes, _stream(dev)))
        ctx.batch, ctx.cfg, ctx.has_prev, ctx.has_beta = batch, cfg, (reslap and L_prev is not None), beta is not None
        ctx.reserve_lease, ctx.flags = lease, flags
        ctx.save_for_backward(X, weight, bias, M_L, alpha, beta, L_prev if reslap else None,
                              Y, res_L, res_W, L_sum)
        ctx.mark_non_differentiable(res_L, res_W)
        ctx.set_materialize_grads(False)      # an unused L_sum must reach the library as "no upstream gradient"
        return Y, res_L, res_W, L_sum

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dY, _dres_L, _dres_W, dL_sum):
        lib = _ffi.lib()
        X, weight, bias, M_L, alpha, beta, L_prev, Y, res_L, res_W, L_sum = ctx.saved_tensors
        reserve = ctx.reserve_lease.tensor
        batch, cfg, plan = ctx.batch, ctx.cfg, ctx.batch.plan
        dev = X.device
        paper = cfg.mode == "paper"
        if dY is None:
            dY = torch.zeros_like(Y)
        if cfg.bf16:
            dY = dY.to(torch.bfloat16)
            dY = dY if dY.is_contiguous() else dY.contiguous()
        else:
            dY = _f32c(dY, "dY")
        if dL_sum is not None:
            dL_sum = _f32c(dL_sum, "dL_sum")
        ldx, ldy = X.shape[1], cfg.Fout
        desc, _, _, wbytes = _sizes(plan, cfg, ldx, ldy, ctx.flags)     # (the forward's flags: autograd runs this on its own thread)
        dX = torch.empty_like(X)
        if ldx > cfg.Fin:
            dX.zero_()
        dweight = torch.empty_like(weight)
        dbias = torch.empty_like(bias)
        dM = torch.empty_like(M_L) if paper else None
        dalpha = torch.empty_like(alpha)
        dbeta = torch.empty_like(beta) if ctx.has_prev else None
        dLp = torch.empty_like(L_prev) if ctx.has_prev else None
        ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
        gb = _gbatch(batch, X, L_prev, ctx.cfg.prev_sym)
        pr = _ffi.Params(weight.data_ptr(), bias.data_ptr(), M_L.data_ptr(), alpha.data_ptr(), _ptr(beta))
        fw = _ffi.Outputs(Y.data_ptr(), res_L.data_ptr(), res_W.data_ptr(), L_sum.data_ptr() if cfg.lsum else None)
        gr = _ffi.Grads(dX.data_ptr(), dweight.data_ptr(), dbias.data_ptr(), _ptr(dM), dalpha.data_ptr(),
                        _ptr(dbeta), _ptr(dLp))
        fk = _active_fork(dev)
        if fk is not None:
            main = torch.cuda.current_stream(dev)
            ev = torch.cuda.Event()
            ev.record(main)                                   # (creates the handle; the library records it again where it forks)
            _ffi.check(lib.agcn_sgc_ll_backward_forked(C.byref(desc), C.byref(gb), C.byref(pr), C.byref(fw),
                                                       dY.data_ptr(), _ptr(dL_sum), C.byref(gr), reserve.data_ptr(),
                                                       reserve.numel(), ws.data_ptr(), wbytes, main.cuda_stream,
                                                       fk["stream"].cuda_stream, int(ev.cuda_event)))
            # what the side stream still reads when this function returns: alive until the join (NOT the gradient tensors: a
            # second reference would make autograd copy them -- on the main stream, before they are written)
            fk["keep"].append((ws, ev, ctx.reserve_lease, X, Y, dY))
            fk["used"] = True
        else:
            _ffi.check(lib.agcn_sgc_ll_backward(C.byref(desc), C.byref(gb), C.byref(pr), C.byref(fw),
                                                dY.data_ptr(), _ptr(dL_sum), C.byref(gr), reserve.data_ptr(),
                                                reserve.numel(), ws.data_ptr(), wbytes, _stream(dev)))
        if ctx.has_beta and dbeta is None:
            dbeta = None   # beta unused (no L_prev): TF reports no gradient either
        return dX, dweight, dbias, dM, dalpha, dbeta, dLp, None, None


class GraphGatherFunction(torch.autograd.Function):
    """out[b] = act(sum of graph b's real rows)   (graphgather.py:40-77)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, X, batch, activation):
        lib = _ffi.lib()
        _require_gpu(X, "node features")
        X = _f32c(X, "node features")
        plan = batch.plan
        if X.dim() != 2 or X.shape[0] != plan.rows:
            raise ValueError("node features must be [rows=%d, F]" % plan.rows)
        F = X.shape[1]
        out = torch.empty((plan.B, F), dtype=torch.float32, device=X.device)
        _ffi.check(lib.agcn_graph_gather_forward(plan.host_ptr, plan.dev_ptr, X.data_ptr(), F, F,
                                                 _ffi.ACT[activation], out.data_ptr(), _stream(X.device)))
        ctx.batch, ctx.activation, ctx.F = batch, activation, F
        ctx.save_for_backward(out)
        return out

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dOut):
        lib = _ffi.lib()
        (out,) = ctx.saved_tensors
        plan = ctx.batch.plan
        dOut = _f32c(dOut, "dOut")
        dX = torch.empty((plan.rows, ctx.F), dtype=torch.float32, device=out.device)
        _ffi.check(lib.agcn_graph_gather_backward(plan.host_ptr, plan.dev_ptr, out.data_ptr(), dOut.data_ptr(),
                                                  ctx.F, ctx.F, _ffi.ACT[ctx.activation], dX.data_ptr(),
                                                  _stream(out.device)))
        return dX, None, None


class RowsDenseFunction(torch.autograd.Function):
    """Y = act([scale *] A . W [+ bias] [+ Add]) over the row space: DenseMol (dense_layer.py:33-50),
    BlockEnd (blockend.py:67-86) and, with the device scalar `scale`, one term of DenseBlockEnd
    (densenet_block.py:113-123)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, A, W, bias, Add, batch, activation, keep_pad_rows, scale=None):
        lib = _ffi.lib()
        _require_gpu(A, "node features")
        A, W = _f32c(A, "node features"), _f32c(W, "weight")
        plan = batch.plan
        Fin, Fout = W.shape
        if A.dim() != 2 or A.shape[0] != plan.rows or A.shape[1] < Fin:
            raise ValueError("node features must be [rows=%d, >=%d]" % (plan.rows, Fin))
        if bias is not None:
            bias = _f32c(bias, "bias")
        if Add is not None:
            Add = _f32c(Add, "Add")
            if tuple(Add.shape) != (plan.rows, Fout):
                raise ValueError("Add must be [rows, Fout]")
        if scale is not None:
            scale = _f32c(scale, "scale")
            if scale.numel() != 1:
                raise ValueError("scale must hold one element")
        Y = torch.empty((plan.rows, Fout), dtype=torch.float32, device=A.device)
        desc = _ffi.DenseDesc(Fin, Fout, _ffi.ACT[activation], 1 if keep_pad_rows else 0, A.shape[1], Fout, Fout, current_flags())
        # (point-cloud row counts: the product runs on the split-bf16 row kernels, which build a parameter image per call)
        wbytes = lib.agcn_rows_dense_forward_workspace_bytes(C.byref(desc), plan.host_ptr)
        ws = torch.empty(wbytes, dtype=torch.uint8, device=A.device) if wbytes else None
        _ffi.check(lib.agcn_rows_dense_scaled_forward_ws(C.byref(desc), plan.host_ptr, plan.dev_ptr, A.data_ptr(),
                                                         W.data_ptr(), _ptr(scale), _ptr(bias), _ptr(Add), Y.data_ptr(),
                                                         _ptr(ws), wbytes, _stream(A.device)))
        ctx.batch, ctx.desc, ctx.has_bias, ctx.has_add = batch, desc, bias is not None, Add is not None
        ctx.has_scale = scale is not None
        if scale is not None:
            ctx.save_for_backward(A, W, Y, scale)
        else:
            ctx.save_for_backward(A, W, Y)
        return Y

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dY):
        lib = _ffi.lib()
        saved = ctx.saved_tensors
        A, W, Y = saved[:3]
        scale = saved[3] if ctx.has_scale else None
        plan, desc = ctx.batch.plan, ctx.desc
        dY = _f32c(dY, "dY")
        dA = torch.empty_like(A)
        if A.shape[1] > desc.Fin:
            dA.zero_()
        dW = torch.empty_like(W)
        db = torch.empty(desc.Fout, dtype=torch.float32, device=A.device) if ctx.has_bias else None
        dAdd = torch.empty((plan.rows, desc.Fout), dtype=torch.float32, device=A.device) if ctx.has_add else None
        ds = torch.empty_like(scale) if scale is not None else None
        wbytes = lib.agcn_rows_dense_backward_workspace_bytes(C.byref(desc), plan.host_ptr)
        ws = torch.empty(wbytes, dtype=torch.uint8, device=A.device)
        _ffi.check(lib.agcn_rows_dense_scaled_backward(
            C.byref(desc), plan.host_ptr, plan.dev_ptr, A.data_ptr(), W.data_ptr(), _ptr(scale), Y.data_ptr(),
            dY.data_ptr(), dA.data_ptr(), _ptr(dAdd), dW.data_ptr(), _ptr(db), _ptr(ds), ws.data_ptr(), wbytes,
            _stream(A.device)))
        return dA, dW, db, dAdd, None, None, None, ds


class MultitaskCEFunction(torch.autograd.Function):
    """T two-class heads + the weighted sigmoid cross-entropy training loss
    (model_operatos.py:710-759,792-864; multitask_classifier.py:41-44,187-209).
    Returns (loss [1], logits [T, B, 2]); the gradient flows through the loss only."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, feat, W, bias, labels, weights, batch_size):
        lib = _ffi.lib()
        _require_gpu(feat, "graph features")
        feat, W, bias = _f32c(feat, "graph features"), _f32c(W, "head weights"), _f32c(bias, "head biases")
        labels, weights = _f32c(labels, "labels"), _f32c(weights, "weights")
        T, F = W.shape[0], W.shape[1]
        B = feat.shape[0]
        if feat.dim() != 2 or feat.shape[1] < F or tuple(W.shape) != (T, F, 2) or tuple(bias.shape) != (T, 2):
            raise ValueError("feat [B, >=F], W [T, F, 2], bias [T, 2] expected")
        if tuple(labels.shape) != (B, T) or tuple(weights.shape) != (B, T):
            raise ValueError("labels and weights must be [B, T]")
        dev = feat.device
        logits = torch.empty((T, B, 2), dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        wbytes = lib.agcn_multitask_ce_workspace_bytes(B, F, T)
        ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
        inv = 1.0 / float(batch_size)
        _ffi.check(lib.agcn_multitask_ce_forward(B, F, T, feat.data_ptr(), feat.shape[1], W.data_ptr(), bias.data_ptr(),
                                                 labels.data_ptr(), weights.data_ptr(), inv, logits.data_ptr(),
                                                 loss.data_ptr(), ws.data_ptr(), wbytes, _stream(dev)))
        ctx.dims, ctx.inv = (B, F, T), inv
        ctx.save_for_backward(feat, W, logits, labels, weights)
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dloss, _dlogits):
        lib = _ffi.lib()
        feat, W, logits, labels, weights = ctx.saved_tensors
        B, F, T = ctx.dims
        dev = feat.device
        dloss = _f32c(dloss.reshape(1), "dloss")
        dfeat = torch.empty_like(feat)
        if feat.shape[1] > F:
            dfeat.zero_()
        dW = torch.empty_like(W)
        db = torch.empty((T, 2), dtype=torch.float32, device=dev)
        wbytes = lib.agcn_multitask_ce_workspace_bytes(B, F, T)
        ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
        _ffi.check(lib.agcn_multitask_ce_backward(B, F, T, feat.data_ptr(), feat.shape[1], W.data_ptr(),
                                                  logits.data_ptr(), labels.data_ptr(), weights.data_ptr(), ctx.inv,
                                                  dloss.data_ptr(), dfeat.data_ptr(), dW.data_ptr(), db.data_ptr(),
                                                  ws.data_ptr(), wbytes, _stream(dev)))
        return dfeat, dW, db, None, None, None


class GraphPoolFunction(torch.autograd.Function):
    """GraphPoolMol (graphpool.py:55-110).  grad='reference': no gradient, like the reference's py_func;
    grad='argmax': route the gradient to the arg-max rows (extension)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, X, batch, grad):
        lib = _ffi.lib()
        _require_gpu(X, "node features")
        X = _f32c(X, "node features")
        plan = batch.plan
        if X.dim() != 2 or X.shape[0] != plan.rows:
            raise ValueError("node features must be [rows=%d, F]" % plan.rows)
        F = X.shape[1]
        out = torch.empty((plan.rows, F), dtype=torch.float32, device=X.device)
        argmax = torch.empty((plan.rows, F), dtype=torch.int32, device=X.device) if grad == "argmax" else None
        gb = _gbatch(batch, X, None)
        _ffi.check(lib.agcn_graph_pool_forward(plan.host_ptr, plan.dev_ptr, C.byref(gb), X.data_ptr(), F, F,
                                               out.data_ptr(), _ptr(argmax), _stream(X.device)))
        ctx.batch, ctx.grad, ctx.F = batch, grad, F
        if argmax is not None:
            ctx.save_for_backward(argmax)
        else:
            ctx.mark_non_differentiable(out)     # like tf.py_func: no gradient
        return out

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dOut):
        if ctx.grad != "argmax":
            return None, None, None
        lib = _ffi.lib()
        (argmax,) = ctx.saved_tensors
        plan = ctx.batch.plan
        dOut = _f32c(dOut, "dOut")
        dX = torch.empty((plan.rows, ctx.F), dtype=torch.float32, device=dOut.device)
        gb = _gbatch(ctx.batch, dOut, None)
        _ffi.check(lib.agcn_graph_pool_backward(plan.host_ptr, plan.dev_ptr, C.byref(gb), argmax.data_ptr(),
                                                dOut.data_ptr(), ctx.F, ctx.F, dX.data_ptr(), _stream(dOut.device)))
        return dX, None, None
