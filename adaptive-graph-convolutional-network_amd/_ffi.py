"""ctypes binding of the C ABI in include/agcn.h.  No compute happens in Python: this module
only marshals pointers.  The library is REQUIRED: there is no CPU or torch fallback, and
loading fails loudly if lib/libagcn_hip.so is missing."""
import ctypes as C
import os

from . import _build

_lib = None


class AgcnError(RuntimeError):
    pass


class PlanInfo(C.Structure):
    _fields_ = [("B", C.c_int32), ("N1", C.c_int32), ("rows", C.c_int32), ("max_n", C.c_int32),
                ("N2", C.c_int64), ("tile_elems", C.c_int64), ("plan_bytes", C.c_int64),
                ("off_n", C.c_int64), ("off_row_off", C.c_int64), ("off_tile_off", C.c_int64),
                ("off_order", C.c_int64), ("off_wtile_off", C.c_int64),
                ("n_slabs", C.c_int32), ("reserved1", C.c_int32),
                ("off_slab_graph", C.c_int64), ("off_slab_row", C.c_int64), ("off_row_graph", C.c_int64),
                ("off_graph_desc", C.c_int64), ("off_slab_desc", C.c_int64)]


class LayerDesc(C.Structure):
    _fields_ = [("Fin", C.c_int32), ("Fout", C.c_int32), ("K", C.c_int32), ("mode", C.c_int32),
                ("variant", C.c_int32), ("activation", C.c_int32), ("ldx", C.c_int32), ("ldy", C.c_int32),
                ("feat_dtype", C.c_int32), ("flags", C.c_uint32)]


class GraphBatch(C.Structure):
    _fields_ = [("plan_host", C.c_void_p), ("plan_dev", C.c_void_p), ("X", C.c_void_p),
                ("L0_dense", C.c_void_p), ("L0_ld", C.c_int32), ("L0_symmetric", C.c_int32),
                ("L0_rowptr", C.c_void_p), ("L0_col", C.c_void_p), ("L0_val", C.c_void_p),
                ("L_prev", C.c_void_p), ("L_prev_symmetric", C.c_int32), ("reserved0", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("M_L", C.c_void_p),
                ("alpha", C.c_void_p), ("beta", C.c_void_p)]


class Outputs(C.Structure):
    _fields_ = [("Y", C.c_void_p), ("res_L", C.c_void_p), ("res_W", C.c_void_p), ("L_sum", C.c_void_p)]


class Grads(C.Structure):
    _fields_ = [("dX", C.c_void_p), ("dweight", C.c_void_p), ("dbias", C.c_void_p), ("dM_L", C.c_void_p),
                ("dalpha", C.c_void_p), ("dbeta", C.c_void_p), ("dL_prev", C.c_void_p)]


class DenseDesc(C.Structure):
    _fields_ = [("Fin", C.c_int32), ("Fout", C.c_int32), ("activation", C.c_int32), ("keep_pad_rows", C.c_int32),
                ("ldx", C.c_int32), ("ldy", C.c_int32), ("ldadd", C.c_int32), ("flags", C.c_uint32)]


class GraphStoreC(C.Structure):
    _fields_ = [("n_graphs", C.c_int64), ("F", C.c_int32), ("reserved", C.c_int32), ("n", C.c_void_p),
                ("row_off", C.c_void_p), ("X", C.c_void_p), ("rowptr", C.c_void_p), ("col", C.c_void_p),
                ("val", C.c_void_p), ("symmetric", C.c_void_p)]


class ResidentStoreC(C.Structure):
    _fields_ = [("host", GraphStoreC), ("X_dev", C.c_void_p), ("rowptr_dev", C.c_void_p), ("col_dev", C.c_void_p),
                ("val_dev", C.c_void_p), ("x_bf16", C.c_int32), ("stage_bf16", C.c_int32)]


class EpochBuffers(C.Structure):
    _fields_ = [("stage_host", C.c_void_p), ("stage_dev", C.c_void_p), ("stage_cap", C.c_size_t),
                ("X", C.c_void_p), ("rowptr", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p),
                ("rows_cap", C.c_int64), ("nnz_cap", C.c_int64), ("tiles_cap", C.c_int64),
                ("Y", C.c_void_p), ("res_L", C.c_void_p), ("res_W", C.c_void_p), ("dY", C.c_void_p),
                ("dX", C.c_void_p), ("dweight", C.c_void_p), ("dbias", C.c_void_p), ("dM_L", C.c_void_p), ("dalpha", C.c_void_p),
                ("reserve", C.c_void_p), ("reserve_bytes", C.c_size_t), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class EpochNeeds(C.Structure):
    _fields_ = [("reserve_bytes", C.c_size_t), ("workspace_bytes", C.c_size_t)]


class StagedBatch(C.Structure):
    _fields_ = [("rows", C.c_int64), ("nnz", C.c_int64), ("max_n", C.c_int32), ("symmetric", C.c_int32),
                ("stage_bytes", C.c_int64), ("plan", PlanInfo)]


# AgcnLayerDesc.flags / AgcnDenseDesc.flags (include/agcn.h, AGCN_FLAG_*): per-call dispatch overrides, field = value + 1
_FLAG_FIELDS = {"spec_mode": (0, 15, 1, 0, 7), "fused_mode": (4, 7, 1, 0, 3), "pg_onepass": (7, 3, 1, 0, 2), "spec_small": (9, 3, 1, 0, 2),
                "rows_split": (11, 3, 1, 0, 1), "slab_mode": (13, 3, 1, 0, 1), "fused_gb": (15, 3, 2, -1, 1), "spec_tail": (17, 3, 1, 0, 1),
                "coop_members": (24, 255, 1, 0, 120)}


def dispatch_flags(base=0, **knobs):
    """The flags word that overrides the named dispatch knobs for one call (None = leave the field of `base` alone):
    dispatch_flags(spec_mode=3) = every kernel path as usual but no member workgroups."""
    f = int(base)
    for name, v in knobs.items():
        shift, mask, bias, lo, hi = _FLAG_FIELDS[name]
        if v is None:
            continue
        v = int(v)
        if not lo <= v <= hi:
            raise ValueError("%s = %d outside %d..%d" % (name, v, lo, hi))
        f = (f & ~(mask << shift)) | ((v + bias) << shift)
    return f


MODE = {"reference": 0, "paper": 1}
VARIANT = {"SGC_LL": 0, "SGC_LL_Reslap": 1}
ACT = {None: 0, "linear": 0, "relu": 1, "tanh": 2}
RESERVE_HEADER_BYTES = 65536      # include/agcn.h: AGCN_RESERVE_HEADER_BYTES (what must be zero when a reserve is first used)

# every symbol include/agcn.h declares (tests/test_abi.py checks the two lists agree)
SYMBOLS = [
    "agcn_plan_bytes", "agcn_plan_bytes_for", "agcn_plan_build", "agcn_plan_info",
    "agcn_sgc_ll_reserve_bytes", "agcn_sgc_ll_forward_workspace_bytes", "agcn_sgc_ll_backward_workspace_bytes",
    "agcn_sgc_ll_forward", "agcn_sgc_ll_backward", "agcn_sgc_ll_backward_nodes", "agcn_sgc_ll_backward_params", "agcn_sgc_ll_backward_forked",
    "agcn_sgc_ll_dispatch_info", "agcn_sgc_ll_reserve_status", "agcn_set_device_status_word", "agcn_device_status", "agcn_get_coop_timeout_ticks", "agcn_set_coop_timeout_ticks", "agcn_tiles_asymmetry", "agcn_epoch_step", "agcn_get_spec_tail", "agcn_set_spec_tail", "agcn_get_rows_split", "agcn_set_rows_split",
    "agcn_graph_gather_forward", "agcn_graph_gather_backward",
    "agcn_rows_dense_forward", "agcn_rows_dense_backward_workspace_bytes", "agcn_rows_dense_backward",
    "agcn_rows_dense_scaled_forward", "agcn_rows_dense_scaled_backward", "agcn_rows_dense_forward_workspace_bytes",
    "agcn_rows_dense_scaled_forward_ws",
    "agcn_multitask_ce_workspace_bytes", "agcn_multitask_ce_forward", "agcn_multitask_ce_backward",
    "agcn_graph_pool_forward", "agcn_graph_pool_backward",
    "agcn_store_batch_sizes", "agcn_store_pack_batch", "agcn_store_stage_bytes", "agcn_store_stage_batch",
    "agcn_cast_rows", "agcn_rows_pack", "agcn_rows_unpack",
    "agcn_version", "agcn_last_error", "agcn_get_lds_limit", "agcn_set_lds_limit",
    "agcn_get_slab_mode", "agcn_set_slab_mode", "agcn_get_fused_mode", "agcn_set_fused_mode",
    "agcn_get_fused_global_b", "agcn_set_fused_global_b", "agcn_get_param_grad_onepass", "agcn_set_param_grad_onepass",
    "agcn_get_spec_mode", "agcn_set_spec_mode", "agcn_get_spec_small", "agcn_set_spec_small",
    "agcn_get_coop_members", "agcn_set_coop_members",
    "agcn_profile_enable", "agcn_profile_kernels", "agcn_profile_kernel_name", "agcn_profile_collect",
]


def _declare(lib):
    P, I32, SZ = C.c_void_p, C.c_int32, C.c_size_t
    lib.agcn_plan_bytes.restype = SZ
    lib.agcn_plan_bytes.argtypes = [I32]
    lib.agcn_plan_bytes_for.restype = SZ
    lib.agcn_plan_bytes_for.argtypes = [P, I32, I32]
    lib.agcn_plan_build.restype = C.c_int
    lib.agcn_plan_build.argtypes = [P, I32, I32, P, SZ]
    lib.agcn_plan_info.restype = C.c_int
    lib.agcn_plan_info.argtypes = [P, C.POINTER(PlanInfo)]
    for name in ("agcn_sgc_ll_reserve_bytes", "agcn_sgc_ll_forward_workspace_bytes",
                 "agcn_sgc_ll_backward_workspace_bytes"):
        f = getattr(lib, name)
        f.restype = SZ
        f.argtypes = [C.POINTER(LayerDesc), P]
    lib.agcn_sgc_ll_forward.restype = C.c_int
    lib.agcn_sgc_ll_forward.argtypes = [C.POINTER(LayerDesc), C.POINTER(GraphBatch), C.POINTER(Params),
                                        C.POINTER(Outputs), P, SZ, P, SZ, P]
    for name in ("agcn_sgc_ll_backward", "agcn_sgc_ll_backward_nodes", "agcn_sgc_ll_backward_params"):
        f = getattr(lib, name)
        f.restype = C.c_int
        f.argtypes = [C.POINTER(LayerDesc), C.POINTER(GraphBatch), C.POINTER(Params),
                      C.POINTER(Outputs), P, P, C.POINTER(Grads), P, SZ, P, SZ, P]
    lib.agcn_sgc_ll_backward_forked.restype = C.c_int
    lib.agcn_sgc_ll_backward_forked.argtypes = [C.POINTER(LayerDesc), C.POINTER(GraphBatch), C.POINTER(Params),
                                                C.POINTER(Outputs), P, P, C.POINTER(Grads), P, SZ, P, SZ, P, P, P]
    lib.agcn_graph_gather_forward.restype = C.c_int
    lib.agcn_graph_gather_forward.argtypes = [P, P, P, I32, I32, I32, P, P]
    lib.agcn_graph_gather_backward.restype = C.c_int
    lib.agcn_graph_gather_backward.argtypes = [P, P, P, P, I32, I32, I32, P, P]
    lib.agcn_rows_dense_forward.restype = C.c_int
    lib.agcn_rows_dense_forward.argtypes = [C.POINTER(DenseDesc), P, P, P, P, P, P, P, P]
    lib.agcn_rows_dense_backward_workspace_bytes.restype = SZ
    lib.agcn_rows_dense_backward_workspace_bytes.argtypes = [C.POINTER(DenseDesc), P]
    lib.agcn_rows_dense_backward.restype = C.c_int
    lib.agcn_rows_dense_backward.argtypes = [C.POINTER(DenseDesc), P, P, P, P, P, P, P, P, P, P, P, SZ, P]
    lib.agcn_rows_dense_scaled_forward.restype = C.c_int
    lib.agcn_rows_dense_scaled_forward.argtypes = [C.POINTER(DenseDesc), P, P, P, P, P, P, P, P, P]
    lib.agcn_rows_dense_forward_workspace_bytes.restype = SZ
    lib.agcn_rows_dense_forward_workspace_bytes.argtypes = [C.POINTER(DenseDesc), P]
    lib.agcn_rows_dense_scaled_forward_ws.restype = C.c_int
    lib.agcn_rows_dense_scaled_forward_ws.argtypes = [C.POINTER(DenseDesc), P, P, P, P, P, P, P, P, P, SZ, P]
    lib.agcn_rows_dense_scaled_backward.restype = C.c_int
    lib.agcn_rows_dense_scaled_backward.argtypes = [C.POINTER(DenseDesc), P, P, P, P, P, P, P, P, P, P, P, P, P, SZ, P]
    lib.agcn_multitask_ce_workspace_bytes.restype = SZ
    lib.agcn_multitask_ce_workspace_bytes.argtypes = [I32, I32, I32]
    lib.agcn_multitask_ce_forward.restype = C.c_int
    lib.agcn_multitask_ce_forward.argtypes = [I32, I32, I32, P, I32, P, P, P, P, C.c_float, P, P, P, SZ, P]
    lib.agcn_multitask_ce_backward.restype = C.c_int
    lib.agcn_multitask_ce_backward.argtypes = [I32, I32, I32, P, I32, P, P, P, P, C.c_float, P, P, P, P, P, SZ, P]
    lib.agcn_rows_pack.restype = C.c_int
    lib.agcn_rows_pack.argtypes = [P, P, P, I32, I32, I32, P, I32, P]
    lib.agcn_rows_unpack.restype = C.c_int
    lib.agcn_rows_unpack.argtypes = [P, P, P, I32, I32, P, I32, I32, P, P]
    lib.agcn_cast_rows.restype = C.c_int
    lib.agcn_cast_rows.argtypes = [P, I32, P, I32, C.c_int64, I32, I32, I32, P]
    lib.agcn_store_batch_sizes.restype = C.c_int
    lib.agcn_store_batch_sizes.argtypes = [C.POINTER(GraphStoreC), P, I32, P, P, P]
    lib.agcn_store_pack_batch.restype = C.c_int
    lib.agcn_store_pack_batch.argtypes = [C.POINTER(GraphStoreC), P, I32, P, P, I32, P, P, P, P, I32]
    lib.agcn_sgc_ll_dispatch_info.restype = C.c_int
    lib.agcn_sgc_ll_dispatch_info.argtypes = [C.POINTER(LayerDesc), C.POINTER(GraphBatch), P, I32]
    lib.agcn_store_stage_bytes.restype = SZ
    lib.agcn_store_stage_bytes.argtypes = [I32, C.c_int64]
    lib.agcn_store_stage_batch.restype = C.c_int
    lib.agcn_store_stage_batch.argtypes = [C.POINTER(ResidentStoreC), P, I32, P, P, SZ, P, I32, C.c_int64, P, P, P,
                                           C.c_int64, C.POINTER(StagedBatch), P]
    lib.agcn_graph_pool_forward.restype = C.c_int
    lib.agcn_graph_pool_forward.argtypes = [P, P, C.POINTER(GraphBatch), P, I32, I32, P, P, P]
    lib.agcn_graph_pool_backward.restype = C.c_int
    lib.agcn_graph_pool_backward.argtypes = [P, P, C.POINTER(GraphBatch), P, P, I32, I32, P, P]
    lib.agcn_version.restype = I32
    lib.agcn_version.argtypes = []
    lib.agcn_last_error.restype = C.c_char_p
    lib.agcn_last_error.argtypes = []
    lib.agcn_get_lds_limit.restype = I32
    lib.agcn_get_lds_limit.argtypes = []
    lib.agcn_set_lds_limit.restype = None
    lib.agcn_set_lds_limit.argtypes = [I32]
    lib.agcn_get_slab_mode.restype = I32
    lib.agcn_get_slab_mode.argtypes = []
    lib.agcn_set_slab_mode.restype = None
    lib.agcn_set_slab_mode.argtypes = [I32]
    lib.agcn_sgc_ll_reserve_status.restype = C.c_int
    lib.agcn_sgc_ll_reserve_status.argtypes = [P, SZ, P, C.POINTER(C.c_int32)]
    lib.agcn_set_device_status_word.restype = C.c_int
    lib.agcn_set_device_status_word.argtypes = [I32, P]
    lib.agcn_device_status.restype = C.c_int
    lib.agcn_device_status.argtypes = [I32, P, C.POINTER(C.c_int32)]
    lib.agcn_epoch_step.restype = C.c_int
    lib.agcn_epoch_step.argtypes = [P, P, I32, P, P, P, I32, P, P, P, P]
    lib.agcn_tiles_asymmetry.restype = C.c_int
    lib.agcn_tiles_asymmetry.argtypes = [P, P, P, I32, P, P]
    lib.agcn_get_spec_tail.restype = I32
    lib.agcn_get_spec_tail.argtypes = []
    lib.agcn_set_spec_tail.restype = None
    lib.agcn_set_spec_tail.argtypes = [I32]
    lib.agcn_get_rows_split.restype = I32
    lib.agcn_get_rows_split.argtypes = []
    lib.agcn_set_rows_split.restype = None
    lib.agcn_set_rows_split.argtypes = [I32]
    lib.agcn_get_coop_timeout_ticks.restype = C.c_uint32
    lib.agcn_get_coop_timeout_ticks.argtypes = []
    lib.agcn_set_coop_timeout_ticks.restype = None
    lib.agcn_set_coop_timeout_ticks.argtypes = [C.c_uint32]
    for nm in ("agcn_get_fused_global_b", "agcn_get_param_grad_onepass", "agcn_get_spec_mode", "agcn_get_spec_small",
               "agcn_get_coop_members"):
        getattr(lib, nm).restype = I32
        getattr(lib, nm).argtypes = []
    for nm in ("agcn_set_fused_global_b", "agcn_set_param_grad_onepass", "agcn_set_spec_mode", "agcn_set_spec_small",
               "agcn_set_coop_members"):
        getattr(lib, nm).restype = None
        getattr(lib, nm).argtypes = [I32]
    lib.agcn_get_fused_mode.restype = I32
    lib.agcn_get_fused_mode.argtypes = []
    lib.agcn_set_fused_mode.restype = None
    lib.agcn_set_fused_mode.argtypes = [I32]
    lib.agcn_profile_enable.restype = None
    lib.agcn_profile_enable.argtypes = [C.c_int]
    lib.agcn_profile_kernels.restype = I32
    lib.agcn_profile_kernels.argtypes = []
    lib.agcn_profile_kernel_name.restype = C.c_char_p
    lib.agcn_profile_kernel_name.argtypes = [I32]
    lib.agcn_profile_collect.restype = C.c_int
    lib.agcn_profile_collect.argtypes = [P, P]


def lib():
    """The loaded C-ABI library.  torch must be imported first so that the HIP runtime that
    is already in the process (torch's libamdhip64.so.7) satisfies the library's dependency."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (loads libamdhip64 before our .so resolves it)
        path = os.environ.get("AGCN_HIP_LIB", _build.LIB_PATH)
        if not os.path.exists(path):
            raise AgcnError(
                "HIP library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % path)
        l = C.CDLL(path)
        _declare(l)
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise AgcnError("agcn error %d: %s" % (rc, lib().agcn_last_error().decode()))


def reserve_status(reserve, stream=None):
    """Waits for `stream` (default: the current one of the reserve's device) and raises AgcnError if a cooperative wait of
    any step that used this reserve tensor ran out (include/agcn.h: agcn_sgc_ll_reserve_status).  Returns 0 otherwise."""
    import torch
    with torch.cuda.device(reserve.device):
        st = (torch.cuda.current_stream(reserve.device) if stream is None else stream).cuda_stream
        n = C.c_int32(0)
        check(lib().agcn_sgc_ll_reserve_status(reserve.data_ptr(), reserve.numel() * reserve.element_size(), st, C.byref(n)))
    return n.value


def dispatch_info(layer, batch):
    """agcn_sgc_ll_dispatch_info for `layer` on `batch` (host logic only): list of AGCN_DISPATCH_INFO_INTS ints;
    [0] = path bits (4 fused | 1 / 2 specialised fwd / bwd, 8 mixed, 16 slab, 32 per-graph, 64 cloud kernels)."""
    from .functional import _Cfg, _desc, _gbatch
    cfg = _Cfg(layer.n_atom_feature, layer.nb_filter, layer.K, layer.laplacian_mode, layer._variant,
               "relu" if layer.activation == "relu" else "linear")
    desc = _desc(cfg, batch.X.shape[1], cfg.Fout)
    gb = _gbatch(batch, batch.X, None)
    out = (C.c_int32 * 8)()
    check(lib().agcn_sgc_ll_dispatch_info(C.byref(desc), C.byref(gb), out, 8))
    return list(out)


def profile_collect():
    """{kernel name: (total_ms, launches)} since the last collect (needs agcn_profile_enable(1))."""
    import numpy as np
    l = lib()
    k = l.agcn_profile_kernels()
    ms = np.zeros(k, dtype=np.float64)
    calls = np.zeros(k, dtype=np.int64)
    check(l.agcn_profile_collect(ms.ctypes.data, calls.ctypes.data))
    return {l.agcn_profile_kernel_name(i).decode(): (float(ms[i]), int(calls[i])) for i in range(k)}
