"""C-ABI library: loads without a GPU, exports every symbol include/agcn.h declares, and its
host-only entry points (plan building, size queries, argument validation) behave.  CPU only:
no kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def amd():
    import agcn_amd
    if agcn_amd._build.is_stale():
        agcn_amd._build.build()
    return agcn_amd


def test_exports_every_declared_symbol(amd):
    hdr = open(os.path.join(ROOT, "include", "agcn.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(agcn_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    lib = amd._ffi.lib()
    for name in declared:
        assert hasattr(lib, name), "libagcn_hip.so does not export %s" % name
    assert sorted(amd._ffi.SYMBOLS) == declared
    assert lib.agcn_version() == 107


def test_dynamic_symbol_table_is_the_header_and_nothing_else(amd):
    """the boundary is the C functions of include/agcn.h.  The library is built with -fvisibility=hidden and
    every declaration carries AGCN_API, so `nm -D --defined-only` lists exactly those names: no mangled C++ symbol, no kernel
    stub, no global."""
    import subprocess
    hdr = open(os.path.join(ROOT, "include", "agcn.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(agcn_[a-z_0-9]+)\s*\(", hdr)))
    out = subprocess.run(["nm", "-D", "--defined-only", amd._build.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    extra = [e for e in exported if e not in declared]
    assert not extra, "symbols exported beyond include/agcn.h: %s" % extra[:10]
    assert exported == declared
    assert all(re.search(r"AGCN_API [^;]*\b%s\(" % n, hdr) for n in declared), "a declaration without AGCN_API"


def test_header_constants_match_the_c_header(amd):
    """Constants the Python side restates from include/agcn.h."""
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "agcn.h")).read()
    m = re.search(r"#define\s+AGCN_RESERVE_HEADER_BYTES\s+(\d+)", text)
    assert m and int(m.group(1)) == amd._ffi.RESERVE_HEADER_BYTES


def test_plan_packed(amd):
    n = np.array([5, 3, 9, 4, 9], dtype=np.int32)
    p = amd.Plan(n)
    assert p.B == 5 and p.N1 == 30 and p.rows == 30 and p.max_n == 9 and p.N2 == int((n.astype(np.int64) ** 2).sum())
    np.testing.assert_array_equal(p.n, n)
    np.testing.assert_array_equal(p.row_off, np.concatenate([[0], np.cumsum(n)]))
    assert p.row_off.dtype == np.int32 and p.tile_off.dtype == np.int64
    # tiles: 16-byte aligned starts, non-overlapping
    assert np.all(p.tile_off % 4 == 0)
    assert np.all(np.diff(p.tile_off) >= n.astype(np.int64) ** 2)
    # schedule: largest first, stable
    np.testing.assert_array_equal(p.order, [2, 4, 0, 3, 1])


def test_plan_padded_matches_reference_slices(amd):
    # graph_topology.py:84-98: every graph owns max_atom rows; data_slice rows are [n, -1]
    n = np.array([4, 9, 5, 17], dtype=np.int32)
    p = amd.Plan(n, 20)
    np.testing.assert_array_equal(p.row_off, np.arange(5) * 20)
    assert p.rows == 80 and p.N1 == 35


def test_plan_rejects_bad_input(amd):
    with pytest.raises(amd._ffi.AgcnError):
        amd.Plan([4, 0, 3])
    with pytest.raises(amd._ffi.AgcnError):
        amd.Plan([4, 30], 20)        # graph larger than max_atom
    with pytest.raises(amd._ffi.AgcnError):
        amd.Plan([5000])             # beyond AGCN_MAX_NODES
    lib = amd._ffi.lib()
    info = amd._ffi.PlanInfo()
    junk = np.zeros(256, dtype=np.uint8)
    assert lib.agcn_plan_info(junk.ctypes.data, C.byref(info)) == -4
    assert b"magic" in lib.agcn_last_error()


def test_size_queries_and_validation(amd):
    lib = amd._ffi.lib()
    p = amd.Plan([18, 48, 7, 23])
    d = amd._ffi.LayerDesc(75, 64, 2, 1, 0, 1, 75, 64, 0, 0)
    r = lib.agcn_sgc_ll_reserve_bytes(C.byref(d), p.host_ptr)
    assert r >= 4 * (p.rows + 4 * p.B + p.rows * 75)
    assert lib.agcn_sgc_ll_forward_workspace_bytes(C.byref(d), p.host_ptr) >= 256
    wb = lib.agcn_sgc_ll_backward_workspace_bytes(C.byref(d), p.host_ptr)
    assert wb >= 4 * (p.rows * 75 * 2 + 2 * 75 * 64)
    # forcing the global-tile path needs scratch for the tiles
    lib.agcn_set_lds_limit(0)
    try:
        assert lib.agcn_sgc_ll_forward_workspace_bytes(C.byref(d), p.host_ptr) >= 4 * int(p.N2)
    finally:
        lib.agcn_set_lds_limit(160 * 1024)
    assert lib.agcn_get_lds_limit() == 160 * 1024
    bad = amd._ffi.LayerDesc(75, 64, 0, 1, 0, 1, 75, 64, 0, 0)     # K = 0
    assert lib.agcn_sgc_ll_reserve_bytes(C.byref(bad), p.host_ptr) == 0
    assert b"K must be" in lib.agcn_last_error()
    # argument validation happens before any device work
    gb = amd._ffi.GraphBatch()
    rc = lib.agcn_sgc_ll_forward(C.byref(d), C.byref(gb), C.byref(amd._ffi.Params()), C.byref(amd._ffi.Outputs()),
                                 None, 0, None, 0, None)
    assert rc == -1


def test_graph_descriptor_table(amd):
    """AgcnGraphDesc records, in schedule order (largest graph first), agree with the plain plan arrays."""
    n = np.array([5, 3, 9, 4, 9, 1], dtype=np.int32)
    p = amd.Plan(n)
    rec = np.dtype([("graph", "<i4"), ("n", "<i4"), ("row0", "<i4"), ("reserved", "<i4"), ("tile_off", "<i8"),
                    ("tile_end", "<i8")])
    assert rec.itemsize == 32
    gd = p.host[p.info.off_graph_desc:p.info.off_graph_desc + 32 * p.B].view(rec)
    np.testing.assert_array_equal(gd["graph"], p.order)
    np.testing.assert_array_equal(gd["n"], n[p.order])
    np.testing.assert_array_equal(gd["row0"], p.row_off[p.order])
    np.testing.assert_array_equal(gd["tile_off"], p.tile_off[p.order])
    np.testing.assert_array_equal(gd["tile_end"], p.tile_off[p.order + 1])
    assert list(n[p.order]) == sorted(n, reverse=True)


def test_slab_descriptor_table(amd):
    """AgcnSlabDesc records (ABI 107), in slab order, agree with slab_graph / slab_row and the per-graph plan arrays; the table lies
    inside the blob agcn_plan_bytes_for sized."""
    n = np.array([5, 33, 9, 64, 17, 1, 130], dtype=np.int32)
    p = amd.Plan(n)
    rec = np.dtype([("graph", "<i4"), ("row", "<i4"), ("n", "<i4"), ("row0", "<i4"), ("tile_off", "<i8"), ("tile_end", "<i8")])
    assert rec.itemsize == 32
    off = p.info.off_slab_desc
    assert off > 0 and off % 16 == 0 and off + 32 * p.n_slabs <= p.info.plan_bytes <= len(p.host)
    sd = p.host[off:off + 32 * p.n_slabs].view(rec)
    assert p.n_slabs == int(((n + 15) // 16).sum())
    np.testing.assert_array_equal(sd["graph"], p.slab_graph)
    np.testing.assert_array_equal(sd["row"], p.slab_row)
    np.testing.assert_array_equal(sd["n"], n[p.slab_graph])
    np.testing.assert_array_equal(sd["row0"], p.row_off[p.slab_graph])
    np.testing.assert_array_equal(sd["tile_off"], p.tile_off[p.slab_graph])
    np.testing.assert_array_equal(sd["tile_end"], p.tile_off[p.slab_graph + 1])
    # slabs of one graph are consecutive, rows 0, 16, ..; graphs in schedule order (largest first)
    assert list(sd["row"][sd["graph"] == 6]) == list(range(0, 130, 16))
    assert sd["graph"][0] == 6 and sd["graph"][-1] == 5


def test_new_entry_points_validate_before_launching(amd):
    """Host-side argument checks of the tail / ingest / layout entry points (no kernel is launched on an error)."""
    lib = amd._ffi.lib()
    # classifier tail
    assert lib.agcn_multitask_ce_workspace_bytes(0, 16, 3) == 0
    assert lib.agcn_multitask_ce_workspace_bytes(256, 256, 12) > 0
    rc = lib.agcn_multitask_ce_forward(8, 16, 3, None, 16, None, None, None, None, C.c_float(0.125), None, None, None, 0, None)
    assert rc != 0 and b"null tensor" in lib.agcn_last_error()
    # row layout conversion needs a packed plan and a stride that holds the largest graph
    padded = amd.Plan(np.array([3, 5], np.int32), 8)
    packed = amd.Plan(np.array([3, 5], np.int32))
    buf = np.zeros(1024, np.float32)
    rc = lib.agcn_rows_pack(padded.host_ptr, padded.host_ptr, buf.ctypes.data, 8, 4, 4, buf.ctypes.data, 4, None)
    assert rc != 0 and b"packed plan" in lib.agcn_last_error()
    rc = lib.agcn_rows_pack(packed.host_ptr, packed.host_ptr, buf.ctypes.data, 4, 4, 4, buf.ctypes.data, 4, None)
    assert rc != 0 and b"stride" in lib.agcn_last_error()
    # storage conversion: only f32 <-> bf16
    rc = lib.agcn_cast_rows(buf.ctypes.data, 0, buf.ctypes.data, 0, 4, 4, 4, 4, None)
    assert rc != 0 and b"unsupported conversion" in lib.agcn_last_error()
    # scaled dense: scale and dscale go together
    d = amd._ffi.DenseDesc(4, 4, 0, 0, 4, 4, 4, 0)
    rc = lib.agcn_rows_dense_scaled_backward(C.byref(d), packed.host_ptr, packed.host_ptr, buf.ctypes.data, buf.ctypes.data,
                                             buf.ctypes.data, buf.ctypes.data, buf.ctypes.data, buf.ctypes.data, None,
                                             buf.ctypes.data, None, None, buf.ctypes.data, buf.nbytes, None)
    assert rc != 0 and b"go together" in lib.agcn_last_error()
    # knobs round-trip
    for setter, getter, values in ((lib.agcn_set_fused_global_b, lib.agcn_get_fused_global_b, (0, 1, -1)),
                                   (lib.agcn_set_param_grad_onepass, lib.agcn_get_param_grad_onepass, (0, 1, 2)),
                                   (lib.agcn_set_fused_mode, lib.agcn_get_fused_mode, (0, 2, 3, 1)),
                                   (lib.agcn_set_spec_mode, lib.agcn_get_spec_mode, (0, 1, 2, 3, 7)),
                                   (lib.agcn_set_spec_small, lib.agcn_get_spec_small, (0, 1, 2)),
                                   (lib.agcn_set_rows_split, lib.agcn_get_rows_split, (0, 1))):
        for v in values:
            setter(v)
            assert getter() == v


def test_missing_library_fails_loudly(amd, monkeypatch):
    monkeypatch.setattr(amd._ffi, "_lib", None)
    monkeypatch.setenv("AGCN_HIP_LIB", "/nonexistent/libagcn_hip.so")
    with pytest.raises(amd._ffi.AgcnError, match="no CPU fallback"):
        amd._ffi.lib()
