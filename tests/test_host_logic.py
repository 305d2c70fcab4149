"""Host-side product code (graph preprocessing, batch packing, layer boundary) on CPU."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def amd():
    import agcn_amd
    if agcn_amd._build.is_stale():
        agcn_amd._build.build()
    return agcn_amd


def _adj(g):
    ptr, flat = g["adj_ptr"], g["adj_flat"]
    return [list(map(int, flat[ptr[i]:ptr[i + 1]])) for i in range(len(ptr) - 1)]


LAP = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "lap_*.npz")))


@pytest.mark.parametrize("name", LAP)
def test_molgraph_matches_reference(amd, name):
    """agcn_amd.MolGraph vs the reference's MolGraph (graph_structure.py:22-55,85-130):
    integer outputs bit exact, float64 values to 1e-15."""
    g = dict(np.load(os.path.join(GOLDEN, name)))
    adj = _adj(g)
    n = len(adj)
    mg = amd.MolGraph(np.zeros((n, 3), np.float32), adj)
    np.testing.assert_array_equal(mg.degree_list, g["degree_list"])
    assert mg.degree_list.dtype == np.int32
    assert mg.has_Lap == bool(g["has_lap"])
    assert (mg.n_node, mg.n_feat, mg.max_deg, mg.min_deg) == (n, 3, 10, 0)
    if not mg.has_Lap:
        assert mg.Laplacian is None
        return
    L = mg.Laplacian
    assert L.dtype == np.float64 and L.format == "csr"
    np.testing.assert_array_equal(L.indptr, g["indptr"])
    np.testing.assert_array_equal(L.indices, g["sorted_indices"])
    np.testing.assert_allclose(L.data, g["sorted_data"], rtol=1e-15, atol=0)
    np.testing.assert_allclose(np.asarray(L.todense()), g["dense"], rtol=1e-15, atol=0)


def test_graph_batch_packing(amd):
    import agcn_oracle as orc
    sb = orc.synth_batch(5, 6)
    graphs = [amd.MolGraph(x, a) for x, a in zip(sb["X"], sb["adj"])]
    gb = amd.GraphBatch.from_graphs(graphs, "cpu")
    p = gb.plan
    assert p.B == 6 and p.rows == p.N1 == int(sb["ns"].sum())
    np.testing.assert_array_equal(gb.X.numpy(), np.concatenate(sb["X"], 0))
    rp, col, val = gb.l0_rowptr.numpy(), gb.l0_col.numpy(), gb.l0_val.numpy()
    assert rp.dtype == np.int32 and col.dtype == np.int32 and val.dtype == np.float32
    for b in range(p.B):                      # CSR over the row space reproduces each dense L (as f32)
        n, r0 = int(p.n[b]), int(p.row_off[b])
        dense = np.zeros((n, n), np.float32)
        for i in range(n):
            for q in range(rp[r0 + i], rp[r0 + i + 1]):
                dense[i, col[q]] = val[q]
        np.testing.assert_array_equal(dense, sb["L0"][b])


def test_padded_feed_batch(amd):
    g = dict(np.load(os.path.join(GOLDEN, "topo_tox4.npz")))
    Xl = [torch.from_numpy(x) for x in g["X"]]
    Ll = [torch.from_numpy(l) for l in g["L"]]
    gb = amd.GraphBatch.from_padded(Xl, Ll, g["data_slice"])
    np.testing.assert_array_equal(gb.plan.n, g["ns"])
    assert gb.plan.row_stride == int(g["max_atom"]) and gb.l0_ld == int(g["max_atom"])
    views = gb.rows(gb.X)
    assert len(views) == 4 and tuple(views[1].shape) == (20, 75)
    assert torch.equal(views[2], Xl[2])


def test_graph_store_batches(amd):
    """Native packed-batch ingest (agcn_store_pack_batch): a batch gathered from the dataset-level store is
    identical, array for array, to packing the same graphs one by one; symmetry flags are exact; the batching
    rule is DiskDataset.iterbatches' (diskdataset.py:263-283)."""
    import scipy.sparse as sp
    from agcn_amd import synth
    data = synth.tox21_batch(5, 70)
    n, X, csr = data["n"], data["X"], list(data["csr"])
    ip, ix, v = csr[3]                      # one graph made asymmetric in value, one in structure
    v = v.copy()
    v[1] += 1e-3
    csr[3] = (ip, ix, v)
    m = sp.csr_matrix((csr[5][2], csr[5][1], csr[5][0]), shape=(n[5], n[5])).tolil()
    i, j = m.nonzero()
    k = [t for t in range(len(i)) if i[t] != j[t]][0]
    m[i[k], j[k]] = 0
    m = m.tocsr()
    m.eliminate_zeros()
    csr[5] = (m.indptr, m.indices, m.data)
    y = np.arange(70 * 3).reshape(70, 3)
    store = amd.GraphStore.from_arrays(n, X, csr, y=y, w=y * 0.5)
    assert len(store) == 70 and store.F == 75
    want = np.ones(70, np.uint8)
    want[[3, 5]] = 0
    np.testing.assert_array_equal(store.symmetric, want)
    for ids in (np.arange(16), np.array([9, 2, 69, 0, 33]), np.array([3]), np.random.RandomState(1).permutation(70)):
        gb = store.batch(ids, "cpu", n_threads=3)
        ref = amd.GraphBatch.from_arrays(n[ids], np.concatenate([X[i] for i in ids], 0), [csr[i] for i in ids], "cpu")
        for key in ("X", "l0_rowptr", "l0_col", "l0_val"):
            assert torch.equal(getattr(gb, key), getattr(ref, key)), key
        np.testing.assert_array_equal(gb.plan.n, ref.plan.n)
        assert gb.l0_symmetric == ref.l0_symmetric == (not ({3, 5} & set(ids.tolist())))
    with pytest.raises(amd.AgcnError):
        store.batch(np.array([0, 70]), "cpu")
    # iterbatches: deterministic order, np.linspace cut points, every sample exactly once
    seen, sizes = [], []
    for gb, yb, wb, idb in store.iterbatches(16, "cpu", deterministic=True):
        assert gb.plan.B == len(idb) and np.array_equal(yb, y[idb]) and np.array_equal(wb, y[idb] * 0.5)
        seen += idb.tolist()
        sizes.append(len(idb))
    pts = np.linspace(0, 70, int(np.ceil(70 / 16.0)) + 1, dtype=int)
    assert seen == list(range(70)) and sizes == np.diff(pts).tolist()
    shuffled = [idb for _, _, _, idb in store.iterbatches(16, "cpu", rng=np.random.RandomState(4))]
    assert sorted(np.concatenate(shuffled).tolist()) == list(range(70))


def test_layer_contract(amd):
    """Names, constructor arguments and read-back attributes the reference's containers rely on
    (tf_graphs.py:41-73,173-194; basic_layer.py:57-67)."""
    a = amd.SGC_LL(64, 75, 256)
    b = amd.SGC_LL_Reslap(128, 64, 256, activation="relu", dropout=None, K=2, save_lap=True, save_output=False)
    gth = amd.GraphGatherMol(256, activation="tanh")
    assert [type(x).__name__ for x in (a, b, gth)] == ["SGC_LL", "SGC_LL_Reslap", "GraphGatherMol"]
    assert a.nb_filter == 64 and a.n_atom_feature == 75 and a.K == 2 and a.batch_size == 256
    assert b.save_lap is True and b.save_output is False and a.save_lap is False
    assert a.name.startswith("sgc_ll_") and b.name.startswith("sgc_ll_reslap_")
    n1 = int(amd.SGC_LL(4, 4, 1).name.rsplit("_", 1)[1])
    assert int(amd.SGC_LL(4, 4, 1).name.rsplit("_", 1)[1]) == n1 + 1     # uid counter per class prefix
    with pytest.raises(TypeError):
        amd.SGC_LL(64, 75, 256, foo=1)                                      # kwargs whitelist
    assert amd.SGC_LL(4, 4, 1, name="x", trainable=False).trainable is False
    a.build()
    assert set(a.vars) == {"weight", "bias", "M_L", "alpha"} and tuple(a.vars["weight"].shape) == (150, 64)
    b.build()
    assert set(b.vars) == {"weight", "bias", "M_L", "alpha", "beta"}
    r = np.sqrt(6.0 / (150 + 64))
    assert float(a.vars["weight"].detach().abs().max()) <= r and float(a.vars["bias"].detach().abs().max()) == 0
    assert float(a.vars["alpha"][0]) == 1.0 and float(b.vars["beta"][0]) == 1.0


def test_no_cpu_fallback(amd):
    import agcn_oracle as orc
    sb = orc.synth_batch(5, 2)
    gb = amd.GraphBatch.from_arrays(sb["ns"], np.concatenate(sb["X"], 0), sb["csr"], "cpu")
    layer = amd.SGC_LL(8, 75, 2)
    with pytest.raises(amd._ffi.AgcnError, match="no CPU fallback"):
        layer({"node_features": gb.X, "graph_batch": gb})


def test_product_does_not_import_oracle():
    """The shipped package must not depend on oracle/ in any way."""
    from conftest import ROOT
    pkg = os.path.join(ROOT, "adaptive-graph-convolutional-network_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "agcn_oracle" not in src and "oracle/" not in src, f


def test_stage_blob_bound_covers_every_split_of_the_rows(amd):
    """agcn_store_stage_bytes(B, rows) -- the size the resident store allocates its pinned / device blobs with -- must
    cover the plan + gather table of ANY batch of at most B graphs and `rows` rows, however unevenly the rows are
    split (the slab tables grow with sum(ceil(n / 16)))."""
    lib = amd._ffi.lib()
    rng = np.random.RandomState(0)
    for B, rows in ((1, 1), (1, 1600), (7, 7), (256, 4617), (256, 25000), (64, 64 * 1600)):
        cap = lib.agcn_store_stage_bytes(B, rows)
        assert cap > 0
        for trial in range(20):
            b = int(rng.randint(1, B + 1))
            # a random split of at most `rows` rows into b graphs of 1..1600 nodes
            n = np.ones(b, dtype=np.int64)
            budget = min(rows, 1600 * b) - b
            extra = rng.multinomial(budget, rng.dirichlet(np.full(b, 0.3))) if budget > 0 else np.zeros(b, np.int64)
            n = np.minimum(n + extra, 1600).astype(np.int32)
            need = lib.agcn_plan_bytes_for(n.ctypes.data, b, 0)
            assert need > 0
            assert (need + 15) // 16 * 16 + 24 * b <= cap, (B, rows, b, int(n.sum()))
    assert lib.agcn_store_stage_bytes(0, 10) == 0 and lib.agcn_store_stage_bytes(4, 3) == 0


def _dispatch(amd, n, Fin, Fout, mode, variant=0, K=2, symmetric=1, csr=True, flags=0):
    """agcn_sgc_ll_dispatch_info on a packed plan of the node counts `n` (host logic: no GPU needed)."""
    import ctypes as C
    lib = amd._ffi.lib()
    n = np.ascontiguousarray(n, dtype=np.int32)
    nbytes = lib.agcn_plan_bytes_for(n.ctypes.data, n.size, 0)
    plan = np.zeros(nbytes, np.uint8)
    assert lib.agcn_plan_build(n.ctypes.data, n.size, 0, plan.ctypes.data, nbytes) == 0
    desc = amd._ffi.LayerDesc(Fin, Fout, K, mode, variant, 1, Fin, Fout, 0, flags)
    one = 1   # any non-NULL "device" pointer: the call only tests them for NULL
    gb = amd._ffi.GraphBatch(plan.ctypes.data, one, one, None, 0, symmetric, one if csr else None, one if csr else None,
                             one if csr else None, None)
    out = (C.c_int32 * 8)()
    assert lib.agcn_sgc_ll_dispatch_info(C.byref(desc), C.byref(gb), out, 8) == 0, lib.agcn_last_error()
    d = dict(zip(("path", "small_first", "f_graphs", "f_members", "f_paired", "b_graphs", "b_members", "b_paired"), list(out)))
    d["bf16"] = bool(d["path"] & 256)      # (bit 256: AgcnLayerDesc.feat_dtype = bf16 is served on this batch)
    d["path"] &= ~256
    return d


def test_dispatch_table(amd):
    """Which kernels a batch takes, decided on the host: checked without a GPU on the batch shapes the
    benchmarks use."""
    from agcn_amd import synth
    rs = np.random.RandomState(0)
    tox = lambda B: synth.node_counts(np.random.RandomState(123), B)   # noqa: E731
    n256 = tox(256)
    big, tiny = int((n256 >= 33).sum()), int((n256 <= 16).sum())
    members = int(np.ceil(n256[n256 >= 33] / 16.0).sum())
    # the headline batch: specialised kernels, every graph >= 33 nodes spread over members, exactly enough pairs of
    # small graphs for the launch to fit 256 CUs, the same plan for the forward and the backward
    # (| 128: paper mode, the whole batch in one 16-wave launch -> the backward launch carries the parameter gradients,
    #  csrc/spec_tail.h)
    prev_tail = amd._ffi.lib().agcn_get_spec_tail()
    amd._ffi.lib().agcn_set_spec_tail(1)
    d = _dispatch(amd, n256, 75, 64, 1)
    assert d["path"] == 7 | 128 and d["small_first"] == 256 and d["bf16"]
    assert (d["f_graphs"], d["f_members"]) == (big, members) == (d["b_graphs"], d["b_members"])
    assert d["f_paired"] == d["b_paired"] == 2 * (256 + members - big - 256) and d["f_paired"] <= tiny
    # reference mode: members in the forward only
    d = _dispatch(amd, n256, 75, 64, 0)
    assert d["path"] == 7 and d["f_members"] == members and d["b_members"] == 0 and d["b_paired"] == 0
    # half a chip of graphs: members, nobody paired
    d = _dispatch(amd, tox(128), 75, 64, 1)
    assert d["path"] == 7 | 128 and d["f_members"] > 0 and d["f_paired"] == 0
    # two rounds of workgroups per CU: no members, small graphs in pairs
    d = _dispatch(amd, tox(600), 75, 64, 1)       # (600 producer units: beyond what a tail workgroup keeps in registers)
    assert d["path"] == 7 and d["small_first"] == 600 and d["f_members"] == 0 and d["f_paired"] > 0
    # 1024 graphs and up: 8-wave launch beside the 16-wave one, nothing else
    d = _dispatch(amd, tox(2048), 75, 64, 1)
    assert d["path"] == 7 and 0 < d["small_first"] < 2048 and d["f_members"] == d["f_paired"] == 0
    # a full batch without enough small graphs: only graphs of five tiles and up are spread
    n = np.concatenate([np.full(250, 40), [96, 80, 64, 40, 40, 40]]).astype(np.int32)
    d = _dispatch(amd, n, 75, 64, 1)
    assert d["path"] == 7 | 128 and d["f_graphs"] == 2 and d["f_members"] == 6 + 5 and d["f_paired"] == 0
    # a molecule beyond the specialised kernels' 96 nodes: the whole batch on the slab kernels (general fused kernels
    # when it is small enough for them); other layer widths / Reslap: general fused kernels
    n = n256.copy()
    n[3] = 200
    assert _dispatch(amd, n, 75, 64, 1)["path"] == 16 and not _dispatch(amd, n, 75, 64, 1)["bf16"]
    assert _dispatch(amd, tox(2048), 75, 64, 1)["bf16"] and not _dispatch(amd, n256, 75, 64, 1, variant=1)["bf16"]
    # ... unless it is within the members' reach (<= 128 nodes; <= 192 on the kernel variant with the wide members, as far as
    # their LDS plan fits; paper mode): specialised kernels, that graph on 8 (10) members
    n[3] = 150
    d = _dispatch(amd, n, 75, 64, 1)
    assert d["path"] & 7 == 7 and d["f_members"] >= 10 and d["b_members"] == d["f_members"]
    d = _dispatch(amd, n, 75, 64, 0)                           # (reference mode: the backward spreads exactly that molecule)
    assert d["path"] & 7 == 7 and d["f_members"] >= 10 and d["b_graphs"] == 1 and d["b_members"] == 10
    d = _dispatch(amd, n, 64, 64, 1, variant=1)
    assert d["path"] == 7 and d["f_members"] >= 10 and d["b_members"] == d["f_members"]
    n[3] = 180      # (75 -> 64: the members' images of 180 rows x 80 features do not fit LDS; 64 -> 64 does)
    assert _dispatch(amd, n, 75, 64, 1)["path"] == 16
    assert _dispatch(amd, n, 64, 64, 1, variant=1)["path"] == 7
    # ... also in a batch too large for members in general (two rounds of workgroups per CU): only that molecule is spread
    nb = np.concatenate([n, n256]).astype(np.int32)
    nb[3] = 150
    d = _dispatch(amd, nb, 75, 64, 1)
    assert d["path"] & 7 == 7 and d["f_graphs"] == 1 and d["f_members"] == 10 and d["b_graphs"] == 1 and d["b_members"] == 10
    d = _dispatch(amd, nb, 75, 64, 0)
    assert d["path"] & 7 == 7 and d["f_members"] == 10 and d["b_members"] == 10
    n[3] = 200      # SGC_LL_Reslap beyond every member: mixed dispatch (that molecule on the per-graph kernels, side stream)
    assert _dispatch(amd, n, 64, 64, 1, variant=1)["path"] == 8
    n[3] = 120
    d = _dispatch(amd, n, 75, 64, 1)
    assert d["path"] == 7 | 128 and d["f_graphs"] >= 1 and d["f_members"] >= 8 and d["b_members"] == d["f_members"]
    d = _dispatch(amd, n, 75, 64, 0)                           # (reference mode: members in the backward for that molecule only)
    assert d["path"] & 7 == 7 and d["b_graphs"] == 1 and d["b_members"] == 8
    amd._ffi.lib().agcn_set_spec_tail(0)
    assert _dispatch(amd, n256, 75, 64, 1)["path"] == 7
    amd._ffi.lib().agcn_set_spec_tail(prev_tail)
    d = _dispatch(amd, n, 128, 128, 1)                         # (128 input features: the members' plan holds a 128-node graph ...)
    assert d["path"] & 7 == 7 and d["b_members"] == d["f_members"] >= 8
    n[3] = 130
    assert _dispatch(amd, n, 128, 128, 1)["path"] == 16        # (... and no larger one)
    n[3] = 120
    assert _dispatch(amd, n256, 80, 64, 1)["path"] == 4
    # SGC_LL_Reslap at the specialised widths: the specialised kernels' Reslap flavour with the same layout as SGC_LL (the largest
    # graphs on members, small graphs paired to make room; reference mode: members in the forward only; a 120-node molecule
    # within the members' reach); other widths: the general fused kernels
    d, d0 = _dispatch(amd, n256, 64, 64, 1, variant=1), _dispatch(amd, n256, 64, 64, 1)
    assert d["path"] == 7 and d["f_members"] == d["b_members"] == d0["f_members"] > 40 and d["f_paired"] == d["b_paired"] == d0["f_paired"] > 0
    d = _dispatch(amd, n256, 64, 64, 0, variant=1)
    assert d["path"] == 7 and d["f_members"] > 40 and d["b_members"] == 0
    d = _dispatch(amd, n, 64, 64, 1, variant=1)
    assert d["path"] == 7 and d["f_members"] >= 8 and d["b_members"] == d["f_members"]
    d = _dispatch(amd, n, 64, 64, 0, variant=1)                    # (reference mode, 120 nodes: that molecule on 8 members in both directions)
    assert d["path"] == 7 and d["f_members"] >= 8 and d["b_graphs"] == 1 and d["b_members"] == 8
    assert _dispatch(amd, n256, 80, 64, 1, variant=1)["path"] == 4
    assert _dispatch(amd, n256, 75, 64, 1, K=3)["path"] == 32
    del rs


def test_member_budget_knob_and_cloud_dispatch(amd):
    """agcn_set_coop_members: the member workgroups of a launch never exceed the budget (all members of the launches in
    flight must be co-resident: epoch lanes divide it), clamped to [0, 120]; and point-cloud batches (Fin = 64, dense
    symmetric L_orig, every graph a multiple of 64 nodes) report the cloud kernels (path bit 64)."""
    from agcn_amd import synth
    lib = amd._ffi.lib()
    n256 = synth.node_counts(np.random.RandomState(123), 256)
    old = lib.agcn_get_coop_members()
    try:
        assert old == 120
        full = _dispatch(amd, n256, 75, 64, 1)["f_members"]
        assert 40 < full <= 120
        lib.agcn_set_coop_members(40)
        d = _dispatch(amd, n256, 75, 64, 1)
        assert d["path"] & ~128 == 7 and 0 < d["f_members"] <= 40 and d["b_members"] <= 40
        lib.agcn_set_coop_members(0)
        d = _dispatch(amd, n256, 75, 64, 1)
        assert d["path"] & ~128 == 7 and d["f_members"] == 0 and d["b_members"] == 0
        lib.agcn_set_coop_members(100000)
        assert lib.agcn_get_coop_members() == 120
        lib.agcn_set_coop_members(-5)
        assert lib.agcn_get_coop_members() == 0
    finally:
        lib.agcn_set_coop_members(old)
    clouds = np.full(4, 1024, np.int32)
    assert _dispatch(amd, clouds, 64, 128, 1)["path"] == 16          # CSR L_orig: the general slab kernels
    import ctypes as C

    def path(n_arr, Fin, Fout, variant=0, l_prev=None, prev_sym=0):
        n_arr = np.ascontiguousarray(n_arr, np.int32)
        nb = lib.agcn_plan_bytes_for(n_arr.ctypes.data, n_arr.size, 0)
        pl = np.zeros(nb, np.uint8)
        assert lib.agcn_plan_build(n_arr.ctypes.data, n_arr.size, 0, pl.ctypes.data, nb) == 0
        desc = amd._ffi.LayerDesc(Fin, Fout, 2, 1, variant, 1, Fin, Fout)
        gb = amd._ffi.GraphBatch(pl.ctypes.data, 1, 1, 1, 0, 1, None, None, None, l_prev, prev_sym, 0)   # dense L_orig, symmetric
        out = (C.c_int32 * 8)()
        assert lib.agcn_sgc_ll_dispatch_info(C.byref(desc), C.byref(gb), out, 8) == 0
        return out[0]
    # SGC_LL_Reslap (the layer of the reference's point-cloud network) and the 128-feature layers take the point-cloud kernels
    # too; a previous Laplacian must be vouched symmetric (AgcnGraphBatch.L_prev_symmetric), else the general per-graph path
    assert path(clouds, 64, 128, variant=1) & 64 and path(clouds, 128, 128, variant=1) & 64 and path(clouds, 128, 64) & 64
    assert path(clouds, 64, 128, variant=1, l_prev=1, prev_sym=1) & 64
    assert path(clouds, 64, 128, variant=1, l_prev=1, prev_sym=0) == 32
    assert path([300, 257], 128, 128, variant=1) == 32            # not multiples of 64
    assert path(clouds, 64, 128) == 16 | 64
    assert path([512, 256, 320, 64], 64, 128) == 16 | 64
    assert path([300, 257, 519, 131], 64, 128) == 16          # not multiples of 64: general slab kernels
    assert path(clouds, 3, 64) == 16                             # the first layer has kernels of its own (Fin <= 4)


def test_hiv_epoch_plan_is_a_balanced_partition():
    """bench.py --workload hiv-epoch: global batches in data-set order (the reference's batching rule on the global batch
    size), each split over the ranks by LPT: a partition of the data set, equal counts per rank and batch (up to one), per-rank
    work within a few per cent of each other."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import agcn_amd
    rs = np.random.RandomState(3)
    n = np.clip(np.rint(rs.lognormal(np.log(23.0), 0.5, size=5000)), 4, 222).astype(np.int32)
    for world in (1, 2, 8):
        per_rank = bench.hiv_epoch_plan(n, world, 256, 75, 64, 2)
        assert len(per_rank) == world and len({len(p) for p in per_rank}) == 1
        allidx = np.sort(np.concatenate([np.concatenate(p) for p in per_rank]))
        assert np.array_equal(allidx, np.arange(len(n)))
        cost = agcn_amd.dist.graph_cost(n, 75, 64, 2)
        for j in range(len(per_rank[0])):
            sizes = [len(per_rank[r][j]) for r in range(world)]
            assert max(sizes) <= 256 and max(sizes) == -(-sum(sizes) // world)      # (never more than the even share, rounded up)
            loads = np.array([cost[per_rank[r][j]].sum() for r in range(world)], dtype=np.float64)
            assert loads.max() / loads.mean() < 1.05


def test_dispatch_flags_are_per_call_and_per_thread(amd):
    """AgcnLayerDesc.flags (ABI 106): the kernel-selecting knobs as a per-call argument.  A field that is set overrides the
    process default for THAT call only -- the process-wide value (agcn_get_*) never moves -- and two threads calling at the
    same time with different flags each get their own dispatch (the override is thread-local inside the library).  Host
    logic only (agcn_sgc_ll_dispatch_info)."""
    import threading
    from agcn_amd import synth
    lib, ffi = amd._ffi.lib(), amd._ffi
    n256 = synth.node_counts(np.random.RandomState(123), 256)
    base = _dispatch(amd, n256, 75, 64, 1)
    assert base["path"] & ~128 == 7 and base["f_members"] > 40
    assert lib.agcn_get_coop_members() == 120 and lib.agcn_get_spec_mode() == 7 and lib.agcn_get_fused_mode() == 1
    # member budget, members off, specialised kernels off, fused kernels off: each for one call
    d = _dispatch(amd, n256, 75, 64, 1, flags=ffi.dispatch_flags(coop_members=40))
    assert d["path"] & ~128 == 7 and 0 < d["f_members"] <= 40 and d["b_members"] <= 40
    d = _dispatch(amd, n256, 75, 64, 1, flags=ffi.dispatch_flags(coop_members=0))
    assert d["f_members"] == 0 and d["b_members"] == 0
    d = _dispatch(amd, n256, 75, 64, 1, flags=ffi.dispatch_flags(spec_mode=3))
    assert d["path"] & ~128 == 7 and d["f_members"] == 0 and d["b_members"] == 0 and d["f_paired"] == 0
    assert _dispatch(amd, n256, 75, 64, 1, flags=ffi.dispatch_flags(spec_mode=0))["path"] == 4
    assert _dispatch(amd, n256, 75, 64, 1, flags=ffi.dispatch_flags(fused_mode=0))["path"] in (16, 32)
    assert _dispatch(amd, n256, 75, 64, 1, flags=ffi.dispatch_flags(fused_mode=0, slab_mode=0))["path"] == 32
    # nothing process-wide moved, and a call without flags is the default dispatch again
    assert lib.agcn_get_coop_members() == 120 and lib.agcn_get_spec_mode() == 7 and lib.agcn_get_fused_mode() == 1
    assert lib.agcn_get_slab_mode() == 1
    assert _dispatch(amd, n256, 75, 64, 1) == base
    # the encoding: (value + 1) per field, 0 = not overridden; out-of-range values are refused on the Python side
    assert ffi.dispatch_flags() == 0 and ffi.dispatch_flags(spec_mode=0) == 1 and ffi.dispatch_flags(fused_gb=-1) == 1 << 15
    assert ffi.dispatch_flags(ffi.dispatch_flags(spec_mode=7), spec_mode=3) == 4
    with pytest.raises(ValueError):
        ffi.dispatch_flags(coop_members=121)
    # two threads, different flags, many calls each, at the same time
    want = {0: base["f_members"], 1: 0}
    flags = {0: 0, 1: ffi.dispatch_flags(spec_mode=3)}
    bad = []

    def worker(k):
        for _ in range(200):
            got = _dispatch(amd, n256, 75, 64, 1, flags=flags[k])["f_members"]
            if got != want[k]:
                bad.append((k, got))
    th = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad[:5]
    # the Python context manager carries the flags into the descriptors this thread builds, and only inside the block
    from agcn_amd import functional as fn
    assert fn.current_flags() == 0
    with amd.dispatch(spec_mode=3, coop_members=10):
        assert fn.current_flags() == ffi.dispatch_flags(spec_mode=3, coop_members=10)
        with amd.dispatch(spec_mode=7):
            assert fn.current_flags() == ffi.dispatch_flags(spec_mode=7, coop_members=10)
        assert fn.no_member_flags() & 15 == 4            # spec_mode 3 (+ 1)
    assert fn.current_flags() == 0
    assert fn.no_member_flags() == ffi.dispatch_flags(spec_mode=3, pg_onepass=1)     # process defaults 7 / 2 -> 3 / 1


def test_reserve_size_classes(amd):
    """functional._size_class: 256 KB classes below 64 MB, at most +12.5 % above (a reserve that grows by a few KB finds the
    pooled buffer of the batch before it); monotone and never below the request."""
    from agcn_amd import functional as fn
    prev = 0
    for r in [1, 65536, 262144, 262145, (64 << 20) - 1, 64 << 20, (64 << 20) + 1, 256 << 20, (256 << 20) + 4096, 1 << 30,
              (1 << 30) + 1, 3 << 30]:
        c = fn._size_class(r)
        assert c >= r and c >= prev and c <= max(r + 262143, r * 1.125 + 1), (r, c)
        prev = c
    assert fn._size_class((256 << 20) + 4096) == fn._size_class((256 << 20) + 8192)
