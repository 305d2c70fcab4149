"""The graph containers (agcn_amd/graphs.py; reference models/tf_modules/tf_graphs.py): layer lists written the way
the reference's networks write them (basic_AGCN.py:17-49, ResAGCN.py:13-57, DenseAGCN.py:13-85) run through `add()`
and give the same results as calling the layers by hand with the reference's input dicts."""
import numpy as np
import pytest
import torch

import agcn_oracle as orc
from helpers import (FEATURE_RTOL, FEATURE_ULPS, assert_close, golden_params, load_golden, make_layer, synth_case, to_np)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def amd():
    import agcn_amd
    return agcn_amd


def _batch(amd, seed=71, ns=(9, 17, 5, 24, 12), Fin=75):
    rs, X, L0, csr, _, _ = synth_case(seed, list(ns), Fin, 8, 2, orc.SGC_LL, atoms=True)
    gb = amd.GraphBatch.from_arrays(np.asarray(ns, np.int32), np.concatenate(X, 0), csr, DEV)
    return rs, X, L0, gb


def _gen(seed):
    return torch.Generator().manual_seed(seed)      # (parameters are drawn on the host, then moved)


def test_dispatch_is_by_class_name_and_order_is_checked(amd):
    m = amd.SequentialGraphMol(75, batch_size=5, max_atom=24)
    m.add(amd.SGC_LL(8, 75, 5, device=DEV))
    m.add(amd.GraphGatherMol(5, activation="tanh"))
    with pytest.raises(AssertionError, match="after a GraphGather"):
        m.add(amd.SGC_LL(8, 8, 5, device=DEV))
    assert [type(l).__name__ for l in m.layers] == ["SGC_LL", "GraphGatherMol"]


def test_simple_agcn_layer_list(amd):
    """SimpleAGCN (basic_AGCN.py:33-47) through SequentialGraphMol == the same layers called by hand."""
    rs, X, L0, gb = _batch(amd)
    B, K, widths = gb.plan.B, 2, [16, 24, 24, 16]
    model = amd.SequentialGraphMol(75, B, 24)
    fin = 75
    for i, w in enumerate(widths):
        model.add(amd.SGC_LL(w, fin, B, K=K, activation="relu", laplacian_mode="paper", device=DEV, generator=_gen(i)))
        fin = w
    model.add(amd.DenseMol(32, fin, activation="relu", device=DEV, generator=_gen(10)))
    model.add(amd.GraphGatherMol(B, activation="tanh"))
    out = model(gb)
    assert tuple(out.shape) == (B, 32)
    assert len(model.get_resL_set()) == 4 * B and len(model.get_resW_set()) == 4 * B   # every graph of every SGC layer
    # by hand, reference-style dicts
    h = gb.X
    for lay in model.layers[:4]:
        h, _, _ = lay({"node_features": h, "graph_batch": gb})
    h = model.layers[4]({"node_features": h, "graph_batch": gb})
    ref = model.layers[5]({"node_features": h, "graph_batch": gb})
    assert torch.equal(out, ref)
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_res_agcn_layer_list(amd):
    """ResAGCN (ResAGCN.py:31-55): two blocks of two SGC_LL + BlockEnd, against the oracle composed the same way."""
    rs, X, L0, gb = _batch(amd, seed=72)
    B, K, w = gb.plan.B, 2, [16, 24, 24, 16]
    model = amd.ResidualGraphMol(75, B, 24)
    model.add(amd.SGC_LL(w[0], 75, B, K=K, activation="relu", laplacian_mode="paper", device=DEV, generator=_gen(1)))
    model.add(amd.SGC_LL(w[1], w[0], B, K=K, activation="relu", laplacian_mode="paper", device=DEV, generator=_gen(2)))
    model.add(amd.BlockEnd(1, w[0], w[1], "relu", 24, B, device=DEV, generator=_gen(3)))
    model.add(amd.SGC_LL(w[2], w[1], B, K=K, activation="relu", laplacian_mode="paper", device=DEV, generator=_gen(4)))
    model.add(amd.SGC_LL(w[3], w[2], B, K=K, activation="relu", laplacian_mode="paper", device=DEV, generator=_gen(5)))
    model.add(amd.BlockEnd(2, w[2], w[3], "relu", 24, B, device=DEV, generator=_gen(6)))
    model.add(amd.DenseMol(32, w[3], activation="relu", device=DEV, generator=_gen(7)))
    model.add(amd.GraphGatherMol(B, activation="tanh"))
    out = model(gb)
    assert len(model.block_outputs) == 3          # first SGC_LL output + two block ends (tf_graphs.py:126-143)

    f64 = lambda t: to_np(t).astype(np.float64)   # noqa: E731
    L64 = [l.astype(np.float64) for l in L0]

    def sgc(hs, lay):
        p = {k: f64(v) for k, v in lay.vars.items()}
        return [c["Y"] for c in orc.layer_forward(hs, L64, p, K, orc.PAPER, orc.SGC_LL)]

    hs = [x.astype(np.float64) for x in X]
    h1 = sgc(hs, model.layers[0])
    h2 = sgc(h1, model.layers[1])
    b1 = [orc.block_end_forward(a, r, f64(model.layers[2].vars["weight"]))[0] for a, r in zip(h2, h1)]
    h3 = sgc(b1, model.layers[3])
    h4 = sgc(h3, model.layers[4])
    b2 = [orc.block_end_forward(a, r, f64(model.layers[5].vars["weight"]))[0] for a, r in zip(h4, b1)]
    hd = [orc.dense_mol_forward(x, f64(model.layers[6].W), f64(model.layers[6].b)) for x in b2]
    o, _ = orc.graph_gather_forward(hd, "tanh")
    assert_close(to_np(out), o, FEATURE_RTOL, "ResAGCN output")   # 8 stages deep
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_dense_agcn_layer_list(amd):
    """DenseAGCN (DenseAGCN.py:30-83): two dense blocks, save_output=True on every SGC_LL."""
    rs, X, L0, gb = _batch(amd, seed=73)
    B, K, w = gb.plan.B, 2, [16, 24, 24, 16]
    model = amd.DenseConnectedGraph(2, 75, B, 24)
    mk = lambda o, i, s: amd.SGC_LL(o, i, B, K=K, activation="relu", save_output=True, laplacian_mode="paper",  # noqa: E731
                                    device=DEV, generator=_gen(s))
    model.add(mk(w[0], 75, 1))
    model.add(mk(w[1], w[0], 2))
    model.add(amd.DenseBlockEnd(0, [w[0], w[1]], w[1], "relu", max_atom=24, batch_size=B, device=DEV, generator=_gen(3)))
    model.add(mk(w[2], w[1], 4))
    model.add(mk(w[3], w[2], 5))
    model.add(amd.DenseBlockEnd(1, [w[2], w[3]], w[3], "relu", max_atom=24, batch_size=B, device=DEV, generator=_gen(6)))
    model.add(amd.DenseMol(32, w[3], activation="relu", device=DEV, generator=_gen(7)))
    model.add(amd.GraphGatherMol(B, activation="tanh"))
    out = model(gb)
    # bookkeeping of tf_graphs.py:266-294: flat source-major lists, dims per source
    assert model.inblock_activations_dim == {0: [w[0], w[1]], 1: [w[2], w[3]]}
    assert len(model.inblock_activations[0]) == 2 * B and len(model.block_outputs) == 2 * B
    assert model.block_outputs_dim == [w[1], w[3]] and model.current_block_id == 2

    f64 = lambda t: to_np(t).astype(np.float64)   # noqa: E731
    L64 = [l.astype(np.float64) for l in L0]

    def sgc(hs, lay):
        p = {k: f64(v) for k, v in lay.vars.items()}
        return [c["Y"] for c in orc.layer_forward(hs, L64, p, K, orc.PAPER, orc.SGC_LL)]

    def dbe(x, ins, outs, lay):
        v = lay.vars
        return [orc.dense_block_end_forward(x[b], [s[b] for s in ins], [f64(t) for t in v["weight_inblock"]],
                                            [s[b] for s in outs], [f64(t) for t in v["weight_outblock"]],
                                            float(v["beta_inblock"].detach()), float(v["beta_outblock"].detach()))[0] for b in range(B)]

    hs = [x.astype(np.float64) for x in X]
    h1 = sgc(hs, model.layers[0])
    h2 = sgc(h1, model.layers[1])
    e1 = dbe(h2, [h1, h2], [], model.layers[2])
    h3 = sgc(e1, model.layers[3])
    h4 = sgc(h3, model.layers[4])
    e2 = dbe(h4, [h3, h4], [e1], model.layers[5])
    hd = [orc.dense_mol_forward(x, f64(model.layers[6].W), f64(model.layers[6].b)) for x in e2]
    o, _ = orc.graph_gather_forward(hd, "tanh")
    assert_close(to_np(out), o, FEATURE_RTOL, "DenseAGCN output")
    out.sum().backward()
    # (the first block has no earlier block outputs: its beta_outblock is unused and, as in TF, gets no gradient)
    unused = model.layers[2].vars["beta_outblock"]
    assert unused.grad is None
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters() if p is not unused)


def test_reslap_container_reference_feed_golden(amd):
    """ResidualGraphMolResLap on the reference's own padded feed: the 2-layer SGC_LL_Reslap chain of
    tests/golden/reslap_chain.npz (generated by the reference's classes, first layer save_lap=True) -- the container
    must hand `stack_laplacians[-batch_size:]` to the second layer (tf_graphs.py:184,193-194)."""
    g = load_golden("reslap_chain.npz")
    ns = g["ns"]
    B, K = len(ns), int(g["K"])
    Fin, Fout, max_atom = int(g["Fin"]), int(g["Fout"]), int(g["max_atom"])
    feed = {"node_features": [torch.from_numpy(g["X"][b]).to(DEV) for b in range(B)],
            "original_laplacian": [torch.from_numpy(g["L"][b]).to(DEV) for b in range(B)],
            "data_slice": g["data_slice"], "lap_slice": g["lap_slice"]}
    model = amd.ResidualGraphMolResLap(Fin, B, max_atom)
    l1 = make_layer(amd, orc.SGC_LL_RESLAP, Fout, Fin, B, K, orc.REFERENCE, golden_params(g, "l1_"), DEV)
    l2 = make_layer(amd, orc.SGC_LL_RESLAP, Fout, Fout, B, K, orc.REFERENCE, golden_params(g, "l2_"), DEV)
    l1.save_lap = True
    model.add(l1)
    model.add(l2)
    out = model(feed)
    assert len(model.get_laplacian()) == B and len(model.get_resL_set()) == 2 * B
    for b, n in enumerate(ns):
        y = to_np(out[b])
        assert y.shape == (max_atom, Fout) and np.all(y[n:] == 0)
        assert_close(y, g["l2_Y"][b], FEATURE_RTOL, "chain Y[%d]" % b)
        assert_close(to_np(model.get_laplacian()[b]), g["l1_L_sum"][b, :n, :n], FEATURE_RTOL, "saved L[%d]" % b)


def test_point_cloud_reslap_chain_through_the_container(amd):
    """The start of the reference's point-cloud network (Point_AGCN.py:37-60: SGC_LL_Reslap(n_filters_1, xyz, save_lap) ->
    SGC_LL_Reslap(n_filters_2, n_filters_1)) on two clouds with a dense symmetric L_orig, through ResidualGraphMolResLap:
    the container hands the first layer's L_sum on as the packed, symmetric-flagged TileViews, so BOTH layers (the first
    through zero-padded feature columns) run on the point-cloud kernels -- checked with the library's profiler -- and the
    chain's output and the gradient reaching the first layer's parameters equal the float64 oracle's."""
    from helpers import GRAD_RTOL, GRAD_ULPS, oracle_run
    sizes, K = [256, 128], 2
    rs = np.random.RandomState(4242)
    Xs, Ls = [], []
    for n in sizes:
        pts = rs.uniform(-1, 1, size=(n, 3)).astype(np.float32)
        d = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
        A = (d < 0.5 * d.mean()).astype(np.float64)
        np.fill_diagonal(A, 0)
        _, _, _, L = orc.graph_laplacian([list(np.nonzero(A[i])[0]) for i in range(n)])
        Xs.append(pts); Ls.append(L.astype(np.float32))
    p1 = orc.init_params(rs, 3, 64, K, orc.SGC_LL_RESLAP)
    p2 = orc.init_params(rs, 64, 32, K, orc.SGC_LL_RESLAP)
    for p in (p1, p2):
        p["alpha"] = np.array([0.9], np.float32)
        p["beta"] = np.array([1.1], np.float32)
    plan = amd.Plan(sizes, 0, DEV)
    l0 = np.zeros(int(plan.tile_off[-1]), np.float32)
    for b, L in enumerate(Ls):
        l0[int(plan.tile_off[b]):int(plan.tile_off[b]) + L.size] = L.reshape(-1)
    gb = amd.GraphBatch(plan, torch.from_numpy(np.concatenate(Xs)).to(DEV), l0_dense=torch.from_numpy(l0).to(DEV), l0_ld=0,
                        l0_symmetric=True)
    model = amd.ResidualGraphMolResLap(3, len(sizes), max(sizes))
    l1 = make_layer(amd, orc.SGC_LL_RESLAP, 64, 3, len(sizes), K, orc.PAPER, p1, DEV)
    l2 = make_layer(amd, orc.SGC_LL_RESLAP, 32, 64, len(sizes), K, orc.PAPER, p2, DEV)
    l1.save_lap = True
    model.add(l1)
    model.add(l2)
    lib = amd._ffi.lib()
    lib.agcn_profile_enable(1)
    try:
        amd._ffi.profile_collect()
        out = model(gb)
        G = [rs.randn(n, 32).astype(np.float32) for n in sizes]
        (out.data * torch.from_numpy(np.concatenate(G)).to(DEV)).sum().backward()
        torch.cuda.synchronize()
        prof = amd._ffi.profile_collect()
    finally:
        lib.agcn_profile_enable(0)
    assert prof["cloud_rlap1_kernel"][1] == 2 and prof["cloud_rbwd_kernel<C>"][1] == 2, prof
    assert prof["sgc_forward_kernel"][1] == 0 and prof["sgc_backward_kernel"][1] == 0
    # oracle: layer 1, then layer 2 with layer 1's summed Laplacian; the gradient of layer 2's L_prev arrives on layer 1's L_sum
    c1, _ = oracle_run(Xs, Ls, p1, K, orc.PAPER, orc.SGC_LL_RESLAP, [np.zeros((n, 64), np.float32) for n in sizes])
    La1 = [c["La"].astype(np.float32) for c in c1]
    Y1 = [c["Y"].astype(np.float32) for c in c1]
    c2, t2 = oracle_run(Y1, Ls, p2, K, orc.PAPER, orc.SGC_LL_RESLAP, G, La1)
    _, t1 = oracle_run(Xs, Ls, p1, K, orc.PAPER, orc.SGC_LL_RESLAP, [g.astype(np.float32) for g in t2["dX"]], None,
                       [g.astype(np.float32) for g in t2["dLp"]])
    r0 = 0
    for b, n in enumerate(sizes):
        assert_close(to_np(out.data)[r0:r0 + n], c2[b]["Y"], FEATURE_RTOL, "chain Y[%d]" % b)
        r0 += n
    assert_close(to_np(l1.vars["weight"].grad), t1["dweight"], GRAD_RTOL, "layer 1 dweight")
    assert_close(to_np(l1.vars["M_L"].grad), t1["dM_L"], GRAD_RTOL, "layer 1 dM_L")


def build_point_cloud_network(amd, P, B, max_atom, mode):
    """The reference's point-cloud network, entry for entry (models/networks/Point_AGCN.py:34-90), through the container it uses."""
    f = [int(p["weight"].shape[1]) for p in P["conv"]]
    raw, K = int(P["conv"][0]["M_L"].shape[0]), P["K"]
    m = amd.DenseConnectedGraphResLap(2, raw, B, max_atom)
    convs = []
    for i, (fin, fout) in enumerate(((raw, f[0]), (f[0], f[1]), (f[1], f[2]), (f[2], f[3]))):
        lay = make_layer(amd, orc.SGC_LL_RESLAP, fout, fin, B, K, mode, P["conv"][i], DEV)
        lay.save_output = True
        lay.save_lap = i in (0, 2)
        convs.append(lay)
    be = []
    for i, (res, outw) in enumerate((([f[0], f[1]], f[1]), ([f[2], f[3]], f[3]))):
        lay = amd.DenseBlockEnd(i, res, outw, "relu", max_atom=max_atom, batch_size=B, device=DEV)
        b = P["be"][i]
        lay.load_vars(b["Win"], b["Wout"], b["b1"], b["b2"])
        be.append(lay)
    dense = amd.DenseMol(int(P["Wd"].shape[1]), f[3], activation="relu", device=DEV)
    dense.load_vars(P["Wd"], P["bd"])
    for lay in (convs[0], convs[1], be[0], convs[2], convs[3], be[1], dense, amd.GraphGatherMol(B, activation="tanh")):
        m.add(lay)
    return m, convs, be, dense


@pytest.mark.parametrize("mode", [orc.PAPER, orc.REFERENCE], ids=["paper", "reference"])
def test_point_cloud_network_whole(amd, mode):
    """The WHOLE point-cloud network of the reference (Point_AGCN.py:34-90: DenseConnectedGraphResLap with the eight-entry layer
    list -- SGC_LL_Reslap 3 -> 64 (save_lap) -> 128 | DenseBlockEnd | 128 -> 128 (save_lap) -> 64 | DenseBlockEnd | DenseMol 256 |
    GraphGatherMol; widths utils/hyper_parameters.py:8-24) on clouds of 1024 points (meshloader.py:34-39), forward + backward,
    against the chained float64 oracle (tests/helpers.py, itself pinned to torch autograd in test_oracle_autograd.py): the
    readout, the input gradient and every parameter gradient of every layer, bounds calibrated on the chain run in float32.
    Every convolution must run on the point-cloud kernels (library profiler: no per-graph kernel launches)."""
    from helpers import (GRAD_RTOL, _scalar_close, point_cloud_network_oracle, point_cloud_network_params)
    sizes = [1024, 1024, 256]
    rs = np.random.RandomState(1024)
    Xs, Ls = [], []
    for n in sizes:
        pts = rs.uniform(-1, 1, size=(n, 3)).astype(np.float32)
        d = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
        A = (d < 0.6 * d.mean()).astype(np.float64)                       # meshloader.py:264-285 threshold rule
        np.fill_diagonal(A, 0)
        _, _, _, L = orc.graph_laplacian([list(np.nonzero(A[i])[0]) for i in range(n)])
        Xs.append(pts); Ls.append(L.astype(np.float32))
    P = point_cloud_network_params(rs)
    B = len(sizes)
    plan = amd.Plan(sizes, 0, DEV)
    l0 = np.zeros(int(plan.tile_off[-1]), np.float32)
    for b, L in enumerate(Ls):
        l0[int(plan.tile_off[b]):int(plan.tile_off[b]) + L.size] = L.reshape(-1)
    gb = amd.GraphBatch(plan, torch.from_numpy(np.concatenate(Xs)).to(DEV), l0_dense=torch.from_numpy(l0).to(DEV), l0_ld=0,
                        l0_symmetric=True)
    model, convs, be, dense = build_point_cloud_network(amd, P, B, max(sizes), mode)
    Gout = rs.randn(B, int(P["Wd"].shape[1])).astype(np.float32)
    Xt = gb.X.clone().requires_grad_(True)
    gb.X = Xt
    conv_Y = []                                   # the kernels' Y of every convolution: settles activation ties (helpers)
    for lay in convs:
        def recording(x, _call=lay.call):
            res = _call(x)
            conv_Y.append(res[0].data.detach())
            return res
        lay.call = recording
    lib = amd._ffi.lib()
    lib.agcn_profile_enable(1)
    try:
        amd._ffi.profile_collect()
        out = model(gb)
        (out * torch.from_numpy(Gout).to(DEV)).sum().backward()
        torch.cuda.synchronize()
        prof = {k: v[1] for k, v in amd._ffi.profile_collect().items() if v[1]}
    finally:
        lib.agcn_profile_enable(0)
    assert tuple(out.shape) == (B, int(P["Wd"].shape[1]))
    assert len(model.get_laplacian()) == 2 * B and len(model.get_resL_set()) == 4 * B
    # four convolutions, all on the point-cloud kernels (one similarity / Laplacian launch each: the 128-feature layers take
    # both feature halves in one run of cloud_lap; their Reslap backward passes still run once per half)
    assert prof["cloud_sim_kernel"] == 4 and prof["cloud_rlap1_kernel"] == 4 and prof["cloud_lap_kernel"] == 4, prof
    assert not any(k.startswith(("sgc_forward", "sgc_backward", "fused_", "spec_", "slab_")) for k in prof), prof
    assert len(conv_Y) == 4
    r = point_cloud_network_oracle(Xs, Ls, P, Gout, mode, np.float64, Y_ties=[to_np(y) for y in conv_Y])
    r32 = point_cloud_network_oracle(Xs, Ls, P, Gout, mode, np.float32, Y_ties=r["Y"])
    assert_close(to_np(out), r["out"], FEATURE_RTOL, "network readout", ref32=r32["out"])
    assert_close(to_np(Xt.grad), r["dX"], GRAD_RTOL, "network dX", ref32=r32["dX"])
    for i, lay in enumerate(convs):
        t, t32 = r["conv_grads"][i], r32["conv_grads"][i]
        scale = float(np.abs(t["dweight"]).max())
        assert_close(to_np(lay.vars["weight"].grad), t["dweight"], GRAD_RTOL, "conv %d dweight" % (i + 1), ref32=t32["dweight"])
        assert_close(to_np(lay.vars["bias"].grad), t["dbias"], GRAD_RTOL, "conv %d dbias" % (i + 1), ref32=t32["dbias"])
        _scalar_close(to_np(lay.vars["alpha"].grad), t["dalpha"], t32["dalpha"], scale, "conv %d dalpha" % (i + 1))
        if i > 0:
            _scalar_close(to_np(lay.vars["beta"].grad), t["dbeta"], t32["dbeta"], scale, "conv %d dbeta" % (i + 1))
        if mode == orc.PAPER:
            assert_close(to_np(lay.vars["M_L"].grad), t["dM_L"], GRAD_RTOL, "conv %d dM_L" % (i + 1), ref32=t32["dM_L"])
        else:
            assert lay.vars["M_L"].grad is None
    assert_close(to_np(dense.W.grad), r["dWd"], GRAD_RTOL, "DenseMol dW", ref32=r32["dWd"])
    for i, lay in enumerate(be):
        for l, w in enumerate(lay.vars["weight_inblock"]):
            assert_close(to_np(w.grad), r["be%d_dWin" % i][l], GRAD_RTOL, "block end %d dWin[%d]" % (i, l), ref32=r32["be%d_dWin" % i][l])
        sc = float(max(np.abs(w).max() for w in r["be%d_dWin" % i]))
        _scalar_close(to_np(lay.vars["beta_inblock"].grad), r["be%d_db1" % i], r32["be%d_db1" % i], sc, "block end %d dbeta_in" % i)
    assert_close(to_np(be[1].vars["weight_outblock"][0].grad), r["be1_dWout"][0], GRAD_RTOL, "block end 1 dWout", ref32=r32["be1_dWout"][0])


def test_reslap_container_step_captures_into_one_hip_graph(amd):
    """A whole ResLap network step (container + autograd forward + backward) as ONE HIP graph (agcn_amd.GraphedStep): replays
    reproduce the eager step bit for bit.  Regression: SGC_LL_Reslap kept `early_laps` (the previous layer's L_sum, i.e. its
    whole autograd graph) as an attribute between calls, so the first layer's AccumulateGrad nodes survived from the warm-up
    into the capture on another stream -- a synchronisation inside the capture (bench.py's point-cloud network line crashed)."""
    from helpers import point_cloud_network_params
    sizes = [128, 64]
    rs = np.random.RandomState(5)
    Xs, Ls = [], []
    for n in sizes:
        pts = rs.uniform(-1, 1, size=(n, 3)).astype(np.float32)
        d = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
        A = (d < 0.6 * d.mean()).astype(np.float64)
        np.fill_diagonal(A, 0)
        _, _, _, L = orc.graph_laplacian([list(np.nonzero(A[i])[0]) for i in range(n)])
        Xs.append(pts); Ls.append(L.astype(np.float32))
    P = point_cloud_network_params(rs, widths=(64, 128, 128, 64), final=32, dense_scale=0.05)
    B = len(sizes)
    plan = amd.Plan(sizes, 0, DEV)
    l0 = np.zeros(int(plan.tile_off[-1]), np.float32)
    for b, L in enumerate(Ls):
        l0[int(plan.tile_off[b]):int(plan.tile_off[b]) + L.size] = L.reshape(-1)
    gb = amd.GraphBatch(plan, torch.from_numpy(np.concatenate(Xs)).to(DEV), l0_dense=torch.from_numpy(l0).to(DEV), l0_ld=0,
                        l0_symmetric=True)
    model, convs, be, dense = build_point_cloud_network(amd, P, B, max(sizes), orc.PAPER)
    G = torch.from_numpy(rs.randn(B, 32).astype(np.float32)).to(DEV)
    params = [v for lay in convs for v in lay.vars.values()] + dense.parameters()
    for lay in be:
        params += [lay.vars["beta_inblock"], lay.vars["beta_outblock"]] + lay.vars["weight_inblock"] + lay.vars["weight_outblock"]

    def step():
        for q in params:
            q.grad = None
        out = model(gb)
        out.backward(G)
        return out.detach()

    want_out = step().clone()
    want = [None if q.grad is None else q.grad.clone() for q in params]
    assert all(lay.early_laps is None for lay in convs)
    gs = amd.GraphedStep(step, DEV)
    for _ in range(2):
        got_out = gs.replay()
        torch.cuda.synchronize()
        assert torch.equal(got_out, want_out)
        for q, w in zip(params, want):
            assert (q.grad is None) == (w is None)
            if w is not None:
                assert torch.equal(q.grad, w)


def test_segmentation_graph_backbone_and_two_heads(amd):
    """SegmentationGraph (seg_graph.py:9-125) with Seg_AGCN's layer list shape (Seg_AGCN.py:75-121: SGC_LL, SGC_LL,
    GraphPoolMol backbone; DenseMol -> GraphGatherMol -> FCL classifier head; SGC_LL, SGC_LL -> Merge segmentation head):
    both heads start from the backbone output, FCL sees `node_features` only, Merge the raw output, and the results equal
    the same layers called by hand."""
    rs, X, L0, gb = _batch(amd, Fin=75)
    B = gb.plan.B

    class FCL(object):                       # dispatched by class name; the trainers' layer itself is out of scope
        def __init__(self):
            self.keys = None

        def __call__(self, x):
            self.keys = sorted(x.keys())
            return x["node_features"] * 2.0

    class Merge(object):
        def __call__(self, x):
            assert not isinstance(x, dict)
            return ("merged", x)

    m = amd.SegmentationGraph(75, batch_size=B, num_point=24)
    with pytest.raises(AssertionError):
        m.add(amd.DenseMol(8, 8, device=DEV), classifer=True)          # no backbone yet
    with pytest.raises(AssertionError):
        m.add(object())                                                 # not a segmentation-graph layer
    bb = [amd.SGC_LL(16, 75, B, K=2, laplacian_mode="paper", device=DEV, generator=_gen(1)),
          amd.SGC_LL(24, 16, B, K=2, laplacian_mode="paper", device=DEV, generator=_gen(2)),
          amd.GraphPoolMol(B)]
    for lay in bb:
        m.add(lay)
    cl = [amd.DenseMol(12, 24, activation="relu", device=DEV, generator=_gen(3)), amd.GraphGatherMol(B, activation="tanh"), FCL()]
    for lay in cl:
        m.add(lay, classifer=True)
    sg = [amd.SGC_LL(16, 24, B, K=2, laplacian_mode="paper", device=DEV, generator=_gen(4)),
          amd.SGC_LL(6, 16, B, K=2, laplacian_mode="paper", device=DEV, generator=_gen(5)), Merge()]
    for lay in sg:
        m.add(lay, segmentation=True)
    cls_out, seg_out = m(gb)
    assert len(m.get_resL_set()) == 2 * B            # the backbone's SGC layers (the head's are not collected: seg_graph.py:66)
    # by hand
    h = gb.X
    for lay in bb[:2]:
        h, _, _ = lay({"node_features": h, "graph_batch": gb})
    h = bb[2]({"node_features": h, "graph_batch": gb})
    assert torch.equal(to_np_t(m.backbone_outputs()), to_np_t(h))
    c = cl[0]({"node_features": h, "graph_batch": gb})
    c = cl[1]({"node_features": c, "graph_batch": gb})
    assert torch.equal(cls_out, c * 2.0) and cl[2].keys == ["node_features"]
    s1, _, _ = sg[0]({"node_features": h, "graph_batch": gb})
    s2, _, _ = sg[1]({"node_features": s1, "graph_batch": gb})
    assert seg_out[0] == "merged" and torch.equal(to_np_t(seg_out[1]), to_np_t(s2))
    assert m.classification_outputs() is cls_out and m.segmentation_outputs() is seg_out


def to_np_t(v):
    return v.data if hasattr(v, "data") and not torch.is_tensor(v) else v


def test_learning_phase_switches_dropout_for_the_whole_container(amd):
    """dropout=p layers depend on the learning phase (dropout.py:34-41, model_operatos.py:46-61): `model.eval()` makes the
    predictions deterministic, `model.train()` brings the dropout back; `uses_learning_phase` says so."""
    rs, X, L0, gb = _batch(amd)
    B = gb.plan.B
    m = amd.SequentialGraphMol(75, B, 24)
    m.add(amd.SGC_LL(16, 75, B, K=2, dropout=0.5, laplacian_mode="paper", device=DEV, generator=_gen(1)))
    m.add(amd.SGC_LL(8, 16, B, K=2, laplacian_mode="paper", device=DEV, generator=_gen(2)))
    assert m.uses_learning_phase and m.layers[0].uses_learning_phase and not m.layers[1].uses_learning_phase
    m.eval()
    a, b = m(gb).data.clone(), m(gb).data.clone()
    assert torch.equal(a, b)
    m.train()
    torch.manual_seed(0)
    c = m(gb).data.clone()
    torch.manual_seed(1)
    d = m(gb).data.clone()
    assert not torch.equal(c, d) and not torch.equal(a, c)
