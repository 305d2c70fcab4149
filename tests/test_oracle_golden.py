"""The oracle (oracle/agcn_oracle.py) against golden vectors produced by the
reference's own layer classes (oracle/refshim/gen_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

import agcn_oracle as orc
from conftest import GOLDEN

RTOL = 2e-6     # oracle-vs-reference float32 tolerance (summation order only)
ATOL = 2e-6


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def _adj(g):
    ptr, flat = g["adj_ptr"], g["adj_flat"]
    return [list(map(int, flat[ptr[i]:ptr[i + 1]])) for i in range(len(ptr) - 1)]


LAP = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "lap_*.npz")))


@pytest.mark.parametrize("name", LAP)
def test_graph_laplacian(name):
    g = _load(name)
    adj = _adj(g)
    # integer work: bit exact
    np.testing.assert_array_equal(orc.degree_list(adj), g["degree_list"])
    assert orc.degree_list(adj).dtype == np.int32
    assert orc.has_laplacian(len(adj)) == bool(g["has_lap"])
    if not g["has_lap"]:
        return
    indptr, indices, data, dense = orc.graph_laplacian(adj)
    assert indptr.dtype == np.int32 and indices.dtype == np.int32 and data.dtype == np.float64
    np.testing.assert_array_equal(indptr, g["indptr"])
    np.testing.assert_array_equal(indices, g["sorted_indices"])
    # float64 values: every product is restated in scipy's order; only the column sum
    # `W.sum(axis=0)` (graph_structure.py:93) follows scipy's internal accumulation order,
    # which leaves <= 1-2 ulp (f64) of slack.  The layer path consumes float32(L).
    np.testing.assert_allclose(data, g["sorted_data"], rtol=1e-15, atol=0)
    np.testing.assert_allclose(dense, g["dense"], rtol=1e-15, atol=0)
    assert np.abs(dense.astype(np.float32) - g["dense"].astype(np.float32)).max() <= 6e-8
    assert np.abs(dense - dense.T).mean() < 1e-9      # graph_structure.py:106


def test_ring_known_answer():
    # ring C_n -> diagonal 2/3, neighbours -1/3
    n = 6
    adj = [[(i - 1) % n, (i + 1) % n] for i in range(n)]
    _, _, _, L = orc.graph_laplacian(adj)
    np.testing.assert_allclose(np.diag(L), 2.0 / 3.0, rtol=1e-15)
    np.testing.assert_allclose(L[0, 1], -1.0 / 3.0, rtol=1e-15)
    assert L[0, 2] == 0


def test_pad_batch_topology():
    g = _load("topo_tox4.npz")
    ns = g["ns"]
    off = np.concatenate([[0], np.cumsum(ns)])
    off2 = np.concatenate([[0], np.cumsum(ns.astype(np.int64) ** 2)])
    feats = [g["feat"][off[b]:off[b + 1]] for b in range(len(ns))]
    laps = [g["lap_dense"][off2[b]:off2[b + 1]].reshape(ns[b], ns[b]) for b in range(len(ns))]
    X, L, ds, ls = orc.pad_batch(feats, laps, int(g["max_atom"]))
    np.testing.assert_array_equal(np.stack(X), g["X"])
    np.testing.assert_array_equal(np.stack(L), g["L"])
    np.testing.assert_array_equal(ds, g["data_slice"])
    np.testing.assert_array_equal(ls, g["lap_slice"])
    assert ds.dtype == np.int32 and ls.dtype == np.int32


def _params(g, prefix):
    p = {}
    for k in ("weight", "bias", "M_L", "alpha", "beta"):
        if prefix + "p_" + k in g:
            p[k] = g[prefix + "p_" + k]
    return p


def _unpad(g):
    ns = g["ns"]
    X = [g["X"][b, :n] for b, n in enumerate(ns)]
    L0 = [g["L"][b, :n, :n] for b, n in enumerate(ns)]
    return ns, X, L0


LAYER = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "sgc_*.npz")) +
               glob.glob(os.path.join(GOLDEN, "reslap_*.npz")))


def _close(a, b, what, rtol=RTOL, atol=ATOL):
    scale = max(1.0, float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale, err_msg=what)


@pytest.mark.parametrize("name", LAYER)
def test_layer_forward_backward(name):
    g = _load(name)
    ns, X, L0 = _unpad(g)
    K, variant, chain = int(g["K"]), int(g["variant"]), int(g["chain"])
    max_atom = int(g["max_atom"])
    p1 = _params(g, "l1_")
    c1 = orc.layer_forward(X, L0, p1, K, orc.REFERENCE, variant)

    def check_fwd(prefix, caches):
        for b, n in enumerate(ns):
            c = caches[b]
            _close(orc.pad_rows(c["Y"], max_atom), g[prefix + "Y"][b], "Y")
            # reference mode: res_L is exactly I (SGC_LL) or I/sqrt(n) (Reslap)
            _close(c["R2"], g[prefix + "res_L"][b, :n, :n], "res_L", atol=1e-7)
            _close(c["S"], g[prefix + "res_W"][b, :n, :n], "res_W")
            assert np.all(np.diag(c["S"]) == 0)
            np.testing.assert_array_equal(c["S"], c["S"].T)
            if variant == orc.SGC_LL_RESLAP:
                _close(c["La"], g[prefix + "L_sum"][b, :n, :n], "L_sum")
            assert np.all(g[prefix + "Y"][b, n:] == 0)     # pad rows exactly 0

    check_fwd("l1_", c1)
    if variant == orc.SGC_LL:
        for b, n in enumerate(ns):
            np.testing.assert_array_equal(c1[b]["R2"], np.eye(n, dtype=np.float32))

    G = [g["G"][b, :n] for b, n in enumerate(ns)]
    if not chain:
        t1 = orc.layer_backward(c1, G, p1, K, orc.REFERENCE, variant)
    else:
        p2 = _params(g, "l2_")
        X2 = [c["Y"] for c in c1]
        Lp = [c["La"] for c in c1]
        c2 = orc.layer_forward(X2, L0, p2, K, orc.REFERENCE, variant, Lp_list=Lp)
        check_fwd("l2_", c2)
        t2 = orc.layer_backward(c2, G, p2, K, orc.REFERENCE, variant)
        t1 = orc.layer_backward(c1, t2["dX"], p1, K, orc.REFERENCE, variant, dLa_extra_list=t2["dLp"])
        for k in ("weight", "bias", "alpha", "beta"):
            assert int(g["l2_has_d_" + k]) == 1
            _close(np.asarray(t2["d" + k]).reshape(g["l2_d_" + k].shape), g["l2_d_" + k], "l2 d" + k,
                   rtol=2e-5, atol=2e-5)
        assert int(g["l2_has_d_M_L"]) == 0 and t2["dM_L"] is None

    for b, n in enumerate(ns):
        _close(t1["dX"][b], g["dX"][b, :n], "dX", rtol=2e-5, atol=2e-5)
        assert np.all(g["dX"][b, n:] == 0)
    for k in ("weight", "bias", "alpha"):
        if k == "alpha" and K == 1:
            # K == 1: L_all never reaches Y, so TF reports no gradient at all; the oracle says 0
            assert int(g["l1_has_d_alpha"]) == 0 and float(t1["dalpha"][0]) == 0.0
            continue
        assert int(g["l1_has_d_" + k]) == 1
        _close(np.asarray(t1["d" + k]).reshape(g["l1_d_" + k].shape), g["l1_d_" + k], "d" + k,
               rtol=2e-5, atol=2e-5)
    # M_L never receives a gradient in the reference
    assert int(g["l1_has_d_M_L"]) == 0 and t1["dM_L"] is None
    if variant == orc.SGC_LL:
        assert float(g["l1_d_alpha"][0]) == 0.0
    else:
        # first Reslap layer: no previous Laplacian => beta unused
        assert int(g["l1_has_d_beta"]) == 0 and t1["dbeta"] is None


def test_similarity_loop_matches_vectorised():
    g = _load("sgc_tox_l1.npz")
    ns, X, _ = _unpad(g)
    M = g["l1_p_M_L"]
    for b in range(len(ns)):
        S, _, _ = orc.similarity(X[b], M)
        _close(orc.similarity_loop(X[b], M), S, "S loop")


@pytest.mark.parametrize("act", ["tanh", "relu"])
def test_graph_gather(act):
    g = _load("gather_%s.npz" % act)
    ns = g["ns"]
    X = [g["X"][b, :n] for b, n in enumerate(ns)]
    out, pre = orc.graph_gather_forward(X, act)
    _close(out, g["out"], "gather out")
    dX = orc.graph_gather_backward(out, pre, g["G"], ns, act)
    for b, n in enumerate(ns):
        _close(dX[b], g["dX"][b, :n], "gather dX")
        assert np.all(g["dX"][b, n:] == 0)


def test_dense_mol_golden():
    g = _load("dense_mol.npz")
    dW = 0
    db = 0
    for b in range(len(g["ns"])):
        _close(orc.dense_mol_forward(g["X"][b], g["W"], g["b"]), g["Y"][b], "DenseMol Y")
        dx, dw, dbb = orc.dense_mol_backward(g["X"][b], g["W"], g["G"][b])
        _close(dx, g["dX"][b], "DenseMol dX", rtol=2e-5, atol=2e-5)
        dW, db = dW + dw, db + dbb
    _close(dW, g["dW"], "DenseMol dW", rtol=2e-5, atol=2e-5)
    _close(db, g["db"], "DenseMol db", rtol=2e-5, atol=2e-5)
    n0 = int(g["ns"][0])
    assert np.allclose(g["Y"][0][n0:], g["b"])        # the bias lands on pad rows (dense_layer.py:42-50)


def test_block_end_golden():
    g = _load("block_end.npz")
    dW = 0
    for b, n in enumerate(g["ns"]):
        y, pre = orc.block_end_forward(g["X"][b, :n], g["R"][b, :n], g["W"])
        _close(y, g["Y"][b, :n], "BlockEnd Y")
        assert np.all(g["Y"][b, n:] == 0)
        dx, dr, dw = orc.block_end_backward(g["R"][b, :n], g["W"], pre, g["G"][b, :n])
        _close(dx, g["dX"][b, :n], "BlockEnd dX", rtol=2e-5, atol=2e-5)
        _close(dr, g["dR"][b, :n], "BlockEnd dR", rtol=2e-5, atol=2e-5)
        dW = dW + dw
    _close(dW, g["dW"], "BlockEnd dW", rtol=2e-5, atol=2e-5)


def test_dense_block_end_golden():
    """densenet_block.py:48-131 executed from the reference: forward, node / weight / beta gradients."""
    g = _load("dense_block_end.npz")
    n_in, n_out = int(g["n_in"]), int(g["n_out"])
    Win = [g["Win%d" % i] for i in range(n_in)]
    Wout = [g["Wout%d" % i] for i in range(n_out)]
    dWi, dWo, db1, db2 = [0] * n_in, [0] * n_out, 0.0, 0.0
    for b, n in enumerate(g["ns"]):
        ins = [g["A%d" % i][b, :n] for i in range(n_in)]
        outs = [g["O%d" % i][b, :n] for i in range(n_out)]
        y, pre = orc.dense_block_end_forward(g["X"][b, :n], ins, Win, outs, Wout, g["beta1"], g["beta2"])
        _close(y, g["Y"][b, :n], "DenseBlockEnd Y", rtol=2e-5, atol=2e-5)
        assert np.all(g["Y"][b, n:] == 0)
        dx, da, dwi, do, dwo, d1, d2 = orc.dense_block_end_backward(ins, Win, outs, Wout, g["beta1"], g["beta2"], pre,
                                                                    g["G"][b, :n])
        _close(dx, g["dX"][b, :n], "DenseBlockEnd dX", rtol=2e-5, atol=2e-5)
        for i in range(n_in):
            _close(da[i], g["dA%d" % i][b, :n], "DenseBlockEnd dA", rtol=2e-5, atol=2e-5)
            dWi[i] = dWi[i] + dwi[i]
        for i in range(n_out):
            _close(do[i], g["dO%d" % i][b, :n], "DenseBlockEnd dO", rtol=2e-5, atol=2e-5)
            dWo[i] = dWo[i] + dwo[i]
        db1, db2 = db1 + d1, db2 + d2
    for i in range(n_in):
        _close(dWi[i], g["dWin%d" % i], "DenseBlockEnd dWin", rtol=2e-5, atol=2e-5)
    for i in range(n_out):
        _close(dWo[i], g["dWout%d" % i], "DenseBlockEnd dWout", rtol=2e-5, atol=2e-5)
    assert abs(db1 - float(g["dbeta1"][0])) <= 2e-5 * abs(float(g["dbeta1"][0])) + 1e-5
    assert abs(db2 - float(g["dbeta2"][0])) <= 2e-5 * abs(float(g["dbeta2"][0])) + 1e-5


def test_multitask_head_golden():
    """multitask_logits + 'cross_entropy' training loss executed from the reference
    (model_operatos.py:792-864, multitask_classifier.py:41-44,187-209)."""
    g = _load("multitask_head.npz")
    B = g["feat"].shape[0]
    loss, logits = orc.multitask_ce_forward(g["feat"], g["W"], g["b"], g["labels"], g["weights"], B)
    _close(logits, g["logits"], "logits", rtol=1e-5, atol=1e-6)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    dfeat, dW, db = orc.multitask_ce_backward(g["feat"], g["W"], logits, g["labels"], g["weights"], B)
    _close(dfeat, g["dfeat"], "dfeat", rtol=2e-5, atol=1e-6)
    _close(dW, g["dW"], "dW", rtol=2e-5, atol=1e-6)
    _close(db, g["db"], "db", rtol=2e-5, atol=1e-6)


def test_graph_pool_golden():
    g = _load("graph_pool.npz")
    for b, n in enumerate(g["ns"]):
        y, arg = orc.graph_pool_forward(g["X"][b, :n], g["L"][b, :n, :n])
        np.testing.assert_array_equal(y, g["Y"][b, :n])          # a max of inputs: bit exact
        assert np.all(g["Y"][b, n:] == 0)
        G = np.random.RandomState(b).randn(n, y.shape[1]).astype(np.float32)
        dX = orc.graph_pool_backward(arg, G)
        assert abs(dX.sum() - G.sum()) < 1e-4                     # routing conserves the gradient mass
