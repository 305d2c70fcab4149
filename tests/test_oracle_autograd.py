"""laplacian_mode="paper" has no counterpart in the reference (its py_func blocks the
gradient), so the oracle's analytic forward/backward for that mode is pinned here against
an independent torch-CPU autograd statement in float64, plus the reference-mode branch
again for cross-checking.  CPU only."""
import numpy as np
import pytest
import torch

import agcn_oracle as orc


def _tf_min(x, y):
    # TF MinimumGrad: ties go to x.  Off ties torch.minimum agrees.
    return torch.minimum(x, y)


def torch_forward(x, L0, Lp, p, K, mode, variant):
    """Straight-line torch statement of SURVEY Appendix A (autograd supplies the backward)."""
    n = x.shape[0]
    one = torch.tensor(1.0, dtype=x.dtype)
    Z = x @ p["M_L"]
    diff = Z[:, None, :] - Z[None, :, :]
    sq = (diff * diff).sum(-1)
    eye = torch.eye(n, dtype=x.dtype)
    # sqrt with a defined-0 gradient at 0 (r_ij == 0 for duplicate rows / diagonal)
    pos = sq > 0
    r = torch.where(pos, torch.sqrt(torch.where(pos, sq, torch.ones_like(sq))), torch.zeros_like(sq))
    S = torch.exp(-r) * (1 - eye)
    if mode == orc.REFERENCE:
        S = S.detach()
    d = S.sum(0) + np.spacing(np.array(0, np.float64 if x.dtype == torch.float64 else np.float32)).item()
    s = 1 / torch.sqrt(d)
    if mode == orc.REFERENCE:
        R = (eye - torch.diag(s) * S * torch.diag(s)).detach()
    else:
        R = eye - s[:, None] * S * s[None, :]
    q = float(n * n) if variant == orc.SGC_LL else 1.0
    R1 = R * _tf_min(torch.rsqrt((R * R).sum()) * q, one)
    alpha = p["alpha"][0]
    R2 = torch.relu(R1) - alpha * torch.relu(-R1)
    if variant == orc.SGC_LL:
        La = R2 + L0
    else:
        Lpre = R2 + L0
        if Lp is not None:
            Lpre = Lpre + Lp * p["beta"][0]
        Lc = Lpre * _tf_min(torch.rsqrt((Lpre * Lpre).sum()), one)
        La = torch.relu(Lc) - alpha * torch.relu(-Lc)
    Xk = [x]
    if K > 1:
        Xk.append(La @ x)
    for k in range(2, K):
        Xk.append(2 * (La @ Xk[k - 1]) - Xk[k - 2])
    C = torch.stack(Xk, dim=2).reshape(n, -1)
    Y = torch.relu(C @ p["weight"] + p["bias"])
    return Y, R2, S, La


def _case(seed, ns, Fin, Fout, K, variant, dup=True):
    rs = np.random.RandomState(seed)
    X, L0 = [], []
    for n in ns:
        x = np.maximum(rs.randn(n, Fin), 0)
        if dup and n >= 5:
            x[3] = x[1]
            x[4] = x[1]
        X.append(x)
        _, _, _, L = orc.graph_laplacian(orc.synth_adj_lists(rs, n))
        L0.append(L)
    p = orc.init_params(rs, Fin, Fout, K, variant)
    p = {k: v.astype(np.float64) for k, v in p.items()}
    p["bias"] = 0.1 * rs.randn(Fout)
    p["alpha"] = np.array([0.6])
    if variant == orc.SGC_LL_RESLAP:
        p["beta"] = np.array([1.4])
    return rs, X, L0, p


@pytest.mark.parametrize("mode", [orc.REFERENCE, orc.PAPER])
@pytest.mark.parametrize("variant", [orc.SGC_LL, orc.SGC_LL_RESLAP])
@pytest.mark.parametrize("K", [1, 2, 3])
def test_oracle_vs_autograd_f64(mode, variant, K):
    ns, Fin, Fout = [4, 7, 12], 6, 5
    rs, X, L0, p = _case(100 + K, ns, Fin, Fout, K, variant)
    Lp = None
    if variant == orc.SGC_LL_RESLAP:
        Lp = [0.2 * rs.randn(n, n) for n in ns]
    caches = orc.layer_forward(X, L0, p, K, mode, variant, Lp_list=Lp)
    G = [rs.randn(n, Fout) for n in ns]
    dLa_extra = [0.3 * rs.randn(n, n) for n in ns] if variant == orc.SGC_LL_RESLAP else None
    tot = orc.layer_backward(caches, G, p, K, mode, variant, dLa_extra)

    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    tX = [torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in X]
    tLp = None if Lp is None else [torch.tensor(l, requires_grad=True) for l in Lp]
    loss = 0
    for b in range(len(ns)):
        Y, R2, S, La = torch_forward(tX[b], torch.tensor(L0[b]), None if tLp is None else tLp[b],
                                     tp, K, mode, variant)
        np.testing.assert_allclose(caches[b]["Y"], Y.detach().numpy(), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(caches[b]["R2"], R2.detach().numpy(), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(caches[b]["S"], S.detach().numpy(), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(caches[b]["La"], La.detach().numpy(), rtol=1e-12, atol=1e-13)
        loss = loss + (Y * torch.tensor(G[b])).sum()
        if dLa_extra is not None:
            loss = loss + (La * torch.tensor(dLa_extra[b])).sum()
    loss.backward()

    def g(t):
        return np.zeros(t.shape) if t.grad is None else t.grad.numpy()

    for b in range(len(ns)):
        np.testing.assert_allclose(tot["dX"][b], g(tX[b]), rtol=1e-9, atol=1e-11)
        if tLp is not None:
            np.testing.assert_allclose(tot["dLp"][b], g(tLp[b]), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(tot["dweight"], g(tp["weight"]), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(tot["dbias"], g(tp["bias"]), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(tot["dalpha"], g(tp["alpha"]), rtol=1e-9, atol=1e-11)
    if variant == orc.SGC_LL_RESLAP:
        np.testing.assert_allclose(tot["dbeta"], g(tp["beta"]), rtol=1e-9, atol=1e-11)
    if mode == orc.PAPER and (K > 1 or dLa_extra is not None):
        np.testing.assert_allclose(tot["dM_L"], g(tp["M_L"]), rtol=1e-8, atol=1e-11)
        assert np.abs(tot["dM_L"]).max() > 0
    else:
        assert tp["M_L"].grad is None or np.all(tp["M_L"].grad.numpy() == 0)


def test_paper_mode_finite_difference():
    """Central differences on M_L and X (no duplicate rows: r_ij = 0 is a kink)."""
    K, variant = 2, orc.SGC_LL
    rs, X, L0, p = _case(7, [6], 4, 3, K, variant, dup=False)
    X[0] = X[0] + 0.05 * rs.rand(*X[0].shape)      # keep away from relu kinks at x == 0
    G = [rs.randn(6, 3)]

    def loss(pp, xx):
        c = orc.layer_forward([xx], L0, pp, K, orc.PAPER, variant)
        return float((c[0]["Y"] * G[0]).sum())

    c = orc.layer_forward(X, L0, p, K, orc.PAPER, variant)
    tot = orc.layer_backward(c, G, p, K, orc.PAPER, variant)
    eps = 1e-6
    for (i, j) in [(0, 0), (1, 2), (3, 1)]:
        pp = {k: v.copy() for k, v in p.items()}
        pp["M_L"][i, j] += eps
        up = loss(pp, X[0])
        pp["M_L"][i, j] -= 2 * eps
        dn = loss(pp, X[0])
        np.testing.assert_allclose(tot["dM_L"][i, j], (up - dn) / (2 * eps), rtol=2e-5, atol=1e-8)
    for (i, j) in [(0, 1), (2, 3), (5, 0)]:
        xx = X[0].copy()
        xx[i, j] += eps
        up = loss(p, xx)
        xx[i, j] -= 2 * eps
        dn = loss(p, xx)
        np.testing.assert_allclose(tot["dX"][0][i, j], (up - dn) / (2 * eps), rtol=2e-5, atol=1e-8)


def test_known_answers():
    """SURVEY 8(c) (i)-(iv),(vi),(vii)."""
    rs = np.random.RandomState(3)
    n, Fin, Fout, K = 6, 5, 4, 2
    x = rs.randn(n, Fin).astype(np.float32)
    L0 = np.zeros((n, n), dtype=np.float32)
    p = orc.init_params(rs, Fin, Fout, K)
    # (i) reference mode => res_L == I for any X, M_L
    c = orc.forward_graph(x, L0, None, p, K, orc.REFERENCE, orc.SGC_LL)
    np.testing.assert_array_equal(c["R2"], np.eye(n, dtype=np.float32))
    # (iv) L_orig = 0, alpha = 1 => Y = relu([X|X] interleaved . W + b)
    Cexp = np.repeat(x, K, axis=1)
    np.testing.assert_allclose(c["Y"], np.maximum(Cexp @ p["weight"] + p["bias"], 0), rtol=1e-6, atol=1e-6)
    # (vii) W symmetric bit-for-bit, diagonal exactly 0
    np.testing.assert_array_equal(c["S"], c["S"].T)
    assert np.all(np.diag(c["S"]) == 0)
    # (ii) M_L = 0 => W_ij = 1 (i != j), d_j = n - 1
    p0 = dict(p)
    p0["M_L"] = np.zeros_like(p["M_L"])
    c0 = orc.forward_graph(x, L0, None, p0, K, orc.PAPER, orc.SGC_LL)
    np.testing.assert_array_equal(c0["S"], 1 - np.eye(n, dtype=np.float32))
    np.testing.assert_allclose(c0["s"], (n - 1) ** -0.5, rtol=1e-6)
    # (iii) n = 2, Fin = 1, M = [[1]], x = (0, a) => W_01 = exp(-|a|)
    a = np.float32(-1.75)
    S, _, _ = orc.similarity(np.array([[0.0], [a]], dtype=np.float32), np.array([[1.0]], dtype=np.float32))
    np.testing.assert_allclose(S[0, 1], np.exp(-abs(a)), rtol=1e-6)


def test_point_cloud_network_oracle_vs_autograd_f64():
    """The chained oracle of the reference's point-cloud network (tests/helpers.py: point_cloud_network_oracle; Point_AGCN.py:34-90:
    four SGC_LL_Reslap in two densely connected blocks, the second and third both reading the FIRST one's saved Laplacian,
    DenseMol, GraphGatherMol) against the same network written with the independent torch statement of a layer: every
    gradient, including the two cross-layer Laplacian edges that meet on layer 1's L_sum."""
    import helpers
    rs = np.random.RandomState(0)
    sizes = [40, 17]
    Xs = [rs.uniform(-1, 1, (n, 3)).astype(np.float32) for n in sizes]
    Ls = []
    for x in Xs:
        n = x.shape[0]
        d = np.sqrt(((x[:, None] - x[None]) ** 2).sum(-1))
        A = (d < 0.6 * d.mean()).astype(float)
        np.fill_diagonal(A, 0)
        Ls.append(orc.graph_laplacian([list(np.nonzero(A[i])[0]) for i in range(n)])[3].astype(np.float32))
    P = helpers.point_cloud_network_params(rs, widths=(16, 24, 24, 16), final=12, dense_scale=0.1)
    G = rs.randn(len(sizes), 12)
    r = helpers.point_cloud_network_oracle(Xs, Ls, P, G, orc.PAPER)
    tp = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True)  # noqa: E731
    pc = [{k: tp(v) for k, v in p.items()} for p in P["conv"]]
    be = [{"Win": [tp(w) for w in b["Win"]], "Wout": [tp(w) for w in b["Wout"]], "b1": tp(b["b1"]), "b2": tp(b["b2"])} for b in P["be"]]
    Wd, bd = tp(P["Wd"]), tp(P["bd"])
    Xt = [tp(x) for x in Xs]
    outs, V, M = [], orc.SGC_LL_RESLAP, orc.PAPER
    for g in range(len(sizes)):
        L0 = torch.tensor(Ls[g].astype(np.float64))
        Y1, _, _, La1 = torch_forward(Xt[g], L0, None, pc[0], 2, M, V)
        Y2, _, _, _ = torch_forward(Y1, L0, La1, pc[1], 2, M, V)
        B0 = torch.relu(Y2 + (Y1 @ be[0]["Win"][0]) * be[0]["b1"][0] + (Y2 @ be[0]["Win"][1]) * be[0]["b1"][0])
        Y3, _, _, La3 = torch_forward(B0, L0, La1, pc[2], 2, M, V)
        Y4, _, _, _ = torch_forward(Y3, L0, La3, pc[3], 2, M, V)
        B1 = torch.relu(Y4 + (Y3 @ be[1]["Win"][0]) * be[1]["b1"][0] + (Y4 @ be[1]["Win"][1]) * be[1]["b1"][0] +
                        (B0 @ be[1]["Wout"][0]) * be[1]["b2"][0])
        outs.append(torch.tanh((B1 @ Wd + bd).sum(0)))
    out = torch.stack(outs)
    (out * torch.tensor(G)).sum().backward()

    def close(a, b, what):
        a, b = np.asarray(a, dtype=np.float64).reshape(-1), b.detach().numpy().reshape(-1)
        assert np.abs(a - b).max() <= 1e-11 * (np.abs(b).max() + 1e-300), what

    close(r["out"], out, "out")
    close(r["dX"], torch.cat([x.grad for x in Xt]), "dX")
    for i in range(4):
        for k in ("weight", "bias", "M_L", "alpha", "beta"):
            g, o = pc[i][k].grad, r["conv_grads"][i]["d" + k]
            assert (g is None) == (o is None), (i, k)
            if g is not None:
                close(o, g, "layer %d d%s" % (i + 1, k))
    close(r["dWd"], Wd.grad, "dWd")
    close(r["dbd"], bd.grad, "dbd")
    for i in range(2):
        for l, w in enumerate(be[i]["Win"]):
            close(r["be%d_dWin" % i][l], w.grad, "block end %d dWin" % i)
        close(r["be%d_db1" % i], be[i]["b1"].grad, "block end %d dbeta1" % i)
    close(r["be1_dWout"][0], be[1]["Wout"][0].grad, "block end 1 dWout")
    close(r["be1_db2"], be[1]["b2"].grad, "block end 1 dbeta2")
