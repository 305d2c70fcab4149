"""Shared helpers for the parity tests (HIP path vs oracle / golden vectors)."""
import os

import numpy as np
import torch

import agcn_oracle as orc
from conftest import GOLDEN

# Parity tolerances.  BASELINE.json north_star: "within 1e-5 rel fp32 on features".  The check is ELEMENT-WISE:
#     |got - want| <= rtol * |want| + atol_ulps * eps_f32 * scale,   scale[i, j] = max(rowmax_i |want|, colmax_j |want|)
# rtol is the stated bar (1e-5 on features, 2e-5 on gradients, which reduce over all nodes of the batch in a different
# fixed order than numpy).  The absolute term exists because the outputs are dot products with cancellation: an
# element that cancels to ~0 cannot be relatively exact in fp32 -- its error is a few ulp of the TERMS, whose size
# the largest entry of the element's row or column stands for (row alone fails for one-column outputs such as
# Fout = 1, where the "row" is the cancelling element itself).  The allowance is calibrated on fp32 arithmetic itself (this oracle run in float32
# against its float64 run on the test cases: Y <= 5 ulp of the row max, dX / dweight <= 5, dM_L <= 8, and the
# degenerate Fin = 1 case 17 ulp on dM_L, 120 on dX): 16 ulp for features, 32 for gradients.
# Tile outputs (res_W, res_L) are products / exponentials without cancellation: pure relative (atol_ulps = 0; values
# below the smallest normal float may be flushed).
FEATURE_RTOL = 1e-5
GRAD_RTOL = 2e-5
FEATURE_ULPS = 16
GRAD_ULPS = 32
EPS32 = float(np.finfo(np.float32).eps)
TINY32 = float(np.finfo(np.float32).tiny)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def rel_err(a, b):
    """max |a - b| / max |b| (tensor-level figure, used for reporting only)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(float(np.abs(b).max()) if b.size else 0.0, 1e-30)
    return float(np.abs(a - b).max() / den) if b.size else 0.0


def assert_close(a, b, rtol, what, atol_ulps=None):
    """Element-wise: |a - b| <= rtol |b| + atol_ulps eps scale (+ the flush-to-zero range), scale = the largest
    |b| in the element's row (last axis) or column (axis -2); whole-vector max for 1-D.  atol_ulps defaults to
    FEATURE_ULPS for rtol <= FEATURE_RTOL and GRAD_ULPS otherwise."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    assert np.all(np.isfinite(a)), "%s: non-finite values" % what
    assert np.all(np.isfinite(b)), "%s: non-finite values in the expected tensor" % what
    if b.size == 0:
        return
    if atol_ulps is None:
        atol_ulps = FEATURE_ULPS if rtol <= FEATURE_RTOL else GRAD_ULPS * max(1.0, rtol / GRAD_RTOL)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    if b64.ndim > 1:
        scale = np.maximum(np.abs(b64).max(axis=-1, keepdims=True), np.abs(b64).max(axis=-2, keepdims=True))
    else:
        scale = np.abs(b64).max()
    bound = rtol * np.abs(b64) + atol_ulps * EPS32 * scale + TINY32
    err = np.abs(a64 - b64)
    bad = err > bound
    if bad.any():
        k = np.unravel_index(np.argmax(err / bound), err.shape)
        raise AssertionError("%s: %d of %d elements outside rtol %.1e + %g ulp of the row/column max; worst at %s: got %.9g, "
                             "want %.9g (|d| = %.3e = %.1f x the bound; rel to tensor max %.2e)"
                             % (what, int(bad.sum()), b.size, rtol, atol_ulps, k, a64[k], b64[k], err[k],
                                float(err[k] / bound[k]), rel_err(a, b)))


def golden_params(g, prefix):
    return {k: g[prefix + "p_" + k] for k in ("weight", "bias", "M_L", "alpha", "beta") if prefix + "p_" + k in g}


def to_np(t):
    return t.detach().cpu().numpy()


def oracle_run(X, L0, p, K, mode, variant, G, Lp=None, dLsum=None, dtype=np.float64):
    """Oracle forward+backward in `dtype` (float64 = ground truth for synthetic cases)."""
    Xd = [x.astype(dtype) for x in X]
    Ld = [l.astype(dtype) for l in L0]
    pd_ = {k: v.astype(dtype) for k, v in p.items()}
    Lpd = None if Lp is None else [l.astype(dtype) for l in Lp]
    caches = orc.layer_forward(Xd, Ld, pd_, K, mode, variant, Lp_list=Lpd)
    ex = None if dLsum is None else [d.astype(dtype) for d in dLsum]
    tot = orc.layer_backward(caches, [g.astype(dtype) for g in G], pd_, K, mode, variant, ex)
    return caches, tot


def pack_tiles_np(tiles, plan):
    out = np.zeros(int(plan.tile_elems), dtype=np.float32)
    for b, t in enumerate(tiles):
        n, off = int(plan.n[b]), int(plan.tile_off[b])
        out[off:off + n * n] = np.asarray(t, dtype=np.float32).reshape(-1)
    return out


def unpack_tiles_np(flat, plan):
    flat = np.asarray(flat)
    return [flat[int(plan.tile_off[b]):int(plan.tile_off[b]) + int(plan.n[b]) ** 2].reshape(int(plan.n[b]), -1)
            for b in range(plan.B)]


def make_layer(agcn_amd, variant, Fout, Fin, B, K, mode, p, device, activation="relu"):
    cls = agcn_amd.SGC_LL_Reslap if variant == orc.SGC_LL_RESLAP else agcn_amd.SGC_LL
    layer = cls(Fout, Fin, B, activation=activation, K=K,
                laplacian_mode="paper" if mode == orc.PAPER else "reference", device=device)
    layer.load_vars(p)
    return layer


def synth_case(seed, ns, Fin, Fout, K, variant, atoms=False, fancy=True):
    """Random graphs + parameters (numpy, float32)."""
    rs = np.random.RandomState(seed)
    X, L0, csr = [], [], []
    for n in ns:
        adj = orc.synth_adj_lists(rs, int(n))
        if atoms and Fin == 75:
            x = orc.synth_atom_features(rs, int(n))
        else:
            x = np.maximum(rs.randn(int(n), Fin), 0).astype(np.float32)
            if n >= 6:
                x[3] = x[1]          # duplicate rows: r_ij == 0
        ip, ix, dat, L = orc.graph_laplacian(adj)
        X.append(x)
        L0.append(L.astype(np.float32))
        csr.append((ip, ix, dat))
    p = orc.init_params(rs, Fin, Fout, K, variant)
    if fancy:
        p["bias"] = (0.1 * rs.randn(Fout)).astype(np.float32)
        p["alpha"] = np.array([0.7], dtype=np.float32)
        if variant == orc.SGC_LL_RESLAP:
            p["beta"] = np.array([1.3], dtype=np.float32)
    G = [rs.randn(int(n), Fout).astype(np.float32) for n in ns]
    return rs, X, L0, csr, p, G


def run_packed(agcn_amd, device, ns, X, csr, p, G, Fin, Fout, K, mode, variant, Lp=None, dLsum=None,
               activation="relu"):
    """Runs the HIP layer on a packed batch (CSR Laplacian); returns numpy results."""
    gb = agcn_amd.GraphBatch.from_arrays(np.asarray(ns, np.int32), np.concatenate(X, 0), csr, device)
    plan = gb.plan
    assert gb.l0_symmetric, "synthetic Laplacians are symmetric: the slab path must be eligible"
    layer = make_layer(agcn_amd, variant, Fout, Fin, len(ns), K, mode, p, device, activation)
    Xt = gb.X.clone().requires_grad_(True)
    x = {"node_features": Xt, "graph_batch": gb}
    Lpt = None
    if variant == orc.SGC_LL_RESLAP:
        if Lp is not None:
            Lpt = torch.from_numpy(pack_tiles_np(Lp, plan)).to(device).requires_grad_(True)
            x["res_lap"] = agcn_amd.TileViews(Lpt, plan)
        else:
            x["res_lap"] = []
        nodes, res_L, res_W, L_sum = layer(x)
    else:
        nodes, res_L, res_W = layer(x)
        L_sum = None
    Y = nodes.data
    loss = (Y * torch.from_numpy(np.concatenate(G, 0)).to(device)).sum()
    if dLsum is not None:
        loss = loss + (L_sum.data * torch.from_numpy(pack_tiles_np(dLsum, plan)).to(device)).sum()
    loss.backward()
    off = plan.row_off
    out = {
        "Y": [to_np(Y)[off[b]:off[b] + ns[b]] for b in range(len(ns))],
        "res_L": unpack_tiles_np(to_np(res_L.data), plan),
        "res_W": unpack_tiles_np(to_np(res_W.data), plan),
        "L_sum": None if L_sum is None else unpack_tiles_np(to_np(L_sum.data), plan),
        "dX": [to_np(Xt.grad)[off[b]:off[b] + ns[b]] for b in range(len(ns))],
        "dLp": None if Lpt is None or Lpt.grad is None else unpack_tiles_np(to_np(Lpt.grad), plan),
    }
    for k, v in layer.vars.items():
        out["d" + k] = None if v.grad is None else to_np(v.grad)
    return out


def compare_with_oracle(out, caches, tot, variant, mode, K, has_prev):
    B = len(caches)
    assert_close(np.concatenate(out["Y"]), np.concatenate([c["Y"] for c in caches]), FEATURE_RTOL, "Y")
    for b in range(B):
        assert_close(out["res_W"][b], caches[b]["S"], FEATURE_RTOL, "res_W[%d]" % b, atol_ulps=0)
        assert_close(out["res_L"][b], caches[b]["R2"], FEATURE_RTOL, "res_L[%d]" % b, atol_ulps=0)
        if variant == orc.SGC_LL_RESLAP:
            assert_close(out["L_sum"][b], caches[b]["La"], FEATURE_RTOL, "L_sum[%d]" % b)
    assert_close(np.concatenate(out["dX"]), np.concatenate(tot["dX"]), GRAD_RTOL, "dX")
    assert_close(out["dweight"], tot["dweight"], GRAD_RTOL, "dweight")
    assert_close(out["dbias"], tot["dbias"], GRAD_RTOL, "dbias")
    da = float(np.asarray(tot["dalpha"]).reshape(-1)[0])
    scale = max(abs(da), float(np.abs(tot["dweight"]).max()))
    assert abs(float(out["dalpha"][0]) - da) <= GRAD_RTOL * 10 * scale, "dalpha %r vs %r" % (out["dalpha"], da)
    if mode == orc.PAPER:
        if tot["dM_L"] is None:
            assert out["dM_L"] is None or np.all(out["dM_L"] == 0)
        else:
            assert_close(out["dM_L"], tot["dM_L"], 5 * GRAD_RTOL, "dM_L")
    else:
        assert out["dM_L"] is None
    if has_prev:
        db = float(np.asarray(tot["dbeta"]).reshape(-1)[0])
        assert abs(float(out["dbeta"][0]) - db) <= GRAD_RTOL * 10 * max(abs(db), scale), "dbeta"
        for b in range(B):
            assert_close(out["dLp"][b], tot["dLp"][b], GRAD_RTOL, "dL_prev[%d]" % b)
