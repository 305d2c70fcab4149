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
# Parameter gradients that reduce over every pair of every graph (dM_L, dalpha, dbeta) carry NO extra multiplier: the
# main parametrised tests run the oracle a second time in float32 and print its error next to the HIP path's
# (assert_close ref32 / _scalar_close): the HIP results sit at 0.0-0.4 of the plain gradient bound, where the fp32
# oracle itself sits (0.0-0.5; once 2.7 on a scalar).
# Tile outputs (res_W, res_L) are products / exponentials without cancellation: pure relative (atol_ulps = 0; values
# below the smallest normal float may be flushed).
MAX_TIES = 4
FEATURE_RTOL = 1e-5
GRAD_RTOL = 2e-5
# The absolute allowances, in ulps of the row / column maximum: TWICE the worst need observed over the whole GPU suite on MI355X
# (profiles/r06_parity_margins.json, `ulps_needed`: the suite records it per comparison): features 3.6 -> 8; dX / dweight / dbias /
# dL_prev 5.5 -> 12; dM_L (a sum over every PAIR of every graph) 19.2 -> 40, flat in the graph size.  Rounds 1-5 ran with 16 / 32 and
# a sqrt(n / 64) size factor that allowed 64 ulps where 7.9 were needed.
FEATURE_ULPS = 8
GRAD_ULPS = 12
DML_ULPS = 40
SCALAR_RTOL = 5e-6       # dalpha / dbeta: observed at most 0.1 x GRAD_RTOL of the weight-gradient scale
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


K32_CAP = 8.0     # the widest a bound is ever calibrated to: 8 x the stated one (the fp32 oracle at 4 x)

# Every comparison leaves its MARGIN behind: (test id, tensor, bound) -> the worst |err| / bound seen, so that the distance
# between what the kernels achieve and what the tests would let pass is a committed number, not an impression
# (tests/conftest.py writes gpurun_out/parity_margins.json at the end of a `-m gpu` session; the round's copy lives in
# profiles/).  A 10 x regression of a kernel's error must fail a test: no bound may be looser than 4 x what was observed.
MARGINS = {}


def size_factor(n):
    """Factor on the absolute (ulp) allowance for outputs that are, or depend on, sums over the n nodes of a large graph
    (64 nodes is the size the default allowance was calibrated on; observed need at n = 1024: 7.9 ulps on features against 3.6 at
    molecule sizes -- the fourth root, (n / 64)^(1/4) = 2 at n = 1024, covers it twice)."""
    return float((max(float(n), 64.0) / 64.0) ** 0.25)


def _record_margin(what, rtol, ulps, ratio, k32=None, factor=1.0, ulps_needed=None):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" (")[0]
    import re as _re
    key = (test, _re.sub(r"\[\d+\]", "[*]", what), float(rtol), None if ulps is None else round(float(ulps), 3))
    m = MARGINS.get(key)
    if m is None:
        m = MARGINS[key] = {"worst": 0.0, "calls": 0, "k32": None, "calibration": 1.0, "ulps_needed": None}
    if ulps_needed is not None:
        m["ulps_needed"] = max(m["ulps_needed"] or 0.0, float(ulps_needed))
    m["worst"] = max(m["worst"], float(ratio))
    m["calls"] += 1
    if k32 is not None:
        m["k32"] = max(m["k32"] or 0.0, float(k32))
    m["calibration"] = max(m["calibration"], float(factor))


def assert_close(a, b, rtol, what, atol_ulps=None, ref32=None, scale_floor=0.0):
    """Element-wise: |a - b| <= rtol |b| + atol_ulps eps scale (+ the flush-to-zero range), scale = the largest
    |b| in the element's row (last axis) or column (axis -2); whole-vector max for 1-D.  atol_ulps defaults to
    FEATURE_ULPS for rtol <= FEATURE_RTOL and GRAD_ULPS otherwise.
    ref32: the SAME quantity from the oracle run in float32.  The bound is then calibrated per case on what fp32
    arithmetic of the reference algorithm itself achieves: with k32 = the worst fp32-oracle error in units of the
    bound, the HIP result may be max(1, 2 k32) bounds away -- never looser than twice the oracle's own fp32 error,
    never tighter than the stated bar -- and both figures are printed (pytest -s / on failure)."""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    assert np.all(np.isfinite(a)), "%s: non-finite values" % what
    assert np.all(np.isfinite(b)), "%s: non-finite values in the expected tensor" % what
    if b.size == 0:
        return
    if "dm_l" in what.lower() and rtol > FEATURE_RTOL:
        atol_ulps = DML_ULPS * max(1.0, rtol / GRAD_RTOL)      # (whatever the caller scaled for the graph size: flat, see DML_ULPS)
    if atol_ulps is None:
        atol_ulps = FEATURE_ULPS if rtol <= FEATURE_RTOL else GRAD_ULPS * max(1.0, rtol / GRAD_RTOL)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    if b64.ndim > 1:
        scale = np.maximum(np.abs(b64).max(axis=-1, keepdims=True), np.abs(b64).max(axis=-2, keepdims=True))
    else:
        scale = np.abs(b64).max()
    # scale_floor: a magnitude the row / column maxima are not allowed to fall below (the batch-wide scale of a per-graph gradient whose
    # tile may be analytically ZERO: the clip adjoint of a 1 x 1 Laplacian tile is a / |a| - a a a / |a|^3)
    scale = np.maximum(scale, float(scale_floor))
    bound = rtol * np.abs(b64) + atol_ulps * EPS32 * scale + TINY32
    err = np.abs(a64 - b64)
    stated = float((err / bound).max())      # in units of the STATED bound (before any fp32-oracle calibration)
    # how many ulps of the row / column maximum the absolute term would have to be for this tensor to pass at this rtol
    need = float((np.maximum(err - rtol * np.abs(b64) - TINY32, 0.0) / (EPS32 * np.maximum(scale, TINY32))).max())
    if ref32 is None:
        _record_margin(what, rtol, atol_ulps, stated, ulps_needed=need)
    if ref32 is not None:
        k32 = float((np.abs(np.asarray(ref32, dtype=np.float64) - b64) / bound).max())
        kgpu = float((err / bound).max())
        print("%-12s fp32 oracle %.2f x the bound, HIP %.2f x the bound (rtol %.0e + %g ulp)" % (what, k32, kgpu, rtol, atol_ulps))
        # calibration, capped: twice the fp32 oracle's own error, never more than K32_CAP x the stated bound -- a case on
        # which the fp32 oracle itself is further off is ill-conditioned test data and must be fixed there, not absorbed
        assert k32 <= K32_CAP / 2.0, ("%s: the fp32 oracle is %.1f x the bound away from the float64 one: ill-conditioned case"
                                      % (what, k32))
        _record_margin(what, rtol, atol_ulps, stated, k32, min(max(1.0, 2.0 * k32), K32_CAP), ulps_needed=need)
        bound = bound * min(max(1.0, 2.0 * k32), K32_CAP)
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


def align_relu_ties(caches, p, Y_got, ulps=64):
    """The activation's derivative at a pre-activation that is zero to within fp32 rounding is decided by that rounding
    (relu'(+3e-8) = 1, relu'(-3e-8) = 0): a float64 oracle and an fp32 kernel may legitimately land on different sides,
    and ONE flipped mask element moves a whole row of dX, a column of dweight and -- through the dense Laplacian -- every
    gradient behind it (seen on tests/test_gpu_parity.py::test_point_cloud_kernels_random_batches[4-reference]:
    Ypre = 3.3e-8, |dY| = 0.82).  Where |Ypre| <= ulps * eps32 * (|C| |W| + |b|) -- the rounding bound of that element's own
    fp32 dot product -- and the signs differ, the oracle's backward takes the side the kernel's Y is on.  Returns the number
    of elements moved (0 in nearly every case)."""
    W, bias = np.abs(np.asarray(p["weight"], dtype=np.float64)), np.abs(np.asarray(p["bias"], dtype=np.float64))
    moved, r0 = 0, 0
    for c in caches:
        n = c["Ypre"].shape[0]
        got_pos = np.asarray(Y_got[r0:r0 + n]) > 0
        bound = ulps * EPS32 * (np.abs(c["C"]).astype(np.float64) @ W + bias)
        flip = (got_pos != (c["Ypre"] > 0)) & (np.abs(c["Ypre"]) <= bound)
        if flip.any():
            tiny = np.finfo(c["Ypre"].dtype).tiny
            c["Ypre"] = np.where(flip, np.where(got_pos, tiny, -tiny).astype(c["Ypre"].dtype), c["Ypre"])
            moved += int(flip.sum())
        r0 += n
    return moved


def oracle_run(X, L0, p, K, mode, variant, G, Lp=None, dLsum=None, dtype=np.float64, Y_tie=None):
    """Oracle forward+backward in `dtype` (float64 = ground truth for synthetic cases).  Y_tie: the kernel's Y (packed
    rows), to settle activation ties at |Ypre| ~ fp32 rounding (align_relu_ties)."""
    Xd = [x.astype(dtype) for x in X]
    Ld = [l.astype(dtype) for l in L0]
    pd_ = {k: v.astype(dtype) for k, v in p.items()}
    Lpd = None if Lp is None else [l.astype(dtype) for l in Lp]
    caches = orc.layer_forward(Xd, Ld, pd_, K, mode, variant, Lp_list=Lpd)
    if Y_tie is not None:
        moved = align_relu_ties(caches, pd_, Y_tie)
        if moved:
            print("oracle_run: %d activation tie(s) settled on the kernel's side" % moved)
        _record_margin("relu ties moved", 0.0, None, float(moved))
        # ties are pre-activations inside fp32 rounding of zero: one in a few hundred thousand elements.  More than a
        # handful means the kernel's Y is WRONG near zero, not tied
        assert moved <= MAX_TIES, "%d activation ties moved to the kernel's side (at most %d expected)" % (moved, MAX_TIES)
    ex = None if dLsum is None else [d.astype(dtype) for d in dLsum]
    tot = orc.layer_backward(caches, [g.astype(dtype) for g in G], pd_, K, mode, variant, ex)
    return caches, tot


def oracle_pair(X, L0, p, K, mode, variant, G, Lp=None, dLsum=None, Y_got=None):
    """The float64 oracle (ground truth; activation ties settled on the side the kernel's Y is on, if given) and the float32
    oracle (the calibration of the parameter-gradient bounds: what the reference's own arithmetic achieves) of one case.
    The float32 run takes ITS ties from the float64 run: a mask element that flips between the two oracles would otherwise
    show up as an fp32-oracle error of thousands of bounds on that graph and silently widen the calibrated bound
    (test_full_size_every_graph_every_gradient[reslap-reference-b256], graph 11, before the calibration was capped).
    Returns (caches, tot, tot32)."""
    caches, tot = oracle_run(X, L0, p, K, mode, variant, G, Lp, dLsum, Y_tie=Y_got)
    ypre64 = np.concatenate([c["Ypre"] for c in caches])
    _, tot32 = oracle_run(X, L0, p, K, mode, variant, G, Lp, dLsum, dtype=np.float32, Y_tie=ypre64)
    return caches, tot, tot32


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


def _scalar_close(got, want, want32, scale, what, graphs=256):
    """Scalar gradients (dalpha, dbeta: sums over every pair of every graph): SCALAR_RTOL of the weight-gradient scale, or
    twice the error the fp32 oracle itself makes on this case, whichever is larger (printed).  The bar was set on 256-graph
    batches; the fp32 rounding of a sum over B graphs grows like sqrt(B), and so does the bar beyond 256 graphs
    (tools/fuzz_spec.py, seed 23: 1030 graphs at 5.05e-6 without an fp32 oracle to calibrate on)."""
    got, want = float(np.asarray(got).reshape(-1)[0]), float(np.asarray(want).reshape(-1)[0])
    # (+ TINY32: a case whose gradients are all zero -- one output feature, every activation clipped -- has no scale at all)
    bar = SCALAR_RTOL * float(np.sqrt(max(int(graphs), 256) / 256.0)) * max(abs(want), scale) + TINY32
    e32 = abs(float(np.asarray(want32).reshape(-1)[0]) - want) if want32 is not None else None
    print("%-12s fp32 oracle %s, HIP %.2f x the bar (%.1e of the gradient scale)"
          % (what, "n/a" if e32 is None else "%.2f x the bar" % (e32 / bar), abs(got - want) / bar, SCALAR_RTOL))
    if e32 is not None:
        assert e32 <= K32_CAP / 2.0 * bar, "%s: the fp32 oracle is %.1f x the bar away from the float64 one: ill-conditioned case" % (what, e32 / bar)
    allowed = min(max(bar, 2.0 * e32), K32_CAP * bar) if e32 is not None else bar
    _record_margin(what, SCALAR_RTOL, None, abs(got - want) / bar, None if e32 is None else e32 / bar, allowed / bar)
    assert abs(got - want) <= allowed, "%s %r vs %r (allowed %.3e)" % (what, got, want, allowed)


def compare_with_oracle(out, caches, tot, variant, mode, K, has_prev, tot32=None):
    """tot32: the oracle's gradients from a float32 run of the same case (oracle_run(..., dtype=np.float32)[1]): the
    parameter-gradient bounds are then calibrated on it (assert_close, ref32)."""
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
    scale = float(np.abs(tot["dweight"]).max())
    _scalar_close(out["dalpha"], tot["dalpha"], None if tot32 is None else tot32["dalpha"], scale, "dalpha", graphs=B)
    if mode == orc.PAPER:
        if tot["dM_L"] is None:
            assert out["dM_L"] is None or np.all(out["dM_L"] == 0)
        elif tot32 is not None:
            assert_close(out["dM_L"], tot["dM_L"], GRAD_RTOL, "dM_L", ref32=tot32["dM_L"])
        else:
            assert_close(out["dM_L"], tot["dM_L"], GRAD_RTOL, "dM_L")
    else:
        assert out["dM_L"] is None
    if has_prev:
        _scalar_close(out["dbeta"], tot["dbeta"], None if tot32 is None else tot32["dbeta"], scale, "dbeta", graphs=B)
        dlp_scale = max(float(np.abs(t).max()) for t in tot["dLp"])
        for b in range(B):
            # (a 1 x 1 tile's clip adjoint is analytically 0 -- and numpy's fp32 evaluation cancels exactly where the kernels keep one
            #  rounding of the operands, tools/fuzz_paths.py case 24: the batch-wide gradient scale is the floor of its bound)
            one = np.asarray(tot["dLp"][b]).size == 1
            assert_close(out["dLp"][b], tot["dLp"][b], GRAD_RTOL, "dL_prev[%d]" % b,
                         ref32=None if tot32 is None else tot32["dLp"][b], scale_floor=dlp_scale if one else 0.0)


# ----------------------------------------------------------------------------------------------------------------------------
# The reference's point-cloud network (models/networks/Point_AGCN.py:34-90) as a chained oracle: DenseConnectedGraphResLap with
#   SGC_LL_Reslap(f1, xyz, save_lap, save_output), SGC_LL_Reslap(f2, f1, save_output), DenseBlockEnd(0, [f1, f2], f2),
#   SGC_LL_Reslap(f3, f2, save_lap, save_output), SGC_LL_Reslap(f4, f3, save_output), DenseBlockEnd(1, [f3, f4], f4),
#   DenseMol(final, f4), GraphGatherMol(tanh)
# The previous Laplacian of a convolution is the most recently SAVED one (tf_graphs.py:346,360-361): layers 2 and 3 both get
# layer 1's, layer 4 gets layer 3's -- so layer 1's L_sum collects two cross-layer gradients.
def point_cloud_network_params(rs, widths=(64, 128, 128, 64), final=256, raw=3, K=2, dense_scale=0.004):
    f1, f2, f3, f4 = widths
    P = {"conv": [], "K": K}
    for fin, fout in ((raw, f1), (f1, f2), (f2, f3), (f3, f4)):
        p = orc.init_params(rs, fin, fout, K, orc.SGC_LL_RESLAP)
        p["alpha"] = np.array([0.9], np.float32)
        p["beta"] = np.array([0.8], np.float32)
        p["bias"] = (0.05 * rs.randn(fout)).astype(np.float32)
        P["conv"].append(p)
    P["be"] = [{"Win": [orc.glorot_uniform(rs, (f1, f2)), orc.glorot_uniform(rs, (f2, f2))], "Wout": [],
                "b1": np.array([0.7], np.float32), "b2": np.array([1.0], np.float32)},
               {"Win": [orc.glorot_uniform(rs, (f3, f4)), orc.glorot_uniform(rs, (f4, f4))], "Wout": [orc.glorot_uniform(rs, (f2, f4))],
                "b1": np.array([0.6], np.float32), "b2": np.array([0.5], np.float32)}]
    # (a row-sum over ~1000 points goes into the tanh of the readout: small dense weights keep it off saturation)
    P["Wd"] = (dense_scale * orc.glorot_uniform(rs, (f4, final))).astype(np.float32)
    P["bd"] = (0.001 * rs.randn(final)).astype(np.float32)
    return P


def point_cloud_network_oracle(Xs, Ls, P, Gout, mode, dtype=np.float64, Y_ties=None):
    """Forward + backward of the network on per-cloud arrays.  Y_ties: per convolution, the kernel's (or the float64 oracle's)
    packed Y, to settle activation ties (align_relu_ties).  Returns a dict: out [B, final], conv Y's, and the gradients."""
    K = P["K"]
    c_ = lambda d: {k: np.asarray(v).astype(dtype) for k, v in d.items()}  # noqa: E731
    pc = [c_(p) for p in P["conv"]]
    X = [x.astype(dtype) for x in Xs]
    L0 = [l.astype(dtype) for l in Ls]
    V = orc.SGC_LL_RESLAP
    ns = [x.shape[0] for x in Xs]

    def conv(i, h, Lp):
        c = orc.layer_forward(h, L0, pc[i], K, mode, V, Lp_list=Lp)
        if Y_ties is not None:
            align_relu_ties(c, pc[i], Y_ties[i])
        return c

    def be_fwd(i, x, ins, outs):
        b = P["be"][i]
        res = [orc.dense_block_end_forward(x[g], [a[g] for a in ins], [w.astype(dtype) for w in b["Win"]], [o[g] for o in outs],
                                           [w.astype(dtype) for w in b["Wout"]], b["b1"].astype(dtype), b["b2"].astype(dtype))
               for g in range(len(ns))]
        return [r[0] for r in res], [r[1] for r in res]

    c1 = conv(0, X, None)
    Y1, La1 = [c["Y"] for c in c1], [c["La"] for c in c1]
    c2 = conv(1, Y1, La1)
    Y2 = [c["Y"] for c in c2]
    B0, pre0 = be_fwd(0, Y2, [Y1, Y2], [])
    c3 = conv(2, B0, La1)
    Y3, La3 = [c["Y"] for c in c3], [c["La"] for c in c3]
    c4 = conv(3, Y3, La3)
    Y4 = [c["Y"] for c in c4]
    B1, pre1 = be_fwd(1, Y4, [Y3, Y4], [B0])
    Wd, bd = P["Wd"].astype(dtype), P["bd"].astype(dtype)
    D = [orc.dense_mol_forward(x, Wd, bd) for x in B1]
    out, pre = orc.graph_gather_forward(D, "tanh")
    res = {"out": out, "Y": [np.concatenate(y) for y in (Y1, Y2, Y3, Y4)], "La1": La1, "La3": La3}
    if Gout is None:
        return res
    gD = orc.graph_gather_backward(out, pre, Gout.astype(dtype), ns, "tanh")
    dm = [orc.dense_mol_backward(x, Wd, g) for x, g in zip(B1, gD)]
    res["dWd"], res["dbd"] = sum(d[1] for d in dm), sum(d[2] for d in dm)
    gB1 = [d[0] for d in dm]

    def be_bwd(i, ins, outs, pres, G):
        b = P["be"][i]
        r = [orc.dense_block_end_backward([a[g] for a in ins], [w.astype(dtype) for w in b["Win"]], [o[g] for o in outs],
                                          [w.astype(dtype) for w in b["Wout"]], b["b1"].astype(dtype), b["b2"].astype(dtype), pres[g], G[g])
             for g in range(len(ns))]
        dx = [x[0] for x in r]
        da = [[x[1][l] for x in r] for l in range(len(ins))]
        do = [[x[3][l] for x in r] for l in range(len(outs))]
        dWi = [sum(x[2][l] for x in r) for l in range(len(ins))]
        dWo = [sum(x[4][l] for x in r) for l in range(len(outs))]
        return dx, da, do, dWi, dWo, sum(x[5] for x in r), sum(x[6] for x in r)

    add = lambda a, b: [u + v for u, v in zip(a, b)]  # noqa: E731
    dx, da, do, res["be1_dWin"], res["be1_dWout"], res["be1_db1"], res["be1_db2"] = be_bwd(1, [Y3, Y4], [B0], pre1, gB1)
    gY4 = add(dx, da[1])
    t4 = orc.layer_backward(c4, gY4, pc[3], K, mode, V)
    gY3 = add(t4["dX"], da[0])
    t3 = orc.layer_backward(c3, gY3, pc[2], K, mode, V, t4["dLp"])            # layer 4's dL_prev arrives on layer 3's L_sum
    gB0 = add(t3["dX"], do[0])
    dx, da, _, res["be0_dWin"], _, res["be0_db1"], _ = be_bwd(0, [Y1, Y2], [], pre0, gB0)
    gY2 = add(dx, da[1])
    t2 = orc.layer_backward(c2, gY2, pc[1], K, mode, V)
    gY1 = add(t2["dX"], da[0])
    t1 = orc.layer_backward(c1, gY1, pc[0], K, mode, V, add(t2["dLp"], t3["dLp"]))   # layers 2 AND 3 read layer 1's L_sum
    res["conv_grads"] = [t1, t2, t3, t4]
    res["dX"] = np.concatenate(t1["dX"])
    return res
