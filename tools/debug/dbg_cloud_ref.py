"""Debug: the 128-feature Reslap cloud backward in reference mode (random_batches[4-reference] failure)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import agcn_amd as amd
import agcn_oracle as orc
from helpers import oracle_run, pack_tiles_np, unpack_tiles_np
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
DEV = "cuda:0"

def run(sizes, Fin, Fout, mode, variant, seed=7004, with_prev=True, fused_off=True, consume=False, with_dls=True, alpha=0.75, beta=0.9, only=None):
    rs = np.random.RandomState(seed)
    if consume:
        sizes = [int(64 * rs.randint(1, 8)) for _ in range(int(rs.randint(1, 8)))]
    K = 2
    Xs, Ls, Gs = [], [], []
    for n in sizes:
        X, L, G = T._cloud_case(rs, n, Fin, Fout, thresh=0.6)
        Xs.append(X); Ls.append(L); Gs.append(G)
    p = orc.init_params(rs, Fin, Fout, K, variant)
    p["alpha"] = np.array([alpha], dtype=np.float32)
    reslap = variant == orc.SGC_LL_RESLAP
    if reslap:
        p["beta"] = np.array([beta], dtype=np.float32)
    Lp = [T._sym_prev(rs, n) for n in sizes] if reslap and with_prev else None
    dLs = [(0.01 * rs.randn(n, n)).astype(np.float32) for n in sizes] if reslap and with_prev else None
    if not with_dls: dLs = None
    if only is not None:
        keep = only
        sizes = [sizes[keep]]; Xs = [Xs[keep]]; Ls = [Ls[keep]]; Gs = [Gs[keep]]
        if Lp is not None: Lp = [Lp[keep]]
        if dLs is not None: dLs = [dLs[keep]]
    plan = amd.Plan(sizes, 0, DEV)
    l0 = np.zeros(int(plan.tile_off[-1]), np.float32)
    for b, L in enumerate(Ls):
        l0[int(plan.tile_off[b]):int(plan.tile_off[b]) + L.size] = L.reshape(-1)
    gb = amd.GraphBatch(plan, torch.from_numpy(np.concatenate(Xs)).to(DEV), l0_dense=torch.from_numpy(l0).to(DEV), l0_ld=0, l0_symmetric=True)
    lib = amd._ffi.lib()
    if fused_off: lib.agcn_set_fused_mode(0)
    lib.agcn_profile_enable(1); amd._ffi.profile_collect()
    layer = T.make_layer(amd, variant, Fout, Fin, len(sizes), K, mode, p, DEV)
    Xt = gb.X.clone().requires_grad_(True)
    x = {"node_features": Xt, "graph_batch": gb}
    if reslap:
        if Lp is not None:
            Lpt = torch.from_numpy(pack_tiles_np(Lp, plan)).to(DEV).requires_grad_(True)
            x["res_lap"] = amd.TileViews(Lpt, plan, symmetric=True)
        else:
            x["res_lap"] = []
        nodes, res_L, res_W, L_sum = layer(x)
        loss = (nodes.data * torch.from_numpy(np.concatenate(Gs)).to(DEV)).sum()
        if dLs is not None:
            loss = loss + (L_sum.data * torch.from_numpy(pack_tiles_np(dLs, plan)).to(DEV)).sum()
    else:
        nodes, res_L, res_W = layer(x)
        loss = (nodes.data * torch.from_numpy(np.concatenate(Gs)).to(DEV)).sum()
    loss.backward(); torch.cuda.synchronize()
    prof = {k: v[1] for k, v in amd._ffi.profile_collect().items() if v[1]}
    lib.agcn_profile_enable(0); lib.agcn_set_fused_mode(1)
    caches, tot = oracle_run(Xs, Ls, p, K, mode, variant, Gs, Lp, dLs)
    dX = Xt.grad.cpu().numpy(); r0 = 0
    Yh = nodes.data.detach().cpu().numpy()
    Yo = np.concatenate([c["Y"] for c in caches]); Ypo = np.concatenate([c["Ypre"] for c in caches])
    Gall = np.concatenate(Gs)
    print("   Y err %.2e; mask mismatches %d of %d; min |Ypre| at mismatches %s" % (np.abs(Yh - Yo).max() / np.abs(Yo).max(), int(((Yh > 0) != (Ypo > 0)).sum()), Yh.size,
          np.sort(np.abs(Ypo[(Yh > 0) != (Ypo > 0)]))[-5:]))
    db_h = (Gall * (Yh > 0)).sum(0); db_o = (Gall * (Ypo > 0)).sum(0); db_got = layer.vars["bias"].grad.cpu().numpy()
    print("   dbias: got vs from-HIP-Y %.2e, got vs oracle %.2e" % (np.abs(db_got - db_h).max(), np.abs(db_got - db_o).max()))
    print("sizes", sizes, Fin, Fout, "mode", mode, "prev", with_prev, "dls", with_dls, "alpha", alpha, "beta", beta, "only", only)
    for nm in ("weight", "bias", "alpha") + (("beta",) if Lp is not None else ()):
        gg = layer.vars[nm].grad.cpu().numpy().reshape(-1); ww = np.asarray(tot["d" + nm]).reshape(-1)
        print("   d%s err %.2e" % (nm, np.abs(gg - ww).max() / (np.abs(ww).max() + 1e-30)))
    for b, n in enumerate(sizes):
        e = np.abs(dX[r0:r0+n] - tot["dX"][b]); s = np.abs(tot["dX"][b]).max()
        h = Fin // 2
        rows_bad = np.nonzero((e > 1e-4 * s).any(1))[0]
        print("  graph %d n=%d: max err/scale cols[:h] %.2e cols[h:] %.2e; bad rows %d (first %s)" % (b, n, e[:, :h].max()/s, e[:, h:].max()/s, len(rows_bad), rows_bad[:8]))
        if len(rows_bad):
            bad = e > 1e-4 * s
            print("     bad per column block of 16:", [int(bad[:, 16*k:16*k+16].sum()) for k in range(Fin // 16)])
            print("     bad per row block of 16:", [int(bad[16*k:16*k+16].sum()) for k in range(n // 16)])
            c = caches[b]
            Yb = c["Y"]; dead = np.nonzero((Yb <= 0).all(1))[0]
            print("     rows with all-dead relu:", len(dead), dead[:8], " all bad rows:", rows_bad)
            Gp = Gs[b] * (c["Ypre"] > 0)
            W = p["weight"].astype(np.float64).reshape(Fin, K, Fout)
            g0 = Gp @ W[:, 0, :].T; g1 = Gp @ W[:, 1, :].T
            want = g0 + c["La"].T @ g1
            print("     oracle dX vs g0 + La^T g1: %.2e" % (np.abs(want - tot["dX"][b]).max() / s))
            got = dX[r0:r0+n]
            print("     got - g0 vs La^T g1: %.2e ; got - La^T g1 vs g0: %.2e" % (np.abs((got - g0) - c["La"].T @ g1).max() / s, 0))
            d = got - want
            # is the error a row-mix of g1 (an La error) or a column mix?
            sol = np.linalg.lstsq(g1, d.T @ np.eye(n), rcond=None) if False else None
            print("     err rows nonzero:", np.nonzero(np.abs(d).max(1) > 1e-5 * s)[0][:40])
            print("     err energy by column", np.round(np.abs(d).max(0)[:16] / s, 4))
            E = d @ np.linalg.pinv(g1)       # d = E g1  ->  E = error in La^T (n x n), if g1 has full row rank (n <= Fin? no)

        if reslap:
            ls = L_sum[b].detach().cpu().numpy(); el = np.abs(ls - caches[b]["La"]).max() / np.abs(caches[b]["La"]).max()
            print("     L_sum err %.2e" % el)
        r0 += n

def poison():
    """fill the caching allocator's free blocks with NaN so that reads of never-written memory show"""
    torch.cuda.synchronize()
    free = []
    for sz in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12):
        for _ in range(6):
            try:
                free.append(torch.full((sz // 4,), float("nan"), device=DEV))
            except Exception:
                break
    torch.cuda.synchronize()
    del free

if __name__ == "__main__":
    R, P = orc.REFERENCE, orc.PAPER
    V = orc.SGC_LL_RESLAP
    run(None, 128, 64, R, V, consume=True, only=3)
