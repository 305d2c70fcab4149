"""Debug: SGC_LL_Reslap forward with / without members (AGCN_RESLAP_COOP), tensor by tensor."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import agcn_amd as amd
    import agcn_oracle as orc
    from helpers import synth_case, run_packed
    mode = orc.REFERENCE if sys.argv[2] == "reference" else orc.PAPER
    ns, Fin, Fout, K = [48, 17], 128, 128, 2
    out_all = {}
    for rep in range(int(sys.argv[3])):
        rs, X, L0, csr, p, G = synth_case(1003, ns, Fin, Fout, K, orc.SGC_LL_RESLAP, atoms=False)
        Lp = [(0.2 * rs.randn(n, n)).astype(np.float32) for n in ns]
        dLs = [(0.3 * rs.randn(n, n)).astype(np.float32) for n in ns]
        out = run_packed(amd, "cuda:0", ns, X, csr, p, G, Fin, Fout, K, mode, orc.SGC_LL_RESLAP, Lp, dLs)
        for k, v in out.items():
            if v is None:
                continue
            if isinstance(v, list):
                v = np.concatenate([np.asarray(x).reshape(-1) for x in v])
            out_all.setdefault(k, []).append(np.asarray(v))
    np.savez(sys.argv[4], **{k: np.stack(v) for k, v in out_all.items()})
else:
    for mode in ("reference", "paper"):
        for coop in ("0", "3"):
            env = dict(os.environ, AGCN_RESLAP_COOP=coop)
            subprocess.check_call([sys.executable, __file__, "child", mode, "12", "/tmp/dbg_%s_%s.npz" % (mode, coop)], env=env)
        a, b = np.load("/tmp/dbg_%s_0.npz" % mode), np.load("/tmp/dbg_%s_3.npz" % mode)
        for k in a.files:
            x, y = a[k], b[k]
            d = np.abs(x - y)
            sc = np.abs(x).max() + 1e-30
            print(mode, k, x.shape, "max |coop - plain| / max per rep:", np.round(d.reshape(d.shape[0], -1).max(1) / sc, 8).tolist())
            # run-to-run variation of the coop path itself
            dv = np.abs(y - y[0:1]).reshape(y.shape[0], -1).max(1) / sc
            print("      coop rep-to-rep:", np.round(dv, 8).tolist())
            if k == "L_sum":
                for rep in range(x.shape[0]):
                    e = np.abs(x[rep][:2304] - y[rep][:2304]).reshape(48, 48).max(1)
                    r = (y[rep][:2304].reshape(48, 48) / (x[rep][:2304].reshape(48, 48) + 1e-30))
                    print("      rep", rep, "rows wrong:", np.nonzero(e > 1e-5)[0].tolist(), "ratio coop/plain of row 0, 16, 32:", r[0, 1], r[16, 1], r[32, 1])
