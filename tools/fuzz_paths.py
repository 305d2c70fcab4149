"""Randomised cross-check of every kernel path against the oracle (float64): random widths, graph sizes,
variants, modes, with and without L_prev / upstream dL_sum, leaky slopes != 1.
usage: python tools/fuzz_paths.py [n_cases] [seed] [big]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for extra in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if extra not in sys.path:
        sys.path.insert(0, extra)
import agcn_oracle as orc  # noqa: E402
from helpers import compare_with_oracle, oracle_run, run_packed, synth_case  # noqa: E402

DEV = "cuda:0"
# name: (fused_mode, slab_mode, fused_global_b, lds_limit)
PATHS = {"auto": (1, 1, -1, 160 * 1024), "fused_lds": (1, 1, 0, 160 * 1024), "fused_gb": (1, 1, 1, 160 * 1024),
         "slab": (0, 1, -1, 160 * 1024), "pergraph_lds": (0, 0, -1, 160 * 1024), "pergraph_global": (0, 0, -1, 0)}


def run(n_cases, seed, fin_choices=(1, 3, 5, 16, 33, 64, 75, 128), verbose=True, mu=np.log(12), sigma=0.7, nmax=110):
    import agcn_amd as amd
    lib = amd._ffi.lib()
    rs0 = np.random.RandomState(seed)
    fails = 0
    try:
        for case in range(n_cases):
            B = int(rs0.randint(1, 9))
            ns = [int(v) for v in np.clip(np.rint(rs0.lognormal(mu, sigma, size=B)), 1, nmax)]
            Fin = int(rs0.choice(list(fin_choices)))
            Fout = int(rs0.choice([1, 4, 17, 64, 128]))
            K = int(rs0.choice([1, 2, 2, 2, 3]))
            variant = orc.SGC_LL_RESLAP if rs0.rand() < 0.4 else orc.SGC_LL
            mode = orc.PAPER if rs0.rand() < 0.6 else orc.REFERENCE
            rs, X, L0, csr, p, G = synth_case(10_000 + case, ns, Fin, Fout, K, variant)
            p["alpha"] = np.array([rs0.choice([1.0, 0.7, 0.2])], np.float32)
            Lp = dLs = None
            if variant == orc.SGC_LL_RESLAP:
                if rs0.rand() < 0.7:
                    Lp = [(0.2 * rs.randn(n, n)).astype(np.float32) for n in ns]
                    p["beta"] = np.array([rs0.choice([1.0, 1.3])], np.float32)
                if rs0.rand() < 0.5:
                    dLs = [(0.3 * rs.randn(n, n)).astype(np.float32) for n in ns]
            caches, tot = oracle_run(X, L0, p, K, mode, variant, G, Lp, dLs)
            tot32 = oracle_run(X, L0, p, K, mode, variant, G, Lp, dLs, dtype=np.float32)[1]
            for name, (fm, sm, gbm, lim) in PATHS.items():
                lib.agcn_set_fused_mode(fm)
                lib.agcn_set_slab_mode(sm)
                lib.agcn_set_fused_global_b(gbm)
                lib.agcn_set_lds_limit(lim)
                try:
                    out = run_packed(amd, DEV, ns, X, csr, p, G, Fin, Fout, K, mode, variant, Lp, dLs)
                    compare_with_oracle(out, caches, tot, variant, mode, K, has_prev=Lp is not None, tot32=tot32)
                except Exception as e:      # noqa: BLE001
                    fails += 1
                    print("FAIL case %d path %s: ns=%s Fin=%d Fout=%d K=%d variant=%d mode=%d prev=%s dLs=%s -> %s"
                          % (case, name, ns, Fin, Fout, K, variant, mode, Lp is not None, dLs is not None, str(e)[:200]))
    finally:
        lib.agcn_set_fused_mode(1)
        lib.agcn_set_slab_mode(1)
        lib.agcn_set_fused_global_b(-1)
        lib.agcn_set_lds_limit(160 * 1024)
    if verbose:
        print("cases %d x %d paths, failures %d" % (n_cases, len(PATHS), fails))
    return fails


if __name__ == "__main__":
    big = len(sys.argv) > 3 and sys.argv[3] == "big"       # graphs up to 400 nodes (slab / global-tile territory)
    kw = dict(mu=np.log(50), sigma=0.9, nmax=400) if big else {}
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0, **kw) else 0)
