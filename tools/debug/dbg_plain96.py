"""A 96-node graph on the PLAIN one-workgroup specialised kernels (members switched off) at every covered layer shape, against
the float64 oracle -- the case fuzz_spec.py found at B = 1030 (no members there), 128 -> 64."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for extra in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if extra not in sys.path:
        sys.path.insert(0, extra)
import agcn_oracle as orc  # noqa: E402
from helpers import compare_with_oracle, oracle_run, run_packed, synth_case  # noqa: E402

import agcn_amd as amd  # noqa: E402
lib = amd._ffi.lib()
for coop in (0, 1):
    lib.agcn_set_spec_mode(7 if coop else 3)
    for (Fin, Fout) in ((75, 64), (64, 128), (128, 128), (128, 64), (64, 64)):
        for nbig in (96, 95, 81, 80):
            ns = [20, nbig, 7, 33, 12, 50]
            rs, X, L0, csr, p, G = synth_case(777, ns, Fin, Fout, 2, orc.SGC_LL, atoms=(Fin == 75))
            try:
                out = run_packed(amd, "cuda:0", ns, X, csr, p, G, Fin, Fout, 2, orc.PAPER, orc.SGC_LL, None, None)
                caches, tot = oracle_run(X, L0, p, 2, orc.PAPER, orc.SGC_LL, G, None, None)
                compare_with_oracle(out, caches, tot, orc.SGC_LL, orc.PAPER, 2, has_prev=False)
                print("ok   coop=%d %d->%d n=%d" % (coop, Fin, Fout, nbig), flush=True)
            except Exception as e:  # noqa: BLE001
                print("FAIL coop=%d %d->%d n=%d: %s" % (coop, Fin, Fout, nbig, str(e)[:200]), flush=True)
lib.agcn_set_spec_mode(7)
