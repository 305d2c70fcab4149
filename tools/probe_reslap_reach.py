#!/usr/bin/env python3
"""SGC_LL_Reslap 64 -> 64 (with L_prev) or SGC_LL 64 -> 64 on HIV-shaped 256-graph batches: one whose largest molecule is within the
cooperative members' reach (<= 128 atoms), one within the WIDE members' (129 .. 192) and one beyond both (that batch's largest
molecule replaced by one of 200 atoms), per agcn_set_fused_mode setting -- what the dispatch costs when a batch leaves the
specialised kernels.  usage: probe_reslap_reach.py reslap|sgc [paper|reference].  One JSON line per case: us per step (HIP-graph replay) + kernels."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import agcn_amd as amd               # noqa: E402
from agcn_amd import synth           # noqa: E402

dev = torch.device("cuda", 0)
lib = amd._ffi.lib()
variant = sys.argv[1] if len(sys.argv) > 1 else "reslap"
lmode = sys.argv[2] if len(sys.argv) > 2 else "paper"      # laplacian_mode


def pick(beyond):
    for seed in range(40, 400):
        nsa = synth.node_counts(np.random.RandomState(seed), 256, mu=np.log(23.0), hi=222)
        big = int((nsa > 96).sum())
        if 1 <= big <= 4 and ((128 < nsa.max() <= 192) if beyond else (nsa.max() <= 128)):
            return seed, int(nsa.max())
    raise SystemExit("no seed")


def timed(fn, steps=200, warmup=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps


from agcn_amd.graph_structure import laplacian_csr   # noqa: E402
for beyond in (False, True, None):
    seed, nmax = pick(bool(beyond))
    data = synth.tox21_batch(seed, 256, 64, mu=np.log(23.0), hi=222)
    if beyond is None:      # beyond every member: a 200-atom molecule in place of the largest one
        rs = np.random.RandomState(5)
        k, nmax = int(np.argmax(data["n"])), 200
        data["n"] = data["n"].copy()
        data["n"][k] = nmax
        data["X"][k] = np.maximum(rs.randn(nmax, 64), 0).astype(np.float32)
        data["csr"][k] = laplacian_csr(synth.molecule_adjacency(rs, nmax))
    gb = amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
    for fm in (1, 3):
        lib.agcn_set_fused_mode(fm)
        try:
            if variant == "reslap":
                layer = amd.SGC_LL_Reslap(64, 64, 256, K=2, laplacian_mode=lmode, device=dev)
                layer.load_vars(synth.glorot_params(np.random.RandomState(123), 64, 64, 2, reslap=True))
                Lp = 0.1 * torch.randn(gb.plan.tile_elems, device=dev)
                st = amd.SgcLLStep(layer, gb, L_prev=Lp, dY=torch.randn(gb.plan.rows, 64, device=dev))
            else:
                layer = amd.SGC_LL(64, 64, 256, K=2, laplacian_mode=lmode, device=dev)
                layer.load_vars(synth.glorot_params(np.random.RandomState(123), 64, 64, 2))
                st = amd.SgcLLStep(layer, gb, dY=torch.randn(gb.plan.rows, 64, device=dev))
            names = st.kernel_names()
            st.capture()
            t = timed(st.replay)
            for _ in range(3):
                st.step()
            torch.cuda.synchronize(dev)
            lib.agcn_profile_enable(1)
            for _ in range(20):
                st.step()
            prof = amd._ffi.profile_collect()
            lib.agcn_profile_enable(0)
            kus = {k: round(1e3 * ms / c, 2) for k, (ms, c) in prof.items() if c > 0}
            print(json.dumps({"variant": variant, "laplacian_mode": lmode, "largest": nmax, "rows": int(gb.plan.rows), "fused_mode": fm,
                              "us_per_step": round(t * 1e6, 1), "kernels": names, "kernel_us": kus}), flush=True)
        finally:
            lib.agcn_set_fused_mode(1)
