"""Stress: the cooperative launch layout (members exchanging through global memory, paired small graphs) must give the
same bits on every one of many thousand launches -- alone, as HIP-graph replays, and with TWO steps in flight on two
streams (members of both launches resident together) -- and never run into a bounded wait (NaN)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agcn_amd as amd  # noqa: E402
from agcn_amd import synth  # noqa: E402
from agcn_amd.engine import _overlapping_streams  # noqa: E402

dev = torch.device("cuda", 0)
scale = float(os.environ.get("AGCN_STRESS_SCALE", "1"))
bad = 0
# (the last case: SGC_LL_Reslap, whose members also exchange the graph-wide clip norms / inner products -- three exchanges in a
#  row per direction, csrc/spec_backward_coop.h: coop_graph_sum)
# (wide: a seed whose batch has a molecule of 129 .. 192 atoms -- the kernel variant with the wide members)
for (B, mu, hi, reslap, wide) in ((256, np.log(16.0), 132, False, False), (256, np.log(23.0), 128, False, False), (100, np.log(30.0), 96, False, False),
                                  (256, np.log(23.0), 128, True, False), (256, np.log(23.0), 222, False, True), (256, np.log(23.0), 222, True, True)):
    Fin = 64 if reslap else 75
    seed = 7
    if wide:
        for seed in range(40, 400):
            nsa = synth.node_counts(np.random.RandomState(seed), B, mu=mu, hi=hi)
            if 128 < nsa.max() <= 160:
                break
    data = synth.tox21_batch(seed, B, Fin, mu=mu, hi=hi)
    gb = amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
    assert (int(data["n"].max()) > 128) == wide
    if reslap:
        layer = amd.SGC_LL_Reslap(64, Fin, B, K=2, laplacian_mode="paper", device=dev)
        layer.load_vars(synth.glorot_params(np.random.RandomState(1), Fin, 64, 2, reslap=True))
        extra = dict(L_prev=0.1 * torch.randn(gb.plan.tile_elems, device=dev), dL_sum=0.1 * torch.randn(gb.plan.tile_elems, device=dev))
        assert amd._ffi.dispatch_info(layer, gb)[6] > 0, "the Reslap batch must have backward members"
    else:
        layer = amd.SGC_LL(64, Fin, B, K=2, laplacian_mode="paper", device=dev)
        layer.load_vars(synth.glorot_params(np.random.RandomState(1), Fin, 64, 2))
        extra = {}
    if wide:
        assert amd._ffi.dispatch_info(layer, gb)[0] & 7 == 7 and amd._ffi.dispatch_info(layer, gb)[6] >= 9, "the wide batch must run on members"
    dY = torch.randn(gb.plan.rows, 64, device=dev)
    steps = [amd.SgcLLStep(layer, gb, dY=dY, **extra) for _ in range(2)]
    steps[0].step()
    torch.cuda.synchronize()
    ref = (steps[0].bucket.clone(), steps[0].dX.clone(), steps[0].Y.clone(), steps[0].res_L.clone())
    assert torch.isfinite(ref[0]).all()
    streams = _overlapping_streams(dev, 2)
    diff = torch.zeros((), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    n_it = max(50, int(6000 * scale))
    for it in range(n_it):
        for k in (0, 1):        # two launches with members side by side
            with torch.cuda.stream(streams[k]):
                steps[k].step()
        if it % 50 == 0:
            for k in (0, 1):
                with torch.cuda.stream(streams[k]):
                    st = steps[k]
                    diff += (st.bucket != ref[0]).sum() + (st.dX != ref[1]).sum() + (st.Y != ref[2]).sum() + \
                        (st.res_L != ref[3]).sum()
    torch.cuda.synchronize()
    # graph replays, ten steps per graph
    steps[0].capture(steps=10)
    for it in range(max(20, int(1500 * scale))):
        steps[0].replay()
        if it % 50 == 0:
            diff += (steps[0].bucket != ref[0]).sum() + (steps[0].dX != ref[1]).sum()
    torch.cuda.synchronize()
    d = int(diff)
    print("B=%d max_n=%d%s: %d mismatching values, %.1f s" % (B, int(data["n"].max()), " Reslap" if reslap else "", d, time.perf_counter() - t0), flush=True)
    bad += d
print("mismatches:", bad)
sys.exit(1 if bad else 0)
