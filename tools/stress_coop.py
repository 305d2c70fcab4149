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
for (B, mu, hi) in ((256, np.log(16.0), 132), (256, np.log(23.0), 128), (100, np.log(30.0), 96)):
    data = synth.tox21_batch(7, B, 75, mu=mu, hi=hi)
    gb = amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
    layer = amd.SGC_LL(64, 75, B, K=2, laplacian_mode="paper", device=dev)
    layer.load_vars(synth.glorot_params(np.random.RandomState(1), 75, 64, 2))
    dY = torch.randn(gb.plan.rows, 64, device=dev)
    steps = [amd.SgcLLStep(layer, gb, dY=dY) for _ in range(2)]
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
    print("B=%d max_n=%d: %d mismatching values, %.1f s" % (B, int(data["n"].max()), d, time.perf_counter() - t0), flush=True)
    bad += d
print("mismatches:", bad)
sys.exit(1 if bad else 0)
