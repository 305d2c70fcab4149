"""Stress: the one-pass parameter-gradient kernel (cross-workgroup hand-off through sc1 stores / loads and an
arrival counter) must give bit-identical gradients on every launch, eager and graph-replayed, while other work
keeps the L2s busy."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agcn_amd as amd  # noqa: E402
from agcn_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
bad = 0
for (B, Fin, Fout, iters) in ((256, 75, 64, 3000), (256, 128, 128, 1500), (2048, 75, 64, 300), (37, 64, 128, 1500)):
    data = synth.tox21_batch(123, B, Fin)
    gb = amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
    layer = amd.SGC_LL(Fout, Fin, B, K=2, laplacian_mode="paper", device=dev)
    layer.load_vars(synth.glorot_params(np.random.RandomState(123), Fin, Fout, 2))
    st = amd.SgcLLStep(layer, gb, dY=torch.randn(gb.plan.rows, Fout, device=dev))
    st.step()
    torch.cuda.synchronize()
    ref = st.bucket.clone()
    noise = torch.randn(64 << 20, device=dev)            # 256 MB of unrelated traffic between steps
    for mode in ("eager", "graph"):
        if mode == "graph":
            st.capture()
        diff = torch.zeros((), dtype=torch.int64, device=dev)
        for it in range(iters):
            st.bucket.zero_()
            if it % 7 == 0:
                noise.mul_(1.0001)
            (st.replay if mode == "graph" else st.step)()
            diff += (st.bucket != ref).sum()          # every launch is checked, on the device
        torch.cuda.synchronize()
        if int(diff) != 0:
            bad += 1
            print("MISMATCH", B, Fin, Fout, mode, int(diff))
    print("ok", B, Fin, Fout, flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
