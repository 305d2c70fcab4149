#!/usr/bin/env python3
"""The reference's residual network on molecules (ResAGCN.py: ResAGCNResLap -- 4 x SGC_LL_Reslap in two residual blocks + 2 x
BlockEnd + DenseMol + GraphGatherMol) on a Tox21-shaped batch of 256 graphs, autograd forward + backward, EAGER launches: the
workload of bench.py's `secondary` line, for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_resnet.py`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import agcn_amd                      # noqa: E402
from agcn_amd import synth           # noqa: E402

dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
K, mode = 2, "paper"
data = synth.tox21_batch(123, 256, 75)
gb = agcn_amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
rs = np.random.RandomState(11)
net = agcn_amd.ResidualGraphMolResLap(75, 256, int(data["n"].max()))
fin, convs, bes = 75, [], []
for i, w in enumerate((64, 128, 128, 64)):
    lay = agcn_amd.SGC_LL_Reslap(w, fin, 256, K=K, laplacian_mode=mode, activation="relu", save_lap=i in (0, 2), device=dev)
    lay.load_vars(synth.glorot_params(rs, fin, w, K, reslap=True))
    convs.append(lay)
    net.add(lay)
    if i in (1, 3):
        be = agcn_amd.BlockEnd(1 + i // 2, fin, w, "relu", int(data["n"].max()), 256, device=dev)
        be.build(dev)
        bes.append(be)
        net.add(be)
    fin = w
dn = agcn_amd.DenseMol(256, 64, activation="relu", device=dev)
dn.build(dev)
net.add(dn)
net.add(agcn_amd.GraphGatherMol(256, activation="tanh"))
G = torch.from_numpy(rs.randn(256, 256).astype(np.float32)).to(dev)
params = [v for lay in convs for v in lay.vars.values()] + dn.parameters() + [b.vars["weight"] for b in bes]
for it in range(steps):
    for q in params:
        q.grad = None
    net(gb).backward(G)
torch.cuda.synchronize()
print("ran", steps, "steps")
