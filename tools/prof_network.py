#!/usr/bin/env python3
"""The reference's whole point-cloud network (Point_AGCN.py: 4 x SGC_LL_Reslap in two dense blocks + 2 x DenseBlockEnd + DenseMol +
GraphGatherMol) on 32 clouds of 1024 points, autograd forward + backward, EAGER launches: the workload of bench.py's `secondary`
line, for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_network.py` (which kernels the 5.3 ms step is made of)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import agcn_amd                      # noqa: E402
from agcn_amd import synth           # noqa: E402

dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
K, mode = 2, "paper"
B, n = 32, 1024
rs = np.random.RandomState(7)
pts = rs.uniform(-1, 1, size=(B, n, 3)).astype(np.float32)
n32 = np.full(B, n, np.int32)
plan = agcn_amd.Plan(n32, 0, dev)
# a symmetric dense Laplacian per cloud (threshold graph on the first cloud's distances, reused: the values do not matter here)
d = np.sqrt(((pts[0][:, None, :] - pts[0][None, :, :]) ** 2).sum(-1))
A = (d < 0.6 * d.mean()).astype(np.float32)
np.fill_diagonal(A, 0)
deg = A.sum(1) + 1e-6
L = np.eye(n, dtype=np.float32) - (A / np.sqrt(deg)[:, None]) / np.sqrt(deg)[None, :]
L = (0.5 * (L + L.T)).astype(np.float32)          # (bit-symmetric: the layer verifies the vouch)
Lt = torch.from_numpy(np.tile(L.reshape(-1), B)).to(dev)
gbn = agcn_amd.GraphBatch(plan, torch.from_numpy(pts.reshape(-1, 3).copy()).to(dev), l0_dense=Lt, l0_ld=0, l0_symmetric=True)
net = agcn_amd.DenseConnectedGraphResLap(2, 3, B, n)
fin, convs = 3, []
for i, w in enumerate((64, 128, 128, 64)):
    lay = agcn_amd.SGC_LL_Reslap(w, fin, B, K=K, laplacian_mode=mode, save_lap=i in (0, 2), save_output=True, device=dev)
    lay.load_vars(synth.glorot_params(rs, fin, w, K, reslap=True))
    convs.append(lay)
    fin = w
be0 = agcn_amd.DenseBlockEnd(0, [64, 128], 128, "relu", max_atom=n, batch_size=B, device=dev)
be1 = agcn_amd.DenseBlockEnd(1, [128, 64], 64, "relu", max_atom=n, batch_size=B, device=dev)
dn = agcn_amd.DenseMol(256, 64, device=dev)
dn.build(dev)
with torch.no_grad():
    dn.W.mul_(0.004)
for lay in (convs[0], convs[1], be0, convs[2], convs[3], be1, dn, agcn_amd.GraphGatherMol(B, activation="tanh")):
    net.add(lay)
Gn = torch.from_numpy(rs.randn(B, 256).astype(np.float32)).to(dev)
be0.inblock_activations_dim, be0.preceding_blocks_dim = [64, 128], []
be1.inblock_activations_dim, be1.preceding_blocks_dim = [128, 64], [128]
be0.build(dev)
be1.build(dev)
params = [v for lay in convs for v in lay.vars.values()] + dn.parameters() + be0.parameters() + be1.parameters()
for it in range(steps + 2):
    for q in params:
        q.grad = None
    net(gbn).backward(Gn)
torch.cuda.synchronize()
print("ran", steps + 2, "steps")
