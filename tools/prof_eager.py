"""Diagnostic: host-side cost of one eager autograd layer step (forward + backward) on a fixed batch."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agcn_amd as amd  # noqa: E402
from agcn_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
data = synth.tox21_batch(123, 256)
gb = amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
layer = amd.SGC_LL(64, 75, 256, K=2, laplacian_mode="paper", device=dev)
layer.load_vars(synth.glorot_params(np.random.RandomState(123), 75, 64, 2))
G = torch.ones(gb.plan.rows, 64, device=dev)
Xt = gb.X.requires_grad_(True)


def step():
    for q in layer.vars.values():
        q.grad = None
    Xt.grad = None
    nodes, _, _ = layer({"node_features": Xt, "graph_batch": gb})
    nodes.data.backward(G)


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("eager step: host %.1f us, wall %.1f us" % (host / 300 * 1e6, tot / 300 * 1e6))
cProfile.run("for _ in range(300): step()", "/tmp/eager.prof")
pstats.Stats("/tmp/eager.prof").sort_stats("tottime").print_stats(22)
