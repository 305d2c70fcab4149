"""Diagnostic: cProfile of the host side of an ingest + layer epoch."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agcn_amd as amd  # noqa: E402
from agcn_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
data = synth.tox21_batch(321, 4096, mu=np.log(23.0), hi=222)
store = amd.GraphStore.from_arrays(data["n"], data["X"], data["csr"])
layer = amd.SGC_LL(64, 75, 256, K=2, laplacian_mode="paper", device=dev)
layer.load_vars(synth.glorot_params(np.random.RandomState(123), 75, 64, 2))


def epoch(pf, with_layer=True):
    for gb, _, _, _ in store.iterbatches(256, dev, deterministic=True, prefetch=pf):
        if with_layer:
            for q in layer.vars.values():
                q.grad = None
            Xt = gb.X.requires_grad_(True)
            nodes, _, _ = layer({"node_features": Xt, "graph_batch": gb})
            nodes.data.backward(torch.ones_like(nodes.data))
    torch.cuda.synchronize(dev)


epoch(0)
epoch(0)
cProfile.run("epoch(0)", "/tmp/ep.prof")
pstats.Stats("/tmp/ep.prof").sort_stats("cumtime").print_stats(28)
