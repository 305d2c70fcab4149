#!/usr/bin/env python3
"""The HIV-shaped autograd layer step of bench.py's `secondary` (256 molecules within the specialised kernels' reach, SGC_LL 75 -> 64,
one HIP graph) with fp32 and with bf16 features, A/B/A/B: us per step of each, four rounds (environment: AGCN_BFPIPE)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import agcn_amd                      # noqa: E402
from agcn_amd import synth           # noqa: E402

dev = torch.device("cuda", 0)
for seed in range(40, 400):
    nsa = synth.node_counts(np.random.RandomState(seed), 256, mu=np.log(23.0), hi=222)
    if nsa.max() <= 128:
        break
data = synth.tox21_batch(seed, 256, 75, mu=np.log(23.0), hi=222)
gb = agcn_amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
steps = {}
for dt, nm in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
    lay = agcn_amd.SGC_LL(64, 75, 256, K=2, laplacian_mode="paper", device=dev)
    lay.load_vars(synth.glorot_params(np.random.RandomState(123), 75, 64, 2))
    X = gb.X.to(dt).clone().requires_grad_(True)
    G = torch.from_numpy(np.random.RandomState(7).randn(gb.plan.rows, 64).astype(np.float32)).to(dev).to(dt)
    ps = list(lay.vars.values()) + [X]

    def step(lay=lay, X=X, G=G, ps=ps):
        for q in ps:
            q.grad = None
        nodes, _, _ = lay({"node_features": X, "graph_batch": gb})
        nodes.data.backward(G)
    steps[nm] = agcn_amd.GraphedStep(step, dev)
res = {k: [] for k in steps}
for rnd in range(4):
    for nm, g in steps.items():
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        res[nm].append((time.perf_counter() - t0) / 200 * 1e6)
print("AGCN_BFPIPE=%s us per step: fp32 %s  bf16 %s  ratio of medians %.4f"
      % (os.environ.get("AGCN_BFPIPE", "default"), np.round(res["fp32"], 2).tolist(), np.round(res["bf16"], 2).tolist(),
         np.median(res["bf16"]) / np.median(res["fp32"])))
