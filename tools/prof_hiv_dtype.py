#!/usr/bin/env python3
"""The HIV-shaped autograd layer step of bench.py's `secondary` (256 molecules, all within the specialised kernels' reach, SGC_LL
75 -> 64) with fp32 or bf16 node features, EAGER, for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_hiv_dtype.py f32|bf16`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import agcn_amd                      # noqa: E402
from agcn_amd import synth           # noqa: E402

dev = torch.device("cuda", 0)
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
for seed in range(40, 400):
    nsa = synth.node_counts(np.random.RandomState(seed), 256, mu=np.log(23.0), hi=222)
    if nsa.max() <= 128:
        break
data = synth.tox21_batch(seed, 256, 75, mu=np.log(23.0), hi=222)
gb = agcn_amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
lay = agcn_amd.SGC_LL(64, 75, 256, K=2, laplacian_mode="paper", device=dev)
lay.load_vars(synth.glorot_params(np.random.RandomState(123), 75, 64, 2))
X = gb.X.to(dt).clone().requires_grad_(True)
G = torch.from_numpy(np.random.RandomState(7).randn(gb.plan.rows, 64).astype(np.float32)).to(dev).to(dt)
ps = list(lay.vars.values()) + [X]
for it in range(200):
    for q in ps:
        q.grad = None
    nodes, _, _ = lay({"node_features": X, "graph_batch": gb})
    nodes.data.backward(G)
torch.cuda.synchronize()
print("ran 200 steps", dt)
if os.environ.get("AGCN_HIP_LIB", "").endswith("_stamps.so"):      # phase stamps of the LAST step (tools/stamps.py's library)
    import ctypes as C
    lib = agcn_amd._ffi.lib()
    lib.agcn_debug_read_stamps.restype = C.c_int
    lib.agcn_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    buf = np.zeros(8 * 1024 * 16, dtype=np.uint64)
    lib.agcn_debug_read_stamps(buf.ctypes.data, buf.size, 1)
    for q in ps:
        q.grad = None
    nodes, _, _ = lay({"node_features": X, "graph_batch": gb})
    nodes.data.backward(G)
    torch.cuda.synchronize()
    lib.agcn_debug_read_stamps(buf.ctypes.data, buf.size, 1)
    s = buf.reshape(8, 1024, 16).astype(np.int64)
    for k, nm in ((1, "forward plain"), (0, "forward members"), (4, "backward plain"), (5, "backward members")):
        blk = s[k]
        used = blk[:, 0] > 0
        if not used.any():
            continue
        t0 = blk[used, 0].min()
        cols = sorted([c for c in range(16) if (blk[used, c] > 0).any()], key=lambda c: np.median(blk[used, c]))
        print(nm, "stamps", cols, "median us", np.round(np.median((blk[used][:, cols] - t0) / 100.0, 0), 2).tolist())
