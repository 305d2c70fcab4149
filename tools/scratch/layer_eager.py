"""Diagnostic: eager forward + backward steps of one SGC_LL layer shape on the Tox21-shaped batch (for rocprofv3
--kernel-trace: run under `timeout`).  usage: layer_eager.py FIN FOUT [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import agcn_amd  # noqa: E402
from agcn_amd import synth  # noqa: E402

Fin, Fout = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
B = 256
data = synth.tox21_batch(123, B)
X = np.concatenate(data["X"], 0)
if Fin != 75:
    X = np.maximum(np.random.RandomState(5).randn(X.shape[0], Fin), 0).astype(np.float32)
gb = agcn_amd.GraphBatch.from_arrays(data["n"], X, data["csr"], "cuda:0")
layer = agcn_amd.SGC_LL(Fout, Fin, B, K=2, laplacian_mode="paper", device="cuda:0")
layer.load_vars(synth.glorot_params(np.random.RandomState(123), Fin, Fout, 2))
st = agcn_amd.SgcLLStep(layer, gb, dY=torch.randn(gb.plan.rows, Fout, device="cuda:0"))
for _ in range(steps):
    st.step()
torch.cuda.synchronize()
print("done")
