import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import agcn_amd as amd
from agcn_amd import synth
dev = torch.device("cuda", 0)
G = 20000
data = synth.tox21_batch(321, G, mu=np.log(23.0), hi=222)
store = amd.GraphStore.from_arrays(data["n"], data["X"], data["csr"])
layer = amd.SGC_LL(64, 75, 256, K=2, laplacian_mode="paper", device=dev)
layer.load_vars(synth.glorot_params(np.random.RandomState(123), 75, 64, 2))
sels = store._selections(256, True, None)
junk = []
for trial in range(6):
    junk.append([torch.cuda.Stream(dev) for _ in range(trial)])      # shifts the pool's round-robin
    for lanes in (2, 3):
        rl = store.to_device(dev, 256, lanes=lanes)
        epl = amd.SgcLLEpochLanes(layer, rl, lanes=lanes)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for sel in sels: epl.step(sel)
            epl.synchronize(); dt = time.perf_counter() - t0
        print("trial %d lanes %d: %.2f M graphs/s" % (trial, lanes, G / dt / 1e6), flush=True)
        del rl, epl
