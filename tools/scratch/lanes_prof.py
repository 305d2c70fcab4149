import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import agcn_amd as amd
from agcn_amd import synth
dev = torch.device("cuda", 0)
G = 41127
data = synth.tox21_batch(321, G, mu=np.log(23.0), hi=222)
store = amd.GraphStore.from_arrays(data["n"], data["X"], data["csr"])
layer = amd.SGC_LL(64, 75, 256, K=2, laplacian_mode="paper", device=dev)
layer.load_vars(synth.glorot_params(np.random.RandomState(123), 75, 64, 2))
sels = store._selections(256, True, None)
for lanes in (2, 4):
    rl = store.to_device(dev, 256, lanes=lanes, slots=4)
    epl = amd.SgcLLEpochLanes(layer, rl, lanes=lanes)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for sel in sels: epl.step(sel)
        t1 = time.perf_counter(); epl.synchronize(); t2 = time.perf_counter()
    print("lanes %d: host %.1f us/batch total %.1f us/batch" % (lanes, (t1 - t0) / len(sels) * 1e6, (t2 - t0) / len(sels) * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for sel in sels: epl.step(sel)
pr.disable(); epl.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
