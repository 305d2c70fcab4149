"""Diagnostic: where a single-lane epoch over the HBM-resident store spends its time (host vs GPU, staging vs step)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agcn_amd as amd
from agcn_amd import synth

dev = torch.device("cuda", 0)
G = int(os.environ.get("G", "16384"))
data = synth.tox21_batch(321, G, mu=np.log(23.0), hi=222)
store = amd.GraphStore.from_arrays(data["n"], data["X"], data["csr"])
layer = amd.SGC_LL(64, 75, 256, K=2, laplacian_mode="paper", device=dev)
layer.load_vars(synth.glorot_params(np.random.RandomState(123), 75, 64, 2))
sels = store._selections(256, True, None)
rs = store.to_device(dev, 256, lanes=2)
ep = amd.SgcLLEpochStep(layer, rs.rows_cap, rs.tiles_cap, 256, device=dev)
nb = len(sels)

def timeit(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(dev); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(dev)
        best = min(best, time.perf_counter() - t0)
    return best

def plain():
    for sel in sels:
        ep.step(rs.stage(sel))
def stage_only():
    for sel in sels:
        rs.stage(sel)
gbs = None
def step_only():
    gb = rs.stage(sels[3])
    for _ in sels:
        ep.step(gb)
pipe = amd.SgcLLEpochPipeline(layer, rs)
def piped():
    for _ in pipe.run(sels):
        pass
    pipe.synchronize()
paths = {}
for sel in sels:
    gb = rs.stage(sel)
    paths[amd._ffi.dispatch_info(layer, gb)[0]] = paths.get(amd._ffi.dispatch_info(layer, gb)[0], 0) + 1
torch.cuda.synchronize(dev)
print("batches", nb, "paths", paths)
for name, fn in (("stage+step (plain)", plain), ("stage only", stage_only), ("step only (one batch)", step_only), ("pipelined", piped)):
    t = timeit(fn)
    print("%-24s %7.1f us per batch  %.2f M graphs/s" % (name, 1e6 * t / nb, G / t / 1e6))
# host-only cost: enqueue without waiting (the queue absorbs a few batches), measured on a short run
t0 = time.perf_counter()
for sel in sels[:8]:
    ep.step(rs.stage(sel))
t_host = (time.perf_counter() - t0) / 8
torch.cuda.synchronize(dev)
print("host enqueue per batch ~%.1f us" % (1e6 * t_host))
