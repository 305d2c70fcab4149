import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import agcn_amd as amd
from agcn_amd import synth, _ffi
dev = torch.device("cuda", 0)
G = 41127
data = synth.tox21_batch(321, G, mu=np.log(23.0), hi=222)
store = amd.GraphStore.from_arrays(data["n"], data["X"], data["csr"])
rs = store.to_device(dev, 256)
layer = amd.SGC_LL(64, 75, 256, K=2, laplacian_mode="paper", device=dev)
layer.load_vars(synth.glorot_params(np.random.RandomState(123), 75, 64, 2))
ep = amd.SgcLLEpochStep(layer, rows_cap=rs.rows_cap, tiles_cap=rs.tiles_cap, batch_cap=256, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for gb, _, _, _ in rs.iterbatches(256, deterministic=True):
        ep.step(gb)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host issue %.1f us/batch, total %.1f us/batch" % ((t1 - t0) / 161 * 1e6, (t2 - t0) / 161 * 1e6))
# kernel mix
lib = _ffi.lib()
lib.agcn_profile_enable(1)
mx = []
for gb, _, _, _ in rs.iterbatches(256, deterministic=True):
    ep.step(gb); mx.append(gb.plan.max_n)
torch.cuda.synchronize()
prof = _ffi.profile_collect()
lib.agcn_profile_enable(0)
print({k: (round(ms / max(c, 1) * 1e3, 1), c) for k, (ms, c) in prof.items() if c})
print("max_n per batch: min %d median %d max %d; batches with max_n<=96: %d of %d" % (min(mx), np.median(mx), max(mx), sum(m <= 96 for m in mx), len(mx)))
# host-only cost split
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for gb, _, _, _ in rs.iterbatches(256, deterministic=True):
    ep.step(gb)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
