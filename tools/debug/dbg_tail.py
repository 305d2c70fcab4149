"""Debug: trailing parameter-gradient workgroups (spec_tail.h): determinism and lanes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import agcn_amd as amd
import agcn_oracle as orc
from helpers import make_layer
from agcn_amd import synth
DEV = "cuda:0"
print("AGCN_SPEC_TAIL", os.environ.get("AGCN_SPEC_TAIL"))
data = synth.tox21_batch(55, 1100, mu=np.log(23.0), hi=222)
store = amd.GraphStore.from_arrays(data["n"], data["X"], data["csr"])
p = orc.init_params(np.random.RandomState(4), 75, 64, 2)
layer = make_layer(amd, orc.SGC_LL, 64, 75, 256, 2, orc.PAPER, p, DEV)
sels = store._selections(256, False, np.random.RandomState(1))
r1 = store.to_device(DEV, 256)
ep = amd.SgcLLEpochStep(layer, r1.rows_cap, r1.tiles_cap, 256, device=DEV)
runs = []
for rep in range(3):
    want = []
    for sel in sels:
        gb = r1.stage(sel)
        if rep == 0:
            print("batch path", amd._ffi.dispatch_info(layer, gb))
        ep.step(gb)
        torch.cuda.synchronize()
        want.append((ep.Y.clone(), ep.dX.clone(), ep.bucket.clone()))
    runs.append(want)
for rep in (1, 2):
    for j in range(len(sels)):
        for k, nm in enumerate(("Y", "dX", "bucket")):
            a, b = runs[0][j][k], runs[rep][j][k]
            if not torch.equal(a, b):
                d = (a - b).abs()
                print("single lane rep %d batch %d %s differs: max %.3e at %d of %d, nan %d" % (rep, j, nm, float(d.max()), int((d > 0).sum()), d.numel(), int(torch.isnan(b).sum())))
print("status", ep.check())
r2 = store.to_device(DEV, 256, lanes=2)
lanes = amd.SgcLLEpochLanes(layer, r2, lanes=2)
got = []
for j, sel in enumerate(sels):
    st = lanes.step(sel)
    if j >= 1:
        k = (j - 1) % 2
        lanes.streams[k].synchronize()
        prev = lanes.steps[k]
        got.append((prev.Y.clone(), prev.dX.clone(), prev.bucket.clone()))
lanes.synchronize(check=False)
got.append((st.Y.clone(), st.dX.clone(), st.bucket.clone()))
for j in range(len(sels)):
    for k, nm in enumerate(("Y", "dX", "bucket")):
        a, b = runs[0][j][k], got[j][k]
        if not torch.equal(a, b):
            d = (a - b).abs()
            idx = torch.nonzero(d > 0).reshape(-1)[:6].tolist()
            print("lanes batch %d %s differs: max %.3e, %d of %d elements, first idx %s, nan %d" % (j, nm, float(d.max()), int((d > 0).sum()), d.numel(), idx, int(torch.isnan(b).sum())))
for stp in lanes.steps:
    try:
        print("lane status", stp.check())
    except Exception as e:
        print("lane status error", e)
print("done")
