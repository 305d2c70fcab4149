"""Debug: the cooperative stress (two lanes + a CU hog) with the trailing parameter-gradient workgroups."""
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
data = synth.tox21_batch(77, 1536, mu=np.log(30.0), hi=128)
store = amd.GraphStore.from_arrays(data["n"], data["X"], data["csr"])
p = synth.glorot_params(np.random.RandomState(3), 75, 64, 2)
layer = make_layer(amd, orc.SGC_LL, 64, 75, 256, 2, orc.PAPER, p, DEV)
sels = store._selections(256, False, np.random.RandomState(1))
r1 = store.to_device(DEV, 256)
ep = amd.SgcLLEpochStep(layer, r1.rows_cap, r1.tiles_cap, 256, device=DEV)
want = []
for sel in sels:
    gb = r1.stage(sel)
    print("path", amd._ffi.dispatch_info(layer, gb))
    ep.step(gb)
    torch.cuda.synchronize()
    want.append((ep.Y.clone(), ep.dX.clone(), ep.bucket.clone()))
print("single status", ep.check())
r2 = store.to_device(DEV, 256, lanes=2)
lanes = amd.SgcLLEpochLanes(layer, r2, lanes=2)
for hog in (False, True, True):
    hog_stream = torch.cuda.Stream(DEV)
    A = torch.randn(4096, 4096, device=DEV)
    got = []
    if hog:
        with torch.cuda.stream(hog_stream):
            for _ in range(60):
                A = torch.mm(A, A) * 1e-4
    for j, sel in enumerate(sels):
        st = lanes.step(sel)
        if j >= 1:
            k = (j - 1) % 2
            lanes.streams[k].synchronize()
            prev = lanes.steps[k]
            got.append((prev.Y.clone(), prev.dX.clone(), prev.bucket.clone()))
    lanes.synchronize(check=False)
    got.append((st.Y.clone(), st.dX.clone(), st.bucket.clone()))
    hog_stream.synchronize()
    print("hog", hog)
    for j in range(len(sels)):
        for k, nm in enumerate(("Y", "dX", "bucket")):
            a, b = want[j][k], got[j][k]
            if not torch.equal(a, b):
                d = (a - b).abs()
                dn = torch.nan_to_num(d)
                bad = (dn > 0) | torch.isnan(d)
                idx = torch.nonzero(bad)[:6].reshape(-1).tolist()
                print("  batch %d %s differs: max %.3e, %d of %d elements, idx %s, nan got %d" % (j, nm, float(dn.max()), int(bad.sum()), d.numel(), idx, int(torch.isnan(b).sum())))
    for stp in lanes.steps:
        try:
            print("  lane status", stp.check())
        except Exception as e:
            print("  lane status error", str(e)[:100])
            w = stp._reserve[8 * 4080:8 * 4080 + 4].view(torch.int32); w.zero_()
print("done")
