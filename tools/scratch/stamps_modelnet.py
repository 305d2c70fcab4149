import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["AGCN_HIP_LIB"] = os.path.join(ROOT, "adaptive-graph-convolutional-network_amd", "lib", "libagcn_hip_stamps.so")
import numpy as np, torch
import agcn_amd
from agcn_amd import synth
from bench_shapes import modelnet_batch
Fin, Fout = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
gb, n = modelnet_batch(agcn_amd, 32, 1024, Fin, dev)
layer = agcn_amd.SGC_LL(Fout, Fin, len(n), K=2, laplacian_mode="paper", device=dev)
layer.load_vars(synth.glorot_params(np.random.RandomState(123), Fin, Fout, 2))
st = agcn_amd.SgcLLStep(layer, gb, dY=torch.randn(gb.plan.rows, Fout, device=dev))
lib = agcn_amd._ffi.lib()
lib.agcn_debug_read_stamps.restype = C.c_int
lib.agcn_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
for _ in range(3): st.step()
buf = np.zeros(8 * 1024 * 16, dtype=np.uint64)
lib.agcn_debug_read_stamps(buf.ctypes.data, buf.size, 1)
st.step()
lib.agcn_debug_read_stamps(buf.ctypes.data, buf.size, 1)
s = buf.reshape(8, 1024, 16).astype(np.int64)
for k in (6, 7):
    blk = s[k]; used = blk[:, 0] > 0
    if not used.any(): continue
    cols = [c for c in range(16) if (blk[used, c] > 0).any()]
    rel = (blk[used][:, cols] - blk[used][:, [cols[0]]]) / 100.0
    print("slot", k, "blocks", used.sum(), "cols", cols)
    print("  median us since stamp0:", np.round(np.median(rel, 0), 2))
