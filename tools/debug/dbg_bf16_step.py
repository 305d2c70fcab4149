"""Debug: the HIV-shaped autograd layer step as a HIP graph, fp32 vs bf16 features (bench.py secondary block)."""
import faulthandler, os, sys, time
faulthandler.enable()
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import agcn_amd
from agcn_amd import synth
dev = torch.device("cuda", 0)
K = 2
def P(*a):
    print(*a); sys.stdout.flush()
for seed in range(40, 400):
    nsa = synth.node_counts(np.random.RandomState(seed), 256, mu=np.log(23.0), hi=222)
    if nsa.max() <= 128:
        break
P("seed", seed, "max", nsa.max())
data = synth.tox21_batch(seed, 256, 75, mu=np.log(23.0), hi=222)
P("data n max", data["n"].max(), (data["n"] == nsa).all())
gbh = agcn_amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
for dt, nm in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
    lay = agcn_amd.SGC_LL(64, 75, 256, K=K, laplacian_mode="paper", device=dev)
    lay.load_vars(synth.glorot_params(np.random.RandomState(123), 75, 64, K))
    P(nm, "dispatch", agcn_amd._ffi.dispatch_info(lay, gbh))
    Xh = gbh.X.to(dt).clone().requires_grad_(True)
    Gh = torch.from_numpy(np.random.RandomState(7).randn(gbh.plan.rows, 64).astype(np.float32)).to(dev).to(dt)
    hp = list(lay.vars.values()) + [Xh]
    def hstep():
        for q in hp:
            q.grad = None
        nodes, _, _ = lay({"node_features": Xh, "graph_batch": gbh})
        nodes.data.backward(Gh)
    hstep(); torch.cuda.synchronize(); P(nm, "eager ok", float(Xh.grad.float().abs().sum()))
    gh = agcn_amd.GraphedStep(hstep, dev)
    P(nm, "captured")
    for _ in range(5): gh.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): gh.replay()
    torch.cuda.synchronize()
    P(nm, "us per step", 1e6 * (time.perf_counter() - t0) / 200)
    del gh
P("done")
