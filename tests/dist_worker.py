"""One rank of the data-parallel GPU test (tests/test_dist_gpu.py): packs its LPT shard of the global batch, runs
the real SgcLLStep forward + backward on its GPU and all-reduces the flat gradient bucket.
usage: dist_worker.py <backend: nccl|gloo> <shared_gpu: 0|1> <out.npz>   (rank / world / master from the env)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def global_batch():
    from agcn_amd import synth
    Fin, Fout, K, B = 75, 64, 2, 32
    data = synth.tox21_batch(321, B, Fin)
    p = synth.glorot_params(np.random.RandomState(5), Fin, Fout, K)
    G = [np.random.RandomState(100 + b).randn(int(nb), Fout).astype(np.float32) for b, nb in enumerate(data["n"])]
    return Fin, Fout, K, data, p, G


def run_shard(idx, dev, mode="paper"):
    """Forward + backward of the graphs `idx` of the global batch; returns the SgcLLStep (bucket = summed grads)."""
    import torch
    import agcn_amd
    Fin, Fout, K, data, p, G = global_batch()
    n = data["n"][idx]
    gb = agcn_amd.GraphBatch.from_arrays(n, np.concatenate([data["X"][i] for i in idx], 0),
                                         [data["csr"][i] for i in idx], dev)
    layer = agcn_amd.SGC_LL(Fout, Fin, len(idx), K=K, laplacian_mode=mode, device=dev)
    layer.load_vars(p)
    dY = torch.from_numpy(np.concatenate([G[i] for i in idx], 0)).to(dev)
    st = agcn_amd.SgcLLStep(layer, gb, dY=dY)
    st.step()
    return st


def run_stack_shard(idx, dev, graphed=True):
    """The 4-layer SimpleAGCN-shaped stack (75-64-128-128-64) + DenseMol(32) + GraphGatherMol on the graphs `idx` of the global
    batch, autograd forward + backward into ONE flat gradient bucket (agcn_amd.dist.GradBucket), as a HIP graph
    (agcn_amd.GraphedStep).  Returns (bucket, replay)."""
    import torch
    import agcn_amd
    from agcn_amd import synth
    Fin, _, K, data, _, _ = global_batch()
    n = data["n"][idx]
    gb = agcn_amd.GraphBatch.from_arrays(n, np.concatenate([data["X"][i] for i in idx], 0), [data["csr"][i] for i in idx], dev)
    rs = np.random.RandomState(11)
    layers, fin = [], Fin
    for w in (64, 128, 128, 64):
        lay = agcn_amd.SGC_LL(w, fin, len(idx), K=K, laplacian_mode="paper", device=dev)
        lay.load_vars(synth.glorot_params(rs, fin, w, K))
        layers.append(lay)
        fin = w
    dense = agcn_amd.DenseMol(32, fin, device=dev)
    r = np.sqrt(6.0 / (fin + 32))
    dense.load_vars(rs.uniform(-r, r, size=(fin, 32)).astype(np.float32), np.zeros(32, np.float32))
    gather = agcn_amd.GraphGatherMol(len(idx), activation="tanh")
    Gall = np.random.RandomState(12).randn(len(data["n"]), 32).astype(np.float32)      # one upstream row per GLOBAL graph
    G = torch.from_numpy(Gall[idx]).to(dev)
    params = [v for lay in layers for v in lay.vars.values()] + dense.parameters()
    bucket = agcn_amd.dist.GradBucket(params)

    def step():
        bucket.zero()
        h = gb.X
        for lay in layers:
            h, _, _ = lay({"node_features": h, "graph_batch": gb})
        h = dense({"node_features": h, "graph_batch": gb})
        gather({"node_features": h, "graph_batch": gb}).backward(G)

    if not graphed:
        step()
        return bucket, step
    gs = agcn_amd.GraphedStep(step, dev)
    gs.replay()
    return bucket, gs.replay


def stack_main(backend, shared, out):
    """Data parallelism of the multi-layer step: every rank runs the HIP-graphed stack on its LPT shard, then ONE all-reduce
    of the flat bucket over all layers' gradients behind the replay (same stream)."""
    import torch
    import torch.distributed as dist
    import agcn_amd
    rank, local, world = agcn_amd.dist.init_from_env(backend)
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    Fin, Fout, K, data, p, G = global_batch()
    shards = agcn_amd.dist.shard_graphs(agcn_amd.dist.graph_cost(data["n"], Fin, Fout, K), world)
    bucket, replay = run_stack_shard(shards[rank], dev)
    for _ in range(2):                      # replay -> all-reduce, twice: the bucket is re-zeroed inside the graph
        replay()
        bucket.all_reduce()
    torch.cuda.synchronize(dev)
    if rank == 0:
        np.savez(out, bucket=bucket.flat.cpu().numpy(), ranks=dist.get_world_size(), nbytes=bucket.nbytes())
    dist.barrier()
    dist.destroy_process_group()


def capture_main(out):
    """bench.py --graph-allreduce's code path on ONE GPU: a one-rank RCCL group (a real ncclAllReduce launch, identity in
    value), the all-reduce of the flat bucket captured into the step's HIP graph BEHIND the step, replays compared with the
    eager bucket bit for bit.  Proves on this stack that RCCL initialises with device_id, that its collective is
    capturable, and that SgcLLStep.capture(extra=...) orders it behind the parameter-gradient kernel."""
    import torch
    import torch.distributed as dist
    import agcn_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    Fin, Fout, K, data, p, G = global_batch()
    st = run_shard(np.arange(len(data["n"])), dev)
    torch.cuda.synchronize(dev)
    want = st.bucket.clone()
    warm = torch.ones(4, device=dev)
    dist.all_reduce(warm)                                  # communicator set-up outside the capture
    torch.cuda.synchronize(dev)
    # TEN steps -- kernels and collective -- per graph launch, as bench.py's default for N > 1 captures them; ten replays
    st.capture(extra=lambda: dist.all_reduce(st.bucket, op=dist.ReduceOp.SUM), steps=10)
    ok = st.steps_per_replay == 10
    for _ in range(10):
        st.bucket.zero_()
        st.replay()
        torch.cuda.synchronize(dev)
        ok = ok and bool(torch.equal(st.bucket, want))
    np.savez(out, ok=int(ok), ranks=dist.get_world_size(), bucket=st.bucket.cpu().numpy(), want=want.cpu().numpy())
    dist.destroy_process_group()


def main():
    if sys.argv[1] == "capture":
        return capture_main(sys.argv[2])
    if sys.argv[1] == "stack":
        return stack_main(sys.argv[2], sys.argv[3] == "1", sys.argv[4])
    backend, shared, out = sys.argv[1], sys.argv[2] == "1", sys.argv[3]
    import torch
    import torch.distributed as dist
    import agcn_amd
    rank, local, world = agcn_amd.dist.init_from_env(backend)
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    Fin, Fout, K, data, p, G = global_batch()
    shards = agcn_amd.dist.shard_graphs(agcn_amd.dist.graph_cost(data["n"], Fin, Fout, K), world)
    st = run_shard(shards[rank], dev)
    agcn_amd.dist.allreduce_bucket(st.bucket)
    torch.cuda.synchronize(dev)
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    if rank == 0:
        np.savez(out, bucket=st.bucket.cpu().numpy(), ranks=int(round(float(ones.item()))))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
