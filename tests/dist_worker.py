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
    st.capture(extra=lambda: dist.all_reduce(st.bucket, op=dist.ReduceOp.SUM), steps=2)
    ok = True
    for _ in range(3):
        st.bucket.zero_()
        st.replay()
        torch.cuda.synchronize(dev)
        ok = ok and bool(torch.equal(st.bucket, want))
    np.savez(out, ok=int(ok), ranks=dist.get_world_size(), bucket=st.bucket.cpu().numpy(), want=want.cpu().numpy())
    dist.destroy_process_group()


def main():
    if sys.argv[1] == "capture":
        return capture_main(sys.argv[2])
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
