"""Data parallelism over the GPUs of one node: graphs are independent units, only the
parameter gradients couple them.  Each rank packs its own shard; the single exchange step is
one RCCL all-reduce (SUM) over the flat gradient bucket.  The reference has no multi-GPU code;
SUM (not mean) is right because its loss divides by the GLOBAL batch size
(models/tf_modules/multitask_classifier.py:207-208).

Consequence for callers: anything that divides by the batch size -- `MultitaskHead(..., batch_size=...)`, a hand-written
loss scale -- must be given the GLOBAL batch size (per-rank graphs x world size) on every rank; the SGC layers
themselves are built with the LOCAL graph count (it is the length of their output lists).  `global_batch_size()`
returns the number to use."""
import numpy as np
import torch
import torch.distributed as dist


def graph_cost(n, Fin, Fout, K):
    """Work model of one graph: n^2 Fin (pairwise distances, L X) + n K Fin Fout (filter GEMM)."""
    n = np.asarray(n, dtype=np.int64)
    return n * n * int(Fin) + n * int(K) * int(Fin) * int(Fout)


def shard_graphs(costs, world_size):
    """Greedy longest-processing-time binning with EQUAL graph counts per rank (weak scaling
    keeps the per-GPU batch fixed).  Deterministic; returns a list of index arrays whose
    concatenation is a permutation of range(len(costs))."""
    costs = np.asarray(costs, dtype=np.int64)
    B = costs.shape[0]
    cap = -(-B // world_size)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    count = np.zeros(world_size, dtype=np.int64)
    bins = [[] for _ in range(world_size)]
    for g in order:
        open_ranks = np.nonzero(count < cap)[0]
        r = open_ranks[np.argmin(load[open_ranks])]
        bins[r].append(int(g))
        load[r] += costs[g]
        count[r] += 1
    return [np.sort(np.asarray(b, dtype=np.int64)) for b in bins]


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  backend 'nccl'
    is RCCL on ROCm; CPU tests use 'gloo'."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def global_batch_size(local_batch_size, group=None):
    """Batch size the loss must divide by under data parallelism (SUM all-reduce): local x world size."""
    if dist.is_available() and dist.is_initialized():
        return int(local_batch_size) * dist.get_world_size(group)
    return int(local_batch_size)


def allreduce_bucket(bucket, group=None):
    """One collective per step over the flat fp32 gradient bucket (61 KB for a 75->64 layer:
    latency-bound on xGMI, so never split it)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)
    return bucket


def flatten_grads(tensors):
    """Pack a list of gradient tensors into one flat buffer and return (flat, views)."""
    flat = torch.cat([t.reshape(-1) for t in tensors])
    views, o = [], 0
    for t in tensors:
        views.append(flat[o:o + t.numel()].view_as(t))
        o += t.numel()
    return flat, views


class GradBucket(object):
    """ONE flat fp32 bucket behind the `.grad` fields of a list of parameters -- every SGC layer's [weight | bias | M_L | alpha
    (| beta)], the dense tail, the task heads (140 821 floats = 563 KB for SimpleAGCN + DenseMol + 12 heads) -- so
    that a whole-network step needs ONE collective.  Each parameter's `.grad` is a view into the bucket; autograd accumulates
    into the views in place, `zero()` (one memset) starts a step, `all_reduce()` (one RCCL all-reduce, SUM: the reference's
    loss divides by the GLOBAL batch size, multitask_classifier.py:207-208) ends it.  Works inside `engine.GraphedStep`:

        bucket = dist.GradBucket(params)
        def step():
            bucket.zero()
            loss(model(batch)).backward()
            bucket.all_reduce()            # captured into the HIP graph behind the last backward kernel (RCCL), or leave it
        gs = GraphedStep(step, dev)        # out of `step` and call bucket.all_reduce() after gs.replay() (same stream)

    Parameters that get no gradient (M_L in reference mode) keep a zero section."""

    def __init__(self, params):
        self.params = [p for p in params if p is not None and p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views, o = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("GradBucket holds fp32 parameters of one device")
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self.attach()

    def attach(self):
        """(Re)points every parameter's .grad at its section of the bucket (after something set it to None)."""
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v

    def zero(self):
        self.attach()
        self.flat.zero_()

    def all_reduce(self, group=None):
        return allreduce_bucket(self.flat, group)

    def nbytes(self):
        return self.flat.numel() * 4
