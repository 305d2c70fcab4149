"""Data-parallel plumbing on CPU: deterministic sharding and the flat-bucket all-reduce over a
world_size-2 gloo group.  The per-rank compute is replaced by a cheap stand-in (numpy sums over
the rank's graphs) because the HIP path cannot run here; what is tested is that shard + SUM
all-reduce reproduces the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


@pytest.fixture(scope="module")
def amd():
    import agcn_amd
    return agcn_amd


def test_shard_graphs_properties(amd):
    rs = np.random.RandomState(0)
    n = np.clip(np.rint(rs.lognormal(np.log(16), 0.5, size=2048)), 4, 132).astype(np.int32)
    cost = amd.dist.graph_cost(n, 75, 64, 2)
    for world in (1, 2, 4, 8):
        shards = amd.dist.shard_graphs(cost, world)
        assert len(shards) == world
        allidx = np.sort(np.concatenate(shards))
        np.testing.assert_array_equal(allidx, np.arange(2048))          # a partition
        assert all(len(s) == 2048 // world for s in shards)              # equal graph counts
        loads = np.array([cost[s].sum() for s in shards], dtype=np.float64)
        assert loads.max() / loads.mean() < 1.01                          # balanced work
        again = amd.dist.shard_graphs(cost, world)
        assert all(np.array_equal(a, b) for a, b in zip(shards, again))  # deterministic


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, feats, out_q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import agcn_amd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, _, w = agcn_amd.dist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    shards = agcn_amd.dist.shard_graphs(agcn_amd.dist.graph_cost(n, 75, 64, 2), world)
    mine = shards[rank]
    # stand-in "gradients": two tensors that are sums over the rank's graphs
    g1 = torch.from_numpy(np.stack([feats[i] for i in mine]).sum(0))
    g2 = torch.tensor([float(n[mine].sum())], dtype=torch.float32)
    flat, views = agcn_amd.dist.flatten_grads([g1, g2])
    agcn_amd.dist.allreduce_bucket(flat)
    if rank == 0:
        out_q.put((views[0].numpy().copy(), views[1].numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_allreduce_world2(amd):
    rs = np.random.RandomState(1)
    B = 16
    n = rs.randint(4, 60, size=B).astype(np.int32)
    feats = [rs.randn(5, 3).astype(np.float32) for _ in range(B)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, feats, q)) for r in range(2)]
    for p in procs:
        p.start()
    g1, g2 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    np.testing.assert_allclose(g1, np.stack(feats).sum(0), rtol=1e-6, atol=1e-6)   # == single process
    assert float(g2[0]) == float(n.sum())                                           # integers: exact


def _oracle_worker(rank, world, port, out_q):
    """One rank of the 8-way data-parallel run of the B = 2048 global batch (BASELINE config 4) with the ORACLE as the
    per-rank compute: LPT shard -> forward + backward -> flat bucket [dweight | dbias | dM_L | dalpha] -> SUM all-reduce."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "oracle"))
    import agcn_amd
    import agcn_oracle as orc
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    agcn_amd.dist.init_from_env("gloo")
    X, L0, p, G, ns = _dp_case(orc)
    shards = agcn_amd.dist.shard_graphs(agcn_amd.dist.graph_cost(ns, 75, 64, 2), world)
    mine = shards[rank]
    flat = _oracle_bucket(orc, [X[i] for i in mine], [L0[i] for i in mine], p, [G[i] for i in mine])
    agcn_amd.dist.allreduce_bucket(flat)
    if rank == 0:
        out_q.put((flat.numpy().copy(), [len(s) for s in shards]))
    dist.barrier()
    dist.destroy_process_group()


def _dp_case(orc):
    sb = orc.synth_batch(123, 2048)
    rs = np.random.RandomState(9)
    p = {k: v.astype(np.float64) for k, v in orc.init_params(rs, 75, 64, 2).items()}
    ns = np.asarray(sb["ns"])
    G = [np.random.RandomState(1000 + b).randn(int(n), 64) for b, n in enumerate(ns)]
    X = [x.astype(np.float64) for x in sb["X"]]
    L0 = [l.astype(np.float64) for l in sb["L0"]]
    return X, L0, p, G, ns


def _oracle_bucket(orc, X, L0, p, G):
    caches = orc.layer_forward(X, L0, p, 2, orc.PAPER, orc.SGC_LL)
    tot = orc.layer_backward(caches, G, p, 2, orc.PAPER, orc.SGC_LL)
    parts = [tot["dweight"], tot["dbias"], tot["dM_L"], np.asarray(tot["dalpha"]).reshape(-1)]
    return torch.from_numpy(np.concatenate([np.asarray(t, np.float64).reshape(-1) for t in parts]))


def test_bucket_equality_world8_full_global_batch(amd):
    """BASELINE config 4 (Tox21 batch = 2048 over 8 GPUs) with everything but the GPU: 8 gloo ranks, each with its LPT
    shard (256 graphs) of the 2048-graph global batch, the oracle as the per-rank compute, ONE SUM all-reduce of the flat
    bucket == the single-process gradients of the whole batch (float64: 1e-9 of the bucket's scale: the order of
    summation is the only difference)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import agcn_oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_oracle_worker, args=(r, 8, port, q)) for r in range(8)]
    for p_ in procs:
        p_.start()
    got, counts = q.get(timeout=600)
    for p_ in procs:
        p_.join(timeout=600)
        assert p_.exitcode == 0
    assert counts == [256] * 8
    X, L0, p, G, ns = _dp_case(orc)
    want = _oracle_bucket(orc, X, L0, p, G).numpy()
    assert got.shape == want.shape == (75 * 2 * 64 + 64 + 75 * 75 + 1,)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9 * np.abs(want).max())


def test_single_process_is_noop(amd):
    t = torch.arange(4, dtype=torch.float32)
    assert torch.equal(amd.dist.allreduce_bucket(t.clone()), t)


def _bucket_worker(rank, world, port, out_q):
    """A multi-layer model (two linear stages + a scalar) trained data-parallel with ONE flat bucket behind every .grad
    (agcn_amd.dist.GradBucket): autograd accumulates into the bucket's views, one SUM all-reduce per step."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import agcn_amd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    agcn_amd.dist.init_from_env("gloo")
    params, X, G = _bucket_case()
    n = np.full(X.shape[0], 10, np.int32)
    mine = agcn_amd.dist.shard_graphs(agcn_amd.dist.graph_cost(n, 4, 3, 2), world)[rank]
    bucket = agcn_amd.dist.GradBucket(params)
    for _ in range(2):                                  # second step: the bucket is re-zeroed, not accumulated
        bucket.zero()
        _bucket_loss(params, X[mine], G[mine]).backward()
        bucket.all_reduce()
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
    if rank == 0:
        out_q.put(bucket.flat.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def _bucket_case():
    g = torch.Generator().manual_seed(3)
    W1 = torch.randn(4, 5, generator=g, dtype=torch.float64).float().requires_grad_(True)
    W2 = torch.randn(5, 3, generator=g).requires_grad_(True)
    a = torch.ones(1, requires_grad=True)
    frozen = torch.randn(2, requires_grad=True)         # never reached by the loss: its section stays zero
    X = torch.randn(16, 4, generator=g)
    G = torch.randn(16, 3, generator=g)
    return [W1, W2, a, frozen], X, G


def _bucket_loss(params, X, G):
    W1, W2, a, _ = params
    return ((torch.relu(X @ W1) @ W2 * a) * G).sum()


@pytest.mark.parametrize("world", [2, 8])
def test_grad_bucket_multilayer_allreduce(amd, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    params, X, G = _bucket_case()
    _bucket_loss(params, X, G).backward()
    want = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    assert np.all(got[-2:] == 0) and got.size == 4 * 5 + 5 * 3 + 1 + 2
