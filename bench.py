#!/usr/bin/env python
"""Headline benchmark: AGCN layer forward+backward graphs/sec on Tox21-shaped batches
(batch 256 per GPU, SGC_LL 75 -> 64, K = 2, fp32), BASELINE.json's metric.

A "step" is one pass of the hot path over one batch: SGC_LL forward (similarity tile, learned
Laplacian, Chebyshev, filter GEMM, all outputs incl. res_L / res_W written) + backward (dX and all
parameter gradients) and, for N > 1, the RCCL all-reduce of the flat gradient bucket.
Inputs are resident in HBM in the packed layout before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode paper|reference] [--no-graph]
N > 1 is launched with torch.distributed.run, one rank per GPU.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.3 TB/s achievable)


def algorithmic_bytes(n, nnz, Fin, Fout, K):
    """minimal HBM bytes of one layer call (CSR Laplacian)."""
    n = n.astype(np.int64)
    N1, N2, B = int(n.sum()), int((n * n).sum()), len(n)
    P = Fin * Fin + K * Fin * Fout + Fout + 1
    L_bytes = 8 * nnz + 4 * (N1 + B)
    fwd = 4 * N1 * Fin + L_bytes + 4 * N1 * Fout + 8 * N2 + 4 * P
    bwd = 4 * N1 * Fout + 8 * N1 * Fin + L_bytes + 8 * P
    return N1, N2, fwd, bwd


def kernel_bytes(N1, N2, nnz, B, Fin, Fout, K, paper):
    """Compulsory HBM bytes of each kernel of the path (its own inputs + outputs, each counted
    once), keyed by the profiling names of agcn_profile_kernel_name()."""
    Lb = 8 * nnz + 4 * (N1 + B)
    W, M = 4 * K * Fin * Fout, 4 * Fin * Fin
    kb = {
        "rows_gemm_kernel<PROJECT>": 8 * N1 * Fin + M,
        # Z + X in, res_W + res_L + X_1..X_{K-1} + s out
        "sgc_forward_kernel": 8 * N1 * Fin + Lb + 8 * N2 + 4 * (K - 1) * N1 * Fin + 4 * N1,
        "rows_gemm_kernel<FILTER>": 4 * K * N1 * Fin + W + 4 * Fout + 4 * N1 * Fout,
        "rows_gemm_kernel<FILTER_BWD>": 8 * N1 * Fout + W + 4 * K * N1 * Fin + 4 * N1 * Fout,
        # g_1.. in, dX read+write, L (res_L + CSR) in; paper adds X, Z, res_W, s in and dZ out
        "sgc_backward_kernel": 4 * (K - 1) * N1 * Fin + 8 * N1 * Fin + 4 * N2 + Lb
                               + ((12 * N1 * Fin + 4 * N2 + 4 * N1) if paper else 0),
        "rows_gemm_kernel<PROJECT_BWD>": 12 * N1 * Fin + M,
        "param_grad_partial_kernel": 4 * K * N1 * Fin + 4 * N1 * Fout + (8 * N1 * Fin if paper else 0),
        "param_grad_reduce_kernel": W + 4 * Fout + (M if paper else 0),
        "param_grad_onepass_kernel": 4 * K * N1 * Fin + 4 * N1 * Fout + (8 * N1 * Fin if paper else 0)
                                     + W + 4 * Fout + (M if paper else 0),
        # fused per-graph path (SGC_LL, K = 2): the layer-level algorithmic traffic of SURVEY 8(d) plus the
        # forward->backward reserve (Z, X_1, s, and r tiles in paper mode) and the Gp / dZ hand-off to the
        # parameter-gradient kernels
        "fused_forward_kernel": 4 * N1 * Fin + Lb + 4 * N1 * Fout + 8 * N2 + M + W + 4 * Fout
                                + 8 * N1 * Fin + 4 * N1 + (4 * N2 if paper else 0),
        "fused_backward_kernel": 8 * N1 * Fout + W + 4 * N2 + Lb + 4 * N1 * Fin + 4 * N1 * Fout
                                 + ((12 * N1 * Fin + 8 * N2 + 4 * N1 + M) if paper else 0),
        # slab-parallel path (SGC_LL, K = 2)
        "slab_sim_kernel": 4 * N1 * Fin + 4 * N2 * (2 if paper else 1) + 4 * N1,
        "slab_lap_kernel": (4 * N2 if paper else 0) + 4 * N2 + Lb + 8 * N1 * Fin + 4 * N1,
        "slab_bwd1_kernel": 4 * N2 + Lb + 12 * N1 * Fin + ((4 * N1 * Fin + 8 * N2 + 4 * N1) if paper else 0),
        "slab_bwd2_kernel": 12 * N2 + 8 * N1 * Fin + 8 * N1,
    }
    return kb


def cpu_baseline(sample, p, G, K, mode_name, budget_s=12.0):
    """The oracle (numpy port of the reference algorithm, vectorised per graph) timed on this
    box's host cores, single thread, same inputs, fwd+bwd.  Reported baseline, not a target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import agcn_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                    # pragma: no cover
        threadpool_limits = None
    mode = orc.PAPER if mode_name == "paper" else orc.REFERENCE
    X = sample["X"]
    L0 = []
    for (ip, ix, dat), nb in zip(sample["csr"], sample["n"]):
        L = np.zeros((nb, nb), np.float32)
        for i in range(nb):
            L[i, ix[ip[i]:ip[i + 1]]] = dat[ip[i]:ip[i + 1]]
        L0.append(L)

    def one():
        caches = orc.layer_forward(X, L0, p, K, mode, orc.SGC_LL)
        orc.layer_backward(caches, G, p, K, mode, orc.SGC_LL)

    import contextlib
    with (threadpool_limits(limits=1) if threadpool_limits else contextlib.nullcontext()):
        one()
        reps, t0 = 0, time.perf_counter()
        while True:
            one()
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or reps >= 20:
                break
    gps = len(X) * reps / dt
    # reference-faithful variant (O(n^2) Python loop with one np.linalg.norm per pair,
    # graphconv.py:168-178), forward only, small sample
    m = min(32, len(X))
    t0 = time.perf_counter()
    orc.layer_forward(X[:m], L0[:m], p, K, orc.REFERENCE, orc.SGC_LL, loop_similarity=True)
    loop_gps = m / (time.perf_counter() - t0)
    return {"value": round(gps, 1), "unit": "graphs/s", "cores": 1, "kind": "port",
            "sample": "same %d-graph batch, %d fwd+bwd passes of oracle/agcn_oracle.py (%s mode, numpy, 1 thread)"
                      % (len(X), reps, mode_name),
            "reference_loop_fwd_graphs_per_s": round(loop_gps, 1), "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="graphs per GPU")
    ap.add_argument("--mode", default="paper", choices=["paper", "reference"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="single-GPU diagnostic: build the global batch of an N-GPU run and time shard --shard of it")
    ap.add_argument("--shard", type=int, default=0)
    ap.add_argument("--skip-other-mode", action="store_true",
                    help="profiling runs: do not also time the other laplacian_mode")
    args = ap.parse_args()

    import agcn_amd
    from agcn_amd import synth

    # AGCN_BENCH_SELFTEST=1 (development only): several ranks share cuda:0 over gloo, to exercise the multi-process
    # control flow on a one-GPU box; the driver's runs use one GPU per rank over RCCL
    selftest = os.environ.get("AGCN_BENCH_SELFTEST") == "1"
    rank, local, world = agcn_amd.dist.init_from_env("gloo" if selftest else "nccl")
    if selftest:
        local = 0
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    agcn_amd._ffi.lib()

    Fin, Fout, K = 75, 64, 2
    B_global = args.batch * world
    emu = args.shard_of if (world == 1 and args.shard_of > 1) else 0
    data = synth.tox21_batch(123, args.batch * (emu or world), Fin)
    shards = agcn_amd.dist.shard_graphs(agcn_amd.dist.graph_cost(data["n"], Fin, Fout, K), emu or world)
    mine = shards[args.shard if emu else rank]
    n = data["n"][mine]
    X = [data["X"][i] for i in mine]
    csr = [data["csr"][i] for i in mine]
    p = synth.glorot_params(np.random.RandomState(123), Fin, Fout, K)
    gb = agcn_amd.GraphBatch.from_arrays(n, np.concatenate(X, 0), csr, dev)
    Gnp = np.random.RandomState(7 + rank).randn(gb.plan.rows, Fout).astype(np.float32)
    dY = torch.from_numpy(Gnp).to(dev)

    def make(mode):
        layer = agcn_amd.SGC_LL(Fout, Fin, len(n), K=K, laplacian_mode=mode, device=dev)
        layer.load_vars(p)
        return agcn_amd.SgcLLStep(layer, gb, dY=dY)

    st = make(args.mode)
    use_graph = not args.no_graph
    if use_graph:
        st.capture()

    def step():
        if use_graph:
            st.replay()
        else:
            st.step()
        if world > 1:
            agcn_amd.dist.allreduce_bucket(st.bucket)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed = timed(step, args.steps, args.warmup)
    value = B_global * args.steps / elapsed

    # ---- per-kernel durations: HIP events recorded by the library on the launch stream around
    # every kernel (agcn_profile_*), over the same number of (eager) steps ---------------------------
    lib = agcn_amd._ffi.lib()
    for _ in range(5):
        st.step()
    torch.cuda.synchronize(dev)
    lib.agcn_profile_enable(1)
    for _ in range(args.steps):
        st.step()
    prof = agcn_amd._ffi.profile_collect()
    lib.agcn_profile_enable(0)
    avg_ms = {k: (ms / c) for k, (ms, c) in prof.items() if c > 0}

    nnz = int(sum(len(c[1]) for c in csr))
    N1, N2, fwd_b, bwd_b = algorithmic_bytes(n, nnz, Fin, Fout, K)
    kb = kernel_bytes(N1, N2, nnz, len(n), Fin, Fout, K, args.mode == "paper")
    dom = max(avg_ms, key=avg_ms.get)
    achieved = kb.get(dom, 0) / (avg_ms[dom] * 1e-3) / 1e9
    step_ms = 1e3 * elapsed / args.steps
    # HBM traffic of the dominant kernel: TCC FETCH_SIZE / WRITE_SIZE collected with rocprofv3 --pmc
    # (tools/traffic.sh, corrected as MI355X_MICROARCH.md prescribes) and committed under profiles/
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.mode)
    if os.path.exists(tpath) and world == 1 and args.batch == 256:
        try:
            tj = json.load(open(tpath))
            for kname, v in tj.items():
                if kname.split("::")[-1].split("<")[0] == dom.split("<")[0] and "<" not in dom:
                    traffic, traffic_src = v["hbm_bytes"], "profiles/traffic_%s.json" % args.mode
        except Exception:
            pass
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": int(kb.get(dom, 0)),
                "avg_launch_ms": {k: round(v, 5) for k, v in avg_ms.items()},
                "kernel_bytes": {k: int(v) for k, v in kb.items() if k in avg_ms},
                "layer_fwd_bytes": int(fwd_b), "layer_bwd_bytes": int(bwd_b),
                "layer_achieved_GBs": round((fwd_b + bwd_b) / (step_ms * 1e-3) / 1e9, 2)}

    # ---- the other laplacian mode, same protocol (reported beside the headline) ------------------
    other = "reference" if args.mode == "paper" else "paper"
    st2 = make(other) if not args.skip_other_mode else None
    if use_graph and st2 is not None:
        st2.capture()

    def step2():
        if use_graph:
            st2.replay()
        else:
            st2.step()
        if world > 1:
            agcn_amd.dist.allreduce_bucket(st2.bucket)

    el2 = timed(step2, args.steps, args.warmup) if st2 is not None else None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        G = [Gnp[int(gb.plan.row_off[b]):int(gb.plan.row_off[b]) + int(n[b])] for b in range(len(n))]
        cpu = cpu_baseline({"X": X, "csr": csr, "n": n}, p, G, K, args.mode)

    if rank == 0:
        line = {
            "metric": "AGCN layer fwd+bwd graphs/sec, Tox21 batch=256 K=2",
            "value": round(value, 1), "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "Tox21-shaped batch=%d per GPU, SGC_LL 75->64, K=2, laplacian_mode=%s, "
                                   "packed rows + CSR original Laplacian, fwd+bwd%s"
                                   % (args.batch, args.mode, " + RCCL grad all-reduce" if world > 1 else ""),
                       "global_batch": B_global, "N1_local": N1, "N2_local": N2, "max_n_local": int(max(n)),
                       "launch": "hip_graph" if use_graph else "eager", "parallelism": "dp%d" % world},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "other_mode": None if el2 is None else {
                "laplacian_mode": other, "value": round(B_global * args.steps / el2, 1),
                "ms_per_step": round(1e3 * el2 / args.steps, 5)},
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
