#!/usr/bin/env python3
"""Secondary workloads of SURVEY.md §8(d) / BASELINE.json `configs`, same protocol as bench.py
(inputs resident, HIP-graph replay, wall clock around K steps bracketed by synchronize):

  tox21-l1 / l2 / l3 / l4 : the four SimpleAGCN layer shapes 75->64, 64->128, 128->128, 128->64 at B=256
  ref-feed                : 75->64 at B=256 in the reference's padded feed layout (max_atom 132, dense Laplacians)
  tox21-2048              : 75->64 at B=2048 (config[3] on one GPU)
  modelnet-3-64 / 64-128  : B graphs of n=1024 points, dense original Laplacian (config[2])
  modelnet-reslap         : the same clouds through SGC_LL_Reslap 64 -> 128 with L_prev (the reference's point-cloud layer)
  epoch                   : ingest (graph store -> packed batch -> upload) + layer over an HIV-shaped dataset
  stack                   : 4 x SGC_LL -> DenseMol 64->256 -> GraphGatherMol(tanh), autograd, eager launches

One JSON line per workload; `--prof` adds the library's per-kernel HIP-event averages.
Not part of the driver contract (bench.py is); results are copied into profiles/."""
import os
# more hardware queues than the default four: this process keeps several streams alive (the library's side streams,
# the epoch lanes), and streams that share a queue run strictly in turn (2 lanes: 1.9 instead of 3.1 M graphs/s)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, steps, warmup, dev):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps


def kernel_profile(amd, fn, steps, dev):
    lib = amd._ffi.lib()
    for _ in range(3):
        fn()
    torch.cuda.synchronize(dev)
    lib.agcn_profile_enable(1)
    for _ in range(steps):
        fn()
    prof = amd._ffi.profile_collect()
    lib.agcn_profile_enable(0)
    return {k: round(1e3 * ms / c, 2) for k, (ms, c) in prof.items() if c > 0}   # us per launch


def flops_layer(n, Fin, Fout, K, paper):
    """SURVEY.md §8(d) flop model."""
    n = np.asarray(n, np.float64)
    N1, N2 = n.sum(), (n * n).sum()
    fwd = 2 * N1 * Fin * Fin + (3 * Fin + 10) * N2 + 2 * (K - 1) * N2 * Fin + 2 * N1 * K * Fin * Fout
    bwd = 2 * (2 * N1 * K * Fin * Fout) + 2 * (2 * (K - 1) * N2 * Fin)
    if paper:
        bwd += 4 * N2 * Fin + 4 * N1 * Fin * Fin
    return fwd + bwd


def bytes_layer(n, nnz, Fin, Fout, K, dense_l0):
    n = np.asarray(n, np.float64)
    N1, N2, B = n.sum(), (n * n).sum(), len(n)
    P = Fin * Fin + K * Fin * Fout + Fout + 1
    Lb = 4 * N2 if dense_l0 else 8 * nnz + 4 * (N1 + B)
    fwd = 4 * N1 * Fin + Lb + 4 * N1 * Fout + 8 * N2 + 4 * P
    bwd = 4 * N1 * Fout + 4 * N1 * Fin + Lb + 4 * N1 * Fin + 8 * P
    return fwd + bwd


def modelnet_batch(amd, B, n, Fin, dev, seed=10):
    """Unit-cube point clouds, mean-distance-threshold adjacency (utils/meshloader.py:264-285), dense feed."""
    from agcn_amd import synth
    X, L = synth.modelnet_clouds(seed, B, n, Fin)
    ds = np.stack([np.full(B, n), np.full(B, -1)], 1).astype(np.int32)
    return amd.GraphBatch.from_padded(torch.from_numpy(X).to(dev), torch.from_numpy(L).to(dev), ds, dev), np.full(B, n, np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="paper", choices=["paper", "reference"])
    ap.add_argument("--only", default="", help="comma list of workload names")
    ap.add_argument("--modelnet-batch", type=int, default=8)
    ap.add_argument("--prof", action="store_true")
    ap.add_argument("--eager", action="store_true", help="eager launches instead of HIP-graph replay (profiler runs)")
    ap.add_argument("--epoch-graphs", type=int, default=8192)
    args = ap.parse_args()
    import agcn_amd as amd
    from agcn_amd import synth

    dev = torch.device("cuda", 0)
    amd._ffi.lib()
    only = set(x for x in args.only.split(",") if x)
    K = 2

    def want(name):
        return not only or name in only

    def run_layer(name, gb, n, nnz, Fin, Fout, dense):
        p = synth.glorot_params(np.random.RandomState(123), Fin, Fout, K)
        layer = amd.SGC_LL(Fout, Fin, len(n), K=K, laplacian_mode=args.mode, device=dev)
        layer.load_vars(p)
        dY = torch.from_numpy(np.random.RandomState(7).randn(gb.plan.rows, Fout).astype(np.float32)).to(dev)
        st = amd.SgcLLStep(layer, gb, dY=dY)
        if args.eager:
            t = timed(st.step, args.steps, args.warmup, dev)
        else:
            st.capture()
            t = timed(st.replay, args.steps, args.warmup, dev)
        fl = flops_layer(n, Fin, Fout, K, args.mode == "paper")
        by = bytes_layer(n, nnz, Fin, Fout, K, dense)
        out = {"workload": name, "mode": args.mode, "graphs": int(len(n)), "N1": int(np.sum(n)),
               "N2": int(np.sum(np.asarray(n, np.int64) ** 2)), "Fin": Fin, "Fout": Fout,
               "us_per_step": round(t * 1e6, 2), "graphs_per_s": round(len(n) / t, 1),
               "algorithmic_GB_s": round(by / t / 1e9, 1), "algorithmic_TFLOP_s": round(fl / t / 1e12, 3)}
        if args.prof:
            out["kernel_us"] = kernel_profile(amd, st.step, 20, dev)
        print(json.dumps(out), flush=True)

    shapes = [("tox21-l1", 75, 64), ("tox21-l2", 64, 128), ("tox21-l3", 128, 128), ("tox21-l4", 128, 64)]
    for name, Fin, Fout in shapes:
        if not want(name):
            continue
        data = synth.tox21_batch(123, 256, Fin)
        gb = amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
        run_layer(name, gb, data["n"], sum(len(c[1]) for c in data["csr"]), Fin, Fout, False)

    if want("reslap"):
        # SGC_LL_Reslap 64 -> 64 with a previous summed Laplacian (ResAGCN-style second layer)
        data = synth.tox21_batch(123, 256, 64)
        gb = amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
        layer = amd.SGC_LL_Reslap(64, 64, 256, K=K, laplacian_mode=args.mode, device=dev)
        layer.load_vars(synth.glorot_params(np.random.RandomState(123), 64, 64, K, reslap=True))
        Lp = 0.1 * torch.randn(gb.plan.tile_elems, device=dev)
        dY = torch.randn(gb.plan.rows, 64, device=dev)
        st = amd.SgcLLStep(layer, gb, L_prev=Lp, dY=dY)
        st.capture()
        t = timed(st.replay, args.steps, args.warmup, dev)
        out = {"workload": "reslap 64->64 with L_prev, B=256", "mode": args.mode, "graphs": 256,
               "us_per_step": round(t * 1e6, 2), "graphs_per_s": round(256 / t, 1)}
        if args.prof:
            out["kernel_us"] = kernel_profile(amd, st.step, 20, dev)
        print(json.dumps(out), flush=True)

    if want("ref-feed"):
        # the reference's own feed layout (graph_topology.py:84-135): every graph padded to max_atom = 132 rows,
        # dense [max_atom, max_atom] Laplacians -- the literal drop-in call
        data = synth.tox21_batch(123, 256, 75)
        n, B, MA = data["n"], 256, 132
        Xp = np.zeros((B, MA, 75), np.float32)
        Lp = np.zeros((B, MA, MA), np.float32)
        for b in range(B):
            Xp[b, :n[b]] = data["X"][b]
            ip, ix, v = data["csr"][b]
            rr = np.repeat(np.arange(n[b]), np.diff(ip))
            Lp[b, rr, ix] = v.astype(np.float32)
        ds = np.stack([n, -np.ones(B)], 1).astype(np.int32)
        gb = amd.GraphBatch.from_padded(torch.from_numpy(Xp).to(dev), torch.from_numpy(Lp).to(dev), ds, dev)
        run_layer("ref-feed (padded lists in, rows packed once by agcn_rows_pack, dense L in place)", gb, n, 0, 75, 64, True)

    if want("tox21-2048"):
        data = synth.tox21_batch(123, 2048, 75)
        gb = amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
        run_layer("tox21-2048", gb, data["n"], sum(len(c[1]) for c in data["csr"]), 75, 64, False)

    for name, Fin, Fout in (("modelnet-3-64", 3, 64), ("modelnet-64-128", 64, 128)):
        if not want(name):
            continue
        gb, n = modelnet_batch(amd, args.modelnet_batch, 1024, Fin, dev)
        run_layer(name, gb, n, 0, Fin, Fout, True)

    if want("modelnet-128-128"):
        # the network's third convolution (l_n_filters = [64, 128, 128, 64]): 128 input features, SGC_LL
        gb, n = modelnet_batch(amd, args.modelnet_batch, 1024, 128, dev)
        run_layer("modelnet-128-128", gb, n, 0, 128, 128, True)

    if want("modelnet-reslap-128"):
        gb, n = modelnet_batch(amd, args.modelnet_batch, 1024, 128, dev)
        layer = amd.SGC_LL_Reslap(128, 128, len(n), K=K, laplacian_mode=args.mode, device=dev)
        layer.load_vars(synth.glorot_params(np.random.RandomState(123), 128, 128, K, reslap=True))
        g = torch.Generator(device="cpu").manual_seed(5)
        Lp = (0.05 * torch.randn(len(n), 1024, 1024, generator=g))
        Lp = (0.5 * (Lp + Lp.transpose(1, 2))).reshape(-1).to(dev)
        dY = torch.from_numpy(np.random.RandomState(7).randn(gb.plan.rows, 128).astype(np.float32)).to(dev)
        st = amd.SgcLLStep(layer, gb, L_prev=Lp, dY=dY, L_prev_symmetric=True)
        if args.eager:
            t = timed(st.step, args.steps, args.warmup, dev)
        else:
            st.capture()
            t = timed(st.replay, args.steps, args.warmup, dev)
        out = {"workload": "modelnet-reslap 128->128 with L_prev", "mode": args.mode, "graphs": int(len(n)),
               "N2": int(np.sum(np.asarray(n, np.int64) ** 2)), "us_per_step": round(t * 1e6, 2), "graphs_per_s": round(len(n) / t, 1)}
        if args.prof:
            out["kernel_us"] = kernel_profile(amd, st.step, 20, dev)
        print(json.dumps(out), flush=True)

    if want("modelnet-reslap"):
        # the layer the reference's point-cloud network is made of (Point_AGCN.py:37-71: SGC_LL_Reslap throughout), hidden width,
        # with the previous layer's (symmetric) Laplacian coming in and a gradient arriving on L_sum
        gb, n = modelnet_batch(amd, args.modelnet_batch, 1024, 64, dev)
        layer = amd.SGC_LL_Reslap(128, 64, len(n), K=K, laplacian_mode=args.mode, device=dev)
        layer.load_vars(synth.glorot_params(np.random.RandomState(123), 64, 128, K, reslap=True))
        g = torch.Generator(device="cpu").manual_seed(5)
        Lp = (0.05 * torch.randn(len(n), 1024, 1024, generator=g))
        Lp = (0.5 * (Lp + Lp.transpose(1, 2))).reshape(-1).to(dev)
        dY = torch.from_numpy(np.random.RandomState(7).randn(gb.plan.rows, 128).astype(np.float32)).to(dev)
        st = amd.SgcLLStep(layer, gb, L_prev=Lp, dY=dY, L_prev_symmetric=True)
        if args.eager:
            t = timed(st.step, args.steps, args.warmup, dev)
        else:
            st.capture()
            t = timed(st.replay, args.steps, args.warmup, dev)
        out = {"workload": "modelnet-reslap 64->128 with L_prev", "mode": args.mode, "graphs": int(len(n)),
               "N2": int(np.sum(np.asarray(n, np.int64) ** 2)), "us_per_step": round(t * 1e6, 2), "graphs_per_s": round(len(n) / t, 1)}
        if args.prof:
            out["kernel_us"] = kernel_profile(amd, st.step, 20, dev)
        print(json.dumps(out), flush=True)

    if want("epoch"):
        # ingest + layer over a dataset: HIV-shaped sizes (mu = ln 23, clip [4, 222]); every batch is gathered
        # from the graph store (native pack, pinned staging, async upload, prefetch thread) and run through
        # SGC_LL 75 -> 64 forward + backward (autograd, eager launches)
        G = args.epoch_graphs
        data = synth.tox21_batch(321, G, mu=np.log(23.0), hi=222)
        t0 = time.perf_counter()
        store = amd.GraphStore.from_arrays(data["n"], data["X"], data["csr"])
        t_store = time.perf_counter() - t0
        layer = amd.SGC_LL(64, 75, 256, K=K, laplacian_mode=args.mode, device=dev)
        layer.load_vars(synth.glorot_params(np.random.RandomState(123), 75, 64, K))
        res = {}
        n_all = data["n"].astype(np.int64)
        ep = amd.SgcLLEpochStep(layer, rows_cap=int(1.5 * 256 * n_all.mean() + 4 * n_all.max()),
                                tiles_cap=int(2.0 * 256 * (n_all * n_all).mean() + 4 * int(n_all.max()) ** 2 + 1024),
                                batch_cap=256, device=dev)
        for what in ("ingest", "ingest+layer", "ingest+layer(static buffers)"):
            for pf in (0, 2):
                for rep in range(2):        # first pass: pinned staging allocation, code-object loading
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for gb, _, _, _ in store.iterbatches(256, dev, deterministic=True, prefetch=pf):
                        if what == "ingest+layer(static buffers)":
                            ep.step(gb)
                        elif what != "ingest":
                            for q in layer.vars.values():
                                q.grad = None
                            Xt = gb.X.requires_grad_(True)
                            layer.batch_size = gb.plan.B      # (np.linspace cuts give batches of 255 / 256 graphs)
                            nodes, _, _ = layer({"node_features": Xt, "graph_batch": gb})
                            nodes.data.backward(torch.ones_like(nodes.data))
                    torch.cuda.synchronize(dev)
                    res["%s prefetch=%d" % (what, pf)] = round(G / (time.perf_counter() - t0), 1)
        # the store RESIDENT in HBM: one upload per dataset, a batch = one table upload + one gather launch
        t0 = time.perf_counter()
        rstore = store.to_device(dev, 256)
        torch.cuda.synchronize(dev)
        res["resident store upload_s"] = round(time.perf_counter() - t0, 3)
        ep2 = amd.SgcLLEpochStep(layer, rows_cap=rstore.rows_cap, tiles_cap=rstore.tiles_cap, batch_cap=256, device=dev)
        for what in ("resident stage", "resident stage+layer(static buffers)"):
            for rep in range(3):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for gb, _, _, _ in rstore.iterbatches(256, deterministic=True):
                    if what != "resident stage":
                        ep2.step(gb)
                torch.cuda.synchronize(dev)
                res[what] = round(G / (time.perf_counter() - t0), 1)
        # BASELINE configs[4]: features at rest in bf16, Laplacian and all arithmetic fp32
        rh = store.to_device(dev, 256, dtype=torch.bfloat16)
        for rep in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for gb, _, _, _ in rh.iterbatches(256, deterministic=True):
                ep2.step(gb)
            torch.cuda.synchronize(dev)
            res["resident stage+layer(static buffers), bf16 features at rest"] = round(G / (time.perf_counter() - t0), 1)
        del rh
        # consecutive batches overlapped on 2 / 3 streams (evaluation passes, gradient accumulation: SgcLLEpochLanes)
        sels = store._selections(256, True, None)
        for lanes in (2, 3):
            rl = store.to_device(dev, 256, lanes=lanes)
            epl = amd.SgcLLEpochLanes(layer, rl, lanes=lanes)
            for rep in range(3):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for sel in sels:
                    epl.step(sel)
                epl.synchronize()
                res["resident stage+layer, %d lanes" % lanes] = round(G / (time.perf_counter() - t0), 1)
            del rl, epl
        print(json.dumps({"workload": "epoch over %d HIV-shaped graphs, batches of 256" % G, "mode": args.mode,
                          "store_build_s": round(t_store, 3), "graphs_per_s": res}), flush=True)

    if want("stack"):
        data = synth.tox21_batch(123, 256, 75)
        B = len(data["n"])
        gb = amd.GraphBatch.from_arrays(data["n"], np.concatenate(data["X"], 0), data["csr"], dev)
        rs = np.random.RandomState(123)
        layers, fin = [], 75
        for w in (64, 128, 128, 64):
            lay = amd.SGC_LL(w, fin, B, K=K, laplacian_mode=args.mode, device=dev)
            lay.load_vars(synth.glorot_params(rs, fin, w, K))
            layers.append(lay)
            fin = w
        dense = amd.DenseMol(256, fin, device=dev)
        dense.build(dev)
        gather = amd.GraphGatherMol(B, activation="tanh")
        G = torch.from_numpy(rs.randn(B, 256).astype(np.float32)).to(dev)
        params = [v for lay in layers for v in lay.vars.values()] + dense.parameters()

        def step():
            for q in params:
                q.grad = None
            h = gb.X
            for lay in layers:
                h, _, _ = lay({"node_features": h, "graph_batch": gb})
            h = dense({"node_features": h, "graph_batch": gb})
            out = gather({"node_features": h, "graph_batch": gb})
            out.backward(G)

        t = timed(step, args.steps, args.warmup, dev)
        out = {"workload": "stack 75-64-128-128-64 + DenseMol 256 + GraphGather, autograd eager",
               "mode": args.mode, "graphs": B, "us_per_step": round(t * 1e6, 2),
               "graphs_per_s": round(B / t, 1)}
        if args.prof:
            out["kernel_us"] = kernel_profile(amd, step, 10, dev)
        print(json.dumps(out), flush=True)
        # the same step (autograd forward + backward of the whole stack) captured into ONE HIP graph
        # (amd.GraphedStep captures on the stream its warm-up ran on: the layers' reserves then come out of the lease pool; a capture
        #  on another stream allocates them inside the graph, with four 64 KB header fills -- 20 us -- replayed every step)
        g = amd.GraphedStep(step, dev)
        t = timed(g.replay, args.steps, args.warmup, dev)
        out = {"workload": "stack 75-64-128-128-64 + DenseMol 256 + GraphGather, one HIP graph",
               "mode": args.mode, "graphs": B, "us_per_step": round(t * 1e6, 2),
               "graphs_per_s": round(B / t, 1)}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
