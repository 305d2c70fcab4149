#!/usr/bin/env python3
"""One-rank RCCL all-reduce latency of the gradient buckets of this path (61 KB: one 75 -> 64 layer; 563 KB: SimpleAGCN + DenseMol +
heads), on the stream behind a kernel and captured into a HIP graph -- what a data-parallel step adds per rank before any wire time
(the collective is latency-bound).  usage: python tools/rccl_latency.py    (one GPU; prints one JSON line)"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
for floats in (15290, 140821):
    b = torch.randn(floats, device=dev)
    for _ in range(20):
        dist.all_reduce(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        dist.all_reduce(b)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 500 * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            dist.all_reduce(b)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        g.replay()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / 1000 * 1e6
    out["%d_bytes" % (4 * floats)] = {"eager_us_per_call": round(eager, 2), "in_graph_us_per_call": round(graphed, 2)}
print(json.dumps(out))
dist.destroy_process_group()
