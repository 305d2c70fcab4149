#!/usr/bin/env python
"""Diagnostic: builds tools/build/libagcn_hip_stamps.so (on demand: `--build`; not shipped in lib/) with -DAGCN_STAMPS (per-workgroup wall-clock stamps
at phase boundaries), runs a few benchmark steps and prints, per kernel, the phase breakdown.
Never used by the product, tests or bench.py."""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "adaptive-graph-convolutional-network_amd")
LIB = os.path.join(ROOT, "tools", "build", "libagcn_hip_stamps.so")


def build():
    src = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.cpp")))
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17",
           "-DAGCN_STAMPS", "-fgpu-rdc", "-o", LIB] + src
    subprocess.check_call(cmd)


def main():
    if "--build" in sys.argv:
        build()
        return
    os.environ["AGCN_HIP_LIB"] = LIB
    import numpy as np
    import torch
    import agcn_amd
    from agcn_amd import synth
    mode = "paper" if "--reference" not in sys.argv else "reference"
    B = int(os.environ.get("AGCN_STAMP_B", 256))
    data = synth.tox21_batch(123, B)
    if os.environ.get("AGCN_STAMP_SHARD_OF"):      # shard 0 of an emulated N-GPU run (holds the largest graph)
        N = int(os.environ["AGCN_STAMP_SHARD_OF"])
        big = synth.tox21_batch(123, B * N)
        mine = agcn_amd.dist.shard_graphs(agcn_amd.dist.graph_cost(big["n"], 75, 64, 2), N)[0]
        data = {"n": big["n"][mine], "X": [big["X"][i] for i in mine], "csr": [big["csr"][i] for i in mine]}
    Fin, Fout = int(os.environ.get("AGCN_STAMP_FIN", 75)), int(os.environ.get("AGCN_STAMP_FOUT", 64))   # layer shape
    p = synth.glorot_params(np.random.RandomState(123), Fin, Fout, 2)
    X = np.concatenate(data["X"], 0)
    if Fin != 75:
        X = np.maximum(np.random.RandomState(5).randn(X.shape[0], Fin), 0).astype(np.float32)   # deeper layer: relu(N(0,1))
    gb = agcn_amd.GraphBatch.from_arrays(data["n"], X, data["csr"], "cuda:0")
    layer = agcn_amd.SGC_LL(Fout, Fin, B, K=2, laplacian_mode=mode, device="cuda:0")
    layer.load_vars(p)
    st = agcn_amd.SgcLLStep(layer, gb, dY=torch.randn(gb.plan.rows, Fout, device="cuda:0"))
    lib = agcn_amd._ffi.lib()
    lib.agcn_debug_read_stamps.restype = C.c_int
    lib.agcn_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    for _ in range(5):
        st.step()
    buf = np.zeros(8 * 1024 * 16, dtype=np.uint64)
    lib.agcn_debug_read_stamps(buf.ctypes.data, buf.size, 1)
    seq = os.environ.get("AGCN_STAMP_SEQ", "")     # e.g. "ff": two forwards in a row, "bf": backward then forward
    reps = int(os.environ.get("AGCN_STAMP_REPS", 1))   # > 1: the sequence is repeated, medians over the repetitions are printed

    def one():
        if seq:
            for ch in seq:
                st.forward() if ch == "f" else st.backward()
        else:
            st.step()
        lib.agcn_debug_read_stamps(buf.ctypes.data, buf.size, 1)
        return buf.reshape(8, 1024, 16).astype(np.int64).copy()
    runs = [one() for _ in range(reps)]
    s = runs[0]
    names = ["PROJECT", "graph_fwd/slab_sim", "FILTER", "FILTER_BWD", "graph_bwd", "PROJECT_BWD", "slab_lap/DENSE", "param_grad"]
    for k, nm in enumerate(names):
        blk = s[k]
        used = blk[:, 0] > 0
        if not used.any():
            continue
        t0 = blk[used, 0].min()
        cols = [c for c in range(16) if (blk[used, c] > 0).any()]
        cols = sorted(cols, key=lambda c: np.median(blk[used, c]))
        print("%-12s blocks %4d  span %.2f us (first start -> last end)" % (nm, used.sum(), (blk[used].max() - t0) / 100.0))
        last = cols[-1]
        rel = (blk[used][:, cols] - t0) / 100.0
        print("   stamp idx      :", cols)
        print("   median (us)    :", np.round(np.median(rel, 0), 2))
        print("   max    (us)    :", np.round(rel.max(0), 2))
        if os.environ.get("AGCN_STAMP_ROUNDS"):      # phase durations of the workgroups that start first vs those that start late
            start = blk[used][:, cols[0]] - t0
            dur = np.diff(blk[used][:, cols], axis=1) / 100.0
            early, late = start <= np.percentile(start, 25), start >= np.percentile(start, 75)
            print("   phase durations, earliest quarter (start <= %.1f us):" % (np.percentile(start, 25) / 100.0), np.round(np.median(dur[early], 0), 2))
            print("   phase durations, latest quarter   (start >= %.1f us):" % (np.percentile(start, 75) / 100.0), np.round(np.median(dur[late], 0), 2))
        # the block that ends last
        w = np.argmax(blk[used, last])
        print("   slowest block  :", np.round(rel[w], 2), "(block %d)" % np.nonzero(used)[0][w])
        if reps > 1:
            meds, maxs = [], []
            for r in runs:
                b = r[k]
                u = b[:, 0] > 0
                rr = (b[u][:, cols] - b[u, 0].min()) / 100.0
                meds.append(np.median(rr, 0)); maxs.append(rr.max(0))
            np.set_printoptions(linewidth=250)
            print("   over %d repetitions: median of medians:" % reps, np.round(np.median(meds, 0), 2))
            print("   over %d repetitions: median of maxima :" % reps, np.round(np.median(maxs, 0), 2))


    # whole launches: the forward = members (slot 0) + one-workgroup bodies (slot 1), the backward = slots 5 + 4
    for nm, slots in (("forward launch", (0, 1)), ("backward launch", (5, 4)), ("param-grad launch", (7,))):
        spans, per = [], {k: ([], []) for k in slots}
        for r in runs:
            v = np.concatenate([r[k][r[k][:, 0] > 0] for k in slots])
            if len(v) == 0:
                continue
            t0 = np.where(v > 0, v, np.iinfo(np.int64).max).min()
            spans.append((v.max() - t0) / 100.0)
            for k in slots:
                b = r[k][r[k][:, 0] > 0]
                if len(b):
                    st0 = np.where(b > 0, b, np.iinfo(np.int64).max).min(1)
                    per[k][0].append((st0.max() - t0) / 100.0)
                    per[k][1].append((b.max() - t0) / 100.0)
        if spans:
            print("%-18s first start -> last stamp, median over %d repetitions: %.2f us" % (nm, len(spans), np.median(spans)))
            for k in slots:
                if per[k][0]:
                    print("    %-20s last workgroup starts at %.2f, last one ends at %.2f us of the launch" % (names[k], np.median(per[k][0]), np.median(per[k][1])))
    # launch boundaries inside the last step of the sequence (absolute 100 MHz clocks): last stamp of one launch -> first stamp of the next
    if seq.endswith("fb") or not seq:
        def ends(r, slots):
            v = np.concatenate([r[k][r[k][:, 0] > 0] for k in slots])
            return np.where(v > 0, v, np.iinfo(np.int64).max).min(), v.max()
        gaps = {"forward -> backward": [], "backward -> param-grad": []}
        for r in runs:
            try:
                f0, f1 = ends(r, (0, 1)); b0, b1 = ends(r, (5, 4)); p0, p1 = ends(r, (7,))
            except ValueError:
                continue
            gaps["forward -> backward"].append((b0 - f1) / 100.0)
            gaps["backward -> param-grad"].append((p0 - b1) / 100.0)
        for k, v in gaps.items():
            if v:
                print("boundary %-24s last stamp -> first stamp of the next launch: median %.2f us" % (k, np.median(v)))


if __name__ == "__main__":
    main()
