"""One worker of the all-core CPU baseline (bench.py `cpu_baseline.vec_all_cores`).  TEST INFRASTRUCTURE ONLY.

Graphs are independent, so the honest way to use every host core is one process per core, each running the
per-graph numpy oracle (agcn_oracle.py, 1 BLAS thread) on its share of the batch.  Protocol: load the shard, one
warm-up pass, print READY, then for every "GO <reps>" line on stdin run `reps` forward+backward passes and print the
elapsed seconds ("QUIT" ends).
usage: cpu_pool_worker.py <batch.npz> <worker> <n_workers> <mode>"""
import os
import sys
import time

for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_k] = "1"
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import agcn_oracle as orc


def main():
    path, w, nw, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    z = np.load(path)
    n = z["n"]
    mine = np.arange(w, len(n), nw)
    off = np.concatenate([[0], np.cumsum(n)])
    toff = np.concatenate([[0], np.cumsum(n.astype(np.int64) ** 2)])
    X = [z["X"][off[b]:off[b + 1]] for b in mine]
    G = [z["G"][off[b]:off[b + 1]] for b in mine]
    L0 = [z["L0"][toff[b]:toff[b + 1]].reshape(n[b], n[b]) for b in mine]
    p = {k: z["p_" + k] for k in ("weight", "bias", "M_L", "alpha")}
    K = int(z["K"])

    def one():
        caches = orc.layer_forward(X, L0, p, K, mode, orc.SGC_LL)
        orc.layer_backward(caches, G, p, K, mode, orc.SGC_LL)

    one()
    sys.stdout.write("READY\n")
    sys.stdout.flush()
    # rounds: "GO <reps>" -> run -> elapsed seconds; "QUIT" ends (a short pilot round calibrates the timed one)
    while True:
        line = sys.stdin.readline().split()
        if not line or line[0] != "GO":
            break
        reps = int(line[1])
        t0 = time.perf_counter()
        for _ in range(reps):
            one()
        sys.stdout.write("%.6f\n" % (time.perf_counter() - t0))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
