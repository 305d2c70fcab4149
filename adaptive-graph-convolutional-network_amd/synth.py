"""Synthetic Tox21-shaped workloads (no dataset or rdkit is available offline).

Shapes follow the reference's constants: 75-dim atom vectors made of one-hot blocks
(utils/feature/graph_features.py:156-180), graphs of 4..132 atoms (graphs with <= 3 nodes are
dropped, models/graph_structure.py:51-55), molecule-like sparse topology (spanning tree + a few
ring closures, max degree 4), batch_size 256, seed 123 (utils/hyper_parameters.py:8-24)."""
import numpy as np

from .graph_structure import laplacian_csr


def node_counts(rs, B, mu=np.log(16.0), sigma=0.5, lo=4, hi=132):
    return np.clip(np.rint(rs.lognormal(mu, sigma, size=B)), lo, hi).astype(np.int32)


def molecule_adjacency(rs, n):
    adj = [[] for _ in range(n)]
    deg = np.zeros(n, dtype=np.int64)

    def link(a, b):
        if a == b or b in adj[a] or deg[a] >= 4 or deg[b] >= 4:
            return False
        adj[a].append(b)
        adj[b].append(a)
        deg[a] += 1
        deg[b] += 1
        return True

    for v in range(1, n):
        for _ in range(16):
            if link(int(rs.randint(0, v)), v):
                break
        else:
            for u in range(v):
                if link(u, v):
                    break
    for _ in range(int(round(0.1 * n))):
        link(int(rs.randint(0, n)), int(rs.randint(0, n)))
    return [sorted(a) for a in adj]


def atom_features(rs, n):
    """One 1 per one-hot block (44 symbol / 11 degree / 5 numH / 7 valence), 2 small ints,
    5 hybridisation, 1 aromatic flag.  The symbol block is skewed to C/N/O, so duplicate rows
    (identical atoms, r_ij = 0) are common -- as in real molecules."""
    x = np.zeros((n, 75), dtype=np.float32)
    p = np.full(44, 0.05 / 41)
    p[0], p[1], p[2] = 0.72, 0.11, 0.12
    x[np.arange(n), rs.choice(44, size=n, p=p / p.sum())] = 1
    off = 44
    for width, hi in ((11, 5), (5, 4), (7, 5)):
        x[np.arange(n), off + rs.randint(0, hi, size=n)] = 1
        off += width
    x[:, off] = rs.randint(0, 2, size=n) * rs.choice([-1, 0, 0, 0, 1], size=n)
    off += 2
    x[np.arange(n), off + rs.randint(0, 3, size=n)] = 1
    off += 5
    x[:, off] = rs.randint(0, 2, size=n)
    return x


def tox21_batch(seed, B, Fin=75, mu=np.log(16.0), hi=132):
    """Returns dict(n int32 [B], X list of [n,Fin] f32, csr list of (indptr, indices, data f64))."""
    rs = np.random.RandomState(seed)
    n = node_counts(rs, B, mu=mu, hi=hi)
    X, csr = [], []
    for nb in n:
        adj = molecule_adjacency(rs, int(nb))
        X.append(atom_features(rs, int(nb)) if Fin == 75 else
                 np.maximum(rs.randn(int(nb), Fin), 0).astype(np.float32))
        csr.append(laplacian_csr(adj))
    return {"n": n, "X": X, "csr": csr}


def glorot_params(rs, Fin, Fout, K, reslap=False):
    """graphconv.py:19-23,75-83."""
    def g(shape):
        r = np.sqrt(6.0 / (shape[0] + shape[1]))
        return rs.uniform(-r, r, size=shape).astype(np.float32)
    p = {"weight": g((Fin * K, Fout)), "bias": np.zeros(Fout, np.float32), "M_L": g((Fin, Fin)),
         "alpha": np.ones(1, np.float32)}
    if reslap:
        p["beta"] = np.ones(1, np.float32)
    return p


def modelnet_clouds(seed, B, n=1024, Fin=3):
    """ModelNet-shaped point clouds (BASELINE.json config 3): B clouds of n unit-cube points
    (utils/data_loader/meshloader.py:34), adjacency by the mean-distance threshold rule (:264-285, ~50 % dense -> the
    original Laplacian is fed DENSE), features = xyz for the first layer (Fin = 3) or relu(N(0, 1)) for a hidden one.
    Returns (X [B, n, Fin] f32, L [B, n, n] f32)."""
    rs = np.random.RandomState(seed)
    Xs, Ls = [], []
    for _ in range(B):
        pts = rs.uniform(-1, 1, size=(n, 3)).astype(np.float32)
        d = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
        A = d < d.mean()
        np.fill_diagonal(A, False)
        adj = [np.nonzero(A[i])[0].tolist() for i in range(n)]
        ip, ix, val = laplacian_csr(adj)
        L = np.zeros((n, n), np.float32)
        L[np.repeat(np.arange(n), np.diff(ip)), ix] = val.astype(np.float32)
        Ls.append(L)
        Xs.append(pts if Fin == 3 else np.maximum(rs.randn(n, Fin), 0).astype(np.float32))
    return np.stack(Xs), np.stack(Ls)


def layer_work(n, nnz, Fin, Fout, K, paper, dense_l0=False, reslap=False):
    """algorithmic HBM bytes and flops of one layer call, (fwd_bytes, bwd_bytes, fwd_flops,
    bwd_flops).  L_bytes = 4 N2 (dense original Laplacian) or 8 nnz + 4 (N1 + B) (CSR); Reslap adds the L_sum write and
    the L_prev read."""
    n = np.asarray(n, np.int64)
    N1, N2, B = int(n.sum()), int((n * n).sum()), len(n)
    P = Fin * Fin + K * Fin * Fout + Fout + 1
    Lb = 4 * N2 if dense_l0 else 8 * int(nnz) + 4 * (N1 + B)
    fwd_b = 4 * N1 * Fin + Lb + 4 * N1 * Fout + 8 * N2 + 4 * P + (8 * N2 if reslap else 0)
    bwd_b = 4 * N1 * Fout + 8 * N1 * Fin + Lb + 8 * P
    fwd_f = 2 * N1 * Fin * Fin + (3 * Fin + 10) * N2 + 2 * (K - 1) * N2 * Fin + 2 * N1 * K * Fin * Fout
    bwd_f = 2 * (2 * N1 * K * Fin * Fout) + 2 * (2 * (K - 1) * N2 * Fin) + ((4 * N2 * Fin + 4 * N1 * Fin * Fin) if paper else 0)
    return fwd_b, bwd_b, fwd_f, bwd_f
