"""Per-sample graph objects: host-side preprocessing that builds the ORIGINAL (intrinsic)
sparse normalised Laplacian of each graph.

Mirrors the interface of the reference's `models/graph_structure.py` (`Graph`, `MolGraph`:
same constructor arguments and attributes -- node_features, n_node, n_feat, degree_list
(int32), adj_lists, max_deg, min_deg, has_Lap, Laplacian as a scipy CSR float64 matrix) but
computes the Laplacian with vectorised numpy directly in CSR form: no networkx graph, no
dense n x n intermediate.

Algorithm (graph_structure.py:85-130, the *doubly* normalised Laplacian):
    A~ = A + I;  Ahat = D~^-1/2 A~ D~^-1/2 (rowsum degree, inf -> 0)
    L  = I - D'^-1/2 Ahat D'^-1/2,  D' = colsum(Ahat) + spacing(0)
Graphs with <= 3 nodes get no Laplacian (graph_structure.py:51-55).
"""
import numpy as np
import scipy.sparse as sp


def laplacian_csr(adj_lists):
    """Returns (indptr int32 [n+1], indices int32 [nnz], data float64 [nnz]) of the original
    Laplacian in canonical CSR (sorted columns, exact zeros pruned like scipy does)."""
    n = len(adj_lists)
    deg = np.fromiter((len(a) for a in adj_lists), dtype=np.int64, count=n)
    src = np.repeat(np.arange(n, dtype=np.int64), deg)
    dst = np.fromiter((j for a in adj_lists for j in a), dtype=np.int64, count=int(deg.sum()))
    if dst.size and (dst.min() < 0 or dst.max() >= n):
        raise ValueError("adjacency list refers to a node outside the graph")
    # undirected simple graph (networkx from_dict_of_lists): symmetrise, collapse duplicates
    ar = np.arange(n, dtype=np.int64)
    key = np.unique(np.concatenate([src * n + dst, dst * n + src, ar * n + ar]))
    r, c = key // n, key % n
    # A + I: off-diagonal entries 1; diagonal 1 (identity) + 1 if the graph has a self loop there
    loops = np.zeros(n, dtype=bool)
    loops[src[src == dst]] = True
    val = np.ones(key.shape[0], dtype=np.float64)
    diag = r == c
    val[diag] += loops[r[diag]]
    rowsum = np.bincount(r, weights=val, minlength=n)
    with np.errstate(divide="ignore"):
        dinv = np.power(rowsum, -0.5)
    dinv[np.isinf(dinv)] = 0.0
    # Ahat = (A~ D)^T D : entry [r, c] = (a[c, r] * dinv[r]) * dinv[c]   (a symmetric)
    w = (val * dinv[r]) * dinv[c]
    d = np.bincount(c, weights=w, minlength=n)
    d = d + np.spacing(np.array(0, np.float64))
    d = 1.0 / np.sqrt(d)
    data = diag.astype(np.float64) - (d[r] * w) * d[c]
    keep = data != 0
    r, c, data = r[keep], c[keep], data[keep]
    indptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=n), out=indptr[1:])
    return indptr, c.astype(np.int32), data


class Graph(object):
    """Adaptive graph object for one data sample (reference: graph_structure.py:19-130)."""

    def __init__(self, node_features, adj_lists, max_deg, min_deg):
        self.node_features = node_features
        self.n_node, self.n_feat = node_features.shape
        self.degree_list = np.array([len(nbrs) for nbrs in adj_lists], dtype=np.int32)
        self.adj_lists = adj_lists
        self.max_deg = max_deg
        self.min_deg = min_deg
        self.has_Lap = True
        if self.n_node > 3:
            self.Laplacian = self.compute_laplacian(adj_lists)
        else:
            self.Laplacian = None
            self.has_Lap = False

    def get_num_nodes(self):
        return self.n_node

    def get_num_features(self):
        return self.n_feat

    def get_max_degree(self):
        return self.max_deg

    def get_original_adj_list(self):
        return self.adj_lists

    def get_node_fearture(self):  # (sic) the reference's spelling
        return self.node_features

    @staticmethod
    def compute_adj_matrix(adj_lists):
        """Symmetric 0/1 adjacency as CSR (graph_structure.py:76-83)."""
        n = len(adj_lists)
        deg = np.fromiter((len(a) for a in adj_lists), dtype=np.int64, count=n)
        src = np.repeat(np.arange(n, dtype=np.int64), deg)
        dst = np.fromiter((j for a in adj_lists for j in a), dtype=np.int64, count=int(deg.sum()))
        key = np.unique(np.concatenate([src * n + dst, dst * n + src]))
        return sp.csr_matrix((np.ones(key.shape[0], dtype=np.int64), (key // n, key % n)), shape=(n, n))

    def compute_laplacian(self, adj_lists):
        indptr, indices, data = laplacian_csr(adj_lists)
        n = len(adj_lists)
        L = sp.csr_matrix((data, indices, indptr), shape=(n, n))
        assert np.abs(L - L.T).mean() < 1e-9    # same invariant as graph_structure.py:106
        return L


class MolGraph(Graph):
    """Molecular graph derived from a SMILES string (graph_structure.py:133-146)."""

    def __init__(self, node_features, adj_lists, max_deg=10, min_deg=0, smiles=None):
        super(MolGraph, self).__init__(node_features, adj_lists, max_deg, min_deg)
        self.smiles = smiles
