"""CPU restatement (numpy) of the reference's AGCN adaptive graph-convolution
hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *oracle*: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker /
reported baseline -- never as (part of) the product path.  The product
(`agcn_amd`) does not import it and fails loudly when its HIP library is
missing.

Pinning: the reference has no tests or golden vectors for this path
(SURVEY.md section 4), so this restatement is pinned against outputs of the
reference's own, unmodified layer classes executed in the build container
behind a fake-`tensorflow` shim (`oracle/refshim/gen_golden.py` ->
`tests/golden/*.npz`; checked by `tests/test_oracle_golden.py`).  The
`laplacian_mode="paper"` branch (matrix-product normalised Laplacian with
gradients to M_L) has no counterpart in the reference (its py_func blocks the
gradient and its `D * W * D` is elementwise); that branch is pinned against
torch-CPU autograd in float64 and finite differences instead
(`tests/test_oracle_autograd.py`).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Arithmetic follows the reference's dtypes: float32 on the
layer path, float64 + int32 for the host-side original Laplacian.
All functions are dtype-generic: feed float64 arrays to get a float64 run.
"""
import numpy as np

REFERENCE = 0   # laplacian_mode="reference": res_L == I (elementwise D*W*D), no grad to M_L
PAPER = 1       # laplacian_mode="paper": I - D^-1/2 W D^-1/2, fully differentiable
SGC_LL = 0
SGC_LL_RESLAP = 1


# ----------------------------------------------------------------------------
# a1  models/graph_structure.py:22-55, 76-130
def degree_list(adj_lists):
    """graph_structure.py:43 -- int32 degree per node = len(adj_lists[i])."""
    return np.array([len(nbrs) for nbrs in adj_lists], dtype=np.int32)


def adjacency_dense(adj_lists):
    """graph_structure.py:76-83.  nx.from_dict_of_lists builds an UNDIRECTED
    simple graph (node order = key order 0..n-1): an edge i-j exists if j is in
    adj_lists[i] or i in adj_lists[j]; duplicates collapse; a self reference
    makes a self loop with weight 1.  Returned as float32 like :127."""
    n = len(adj_lists)
    A = np.zeros((n, n), dtype=np.float32)
    for i, nbrs in enumerate(adj_lists):
        for j in nbrs:
            A[i, j] = 1.0
            A[j, i] = 1.0
    return A


def graph_laplacian(adj_lists):
    """graph_structure.py:85-130: the *doubly* normalised original Laplacian.

      A~   = A + I                               (:124, float64 from sp.eye)
      Ahat = D~^-1/2 A~ D~^-1/2, D~ = rowsum(A~) (:110-120, inf -> 0)
      L    = I - D'^-1/2 Ahat D'^-1/2, D' = colsum(Ahat) + spacing(0)  (:87-104)

    Returns (indptr int32, indices int32, data float64, dense float64) in
    canonical CSR (sorted column indices; structure = the non-zeros of L:
    scipy's sparse `I - D*W*D` prunes entries that cancel to exactly 0, e.g.
    the diagonal of an isolated node).
    Multiplication order follows scipy's: ((a_ij * dinv_j) * dinv_i) for Ahat
    (:119, (A D)^T D) and ((d_i * w_ij) * d_j) for D*W*D (:104).
    """
    n = len(adj_lists)
    A = adjacency_dense(adj_lists).astype(np.float64) + np.eye(n)
    rowsum = A.sum(axis=1)
    with np.errstate(divide="ignore"):
        dinv = np.power(rowsum, -0.5)
    dinv[np.isinf(dinv)] = 0.0
    # normalized_adj = (A . D)^T . D   -> entry [j, i] = (A[i, j] * dinv[j]) * dinv[i]
    W = ((A * dinv[None, :]).T) * dinv[None, :]
    d = W.sum(axis=0)
    d = d + np.spacing(np.array(0, W.dtype))
    d = 1.0 / np.sqrt(d)
    L = np.eye(n) - (d[:, None] * W) * d[None, :]
    mask = L != 0
    indptr = np.zeros(n + 1, dtype=np.int32)
    indptr[1:] = np.cumsum(mask.sum(axis=1))
    indices = np.nonzero(mask)[1].astype(np.int32)
    data = L[mask]
    return indptr, indices, data, L


def has_laplacian(n_node):
    """graph_structure.py:51-55 -- graphs with <= 3 nodes have no Laplacian."""
    return n_node > 3


# ----------------------------------------------------------------------------
# a2  models/tf_modules/graph_topology.py:84-135
def pad_batch(node_features, laplacians_dense, max_atom):
    """graph_topology.py:84-98,100-135: zero-pad X to [max_atom, Fin] and the
    densified Laplacian to [max_atom, max_atom]; data_slice rows [n, -1] and
    lap_slice rows [n, n], both int32.  Values are float32 at the feed
    (placeholders are float32, graph_topology.py:58-73)."""
    Xp, Lp, ds, ls = [], [], [], []
    for x, L in zip(node_features, laplacians_dense):
        n = x.shape[0]
        xp = np.zeros((max_atom, x.shape[1]), dtype=np.float32)
        xp[:n] = x
        lp = np.zeros((max_atom, max_atom), dtype=np.float32)
        lp[:n, :n] = np.asarray(L).astype(np.float32)
        Xp.append(xp)
        Lp.append(lp)
        ds.append([n, -1])
        ls.append([n, n])
    return Xp, Lp, np.asarray(ds, dtype=np.int32), np.asarray(ls, dtype=np.int32)


# ----------------------------------------------------------------------------
# a3  models/layers/graphconv.py:19-29,66-83; graphconv_reslap.py:21-43
def glorot_uniform(rs, shape):
    """graphconv.py:19-23: U(-r, r), r = sqrt(6 / (shape[0] + shape[1]))."""
    r = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rs.uniform(-r, r, size=shape).astype(np.float32)


def init_params(rs, Fin, Fout, K, variant=SGC_LL):
    """graphconv.py:75-83 / graphconv_reslap.py:30-43."""
    p = {
        "weight": glorot_uniform(rs, (Fin * K, Fout)),
        "bias": np.zeros((Fout,), dtype=np.float32),
        "M_L": glorot_uniform(rs, (Fin, Fin)),
        "alpha": np.ones((1,), dtype=np.float32),
    }
    if variant == SGC_LL_RESLAP:
        p["beta"] = np.ones((1,), dtype=np.float32)
    return p


# ----------------------------------------------------------------------------
# operators
def lrelu(x, alpha):
    """operators/model_operatos.py:534-558 with a tensor alpha (`alpha != 0.`
    is always true): relu(x) - alpha * relu(-x)."""
    return np.maximum(x, 0) - alpha * np.maximum(-x, 0)


def lrelu_grad(x, alpha):
    """d lrelu / dx with TF's relu gradient (1 where arg > 0, else 0):
    [x > 0] + alpha * [x < 0]   (0 at x == 0)."""
    return (x > 0).astype(x.dtype) + alpha * (x < 0).astype(x.dtype)


def clip_scale(T, average):
    """TF-0.12 clip_by_norm(T, 1) (average=False; graphconv_reslap.py:185,194)
    or clip_by_average_norm(T, 1) (average=True; graphconv.py:212):
        scale = min(rsqrt(sum(T*T)) [* numel], 1)
    Returns (scale, sumsq)."""
    dt = T.dtype.type
    ss = (T * T).sum(dtype=T.dtype)
    with np.errstate(divide="ignore"):
        inv = dt(1.0) / np.sqrt(ss)
    if average:
        inv = inv * dt(T.size)
    return np.minimum(inv, dt(1.0)), ss


def clip_backward(D, T, ss, average):
    """Gradient of T * min(q * rsqrt(ss), 1), q = numel (average) or 1.
    TF MinimumGrad sends the gradient to the norm branch when
    q * rsqrt(ss) <= 1; otherwise the scale is the constant 1."""
    dt = T.dtype.type
    q = dt(T.size) if average else dt(1.0)
    with np.errstate(divide="ignore"):
        inv = dt(1.0) / np.sqrt(ss)
    if q * inv <= dt(1.0):
        # y = q * T * ss^-1/2 ; dy/dT . D = q * (D * inv - T * <D,T> * inv^3)
        dot = (D * T).sum(dtype=T.dtype)
        return q * (D * inv - T * (dot * inv * inv * inv))
    return D


# ----------------------------------------------------------------------------
# a5  py_func body   graphconv.py:163-209  (twin graphconv_reslap.py:136-182)
def similarity(x, M):
    """graphconv.py:164-180 vectorised: Z = x.M; W_ij = exp(-||Z_i - Z_j||_2),
    W_ii = 0.  Returns (S, Z, r)."""
    Z = x @ M
    n = Z.shape[0]
    if n <= 256:
        diff = Z[:, None, :] - Z[None, :, :]
        r = np.sqrt((diff * diff).sum(axis=-1, dtype=Z.dtype))
    else:
        # the same arithmetic in blocks of rows (every pair's sum over the features is its own reduction along the last
        # axis: identical bits), so that the [rows, n, F] temporaries stay in cache at point-cloud sizes (n = 1024, F = 128 in
        # float64: 1 GB each otherwise)
        r = np.empty((n, n), dtype=Z.dtype)
        step = max(1, (1 << 21) // max(1, n * Z.shape[1]))
        for i0 in range(0, n, step):
            diff = Z[i0:i0 + step, None, :] - Z[None, :, :]
            r[i0:i0 + step] = np.sqrt((diff * diff).sum(axis=-1, dtype=Z.dtype))
    S = np.exp(-r)
    np.fill_diagonal(S, 0)
    return S.astype(x.dtype), Z, r


def similarity_loop(x, M):
    """graphconv.py:164-180 exactly as written (O(n^2) Python loop, one
    np.linalg.norm per pair).  Used for the reference-faithful CPU baseline."""
    x_w = np.dot(x, M)
    D = [[] for _ in range(x_w.shape[0])]
    for i in range(x_w.shape[0]):
        for j in range(x_w.shape[0]):
            if j == i:
                D[i].append(0.0)
                continue
            dist = np.linalg.norm(x_w[i] - x_w[j])
            D[i].append(1 * np.exp(-1 * dist))
    return np.asarray(D).astype(np.float32)


def residual_laplacian(S, mode):
    """graphconv.py:195-200 / graphconv_reslap.py:168-173.
    d = colsum(S) + spacing(0); s = d^-1/2.
    reference: L = I - diag(s) * S * diag(s) with numpy ELEMENTWISE `*` (== I).
    paper:     L = I - diag(s) . S . diag(s) (matrix products).
    Returns (R, s)."""
    n = S.shape[0]
    d = S.sum(axis=0, dtype=S.dtype)
    d = d + np.spacing(np.array(0, S.dtype))
    s = (1 / np.sqrt(d)).astype(S.dtype)
    I = np.identity(n, dtype=S.dtype)
    if mode == REFERENCE:
        D = np.diag(s)
        R = I - D * S * D
    else:
        R = I - (s[:, None] * S) * s[None, :]
    return R.astype(S.dtype), s


# ----------------------------------------------------------------------------
# a4-a10 forward for one graph
def forward_graph(x, L0, Lp, p, K, mode=REFERENCE, variant=SGC_LL, loop_similarity=False):
    """One graph through specgraph_LL (graphconv.py:145-251) or
    specgraph_LL_reslap (graphconv_reslap.py:119-229) + activation
    (graphconv.py:117-123).  x [n,Fin], L0 [n,n] (already sliced, a4),
    Lp [n,n] or None.  Returns a cache dict with every intermediate."""
    dt = x.dtype.type
    alpha = dt(p["alpha"][0])
    c = {}
    if loop_similarity:
        S = similarity_loop(x, p["M_L"])
        Z = r = None
    else:
        S, Z, r = similarity(x, p["M_L"])
    R, s = residual_laplacian(S, mode)
    reslap = variant == SGC_LL_RESLAP
    scR, ssR = clip_scale(R, average=not reslap)           # graphconv.py:212 / reslap:185
    R1 = R * scR
    R2 = lrelu(R1, alpha)                                  # graphconv.py:213 / reslap:186
    if not reslap:
        La = R2 + L0                                       # graphconv.py:216
        Lpre = Lc = None
        scL = ssL = None
    else:
        Lpre = R2 + L0                                     # reslap:193
        if Lp is not None:
            Lpre = Lpre + Lp * dt(p["beta"][0])            # reslap:190
        scL, ssL = clip_scale(Lpre, average=False)         # reslap:194
        Lc = Lpre * scL
        La = lrelu(Lc, alpha)                              # reslap:195
    # Chebyshev basis, graphconv.py:222-236
    Xk = [x]
    if K > 1:
        Xk.append(La @ x)
    for k in range(2, K):
        Xk.append(dt(2) * (La @ Xk[k - 1]) - Xk[k - 2])
    # graphconv.py:238-247: [K,n,Fin] -> [n,Fin,K] -> [n,Fin*K]; column = f*K + k
    C = np.stack(Xk, axis=2).reshape(x.shape[0], -1)
    Ypre = C @ p["weight"] + p["bias"]
    Y = np.maximum(Ypre, 0)                                # graphconv.py:118-123
    c.update(x=x, L0=L0, Lp=Lp, S=S, Z=Z, r=r, s=s, R=R, scR=scR, ssR=ssR, R1=R1, R2=R2,
             Lpre=Lpre, scL=scL, ssL=ssL, Lc=Lc, La=La, Xk=Xk, C=C, Ypre=Ypre, Y=Y)
    return c


# a11 backward for one graph (TF autodiff restated analytically)
def backward_graph(c, G, p, K, mode=REFERENCE, variant=SGC_LL, dLa_extra=None):
    """Gradients of l = <Y, G> (+ <La, dLa_extra> for the cross-layer Reslap
    edge: a later layer's dLp lands on this layer's L_sum output).
    Returns dict: dX, dweight, dbias, dM_L, dalpha, dbeta, dLp."""
    x = c["x"]
    dt = x.dtype.type
    n, Fin = x.shape
    alpha = dt(p["alpha"][0])
    reslap = variant == SGC_LL_RESLAP
    g = {}
    Gp = G * (c["Ypre"] > 0)
    g["dweight"] = c["C"].T @ Gp
    g["dbias"] = Gp.sum(axis=0, dtype=x.dtype)
    dC = Gp @ p["weight"].T
    gk = [np.ascontiguousarray(dC[:, k::K]) for k in range(K)]
    La, Xk = c["La"], c["Xk"]
    dLa = np.zeros_like(La)
    for k in range(K - 1, 1, -1):
        dLa += dt(2) * (gk[k] @ Xk[k - 1].T)
        gk[k - 1] = gk[k - 1] + dt(2) * (La.T @ gk[k])
        gk[k - 2] = gk[k - 2] - gk[k]
    if K > 1:
        dLa += gk[1] @ Xk[0].T
        gk[0] = gk[0] + La.T @ gk[1]
    dX = gk[0]
    if dLa_extra is not None:
        dLa = dLa + dLa_extra
    dalpha = dt(0)
    g["dbeta"] = None
    g["dLp"] = None
    if reslap:
        Lc, Lpre = c["Lc"], c["Lpre"]
        dalpha += (dLa * np.minimum(Lc, 0)).sum(dtype=x.dtype)
        dLc = dLa * lrelu_grad(Lc, alpha)
        dLpre = clip_backward(dLc, Lpre, c["ssL"], average=False)
        if c["Lp"] is not None:
            g["dbeta"] = np.array([(dLpre * c["Lp"]).sum(dtype=x.dtype)], dtype=x.dtype)
            g["dLp"] = dLpre * dt(p["beta"][0])
        dR2 = dLpre
    else:
        dR2 = dLa
    R1 = c["R1"]
    dalpha += (dR2 * np.minimum(R1, 0)).sum(dtype=x.dtype)
    g["dalpha"] = np.array([dalpha], dtype=x.dtype)
    g["dM_L"] = None
    if mode == PAPER:
        dR1 = dR2 * lrelu_grad(R1, alpha)
        dR = clip_backward(dR1, c["R"], c["ssR"], average=not reslap)
        S, s, Z, r = c["S"], c["s"], c["Z"], c["r"]
        dA = -dR                                   # A_ij = s_i S_ij s_j
        ds = (dA * S * s[None, :]).sum(axis=1, dtype=x.dtype) + \
             (dA * S * s[:, None]).sum(axis=0, dtype=x.dtype)
        # s = d^-1/2  =>  ds/dd = -1/2 d^-3/2.  A node whose similarities all underflow has d = spacing(0),
        # s^3 = inf and ds = 0: it is cut out of the degree gradient (dd = 0) instead of producing inf * 0 = NaN
        with np.errstate(over="ignore", invalid="ignore"):
            s3 = s * s * s
            dd = np.where(np.isfinite(s3) & (ds != 0), dt(-0.5) * s3 * ds, dt(0)).astype(x.dtype)
        with np.errstate(over="ignore", invalid="ignore"):
            dS = dA * s[:, None] * s[None, :] + dd[None, :]    # d_j = sum_i S_ij (inf where s_i s_j overflows: S = 0 there)
            dr = np.where(S == 0, dt(0), -S * dS).astype(x.dtype)   # S = exp(-r); S == 0 (diagonal, underflow): 0, never inf * 0
        with np.errstate(divide="ignore", invalid="ignore"):
            cij = np.where(r > 0, (dr + dr.T) / r, dt(0)).astype(x.dtype)
        np.fill_diagonal(cij, 0)
        # dZ_i = sum_j c_ij (Z_i - Z_j) = rowsum(c)_i Z_i - (c Z)_i ; defined 0 where r_ij == 0
        dZ = cij.sum(axis=1, dtype=x.dtype)[:, None] * Z - cij @ Z
        g["dM_L"] = x.T @ dZ
        dX = dX + dZ @ p["M_L"].T
    g["dX"] = dX
    return g


# ----------------------------------------------------------------------------
# batch level (what one layer call does: graphconv.py:85-125, reslap:45-89)
def layer_forward(X_list, L0_list, p, K, mode=REFERENCE, variant=SGC_LL, Lp_list=None,
                  loop_similarity=False):
    """X_list / L0_list: per-graph UNPADDED arrays ([n,Fin], [n,n]).
    Returns list of caches (Y, R2 == res_L, S == res_W, La == L_sum inside)."""
    out = []
    for b in range(len(X_list)):
        Lp = None if not Lp_list else Lp_list[b]
        out.append(forward_graph(X_list[b], L0_list[b], Lp, p, K, mode, variant, loop_similarity))
    return out


def layer_backward(caches, G_list, p, K, mode=REFERENCE, variant=SGC_LL, dLa_extra_list=None):
    """Sums parameter gradients over the batch in graph order."""
    tot = {"dweight": 0, "dbias": 0, "dalpha": 0, "dM_L": None, "dbeta": None}
    dX, dLp = [], []
    for b, c in enumerate(caches):
        ex = None if dLa_extra_list is None else dLa_extra_list[b]
        g = backward_graph(c, G_list[b], p, K, mode, variant, ex)
        dX.append(g["dX"])
        dLp.append(g["dLp"])
        for k in ("dweight", "dbias", "dalpha"):
            tot[k] = tot[k] + g[k]
        for k in ("dM_L", "dbeta"):
            if g[k] is not None:
                tot[k] = g[k] if tot[k] is None else tot[k] + g[k]
    tot["dX"] = dX
    tot["dLp"] = dLp
    return tot


def pad_rows(Y, max_atom):
    """graphconv.py:249-250: zero-pad rows to max_atom (pad rows stay 0 through relu)."""
    out = np.zeros((max_atom, Y.shape[1]), dtype=Y.dtype)
    out[:Y.shape[0]] = Y
    return out


# ----------------------------------------------------------------------------
# N1 readout  models/layers/graphgather.py:40-77
def graph_gather_forward(X_list, activation="tanh"):
    """Row-sum of each graph's real rows -> [B, F], then the activation
    (graphgather.py:68-77, :48)."""
    pre = np.stack([x.sum(axis=0, dtype=x.dtype) for x in X_list])
    if activation == "tanh":
        return np.tanh(pre), pre
    if activation == "relu":
        return np.maximum(pre, 0), pre
    if activation in (None, "linear"):
        return pre, pre
    raise ValueError(activation)


def graph_gather_backward(out, pre, G, ns, activation="tanh"):
    if activation == "tanh":
        gp = G * (1 - out * out)
    elif activation == "relu":
        gp = G * (pre > 0)
    else:
        gp = G
    return [np.repeat(gp[b:b + 1], n, axis=0) for b, n in enumerate(ns)]


# ----------------------------------------------------------------------------
# N1 / N2 / N4 "next" rows
def dense_mol_forward(X_pad, W, b):
    """models/layers/dense_layer.py:33-50: xx = matmul(xx, W) + b on the PADDED [max_atom, Fin] tensor
    (so pad rows come out as b); the stored activation is never applied."""
    return X_pad @ W + b


def dense_mol_backward(X_pad, W, G):
    """Returns (dX, dW, db) for one padded graph (sum the parameter gradients over the batch)."""
    return G @ W.T, X_pad.T @ G, G.sum(axis=0, dtype=G.dtype)


def block_end_forward(x, x_res, W, activation="relu"):
    """models/layers/blockend.py:67-86 on the sliced rows [n, F]: act(x_res . W + x) (pad rows are 0)."""
    pre = x_res @ W + x
    return (np.maximum(pre, 0) if activation == "relu" else pre), pre


def block_end_backward(x_res, W, pre, G, activation="relu"):
    """Returns (dx, dx_res, dW)."""
    Gp = G * (pre > 0) if activation == "relu" else G
    return Gp, Gp @ W.T, x_res.T @ Gp


def dense_block_end_forward(x, ins, W_in, outs, W_out, beta1, beta2, activation="relu"):
    """models/layers/densenet_block.py:99-131 on the sliced rows [n, F] of one graph:
    x + sum_l beta1 * (a_l . W_in_l) + sum_b beta2 * (o_b . W_out_b), accumulated in the reference's order
    (:113-123), then the activation (:92); pad rows are 0 (:126).  Returns (y, pre)."""
    b1, b2 = x.dtype.type(np.asarray(beta1).reshape(-1)[0]), x.dtype.type(np.asarray(beta2).reshape(-1)[0])
    pre = x.copy()
    for a, W in zip(ins, W_in):
        pre = pre + (a @ W) * b1
    for o, W in zip(outs, W_out):
        pre = pre + (o @ W) * b2
    return (np.maximum(pre, 0) if activation == "relu" else pre), pre


def dense_block_end_backward(ins, W_in, outs, W_out, beta1, beta2, pre, G, activation="relu"):
    """Returns (dx, [da_l], [dW_in_l], [do_b], [dW_out_b], dbeta1, dbeta2)."""
    b1, b2 = G.dtype.type(np.asarray(beta1).reshape(-1)[0]), G.dtype.type(np.asarray(beta2).reshape(-1)[0])
    Gp = G * (pre > 0) if activation == "relu" else G
    da = [b1 * (Gp @ W.T) for W in W_in]
    dWi = [b1 * (a.T @ Gp) for a in ins]
    do = [b2 * (Gp @ W.T) for W in W_out]
    dWo = [b2 * (o.T @ Gp) for o in outs]
    db1 = sum(((a @ W) * Gp).sum() for a, W in zip(ins, W_in))
    db2 = sum(((o @ W) * Gp).sum() for o, W in zip(outs, W_out))
    return Gp, da, dWi, do, dWo, db1, db2


def multitask_ce_forward(feat, W, b, labels, weights, batch_size):
    """Classifier tail: T heads logits_t = feat . W[t] + b[t]  (model_operatos.py:710-759 fully_connected_layer,
    :792-864 multitask_logits/logits) and the 'cross_entropy' training loss (multitask_classifier.py:41-44):
    per task sum_{b,c} w[b,t] * sce(logits_t[b,c], one_hot(y[b,t], 2)[c]), summed over tasks, divided by
    batch_size (:187-209).  sce(x, z) = max(x,0) - x z + log(1 + exp(-|x|)).
    feat [B,F], W [T,F,2], b [T,2], labels/weights [B,T].  Returns (loss, logits [T,B,2])."""
    T = W.shape[0]
    logits = np.stack([feat @ W[t] + b[t] for t in range(T)])
    y = labels.astype(np.int64)
    z = np.stack([(y == c) for c in range(2)], axis=-1).astype(feat.dtype)          # [B,T,2]; out of range -> zeros
    x = np.transpose(logits, (1, 0, 2))                                              # [B,T,2]
    sce = np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))
    loss = (sce * weights[:, :, None].astype(feat.dtype)).sum() / feat.dtype.type(batch_size)
    return loss, logits


def multitask_ce_backward(feat, W, logits, labels, weights, batch_size, dloss=1.0):
    """Returns (dfeat [B,F], dW [T,F,2], db [T,2])."""
    T = W.shape[0]
    y = labels.astype(np.int64)
    z = np.stack([(y == c) for c in range(2)], axis=-1).astype(feat.dtype)
    x = np.transpose(logits, (1, 0, 2))
    dx = (1.0 / (1.0 + np.exp(-x)) - z) * weights[:, :, None] * (dloss / batch_size)   # [B,T,2]
    dfeat = sum(dx[:, t, :] @ W[t].T for t in range(T))
    dW = np.stack([feat.T @ dx[:, t, :] for t in range(T)])
    db = np.stack([dx[:, t, :].sum(axis=0) for t in range(T)])
    return dfeat, dW, db


def graph_pool_forward(x, L):
    """models/layers/graphpool.py:91-103 as written: per node, max over the rows of x selected by the
    non-zeros of the node's Laplacian row; an empty row keeps x[i].  Returns (pooled, argmax)."""
    acc, arg = [], []
    for i, l in enumerate(list(L)):
        idx = np.nonzero(l)[0]
        if len(idx) != 0:
            rows = x[idx]
            acc.append(np.amax(rows, axis=0))
            arg.append(idx[np.argmax(rows, axis=0)])       # first maximum in index order
        else:
            acc.append(x[i])
            arg.append(np.full(x.shape[1], i))
    return np.vstack(acc), np.vstack(arg)


def graph_pool_backward(argmax, G):
    """Arg-max routing (extension: the reference's py_func passes no gradient)."""
    dX = np.zeros_like(G)
    n, F = G.shape
    for i in range(n):
        for f in range(F):
            dX[argmax[i, f], f] += G[i, f]
    return dX


# ----------------------------------------------------------------------------
# synthetic Tox21-shaped batches; used by tests and bench
ONE_HOT_BLOCKS = (44, 11, 5, 7)   # symbol, degree, numH, implicit valence (graph_features.py:160-179)


def synth_node_counts(rs, B, mu=np.log(16.0), sigma=0.5, lo=4, hi=132):
    return np.clip(np.rint(rs.lognormal(mu, sigma, size=B)), lo, hi).astype(np.int32)


def synth_adj_lists(rs, n):
    """Random spanning tree + round(0.1 n) ring-closing edges, max degree 4."""
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
            u = int(rs.randint(0, v))
            if link(u, v):
                break
        else:
            for u in range(v):
                if link(u, v):
                    break
    for _ in range(int(round(0.1 * n))):
        link(int(rs.randint(0, n)), int(rs.randint(0, n)))
    return [sorted(a) for a in adj]


def synth_atom_features(rs, n):
    """75-dim atom vectors: one 1 per one-hot block (44/11/5/7), 2 small ints,
    hybridisation one-hot (5), aromatic flag (utils/feature/graph_features.py:156-180).
    Symbol block is heavily skewed (C/N/O) so duplicate rows are common."""
    x = np.zeros((n, 75), dtype=np.float32)
    sym = rs.choice(44, size=n, p=_symbol_p())
    x[np.arange(n), sym] = 1
    off = 44
    for width, hi in ((11, 5), (5, 4), (7, 5)):
        idx = rs.randint(0, hi, size=n)
        x[np.arange(n), off + idx] = 1
        off += width
    x[:, off] = rs.randint(0, 2, size=n) * rs.choice([-1, 0, 0, 0, 1], size=n)   # formal charge
    x[:, off + 1] = 0                                                             # radical electrons
    off += 2
    x[np.arange(n), off + rs.randint(0, 3, size=n)] = 1
    off += 5
    x[:, off] = rs.randint(0, 2, size=n)
    return x


def _symbol_p():
    p = np.full(44, 0.05 / 41)
    p[0], p[1], p[2] = 0.72, 0.11, 0.12
    return p / p.sum()


def synth_batch(seed, B, Fin=75, mu=np.log(16.0), hi=132, features="atoms"):
    """Returns dict(ns, X_list, adj, L0_list (float32 dense), csr list)."""
    rs = np.random.RandomState(seed)
    ns = synth_node_counts(rs, B, mu=mu, hi=hi)
    X, adj, L0, csr = [], [], [], []
    for n in ns:
        a = synth_adj_lists(rs, int(n))
        if features == "atoms" and Fin == 75:
            x = synth_atom_features(rs, int(n))
        else:
            x = np.maximum(rs.randn(int(n), Fin), 0).astype(np.float32)
        indptr, indices, data, L = graph_laplacian(a)
        X.append(x)
        adj.append(a)
        L0.append(L.astype(np.float32))
        csr.append((indptr, indices, data))
    return {"ns": ns, "X": X, "adj": adj, "L0": L0, "csr": csr}
