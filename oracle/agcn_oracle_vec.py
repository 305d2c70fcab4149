"""All-core CPU restatement of the SGC_LL layer (torch-CPU, batch-vectorised).  TEST INFRASTRUCTURE ONLY.

Same rules as `agcn_oracle.py`: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import this file, as a checker / reported baseline, never as the product.

SURVEY.md section 8(d) asks for two CPU baselines timed on the GPU box's host: the
reference-faithful 1-core loop (`agcn_oracle.py`, `loop_similarity=True`) and a vectorised one that
uses ALL host cores ("CPU-vec", `torch.set_num_threads(os.cpu_count())`).  This is the second: the
ragged batch is zero-padded to [B, n_max, .] and every stage is a batched torch op (bmm / broadcast),
forward by the formulas of `agcn_oracle.forward_graph` (graphconv.py:163-251), backward by torch
autograd with the same conventions (d||.||/dz := 0 at r == 0; TF relu gradient 0 at 0).
It is checked against the per-graph numpy oracle in `tests/test_oracle_vec.py`.
"""
import numpy as np
import torch

REFERENCE, PAPER = 0, 1


def pad_batch(X_list, L0_list, G_list=None):
    """Per-graph arrays -> zero-padded batch tensors + node mask."""
    B = len(X_list)
    n = np.array([x.shape[0] for x in X_list], dtype=np.int64)
    nmax, Fin = int(n.max()), X_list[0].shape[1]
    X = np.zeros((B, nmax, Fin), np.float32)
    L0 = np.zeros((B, nmax, nmax), np.float32)
    mask = np.zeros((B, nmax), np.float32)
    for b in range(B):
        X[b, :n[b]] = X_list[b]
        L0[b, :n[b], :n[b]] = L0_list[b]
        mask[b, :n[b]] = 1
    out = {"X": torch.from_numpy(X), "L0": torch.from_numpy(L0), "mask": torch.from_numpy(mask), "n": n}
    if G_list is not None:
        G = np.zeros((B, nmax, G_list[0].shape[1]), np.float32)
        for b in range(B):
            G[b, :n[b]] = G_list[b]
        out["G"] = torch.from_numpy(G)
    return out


def params_to_torch(p):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).clone() for k, v in p.items()}


def _lrelu(x, alpha):
    # operators/model_operatos.py:534-558 (tensor alpha)
    return torch.relu(x) - alpha * torch.relu(-x)


def forward(bt, p, K, mode):
    """SGC_LL forward over the padded batch (graphconv.py:127-252), K == 2.
    Returns dict(Y, res_L, res_W); padded entries are 0."""
    assert K == 2
    X, L0, mask = bt["X"], bt["L0"], bt["mask"]
    B, nmax, Fin = X.shape
    pair = mask[:, :, None] * mask[:, None, :]
    eye = torch.eye(nmax, dtype=X.dtype)[None]
    Z = X @ p["M_L"]                                             # graphconv.py:164
    diff = Z[:, :, None, :] - Z[:, None, :, :]
    r2 = (diff * diff).sum(-1)
    pos = r2 > 0
    r = torch.where(pos, torch.sqrt(torch.where(pos, r2, torch.ones_like(r2))), torch.zeros_like(r2))
    S = torch.exp(-r) * (1 - eye) * pair                         # :168-180
    I = eye * pair
    if mode == PAPER:
        d = S.sum(1) + float(np.spacing(np.float32(0)))          # :195-196 (column sums)
        # (padded nodes: a finite stand-in inside the untaken branch keeps autograd free of 0 * inf)
        s = torch.where(mask > 0, 1.0 / torch.sqrt(torch.where(mask > 0, d, torch.ones_like(d))), torch.zeros_like(d))
        R = I - (s[:, :, None] * S) * s[:, None, :]
    else:
        R = I                                                    # elementwise D*W*D == 0 (:198-200)
    # clip_by_average_norm(R, 1) never clips (||R||_F <= n <= n^2): scale exactly 1 (:212)
    R2 = _lrelu(R, p["alpha"][0])                                # :213
    La = R2 + L0                                                 # :216
    X1 = La @ X                                                  # :231
    W = p["weight"]
    Ypre = X @ W[0::K] + X1 @ W[1::K] + p["bias"]               # :238-247 (column f*K + k)
    Y = torch.relu(Ypre) * mask[:, :, None]                      # :118-123, pad rows 0 (:249)
    return {"Y": Y, "res_L": R2, "res_W": S}


def step(bt, p, K, mode):
    """Forward + backward of l = <Y, G>; returns (outputs, grads dict)."""
    names = ["weight", "bias", "alpha"] + (["M_L"] if mode == PAPER else [])
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    X = bt["X"].clone().requires_grad_(True)
    b2 = dict(bt)
    b2["X"] = X
    out = forward(b2, q, K, mode)
    (out["Y"] * bt["G"]).sum().backward()
    grads = {"d" + k: v.grad for k, v in leaves.items()}
    grads["dX"] = X.grad
    return out, grads
