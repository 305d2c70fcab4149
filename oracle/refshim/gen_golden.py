"""Generate golden vectors by executing the reference's OWN layer classes
(unmodified, imported from /root/reference behind fake_tensorflow.py) and
its own graph_structure.py / graph_topology.py.

BUILD-CONTAINER ONLY.  Usage:
    PYTHONDONTWRITEBYTECODE=1 python oracle/refshim/gen_golden.py

Writes small .npz files (inputs + expected outputs, data only) into
tests/golden/.  The reference source itself never leaves /root/reference.

What each file pins (reference file:line):
  lap_*.npz      graph_structure.py:22-55,85-130   degree_list, CSR indptr/indices/data (f64)
  topo_*.npz     tf_modules/graph_topology.py:84-135  padded feeds + int32 slices
  sgc_*.npz      layers/graphconv.py:85-252          SGC_LL fwd + TF-autodiff grads
  reslap_*.npz   layers/graphconv_reslap.py:45-230   SGC_LL_Reslap (single layer and 2-layer chain)
  gather_*.npz   layers/graphgather.py:40-77         GraphGatherMol fwd + grads
  dense_mol / block_end / graph_pool .npz   layers/dense_layer.py, blockend.py, graphpool.py
  dense_block_end.npz  layers/densenet_block.py:48-131  DenseBlockEnd fwd + grads (incl. both betas)
  multitask_head.npz   operators/model_operatos.py:792-864 + tf_modules/multitask_classifier.py:41-44,187-209
                       12-task-style heads + weighted sigmoid cross-entropy, fwd + grads
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import load_reference  # noqa: E402
import agcn_oracle as orc  # noqa: E402  (only for the synthetic topology / feature generators)

OUT = os.environ.get("AGCN_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # (override: the regeneration test)


def _graphs(rs, ns, Fin, gs, atoms):
    graphs = []
    for n in ns:
        adj = orc.synth_adj_lists(rs, int(n))
        if atoms and Fin == 75:
            x = orc.synth_atom_features(rs, int(n))
        else:
            x = np.maximum(rs.randn(int(n), Fin), 0).astype(np.float32)
            if n >= 6:              # force duplicate rows (r_ij == 0 path)
                x[3] = x[1]
        graphs.append(gs.MolGraph(x, adj))
    return graphs


def _feed(ftf, topo_cls, graphs, Fin, B, max_atom):
    topo = topo_cls(n_feat=Fin, batch_size=B, max_atom=max_atom)
    fd = topo.batch_to_feed_dict(np.array(graphs, dtype=object))
    X = [np.asarray(fd[p], dtype=np.float32) for p in topo.get_atom_features_placeholder()]
    L = [np.asarray(fd[p], dtype=np.float32) for p in topo.get_laplacians_placeholder()]
    ds = np.asarray(fd[topo.get_dataslice_placeholders()])
    ls = np.asarray(fd[topo.get_lapslice_placeholders()])
    assert ds.dtype == np.int32 and ls.dtype == np.int32
    return X, L, ds, ls


def _pad_sq(t_list, ns, max_atom):
    out = np.zeros((len(ns), max_atom, max_atom), dtype=np.float32)
    for b, (t, n) in enumerate(zip(t_list, ns)):
        a = t.t.detach().numpy() if hasattr(t, "t") else np.asarray(t)
        assert a.shape == (n, n), (a.shape, n)
        out[b, :n, :n] = a
    return out


def _grad(v):
    return None if v.t.grad is None else v.t.grad.detach().numpy().copy()


def _param_overrides(rs, Fin, Fout, K, reslap, fancy):
    """fancy => non-trivial bias/alpha/beta so that every term is exercised;
    otherwise the reference's own initial values (bias 0, alpha = beta = 1)."""
    ov = {}
    if fancy:
        ov["bias"] = (0.1 * rs.randn(Fout)).astype(np.float32)
        ov["alpha"] = np.array([0.7], dtype=np.float32)
        if reslap:
            ov["beta"] = np.array([1.3], dtype=np.float32)
    return ov


def run_layer_case(ftf, layers, gs, topo_cls, name, variant, ns, Fin, Fout, K, seed, fancy,
                   atoms=False, chain=False):
    rs = np.random.RandomState(seed)
    B = len(ns)
    max_atom = int(max(ns)) + int(rs.randint(0, 4))
    graphs = _graphs(rs, ns, Fin, gs, atoms)
    Xn, Ln, ds, ls = _feed(ftf, topo_cls, graphs, Fin, B, max_atom)
    X = [ftf.Tensor(torch.from_numpy(x.copy()).requires_grad_(True)) for x in Xn]
    L = [ftf.Tensor(torch.from_numpy(l.copy())) for l in Ln]
    dsT, lsT = ftf.constant(ds), ftf.constant(ls)
    reslap = variant == "reslap"
    save = dict(variant=np.array(1 if reslap else 0), ns=np.asarray(ns, np.int32), K=np.array(K),
                Fin=np.array(Fin), Fout=np.array(Fout), max_atom=np.array(max_atom),
                X=np.stack(Xn), L=np.stack(Ln), data_slice=ds, lap_slice=ls, chain=np.array(int(chain)))

    ftf.reset_state(seed + 1000, _param_overrides(rs, Fin, Fout, K, reslap, fancy))
    if not reslap:
        l1 = layers.SGC_LL(Fout, Fin, B, activation="relu", K=K)
        inp = {"node_features": X, "original_laplacian": L, "data_slice": dsT, "lap_slice": lsT}
        out1 = l1(inp)
        assert len(out1) == 3
        Y, resL, resW = out1
        Lsum = None
    else:
        l1 = layers.SGC_LL_Reslap(Fout, Fin, B, activation="relu", K=K, save_lap=True)
        inp = {"node_features": X, "original_laplacian": L, "data_slice": dsT, "lap_slice": lsT,
               "res_lap": [][-B:]}            # tf_graphs.py:184 on an empty stack
        out1 = l1(inp)
        assert len(out1) == 4
        Y, resL, resW, Lsum = out1

    def dump(prefix, layer, Yl, rL, rW, Ls):
        for k, v in layer.vars.items():
            save[prefix + "p_" + k] = v.t.detach().numpy().copy()
        save[prefix + "Y"] = np.stack([y.t.detach().numpy() for y in Yl])
        save[prefix + "res_L"] = _pad_sq(rL, ns, max_atom)
        save[prefix + "res_W"] = _pad_sq(rW, ns, max_atom)
        if Ls is not None:
            save[prefix + "L_sum"] = _pad_sq(Ls, ns, max_atom)

    dump("l1_", l1, Y, resL, resW, Lsum)
    final = Y
    if chain:
        ftf.reset_state(seed + 2000, _param_overrides(rs, Fout, Fout, K, True, fancy))
        l2 = layers.SGC_LL_Reslap(Fout, Fout, B, activation="relu", K=K)
        inp2 = {"node_features": Y, "original_laplacian": L, "data_slice": dsT, "lap_slice": lsT,
                "res_lap": list(Lsum)[-B:]}   # tf_graphs.py:184,193-194
        Y2, resL2, resW2, Lsum2 = l2(inp2)
        dump("l2_", l2, Y2, resL2, resW2, Lsum2)
        final = Y2

    G = rs.randn(B, max_atom, Fout).astype(np.float32)
    save["G"] = G
    loss = sum((y.t * torch.from_numpy(G[b])).sum() for b, y in enumerate(final))
    loss.backward()
    save["dX"] = np.stack([_grad(x) for x in X])
    for prefix, layer in (("l1_", l1),) + ((("l2_", l2),) if chain else ()):
        for k, v in layer.vars.items():
            g = _grad(v)
            save[prefix + "has_d_" + k] = np.array(int(g is not None))
            save[prefix + "d_" + k] = g if g is not None else np.zeros_like(v.t.detach().numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)
    print("wrote", name, "B=%d max_atom=%d" % (B, max_atom))


def run_lap_cases(gs):
    rs = np.random.RandomState(7)
    cases = {
        "ring6": [[(i - 1) % 6, (i + 1) % 6] for i in range(6)],
        "path4": [[1], [0, 2], [1, 3], [2]],
        "star5": [[1, 2, 3, 4], [0], [0], [0], [0]],
        "asym5": [[1], [2], [3], [4], []],              # one-directional lists: symmetrised by networkx
        "isolated6": [[1], [0], [], [4, 5], [3], [3]],  # node 2 has degree 0
        "selfloop4": [[0, 1], [0, 2], [1, 3], [2]],     # explicit self reference
        "tiny3": [[1], [0, 2], [1]],                    # n <= 3 -> no Laplacian
    }
    for n in (4, 5, 7, 18, 48, 132):
        cases["tree%d" % n] = orc.synth_adj_lists(rs, n)
    for name, adj in cases.items():
        n = len(adj)
        g = gs.MolGraph(np.zeros((n, 3), dtype=np.float32), adj)
        save = dict(n=np.array(n), has_lap=np.array(int(g.has_Lap)),
                    degree_list=g.degree_list,
                    adj_flat=np.array([j for a in adj for j in a], dtype=np.int64),
                    adj_ptr=np.cumsum([0] + [len(a) for a in adj]).astype(np.int64))
        if g.has_Lap:
            Lm = g.Laplacian
            assert Lm.dtype == np.float64
            Ls = Lm.copy()
            Ls.sort_indices()
            save.update(indptr=Lm.indptr, indices=Lm.indices, data=Lm.data,
                        sorted_indices=Ls.indices, sorted_data=Ls.data,
                        dense=np.asarray(Lm.todense()))
        np.savez_compressed(os.path.join(OUT, "lap_%s.npz" % name), **save)
    print("wrote %d lap cases" % len(cases))


def run_topo_case(ftf, gs, topo_cls):
    rs = np.random.RandomState(11)
    ns = [4, 9, 5, 17]
    graphs = _graphs(rs, ns, 75, gs, atoms=True)
    X, L, ds, ls = _feed(ftf, topo_cls, graphs, 75, len(ns), 20)
    np.savez_compressed(
        os.path.join(OUT, "topo_tox4.npz"), ns=np.asarray(ns, np.int32), max_atom=np.array(20),
        feat=np.concatenate([g.node_features for g in graphs]),
        lap_dense=np.concatenate([np.asarray(g.Laplacian.todense()).reshape(-1) for g in graphs]),
        X=np.stack(X), L=np.stack(L), data_slice=ds, lap_slice=ls)
    print("wrote topo_tox4")


def run_gather_case(ftf, layers):
    rs = np.random.RandomState(13)
    ns, F, max_atom = [4, 7, 5, 12, 6], 9, 14
    Xn = np.zeros((len(ns), max_atom, F), dtype=np.float32)
    for b, n in enumerate(ns):
        Xn[b, :n] = rs.randn(n, F)
        Xn[b, n:] = 5.0          # garbage in the pad rows must not reach the sum (DenseMol bias lands there)
    X = [ftf.Tensor(torch.from_numpy(Xn[b].copy()).requires_grad_(True)) for b in range(len(ns))]
    ds = np.asarray([[n, -1] for n in ns], dtype=np.int32)
    for act in ("tanh", "relu"):
        for x in X:
            x.t.grad = None
        layer = layers.GraphGatherMol(len(ns), activation=act)
        out = layer({"node_features": X, "data_slice": ftf.constant(ds)})
        G = rs.randn(len(ns), F).astype(np.float32)
        (out.t * torch.from_numpy(G)).sum().backward()
        np.savez_compressed(os.path.join(OUT, "gather_%s.npz" % act), ns=np.asarray(ns, np.int32),
                            X=Xn, data_slice=ds, out=out.t.detach().numpy(), G=G,
                            dX=np.stack([_grad(x) for x in X]))
    print("wrote gather cases")


def run_next_rows(ftf, layers, gs, topo_cls):
    """DenseMol (dense_layer.py:33-50), BlockEnd (blockend.py:40-86), GraphPoolMol (graphpool.py:38-110)."""
    rs = np.random.RandomState(21)
    ns, Fin, Fout = [5, 9, 4, 12], 7, 6
    B, max_atom = len(ns), 13
    graphs = _graphs(rs, ns, Fin, gs, atoms=False)
    Xn, Ln, ds, ls = _feed(ftf, topo_cls, graphs, Fin, B, max_atom)
    dsT, lsT = ftf.constant(ds), ftf.constant(ls)

    # DenseMol
    X = [ftf.Tensor(torch.from_numpy(x.copy()).requires_grad_(True)) for x in Xn]
    dm = layers.DenseMol(Fout, Fin, activation="relu")
    # non-zero bias through the shim's variable override (the layer creates its variables inside __call__)
    ftf.reset_state(5, {"%s_b" % dm.name: (0.1 * rs.randn(Fout)).astype(np.float32)})
    ref = dm({"node_features": X})
    G = rs.randn(B, max_atom, Fout).astype(np.float32)
    sum((o.t * torch.from_numpy(G[b])).sum() for b, o in enumerate(ref)).backward()
    np.savez_compressed(os.path.join(OUT, "dense_mol.npz"), ns=np.asarray(ns, np.int32), X=np.stack(Xn), data_slice=ds,
                        W=dm.W.t.detach().numpy(), b=dm.b.t.detach().numpy(), G=G,
                        Y=np.stack([o.t.detach().numpy() for o in ref]), dX=np.stack([_grad(x) for x in X]),
                        dW=_grad(dm.W), db=_grad(dm.b))

    # BlockEnd
    Fres = 5
    X = [ftf.Tensor(torch.from_numpy(x.copy()).requires_grad_(True)) for x in np.stack(Xn)[:, :, :Fout].copy()]
    Xn = np.stack(Xn)
    Rn = np.zeros((B, max_atom, Fres), dtype=np.float32)
    for b, n in enumerate(ns):
        Rn[b, :n] = rs.randn(n, Fres)
    R = [ftf.Tensor(torch.from_numpy(r.copy()).requires_grad_(True)) for r in Rn]
    ftf.reset_state(6)
    be = layers.BlockEnd(0, Fres, Fout, activation="relu", max_atom=max_atom, batch_size=B)
    out = list(be({"node_features": X, "data_slice": dsT, "block_outputs": R}))
    G = rs.randn(B, max_atom, Fout).astype(np.float32)
    sum((o.t * torch.from_numpy(G[b])).sum() for b, o in enumerate(out)).backward()
    np.savez_compressed(os.path.join(OUT, "block_end.npz"), ns=np.asarray(ns, np.int32), data_slice=ds,
                        X=np.stack([x.t.detach().numpy() for x in X]), R=Rn, W=be.vars["weight"].t.detach().numpy(),
                        G=G, Y=np.stack([o.t.detach().numpy() for o in out]), dX=np.stack([_grad(x) for x in X]),
                        dR=np.stack([_grad(r) for r in R]), dW=_grad(be.vars["weight"]))

    # GraphPoolMol (forward only: py_func passes no gradient)
    X = [ftf.Tensor(torch.from_numpy(x.copy()).requires_grad_(True)) for x in Xn]
    L = [ftf.Tensor(torch.from_numpy(l.copy())) for l in Ln]
    gp = layers.GraphPoolMol(B)
    out = gp({"node_features": X, "original_laplacian": L, "data_slice": dsT, "lap_slice": lsT})
    assert all(o.t.grad_fn is None for o in out), "GraphPoolMol must not carry a gradient in the reference"
    np.savez_compressed(os.path.join(OUT, "graph_pool.npz"), ns=np.asarray(ns, np.int32), X=np.stack(Xn),
                        L=np.stack(Ln), data_slice=ds, lap_slice=ls, Y=np.stack([o.t.detach().numpy() for o in out]))
    print("wrote dense_mol, block_end, graph_pool")


def run_tail_rows(ftf, layers):
    """DenseBlockEnd (densenet_block.py:48-131) and the classifier tail: multitask_logits
    (model_operatos.py:710-759,792-864) + the 'cross_entropy' training loss
    (multitask_classifier.py:41-44,187-209), both executed from the reference's own code."""
    import importlib
    import types
    rs = np.random.RandomState(31)
    ns, Fout = [5, 9, 4, 12], 6
    B, max_atom = len(ns), 13
    ds = np.stack([np.asarray(ns), -np.ones(B)], 1).astype(np.int32)        # graph_topology.py:90
    dsT = ftf.constant(ds)

    def padded(F):
        a = np.zeros((B, max_atom, F), dtype=np.float32)
        for b, n in enumerate(ns):
            a[b, :n] = rs.randn(n, F)
        return a

    def leaves(a):
        return [ftf.Tensor(torch.from_numpy(x.copy()).requires_grad_(True)) for x in a]

    # ---- DenseBlockEnd: x + beta1 * sum_l a_l W_in_l + beta2 * sum_b o_b W_out_b, relu, pad rows 0 ----
    in_dims, out_dims = [6, 5], [7]
    Xn = padded(Fout)
    An = [padded(F) for F in in_dims]
    On = [padded(F) for F in out_dims]
    X, A, O = leaves(Xn), [leaves(a) for a in An], [leaves(o) for o in On]
    ftf.reset_state(8, {"beta1": np.array([0.7], np.float32), "beta2": np.array([1.3], np.float32)})
    dbe = layers.DenseBlockEnd(0, in_dims, Fout, activation="relu", max_atom=max_atom, batch_size=B)
    out = list(dbe({"node_features": X, "data_slice": dsT,
                    "inblock_activations": [t for a in A for t in a], "inblock_activations_dim": in_dims,
                    "block_outputs": [t for o in O for t in o], "block_outputs_dim": out_dims}))
    G = rs.randn(B, max_atom, Fout).astype(np.float32)
    sum((o.t * torch.from_numpy(G[b])).sum() for b, o in enumerate(out)).backward()
    rec = dict(ns=np.asarray(ns, np.int32), data_slice=ds, X=Xn, G=G,
               Y=np.stack([o.t.detach().numpy() for o in out]), dX=np.stack([_grad(x) for x in X]),
               beta1=dbe.vars["beta_inblock"].t.detach().numpy(), beta2=dbe.vars["beta_outblock"].t.detach().numpy(),
               dbeta1=_grad(dbe.vars["beta_inblock"]), dbeta2=_grad(dbe.vars["beta_outblock"]),
               n_in=np.int32(len(in_dims)), n_out=np.int32(len(out_dims)))
    for i in range(len(in_dims)):
        rec["A%d" % i] = An[i]
        rec["dA%d" % i] = np.stack([_grad(t) for t in A[i]])
        rec["Win%d" % i] = dbe.vars["weight_inblock"][i].t.detach().numpy()
        rec["dWin%d" % i] = _grad(dbe.vars["weight_inblock"][i])
    for i in range(len(out_dims)):
        rec["O%d" % i] = On[i]
        rec["dO%d" % i] = np.stack([_grad(t) for t in O[i]])
        rec["Wout%d" % i] = dbe.vars["weight_outblock"][i].t.detach().numpy()
        rec["dWout%d" % i] = _grad(dbe.vars["weight_outblock"][i])
    np.savez_compressed(os.path.join(OUT, "dense_block_end.npz"), **rec)

    # ---- multitask heads + weighted sigmoid cross-entropy ---------------------------------------------
    # multitask_classifier.py imports matplotlib / the trainer base classes at module level: give it
    # empty stand-ins for THOSE (they are not on the path) and run its own get_loss_fn / add_training_loss.
    for name, attrs in (("AGCN.utils", {}), ("AGCN.utils.save", {"log": lambda *a, **k: None}),
                        ("AGCN.models.tf_modules.basic_model", {"Model": object}),
                        ("AGCN.models.tf_modules.evaluation", {"Evaluator": object})):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules.setdefault(name, m)
    pkg = types.ModuleType("AGCN.models.tf_modules")
    pkg.__path__ = [os.path.join(load_reference.REF_ROOT, "models", "tf_modules")]
    sys.modules.setdefault("AGCN.models.tf_modules", pkg)
    mc = importlib.import_module("AGCN.models.tf_modules.multitask_classifier")
    model_ops = importlib.import_module("AGCN.models.operators.model_operatos")

    Bh, F, T = 7, 10, 3
    feat_n = rs.randn(Bh, F).astype(np.float32)
    labels = rs.randint(0, 2, size=(Bh, T)).astype(np.float32)
    weights = (rs.rand(Bh, T) * (rs.rand(Bh, T) > 0.25)).astype(np.float32)      # some masked (0) entries
    feat = ftf.Tensor(torch.from_numpy(feat_n.copy()).requires_grad_(True))
    ftf.reset_state(9)
    # the reference's initialisers (stddev 0.01 weights, zero biases) make every logit ~0; swap the two
    # initialiser ops for larger random draws while ITS multitask_logits builds the heads
    tn, zr = ftf.truncated_normal, ftf.zeros
    ftf.truncated_normal = lambda shape, **k: ftf.Tensor(torch.from_numpy(
        (0.5 * rs.randn(*[int(v) for v in shape])).astype(np.float32)))
    ftf.zeros = lambda shape, **k: ftf.Tensor(torch.from_numpy(
        (0.3 * rs.randn(*[int(v) for v in shape])).astype(np.float32)))
    try:
        logits = model_ops.multitask_logits(feat, T, n_feature=F)
    finally:
        ftf.truncated_normal, ftf.zeros = tn, zr
    created = list(ftf.STATE.created)           # [(name, var)]: w, b per task in creation order
    assert len(created) == 2 * T
    me = types.SimpleNamespace(n_tasks=T, batch_size=Bh, label_placeholder=ftf.constant(labels),
                               weight_placeholder=ftf.constant(weights))
    loss = mc.MultitaskGraphClassifier.add_training_loss(me, "cross_entropy", logits)
    loss.t.backward()
    np.savez_compressed(os.path.join(OUT, "multitask_head.npz"), feat=feat_n, labels=labels, weights=weights,
                        W=np.stack([created[2 * t][1].t.detach().numpy() for t in range(T)]),
                        b=np.stack([created[2 * t + 1][1].t.detach().numpy() for t in range(T)]),
                        logits=np.stack([l.t.detach().numpy() for l in logits]), loss=loss.t.detach().numpy(),
                        dfeat=_grad(feat), dW=np.stack([_grad(created[2 * t][1]) for t in range(T)]),
                        db=np.stack([_grad(created[2 * t + 1][1]) for t in range(T)]))
    print("wrote dense_block_end, multitask_head")


def main():
    assert load_reference.available(), "reference tree not found"
    os.makedirs(OUT, exist_ok=True)
    ftf, layers, gs = load_reference.load()
    import importlib.util
    # tf_modules/__init__.py drags in h5py/rdkit; load the one file directly instead
    spec = importlib.util.spec_from_file_location(
        "agcn_ref_graph_topology",
        os.path.join(load_reference.REF_ROOT, "models", "tf_modules", "graph_topology.py"))
    topo_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(topo_mod)
    topo_cls = topo_mod.GraphTopologyMol

    run_lap_cases(gs)
    run_topo_case(ftf, gs, topo_cls)
    run_gather_case(ftf, layers)
    run_next_rows(ftf, layers, gs, topo_cls)
    run_tail_rows(ftf, layers)
    if os.environ.get("AGCN_GOLDEN_ONLY") == "tail":
        return

    L = (ftf, layers, gs, topo_cls)
    #                 name                 variant   ns                 Fin Fout K seed fancy
    run_layer_case(*L, "sgc_small",        "sgc",    [4, 5, 7],          5,  4, 2, 1, False)
    run_layer_case(*L, "sgc_small_fancy",  "sgc",    [4, 5, 7, 6],       5,  4, 2, 2, True)
    run_layer_case(*L, "sgc_k1",           "sgc",    [5, 9],             6,  8, 1, 3, True)
    run_layer_case(*L, "sgc_k3",           "sgc",    [6, 11, 4],         7,  5, 3, 4, True)
    run_layer_case(*L, "sgc_tox_l1",       "sgc",    [18, 48, 7, 23],   75, 64, 2, 5, False, atoms=True)
    run_layer_case(*L, "sgc_tox_l3",       "sgc",    [18, 33, 5],      128, 128, 2, 6, True)
    run_layer_case(*L, "sgc_n132",         "sgc",    [132, 4],          75, 64, 2, 7, True, atoms=True)
    run_layer_case(*L, "reslap_small",     "reslap", [4, 5, 7],          5,  4, 2, 8, False)
    run_layer_case(*L, "reslap_fancy",     "reslap", [6, 9, 4, 13],      7,  6, 2, 9, True)
    run_layer_case(*L, "reslap_k1",        "reslap", [5, 8],             6,  4, 1, 10, True)
    run_layer_case(*L, "reslap_tox",       "reslap", [18, 48, 7],       75, 64, 2, 11, True, atoms=True)
    run_layer_case(*L, "reslap_chain",     "reslap", [4, 9, 6, 12],      7,  6, 2, 12, True, chain=True)
    run_layer_case(*L, "reslap_chain_tox", "reslap", [18, 30, 5],       75, 64, 2, 13, False, atoms=True,
                   chain=True)
    # BASELINE.json config 1: "Tox21 batch=32, K=2 Chebyshev, repo's TF CPU path" -- a whole 32-graph batch of the
    # synthetic Tox21-shaped generator's sizes (lognormal(ln 16, 0.5) clipped to [4, 132], seed 123)
    ns32 = np.clip(np.rint(np.random.RandomState(123).lognormal(np.log(16.0), 0.5, size=32)), 4, 132).astype(int)
    run_layer_case(*L, "sgc_tox_b32",      "sgc",    [int(v) for v in ns32], 75, 64, 2, 14, False, atoms=True)


if __name__ == "__main__":
    main()
