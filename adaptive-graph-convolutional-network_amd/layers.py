"""Host-side mirror of the reference's layer classes for the hot path.

Class names, constructor arguments, the dict-in / tuple-of-lists-out call contract and the
read-back attributes are those of `models/layers/graphconv.py` (SGC_LL),
`graphconv_reslap.py` (SGC_LL_Reslap), `graphgather.py` (GraphGatherMol) and
`basic_layer.py` (Layer), because the reference's containers dispatch on
`type(layer).__name__` (models/tf_modules/tf_graphs.py:41,45,56,173,178).

Two input forms are accepted for `x['node_features']`:
  * the reference feed: a list of B [max_atom, Fin] tensors (+ 'original_laplacian' list of
    [max_atom, max_atom], 'data_slice' [B,2] rows [n,-1], 'lap_slice' [B,2] rows [n,n]);
  * the output of a previous layer of this package (a lazy `RowViews` over packed device
    memory), which keeps the batch packed on the device between layers.
Returned lists are lazy sequences of per-graph views (`RowViews` / `TileViews`).

Extension over the reference: `laplacian_mode` ("reference" = what the repository computes,
res_L == I and no gradient to M_L; "paper" = I - D^-1/2 W D^-1/2, differentiable).
"""
from collections import defaultdict

import os

import numpy as np
import torch

from .batch import GraphBatch, RowViews, TileViews
from .functional import (CastRowsFunction, GraphGatherFunction, GraphPoolFunction, MultitaskCEFunction,
                         PackRowsFunction, RowsDenseFunction, SgcLLFunction, _Cfg)

_UID_PREFIXES = defaultdict(int)


def _in32(X):
    """bf16-stored features are widened once at the layer input; every kernel computes in fp32."""
    return CastRowsFunction.apply(X, torch.float32) if X.dtype == torch.bfloat16 else X


def _out_as(Y, dtype):
    """... and the layer output goes back to the storage dtype the features came in."""
    return CastRowsFunction.apply(Y, torch.bfloat16) if dtype == torch.bfloat16 else Y


def get_uid(prefix=""):
    """models/operators/model_operatos.py:156-168."""
    _UID_PREFIXES[prefix] += 1
    return _UID_PREFIXES[prefix]


# AGCN_DEBUG_CHECKS=1 / set_debug_checks(True): symmetry vouches are verified on every call, not once per vouched object
_DEBUG_CHECKS = [os.environ.get("AGCN_DEBUG_CHECKS", "0") not in ("", "0")]


def set_debug_checks(on):
    """Verify caller-vouched symmetry flags on EVERY layer call (a launch + a 4-byte read each), not only on a layer object's
    first call.  Returns the previous setting."""
    prev = _DEBUG_CHECKS[0]
    _DEBUG_CHECKS[0] = bool(on)
    return prev


class Layer(object):
    """Base layer: naming + kwargs whitelist of models/layers/basic_layer.py:44-79."""

    def __init__(self, **kwargs):
        if not hasattr(self, "uses_learning_phase"):
            self.uses_learning_phase = False
        if not hasattr(self, "losses"):
            self.losses = []
        allowed_kwargs = {"input_shape", "batch_input_shape", "input_dtype", "name", "trainable"}
        for kwarg in kwargs.keys():
            if kwarg not in allowed_kwargs:
                raise TypeError("Keyword argument not understood:", kwarg)
        name = kwargs.get("name")
        if not name:
            prefix = self.__class__.__name__.lower()
            name = prefix + "_" + str(get_uid(prefix))
        self.name = name
        self.trainable = kwargs.get("trainable", True)

    def call(self, x):
        return x

    def __call__(self, x):
        return self.call(x)

    # the reference's learning phase is a global switch (model_operatos.py:46-61, in_train_phase); here it is a flag on
    # every layer, flipped per layer or -- containers -- for a whole network
    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)


def glorot(shape, generator=None, device=None):
    """graphconv.py:19-23: U(-r, r), r = sqrt(6 / (fan_in + fan_out))."""
    r = float(np.sqrt(6.0 / (shape[0] + shape[1])))
    t = torch.empty(shape, dtype=torch.float32)
    t.uniform_(-r, r, generator=generator)
    return t.to(device) if device is not None else t


def _resolve_batch(x):
    """Returns (GraphBatch, X2d) for either input form."""
    nf = x["node_features"]
    gb = x.get("graph_batch")
    if isinstance(nf, RowViews):
        if gb is None:
            gb = getattr(nf, "batch", None)
        if gb is None:
            raise ValueError("packed node_features need x['graph_batch']")
        return gb, nf.data
    if gb is not None and torch.is_tensor(nf) and nf.dim() == 2:
        return gb, nf
    # reference feed (lists of padded tensors)
    gb = GraphBatch.from_padded(nf, x["original_laplacian"], x["data_slice"])
    ls = x.get("lap_slice")
    if ls is not None:
        ls = ls.detach().cpu().numpy() if torch.is_tensor(ls) else np.asarray(ls)
        if not (np.array_equal(ls[:, 0], gb.plan.n) and np.array_equal(ls[:, 1], gb.plan.n)):
            raise ValueError("lap_slice does not agree with data_slice")
    return gb, gb.X


def _pack_tiles(res_lap, plan, device):
    """res_lap (list of B [n,n] tensors, or TileViews) -> packed tile tensor or None."""
    if res_lap is None or len(res_lap) == 0:     # `if early_Laps:` graphconv_reslap.py:188
        return None
    if isinstance(res_lap, TileViews) and res_lap.plan is plan:
        return res_lap.data
    if len(res_lap) != plan.B:
        raise ValueError("res_lap must hold one Laplacian per graph")
    out = torch.zeros(plan.tile_elems, dtype=torch.float32, device=device)
    for b, t in enumerate(res_lap):
        n, off = int(plan.n[b]), int(plan.tile_off[b])
        out[off:off + n * n] = t.reshape(-1)
    return out


class SGC_LL(Layer):
    """Spectral graph convolution with a learned residual Laplacian (graphconv.py:32-252).

    SGC_LL(output_dim, input_dim, batch_size, activation='relu', dropout=None, K=2,
           save_lap=False, save_output=False, **kwargs)
    call(x) -> (activated_nodes, res_L_updated, res_W_updated)
    """
    _variant = "SGC_LL"

    def __init__(self, output_dim, input_dim, batch_size, activation="relu", dropout=None, K=2,
                 save_lap=False, save_output=False, laplacian_mode="reference", device=None,
                 generator=None, **kwargs):
        super(SGC_LL, self).__init__(**kwargs)
        if dropout is not None and not (0.0 <= float(dropout) < 1.0):
            raise ValueError("dropout is the fraction of activations to drop, in [0, 1)")
        if activation not in ("relu", "linear", None):
            raise ValueError("Invalid activation function: %s" % (activation,))
        if laplacian_mode not in ("reference", "paper"):
            raise ValueError("laplacian_mode must be 'reference' or 'paper'")
        self.dropout = dropout
        self.training = True        # the learning phase of the reference (model_operatos.in_train_phase): dropout only
        # (model_operatos.py:46-61, dropout.py:34-41: a layer with 0 < p < 1 depends on the learning phase)
        self.uses_learning_phase = bool(dropout) and 0.0 < float(dropout) < 1.0
        self.activation = activation
        self.batch_size = batch_size
        self.nb_filter = output_dim
        self.n_atom_feature = input_dim
        self.vars = {}
        self.bias = True
        self.K = K
        self.save_lap = save_lap
        self.early_laps = None
        self.save_output = save_output
        self.laplacian_mode = laplacian_mode
        self._device = device
        self._generator = generator
        self._built = False

    def build(self, device=None):
        """Creates the parameters once (the reference re-creates them on every call,
        graphconv.py:66-83,92; same names in `.vars`)."""
        if self._built:
            return
        device = self._device if self._device is not None else device
        g = self._generator
        req = bool(self.trainable)
        Fin, K, Fout = self.n_atom_feature, self.K, self.nb_filter
        self.vars["weight"] = glorot([Fin * K, Fout], g, device).requires_grad_(req)
        self.vars["bias"] = torch.zeros([Fout], dtype=torch.float32, device=device).requires_grad_(req)
        self.vars["M_L"] = glorot([Fin, Fin], g, device).requires_grad_(req)
        self.vars["alpha"] = torch.ones((1,), dtype=torch.float32, device=device).requires_grad_(req)
        if self._variant == "SGC_LL_Reslap":
            self.vars["beta"] = torch.ones((1,), dtype=torch.float32, device=device).requires_grad_(req)
        self._built = True

    def parameters(self):
        self.build()
        return list(self.vars.values())

    def load_vars(self, arrays):
        """Set parameters from numpy arrays / tensors (same keys as `.vars`)."""
        dev = self._device
        for k, v in arrays.items():
            t = torch.as_tensor(np.asarray(v), dtype=torch.float32) if not torch.is_tensor(v) else v.detach().clone()
            self.vars[k] = t.to(dev).requires_grad_(bool(self.trainable))
        self._built = True

    def _apply(self, x, L_prev, prev_sym=False):
        gb, X = _resolve_batch(x)
        self.build(X.device)
        cfg = _Cfg(self.n_atom_feature, self.nb_filter, self.K, self.laplacian_mode, self._variant,
                   "relu" if self.activation == "relu" else "linear", lsum=self._variant == "SGC_LL_Reslap",
                   prev_sym=prev_sym and L_prev is not None)
        if gb.plan.B != self.batch_size:
            raise ValueError("batch holds %d graphs, layer was built for batch_size=%d" % (gb.plan.B, self.batch_size))
        v = self.vars
        if X.dtype == torch.bfloat16 and self._bf16_served(gb, X, cfg):
            # bf16 features straight into (and out of) the specialised kernels: no conversion launches (include/agcn.h:
            # AgcnLayerDesc.feat_dtype); everything between the loads and the stores is fp32
            cfgb = _Cfg(cfg.Fin, cfg.Fout, cfg.K, cfg.mode, cfg.variant, cfg.activation, lsum=cfg.lsum, prev_sym=cfg.prev_sym,
                        bf16=True)
            self._verify_vouches(gb, L_prev, cfg.prev_sym)
            Y, res_L, res_W, L_sum = SgcLLFunction.apply(X, v["weight"], v["bias"], v["M_L"], v["alpha"], v.get("beta"), L_prev, gb,
                                                         cfgb)
            if self.dropout and self.training:
                Y = torch.nn.functional.dropout(Y, p=float(self.dropout), training=True)
            nodes = RowViews(Y, gb.plan, gb)
            sum_sym = bool(gb.l0_symmetric) and (L_prev is None or cfg.prev_sym)
            outs = (TileViews(res_L, gb.plan, True), TileViews(res_W, gb.plan, True), TileViews(L_sum, gb.plan, sum_sym))
            for t in outs:
                t._verified = True
            return (gb, nodes) + outs
        X32, W, M = _in32(X), v["weight"], v["M_L"]
        pad = self._point_cloud_pad(gb, X32, cfg)
        if pad:
            # Narrow layers on point clouds (the first layers of the reference's point-cloud network, xyz -> n_filters_1:
            # Point_AGCN.py:37-46) run on the point-cloud kernels of the 64-feature width with zero columns appended to X,
            # zero rows / columns to M_L and zero rows to W (row f * K + k: the rows of the features behind Fin come last).
            # Zero features change nothing: Z, the distances, X_1's columns and every gradient keep their values, and
            # autograd's adjoint of the padding hands back exactly the un-padded blocks.
            F = torch.nn.functional
            X32 = F.pad(X32, (0, pad))
            W = F.pad(W, (0, 0, 0, pad * self.K))
            M = F.pad(M, (0, pad, 0, pad))
            cfg = _Cfg(self.n_atom_feature + pad, self.nb_filter, self.K, self.laplacian_mode, self._variant,
                       cfg.activation, lsum=cfg.lsum, prev_sym=cfg.prev_sym)
        self._verify_vouches(gb, L_prev, cfg.prev_sym)
        Y, res_L, res_W, L_sum = SgcLLFunction.apply(X32, W, v["bias"], M, v["alpha"], v.get("beta"), L_prev, gb, cfg)
        if self.dropout and self.training:
            # graphconv.py:121-122 / dropout.py:34-41: Dropout(p) on the activated nodes in the training phase, i.e.
            # tf.nn.dropout(x, keep_prob = 1 - p): kept entries scaled by 1 / (1 - p).  (No reference network enables it;
            # an element-wise torch op behind the layer, differentiable, not part of the fused kernels.)
            Y = torch.nn.functional.dropout(Y, p=float(self.dropout), training=True)
        nodes = RowViews(_out_as(Y, X.dtype), gb.plan, gb)
        # res_W and res_L are symmetric by construction; the summed Laplacian when L_orig and L_prev are
        sum_sym = bool(gb.l0_symmetric) and (L_prev is None or cfg.prev_sym)
        outs = (TileViews(res_L, gb.plan, True), TileViews(res_W, gb.plan, True), TileViews(L_sum, gb.plan, sum_sym))
        for t in outs:
            t._verified = True           # (derived from inputs that were computed symmetric or have been verified)
        return (gb, nodes) + outs

    def _verify_vouches(self, gb, L_prev, prev_sym):
        """A symmetry flag somebody VOUCHED for -- `GraphBatch(..., l0_dense=..., l0_symmetric=True)`, `TileViews(...,
        symmetric=True)` / `x["res_lap_symmetric"]` -- is checked on the device ONCE PER VOUCHED OBJECT
        (agcn_tiles_asymmetry: one launch + one 4-byte read; the object then carries `_verified`, so a static batch replayed
        step after step costs nothing after its first call and a NEW vouched batch / tile set is always checked), and on
        every call under AGCN_DEBUG_CHECKS=1 / `agcn_amd.set_debug_checks(True)`: the point-cloud kernels take La^T g_1 as La g_1 on the strength of those flags,
        so a false vouch would be a silently wrong dX.  Flags this package computed itself are not re-checked (outside the debug
        mode); a vouch that comes as `x["res_lap_symmetric"]` beside a plain list of tiles has no object to remember the
        verdict on and is checked on every call -- hand in a `TileViews`."""
        todo = []
        laps = getattr(self, "early_laps", None)
        dbg = _DEBUG_CHECKS[0]
        if prev_sym and L_prev is not None and (dbg or not getattr(laps, "_verified", False)):
            todo.append(("the previous Laplacian (res_lap, vouched symmetric)", L_prev, 0, laps))
        if gb.l0_symmetric and gb.l0_dense is not None and (dbg or not gb._l0_verified):
            todo.append(("the dense original Laplacian (l0_symmetric=True)", gb.l0_dense, gb.l0_ld, gb))
        if not todo:
            return
        if torch.cuda.is_current_stream_capturing():
            from . import _ffi
            raise _ffi.AgcnError("a vouched-symmetric object (%s) reached a layer for the first time inside a stream capture; "
                                 "the check reads a word back from the device: run the step once before capturing it"
                                 % todo[0][0])
        from . import _ffi
        from .functional import _stream
        dev = gb.plan.dev.device
        counts = torch.zeros(len(todo), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            for k, (_, t, ld, _) in enumerate(todo):
                _ffi.check(_ffi.lib().agcn_tiles_asymmetry(gb.plan.host_ptr, gb.plan.dev_ptr, t.data_ptr(), int(ld),
                                                          counts[k:].data_ptr(), _stream(dev)))
        bad = counts.tolist()
        for (what, _, _, obj), c in zip(todo, bad):
            if c:
                raise _ffi.AgcnError("%s is NOT symmetric: %d element pair(s) differ from their transposes; the flag must be "
                                     "dropped (the layer then takes the general kernels)" % (what, c))
            if obj is gb:
                gb._l0_verified = True
            elif obj is not None and hasattr(obj, "_verified"):
                obj._verified = True

    def _bf16_served(self, gb, X, cfg):
        """True when the library's kernels for this batch read / write bf16 features themselves (dispatch bit 256: every graph
        on the specialised kernels).  Host logic only, cached on the plan."""
        if self._variant != "SGC_LL" or not X.is_cuda or X.shape[1] != cfg.Fin:
            return False
        from . import _ffi
        from .functional import _desc, _gbatch, current_flags
        lib = _ffi.lib()
        key = ("bf16_served", cfg.Fin, cfg.Fout, cfg.K, cfg.mode, lib.agcn_get_lds_limit(), lib.agcn_get_fused_mode(),
               lib.agcn_get_spec_mode(), lib.agcn_get_slab_mode(), current_flags())
        cache = gb.plan.__dict__.setdefault("_size_cache", {})
        hit = cache.get(key)
        if hit is None:
            out = (_ffi.C.c_int32 * 8)()
            desc = _desc(cfg, X.shape[1], cfg.Fout)
            _ffi.check(lib.agcn_sgc_ll_dispatch_info(_ffi.C.byref(desc), _ffi.C.byref(_gbatch(gb, X, None)), out, 8))
            hit = cache[key] = bool(out[0] & 256)
        return hit

    def _point_cloud_pad(self, gb, X, cfg):
        """Feature columns to append so that the layer takes the point-cloud kernels (csrc/cloud_kernels.hip, 64 input
        features), or 0.  Asked of the library's own dispatch logic: only batches that would otherwise fall to the general
        slab kernels or the one-workgroup-per-graph fallback, and that have the shape the point-cloud kernels take (K = 2,
        dense symmetric L_orig on packed rows, every graph a multiple of 64 nodes)."""
        Fin = self.n_atom_feature
        if self.K != 2 or Fin >= 64 or gb.l0_dense is None or not gb.l0_symmetric or gb.l0_rowptr is not None:
            return 0
        if self._variant == "SGC_LL" and Fin <= 4:
            return 0                      # the streaming small-feature kernels are faster there
        n = np.asarray(gb.plan.n_host if hasattr(gb.plan, "n_host") else gb.plan.n)
        if n.size == 0 or np.any(n % 64 != 0):
            return 0
        from . import _ffi
        from .functional import _desc, _gbatch
        out = (_ffi.C.c_int32 * 8)()
        desc = _desc(cfg, X.shape[1], cfg.Fout)
        # (host logic only; whether a previous Laplacian comes in changes nothing unless it is not known to be symmetric)
        if self._variant == "SGC_LL_Reslap" and getattr(self, "_prev_unsym", False):
            return 0
        _ffi.check(_ffi.lib().agcn_sgc_ll_dispatch_info(_ffi.C.byref(desc), _ffi.C.byref(_gbatch(gb, X, None)), out, 8))
        return 64 - Fin if out[0] in (16, 32) else 0

    def call(self, x):
        _, nodes, res_L, res_W, _ = self._apply(x, None)
        return nodes, res_L, res_W


class SGC_LL_Reslap(SGC_LL):
    """SGC_LL plus beta * (previous layer's summed Laplacian) (graphconv_reslap.py:15-230).
    call(x) -> (activated_atoms, res_L_updated, res_W_updated, L_sum_updated)."""
    _variant = "SGC_LL_Reslap"

    def call(self, x):
        gb, X = _resolve_batch(x)
        self.early_laps = x["res_lap"]
        L_prev = _pack_tiles(self.early_laps, gb.plan, X.device)
        # symmetric previous Laplacians (this package's own outputs carry the flag; a caller may vouch with
        # x["res_lap_symmetric"] = True): the point-cloud kernels need to know, everything else ignores it
        prev_sym = bool(x.get("res_lap_symmetric", getattr(self.early_laps, "symmetric", False)))
        self._prev_unsym = L_prev is not None and not prev_sym
        x2 = dict(x)
        x2["graph_batch"], x2["node_features"] = gb, X
        try:
            _, nodes, res_L, res_W, L_sum = self._apply(x2, L_prev, prev_sym)
        finally:
            # (the attribute mirrors graphconv_reslap.py:68 for the duration of the call only: kept, it would hold the PREVIOUS
            #  layer's autograd graph -- and with it that layer's AccumulateGrad nodes -- alive until this layer's next call,
            #  across iterations and streams: a whole-network step then could not be captured into a HIP graph)
            self.early_laps = None
        return nodes, res_L, res_W, L_sum


class GraphGatherMol(Layer):
    """Per-graph row-sum readout + activation (graphgather.py:11-77).
    GraphGatherMol(batch_size, activation='relu'); call(x) -> tensor [batch_size, n_feat]."""

    def __init__(self, batch_size, activation="relu", **kwargs):
        super(GraphGatherMol, self).__init__(**kwargs)
        if activation not in ("relu", "tanh", "linear", None):
            raise ValueError("Invalid activation function: %s" % (activation,))
        self.activation = activation
        self.batch_size = batch_size

    def build(self, input_shape=None):
        pass

    def get_output_shape_for(self, input_shape):
        return self.batch_size, input_shape[0][1]

    def call(self, x, mask=None):
        nf = x["node_features"]
        gb = x.get("graph_batch")
        if isinstance(nf, RowViews):
            gb = gb if gb is not None else getattr(nf, "batch", None)
            X = nf.data
        elif gb is not None and torch.is_tensor(nf):
            X = nf
        else:
            X3 = nf if torch.is_tensor(nf) else torch.stack(list(nf))
            ds = x["data_slice"]
            ds = ds.detach().cpu().numpy() if torch.is_tensor(ds) else np.asarray(ds)
            from .batch import Plan
            plan = Plan(ds[:, 0], X3.shape[1], X3.device)
            gb = GraphBatch(plan, X3.reshape(-1, X3.shape[2]))
            X = gb.X
        if gb is None:
            raise ValueError("packed node_features need x['graph_batch']")
        return _out_as(GraphGatherFunction.apply(_in32(X), gb, self.activation), X.dtype)


def _rows_of(x, key="node_features", pack=None):
    """(GraphBatch, 2-D tensor) for a node-feature input that is either a RowViews of this package,
    a 2-D tensor + x['graph_batch'], or the reference's list of padded tensors (+ 'data_slice')."""
    nf = x[key]
    gb = x.get("graph_batch")
    if isinstance(nf, RowViews):
        return (gb if gb is not None else getattr(nf, "batch", None)), nf.data
    if gb is not None and torch.is_tensor(nf) and nf.dim() == 2:
        return gb, nf
    X3 = nf if torch.is_tensor(nf) else torch.stack(list(nf))
    ds = x["data_slice"]
    ds = ds.detach().cpu().numpy() if torch.is_tensor(ds) else np.asarray(ds)
    gb = GraphBatch.from_padded(X3, x.get("original_laplacian"), ds, pack=pack)
    return gb, gb.X


def _row_views(data, gb, pad_fill=None):
    return RowViews(data, gb.plan, gb, pad_fill)


class DenseMol(Layer):
    """Per-node dense layer (dense_layer.py:13-50): xx . W + b for every row INCLUDING pad rows; the
    activation given to the constructor is stored but never applied, exactly like the reference.
    DenseMol(output_dim, input_dim, init='glorot_uniform', activation='relu', bias=True)."""

    def __init__(self, output_dim, input_dim, init="glorot_uniform", activation="relu", bias=True, device=None,
                 generator=None, **kwargs):
        if init != "glorot_uniform":
            raise ValueError("only glorot_uniform is reached by the reference networks")
        self.init = init
        self.activation = activation          # stored, never applied (dense_layer.py:23,42-50)
        self.output_dim = output_dim
        self.input_dim = input_dim
        self.bias = bias
        self.input_shape = (self.input_dim,)
        if self.input_dim:
            kwargs["input_shape"] = (self.input_dim,)
        super(DenseMol, self).__init__(**kwargs)
        self._device, self._generator = device, generator
        self.W = self.b = None

    def build(self, device=None):
        if self.W is None:
            device = self._device if self._device is not None else device
            self.W = glorot([self.input_dim, self.output_dim], self._generator, device).requires_grad_(bool(self.trainable))
            self.b = torch.zeros(self.output_dim, dtype=torch.float32, device=device).requires_grad_(bool(self.trainable))

    def load_vars(self, W, b):
        dev = self._device
        self.W = torch.as_tensor(np.asarray(W), dtype=torch.float32).to(dev).requires_grad_(bool(self.trainable))
        self.b = torch.as_tensor(np.asarray(b), dtype=torch.float32).to(dev).requires_grad_(bool(self.trainable))

    def parameters(self):
        self.build()
        return [self.W, self.b]

    def __call__(self, x):
        # Fed the raw padded lists, the layer works on ALL max_atom rows like the reference (pad rows give the bias
        # and carry gradient into it, dense_layer.py:42-50).  Fed the packed output of a previous layer, pad rows
        # exist only in the lazily padded view (= bias, no gradient) -- the same thing for every reference network,
        # whose only consumer of this layer, GraphGatherMol, slices the real rows (graphgather.py:50-58).
        gb, X = _rows_of(x, pack=False)
        self.build(X.device)
        Y = RowsDenseFunction.apply(_in32(X), self.W, self.b if self.bias else None, None, gb, "linear", True)
        # pad rows of the reference's output hold the bias (xx . W + b on all max_atom rows, dense_layer.py:42-50)
        return _row_views(_out_as(Y, X.dtype), gb, self.b if self.bias else None)


class BlockEnd(Layer):
    """Last layer of a residual block (blockend.py:13-86): act(x + x_res . W), pad rows 0.
    BlockEnd(block_id, res_n_features, n_features, activation='relu', max_atom=128, batch_size=256);
    input keys: node_features, data_slice, block_outputs."""

    def __init__(self, block_id, res_n_features, n_features, activation="relu", max_atom=128, batch_size=256,
                 device=None, generator=None, **kwargs):
        super(BlockEnd, self).__init__(**kwargs)
        if activation not in ("relu", "linear", None):
            raise ValueError("Invalid activation function: %s" % (activation,))
        self.max_atom, self.batch_size, self.block_id = max_atom, batch_size, block_id
        self.activation = activation
        self.res_n_features, self.n_features = res_n_features, n_features
        self.vars = {}
        self._device, self._generator = device, generator

    def build(self, device=None):
        if "weight" not in self.vars:
            device = self._device if self._device is not None else device
            self.vars["weight"] = glorot([self.res_n_features, self.n_features], self._generator,
                                         device).requires_grad_(bool(self.trainable))

    def load_vars(self, arrays):
        for k, v in arrays.items():
            self.vars[k] = torch.as_tensor(np.asarray(v), dtype=torch.float32).to(self._device).requires_grad_(
                bool(self.trainable))

    def parameters(self):
        self.build()
        return list(self.vars.values())

    def call(self, x):
        gb, X = _rows_of(x)
        x2 = dict(x)
        x2["graph_batch"] = gb
        Xres = _source_rows(x["block_outputs"], self.res_n_features, gb)
        self.build(X.device)
        Y = RowsDenseFunction.apply(_in32(Xres), self.vars["weight"], None, _in32(X).contiguous(), gb,
                                    "relu" if self.activation == "relu" else "linear", False)
        return _row_views(_out_as(Y, X.dtype), gb)


def _source_rows(src, F, gb):
    """One [rows, F] tensor for a residual source given as a RowViews, a 2-D tensor, or B per-graph tensors
    (what the reference's containers accumulate with `list += layer_output`, tf_graphs.py:268,292).  Per-graph
    views produced by this package are recognised and mapped back to their packed parent without a copy."""
    plan = gb.plan
    stride = getattr(gb, "pad_stride", 0) if plan.rows == plan.N1 else 0
    if isinstance(src, RowViews):
        return src.data
    if torch.is_tensor(src):
        src2 = src if src.dim() == 2 else src.reshape(-1, src.shape[-1])
        if stride and src2.shape[0] == plan.B * stride and src2.shape[0] != plan.rows:
            return PackRowsFunction.apply(src2, gb, stride)      # padded 3-D / 2-D source, packed batch
        return src2
    ts = list(src)
    if len(ts) != plan.B:
        raise ValueError("a residual source must hold one tensor per graph (%d), got %d" % (plan.B, len(ts)))
    base = ts[0]._base
    if base is not None and base.dim() == 2 and base.shape[0] == plan.rows and base.is_contiguous():
        w = base.shape[1] * base.element_size()
        if all(t._base is base and t.data_ptr() == base.data_ptr() + int(plan.row_off[b]) * w
               for b, t in enumerate(ts)):
            return base
    if stride and all(t.shape[0] == stride for t in ts):   # the reference's padded per-graph tensors
        return PackRowsFunction.apply(torch.stack(ts).reshape(-1, ts[0].shape[-1]), gb, stride)
    if plan.rows == plan.N1:      # packed rows
        return torch.cat([t[:int(plan.n[b])] for b, t in enumerate(ts)], 0)
    return torch.stack(ts).reshape(-1, ts[0].shape[-1])


class DenseBlockEnd(Layer):
    """Last layer of a densely connected block (densenet_block.py:15-131):
        act(x + beta_inblock * sum_l a_l . W_in_l + beta_outblock * sum_b o_b . W_out_b), pad rows 0.
    DenseBlockEnd(block_id, res_n_features_list, output_n_features, activation='relu', K=2, max_atom=128,
    batch_size=256); input keys: node_features, data_slice, inblock_activations(+_dim), block_outputs(+_dim).
    The two activation lists are flat, source-major (entry l * batch_size + graph, :114,120), as the
    reference's container builds them; a list of per-source RowViews / 2-D tensors is accepted as well."""

    def __init__(self, block_id, res_n_features_list, output_n_features, activation="relu", K=2, max_atom=128,
                 batch_size=256, device=None, generator=None, **kwargs):
        super(DenseBlockEnd, self).__init__(**kwargs)
        if activation not in ("relu", "linear", None):
            raise ValueError("Invalid activation function: %s" % (activation,))
        assert type(res_n_features_list) == list
        self.max_atom, self.batch_size, self.block_id = max_atom, batch_size, block_id
        self.activation, self.K = activation, K
        self.res_n_features_list = res_n_features_list
        self.output_n_features = output_n_features
        self.vars = {}
        self.inblock_activations_dim, self.preceding_blocks_dim = [], []
        self._device, self._generator = device, generator

    def build(self, device=None):
        if "beta_inblock" in self.vars:
            return
        device = self._device if self._device is not None else device
        tr = bool(self.trainable)
        mk = lambda F: glorot([F, self.output_n_features], self._generator, device).requires_grad_(tr)  # noqa: E731
        # creation order of the reference: out-of-block weights, in-block weights, beta1, beta2 (:50-68)
        self.vars["weight_outblock"] = [mk(F) for F in self.preceding_blocks_dim]
        self.vars["weight_inblock"] = [mk(F) for F in self.inblock_activations_dim]
        self.vars["beta_inblock"] = torch.ones(1, dtype=torch.float32, device=device).requires_grad_(tr)
        self.vars["beta_outblock"] = torch.ones(1, dtype=torch.float32, device=device).requires_grad_(tr)

    def load_vars(self, weight_inblock, weight_outblock, beta_inblock, beta_outblock):
        dev, tr = self._device, bool(self.trainable)
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32).to(dev).requires_grad_(tr)  # noqa: E731
        self.vars = {"weight_inblock": [t(w) for w in weight_inblock], "weight_outblock": [t(w) for w in weight_outblock],
                     "beta_inblock": t(beta_inblock).reshape(1).detach().requires_grad_(tr),
                     "beta_outblock": t(beta_outblock).reshape(1).detach().requires_grad_(tr)}
        self.inblock_activations_dim = [int(w.shape[0]) for w in self.vars["weight_inblock"]]
        self.preceding_blocks_dim = [int(w.shape[0]) for w in self.vars["weight_outblock"]]

    def parameters(self):
        self.build()
        return (list(self.vars["weight_inblock"]) + list(self.vars["weight_outblock"]) +
                [self.vars["beta_inblock"], self.vars["beta_outblock"]])

    @staticmethod
    def _split(lst, dims, B):
        """flat source-major list of len(dims) * B tensors -> len(dims) sources"""
        lst = lst if isinstance(lst, (list, tuple)) else list(lst)
        if len(lst) == len(dims):                                        # one entry per source
            return list(lst)
        if len(lst) != len(dims) * B:                                   # densenet_block.py:108
            raise ValueError("expected %d x %d tensors, got %d" % (len(dims), B, len(lst)))
        return [lst[i * B:(i + 1) * B] for i in range(len(dims))]

    def call(self, x):
        gb, X = _rows_of(x)
        B = gb.plan.B
        in_dims, out_dims = list(x["inblock_activations_dim"]), list(x["block_outputs_dim"])
        if "beta_inblock" not in self.vars:
            self.inblock_activations_dim, self.preceding_blocks_dim = in_dims, out_dims
        self.build(X.device)
        if [int(w.shape[0]) for w in self.vars["weight_inblock"]] != in_dims or \
                [int(w.shape[0]) for w in self.vars["weight_outblock"]] != out_dims:
            raise ValueError("source widths do not match the layer's weights")
        ins = self._split(x["inblock_activations"], in_dims, B)
        outs = self._split(x["block_outputs"], out_dims, B)
        terms = [(_source_rows(a, F, gb), w, self.vars["beta_inblock"])
                 for a, F, w in zip(ins, in_dims, self.vars["weight_inblock"])]
        terms += [(_source_rows(o, F, gb), w, self.vars["beta_outblock"])
                  for o, F, w in zip(outs, out_dims, self.vars["weight_outblock"])]
        if not terms:
            raise ValueError("DenseBlockEnd needs at least one saved activation (tf_graphs.py:281)")
        act = "relu" if self.activation == "relu" else "linear"
        acc = _in32(X)
        for i, (A, W, beta) in enumerate(terms):
            last = i == len(terms) - 1
            acc = RowsDenseFunction.apply(_in32(A), W, None, acc.contiguous(), gb, act if last else "linear", False, beta)
        return _row_views(_out_as(acc, X.dtype), gb)


class MultitaskHead(object):
    """The classifier tail of MultitaskGraphClassifier: `multitask_logits(feat, n_tasks, n_feature=...)`
    (model_operatos.py:792-829: one fully connected 2-class head per task, w ~ truncated_normal(stddev 0.01),
    b = 0, :744-750) and `add_training_loss('cross_entropy', logits)` (multitask_classifier.py:41-44,187-209:
    per-task weighted sigmoid cross-entropy against one-hot labels, summed, divided by batch_size).
    Under data parallelism (`dist.allreduce_bucket` SUMS the per-rank gradients) `batch_size` must be the GLOBAL batch
    size, `dist.global_batch_size(local)`: the reference's loss divides by the whole batch
    (multitask_classifier.py:207-208); the feature matrix a rank passes in holds its local graphs only."""

    def __init__(self, n_tasks, n_feature, batch_size, device=None, generator=None, trainable=True):
        self.n_tasks, self.n_feature, self.batch_size = n_tasks, n_feature, batch_size
        w = torch.empty((n_tasks, n_feature, 2), dtype=torch.float32)
        torch.nn.init.trunc_normal_(w, mean=0.0, std=0.01, a=-0.02, b=0.02, generator=generator)
        self.vars = {"w": w.to(device).requires_grad_(bool(trainable)),
                     "b": torch.zeros((n_tasks, 2), dtype=torch.float32, device=device).requires_grad_(bool(trainable))}

    def load_vars(self, w, b):
        dev, tr = self.vars["w"].device, self.vars["w"].requires_grad
        self.vars["w"] = torch.as_tensor(np.asarray(w), dtype=torch.float32).to(dev).requires_grad_(tr)
        self.vars["b"] = torch.as_tensor(np.asarray(b), dtype=torch.float32).to(dev).requires_grad_(tr)

    def parameters(self):
        return [self.vars["w"], self.vars["b"]]

    def loss(self, feat, labels, weights):
        """feat [B, n_feature]; labels, weights [B, n_tasks] (label_placeholder / weight_placeholder).
        Returns (loss 0-dim tensor, list of n_tasks logits tensors [B, 2])."""
        loss, logits = MultitaskCEFunction.apply(feat, self.vars["w"], self.vars["b"], labels, weights,
                                                 self.batch_size)
        return loss.reshape(()), [logits[t] for t in range(self.n_tasks)]

    def logits(self, feat):
        """Prediction path: labels and weights all zero, as construct_feed_dict does when none are given
        (multitask_classifier.py:172-176)."""
        z = torch.zeros((feat.shape[0], self.n_tasks), dtype=torch.float32, device=feat.device)
        return self.loss(feat, z, z)[1]


class GraphPoolMol(Layer):
    """Neighbourhood max-pooling (graphpool.py:11-110).  GraphPoolMol(batch_size).  grad: 'reference' (the
    reference's py_func passes no gradient) or 'argmax' (extension)."""

    def __init__(self, batch_size, grad="reference", **kwargs):
        super(GraphPoolMol, self).__init__(**kwargs)
        if grad not in ("reference", "argmax"):
            raise ValueError("grad must be 'reference' or 'argmax'")
        self.batch_size = batch_size
        self.sparse_inputs = True
        self.grad = grad

    def get_output_shape_for(self, input_shape):
        return input_shape[0]

    def call(self, x, mask=None):
        gb, X = _rows_of(x)
        if gb.l0_dense is None and gb.l0_rowptr is None:
            raise ValueError("GraphPoolMol needs the original Laplacian")
        return _row_views(GraphPoolFunction.apply(X, gb, self.grad), gb)
