"""A tiny stand-in for the TensorFlow-0.12 API surface that the reference's
layer code touches, backed by torch-CPU autograd.

TEST INFRASTRUCTURE ONLY.  It exists so that `oracle/refshim/gen_golden.py`
can import and execute the reference's own, unmodified layer classes
(`/root/reference/models/layers/*.py`) in the build container and record
their inputs/outputs/gradients as golden vectors under `tests/golden/`.
Nothing in the product path (`agcn_amd`, `bench.py`) imports this file and it
never runs on the GPU box (the reference does not exist there).

Only the ops reached by the hot path are provided:
graphconv.py, graphconv_reslap.py, graphgather.py, graphpool.py, dense_layer.py,
blockend.py, densenet_block.py and operators/{activations,model_operatos}.py.

Semantics notes (TF 0.12, CPU):
  * clip_by_norm(t, c):          t * c * min(rsqrt(sum(t*t)), 1/c)
  * clip_by_average_norm(t, c):  t * c * min(rsqrt(sum(t*t)) * numel, 1/c)
  * minimum() gradient: ties go to the first argument (x <= y mask)
  * nn.relu gradient: 1 where x > 0, else 0
  * py_func: numpy round trip, no gradient
  * concat(dim, values): the pre-1.0 argument order (graphconv.py:227)
  * Tensor has no __ne__/__eq__ overload, so `alpha != 0.` is always True
"""
import contextlib
import sys
import types

import numpy as np
import torch

torch.set_grad_enabled(True)


# --------------------------------------------------------------------------
# dtypes
class DType(object):
    def __init__(self, tdtype, name):
        self.t = tdtype
        self.name = name

    @property
    def base_dtype(self):
        return self

    def __eq__(self, other):
        if isinstance(other, str):
            return self.name == other
        return isinstance(other, DType) and other.t == self.t

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self.name)

    def __repr__(self):
        return "tf." + self.name


float32 = DType(torch.float32, "float32")
float64 = DType(torch.float64, "float64")
int32 = DType(torch.int32, "int32")
int64 = DType(torch.int64, "int64")
bool = DType(torch.bool, "bool")  # noqa: A001 (mirrors tf.bool)
_DT = {torch.float32: float32, torch.float64: float64, torch.int32: int32,
       torch.int64: int64, torch.bool: bool}


def _as_dtype(d):
    if d is None:
        return None
    if isinstance(d, DType):
        return d
    if isinstance(d, str):
        return {"float32": float32, "float64": float64, "int32": int32,
                "int64": int64, "bool": bool}[d]
    raise TypeError(d)


# --------------------------------------------------------------------------
# Tensor wrapper
class _Shape(object):
    def __init__(self, dims):
        self.dims = list(dims)

    def as_list(self):
        return list(self.dims)

    def __getitem__(self, i):
        return self.dims[i]

    def __iter__(self):
        return iter(self.dims)

    def __len__(self):
        return len(self.dims)


class Tensor(object):
    """Wraps a torch tensor; arithmetic is always out of place."""
    __array_priority__ = 1000

    def __init__(self, t, name=None):
        assert isinstance(t, torch.Tensor)
        self.t = t
        self.name = name

    @property
    def dtype(self):
        return _DT[self.t.dtype]

    def get_shape(self):
        return _Shape(self.t.shape)

    @property
    def shape(self):
        return _Shape(self.t.shape)

    # arithmetic ------------------------------------------------------------
    def __add__(self, o):
        return Tensor(self.t + _raw(o, like=self))

    __radd__ = __add__

    def __sub__(self, o):
        return Tensor(self.t - _raw(o, like=self))

    def __rsub__(self, o):
        return Tensor(_raw(o, like=self) - self.t)

    def __mul__(self, o):
        return Tensor(self.t * _raw(o, like=self))

    __rmul__ = __mul__

    def __truediv__(self, o):
        return Tensor(self.t / _raw(o, like=self))

    def __neg__(self):
        return Tensor(-self.t)

    def __gt__(self, o):
        return Tensor(self.t > _raw(o, like=self))

    def __lt__(self, o):
        return Tensor(self.t < _raw(o, like=self))

    # `-=` etc. fall back to __sub__ (no __isub__) => out of place, like TF.
    def __bool__(self):
        raise TypeError("tf.Tensor is not a Python bool")

    def __repr__(self):
        return "Tensor(%r)" % (self.t,)


def _raw(x, like=None):
    if isinstance(x, Tensor):
        return x.t
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x))
    if isinstance(x, (int, float, np.floating, np.integer)):
        if like is not None:
            return torch.tensor(x, dtype=like.t.dtype)
        return torch.tensor(x)
    if isinstance(x, (list, tuple)):
        if len(x) and isinstance(x[0], (Tensor, torch.Tensor)):
            return torch.stack([_raw(v) for v in x])
        return torch.tensor(np.asarray(x))
    raise TypeError(type(x))


def _ints(x):
    """Turn a shape-like (Tensor / list of Tensors / ints) into python ints."""
    if isinstance(x, Tensor):
        return [int(v) for v in x.t.reshape(-1).tolist()]
    out = []
    for v in x:
        if isinstance(v, Tensor):
            out.append(int(v.t.reshape(-1)[0].item()) if v.t.numel() == 1 else None)
        else:
            out.append(int(v))
    return out


# --------------------------------------------------------------------------
# variable creation, with a hook for deterministic / overridden values
class _State(object):
    rng = np.random.RandomState(0)
    overrides = {}      # variable name -> ndarray
    created = []        # (name, Tensor) in creation order


STATE = _State()


def reset_state(seed=0, overrides=None):
    STATE.rng = np.random.RandomState(seed)
    STATE.overrides = dict(overrides or {})
    STATE.created = []


def _make_var(value, name):
    if name in STATE.overrides:
        ov = np.asarray(STATE.overrides[name], dtype=np.float32)
        assert tuple(ov.shape) == tuple(value.shape), (name, ov.shape, value.shape)
        value = torch.from_numpy(ov.copy())
    leaf = value.detach().clone().requires_grad_(True)
    v = Tensor(leaf, name=name)
    STATE.created.append((name, v))
    return v


def Variable(initial_value, *args, **kwargs):
    # model_operatos.zeros() passes (value, dtype, name) positionally (model_operatos.py:495-496):
    # take the first string among the positionals as the name
    name = kwargs.get("name")
    for a in args:
        if name is None and isinstance(a, str):
            name = a
    return _make_var(_raw(initial_value).to(torch.float32), name)


def get_variable(name, shape=None, initializer=None, dtype=None, **_):
    shape = tuple(int(s) for s in (shape if shape is not None else ()))
    init = initializer(shape) if initializer is not None else torch.zeros(shape)
    return _make_var(_raw(init).to(torch.float32), name)


def constant_initializer(value=0.0, **_):
    return lambda shape: torch.full(tuple(int(s) for s in shape), float(value), dtype=torch.float32)


def truncated_normal_initializer(mean=0.0, stddev=1.0, **_):
    def f(shape):
        v = STATE.rng.normal(mean, stddev, size=shape)
        v = np.clip(v, mean - 2 * stddev, mean + 2 * stddev)
        return torch.from_numpy(v.astype(np.float32))
    return f


def random_uniform_initializer(minval=0.0, maxval=1.0, dtype=None, seed=None, **_):
    def f(shape):
        v = STATE.rng.uniform(minval, maxval, size=tuple(int(s) for s in shape))
        return Tensor(torch.from_numpy(v.astype(np.float32)))
    return f


def random_normal_initializer(mean=0.0, stddev=1.0, dtype=None, seed=None, **_):
    def f(shape):
        v = STATE.rng.normal(mean, stddev, size=tuple(int(s) for s in shape))
        return Tensor(torch.from_numpy(v.astype(np.float32)))
    return f


def random_uniform(shape, minval=0.0, maxval=1.0, dtype=None, **_):
    v = STATE.rng.uniform(minval, maxval, size=tuple(int(s) for s in shape))
    return Tensor(torch.from_numpy(v.astype(np.float32)))


def truncated_normal(shape, mean=0.0, stddev=1.0, **_):
    return Tensor(truncated_normal_initializer(mean, stddev)(tuple(int(s) for s in shape)))


def zeros(shape, dtype=None, **_):
    return Tensor(torch.zeros(tuple(int(s) for s in shape), dtype=torch.float32))


def ones(shape, dtype=None, **_):
    return Tensor(torch.ones(tuple(int(s) for s in shape), dtype=torch.float32))


@contextlib.contextmanager
def variable_scope(*_, **__):
    yield


name_scope = variable_scope


def global_variables_initializer():
    return None


# --------------------------------------------------------------------------
# plain ops
def constant(value, dtype=None, shape=None, name=None):
    arr = np.asarray(value)
    if dtype is not None:
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(_as_dtype(dtype).t)
    elif arr.dtype.kind == "f":
        t = torch.from_numpy(np.ascontiguousarray(arr.astype(np.float32)))
    elif arr.dtype.kind in "iu":
        t = torch.from_numpy(np.ascontiguousarray(arr.astype(np.int32)))
    else:
        t = torch.from_numpy(np.ascontiguousarray(arr))
    return Tensor(t)


def convert_to_tensor(x, dtype=None, **_):
    if isinstance(x, Tensor):
        return x
    return constant(x, dtype=dtype)


def cast(x, dtype, **_):
    return Tensor(_raw(x).to(_as_dtype(dtype).t))


def gather(params, indices, **_):
    return Tensor(_raw(params)[_raw(indices).long()])


def slice(input_, begin, size, **_):  # noqa: A001
    t = _raw(input_)
    b = _ints(begin)
    s = _ints(size)
    idx = []
    for d, (b0, s0) in enumerate(zip(b, s)):
        idx.append(np.s_[b0:] if s0 == -1 else np.s_[b0:b0 + s0])
    return Tensor(t[tuple(idx)])


def reduce_sum(x, axis=None, keep_dims=False, **_):
    t = _raw(x)
    if axis is None:
        return Tensor(t.sum())
    return Tensor(t.sum(dim=axis, keepdim=keep_dims))


def reduce_max(x, axis=None, keep_dims=False, **_):
    t = _raw(x)
    if axis is None:
        return Tensor(t.max())
    return Tensor(t.max(dim=axis, keepdim=keep_dims).values)


def reduce_mean(x, axis=None, keep_dims=False, **_):
    t = _raw(x)
    if axis is None:
        return Tensor(t.mean())
    return Tensor(t.mean(dim=axis, keepdim=keep_dims))


def matmul(a, b, **_):
    return Tensor(torch.matmul(_raw(a), _raw(b)))


def transpose(x, perm=None, **_):
    t = _raw(x)
    if perm is None:
        perm = list(range(t.dim()))[::-1]
    return Tensor(t.permute(*[int(p) for p in perm]))


def stack(values, axis=0, **_):
    vals = list(values)
    if all(isinstance(v, (int, np.integer)) or
           (isinstance(v, Tensor) and v.t.dim() == 0) for v in vals) and \
            any(not isinstance(v, Tensor) for v in vals):
        # shape-building use (graphconv.py:239-240): return python ints
        return _ints(vals)
    if all(isinstance(v, Tensor) and v.t.dim() == 0 and
           v.t.dtype in (torch.int32, torch.int64) for v in vals):
        return _ints(vals)
    return Tensor(torch.stack([_raw(v) for v in vals], dim=axis))


pack = stack


def concat(concat_dim, values, **_):
    # TF <= 0.12 argument order: concat(dim, values)
    if not isinstance(concat_dim, (int, np.integer)):
        concat_dim, values = values, concat_dim
    return Tensor(torch.cat([_raw(v) for v in values], dim=int(concat_dim)))


def reshape(x, shape, **_):
    return Tensor(_raw(x).reshape(*_ints(shape)))


def squeeze(x, axis=None, **_):
    t = _raw(x)
    return Tensor(t.squeeze() if axis is None else t.squeeze(axis))


def expand_dims(x, axis, **_):
    return Tensor(_raw(x).unsqueeze(axis))


def pad(x, paddings, mode="CONSTANT", **_):
    t = _raw(x)
    assert mode == "CONSTANT"
    p = [_ints(row) for row in paddings]
    flat = []
    for lo, hi in reversed(p):
        flat += [lo, hi]
    return Tensor(torch.nn.functional.pad(t, flat))


def add(a, b, **_):
    return Tensor(_raw(a) + _raw(b))


def subtract(a, b, **_):
    return Tensor(_raw(a) - _raw(b))


sub = subtract


def multiply(a, b, **_):
    return Tensor(_raw(a) * _raw(b))


mul = multiply


def add_n(values, **_):
    out = _raw(values[0])
    for v in values[1:]:
        out = out + _raw(v)
    return Tensor(out)


def exp(x, **_):
    return Tensor(torch.exp(_raw(x)))


def abs(x, **_):  # noqa: A001
    return Tensor(torch.abs(_raw(x)))


def where(cond, a, b, **_):
    return Tensor(torch.where(_raw(cond), _raw(a), _raw(b)))


class _TFMinimum(torch.autograd.Function):
    """min(x, y) whose gradient goes to x where x <= y (TF's MinimumGrad)."""

    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x <= y)
        ctx.xs, ctx.ys = x.shape, y.shape
        return torch.minimum(x, y)

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        gx = torch.where(m, g, torch.zeros_like(g))
        gy = torch.where(m, torch.zeros_like(g), g)
        return gx.sum_to_size(ctx.xs), gy.sum_to_size(ctx.ys)


def minimum(x, y, **_):
    a, b = _raw(x), _raw(y)
    if not isinstance(y, (Tensor, torch.Tensor)):
        b = b.to(a.dtype)
    return Tensor(_TFMinimum.apply(a, b))


def rsqrt(x, **_):
    return Tensor(torch.rsqrt(_raw(x)))


def clip_by_norm(t, clip_norm, **_):
    x = _raw(t)
    c = torch.tensor(float(clip_norm), dtype=x.dtype)
    l2norm_inv = torch.rsqrt((x * x).sum())
    scale = _TFMinimum.apply(l2norm_inv, torch.tensor(1.0, dtype=x.dtype) / c)
    return Tensor(x * c * scale)


def clip_by_average_norm(t, clip_norm, **_):
    x = _raw(t)
    c = torch.tensor(float(clip_norm), dtype=x.dtype)
    n_element = torch.tensor(float(x.numel()), dtype=x.dtype)
    l2norm_inv = torch.rsqrt((x * x).sum())
    scale = _TFMinimum.apply(l2norm_inv * n_element, torch.tensor(1.0, dtype=x.dtype) / c)
    return Tensor(x * c * scale)


def clip_by_value(t, lo, hi, **_):
    return Tensor(torch.minimum(torch.maximum(_raw(t), _raw(lo)), _raw(hi)))


def py_func(func, inp, Tout, **_):
    args = [_raw(a).detach().cpu().numpy() for a in inp]
    outs = func(*args)
    single = not isinstance(Tout, (list, tuple))
    if single:
        outs, Tout = [outs], [Tout]
    res = []
    for o, dt in zip(outs, Tout):
        o = np.asarray(o)
        want = {float32: np.float32, float64: np.float64, int32: np.int32,
                int64: np.int64}[_as_dtype(dt)]
        assert o.dtype == want, "py_func dtype mismatch: %s vs %s" % (o.dtype, want)
        res.append(Tensor(torch.from_numpy(np.ascontiguousarray(o))))
    return res[0] if single else res


class _Placeholder(object):
    """Hashable feed-dict key (graph_topology.py:58-78); never evaluated."""

    def __init__(self, dtype, shape, name):
        self.dtype, self.shape, self.name = dtype, shape, name


def placeholder(dtype=None, shape=None, name=None):
    return _Placeholder(dtype, shape, name)


def get_default_graph():
    return None


class Graph(object):
    @contextlib.contextmanager
    def as_default(self):
        yield self


# --------------------------------------------------------------------------
# tf.nn
nn = types.ModuleType("tensorflow.nn")


def _relu(x, **_):
    return Tensor(torch.relu(_raw(x)))


def _tanh(x, **_):
    return Tensor(torch.tanh(_raw(x)))


def _sigmoid(x, **_):
    return Tensor(torch.sigmoid(_raw(x)))


def _bias_add(x, b, **_):
    return Tensor(_raw(x) + _raw(b))


def _dropout(x, keep_prob, **_):
    raise RuntimeError("dropout is never enabled by any reference network")


def _softmax(x, **_):
    return Tensor(torch.softmax(_raw(x), dim=-1))


def _elu(x, **_):
    return Tensor(torch.nn.functional.elu(_raw(x)))


nn.relu = _relu
nn.tanh = _tanh
nn.sigmoid = _sigmoid
nn.bias_add = _bias_add
nn.dropout = _dropout
nn.softmax = _softmax
nn.elu = _elu
tanh = _tanh
sigmoid = _sigmoid


# ---- the few extra ops the classifier head / loss use (model_operatos.py:710-759,
# multitask_classifier.py:41-44,187-209) --------------------------------------------------
def split(split_dim, num_split, value, **_):
    """TF 0.12 argument order: tf.split(split_dim, num_split, value)."""
    return [Tensor(c) for c in torch.chunk(_raw(value), int(num_split), dim=int(split_dim))]


def to_int32(x, **_):
    return Tensor(_raw(x).to(torch.int32))


def to_float(x, **_):
    return Tensor(_raw(x).to(torch.float32))


def one_hot(indices, depth, **_):
    idx = _raw(indices).to(torch.int64)
    out = torch.zeros(tuple(idx.shape) + (int(depth),), dtype=torch.float32)
    ok = (idx >= 0) & (idx < int(depth))          # out-of-range index -> all-zero row, like TF
    out[ok, idx[ok]] = 1.0
    return Tensor(out)


def div(a, b, **_):
    return Tensor(_raw(a) / _raw(b))


def _xw_plus_b(x, w, b, **_):
    return Tensor(torch.matmul(_raw(x), _raw(w)) + _raw(b))


def _sigmoid_cross_entropy_with_logits(logits, targets, **_):
    """TF 0.12 positional order (logits, targets): max(x, 0) - x z + log(1 + exp(-|x|))."""
    x, z = _raw(logits), _raw(targets)
    return Tensor(torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-torch.abs(x))))


nn.xw_plus_b = _xw_plus_b
nn.sigmoid_cross_entropy_with_logits = _sigmoid_cross_entropy_with_logits


# --------------------------------------------------------------------------
def install():
    """Register this module as `tensorflow` (+ the sub-modules the reference
    imports: tensorflow.python.training.moving_averages)."""
    me = sys.modules[__name__]
    sys.modules["tensorflow"] = me
    sys.modules["tensorflow.nn"] = nn
    py = types.ModuleType("tensorflow.python")
    tr = types.ModuleType("tensorflow.python.training")
    ma = types.ModuleType("tensorflow.python.training.moving_averages")
    py.training = tr
    tr.moving_averages = ma
    me.python = py
    sys.modules["tensorflow.python"] = py
    sys.modules["tensorflow.python.training"] = tr
    sys.modules["tensorflow.python.training.moving_averages"] = ma
    return me
