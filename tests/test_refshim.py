"""Pins the fake-TensorFlow shim (oracle/refshim/fake_tensorflow.py) that the golden vectors were generated
behind: its TF-0.12 helpers are checked against hand-computed values from TF's documented formulas (SURVEY.md
Appendix A), including the tie case of MinimumGrad and the relu gradient at 0; and, where the reference tree is
present, the committed goldens are regenerated from the reference's own classes and compared bit for bit."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "refshim"))
import fake_tensorflow as ftf  # noqa: E402


def _leaf(v):
    return ftf.Tensor(torch.tensor(v, dtype=torch.float32, requires_grad=True))


def _grad_of(y, x, g):
    (y.t * torch.tensor(g, dtype=torch.float32)).sum().backward()
    return x.t.grad.numpy()


def test_clip_by_norm_values_and_gradients():
    # tf.clip_by_norm(t, c) = t * c * min(rsqrt(sum t^2), 1 / c)   (TF 0.12 clip_ops.py)
    t = _leaf([3.0, 4.0])                                   # ||t|| = 5 > 1: scaled to norm 1
    y = ftf.clip_by_norm(t, 1.0)
    np.testing.assert_allclose(y.t.detach().numpy(), [0.6, 0.8], rtol=1e-6)
    # d(t / ||t||) . g = (g - that <g, that>) / ||t||, that = t / ||t||: g = (1, 0) -> (1 - 0.36, -0.48) / 5
    np.testing.assert_allclose(_grad_of(y, t, [1.0, 0.0]), [0.128, -0.096], rtol=1e-5)
    t = _leaf([0.3, 0.4])                                   # ||t|| = 0.5 < 1: untouched, identity gradient
    y = ftf.clip_by_norm(t, 1.0)
    np.testing.assert_allclose(y.t.detach().numpy(), [0.3, 0.4], rtol=1e-6)
    np.testing.assert_allclose(_grad_of(y, t, [1.0, -2.0]), [1.0, -2.0], rtol=1e-6)
    t = _leaf([1.0, 0.0])                                   # ||t|| == 1 exactly: min(1, 1) is a tie ->
    y = ftf.clip_by_norm(t, 1.0)                            # MinimumGrad (x <= y) takes the NORM branch
    np.testing.assert_allclose(y.t.detach().numpy(), [1.0, 0.0], rtol=1e-6)
    np.testing.assert_allclose(_grad_of(y, t, [1.0, 1.0]), [0.0, 1.0], atol=1e-6)   # projection, not identity
    t = _leaf([6.0, 8.0])                                   # clip_norm = 2: 10 -> 2
    np.testing.assert_allclose(ftf.clip_by_norm(t, 2.0).t.detach().numpy(), [1.2, 1.6], rtol=1e-6)


def test_clip_by_average_norm_values_and_gradients():
    # tf.clip_by_average_norm(t, c) = t * c * min(rsqrt(sum t^2) * numel, 1 / c)
    t = _leaf([3.0, 4.0])                                   # average norm 5 / 2 = 2.5 > 1 -> t / 2.5
    y = ftf.clip_by_average_norm(t, 1.0)
    np.testing.assert_allclose(y.t.detach().numpy(), [1.2, 1.6], rtol=1e-6)
    # y = 2 t / ||t||: gradient = 2 (g - that <g, that>) / ||t||
    np.testing.assert_allclose(_grad_of(y, t, [1.0, 0.0]), [0.256, -0.192], rtol=1e-5)
    t = _leaf([[1.0, 0.0], [0.0, 1.0]])                     # the identity: ||.||_F / numel = sqrt(2) / 4 < 1: no-op
    y = ftf.clip_by_average_norm(t, 1.0)
    np.testing.assert_array_equal(y.t.detach().numpy(), np.eye(2, dtype=np.float32))
    np.testing.assert_allclose(_grad_of(y, t, [[1.0, 2.0], [3.0, 4.0]]), [[1.0, 2.0], [3.0, 4.0]], rtol=1e-6)


def test_minimum_gradient_tie_goes_to_first_argument():
    # TF _MinimumGrad: xmask = less_equal(x, y)
    x, y = _leaf([1.0, 2.0, 3.0]), _leaf([2.0, 2.0, 1.0])
    m = ftf.minimum(x, y)
    np.testing.assert_array_equal(m.t.detach().numpy(), [1.0, 2.0, 1.0])
    (m.t * torch.tensor([1.0, 10.0, 100.0])).sum().backward()
    np.testing.assert_array_equal(x.t.grad.numpy(), [1.0, 10.0, 0.0])
    np.testing.assert_array_equal(y.t.grad.numpy(), [0.0, 0.0, 100.0])


def test_relu_gradient_is_zero_at_zero():
    # TF ReluGrad passes the gradient where the OUTPUT is > 0
    x = _leaf([-1.0, 0.0, 2.0])
    y = ftf.nn.relu(x)
    np.testing.assert_array_equal(y.t.detach().numpy(), [0.0, 0.0, 2.0])
    np.testing.assert_array_equal(_grad_of(y, x, [5.0, 5.0, 5.0]), [0.0, 0.0, 5.0])


def test_old_argument_orders_and_shape_helpers():
    a = ftf.constant(np.arange(6, dtype=np.float32).reshape(2, 3))
    b = ftf.constant(np.arange(6, 12, dtype=np.float32).reshape(2, 3))
    # TF <= 0.12: tf.concat(concat_dim, values)   (graphconv.py:227)
    np.testing.assert_array_equal(ftf.concat(0, [a, b]).t.numpy(), np.arange(12, dtype=np.float32).reshape(4, 3))
    np.testing.assert_array_equal(ftf.concat(1, [a, b]).t.numpy(),
                                  np.concatenate([a.t.numpy(), b.t.numpy()], 1))
    # tf.slice with size -1 = "to the end"   (graphconv.py:151-154: data_slice rows are [n, -1])
    np.testing.assert_array_equal(ftf.slice(a, [0, 1], [1, -1]).t.numpy(), [[1.0, 2.0]])
    np.testing.assert_array_equal(ftf.slice(a, [0, 0], [2, 2]).t.numpy(), [[0.0, 1.0], [3.0, 4.0]])
    # tf.stack / tf.pack of python ints builds a shape   (graphconv.py:239-240)
    assert ftf.stack([2, 3]) == [2, 3] and ftf.pack([4, 5, 6]) == [4, 5, 6]
    st = ftf.stack([a, b])
    assert tuple(st.t.shape) == (2, 2, 3)
    # transpose with perm, reshape with a stacked shape, pad CONSTANT
    np.testing.assert_array_equal(ftf.transpose(st, perm=[1, 2, 0]).t.numpy(), np.transpose(st.t.numpy(), (1, 2, 0)))
    np.testing.assert_array_equal(ftf.reshape(a, ftf.stack([3, 2])).t.numpy(), a.t.numpy().reshape(3, 2))
    np.testing.assert_array_equal(ftf.pad(a, [[0, 1], [0, 0]]).t.numpy(),
                                  np.pad(a.t.numpy(), [(0, 1), (0, 0)]))
    assert a.get_shape().as_list() == [2, 3]
    assert a.dtype.base_dtype == ftf.float32


def test_py_func_blocks_the_gradient():
    x = _leaf([[1.0, 2.0], [3.0, 4.0]])
    out = ftf.py_func(lambda v: (v * 2).astype(np.float32), [x], ftf.float32)
    assert not out.t.requires_grad                     # graphconv.py:211: no gradient through the similarity
    np.testing.assert_array_equal(out.t.numpy(), [[2.0, 4.0], [6.0, 8.0]])


def test_in_place_operators_do_not_alias():
    # model_operatos.py:557 (`x -= alpha * negative_part`) must not modify the tensor it was derived from
    a = ftf.constant(np.array([1.0, 2.0], np.float32))
    b = a
    b -= ftf.constant(np.array([1.0, 1.0], np.float32))
    np.testing.assert_array_equal(a.t.numpy(), [1.0, 2.0])
    np.testing.assert_array_equal(b.t.numpy(), [0.0, 1.0])


@pytest.mark.skipif(not os.path.isdir("/root/reference/models/layers"), reason="the reference tree is not on this box")
def test_goldens_regenerate_bit_for_bit(tmp_path):
    """Re-runs oracle/refshim/gen_golden.py (the reference's own classes behind the shim) into a scratch directory:
    every committed tests/golden/*.npz must come back identical, array by array, bit for bit."""
    env = dict(os.environ, AGCN_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "refshim", "gen_golden.py")], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert res.returncode == 0, res.stdout.decode()[-2000:]
    committed = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    assert len(committed) >= 30
    for path in committed:
        fresh = os.path.join(str(tmp_path), os.path.basename(path))
        assert os.path.exists(fresh), "not regenerated: " + os.path.basename(path)
        a, b = np.load(path), np.load(fresh)
        assert sorted(a.files) == sorted(b.files), os.path.basename(path)
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (path, k)
            assert a[k].tobytes() == b[k].tobytes(), "%s: array %s differs" % (os.path.basename(path), k)
