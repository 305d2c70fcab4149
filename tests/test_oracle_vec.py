"""The all-core CPU baseline (oracle/agcn_oracle_vec.py, torch-CPU over the zero-padded batch) restates the same
layer as the per-graph numpy oracle: checked here so that the baseline bench.py reports is a correct one."""
import numpy as np
import pytest

from helpers import orc

orv = pytest.importorskip("agcn_oracle_vec")


@pytest.mark.parametrize("mode", [orc.REFERENCE, orc.PAPER])
def test_vec_oracle_matches_numpy_oracle(mode):
    sb = orc.synth_batch(77, 12)
    rs = np.random.RandomState(3)
    Fin, Fout, K = 75, 64, 2
    p = orc.init_params(rs, Fin, Fout, K)
    p["alpha"] = np.array([0.7], np.float32)
    G = [rs.randn(int(n), Fout).astype(np.float32) for n in sb["ns"]]
    caches = orc.layer_forward(sb["X"], sb["L0"], p, K, mode, orc.SGC_LL)
    tot = orc.layer_backward(caches, G, p, K, mode, orc.SGC_LL)
    bt = orv.pad_batch(sb["X"], sb["L0"], G)
    out, grads = orv.step(bt, orv.params_to_torch(p), K, mode)
    for b, n in enumerate(sb["ns"]):
        n = int(n)
        np.testing.assert_allclose(out["Y"][b, :n].detach().numpy(), caches[b]["Y"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(out["res_W"][b, :n, :n].detach().numpy(), caches[b]["S"], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(out["res_L"][b, :n, :n].detach().numpy(), caches[b]["R2"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(grads["dX"][b, :n].numpy(), tot["dX"][b], rtol=1e-3, atol=2e-5 * np.abs(tot["dX"][b]).max())
    sw = np.abs(tot["dweight"]).max()
    np.testing.assert_allclose(grads["dweight"].numpy(), tot["dweight"], rtol=1e-3, atol=2e-5 * sw)
    np.testing.assert_allclose(grads["dbias"].numpy(), tot["dbias"], rtol=1e-3, atol=2e-5 * sw)
    if mode == orc.PAPER:
        np.testing.assert_allclose(grads["dM_L"].numpy(), tot["dM_L"], rtol=2e-3, atol=5e-5 * np.abs(tot["dM_L"]).max())
        assert abs(float(grads["dalpha"][0]) - float(tot["dalpha"][0])) <= 1e-4 * max(1.0, abs(float(tot["dalpha"][0])))
