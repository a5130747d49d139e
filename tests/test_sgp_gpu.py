"""Sparse GP (FITC / VFE) on the GPU through the C ABI vs the CPU oracle (oracle/sgp_oracle.py -- mathematically
pinned, reference-unpinned, see its header).  Tolerances: likelihood 1e-8 relative, predictions 1e-6."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KINDS = ["SquaredExponential", "AbsoluteExponential", "Matern32", "Matern52"]


@pytest.fixture(scope="module")
def egx():
    import egobox_amd
    return egobox_amd


@pytest.fixture(scope="module")
def S():
    from oracle import sgp_oracle
    return sgp_oracle


def _problem(n, nz, d, seed=0, noise=0.05):
    rng = np.random.default_rng(seed)
    x = rng.random((n, d)) * 2 - 1
    y = np.sin(3 * x[:, 0]) + 0.5 * np.cos(2 * x[:, -1]) + noise * rng.standard_normal(n)
    z = x[rng.permutation(n)[:nz]].copy()
    return x, y, z


@pytest.mark.parametrize("method", [0, 1])
@pytest.mark.parametrize("corr", range(4))
def test_likelihood_and_predictions_vs_oracle(egx, S, corr, method):
    x, y, z = _problem(700, 40, 3, seed=corr)
    theta, sigma2, noise = np.array([1.3, 0.8, 1.1]), 0.9, 0.02
    ref = S.SparseGpOracle(x, y, z, theta, sigma2, noise, corr=KINDS[corr], method=[S.FITC, S.VFE][method])
    with egx.SgpHandle(x, y, z, corr=corr, method=method) as h:
        lk, st = h.likelihood(theta, sigma2, noise)
        assert st == 0 and lk == pytest.approx(ref.likelihood, rel=1e-8)
        h.finalize(theta, sigma2, noise)
        xq = np.random.default_rng(9).random((333, 3)) * 2 - 1
        np.testing.assert_allclose(h.predict(xq), ref.predict(xq), rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(h.predict_var(xq), ref.predict_var(xq), rtol=1e-6, atol=1e-9)
        s = h.state(with_inv=True)
        assert s["likelihood"] == lk and s["sigma2"] == sigma2 and s["noise"] == noise
        np.testing.assert_allclose(s["w_vec"], ref.w_vec[:, 0], rtol=1e-6, atol=1e-6 * np.abs(ref.w_vec).max())
        np.testing.assert_allclose(s["w_inv"], ref.w_inv, rtol=1e-6, atol=1e-6 * np.abs(ref.w_inv).max())
        assert h.predict(np.zeros((0, 3))).shape == (0,)


def test_more_inducing_points_than_one_block_and_split_k(egx, S):
    """nz = 300 (z_pad = 384: two Cholesky blocks), n = 5000 (several K splits of the Gram product), d = 6."""
    x, y, z = _problem(5000, 300, 6, seed=5)
    theta, sigma2, noise = np.full(6, 0.7), 1.3, 0.01
    for method, name in ((0, S.FITC), (1, S.VFE)):
        ref = S.SparseGpOracle(x, y, z, theta, sigma2, noise, corr=KINDS[3], method=name)
        with egx.SgpHandle(x, y, z, corr=3, method=method) as h:
            lk, st = h.likelihood(theta, sigma2, noise)
            assert st == 0 and lk == pytest.approx(ref.likelihood, rel=1e-8)
            h.finalize(theta, sigma2, noise)
            xq = np.random.default_rng(1).random((100, 6)) * 2 - 1
            np.testing.assert_allclose(h.predict(xq), ref.predict(xq), rtol=1e-6, atol=1e-8)
            np.testing.assert_allclose(h.predict_var(xq), ref.predict_var(xq), rtol=1e-6, atol=1e-9)


def test_status_and_errors(egx):
    x, y, z = _problem(200, 10, 2)
    with egx.SgpHandle(x, y, z) as h:
        assert h.likelihood([float("nan"), 1.0], 1.0, 0.01)[1] == egx._lib.STATUS_NAN_THETA
        assert h.likelihood([1.0, 1.0], -1.0, 0.01)[1] == egx._lib.STATUS_NAN_THETA
        with pytest.raises(egx.NotFittedError):
            h.predict(x[:2])
        with pytest.raises(egx.InvalidValueError):
            h.likelihood([1.0, 1.0, 1.0], 1.0, 0.01)
    zdup = np.vstack([z, z[:1]])  # duplicated inducing point, no nugget: Kmm singular
    with egx.SgpHandle(x, y, zdup, nugget=0.0) as h:
        lk, st = h.likelihood([1.0, 1.0], 1.0, 0.01)
        assert st == egx._lib.STATUS_NOT_POSITIVE_DEFINITE or np.isfinite(lk)
    with pytest.raises(egx.InvalidValueError):
        egx.SgpHandle(x, y, np.zeros((300, 2)))  # more inducing points than training points


def test_fit_improves_the_likelihood_and_recovers_the_noise_level(egx, S):
    """The reference's tutorial problem shape (doc/SparseGpx_Tutorial.ipynb cells 9-16): 200 noisy samples of a 1-d
    function, 30 inducing points, noise variance estimated.  theta* / likelihood are optimiser- and draw-dependent
    (parity-unpinned); what must hold: the fit is at least as good as its start, matches the oracle at the fitted
    parameters, and the estimated noise is of the right order."""
    rng = np.random.RandomState(0)
    xt = 2 * rng.rand(200, 1) - 1
    f = lambda x: np.sin(3 * np.pi * x) + 0.3 * np.cos(9 * np.pi * x) + 0.5 * np.sin(7 * np.pi * x)  # noqa: E731
    yt = (f(xt) + rng.normal(0.0, np.sqrt(0.01), size=(200, 1))).ravel()
    theta0 = 1.0 / np.std(xt, axis=0) ** 2
    sgp = egx.SparseGaussianProcess.params(egx.Inducings.Randomized(30)).theta_init(theta0) \
        .theta_bounds([(1e-8, 1e2)]).seed(42).n_start(4).fit(xt, yt)
    th, s2, nv, lk = sgp.theta(), sgp.variance(), sgp.noise_variance(), sgp.likelihood()
    ref = S.SparseGpOracle(xt, yt, sgp.inducings(), th, s2, nv)
    # at the optimum the smooth kernel makes Kmm nearly singular (30 points on a line, nugget 2e-14): two correct
    # evaluations agree to ~1e-6 there, not 1e-8 (the well-conditioned cases above hold 1e-8)
    assert lk == pytest.approx(ref.likelihood, rel=1e-5)
    start = S.SparseGpOracle(xt, yt, sgp.inducings(), theta0, float(np.std(yt, ddof=1) ** 2), 1e-2)
    assert lk >= start.likelihood
    assert 1e-3 < nv < 1e-1
    x = np.linspace(-1, 1, 101).reshape(-1, 1)
    assert np.sqrt(np.mean((sgp.predict(x) - f(x).ravel()) ** 2)) < 0.15
    assert "SGP(" in str(sgp)
    sgp.close()
    gx = egx.SparseGpx.builder(nz=30, seed=0, n_start=1).fit(xt, yt.reshape(-1, 1))
    assert gx.thetas().shape == (1, 1) and gx.predict_var(x).shape == (101,)
    with pytest.raises(ValueError):
        egx.SparseGpx.builder(nz=30, seed=0).fit(xt, np.hstack([yt.reshape(-1, 1)] * 2))


def test_numerical_gradients_like_the_reference(egx, S):
    """sparse_algorithm.rs:298-336 differentiates predict / predict_var by central differences with step sqrt(eps)."""
    x, y, z = _problem(400, 25, 2, seed=2)
    theta, sigma2, noise = np.array([1.1, 0.9]), 1.0, 0.02
    ref = S.SparseGpOracle(x, y, z, theta, sigma2, noise, corr=KINDS[3])
    sgp = egx.SparseGaussianProcess(egx.SgpHandle(x, y, z, corr=3), egx.SgpParams(egx.Matern52Corr(), egx.Inducings.Located(z)))
    sgp._h.finalize(theta, sigma2, noise)
    xq = np.random.default_rng(3).random((7, 2)) - 0.5
    h = 1e-5  # a larger step for the oracle's own differences: the two agree to the truncation/rounding error
    for fn_gpu, fn_ref, atol in ((sgp.predict_gradients, ref.predict, 5e-6), (sgp.predict_var_gradients, ref.predict_var, 5e-5)):
        g = fn_gpu(xq)
        assert g.shape == (7, 2)
        for k in range(2):
            dq = np.zeros(2)
            dq[k] = h
            fd = (fn_ref(xq + dq) - fn_ref(xq - dq)) / (2 * h)
            # step sqrt(eps) ~ 1.5e-8: rounding of the predictions (~1e-14 absolute) is amplified by 1 / (2 h) ~ 3e7
            np.testing.assert_allclose(g[:, k], fd, rtol=2e-5, atol=atol)
    sgp.close()


def test_sparse_gpx_save_load_roundtrip(egx, tmp_path):
    x, y, z = _problem(500, 20, 2, seed=4)
    gx = egx.SparseGpx.builder(z=z, n_start=0, max_eval=30, method=egx.SparseMethod.VFE).fit(x, y)
    xq = np.random.default_rng(5).random((40, 2)) * 2 - 1
    y0, v0 = gx.predict(xq), gx.predict_var(xq)
    path = tmp_path / "sgp.json"
    assert gx.save(str(path))
    import json
    d = json.load(open(path))
    e = d["experts"][0]
    assert d["gp_type"] == "SparseGp" and e["method"] == "Vfe" and e["w_data"]["inv"]["dim"] == [20, 20]
    back = egx.SparseGpx.load(str(path))
    np.testing.assert_allclose(back.predict(xq), y0, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(back.predict_var(xq), v0, rtol=1e-9, atol=1e-12)
    assert back.likelihoods()[0] == pytest.approx(gx.likelihoods()[0], rel=1e-10)
    with pytest.raises(NotImplementedError):
        gx.save(str(tmp_path / "sgp.bin"))
