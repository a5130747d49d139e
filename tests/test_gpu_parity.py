"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors.

Run on the GPU box: python -m pytest tests -m gpu.  Tolerances are the ones BASELINE.json states:
1e-8 relative on the log-likelihood, 1e-6 relative on predictions (absolute floor for variances that the
reference clamps at 0, crates/gp/src/algorithm.rs:278).
"""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LK_RTOL = 1e-8
PRED_RTOL = 1e-6


@pytest.fixture(scope="module")
def egx():
    import egobox_amd
    return egobox_amd


@pytest.fixture(scope="module")
def O():
    from oracle import gp_oracle
    return gp_oracle


KINDS = ["SquaredExponential", "AbsoluteExponential", "Matern32", "Matern52"]
MEANS = ["Constant", "Linear", "Quadratic"]


def _data(n, d, seed=0):
    from egobox_amd import workload
    return workload.make_training_set(n, d, seed=seed)


# ------------------------------------------------------------------ kernel level
def test_mfma_layout_probe(egx):
    assert egx.mfma_probe() < 1e-12


@pytest.mark.parametrize("kind", range(4))
@pytest.mark.parametrize("n,d", [(5, 1), (130, 3), (300, 7), (257, 64)])
def test_corr_matrix_matches_value(egx, O, kind, n, d):
    rng = np.random.default_rng(n + d)
    xn, _, _ = O.normalize(rng.random((n, d)))
    theta = 0.2 + rng.random(d)
    got = egx.corr_matrix(kind, xn, theta)
    dd, idx = O.diff_matrix(xn)
    want = O.assemble_r(O.corr_value(KINDS[kind], dd, theta, np.eye(d)), idx, n, O.DEFAULT_NUGGET)
    np.testing.assert_allclose(got, want, rtol=2e-13, atol=1e-300)
    assert np.all(np.diag(got) == 1.0 + O.DEFAULT_NUGGET)


@pytest.mark.parametrize("kind", range(4))
def test_cross_corr_matches_value(egx, O, kind):
    rng = np.random.default_rng(5)
    xt = rng.standard_normal((200, 6))
    xq = rng.standard_normal((131, 6))
    theta = 0.1 + rng.random(6)
    got = egx.cross_corr(kind, xq, xt, theta)
    want = O.corr_value(KINDS[kind], O.pairwise_differences(xq, xt), theta, np.eye(6)).reshape(131, 200)
    np.testing.assert_allclose(got, want, rtol=2e-13, atol=1e-300)


def test_kernel_value_kats(egx, O, golden_dir):
    """The reference's own kernel known-answers (correlation_models.rs:598-641, 719-726) on the GPU."""
    kat = json.load(open(os.path.join(golden_dir, "kat.json")))
    for name, kind in (("sqexp_2d", 0), ("matern32_2d", 2), ("matern52_2d", 3)):
        k = kat[name]
        xt = np.array(k["xt"])
        theta = np.sqrt(k["theta_sq"]) if "theta_sq" in k else np.array(k["theta"])
        r = egx.corr_matrix(kind, xt, theta)
        got = [r[0, 1], r[0, 2], r[1, 2]]
        np.testing.assert_allclose(got, k["expected"], atol=k["tol"], rtol=0)
        np.testing.assert_allclose(got, k["expected"], rtol=2e-8)
    k = kat["sqexp_1d"]
    r = egx.corr_matrix(0, np.array(k["xt"]), np.sqrt(k["theta_sq"]))
    got = [r[i, j] for i in range(5) for j in range(i + 1, 5)]
    np.testing.assert_allclose(got, k["expected"], rtol=1e-13)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 128, 200, 256, 257, 500, 1025])
def test_potrf_vs_lapack(egx, n):
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n))
    spd = a @ a.T + n * np.eye(n)
    got, info = egx.potrf(spd)
    assert info == 0
    want = np.linalg.cholesky(spd)
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-11 * np.abs(want).max())
    assert np.all(np.triu(got, 1) == 0.0)
    # asymmetric check of the factorisation itself (transpose detecting)
    np.testing.assert_allclose(got @ got.T, spd, rtol=1e-12, atol=1e-12 * n)


def test_potrf_reports_first_bad_pivot(egx):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((300, 300))
    spd = a @ a.T + 300 * np.eye(300)
    spd[150, 150] = -1.0
    _, info = egx.potrf(spd)
    assert info == 151
    _, info = egx.potrf(np.array([[1.0, 2.0], [2.0, 1.0]]))
    assert info == 2


# ------------------------------------------------------------------ golden vectors through the C ABI
def test_golden_b_through_cabi(egx, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "golden_b.json")))
    gp = egx.GaussianProcess.params(egx.LinearMean(), egx.Matern52Corr()).nugget(g["nugget"]) \
        .theta_tuning(egx.ThetaTuning.Fixed(g["theta"])).fit(np.array(g["training_x"]), np.array(g["training_y"]))
    ip = gp.inner_params(with_chol=True)
    assert ip["likelihood"] == pytest.approx(g["likelihood"], rel=1e-12)
    assert ip["sigma2"] == pytest.approx(g["sigma2"], rel=1e-11)
    np.testing.assert_allclose(ip["r_chol"], np.array(g["r_chol"]), atol=1e-14)
    np.testing.assert_allclose(ip["ft"], np.array(g["ft"]), atol=1e-13)
    np.testing.assert_allclose(ip["ft_qr_r"], np.array(g["ft_qr_r"]), atol=1e-13)
    np.testing.assert_allclose(ip["beta"], np.array(g["beta"]), atol=1e-13)
    np.testing.assert_allclose(ip["gamma"], np.array(g["gamma"]), atol=1e-12)
    np.testing.assert_allclose(ip["xt_norm"], np.array(g["xt_norm"]["data"]), atol=1e-15)
    np.testing.assert_allclose(ip["x_std"], g["xt_norm"]["std"], rtol=1e-15)
    np.testing.assert_allclose(ip["y_mean"], g["yt_norm"]["mean"], rtol=1e-14)
    gp.close()


def test_golden_a_and_python_pins_through_gpx(egx, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "golden_a.json")))
    k = json.load(open(os.path.join(golden_dir, "kat.json")))["python_kriging"]
    gpx = egx.Gpx.builder(theta_init=[g["theta_printed_8_digits"]], n_start=-1).fit(np.array(g["xt"]), np.array(g["yt"]))
    assert gpx.likelihoods()[0] == pytest.approx(g["likelihood"], abs=1e-12)
    assert gpx.variances()[0] == pytest.approx(g["variance"], rel=1e-8)
    assert gpx.thetas()[0, 0] == g["theta_printed_8_digits"]
    assert gpx.dims() == (1, 1)
    assert gpx.predict(np.array([[1.0]]))[0] == pytest.approx(k["predict_1.0"], abs=1e-7)
    assert gpx.predict_var(np.array([[1.0]]))[0] == pytest.approx(k["var_1.0"], abs=1e-7)
    assert gpx.predict(np.array([[1.1]]))[0] == pytest.approx(k["predict_1.1"], abs=k["delta"])
    assert gpx.predict_var(np.array([[1.1]]))[0] == pytest.approx(k["var_1.1"], abs=k["delta"])
    # python/egobox/tests/test_gpmix.py:47-52
    assert gpx.predict_gradients(np.array([[1.1]]))[0, 0] == pytest.approx(k["predict_gradients_1.1"], abs=k["delta"])
    assert gpx.predict_var_gradients(np.array([[1.1]]))[0, 0] == pytest.approx(k["predict_var_gradients_1.1"], abs=k["delta"])
    xd, yd = gpx.training_data()
    np.testing.assert_array_equal(xd[:, 0], g["xt"])
    np.testing.assert_array_equal(yd, g["yt"])


# ------------------------------------------------------------------ likelihood / fit parity vs the oracle
CASES = [
    # n, d, mean, corr, theta scale (x 0.5/sqrt(d)); scales chosen so the oracle's smallest Cholesky pivot is
    # >> sqrt(nugget) and the oracle agrees with itself under a row permutation to < 1e-10:
    # 1e-8 parity is only meaningful where the problem is well posed.
    (5, 1, 0, 0, 1.0), (3, 1, 0, 0, 1.0), (127, 2, 0, 3, 3.0), (128, 2, 1, 0, 8.0), (129, 3, 0, 2, 1.0),
    (257, 4, 2, 0, 4.0), (640, 5, 1, 3, 1.0), (1000, 8, 0, 0, 1.0), (1500, 16, 0, 3, 1.0), (2048, 8, 0, 1, 1.0),
    (1111, 32, 1, 2, 1.0), (900, 64, 0, 0, 1.0),
]


def _check_fit(egx, O, x, y, theta, mean, corr, lk_rtol=LK_RTOL, inner=True):
    d = x.shape[1]
    ref = O.fit_fixed(x, y, theta, mean=MEANS[mean], corr=KINDS[corr])
    with egx.GpHandle(x, y, mean=mean, corr=corr) as h:
        lk, st = h.likelihood(theta)
        assert st == 0
        assert lk == pytest.approx(ref.likelihood, rel=lk_rtol)
        h.finalize(theta)
        ip = h.inner(with_chol=True)
        assert ip["likelihood"] == pytest.approx(ref.likelihood, rel=lk_rtol)
        if inner:
            assert ip["sigma2"] == pytest.approx(ref.inner.sigma2, rel=1e-7)
            scale = np.abs(ref.inner.r_chol).max()
            np.testing.assert_allclose(ip["r_chol"], ref.inner.r_chol, rtol=1e-7, atol=1e-9 * scale)
            np.testing.assert_allclose(ip["beta"], ref.inner.beta, rtol=1e-6, atol=1e-8 * np.abs(ref.inner.beta).max())
            np.testing.assert_allclose(ip["gamma"], ref.inner.gamma, rtol=1e-5,
                                       atol=1e-6 * np.abs(ref.inner.gamma).max())
            np.testing.assert_allclose(np.abs(ip["ft_qr_r"]), np.abs(ref.inner.ft_qr_r), rtol=1e-7,
                                       atol=1e-9 * np.abs(ref.inner.ft_qr_r).max())
        assert np.all(np.diag(ip["ft_qr_r"]) > 0)
        assert np.all(np.triu(ip["r_chol"], 1) == 0.0)
        # predictions (also exercises chunk tails: 333 is not a multiple of any tile)
        xq = x.min(0) + (x.max(0) - x.min(0)) * np.random.default_rng(7).random((333, d))
        yp, vp = h.predict_valvar(xq)
        if inner:
            yr, vr = ref.predict(xq), ref.predict_var(xq)
            np.testing.assert_allclose(yp, yr, rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(yr).max())
            np.testing.assert_allclose(vp, vr, rtol=PRED_RTOL,
                                       atol=1e-9 * ref.inner.sigma2 + PRED_RTOL * np.abs(vr).max())
        np.testing.assert_array_equal(h.predict(xq), yp)  # predict_valvar == (predict, predict_var)
        np.testing.assert_array_equal(h.predict_var(xq), vp)  # crates/moe/src/algorithm.rs:1548-1554
        assert np.all(vp >= 0.0)
    return ref


@pytest.mark.parametrize("n,d,mean,corr,scale", CASES)
def test_fixed_theta_fit_matches_oracle(egx, O, n, d, mean, corr, scale):
    x, y = _data(n, d, seed=n)
    theta = scale * egx.workload.default_theta(d) * (1.0 + 0.3 * np.arange(d) / max(1, d - 1))
    _check_fit(egx, O, x, y, theta, mean, corr)


def test_config1_kriging_n256_d1(egx, O):
    """BASELINE configs[0] shape: Kriging (constant mean, squared exponential), n=256, d=1, through the
    same calls as crates/gp/examples/kriging.rs (regular grid so that the problem is well posed)."""
    x = np.linspace(0.0, 1.0, 256).reshape(-1, 1)
    y = np.sin(12.0 * x[:, 0]) + x[:, 0] ** 2
    ref = _check_fit(egx, O, x, y, np.array([60.0]), 0, 0)
    assert np.diag(ref.inner.r_chol).min() > 0.1
    gp = egx.Kriging.params().theta_tuning(egx.ThetaTuning.Fixed([60.0])).fit(x, y)
    assert gp.likelihood() == pytest.approx(ref.likelihood, rel=LK_RTOL)
    assert gp.variance() == pytest.approx(ref.inner.sigma2, rel=1e-7)
    gp.close()


def test_ill_conditioned_case_within_oracle_self_consistency(egx, O):
    """n=256 LHS points in 1-D with a smooth kernel: cond(R) ~ 1/nugget.  Two LAPACK-grade evaluations of
    the SAME likelihood (rows permuted) differ by ~1e-2 relative here, so 1e-8 parity is undefined; assert
    instead that the GPU is as close to the oracle as the oracle is to itself, and that nothing blows up."""
    x, y = _data(256, 1, seed=256)
    theta = np.array([0.5])
    ref = O.fit_fixed(x, y, theta)
    perm = np.random.default_rng(1).permutation(256)
    ref2 = O.fit_fixed(x[perm], y[perm], theta)
    self_err = abs(ref.likelihood - ref2.likelihood) / abs(ref.likelihood)
    with egx.GpHandle(x, y) as h:
        lk, st = h.likelihood(theta)
    if st == 0:
        assert abs(lk - ref.likelihood) / abs(ref.likelihood) <= max(1e-8, 50.0 * self_err)
    else:
        assert st == egx._lib.STATUS_NOT_POSITIVE_DEFINITE


def test_config2_dense_sqexp_n4096_d8(egx, O, arbiter):
    """BASELINE configs[1] exactly as SURVEY 8d states it: n=4096, d=8 squared exponential, theta_j = 0.5/sqrt(d).
    The smallest pivot is ~5e-6 (only ~30x sqrt(nugget)): two double-precision evaluations of this likelihood differ
    by ~1e-6 (LAPACK disagrees with itself under a row permutation), so the judge is the extended-precision ARBITER
    (oracle/arbiter_ld.c, x87 long double end to end, tests/golden/arbiter.json): the HIP path must be no further from
    it than 10x LAPACK's own distance (the larger of its two recorded ones), both printed."""
    x, y = _data(4096, 8, seed=42)
    theta = egx.workload.default_theta(8)
    rec, recp = arbiter["config2_n4096_d8_sqexp"], arbiter["perm_n4096_d8_sqexp"]
    np.testing.assert_array_equal(theta, np.array(rec["theta"]))
    truth = rec["truth_likelihood"]
    assert abs(recp["truth_likelihood"] - truth) / abs(truth) < 1e-8   # the arbiter does not care about the row order
    ref = O.fit_fixed(x, y, theta)
    lap_err = abs(ref.likelihood - truth) / abs(truth)
    assert lap_err == pytest.approx(rec["lapack_rel_error"], rel=0.5)  # (what the fixture's generator saw; BLAS build dependent)
    lap_worst = max(lap_err, rec["lapack_rel_error"], recp["lapack_rel_error"])
    with egx.GpHandle(x, y) as h:
        h.finalize(theta)
        lk, s2 = h.fitted_scalars()
        gpu_err = abs(lk - truth) / abs(truth)
        print(f"config2 at theta = 0.5/sqrt(d): long double {truth!r}; LAPACK {ref.likelihood!r} rel err {lap_err:.2e} "
              f"(permuted rows {recp['lapack_rel_error']:.2e}); HIP {lk!r} rel err {gpu_err:.2e}")
        assert gpu_err <= max(LK_RTOL, 10.0 * lap_worst)
        tol = max(LK_RTOL, 10.0 * lap_worst)
        assert s2 == pytest.approx(rec["truth_sigma2"], rel=max(1e-7, 100 * tol))
        xq = np.random.default_rng(3).random((1000, 8))
        yr, vr = ref.predict(xq), ref.predict_var(xq)
        np.testing.assert_allclose(h.predict(xq), yr, rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(yr).max())
        np.testing.assert_allclose(h.predict_var(xq), vr, rtol=100 * tol, atol=100 * tol * vr.max())


def test_config2_size_well_conditioned_matern(egx, O):
    """Same size with the Matern-5/2 kernel (well conditioned): the straight 1e-8 / 1e-6 bars."""
    x, y = _data(4096, 8, seed=43)
    theta = egx.workload.default_theta(8)
    ref = O.fit_fixed(x, y, theta, corr="Matern52")
    with egx.GpHandle(x, y, corr=3) as h:
        h.finalize(theta)
        lk, s2 = h.fitted_scalars()
        assert lk == pytest.approx(ref.likelihood, rel=LK_RTOL)
        assert s2 == pytest.approx(ref.inner.sigma2, rel=1e-7)
        xq = np.random.default_rng(3).random((1000, 8))
        np.testing.assert_allclose(h.predict(xq), ref.predict(xq), rtol=PRED_RTOL)
        vr = ref.predict_var(xq)
        np.testing.assert_allclose(h.predict_var(xq), vr, rtol=PRED_RTOL, atol=PRED_RTOL * vr.max())


def test_len1_theta_broadcast(egx, O):
    x, y = _data(300, 4, seed=2)
    with egx.GpHandle(x, y) as h:
        lk1, _ = h.likelihood([0.4])
        lk4, _ = h.likelihood([0.4] * 4)
        assert lk1 == lk4
        with pytest.raises(egx.InvalidValueError):
            h.likelihood([0.4, 0.4])


def test_kpls_weights_collapse(egx, O):
    """w_star path (KPLS generalisation) for all four kernels."""
    x, y = _data(400, 5, seed=9)
    rng = np.random.default_rng(1)
    w = rng.standard_normal((5, 2))
    theta = np.array([0.7, 0.3])
    for corr in range(4):
        ref = O.fit_fixed(x, y, theta, corr=KINDS[corr], w_star=w)
        with egx.GpHandle(x, y, corr=corr, w_star=w) as h:
            assert h.h == 2
            lk, st = h.likelihood(theta)
            assert st == 0 and lk == pytest.approx(ref.likelihood, rel=LK_RTOL)
            h.finalize(theta)
            xq = rng.random((50, 5))
            np.testing.assert_allclose(h.predict(xq), ref.predict(xq), rtol=PRED_RTOL)


# ------------------------------------------------------------------ status channel / errors
def test_status_channel(egx, O):
    x = np.array([[0.0], [1.0], [1.0], [2.0], [3.0]])  # duplicated row
    y = np.array([0.0, 1.0, 1.0, 0.5, 0.2])
    with egx.GpHandle(x, y, nugget=-1e-3) as h:  # duplicate rows + negative nugget: pivot < 0
        lk, st = h.likelihood([1.0])
        assert st == egx._lib.STATUS_NOT_POSITIVE_DEFINITE and lk == -math.inf
        with pytest.raises(egx.LinalgError):
            h.finalize([1.0])
        with pytest.raises(egx.NotFittedError):
            h.predict(np.array([[0.5]]))
        lk, st = h.likelihood([float("nan")])
        assert st == egx._lib.STATUS_NAN_THETA and lk == -math.inf
    # ill-conditioned regression basis: quadratic mean on collinear data
    xc = np.linspace(0, 1, 40).reshape(-1, 1) * np.ones((1, 2))
    yc = np.sin(xc[:, 0])
    ref_lk, ref_st = O.likelihood_at(xc, yc, [0.5, 0.5], mean="Quadratic")
    with egx.GpHandle(xc, yc, mean=2) as h:
        lk, st = h.likelihood([0.5, 0.5])
        assert ref_st in (2, 3)
        assert st in (egx._lib.STATUS_ILL_CONDITIONED_F, egx._lib.STATUS_ILL_CONDITIONED_FT)
        with pytest.raises(egx.LikelihoodComputationError):
            h.finalize([0.5, 0.5])


def test_batch_equals_single_and_uses_workspaces(egx):
    x, y = _data(700, 6, seed=4)
    thetas = egx.theta_sweep_candidates(7, 6, seed=3)
    with egx.GpHandle(x, y, n_workspaces=3) as h:
        lkb, stb = h.likelihood_batch(thetas)
        for i in range(7):
            lk, st = h.likelihood(thetas[i])
            assert st == stb[i]
            if st == 0:
                assert lk == lkb[i]  # same kernels, same order: bit identical


@pytest.mark.parametrize("n,d,mean,corr", [(700, 6, 0, 0), (2100, 5, 1, 3), (4200, 8, 0, 0)])
def test_lockstep_groups_are_bit_identical_to_single_candidates(egx, n, d, mean, corr):
    """Round 3: the candidates of a batch are factored in LOCK-STEP groups by one launch sequence (grid.z = candidate).
    A candidate must get the bits it gets alone -- whatever the group width, its slot in the group, a ragged last
    group, a NaN theta or a not-positive-definite companion next to it (n = 4200 crosses the look-ahead and the
    stream-kernel thresholds, the linear mean takes the device GLS route)."""
    x, y = _data(n, d, seed=11)
    k = 7 if n > 3000 else 11
    thetas = egx.theta_sweep_candidates(k, d, seed=5)
    thetas[3] = np.nan                      # answered at once, takes no slot
    thetas[5] = 1e-3 if corr == 0 else thetas[5]   # sq-exp with tiny theta: R ~ all ones, not positive definite
    with egx.GpHandle(x, y, mean=mean, corr=corr, n_workspaces=6) as h:
        h.set_lockstep(1)
        ref_lk, ref_st = h.likelihood_batch(thetas)
        assert ref_st[3] == egx._lib.STATUS_NAN_THETA
        for width in (2, 3, 4, 6):
            assert h.set_lockstep(width) == width
            lk, st = h.likelihood_batch(thetas)
            np.testing.assert_array_equal(st, ref_st)
            ok = st == 0
            assert ok.sum() >= 3
            np.testing.assert_array_equal(lk[ok], ref_lk[ok])   # bit identical
        assert h.set_lockstep(0) == 6   # the default: min(n_workspaces, 12) below n_pad 14336
        # a fit keeps workspace 0; the batch then uses slots of the remaining five
        h.finalize(thetas[0])
        lk, st = h.likelihood_batch(thetas)
        np.testing.assert_array_equal(lk[st == 0], ref_lk[ref_st == 0])
        assert h.fitted_scalars()[0] == ref_lk[0]


def test_one_shot_fits_are_warm_fits_and_the_pool_is_bounded(egx):
    """The reference creates a model per fit (algorithm.rs:785-794; one per expert in egobox-moe).  Destroyed handles
    leave their device resources in the library's pool: 20 create -> fit -> destroy cycles on DIFFERENT data of one
    shape must (a) hit the pool from the second cycle on, (b) cost at most 1.2x a fit on a resident handle plus the
    upload, (c) not grow device memory, (d) give the results a fresh process gives; egx_trim returns the memory."""
    import time
    import torch
    n, d = 4096, 8
    theta = egx.workload.default_theta(d) * 4.0
    egx.trim()
    x0, y0 = _data(n, d, seed=100)
    with egx.GpHandle(x0, y0) as h:           # resident handle: the yardstick
        h.finalize(theta)
        t0 = time.perf_counter()
        for _ in range(5):
            h.finalize(theta)
        resident = (time.perf_counter() - t0) / 5
        ref_lk = h.fitted_scalars()[0]
    # The baseline is taken with the pool EMPTY but after this shape's first use: what the HIP runtime allocates once per
    # hardware queue -- code objects, and since round 5 the scratch buffer in which the chain kernel's called roles save
    # registers (168 bytes x 64 lanes x every wave slot of the chip, ~90 MB per queue) -- is not the pool's to return.
    egx.trim()
    free0 = torch.cuda.mem_get_info()[0]
    with egx.GpHandle(x0, y0) as h:           # (leaves its resources in the pool for the cycles below)
        h.finalize(theta)
    stats0 = egx.pool_stats()
    assert stats0["cached_bytes"] > 0
    cycle, used = [], []
    for c in range(20):
        x, y = _data(n, d, seed=100 + c % 3)
        t0 = time.perf_counter()
        h = egx.GpHandle(x, y)
        h.finalize(theta)
        lk = h.fitted_scalars()[0]
        h.close()
        cycle.append(time.perf_counter() - t0)
        used.append(free0 - torch.cuda.mem_get_info()[0])
        if c % 3 == 0:
            assert lk == ref_lk               # same data, same bits, whatever the workspace held before
    stats = egx.pool_stats()
    assert stats["hits"] - stats0["hits"] == 20 and stats["misses"] == stats0["misses"]
    print(f"resident fit {resident * 1e3:.2f} ms, create+fit+destroy cycles {np.median(cycle) * 1e3:.2f} ms (median), "
          f"{max(cycle) * 1e3:.2f} ms (max)")
    assert np.median(cycle) <= 1.2 * resident + 2e-3   # + normalisation, the k-major copies and two uploads on the host
    assert max(used) - min(used) < 64 << 20            # no growth
    freed = egx.trim()
    assert freed >= stats["cached_bytes"] > 0 and egx.pool_stats()["cached_bytes"] == 0
    # (200 MB, not 64: when this test is the first of its process the cycles may have touched one or two hardware queues more
    #  than the baseline had -- ~90 MB of runtime scratch each, see above; a leak PER CYCLE is what the line before the trim catches)
    assert free0 - torch.cuda.mem_get_info()[0] < 200 << 20


def test_concurrent_likelihood_calls_on_one_handle(egx):
    """SURVEY 8b threading: the objective closure is called concurrently from rayon workers on the same training
    set -> egx_gp_likelihood from many threads must overlap on the workspace pool and return what the serial
    calls return; a fitted model in workspace 0 survives as long as another workspace is free."""
    from concurrent.futures import ThreadPoolExecutor
    x, y = _data(900, 5, seed=21)
    thetas = egx.theta_sweep_candidates(24, 5, seed=5)
    with egx.GpHandle(x, y, corr=2, n_workspaces=3) as h:
        serial = [h.likelihood(t) for t in thetas]
        h.finalize(thetas[0])
        yfit = h.predict(x[:5])
        with ThreadPoolExecutor(2) as pool:  # 2 threads, workspaces 2 and 1: workspace 0 (the fit) is never taken
            par = list(pool.map(h.likelihood, thetas))
        assert par == serial
        np.testing.assert_array_equal(h.predict(x[:5]), yfit)
        with ThreadPoolExecutor(6) as pool:  # more threads than workspaces: callers queue on the pool
            par = list(pool.map(h.likelihood, thetas))
        assert par == serial


# ------------------------------------------------------------------ new capability: theta gradient
@pytest.mark.parametrize("corr", range(4))
def test_likelihood_gradient(egx, O, corr):
    x, y = _data(300, 3, seed=6)
    theta = np.array([0.8, 1.3, 0.5]) * (4.0 if corr == 0 else 1.0)  # keep the smooth kernel well conditioned
    lk_ref, g_ref = O.likelihood_grad(x, y, theta, corr=KINDS[corr])
    with egx.GpHandle(x, y, corr=corr) as h:
        lk, g, st = h.likelihood_grad(theta)
        assert st == 0 and lk == pytest.approx(lk_ref, rel=LK_RTOL)
        np.testing.assert_allclose(g, g_ref, rtol=1e-6, atol=1e-6 * np.abs(g_ref).max())
        # and against finite differences of the parity-checked likelihood
        for k in range(3):
            e = np.zeros(3)
            e[k] = 1e-5
            fd = (h.likelihood(theta + e)[0] - h.likelihood(theta - e)[0]) / 2e-5
            assert g[k] == pytest.approx(fd, rel=1e-4, abs=1e-5)


# ------------------------------------------------------------------ tuned fit
def test_full_fit_improves_likelihood(egx):
    x, y = _data(200, 2, seed=8)
    gp = egx.Kriging.params().n_start(3).max_eval(60).fit(x, y)
    with egx.GpHandle(x, y) as h:
        lk0, _ = h.likelihood([0.1])
    assert gp.likelihood() >= lk0 - 1e-9
    th = gp.theta()
    assert np.all(th >= 1e-2 * (1 - 1e-12)) and np.all(th <= 1e1 * (1 + 1e-12))
    assert gp.n_evals >= 4 and np.all(np.isfinite(gp.predict(x)))
    with egx.GpHandle(x, y) as h:  # the stored likelihood is the likelihood at the stored theta
        lk1, st1 = h.likelihood(th)
    assert st1 == 0 and lk1 == gp.likelihood()
    gp.close()


def test_default_tuned_fit_lands_on_the_reference_theta_golden_a(egx, golden_dir):
    """The reference's notebook (doc/Gpx_Tutorial.ipynb cells 9-14): default Kriging fit (11 COBYLA runs) prints
    theta* = 1.83209405, likelihood 0.57817407, variance 0.30494059.  egx_gp_fit (csrc/cobyla.h, all starts in
    lock-step through one likelihood batch per round) lands on the same optimum: theta to 1e-3 relative -- ftol_rel =
    1e-4 stops that early --, likelihood to 1e-6, variance to 1e-3."""
    with open(os.path.join(golden_dir, "golden_a.json")) as f:
        ga = json.load(f)
    xt, yt = np.array(ga["xt"]).reshape(-1, 1), np.array(ga["yt"])
    gp = egx.Gpx.builder().fit(xt, yt)
    assert gp.thetas()[0][0] == pytest.approx(ga["theta_printed_8_digits"], rel=1e-3)
    assert gp.likelihoods()[0] == pytest.approx(ga["likelihood"], abs=1e-6)
    assert gp.variances()[0] == pytest.approx(ga["variance"], rel=1e-3)
    e = gp._experts[0]
    assert 11 <= e.n_evals <= 11 * 25   # clamp(10 h, 25, max_eval = 50) per start, algorithm.rs:933-936
    e.close()


def test_multistart_threads_are_deterministic(egx):
    """Starts are independent optimisations reduced by min: running them on 1 or 3 workspaces (host threads +
    streams) must give the same theta and likelihood."""
    x, y = _data(300, 3, seed=12)
    res = []
    for nws in (1, 3):
        gp = egx.GaussianProcess.params(egx.ConstantMean(), egx.Matern52Corr()).n_start(4).max_eval(40) \
            .n_workspaces(nws).fit(x, y)
        res.append((gp.theta().copy(), gp.likelihood(), gp.n_evals))
        gp.close()
    np.testing.assert_array_equal(res[0][0], res[1][0])
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]


def test_api_state_semantics(egx):
    x, y = _data(200, 2, seed=13)
    with egx.GpHandle(x, y, corr=3) as h:  # Matern-5/2: well conditioned at these thetas
        assert (h.n, h.d, h.p, h.h) == (200, 2, 1, 2)
        with pytest.raises(egx.NotFittedError):
            h.inner()
        h.finalize([0.8, 0.9])
        y0 = h.predict(x[:3])
        assert h.predict(np.zeros((0, 2))).shape == (0,)                # empty query batch
        np.testing.assert_array_equal(h.predict(x[:1]), y0[:1])          # single point == batched
        h.finalize([0.8, 0.9])                                            # re-fit is idempotent (bit identical)
        np.testing.assert_array_equal(h.predict(x[:3]), y0)
        h.likelihood([0.5, 0.5])                                          # re-uses workspace 0: the fit is gone
        with pytest.raises(egx.NotFittedError):
            h.predict(x[:3])
        with pytest.raises(egx.InvalidValueError):
            h.predict(np.zeros((3, 5)))
    # concurrent calls on one handle from several host threads are serialised, not corrupted
    from concurrent.futures import ThreadPoolExecutor
    with egx.GpHandle(x, y, corr=3, n_workspaces=2) as h:
        ths = [np.array([0.3 + 0.1 * i, 0.7]) for i in range(6)]
        want = [h.likelihood(t)[0] for t in ths]
        with ThreadPoolExecutor(3) as pool:
            got = list(pool.map(lambda t: h.likelihood(t)[0], ths))
        assert got == want


def test_gpx_save_load_roundtrip(egx, tmp_path):
    x, y = _data(50, 2, seed=11)
    gpx = egx.Gpx.builder(regr_spec=egx.RegressionSpec.LINEAR, corr_spec=egx.CorrelationSpec.MATERN52,
                          theta_init=[0.5, 0.7], n_start=-1).fit(x, y)
    f = str(tmp_path / "gp.json")
    assert gpx.save(f)
    d = json.load(open(f))
    assert d["experts"][0]["type_fullgp"] == "GpLinearMatern52Surrogate"
    xq = np.random.default_rng(0).random((20, 2))
    for refit in (False, True):  # upload the stored factor as is / re-factor at the stored theta
        gpx2 = egx.Gpx.load(f, refit=refit)
        np.testing.assert_allclose(gpx2.predict(xq), gpx.predict(xq), rtol=1e-12)
        np.testing.assert_allclose(gpx2.predict_var(xq), gpx.predict_var(xq), rtol=1e-9, atol=1e-12 * gpx.variances()[0])
        assert gpx2.likelihoods()[0] == pytest.approx(gpx.likelihoods()[0], rel=1e-12)
    assert "Mixture[Hard](Linear_Matern52GP(mean=LinearMean, corr=Matern52" in str(gpx)


def test_load_reference_serialized_model(egx, O, golden_dir):
    """A model serialized by the REFERENCE (egobox 0.32.0, doc/Gpx_Tutorial.ipynb cell 31; same serde schema as
    crates/moe/src/surrogates.rs save/load) is ingested as is -- stored r_chol, gamma, beta, ft, ft_qr_r uploaded,
    nothing re-factored -- and predicts like the oracle evaluated on the reference's own stored parameters."""
    g = json.load(open(os.path.join(golden_dir, "golden_b.json")))
    nd = lambda a: {"v": 1, "dim": list(np.shape(a)), "data": np.asarray(a, dtype=float).ravel().tolist()}
    expert = {
        "type_fullgp": g["type_fullgp"], "theta": nd(g["theta"]), "likelihood": g["likelihood"],
        "inner_params": {"sigma2": g["sigma2"], "beta": nd(g["beta"]), "gamma": nd(g["gamma"]),
                         "r_chol": nd(g["r_chol"]), "ft": nd(g["ft"]), "ft_qr_r": nd(g["ft_qr_r"])},
        "w_star": nd(g["w_star"]),
        "xt_norm": {k: nd(v) for k, v in g["xt_norm"].items()}, "yt_norm": {k: nd(v) for k, v in g["yt_norm"].items()},
        "training_data": [nd(g["training_x"]), nd(g["training_y"])],
        "params": {"theta_tuning": {"Full": {}}, "mean": "LinearMean", "corr": "Matern52", "kpls_dim": None,
                   "n_start": 10, "max_eval": 1000, "nugget": g["nugget"]},
    }
    gpx = egx.Gpx.from_dict({"recombination": "Hard", "experts": [expert], "gp_type": "FullGp"})
    assert gpx.likelihoods()[0] == g["likelihood"] and gpx.variances()[0] == g["sigma2"]
    assert gpx.thetas()[0, 0] == g["theta"][0]
    # oracle object carrying the reference's stored parameters (no fitting anywhere)
    xn, ym, ys = np.array(g["xt_norm"]["data"]), np.array(g["yt_norm"]["mean"]), np.array(g["yt_norm"]["std"])
    inner = O.GpInnerParams(sigma2=g["sigma2"], beta=np.array(g["beta"]), gamma=np.array(g["gamma"]),
                            r_chol=np.array(g["r_chol"]), ft=np.array(g["ft"]), ft_qr_r=np.array(g["ft_qr_r"]))
    ref = O.GaussianProcessOracle(theta=np.array(g["theta"]), likelihood=g["likelihood"], inner=inner,
                                  w_star=np.eye(1), xt_norm=xn, x_mean=np.array(g["xt_norm"]["mean"]),
                                  x_std=np.array(g["xt_norm"]["std"]), yt_norm=np.array(g["yt_norm"]["data"]),
                                  y_mean=ym, y_std=ys, mean=O.LINEAR, corr=O.MATERN52, nugget=g["nugget"])
    xq = np.linspace(-10, 10, 500).reshape(-1, 1)  # the notebook's own plotting grid (cell 37)
    yr, vr = ref.predict(xq), ref.predict_var(xq)
    np.testing.assert_allclose(gpx.predict(xq), yr, rtol=1e-9, atol=1e-9 * np.abs(yr).max())
    np.testing.assert_allclose(gpx.predict_var(xq), vr, rtol=1e-7, atol=1e-9 * g["sigma2"])


# ------------------------------------------------------------------ full size (BASELINE metric size): properties
@pytest.mark.parametrize("corr", [0, 3])
def test_full_size_properties_n16384_d32(egx, corr):
    """No oracle run at this size fits the test budget; assert size-independent properties instead:
    permutation invariance of the likelihood, interpolation at training points, zero variance there,
    and agreement between the batch and single-candidate paths."""
    n, d = 16384, 32
    x, y = _data(n, d, seed=42)
    theta = egx.workload.default_theta(d)
    perm = np.random.default_rng(0).permutation(n)
    with egx.GpHandle(x, y, corr=corr) as h:
        h.finalize(theta)
        lk, s2 = h.fitted_scalars()
        assert np.isfinite(lk) and s2 > 0
        idx = np.arange(0, n, 97)
        yp, vp = h.predict_valvar(x[idx])
        np.testing.assert_allclose(yp, y[idx], rtol=1e-6, atol=1e-6 * np.abs(y).max())
        assert np.all(vp >= 0) and np.all(vp <= 1e-6 * s2)
        lkb, stb = h.likelihood_batch(theta[None, :])
        assert stb[0] == 0 and lkb[0] == lk
    with egx.GpHandle(x[perm], y[perm], corr=corr) as h2:
        lk2, st2 = h2.likelihood(theta)
        assert st2 == 0 and lk2 == pytest.approx(lk, rel=LK_RTOL)


# ------------------------------------------------------------------ widening: mixture of GPU experts
@pytest.mark.parametrize("recomb", ["smooth", "hard"])
def test_mixture_of_gpu_experts_matches_oracle(egx, O, recomb):
    from egobox_amd.moe import GaussianMixture, GpMixture
    from oracle import moe_oracle as MO
    rng = np.random.default_rng(1)
    gpu_experts, cpu_experts = [], []
    for c in range(3):
        x = rng.random((300, 2)) + [c, 0.0]
        y = np.sin(3 * x[:, 0]) + x[:, 1] * (c + 1)
        theta = [1.5, 1.0]
        gpu_experts.append(egx.GaussianProcess.params(egx.ConstantMean(), egx.Matern52Corr())
                           .theta_tuning(egx.ThetaTuning.Fixed(theta)).fit(x, y))
        cpu_experts.append(O.fit_fixed(x, y, theta, corr=O.MATERN52))
    means = np.array([[0.5, 0.5], [1.5, 0.5], [2.5, 0.5]])
    covs = np.array([np.eye(2) * 0.2] * 3)
    w = np.array([0.3, 0.3, 0.4])
    mix = GpMixture(gpu_experts, GaussianMixture(w, means, covs, 0.8), recomb)
    gmo = MO.GaussianMixtureOracle(w, means, covs, 0.8)
    xq = np.random.default_rng(2).random((500, 2)) * [3.0, 1.0]
    val, var = mix.predict_valvar(xq)
    if recomb == "smooth":
        want_val, want_var = MO.predict_smooth(cpu_experts, gmo, xq), MO.predict_var_smooth(cpu_experts, gmo, xq)
    else:
        want_val, want_var = MO.predict_hard(cpu_experts, gmo, xq), MO.predict_var_hard(cpu_experts, gmo, xq)
    np.testing.assert_allclose(val, want_val, rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(want_val).max())
    np.testing.assert_allclose(var, want_var, rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(want_var).max() + 1e-12)
    for e in gpu_experts:
        e.close()


# ------------------------------------------------------------------ more edge cases
def test_quadratic_mean_with_more_than_128_basis_columns(egx, O):
    """Quadratic mean at d = 15 has p = 136 basis columns: the appended right-hand-side block is 256 rows."""
    rng = np.random.default_rng(21)
    x = rng.random((700, 15))
    y = np.sin(x).sum(axis=1) + (x[:, :3] ** 2).sum(axis=1)
    theta = np.full(15, 0.6)
    ref = O.fit_fixed(x, y, theta, mean="Quadratic", corr="Matern52")
    with egx.GpHandle(x, y, mean=2, corr=3) as h:
        assert h.p == 136
        lk, st = h.likelihood(theta)
        assert st == 0 and lk == pytest.approx(ref.likelihood, rel=1e-7)  # p x p GLS conditioning, not the kernels
        h.finalize(theta)
        xq = rng.random((200, 15))
        np.testing.assert_allclose(h.predict(xq), ref.predict(xq), rtol=1e-6, atol=1e-7)
        vr = ref.predict_var(xq)
        np.testing.assert_allclose(h.predict_var(xq), vr, rtol=1e-5, atol=1e-6 * vr.max())


def test_duplicate_training_points_follow_the_oracle_status(egx, O):
    """Collisions: the reference only warns on duplicated rows (algorithm.rs:857-865); with the default nugget the
    factorisation may or may not survive -- either way the status channel must agree with a LAPACK evaluation in
    kind (0 or NOT_POSITIVE_DEFINITE) and never raise."""
    rng = np.random.default_rng(5)
    x = rng.random((300, 3))
    x[17] = x[3]
    x[250] = x[100]
    y = x.sum(axis=1)
    y[17], y[250] = y[3], y[100]
    for corr in (0, 3):
        lk_ref, st_ref = O.likelihood_at(x, y, [1.0, 1.0, 1.0], corr=KINDS[corr])
        with egx.GpHandle(x, y, corr=corr) as h:
            lk, st = h.likelihood([1.0, 1.0, 1.0])
        assert st in (0, 1) and st_ref in (0, 1)
        if st == 0:
            assert np.isfinite(lk)


def test_non_power_of_two_large_n_properties(egx):
    """n = 20000 (n_pad = 20096, 3.2 GB workspace: byte offsets beyond 2^31): interpolation and zero variance at
    training points, permutation invariance of the likelihood."""
    n, d = 20000, 6
    x, y = _data(n, d, seed=77)
    theta = egx.workload.default_theta(d) * 2.0
    with egx.GpHandle(x, y, corr=3) as h:
        h.finalize(theta)
        lk, s2 = h.fitted_scalars()
        idx = np.arange(0, n, 401)
        yp, vp = h.predict_valvar(x[idx])
        np.testing.assert_allclose(yp, y[idx], rtol=1e-6, atol=1e-6 * np.abs(y).max())
        assert np.all(vp >= 0) and np.all(vp <= 1e-6 * s2)
    perm = np.random.default_rng(0).permutation(n)
    with egx.GpHandle(x[perm], y[perm], corr=3) as h2:
        lk2, st2 = h2.likelihood(theta)
        assert st2 == 0 and lk2 == pytest.approx(lk, rel=LK_RTOL)


def test_lbfgs_fit_improves_on_start_and_on_derivative_free(egx):
    """Gradient-based tuned fit (new): from the default start theta0 = 0.1 the projected L-BFGS run ends with a
    much smaller projected gradient and a likelihood at least as good as the derivative-free optimiser's (which
    stops at the reference's clamp(10 h, 25, max_eval) budget).  (The optimum of this smooth response sits on the
    lower theta bound in one direction, where the likelihood is flat and noisy: scipy's L-BFGS-B on the CPU oracle
    stops there with a larger residual gradient than this implementation.)"""
    x, y = _data(400, 3, seed=31)
    y = np.sin(6 * x[:, 0]) + x[:, 1] ** 2 + 0.5 * x[:, 2]
    base = lambda: egx.GaussianProcess.params(egx.ConstantMean(), egx.Matern52Corr()).n_start(0)
    g1 = base().optimizer("lbfgs").max_eval(200).fit(x, y)
    g2 = base().max_eval(200).fit(x, y)
    th, lk = g1.theta(), g1.likelihood()
    assert np.all(th >= 1e-2 * (1 - 1e-9)) and np.all(th <= 1e1 * (1 + 1e-9))
    assert lk >= g2.likelihood() - 1e-6 * abs(g2.likelihood())
    assert g1.n_evals > 0 and g2.n_evals > 0

    def projected_grad(h, theta):
        l, g, st = h.likelihood_grad(theta)
        assert st == 0
        gx = theta * np.log(10.0) * g  # dL/dlog10(theta); maximisation: at the lower bound only gx > 0 counts
        at_lo, at_hi = theta <= 1e-2 * 1.0001, theta >= 1e1 * 0.9999
        gx = np.where((at_lo & (gx < 0)) | (at_hi & (gx > 0)), 0.0, gx)
        return l, np.abs(gx).max()

    with egx.GpHandle(x, y, corr=3) as h:
        l0, pg0 = projected_grad(h, np.full(3, 0.1))
        l1, pg1 = projected_grad(h, th)
        assert l1 == pytest.approx(lk, rel=1e-12) and l1 > l0
        assert pg1 <= 0.5 * pg0
    g1.close()
    g2.close()


# ------------------------------------------------------------------ x-gradients of the predictions
def _grad_tol(ref):
    return dict(rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(ref).max())


@pytest.mark.parametrize("mean", range(3))
@pytest.mark.parametrize("corr", range(4))
def test_prediction_gradients_vs_oracle(egx, O, mean, corr):
    x, y = _data(300, 3, seed=31)
    theta = np.array([0.8, 1.3, 0.6]) * (4.0 if corr == 0 else 1.0)
    ref = O.fit_fixed(x, y, theta, mean=MEANS[mean], corr=KINDS[corr])
    xq = np.random.default_rng(3).random((37, 3))
    with egx.GpHandle(x, y, mean=mean, corr=corr) as h:
        h.finalize(theta)
        gy, gv = h.predict_gradients(xq), h.predict_var_gradients(xq)
        ry, rv = ref.predict_valvar_gradients(xq)
        np.testing.assert_allclose(gy, ry, **_grad_tol(ry))
        np.testing.assert_allclose(gv, rv, **_grad_tol(rv))
        gy2, gv2 = h.predict_valvar_gradients(xq)
        np.testing.assert_array_equal(gy2, gy)
        np.testing.assert_array_equal(gv2, gv)
        assert h.predict_gradients(np.zeros((0, 3))).shape == (0, 3)


def test_reference_bug_var_derivatives_kat_through_cabi(egx, golden_dir):
    """algorithm.rs:1723-1797: the reference's fixed data set, d var / d x vs central differences of predict_var."""
    import json
    b = json.load(open(os.path.join(golden_dir, "kat.json")))["bug_var_derivatives"]
    gp = egx.GaussianProcess.params(egx.ConstantMean(), egx.SquaredExponentialCorr()) \
        .theta_tuning(egx.ThetaTuning.Fixed(np.sqrt(2.0 * np.array(b["theta_sq_half"])))) \
        .fit(np.array(b["xt"]), np.array(b["yt"]))
    xa, xb = b["x"]
    e = b["e"]
    v = gp.predict_var(np.array([[xa + e, xb], [xa - e, xb], [xa, xb + e], [xa, xb - e]]))
    g = gp.predict_var_gradients(np.array([[xa, xb]]))
    assert g[0, 0] == pytest.approx((v[0] - v[1]) / (2 * e), abs=b["epsilon"])
    assert g[0, 1] == pytest.approx((v[2] - v[3]) / (2 * e), abs=b["epsilon"])
    gp.close()


def test_prediction_gradients_kpls_weights(egx, O):
    x, y = _data(250, 5, seed=33)
    rng = np.random.default_rng(2)
    w = rng.standard_normal((5, 2))
    theta = np.array([0.7, 0.4])
    xq = rng.random((20, 5))
    for corr in range(4):
        ref = O.fit_fixed(x, y, theta, corr=KINDS[corr], w_star=w)
        with egx.GpHandle(x, y, corr=corr, w_star=w) as h:
            h.finalize(theta)
            ry, rv = ref.predict_valvar_gradients(xq)
            np.testing.assert_allclose(h.predict_gradients(xq), ry, **_grad_tol(ry))
            np.testing.assert_allclose(h.predict_var_gradients(xq), rv, **_grad_tol(rv))


def test_prediction_gradients_medium_size_and_state(egx, O):
    """n = 2000, d = 8 (one pass of 8 register sums), m spanning several workgroups; one point (EGO's use: the training
    range is split over workgroups); the cached C^-T is rebuilt after a refit at another theta."""
    x, y = _data(2000, 8, seed=35)
    theta = np.full(8, 0.9)
    xq = np.random.default_rng(4).random((300, 8))
    with egx.GpHandle(x, y, corr=3) as h:
        for th in (theta, theta * 1.7):
            ref = O.fit_fixed(x, y, th, corr=KINDS[3])
            h.finalize(th)
            ry, rv = ref.predict_valvar_gradients(xq[:40])
            gy, gv = h.predict_valvar_gradients(xq)
            np.testing.assert_allclose(gy[:40], ry, **_grad_tol(ry))
            np.testing.assert_allclose(gv[:40], rv, **_grad_tol(rv))
            g1 = h.predict_var_gradients(xq[7:8])
            np.testing.assert_allclose(g1, gv[7:8], rtol=1e-9, atol=1e-12 * np.abs(gv).max())
        with pytest.raises(egx.NotFittedError):
            h.likelihood(theta)  # single workspace: the evaluation takes workspace 0 and un-fits the model
            h.predict_gradients(xq[:1])


def test_prediction_gradients_d32_two_chunks_of_registers(egx):
    """d = 40 > 32: two register passes; checked against central differences of the GPU's own predictions."""
    x, y = _data(500, 40, seed=37)
    theta = np.full(40, 0.35)
    xq = np.random.default_rng(6).random((3, 40))
    with egx.GpHandle(x, y, corr=2) as h:
        h.finalize(theta)
        gy, gv = h.predict_valvar_gradients(xq)
        e = 1e-6
        for k in (0, 17, 33, 39):
            dq = np.zeros(40)
            dq[k] = e
            fy = (h.predict(xq + dq) - h.predict(xq - dq)) / (2 * e)
            fv = (h.predict_var(xq + dq) - h.predict_var(xq - dq)) / (2 * e)
            np.testing.assert_allclose(gy[:, k], fy, rtol=1e-5, atol=1e-6 * np.abs(gy).max())
            np.testing.assert_allclose(gv[:, k], fv, rtol=1e-5, atol=1e-6 * np.abs(gv).max())


def test_partial_theta_tuning(egx, O):
    """ThetaTuning::Partial (algorithm.rs:822-826, 873-960): inactive components keep their initial value, the
    active ones move and the likelihood at the result is what the oracle computes there."""
    x, y = _data(300, 3, seed=41)
    init = np.array([0.7, 1.1, 0.9])
    gp = egx.GaussianProcess.params(egx.ConstantMean(), egx.Matern52Corr()) \
        .theta_tuning(egx.ThetaTuning.Partial(init, [(1e-2, 1e1)], [0, 2])).n_start(2).max_eval(40).fit(x, y)
    th = gp.theta()
    assert th[1] == init[1] and (th[0] != init[0] or th[2] != init[2])
    lk0, st0 = O.likelihood_at(x, y, init, corr=KINDS[3])[:2]
    ref = O.fit_fixed(x, y, th, corr=KINDS[3])
    assert gp.likelihood() == pytest.approx(ref.likelihood, rel=LK_RTOL)
    assert gp.likelihood() >= lk0  # start 0 is the user's theta: the optimum cannot be worse
    gp.close()
    with pytest.raises(egx.InvalidValueError):
        egx.GaussianProcess.params(egx.ConstantMean(), egx.Matern52Corr()) \
            .theta_tuning(egx.ThetaTuning.Partial(init, [(1e-2, 1e1)], [0, 5])).fit(x, y)


@pytest.mark.parametrize("recomb", ["smooth", "hard"])
def test_mixture_gradients_of_gpu_experts_match_oracle(egx, O, recomb):
    from egobox_amd.moe import GaussianMixture, GpMixture
    from oracle import moe_oracle as MO
    rng = np.random.default_rng(1)
    gpu_experts, cpu_experts = [], []
    for c in range(3):
        x = rng.random((200, 2)) + [c, 0.0]
        y = np.sin(3 * x[:, 0]) + x[:, 1] * (c + 1)
        theta = [1.5, 1.0]
        gpu_experts.append(egx.GaussianProcess.params(egx.ConstantMean(), egx.Matern52Corr())
                           .theta_tuning(egx.ThetaTuning.Fixed(theta)).fit(x, y))
        cpu_experts.append(O.fit_fixed(x, y, theta, corr=O.MATERN52))
    means = np.array([[0.5, 0.5], [1.5, 0.5], [2.5, 0.5]])
    covs = np.array([np.eye(2) * 0.2] * 3)
    w = np.array([0.3, 0.3, 0.4])
    mix = GpMixture(gpu_experts, GaussianMixture(w, means, covs, 0.8), recomb)
    gmo = MO.GaussianMixtureOracle(w, means, covs, 0.8)
    xq = np.random.default_rng(2).random((60, 2)) * [3.0, 1.0]
    gy, gv = mix.predict_valvar_gradients(xq)
    if recomb == "smooth":
        wy, wv = MO.predict_gradients_smooth(cpu_experts, gmo, xq), MO.predict_var_gradients_smooth(cpu_experts, gmo, xq)
    else:
        wy, wv = MO.predict_gradients_hard(cpu_experts, gmo, xq), MO.predict_var_gradients_hard(cpu_experts, gmo, xq)
    np.testing.assert_allclose(gy, wy, rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(wy).max())
    np.testing.assert_allclose(gv, wv, rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(wv).max())
    for e in gpu_experts:
        e.close()


def test_likelihood_gradient_several_panel_groups(egx, O):
    """n_pad = 1536 = three groups of two 256-wide panels: the two-level blocking of the triangular solves
    (identity right-hand sides, rows skipped below the current block) and the ktri GEMM behind the theta-gradient."""
    x, y = _data(1500, 4, seed=43)
    theta = np.array([0.9, 1.2, 0.7, 1.0])
    lk_ref, g_ref = O.likelihood_grad(x, y, theta, corr=KINDS[3])
    with egx.GpHandle(x, y, corr=3) as h:
        lk, g, st = h.likelihood_grad(theta)
        assert st == 0 and lk == pytest.approx(lk_ref, rel=LK_RTOL)
        np.testing.assert_allclose(g, g_ref, rtol=1e-6, atol=1e-6 * np.abs(g_ref).max())


# ------------------------------------------------------------------ exact closed form at ANY size (no oracle run needed)
def _ou_closed_form(x, y, theta):
    """Constant-mean kriging with the absolute-exponential kernel in ONE dimension is an Ornstein-Uhlenbeck process:
    on the sorted points C^-1 v is the innovations transform  w_1 = v_1, w_{i+1} = (v_{i+1} - rho_i v_i) / sqrt(1 - rho_i^2),
    rho_i = exp(-theta dx_i), and diag C = (1, sqrt(1 - rho_i^2)).  O(n), exact -> likelihood, sigma2, beta as the
    reference defines them (algorithm.rs:1006-1048) at sizes no dense CPU factorisation reaches in a test."""
    x = np.asarray(x, dtype=np.float64).ravel()
    xs, ys = x.std(ddof=1), y.std(ddof=1)
    order = np.argsort(x)
    xn = ((x - x.mean()) / xs)[order].astype(np.longdouble)
    yn = ((y - y.mean()) / ys)[order].astype(np.longdouble)
    rho = np.exp(-np.longdouble(theta) * np.diff(xn))
    s = np.sqrt(1.0 - rho * rho)

    def whiten(v):
        return np.concatenate([v[:1], (v[1:] - rho * v[:-1]) / s])

    ft, yt = whiten(np.ones_like(xn)), whiten(yn)
    beta = ft.dot(yt) / ft.dot(ft)
    r = yt - ft * beta
    n = x.size
    sigma2 = r.dot(r) / n
    lk = -n * (np.log10(sigma2) + 2.0 / n * np.log10(s).sum())
    return float(lk), float(sigma2 * ys * ys), float(beta), float(s.min())


@pytest.mark.parametrize("n", [5000, 16384, 49152, 98304])
def test_ornstein_uhlenbeck_closed_form_any_size(egx, n):
    """n = 98304 is a 77 GB correlation matrix (the handle is sized for the 288 GB of an MI355X; 131072 = 137 GB runs
    too, tools/ou_capacity.py): likelihood, variance and beta against the exact O(n) closed form, rows shuffled."""
    rng = np.random.default_rng(n)
    x = np.sort(rng.random(n)) + np.arange(n) * (0.5 / n)  # strictly increasing, gaps >= 0.5 / n
    y = np.sin(7.0 * x) + 0.3 * np.cos(23.0 * x) + 0.05 * rng.standard_normal(n)
    theta = 40.0
    lk, s2, beta, min_pivot = _ou_closed_form(x, y, theta)
    assert min_pivot > 1e-3
    perm = rng.permutation(n)
    with egx.GpHandle(x[perm].reshape(-1, 1), y[perm], corr=1, nugget=0.0) as h:
        h.finalize([theta])
        lk_gpu, s2_gpu = h.fitted_scalars()
        assert lk_gpu == pytest.approx(lk, rel=LK_RTOL)
        assert s2_gpu == pytest.approx(s2, rel=1e-7)
        assert float(h.inner()["beta"][0, 0]) == pytest.approx(beta, rel=1e-6, abs=1e-9)
        idx = np.arange(0, n, 997)
        yp, vp = h.predict_valvar(x[idx].reshape(-1, 1))
        np.testing.assert_allclose(yp, y[idx], rtol=1e-6, atol=1e-6 * np.abs(y).max())
        assert np.all(vp >= 0) and np.all(vp <= 1e-6 * s2)


def test_plain_c_host_drives_the_boundary(tmp_path):
    """tests/c_host/golden_a_driver.c: a C99 program (no Python, no torch) links libegx_gp_hip.so and reproduces the
    reference's notebook / Python-test values through the C ABI -- the drop-in boundary as a compiled host sees it."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "golden_a_driver"
    libdir = os.path.join(root, "egobox_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{os.path.join(root, 'include')}",
                    os.path.join(root, "tests", "c_host", "golden_a_driver.c"), f"-L{libdir}", "-legx_gp_hip", "-lm",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert out.stdout.startswith("OK")


def test_reference_style_cpp_host_tests(tmp_path):
    """tests/c_host/reference_style_tests.cpp: the reference's own test shapes (test_gp! over 3 means x 4 kernels,
    golden A, test_bug_var_derivatives, the theta0-length check) in C++ through include/egx_gp.hpp."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "reference_style_tests"
    libdir = os.path.join(root, "egobox_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", f"-I{os.path.join(root, 'include')}",
                    os.path.join(root, "tests", "c_host", "reference_style_tests.cpp"), f"-L{libdir}", "-legx_gp_hip",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert out.stdout.startswith("OK")


def test_a_first_handle_of_a_new_shape_reuses_idle_streams(egx):
    """The HIP runtime takes ~3 ms to create a stream and a workspace has four (profiles/r06_first_handle_of_a_shape_costs.txt:
    14.5 ms per workspace, whatever n): idle workspaces -- pooled or freed -- hand their streams to a free list, and a handle of
    ANY shape takes them from there.  Models of different sizes created and dropped one after the other (a mixture's clusters,
    crates/moe/src/algorithm.rs:167-262) pay the runtime once; egx_trim destroys the idle streams and the library keeps working."""
    import time
    d = 3
    theta = np.full(d, 1.0)

    def cycle(n, nws):
        x, y = _data(n, d, seed=n)
        t0 = time.perf_counter()
        h = egx.GpHandle(x, y, corr=1, n_workspaces=nws)
        t = time.perf_counter() - t0
        lk = h.likelihood_batch(np.stack([theta * (1 + 0.01 * c) for c in range(nws)]))[0]
        h.finalize(theta)
        out = (h.fitted_scalars()[0], lk.copy())
        h.close()
        return t, out

    egx.trim()
    t_first, ref = cycle(900, 4)                       # creates four stream sets (or finds none idle)
    times = [cycle(n, 4)[0] for n in (1300, 1700, 2300, 2900)]   # four NEW shapes: pool misses, idle streams
    print(f"first handle {t_first * 1e3:.1f} ms, first handles of four other shapes {[round(t * 1e3, 2) for t in times]} ms")
    assert max(times) < 0.025                          # (4 x 14.5 ms when every shape created its own streams)
    egx.trim()                                         # pooled entries freed, idle streams destroyed
    t_again, again = cycle(900, 4)
    assert again[0] == ref[0]
    np.testing.assert_array_equal(again[1], ref[1])
    egx.trim()


def test_no_device_memory_leak_over_handle_lifecycles(egx, O):
    """Create / fit / predict / gradients / destroy many handles (dense and sparse): free device memory returns to its
    level (every hipMalloc of a handle is owned by it)."""
    import torch
    x, y = _data(1500, 4, seed=51)
    z = x[:40].copy()
    theta = np.full(4, 0.9)

    def cycle():
        with egx.GpHandle(x, y, corr=3, n_workspaces=2) as h:
            h.finalize(theta)
            h.predict_valvar(x[:10])
            h.predict_valvar_gradients(x[:3])
            h.likelihood_grad(theta)
        with egx.SgpHandle(x, y, z) as s:
            s.finalize(theta, 1.0, 0.01)
            s.predict_var(x[:10])

    cycle()
    egx.trim()  # (destroyed handles park their device resources in the library's pool: compare with it emptied)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(25):
        cycle()
    egx.trim()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    # no GROWTH (the runtime's own sub-allocator may hand back a few 2 MiB blocks more or less than before: observed
    # +68 MiB free after the 25 cycles)
    assert free0 - free1 < 64 << 20, (free0, free1)
    # ... and over 20 DISTINCT shapes (every one leaves pooled slabs and, for the sizes that take chain launches, a device task
    # list per shape behind): egx_trim gives all of it back -- the pool reads zero and the free memory is where it was
    for j in range(20):
        n = 300 + 137 * j
        xj, yj = _data(n, 3, seed=80 + j)
        with egx.GpHandle(xj, yj, corr=1, n_workspaces=1 + j % 3) as h:   # (absolute exponential: well conditioned at any density)
            h.finalize(np.full(3, 1.0))
            h.predict_valvar(xj[:5])
    assert egx.pool_stats()["cached_bytes"] > 0
    freed = egx.trim()
    assert freed > 0 and egx.pool_stats()["cached_bytes"] == 0
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    assert free0 - free2 < 64 << 20, (free0, free2)


def test_randomised_parity_sweep(egx, O):
    """40 seeded random configurations (n around the 64 / 128 / 256 tile edges, d 1..12, every mean x kernel, with and
    without KPLS weights): likelihood, beta, predictions, variances and x-gradients against the oracle wherever the
    oracle's smallest Cholesky pivot says the problem is well posed."""
    rng = np.random.default_rng(2024)
    edges = [63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 300, 383, 385, 511, 513]
    checked = 0
    for case in range(40):
        n = int(rng.choice(edges)) + int(rng.integers(0, 3))
        d = int(rng.integers(1, 13))
        mean = int(rng.integers(0, 3)) if d <= 6 else int(rng.integers(0, 2))
        corr = int(rng.integers(0, 4))
        if O.regression_value(MEANS[mean], np.zeros((1, d))).shape[1] >= n // 2:
            mean = 0
        x = rng.random((n, d)) * rng.uniform(0.5, 20.0, size=d) + rng.uniform(-5, 5, size=d)
        y = np.sin(x @ rng.standard_normal(d) / np.sqrt(d)) + 0.1 * (x[:, 0] - x[:, 0].mean()) ** 2
        use_w = d >= 3 and case % 4 == 0
        h_dim = int(rng.integers(1, d)) if use_w else d
        w = rng.standard_normal((d, h_dim)) if use_w else None
        theta = rng.uniform(0.5, 2.0, size=h_dim) * (3.0 if corr == 0 else 1.0)
        ref = O.fit_fixed(x, y, theta, mean=MEANS[mean], corr=KINDS[corr], w_star=w)
        if np.min(np.diag(ref.inner.r_chol)) < 1e-3:
            continue
        checked += 1
        xq = rng.random((17, d)) * (x.max(axis=0) - x.min(axis=0)) + x.min(axis=0)
        with egx.GpHandle(x, y, mean=mean, corr=corr, w_star=w) as h:
            lk, st = h.likelihood(theta)
            assert st == 0 and lk == pytest.approx(ref.likelihood, rel=LK_RTOL), (case, n, d, mean, corr)
            h.finalize(theta)
            np.testing.assert_allclose(h.inner()["beta"], ref.inner.beta, rtol=1e-6, atol=1e-8 * np.abs(ref.inner.beta).max())
            yv, vv = h.predict_valvar(xq)
            ry, rv = ref.predict(xq), ref.predict_var(xq)
            np.testing.assert_allclose(yv, ry, rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(ry).max())
            np.testing.assert_allclose(vv, rv, rtol=PRED_RTOL, atol=PRED_RTOL * ref.inner.sigma2)
            gy, gv = h.predict_valvar_gradients(xq[:5])
            wy, wv = ref.predict_valvar_gradients(xq[:5])
            np.testing.assert_allclose(gy, wy, rtol=1e-5, atol=1e-5 * np.abs(wy).max())
            np.testing.assert_allclose(gv, wv, rtol=1e-5, atol=1e-5 * max(np.abs(wv).max(), 1e-12))
    assert checked >= 25


def test_few_query_paths_agree_with_batched_paths_and_oracle(egx, O):
    """m <= 8 takes the few-query paths (split-range mean, GEMV passes over the cached C^-T -- from the third small
    predict_var call after a fit, or after any gradient call --, lanes-over-training-points gradient kernel); m > 8 the
    batched ones.  Both must match the oracle, for every kernel and a non-trivial trend."""
    x, y = _data(700, 5, seed=61)
    rng = np.random.default_rng(8)
    xq = rng.random((20, 5))
    for corr in range(4):
        theta = np.full(5, 0.9) * (3.0 if corr == 0 else 1.0)
        ref = O.fit_fixed(x, y, theta, mean="Linear", corr=KINDS[corr])
        with egx.GpHandle(x, y, mean=1, corr=corr) as h:
            h.finalize(theta)
            ry, rv = ref.predict(xq), ref.predict_var(xq)
            wy, wv = ref.predict_valvar_gradients(xq)
            for rep in range(4):  # the third small predict_var call switches to the C^-T path
                small = h.predict_var(xq[:3])
                np.testing.assert_allclose(small, rv[:3], rtol=PRED_RTOL, atol=PRED_RTOL * ref.inner.sigma2)
            np.testing.assert_allclose(h.predict(xq[:2]), ry[:2], rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(ry).max())
            np.testing.assert_allclose(h.predict(xq), ry, rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(ry).max())
            np.testing.assert_allclose(h.predict_var(xq), rv, rtol=PRED_RTOL, atol=PRED_RTOL * ref.inner.sigma2)
            ys, vs = h.predict_valvar(xq[:4])  # value + variance of a few points: both few-query paths in one call
            np.testing.assert_allclose(ys, ry[:4], rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(ry).max())
            np.testing.assert_allclose(vs, rv[:4], rtol=PRED_RTOL, atol=PRED_RTOL * ref.inner.sigma2)
            gy_s, gv_s = h.predict_valvar_gradients(xq[:8])
            gy_b, gv_b = h.predict_valvar_gradients(xq)
            np.testing.assert_allclose(gy_s, wy[:8], rtol=1e-6, atol=1e-6 * np.abs(wy).max())
            np.testing.assert_allclose(gv_s, wv[:8], rtol=1e-6, atol=1e-6 * np.abs(wv).max())
            np.testing.assert_allclose(gy_b, wy, rtol=1e-6, atol=1e-6 * np.abs(wy).max())
            np.testing.assert_allclose(gv_b, wv, rtol=1e-6, atol=1e-6 * np.abs(wv).max())
            h.finalize(theta * 1.3)  # a refit invalidates the cached C^-T
            ref2 = O.fit_fixed(x, y, theta * 1.3, mean="Linear", corr=KINDS[corr])
            np.testing.assert_allclose(h.predict_var_gradients(xq[:2]), ref2.predict_var_gradients(xq[:2]), rtol=1e-6,
                                       atol=1e-6 * np.abs(wv).max())


def test_reference_constant_function(egx):
    """algorithm.rs:1216-1237: a constant response (zero sample std -> divisor 1, sigma2 = 0): predictions are the
    constant to 1e-6, with the full-dimension kernel and with a supplied (d x 1) KPLS rotation like the reference's
    kpls_dim(Some(1)), tuned and at a fixed theta."""
    rng = np.random.default_rng(42)
    xt = rng.random((5, 3))
    yt = np.full(5, 3.1)
    xtest = np.random.default_rng(43).random((5, 3))
    for w in (None, np.array([[0.6], [0.5], [0.4]])):
        for tuning in (egx.ThetaTuning.Fixed([0.1]), egx.ThetaTuning.Full([0.1], [(1e-2, 1e1)])):
            p = egx.GaussianProcess.params(egx.ConstantMean(), egx.SquaredExponentialCorr()).theta_tuning(tuning).n_start(2)
            if w is not None:
                p.kpls_dim(1).kpls_weights(w)
            gp = p.fit(xt, yt)
            yp, vp = gp.predict_valvar(xtest)
            np.testing.assert_allclose(yp, 3.1, atol=1e-6)
            assert np.all(np.isfinite(vp)) and np.all(vp >= 0.0) and np.all(vp <= 1e-12)
            gp.close()


def test_reference_fixed_theta(egx):
    """algorithm.rs:1641-1657: the default (tuned) fit moves theta off its initial guess; refitting with
    ThetaTuning::Fixed(theta*) returns exactly theta*."""
    xt = np.array([[0.0], [1.0], [2.0], [3.0], [4.0]])
    yt = np.array([0.0, 1.0, 1.5, 0.9, 1.0])
    gp = egx.Kriging.params().fit(xt, yt)
    expected = gp.theta().copy()
    assert abs(expected[0] - egx.ThetaTuning.DEFAULT_INIT) > 1e-6
    gp2 = egx.Kriging.params().theta_tuning(egx.ThetaTuning.Fixed(expected)).fit(xt, yt)
    np.testing.assert_array_equal(gp2.theta(), expected)
    assert gp2.likelihood() == pytest.approx(gp.likelihood(), rel=1e-12)
    gp.close()
    gp2.close()


def test_concurrent_mixed_calls_on_one_handle(egx):
    """Threads hammering ONE fitted handle with predictions, gradients and (on the spare workspace) likelihood
    evaluations: every result equals the serial one (exclusive lock for the fitted state, shared lock + workspace pool for
    the likelihood; Python releases the GIL around every C call)."""
    from concurrent.futures import ThreadPoolExecutor
    x, y = _data(800, 4, seed=71)
    theta = np.full(4, 0.9)
    rng = np.random.default_rng(9)
    xqs = [rng.random((m, 4)) for m in (1, 3, 50, 200)]
    thetas = [theta * f for f in (0.8, 1.0, 1.2, 1.5)]
    with egx.GpHandle(x, y, corr=3, n_workspaces=2) as h:
        h.finalize(theta)
        serial = {"p": [h.predict(q) for q in xqs], "v": [h.predict_var(q) for q in xqs],
                  "g": [h.predict_valvar_gradients(q[:5]) for q in xqs], "l": [h.likelihood(t) for t in thetas]}

        def job(i):
            kind, k = "pvgl"[i % 4], (i // 4) % 4
            if kind == "p":
                return kind, k, h.predict(xqs[k])
            if kind == "v":
                return kind, k, h.predict_var(xqs[k])
            if kind == "g":
                return kind, k, h.predict_valvar_gradients(xqs[k][:5])
            return kind, k, h.likelihood(thetas[k])

        with ThreadPoolExecutor(6) as pool:
            results = list(pool.map(job, range(64)))
        for kind, k, val in results:
            want = serial[kind][k]
            if kind == "g":
                np.testing.assert_array_equal(val[0], want[0])
                np.testing.assert_array_equal(val[1], want[1])
            elif kind == "l":
                assert val == want
            elif kind == "v":  # few-point variances switch to the cached C^-T path on the third call: same value to rounding
                np.testing.assert_allclose(val, want, rtol=1e-9, atol=1e-12)
            else:
                np.testing.assert_array_equal(val, want)
        np.testing.assert_array_equal(h.predict(xqs[2]), serial["p"][2])  # the fit survived (workspace 1 served the likelihoods)


def test_batch_likelihood_keeps_a_fitted_model(egx):
    """egx_gp_likelihood_batch on a fitted handle with spare workspaces pipelines over those and leaves the fit alone; with a
    single workspace it has to take it."""
    x, y = _data(500, 3, seed=73)
    thetas = egx.theta_sweep_candidates(5, 3, seed=1)
    with egx.GpHandle(x, y, corr=3, n_workspaces=3) as h:
        h.finalize([0.9, 0.9, 0.9])
        y0 = h.predict(x[:4])
        lk, st = h.likelihood_batch(thetas)
        np.testing.assert_array_equal(h.predict(x[:4]), y0)
        lk1 = [h.likelihood(t)[0] for t in thetas]
        np.testing.assert_array_equal(lk, lk1)
    with egx.GpHandle(x, y, corr=3, n_workspaces=1) as h1:
        h1.finalize([0.9, 0.9, 0.9])
        h1.likelihood_batch(thetas)
        with pytest.raises(egx.NotFittedError):
            h1.predict(x[:4])


def test_device_allocation_failure_is_an_error_not_a_crash(egx):
    """n = 200 000 would need a 320 GB correlation matrix (> the 288 GB of the GPU): create fails with the HIP error
    code and message, nothing leaks, and the library keeps working."""
    import torch
    n = 200_000
    x = np.linspace(0.0, 1.0, n).reshape(-1, 1)
    y = np.sin(x[:, 0])
    with egx.GpHandle(x[:300], y[:300], corr=3) as h:  # first use loads code objects / runtime pools: not part of the check
        h.likelihood([5.0])
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    assert egx.pool_stats()["cached_bytes"] > 0  # the closed handle's resources sit in the pool
    with pytest.raises(egx.EgxError) as ei:
        egx.GpHandle(x, y)
    assert ei.value.rc == egx._lib.ERR_HIP and "out of memory" in str(ei.value)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20  # nothing leaked ...
    # ... and before it reported out-of-memory the library gave back what it held itself (round 3's pool kept up to 48 GB
    # through such a failure: ADVICE r3)
    assert egx.pool_stats()["cached_bytes"] == 0
    with egx.GpHandle(x[:300], y[:300], corr=3) as h:  # still alive
        assert h.likelihood([5.0])[1] == 0


def test_not_positive_definite_in_a_large_matrix(egx):
    """A bad pivot deep inside a multi-group factorisation (look-ahead streams, chip-filling updates): the status channel
    reports NOT_POSITIVE_DEFINITE, nothing hangs, and the handle evaluates a good theta right afterwards."""
    x, y = _data(6000, 3, seed=77)
    x = x.copy()
    x[4321] = x[1234]  # duplicated point + negative nugget -> a negative pivot at column ~4321
    with egx.GpHandle(x, y, corr=3, nugget=-1e-6) as h:
        lk, st = h.likelihood([1.0, 1.0, 1.0])
        assert st == egx._lib.STATUS_NOT_POSITIVE_DEFINITE and lk == -math.inf
        with pytest.raises(egx.LinalgError):
            h.finalize([1.0, 1.0, 1.0])
    with egx.GpHandle(x, y, corr=3) as h2:  # default nugget: the duplicate is regularised
        lk2, st2 = h2.likelihood([1.0, 1.0, 1.0])
        assert st2 == 0 and np.isfinite(lk2)


def test_reference_kpls_griewank(egx):
    """algorithm.rs:1325-1375: KPLS (3 components of a PLS regression, now computed on the host by egobox_amd/kpls.py) on
    Griewank in 5 dimensions, 100 training points, default tuned fit: normalised RMS error of 100 test predictions
    below 1e-2, the reference's own bound."""
    from egobox_amd import workload
    xt = workload.lhs(100, 5, 42)
    xtest = workload.lhs(100, 5, 0)
    gp = egx.GaussianProcess.params(egx.ConstantMean(), egx.SquaredExponentialCorr()).kpls_dim(3).fit(
        -600.0 + 1200.0 * xt, workload.griewank(xt))
    assert gp.theta().shape == (3,)
    ytest, ytrue = gp.predict(-600.0 + 1200.0 * xtest), workload.griewank(xtest)
    assert np.linalg.norm(ytrue - ytest) / np.linalg.norm(ytrue) < 1e-2
    gp.close()


# ------------------------------------------------------------------ round 4: input dimensions beyond 64
# (the reference benches its kernels at dim 100, crates/gp/benches/corr.rs:11-15, and KPLS exists for d >> 64,
#  crates/gp/src/algorithm.rs:843-855: the correlation kernels stage the dimensions in chunks of 64)
@pytest.mark.parametrize("kind", range(4))
@pytest.mark.parametrize("n,d", [(150, 100), (130, 200), (70, 65)])
def test_corr_kernels_beyond_64_dimensions(egx, O, kind, n, d):
    rng = np.random.default_rng(n + d + kind)
    xn, _, _ = O.normalize(rng.random((n, d)))
    theta = (0.2 + rng.random(d)) / np.sqrt(d)
    got = egx.corr_matrix(kind, xn, theta)
    dd, idx = O.diff_matrix(xn)
    want = O.assemble_r(O.corr_value(KINDS[kind], dd, theta, np.eye(d)), idx, n, O.DEFAULT_NUGGET)
    np.testing.assert_allclose(got, want, rtol=5e-13, atol=1e-300)
    xq, _, _ = O.normalize(rng.random((97, d)))
    gc = egx.cross_corr(kind, xq, xn, theta)
    wc = O.corr_value(KINDS[kind], O.pairwise_differences(xq, xn), theta, np.eye(d)).reshape(97, n)
    np.testing.assert_allclose(gc, wc, rtol=5e-13, atol=1e-300)


@pytest.mark.parametrize("corr", range(4))
@pytest.mark.parametrize("d", [100, 200])
def test_fit_predict_and_gradients_beyond_64_dimensions(egx, O, corr, d):
    """Fit at fixed theta, predict / predict_var (the LDS-only prediction kernel with chunked dimensions + the device-side
    query normalisation over two to four chunks), the theta-gradient (its trace kernel's chunked passes and four to seven
    output chunks) and the x-gradients (d = 100 / 200: the kernel form that reads the query from global memory) against
    the oracle."""
    n = 300
    x, y = _data(n, d, seed=50 + d)
    theta = egx.workload.default_theta(d) * (1.0 + 0.3 * np.arange(d) / (d - 1)) * (2.0 if corr == 0 else 1.0)
    ref = _check_fit(egx, O, x, y, theta, 0, corr)
    lk_ref, g_ref = O.likelihood_grad(x, y, theta, corr=KINDS[corr])
    xq = x.min(0) + (x.max(0) - x.min(0)) * np.random.default_rng(3).random((40, d))
    with egx.GpHandle(x, y, corr=corr, n_workspaces=2) as h:
        lk, g, st = h.likelihood_grad(theta)
        assert st == 0 and lk == pytest.approx(lk_ref, rel=LK_RTOL)
        np.testing.assert_allclose(g, g_ref, rtol=1e-6, atol=1e-6 * np.abs(g_ref).max())
        lkb, gb, stb = h.likelihood_grad_batch(np.stack([theta, theta * 1.01]))
        assert lkb[0] == lk and np.array_equal(gb[0], g)
        h.finalize(theta)
        # x-gradients: directional central differences of the oracle-checked predictions (the oracle's own jacobians cost
        # minutes at this d), batched form and the few-query path
        gy, gv = h.predict_valvar_gradients(xq)
        one = h.predict_valvar_gradients(xq[:1])
        rng = np.random.default_rng(9)
        span = x.max(0) - x.min(0)
        for a in range(6):
            v = rng.standard_normal(d) * span
            v /= np.linalg.norm(v)
            # (the absolute exponential has a kink at every training coordinate: a step of 1e-5 crosses one in a few per
            #  cent of the directions -- measured with the oracle's predictions -- so it gets a step a hundred times shorter)
            eps = 1e-7 if corr == 1 else 1e-5
            yp, vp = h.predict_valvar(np.stack([xq[a] + eps * v, xq[a] - eps * v]))
            fy, fv = (yp[0] - yp[1]) / (2 * eps), (vp[0] - vp[1]) / (2 * eps)
            loose = 20.0 if corr == 1 else 1.0
            assert gy[a] @ v == pytest.approx(fy, rel=2e-5 * loose, abs=1e-6 * loose * np.abs(gy).max())
            assert gv[a] @ v == pytest.approx(fv, rel=2e-4 * loose, abs=1e-5 * loose * np.abs(gv).max())
        np.testing.assert_allclose(one[0], gy[:1], rtol=1e-7, atol=1e-9 * np.abs(gy).max())
        np.testing.assert_allclose(one[1], gv[:1], rtol=1e-6, atol=1e-8 * np.abs(gv).max())


def test_kpls_with_100_input_dimensions(egx, O):
    """KPLS is what the reference offers for d >> 64 (algorithm.rs:843-855): d = 100 reduced to h = 3, all four kernels,
    likelihood + predictions against the oracle, and the theta-gradient (h outputs) against finite differences."""
    n, d, hk = 400, 100, 3
    x, y = _data(n, d, seed=77)
    rng = np.random.default_rng(4)
    w = rng.standard_normal((d, hk))
    w /= np.linalg.norm(w, axis=0)
    theta = np.array([0.6, 0.9, 0.4])
    xq = rng.random((50, d))
    for corr in range(4):
        ref = O.fit_fixed(x, y, theta, corr=KINDS[corr], w_star=w)
        with egx.GpHandle(x, y, corr=corr, w_star=w, n_workspaces=2) as h:
            assert h.h == hk
            lk, st = h.likelihood(theta)
            assert st == 0 and lk == pytest.approx(ref.likelihood, rel=LK_RTOL)
            lkg, g, stg = h.likelihood_grad(theta)
            assert stg == 0 and lkg == lk
            for l in range(hk):
                e = np.zeros(hk)
                e[l] = 1e-5 * theta[l]
                lks, sts = h.likelihood_batch(np.stack([theta + e, theta - e]))
                fd = (lks[0] - lks[1]) / (2 * e[l])
                assert g[l] == pytest.approx(fd, rel=5e-5, abs=1e-6 * np.abs(g).max())
            h.finalize(theta)
            yr, vr = ref.predict(xq), ref.predict_var(xq)
            yp, vp = h.predict_valvar(xq)
            np.testing.assert_allclose(yp, yr, rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(yr).max())
            np.testing.assert_allclose(vp, vr, rtol=PRED_RTOL, atol=1e-9 * ref.inner.sigma2 + PRED_RTOL * np.abs(vr).max())


@pytest.mark.parametrize("corr,d", [(0, 300), (3, 300), (0, 700), (2, 1100)])
def test_x_gradients_beyond_256_dimensions(egx, corr, d):
    """Round 5: the batched x-gradient kernel narrows its LDS slab of training points beyond d ~ 300 (64 points wide at
    d = 300, 16 at d = 700, 8 at d = 1100) instead of refusing d > 256.  Checked by directional central differences of
    the predictions (themselves oracle-checked at d = 100 / 200 above), batched form and the few-query form against each
    other -- and, round 6, at d = 300 against the ORACLE's jacobians (algorithm.rs:510-617 restated) on query rows 0 and 1
    (tests/golden/large_n.json `xgrad_n200_d300_*`, make_large_n.py --only xgrad300)."""
    n = 200
    x, y = _data(n, d, seed=5)
    theta = egx.workload.default_theta(d) * (2.0 if corr == 0 else 1.0)
    xq = x.min(0) + (x.max(0) - x.min(0)) * np.random.default_rng(1).random((130, d))
    with egx.GpHandle(x, y, corr=corr) as h:
        h.finalize(theta)
        gy, gv = h.predict_valvar_gradients(xq)
        assert gy.shape == (130, d) and np.all(np.isfinite(gy)) and np.all(np.isfinite(gv))
        if d == 300:
            import json
            import os
            with open(os.path.join(os.path.dirname(__file__), "golden", "large_n.json")) as f:
                fx = json.load(f)["xgrad_n200_d300_" + ("SquaredExponential" if corr == 0 else "Matern52")]
            assert fx["corr_id"] == corr and np.array_equal(np.array(fx["theta"]), theta)
            lk, st = h.likelihood(theta)
            assert st == 0 and abs(lk - fx["likelihood"]) <= 1e-8 * abs(fx["likelihood"])
            h.finalize(theta)
            ry, rv = np.array(fx["predict_gradients"]), np.array(fx["predict_var_gradients"])
            np.testing.assert_allclose(gy[:2], ry, rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(ry).max())
            np.testing.assert_allclose(gv[:2], rv, rtol=PRED_RTOL, atol=PRED_RTOL * np.abs(rv).max())
        one = h.predict_valvar_gradients(xq[:1])
        np.testing.assert_allclose(one[0], gy[:1], rtol=1e-7, atol=1e-9 * np.abs(gy).max())
        np.testing.assert_allclose(one[1], gv[:1], rtol=1e-6, atol=1e-8 * np.abs(gv).max())
        rng = np.random.default_rng(9)
        span = x.max(0) - x.min(0)
        for a in (0, 1, 64, 129):
            v = rng.standard_normal(d) * span
            v /= np.linalg.norm(v)
            eps = 1e-5
            yp, vp = h.predict_valvar(np.stack([xq[a] + eps * v, xq[a] - eps * v]))
            fy, fv = (yp[0] - yp[1]) / (2 * eps), (vp[0] - vp[1]) / (2 * eps)
            assert gy[a] @ v == pytest.approx(fy, rel=2e-5, abs=1e-6 * np.abs(gy).max())
            assert gv[a] @ v == pytest.approx(fv, rel=2e-4, abs=1e-5 * np.abs(gv).max())


def test_x_gradients_refuse_what_does_not_fit_lds(egx):
    """d * (hcols + 1) > 20480 doubles (coefficients + ONE training point) cannot be staged in the 160 KB of LDS: refused
    with a message, never wrong.  Just below that the slab is one training point wide and few queries take the batched
    form (the few-query kernel needs d * (hcols + 5) doubles)."""
    n = 64
    for d, fits in ((10000, True), (10300, False)):
        x, y = _data(n, d, seed=6)
        with egx.GpHandle(x, y, corr=0) as h:
            h.finalize(egx.workload.default_theta(d))
            xq = x[:3] + 1e-3
            assert np.all(np.isfinite(h.predict(xq)))
            if fits:
                g = h.predict_gradients(xq)
                v = np.zeros(d)
                v[::7] = 1.0 / np.sqrt(len(v[::7]))
                eps = 1e-5
                yp = h.predict(np.stack([xq[1] + eps * v, xq[1] - eps * v]))
                assert g[1] @ v == pytest.approx((yp[0] - yp[1]) / (2 * eps), rel=1e-4, abs=1e-6 * np.abs(g).max())
            else:
                with pytest.raises(egx.EgxError, match="LDS"):
                    h.predict_gradients(xq)


def test_shrink_gives_the_multistart_workspaces_back(egx):
    """egx_gp_shrink: a handle that ran an optimisation on several workspaces keeps its fitted factor
    (workspace 0) and frees the rest -- the device memory comes back, predictions are bit for bit unchanged, and the
    handle keeps working (likelihood batches on what is left, a gradient that re-allocates its scratch)."""
    import torch
    n, d = 3000, 4
    x, y = _data(n, d, seed=12)
    theta = egx.workload.default_theta(d) * 2.0
    xq = np.random.default_rng(0).random((64, d))
    with egx.GpHandle(x, y, corr=3, n_workspaces=6) as h:
        h.finalize(theta)
        lk0, g0, _ = h.likelihood_grad(theta * 1.1)  # allocates the C^-T scratch
        before = h.predict_valvar(xq)
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        h.shrink(2)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        per_ws = 8 * 3072 * (3072 + 128)
        assert free1 - free0 >= 4 * per_ws  # four matrices + the gradient scratch came back
        after = h.predict_valvar(xq)
        assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
        lks, sts = h.likelihood_batch(np.stack([theta, theta * 1.2, theta * 0.9]))
        assert np.all(sts == 0) and np.array_equal(h.predict(xq), before[0])  # the fit survived (workspace 1 did the work)
        lk1, g1, _ = h.likelihood_grad(theta * 1.1)
        assert lk1 == lk0 and np.array_equal(g1, g0)
        h.shrink(1)
        h.shrink(5)  # more than there are: nothing to do
        assert np.array_equal(h.predict(xq), before[0])
    with pytest.raises(egx.InvalidValueError):
        with egx.GpHandle(x[:100], y[:100]) as h2:
            h2.shrink(0)
