"""Mixture recombination: the Gaussian-mixture restatements (oracle AND product host code)
against the reference's own known answers, and the recombination logic with CPU-oracle experts."""
import numpy as np
import pytest

from egobox_amd.moe import GaussianMixture, GpMixture
from oracle import gp_oracle as O
from oracle import moe_oracle as MO

# crates/moe/src/gaussian_mixture.rs:378-398 test_pdfs
PDF_KATS = [
    ([[0.0, 0.0]], [[[1.0, 0.0], [0.0, 1.0]]], [1.0, 1.0], 0.05854983152431917),
    ([[0.0, 0.0]], [[[1.0, 0.0], [0.0, 1.0]]], [1.0, 2.0], 0.013064233284684921),
    ([[0.5, -0.2]], [[[2.0, 0.3], [0.3, 0.5]]], [-1.0, 2.0], 0.00014842259203296995),
]


@pytest.mark.parametrize("cls", [GaussianMixture, MO.GaussianMixtureOracle])
@pytest.mark.parametrize("means,covs,x,expected", PDF_KATS)
def test_pdfs_known_answers(cls, means, covs, x, expected):
    g = cls([1.0], means, covs)
    assert g.pdfs(np.array(x))[0] == pytest.approx(expected, rel=1e-13)


@pytest.mark.parametrize("cls", [GaussianMixture, MO.GaussianMixtureOracle])
def test_one_cluster(cls):
    # gaussian_mixture.rs:348-364 test_gmx_one_cluster
    g = cls([1.0], [[4.0, 4.0]], [[[3.0, 0.0], [0.0, 3.0]]], 1.0)
    obs = np.repeat(np.linspace(0, 4, 11)[:, None], 2, axis=1)
    assert np.all(g.predict(obs) == 0)
    assert np.all(g.predict_probas(obs) == 1.0)


def test_product_gmx_matches_oracle():
    rng = np.random.default_rng(0)
    k, nx = 4, 3
    means = rng.standard_normal((k, nx)) * 2
    a = rng.standard_normal((k, nx, nx))
    covs = np.einsum("kij,klj->kil", a, a) + 0.5 * np.eye(nx)
    w = rng.random(k)
    w /= w.sum()
    x = rng.standard_normal((200, nx)) * 3
    for hf in (1.0, 0.99, 0.3):
        g, go = GaussianMixture(w, means, covs, hf), MO.GaussianMixtureOracle(w, means, covs, hf)
        np.testing.assert_allclose(g.predict_probas(x), go.predict_probas(x), rtol=1e-12, atol=1e-300)
        np.testing.assert_array_equal(g.predict(x), go.predict(x))
        # responsibilities sum to one wherever the mixture density is above f64 epsilon (below it the reference
        # leaves log_prob_norm at 0, gaussian_mixture.rs:244-250, and the "probabilities" are the tiny densities)
        tot = g.predict_probas(x).sum(axis=1)
        assert np.all((np.abs(tot - 1.0) < 1e-12) | (tot < 1e-15))
    # two symmetric clusters: the mid point is a coin flip (gaussian_mixture.rs:331-346 setup)
    g = GaussianMixture([0.5, 0.5], [[0.0, 0.0], [4.0, 4.0]], [[[3.0, 0], [0, 3.0]]] * 2, 0.99)
    np.testing.assert_allclose(g.predict_probas(np.array([[2.0, 2.0]])), [[0.5, 0.5]], rtol=1e-12)


def _experts():
    rng = np.random.default_rng(1)
    experts = []
    for c in range(3):
        x = rng.random((40, 2)) + [c, 0.0]
        y = np.sin(3 * x[:, 0]) + x[:, 1] * (c + 1)
        experts.append(O.fit_fixed(x, y, [1.5, 1.0], corr=O.MATERN52))
    means = np.array([[0.5, 0.5], [1.5, 0.5], [2.5, 0.5]])
    covs = np.array([np.eye(2) * 0.2] * 3)
    return experts, means, covs


@pytest.mark.parametrize("recomb", ["smooth", "hard"])
def test_recombination_with_oracle_experts(recomb):
    """The product's GpMixture logic (routing, batching, weighting) with CPU-oracle experts standing in for the
    GPU ones equals the reference-shaped per-row / per-expert recombination."""
    experts, means, covs = _experts()
    w = np.array([0.3, 0.3, 0.4])
    gmx, gmo = GaussianMixture(w, means, covs, 0.8), MO.GaussianMixtureOracle(w, means, covs, 0.8)
    xq = np.random.default_rng(2).random((57, 2)) * [3.0, 1.0]
    mix = GpMixture(experts, gmx, recomb)
    val, var = mix.predict_valvar(xq)
    if recomb == "smooth":
        want_val, want_var = MO.predict_smooth(experts, gmo, xq), MO.predict_var_smooth(experts, gmo, xq)
    else:
        want_val, want_var = MO.predict_hard(experts, gmo, xq), MO.predict_var_hard(experts, gmo, xq)
    np.testing.assert_allclose(val, want_val, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(var, want_var, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(mix.predict(xq), want_val, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(mix.predict_var(xq), want_var, rtol=1e-8, atol=1e-12)


def test_expert_sharding_covers_all_experts():
    """Expert e on rank e mod G: the per-rank partial results sum to the single-process result."""
    experts, means, covs = _experts()
    gmx = GaussianMixture([1 / 3] * 3, means, covs)
    xq = np.random.default_rng(3).random((31, 2)) * [3.0, 1.0]
    for recomb in ("smooth", "hard"):
        full = GpMixture(experts, gmx, recomb).predict_valvar(xq)
        parts = []
        for r in range(2):
            m = GpMixture([e if i % 2 == r else None for i, e in enumerate(experts)], gmx, recomb, rank=r, world=2)
            m.world, m._allreduce = 2, (lambda *a: a)  # no process group here: add the partials by hand
            parts.append(m.predict_valvar(xq))
        np.testing.assert_allclose(parts[0][0] + parts[1][0], full[0], rtol=1e-13)
        np.testing.assert_allclose(parts[0][1] + parts[1][1], full[1], rtol=1e-13)


# ---------------------------------------------------------------- gradients of the mixture (algorithm.rs:691-783, 942-1010)
def test_probas_derivatives_match_oracle_and_finite_differences():
    _, means, covs = _experts()
    w = np.array([0.3, 0.3, 0.4])
    covs = covs.copy()
    covs[1] = [[0.3, 0.05], [0.05, 0.15]]
    gmx, gmo = GaussianMixture(w, means, covs, 0.8), MO.GaussianMixtureOracle(w, means, covs, 0.8)
    xq = np.random.default_rng(3).random((11, 2)) * [3.0, 1.0]
    pd = gmx.predict_probas_derivatives(xq)
    np.testing.assert_allclose(pd, gmo.predict_probas_derivatives(xq), rtol=1e-10, atol=1e-13)
    e = 1e-6
    for k in range(2):
        dq = np.zeros(2)
        dq[k] = e
        fd = (gmo.predict_probas(xq + dq) - gmo.predict_probas(xq - dq)) / (2 * e)
        np.testing.assert_allclose(pd[:, :, k], fd, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(pd.sum(axis=1), 0.0, atol=1e-12)  # responsibilities sum to one


@pytest.mark.parametrize("recomb", ["smooth", "hard"])
def test_gradient_recombination_with_oracle_experts(recomb):
    """Batched product logic vs the reference-shaped point-by-point recombination, and (smooth) vs central
    differences of the mixture's own predictions -- what the reference's test_variance_derivatives asserts."""
    experts, means, covs = _experts()
    w = np.array([0.3, 0.3, 0.4])
    gmx, gmo = GaussianMixture(w, means, covs, 0.8), MO.GaussianMixtureOracle(w, means, covs, 0.8)
    xq = np.random.default_rng(2).random((23, 2)) * [3.0, 1.0]
    mix = GpMixture(experts, gmx, recomb)
    gy, gv = mix.predict_valvar_gradients(xq)
    if recomb == "smooth":
        wy, wv = MO.predict_gradients_smooth(experts, gmo, xq), MO.predict_var_gradients_smooth(experts, gmo, xq)
    else:
        wy, wv = MO.predict_gradients_hard(experts, gmo, xq), MO.predict_var_gradients_hard(experts, gmo, xq)
    np.testing.assert_allclose(gy, wy, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(gv, wv, rtol=1e-8, atol=1e-11)
    np.testing.assert_array_equal(mix.predict_gradients(xq), gy)
    np.testing.assert_array_equal(mix.predict_var_gradients(xq), gv)
    if recomb == "smooth":
        e = 1e-6
        for k in range(2):
            dq = np.zeros(2)
            dq[k] = e
            fy = (mix.predict(xq + dq) - mix.predict(xq - dq)) / (2 * e)
            fv = (mix.predict_var(xq + dq) - mix.predict_var(xq - dq)) / (2 * e)
            np.testing.assert_allclose(gy[:, k], fy, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(gv[:, k], fv, rtol=1e-5, atol=1e-7)


def test_precisions_chol_through_the_c_abi():
    """egx_gmx_precisions_chol is host arithmetic (no GPU): (chol(cov)^-1)^T per cluster as gaussian_mixture.rs:182-205,
    against the oracle's; a covariance that is not positive definite is a LinalgError."""
    from egobox_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(3)
    for k, nx in ((1, 1), (3, 2), (8, 16), (2, 40)):
        a = rng.standard_normal((k, nx, nx))
        covs = np.ascontiguousarray(np.einsum("kij,klj->kil", a, a) + 0.5 * np.eye(nx))
        out = np.empty((k, nx, nx))
        L.check(lib.egx_gmx_precisions_chol(L.dptr(covs), k, nx, L.dptr(out)))
        w = np.full(k, 1.0 / k)
        ref = MO.GaussianMixtureOracle(w, np.zeros((k, nx)), covs).precisions_chol
        np.testing.assert_allclose(out, ref, rtol=1e-11, atol=1e-13 * np.abs(ref).max())
        assert np.all(np.tril(out, -1) == 0.0)  # upper triangular, as the reference stores it
    bad = np.array([[[1.0, 2.0], [2.0, 1.0]]])
    with pytest.raises(L.LinalgError):
        L.check(lib.egx_gmx_precisions_chol(L.dptr(bad), 1, 2, L.dptr(np.empty((1, 2, 2)))))
