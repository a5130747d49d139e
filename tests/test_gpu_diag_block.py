"""The diagonal-block kernel of the blocked Cholesky (k_potf2_reg, csrc/kernels_chol.hip) through egx_potrf, against LAPACK:
the cases that are specific to its construction -- the explicit 16x16 inverses of its in-block triangular solves (with one
refinement step for ill-conditioned tiles), the `info` of a failed pivot in any strip, partial blocks, and the other
variants of the kernel (EGX_POTF2_REG = 8 / 12 waves, 0 = the LDS-tile kernel of round 1), each in a process of its own.

The reference factors with LAPACK / linfa-linalg `cholesky()` (algorithm.rs:1004, :1077): dpotrf here is that oracle.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.linalg as sl

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def egx():
    import egobox_amd
    return egobox_amd


def _collinear_kernel(n, nugget=100 * np.finfo(float).eps):
    """Squared-exponential correlation matrix of n collinear points: numerical rank ~ 12, positive definite by its nugget
    alone (smallest pivots ~ 4e-11) -- the matrix of test_status_channel's second half."""
    x = np.linspace(0.0, 1.0, n)
    xn = (x - x.mean()) / x.std(ddof=1)
    return np.exp(-((xn[:, None] - xn[None, :]) ** 2)) + nugget * np.eye(n)


def _residual(a, l):
    return np.abs(l @ l.T - a).max() / np.abs(a).max()


@pytest.mark.parametrize("n", [24, 40, 64, 100, 200, 256])
def test_rank_deficient_kernel_matrix_stays_positive_definite(egx, n):
    """LAPACK factors it; a triangular solve by an explicitly inverted 16x16 tile WITHOUT refinement loses a pivot in
    strip 1 or 2 (residual eps * cond(L16) ~ 1e-9 against pivots of 4e-11).  The backward error of what comes out must
    be the usual few eps, not eps * cond.  (Up to one diagonal block: below it the panel solve k_panel_trsm applies
    explicit 64x64 inverses without refinement, backward error ~ eps * cond(L64), see the next test and DESIGN.md section 7.)"""
    a = _collinear_kernel(n)
    want = np.linalg.cholesky(a)
    got, info = egx.potrf(a)
    assert info == 0
    assert _residual(a, got) < 50 * np.finfo(float).eps
    assert _residual(a, got) < 10 * max(_residual(a, want), np.finfo(float).eps)
    assert np.all(np.triu(got, 1) == 0.0)


@pytest.mark.parametrize("n,bad", [(30, 0), (30, 15), (30, 16), (30, 17), (300, 100), (300, 255), (300, 256), (300, 257),
                                   (300, 299), (700, 511), (700, 640)])
def test_info_is_lapacks_for_a_pivot_failing_anywhere(egx, n, bad):
    """First non-positive pivot in the first / last column of a 16-column strip, of a 256 block, of the matrix."""
    rng = np.random.default_rng(n + bad)
    g = rng.standard_normal((n, n))
    a = g @ g.T / n + 0.1 * np.eye(n)
    a[bad, bad] = -abs(a[bad, bad]) if bad == 0 else a[bad, :bad] @ np.linalg.solve(a[:bad, :bad], a[:bad, bad]) - 1e-3
    _, lapack_info = sl.lapack.dpotrf(a, lower=1)
    assert lapack_info == bad + 1
    _, info = egx.potrf(a)
    assert info == lapack_info


@pytest.mark.parametrize("n", [16, 17, 128, 129, 255, 256, 257, 383, 384, 385])
def test_partial_blocks_and_strips(egx, n):
    """n around the strip (16), diagonal-block (128 / 256) and padding (128) boundaries; a matrix whose factor has entries
    of every magnitude down to 1e-6."""
    rng = np.random.default_rng(n)
    t = np.sort(rng.random(n)) * 3.0
    a = np.exp(-((t[:, None] - t[None, :]) ** 2) / 0.02) + 1e-10 * np.eye(n)
    got, info = egx.potrf(a)
    assert info == 0
    # inside one diagonal block the backward error is LAPACK's (measured 2.5 - 4 eps, LAPACK 1.5 - 3; round 1's kernel:
    # 70 - 1600 eps); with a panel solve below it, eps * cond(L64) ~ 3000 eps on this matrix (cond(L) = 5e5)
    assert _residual(a, got) < (50 if n <= 256 else 2e4) * np.finfo(float).eps
    # (no entry-wise comparison with LAPACK's factor: two backward-stable factors of a matrix with cond 3e11 differ by
    #  eps * cond ~ 1e-4; well-conditioned matrices are compared entry-wise in test_gpu_parity.py::test_potrf_vs_lapack)
    assert np.all(np.triu(got, 1) == 0.0) and np.all(np.diag(got) > 0.0)


_VARIANT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import egobox_amd as egx
rng = np.random.default_rng(5)
out = []
for n in (100, 300, 700):
    g = rng.standard_normal((n, n))
    a = g @ g.T / n + 0.05 * np.eye(n)
    l, info = egx.potrf(a)
    out.append((info, float(np.abs(l - np.linalg.cholesky(a)).max())))
x = np.linspace(0.0, 1.0, 40); xn = (x - x.mean()) / x.std(ddof=1)
a = np.exp(-((xn[:, None] - xn[None, :]) ** 2)) + 100 * np.finfo(float).eps * np.eye(40)
l, info = egx.potrf(a)
out.append((info, float(np.abs(l @ l.T - a).max())))
a[17, 17] = -1.0
out.append((egx.potrf(a)[1], 0.0))
print("RESULT", out)
"""


@pytest.mark.parametrize("group", ["1", "2", "4"])
def test_every_group_width_of_the_factorisation(group):
    """EGX_POTRF_GROUP (panels per trailing update) is read once per process: each width gets its own."""
    env = dict(os.environ, EGX_POTRF_GROUP=group)
    r = subprocess.run([sys.executable, "-c", _VARIANT % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1]
    res = eval(line[len("RESULT"):])  # noqa: S307 - our own output
    for info, err in res[:3]:
        assert info == 0 and err < 1e-12
    assert res[3][0] == 0 and res[3][1] < 1e-14
    assert res[4][0] == 18


def test_fuzz_against_lapack(egx):
    """250 random matrices, n = 1 .. 1600, four families (Wishart + shift, 1-D Gaussian kernel with nugget 1e-13 .. 1e-6,
    product-exponential kernel in 1 - 5 dimensions, rank n / 2 + shift 1e-14 .. 1e-10), rows permuted: the same success /
    failure as dpotrf wherever LAPACK's smallest pivot is above rounding level, LAPACK's `info` on failure, and a backward
    error of a few eps (2 200 cases of this generator on the GPU box: worst 15 eps, no disagreement of any kind)."""
    rng = np.random.default_rng(20260927)
    eps = np.finfo(float).eps
    worst = 0.0
    for _ in range(250):
        n = int(rng.integers(1, 1600))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            g = rng.standard_normal((n, n))
            a = g @ g.T / n + 10.0 ** rng.uniform(-6, 0) * np.eye(n)
        elif kind == 1:
            t = np.sort(rng.random(n)) * rng.uniform(0.5, 20)
            a = np.exp(-((t[:, None] - t[None, :]) ** 2) / 10.0 ** rng.uniform(-3, 0)) + 10.0 ** rng.uniform(-13, -6) * np.eye(n)
        elif kind == 2:
            d = int(rng.integers(1, 6))
            x = rng.random((n, d))
            a = np.exp(-(np.abs(x[:, None, :] - x[None, :, :]) * 10.0 ** rng.uniform(-2, 1, d)).sum(-1)) + 1e-12 * np.eye(n)
        else:
            g = rng.standard_normal((n, max(1, n // 2)))
            a = g @ g.T / n + 10.0 ** rng.uniform(-14, -10) * np.eye(n)
        p = rng.permutation(n)
        a = a[np.ix_(p, p)]
        lw, lapack_info = sl.lapack.dpotrf(a, lower=1)
        got, info = egx.potrf(a)
        if lapack_info == 0 and info == 0:
            worst = max(worst, _residual(a, got) / eps)
            assert np.all(np.triu(got, 1) == 0.0)
        elif (lapack_info == 0) != (info == 0):
            margin = (np.diag(lw).min() ** 2 / np.abs(a).max()) if lapack_info == 0 else 0.0
            assert margin < 1e-11, (n, kind, lapack_info, info, margin)
        else:
            assert info == lapack_info
    assert worst < 100.0, worst
