"""Host-side mirror logic (builder semantics, multistart candidates, spec handling) -- no GPU needed."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def egx():
    import egobox_amd
    return egobox_amd


def test_builder_defaults_match_reference(egx):
    p = egx.GaussianProcess.params(egx.ConstantMean(), egx.SquaredExponentialCorr())
    assert p._n_start == 10 and p._max_eval == 1000            # crates/gp/src/lib.rs constants
    assert p._nugget == 100.0 * np.finfo(float).eps             # parameters.rs:118
    t = p._theta_tuning
    assert t.kind == "Full" and t.init.tolist() == [0.1] and t.bounds == [(1e-2, 1e1)]  # parameters.rs:36-51


def test_builder_theta_init_bounds_transitions(egx):
    # parameters.rs:193-234
    p = egx.Kriging.params().theta_init([0.5, 0.6])
    assert p._theta_tuning.kind == "Full" and p._theta_tuning.init.tolist() == [0.5, 0.6]
    p.theta_bounds([(0.1, 1.0)])
    assert p._theta_tuning.bounds == [(0.1, 1.0)] and p._theta_tuning.init.tolist() == [0.5, 0.6]
    p.theta_tuning(egx.ThetaTuning.Fixed([0.3]))
    p.theta_bounds([(0.2, 2.0)])          # no-op when Fixed
    assert p._theta_tuning.kind == "Fixed" and p._theta_tuning.bounds is None
    p.theta_init([0.9])
    assert p._theta_tuning.kind == "Fixed" and p._theta_tuning.init.tolist() == [0.9]
    p2 = egx.Kriging.params().theta_tuning(egx.ThetaTuning.Partial([0.1, 0.1], [(0.01, 10)], [0])).theta_init([0.2, 0.2])
    assert p2._theta_tuning.kind == "Full"
    assert egx.Kriging.params().max_eval(3)._max_eval == 25     # parameters.rs:251-254


def test_builder_check(egx):
    with pytest.raises(egx.InvalidValueError, match="canot be 0"):
        egx.Kriging.params().kpls_dim(0).check()                # parameters.rs:290-294
    with pytest.raises(egx.InvalidValueError, match="Dimension reduction"):
        egx.Kriging.params().theta_init([0.1, 0.1]).kpls_dim(3).check()  # parameters.rs:296-304
    egx.Kriging.params().theta_init([0.1]).kpls_dim(3).check()  # len-1 theta: no constraint


def test_model_markers(egx):
    assert str(egx.ConstantMean()) == "ConstantMean" and str(egx.Matern52Corr()) == "Matern52"
    assert egx.LinearMean() == egx.LinearMean() and egx.LinearMean() != egx.QuadraticMean()
    assert [c().code for c in (egx.SquaredExponentialCorr, egx.AbsoluteExponentialCorr, egx.Matern32Corr,
                               egx.Matern52Corr)] == [0, 1, 2, 3]


def test_prepare_multistart(egx):
    # optimization.rs:26-71: log10 space, row 0 = user theta0, rows 1.. inside the log10 bounds
    starts, bl = egx.prepare_multistart(10, np.array([0.1, 0.5, 2.0]), [(1e-2, 1e1)] * 3)
    assert starts.shape == (11, 3)
    np.testing.assert_allclose(starts[0], np.log10([0.1, 0.5, 2.0]))
    assert bl == [(-2.0, 1.0)] * 3
    assert np.all(starts[1:] >= -2.0) and np.all(starts[1:] <= 1.0)
    # LHS property: one point per stratum in every column
    for j in range(3):
        strata = np.floor((starts[1:, j] + 2.0) / 3.0 * 10).astype(int)
        assert sorted(strata.tolist()) == list(range(10))
    s2, _ = egx.prepare_multistart(10, np.array([0.1, 0.5, 2.0]), [(1e-2, 1e1)] * 3)
    np.testing.assert_array_equal(starts, s2)                   # seeded (42): reproducible
    s0, _ = egx.prepare_multistart(0, np.array([0.3]), [(1e-2, 1e1)])
    assert s0.shape == (1, 1)
    s1, _ = egx.prepare_multistart(1, np.array([0.3]), [(1e-2, 1e1)])
    assert s1.shape == (2, 1) and -2.0 <= s1[1, 0] <= 1.0


def test_theta_sweep_candidates(egx):
    c = egx.theta_sweep_candidates(512, 32)
    assert c.shape == (512, 32)
    assert np.all(c[0] == 0.1)
    assert c.min() >= 1e-2 * (1 - 1e-12) and c.max() <= 1e1 * (1 + 1e-12)


def test_gpx_spec_handling(egx):
    with pytest.raises(NotImplementedError):
        egx.Gpx.builder(regr_spec=egx.RegressionSpec.ALL).fit(np.random.rand(5, 1), np.random.rand(5))
    with pytest.raises(NotImplementedError):
        egx.Gpx.builder(n_clusters=3).fit(np.random.rand(5, 1), np.random.rand(5))
    with pytest.raises(NotImplementedError):
        egx.Gpx.builder(corr_spec=egx.CorrelationSpec.SQUARED_EXPONENTIAL | egx.CorrelationSpec.MATERN52) \
            .fit(np.random.rand(5, 1), np.random.rand(5))


def test_workload_generators(egx):
    x = egx.workload.lhs(100, 3, seed=1)
    assert x.shape == (100, 3) and x.min() >= 0 and x.max() <= 1
    for j in range(3):
        assert sorted(np.floor(x[:, j] * 100).astype(int).tolist()) == list(range(100))
    np.testing.assert_array_equal(x, egx.workload.lhs(100, 3, seed=1))
    g = egx.workload.griewank(np.full((1, 4), 0.5))  # x = 0 -> griewank = 0
    assert abs(g[0]) < 1e-12
    from oracle import gp_oracle as O
    np.testing.assert_array_equal(O.lhs_classic(50, 2, 7), egx.workload.lhs(50, 2, 7))
    np.testing.assert_allclose(O.griewank(x), egx.workload.griewank(x))


def test_host_cpp_under_address_and_ub_sanitizers(tmp_path):
    """egobox_amd/csrc/host_math.h (normalisation, trend basis and its jacobian, Householder QR with
    positive diagonal, Jacobi singular values, the box-constrained Nelder-Mead) built with -fsanitize=address,undefined
    and checked against independent formulas (tests/c_host/host_math_sanitized.cpp)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "host_math_sanitized"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Werror", "-fsanitize=address,undefined",
                    "-fno-omit-frame-pointer", os.path.join(root, "tests", "c_host", "host_math_sanitized.cpp"),
                    "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert out.stdout.startswith("OK") and "runtime error" not in out.stderr


def test_griewank_known_answer(golden_dir):
    """The synthetic workload's target function against the reference's own pin (algorithm.rs:1319-1323): both the
    oracle's and the product's griewank take unit-cube inputs and map them to [-600, 600]^d."""
    import json
    import os
    import egobox_amd.workload as W
    from oracle import gp_oracle as O
    k = json.load(open(os.path.join(golden_dir, "kat.json")))["griewank"]
    x01 = (np.array(k["x"]) + 600.0) / 1200.0
    np.testing.assert_allclose(W.griewank(x01), k["expected"], atol=k["tol"])
    np.testing.assert_allclose(O.griewank(x01), k["expected"], atol=k["tol"])


def test_pls_rotations_match_scikit_learn_up_to_sign():
    """egobox_amd/kpls.py restates the PLS1 regression the reference takes from linfa-pls (a port of scikit-learn's);
    scikit-learn is only the checker here.  Column signs are a convention the kernels never see."""
    sk = pytest.importorskip("sklearn.cross_decomposition")
    from egobox_amd.kpls import pls_rotations
    rng = np.random.default_rng(0)
    x = rng.random((100, 5)) * 1200 - 600
    i = np.arange(1, 6)
    y = (x * x).sum(1) / 4000 - np.prod(np.cos(x / np.sqrt(i)), axis=1) + 1
    for k in (1, 3, 5):
        mine = pls_rotations(x, y, k)
        ref = sk.PLSRegression(n_components=k).fit(x, y).x_rotations_
        np.testing.assert_allclose(np.abs(mine), np.abs(ref), rtol=1e-8, atol=1e-10)
    assert np.all(pls_rotations(x, np.full(100, 3.1), 2) == 0.0)  # constant response -> zero weights (algorithm.rs:846-850)


def test_one_launch_back_substitution_cannot_deadlock_on_four_resident_workgroups():
    """The scheduling argument of k_trsv_t_fused (egobox_amd/csrc/kernels_chol.hip), restated and run: segments (64 columns) are
    handed out by a START ticket from the last one down; a workgroup waits (a) for the solutions x_b of every block behind its own,
    published by the segment workgroups of those blocks, and (b) for the finished right-hand-side pieces of the other segments of
    its own block before it publishes its rows of x_blk.  With R workgroups resident at a time (a new one starts, in ticket
    order, whenever one leaves) every segment finishes for every R >= 4 -- and R = 3 shows the rule is tight."""
    def run(nseg, resident):
        nblk = (nseg + 3) // 4
        segs_of = lambda b: [s for s in range(4 * b, min(4 * b + 4, nseg))]  # noqa: E731
        r_final, x_rows = set(), set()      # segments whose piece of rho is final / whose rows of x are published
        x_done = lambda b: all(s in x_rows for s in segs_of(b))  # noqa: E731
        state = {}                          # running workgroup: segment -> next block it waits for (None: in its mat-vec)
        next_ticket, finished = 0, 0
        while finished < nseg:
            while len(state) < resident and next_ticket < nseg:   # the dispatcher starts workgroups as slots free up
                seg = nseg - 1 - next_ticket
                state[seg] = nblk - 1
                next_ticket += 1
            progressed = False
            for seg in list(state):
                blk = seg // 4
                b = state[seg]
                while b is not None and b > blk and x_done(b):     # apply the blocks' solutions in order
                    b -= 1
                    progressed = True
                if b is not None and b == blk:                     # own block is next: publish the piece, then the mat-vec
                    r_final.add(seg)
                    b = None
                    progressed = True
                state[seg] = b
                if b is None and all(s in r_final for s in segs_of(blk)):
                    x_rows.add(seg)
                    del state[seg]
                    finished += 1
                    progressed = True
            if not progressed:
                return False
        return True
    for nseg in (2, 4, 6, 34, 64, 130, 256):        # n_pad / 64, incl. a last block of two segments
        for resident in (4, 5, 8, 33, 256):
            assert run(nseg, resident), (nseg, resident)
    assert not run(64, 3)
