"""The pipelined chain kernel (csrc/kernels_pipe.hip: the serial chain of a group of Cholesky panels -- diagonal blocks,
panel solves, in-group updates: the panel step of `cholesky()`, crates/gp/src/algorithm.rs:1004 -- as ONE persistent launch
with device-side hand-offs) through the C ABI: against the separate-launch chain and LAPACK, the schedule a handle reports,
a lost pivot, and the bounded waits (a hand-off that never arrives is an ERROR within the bound, not a hang)."""
import time

import numpy as np
import pytest
import scipy.linalg as sl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def egx():
    import egobox_amd
    return egobox_amd


def _data(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(n, d))
    y = np.sin(3 * x[:, 0]) + x[:, 1:].sum(axis=1) ** 2 + 0.1 * rng.standard_normal(n)
    return x, y


@pytest.fixture()
def knobs(egx):
    """egx_set_tuning settings of one test, restored afterwards"""
    saved = {}

    def setk(name, value):
        old = egx.set_tuning(name, value)
        saved.setdefault(name, old)
        return old
    yield setk
    for k, v in saved.items():
        egx.set_tuning(k, v)


@pytest.mark.parametrize("n", [300, 1000, 2100, 4096, 5000])
def test_chain_launches_agree_with_separate_launches_and_lapack(egx, knobs, n):
    """egx_potrf on a kernel matrix: one chain launch for the whole factorisation (the default up to 4096 columns), chain
    launches per group of panels (pipe = 2), separate launches (pipe = 0): the same factor to rounding, LAPACK's
    residual."""
    rng = np.random.default_rng(n)
    pts = rng.uniform(size=(n, 3))
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    a = np.exp(-6.0 * d2) + 1e-8 * np.eye(n)
    want = sl.cholesky(a, lower=True)
    res_lapack = np.abs(want @ want.T - a).max()
    facs = {}
    for name, settings in (("whole", {"pipe": 1}), ("groups", {"pipe": 2}), ("separate", {"pipe": 0})):
        for k, v in settings.items():
            knobs(k, v)
        got, info = egx.potrf(a)
        assert info == 0
        assert np.abs(got @ got.T - a).max() <= 10 * max(res_lapack, 1e-15), name
        facs[name] = got
    scale = np.abs(want).max()
    assert np.abs(facs["whole"] - facs["separate"]).max() <= 2e-5 * scale     # (cond ~ 1e8: the factors agree to 1e-7)
    assert np.abs(facs["groups"] - facs["separate"]).max() <= 2e-5 * scale
    if n <= 256:
        np.testing.assert_array_equal(facs["groups"], facs["separate"])      # one panel: the same arithmetic, the same bits


@pytest.mark.parametrize("n,bad", [(700, 0), (700, 255), (700, 256), (700, 300), (700, 699), (2000, 1500)])
def test_info_is_lapacks_under_chain_launches(egx, knobs, n, bad):
    rng = np.random.default_rng(n + bad)
    g = rng.standard_normal((n, n))
    a = g @ g.T / n + 0.1 * np.eye(n)
    a[bad, bad] = -1.0
    _, info_lapack = sl.lapack.dpotrf(a, lower=1)
    for form in (1, 2):
        knobs("pipe", form)
        _, info = egx.potrf(a)
        assert info == info_lapack == bad + 1


def test_schedule_is_reported_and_survives_shrink(egx):
    """egx_gp_get_schedule: a one-workspace handle up to 7168 columns factors as ONE chain launch, a small handle with many
    workspaces per group, a larger one with several workspaces by separate launches; egx_gp_shrink does not change what the handle was created with."""
    x, y = _data(900, 4, 1)
    with egx.GpHandle(x, y, n_workspaces=1) as h:
        s = h.schedule()
        assert s["pipelined_chain"] == 1 and s["whole_factorisation_launch"] == 1 and s["left_looking"] == 0
    with egx.GpHandle(x, y, n_workspaces=12) as h:      # 12 x 4 panels > 32 diagonal blocks
        s = h.schedule()
        assert s["pipelined_chain"] == 1 and s["whole_factorisation_launch"] == 0
        th = np.full(4, 0.3)
        lk0, st0 = h.likelihood(th)
        h.finalize(th)
        h.shrink(1)
        assert h.schedule() == {**s, "lockstep": 1}      # the width is capped, the schedule stays
        lk1, st1 = h.likelihood(th)
        assert st0 == st1 == 0 and lk0 == lk1           # ... and so do the bits
    x, y = egx.workload.make_training_set(4200, 8, 3)   # (the benchmark's well-conditioned family)
    th = egx.workload.default_theta(8) * 3.0
    with egx.GpHandle(x, y, n_workspaces=1) as h:       # a lone matrix up to 7168 columns: still one launch (17 diagonal blocks)
        s = h.schedule()
        assert s["pipelined_chain"] == 1 and s["whole_factorisation_launch"] == 1
        lk_whole, st = h.likelihood(th)
        assert st == 0
    with egx.GpHandle(x, y, n_workspaces=2) as h:       # 2 x 17 diagonal blocks > 32, beyond 4096 columns: separate launches
        s = h.schedule()
        assert s["pipelined_chain"] == 0 and s["whole_factorisation_launch"] == 0
        lk_sep, st = h.likelihood(th)
        assert st == 0 and abs(lk_whole - lk_sep) <= 1e-9 * abs(lk_sep)


@pytest.mark.parametrize("n", [5600, 6400, 8192])
def test_flow_launch_agrees_with_separate_launches_and_lapack(egx, knobs, n):
    """Round 6: a lone matrix between 5376 and 14080 padded columns factors as ONE flow launch (csrc/pipe_flow.h, k_potrf_flow:
    critical heads / tails per stage, bulk-class rounds per column with 128 x 256 LDS-DMA tiles, look-before-claim).  egx_potrf
    takes the schedule of a one-workspace handle: the factor against LAPACK and against the separate launches (pipe = 0), the
    same bits on a second call (the schedule is dynamic, the arithmetic is not), LAPACK's info for a lost pivot."""
    rng = np.random.default_rng(n)
    pts = rng.uniform(size=(n, 3))
    a = np.exp(-6.0 * ((pts[:, None, 0] - pts[None, :, 0]) ** 2 + (pts[:, None, 1] - pts[None, :, 1]) ** 2 + (pts[:, None, 2] - pts[None, :, 2]) ** 2))
    a[np.diag_indices(n)] += 1e-8
    want = sl.cholesky(a, lower=True)
    res_lapack = np.abs(want @ want.T - a).max()
    s0 = egx.chain_stats()
    got, info = egx.potrf(a)
    assert info == 0 and np.abs(got @ got.T - a).max() <= 10 * max(res_lapack, 1e-15)
    again, _ = egx.potrf(a)
    np.testing.assert_array_equal(got, again)
    knobs("pipe", 0)
    sep, info = egx.potrf(a)
    knobs("pipe", 1)
    assert info == 0 and np.abs(got - sep).max() <= 2e-5 * np.abs(want).max()
    bad = n // 2 + 77
    a[bad, bad] = -1.0
    _, info_lapack = sl.lapack.dpotrf(a, lower=1)
    _, info = egx.potrf(a)
    assert info == info_lapack == bad + 1
    assert egx.chain_stats() == s0          # no launch ran into its wait bound


def test_a_lone_handle_takes_the_flow_launch_and_its_likelihood_is_the_separate_launches_to_rounding(egx, knobs):
    x, y = egx.workload.make_training_set(6000, 8, 11)       # (the benchmark's well-conditioned family: the 1e-8 bar applies)
    th = egx.workload.default_theta(8) * 3.0
    xq = np.random.default_rng(3).random((200, 8))
    with egx.GpHandle(x, y) as h:
        s = h.schedule()
        assert s["flow"] == 1 and s["lockstep"] == 1
        lk, st = h.likelihood(th)
        lk2, _ = h.likelihood(th)
        assert st == 0 and lk == lk2
        g_lk, g, gst = h.likelihood_grad(th)                   # the C^-T rider follows the launch group by group
        assert gst == 0 and g_lk == pytest.approx(lk, rel=1e-12)
        h.finalize(th)
        yp, vp = h.predict_valvar(xq)
    with egx.GpHandle(x, y, n_workspaces=2) as h2:             # two workspaces: separate launches (no flow, no chain launch at this size)
        assert h2.schedule()["flow"] == 0
        lk_sep, st = h2.likelihood(th)
        g2_lk, g2, _ = h2.likelihood_grad(th)
        h2.finalize(th)
        yp2, vp2 = h2.predict_valvar(xq)
    assert st == 0 and abs(lk - lk_sep) <= 1e-10 * abs(lk_sep)
    np.testing.assert_allclose(g, g2, rtol=1e-6, atol=1e-8 * np.abs(g2).max())
    np.testing.assert_allclose(yp, yp2, rtol=1e-8)
    np.testing.assert_allclose(vp, vp2, rtol=1e-6, atol=1e-12)


def test_likelihood_does_not_depend_on_the_chain_form_beyond_rounding(egx, knobs):
    x, y = egx.workload.make_training_set(1500, 8, 42)     # (the benchmark's well-conditioned family: the 1e-8 bar applies)
    th = egx.workload.default_theta(8) * 3.0
    vals = []
    for settings in ({"pipe": 1}, {"pipe": 2}, {"pipe": 0}):
        for k, v in settings.items():
            knobs(k, v)
        with egx.GpHandle(x, y) as h:
            lk, st = h.likelihood(th)
            assert st == 0
            vals.append(lk)
    assert abs(vals[0] - vals[2]) <= 1e-10 * abs(vals[2]) and abs(vals[1] - vals[2]) <= 1e-10 * abs(vals[2])


_FORCED_TIMEOUT_SCRIPT = r"""
import sys, time, json
import numpy as np
import egobox_amd as egx
x, y = egx.workload.make_training_set(1000, 4, 4)   # (the benchmark's well-conditioned family: chain and separate launches agree to 1e-10)
th = egx.workload.default_theta(4) * 3.0
out = {}
with egx.GpHandle(x, y) as h:
    assert h.schedule()["pipelined_chain"] == 1
    good, st = h.likelihood(th)
    assert st == 0
    egx.set_tuning("pipe", 0)
    with egx.GpHandle(x, y) as hs:                     # what the fallback computes: the separate-launch schedule
        sep, st = hs.likelihood(th)
    egx.set_tuning("pipe", 1)
    egx.set_tuning("pipe_timeout_ms", 25)
    egx.set_tuning("pipe_stall", 1)                    # test build only: the diagonal role publishes into a scratch word
    s0 = egx.chain_stats()
    t0 = time.perf_counter()
    lk, st = h.likelihood(th)                          # product semantics: SUCCESS through the fallback
    out["fallback_seconds"] = time.perf_counter() - t0
    s1 = egx.chain_stats()
    out["fallback"] = [lk, int(st), sep, good, s1["aborted"] - s0["aborted"], s1["retried"] - s0["retried"]]
    h.finalize(th)                                     # ... also for a fit (the factor the predictions use)
    out["fit_after_fallback"] = h.fitted_scalars()[0]
    egx.set_tuning("pipe_retry", 0)                    # fallback disabled: the error, within the bound, not a hang
    t0 = time.perf_counter()
    try:
        h.likelihood(th)
        out["error"] = None
    except egx.EgxError as e:
        out["error"] = str(e)
    out["error_seconds"] = time.perf_counter() - t0
    egx.set_tuning("pipe_stall", 0)
    egx.set_tuning("pipe_retry", 1)
    egx.set_tuning("pipe_timeout_ms", 2000)
    again, st = h.likelihood(th)
    out["again"] = [again, int(st)]
a = np.eye(600) + 0.01
egx.set_tuning("pipe_timeout_ms", 25)
egx.set_tuning("pipe_stall", 1)
got, info = egx.potrf(a)                               # egx_potrf retries too
out["potrf_resid"] = float(np.abs(got @ got.T - a).max())
egx.set_tuning("pipe_retry", 0)
try:
    egx.potrf(a)
    out["potrf_error"] = None
except egx.EgxError as e:
    out["potrf_error"] = str(e)
print("RESULT " + json.dumps(out))
"""


def _run_script(script, env_extra, timeout=300):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), **env_extra)
    return subprocess.run([sys.executable, "-c", script], env=env, cwd=root, capture_output=True, text=True, timeout=timeout)


def test_a_hand_off_that_never_arrives_is_retried_by_separate_launches_and_an_error_only_without_the_fallback(egx):
    """The reference's cholesky() (algorithm.rs:1004) never fails a positive-definite matrix, and the only errors its objective
    sees are numerical (:893-896).  A chain launch whose bounded wait runs out (forced here: the TEST build of the library --
    EGX_TEST_LIBRARY=1, in a process of its own -- lets the diagonal role publish into a scratch word) is therefore run once more
    by separate launches: the caller gets the separate-launch likelihood, egx_chain_stats counts it; with the fallback switched
    off it is EGX_ERR_HIP within the bound, not a hang, and the next evaluation on the same handle is clean."""
    import json
    r = _run_script(_FORCED_TIMEOUT_SCRIPT, {"EGX_TEST_LIBRARY": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    lk, st, sep, good, aborted, retried = out["fallback"]
    # (the fallback IS the separate-launch schedule: the same bits as a handle created with chain launches off; against the chain
    #  launch's own value the two orders of summation agree within the parity bar, 6e-9 here)
    assert st == 0 and lk == sep and abs(lk - good) <= 1e-8 * abs(good)
    assert aborted == 1 and retried == 1 and out["fallback_seconds"] < 5.0
    assert abs(out["fit_after_fallback"] - good) <= 1e-8 * abs(good)
    assert out["error"] is not None and "pipelined chain kernel" in out["error"] and out["error_seconds"] < 5.0
    assert out["again"] == [good, 0]
    assert out["potrf_resid"] < 1e-12 and out["potrf_error"] is not None


def test_the_product_library_has_no_test_hooks(egx):
    with pytest.raises(egx.InvalidValueError, match="unknown knob"):
        egx.set_tuning("pipe_stall", 1)


_SHARED_GPU_SCRIPT = r"""
import sys, time, json
import numpy as np
import egobox_amd as egx
x, y = egx.workload.make_training_set(4096, 8, 3)
th = egx.workload.default_theta(8) * 3.0
n_ok = 0
with egx.GpHandle(x, y) as h:
    assert h.schedule()["whole_factorisation_launch"] == 1
    ref, st = h.likelihood(th)
    t_end = time.perf_counter() + float(sys.argv[1])
    while time.perf_counter() < t_end:
        lk, st = h.likelihood(th)
        assert st == 0 and abs(lk - ref) <= 1e-9 * abs(ref), (lk, ref)
        n_ok += 1
print("RESULT " + json.dumps({"evaluations": n_ok, **egx.chain_stats()}))
"""


def test_two_processes_share_the_gpu_with_a_tight_wait_bound_and_nobody_fails(egx):
    """Two processes on ONE GPU, each looping whole-factorisation chain launches at n = 4096 for 10 s with a 50 ms wait bound:
    time slicing between the processes may pre-empt a launch beyond the bound -- every evaluation still succeeds (through the
    fallback when it must; the retries are counted and reported, not asserted)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), EGX_PIPE_TIMEOUT_MS="50")
    procs = [subprocess.Popen([sys.executable, "-c", _SHARED_GPU_SCRIPT, "10"], env=env, cwd=root, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    res = [json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:]) for so, _ in outs]
    print("shared GPU, 50 ms bound:", res)
    assert all(r["evaluations"] > 100 for r in res)


def test_three_lockstep_slots_with_look_ahead_chain_launches_in_flight(egx):
    """n_pad = 4096 with 36 workspaces: three lock-step slots of twelve, each with a look-ahead chain launch whose first panel
    waits for the rest of its columns' update.  With three such launches in flight the 3 x 96 workgroups that would spin on the
    device could leave that update no compute unit to run on: from three sequences on the wait is the stream's.
    Every candidate must come back, with the bits it gets in a batch of one slot."""
    x, y = egx.workload.make_training_set(4000, 8, 5)
    base = egx.workload.default_theta(8) * 3.0
    thetas = np.stack([base * (1.0 + 0.01 * j) for j in range(36)])
    with egx.GpHandle(x, y, n_workspaces=36) as h:
        s = h.schedule()
        assert s["pipelined_chain"] == 1 and s["whole_factorisation_launch"] == 0 and s["lockstep"] == 12
        a0 = egx.chain_stats()["aborted"]
        lk, st = h.likelihood_batch(thetas)
        assert np.all(st == 0) and egx.chain_stats()["aborted"] == a0
        lk1, st1 = h.likelihood_batch(thetas[:12])
        np.testing.assert_array_equal(lk[:12], lk1)


# ---- lock-step across MODELS (egx_gp_create_group / egx_gp_finalize_multi): the expert loop of egobox-moe
#      (crates/moe/src/algorithm.rs:167-177), EGO's objective + constraint surrogates (crates/ego/src/solver/solver_impl.rs:370-391)
# (a group takes the schedule of a LONE handle of its shape -- that is what makes a member's fit bit for bit its lone fit --, so
#  up to 7168 columns it is ONE whole-factorisation launch with grid.z = members; k = 12 at n = 1000 is the widest such launch:
#  48 diagonal blocks, beyond the 32 a handle's own workspaces are allowed.  The "chain launch per group of panels" row of
#  schedule.h is taken by handles with many WORKSPACES only, never by a group: test_three_lockstep_slots_... runs it.)
@pytest.mark.parametrize("n,d,k,mean,corr", [(700, 4, 5, 0, 0), (2100, 5, 3, 1, 3), (4200, 6, 3, 0, 0), (1000, 3, 12, 0, 0)])
def test_models_of_a_group_fitted_in_lock_step_are_their_lone_fits_bit_for_bit(egx, n, d, k, mean, corr):
    sets = [_data(n, d, 50 + j) for j in range(k)]
    thetas = np.stack([np.full(d, 0.35 + 0.05 * j) for j in range(k)])
    xq = np.random.default_rng(1).uniform(size=(64, d))
    lone = []
    for (x, y), th in zip(sets, thetas):
        with egx.GpHandle(x, y, mean=mean, corr=corr) as h:
            h.finalize(th)
            lone.append((h.fitted_scalars(), h.predict(xq), h.predict_var(xq)))
    hs = egx.GpHandle.create_group(np.stack([s[0] for s in sets]), np.stack([s[1] for s in sets]), mean=mean, corr=corr)
    try:
        assert all(h.schedule() == hs[0].schedule() for h in hs)
        egx.finalize_multi(hs, thetas)
        for h, want in zip(hs, lone):
            assert h.fitted_scalars() == want[0]
            np.testing.assert_array_equal(h.predict(xq), want[1])
            np.testing.assert_array_equal(h.predict_var(xq), want[2])
        # the reduced likelihood alone, members in another order and next to a handle that belongs to no group
        lk_ref = [h.likelihood(th)[0] for h, th in zip(hs, thetas * 1.1)]
        with egx.GpHandle(*sets[0], mean=mean, corr=corr) as free:
            order = [2, 0, 1] if k >= 3 else list(range(k))
            lk, st = egx.likelihood_multi([hs[i] for i in order] + [free], np.concatenate([thetas[order] * 1.1, thetas[:1] * 1.1]))
            assert np.all(st == 0)
            for q, i in enumerate(order):
                assert lk[q] == lk_ref[i]
            assert lk[-1] == lk_ref[0]                 # the same training set, outside the group: the same bits
        with pytest.raises(egx.InvalidValueError, match="distinct"):
            egx.finalize_multi([hs[0], hs[0]], thetas[:2])
        # members go one by one; the others keep working, alone and together
        hs[1].close()
        egx.finalize_multi([hs[0], hs[2]], thetas[[0, 2]])
        assert hs[0].fitted_scalars() == lone[0][0] and hs[2].fitted_scalars() == lone[2][0]
        hs[0].finalize(thetas[0])
        assert hs[0].fitted_scalars() == lone[0][0]
    finally:
        for h in hs:
            h.close()


@pytest.mark.parametrize("n,d,k", [(700, 3, 5), (2100, 4, 3)])
def test_tuned_fits_of_a_group_are_the_lone_tuned_fits_bit_for_bit(egx, n, d, k):
    """egx_gp_fit_multi (round 6): ThetaTuning::Full for the members of a group -- the tuned fit the expert loop runs per cluster
    (crates/moe/src/algorithm.rs:167-177 -> :209-262 -> crates/gp/src/algorithm.rs:921-945) -- with the k x n_starts COBYLA
    machines advanced in lock-step and every round's trial points evaluated as lock-step launch sequences across the models.
    Each member must end exactly where egx_gp_fit on a one-workspace handle of its training set ends: theta*, likelihood,
    evaluations and predictions bit for bit."""
    sets = [_data(n, d, 90 + j) for j in range(k)]
    lo, hi = np.full(d, 0.05), np.full(d, 5.0)
    rng = np.random.default_rng(7)
    starts = 10.0 ** rng.uniform(np.log10(0.1), np.log10(2.0), size=(4, d))
    xq = rng.uniform(size=(50, d))
    lone = []
    for x, y in sets:
        with egx.GpHandle(x, y, n_workspaces=1) as h:
            ne = h.fit(starts, lo, hi, 30)
            lone.append((ne, h.fitted_scalars(), h.inner()["theta"].copy(), h.predict(xq), h.predict_var(xq)))
    hs = egx.GpHandle.create_group(np.stack([s[0] for s in sets]), np.stack([s[1] for s in sets]))
    try:
        nes = egx.fit_multi(hs, np.tile(starts, (k, 1, 1)), lo, hi, 30)
        for h, ne, want in zip(hs, nes, lone):
            assert int(ne) == want[0] and h.fitted_scalars() == want[1]
            np.testing.assert_array_equal(h.inner()["theta"], want[2])
            np.testing.assert_array_equal(h.predict(xq), want[3])
            np.testing.assert_array_equal(h.predict_var(xq), want[4])
        assert len({tuple(w[2]) for w in lone}) == k            # (the members really were tuned apart)
        with pytest.raises(egx.InvalidValueError, match="distinct"):
            egx.fit_multi([hs[0], hs[0]], np.tile(starts, (2, 1, 1)), lo, hi, 30)
    finally:
        for h in hs:
            h.close()
    # ... and through the builder: GpParams.fit_group with ThetaTuning.Full, GpMixture.fit_experts on top of it
    # (`fit` tunes on several workspaces -- another schedule row, 1e-10 per evaluation.  COBYLA run to convergence stops within
    #  its ftol_rel = 1e-4 of the same optimum either way; cut off after 30 evaluations it may stand somewhere else entirely -- one
    #  flipped comparison early on: n = 2100, seed 91 gave 2457.8 on one or two workspaces and 2065.0 on four, and 2538.544 /
    #  2538.553 with 100 evaluations, profiles/r06_tuned_fit_by_workspaces.txt -- hence max_eval = 100 here)
    params = egx.GaussianProcess.params(egx.ConstantMean(), egx.SquaredExponentialCorr()).n_start(3).max_eval(100)
    gps = params.fit_group(np.stack([s[0] for s in sets]), np.stack([s[1] for s in sets]))
    try:
        assert all(g.n_evals > 4 for g in gps) and all(np.isfinite(g.likelihood()) for g in gps)
        ref = [params.fit(*s) for s in sets[:2]]
        for g, r in zip(gps, ref):
            assert g.likelihood() == pytest.approx(r.likelihood(), rel=1e-4)
        for r in ref:
            r.close()
    finally:
        for g in gps:
            g.close()


def test_expert_loop_in_lock_step(egx):
    """GpMixture.fit_experts: clusters of equal size through fit_group, the odd one through fit; predictions of the mixture
    equal those of a mixture of separately fitted experts."""
    from egobox_amd.moe import GaussianMixture, GpMixture
    d = 3
    sizes = [500, 500, 380, 500]
    sets = [_data(n, d, 70 + j) for j, n in enumerate(sizes)]
    params = egx.GaussianProcess.params(egx.ConstantMean(), egx.SquaredExponentialCorr()).theta_tuning(egx.ThetaTuning.Fixed(np.full(d, 0.4)))
    rng = np.random.default_rng(2)
    w = rng.random(4) + 0.5
    gmx = GaussianMixture(w / w.sum(), rng.random((4, d)), np.array([np.eye(d) * 0.3] * 4), 0.9)
    mix = GpMixture.fit_experts(params, [s[0] for s in sets], [s[1] for s in sets], gmx, "smooth")
    ref = GpMixture([params.fit(*s) for s in sets], gmx, "smooth")
    xq = rng.uniform(size=(200, d))
    np.testing.assert_array_equal(mix.predict(xq), ref.predict(xq))
    np.testing.assert_array_equal(mix.predict_var(xq), ref.predict_var(xq))
    for e in mix.experts + ref.experts:
        e.close()


def test_one_shot_groups_are_warm_groups(egx):
    """The expert loop of egobox-moe creates its experts per fit (crates/moe/src/algorithm.rs:167-177 -> params.train).  A
    destroyed GROUP leaves its slabs and its members' workspaces (streams, events, pinned buffers, training-set buffers) in the
    library's pool like a destroyed lone handle does: create_group -> fit -> close cycles on different data of one shape hit the
    pool from the second cycle on, give the bits of the first cycle for the same data, cost little more than the fit on resident
    members, do not grow device memory; egx_trim gives everything back."""
    import torch
    k, n, d = 4, 2100, 3
    th = np.tile(np.full(d, 0.6), (k, 1))

    def sets(seed):
        dd = [_data(n, d, seed + j) for j in range(k)]
        return np.stack([s[0] for s in dd]), np.stack([s[1] for s in dd])

    egx.trim()
    xs, ys = sets(300)
    hs = egx.GpHandle.create_group(xs, ys, corr=1)   # (absolute exponential: well conditioned at any density)
    egx.finalize_multi(hs, th)
    t0 = time.perf_counter()
    for _ in range(3):
        egx.finalize_multi(hs, th)
    resident = (time.perf_counter() - t0) / 3
    ref = [h.fitted_scalars()[0] for h in hs]
    for h in hs:
        h.close()
    stats0 = egx.pool_stats()
    assert stats0["cached_bytes"] >= k * 8 * n * n   # the group's matrices are in the pool, not freed
    free0 = torch.cuda.mem_get_info()[0]
    cycle, used = [], []
    for c in range(8):
        xs, ys = sets(300 + 10 * (c % 2))
        t0 = time.perf_counter()
        hs = egx.GpHandle.create_group(xs, ys, corr=1)
        egx.finalize_multi(hs, th)
        lk = [h.fitted_scalars()[0] for h in hs]
        for h in hs:
            h.close()
        cycle.append(time.perf_counter() - t0)
        used.append(free0 - torch.cuda.mem_get_info()[0])
        if c % 2 == 0:
            assert lk == ref                  # same data, same bits, whatever the slabs held before
    stats = egx.pool_stats()
    assert stats["misses"] == stats0["misses"] and stats["hits"] - stats0["hits"] == 8 * (k + 1)   # k members + the slabs
    print(f"resident group fit {resident * 1e3:.2f} ms, create_group + fit + close cycles {np.median(cycle) * 1e3:.2f} ms (median)")
    assert np.median(cycle) <= 1.3 * resident + 5e-3   # + the members' normalisation, k-major copies and uploads on the host
    assert max(used) - min(used) < 64 << 20
    # a lone handle of the members' shape does not take a member's entry (its slabs are the group's), nor the other way round
    x1, y1 = _data(n, d, 300)
    with egx.GpHandle(x1, y1, corr=1) as h:
        assert egx.pool_stats()["misses"] == stats["misses"] + 1
        h.finalize(th[0])
        assert h.fitted_scalars()[0] == ref[0]   # (n_pad <= 4096 on one workspace: the schedule row of the group's members)
    freed = egx.trim()
    assert freed >= stats["cached_bytes"] > 0 and egx.pool_stats()["cached_bytes"] == 0
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 + k * 8 * n * n - (64 << 20)   # the pooled matrices went back to the device


_BOUNDED_GROUP_POOL_SCRIPT = r"""
import json, sys
import numpy as np
import egobox_amd as egx

def data(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(n, d))
    return x, np.sin(3 * x[:, 0]) + x[:, 1:].sum(axis=1) ** 2 + 0.1 * rng.standard_normal(n)

k, d = 3, 3
th = np.tile(np.full(d, 0.6), (k, 1))
out = {"lk": [], "cached": []}
for n in (1500, 1500, 2300, 1500, 2300):          # 3 x 19 MB and 3 x 44 MB of matrices against a 0.08 GB bound
    sets = [data(n, d, 400 + j) for j in range(k)]
    hs = egx.GpHandle.create_group(np.stack([s[0] for s in sets]), np.stack([s[1] for s in sets]), corr=1)
    egx.finalize_multi(hs, th)
    out["lk"].append([n] + [h.fitted_scalars()[0] for h in hs])
    for h in hs:
        h.close()
    out["cached"].append(egx.pool_stats()["cached_bytes"])
out["stats"] = egx.pool_stats()
print("RESULT " + json.dumps(out))
"""


def test_group_pool_stays_within_its_bound(egx):
    """EGX_POOL_MAX_GB bounds what destroyed handles AND destroyed groups leave behind (read once per process, hence a process of
    its own): with 0.08 GB a group of three n = 1500 models is kept, a group of three n = 2300 models (134 MB of matrices) is freed
    on the spot and pushes nothing else out beyond the bound; results do not depend on what the pool held."""
    import json
    r = _run_script(_BOUNDED_GROUP_POOL_SCRIPT, {"EGX_POOL_MAX_GB": "0.08"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert all(c <= 0.08 * 2 ** 30 for c in out["cached"]), out["cached"]
    assert out["cached"][0] > 50e6              # the first group's slabs (3 x 19 MB) were kept ...
    assert out["stats"]["hits"] >= 4            # ... and adopted by the second group (slabs + three members)
    by_n = {}
    for row in out["lk"]:
        by_n.setdefault(row[0], []).append(row[1:])
    for rows in by_n.values():
        assert all(r_ == rows[0] for r_ in rows)   # the same data, the same bits, whatever was pooled


@pytest.mark.parametrize("n,d", [(300, 3), (1000, 3), (2100, 4), (5000, 4), (9000, 5)])
def test_one_launch_back_substitution_gives_the_bits_of_the_launch_per_block_form(egx, knobs, n, d):
    """gamma = C^-T rho (crates/gp/src/algorithm.rs:1034) as ONE launch (k_trsv_t_fused, round 6: a workgroup per 64 columns of
    the right-hand side, segments handed out from the last one down by a start ticket, the blocks' solutions published through
    agent-scope counters) against the 2 n / 256 launches of rounds 1-5 (egx_set_tuning "trsv_fused" = 0): the same sums in the
    same order -- gamma, hence every prediction, bit for bit; a padded size that is not a multiple of 256 (n = 2100, 5000, 9000:
    a last block of 128 columns) included; and against the oracle's gamma."""
    from oracle import gp_oracle as O
    x, y = _data(n, d, 11 + n)
    theta = np.full(d, 2.0)
    xq = np.random.default_rng(1).uniform(size=(40, d))
    out = {}
    for fused in (1, 0, 1):
        knobs("trsv_fused", fused)
        with egx.GpHandle(x, y, n_workspaces=1) as h:
            h.finalize(theta)
            out.setdefault(fused, []).append((h.inner()["gamma"].copy(), h.predict(xq), h.predict_var(xq)))
    for got in out[1]:
        for a, b in zip(got, out[0][0]):
            np.testing.assert_array_equal(a, b)
    if n <= 2100:
        want = np.asarray(O.fit_fixed(x, y, theta).inner.gamma).ravel()
        np.testing.assert_allclose(out[1][0][0].ravel(), want, rtol=1e-6, atol=1e-7 * np.abs(want).max())


def test_one_launch_back_substitution_of_a_group_and_under_load(egx, knobs):
    """... for the members of a group (blockIdx.y = model: egx_gp_finalize_multi) and with other launches keeping the chip busy
    (the segments' start tickets, not the dispatch order, decide who waits for whom): the bits of the launch-per-block form."""
    k, n, d = 5, 2100, 3
    sets = [_data(n, d, 40 + j) for j in range(k)]
    thetas = np.full((k, d), 2.0) * (1.0 + 0.1 * np.arange(k))[:, None]
    res = {}
    for fused in (1, 0):
        knobs("trsv_fused", fused)
        hs = egx.GpHandle.create_group(np.stack([s[0] for s in sets]), np.stack([s[1] for s in sets]))
        try:
            egx.finalize_multi(hs, thetas)
            res[fused] = [h.inner()["gamma"].copy() for h in hs]
        finally:
            for h in hs:
                h.close()
    for a, b in zip(res[1], res[0]):
        np.testing.assert_array_equal(a, b)
    # under load: a twelve-workspace handle evaluates a batch on its own streams while another handle finalizes, ten times over
    knobs("trsv_fused", 1)
    xb, yb = _data(4096, 4, 5)
    x1, y1 = sets[0]
    import threading
    with egx.GpHandle(xb, yb, n_workspaces=12) as hb, egx.GpHandle(x1, y1, n_workspaces=1) as h1:
        cands = np.full((24, 4), 2.0) * (1.0 + 0.01 * np.arange(24))[:, None]
        stop = threading.Event()

        def load():
            while not stop.is_set():
                hb.likelihood_batch(cands)
        t = threading.Thread(target=load)
        t.start()
        try:
            for _ in range(10):
                h1.finalize(thetas[0])
                np.testing.assert_array_equal(h1.inner()["gamma"], res[0][0])
        finally:
            stop.set()
            t.join()


_TRSV_STALL_SCRIPT = r"""
import json, time
import numpy as np
import egobox_amd as egx
rng = np.random.default_rng(5)
x = rng.uniform(size=(2100, 3))
y = np.sin(3 * x[:, 0]) + x[:, 1:].sum(axis=1) ** 2 + 0.1 * rng.standard_normal(2100)
th = np.full(3, 2.0)
xq = rng.uniform(size=(20, 3))
out = {}
with egx.GpHandle(x, y) as h:
    h.finalize(th)
    good = (h.inner()["gamma"].copy(), h.predict(xq))
    egx.set_tuning("pipe_timeout_ms", 25)
    egx.set_tuning("trsv_stall", 1)                    # the last segment of the one-launch back-substitution never publishes
    s0 = egx.chain_stats()
    t0 = time.perf_counter()
    h.finalize(th)                                     # ... the fit still succeeds: launch per block, once more
    out["fallback_seconds"] = time.perf_counter() - t0
    s1 = egx.chain_stats()
    out["counted"] = [s1["aborted"] - s0["aborted"], s1["retried"] - s0["retried"]]
    out["same_gamma"] = bool(np.array_equal(h.inner()["gamma"], good[0])) and bool(np.array_equal(h.predict(xq), good[1]))
    egx.set_tuning("pipe_retry", 0)
    t0 = time.perf_counter()
    try:
        h.finalize(th)
        out["error"] = None
    except egx.EgxError as e:
        out["error"] = str(e)
    out["error_seconds"] = time.perf_counter() - t0
    egx.set_tuning("trsv_stall", 0)
    egx.set_tuning("pipe_retry", 1)
    egx.set_tuning("pipe_timeout_ms", 2000)
    h.finalize(th)
    out["again"] = bool(np.array_equal(h.inner()["gamma"], good[0]))
print("RESULT " + json.dumps(out))
"""


def test_a_back_substitution_whose_hand_off_never_arrives_is_run_again_launch_per_block(egx):
    """The one-launch back-substitution bounds its waits like the chain launch: a segment that never publishes (forced: the TEST
    build's "trsv_stall") leaves NaN in gamma within the bound, finalize runs the launch-per-block form from the untouched
    right-hand side -- the same gamma, the same predictions, counted by egx_chain_stats -- and only with the fallback off it is
    EGX_ERR_HIP; never a hang, never a wrong model; the next fit on the handle is clean."""
    import json
    r = _run_script(_TRSV_STALL_SCRIPT, {"EGX_TEST_LIBRARY": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert out["counted"] == [1, 1] and out["same_gamma"] and out["fallback_seconds"] < 5.0
    assert out["error"] is not None and "back-substitution" in out["error"] and out["error_seconds"] < 5.0
    assert out["again"]
