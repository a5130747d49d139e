"""The N>1 path on CPU: world_size-2 gloo run of the theta-sweep sharding + all-gather, with the CPU oracle
standing in for the per-rank GPU evaluator (the oracle is only the checker here)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from egobox_amd.sweep import sweep_likelihood, best_candidate
    from egobox_amd.multistart import theta_sweep_candidates
    from oracle import gp_oracle as O
    rng = np.random.default_rng(0)
    x = rng.random((40, 2))
    y = np.sin(4 * x[:, 0]) + x[:, 1]
    thetas = theta_sweep_candidates(7, 2, seed=5)
    thetas[3, 0] = np.nan  # exercises the status channel through the collective
    calls = []

    def evaluate(th):
        calls.append(len(th))
        out = [O.likelihood_at(x, y, t) for t in th]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out], dtype=np.int32)

    lk, st = sweep_likelihood(evaluate, thetas)
    q.put((rank, lk, st, calls, best_candidate(lk, st)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sweep_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    (_, lk0, st0, calls0, b0), (_, lk1, st1, calls1, b1) = res
    np.testing.assert_array_equal(lk0, lk1)       # every rank holds the full result
    np.testing.assert_array_equal(st0, st1)
    assert calls0 == [4] and calls1 == [3]        # 7 candidates: ranks got 4 and 3 (k mod G)
    assert st0[3] == 4 and lk0[3] == np.inf       # NaN theta -> status 4 (oracle convention: +inf objective)
    assert b0 == b1 and b0 >= 0
    # reference values computed serially
    from egobox_amd.multistart import theta_sweep_candidates
    from oracle import gp_oracle as O
    rng = np.random.default_rng(0)
    x = rng.random((40, 2))
    y = np.sin(4 * x[:, 0]) + x[:, 1]
    thetas = theta_sweep_candidates(7, 2, seed=5)
    for i in (0, 1, 2, 4, 5, 6):
        assert lk0[i] == O.likelihood_at(x, y, thetas[i])[0]


def test_shard_indices_cover_everything():
    from egobox_amd.sweep import shard_indices, expert_to_rank
    for k in (0, 1, 7, 512):
        for g in (1, 2, 4, 8):
            allidx = np.concatenate([shard_indices(k, r, g) for r in range(g)])
            assert sorted(allidx.tolist()) == list(range(k))
    assert expert_to_rank(8, 8) == list(range(8))
    assert expert_to_rank(8, 4) == [0, 1, 2, 3, 0, 1, 2, 3]


def test_sweep_single_process():
    from egobox_amd.sweep import sweep_likelihood
    th = np.arange(6.0).reshape(3, 2)
    lk, st = sweep_likelihood(lambda t: (t.sum(1), np.zeros(len(t), dtype=np.int32)), th)
    np.testing.assert_array_equal(lk, [1.0, 5.0, 9.0])


def _moe_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from egobox_amd.moe import GaussianMixture, GpMixture
    from oracle import gp_oracle as O
    rng = np.random.default_rng(1)
    experts = []
    for c in range(3):
        x = rng.random((30, 2)) + [c, 0.0]
        y = np.sin(3 * x[:, 0]) + x[:, 1] * (c + 1)
        # expert e lives on rank e mod G (BASELINE config 5); the others are absent on this rank
        experts.append(O.fit_fixed(x, y, [1.5, 1.0], corr=O.MATERN52) if c % world == rank else None)
    gmx = GaussianMixture([0.3, 0.3, 0.4], [[0.5, 0.5], [1.5, 0.5], [2.5, 0.5]], [np.eye(2) * 0.2] * 3, 0.8)
    xq = np.random.default_rng(2).random((25, 2)) * [3.0, 1.0]
    out = {}
    for recomb in ("smooth", "hard"):
        mix = GpMixture(experts, gmx, recomb, rank=rank, world=world)
        out[recomb] = mix.predict_valvar(xq)
        out[recomb + "_grad"] = mix.predict_valvar_gradients(xq[:9])  # (m, nx) matrices through the same all-reduce
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_moe_experts_sharded_over_two_ranks_gloo():
    """Config-5 shape on CPU: experts sharded e mod G over 2 gloo ranks, one all-reduce of the weighted vectors;
    every rank ends with the single-process result."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_moe_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from egobox_amd.moe import GaussianMixture, GpMixture
    from oracle import gp_oracle as O
    rng = np.random.default_rng(1)
    experts = []
    for c in range(3):
        x = rng.random((30, 2)) + [c, 0.0]
        y = np.sin(3 * x[:, 0]) + x[:, 1] * (c + 1)
        experts.append(O.fit_fixed(x, y, [1.5, 1.0], corr=O.MATERN52))
    gmx = GaussianMixture([0.3, 0.3, 0.4], [[0.5, 0.5], [1.5, 0.5], [2.5, 0.5]], [np.eye(2) * 0.2] * 3, 0.8)
    xq = np.random.default_rng(2).random((25, 2)) * [3.0, 1.0]
    for recomb in ("smooth", "hard"):
        want = GpMixture(experts, gmx, recomb).predict_valvar(xq)
        want_g = GpMixture(experts, gmx, recomb).predict_valvar_gradients(xq[:9])
        for r in range(2):
            np.testing.assert_allclose(res[r][recomb][0], want[0], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(res[r][recomb][1], want[1], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(res[r][recomb + "_grad"][0], want_g[0], rtol=1e-11, atol=1e-12)
            np.testing.assert_allclose(res[r][recomb + "_grad"][1], want_g[1], rtol=1e-11, atol=1e-12)


def test_library_sweep_partition_arithmetic(tmp_path):
    """The C++ sharding / payload layout of egx_sweep_likelihood (csrc/sweep_shard.h) for simulated worlds of 1-8 ranks
    and ragged candidate counts, under ASan + UBSan (tests/c_host/sweep_shard_test.cpp)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "sweep_shard_test"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-fsanitize=address,undefined",
                    os.path.join(root, "tests", "c_host", "sweep_shard_test.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK"), (out.stdout, out.stderr)
