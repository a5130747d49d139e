"""Theta-sweep / multistart sharding across the GPUs of one node (BASELINE config 4).

The reference runs its multistart likelihood evaluations as independent rayon tasks
(crates/gp/src/algorithm.rs:928-945); evaluations at different theta share nothing but the (replicated,
4 MiB) training set.  Here: one process per GPU, candidate k goes to rank k mod G, every rank evaluates
its shard on its own device, and ONE all-gather of (likelihood f64, status) over RCCL/xGMI assembles the
result everywhere.  Payload is 16 B per candidate (8 KB for 512): latency class, never link bound.
"""
from __future__ import annotations

import numpy as np


def shard_indices(k, rank, world):
    """Static round-robin partition: candidates rank, rank+world, ..."""
    return np.arange(rank, k, world)


def sweep_likelihood(evaluate, thetas, rank=None, world=None, device=None):
    """Evaluate `thetas` (k x h) sharded over the ranks of the default process group.

    evaluate(thetas_shard) -> (lkh (ks,), status (ks,)) runs on this rank's device
    (normally `GpHandle.likelihood_batch`).  Returns (lkh (k,), status (k,)) on every rank.
    `device`: torch device of the collective payload ("cuda:<local_rank>" under RCCL, None for gloo).
    """
    import torch
    import torch.distributed as dist
    thetas = np.ascontiguousarray(thetas, dtype=np.float64)
    k = thetas.shape[0]
    distributed = dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank() if distributed else 0
    if world is None:
        world = dist.get_world_size() if distributed else 1
    mine = shard_indices(k, rank, world)
    if mine.size:
        lk, st = evaluate(thetas[mine])
    else:
        lk, st = np.empty(0), np.empty(0, dtype=np.int32)
    if world == 1:
        return np.asarray(lk, dtype=np.float64), np.asarray(st, dtype=np.int32)
    # fixed-size payload per rank: ceil(k / world) x {lkh, status}
    per = (k + world - 1) // world
    buf = torch.full((per, 2), float("nan"), dtype=torch.float64)
    buf[:mine.size, 0] = torch.from_numpy(np.asarray(lk, dtype=np.float64))
    buf[:mine.size, 1] = torch.from_numpy(np.asarray(st, dtype=np.float64))
    if device is not None:
        buf = buf.to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)  # ncclAllGather over xGMI on the GPU box; gloo in the CPU tests
    lkh = np.empty(k)
    status = np.empty(k, dtype=np.int32)
    for r in range(world):
        idx = shard_indices(k, r, world)
        pr = parts[r].cpu().numpy()
        lkh[idx] = pr[:idx.size, 0]
        status[idx] = pr[:idx.size, 1].astype(np.int32)
    return lkh, status


def best_candidate(lkh, status):
    """arg-max of the likelihood over candidates that evaluated cleanly (the reduce of algorithm.rs:942-945)."""
    lkh = np.asarray(lkh)
    ok = (np.asarray(status) == 0) & np.isfinite(lkh)
    if not ok.any():
        return -1
    return int(np.flatnonzero(ok)[np.argmax(lkh[ok])])


def expert_to_rank(n_experts, world):
    """MoE config 5: expert e -> rank e mod G (crates/moe/src/algorithm.rs:167-177 trains them serially)."""
    return [e % world for e in range(n_experts)]
