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


class Sweep:
    """`egx_sweep*` of include/egx_gp.h: this rank's replica of the training set + the RCCL communicator.

    The collective lives INSIDE libegx_gp_hip.so (ncclAllGather on the library's own stream); this class only
    moves the 128-byte unique id between the ranks.  `id_bytes`: None builds no communicator (world must be 1),
    "new" asks the library for a fresh id (rank 0), bytes = the id received from rank 0.
    """

    def __init__(self, x, y, mean=0, corr=0, nugget=None, device=-1, rank=0, world=1, id_bytes=None, n_workspaces=2):
        import ctypes as C
        from . import _lib as L
        lib = L.load()
        self._lib, self._C, self._L = lib, C, L
        x = L.as_f64(x, 2)
        y = L.as_f64(np.asarray(y).reshape(-1), 1)
        cfg = L.GpConfig()
        lib.egx_gp_config_default(C.byref(cfg))
        cfg.corr, cfg.mean, cfg.device, cfg.n_workspaces = int(corr), int(mean), int(device), int(n_workspaces)
        if nugget is not None:
            cfg.nugget = float(nugget)
        if isinstance(id_bytes, str) and id_bytes == "new":
            id_bytes = self.unique_id()
        self.id_bytes = id_bytes
        idbuf = None
        if id_bytes is not None:
            if len(id_bytes) != 128:
                raise L.InvalidValueError(L.ERR_INVALID_VALUE, "the RCCL unique id has 128 bytes")
            idbuf = C.create_string_buffer(bytes(id_bytes), 128)
        self._h = C.c_void_p()
        L.check(lib.egx_sweep_create(C.byref(cfg), L.dptr(x), L.dptr(y), x.shape[0], x.shape[1],
                                     C.cast(idbuf, C.c_void_p) if idbuf is not None else None, int(rank), int(world),
                                     C.byref(self._h)))
        self.rank, self.world, self.d = int(rank), int(world), x.shape[1]

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib as L
        buf = C.create_string_buffer(128)
        L.check(L.load().egx_sweep_unique_id(C.cast(buf, C.c_void_p)))
        return buf.raw

    def info(self):
        C = self._C
        r, w, rr, v, na = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        self._L.check(self._lib.egx_sweep_info(self._h, C.byref(r), C.byref(w), C.byref(rr), C.byref(v), C.byref(na)))
        return {"rank": r.value, "world": w.value, "rccl_ranks": rr.value, "rccl_version": v.value,
                "n_allgathers": na.value}

    def likelihood(self, thetas, raise_on_peer_failure=True):
        """COLLECTIVE: (k x h) candidates -> (lkh (k,), status (k,)), complete on every rank.
        When ANOTHER rank failed the call raises `PeerError` -- or, with raise_on_peer_failure=False, returns the arrays
        anyway: the survivors' candidates are valid, the failed rank's carry STATUS_RANK_FAILED."""
        L = self._L
        thetas = L.as_f64(thetas, 2)
        k = thetas.shape[0]
        lk = np.empty(k)
        st = np.empty(k, dtype=np.int32)
        rc = self._lib.egx_sweep_likelihood(self._h, L.dptr(thetas), k, thetas.shape[1], L.dptr(lk),
                                            st.ctypes.data_as(L.c_int32_p))
        if not (rc == L.ERR_PEER and not raise_on_peer_failure):
            L.check(rc)
        return lk, st

    def local_likelihood_batch(self, thetas):
        """NOT a collective: `egx_gp_likelihood_batch` on this rank's handle (the evaluation a rank does inside
        `likelihood`, without the all-gather) -> (lkh (k,), status (k,))."""
        L = self._L
        thetas = L.as_f64(thetas, 2)
        k = thetas.shape[0]
        lk = np.empty(k)
        st = np.empty(k, dtype=np.int32)
        L.check(self._lib.egx_gp_likelihood_batch(self._lib.egx_sweep_handle(self._h), L.dptr(thetas), k, thetas.shape[1],
                                                  L.dptr(lk), st.ctypes.data_as(L.c_int32_p)))
        return lk, st

    def fit(self, theta0s, lo, hi, max_eval=1000):  # GP_COBYLA_MAX_EVAL, crates/gp/src/lib.rs
        """COLLECTIVE tuned fit (egx_sweep_fit): the multistart COBYLA runs of `GpHandle.fit` with start s on rank
        s mod world, one all-gather of the starts' results, every rank's replica finalized at the winner -- the same bits
        as the one-GPU fit.  Returns the evaluations of all starts; `model()` is the fitted replica."""
        L, C = self._L, self._C
        theta0s = L.as_f64(theta0s, 2)
        lo = L.as_f64(np.atleast_1d(lo), 1)
        hi = L.as_f64(np.atleast_1d(hi), 1)
        ne = C.c_int64()
        L.check(self._lib.egx_sweep_fit(self._h, L.dptr(theta0s), theta0s.shape[0], L.dptr(lo), L.dptr(hi), lo.size,
                                        int(max_eval), C.byref(ne)))
        return ne.value

    def model(self):
        """This rank's replica (egx_sweep_handle) as a `GpHandle` view: predict / fitted_scalars / inner on the model a
        `fit` or the caller's `finalize` left resident.  Owned by the sweep."""
        from .gp import GpHandle
        return GpHandle._borrow(self._lib.egx_sweep_handle(self._h), self)

    def set_lockstep(self, width):
        """Lock-step width of this rank's likelihood batches (egx_gp_set_lockstep on the sweep's handle)."""
        h = self._lib.egx_sweep_handle(self._h)
        self._L.check(self._lib.egx_gp_set_lockstep(h, int(width)))
        return self._lib.egx_gp_get_lockstep(h)

    def set_assignment(self, dynamic):
        """0 / False: candidate c -> rank c mod world (default); 1 / True: ranks pull candidates from a node-wide counter
        as their workspaces free up.  Every rank must select the same mode."""
        self._L.check(self._lib.egx_sweep_set_assignment(self._h, 1 if dynamic else 0))

    def last_balance(self):
        """(candidates evaluated by every rank in the last likelihood() call, this rank's evaluation seconds)."""
        C = self._C
        per = np.zeros(self.world, dtype=np.int64)
        sec = C.c_double()
        self._L.check(self._lib.egx_sweep_last_balance(self._h, per.ctypes.data_as(self._L.c_int64_p), C.byref(sec)))
        return per, sec.value

    def allgather(self, v):
        """COLLECTIVE: (count,) doubles per rank -> (world, count)."""
        L = self._L
        v = L.as_f64(np.asarray(v).reshape(-1), 1)
        out = np.empty((self.world, v.size))
        L.check(self._lib.egx_sweep_allgather(self._h, L.dptr(v), v.size, L.dptr(out)))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.egx_sweep_destroy(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def rendezvous_sweep(x, y, device, **kw):
    """One `Sweep` per rank of the default torch.distributed group: rank 0 draws the RCCL unique id, the group's
    store-backed object broadcast carries its 128 bytes to the other ranks (the only thing torch does here)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return Sweep(x, y, device=device, rank=0, world=1, id_bytes="new", **kw)
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [Sweep.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return Sweep(x, y, device=device, rank=rank, world=world, id_bytes=box[0], **kw)


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
