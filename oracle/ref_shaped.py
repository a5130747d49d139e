"""TEST INFRASTRUCTURE ONLY: ctypes loader of oracle/lib/libref_shaped.so (oracle/ref_shaped.c), the single-thread
"reference default build"-shaped restatement.  Imported by tests/ and bench.py's cpu_baseline leg only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "lib", "libref_shaped.so")
CORR = {"squared_exponential": 0, "absolute_exponential": 1, "matern32": 2, "matern52": 3}
_lib = None


def available():
    return os.path.exists(LIB)


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        _lib.ref_shaped_likelihood.restype = C.c_int
        _lib.ref_shaped_likelihood.argtypes = [dp, dp, C.c_int64, C.c_int64, dp, C.c_int, C.c_double, dp, dp, dp]
    return _lib


def likelihood(x, y, theta, corr="squared_exponential", nugget=100.0 * np.finfo(float).eps, want_factor=False):
    """-> dict(status, likelihood, sigma2, beta, min_pivot[, r_chol, gamma]); status 1 = not positive definite."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    y = np.ascontiguousarray(y, dtype=np.float64).ravel()
    n, d = x.shape
    theta = np.ascontiguousarray(np.broadcast_to(np.asarray(theta, dtype=np.float64), (d,)))
    out = np.zeros(4)
    rch = np.empty((n, n)) if want_factor else None
    gam = np.empty(n) if want_factor else None
    dp = C.POINTER(C.c_double)
    p = lambda a: a.ctypes.data_as(dp) if a is not None else None  # noqa: E731
    rc = _load().ref_shaped_likelihood(p(x), p(y), n, d, p(theta), CORR[corr] if isinstance(corr, str) else int(corr),
                                       float(nugget), p(out), p(rch), p(gam))
    if rc < 0:
        raise RuntimeError(f"ref_shaped_likelihood failed: {rc}")
    res = {"status": rc, "likelihood": out[0], "sigma2": out[1], "beta": out[2], "min_pivot": out[3]}
    if want_factor and rc == 0:
        res["r_chol"], res["gamma"] = rch, gam
    return res
