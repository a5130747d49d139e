"""TEST INFRASTRUCTURE ONLY: ctypes loader of oracle/lib/libarbiter_ld.so (oracle/arbiter_ld.c), the x87 long double
end-to-end likelihood used as the ARBITER where two double-precision evaluations disagree (ill-conditioned R).
Imported by tests/ and tests/golden/make_arbiter.py only."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "lib", "libarbiter_ld.so")
CORR = {"SquaredExponential": 0, "AbsoluteExponential": 1, "Matern32": 2, "Matern52": 3}
_lib = None


def available():
    return os.path.exists(LIB)


def likelihood(x, y, theta, corr="SquaredExponential", nugget=100.0 * np.finfo(float).eps):
    """-> dict(status, likelihood, likelihood_low_part, sigma2, beta, min_pivot); status 1 = not positive definite even
    in extended precision."""
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        _lib.arbiter_ld_likelihood.restype = C.c_int
        _lib.arbiter_ld_likelihood.argtypes = [dp, dp, C.c_int64, C.c_int64, dp, C.c_int, C.c_double, dp]
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    y = np.ascontiguousarray(y, dtype=np.float64).ravel()
    n, d = x.shape
    theta = np.ascontiguousarray(np.broadcast_to(np.asarray(theta, dtype=np.float64), (d,)))
    out = np.zeros(5)
    dp = C.POINTER(C.c_double)
    rc = _lib.arbiter_ld_likelihood(x.ctypes.data_as(dp), y.ctypes.data_as(dp), n, d, theta.ctypes.data_as(dp),
                                    CORR[corr] if isinstance(corr, str) else int(corr), float(nugget),
                                    out.ctypes.data_as(dp))
    if rc < 0:
        raise RuntimeError(f"arbiter_ld_likelihood failed: {rc}")
    return {"status": rc, "likelihood": out[0], "likelihood_low_part": out[4], "sigma2": out[1], "beta": out[2],
            "min_pivot": out[3]}
