"""ctypes binding of libegx_gp_hip.so (include/egx_gp.h).

This is the same C ABI a Rust `extern "C"` shim would bind.  The library is
built in-tree by `__graft_entry__.build()` / `make -C egobox_amd/csrc`; if it is missing the
import fails loudly -- there is no Python or CPU fallback for the compute path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libegx_gp_hip.so")
#: the same library built with -DEGX_TEST_HOOKS (the "pipe_stall" knob, grid / trace overrides of the chain launch): loaded
#: instead of the product library only when EGX_TEST_LIBRARY=1 is set -- by tests that force a hand-off to fail, in a
#: process of their own.  Never the default, never used by bench.py.
TEST_LIB_PATH = os.path.join(_HERE, "lib", "_dev", "libegx_gp_hip_testhooks.so")
#: EGX_TEST_LIBRARY=trace: the profiling build with per-tile stamps in the left-looking group updates (tools/dev_build.sh trace;
#: tools/long_update_attribution.py) -- built by hand, never by build(), never the default
TRACE_LIB_PATH = os.path.join(_HERE, "lib", "_dev", "libegx_gp_hip_trace.so")

# return codes / status values (egx_rc, egx_status)
SUCCESS, ERR_INVALID_VALUE, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_FITTED, ERR_LINALG, ERR_LIKELIHOOD, ERR_UNSUPPORTED, ERR_PEER = range(9)
STATUS_OK, STATUS_NOT_POSITIVE_DEFINITE, STATUS_ILL_CONDITIONED_FT, STATUS_ILL_CONDITIONED_F, STATUS_NAN_THETA, STATUS_RANK_FAILED = range(6)

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class GpConfig(C.Structure):
    _fields_ = [("corr", C.c_int32), ("mean", C.c_int32), ("nugget", C.c_double), ("device", C.c_int32),
                ("n_workspaces", C.c_int32), ("w_star", c_double_p), ("kpls_dim", C.c_int64)]


class InnerView(C.Structure):
    _fields_ = [(k, c_double_p) for k in (
        "theta", "likelihood", "sigma2", "beta", "gamma", "r_chol", "ft", "ft_qr_r", "x_mean", "x_std",
        "y_mean", "y_std", "xt_norm", "yt_norm")]


class SgpConfig(C.Structure):
    _fields_ = [("corr", C.c_int32), ("method", C.c_int32), ("nugget", C.c_double), ("device", C.c_int32)]


class Timings(C.Structure):
    _fields_ = [("corr_build_ms", C.c_double), ("potrf_ms", C.c_double), ("potrf_syrk_ms", C.c_double),
                ("solve_ms", C.c_double), ("host_ms", C.c_double), ("total_ms", C.c_double),
                ("potrf_flops", C.c_int64), ("corr_bytes", C.c_int64), ("syrk_launches", C.c_int64),
                ("syrk_flops", C.c_int64)]


#: every symbol include/egx_gp.h declares: (name, restype, argtypes)
SIGNATURES = [
    ("egx_abi_version", C.c_int32, []),
    ("egx_last_error", C.c_char_p, []),
    ("egx_device_count", C.c_int32, []),
    ("egx_gp_config_default", None, [C.POINTER(GpConfig)]),
    ("egx_trim", C.c_int64, []),
    ("egx_set_tuning", C.c_int32, [C.c_char_p, C.c_int32, c_int32_p]),
    ("egx_pool_stats", None, [c_int64_p, c_int64_p, c_int64_p]),
    ("egx_chain_stats", None, [c_int64_p, c_int64_p]),
    ("egx_normalize", C.c_int32, [c_double_p, C.c_int64, C.c_int64, c_double_p, c_double_p, c_double_p]),
    ("egx_regression_ncols", C.c_int64, [C.c_int32, C.c_int64]),
    ("egx_regression_basis", C.c_int32, [C.c_int32, c_double_p, C.c_int64, C.c_int64, c_double_p]),
    ("egx_gp_create", C.c_int32, [C.POINTER(GpConfig), c_double_p, c_double_p, C.c_int64, C.c_int64,
                                  C.POINTER(C.c_void_p)]),
    ("egx_gp_destroy", None, [C.c_void_p]),
    ("egx_gp_dims", C.c_int32, [C.c_void_p, c_int64_p, c_int64_p, c_int64_p, c_int64_p]),
    ("egx_gp_get_training_data", C.c_int32, [C.c_void_p, c_double_p, c_double_p]),
    ("egx_gp_likelihood", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p, c_int32_p]),
    ("egx_gp_likelihood_batch", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, C.c_int64, c_double_p, c_int32_p]),
    ("egx_gp_set_lockstep", C.c_int32, [C.c_void_p, C.c_int32]),
    ("egx_gp_get_lockstep", C.c_int32, [C.c_void_p]),
    ("egx_gp_get_schedule", C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]),
    ("egx_gp_shrink", C.c_int32, [C.c_void_p, C.c_int32]),
    ("egx_gp_likelihood_grad", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p, c_double_p, c_int32_p]),
    ("egx_gp_likelihood_grad_batch", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, C.c_int64, c_double_p, c_double_p,
                                                 c_int32_p]),
    ("egx_gp_finalize", C.c_int32, [C.c_void_p, c_double_p, C.c_int64]),
    ("egx_gp_create_group", C.c_int32, [C.c_void_p, c_double_p, c_double_p, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]),
    ("egx_gp_finalize_multi", C.c_int32, [C.POINTER(C.c_void_p), C.c_int32, c_double_p, C.c_int64]),
    ("egx_gp_likelihood_multi", C.c_int32, [C.POINTER(C.c_void_p), C.c_int32, c_double_p, C.c_int64, c_double_p, C.POINTER(C.c_int32)]),
    ("egx_gp_fit_multi", C.c_int32, [C.POINTER(C.c_void_p), C.c_int32, c_double_p, C.c_int64, c_double_p, c_double_p, C.c_int64, C.c_int64,
                                     c_int64_p]),
    ("egx_gp_fit", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p, c_double_p, C.c_int64, C.c_int64,
                               c_int64_p]),
    ("egx_gp_fit_partial", C.c_int32, [C.c_void_p, c_double_p, c_int64_p, C.c_int64, c_double_p, C.c_int64, c_double_p,
                                       c_double_p, C.c_int64, C.c_int64, c_int64_p]),
    ("egx_gp_fit_lbfgs", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p, c_double_p, C.c_int64, C.c_int64,
                                     c_int64_p]),
    ("egx_gp_predict", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p]),
    ("egx_gp_predict_var", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p]),
    ("egx_gp_predict_valvar", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p, c_double_p]),
    ("egx_gp_predict_gradients", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p]),
    ("egx_gp_predict_var_gradients", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p]),
    ("egx_gp_predict_valvar_gradients", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p, c_double_p]),
    ("egx_gp_get_inner", C.c_int32, [C.c_void_p, C.POINTER(InnerView)]),
    ("egx_gp_set_inner", C.c_int32, [C.c_void_p, C.POINTER(InnerView)]),
    ("egx_corr_matrix", C.c_int32, [C.c_int32, c_double_p, C.c_int64, C.c_int64, c_double_p, C.c_double, c_double_p]),
    ("egx_cross_corr", C.c_int32, [C.c_int32, c_double_p, C.c_int64, c_double_p, C.c_int64, C.c_int64, c_double_p,
                                   c_double_p]),
    ("egx_potrf", C.c_int32, [c_double_p, C.c_int64, c_int32_p]),
    ("egx_sweep_unique_id", C.c_int32, [C.c_void_p]),
    ("egx_sweep_create", C.c_int32, [C.POINTER(GpConfig), c_double_p, c_double_p, C.c_int64, C.c_int64, C.c_void_p,
                                     C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ("egx_sweep_destroy", None, [C.c_void_p]),
    ("egx_sweep_handle", C.c_void_p, [C.c_void_p]),
    ("egx_sweep_info", C.c_int32, [C.c_void_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int64_p]),
    ("egx_sweep_likelihood", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, C.c_int64, c_double_p, c_int32_p]),
    ("egx_sweep_set_assignment", C.c_int32, [C.c_void_p, C.c_int32]),
    ("egx_sweep_last_balance", C.c_int32, [C.c_void_p, c_int64_p, c_double_p]),
    ("egx_sweep_allgather", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p]),
    ("egx_sweep_fit", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p, c_double_p, C.c_int64, C.c_int64, c_int64_p]),
    ("egx_moe_predict_valvar", C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), c_int32_p, C.c_int64, C.c_int64, c_double_p,
                                           c_double_p, C.c_int64, C.c_int64, C.c_int32, c_double_p, c_double_p]),
    ("egx_gmx_precisions_chol", C.c_int32, [c_double_p, C.c_int64, C.c_int64, c_double_p]),
    ("egx_gmx_predict_probas", C.c_int32, [C.c_int32, c_double_p, c_double_p, c_double_p, C.c_int64, C.c_int64, C.c_double,
                                           c_double_p, C.c_int64, c_double_p]),
    ("egx_gmx_predict_probas_derivatives", C.c_int32, [C.c_int32, c_double_p, c_double_p, c_double_p, C.c_int64, C.c_int64,
                                                       C.c_double, c_double_p, C.c_int64, c_double_p]),
    ("egx_moe_predict_valvar_gradients", C.c_int32, [C.c_void_p, C.POINTER(C.c_void_p), c_int32_p, C.c_int64, C.c_int64,
                                                     c_double_p, c_double_p, c_double_p, C.c_int64, C.c_int64, C.c_int32,
                                                     c_double_p, c_double_p]),
    ("egx_gp_last_timings", C.c_int32, [C.c_void_p, C.POINTER(Timings)]),
    ("egx_mfma_probe", C.c_int32, [c_double_p]),
    ("egx_sgp_config_default", None, [C.POINTER(SgpConfig)]),
    ("egx_sgp_create", C.c_int32, [C.POINTER(SgpConfig), c_double_p, c_double_p, C.c_int64, C.c_int64, c_double_p,
                                   C.c_int64, C.POINTER(C.c_void_p)]),
    ("egx_sgp_destroy", None, [C.c_void_p]),
    ("egx_sgp_dims", C.c_int32, [C.c_void_p, c_int64_p, c_int64_p, c_int64_p]),
    ("egx_sgp_likelihood", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, C.c_double, C.c_double, c_double_p, c_int32_p]),
    ("egx_sgp_finalize", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, C.c_double, C.c_double]),
    ("egx_sgp_fit", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p, c_double_p, C.c_int32, C.c_double,
                                C.c_int64, c_int64_p]),
    ("egx_sgp_predict", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p]),
    ("egx_sgp_predict_var", C.c_int32, [C.c_void_p, c_double_p, C.c_int64, c_double_p]),
    ("egx_sgp_get_state", C.c_int32, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
]


class EgxError(RuntimeError):
    """Base error; `.rc` is the egx_rc.  Subclasses mirror GpError (crates/gp/src/errors.rs:8-40)."""

    def __init__(self, rc, msg):
        super().__init__(msg)
        self.rc = rc


class InvalidValueError(EgxError, ValueError):
    pass


class NoDeviceError(EgxError):
    pass


class NotFittedError(EgxError):
    pass


class LinalgError(EgxError):
    pass


class LikelihoodComputationError(EgxError):
    pass


class PeerError(EgxError):
    """A collective call in which another rank failed or did not answer (EGX_ERR_PEER)."""


_ERR = {ERR_INVALID_VALUE: InvalidValueError, ERR_NO_DEVICE: NoDeviceError, ERR_NOT_FITTED: NotFittedError,
        ERR_LINALG: LinalgError, ERR_LIKELIHOOD: LikelihoodComputationError, ERR_PEER: PeerError}

_lib = None


def load():
    """Load (once) and type the shared library.  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    which = os.environ.get("EGX_TEST_LIBRARY")
    if which == "1":
        path = TEST_LIB_PATH
    elif which == "trace":
        path = TRACE_LIB_PATH
    elif which:  # any other scratch build of tools/dev_build.sh <name> <defines>: A/B measurements (tools/ab_lib.py)
        path = os.path.join(_HERE, "lib", "_dev", f"libegx_gp_hip_{which}.so")
    else:
        path = LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C egobox_amd/csrc`).  egobox_amd has no fallback compute path.")
    lib = C.CDLL(path)
    for name, res, args in SIGNATURES:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != SUCCESS:
        msg = load().egx_last_error().decode("utf-8", "replace")
        raise _ERR.get(rc, EgxError)(rc, msg)


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def as_f64(a, ndim=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if ndim is not None and a.ndim != ndim:
        raise InvalidValueError(ERR_INVALID_VALUE, f"expected a {ndim}-d array, got shape {a.shape}")
    return a
