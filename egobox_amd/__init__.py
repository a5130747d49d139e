"""egobox_amd -- MI355X-native (gfx950) kriging hot path behind egobox-gp's builder / fit / predict API.

Importing the package loads `egobox_amd/lib/libegx_gp_hip.so` (C ABI: include/egx_gp.h); the import fails
if the library has not been built, and model construction fails with NoDeviceError without a GPU: there is
no CPU or PyTorch fallback for the compute path.
"""
from . import _lib
from ._lib import (EgxError, InvalidValueError, LikelihoodComputationError, LinalgError, NoDeviceError,
                   NotFittedError, PeerError)

_lib.load()  # fail loudly at import when the HIP library is missing

from .gp import (AbsoluteExponentialCorr, ConstantMean, GaussianProcess, GpHandle, GpParams, Kriging,  # noqa: E402
                 LinearMean, Matern32Corr, Matern52Corr, QuadraticMean, SquaredExponentialCorr, ThetaTuning,
                 chain_stats, corr_matrix, cross_corr, finalize_multi, fit_multi, likelihood_multi, mfma_probe, normalize, pool_stats, potrf,
                 regression_basis, set_tuning, trim)
from .gpx import CorrelationSpec, GpMix, Gpx, Recombination, RegressionSpec  # noqa: E402
from .multistart import prepare_multistart, theta_sweep_candidates  # noqa: E402
from .sgp import (Inducings, ParamTuning, SgpHandle, SgpParams, SparseGaussianProcess, SparseGpMix, SparseGpx,  # noqa: E402
                  SparseMethod)
from . import moe, workload  # noqa: E402
from .sweep import Sweep, best_candidate, rendezvous_sweep, shard_indices, sweep_likelihood  # noqa: E402

__all__ = [n for n in dir() if not n.startswith("_")]
