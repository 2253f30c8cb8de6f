"""wgpu_montecarlo -- MI355X-native drop-in for NightingaleCen/wgpu-monte-carlo's Python API.

    from wgpu_montecarlo import MonteCarloIntegrator, Distribution
    result = MonteCarloIntegrator().integrate([lambda x: x, lambda x: x**2],
                                              Distribution.normal(0.0, 1.0), n_samples=10**9)

Same names as the reference package (python/wgpu_montecarlo/__init__.py:61-71); the compute path is a
fused HIP kernel for gfx950 behind the C ABI in include/mcx.h.
"""
from .frontend import TranspilerError
from .transpile import PythonToWGSL, PythonToHIP, transpile_function, transpile_function_hip
from .distributions import Distribution, DistributionType
from .api import (
    IntegrationResult,
    MonteCarloIntegrator,
    integrate,
    integrate_importance_sampling,
    integrate_mcmc,
)

try:
    from . import runtime as _runtime

    _runtime.load()
    HAS_NATIVE_EXTENSION = True
except ImportError:  # libmcx.so not built: constructing an integrator raises ImportError
    HAS_NATIVE_EXTENSION = False

# name the reference exports for "the native half is importable" (reference __init__.py:49-57)
HAS_RUST_EXTENSION = HAS_NATIVE_EXTENSION

__version__ = "0.2.0+mi355x.1"

__all__ = [
    "MonteCarloIntegrator",
    "Distribution",
    "IntegrationResult",
    "PythonToWGSL",
    "transpile_function",
    "TranspilerError",
    "integrate",
    "integrate_importance_sampling",
    "integrate_mcmc",
]
