"""Probability distributions and their lookup tables (host side).

Drop-in for the reference's `Distribution` (python/wgpu_montecarlo/__init__.py:79-608): same factory
names, same `dist_type` / `params` / `_x_table` / `_cdf_table` / `_pdf_table` attributes, same table
construction rules, so the tables handed to the GPU are the ones the reference would upload:

  * support search          reference __init__.py:88-206   (scan grid -> hill climb -> expand)
  * CDF table               reference __init__.py:209-251  (>= 1000 points, trapezoid, normalised to 1)
  * PDF table               reference __init__.py:549-570
  * log-PDF table for MH    reference __init__.py:572-608  (log(max(p,1e-16)), -100 where p <= 0,
                                                            uniform end point patched)
Checked against tables captured from the reference in tests/golden/ (tools/make_golden.py).
"""
from __future__ import annotations

import math
import threading
import types
from enum import Enum, auto
from typing import Callable, Optional, Sequence, Tuple, Union

import numpy as np


class DistributionType(Enum):
    UNIFORM = auto()
    NORMAL = auto()
    EXPONENTIAL = auto()
    CUSTOM = auto()


_MIN_TABLE_POINTS = 1000
_SCAN_ERRORS = (ValueError, TypeError, OverflowError)


def _scan_grid() -> list:
    """Half-integer grid on [-4, 4] plus +-2^4 .. +-2^10."""
    pts = {k * 0.5 for k in range(-8, 9)}
    for e in range(4, 11):
        pts.update((float(2 ** e), -float(2 ** e)))
    return sorted(pts)


def _find_support(pdf: Callable, threshold_ratio: float = 1e-5, max_hard_limit: float = 10000.0) -> tuple:
    """Effective support of a PDF: locate a positive point, climb to the mode, expand until the
    density falls under `threshold_ratio` of the peak (doubling steps)."""
    start = None
    for x in _scan_grid():
        try:
            v = pdf(x)
        except _SCAN_ERRORS:
            continue
        if v > 0 and math.isfinite(v):
            start = (x, v)
            break
    if start is None:
        raise ValueError(
            "PDF is zero everywhere in scanned range [-4, 4] (step=0.5) and [-1024, 1024] (exponential).\n"
            "This may happen if your distribution is:\n"
            "  - Bounded and located outside [-4, 4] (e.g., Uniform(10, 10.1))\n"
            "  - Heavily shifted (e.g., N(1000, 1)) but not detected by exponential scan\n\n"
            "Solution: Manually specify the support parameter:\n"
            "  dist = Distribution.from_pdf(your_pdf, support=(x_min, x_max))\n\n"
            "Example for Uniform(5, 10):\n"
            "  def my_pdf(x):\n"
            "      return 0.2 if 5 <= x < 10 else 0.0\n"
            "  dist = Distribution.from_pdf(my_pdf, support=(5.0, 10.0))"
        )
    peak_x, peak_v = start
    stride = 1.0
    for _ in range(100):
        lo_v = pdf(peak_x - stride) if peak_x - stride > -max_hard_limit else 0
        hi_v = pdf(peak_x + stride) if peak_x + stride < max_hard_limit else 0
        if lo_v > peak_v:
            peak_x, peak_v = peak_x - stride, lo_v
        elif hi_v > peak_v:
            peak_x, peak_v = peak_x + stride, hi_v
        else:
            stride /= 2
            if stride < 1e-6:
                break
    cutoff = peak_v * threshold_ratio

    def expand(direction: float) -> float:
        edge, stride = peak_x, 0.1
        while (edge > -max_hard_limit) if direction < 0 else (edge < max_hard_limit):
            try:
                v = pdf(edge + direction * stride)
            except _SCAN_ERRORS:
                break
            edge = edge + direction * stride
            if v <= 0 or v < cutoff:
                break
            stride *= 2
        return edge

    return expand(-1.0), expand(+1.0)


def _compute_cdf_table(pdf: Callable, x_min: float, x_max: float, n_points: int = 2048) -> tuple:
    """(x_grid, cdf) on a uniform grid of max(n_points, 1000) points; trapezoid rule; cdf[-1] == 1."""
    n = max(n_points, _MIN_TABLE_POINTS)
    grid = np.linspace(x_min, x_max, n)
    dens = np.array([pdf(x) for x in grid])
    dens = np.clip(np.nan_to_num(dens, nan=0.0, posinf=0.0, neginf=0.0), 0, None)
    step = (x_max - x_min) / (n - 1)
    cdf = np.zeros(n)
    cdf[1:] = np.cumsum((dens[:-1] + dens[1:]) / 2) * step
    total = cdf[-1]
    if total <= 0:
        raise ValueError("PDF integral is zero. Please check the PDF function or support range.")
    return grid, cdf / total


_CDF_TABLES: dict = {}        # (density key, support, size) -> (x f32, cdf f32), read-only; bounded, oldest out
_SUPPORTS: dict = {}          # density key -> (x_min, x_max) found by _find_support
_PDF_TABLES: dict = {}        # (density key, x grid bytes) -> pdf f32 on that grid, read-only
_TABLE_LOCK = threading.Lock()    # updates of the three (lookups are lock-free)


def _remember(cache: dict, key, value, limit: int = 64) -> None:
    with _TABLE_LOCK:
        while len(cache) >= limit:
            cache.pop(next(iter(cache)))             # oldest entry (dicts keep insertion order)
        cache[key] = value


_PLAIN = (int, float, bool, str, complex, type(None), np.generic, types.ModuleType, types.BuiltinFunctionType, np.ufunc)


def _density_key(pdf: Callable):
    """A hashable key that is equal for two density callables only if they compute the same function: the code object, every
    captured value, the defaults and the value of every global the code names -- accepted only where all of those are plain values
    (numbers, strings, modules, builtin functions); anything else (a captured list, an object with state, a nested function) gives
    no key, and no caching. Independent of the transpiler's subset: Beta's density, say, is outside it."""
    code = getattr(pdf, "__code__", None)
    if code is None or any(isinstance(c, types.CodeType) for c in code.co_consts):
        return None
    values = [c.cell_contents for c in (pdf.__closure__ or ())]
    values += list(pdf.__defaults__ or ()) + [v for _, v in sorted((pdf.__kwdefaults__ or {}).items())]
    g = pdf.__globals__
    named = tuple((n, g[n]) for n in code.co_names if n in g)
    values += [v for _, v in named]
    if not all(isinstance(v, _PLAIN) for v in values):
        return None
    return (code, tuple(values))


def _cached_cdf_table(pdf: Callable, x_min: float, x_max: float, n_points: int):
    """_compute_cdf_table as float32 arrays, remembered per density: `Distribution.beta(2, 5)` or `from_pdf(lambda ...)` written
    inline in every call (the reference's examples do) evaluates the density at 2048 points in Python each time -- 2 ms against a
    0.05 ms GPU call -- and, as new arrays, would also look like a new distribution to the plan cache. The arrays of a cached
    entry are shared and read-only."""
    try:
        key = _density_key(pdf)
        key = None if key is None else (key, float(x_min), float(x_max), int(n_points))
        hit = _CDF_TABLES.get(key) if key is not None else None
    except (TypeError, ValueError):                     # an unhashable value after all, an empty closure cell: no key, no cache
        key, hit = None, None
    if hit is None:
        grid, cdf = _compute_cdf_table(pdf, x_min, x_max, n_points)
        hit = (grid.astype(np.float32), cdf.astype(np.float32))
        if key is not None:
            for a in hit:
                a.setflags(write=False)
            _remember(_CDF_TABLES, key, hit)
    return hit


class Distribution:
    """A sampling / target distribution. Use the factory methods."""

    def __init__(self, dist_type: DistributionType, params: dict, pdf_func: Callable[[float], float],
                 x_table: Optional[np.ndarray] = None, cdf_table: Optional[np.ndarray] = None,
                 pdf_table: Optional[np.ndarray] = None):
        self.dist_type = dist_type
        self.params = params
        self._pdf_func = pdf_func
        self._x_table = x_table
        self._cdf_table = cdf_table
        self._pdf_table = pdf_table

    def pdf(self, x: float) -> float:
        return self._pdf_func(x)

    # ---- analytic families -------------------------------------------------------------------
    @staticmethod
    def uniform(min: float = 0.0, max: float = 1.0) -> "Distribution":
        """U(min, max), half-open; sampled as min + u*(max-min) on the GPU."""
        width = max - min

        def pdf(x: float) -> float:
            return 1.0 / width if (min <= x) and (x < max) else 0.0

        return Distribution(DistributionType.UNIFORM, {"min": min, "max": max, "support": (min, max)}, pdf)

    @staticmethod
    def normal(mean: float = 0.0, std: float = 1.0) -> "Distribution":
        """N(mean, std); Box-Muller on the GPU; support recorded as mean +- 7 std."""
        sigma = std
        sqrt_2pi = np.sqrt(2 * np.pi)

        def pdf(x: float) -> float:
            z = (x - mean) / sigma
            return np.exp(-0.5 * z * z) / (sigma * sqrt_2pi)

        return Distribution(DistributionType.NORMAL,
                            {"mean": mean, "std": std, "support": (mean - 7 * std, mean + 7 * std)}, pdf)

    @staticmethod
    def exponential(lambda_param: float = 1.0) -> "Distribution":
        """Exp(lambda); inverse-CDF sampling on the GPU; support recorded as (0, 10/lambda)."""

        def pdf(x: float) -> float:
            return lambda_param * math.exp(-lambda_param * x) if x >= 0 else 0.0

        return Distribution(DistributionType.EXPONENTIAL,
                            {"lambda": lambda_param, "support": (0.0, 10.0 / lambda_param)}, pdf)

    @staticmethod
    def beta(alpha: float, beta_param: float, table_size: int = 2048) -> "Distribution":
        """Beta(alpha, beta) through a CDF table on [0, 1]."""
        try:
            from scipy.special import beta as beta_fn
        except ImportError:
            raise ImportError("scipy is required for Beta distribution. Install with: pip install scipy")
        norm = beta_fn(alpha, beta_param)

        def pdf(x: float) -> float:
            if 0 < x < 1:
                return (x ** (alpha - 1)) * ((1 - x) ** (beta_param - 1)) / norm
            return 0.0

        return Distribution.from_pdf(pdf, support=(0.0, 1.0), table_size=table_size)

    # ---- table based -------------------------------------------------------------------------
    @staticmethod
    def from_pdf(pdf_func: Callable[[float], float], support: Optional[tuple] = None,
                 table_size: int = 2048) -> "Distribution":
        """Custom distribution from a PDF callable (support auto-detected unless given)."""
        if not callable(pdf_func):
            raise TypeError("pdf_func must be callable")
        if support is not None:
            x_min, x_max = support
        else:                                            # the scan evaluates the density a few hundred times: remembered like the table
            try:
                skey = _density_key(pdf_func)
                found = _SUPPORTS.get(skey) if skey is not None else None
            except (TypeError, ValueError):
                skey, found = None, None
            if found is None:
                found = _find_support(pdf_func)
                if skey is not None:
                    _remember(_SUPPORTS, skey, found)
            x_min, x_max = found
        xs, cdf = _cached_cdf_table(pdf_func, x_min, x_max, table_size)
        return Distribution(DistributionType.CUSTOM, {"table_size": len(xs), "support": (x_min, x_max)},
                            pdf_func, x_table=xs, cdf_table=cdf)

    @staticmethod
    def from_pdf_table(x_table: Union[np.ndarray, Sequence[float]], pdf_table: Union[np.ndarray, Sequence[float]],
                       cdf_table: Optional[Union[np.ndarray, Sequence[float]]] = None) -> "Distribution":
        """Custom distribution from tabulated (x, pdf) and optionally a CDF."""
        xs = np.asarray(x_table, dtype=np.float32)
        ps = np.asarray(pdf_table, dtype=np.float32)
        if xs.ndim != 1 or ps.ndim != 1:
            raise ValueError("x_table and pdf_table must be 1D arrays")
        if len(xs) != len(ps):
            raise ValueError("x_table and pdf_table must have the same length")
        if len(xs) < 2:
            raise ValueError("Tables must have at least 2 points")
        if not np.all(np.diff(xs) > 0):
            raise ValueError("x_table must be sorted in ascending order")
        if np.any(ps < 0):
            raise ValueError("pdf_table must contain non-negative values")
        n = len(xs)
        lo, hi = float(xs[0]), float(xs[-1])
        if cdf_table is not None:
            cdf = np.asarray(cdf_table, dtype=np.float32)
            if len(cdf) != n:
                raise ValueError("cdf_table must have same length as x_table")
        else:
            # running trapezoid sum carried in float32, like the reference's element loop
            cdf = np.zeros(n, dtype=np.float32)
            for i in range(1, n):
                cdf[i] = cdf[i - 1] + np.float32(0.5) * (ps[i] + ps[i - 1]) * (xs[i] - xs[i - 1])
            if cdf[-1] > 0:
                cdf = cdf / cdf[-1]
        dens = ps.copy()

        def pdf_func(x: float) -> float:
            if x < lo or x > hi:
                return 0.0
            j = np.searchsorted(xs, x)
            if j == 0:
                return float(dens[0])
            if j >= n:
                return float(dens[-1])
            t = (x - xs[j - 1]) / (xs[j] - xs[j - 1])
            return float((1 - t) * dens[j - 1] + t * dens[j])

        return Distribution(DistributionType.CUSTOM, {"table_size": n, "support": (lo, hi)}, pdf_func,
                            x_table=xs, cdf_table=cdf, pdf_table=ps)

    # ---- derived tables ----------------------------------------------------------------------
    def get_or_compute_pdf_table(self) -> Tuple[np.ndarray, np.ndarray]:
        """(x_table, pdf_table) as float32, evaluating the PDF on the x grid if needed."""
        if self._pdf_table is not None and self._x_table is not None:
            return self._x_table, self._pdf_table
        if self._x_table is None:
            x_min, x_max = self.params.get("support", (-5.0, 5.0))
            self._x_table = np.linspace(x_min, x_max, self.params.get("table_size", 2048), dtype=np.float32)
        # the 2048 density evaluations, once per distinct density and grid (an inline Distribution.normal(0, 2) as the proposal of
        # every integrate_mcmc call is a new object each time)
        try:
            key = _density_key(self._pdf_func)
            key = None if key is None else (key, self._x_table.tobytes())
            hit = _PDF_TABLES.get(key) if key is not None else None
        except (TypeError, ValueError):
            key, hit = None, None
        if hit is None:
            hit = np.array([self._pdf_func(float(x)) for x in self._x_table], dtype=np.float32)
            if key is not None:
                hit.setflags(write=False)
                _remember(_PDF_TABLES, key, hit)
        self._pdf_table = hit
        return self._x_table, self._pdf_table

    def get_log_pdf_table(self, min_log_value: float = -100.0) -> Tuple[np.ndarray, np.ndarray]:
        """(x_table, log_pdf_table) for Metropolis-Hastings."""
        xs, dens = self.get_or_compute_pdf_table()
        with np.errstate(divide="ignore", invalid="ignore"):
            logs = np.where(dens > 0, np.log(np.maximum(dens, 1e-16)), min_log_value).astype(np.float32)
        if self.dist_type == DistributionType.UNIFORM:
            # the grid includes x = max where the half-open PDF is 0: give it the interior value
            width = self.params.get("max", 1.0) - self.params.get("min", 0.0)
            if width > 0:
                logs[-1] = np.log(1.0 / width)
        return xs, logs
