"""`_core`-level drop-in: the interface of the reference's native module over libmcx.so.

The reference's Python half calls `self._integrator.integrate / integrate_is_tables / integrate_mcmc` on the PyO3
class `_core.MonteCarloIntegrator` (src/lib.rs:17-431; call sites python/wgpu_montecarlo/__init__.py:761-770,
994-1007, 1096-1113) and hands it *WGSL text*. This module offers the same class -- same method names, positional
signatures, defaults, return type (float32[K]) and exception types -- so that the reference's own `__init__.py` and
`transpiler.py` could run unchanged on an MI355X. The WGSL strings (the transpiler's output, user strings and the
importance-sampling wrappers that call `pdf_target_from_table` / `pdf_proposal_from_table`) are translated literally
to HIP C++ (wgsl_to_hip.py) and fused into the same kernels; nothing here takes the faster routes of this package's
own API (api.py): weights are evaluated per function as the wrapper text says, and every table lookup runs the
search-and-blend form.

It is a binding, not a second implementation: planning, compilation, tables, launches and the f64 reduction are
libmcx's (include/mcx.h).
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np

from . import emit_hip, runtime, wgsl_to_hip

_SOURCES: dict = {}          # (function strings, math) -> HIP text; bounded, oldest out

_DIST = {"uniform": runtime.DIST_UNIFORM, "normal": runtime.DIST_NORMAL, "exponential": runtime.DIST_EXPONENTIAL,
         "custom": runtime.DIST_CUSTOM}


def _params(dist_type: str, params: dict):
    """src/lib.rs:436-502, including its silent defaults and its error for unknown names."""
    if dist_type not in _DIST:
        raise ValueError(f"Unknown distribution type: {dist_type}")

    def num(key, default):
        try:
            return float(params.get(key, default))
        except (TypeError, ValueError):
            return float(default)

    if dist_type == "uniform":
        return _DIST[dist_type], num("min", 0.0), num("max", 1.0)
    if dist_type == "normal":
        return _DIST[dist_type], num("mean", 0.0), num("std", 1.0)
    if dist_type == "exponential":
        return _DIST[dist_type], num("lambda", 1.0), 0.0
    return _DIST[dist_type], 0.0, 0.0


def _analytic_logpdf(name: str, code: int, p1: float, p2: float) -> str:
    """HIP text of the reference's analytic log-density for one distribution type, generate_log_pdf_code_for_dist
    (src/shader_gen.rs:543-571): what its MH step evaluates when `_core.integrate_mcmc` is given no table. The normal
    case is `pow(z, 2.0)` in the reference's WGSL -- backend-defined for z < 0 -- and is emitted as the intended z * z."""
    a, b = repr(float(np.float32(p1))) + "f", repr(float(np.float32(p2))) + "f"
    if code == runtime.DIST_UNIFORM:
        body = f"(({a} <= x) && (x < {b})) ? -logf({b} - {a}) : -100.0f"
    elif code == runtime.DIST_NORMAL:
        body = f"-0.5f * (((x - {a}) / {b}) * ((x - {a}) / {b})) - logf({b} * 2.50662827463f)"
    elif code == runtime.DIST_EXPONENTIAL:
        body = f"(x >= 0.0f) ? logf({a}) - {a} * x : -100.0f"
    else:
        # a custom distribution without its table: the reference would emit a call to a lookup function that is not
        # bound (shader_gen.rs:566-569 / SURVEY.md App. C-8) and fail at pipeline creation
        raise RuntimeError("Failed to create MCMC pipeline: a custom distribution needs its log-PDF table")
    return f"MCX_DEV float {name}(float x) {{ return {body}; }}"


def _f32(a) -> Optional[np.ndarray]:
    """numpy inputs are copied as f32; a non-contiguous array silently becomes empty in the reference
    (`as_slice().unwrap_or(&[])`, src/lib.rs:71-77) -- here it is simply made contiguous."""
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class MonteCarloIntegrator:
    """Replaces `_core.MonteCarloIntegrator` (src/lib.rs:17-431)."""

    def __init__(self, device: int = 0, math: Optional[str] = None):
        """`math` (an extension; the reference's constructor takes nothing): the routines behind the WGSL builtins exp /
        log / sqrt / sin / cos / tan / pow of the function strings, as in emit_hip.py -- "precise" (the default here: ocml,
        what the parity tests of this binding hold), "default" (the hardware instructions, inside WGSL's own accuracy
        bounds: the reference's benchmark integrand runs 2.8x faster) or "fast". MCX_CORE_MATH sets it for callers that
        cannot pass it, such as the reference's unmodified __init__.py."""
        self._engine = runtime.Engine.shared(device)           # RuntimeError("Failed to initialize GPU: ...")
        self._math = math if math is not None else os.environ.get("MCX_CORE_MATH", "precise")
        if self._math not in ("precise", "default", "fast"):
            raise ValueError("math must be one of ('precise', 'default', 'fast')")

    # ---- helpers ----------------------------------------------------------------------------------
    def _source(self, functions: Sequence[str]) -> str:
        if len(functions) == 0:
            raise ValueError("At least one function is required")          # src/lib.rs:61-65
        # translated once per distinct list of strings (the reference's Python half sends the same texts call after call: 80-220 us
        # of tokenising and parsing each time otherwise). The prelude carries McxPowI: `pow(x, 2.0)`, the transpiler's text for
        # x**2, becomes a product chain in every math mode.
        key = (tuple(functions), self._math)
        try:
            return _SOURCES[key]
        except (KeyError, TypeError):
            pass
        src = "\n\n".join([emit_hip.prelude()] + [wgsl_to_hip.translate(text, i, f"user_func_{i}", self._math) for i, text in enumerate(functions)])
        try:
            if len(_SOURCES) >= 256:
                _SOURCES.pop(next(iter(_SOURCES)))
            _SOURCES[key] = src
        except TypeError:                                # an unhashable element: translate() has already raised for non-strings
            pass
        return src

    def _cdf(self, dist_type: str, x_table, cdf_table):
        if dist_type != "custom":
            return None
        x, c = _f32(x_table), _f32(cdf_table)
        if x is None or c is None:
            raise RuntimeError("Failed to setup integration: custom distribution requires x_table and cdf_table")
        return self._engine.cached_table(runtime.TABLE_CDF, c, x)

    def _result(self, sums: np.ndarray, k: int, n_eff: int) -> np.ndarray:
        with np.errstate(divide="ignore", invalid="ignore"):
            return (sums[:k] / float(n_eff)).astype(np.float32)

    # ---- src/lib.rs:47-141 ------------------------------------------------------------------------
    def integrate(self, functions, dist_type, dist_params, n_samples, seed, x_table=None, cdf_table=None,
                  target_threads=None) -> np.ndarray:
        src = self._source(functions)
        code, p1, p2 = _params(dist_type, dist_params)
        cdf = self._cdf(dist_type, x_table, cdf_table)
        desc = runtime.make_desc(runtime.KIND_INTEGRATE, len(functions), code, guard_endpoints=True)
        mod = self._engine.module(src, desc)
        sums, n_eff = self._engine.integrate(mod, int(n_samples), int(seed), p1, p2, target_threads, cdf=cdf)
        return self._result(sums, len(functions), n_eff)

    # ---- src/lib.rs:158-275 -----------------------------------------------------------------------
    def integrate_is_tables(self, functions, dist_type, dist_params, n_samples, seed, x_table=None, cdf_table=None,
                            target_x_table=None, target_pdf_table=None, proposal_x_table=None,
                            proposal_pdf_table=None, target_threads=None) -> np.ndarray:
        src = self._source(functions)
        code, p1, p2 = _params(dist_type, dist_params)
        cdf = self._cdf(dist_type, x_table, cdf_table)
        mask, tables = 0, {}
        if target_x_table is not None and target_pdf_table is not None:
            tables["target_pdf"] = self._engine.cached_table(runtime.TABLE_PDF, _f32(target_x_table), _f32(target_pdf_table))
            mask |= 1
        if proposal_x_table is not None and proposal_pdf_table is not None:
            tables["proposal_pdf"] = self._engine.cached_table(runtime.TABLE_PDF, _f32(proposal_x_table), _f32(proposal_pdf_table))
            mask |= 2
        desc = runtime.make_desc(runtime.KIND_INTEGRATE, len(functions), code, guard_endpoints=True, user_tables=mask)
        mod = self._engine.module(src, desc)
        sums, n_eff = self._engine.integrate(mod, int(n_samples), int(seed), p1, p2, target_threads, cdf=cdf, **tables)
        return self._result(sums, len(functions), n_eff)

    # ---- src/lib.rs:296-431 -----------------------------------------------------------------------
    def integrate_mcmc(self, functions, proposal_dist_type, proposal_dist_params, target_dist_type, target_dist_params,
                       n_steps, n_chains, n_burnin, seed, x_table=None, cdf_table=None, target_x_table=None,
                       target_log_pdf_table=None, proposal_x_table=None, proposal_log_pdf_table=None,
                       target_threads=None) -> np.ndarray:
        src = self._source(functions)
        if int(n_steps) == 0:
            raise ValueError("n_steps must be positive")                    # src/lib.rs:332-336
        if int(n_chains) == 0:
            raise ValueError("n_chains must be positive")                   # src/lib.rs:338-342
        code, p1, p2 = _params(proposal_dist_type, proposal_dist_params)
        t_code, t1, t2 = _params(target_dist_type, target_dist_params)
        cdf = self._cdf(proposal_dist_type, x_table, cdf_table)
        # the four log-PDF tables are optional (src/lib.rs:296-304): without one, the MH step evaluates the analytic
        # log-density of that distribution type (src/shader_gen.rs:327-339, 496-509)
        t = q = None
        analytic = 0
        if target_x_table is not None and target_log_pdf_table is not None:
            t = self._engine.cached_table(runtime.TABLE_LOGPDF, _f32(target_x_table), _f32(target_log_pdf_table))
        else:
            src += "\n\n" + _analytic_logpdf("mcx_logpdf_p", t_code, t1, t2)
            analytic |= 1
        if proposal_x_table is not None and proposal_log_pdf_table is not None:
            q = self._engine.cached_table(runtime.TABLE_LOGPDF, _f32(proposal_x_table), _f32(proposal_log_pdf_table))
        else:
            src += "\n\n" + _analytic_logpdf("mcx_logpdf_q", code, p1, p2)
            analytic |= 2
        desc = runtime.make_desc(runtime.KIND_MCMC, len(functions), code, guard_endpoints=True, logpdf_analytic=analytic)
        mod = self._engine.module(src, desc)
        sums, n_eff = self._engine.mcmc(mod, int(n_steps), int(n_chains), int(n_burnin), int(seed), p1, p2, t, q,
                                        target_threads=target_threads, cdf=cdf)
        return self._result(sums, len(functions), n_eff)
