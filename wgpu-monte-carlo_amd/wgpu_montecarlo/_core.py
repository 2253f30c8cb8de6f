"""`_core`-level drop-in: the interface of the reference's native module over libmcx.so.

The reference's Python half calls `self._integrator.integrate / integrate_is_tables / integrate_mcmc` on the PyO3
class `_core.MonteCarloIntegrator` (src/lib.rs:17-431; call sites python/wgpu_montecarlo/__init__.py:761-770,
994-1007, 1096-1113) and hands it *WGSL text*. This module offers the same class -- same method names, positional
signatures, defaults, return type (float32[K]) and exception types -- so that the reference's own `__init__.py` and
`transpiler.py` could run unchanged on an MI355X. The WGSL strings (the transpiler's output, user strings and the
importance-sampling wrappers that call `pdf_target_from_table` / `pdf_proposal_from_table`) are planned and translated to
HIP C++ inside libmcx (include/mcx.h: mcx_wgsl_plan, mcx_module_desc_fit) and fused into the same kernels; what is left here is
the reference's argument conventions: its parameter dicts with their silent defaults, numpy tables, exception types, float32[K].

Two ways of compiling what arrives, chosen by `math` (constructor keyword, or MCX_CORE_MATH):
  "precise"  literal: every string is one function of the kernel, evaluated as written -- the importance weight per
             function, table lookups by search and blend, ocml builtins.
  "default"  (and "fast") the plan this package's own API builds for the same call (api.py): the K importance-sampling
             wrappers the reference's Python half generates (python/wgpu_montecarlo/__init__.py:893-905, 968-980) are
             recognised by their fixed text and compiled as K integrands + ONE weight p/q per sample; strict-grid PDF /
             log-PDF tables take the one-read-one-FMA cell form; a normal proposal's log q (MH) comes from its own deviate;
             custom CDF tables use their bucket-direct form; the hardware builtins. Text that does not match the
             generator's format is compiled literally. Same parity bounds as the API's default (tests/test_gpu_core_binding.py
             holds both modes to the oracle).

It is a binding, not a second implementation: recognition of the payload, translation, planning, compilation, tables,
launches and the f64 reduction are libmcx's (include/mcx.h) -- a Rust or C host of the reference's src/lib.rs makes the same calls
(tests/cabi_client.c).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np

from . import runtime

_PLANS: dict = {}            # one payload shape -> (HIP text, desc) as libmcx planned it; bounded, oldest out

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


def _f32(a) -> Optional[np.ndarray]:
    """numpy inputs are copied as f32; a non-contiguous array silently becomes empty in the reference
    (`as_slice().unwrap_or(&[])`, src/lib.rs:71-77) -- here it is simply made contiguous."""
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class MonteCarloIntegrator:
    """Replaces `_core.MonteCarloIntegrator` (src/lib.rs:17-431)."""

    def __init__(self, device: int = 0, math: Optional[str] = None):
        """`math` (an extension; the reference's constructor takes nothing) picks how the strings are compiled, see the
        module docstring: "default" (the plan this package's API builds for the same call; also what MCX_CORE_MATH unset
        means), "fast", or "precise" (literal, ocml builtins). MCX_CORE_MATH sets it for callers that cannot pass it, such
        as the reference's unmodified __init__.py. BASELINE configs[1..3] through the binding, default / precise: 0.47 /
        0.48, 0.79 / 4.6, 8.3 / 17.6 ms (profiles/r03_core_binding_payloads.txt)."""
        self._engine = runtime.Engine.shared(device)           # RuntimeError("Failed to initialize GPU: ...")
        self._math = math if math is not None else os.environ.get("MCX_CORE_MATH", "default")
        if self._math not in ("precise", "default", "fast"):
            raise ValueError("math must be one of ('precise', 'default', 'fast')")

    # ---- planning ---------------------------------------------------------------------------------
    def _plan(self, kind, functions, code, p1, p2, have_target, have_proposal, t_code=0, t1=0.0, t2=0.0):
        """(HIP text, desc) of one payload: libmcx's own planning (include/mcx.h: mcx_wgsl_plan), once per distinct call
        shape -- the reference's Python half sends the same texts call after call."""
        if len(functions) == 0:
            raise ValueError("At least one function is required")          # src/lib.rs:61-65
        try:
            key = (kind, tuple(functions), code, p1, p2, self._math, have_target, have_proposal, t_code, t1, t2)
            hit = _PLANS.get(key)
        except TypeError:                                # an unhashable element: the planner refuses non-strings itself
            key, hit = None, None
        if hit is None:
            hit = runtime.wgsl_plan(kind, functions, code, p1, p2, self._math, have_target, have_proposal, t_code, t1, t2)
            if key is not None:
                if len(_PLANS) >= 256:
                    _PLANS.pop(next(iter(_PLANS)))
                _PLANS[key] = hit
        src, desc = hit
        return src, runtime.ModuleDesc.from_buffer_copy(bytes(desc))        # the caller fits its own copy

    def _module(self, src: str, desc, p1: float, p2: float, cdf, t0=None, t1=None):
        """math = "precise": the module as planned. Otherwise with libmcx's table decisions for the call (mcx_module_desc_fit:
        cell form on strict grids, sentinel pads, LDS staging, bucket-direct sampling), held against the code object's real
        static LDS exactly as api.py builds its plans."""
        if self._math == "precise":
            return self._engine.module(src, desc)
        from . import api

        pad_bytes = runtime.module_desc_fit(desc, cdf, t0, t1, p1, p2)
        return api.build_module(self._engine, src, desc, cdf, t0, t1, extra_bytes=pad_bytes)

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

    def _integrate(self, functions, dist_type, dist_params, n_samples, seed, x_table, cdf_table, target, proposal, target_threads):
        """integrate / integrate_is_tables: `target` / `proposal` = (x, pdf) tables or None."""
        if len(functions) == 0:
            raise ValueError("At least one function is required")          # src/lib.rs:61-65
        code, p1, p2 = _params(dist_type, dist_params)
        cdf = self._cdf(dist_type, x_table, cdf_table)
        p_tab = self._engine.cached_table(runtime.TABLE_PDF, _f32(target[0]), _f32(target[1])) if target is not None else None
        q_tab = self._engine.cached_table(runtime.TABLE_PDF, _f32(proposal[0]), _f32(proposal[1])) if proposal is not None else None
        src, desc = self._plan(runtime.KIND_INTEGRATE, functions, code, p1, p2, p_tab is not None, q_tab is not None)
        mod = self._module(src, desc, p1, p2, cdf, p_tab, q_tab)
        # the PDF tables are bound the same way whether the module weights with them (desc.weight) or its functions read them
        # (desc.user_tables)
        sums, n_eff = self._engine.integrate(mod, int(n_samples), int(seed), p1, p2, target_threads, cdf=cdf,
                                             target_pdf=p_tab, proposal_pdf=q_tab)
        return self._result(sums, len(functions), n_eff)

    # ---- src/lib.rs:47-141 ------------------------------------------------------------------------
    def integrate(self, functions, dist_type, dist_params, n_samples, seed, x_table=None, cdf_table=None,
                  target_threads=None) -> np.ndarray:
        return self._integrate(functions, dist_type, dist_params, n_samples, seed, x_table, cdf_table, None, None, target_threads)

    # ---- src/lib.rs:158-275 -----------------------------------------------------------------------
    def integrate_is_tables(self, functions, dist_type, dist_params, n_samples, seed, x_table=None, cdf_table=None,
                            target_x_table=None, target_pdf_table=None, proposal_x_table=None,
                            proposal_pdf_table=None, target_threads=None) -> np.ndarray:
        target = (target_x_table, target_pdf_table) if target_x_table is not None and target_pdf_table is not None else None
        proposal = (proposal_x_table, proposal_pdf_table) if proposal_x_table is not None and proposal_pdf_table is not None else None
        return self._integrate(functions, dist_type, dist_params, n_samples, seed, x_table, cdf_table, target, proposal, target_threads)

    # ---- src/lib.rs:296-431 -----------------------------------------------------------------------
    def integrate_mcmc(self, functions, proposal_dist_type, proposal_dist_params, target_dist_type, target_dist_params,
                       n_steps, n_chains, n_burnin, seed, x_table=None, cdf_table=None, target_x_table=None,
                       target_log_pdf_table=None, proposal_x_table=None, proposal_log_pdf_table=None,
                       target_threads=None) -> np.ndarray:
        if len(functions) == 0:
            raise ValueError("At least one function is required")
        if int(n_steps) == 0:
            raise ValueError("n_steps must be positive")                    # src/lib.rs:332-336
        if int(n_chains) == 0:
            raise ValueError("n_chains must be positive")                   # src/lib.rs:338-342
        code, p1, p2 = _params(proposal_dist_type, proposal_dist_params)
        t_code, t1, t2 = _params(target_dist_type, target_dist_params)
        cdf = self._cdf(proposal_dist_type, x_table, cdf_table)
        have_t = target_x_table is not None and target_log_pdf_table is not None
        have_q = proposal_x_table is not None and proposal_log_pdf_table is not None
        # the log-PDF tables are optional (src/lib.rs:296-304): the plan says which of them the module reads (a missing one becomes
        # the analytic log-density of that distribution type; a normal proposal's log q comes from its own deviate unless math is
        # "precise")
        src, desc = self._plan(runtime.KIND_MCMC, functions, code, p1, p2, have_t, have_q, t_code, t1, t2)
        t = self._engine.cached_table(runtime.TABLE_LOGPDF, _f32(target_x_table), _f32(target_log_pdf_table)) if have_t else None
        q = None
        if have_q and not desc.q_sampler:
            q = self._engine.cached_table(runtime.TABLE_LOGPDF, _f32(proposal_x_table), _f32(proposal_log_pdf_table))
        if self._math != "precise":
            # the workgroup size a small chain count wants (one chain per thread)
            hint = runtime.mcmc_block_hint(runtime.mcmc_dispatch_config(int(n_chains), target_threads).total_threads)
            desc.block = 0 if hint >= 1024 else hint
        mod = self._module(src, desc, p1, p2, cdf, t, q)
        sums, n_eff = self._engine.mcmc(mod, int(n_steps), int(n_chains), int(n_burnin), int(seed), p1, p2, t, q,
                                        target_threads=target_threads, cdf=cdf)
        return self._result(sums, len(functions), n_eff)
