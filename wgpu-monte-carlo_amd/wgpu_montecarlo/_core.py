"""`_core`-level drop-in: the interface of the reference's native module over libmcx.so.

The reference's Python half calls `self._integrator.integrate / integrate_is_tables / integrate_mcmc` on the PyO3
class `_core.MonteCarloIntegrator` (src/lib.rs:17-431; call sites python/wgpu_montecarlo/__init__.py:761-770,
994-1007, 1096-1113) and hands it *WGSL text*. This module offers the same class -- same method names, positional
signatures, defaults, return type (float32[K]) and exception types -- so that the reference's own `__init__.py` and
`transpiler.py` could run unchanged on an MI355X. The WGSL strings (the transpiler's output, user strings and the
importance-sampling wrappers that call `pdf_target_from_table` / `pdf_proposal_from_table`) are translated to HIP C++
(wgsl_to_hip.py) and fused into the same kernels.

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

It is a binding, not a second implementation: planning, compilation, tables, launches and the f64 reduction are
libmcx's (include/mcx.h).
"""
from __future__ import annotations

import os
import re
from typing import Optional, Sequence

import numpy as np

from . import emit_hip, runtime, wgsl_to_hip

_SOURCES: dict = {}          # (function strings, math) -> HIP text; bounded, oldest out

# The importance-sampling wrapper the reference generates per function i (python/wgpu_montecarlo/__init__.py:893-899, 968-974):
# `fn _is_wrapper_i(x) { let f_val = _is_f_orig_i(x); let p = <P>; let q = <Q>; return f_val * p / q; }` followed by the
# definitions of _is_pdf_p_i / _is_pdf_q_i (when analytic) and of _is_f_orig_i (the user's function, possibly with helpers).
_WRAPPER = re.compile(
    r"^\s*fn\s+_is_wrapper_(\d+)\s*\(\s*x\s*:\s*f32\s*\)\s*->\s*f32\s*\{\s*"
    r"let\s+f_val\s*=\s*_is_f_orig_\1\s*\(\s*x\s*\)\s*;\s*"
    r"let\s+p\s*=\s*(pdf_target_from_table|_is_pdf_p_\1)\s*\(\s*x\s*\)\s*;\s*"
    r"let\s+q\s*=\s*(pdf_proposal_from_table|_is_pdf_q_\1)\s*\(\s*x\s*\)\s*;\s*"
    r"return\s+f_val\s*\*\s*p\s*/\s*q\s*;\s*\}", re.S)
_FN_START = re.compile(r"(?m)^[ \t]*fn\s+([A-Za-z_][A-Za-z_0-9]*)\s*\(")


def _split_weighted(functions):
    """(f_texts, p_text or None, q_text or None) when EVERY string is one of the reference's importance-sampling wrappers
    around the same p and q (None = that density is read from its table); None when any of them is anything else."""
    f_texts, p_seen, q_seen = [], set(), set()
    for text in functions:
        m = _WRAPPER.match(text) if isinstance(text, str) else None
        if m is None:
            return None
        i, p_call, q_call = m.group(1), m.group(2), m.group(3)
        rest = text[m.end():]
        starts = [(g.start(), g.group(1)) for g in _FN_START.finditer(rest)]
        parts = {}                                  # the three named definitions; helpers stay with the one they follow
        order = []
        for j, (pos, name) in enumerate(starts):
            end = starts[j + 1][0] if j + 1 < len(starts) else len(rest)
            if name in (f"_is_pdf_p_{i}", f"_is_pdf_q_{i}", f"_is_f_orig_{i}"):
                order.append(name)
                parts[name] = rest[pos:end]
            elif order:
                parts[order[-1]] += rest[pos:end]
            else:
                return None
        if f"_is_f_orig_{i}" not in parts or rest[:starts[0][0] if starts else 0].strip():
            return None
        p_text = parts.get(f"_is_pdf_p_{i}")
        q_text = parts.get(f"_is_pdf_q_{i}")
        if (p_text is None) != (p_call == "pdf_target_from_table") or (q_text is None) != (q_call == "pdf_proposal_from_table"):
            return None
        p_seen.add(None if p_text is None else p_text.replace(f"_is_pdf_p_{i}", "_is_pdf_p").strip())
        q_seen.add(None if q_text is None else q_text.replace(f"_is_pdf_q_{i}", "_is_pdf_q").strip())
        f_texts.append(parts[f"_is_f_orig_{i}"])
    if len(p_seen) != 1 or len(q_seen) != 1:
        return None
    return f_texts, next(iter(p_seen)), next(iter(q_seen))


_MOMENT_X = re.compile(r"^\s*fn\s+\w+\s*\(\s*x\s*:\s*f32\s*\)\s*->\s*f32\s*\{\s*return\s+x\s*;\s*\}\s*$")
_MOMENT_POW = re.compile(r"^\s*fn\s+\w+\s*\(\s*x\s*:\s*f32\s*\)\s*->\s*f32\s*\{\s*return\s+pow\s*\(\s*x\s*,\s*(\d+)(?:\.0*)?\s*\)\s*;\s*\}\s*$")


def _moment_family(functions) -> bool:
    """Are the K >= 8 strings exactly the transpiler's text for x, x**2, .., x**K (`return x;`, `return pow(x, k.0);`)? Then
    the kernel accumulates the power sums of two / four samples at a time (desc.moment_family), as api._moment_family
    decides from the IR for the same workload (BASELINE configs[4])."""
    if len(functions) < 8 or not all(isinstance(t, str) for t in functions) or not _MOMENT_X.match(functions[0]):
        return False
    for i, text in enumerate(functions[1:], start=2):
        m = _MOMENT_POW.match(text)
        if not m or int(m.group(1)) != i:
            return False
    return True


def _is_normal_pdf_text(text: Optional[str], mean: float, std: float) -> bool:
    """Is `text` the closure Distribution.normal(mean, std) hands the transpiler (python/wgpu_montecarlo/__init__.py:343-347:
    exp(-0.5 z z) / (sigma sqrt_2pi), z = (x - mean) / sigma), for exactly the parameters the call samples with? Then 1/q is a
    function of the deviate the sampler already holds (desc.q_sampler), as in api.py."""
    if text is None:
        return False
    consts = dict(re.findall(r"const\s+(\w+)\s*:\s*f32\s*=\s*([-+0-9.eE]+)\s*;", text))
    body = re.sub(r"\s+", "", text)
    try:
        return (set(consts) == {"mean", "sigma", "sqrt_2pi"} and float(consts["mean"]) == float(mean) and float(consts["sigma"]) == float(std)
                and abs(float(consts["sqrt_2pi"]) - 2.5066282746310002) < 1e-12
                and "varz=((x-mean)/sigma);return(exp((((-0.5)*z)*z))/(sigma*sqrt_2pi));" in body)
    except ValueError:
        return False


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
        """`math` (an extension; the reference's constructor takes nothing) picks how the strings are compiled, see the
        module docstring: "default" (the plan this package's API builds for the same call; also what MCX_CORE_MATH unset
        means), "fast", or "precise" (literal, ocml builtins). MCX_CORE_MATH sets it for callers that cannot pass it, such
        as the reference's unmodified __init__.py. BASELINE configs[1..3] through the binding, default / precise: 0.47 /
        0.48, 0.79 / 4.6, 8.3 / 17.6 ms (profiles/r03_core_binding_payloads.txt)."""
        self._engine = runtime.Engine.shared(device)           # RuntimeError("Failed to initialize GPU: ...")
        self._math = math if math is not None else os.environ.get("MCX_CORE_MATH", "default")
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

    # ---- planning ---------------------------------------------------------------------------------
    @property
    def _literal(self) -> bool:
        return self._math == "precise"

    def _weighted_source(self, split, k: int) -> str:
        """K integrands + the densities of ONE weight: user_func_i from the wrappers' _is_f_orig_i, mcx_pdf_p / mcx_pdf_q from
        their analytic densities (libmcx forms p / q once per sample: desc.weight)."""
        f_texts, p_text, q_text = split
        key = ("weighted", tuple(f_texts), p_text, q_text, self._math)
        if key in _SOURCES:
            return _SOURCES[key]
        parts = [emit_hip.prelude()] + [wgsl_to_hip.translate(t, i, f"user_func_{i}", self._math) for i, t in enumerate(f_texts)]
        if p_text is not None:
            parts.append(wgsl_to_hip.translate(p_text, k, "mcx_pdf_p", self._math))
        if q_text is not None:
            parts.append(wgsl_to_hip.translate(q_text, k + 1, "mcx_pdf_q", self._math))
        if len(_SOURCES) >= 256:
            _SOURCES.pop(next(iter(_SOURCES)))
        _SOURCES[key] = "\n\n".join(parts)
        return _SOURCES[key]

    def _fitted_module(self, src: str, desc, p1: float, p2: float, cdf, t0=None, t1=None):
        """The module of `desc` with libmcx's own table decisions for the call (mcx_module_desc_fit: cell form on strict grids, the
        sentinel pads that spare the index clamp, LDS staging within the budget, bucket-direct sampling) -- then held against the
        code object's real static LDS, exactly as api.py builds its plans."""
        from . import api

        pad_bytes = runtime.module_desc_fit(desc, cdf, t0, t1, p1, p2)
        return api.build_module(self._engine, src, desc, cdf, t0, t1, extra_bytes=pad_bytes)

    def _integrate(self, functions, dist_type, dist_params, n_samples, seed, x_table, cdf_table, target, proposal, target_threads):
        """integrate / integrate_is_tables: `target` / `proposal` = (x, pdf) tables or None."""
        if len(functions) == 0:
            raise ValueError("At least one function is required")          # src/lib.rs:61-65
        code, p1, p2 = _params(dist_type, dist_params)
        cdf = self._cdf(dist_type, x_table, cdf_table)
        k = len(functions)
        p_tab = self._engine.cached_table(runtime.TABLE_PDF, _f32(target[0]), _f32(target[1])) if target is not None else None
        q_tab = self._engine.cached_table(runtime.TABLE_PDF, _f32(proposal[0]), _f32(proposal[1])) if proposal is not None else None
        split = None if self._literal else _split_weighted(functions)
        if split is not None and ((split[1] is None) != (p_tab is not None) or (split[2] is None) != (q_tab is not None)):
            split = None                                     # a wrapper reads a table the call did not bring (or the reverse): literal
        if split is not None:
            # the reference's importance-sampling call: one weight p / q per sample instead of K evaluations of its text
            q_sampler = q_tab is None and code == runtime.DIST_NORMAL and _is_normal_pdf_text(split[2], p1, p2)
            src = self._weighted_source((split[0], split[1], None if q_sampler else split[2]), k)
            desc = runtime.make_desc(runtime.KIND_INTEGRATE, k, code, weight=True, p_table=p_tab is not None, q_table=q_tab is not None,
                                     guard_endpoints=True, q_sampler=q_sampler)
            mod = self._fitted_module(src, desc, p1, p2, cdf, p_tab, q_tab)
            sums, n_eff = self._engine.integrate(mod, int(n_samples), int(seed), p1, p2, target_threads, cdf=cdf,
                                                 target_pdf=p_tab, proposal_pdf=q_tab)
            return self._result(sums, k, n_eff)
        src = self._source(functions)
        mask = (1 if p_tab is not None else 0) | (2 if q_tab is not None else 0)
        tables = {}
        if p_tab is not None:
            tables["target_pdf"] = p_tab
        if q_tab is not None:
            tables["proposal_pdf"] = q_tab
        if self._literal:
            desc = runtime.make_desc(runtime.KIND_INTEGRATE, k, code, guard_endpoints=True, user_tables=mask)
            mod = self._engine.module(src, desc)
        else:
            desc = runtime.make_desc(runtime.KIND_INTEGRATE, k, code, guard_endpoints=True, user_tables=mask,
                                     moment_family=mask == 0 and k <= 32 and _moment_family(functions))
            # (user functions may look the tables up at any argument, not only at the draw: with user_tables the index clamp stays)
            mod = self._fitted_module(src, desc, p1, p2, cdf, p_tab, q_tab)
        sums, n_eff = self._engine.integrate(mod, int(n_samples), int(seed), p1, p2, target_threads, cdf=cdf, **tables)
        return self._result(sums, k, n_eff)

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
        q_sampler = not self._literal and code == runtime.DIST_NORMAL
        if q_sampler:
            pass                                            # neither the table nor the analytic text: log q from the deviate
        elif proposal_x_table is not None and proposal_log_pdf_table is not None:
            q = self._engine.cached_table(runtime.TABLE_LOGPDF, _f32(proposal_x_table), _f32(proposal_log_pdf_table))
        else:
            src += "\n\n" + _analytic_logpdf("mcx_logpdf_q", code, p1, p2)
            analytic |= 2
        if self._literal:
            desc = runtime.make_desc(runtime.KIND_MCMC, len(functions), code, guard_endpoints=True, logpdf_analytic=analytic)
            mod = self._engine.module(src, desc)
        else:
            # api.py's plan for the same call: a normal proposal's log q is -z^2/2 + const of the deviate the sampler holds
            # (its table -- the reference's 2048 points of the same function -- is not interpolated), cell tables, pads,
            # the workgroup size a small chain count wants
            padded = runtime.mcmc_dispatch_config(int(n_chains), target_threads).total_threads
            hint = runtime.mcmc_block_hint(padded)
            desc = runtime.make_desc(runtime.KIND_MCMC, len(functions), code, guard_endpoints=True, logpdf_analytic=analytic,
                                     q_sampler=q_sampler, block=0 if hint >= 1024 else hint)
            mod = self._fitted_module(src, desc, p1, p2, cdf, t, q)
        sums, n_eff = self._engine.mcmc(mod, int(n_steps), int(n_chains), int(n_burnin), int(seed), p1, p2, t, q,
                                        target_threads=target_threads, cdf=cdf)
        return self._result(sums, len(functions), n_eff)
