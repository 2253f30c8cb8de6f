"""`_core`-level drop-in: the interface of the reference's native module over libmcx.so.

The reference's Python half calls `self._integrator.integrate / integrate_is_tables / integrate_mcmc` on the PyO3
class `_core.MonteCarloIntegrator` (src/lib.rs:17-431; call sites python/wgpu_montecarlo/__init__.py:761-770,
994-1007, 1096-1113) and hands it *WGSL text*. This module offers the same class -- same method names, positional
signatures, defaults, return type (float32[K]) and exception types -- so that the reference's own `__init__.py` and
`transpiler.py` could run unchanged on an MI355X. The WGSL strings (the transpiler's output, user strings and the
importance-sampling wrappers that call `pdf_target_from_table` / `pdf_proposal_from_table`) are planned and translated to
HIP C++ inside libmcx and fused into the same kernels. libmcx has the reference's native object itself -- mcx_core_create /
mcx_core_integrate / mcx_core_mcmc (include/mcx.h, csrc/mcx_core.cpp): an engine, its resident tables found again by content,
its planned and compiled modules found again by payload --; what is left here is the reference's Python argument conventions:
parameter dicts with their silent defaults, numpy tables, exception types, float32[K].

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

import atexit
import ctypes as C
import os
import threading
from typing import Optional

import numpy as np

from . import runtime

_CORES: dict = {}            # (device, math) -> mcx_core handle: every integrator of a process shares the engine, the resident tables and
_CORES_LOCK = threading.Lock()   # the compiled modules of its device (the reference's convenience functions build an integrator per call)
_MATH = {"precise": 0, "default": 1, "fast": 2}
_ENCODED: dict = {}          # tuple of WGSL strings -> the char* array handed to libmcx; bounded, oldest out

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


class CoreTables(C.Structure):
    """include/mcx.h: mcx_core_tables"""
    _fields_ = [("struct_size", C.c_uint32), ("n_cdf", C.c_uint32), ("n_target", C.c_uint32), ("n_proposal", C.c_uint32),
                ("x_table", C.POINTER(C.c_float)), ("cdf_table", C.POINTER(C.c_float)), ("target_x", C.POINTER(C.c_float)),
                ("target_v", C.POINTER(C.c_float)), ("proposal_x", C.POINTER(C.c_float)), ("proposal_v", C.POINTER(C.c_float))]


def _shared_core(device: int, math: str):
    with _CORES_LOCK:
        h = _CORES.get((device, math))
        if h is None:
            lib = runtime.load()
            h = C.c_void_p()
            runtime.check(lib.mcx_core_create(int(device), _MATH[math], C.byref(h)))    # RuntimeError("Failed to initialize GPU: ...")
            _CORES[(device, math)] = h
        return h


@atexit.register
def _destroy_cores():
    with _CORES_LOCK:
        lib = runtime.load() if _CORES else None
        for h in _CORES.values():
            lib.mcx_core_destroy(h)
        _CORES.clear()


class MonteCarloIntegrator:
    """Replaces `_core.MonteCarloIntegrator` (src/lib.rs:17-431)."""

    def __init__(self, device: int = 0, math: Optional[str] = None):
        """`math` (an extension; the reference's constructor takes nothing) picks how the strings are compiled, see the
        module docstring: "default" (the plan this package's API builds for the same call; also what MCX_CORE_MATH unset
        means), "fast", or "precise" (literal, ocml builtins). MCX_CORE_MATH sets it for callers that cannot pass it, such
        as the reference's unmodified __init__.py. BASELINE configs[1..3] through the binding, default / precise: 0.47 /
        0.48, 0.78 / 4.6, 8.2 / 17.6 ms (profiles/r03_core_binding_payloads.txt)."""
        self._math = math if math is not None else os.environ.get("MCX_CORE_MATH", "default")
        if self._math not in _MATH:
            raise ValueError("math must be one of ('precise', 'default', 'fast')")
        self._lib = runtime.load()
        self._declare(self._lib)
        self._core = _shared_core(int(device), self._math)

    @staticmethod
    def _declare(lib):
        fp, u32 = C.POINTER(C.c_float), C.c_uint32
        lib.mcx_core_create.argtypes = [C.c_int, C.c_int32, C.POINTER(C.c_void_p)]
        lib.mcx_core_destroy.argtypes = [C.c_void_p]
        lib.mcx_core_destroy.restype = None
        lib.mcx_core_integrate.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_uint64, u32,
                                           C.POINTER(CoreTables), C.c_int64, fp]
        lib.mcx_core_mcmc.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_float,
                                      C.c_float, u32, u32, u32, u32, C.POINTER(CoreTables), C.c_int64, fp]

    # ---- argument conventions ---------------------------------------------------------------------
    @staticmethod
    def _strings(functions):
        if len(functions) == 0:
            raise ValueError("At least one function is required")          # src/lib.rs:61-65
        try:
            return _ENCODED[tuple(functions)]                              # the reference's Python half sends the same texts call after call
        except (KeyError, TypeError):
            pass
        if not all(isinstance(f, str) for f in functions):
            raise TypeError("functions must be WGSL strings")
        texts = (C.c_char_p * len(functions))(*[f.encode() for f in functions])
        if len(_ENCODED) >= 256:
            _ENCODED.pop(next(iter(_ENCODED)), None)
        _ENCODED[tuple(functions)] = texts
        return texts

    @staticmethod
    def _tables(x_table, cdf_table, target_x, target_v, proposal_x, proposal_v):
        """numpy inputs are taken as contiguous f32 (a non-contiguous array silently becomes empty in the reference,
        `as_slice().unwrap_or(&[])`, src/lib.rs:71-77 -- here it is simply made contiguous); a pair counts only if both halves came."""
        fp = C.POINTER(C.c_float)
        keep = []

        def pair(a, b):
            if a is None or b is None:
                return None, None, 0
            a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
            if a.shape != b.shape or a.ndim != 1:
                raise ValueError("table keys and values must be 1D arrays of the same length")
            keep.extend((a, b))
            return a.ctypes.data_as(fp), b.ctypes.data_as(fp), len(a)

        x, c, n_cdf = pair(x_table, cdf_table)
        tx, tv, n_t = pair(target_x, target_v)
        px, pv, n_p = pair(proposal_x, proposal_v)
        return CoreTables(C.sizeof(CoreTables), n_cdf, n_t, n_p, x, c, tx, tv, px, pv), keep

    def _check(self, rc: int) -> None:
        if rc == -5:                                                       # MCX_E_TRANSLATE
            from .frontend import TranspilerError

            raise TranspilerError(runtime.last_error())
        runtime.check(rc)                                                   # ValueError / RuntimeError as the reference raises them

    def _integrate(self, functions, dist_type, dist_params, n_samples, seed, x_table, cdf_table, tx, tv, px, pv, target_threads):
        texts = self._strings(functions)
        code, p1, p2 = _params(dist_type, dist_params)
        tables, keep = self._tables(x_table, cdf_table, tx, tv, px, pv)
        out = np.zeros(len(functions), dtype=np.float32)
        self._check(self._lib.mcx_core_integrate(self._core, texts, len(functions), code, p1, p2, int(n_samples), int(seed) & 0xFFFFFFFF,
                                                 C.byref(tables), int(target_threads or 0), out.ctypes.data_as(C.POINTER(C.c_float))))
        del keep
        return out

    # ---- src/lib.rs:47-141 ------------------------------------------------------------------------
    def integrate(self, functions, dist_type, dist_params, n_samples, seed, x_table=None, cdf_table=None,
                  target_threads=None) -> np.ndarray:
        return self._integrate(functions, dist_type, dist_params, n_samples, seed, x_table, cdf_table, None, None, None, None, target_threads)

    # ---- src/lib.rs:158-275 -----------------------------------------------------------------------
    def integrate_is_tables(self, functions, dist_type, dist_params, n_samples, seed, x_table=None, cdf_table=None,
                            target_x_table=None, target_pdf_table=None, proposal_x_table=None,
                            proposal_pdf_table=None, target_threads=None) -> np.ndarray:
        return self._integrate(functions, dist_type, dist_params, n_samples, seed, x_table, cdf_table, target_x_table, target_pdf_table,
                               proposal_x_table, proposal_pdf_table, target_threads)

    # ---- src/lib.rs:296-431 -----------------------------------------------------------------------
    def integrate_mcmc(self, functions, proposal_dist_type, proposal_dist_params, target_dist_type, target_dist_params,
                       n_steps, n_chains, n_burnin, seed, x_table=None, cdf_table=None, target_x_table=None,
                       target_log_pdf_table=None, proposal_x_table=None, proposal_log_pdf_table=None,
                       target_threads=None) -> np.ndarray:
        texts = self._strings(functions)
        code, p1, p2 = _params(proposal_dist_type, proposal_dist_params)
        t_code, t1, t2 = _params(target_dist_type, target_dist_params)
        tables, keep = self._tables(x_table, cdf_table, target_x_table, target_log_pdf_table, proposal_x_table, proposal_log_pdf_table)
        out = np.zeros(len(functions), dtype=np.float32)
        self._check(self._lib.mcx_core_mcmc(self._core, texts, len(functions), code, p1, p2, t_code, t1, t2, int(n_steps), int(n_chains),
                                            int(n_burnin), int(seed) & 0xFFFFFFFF, C.byref(tables), int(target_threads or 0),
                                            out.ctypes.data_as(C.POINTER(C.c_float))))
        del keep
        return out
