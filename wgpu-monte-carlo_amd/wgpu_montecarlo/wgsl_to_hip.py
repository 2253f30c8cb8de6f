"""Raw WGSL function strings -> HIP C++ device functions: the binding of libmcx's translator (csrc/mcx_wgsl.cpp,
include/mcx.h: mcx_wgsl_translate).

The reference accepts WGSL strings wherever it accepts Python callables (python/wgpu_montecarlo/__init__.py:740-742), and its
Python half hands nothing but WGSL text to its native module: its transpiler's output, user strings, and the importance-sampling
wrappers that call `pdf_target_from_table` / `pdf_proposal_from_table` (:893-905, 968-980). The reference's native half compiles
that text with naga; libmcx translates the scalar subset those sources use to `MCX_DEV` functions and compiles them with hiprtc
into the fused kernels. The translator lives in libmcx so that any binding of the C ABI -- this package, `_core.py`, a Rust or C
host -- hands over the reference's text unchanged; tests/wgsl_reference_translator.py is an independent Python restatement the
test-suite holds it to.
"""
from __future__ import annotations

import ctypes as C

from . import runtime
from .frontend import TranspilerError

MATH_CODES = {"precise": 0, "default": 1, "fast": 2}
E_TRANSLATE = -5


def prelude() -> str:
    """Helpers a translation unit of translated strings needs ahead of them (McxPowI: `pow(x, 2.0)` is a product chain)."""
    lib = runtime.load()
    lib.mcx_wgsl_prelude.restype = C.c_char_p
    return lib.mcx_wgsl_prelude().decode()


def translate(wgsl: str, slot: int, entry_name: str, math: str = "precise") -> str:
    """Translate one WGSL function string (entry function first, optional helpers after it) to HIP: the entry becomes
    `entry_name`, helpers are prefixed `mcx_uf<slot>_`. `math` selects the routines behind exp / log / sqrt / sin / cos / tan /
    pow / sinh / cosh as in emit_hip.py; `/` stays the C operator in every mode, and a literal whole exponent of pow() is a
    product chain in every mode. Raises TranspilerError for text outside the scalar subset."""
    if math not in MATH_CODES:
        raise ValueError(f"math must be one of {tuple(MATH_CODES)}")
    if not isinstance(wgsl, str):
        raise TypeError(f"WGSL function string expected, got {type(wgsl)}")
    lib = runtime.load()
    out = C.c_void_p()
    rc = lib.mcx_wgsl_translate(wgsl.encode(), int(slot), entry_name.encode(), MATH_CODES[math], C.byref(out))
    if rc == E_TRANSLATE:
        raise TranspilerError(runtime.last_error())
    runtime.check(rc)
    try:
        return C.string_at(out).decode()
    finally:
        lib.mcx_free(out)
