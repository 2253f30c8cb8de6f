"""`transpile_function` / `PythonToWGSL` (API compatibility) and their HIP counterparts."""
from __future__ import annotations

from typing import Callable

from . import emit_hip, emit_wgsl, frontend
from .frontend import TranspilerError  # noqa: F401  (re-export)


class PythonToWGSL:
    """Python callable -> WGSL text (same output format as the reference's class of this name)."""

    def transpile(self, func: Callable) -> str:
        return emit_wgsl.emit_function(frontend.lower(func))


class PythonToHIP:
    """Python callable -> HIP C++ device function (what the MI355X kernels are built from)."""

    def __init__(self, math="default"):
        self.math = math

    def transpile(self, func: Callable, name: str = "user_func_0") -> str:
        return emit_hip.emit_function(frontend.lower(func), name, self.math)


def transpile_function(func: Callable) -> str:
    """WGSL text of `func` (reference: transpiler.py:808-811)."""
    return PythonToWGSL().transpile(func)


def transpile_function_hip(func: Callable, name: str = "user_func_0", math="default", fast_math: bool = False) -> str:
    """HIP C++ text of `func` (math: "precise" | "default" | "fast")."""
    return PythonToHIP("fast" if fast_math else math).transpile(func, name)
