"""IR -> HIP C++ device functions for the fused gfx950 kernels (the product path).

Replaces the WGSL the reference hands to `_core` (python/wgpu_montecarlo/transpiler.py output +
shader_gen.rs:229-261 renaming). Each user function becomes

    __device__ __forceinline__ float user_func_<i>(float x) { ... }

and is inlined into the kernel skeleton (csrc/device/mcx_kernels.hpp) by hiprtc.

Device semantics kept from the reference's WGSL:
  `%` is truncated (fmodf), `round` is half-to-even (rintf), `x**k` has C powf semantics for negative
  bases; small integer literal exponents become a shared multiply chain (mcx_powi<N>) so that K fused
  moments x**1..x**K cost K-1 multiplies in total after common-subexpression elimination.
"""
from __future__ import annotations

import math
from typing import List

from . import frontend as ir
from .frontend import TranspilerError

_PRELUDE_HELPERS = """
template <int N> struct McxPowI {
    static MCX_DEV float of(float x) {
        if constexpr (N == 0) return 1.0f;
        else if constexpr (N == 1) return x;
        else if constexpr (N % 2 == 0) { float h = McxPowI<N / 2>::of(x); return h * h; }
        else return McxPowI<N - 1>::of(x) * x;
    }
};
"""

# python/WGSL-side name -> HIP expression template
_FUNCS = {
    "abs": "fabsf({0})", "sin": "sinf({0})", "cos": "cosf({0})", "tan": "tanf({0})",
    "asin": "asinf({0})", "acos": "acosf({0})", "atan": "atanf({0})",
    "sinh": "sinhf({0})", "cosh": "coshf({0})", "tanh": "tanhf({0})",
    "sqrt": "sqrtf({0})", "exp": "expf({0})", "exp2": "exp2f({0})", "log": "logf({0})", "log2": "log2f({0})",
    "floor": "floorf({0})", "ceil": "ceilf({0})", "round": "rintf({0})", "trunc": "truncf({0})",
    "fract": "mcx_fract({0})", "sign": "mcx_sign({0})",
    "min": "fminf({0}, {1})", "max": "fmaxf({0}, {1})", "clamp": "mcx_clamp({0}, {1}, {2})",
    "mix": "mcx_mix({0}, {1}, {2})", "step": "mcx_step({0}, {1})", "smoothstep": "mcx_smoothstep({0}, {1}, {2})",
    "pow": "powf({0}, {1})", "power": "powf({0}, {1})",
}
_ARITY = {"min": 2, "max": 2, "clamp": 3, "mix": 3, "step": 2, "smoothstep": 3, "pow": 2, "power": 2}

# math modes (MonteCarloIntegrator(math=...)):
#   "precise": ocml functions and IEEE division everywhere.
#   "default": exp / log / sqrt / division use the hardware instructions (v_exp_f32, v_log_f32, v_sqrt_f32,
#              v_rcp_f32: 1-2 ulp, exp degrading like 2|x| ulp); sin / cos / tan use v_sin_f32 / v_cos_f32 behind a
#              two-constant argument reduction (absolute error <= 4e-7 for |x| < 1e6, ocml beyond: device/mcx_device.hpp
#              mcx_sin) and pow is exp2(y * log2|x|) with powf's sign and NaN rules (relative error about
#              1.2e-7 * (1 + |y log2 x|), mcx_pow); sinh / cosh are (exp(x) -+ exp(-x)) / 2 on v_exp_f32 (mcx_sinh). All of it is inside the accuracy WGSL itself promises for these
#              operations (division 2.5 ulp, exp 3 + 2|x| ulp, log 3 ulp, sqrt via inverseSqrt 2 ulp, sin / cos 2^-11
#              absolute on [-pi, pi], pow "as exp2(y * log2 x)"), i.e. inside what the reference's own backends deliver.
#              Measured: profiles/r03_trig_pow_accuracy.txt.
#   "fast":    sin / cos / tan as __sinf / __cosf / __tanf: one multiply and the instruction, the error grows
#              like 7e-8 * |x| and the argument range is |x| < ~1600.
_NATIVE_FUNCS = {
    "exp": "__expf({0})", "exp2": "__builtin_amdgcn_exp2f({0})", "log": "__logf({0})",
    "log2": "__builtin_amdgcn_logf({0})", "sqrt": "__builtin_amdgcn_sqrtf({0})",
}
_DEFAULT_FUNCS = dict(_NATIVE_FUNCS, sin="mcx_sin({0})", cos="mcx_cos({0})", tan="mcx_tan({0})", sinh="mcx_sinh({0})", cosh="mcx_cosh({0})",
                      pow="mcx_pow({0}, {1})", power="mcx_pow({0}, {1})")
_FAST_FUNCS = dict(_DEFAULT_FUNCS, sin="__sinf({0})", cos="__cosf({0})", tan="__tanf({0})")
MATH_MODES = ("precise", "default", "fast")

_CONST_VALUES = {
    "pi": math.pi, "e": math.e, "tau": math.tau, "euler_gamma": 0.5772156649015329,
    "inf": math.inf, "nan": math.nan,
}

_CXX_RESERVED = {
    "alignas", "alignof", "and", "asm", "auto", "bool", "break", "case", "catch", "char", "class", "const",
    "continue", "default", "delete", "do", "double", "else", "enum", "explicit", "export", "extern", "float",
    "for", "friend", "goto", "if", "inline", "int", "long", "mutable", "namespace", "new", "not", "operator",
    "or", "private", "protected", "public", "register", "return", "short", "signed", "sizeof", "static",
    "struct", "switch", "template", "this", "throw", "try", "typedef", "typename", "union", "unsigned",
    "using", "virtual", "void", "volatile", "while", "xor", "true", "false", "nullptr",
}


def _ident(name: str) -> str:
    if name in _CXX_RESERVED or name.startswith("mcx_") or name.startswith("__"):
        return name + "_v"
    return name


def float_literal(value: float) -> str:
    if math.isnan(value):
        return "NAN"
    if math.isinf(value):
        return "INFINITY" if value > 0 else "(-INFINITY)"
    text = repr(float(value))
    if "e" not in text and "." not in text:
        text += ".0"
    return text + "f"


def _mode(math) -> str:
    if math is True:
        return "fast"
    if math is False or math is None:
        return "default"
    if math not in MATH_MODES:
        raise ValueError(f"math must be one of {MATH_MODES}")
    return math


class _HipPrinter:
    def __init__(self, math, consts=None) -> None:
        self.consts = consts or {}
        self.mode = _mode(math)
        self.table = {"precise": {}, "default": _DEFAULT_FUNCS, "fast": _FAST_FUNCS}[self.mode]

    def expr(self, node) -> str:
        if isinstance(node, ir.Num):
            return float_literal(node.value)
        if isinstance(node, ir.BoolLit):
            return "true" if node.value else "false"
        if isinstance(node, ir.Var):
            return _ident(node.name)
        if isinstance(node, ir.NamedConst):
            return float_literal(_CONST_VALUES[node.key[1]])
        if isinstance(node, ir.Bin):
            left, right = self.expr(node.left), self.expr(node.right)
            if node.op == "%":
                return f"fmodf({left}, {right})"
            if node.op == "/" and self.mode != "precise":
                return f"mcx_div({left}, {right})"
            return f"({left} {node.op} {right})"
        if isinstance(node, ir.Pow):
            base = self.expr(node.base)
            exponent = node.exponent
            sign = 1.0
            if isinstance(exponent, ir.Unary) and exponent.op == "-" and isinstance(exponent.operand, (ir.Num, ir.Var)):
                exponent, sign = exponent.operand, -1.0
            if isinstance(exponent, ir.Var) and exponent.name in self.consts:
                # a captured constant (closure / global / bound default) is as good as a literal
                exponent = ir.Num(self.consts[exponent.name])
                if exponent.value < 0:
                    exponent, sign = ir.Num(-exponent.value), -sign
            if isinstance(exponent, ir.Num) and float(exponent.value).is_integer() and 0 <= exponent.value <= 64:
                n = int(exponent.value)
                chain = f"McxPowI<{n}>::of({base})"
                if sign > 0:
                    return chain
                return f"(1.0f / {chain})" if self.mode == "precise" else f"mcx_div(1.0f, {chain})"
            return self.table.get("pow", _FUNCS["pow"]).format(base, self.expr(node.exponent))
        if isinstance(node, ir.Unary):
            return f"({node.op}{self.expr(node.operand)})"
        if isinstance(node, ir.Cmp):
            return f"({self.expr(node.left)} {node.op} {self.expr(node.right)})"
        if isinstance(node, ir.Logic):
            return "(" + f" {node.op} ".join(self.expr(v) for v in node.values) + ")"
        if isinstance(node, ir.Select):
            return f"(({self.expr(node.test)}) ? mcx_b2f({self.expr(node.body)}) : mcx_b2f({self.expr(node.orelse)}))"
        if isinstance(node, ir.Call):
            template = self.table.get(node.name) or _FUNCS.get(node.name)
            if template is None:
                raise TranspilerError(
                    f"Unsupported function call: {node.name}. Supported functions: {', '.join(sorted(_FUNCS))}"
                )
            want = _ARITY.get(node.name, 1)
            if len(node.args) != want:
                raise TranspilerError(f"{node.name}() takes {want} argument(s), got {len(node.args)}")
            return template.format(*[self.expr(a) for a in node.args])
        raise TranspilerError(f"Unsupported expression type: {type(node).__name__}")

    def block(self, stmts, indent: int, hoisted) -> List[str]:
        pad = "    " * indent
        out: List[str] = []
        for st in stmts:
            if isinstance(st, ir.Return):
                if st.value is None:
                    out.append(f"{pad}return 0.0f;")
                else:
                    out.append(f"{pad}return mcx_b2f({self.expr(st.value)});")
            elif isinstance(st, ir.Assign):
                out.append(f"{pad}{_ident(st.name)} = mcx_b2f({self.expr(st.value)});")
            elif isinstance(st, ir.If):
                out.append(f"{pad}if ({self.expr(st.test)}) {{")
                out += self.block(st.body, indent + 1, hoisted)
                if st.orelse:
                    out.append(f"{pad}}} else {{")
                    out += self.block(st.orelse, indent + 1, hoisted)
                out.append(f"{pad}}}")
            elif isinstance(st, ir.While):
                out.append(f"{pad}while ({self.expr(st.test)}) {{")
                out += self.block(st.body, indent + 1, hoisted)
                out.append(f"{pad}}}")
            elif isinstance(st, ir.ExprStmt):
                out.append(f"{pad}(void)({self.expr(st.value)});")
            else:
                raise TranspilerError(f"Unsupported statement type: {type(st).__name__}")
        return out


def _assigned_names(stmts, acc: List[str]) -> None:
    for st in stmts:
        if isinstance(st, ir.Assign) and st.name not in acc:
            acc.append(st.name)
        elif isinstance(st, ir.If):
            _assigned_names(st.body, acc)
            _assigned_names(st.orelse, acc)
        elif isinstance(st, ir.While):
            _assigned_names(st.body, acc)


def emit_function(fn: ir.Function, name: str, math="default") -> str:
    """One IR function as a HIP device function called `name` (math: "precise" | "default" | "fast")."""
    printer = _HipPrinter(math, fn.consts)
    params = ", ".join(f"float {_ident(p)}" for p in fn.params)
    lines = [f"MCX_DEV float {name}({params}) {{"]
    for cname, cvalue in fn.consts.items():
        if cname in fn.params:
            continue
        lines.append(f"    const float {_ident(cname)} = {float_literal(cvalue)};")
    local_names: List[str] = []
    _assigned_names(fn.body, local_names)
    # Python locals are function-scoped: declare them once at the top (parameters are already declared)
    for lname in local_names:
        if lname not in fn.params and lname not in fn.consts:
            lines.append(f"    float {_ident(lname)} = 0.0f;")
    lines += printer.block(fn.body, 1, local_names)
    lines.append("    return 0.0f;")
    lines.append("}")
    return "\n".join(lines)


def prelude() -> str:
    """Helpers every emitted translation unit needs (placed after mcx_device.hpp)."""
    return _PRELUDE_HELPERS
