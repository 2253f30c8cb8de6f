"""IR -> WGSL text.

Not used by the MI355X compute path. It exists so that `transpile_function` / `PythonToWGSL` keep
returning the WGSL strings users of the reference know (formats asserted by the reference's
tests/test_transpiler.py, e.g. `fn step(x: f32, rng: f32) -> f32`, `var a = (x * 2.0)`,
`pow(x, 2.0)`, `select(0.0, 1.0, (x > 0.5))`, `const a: f32 = 1.5;`), and so that WGSL strings
produced here can be fed back to `integrate(...)` like any other raw-WGSL function.
"""
from __future__ import annotations

from typing import List

from . import frontend as ir
from .frontend import TranspilerError

# decimal texts of the reference's CONSTANTS_MAP (transpiler.py:114-126)
_CONST_TEXT = {
    "pi": "3.1415926535897932384626433832795",
    "e": "2.7182818284590452353602874713527",
    "tau": "6.283185307179586476925286766559",
    "euler_gamma": "0.577215664901532860606512090082",
    "inf": "1e300",
    "nan": "nan",
}
_RENAMED_CALLS = {"power": "pow"}


def _expr(node) -> str:
    if isinstance(node, ir.Num):
        return str(float(node.value))
    if isinstance(node, ir.BoolLit):
        return "true" if node.value else "false"
    if isinstance(node, ir.Var):
        return node.name
    if isinstance(node, ir.NamedConst):
        return _CONST_TEXT[node.key[1]]
    if isinstance(node, ir.Bin):
        return f"({_expr(node.left)} {node.op} {_expr(node.right)})"
    if isinstance(node, ir.Pow):
        return f"pow({_expr(node.base)}, {_expr(node.exponent)})"
    if isinstance(node, ir.Unary):
        return f"({node.op}{_expr(node.operand)})"
    if isinstance(node, ir.Cmp):
        return f"({_expr(node.left)} {node.op} {_expr(node.right)})"
    if isinstance(node, ir.Logic):
        return "(" + f" {node.op} ".join(_expr(v) for v in node.values) + ")"
    if isinstance(node, ir.Select):
        return f"select({_expr(node.orelse)}, {_expr(node.body)}, {_expr(node.test)})"
    if isinstance(node, ir.Call):
        callee = _RENAMED_CALLS.get(node.name, node.name)
        return f"{callee}({', '.join(_expr(a) for a in node.args)})"
    raise TranspilerError(f"Unsupported expression type: {type(node).__name__}")


def _stmts(stmts) -> List[str]:
    out: List[str] = []
    for st in stmts:
        if isinstance(st, ir.Return):
            if st.value is None:
                out.append("return;")
            elif st.boolean:
                out.append(f"return select(0.0, 1.0, {_expr(st.value)});")
            else:
                out.append(f"return {_expr(st.value)};")
        elif isinstance(st, ir.Assign):
            out.append((f"var {st.name} = " if st.declares else f"{st.name} = ") + _expr(st.value) + ";")
        elif isinstance(st, ir.If):
            out.append(f"if ({_expr(st.test)}) {{")
            out += ["    " + line for line in _stmts(st.body)]
            if st.orelse:
                out.append("} else {")
                out += ["    " + line for line in _stmts(st.orelse)]
            out.append("}")
        elif isinstance(st, ir.While):
            out.append(f"while ({_expr(st.test)}) {{")
            out += ["    " + line for line in _stmts(st.body)]
            out.append("}")
        elif isinstance(st, ir.ExprStmt):
            out.append(_expr(st.value) + ";")
        else:
            raise TranspilerError(f"Unsupported statement type: {type(st).__name__}")
    return out


def emit_function(fn: ir.Function) -> str:
    signature = ", ".join(f"{p}: f32" for p in fn.params)
    lines = [f"const {name}: f32 = {value};" for name, value in fn.consts.items()]
    lines += _stmts(fn.body)
    return f"fn {fn.name}({signature}) -> f32 {{\n    " + "\n    ".join(lines) + "\n}"
