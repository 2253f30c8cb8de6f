"""Test infrastructure: an independent Python restatement of libmcx's WGSL -> HIP translator (csrc/mcx_wgsl.cpp,
include/mcx.h: mcx_wgsl_translate). It was the product's translator until the translation moved into libmcx; it stays here so
that tests/test_wgsl_translator.py can hold the C++ translator to it, text for text, on every WGSL string the test-suite knows
(the reference transpiler's recorded output, the reference's recorded `_core` payloads, the hand-written strings of the GPU
tests) in every math mode. Not imported by the package."""
from __future__ import annotations

import re
from typing import List, Optional, Tuple

from wgpu_montecarlo.frontend import TranspilerError

_TOKEN = re.compile(
    r"\s*(?:(//[^\n]*|/\*.*?\*/)|"
    r"(0[xX][0-9a-fA-F]+[iu]?|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fhiu]?)|"
    r"([A-Za-z_][A-Za-z_0-9]*)|"
    r"(->|<<=|>>=|<<|>>|<=|>=|==|!=|&&|\|\||\+=|-=|\*=|/=|%=|&=|\|=|\^=|\+\+|--|[-+*/%<>=!&|^~(){}\[\],;:.@]))",
    re.S,
)

_TYPES = {"f32": "float", "f16": "float", "i32": "int", "u32": "unsigned int", "bool": "bool"}

# the table lookups the reference's importance-sampling wrappers call (python/wgpu_montecarlo/__init__.py:968-974;
# defined by its shader template, src/distribution.rs:181-223): device accessors of a module built with user_tables
_TABLE_CALLS = {"pdf_target_from_table": "mcx_user_pdf_target", "pdf_proposal_from_table": "mcx_user_pdf_proposal"}

_BUILTINS = {
    "abs": "fabsf", "sin": "sinf", "cos": "cosf", "tan": "tanf", "asin": "asinf", "acos": "acosf",
    "atan": "atanf", "atan2": "atan2f", "sinh": "sinhf", "cosh": "coshf", "tanh": "tanhf",
    "asinh": "asinhf", "acosh": "acoshf", "atanh": "atanhf", "sqrt": "sqrtf", "inverseSqrt": "rsqrtf",
    "exp": "expf", "exp2": "exp2f", "log": "logf", "log2": "log2f", "floor": "floorf", "ceil": "ceilf",
    "round": "rintf", "trunc": "truncf", "fract": "mcx_fract", "sign": "mcx_sign", "min": "fminf",
    "max": "fmaxf", "clamp": "mcx_clamp", "mix": "mcx_mix", "step": "mcx_step",
    "smoothstep": "mcx_smoothstep", "pow": "powf", "fma": "fmaf", "saturate": "__saturatef",
    "degrees": "mcx_degrees", "radians": "mcx_radians",
}

# math modes of emit_hip.py applied to the float-only builtins ("precise", the default here, leaves the ocml routines above):
# the hardware exp / log / sqrt, the range-reduced hardware sin / cos / tan and exp2(y log2|x|) for pow (device/mcx_device.hpp)
_MATH_BUILTINS = {
    "precise": {},
    "default": {"sin": "mcx_sin", "cos": "mcx_cos", "tan": "mcx_tan", "sinh": "mcx_sinh", "cosh": "mcx_cosh", "pow": "mcx_pow", "exp": "__expf", "exp2": "__builtin_amdgcn_exp2f",
                "log": "__logf", "log2": "__builtin_amdgcn_logf", "sqrt": "__builtin_amdgcn_sqrtf"},
}
_MATH_BUILTINS["fast"] = dict(_MATH_BUILTINS["default"], sin="__sinf", cos="__cosf", tan="__tanf")

_BINARY_LEVELS: List[Tuple[str, ...]] = [
    ("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">="), ("<<", ">>"),
    ("+", "-"), ("*", "/", "%"),
]


_WHOLE = re.compile(r"^\(*(-?)\(*(\d+)(?:\.0*)?f?\)*$")


def _whole_exponent(text: str) -> Optional[int]:
    """The value of a translated literal such as `2.0f`, `(-3.0f)` or `4` if it is a whole number of at most 64 in magnitude."""
    m = _WHOLE.match(text.replace(" ", ""))
    if not m or text.count("(") != text.count(")"):
        return None
    n = int(m.group(2))
    return None if n > 64 else (-n if m.group(1) else n)


def _var_name(name: str) -> str:
    """A WGSL variable may be called mcx_something: keep it out of the device library's namespace, at every mention."""
    return name + "_v" if name.startswith("mcx_") else name


def _tokenize(text: str) -> List[Tuple[str, str]]:
    tokens: List[Tuple[str, str]] = []
    pos = 0
    text = text.rstrip()
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise TranspilerError(f"WGSL function string: cannot tokenize near '{text[pos:].lstrip()[:20]}'")
        pos = m.end()
        if m.group(1):
            continue
        if m.group(2):
            tokens.append(("num", m.group(2)))
        elif m.group(3):
            tokens.append(("id", m.group(3)))
        else:
            tokens.append(("op", m.group(4)))
    return tokens


def _number(text: str) -> str:
    if text.lower().startswith("0x"):
        return text.rstrip("iu") + ("u" if text.endswith("u") else "")
    suffix = text[-1] if text[-1] in "fhiu" else ""
    body = text[:-1] if suffix else text
    is_float = any(c in body for c in ".eE") or suffix in ("f", "h")
    if is_float:
        if "." not in body and "e" not in body.lower():
            body += ".0"
        return body + "f"
    return body + ("u" if suffix == "u" else "")


class _Parser:
    def __init__(self, tokens, slot: int, math: str = "precise") -> None:
        self.toks = tokens
        self.i = 0
        self.slot = slot
        self.local_functions: List[str] = []
        if math not in _MATH_BUILTINS:
            raise ValueError(f"math must be one of {tuple(_MATH_BUILTINS)}")
        self.builtins = dict(_BUILTINS, **_MATH_BUILTINS[math])

    # ---- token helpers ----
    def peek(self, k: int = 0):
        j = self.i + k
        return self.toks[j] if j < len(self.toks) else ("eof", "")

    def take(self):
        tok = self.peek()
        self.i += 1
        return tok

    def accept(self, value: str) -> bool:
        if self.peek()[1] == value and self.peek()[0] != "num":
            self.i += 1
            return True
        return False

    def expect(self, value: str) -> None:
        if not self.accept(value):
            raise TranspilerError(f"WGSL function string: expected '{value}' but found '{self.peek()[1]}'")

    def ident(self) -> str:
        kind, value = self.take()
        if kind != "id":
            raise TranspilerError(f"WGSL function string: expected an identifier, found '{value}'")
        return value

    def type_name(self) -> str:
        name = self.ident()
        if name not in _TYPES:
            raise TranspilerError(f"WGSL function string: unsupported type '{name}' (scalar f32/i32/u32/bool only)")
        return _TYPES[name]

    def fn_name(self, name: str) -> str:
        return f"mcx_uf{self.slot}_{name}"

    # ---- expressions ----
    def expression(self, level: int = 0) -> str:
        if level == len(_BINARY_LEVELS):
            return self.unary()
        left = self.expression(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in _BINARY_LEVELS[level]:
            op = self.take()[1]
            right = self.expression(level + 1)
            left = f"mcx_mod({left}, {right})" if op == "%" else f"({left} {op} {right})"
        return left

    def _nest(self):
        """The descent is bounded: parentheses, unary operators and blocks nest at most 200 deep."""
        self.nesting = getattr(self, "nesting", 0) + 1
        if self.nesting > 200:
            raise TranspilerError("WGSL function string: nesting deeper than 200 levels")

    def unary(self) -> str:
        self._nest()
        try:
            return self._unary()
        finally:
            self.nesting -= 1

    def _unary(self) -> str:
        if self.peek() == ("op", "-"):
            self.take()
            return f"(-{self.unary()})"
        if self.peek() == ("op", "!"):
            self.take()
            return f"(!{self.unary()})"
        if self.peek() == ("op", "~"):
            self.take()
            return f"(~{self.unary()})"
        return self.primary()

    def call_args(self) -> List[str]:
        args: List[str] = []
        if not self.accept(")"):
            while True:
                args.append(self.expression())
                if self.accept(")"):
                    break
                self.expect(",")
        return args

    def primary(self) -> str:
        kind, value = self.take()
        if kind == "num":
            return _number(value)
        if kind == "op" and value == "(":
            inner = self.expression()
            self.expect(")")
            return f"({inner})"
        if kind != "id":
            raise TranspilerError(f"WGSL function string: unexpected token '{value}'")
        if value in ("true", "false"):
            return value
        if self.accept("("):
            args = self.call_args()
            if value in _TYPES:
                if len(args) != 1:
                    raise TranspilerError(f"WGSL function string: {value}() takes one argument")
                return f"(({_TYPES[value]})({args[0]}))"
            if value == "select":
                if len(args) != 3:
                    raise TranspilerError("WGSL function string: select() takes three arguments")
                return f"(({args[2]}) ? ({args[1]}) : ({args[0]}))"
            if value == "pow" and len(args) == 2 and _whole_exponent(args[1]) is not None:
                # a literal whole exponent (what the reference's transpiler writes for x**k: `pow(x, 2.0)`) is a product
                # chain, as on the Python path (emit_hip.py); the caller supplies McxPowI (emit_hip.prelude())
                n = _whole_exponent(args[1])
                chain = f"McxPowI<{abs(n)}>::of({args[0]})"
                return chain if n >= 0 else f"(1.0f / {chain})"
            if value in self.builtins:
                return f"{self.builtins[value]}({', '.join(args)})"
            if value in _TABLE_CALLS and value not in self.local_functions:
                return f"{_TABLE_CALLS[value]}({', '.join(args)})"
            if value in self.local_functions:
                return f"{self.fn_name(value)}({', '.join(args)})"
            if value.startswith("vec") or value.startswith("mat") or value == "array":
                raise TranspilerError(f"WGSL function string: '{value}' is not supported (scalar code only)")
            # forward reference to a helper defined later in the same string
            return f"{self.fn_name(value)}({', '.join(args)})"
        if self.peek() == ("op", ".") or self.peek() == ("op", "["):
            raise TranspilerError("WGSL function string: member / index access is not supported (scalar code only)")
        return _var_name(value)

    # ---- statements ----
    def block(self, indent: int) -> List[str]:
        self._nest()
        try:
            return self._block(indent)
        finally:
            self.nesting -= 1

    def _block(self, indent: int) -> List[str]:
        self.expect("{")
        out: List[str] = []
        while not self.accept("}"):
            if self.peek()[0] == "eof":
                raise TranspilerError("WGSL function string: unbalanced braces")
            out += self.statement(indent)
        return out

    def simple_statement(self) -> str:
        """let / var / const / assignment / increment / call, without the trailing ';'"""
        kind, value = self.peek()
        if kind == "id" and value in ("let", "var", "const"):
            self.take()
            name = _var_name(self.ident())
            ctype = "auto"
            if self.accept(":"):
                ctype = self.type_name()
            if self.accept("="):
                init = self.expression()
                if ctype == "auto" and re.fullmatch(r"\(?-?\d+\)?", init):
                    ctype = "int"
                prefix = "const " if value == "const" else ""
                return f"{prefix}{ctype} {name} = {init}"
            if ctype == "auto":
                raise TranspilerError("WGSL function string: a declaration needs a type or an initialiser")
            return f"{ctype} {name} = 0"
        called = self.ident()
        if self.accept("("):
            args = self.call_args()
            callee = self.builtins.get(called) or self.fn_name(called)
            return f"{callee}({', '.join(args)})"
        target = _var_name(called)
        kind, op = self.take()
        if op in ("++", "--"):
            return f"{target}{op}"
        if op in ("=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<=", ">>="):
            value_text = self.expression()
            if op == "%=":
                return f"{target} = mcx_mod({target}, {value_text})"
            return f"{target} {op} {value_text}"
        raise TranspilerError(f"WGSL function string: unsupported statement near '{target} {op}'")

    def statement(self, indent: int) -> List[str]:
        pad = "    " * indent
        kind, value = self.peek()
        if kind == "op" and value == "{":
            return [pad + "{"] + self.block(indent + 1) + [pad + "}"]
        if kind == "op" and value == ";":
            self.take()
            return []
        if kind == "id" and value == "return":
            self.take()
            if self.accept(";"):
                return [pad + "return 0.0f;"]
            expr = self.expression()
            self.expect(";")
            return [pad + f"return mcx_b2f({expr});"]
        if kind == "id" and value == "if":
            self.take()
            cond = self.expression()
            out = [pad + f"if ({cond}) {{"] + self.block(indent + 1)
            while self.peek() == ("id", "else"):
                self.take()
                if self.peek() == ("id", "if"):
                    self.take()
                    cond = self.expression()
                    out += [pad + f"}} else if ({cond}) {{"] + self.block(indent + 1)
                else:
                    out += [pad + "} else {"] + self.block(indent + 1)
                    break
            return out + [pad + "}"]
        if kind == "id" and value == "while":
            self.take()
            cond = self.expression()
            return [pad + f"while ({cond}) {{"] + self.block(indent + 1) + [pad + "}"]
        if kind == "id" and value == "loop":
            self.take()
            return [pad + "while (true) {"] + self.block(indent + 1) + [pad + "}"]
        if kind == "id" and value == "for":
            self.take()
            self.expect("(")
            init = "" if self.peek() == ("op", ";") else self.simple_statement()
            self.expect(";")
            cond = "" if self.peek() == ("op", ";") else self.expression()
            self.expect(";")
            step = "" if self.peek() == ("op", ")") else self.simple_statement()
            self.expect(")")
            return [pad + f"for ({init}; {cond}; {step}) {{"] + self.block(indent + 1) + [pad + "}"]
        if kind == "id" and value in ("break", "continue"):
            self.take()
            self.expect(";")
            return [pad + value + ";"]
        text = self.simple_statement()
        self.expect(";")
        return [pad + text + ";"]

    # ---- functions ----
    def function(self, emitted_name: str) -> Tuple[str, str]:
        while self.accept("@"):           # attributes such as @must_use
            self.ident()
            if self.accept("("):
                self.call_args()
        if self.peek() != ("id", "fn"):
            raise TranspilerError("WGSL function string must start with 'fn'")
        self.take()
        original = self.ident()
        self.expect("(")
        params: List[str] = []
        if not self.accept(")"):
            while True:
                pname = _var_name(self.ident())
                self.expect(":")
                params.append(f"{self.type_name()} {pname}")
                if self.accept(")"):
                    break
                self.expect(",")
        rtype = "float"
        if self.accept("->"):
            rtype = self.type_name()
        self.local_functions.append(original)
        body = self.block(1)
        fallback = "    return 0;" if rtype != "void" else ""
        text = f"MCX_DEV {rtype} {emitted_name}({', '.join(params)}) {{\n" + "\n".join(body + [fallback]) + "\n}"
        return original, text


def translate(wgsl: str, slot: int, entry_name: str, math: str = "precise") -> str:
    """Translate one WGSL function string (entry function first, optional helpers after it). `math` selects the
    routines behind exp / log / sqrt / sin / cos / tan / pow as in emit_hip.py; `/` stays the C operator in every mode
    (the translator does not type expressions, and an integer quotient must stay one)."""
    tokens = _tokenize(wgsl)
    if not tokens:
        raise TranspilerError("empty WGSL function string")
    parser = _Parser(tokens, slot, math)
    # pre-scan helper names so that calls are prefixed consistently
    names = [tokens[j + 1][1] for j in range(len(tokens) - 1) if tokens[j] == ("id", "fn") and tokens[j + 1][0] == "id"]
    if not names:
        raise TranspilerError("WGSL function string must contain a function definition ('fn name(...)')")
    parser.local_functions = list(names)
    pieces: List[str] = []
    declarations: List[str] = []
    first = True
    while parser.peek()[0] != "eof":
        if parser.peek() == ("op", ";"):
            parser.take()
            continue
        if len(pieces) >= len(names):                       # more top-level items than `fn name` pairs: the next one is not a function
            raise TranspilerError("WGSL function string must start with 'fn'")
        name_for_emit = entry_name if first else parser.fn_name(names[len(pieces)])
        _, text = parser.function(name_for_emit)
        declarations.append(text.split("{", 1)[0].rstrip() + ";")
        pieces.append(text)
        first = False
    entry_calls = parser.fn_name(names[0])
    # helpers may call the entry by its WGSL name: provide the alias
    alias = ""
    if any(entry_calls + "(" in p for p in pieces):
        sig = declarations[0][:-1]
        params = sig[sig.index("(") + 1: sig.rindex(")")]
        arg_names = ", ".join(p.split()[-1] for p in params.split(",") if p.strip())
        rtype = sig[sig.index(" ") + 1: sig.rindex(" ", 0, sig.index("("))]        # the whole type ("unsigned int"), not its first word
        alias = f"MCX_DEV {rtype} {entry_calls}({params}) {{ return {entry_name}({arg_names}); }}\n"
        declarations.append(f"MCX_DEV {rtype} {entry_calls}({params});")
    return "\n".join(declarations) + "\n" + "\n".join(pieces) + "\n" + alias
