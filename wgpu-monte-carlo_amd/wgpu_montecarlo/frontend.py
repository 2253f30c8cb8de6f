"""Function front-end: a restricted-Python callable -> a small typed IR.

The IR is printed twice: as HIP C++ device functions for the fused gfx950 kernels (emit_hip.py, the
product path) and as WGSL text (emit_wgsl.py, kept so that `transpile_function` stays a drop-in).

The accepted / rejected subset and the error texts follow the reference's transpiler
(/root/reference/python/wgpu_montecarlo/transpiler.py), because the reject set is
behaviour: `integrate_importance_sampling` decides between the analytic and the table path by whether
the PDF closures are accepted (reference __init__.py:825-864). Notable accidents kept on purpose:
  * free-variable capture excludes `dir(dict)` names, not the real builtins (transpiler.py:252-257), so
    `abs(x)` is "Undefined variable(s): abs" while a closure variable called `min` is captured;
  * a docstring or an augmented assignment makes a function non-transpilable;
  * unknown call names pass through to WGSL unchanged (the HIP printer rejects them instead).
"""
from __future__ import annotations

import ast
import inspect
import linecache
import textwrap
import uuid
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Set, Tuple


class TranspilerError(Exception):
    """Raised when a function is outside the supported subset."""


# ------------------------------------------------------------------------------------------------
# IR
# ------------------------------------------------------------------------------------------------
@dataclass
class Num:
    value: float


@dataclass
class BoolLit:
    value: bool


@dataclass
class Var:
    name: str


@dataclass
class NamedConst:
    """math.pi / numpy.e ... -- keeps the reference's decimal text for the WGSL printer."""
    key: Tuple[str, str]


@dataclass
class Bin:
    op: str          # + - * / %
    left: object
    right: object


@dataclass
class Pow:
    base: object
    exponent: object


@dataclass
class Unary:
    op: str          # - +
    operand: object


@dataclass
class Cmp:
    op: str          # > < >= <= == !=
    left: object
    right: object


@dataclass
class Logic:
    op: str          # && ||
    values: List[object]


@dataclass
class Select:
    """`body if test else orelse`"""
    test: object
    body: object
    orelse: object


@dataclass
class Call:
    name: str        # resolved python-side name (e.g. "sin", "power", or an unknown pass-through name)
    args: List[object]


@dataclass
class Return:
    value: Optional[object]
    boolean: bool = False     # top node was a comparison / and-or: wrapped as select(0.0, 1.0, ..)


@dataclass
class Assign:
    name: str
    value: object
    declares: bool            # first assignment of this name


@dataclass
class If:
    test: object
    body: List[object]
    orelse: List[object]


@dataclass
class While:
    test: object
    body: List[object]


@dataclass
class ExprStmt:
    value: object


@dataclass
class Function:
    name: str
    params: List[str]
    consts: Dict[str, float] = field(default_factory=dict)   # captured free variables, insertion ordered
    body: List[object] = field(default_factory=list)


_BIN_OPS = {"Add": "+", "Sub": "-", "Mult": "*", "Div": "/", "Mod": "%"}
_CMP_OPS = {"Gt": ">", "Lt": "<", "GtE": ">=", "LtE": "<=", "Eq": "==", "NotEq": "!="}

# names the reference maps to WGSL builtins (transpiler.py:82-112)
KNOWN_FUNCTIONS = (
    "abs sin cos tan asin acos atan sinh cosh tanh sqrt exp exp2 log log2 floor ceil round trunc fract "
    "sign min max clamp mix step smoothstep pow power"
).split()

# (module, attribute) pairs usable as constants (transpiler.py:114-126)
CONSTANT_KEYS = {
    ("math", "pi"), ("math", "e"), ("math", "tau"), ("math", "inf"), ("math", "nan"),
    ("numpy", "pi"), ("numpy", "e"), ("numpy", "tau"), ("numpy", "euler_gamma"), ("numpy", "inf"),
    ("numpy", "nan"),
}

_DEFAULT_MODULE_ALIASES = {"np": "numpy", "numpy": "numpy", "math": "math"}

# the reference's "builtins" exclusion is dir() of the module-level __builtins__, which inside an
# imported module is a plain dict
_EXCLUDED_AS_BUILTINS = set(dir(dict))


# ------------------------------------------------------------------------------------------------
# source recovery
# ------------------------------------------------------------------------------------------------
_pinned_files: Set[str] = set()


def _pin_source(func: Callable) -> None:
    """Keep the first-seen text of the defining file in linecache so later edits do not shift lines."""
    try:
        path = inspect.getsourcefile(func)
    except TypeError:
        return
    if not path or path in _pinned_files:
        return
    _pinned_files.add(path)
    if path in linecache.cache:
        return
    try:
        with open(path, "r") as fh:
            lines = fh.readlines()
    except OSError:
        return
    linecache.cache[path] = (sum(len(l) for l in lines), None, lines, path)


_FILE_TREES: Dict[str, tuple] = {}       # path -> (mtime, tree)


def _file_tree(func: Callable) -> Optional[ast.AST]:
    try:
        path = inspect.getsourcefile(func)
    except TypeError:
        return None
    if not path:
        return None
    try:
        import os

        mtime = os.stat(path).st_mtime_ns
        hit = _FILE_TREES.get(path)
        if hit is not None and hit[0] == mtime:
            return hit[1]
        with open(path, "r") as fh:
            tree = ast.parse(fh.read())
        if len(_FILE_TREES) > 64:
            _FILE_TREES.clear()
        _FILE_TREES[path] = (mtime, tree)
        return tree
    except (OSError, SyntaxError, ValueError):
        return None


def _parse_fragment(text: str) -> Optional[ast.AST]:
    """inspect.getsource of a lambda may return a fragment such as
    `[lambda x: x, lambda x: x**2], dist, n_samples=10)`; find a parseable prefix."""
    stripped = text.strip()
    for opener, closer in (("[", "]"), ("(", ")")):
        if stripped.startswith(opener):
            depth = 0
            for pos, ch in enumerate(stripped):
                if ch == opener:
                    depth += 1
                elif ch == closer:
                    depth -= 1
                    if depth == 0:
                        try:
                            return ast.parse(stripped[: pos + 1])
                        except SyntaxError:
                            break
    try:
        return ast.parse("__mcx_fragment__ = " + stripped)
    except SyntaxError:
        pass
    # last resort: cut at each "lambda" keyword and take the longest prefix that parses as an expression
    found = stripped.find("lambda")
    while found != -1:
        tail = stripped[found:]
        for end in range(len(tail), 6, -1):
            for cut in (tail[:end].rstrip(", \n"), tail[:end].rstrip(",) \n]")):     # keep the brackets the lambda itself closes
                try:
                    return ast.parse(cut, mode="exec")
                except SyntaxError:
                    continue
        found = stripped.find("lambda", found + 1)
    return None


def _code_signature(code) -> tuple:
    consts = tuple(c for c in code.co_consts if not hasattr(c, "co_code"))
    return (code.co_code, consts, code.co_names, code.co_varnames, code.co_freevars)


def _pick_lambda(candidates: List[ast.Lambda], func: Callable, margin: int = 0) -> ast.Lambda:
    """Several lambdas share the source line(s): find the one whose compiled body is `func`.

    The reference needs Python >= 3.11 (co_positions) for this; comparing compiled code objects works
    on every version."""
    target = func.__code__
    if hasattr(target, "co_positions"):          # Python >= 3.11: the body's start column identifies it
        column = None
        for pos in target.co_positions():
            if pos[2] is not None and pos[2] > 0:
                column = pos[2]
                break
        if column is not None:
            column -= margin
            return min(candidates, key=lambda lam: abs(getattr(lam.body, "col_offset", 0) - column))
    target_sig = _code_signature(target)
    names_only = (target.co_names, target.co_varnames)
    loose = None
    for lam in candidates:
        try:
            expr = ast.Expression(body=lam)
            ast.fix_missing_locations(expr)
            outer = compile(expr, "<mcx-lambda-match>", "eval")
        except (SyntaxError, ValueError, TypeError):
            continue
        inner = [c for c in outer.co_consts if hasattr(c, "co_code")]
        if not inner:
            continue
        code = inner[0]
        # free variables compile as globals outside their closure: compare the parts that survive
        if _code_signature(code) == target_sig:
            return lam
        if len(code.co_code) == len(target.co_code) and (
            set(code.co_names) | set(code.co_freevars)
        ) == (set(target.co_names) | set(target.co_freevars)) and code.co_varnames == target.co_varnames:
            same_consts = tuple(c for c in code.co_consts if not hasattr(c, "co_code")) == tuple(
                c for c in target.co_consts if not hasattr(c, "co_code"))
            if same_consts and loose is None:
                loose = lam
    if loose is not None:
        return loose
    # fall back to argument names + body shape
    for lam in candidates:
        if tuple(a.arg for a in lam.args.args) == names_only[1][: len(lam.args.args)]:
            return lam
    raise TranspilerError(
        "Multiple lambdas on the same line detected and none matches the compiled function. "
        "Please define each lambda on a separate line."
    )


# ------------------------------------------------------------------------------------------------
# lowering
# ------------------------------------------------------------------------------------------------
class _Lowering:
    def __init__(self) -> None:
        self.imports: Dict[str, str] = {}                  # alias -> "module.name"
        self.module_aliases: Dict[str, str] = dict(_DEFAULT_MODULE_ALIASES)
        self.declared: Set[str] = set()

    # ---- imports (function-local first, then the whole defining file) -------------------------
    def scan_imports(self, tree: ast.AST, overwrite: bool) -> None:
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                for alias in node.names:
                    bound = alias.asname or alias.name
                    if overwrite or bound not in self.module_aliases:
                        self.module_aliases[bound] = alias.name
            elif isinstance(node, ast.ImportFrom):
                module = node.module or ""
                for alias in node.names:
                    bound = alias.asname or alias.name
                    if overwrite or bound not in self.imports:
                        self.imports[bound] = f"{module}.{alias.name}"

    # ---- free variables ------------------------------------------------------------------------
    def capture(self, func: Callable, params: Set[str], used: Set[str], assigned: Set[str]) -> Dict[str, float]:
        skip = params | assigned | set(self.imports) | set(self.module_aliases) | _EXCLUDED_AS_BUILTINS
        cells: Dict[str, object] = {}
        if func.__closure__:
            for name, cell in zip(func.__code__.co_freevars, func.__closure__):
                try:
                    cells[name] = cell.cell_contents
                except ValueError:      # empty cell
                    pass
        captured: Dict[str, float] = {}
        missing: List[str] = []
        for name in used - skip:
            if name in cells:
                value = cells[name]
            elif name in func.__globals__:
                value = func.__globals__[name]
            else:
                value = None
            if value is None:
                missing.append(name)
            elif callable(value):
                continue
            elif isinstance(value, bool):
                captured[name] = 1.0 if value else 0.0
            elif isinstance(value, (int, float)):
                captured[name] = float(value)
            else:
                raise TranspilerError(
                    f"Unsupported external variable type for '{name}': {type(value).__name__}. "
                    f"Only int, float, and bool are supported."
                )
        if missing:
            raise TranspilerError(
                f"Undefined variable(s): {', '.join(missing)}. "
                f"Variables must be defined in global scope, imported, or passed as parameters."
            )
        return captured

    # ---- expressions ---------------------------------------------------------------------------
    def expr(self, node: ast.AST):
        if isinstance(node, ast.Name):
            full = self.imports.get(node.id)
            if full is not None:
                parts = full.split(".")
                if len(parts) == 2 and (parts[0], parts[1]) in CONSTANT_KEYS:
                    return NamedConst((parts[0], parts[1]))
            return Var(node.id)
        if isinstance(node, ast.Constant):
            if isinstance(node.value, bool):
                return BoolLit(node.value)
            if isinstance(node.value, (int, float)):
                return Num(float(node.value))
            raise TranspilerError(f"Unsupported constant type: {type(node.value)}")
        if isinstance(node, ast.BinOp):
            left, right = self.expr(node.left), self.expr(node.right)
            kind = type(node.op).__name__
            if kind == "Pow":
                return Pow(left, right)
            if kind not in _BIN_OPS:
                raise TranspilerError(f"Unsupported binary operator: {kind}")
            return Bin(_BIN_OPS[kind], left, right)
        if isinstance(node, ast.UnaryOp):
            operand = self.expr(node.operand)
            kind = type(node.op).__name__
            if kind == "USub":
                return Unary("-", operand)
            if kind == "UAdd":
                return Unary("+", operand)
            raise TranspilerError(f"Unsupported unary operator: {kind}")
        if isinstance(node, ast.Call):
            return self.call(node)
        if isinstance(node, ast.IfExp):
            return Select(self.expr(node.test), self.expr(node.body), self.expr(node.orelse))
        if isinstance(node, ast.Compare):
            if len(node.ops) != 1 or len(node.comparators) != 1:
                raise TranspilerError("Only simple comparisons supported (e.g., x > y)")
            left, right = self.expr(node.left), self.expr(node.comparators[0])
            kind = type(node.ops[0]).__name__
            if kind not in _CMP_OPS:
                raise TranspilerError(f"Unsupported comparison operator: {kind}")
            return Cmp(_CMP_OPS[kind], left, right)
        if isinstance(node, ast.BoolOp):
            values = [self.expr(v) for v in node.values]
            if isinstance(node.op, ast.And):
                return Logic("&&", values)
            if isinstance(node.op, ast.Or):
                return Logic("||", values)
            raise TranspilerError(f"Unsupported boolean operator: {type(node.op).__name__}")
        if isinstance(node, ast.Attribute):
            return self.attribute(node)
        raise TranspilerError(f"Unsupported expression type: {type(node).__name__}")

    def _supported_modules(self) -> str:
        mods = set(self.module_aliases.values()) | {v.split(".")[0] for v in self.imports.values()}
        return ", ".join(sorted(mods))

    def attribute(self, node: ast.Attribute):
        if isinstance(node.value, ast.Name):
            owner = node.value.id
            if owner in self.module_aliases:
                key = (self.module_aliases[owner], node.attr)
                if key in CONSTANT_KEYS:
                    return NamedConst(key)
                raise TranspilerError(
                    f"Unknown constant: {owner}.{node.attr}. "
                    f"Available constants: {', '.join(f'{m}.{c}' for m, c in sorted(CONSTANT_KEYS))}"
                )
            if owner in self.imports:
                key = (self.imports[owner].split(".")[-1], node.attr)
                if key in CONSTANT_KEYS:
                    return NamedConst(key)
            else:
                raise TranspilerError(
                    f"Unsupported module: {owner}. Supported modules: {self._supported_modules()}"
                )
        raise TranspilerError(f"Unsupported attribute access: {node.attr}")

    def call(self, node: ast.Call):
        target = node.func
        if isinstance(target, ast.Name):
            name = target.id
            if name in self.imports:
                name = self.imports[name].split(".")[-1]
        elif isinstance(target, ast.Attribute):
            if not isinstance(target.value, ast.Name):
                raise TranspilerError(f"Unsupported attribute access: {target.attr}")
            owner = target.value.id
            if owner in self.module_aliases:
                name = target.attr
            elif owner in self.imports:
                name = self.imports[owner].split(".")[-1]
            else:
                raise TranspilerError(
                    f"Unsupported module: {owner}. Supported modules: {self._supported_modules()}"
                )
        else:
            raise TranspilerError("Unsupported function call")
        return Call(name, [self.expr(a) for a in node.args])

    # ---- statements ----------------------------------------------------------------------------
    def stmt(self, node: ast.AST):
        if isinstance(node, ast.Return):
            if node.value is None:
                return Return(None)
            return Return(self.expr(node.value), boolean=isinstance(node.value, (ast.Compare, ast.BoolOp)))
        if isinstance(node, ast.Assign):
            if len(node.targets) != 1:
                raise TranspilerError("Multiple assignment targets not supported")
            target = node.targets[0]
            if not isinstance(target, ast.Name):
                raise TranspilerError("Only simple variable assignment supported")
            value = self.expr(node.value)
            first = target.id not in self.declared
            self.declared.add(target.id)
            return Assign(target.id, value, first)
        if isinstance(node, ast.If):
            test = self.expr(node.test)
            return If(test, [self.stmt(s) for s in node.body], [self.stmt(s) for s in node.orelse])
        if isinstance(node, ast.For):
            raise TranspilerError("For loops not yet implemented")
        if isinstance(node, ast.While):
            test = self.expr(node.test)
            return While(test, [self.stmt(s) for s in node.body])
        if isinstance(node, ast.Expr):
            return ExprStmt(self.expr(node.value))
        raise TranspilerError(f"Unsupported statement type: {type(node).__name__}")


def _names_in(nodes) -> Tuple[Set[str], Set[str]]:
    used: Set[str] = set()
    assigned: Set[str] = set()
    for root in nodes:
        for node in ast.walk(root):
            if isinstance(node, ast.Name):
                used.add(node.id)
            elif isinstance(node, ast.Assign):
                for tgt in node.targets:
                    if isinstance(tgt, ast.Name):
                        assigned.add(tgt.id)
                    elif isinstance(tgt, ast.Tuple):
                        assigned.update(e.id for e in tgt.elts if isinstance(e, ast.Name))
            elif isinstance(node, ast.For) and isinstance(node.target, ast.Name):
                assigned.add(node.target.id)
    return used, assigned


class _Lowered:
    """Everything about a function that is fixed by its code object; captured constants are not."""

    __slots__ = ("name", "params", "body", "used", "assigned", "imports", "module_aliases")

    def __init__(self, name, params, body, used, assigned, imports, module_aliases):
        self.name, self.params, self.body = name, params, body
        self.used, self.assigned = used, assigned
        self.imports, self.module_aliases = imports, module_aliases


_LOWER_CACHE: Dict[object, object] = {}       # code object -> _Lowered | TranspilerError
_LOWER_CACHE_LIMIT = 4096


def lower(func: Callable, bind_defaults: bool = False) -> Function:
    """Lower a Python function or lambda to the IR (raises TranspilerError outside the subset).

    bind_defaults=True (the HIP path) turns trailing parameters that have default values into captured
    constants -- `[lambda x, k=k: x**k for k in range(1, 5)]` works -- where the reference emits a
    two-parameter WGSL function that fails at shader compilation.

    The structural part (source recovery, parsing, import analysis, IR) is memoised per code object -- the
    reference repeats it on every call (transpiler.py:160-182, 369-403); free variables are re-captured
    from the closure / globals each time, so changed values are picked up."""
    if not callable(func) or not hasattr(func, "__code__"):
        raise TranspilerError(f"Could not get source code: {type(func).__name__} is not a Python function")
    code = func.__code__
    rec = _LOWER_CACHE.get(code)
    if rec is None:
        try:
            rec = _lower_structure(func)
        except TranspilerError as exc:
            if not str(exc).startswith(("Undefined variable", "Unsupported external variable")):
                if len(_LOWER_CACHE) < _LOWER_CACHE_LIMIT:
                    _LOWER_CACHE[code] = exc          # structural rejection: permanent for this code object
            raise
        if len(_LOWER_CACHE) >= _LOWER_CACHE_LIMIT:
            _LOWER_CACHE.clear()
        _LOWER_CACHE[code] = rec
    elif isinstance(rec, TranspilerError):
        raise TranspilerError(str(rec))
    low = _Lowering()
    low.imports, low.module_aliases = rec.imports, rec.module_aliases
    consts = low.capture(func, set(rec.params), rec.used, rec.assigned)
    params = list(rec.params)
    defaults = getattr(func, "__defaults__", None) or ()
    if bind_defaults and defaults and len(params) > 1:
        bound = params[max(1, len(params) - len(defaults)):]
        values = defaults[len(defaults) - len(bound):]
        for name, value in zip(bound, values):
            if isinstance(value, bool):
                consts[name] = 1.0 if value else 0.0
            elif isinstance(value, (int, float)):
                consts[name] = float(value)
            else:
                raise TranspilerError(
                    f"Unsupported external variable type for '{name}': {type(value).__name__}. "
                    f"Only int, float, and bool are supported."
                )
        params = params[: len(params) - len(bound)]
    return Function(rec.name, params, consts, rec.body)


_FINGERPRINT_NAMES: Dict[object, tuple] = {}      # code object -> global names whose values the function reads


def fingerprint(func: Callable, bind_defaults: bool = True) -> tuple:
    """A hashable key that is equal for two callables exactly when lower() gives them the same IR and the same
    captured constants: (code object, closure values, defaults, values of the globals it reads). A repeat call with
    the same function then costs one dict lookup instead of a lowering + emission (the reference re-transpiles on
    every call, python/wgpu_montecarlo/__init__.py:740-746). Raises TranspilerError like lower(); TypeError when a
    captured value is not hashable (lower() rejects those too)."""
    code = getattr(func, "__code__", None)
    if code is None:
        raise TranspilerError(f"Could not get source code: {type(func).__name__} is not a Python function")
    names = _FINGERPRINT_NAMES.get(code)
    if names is None:
        lower(func, bind_defaults)                              # structural checks + fills _LOWER_CACHE
        rec = _LOWER_CACHE[code]
        skip = set(rec.params) | rec.assigned | set(rec.imports) | set(rec.module_aliases) | _EXCLUDED_AS_BUILTINS
        names = tuple(sorted(n for n in rec.used - skip if n not in code.co_freevars))
        if len(_FINGERPRINT_NAMES) >= _LOWER_CACHE_LIMIT:
            _FINGERPRINT_NAMES.clear()
        _FINGERPRINT_NAMES[code] = names
    cells = func.__closure__
    g = func.__globals__
    try:
        kw = func.__kwdefaults__
        return (code, tuple(c.cell_contents for c in cells) if cells else None, func.__defaults__,
                tuple(g.get(n) for n in names) if names else None, tuple(sorted(kw.items())) if kw else None)
    except ValueError:                                          # an empty closure cell
        raise TypeError("closure cell is empty")


def _lower_structure(func: Callable) -> _Lowered:
    is_lambda = func.__name__ == "<lambda>"
    if is_lambda:
        _pin_source(func)
    try:
        source = inspect.getsource(func)
    except (OSError, TypeError) as exc:
        raise TranspilerError(f"Could not get source code: {exc}")
    raw_first = source.splitlines()[0] if source else ""
    source = textwrap.dedent(source)
    dedented_first = source.splitlines()[0] if source else ""
    margin = len(raw_first) - len(dedented_first)      # columns removed by dedent

    low = _Lowering()
    if is_lambda:
        try:
            tree = ast.parse(source)
        except SyntaxError:
            tree = _parse_fragment(source)
        if tree is None:
            raise TranspilerError(
                "Could not parse lambda source. "
                "This may happen when lambdas are passed directly in function calls. "
                "Consider assigning the lambda to a variable first."
            )
    else:
        try:
            tree = ast.parse(source)
        except SyntaxError as exc:
            raise TranspilerError(f"Invalid Python syntax: {exc}")

    low.scan_imports(tree, overwrite=True)
    whole_file = _file_tree(func)
    if whole_file is not None:
        low.scan_imports(whole_file, overwrite=False)

    if is_lambda:
        lambdas = [n for n in ast.walk(tree) if isinstance(n, ast.Lambda)]
        if not lambdas:
            raise TranspilerError("No lambda definition found")
        node = lambdas[0] if len(lambdas) == 1 else _pick_lambda(lambdas, func, margin)
        params = [a.arg for a in node.args.args]
        used, _ = _names_in([node.body])
        low.capture(func, set(params), used, set())          # the reference checks free variables first
        body_is_bool = isinstance(node.body, (ast.Compare, ast.BoolOp))
        body = [Return(low.expr(node.body), boolean=body_is_bool)]
        return _Lowered(f"user_func_{uuid.uuid4().hex[:8]}", params, body, used, set(), low.imports,
                        low.module_aliases)

    definition = next((n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)), None)
    if definition is None:
        raise TranspilerError("No function definition found")
    params = [a.arg for a in definition.args.args]
    used, assigned = _names_in(definition.body)
    low.capture(func, set(params), used, assigned)           # the reference checks free variables first
    body = [low.stmt(s) for s in definition.body]
    return _Lowered(definition.name, params, body, used, assigned, low.imports, low.module_aliases)
