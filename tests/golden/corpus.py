"""Function corpus for front-end parity: each entry is run through the reference's transpiler (by
tools/make_golden.py, in the build container only) and through this package's front-end (by
tests/test_frontend.py); accept/reject decisions, error texts and WGSL output must agree.

One lambda per line: the reference needs Python >= 3.11 to tell same-line lambdas apart.
"""
import math
from math import pi as PI, sin, cos as cosine, sqrt

import numpy as np

SCALE = 1.5
OFFSET = 2
FLAG = True
VECTOR = [1.0, 2.0]
NP32 = np.float32(0.25)
NP64 = np.float64(0.75)


def identity(x):
    return x


def affine(x):
    y = x * SCALE
    y = y + OFFSET
    return y


def two_params(x, rng):
    return x + (rng - 0.5) * 0.1


def branchy(x):
    if x > 0.5:
        return 1.0
    else:
        return 0.0


def branch_no_else(x):
    y = 0.0
    if x < 0:
        y = -x
    return y


def bool_return(x):
    return x > 0.5


def bool_and(x):
    return (x > 0.0) and (x < 1.0)


def ternary(x):
    return x if x > 0 else -x


def while_loop(x):
    acc = 0.0
    i = 0.0
    while i < 3:
        acc = acc + x
        i = i + 1
    return acc


def uses_math(x):
    return math.exp(-x * x / 2.0) / math.sqrt(2.0 * math.pi)


def uses_numpy(x):
    return np.sin(x) + np.power(x, 3) + np.abs(x)


def uses_from_imports(x):
    return sin(x) * cosine(x) + sqrt(PI)


def modulo(x):
    return x % 2.0


def power_ops(x):
    return x**2 + x**0.5 + 2**x


def unary_ops(x):
    return +x - (-x)


def constants(x):
    return math.e + math.tau + np.euler_gamma + x


def uses_flag(x):
    return x * FLAG


def uses_np64(x):
    return x * NP64


def clamp_mix(x):
    return max(min(x, 1.0), 0.0)


def unknown_function_passthrough(x):
    return math.erf(x)


# ---- rejected ----------------------------------------------------------------------------------
def with_docstring(x):
    """A docstring is an expression statement holding a str constant."""
    return x


def aug_assign(x):
    y = x
    y += 1
    return y


def for_loop(x):
    acc = 0.0
    for i in range(3):
        acc = acc + x
    return acc


def chained_compare(x):
    return 1.0 if 0 < x < 1 else 0.0


def builtin_abs(x):
    return abs(x)


def builtin_float(x):
    return float(x) * 2


def uses_list(x):
    return x * VECTOR


def uses_np32(x):
    return x * NP32


def undefined_name(x):
    return x * NOT_DEFINED_ANYWHERE  # noqa: F821


def tuple_assign(x):
    a, b = x, x
    return a + b


def not_operator(x):
    return 1.0 if not (x > 0) else 0.0


def floor_div(x):
    return x // 2


def unknown_constant(x):
    return x * math.foo  # noqa


def string_constant(x):
    return "a"


def unsupported_module(x):
    import os

    return x * os.sep


def subscript(x):
    return VECTOR[0] * x


def make_closure(a, b):
    def inner(x):
        return a * x + b

    return inner


def make_closure_shadow(a):
    def inner(x):
        a2 = a * 2.0
        return a2 * x

    return inner


lam_identity = lambda x: x
lam_square = lambda x: x**2
lam_cmp = lambda x: x > 0.5
lam_math = lambda x: math.sin(x) * math.cos(x)
lam_global = lambda x: x * SCALE + OFFSET
lam_ifexp = lambda x: 1.0 if x >= 0 else -1.0
lam_bool_const = lambda x: True
lam_two = lambda x, y: x * y


def corpus():
    """name -> callable (closures instantiated here so both sides see the same objects)."""
    items = {name: obj for name, obj in globals().items()
             if callable(obj) and getattr(obj, "__module__", None) == __name__
             and name not in ("corpus", "make_closure", "make_closure_shadow")}
    items["closure_ab"] = make_closure(1.5, -2)
    items["closure_shadow"] = make_closure_shadow(3)
    return dict(sorted(items.items()))
