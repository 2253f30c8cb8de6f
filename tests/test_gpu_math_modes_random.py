"""math="default" against math="precise" on randomly generated integrands: expression trees over x, constants, + - * /, and the
builtins whose default form is a hardware instruction sequence (sin, cos, tan on a bounded argument, exp, log, sqrt, pow with
positive and negative bases, sinh, cosh). Uniform sampling gives both modes the same samples, so the means may differ only by the
per-value error of the builtins (<= 4e-7 absolute for sin / cos, ~1e-7 * (1 + |y log2 x|) relative for pow, ...) propagated through
the tree: bounded here by 2e-5 of the integrand's scale. Generated as a module file (the front end reads function sources)."""
import importlib.util
import math
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def expr(rng, depth):
    if depth == 0 or rng.random() < 0.2:
        return rng.choice(["x", "x", f"{rng.uniform(-2, 2):.3f}", "(0.5 * x)", "(x + 1.5)"])
    kind = rng.randrange(12)
    a, b = expr(rng, depth - 1), expr(rng, depth - 1)
    if kind == 0:
        return f"({a} + {b})"
    if kind == 1:
        return f"({a} * {b})"
    if kind == 2:
        return f"({a} - {b})"
    if kind == 3:
        return f"({a} / (1.5 + ({b}) ** 2))"
    if kind == 4:
        return f"math.sin({a})"
    if kind == 5:
        return f"math.cos({a})"
    if kind == 6:
        return f"math.exp(-(({a}) ** 2))"
    if kind == 7:
        return f"(abs({a}) + 0.25) ** {rng.uniform(-2.5, 3.5):.3f}"
    if kind == 8:
        return f"math.sqrt(abs({a})) * math.log(1.0 + ({b}) ** 2)"
    if kind == 9:
        return f"(math.sinh(0.3 * math.sin({a})) + math.cosh(0.3 * math.cos({b})))"
    if kind == 10:
        return f"math.tan(0.4 * math.sin({a}))"
    return f"({a}) ** {rng.choice([2, 3, 5])}"              # a whole power of a possibly negative base


def test_random_integrands_agree_between_default_and_precise(tmp_path):
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    rng = random.Random(2025)
    bodies = [expr(rng, 4) for _ in range(48)]
    src = "import math\n\nimport numpy as np\n\n" + "\n\n".join(f"def f{i}(x):\n    return {b}\n" for i, b in enumerate(bodies))
    path = tmp_path / "random_integrands.py"
    path.write_text(src.replace("abs(", "np.abs("))
    spec = importlib.util.spec_from_file_location("random_integrands", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fns = [getattr(mod, f"f{i}") for i in range(len(bodies))]
    dist = Distribution.uniform(-3.0, 3.0)
    got = {}
    for mode in ("precise", "default"):
        mc = MonteCarloIntegrator(math=mode)
        vals = []
        for start in range(0, len(fns), 16):
            vals += mc.integrate(fns[start:start + 16], dist, n_samples=400_000, seed=11).values.tolist()
        got[mode] = np.array(vals)
    xs = np.linspace(-3, 3, 4001)
    for i, f in enumerate(fns):
        scale = max(1.0, float(np.max(np.abs([f(float(x)) for x in xs]))))
        assert np.isfinite(got["precise"][i]) and abs(got["default"][i] - got["precise"][i]) <= 2e-5 * scale, (bodies[i], got["default"][i], got["precise"][i], scale)
    assert np.count_nonzero(got["default"] != got["precise"]) > 10          # the modes do compile different code
