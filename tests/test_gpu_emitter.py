"""Emitter semantics end to end: every accepted function of the front-end corpus (tests/golden/corpus.py) is
fused into GPU kernels and its expectation compared with the SAME Python function applied to the oracle's
samples of the same counter stream. Differences allowed: f32 evaluation on the device vs f64 in Python, and
the documented device semantics (`%` truncates like WGSL / fmodf)."""
import json
import math
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLDEN))
import corpus  # noqa: E402

EXPECTED = json.loads((GOLDEN / "transpiler_corpus.json").read_text())
SKIP = {"unknown_function_passthrough", "two_params", "lam_two"}       # not single-argument / not HIP-emittable
NAMES = [n for n in sorted(EXPECTED) if EXPECTED[n]["ok"] and n not in SKIP]


def python_value(name, fn, x):
    if name == "modulo":
        return math.fmod(x, 2.0)                # device `%` is truncated (WGSL semantics), Python's is floored
    v = fn(x)
    if isinstance(v, complex):                  # (-0.1) ** 0.5 is complex in Python, NaN in powf
        return float("nan")
    return float(v)


@pytest.mark.parametrize("dist", ["normal", "uniform"])
def test_corpus_functions_evaluate_like_python(integrator, dist):
    from wgpu_montecarlo import Distribution

    funcs = corpus.corpus()
    if dist == "normal":
        d, code, p1, p2 = Distribution.normal(0.3, 1.2), oracle.NORMAL, 0.3, 1.2
    else:
        d, code, p1, p2 = Distribution.uniform(0.05, 3.0), oracle.UNIFORM, 0.05, 3.0
    n = 65536 * 2
    xs = oracle.samples(code, p1, p2, n_samples=n, seed=17, guard=1).astype(np.float64).ravel()
    batch = [funcs[name] for name in NAMES]
    got = []
    for start in range(0, len(batch), 16):                     # K <= 16 per fused kernel
        res = integrator.integrate(batch[start:start + 16], d, n_samples=n, seed=17)
        assert res.meta["n_eff"] == xs.size
        got.extend(res.values.tolist())
    for name, value in zip(NAMES, got):
        with np.errstate(all="ignore"):
            want = np.mean([python_value(name, funcs[name], float(x)) for x in xs])
        if not np.isfinite(want):                              # e.g. x**0.5 of a negative sample: NaN on both sides
            assert not np.isfinite(value), (name, value, want)
            continue
        assert value == pytest.approx(want, rel=2e-4, abs=2e-5), (name, value, want)


def test_wgsl_strings_evaluate_like_python(integrator):
    from wgpu_montecarlo import Distribution

    n = 65536 * 2
    xs = oracle.samples(oracle.NORMAL, 0.0, 1.0, n_samples=n, seed=23, guard=1).astype(np.float64).ravel()
    cases = [
        ("fn f(x: f32) -> f32 { return x * x; }", lambda x: x * x),
        ("fn g(x: f32) -> f32 { let a = 2.0; var s = 0.0; for (var i = 0; i < 3; i++) { s += x * a; } return s; }", lambda x: 6.0 * x),
        ("fn h(x: f32) -> f32 { if (x > 0.5) { return 1.0; } else if (x < -0.5) { return -1.0; } return 0.0; }",
         lambda x: 1.0 if x > 0.5 else (-1.0 if x < -0.5 else 0.0)),
        ("fn s(x: f32) -> f32 { return select(0.0, sqrt(abs(x)), x > 0.0) + clamp(x, -1.0, 1.0) + f32(u32(3.7)); }",
         lambda x: (math.sqrt(abs(x)) if x > 0 else 0.0) + max(-1.0, min(1.0, x)) + 3.0),
        ("fn m(x: f32) -> f32 { return helper(x) % 2.0 + pow(abs(x), 1.5); }\nfn helper(y: f32) -> f32 { return y * 3.0; }",
         lambda x: math.fmod(3.0 * x, 2.0) + abs(x) ** 1.5),
        ("fn w(x: f32) -> f32 { var i = 0u; var acc = 0.0; while (i < 4u) { acc = acc + exp(-abs(x)); i = i + 1u; } return acc; }",
         lambda x: 4.0 * math.exp(-abs(x))),
    ]
    res = integrator.integrate([c[0] for c in cases], Distribution.normal(0.0, 1.0), n_samples=n, seed=23)
    for (src, fn), value in zip(cases, res.values):
        want = np.mean([fn(float(x)) for x in xs])
        assert value == pytest.approx(want, rel=2e-4, abs=2e-5), (src, value, want)
