"""The `_core`-level drop-in (wgpu_montecarlo/_core.py): the reference's native-module interface (src/lib.rs:17-431)
driven the way the reference's Python half drives it -- WGSL text in, float32[K] out. The WGSL strings are produced
here in the format the reference's transpiler and its importance-sampling wrapper generator emit
(python/wgpu_montecarlo/__init__.py:893-899, 968-980); results are held to this package's own API and to the oracle.
"""
import math
import re

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _wgsl(fn, name):
    """Transpile `fn` and rename the function like the reference's _rename_wgsl_function does."""
    from wgpu_montecarlo import transpile_function

    return re.sub(r"fn\s+\w+\s*\(", f"fn {name}(", transpile_function(fn), count=1)


@pytest.fixture(scope="module")
def core():
    from wgpu_montecarlo import _core

    return _core.MonteCarloIntegrator()


def test_integrate_takes_the_transpilers_wgsl(core, integrator):
    from wgpu_montecarlo import Distribution, transpile_function

    fns = [lambda x: x, lambda x: x**2, lambda x: math.sin(x) * x]
    texts = [transpile_function(f) for f in fns]
    got = core.integrate(texts, "normal", {"mean": 0.5, "std": 1.5, "support": (-10.0, 11.0)}, 1_000_000, 42)
    assert got.dtype == np.float32 and got.shape == (3,)
    want = integrator.integrate(fns, Distribution.normal(0.5, 1.5), n_samples=1_000_000, seed=42).values
    assert np.allclose(got, want, rtol=2e-6, atol=2e-6)
    # custom distribution: x_table / cdf_table as the reference passes them (src/lib.rs:71-77)
    beta = Distribution.beta(2.0, 5.0)
    got = core.integrate(texts[:2], "custom", {"table_size": 2048, "support": (0.0, 1.0)}, 500_000, 7, beta._x_table,
                         beta._cdf_table, 4096)
    want = integrator.__class__(target_threads=4096).integrate(fns[:2], beta, n_samples=500_000, seed=7).values
    assert np.allclose(got, want, rtol=3e-6, atol=3e-6)


def test_integrate_is_tables_runs_the_wrapper_text_literally(core, integrator):
    """C3's shape: target from a 512-point table, analytic normal proposal. The wrapper calls pdf_target_from_table(x),
    which the translator maps onto the table staged by the launch (desc.user_tables)."""
    from wgpu_montecarlo import Distribution

    xs = np.linspace(0, 10, 512)
    target, proposal = Distribution.from_pdf_table(xs, np.exp(-xs)), Distribution.normal(2.0, 3.0)
    fns = [lambda x: x, lambda x: x**2]
    texts = []
    for i, f in enumerate(fns):
        texts.append(f"""
fn _is_wrapper_{i}(x: f32) -> f32 {{
    let f_val = _is_f_orig_{i}(x);
    let p = pdf_target_from_table(x);
    let q = _is_pdf_q_{i}(x);
    return f_val * p / q;
}}


{_wgsl(proposal._pdf_func, f"_is_pdf_q_{i}")}
{_wgsl(f, f"_is_f_orig_{i}")}
""")
    tx, tp = target.get_or_compute_pdf_table()
    got = core.integrate_is_tables(texts, "normal", {"mean": 2.0, "std": 3.0}, 2_000_000, 3, None, None, tx, tp, None, None, None)
    assert got.dtype == np.float32 and got.shape == (2,)
    want = integrator.integrate_importance_sampling(fns, target, proposal, n_samples=2_000_000, seed=3).values
    assert np.allclose(got, want, rtol=2e-5, atol=2e-5), (got, want)
    s2pi = float(np.float32(np.sqrt(2 * np.pi)))
    ref = oracle.integrate([(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2)], oracle.NORMAL, 2.0, 3.0, n_samples=2_000_000, seed=3,
                           guard=1, p=(oracle.PDF_TABLE, target._x_table, target._pdf_table), q=(oracle.PDF_NORMAL, 2.0, 3.0, s2pi))
    assert np.allclose(got, ref["sums"] / ref["n_eff"], rtol=2e-5, atol=2e-5)


def test_integrate_mcmc_payload(core, integrator):
    from wgpu_montecarlo import Distribution, transpile_function

    target = Distribution.from_pdf(lambda x: 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2)), support=(-10, 10))
    proposal = Distribution.normal(0.0, 2.0)
    fns = [lambda x: x, lambda x: x**2]
    tx, tl = target.get_log_pdf_table()
    px, pl = proposal.get_log_pdf_table()
    got = core.integrate_mcmc([transpile_function(f) for f in fns], "normal", {"mean": 0.0, "std": 2.0}, "custom",
                              {"table_size": 2048, "support": (-10, 10)}, 1500, 1000, 100, 42, None, None, tx, tl, px, pl, None)
    assert got.dtype == np.float32 and got.shape == (2,)
    ref = oracle.mcmc([(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2)], oracle.NORMAL, 0.0, 2.0, tx, tl, px, pl, n_steps=1500,
                      n_chains=1000, n_burnin=100, seed=42, guard=1)
    assert np.allclose(got, ref["sums"][:2] / ref["n_eff"], rtol=2e-4, atol=2e-4)
    assert abs(got[0]) < 0.1 and abs(got[1] - 5.0) < 0.2


def test_error_mapping(core):
    with pytest.raises(ValueError, match="At least one function"):
        core.integrate([], "normal", {}, 1000, 1)
    with pytest.raises(ValueError, match="Unknown distribution type"):
        core.integrate(["fn f(x: f32) -> f32 { return x; }"], "cauchy", {}, 1000, 1)
    with pytest.raises(ValueError, match="n_steps must be positive"):
        core.integrate_mcmc(["fn f(x: f32) -> f32 { return x; }"], "normal", {}, "normal", {}, 0, 10, 0, 1)
    with pytest.raises(RuntimeError):
        core.integrate(["fn f(x: f32) -> f32 { return nope(x); }"], "normal", {}, 1000, 1)
