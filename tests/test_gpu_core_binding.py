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


@pytest.fixture(scope="module", params=["precise", "default"])
def core(request):
    """Both ways the binding compiles what it is handed (wgpu_montecarlo/_core.py): literally, and as the plan this package's
    API builds for the same call. Every test below holds both to the same bounds."""
    from wgpu_montecarlo import _core

    return _core.MonteCarloIntegrator(math=request.param)


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


def _golden_calls():
    import json
    from pathlib import Path

    gdir = Path(__file__).resolve().parent / "golden"
    meta = json.loads((gdir / "boundary_payloads.json").read_text())
    arrays = np.load(gdir / "boundary_payloads.npz")
    calls = []
    for entry in meta:
        args = []
        for a in entry["args"]:
            if isinstance(a, dict) and "array" in a:
                args.append(arrays[a["array"]])
            elif isinstance(a, dict) and "wgsl" in a:
                args.append(list(a["wgsl"]))
            else:
                args.append(a)
        calls.append((entry["method"], args))
    return calls


ORC_DIST = {"normal": oracle.NORMAL, "uniform": oracle.UNIFORM, "exponential": oracle.EXPONENTIAL, "custom": oracle.CUSTOM}


def test_replay_of_the_payloads_the_reference_emitted(core):
    """tests/golden/boundary_payloads.{json,npz} hold what the REFERENCE's own Python half handed to `_core` for the
    BASELINE configs (captured by tools/make_golden.py with a recorder in place of the native module): the WGSL
    strings its transpiler and its importance-sampling wrapper generator emitted (python/wgpu_montecarlo/
    __init__.py:893-905, 968-980; names and const order normalised), the parameter dicts and the float32 tables.
    They are replayed VERBATIM through this package's `_core` (sizes reduced to what the oracle covers in seconds)
    and the float32[K] results are held to the oracle on the same stream."""
    calls = _golden_calls()
    assert [m for m, _ in calls] == ["integrate", "integrate", "integrate_is_tables", "integrate_mcmc", "integrate", "integrate"]
    pows = lambda k: [(oracle.FN_IDENTITY, 0)] + [(oracle.FN_POW, j) for j in range(2, k + 1)]
    s2pi = float(np.float32(2.5066282746310002))

    # C1 / C2: integrate(functions, 'normal', {...}, n, seed, None, None, None)
    for idx, n in ((0, 1_000_000), (1, 3_000_000)):
        fns, dist, params, _, seed, *rest = calls[idx][1]
        got = core.integrate(fns, dist, params, n, seed, *rest)
        ref = oracle.integrate(pows(len(fns)), ORC_DIST[dist], params["mean"], params["std"], n_samples=n, seed=seed, guard=1)
        assert got.dtype == np.float32 and np.allclose(got, ref["sums"] / ref["n_eff"], rtol=2e-5, atol=2e-5), (idx, got)

    # C3: integrate_is_tables(functions, 'normal', params, n, seed, None, None, target_x, target_pdf, None, None, None)
    fns, dist, params, _, seed, x_t, cdf_t, tx, tp, px, pp, tt = calls[2][1]
    assert all("pdf_target_from_table(x)" in f and "_is_pdf_q_" in f for f in fns) and px is None and pp is None
    got = core.integrate_is_tables(fns, dist, params, 2_000_000, seed, x_t, cdf_t, tx, tp, px, pp, tt)
    ref = oracle.integrate(pows(4), oracle.NORMAL, 2.0, 3.0, n_samples=2_000_000, seed=seed, guard=1,
                           p=(oracle.PDF_TABLE, tx, tp), q=(oracle.PDF_NORMAL, 2.0, 3.0, s2pi))
    assert np.allclose(got, ref["sums"] / ref["n_eff"], rtol=2e-5, atol=2e-5), got

    # C4: integrate_mcmc(functions, 'normal', {...}, 'custom', {...}, n_steps, n_chains, n_burnin, seed, None, None, 4 tables, None)
    fns, pd, pp_, td, tp_, n_steps, n_chains, n_burnin, seed, x_t, cdf_t, tx, tl, px, pl, tt = calls[3][1]
    assert (n_steps, n_chains, n_burnin) == (10_000, 1_048_576, 1000) and len(tx) == len(px) == 2048
    got = core.integrate_mcmc(fns, pd, pp_, td, tp_, 1500, 1000, 100, seed, x_t, cdf_t, tx, tl, px, pl, tt)
    ref = oracle.mcmc(pows(2), oracle.NORMAL, pp_["mean"], pp_["std"], tx, tl, px, pl, n_steps=1500, n_chains=1000, n_burnin=100,
                      seed=seed, guard=1)
    assert np.allclose(got, ref["sums"][:2] / ref["n_eff"], rtol=2e-4, atol=2e-4), got

    # C5's sampler: integrate(functions, 'custom', {...}, n, seed, x_table, cdf_table, None) -- 2048-point Beta(2,5) tables
    fns, dist, params, _, seed, x_t, cdf_t, tt = calls[4][1]
    got = core.integrate(fns, dist, params, 3_000_000, seed, x_t, cdf_t, tt)
    ref = oracle.integrate(pows(2), oracle.CUSTOM, 0.0, 0.0, n_samples=3_000_000, seed=seed, guard=1, cdf_table=cdf_t, x_table=x_t)
    assert np.allclose(got, ref["sums"] / ref["n_eff"], rtol=2e-5, atol=2e-5), got
    assert abs(got[0] - 2 / 7) < 1e-3

    # analytic / analytic importance sampling: both PDFs arrive as WGSL text inside the wrapper
    fns, dist, params, n, seed, *rest = calls[5][1]
    assert "_is_pdf_p_0" in fns[0] and "_is_pdf_q_0" in fns[0]
    got = core.integrate(fns, dist, params, 2_000_000, seed, *rest)
    ref = oracle.integrate(pows(1), oracle.NORMAL, 0.5, 1.5, n_samples=2_000_000, seed=seed, guard=1,
                           p=(oracle.PDF_NORMAL, 0.0, 1.0, s2pi), q=(oracle.PDF_NORMAL, 0.5, 1.5, s2pi))
    assert np.allclose(got, ref["sums"] / ref["n_eff"], rtol=2e-5, atol=2e-5), got


@pytest.mark.parametrize("case", ["normal|normal", "exponential|exponential", "uniform|normal", "table|normal", "normal|table"])
def test_integrate_mcmc_without_tables_uses_the_analytic_log_densities(core, case):
    """The four log-PDF tables of `_core.integrate_mcmc` are optional (src/lib.rs:296-304): without one the MH step
    evaluates generate_log_pdf_code_for_dist's expression for that distribution type (src/shader_gen.rs:543-571,
    -100 outside the support). Held to the oracle's restatement of the same fallback, same stream."""
    from wgpu_montecarlo import Distribution

    fns = ["fn f(x: f32) -> f32 { return x; }", "fn g(x: f32) -> f32 { return x * x; }"]
    orc_fns = [(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2)]
    tgt, prop = case.split("|")
    t_args = dict(normal=("normal", {"mean": 0.3, "std": 1.0}, (oracle.NORMAL, 0.3, 1.0)),
                  exponential=("exponential", {"lambda": 1.5}, (oracle.EXPONENTIAL, 1.5, 0.0)),
                  uniform=("uniform", {"min": -1.0, "max": 2.0}, (oracle.UNIFORM, -1.0, 2.0)),
                  table=("normal", {"mean": 0.3, "std": 1.0}, None))[tgt]
    p_args = dict(normal=("normal", {"mean": 0.0, "std": 2.0}, oracle.NORMAL, 0.0, 2.0),
                  exponential=("exponential", {"lambda": 0.7}, oracle.EXPONENTIAL, 0.7, 0.0),
                  table=("normal", {"mean": 0.0, "std": 2.0}, oracle.NORMAL, 0.0, 2.0))[prop]
    tx = tl = px = pl = None
    if tgt == "table":
        tx, tl = Distribution.normal(0.3, 1.0).get_log_pdf_table()
    if prop == "table":
        px, pl = Distribution.normal(0.0, 2.0).get_log_pdf_table()
    got = core.integrate_mcmc(fns, p_args[0], p_args[1], t_args[0], t_args[1], 1200, 1000, 100, 17, None, None, tx, tl, px, pl, None)
    ref = oracle.mcmc(orc_fns, p_args[2], p_args[3], p_args[4], tx, tl, px, pl, n_steps=1200, n_chains=1000, n_burnin=100, seed=17,
                      guard=1, target_analytic=t_args[2])
    assert np.all(np.isfinite(got))
    assert np.allclose(got, ref["sums"][:2] / ref["n_eff"], rtol=2e-4, atol=2e-4), (case, got, ref["sums"][:2] / ref["n_eff"])
    truth = dict(normal=(0.3, 1.09), exponential=(1 / 1.5, 2 / 1.5**2), uniform=(0.5, 1.0), table=(0.3, 1.09))[tgt]
    assert abs(got[0] - truth[0]) < 0.02 and abs(got[1] - truth[1]) < 0.05
    if tgt == "table" or prop == "table":
        return
    with pytest.raises(RuntimeError, match="custom distribution needs its log-PDF table"):
        core.integrate_mcmc(fns, "normal", {}, "custom", {"table_size": 8}, 10, 256, 0, 1)


def test_error_mapping(core):
    with pytest.raises(ValueError, match="At least one function"):
        core.integrate([], "normal", {}, 1000, 1)
    with pytest.raises(ValueError, match="Unknown distribution type"):
        core.integrate(["fn f(x: f32) -> f32 { return x; }"], "cauchy", {}, 1000, 1)
    with pytest.raises(ValueError, match="n_steps must be positive"):
        core.integrate_mcmc(["fn f(x: f32) -> f32 { return x; }"], "normal", {}, "normal", {}, 0, 10, 0, 1)
    with pytest.raises(RuntimeError):
        core.integrate(["fn f(x: f32) -> f32 { return nope(x); }"], "normal", {}, 1000, 1)


def test_the_binding_takes_a_math_mode(monkeypatch):
    """`math` (and MCX_CORE_MATH for callers that cannot pass it) selects how the binding compiles the strings, here the
    routines behind their WGSL builtins; "precise" is the literal, ocml one. Uniform sampling: the same samples in every mode."""
    from wgpu_montecarlo import _core

    texts = ["fn f(x: f32) -> f32 { return x / (exp(sin(x)) + 2.0 + cos(exp(x))); }",
             "fn g(x: f32) -> f32 { return pow(abs(x), 1.5) + tan(0.3 * x) + sqrt(abs(x)) * log(1.0 + x * x); }"]
    args = (texts, "uniform", {"min": -3.0, "max": 3.0}, 1_000_000, 11)
    monkeypatch.delenv("MCX_CORE_MATH", raising=False)
    default = _core.MonteCarloIntegrator().integrate(*args)
    assert np.array_equal(default, _core.MonteCarloIntegrator(math="default").integrate(*args))
    precise = _core.MonteCarloIntegrator(math="precise").integrate(*args)
    assert np.allclose(default, precise, rtol=0, atol=2e-6) and not np.array_equal(default, precise)
    monkeypatch.setenv("MCX_CORE_MATH", "precise")
    assert np.array_equal(_core.MonteCarloIntegrator().integrate(*args), precise)
    assert np.allclose(_core.MonteCarloIntegrator(math="fast").integrate(*args), precise, rtol=0, atol=5e-6)
    with pytest.raises(ValueError):
        _core.MonteCarloIntegrator(math="quick")


def test_the_core_objects_caches_are_bounded():
    """mcx_core keeps resident tables (64, by content) and compiled modules (128, by payload): more distinct ones than that are
    evicted and rebuilt, results stay right, and device memory does not grow with the number of distinct calls."""
    import torch

    from wgpu_montecarlo import _core

    core = _core.MonteCarloIntegrator()
    free0 = None
    for lap in range(2):
        for j in range(70):                                  # 70 different target tables through 64 slots
            xs = np.linspace(-6.0, 6.0, 300 + j).astype(np.float32)
            ps = (np.exp(-0.5 * xs.astype(np.float64) ** 2) / np.sqrt(2 * np.pi)).astype(np.float32)
            text = (f"\nfn _is_wrapper_0(x: f32) -> f32 {{\n    let f_val = _is_f_orig_0(x);\n    let p = pdf_target_from_table(x);\n"
                    f"    let q = _is_pdf_q_0(x);\n    return f_val * p / q;\n}}\n\n\nfn _is_pdf_q_0(x: f32) -> f32 {{\n    const mean: f32 = 0.0;\n"
                    f"    const sigma: f32 = 2.0;\n    const sqrt_2pi: f32 = 2.5066282746310002;\n    var z = ((x - mean) / sigma);\n"
                    f"    return (exp((((-0.5) * z) * z)) / (sigma * sqrt_2pi));\n}}\nfn _is_f_orig_0(x: f32) -> f32 {{\n    return pow(x, 2.0);\n}}\n")
            got = core.integrate_is_tables([text], "normal", {"mean": 0.0, "std": 2.0}, 200_000, 3, None, None, xs, ps, None, None, None)
            assert abs(got[0] - 1.0) < 0.05, (lap, j, got)
        for j in range(140):                                 # 140 different payloads through 128 slots
            got = core.integrate([f"fn f(x: f32) -> f32 {{ return x * {j}.0 + 1.0; }}"], "uniform", {"min": 0.0, "max": 1.0}, 100_000, 5)
            assert abs(got[0] - (0.5 * j + 1.0)) < 0.01 * (j + 1), (lap, j, got)
        torch.cuda.synchronize()
        free = torch.cuda.mem_get_info(0)[0]
        if free0 is None:
            free0 = free
    assert free0 - free < 64 * 2**20, (free0, free)          # the second lap rebuilt what the first evicted: no growth
