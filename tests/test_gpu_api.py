"""Public-API behaviour on the GPU, scenario by scenario as the reference's own GPU tests exercise it
(tests/test_integrator.py, test_distributions.py, test_importance_sampling.py, test_mcmc.py in the
reference; cited per test). Accuracy bars are the reference's: |E - truth| < 0.01 at n = 1e7 etc.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mc(integrator):
    return integrator


def D():
    from wgpu_montecarlo import Distribution

    return Distribution


# ---- tests/test_integrator.py ------------------------------------------------------------------
def test_result_object_shape(mc):                                      # :24-46, :293-357
    res = mc.integrate([lambda x: x, lambda x: x**2, lambda x: x**3], D().normal(0.0, 1.0), n_samples=1_000_000, seed=42)
    assert len(res) == 3 and res.n_functions == 3 and res.n_samples == 1_000_000
    assert res.values.dtype == np.float64 and res.values.shape == (3,)
    assert res[1] == res.values[1]
    assert "IntegrationResult" in repr(res)
    assert abs(res.values[0]) < 0.1 and abs(res.values[1] - 1.0) < 0.1 and abs(res.values[2]) < 0.1


def test_lambdas_written_inline_and_unpacked(mc):                      # :92-160
    res = mc.integrate(
        [lambda x: x, lambda x: x**2, lambda x: x**3, lambda x: x**4],
        D().normal(0.0, 1.0),
        n_samples=1_000_000,
        seed=42,
    )
    assert np.all(np.abs(res.values - [0, 1, 0, 3]) < 0.1)
    f1, f2 = lambda x: x, lambda x: x**2
    res = mc.integrate([f1, f2], D().normal(0.0, 1.0), n_samples=1_000_000, seed=42)
    assert abs(res.values[0]) < 0.1 and abs(res.values[1] - 1.0) < 0.1


def test_named_functions_closures_and_globals(mc):                     # :259-307
    scale = 2.0

    def scaled_square(x):
        y = x * scale
        return y * y

    def indicator(x):
        if x > 0.0:
            return 1.0
        return 0.0

    res = mc.integrate([scaled_square, indicator, lambda x: np.abs(x)], D().normal(0.0, 1.0), n_samples=2_000_000)
    assert abs(res.values[0] - 4.0) < 0.05 and abs(res.values[1] - 0.5) < 0.01
    assert abs(res.values[2] - math.sqrt(2 / math.pi)) < 0.01



def test_accuracy_at_1e7(mc):                                          # :165-257
    n = 10_000_000
    r = mc.integrate([lambda x: x, lambda x: x**2], D().normal(0.0, 1.0), n_samples=n)
    assert abs(r.values[0]) < 0.01 and abs(r.values[1] - 1.0) < 0.01
    r = mc.integrate([lambda x: x, lambda x: x**2], D().uniform(0.0, 1.0), n_samples=n)
    assert abs(r.values[0] - 0.5) < 0.01 and abs(r.values[1] - 1 / 3) < 0.01
    r = mc.integrate([lambda x: x, lambda x: x**2], D().exponential(2.0), n_samples=n)
    assert abs(r.values[0] - 0.5) < 0.01 and abs(r.values[1] - 0.5) < 0.01
    r = mc.integrate([lambda x: math.sin(x), lambda x: math.cos(x)], D().uniform(0.0, 2 * math.pi), n_samples=n)
    assert abs(r.values[0]) < 0.01 and abs(r.values[1]) < 0.01
    r = mc.integrate([lambda x: x, lambda x: (x - 5.0) ** 2], D().normal(5.0, 2.0), n_samples=n)
    assert abs(r.values[0] - 5.0) < 0.01 and abs(r.values[1] - 4.0) < 0.02


def test_convenience_function_and_target_threads():                    # :309-357; test_importance_sampling.py:414-428
    from wgpu_montecarlo import integrate

    a = integrate([lambda x: x**2], D().normal(0.0, 1.0), n_samples=1_000_000, seed=7)
    b = integrate([lambda x: x**2], D().normal(0.0, 1.0), n_samples=1_000_000, seed=7, target_threads=32768)
    assert a.meta["n_eff"] == 65536 * 16 and b.meta["n_eff"] == 32768 * 31
    assert abs(a.values[0] - 1.0) < 0.01 and abs(b.values[0] - 1.0) < 0.01
    assert a.values[0] != b.values[0]                     # T changes the sample indexing, as in the reference


# ---- tests/test_distributions.py -----------------------------------------------------------------
def test_beta_moments_and_table_vs_direct(mc):                         # :78-157
    r = mc.integrate([lambda x: x, lambda x: x**2, lambda x: x**3], D().beta(2.0, 5.0), n_samples=10_000_000)
    assert np.all(np.abs(r.values - [2 / 7, 3 / 28, 1 / 21]) < 0.01)
    table_u = D().from_pdf(lambda x: 1.0 if 0 <= x <= 1 else 0.0, support=(0.0, 1.0))
    a = mc.integrate([lambda x: x, lambda x: x**2], table_u, n_samples=5_000_000)
    b = mc.integrate([lambda x: x, lambda x: x**2], D().uniform(0.0, 1.0), n_samples=5_000_000)
    assert np.all(np.abs(a.values - b.values) < 0.01)
    small = D().beta(2.0, 2.0, table_size=100)                          # :351-360 minimum 1000 points
    assert small.params["table_size"] == 1000
    r = mc.integrate([lambda x: x], small, n_samples=2_000_000)
    assert abs(r.values[0] - 0.5) < 0.01


# ---- tests/test_importance_sampling.py -----------------------------------------------------------
def test_importance_sampling_cases(mc):                                # :34-131
    fns = [lambda x: x, lambda x: x**2, lambda x: x**4]
    r = mc.integrate_importance_sampling(fns, D().normal(0.0, 1.0), D().normal(0.5, 1.5), n_samples=5_000_000)
    assert abs(r.values[0]) < 0.05 and abs(r.values[1] - 1.0) < 0.05 and abs(r.values[2] - 3.0) < 0.3
    r = mc.integrate_importance_sampling([lambda x: x], D().exponential(1.0), D().exponential(0.5), n_samples=5_000_000)
    assert abs(r.values[0] - 1.0) < 0.05
    r = mc.integrate_importance_sampling([lambda x: x, lambda x: x**2], D().uniform(0.0, 1.0), D().uniform(-0.5, 1.5),
                                         n_samples=5_000_000)
    assert abs(r.values[0] - 0.5) < 0.05 and abs(r.values[1] - 1 / 3) < 0.05
    # rare-event tail probability P(X > 3), X ~ N(0,1), proposal centred on the tail
    r = mc.integrate_importance_sampling([lambda x: x > 3.0], D().normal(0.0, 1.0), D().normal(3.5, 1.0),
                                         n_samples=5_000_000)
    assert abs(r.values[0] - 0.0013499) < 2e-5


def test_importance_sampling_custom_pdfs(mc):                          # :155-363
    def tri(x):                                                         # transpilable: if / return body
        if x < 0.0:
            return 0.0
        if x > 1.0:
            return 0.0
        return 2.0 * x

    target = D().from_pdf(tri, support=(0.0, 1.0))
    r = mc.integrate_importance_sampling([lambda x: x], target, D().uniform(0.0, 1.0), n_samples=5_000_000)
    assert abs(r.values[0] - 2 / 3) < 0.01

    def not_transpilable(x):                                            # int() -> table path (reference :291-299)
        return float(int(0.0 <= x <= 1.0)) * 2.0 * x

    r = mc.integrate_importance_sampling([lambda x: x], D().from_pdf(not_transpilable, support=(0.0, 1.0)),
                                         D().uniform(0.0, 1.0), n_samples=2_000_000)
    assert len(r.values) == 1 and abs(r.values[0] - 2 / 3) < 0.01
    for size in (100, 500, 1000):                                       # :348-363
        x = np.linspace(-5, 5, size)
        t = D().from_pdf_table(x, np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi))
        r = mc.integrate_importance_sampling([lambda x: x**2], t, D().normal(0.0, 2.0), n_samples=2_000_000)
        assert abs(r.values[0] - 1.0) < 0.05
    # WGSL-string integrand inside importance sampling (:133-148)
    r = mc.integrate_importance_sampling(["fn f(x: f32) -> f32 { return x * x; }"], D().normal(0.0, 1.0),
                                         D().normal(0.0, 2.0), n_samples=2_000_000)
    assert abs(r.values[0] - 1.0) < 0.05


# ---- tests/test_mcmc.py ---------------------------------------------------------------------------
def test_mcmc_cases(mc):                                               # :91-299
    fns = [lambda x: x, lambda x: x**2]
    r = mc.integrate_mcmc(fns, D().normal(0.0, 1.0), D().normal(0.0, 1.5), n_steps=10_000, n_chains=1024, n_burnin=1000)
    assert abs(r.values[0]) < 0.1 and abs(r.values[1] - 1.0) < 0.1
    assert r.n_samples == 1024 * 10_000 and 0.7 < r.meta["accept_rate"] < 0.8
    r = mc.integrate_mcmc(fns, D().normal(0.0, 1.0), D().normal(0.0, 1.0), n_steps=2000, n_chains=256, n_burnin=0)
    assert r.meta["accept_rate"] == 1.0                                 # p == q: every proposal accepted (:94)
    r = mc.integrate_mcmc(fns, D().exponential(1.0), D().exponential(0.5), n_steps=5000, n_chains=512)
    assert abs(r.values[0] - 1.0) < 0.1 and abs(r.values[1] - 2.0) < 0.3
    r = mc.integrate_mcmc(fns, D().uniform(0.0, 1.0), D().uniform(0.0, 1.0), n_steps=2000, n_chains=256)
    assert abs(r.values[0] - 0.5) < 0.05
    r = mc.integrate_mcmc([lambda x: x], D().normal(0.0, 1.0), D().normal(0.0, 2.0), n_steps=3000, n_chains=1)
    assert r.meta["n_eff"] == 256 * 3000 and abs(r.values[0]) < 0.1     # one chain still runs 256 (:283-299)


def test_mcmc_custom_targets_and_reproducibility(mc):                  # :319-392
    bim = D().from_pdf(lambda x: 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2)), support=(-10, 10))
    a = mc.integrate_mcmc([lambda x: x, lambda x: x**2], bim, D().normal(0.0, 2.0), n_steps=5000, n_chains=1024, seed=9)
    b = mc.integrate_mcmc([lambda x: x, lambda x: x**2], bim, D().normal(0.0, 2.0), n_steps=5000, n_chains=1024, seed=9)
    assert np.array_equal(a.values, b.values)                            # same seed -> identical (:319-344)
    assert abs(a.values[0]) < 0.3 and abs(a.values[1] - 5.0) < 0.3
    c = mc.integrate_mcmc([lambda x: x, lambda x: x**2], bim, D().normal(0.0, 2.0), n_steps=5000, n_chains=1024, seed=10)
    assert not np.array_equal(a.values, c.values)
    r = mc.integrate_mcmc([lambda x: x], D().beta(2.0, 5.0), D().uniform(0.0, 1.0), n_steps=5000, n_chains=512)
    assert abs(r.values[0] - 2 / 7) < 0.02


def test_closure_values_are_recaptured_between_calls(mc):
    """The lowering is cached per code object; captured constants must not be."""
    def make(a):
        return lambda x: a * x * x

    r2 = mc.integrate([make(2.0)], D().normal(0.0, 1.0), n_samples=1_000_000, seed=1)
    r5 = mc.integrate([make(5.0)], D().normal(0.0, 1.0), n_samples=1_000_000, seed=1)
    assert r5.values[0] == pytest.approx(2.5 * r2.values[0], rel=1e-6)


def test_math_modes_agree(mc):
    from wgpu_montecarlo import MonteCarloIntegrator

    fns = [lambda x: math.exp(-x * x / 2) / math.sqrt(2 * math.pi), lambda x: math.sin(x) / (1.0 + x * x)]
    vals = {m: MonteCarloIntegrator(math=m).integrate(fns, D().normal(0.0, 2.0), n_samples=2_000_000, seed=3).values
            for m in ("precise", "default", "fast")}
    assert np.allclose(vals["default"], vals["precise"], atol=2e-6)
    assert np.allclose(vals["fast"], vals["precise"], atol=2e-5)


def test_default_trig_and_pow_stay_within_their_stated_error_on_wide_arguments():
    """math="default" compiles sin / cos / tan to the range-reduced hardware instructions and pow to exp2(y log2|x|)
    (device/mcx_device.hpp mcx_sin .. mcx_pow). Mean absolute differences are compared sample by sample with "precise"
    (ocml) on the same counter stream: arguments from a few to 1e5 radians, past the 1e6 switch to ocml, negative bases
    with integral and fractional exponents. The pointwise bounds are in profiles/r03_trig_pow_accuracy.txt."""
    from wgpu_montecarlo import MonteCarloIntegrator

    def identities(scale):
        return [lambda x: np.abs(math.sin(scale * x) ** 2 + math.cos(scale * x) ** 2 - 1.0),
                lambda x: math.sin(scale * x), lambda x: math.cos(scale * x),
                lambda x: math.cos(scale * x) * math.tan(scale * x) - math.sin(scale * x)]

    for scale, dist in ((1.0, D().uniform(-10.0, 10.0)), (1000.0, D().uniform(-100.0, 100.0)), (3.0e6, D().uniform(-1.0, 1.0))):   # uniform: the same samples in both modes
        got = {m: MonteCarloIntegrator(math=m).integrate(identities(scale), dist, n_samples=1_000_000, seed=9).values for m in ("precise", "default")}
        assert got["default"][0] < 1.5e-6, (scale, got)                       # mean |sin^2 + cos^2 - 1|: two errors of <= 4e-7 each, doubled
        assert np.allclose(got["default"][1:3], got["precise"][1:3], atol=5e-7), (scale, got)
        assert abs(got["default"][3]) < 2e-6, (scale, got)
    powers = [lambda x: np.abs(x) ** 2.5, lambda x: x ** (2.0 + 0.0 * x), lambda x: x ** (3.0 + 0.0 * x), lambda x: (1.0 + x * x) ** (-0.75),
              lambda x: 2.0 ** x, lambda x: np.abs(x) ** 0.5, lambda x: x ** (0.0 * x), lambda x: (0.0 * x) ** (x * x)]
    got = {m: MonteCarloIntegrator(math=m).integrate(powers, D().uniform(-4.0, 5.0), n_samples=1_000_000, seed=9).values for m in ("precise", "default")}
    assert np.all(np.isfinite(got["precise"])) and np.allclose(got["default"], got["precise"], rtol=3e-6, atol=1e-7), got
    hyper = [lambda x: math.sinh(x), lambda x: math.cosh(x), lambda x: math.tanh(x) + 10.0 * math.tanh(0.02 * x) + math.tanh(20.0 * x), lambda x: math.sinh(0.01 * x) * 100.0, lambda x: math.cosh(0.5 * x) / (1.0 + math.sinh(np.abs(x)))]
    got = {m: MonteCarloIntegrator(math=m).integrate(hyper, D().uniform(-4.0, 5.0), n_samples=1_000_000, seed=9).values for m in ("precise", "default")}
    assert np.allclose(got["default"], got["precise"], rtol=2e-6, atol=2e-6), got
    r = MonteCarloIntegrator().integrate([lambda x: x ** 0.5], D().normal(0.0, 1.0), n_samples=100_000)
    assert np.isnan(r.values[0])                                                # fractional power of a negative base, as powf


def test_std_error_output():
    """std_error=True (extension): sum (f w)^2 accumulated in the same pass; standard errors match theory."""
    from wgpu_montecarlo import MonteCarloIntegrator

    mc = MonteCarloIntegrator(std_error=True)
    plain = MonteCarloIntegrator()
    fns = [lambda x: x, lambda x: x**2, lambda x: x**3, lambda x: x**4]
    r = mc.integrate(fns, D().normal(0.0, 1.0), n_samples=10_000_000, seed=42)
    # same samples, same sums up to the compiler's choice of fused multiply-adds in the two modules
    assert np.allclose(r.values, plain.integrate(fns, D().normal(0.0, 1.0), n_samples=10_000_000, seed=42).values,
                       rtol=0, atol=1e-7)
    want = np.sqrt(np.array([1.0, 2.0, 15.0, 96.0]) / r.meta["n_eff"])
    assert np.allclose(r.meta["std_error"], want, rtol=0.02)
    assert np.all(np.abs(r.values - [0, 1, 0, 3]) < 4 * r.meta["std_error"])
    r = mc.integrate_importance_sampling([lambda x: x > 3.0], D().normal(0.0, 1.0), D().normal(3.5, 1.0), n_samples=5_000_000)
    assert abs(r.values[0] - 0.0013499) < 4 * r.meta["std_error"][0] and r.meta["std_error"][0] < 3e-6
    assert "std_error" not in plain.integrate(fns, D().normal(0.0, 1.0), n_samples=100_000).meta


def test_convenience_calls_reuse_the_device_engine():
    """integrate() builds a new MonteCarloIntegrator per call (as in the reference); engines, loaded modules and
    resident tables are shared per device, so the second call costs no device initialisation or module load."""
    import time

    from wgpu_montecarlo import MonteCarloIntegrator, integrate

    f = lambda x: x * x
    integrate([f], D().normal(0.0, 1.0), n_samples=100_000)
    t0 = time.perf_counter()
    for _ in range(20):
        r = integrate([f], D().normal(0.0, 1.0), n_samples=100_000)
    per_call_ms = (time.perf_counter() - t0) / 20 * 1e3
    assert abs(r.values[0] - 1.0) < 0.02
    assert MonteCarloIntegrator()._engine is MonteCarloIntegrator()._engine
    assert per_call_ms < 5.0, per_call_ms


def test_oversubscribed_reference_stream_warns():
    from wgpu_montecarlo import MonteCarloIntegrator

    mc = MonteCarloIntegrator()
    with pytest.warns(UserWarning, match="exceeds the 2\\^32 counter space"):
        r = mc.integrate([lambda x: x * x], D().normal(0.0, 1.0), n_samples=5_000_000_000)
    assert abs(r.values[0] - 1.0) < 1e-3
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("error")
        MonteCarloIntegrator(rng="philox").integrate([lambda x: x * x], D().normal(0.0, 1.0), n_samples=5_000_000_000)


def test_concurrent_host_threads_share_one_engine(mc):
    """SURVEY 8(b) threading row: ctypes releases the GIL, the engine serialises launches with its own mutex. Four
    threads hammer the shared engine with different calls; every result equals the single-threaded one."""
    import threading

    Distribution = D()
    fns = [lambda x: x, lambda x: x * x]
    jobs = [("k1", Distribution.normal(0.0, 1.0), 300_001, 3), ("k1", Distribution.beta(2.0, 5.0), 200_003, 4),
            ("k1", Distribution.uniform(-1.0, 2.0), 100_001, 5), ("k3", None, 0, 6)]

    def run(job):
        kind, dist, n, seed = job
        if kind == "k1":
            return mc.integrate(fns, dist, n_samples=n, seed=seed).values
        return mc.integrate_mcmc(fns, Distribution.normal(0.3, 1.0), Distribution.normal(0.0, 2.0), n_steps=200,
                                 n_chains=512, n_burnin=20, seed=seed).values

    expected = [run(j) for j in jobs]
    errors = []

    def worker(i):
        try:
            for rep in range(25):
                j = (i + rep) % len(jobs)
                got = run(jobs[j])
                if not np.array_equal(got, expected[j]):
                    errors.append((i, rep, j, got, expected[j]))
        except Exception as exc:      # noqa: BLE001
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_auto_stream_uses_the_parity_stream_within_its_counter_space_and_philox_beyond():
    """MonteCarloIntegrator(rng="auto"): a call of <= 2^32 draws is the reference stream's call bit for bit; a larger one
    (here 6e9 samples of K = 2) runs on Philox, says so in meta["rng"], does not warn, and lands within 3 sigma."""
    import warnings

    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    fns = [lambda x: x, lambda x: x * x]
    dist = Distribution.normal(0.0, 1.0)
    auto, ref = MonteCarloIntegrator(rng="auto"), MonteCarloIntegrator(rng="pcg_ref")
    a, b = auto.integrate(fns, dist, n_samples=10**7, seed=3), ref.integrate(fns, dist, n_samples=10**7, seed=3)
    assert a.meta["rng"] == b.meta["rng"] == "pcg_ref" and np.array_equal(a.values, b.values)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # the oversubscription warning must not fire for the auto policy
        big = auto.integrate(fns, dist, n_samples=6 * 10**9, seed=3)
    assert big.meta["rng"] == "philox"
    sigma = np.sqrt(np.array([1.0, 2.0]) / big.meta["n_eff"])
    assert np.all(np.abs(big.values - [0.0, 1.0]) < 3.5 * sigma), big.values
    mh = auto.integrate_mcmc(fns, Distribution.normal(0.5, 1.0), Distribution.normal(0.0, 2.0), n_steps=50, n_chains=4096, n_burnin=10)
    assert mh.meta["rng"] == "pcg_ref"


def test_repeat_calls_stay_cheap_however_the_call_is_written(mc):
    """A repeat call is a dictionary lookup and a launch: tens of microseconds, whether the distribution object is kept or built
    inline in the call (as the reference's examples and benchmark write it), for plain, importance-sampling and MCMC calls alike.
    (A plan-cache key that parsed a density's source on every call once made the importance-sampling call 1.25 ms: this is its
    regression test. Bound: 0.4 ms per call; measured 0.03-0.07.)"""
    import time

    fns = [lambda x: x, lambda x: x**2]
    xs = np.linspace(0, 10, 512)
    target = D().from_pdf_table(xs, np.exp(-xs))
    bimodal = D().from_pdf(lambda x: 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2)), support=(-10, 10))
    calls = {
        "integrate, inline normal": lambda: mc.integrate(fns, D().normal(0.0, 1.0), n_samples=100_000),
        "integrate, inline beta": lambda: mc.integrate(fns, D().beta(2.0, 5.0), n_samples=100_000),
        "importance sampling, table target, inline normal": lambda: mc.integrate_importance_sampling(fns, target, D().normal(2.0, 3.0), n_samples=100_000),
        "mcmc, table target, inline normal": lambda: mc.integrate_mcmc(fns, bimodal, D().normal(0.0, 2.0), n_steps=100, n_chains=4096, n_burnin=10),
    }
    for name, call in calls.items():
        call()
        call()
        times = []
        for _ in range(30):
            t0 = time.perf_counter()
            call()
            times.append(time.perf_counter() - t0)
        assert float(np.median(times)) < 4e-4, (name, float(np.median(times)))
