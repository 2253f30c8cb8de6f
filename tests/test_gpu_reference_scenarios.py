"""Every GPU scenario of the reference's own test-suite, one table row each, run through this package on the MI355X.

The reference's tests cannot travel to the GPU box (and its sources are not copied): each row below RESTATES one of its
test functions -- the same call, the same sizes and seeds, the same acceptance bar -- and names it (file::Class::test and
the line it starts at in /root/reference/tests). tests/test_reference_suite.py runs the originals in the build container
(where, without a GPU, all of these stop at "Failed to initialize GPU"); this file is what shows they hold on the device.
Rows are data: (reference test id, build-and-run, [checks on result.values]); nothing here knows about libmcx.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _api():
    import wgpu_montecarlo as w

    return w


# ---- integrands (named: several reference tests use module-level functions, test_distributions.py:21-34) ---------------
def ident(x):
    return x


def square(x):
    return x * x


def cube(x):
    return x * x * x


def fourth(x):
    return x * x * x * x


def gauss_pdf(x):
    return math.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


def box_pdf(x):
    return 1.0 if 0 <= x < 1 else 0.0


COEFF = 2.0
near = lambda i, want, tol: (lambda v: abs(v[i] - want) < tol)
var_near = lambda want, tol: (lambda v: abs((v[1] - v[0] ** 2) - want) < tol)
count = lambda k: (lambda v: len(v) == k)

# ---- K1: integrate() ---------------------------------------------------------------------------------------------------
# (reference test, distribution factory, functions, n_samples, seed, checks); "conv" rows go through the module-level
# integrate() convenience function like the reference test does
N01 = lambda D: D.normal(mean=0.0, std=1.0)
U01 = lambda D: D.uniform(min=0.0, max=1.0)
K1 = [
    ("test_integrator.py::TestMonteCarloIntegrator::test_single_function:24", N01, [lambda x: x], 10**6, 42, [count(1), near(0, 0.0, 0.1)]),
    ("test_integrator.py::TestMonteCarloIntegrator::test_multiple_functions:35", N01, [lambda x: x, lambda x: x**2, lambda x: x**3], 10**6, 42,
     [count(3), near(0, 0.0, 0.1), near(1, 1.0, 0.1), near(2, 0.0, 0.1)]),
    ("test_integrator.py::TestMonteCarloIntegrator::test_wgsl_string_function:48", N01, ["fn f(x: f32) -> f32 { return x * x; }"], 10**6, 42,
     [near(0, 1.0, 0.1)]),
    ("test_integrator.py::TestMonteCarloIntegrator::test_mixed_callable_and_wgsl:58", N01, [lambda x: x, "fn f(x: f32) -> f32 { return x * x; }"],
     10**6, 42, [count(2), near(0, 0.0, 0.1), near(1, 1.0, 0.1)]),
    ("test_integrator.py::TestInlineLambdas::test_inline_lambdas_in_function_call:94", N01, [lambda x: x, lambda x: x**2], 10**6, 42,
     [near(0, 0.0, 0.1), near(1, 1.0, 0.1)]),
    ("test_integrator.py::TestInlineLambdas::test_inline_lambdas_four_functions:106", N01,
     [lambda x: x, lambda x: x**2, lambda x: x**3, lambda x: x**4], 10**6, 42, [near(0, 0.0, 0.1), near(1, 1.0, 0.1), near(2, 0.0, 0.1), near(3, 3.0, 0.1)]),
    ("test_integrator.py::TestInlineLambdas::test_inline_lambdas_with_global_variables:135", N01, [lambda x: COEFF * x, lambda x: COEFF * x**2], 10**6, 42,
     [near(0, 0.0, 0.1), near(1, COEFF, 0.1)]),
    ("test_integrator.py::TestInlineLambdas::test_inline_lambdas_with_constants:152", N01, [lambda x: math.pi, lambda x: math.e], 10**6, 42,
     [near(0, math.pi, 0.1), near(1, math.e, 0.1)]),
    ("test_integrator.py::TestIntegrationAccuracy::test_normal_mean_and_variance:181", N01, [lambda x: x, lambda x: x * x], 10**7, 42,
     [near(0, 0.0, 0.01), var_near(1.0, 0.01)]),
    ("test_integrator.py::TestIntegrationAccuracy::test_uniform_mean_and_variance:196", U01, [lambda x: x, lambda x: x * x], 10**7, 42,
     [near(0, 0.5, 0.01), var_near(1.0 / 12.0, 0.01)]),
    ("test_integrator.py::TestIntegrationAccuracy::test_exponential_mean_and_variance:211", lambda D: D.exponential(lambda_param=2.0),
     [lambda x: x, lambda x: x * x], 10**7, 42, [near(0, 0.5, 0.01), var_near(0.25, 0.01)]),
    ("test_integrator.py::TestIntegrationAccuracy::test_moment_calculations:230", N01,
     [lambda x: x, lambda x: x * x, lambda x: x * x * x, lambda x: x * x * x * x], 10**7, 42,
     [near(0, 0.0, 0.01), near(1, 1.0, 0.01), near(2, 0.0, 0.01), near(3, 3.0, 0.01)]),
    ("test_integrator.py::TestIntegrationAccuracy::test_trigonometric_expectations:248", lambda D: D.uniform(min=0.0, max=2 * math.pi),
     [lambda x: math.sin(x), lambda x: math.cos(x)], 10**7, 42, [near(0, 0.0, 0.01), near(1, 0.0, 0.01)]),
    ("test_integrator.py::TestIntegrationWithConstants::test_pi_constant:314", U01, [lambda x: math.pi * x], 10**7, 42, [near(0, math.pi * 0.5, 0.01)]),
    ("test_integrator.py::TestIntegrationWithConstants::test_e_constant:325", N01, [lambda x: math.e], 10**7, 42, [near(0, math.e, 0.01)]),
    ("test_integrator.py::TestIntegratorConvenience::test_integrate_function:340 conv", N01, [lambda x: x, lambda x: x * x], 10**6, 42,
     [count(2), near(0, 0.0, 0.1), near(1, 1.0, 0.1)]),
    ("test_distributions.py::TestBetaDistribution::test_beta_2_5:78", lambda D: D.beta(2.0, 5.0, table_size=2048), [ident, square, cube], 10**7, 42,
     [near(0, 2 / 7, 0.01), near(1, 6 / 56, 0.01), near(2, 24 / 504, 0.01)]),
    ("test_distributions.py::TestBetaDistribution::test_beta_convenience_method:112 conv", lambda D: D.beta(3.0, 2.0, table_size=2048), [ident, square],
     5 * 10**6, 123, [near(0, 0.6, 0.02), var_near(0.4 - 0.36, 0.02)]),
    ("test_distributions.py::TestBetaDistribution::test_table_vs_direct:134 conv (table)", lambda D: D.from_pdf(box_pdf, support=(0.0, 1.0)), [ident, square],
     10**6, 42, [near(0, 0.5, 0.01), near(1, 1 / 3, 0.01)]),
    ("test_distributions.py::TestBetaDistribution::test_table_vs_direct:134 conv (direct)", lambda D: D.uniform(0.0, 1.0), [ident, square], 10**6, 42,
     [near(0, 0.5, 0.01), near(1, 1 / 3, 0.01)]),
    ("test_distributions.py::TestUniformDistribution::test_uniform_mean:164 conv", U01, [ident], 10**7, 42, [near(0, 0.5, 0.01)]),
    ("test_distributions.py::TestUniformDistribution::test_uniform_variance:171 conv", U01, [ident, square], 10**7, 42, [var_near(1 / 12, 0.01)]),
    ("test_distributions.py::TestNormalDistribution::test_normal_mean:188 conv", N01, [ident], 10**7, 42, [near(0, 0.0, 0.01)]),
    ("test_distributions.py::TestNormalDistribution::test_normal_variance:195 conv", N01, [ident, square], 10**7, 42, [var_near(1.0, 0.01)]),
    ("test_distributions.py::TestNormalDistribution::test_normal_higher_moments:206 conv", N01, [fourth], 10**7, 42, [near(0, 3.0, 0.01)]),
    ("test_distributions.py::TestNormalDistribution::test_normal_with_mean_and_std:213 conv", lambda D: D.normal(mean=5.0, std=2.0), [ident, square], 10**7, 42,
     [near(0, 5.0, 0.01), var_near(4.0, 0.01)]),
    ("test_distributions.py::TestExponentialDistribution::test_exponential_mean:233 conv", lambda D: D.exponential(lambda_param=2.0), [ident], 10**7, 42,
     [near(0, 0.5, 0.01)]),
    ("test_distributions.py::TestExponentialDistribution::test_exponential_variance:242 conv", lambda D: D.exponential(lambda_param=2.0), [ident, square],
     10**7, 42, [var_near(0.25, 0.01)]),
    ("test_distributions.py::TestCustomDistribution::test_custom_distribution_integration:296 conv", lambda D: D.from_pdf(gauss_pdf), [ident, square],
     5 * 10**6, 42, [near(0, 0.0, 0.02), near(1, 1.0, 0.02)]),
    ("test_distributions.py::TestCustomDistribution::test_custom_distribution_with_manual_support:309 conv", lambda D: D.from_pdf(box_pdf, support=(0.0, 1.0)),
     [ident], 10**6, 42, [near(0, 0.5, 0.02)]),
    ("test_distributions.py::TestVariableTableSize::test_table_size_4096_accuracy:339 conv",
     lambda D: D.from_pdf(gauss_pdf, support=(-5.0, 5.0), table_size=4096), [square], 5 * 10**6, 42, [near(0, 1.0, 0.02)]),
]


@pytest.mark.parametrize("ref,make_dist,fns,n,seed,checks", K1, ids=[row[0] for row in K1])
def test_reference_integrate_scenario(ref, make_dist, fns, n, seed, checks):
    w = _api()
    dist = make_dist(w.Distribution)
    if ref.split(" ")[-1].startswith("conv") or " conv" in ref:
        res = w.integrate(fns, dist, n_samples=n, seed=seed)
    else:
        res = w.MonteCarloIntegrator().integrate(fns, dist, n_samples=n, seed=seed)
    assert all(check(res.values) for check in checks), (ref, res.values)


def test_reference_integrate_closures_and_locals():
    """test_integrator.py: TestIntegrationAccuracy::test_polynomial_expectation:169, TestInlineLambdas::
    test_tuple_unpacking_lambdas:123, TestIntegrationWithGlobalVariables::test_global_variable_in_lambda:264,
    ::test_global_variable_polynomial:278, ::test_closure_capture:293 -- integrands that capture LOCAL variables of the test
    function (they cannot live in a module-level table)."""
    w = _api()
    mc, dist = w.MonteCarloIntegrator(), w.Distribution.normal(mean=0.0, std=1.0)
    a, b, c = 1.0, 2.0, 3.0
    assert abs(mc.integrate([lambda x: a * x * x + b * x + c], dist, n_samples=10**7, seed=42).values[0] - 4.0) < 0.1
    f1, f2 = lambda x: x, lambda x: x**2
    v = mc.integrate([f1, f2], dist, n_samples=10**6, seed=42).values
    assert abs(v[0]) < 0.1 and abs(v[1] - 1.0) < 0.1
    p, q = 2.0, 1.0
    func = lambda x: p * x + q
    assert abs(mc.integrate([func], dist, n_samples=10**7, seed=42).values[0] - 1.0) < 0.01
    coeff_a, coeff_b, coeff_c = 1.0, 2.0, 3.0
    poly = lambda x: coeff_a * x * x + coeff_b * x + coeff_c
    assert abs(mc.integrate([poly], dist, n_samples=10**7, seed=42).values[0] - 4.0) < 0.1

    def make_func(s, t):
        return lambda x: s * x + t

    assert abs(mc.integrate([make_func(2.0, 1.0)], dist, n_samples=10**7, seed=42).values[0] - 1.0) < 0.01


def test_reference_integrate_errors():
    """test_integrator.py::TestMonteCarloIntegrator::test_init:19, ::test_empty_functions_error:73, ::test_invalid_function_type_error:81."""
    w = _api()
    mc, dist = w.MonteCarloIntegrator(), w.Distribution.normal(mean=0.0, std=1.0)
    assert mc is not None
    with pytest.raises(ValueError):
        mc.integrate([], dist, n_samples=1000)
    with pytest.raises(TypeError):
        mc.integrate([123], dist, n_samples=1000)


# ---- K2: integrate_importance_sampling() -----------------------------------------------------------------------------------
def trunc_exp_pdf(x: float) -> float:
    if (x >= 0) and (x < 5):
        return math.exp(-x)
    return 0.0


def shifted_exp_pdf(x: float) -> float:
    if (x >= 1) and (x < 6):
        return math.exp(-(x - 1))
    return 0.0


def flat_1_6_pdf(x: float) -> float:
    if (x >= 1) and (x < 6):
        return 0.2
    return 0.0


def gauss_2_05_pdf(x: float) -> float:
    z = (x - 2.0) / 0.5
    return math.exp(-0.5 * z * z) / (0.5 * math.sqrt(2 * math.pi))


def trunc_normal_pdf(x: float) -> float:
    if (x >= -2) and (x < 2):
        z = x / 1.0
        return math.exp(-0.5 * z * z)
    return 0.0


def power_law_pdf(x: float) -> float:
    if (x >= 1) and (x < 10):
        return 0.1 * math.pow(x, -0.5)
    return 0.0


def steps_pdf(x: float) -> float:                    # outside the transpiler's subset (int(), %): forces the PDF-table path
    return float(int(x) % 2) * 0.5 + 0.1


def steps2_pdf(x: float) -> float:
    return float(int(x * 2) % 3) * 0.3 + 0.1


def ramp_pdf(x: float) -> float:
    if (x >= 0) and (x < 2):
        return 0.5 * x
    return 0.0


K2 = [
    ("test_importance_sampling.py::TestImportanceSamplingBasic::test_identical_distributions:23", lambda D: (D.normal(0.0, 1.0), D.normal(0.0, 1.0)),
     [lambda x: x], 10**6, [near(0, 0.0, 0.01)]),
    ("test_importance_sampling.py::TestImportanceSamplingBasic::test_shifted_proposal:34", lambda D: (D.normal(0.0, 1.0), D.normal(1.0, 1.0)),
     [lambda x: x * x], 5 * 10**6, [near(0, 1.0, 0.05)]),
    ("test_importance_sampling.py::TestImportanceSamplingBasic::test_different_variance_proposal:49", lambda D: (D.normal(0.0, 1.0), D.normal(0.0, 2.0)),
     [lambda x: x * x], 5 * 10**6, [near(0, 1.0, 0.1)]),
    ("test_importance_sampling.py::TestImportanceSamplingMixedDistributions::test_normal_target_uniform_proposal:69",
     lambda D: (D.normal(0.5, 0.2), D.uniform(0.0, 1.0)), [lambda x: x], 5 * 10**6, [near(0, 0.5, 0.1)]),
    ("test_importance_sampling.py::TestImportanceSamplingMixedDistributions::test_uniform_target_uniform_proposal:80",
     lambda D: (D.uniform(0.0, 0.5), D.uniform(0.0, 1.0)), [lambda x: x], 5 * 10**6, [near(0, 0.25, 0.05)]),
    ("test_importance_sampling.py::TestImportanceSamplingMixedDistributions::test_exponential_mixture:95",
     lambda D: (D.exponential(2.0), D.exponential(1.0)), [lambda x: x], 5 * 10**6, [near(0, 0.5, 0.1)]),
    ("test_importance_sampling.py::TestImportanceSamplingMultipleFunctions::test_multiple_functions_same_weight:115",
     lambda D: (D.normal(0.0, 1.0), D.normal(0.5, 1.5)), [lambda x: x, lambda x: x * x, lambda x: x * x * x], 5 * 10**6,
     [near(0, 0.0, 0.1), near(1, 1.0, 0.1), near(2, 0.0, 0.1)]),
    ("test_importance_sampling.py::TestImportanceSamplingMultipleFunctions::test_mixed_callable_and_wgsl:133",
     lambda D: (D.normal(0.0, 1.0), D.normal(0.5, 1.0)), [lambda x: x, "fn f(x: f32) -> f32 { return x * x; }"], 5 * 10**6,
     [near(0, 0.0, 0.1), near(1, 1.0, 0.1)]),
    ("test_importance_sampling.py::TestImportanceSamplingCustomPDF::test_custom_target_transpilable_pdf:155",
     lambda D: (D.from_pdf(trunc_exp_pdf, support=(0.0, 5.0)), D.uniform(0.0, 5.0)), [lambda x: x], 5 * 10**6, [near(0, 1.0 - 6.0 * math.exp(-5.0), 0.15)]),
    ("test_importance_sampling.py::TestImportanceSamplingCustomPDF::test_custom_proposal_transpilable_pdf:180",
     lambda D: (D.uniform(0.0, 5.0), D.from_pdf(trunc_exp_pdf, support=(0.0, 5.0))), [lambda x: x], 5 * 10**6, [near(0, 2.5, 0.3)]),
    ("test_importance_sampling.py::TestImportanceSamplingCustomPDF::test_custom_both_transpilable_pdf:199",
     lambda D: (D.from_pdf(shifted_exp_pdf, support=(1.0, 6.0)), D.from_pdf(flat_1_6_pdf, support=(1.0, 6.0))), [lambda x: 1.0], 5 * 10**6,
     [near(0, 1.0, 0.1)]),
    ("test_importance_sampling.py::TestImportanceSamplingCustomPDF::test_custom_pdf_with_math_functions:222",
     lambda D: (D.from_pdf(gauss_2_05_pdf, support=(0.0, 4.0)), D.uniform(0.0, 4.0)), [lambda x: x], 5 * 10**6, [near(0, 2.0, 0.2)]),
    ("test_importance_sampling.py::TestImportanceSamplingCustomPDF::test_custom_pdf_truncated_normal_moments:239",
     lambda D: (D.from_pdf(trunc_normal_pdf, support=(-2.0, 2.0)), D.uniform(-2.0, 2.0)), [lambda x: x, lambda x: x * x], 5 * 10**6,
     [near(0, 0.0, 0.1), lambda v: v[1] > 0]),
    ("test_importance_sampling.py::TestImportanceSamplingCustomPDF::test_custom_pdf_with_power_function:263",
     lambda D: (D.from_pdf(power_law_pdf, support=(1.0, 10.0)), D.uniform(1.0, 10.0)), [lambda x: x], 5 * 10**6, [lambda v: 1.0 < v[0] < 10.0]),
    ("test_importance_sampling.py::TestImportanceSamplingWithPDFTables::test_non_transpilable_target_uses_table:287",
     lambda D: (D.from_pdf(steps_pdf, support=(0.0, 10.0)), D.uniform(0.0, 10.0)), [lambda x: 1.0], 10**6, [count(1)]),
    ("test_importance_sampling.py::TestImportanceSamplingWithPDFTables::test_non_transpilable_proposal_uses_table:302",
     lambda D: (D.normal(0.5, 0.2), D.from_pdf(steps_pdf, support=(0.0, 10.0))), [lambda x: 1.0], 10**6, [count(1)]),
    ("test_importance_sampling.py::TestImportanceSamplingWithPDFTables::test_both_non_transpilable_uses_tables:317",
     lambda D: (D.from_pdf(steps_pdf, support=(0.0, 10.0)), D.from_pdf(steps2_pdf, support=(0.0, 10.0))), [lambda x: 1.0], 10**6, [count(1)]),
    ("test_importance_sampling.py::TestImportanceSamplingWithPDFTables::test_from_pdf_table_api:335",
     lambda D: (D.from_pdf_table(np.linspace(0, 10, 512), np.exp(-np.linspace(0, 10, 512))), D.uniform(0, 10)), [lambda x: 1.0], 10**6, [count(1)]),
    ("test_importance_sampling.py::TestConvenienceFunction::test_integrate_importance_sampling_function:397 conv",
     lambda D: (D.normal(0.0, 1.0), D.normal(0.0, 1.0)), [lambda x: x, lambda x: x * x], 10**6, [count(2), near(0, 0.0, 0.1), near(1, 1.0, 0.1)]),
    ("test_importance_sampling.py::TestConvenienceFunction::test_convenience_with_custom_pdf:430 conv",
     lambda D: (D.from_pdf(ramp_pdf, support=(0.0, 2.0)), D.uniform(0.0, 2.0)), [lambda x: x], 5 * 10**6, [near(0, 4.0 / 3.0, 0.2)]),
]


@pytest.mark.parametrize("ref,make,fns,n,checks", K2, ids=[row[0] for row in K2])
def test_reference_importance_sampling_scenario(ref, make, fns, n, checks):
    w = _api()
    target, proposal = make(w.Distribution)
    if ref.endswith("conv"):
        res = w.integrate_importance_sampling(fns, target, proposal, n_samples=n, seed=42)
    else:
        res = w.MonteCarloIntegrator().integrate_importance_sampling(fns, target, proposal, n_samples=n, seed=42)
    assert np.all(np.isfinite(res.values)) and all(check(res.values) for check in checks), (ref, res.values)


def test_reference_importance_sampling_table_sizes_threads_and_errors():
    """test_importance_sampling.py::TestImportanceSamplingWithPDFTables::test_arbitrary_table_size:348 (100-, 500-, 1000-point
    PDF tables of a non-transpilable density), ::TestConvenienceFunction::test_with_target_threads_parameter:414
    (target_threads=32768), ::TestImportanceSamplingErrors::test_empty_functions_error:370, ::test_invalid_function_type_error:381."""
    w = _api()
    D = w.Distribution

    def teeth_pdf(x: float) -> float:
        return float(int(x * 10) % 7) * 0.1 + 0.05

    for size in (100, 500, 1000):
        res = w.MonteCarloIntegrator().integrate_importance_sampling([lambda x: 1.0], D.from_pdf(teeth_pdf, support=(0.0, 10.0), table_size=size),
                                                                     D.uniform(0.0, 10.0), n_samples=100_000, seed=42)
        assert len(res.values) == 1 and np.isfinite(res.values[0])
    res = w.integrate_importance_sampling([lambda x: x], D.normal(0.0, 1.0), D.normal(0.0, 1.0), n_samples=100_000, seed=42, target_threads=32768)
    assert abs(res.values[0]) < 0.1 and res.meta["n_eff"] == 32768 * 4
    with pytest.raises(ValueError):
        w.MonteCarloIntegrator().integrate_importance_sampling([], D.normal(0.0, 1.0), D.normal(0.0, 1.0), n_samples=1000)
    with pytest.raises(TypeError):
        w.MonteCarloIntegrator().integrate_importance_sampling([123], D.normal(0.0, 1.0), D.normal(0.0, 1.0), n_samples=1000)


# ---- K3: integrate_mcmc() ------------------------------------------------------------------------------------------------
def bimodal_pdf(x):
    return 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2))


# (reference test, (target, proposal) factory, functions, n_steps, n_chains, n_burnin, seed, checks)
K3 = [
    ("test_mcmc.py::TestMcmcBasic::test_mcmc_normal_mean:91", lambda D: (D.normal(0.0, 1.0), D.normal(0.0, 1.0)), [lambda x: x], 5000, 256, 500, 42,
     [near(0, 0.0, 0.15)]),
    ("test_mcmc.py::TestMcmcBasic::test_mcmc_normal_second_moment:109", lambda D: (D.normal(0.0, 1.0), D.normal(0.0, 1.5)), [lambda x: x**2], 10000, 512, 1000, 42,
     [near(0, 1.0, 0.15)]),
    ("test_mcmc.py::TestMcmcBasic::test_mcmc_multiple_functions:127", lambda D: (D.normal(0.0, 1.0), D.normal(0.0, 1.0)),
     [lambda x: x, lambda x: x**2, lambda x: x**3], 5000, 256, 500, 42, [count(3), near(0, 0.0, 0.15), near(1, 1.0, 0.15), near(2, 0.0, 0.2)]),
    ("test_mcmc.py::TestProposalDistribution::test_same_proposal_as_target:223", lambda D: (D.normal(0.0, 1.0), D.normal(0.0, 1.0)),
     [lambda x: x, lambda x: x**2], 10000, 256, 500, 42, [near(0, 0.0, 0.1), near(1, 1.0, 0.1)]),
    ("test_mcmc.py::TestProposalDistribution::test_wider_proposal:242", lambda D: (D.normal(0.0, 1.0), D.normal(0.0, 2.0)), [lambda x: x**2], 10000, 256, 1000, 42,
     [near(0, 1.0, 0.2)]),
    ("test_mcmc.py::TestProposalDistribution::test_uniform_proposal:260", lambda D: (D.normal(0.0, 1.0), D.uniform(-5.0, 5.0)), [lambda x: x**2], 10000, 256, 1000, 42,
     [lambda v: 0.5 < v[0] < 1.5]),
    ("test_mcmc.py::TestMultiChain::test_single_chain:283", lambda D: (D.normal(0.0, 1.0), D.normal(0.0, 1.0)), [lambda x: x**2], 10000, 1, 1000, 42,
     [near(0, 1.0, 0.3)]),
    ("test_mcmc.py::TestMultiChain::test_many_chains:301", lambda D: (D.normal(0.0, 1.0), D.normal(0.0, 1.0)), [lambda x: x**2], 1000, 4096, 200, 42,
     [near(0, 1.0, 0.1)]),
    ("test_mcmc.py::TestCustomDistribution::test_custom_target_distribution:351", lambda D: (D.from_pdf(bimodal_pdf, support=(-10.0, 10.0)), D.normal(0.0, 2.0)),
     [lambda x: x], 10000, 256, 1000, 42, [near(0, 0.0, 0.5)]),
    ("test_mcmc.py::TestCustomDistribution::test_beta_distribution:374", lambda D: (D.beta(2.0, 5.0), D.uniform(0.0, 1.0)), [lambda x: x], 10000, 256, 1000, 42,
     [near(0, 2.0 / 7.0, 0.1)]),
]


@pytest.mark.parametrize("ref,make,fns,n_steps,n_chains,n_burnin,seed,checks", K3, ids=[row[0] for row in K3])
def test_reference_mcmc_scenario(ref, make, fns, n_steps, n_chains, n_burnin, seed, checks):
    w = _api()
    target, proposal = make(w.Distribution)
    res = w.MonteCarloIntegrator().integrate_mcmc(fns, target, proposal, n_steps=n_steps, n_chains=n_chains, n_burnin=n_burnin, seed=seed)
    assert isinstance(res, w.IntegrationResult) and res.n_functions == len(fns)
    assert all(check(res.values) for check in checks), (ref, res.values)


def test_reference_mcmc_result_shape_burnin_seed_and_errors():
    """test_mcmc.py::TestMcmcBasic::test_mcmc_returns_integration_result:150 (n_samples = n_steps x n_chains, default seed),
    ::TestBurnIn::test_zero_burnin_allowed:172, ::test_burnin_doesnt_affect_sample_count:190, ::TestMultiChain::
    test_reproducibility_with_seed:319, ::TestErrorHandling::test_empty_function_list:399, ::test_zero_n_steps:407,
    ::test_zero_n_chains:417, ::test_negative_burnin:427, ::test_invalid_function_type:437, ::TestConvenienceFunction::
    test_convenience_function_basic:450."""
    w = _api()
    mc, D = w.MonteCarloIntegrator(), w.Distribution
    target, proposal = D.normal(0.0, 1.0), D.normal(0.0, 1.0)
    res = mc.integrate_mcmc([lambda x: x], target, proposal, n_steps=1000, n_chains=64, n_burnin=100)
    assert isinstance(res, w.IntegrationResult) and res.n_samples == 1000 * 64
    zero = mc.integrate_mcmc([lambda x: x], target, proposal, n_steps=1000, n_chains=64, n_burnin=0, seed=42)
    burned = mc.integrate_mcmc([lambda x: x], target, proposal, n_steps=1000, n_chains=64, n_burnin=1000, seed=42)
    assert zero.n_functions == 1 and zero.n_samples == burned.n_samples
    one = mc.integrate_mcmc([lambda x: x**2], target, proposal, n_steps=1000, n_chains=64, n_burnin=100, seed=12345)
    two = mc.integrate_mcmc([lambda x: x**2], target, proposal, n_steps=1000, n_chains=64, n_burnin=100, seed=12345)
    np.testing.assert_array_almost_equal(one.values, two.values)
    with pytest.raises(ValueError, match="At least one function"):
        mc.integrate_mcmc([], target, proposal)
    with pytest.raises(ValueError, match="n_steps must be positive"):
        mc.integrate_mcmc([lambda x: x], target, proposal, n_steps=0, n_chains=64, n_burnin=100)
    with pytest.raises(ValueError, match="n_chains must be positive"):
        mc.integrate_mcmc([lambda x: x], target, proposal, n_steps=100, n_chains=0, n_burnin=100)
    with pytest.raises(ValueError, match="n_burnin must be non-negative"):
        mc.integrate_mcmc([lambda x: x], target, proposal, n_steps=100, n_chains=64, n_burnin=-1)
    with pytest.raises(TypeError):
        mc.integrate_mcmc([123], target, proposal)
    conv = w.integrate_mcmc([lambda x: x**2], target, proposal, n_steps=5000, n_chains=256, n_burnin=500)
    assert isinstance(conv, w.IntegrationResult) and abs(conv.values[0] - 1.0) < 0.15
