"""Pins for the CPU oracle (oracle/mcx_oracle.c).

The reference holds NO numeric golden vectors for this path: its GPU tests assert
statistical tolerances against analytic truth. The oracle is therefore pinned by
  (1) integer known-answer vectors derived from the reference source (pcg_hash constants,
      distribution.rs:62-73; dispatch arithmetic, engine.rs:157-181, 821-832) -- the values below were
      produced independently by the survey's numpy restatement and by this C one;
  (2) every accuracy expectation of the reference's own tests, evaluated on the oracle's output
      (file:line cited per test).
Float parity with a real WGSL backend beyond those tolerances is UNPINNED (implementation-defined
log/sin/cos/pow precision in naga -> Metal/Vulkan/DX12).
"""
import math

import numpy as np
import pytest

import oracle

MOM = [(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2), (oracle.FN_POW, 3), (oracle.FN_POW, 4)]


def test_pcg_hash_known_answers():
    kat = {0: 129708002, 1: 2831084092, 2: 2055130248, 3: 2131687100, 42: 1223963391,
           4294967295: 3861530882, 449710063: 0}
    for v, h in kat.items():
        assert oracle.pcg_hash(v) == h


def test_pcg_hash_is_wrapping_u32_arithmetic():
    def py(v):
        s = (v * 747796405 + 2891336453) & 0xFFFFFFFF
        w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
        return ((w >> 22) ^ w) & 0xFFFFFFFF
    rng = np.random.default_rng(0)
    for v in rng.integers(0, 2**32, size=2000):
        assert oracle.pcg_hash(int(v)) == py(int(v))


def test_counter_known_answers():
    kat = [((42, 0, 0), 42, 1223963391), ((42, 0, 1), 15485905, 1068524562), ((42, 1, 0), 7199411, 2858281791),
           ((42, 65535, 30517), 3794890804, 2007115843), ((12345, 255, 1000002), 77753790, 533615206),
           ((1000041, 7, 11000), 2892164080, 130712447), ((4294967295, 1, 1), 22685231, 1222289643)]
    for (seed, idx, it), comb, h in kat:
        assert oracle.combined(seed, idx, it) == comb
        assert oracle.pcg_hash(comb) == h


def test_uniform_known_answers_and_endpoints():
    assert oracle.random_uniform(42, 0, 0) == pytest.approx(0.28497618, abs=1e-8)
    assert oracle.random_uniform(42, 1, 0) == pytest.approx(0.6654956, abs=1e-7)
    assert oracle.random_uniform(42, 2, 0) == pytest.approx(0.714696, abs=1e-6)
    # closed interval: u = float(h) * 2^-32, so 0 iff h == 0 and 1.0 iff h >= 0xFFFFFF80
    assert np.float32(0xFFFFFF80) / np.float32(4294967296.0) == np.float32(1.0)
    assert np.float32(0xFFFFFF7F) / np.float32(4294967296.0) < np.float32(1.0)


def test_dispatch_known_answers():
    for n, L in [(10**6, 16), (10**7, 153), (10**8, 1526), (10**9, 15259), (10**10, 152588)]:
        cfg = oracle.dispatch_config(n)
        assert (cfg["total_threads"], cfg["loops_per_thread"], cfg["workgroup_count"]) == (65536, L, 256)
    assert oracle.dispatch_config(10**6, 32768)["total_threads"] == 32768
    assert oracle.dispatch_config(10**6, 1000)["total_threads"] == 1024
    for chains, padded in [(1, 256), (64, 256), (4096, 4096), (1048576, 1048576), (257, 512)]:
        assert oracle.mcmc_dispatch_config(chains)["total_threads"] == padded
    # target_threads overrides n_chains (engine.rs:860)
    assert oracle.mcmc_dispatch_config(1024, 100)["total_threads"] == 256


def test_box_muller_pairs_and_cache():
    """even i draws counters (2i, 2i+1) and caches z1; odd i returns the cache (distribution.rs:90-114)."""
    xs = oracle.samples(oracle.NORMAL, 0.0, 1.0, n_samples=65536 * 5, seed=42, nidx=3)
    assert xs.shape == (3, 5)
    for idx in range(3):
        for j in (0, 1, 2):
            u1 = np.float32(oracle.random_uniform(42, idx, 4 * j))
            u2 = np.float32(oracle.random_uniform(42, idx, 4 * j + 1))
            r = np.sqrt(np.float32(-2.0) * np.log(u1, dtype=np.float32), dtype=np.float32)
            th = np.float32(6.283185307179586) * u2
            assert xs[idx, 2 * j] == pytest.approx(float(r * np.cos(th, dtype=np.float32)), abs=2e-6)
            if 2 * j + 1 < 5:
                assert xs[idx, 2 * j + 1] == pytest.approx(float(r * np.sin(th, dtype=np.float32)), abs=2e-6)


def test_reference_accuracy_expectations_normal():
    """tests/test_integrator.py:181-194, 230-246 (|E - truth| < 0.01 at 1e7); SURVEY App. A.4 values."""
    r = oracle.integrate(MOM, oracle.NORMAL, 0.0, 1.0, n_samples=10**7, seed=42)
    assert np.all(np.abs(r["ref"] - [0, 1, 0, 3]) < [0.01, 0.01, 0.01, 0.03])
    assert r["ref"] == pytest.approx([-3.59e-4, 0.99975, -1.01e-3, 2.99904], abs=2e-5)
    r6 = oracle.integrate(MOM[:2], oracle.NORMAL, 0.0, 1.0, n_samples=10**6, seed=42)      # BASELINE C1
    assert r6["n_eff"] == 1_048_576
    assert r6["ref"] == pytest.approx([-4.11e-4, 1.00185], abs=1e-5)
    r5 = oracle.integrate([(oracle.FN_IDENTITY, 0), (oracle.FN_SQ, 0)], oracle.NORMAL, 5.0, 2.0, n_samples=10**7, seed=42)
    assert abs(r5["ref"][0] - 5.0) < 0.01 and abs(r5["ref"][1] - 29.0) < 0.1   # tests/test_distributions.py:213-226


def test_reference_accuracy_expectations_uniform_exponential():
    """tests/test_integrator.py:196-228, 248-257."""
    r = oracle.integrate(MOM[:2], oracle.UNIFORM, 0.0, 1.0, n_samples=10**7, seed=42)
    assert abs(r["ref"][0] - 0.5) < 0.01 and abs(r["ref"][1] - 1 / 3) < 0.01
    r = oracle.integrate(MOM[:2], oracle.EXPONENTIAL, 2.0, 0.0, n_samples=10**7, seed=42)
    assert abs(r["ref"][0] - 0.5) < 0.01 and abs(r["ref"][1] - 0.5) < 0.01
    r = oracle.integrate([(oracle.FN_SIN, 0), (oracle.FN_COS, 0)], oracle.UNIFORM, 0.0, 2 * math.pi, n_samples=10**7, seed=42)
    assert abs(r["ref"][0]) < 0.01 and abs(r["ref"][1]) < 0.01


def test_table_lookup_semantics():
    x = np.linspace(0.0, 1.0, 11, dtype=np.float32)
    v = (x * 10).astype(np.float32)
    assert oracle.table_lookup(x, v, -0.01, -100.0) == -100.0          # outside: log-pdf tables
    assert oracle.table_lookup(x, v, 1.01, 0.0) == 0.0                  # outside: pdf tables
    assert oracle.table_lookup(x, v, 0.0, 0.0) == 0.0
    assert oracle.table_lookup(x, v, 1.0, 0.0) == pytest.approx(10.0)
    assert oracle.table_lookup(x, v, 0.55, 0.0) == pytest.approx(5.5, abs=1e-5)
    cdf = np.array([0.0, 0.25, 0.25, 1.0], dtype=np.float32)             # flat segment: the 1e-10 guard
    xt = np.array([0.0, 1.0, 2.0, 3.0], dtype=np.float32)
    assert oracle.sample_cdf(0.0, cdf, xt) == 0.0
    assert oracle.sample_cdf(0.125, cdf, xt) == pytest.approx(0.5)
    assert oracle.sample_cdf(0.25, cdf, xt) == pytest.approx(1.0)
    assert oracle.sample_cdf(0.625, cdf, xt) == pytest.approx(2.5)
    assert oracle.sample_cdf(1.0, cdf, xt) == pytest.approx(3.0)


def test_reference_accuracy_expectations_beta_and_mcmc():
    """tests/test_distributions.py:78-110 (Beta(2,5) moments within 0.01 at 1e7);
    tests/test_mcmc.py:91-148, 351-372 (MH accuracy); SURVEY App. A.4."""
    from wgpu_montecarlo import Distribution

    d = Distribution.beta(2.0, 5.0)
    r = oracle.integrate(MOM[:3], oracle.CUSTOM, n_samples=10**7, seed=42, cdf_table=d._cdf_table, x_table=d._x_table)
    assert r["ref"] == pytest.approx([0.285662, 0.107111, 0.047600], abs=2e-5)
    assert np.all(np.abs(r["ref"] - [2 / 7, 3 / 28, 1 / 21]) < 0.01)

    def run(target, proposal, code, p1, p2, **kw):
        tx, tl = target.get_log_pdf_table()
        px, pl = proposal.get_log_pdf_table()
        return oracle.mcmc([(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2)], code, p1, p2, tx, tl, px, pl, **kw)

    r = run(Distribution.normal(0, 1), Distribution.normal(0, 1.5), oracle.NORMAL, 0.0, 1.5,
            n_steps=10_000, n_chains=512, n_burnin=1000)
    assert abs(r["ref"][1] - 1.0) < 0.01 and abs(r["ref"][0]) < 0.01
    assert r["sums"][2] / (512 * 11_000) == pytest.approx(0.749, abs=0.01)
    same = run(Distribution.normal(0, 1), Distribution.normal(0, 1), oracle.NORMAL, 0.0, 1.0,
               n_steps=500, n_chains=256, n_burnin=0)
    assert same["sums"][2] == 256 * 500           # p == q: every step accepted (tests/test_mcmc.py:94)
    bim = Distribution.from_pdf(lambda x: 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2)),
                                support=(-10, 10))
    r = run(bim, Distribution.normal(0, 2), oracle.NORMAL, 0.0, 2.0, n_steps=10_000, n_chains=256, n_burnin=1000)
    assert abs(r["ref"][0]) < 0.3 and abs(r["ref"][1] - 5.0) < 0.3
    assert r["n_eff"] == 256 * 10_000


def test_importance_sampling_algebra():
    """tests/test_importance_sampling.py:34-62: E_p[x^2] for p = N(0,1) sampled from q = N(0.5, 1.5)."""
    s2pi = float(np.float32(np.sqrt(2 * np.pi)))
    r = oracle.integrate(MOM[:2], oracle.NORMAL, 0.5, 1.5, n_samples=5_000_000, seed=42,
                         p=(oracle.PDF_NORMAL, 0.0, 1.0, s2pi), q=(oracle.PDF_NORMAL, 0.5, 1.5, s2pi))
    assert abs(r["ref"][0]) < 0.05 and abs(r["ref"][1] - 1.0) < 0.05


def test_philox4x32_10_random123_known_answers():
    """libmcx's opt-in stream is not from the reference; it is pinned by the published Random123 vectors
    (kat_vectors: philox4x32 10)."""
    assert oracle.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert oracle.philox4x32_10((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert oracle.philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == (
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)


def test_philox_stream_statistics():
    r = oracle.integrate(MOM, oracle.NORMAL, 0.0, 1.0, n_samples=10**7, seed=42, rng=1, guard=1)
    sigma = np.sqrt(np.array([1, 2, 15, 96]) / r["n_eff"])
    assert np.all(np.abs(r["sums"] / r["n_eff"] - [0, 1, 0, 3]) < 4 * sigma)
    r = oracle.integrate(MOM[:2], oracle.UNIFORM, 0.0, 1.0, n_samples=10**7, seed=1, rng=1, guard=1)
    assert abs(r["sums"][0] / r["n_eff"] - 0.5) < 4e-4 and abs(r["sums"][1] / r["n_eff"] - 1 / 3) < 4e-4


def test_numpy_restatement_agrees_with_the_c_oracle_on_the_same_stream():
    """oracle/numpy_port.py was written independently of mcx_oracle.c from the same reference lines
    (src/distribution.rs:62-114, src/shader_gen.rs:93-117): same counters, same f32 terms -> the f64 sums agree to the
    rounding of libm vs numpy transcendentals. Even L, odd L (the discarded second half), a general N(mean, std)."""
    from oracle import numpy_port as npp

    pows = lambda k: [(oracle.FN_IDENTITY, 0)] + [(oracle.FN_POW, j) for j in range(2, k + 1)]
    for n, seed, mean, std, k in [(1_000_000, 42, 0.0, 1.0, 4), (65536 * 5, 7, 0.0, 1.0, 2), (300_000, 99, 2.0, 3.0, 3)]:
        sums, n_eff = npp.normal_moments(k, n, seed=seed, mean=mean, std=std)
        ref = oracle.integrate(pows(k), oracle.NORMAL, mean, std, n_samples=n, seed=seed, guard=1)
        assert n_eff == ref["n_eff"]
        scale = np.array([abs(mean) + std] * k) ** np.arange(1, k + 1) * n_eff
        assert np.all(np.abs(sums - ref["sums"]) <= 2e-6 * scale), (sums, ref["sums"])
    # integer side: App. A.2 hash vectors through the numpy hash
    for v in (0, 1, 42, 0xFFFFFFFF, 123456789):
        assert int(npp.pcg_hash(np.uint32(v))[0]) == oracle.pcg_hash(v)
