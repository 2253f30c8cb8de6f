"""The plan cache of the public calls (api._cached_plan) and the GPU-less planner (MonteCarloIntegrator.planner): a
repeat call is one dict lookup -- the reference re-transpiles and re-compiles on every call (src/engine.rs:325-331,
python/wgpu_montecarlo/__init__.py:740-746) -- but only while everything the plan depends on is unchanged: the functions'
code AND captured values, the distributions, the integrator's mode, the tuning knobs. Runs without a GPU: the planner
compiles (hiprtc needs none) and cannot launch."""
import numpy as np
import pytest

from wgpu_montecarlo import Distribution, MonteCarloIntegrator, TranspilerError
from wgpu_montecarlo import runtime as rt

SCALE = 2.0


def _plan(mc, fns, dist):
    return mc._cached_plan("integrate", fns, (dist,), None, lambda: mc._plan_integrate(fns, dist))


def test_repeat_calls_hit_and_changed_captures_miss(monkeypatch):
    mc = MonteCarloIntegrator.planner()
    dist = Distribution.normal(0.0, 1.0)

    def family(k):
        return [lambda x: SCALE * x, lambda x: x**k]

    a = _plan(mc, family(3), dist)
    assert _plan(mc, family(3), dist) is a                       # new lambda objects, same code + captured values
    assert _plan(mc, family(4), dist) is not a                   # closure value changed
    monkeypatch.setitem(globals(), "SCALE", 5.0)
    b = _plan(mc, family(3), dist)
    assert b is not a and "5.0" in b.module.user_src             # a global the function reads changed
    monkeypatch.setitem(globals(), "SCALE", 2.0)
    assert _plan(mc, family(3), dist) is a
    # another distribution object with the same parameters is the same key (by value: the reference's examples write
    # `Distribution.normal(0, 1)` inline in every call); other parameters are another plan
    assert _plan(mc, family(3), Distribution.normal(0.0, 1.0)) is a
    assert _plan(mc, family(3), Distribution.uniform(0.0, 1.0)) is not a
    assert _plan(mc, family(3), Distribution.normal(0.5, 1.0)).desc.unit_params == 0
    n_plans = len(mc._engine._plans)
    for _ in range(5):
        _plan(mc, family(3), Distribution.normal(0.0, 1.0))
        _plan(mc, family(3), Distribution.exponential(2.0))
    assert len(mc._engine._plans) == n_plans + 1                  # one new entry (exponential(2)), not ten
    # table-backed distributions are keyed by their density and their arrays; the tables of a repeated inline density are built
    # once (distributions._cached_cdf_table), so a second Distribution.beta(2, 5) is the same plan too
    beta = Distribution.beta(2.0, 5.0)
    assert _plan(mc, family(3), beta) is _plan(mc, family(3), beta)
    assert _plan(mc, family(3), Distribution.beta(2.0, 5.0)) is _plan(mc, family(3), beta)
    assert _plan(mc, family(3), Distribution.beta(2.0, 5.5)) is not _plan(mc, family(3), beta)
    # a density that captures arrays has no value: keyed by the object
    xs = np.linspace(-3, 3, 200)
    t1, t2 = Distribution.from_pdf_table(xs, np.exp(-xs * xs)), Distribution.from_pdf_table(xs, np.exp(-xs * xs))
    assert _plan(mc, family(3), t1) is _plan(mc, family(3), t1) and _plan(mc, family(3), t2) is not _plan(mc, family(3), t1)
    # the integrator's mode is part of the key; plans are shared per engine, not per integrator
    other = MonteCarloIntegrator.planner(rng="philox")
    other._engine = mc._engine
    assert _plan(other, family(3), dist) is not a
    same_mode = MonteCarloIntegrator.planner()
    same_mode._engine = mc._engine
    assert _plan(same_mode, family(3), dist) is a
    # a tuning knob flipped at run time is not answered from the cache
    beta = Distribution.beta(2.0, 5.0)
    d1 = _plan(mc, family(3), beta)
    monkeypatch.setenv("MCX_NO_DIRECT", "1")
    d2 = _plan(mc, family(3), beta)
    assert d1.desc.cdf_direct == 1 and d2.desc.cdf_direct == 0 and d2 is not d1
    monkeypatch.delenv("MCX_NO_DIRECT")
    assert _plan(mc, family(3), beta) is d1


def test_what_cannot_be_keyed_takes_the_uncached_path_and_raises_what_it_always_raised():
    mc = MonteCarloIntegrator.planner()
    dist = Distribution.uniform(0.0, 1.0)
    table = np.arange(3.0)
    with pytest.raises(TranspilerError, match="Unsupported external variable type"):
        _plan(mc, [lambda x: x + table], dist)                   # an ndarray capture: unhashable AND outside the subset
    with pytest.raises(TypeError):
        _plan(mc, [3.0], dist)
    wgsl = "fn f(x: f32) -> f32 { return x * x; }"
    assert _plan(mc, [wgsl], dist) is _plan(mc, [wgsl], dist)    # raw WGSL strings key by their text
    assert len(mc._engine._plans) == 1


def test_the_cache_is_bounded():
    from wgpu_montecarlo import api

    mc = MonteCarloIntegrator.planner()
    dist = Distribution.exponential(1.0)
    old = api._PLAN_CACHE_ENTRIES
    api._PLAN_CACHE_ENTRIES = 3
    try:
        shifted = lambda c: [lambda x, c=float(c): x + c]                     # one code object, the constant as a default
        plans = [_plan(mc, shifted(c), dist) for c in range(5)]
        assert len(mc._engine._plans) == 3
        assert _plan(mc, shifted(4), dist) is plans[4]                        # the newest survived
        assert _plan(mc, shifted(0), dist) is not plans[0]                    # the oldest was dropped and is rebuilt
    finally:
        api._PLAN_CACHE_ENTRIES = old


def test_planner_builds_the_baseline_modules_without_a_gpu_and_cannot_launch():
    """What __graft_entry__.build() relies on: the code objects bench.py launches exist in the cache before the GPU box
    ever sees them, built from the same descs a GPU call derives (C3: 512 threads, padded unclamped cells; C4: 16-bit
    cell addresses; C5: bucket-direct records + moment family)."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import baseline_configs as bc

    mc = MonteCarloIntegrator.planner()
    want = {"c2": dict(block=0, unit_params=1), "c3": dict(block=512, cell_tables=1, cell_noclamp=1, q_sampler=1, cell_addr16=0),
            "c4": dict(block=512, cell_tables=1, cell_noclamp=1, q_sampler=1, cell_addr16=1),
            "c5": dict(block=0, cdf_direct=1, moment_family=1, tables_lds=1)}
    for name, fields in want.items():
        prepared = bc.get(name, Distribution).prepare(mc)
        desc, mod = prepared._plan.desc, prepared._plan.module
        assert {f: getattr(desc, f) for f in fields} == fields, name
        assert mod.code_object.exists() and mod.key == rt.module_key(mod.user_src, desc)
    with pytest.raises(RuntimeError, match="cannot launch"):
        mc.integrate([lambda x: x], Distribution.normal(0.0, 1.0), n_samples=1000)


def test_lds_decisions_are_rechecked_against_the_code_objects_real_static_lds(monkeypatch):
    """mcx_module_desc_fit decides cell_addr16 / cell_noclamp / tables_lds with an upper bound of the kernel's static LDS;
    the launch checks the real figure and refuses. If the real figure were ever larger, the plan must fall back to the
    slower form instead of turning a valid call into MCX_E_INVALID: simulated here with modules that report more."""
    xs = np.linspace(0, 10, 512)
    target, proposal = Distribution.from_pdf_table(xs, np.exp(-xs)), Distribution.normal(2.0, 3.0)
    fns = [lambda x: x, lambda x: x * x]
    mc = MonteCarloIntegrator.planner()
    plan = mc._plan_mcmc(fns, target, proposal, block=0)
    assert plan.desc.cell_addr16 == 1 and plan.desc.tables_lds == 1
    real = rt.HostModule.__init__

    def inflated(extra):
        def init(self, user_src, desc):
            real(self, user_src, desc)
            self.static_lds += extra
        return init

    monkeypatch.setattr(rt.HostModule, "__init__", inflated(60 * 1024))       # beyond 64 KiB in total: no 16-bit addresses
    plan = mc._plan_mcmc(fns, target, proposal, block=0)
    assert plan.desc.cell_addr16 == 0 and plan.desc.tables_lds == 1 and plan.desc.cell_noclamp == 1
    monkeypatch.setattr(rt.HostModule, "__init__", inflated(158 * 1024))      # beyond the CU: tables from HBM / L2
    plan = mc._plan_mcmc(fns, target, proposal, block=0)
    assert (plan.desc.tables_lds, plan.desc.cell_noclamp, plan.desc.cell_addr16) == (0, 0, 0)


def test_auto_stream_policy_picks_philox_only_beyond_the_reference_streams_counter_space():
    """rng="auto": the reference's hash for every call that stays within its 2^32 inputs (parity stream), Philox for the
    calls that draw more. The choice is per call, and the two streams' plans are separate cache entries."""
    from wgpu_montecarlo import runtime as rt

    auto = MonteCarloIntegrator.planner(rng="auto")
    assert auto._pick_rng(10**9) == rt.RNG_PCG_REF and auto._pick_rng(2**32) == rt.RNG_PCG_REF
    assert auto._pick_rng(2**32 + 1) == rt.RNG_PHILOX and auto._pick_rng(10**10) == rt.RNG_PHILOX
    assert MonteCarloIntegrator.planner(rng="pcg_ref")._pick_rng(10**12) == rt.RNG_PCG_REF
    assert MonteCarloIntegrator.planner(rng="philox")._pick_rng(10) == rt.RNG_PHILOX
    dist = Distribution.normal(0.0, 1.0)
    fns = [lambda x: x * x]
    small = auto._cached_plan("integrate", fns, (dist,), None, lambda: auto._plan_integrate(fns, dist, rt.RNG_PCG_REF), rt.RNG_PCG_REF)
    large = auto._cached_plan("integrate", fns, (dist,), None, lambda: auto._plan_integrate(fns, dist, rt.RNG_PHILOX), rt.RNG_PHILOX)
    assert small.desc.rng == rt.RNG_PCG_REF and large.desc.rng == rt.RNG_PHILOX and small is not large
    with pytest.raises(ValueError, match="'pcg_ref'.*'philox' or 'auto'"):
        MonteCarloIntegrator.planner(rng="xoshiro")


def test_the_committed_issue_model_describes_the_code_objects_this_tree_builds():
    """profiles/rNN_issue_model.json names modules by the cache key of their code object; bench.py quotes `ops_per_unit` and
    the modelled cycles only for a module it finds there. Any edit of the device sources changes the keys: this test fails
    until `python tools/issue_model.py` has been re-run, so the bench line never falls back to stale constants silently."""
    import json
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "tools"))
    import baseline_configs as bc

    files = sorted((root / "profiles").glob("r[0-9][0-9]_issue_model.json"))
    assert files, "no profiles/rNN_issue_model.json"
    modules = json.loads(files[-1].read_text())["modules"]
    for rng in ("pcg_ref", "philox"):
        mc = MonteCarloIntegrator.planner(rng=rng)
        for name in ("c2", "c3", "c4", "c5"):
            key = bc.get(name, Distribution).prepare(mc)._plan.module.key
            assert key in modules, f"{files[-1].name} has no entry for {name} / {rng} ({key}): re-run tools/issue_model.py"
            entry = modules[key]
            assert entry["config"] == name and entry["rng"] == rng
            ops, source = bc.ops_per_unit(name, key)
            assert ops == entry["survey_weighted_ops_per_unit"] and "this code object" in source
    assert bc.ops_per_unit("c2")[0] == pytest.approx(22.0) and bc.ops_per_unit("c5", "no-such-key")[0] == bc.OPS_PER_UNIT["c5"]
