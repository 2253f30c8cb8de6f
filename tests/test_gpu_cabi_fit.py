"""include/mcx.h: mcx_module_desc_fit / mcx_module_desc_fit_host / mcx_module_build_fitted -- libmcx's own performance planning
of a call: which field of mcx_module_desc to set for which tables. The Python host layer calls it for its own plans
(wgpu_montecarlo/api.py, _core.py), a C binding calls it directly (tests/cabi_client.c). Here: the plan made from resident
tables and the plan made without a device (from mcx_table_analyse alone: MonteCarloIntegrator.planner()) are the same desc, byte
for byte, on the BASELINE workloads -- the GPU-less build step compiles exactly the modules the GPU call loads --, and the
branches those workloads do not take (no cell form, too large for LDS, user tables, random-walk proposals)."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))


def _fields(d):
    return {name: getattr(d, name) for name, _ in d._fields_}


@pytest.mark.parametrize("name", ["c2", "c3", "c4", "c5"])
def test_the_plan_without_a_device_is_the_plan_on_the_gpu(integrator, name):
    import baseline_configs as bc
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator
    from wgpu_montecarlo import runtime as rt

    wl = bc.get(name, Distribution)
    plan = wl.prepare(integrator)._plan
    host = wl.prepare(MonteCarloIntegrator.planner())._plan
    got = plan.desc
    assert _fields(host.desc) == _fields(got) and host.module.key == rt.module_key(plan.module.user_src, got)
    tb = plan.tables
    pair = (tb.get("target_logpdf"), tb.get("proposal_logpdf")) if plan.kind == "mcmc" else (tb.get("target_pdf"), tb.get("proposal_pdf"))
    again = rt.ModuleDesc.from_buffer_copy(bytes(got))                         # fitting a fitted desc changes nothing
    rt.module_desc_fit(again, tb.get("cdf"), pair[0], pair[1], plan.p1, plan.p2)
    assert _fields(again) == _fields(got)
    if name == "c3":
        assert got.cell_tables and got.cell_noclamp and got.block == 512 and got.tables_lds
    if name == "c4":
        assert got.cell_tables and got.cell_addr16
    if name == "c5":
        assert got.cdf_direct and got.tables_lds and not got.cell_tables


def test_the_other_branches(integrator):
    from wgpu_montecarlo import Distribution
    from wgpu_montecarlo import runtime as rt

    eng = integrator._engine
    xs = np.linspace(-4, 4, 400)
    grid = eng.cached_table(rt.TABLE_PDF, xs.astype(np.float32), np.exp(-0.5 * xs * xs).astype(np.float32))
    ragged_x = np.sort(np.random.default_rng(1).uniform(-4, 4, 300)).astype(np.float32)
    ragged = eng.cached_table(rt.TABLE_PDF, ragged_x, np.exp(-0.5 * ragged_x * ragged_x).astype(np.float32))
    base = lambda **kw: rt.make_desc(rt.KIND_INTEGRATE, 2, rt.DIST_NORMAL, weight=True, p_table=True, **kw)
    d = base()
    pad = rt.module_desc_fit(d, None, grid, None, 0.0, 1.0)
    assert d.cell_tables and d.cell_noclamp and pad > 0 and d.unit_params and d.block == 512
    d = base()
    assert rt.module_desc_fit(d, None, ragged, None, 0.5, 1.0) == 0 and not d.cell_tables and not d.cell_noclamp and not d.unit_params
    d = base(precise_sampler=True)
    rt.module_desc_fit(d, None, grid, None, 0.0, 1.0)
    assert not d.cell_tables                                                    # the literal search + blend stays
    d = rt.make_desc(rt.KIND_INTEGRATE, 2, rt.DIST_NORMAL, user_tables=1)
    assert rt.module_desc_fit(d, None, grid, None, 0.0, 1.0) == 0 and d.cell_tables and not d.cell_noclamp      # lookups at any argument: the clamp stays
    d = rt.make_desc(rt.KIND_MCMC, 2, rt.DIST_NORMAL, walk=rt.WALK_RANDOM_SYMMETRIC)
    logs = eng.cached_table(rt.TABLE_LOGPDF, xs.astype(np.float32), (-0.5 * xs * xs).astype(np.float32))
    assert rt.module_desc_fit(d, None, logs, None, 0.0, 0.5) == 0 and d.cell_tables and not d.cell_noclamp and d.cell_addr16
    big_x = np.linspace(-8, 8, 30000)
    big = eng.cached_table(rt.TABLE_PDF, big_x.astype(np.float32), np.exp(-0.5 * big_x * big_x).astype(np.float32))
    d = base()
    rt.module_desc_fit(d, None, big, None, 0.0, 1.0)
    assert not d.tables_lds and not d.cell_noclamp                               # 240 KB of cells: read from HBM / L2
    # the fitted module runs, and gives what the Python API gives for the same call
    target = Distribution.from_pdf_table(xs, np.exp(-0.5 * xs * xs) / np.sqrt(2 * np.pi))
    fns = [lambda x: x, lambda x: x * x]
    want = integrator.integrate_importance_sampling(fns, target, Distribution.normal(0.5, 1.5), n_samples=1_000_000, seed=5)
    assert abs(want.values[1] - 1.0) < 0.02


def test_host_and_device_fits_agree_on_random_calls(integrator):
    """mcx_module_desc_fit (resident tables) and mcx_module_desc_fit_host (mcx_table_analyse facts + keys) are two doors to one
    planner: on 200 random calls -- table sizes from 64 to 40 000 points, strict grids and ragged ones, every sampler with random
    parameters, custom CDF samplers with and without the bucket-direct form, integrate and MCMC modules, random-walk proposals,
    user tables, precise samplers, K from 1 to 32 -- they fill the same desc and report the same pad bytes."""
    from wgpu_montecarlo import runtime as rt

    eng = integrator._engine
    rng = np.random.default_rng(31)

    def table(kind, n, ragged):
        xs = np.sort(rng.uniform(-5, 5, n)) if ragged else np.linspace(-rng.uniform(2, 8), rng.uniform(2, 8), n)
        xs = xs.astype(np.float32)
        vals = np.exp(-0.5 * xs.astype(np.float64) ** 2)
        if kind == rt.TABLE_LOGPDF:
            vals = np.log(np.maximum(vals, 1e-30))
        return (eng.cached_table(kind, xs, vals.astype(np.float32)), rt.HostTable(kind, xs, vals.astype(np.float32)))

    def cdf_table(n):
        xs = np.linspace(0.0, 1.0, n).astype(np.float32)
        dens = xs.astype(np.float64) * (1 - xs.astype(np.float64)) ** 4 + 1e-3
        c = np.concatenate([[0.0], np.cumsum((dens[1:] + dens[:-1]) / 2)])
        c = (c / c[-1]).astype(np.float32)
        return (eng.cached_table(rt.TABLE_CDF, c, xs), rt.HostTable(rt.TABLE_CDF, c, xs))

    checked = with_pads = 0
    for _ in range(200):
        mcmc = rng.random() < 0.4
        kind = rt.KIND_MCMC if mcmc else rt.KIND_INTEGRATE
        tkind = rt.TABLE_LOGPDF if mcmc else rt.TABLE_PDF
        dist = int(rng.choice([rt.DIST_UNIFORM, rt.DIST_NORMAL, rt.DIST_EXPONENTIAL, rt.DIST_CUSTOM]))
        p1, p2 = {rt.DIST_UNIFORM: (-float(rng.uniform(0, 4)), float(rng.uniform(0.5, 4))), rt.DIST_NORMAL: (float(rng.choice([0.0, 0.5])), float(rng.choice([1.0, 0.7]))),
                  rt.DIST_EXPONENTIAL: (float(rng.choice([1.0, 2.5])), 0.0), rt.DIST_CUSTOM: (0.0, 0.0)}[dist]
        cdf = cdf_table(int(rng.choice([300, 2048, 5000]))) if dist == rt.DIST_CUSTOM else (None, None)
        sizes = [int(rng.choice([64, 512, 2048, 9000, 40000])) for _ in range(2)]
        t0 = table(tkind, sizes[0], rng.random() < 0.25) if rng.random() < 0.85 else (None, None)
        t1 = table(tkind, sizes[1], rng.random() < 0.25) if rng.random() < 0.4 else (None, None)
        kw = dict(precise_sampler=bool(rng.random() < 0.15), second_moments=bool(rng.random() < 0.2 and not mcmc))
        if mcmc:
            kw["walk"] = int(rng.choice([rt.WALK_INDEPENDENT, rt.WALK_INDEPENDENT, rt.WALK_RANDOM_SYMMETRIC]))
            kw["block"] = int(rng.choice([0, 0, 256]))
        elif rng.random() < 0.3 and t0[0] is not None:
            kw["user_tables"] = 1 | (2 if t1[0] is not None else 0)
        else:
            kw.update(weight=t0[0] is not None or t1[0] is not None, p_table=t0[0] is not None, q_table=t1[0] is not None)
        k = int(rng.choice([1, 2, 4, 16, 32]))
        if kw.get("second_moments") and k > 16:
            k = 16
        a = rt.make_desc(kind, k, dist, **kw)
        b = rt.ModuleDesc.from_buffer_copy(bytes(a))
        pad_a = rt.module_desc_fit(a, cdf[0], t0[0], t1[0], p1, p2)
        pad_b = rt.module_desc_fit(b, cdf[1], t0[1], t1[1], p1, p2)
        assert _fields(a) == _fields(b) and pad_a == pad_b, (_fields(a), _fields(b), pad_a, pad_b)
        checked += 1
        with_pads += pad_a > 0
    assert checked == 200 and with_pads >= 10
