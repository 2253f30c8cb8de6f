"""desc.cell_noclamp on the host (no GPU): the sentinel cells the launch stages either side of a cell table must cover
every index the device's unclamped lookup can form -- floor(fma(x, scale, c0 + pad_l)) in f32 -- for every x the sampler
can produce (include/mcx.h: mcx_cell_pads_host; csrc/mcx_runtime.cpp: sampler_reach, cell_pads)."""
import math

import numpy as np
import pytest

from wgpu_montecarlo import runtime as rt

F = np.float32


def _extreme_draws(dist, p1, p2, x_range):
    """The most extreme x each sampler can return, evaluated in f32 the way device/mcx_device.hpp does."""
    if dist == rt.DIST_UNIFORM:
        span = F(p2) - F(p1)
        return [F(p1), F(F(1.0) * span + F(p1)), F(F(float.fromhex("0x1.fffffep-1")) * span + F(p1))]
    if dist == rt.DIST_NORMAL:
        # Box-Muller with f1 >= 0.5: r^2 = (32 - log2 f1) 2 ln 2 <= 33 * 2 ln 2; |cos|, |sin| <= 1 (+ 1 ulp of v_sin / v_cos)
        r = F(math.sqrt(33.0 * 2.0 * math.log(2.0))) * F(1.0 + 2.0**-20)
        return [F(F(p2) * z + F(p1)) for z in (r, -r)]
    if dist == rt.DIST_EXPONENTIAL:
        return [F(0.0), F(F(-math.log(1e-7)) * F(1.0 + 2.0**-20) / F(p1))]
    return [F(x_range[0]), F(x_range[1])]


@pytest.mark.parametrize("seed", range(40))
def test_pads_cover_every_index_the_unclamped_lookup_can_form(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([2, 3, 17, 300, 512, 1000, 2048, 4096]))
    x0 = float(rng.uniform(-50, 50))
    width = float(10 ** rng.uniform(-2, 2))
    keys = np.linspace(x0, x0 + width, n).astype(np.float32)
    scale, c0 = rt.table_cell_map(keys)
    dist = int(rng.integers(0, 4))
    p1, p2, x_range = 0.0, 1.0, None
    if dist == rt.DIST_UNIFORM:
        p1 = x0 + float(rng.uniform(-2, 1)) * width
        p2 = p1 + float(rng.uniform(0.1, 3)) * width
    elif dist == rt.DIST_NORMAL:
        p1, p2 = x0 + float(rng.uniform(-1, 2)) * width, float(rng.uniform(0.01, 0.4)) * width
    elif dist == rt.DIST_EXPONENTIAL:
        p1 = float(rng.uniform(0.05, 20)) / width
    else:
        lo = x0 + float(rng.uniform(-2, 0.5)) * width
        x_range = (lo, lo + float(rng.uniform(0.1, 3)) * width)
    pads = rt.cell_pads_host(keys, dist, p1, p2, x_range)
    if pads is None:                                   # too wide a range: the clamped module is built instead
        cells_per_unit = (n - 1) / width
        assert any(abs(float(x) - x0) * cells_per_unit > 3000 or abs(float(x) - x0 - width) * cells_per_unit > 3000
                   for x in _extreme_draws(dist, p1, p2, x_range)), (seed, dist, p1, p2)
        return
    pad_l, pad_r = pads
    assert 1 <= pad_l <= 4096 and 1 <= pad_r <= 4096
    staged = n + 1 + pad_l + pad_r                     # sentinels + n - 1 cells + sentinels
    c0p = F(c0 + F(pad_l))
    for x in _extreme_draws(dist, p1, p2, x_range) + [F(keys[0]), F(keys[-1])]:
        t = F(np.float64(x) * np.float64(scale) + np.float64(c0p))          # one rounding, like v_fma_f32
        idx = int(math.floor(float(t)))
        assert 0 <= idx < staged, (seed, dist, float(x), idx, staged, pads)
    # in-table x still land on their own cell: pad_l + 1 + c
    mid = F(0.5 * (float(keys[n // 2 - 1]) + float(keys[n // 2]))) if n > 2 else F(0.5 * (float(keys[0]) + float(keys[1])))
    t = F(np.float64(mid) * np.float64(scale) + np.float64(c0p))
    assert int(math.floor(float(t))) == pad_l + 1 + (n // 2 - 1 if n > 2 else 0)


def test_unbounded_or_degenerate_ranges_are_refused():
    keys = np.linspace(0.0, 1.0, 512).astype(np.float32)
    assert rt.cell_pads_host(keys, rt.DIST_NORMAL, 0.5, 0.1) is not None
    assert rt.cell_pads_host(keys, rt.DIST_NORMAL, 0.5, 0.1, guard=False) is None          # u1 = 0: infinite deviate
    assert rt.cell_pads_host(keys, rt.DIST_NORMAL, 0.5, float("inf")) is None
    assert rt.cell_pads_host(keys, rt.DIST_EXPONENTIAL, 0.0, 0.0) is None
    assert rt.cell_pads_host(keys, rt.DIST_EXPONENTIAL, -2.0, 0.0) is None
    assert rt.cell_pads_host(keys, rt.DIST_CUSTOM, 0.0, 0.0) is None                       # no CDF table range given
    assert rt.cell_pads_host(keys, rt.DIST_CUSTOM, 0.0, 0.0, x_range=(-1.0, 2.0)) == (511, 511)   # 509.x cells each side (the index map is shrunk by 2 eps), floor + 1
    assert rt.cell_pads_host(keys, rt.DIST_UNIFORM, 0.25, 0.75) == (1, 1)
    assert rt.cell_pads_host(keys, rt.DIST_NORMAL, 0.5, 10.0) is None                      # 6.8 sigma = 34 000 cells


def test_default_launch_geometry():
    """Workgroups per launch (mcx_default_launch_blocks): 4096 without staged tables; with tables one workgroup per
    6 x lds_bytes samples, clamped to [2^20 / block, 4096] (profiles/r02b_launch_geometry_vs_call_size.txt), in whole
    rounds of the workgroups the chip holds at once (round 3: 2880 workgroups = 5.6 rounds were 5 % slower than 2560 = 5)."""
    f = rt.default_launch_blocks
    assert f(10**9, 0, 256) == 4096 and f(10**3, 0, 256) == 4096            # the unit count caps it later (plan_integrate)
    assert f(10**7, 73728, 1024) == 1024 and f(3 * 10**8, 73728, 1024) == 1024
    assert 10**9 // (6 * 73728) == 2260 and f(10**9, 73728, 1024) == 2048      # 4 whole rounds of 512 resident workgroups
    assert f(1_250_000_000, 73728, 1024) == 2560                                 # C5's 8-GPU shard: 2825 -> 5 rounds
    assert f(3 * 10**9, 73728, 1024) == 4096 and f(10**10, 73728, 1024) == 4096
    assert f(10**8, 17920, 512) == 2048 and f(10**9, 17920, 512) == 4096
    assert f(10**9, 0, 64) == 16384                                          # small workgroups: never below 2^20 threads
    assert f(10**9, 4096, 64) == 16384
    assert f(10**9, 0, 0) == 0
    blocks = [f(n, 73728, 1024) for n in (10**6, 10**7, 10**8, 10**9, 10**10, 10**11)]
    assert blocks == sorted(blocks)


def test_custom_sampler_reach_is_known_only_when_draws_stay_inside_the_x_column():
    """cell_noclamp took the reach of a custom sampling distribution as [min x, max x] of its CDF table. That
    holds only if the search ends in the cell that holds u (guide / bucket-direct form: monotone, n <= 4096 -- the
    reference's 12-step capped search, src/distribution.rs:128-158, can stop short beyond that) and no u lies beyond
    the last node (cdf[n-1] >= 1), else the last cell's line is extrapolated. mcx_table_facts.reach_known says which."""
    x = np.linspace(0.0, 1.0, 2048).astype(np.float32)
    cdf = (x.astype(np.float64) ** 2).astype(np.float32)
    cdf[-1] = 1.0
    assert rt.table_facts(rt.TABLE_CDF, cdf, x).reach_known == 1
    short = (cdf * np.float32(0.98)).astype(np.float32)           # monotone, but u in (0.98, 1] extrapolates the last cell
    f = rt.table_facts(rt.TABLE_CDF, short, x)
    assert f.reach_known == 0 and f.guide_bits > 0
    # n > 4096: no guide, no bucket-direct records -> the capped search itself runs on the device
    xb = np.linspace(0.0, 1.0, 8192).astype(np.float32)
    cb = (xb.astype(np.float64) ** 2).astype(np.float32)
    cb[-1] = 1.0
    f = rt.table_facts(rt.TABLE_CDF, cb, xb)
    assert f.guide_bits == 0 and f.direct_bits == 0 and f.reach_known == 0
    # the capped search does stop short there: emulate distribution.rs:128-147 on the 8192-point table
    lo, hi, u = 0, 8191, np.float32(0.999)
    for _ in range(12):
        if lo >= hi:
            break
        mid = (lo + hi) // 2
        if cb[mid] < u:
            lo = mid + 1
        else:
            hi = mid
    true_lb = int(np.searchsorted(cb, u, side="left"))
    assert lo != true_lb                                          # not the cell that holds u
    # non-monotone cdf: no guide either
    bad = cdf.copy()
    bad[100] = bad[99] - np.float32(1e-3)
    assert rt.table_facts(rt.TABLE_CDF, bad, x).reach_known == 0


def test_table_facts_match_what_the_planner_needs():
    """mcx_table_analyse (no GPU): the figures mcx_module_desc_fit decides with -- C3's 512-point PDF table, C4's 2048-point
    log-PDF table, Beta(2,5)'s 2048-point CDF table."""
    from wgpu_montecarlo import Distribution

    xs = np.linspace(0, 10, 512)
    t = Distribution.from_pdf_table(xs, np.exp(-xs))
    f = rt.table_facts(rt.TABLE_PDF, *t.get_or_compute_pdf_table())
    assert (f.n, f.has_cells, f.direct_bits, f.lds_bytes) == (512, 1, 0, 513 * 8) and f.inv_dk > 0
    beta = Distribution.beta(2.0, 5.0)
    f = rt.table_facts(rt.TABLE_CDF, beta._cdf_table, beta._x_table)
    assert (f.n, f.has_cells, f.direct_bits, f.reach_known) == (2048, 0, 13, 1)
    assert f.lds_bytes == 8192 * 8                                 # the bucket-direct records are the larger staged form
    assert f.value_min == pytest.approx(float(beta._x_table.min())) and f.value_max == pytest.approx(float(beta._x_table.max()))
    ragged = np.array([0.0, 0.1, 0.5, 0.6, 2.0], dtype=np.float32)
    f = rt.table_facts(rt.TABLE_LOGPDF, ragged, -ragged)
    assert f.has_cells == 0 and f.inv_dk == 0.0 and f.lds_bytes == 5 * 8
    with pytest.raises(ValueError, match="at least 2 points"):
        rt.table_facts(rt.TABLE_PDF, [0.0], [1.0])
