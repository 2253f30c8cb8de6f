"""The C-ABI shared library: loads, exports every symbol include/mcx.h declares, and its pure-host
entry points (planning, source assembly, hiprtc precompile) work without a GPU."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import oracle
from wgpu_montecarlo import runtime as rt

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "mcx.h").read_text()


def declared_symbols():
    names = set(re.findall(r"\b(mcx_[a-z0-9_]+)\s*\(", HEADER))
    inline = set(re.findall(r"static inline \w+ (mcx_[a-z0-9_]+)\s*\(", HEADER))      # the mcx_*_init helpers live in the header
    assert inline == {"mcx_module_desc_init", "mcx_integrate_params_init", "mcx_mcmc_params_init", "mcx_wgsl_program_init", "mcx_core_tables_init"}
    return sorted(names - inline)


def test_library_loads_and_exports_every_declared_symbol():
    lib = rt.load()
    declared = declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"libmcx.so does not export {name}"
    assert sorted(rt.EXPORTED_SYMBOLS) == declared
    assert b"gfx950" in lib.mcx_version()


def test_header_cites_the_reference_interfaces():
    for cite in ("src/lib.rs:47-141", "src/lib.rs:158-275", "src/lib.rs:296-431", "src/engine.rs:157-181",
                 "src/engine.rs:821-832", "src/engine.rs:91-131", "src/shader_gen.rs:45"):
        assert cite in HEADER


@pytest.mark.parametrize("n,target", [(10**6, None), (10**7, None), (10**9, None), (10**10, None), (1000, None),
                                      (10**6, 32768), (10**6, 1000), (1, 1), (2**40, 65536)])
def test_dispatch_config_bit_exact_with_oracle(n, target):
    d = rt.dispatch_config(n, target)
    o = oracle.dispatch_config(n, target)
    assert (d.workgroup_size, d.workgroup_count, d.loops_per_thread, d.total_threads) == (
        o["workgroup_size"], o["workgroup_count"], o["loops_per_thread"], o["total_threads"])


def test_dispatch_known_answers():
    assert rt.dispatch_config(10**9).loops_per_thread == 15259
    assert rt.dispatch_config(10**10).loops_per_thread == 152588
    assert rt.dispatch_config(10**9).total_threads * 15259 == 1_000_013_824
    for chains, padded in [(1, 256), (64, 256), (4096, 4096), (1048576, 1048576)]:
        assert rt.mcmc_dispatch_config(chains).total_threads == padded
    assert rt.mcmc_dispatch_config(1024, 100).total_threads == 256


@pytest.mark.parametrize("dist", [rt.DIST_UNIFORM, rt.DIST_NORMAL, rt.DIST_CUSTOM])
@pytest.mark.parametrize("n,target", [(10**9, None), (3_000_001, None), (70_000, None), (500, 256), (10**6, 1000)])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shards_partition_the_logical_grid(dist, n, target, world):
    """Union of the ranks' (idx, unit) rectangles == the whole grid, no overlap."""
    d = rt.dispatch_config(n, target)
    units = (d.loops_per_thread + 1) // 2 if dist == rt.DIST_NORMAL else d.loops_per_thread
    cells = 0
    seen_units, seen_idx = [], []
    for r in range(world):
        s = rt.shard_integrate(d, dist, r, world)
        assert s.idx_count % 256 == 0 and s.idx_begin % 256 == 0
        cells += s.idx_count * (s.unit_end - s.unit_begin)
        seen_units.append((s.unit_begin, s.unit_end))
        seen_idx.append((s.idx_begin, s.idx_begin + s.idx_count))
    assert cells == d.total_threads * units
    if units >= world:
        assert seen_units[0][0] == 0 and seen_units[-1][1] == units
        assert all(seen_units[i][1] == seen_units[i + 1][0] for i in range(world - 1))
    else:
        assert seen_idx[0][0] == 0 and seen_idx[-1][1] == d.total_threads
        assert all(seen_idx[i][1] == seen_idx[i + 1][0] for i in range(world - 1))


@pytest.mark.parametrize("world", [1, 2, 3, 8, 64])
def test_chain_shards(world):
    for total in (256, 4096, 1048576):
        spans = [rt.shard_chains(total, r, world) for r in range(world)]
        assert sum(n for _, n in spans) == total
        assert all(b % 256 == 0 and n % 256 == 0 for b, n in spans)
        assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        rt.shard_chains(100, 0, 1)
    with pytest.raises(ValueError):
        rt.shard_chains(256, 2, 2)


def test_module_source_and_precompile_without_gpu():
    from wgpu_montecarlo.api import functions_to_hip

    f1 = lambda x: x
    f2 = lambda x: x**2
    src = functions_to_hip([f1, f2])
    desc = rt.make_desc(rt.KIND_INTEGRATE, 2, rt.DIST_NORMAL)
    text = rt.module_source(src, desc)
    for piece in ("#define MCX_K 2", "#define MCX_DIST 1", "mcx_integrate_kernel", "mcx_fold_kernel",
                  "user_func_1", "mcx_pcg_out", "acc[1 * S] += mcx_b2f(user_func_1(x)) * w;"):
        assert piece in text
    rt.precompile(src, desc)
    assert rt.precompile(src, desc) in (1, 2)       # second time: memory or disk cache hit
    with pytest.raises(RuntimeError, match="hiprtc"):
        rt.precompile("MCX_DEV float user_func_0(float x) { return nope(x); }", rt.make_desc(rt.KIND_INTEGRATE, 1, 1))
    with pytest.raises(ValueError, match="At least one function is required"):
        rt.precompile("", rt.make_desc(rt.KIND_INTEGRATE, 0, 1))
    with pytest.raises(ValueError):
        rt.precompile("", rt.make_desc(rt.KIND_INTEGRATE, 1, 9))


def test_no_gpu_means_loud_failure():
    """There is no CPU fallback: without a GPU the engine refuses to exist (src/lib.rs:26-28 message)."""
    if rt.device_count() > 0:
        pytest.skip("a GPU is visible")
    from wgpu_montecarlo import MonteCarloIntegrator

    with pytest.raises(RuntimeError, match="Failed to initialize GPU"):
        MonteCarloIntegrator()
    with pytest.raises(RuntimeError, match="Failed to initialize GPU"):
        rt.Engine(0)


def test_one_hip_runtime_shared_with_torch():
    """libmcx links no HIP runtime; it binds to the instance torch uses (PyTorch-ROCm bundles its own)."""
    import subprocess
    import sys

    out = subprocess.run(["ldd", str(rt.LIB_PATH)], capture_output=True, text=True).stdout
    assert "amdhip64" not in out and "hsa-runtime" not in out, out
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from wgpu_montecarlo import runtime as rt\n"
        "first = rt.hip_runtime()\n"
        "import torch\n"
        "torch.cuda.is_available()\n"
        "maps = set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l)\n"
        "print(first); print(len(maps)); print(sorted(maps))\n" % str(ROOT / "wgpu-monte-carlo_amd"))
    for order in (code, "import torch\n" + code):
        res = subprocess.run([sys.executable, "-c", order], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = res.stdout.strip().splitlines()
        assert lines[-2] == "1", lines            # exactly one libamdhip64 mapped in the process
        assert "torch" in lines[-1]


def test_product_never_imports_the_oracle():
    pkg = ROOT / "wgpu-monte-carlo_amd"
    for path in list(pkg.rglob("*.py")) + list(pkg.rglob("*.cpp")) + list(pkg.rglob("*.hpp")) + list(pkg.rglob("*.h")):
        text = path.read_text()
        assert "import oracle" not in text and "liboracle" not in text and "mcx_oracle" not in text, path


def test_table_cells_strict_grid_and_accuracy():
    """mcx_table_cells: slope-intercept cells only for f32-linspace keys; same interpolant as the reference's lookup."""
    import oracle

    xs = np.linspace(-10.0, 10.0, 2048).astype(np.float32)
    vs = (-0.5 * xs.astype(np.float64) ** 2 - 0.9).astype(np.float32)
    cells = rt.table_cells(xs, vs)
    assert cells is not None and cells.shape == (2047, 2)
    rng = np.random.default_rng(3)
    x = rng.uniform(-10, 10, 4000).astype(np.float32)
    inv_dk = np.float32(2047 / 20.0)
    g = np.clip((x - xs[0]) * inv_dk, 0, 2046).astype(np.int64)
    got = (cells[g, 1].astype(np.float64) * x + cells[g, 0]).astype(np.float32)      # fmaf
    want = np.array([oracle.table_lookup(xs, vs, float(v), -100.0) for v in x])
    assert np.max(np.abs(got - want)) < 2e-5            # |v| up to 50: a few f32 ulps of the table values
    # nearly uniform is not enough (the arithmetic cell guess is not verified on the cell path)
    bumpy = xs.copy()
    bumpy[1000] += 0.1 * (xs[1] - xs[0])
    assert rt.table_cells(bumpy, vs) is None
    assert rt.table_cells(np.array([0.0, 1.0, 3.0], np.float32), np.zeros(3, np.float32)) is None
    assert rt.table_cells(np.array([1.0, 1.0], np.float32), np.zeros(2, np.float32)) is None
    # reference-built tables are strict grids
    from wgpu_montecarlo import Distribution

    tx, tl = Distribution.normal(0.0, 2.0).get_log_pdf_table()
    assert rt.table_cells(tx, tl) is not None
    d = Distribution.from_pdf_table(np.linspace(0, 10, 512), np.exp(-np.linspace(0, 10, 512)))
    assert rt.table_cells(d._x_table, d._pdf_table) is not None


@pytest.mark.parametrize("lo,hi,n", [(-9.0, 9.0, 2048), (0.0, 10.0, 512), (-10.0, 10.0, 4096), (100.0, 101.0, 300), (-6.0, 6.0, 65536)])
def test_cell_index_map_keeps_both_table_ends_inside(lo, hi, n):
    """idx = floor(fma(x, scale, c0)) clamped to [0, n]: 0 / n are the outside sentinels, 1 + c is cell c. Every key,
    both end points included, must land in a real cell next to it; points clearly outside must hit a sentinel."""
    keys = np.linspace(lo, hi, n).astype(np.float32)
    scale, c0 = rt.table_cell_map(keys)
    fma = lambda x: (x.astype(np.float64) * float(scale) + float(c0)).astype(np.float32)
    idx = np.clip(np.floor(fma(keys)), 0, n).astype(np.int64)
    assert idx[0] == 1 and idx[-1] == n - 1
    want = np.arange(n) + 1                    # node i opens cell i (padded index i + 1); the last node closes cell n - 2
    assert np.all((idx == want) | (idx == want - 1))
    dk = (hi - lo) / (n - 1)
    rng = np.random.default_rng(1)
    x = rng.uniform(lo, hi, 200_000).astype(np.float32)
    true_cell = np.clip(np.floor((x.astype(np.float64) - lo) / dk), 0, n - 2).astype(np.int64) + 1
    got = np.clip(np.floor(fma(x)), 0, n).astype(np.int64)
    assert np.all(np.abs(got - true_cell) <= 1)
    frac = (x.astype(np.float64) - lo) / dk % 1.0
    off = got != true_cell                      # only next to a node
    assert np.all(np.minimum(frac[off], 1 - frac[off]) < 0.08)
    outside = np.array([lo - 0.2 * dk - abs(lo) * 1e-6, hi + 0.2 * dk + abs(hi) * 1e-6, lo - 5 * dk, hi + 5 * dk], np.float32)
    got_out = np.clip(np.floor(fma(outside)), 0, n).astype(np.int64)
    assert list(got_out) == [0, n, 0, n]


def test_default_device_follows_mcx_devices(monkeypatch):
    """MCX_DEVICES maps LOCAL_RANK onto a list of device indices; no GPU needed to resolve it."""
    from wgpu_montecarlo import api

    monkeypatch.setenv("MCX_DEVICES", "4, 5,6,7")
    monkeypatch.setenv("LOCAL_RANK", "2")
    assert api._default_device() == 6
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert api._default_device() == 5                      # wraps: more ranks than listed devices share them
    monkeypatch.setenv("MCX_DEVICES", "a,b")
    with pytest.raises(ValueError, match="MCX_DEVICES"):
        api._default_device()
    monkeypatch.delenv("MCX_DEVICES")
    monkeypatch.setenv("LOCAL_RANK", "0")
    assert api._default_device() == 0


def test_mcmc_block_hint_keeps_every_cu_busy():
    """One chain per thread: the workgroup size of an MCMC launch follows the chain count of the rank's shard (C4 over
    1 / 2 / 4 / 8 GPUs: 1 048 576 / 524 288 / 262 144 / 131 072 chains) so that 256 CUs get >= 4 workgroups each."""
    assert [rt.mcmc_block_hint(c) for c in (1_048_576, 524_288, 262_144, 131_072, 65_536, 256, 0)] == [1024, 512, 256, 256, 256, 256, 256]
    assert rt.mcmc_block_hint(2**31) == 1024


def test_versioned_structs_accept_older_layouts_and_refuse_newer_ones():
    """include/mcx.h "ABI versioning": struct_size first; shorter = an older caller (missing fields read as 0), longer =
    a caller built against a newer header (refused), 0 = never initialised (refused)."""
    lib = rt.load()
    assert int(lib.mcx_abi_version()) == rt.ABI_VERSION == 4
    desc = rt.make_desc(rt.KIND_MCMC, 2, rt.DIST_NORMAL, second_moments=True, walk=rt.WALK_ADAPTIVE)
    assert desc.struct_size == C.sizeof(rt.ModuleDesc)
    assert rt.result_rows(desc) == 3 * 2 + 1 + 1

    class FirstRelease(C.Structure):               # the desc up to `unit_params`
        _fields_ = rt.ModuleDesc._fields_[:13]
    old = FirstRelease(C.sizeof(FirstRelease), rt.KIND_MCMC, 2, rt.DIST_NORMAL)
    old.guard_endpoints, old.tables_lds = 1, 1
    rows = lib.mcx_result_rows(C.cast(C.byref(old), C.POINTER(rt.ModuleDesc)))
    assert rows == 3                                # second_moments / walk are not in that layout: they read as 0

    class Newer(C.Structure):
        _fields_ = rt.ModuleDesc._fields_ + [("from_the_future", C.c_int32 * 3)]
    new = Newer()
    C.memmove(C.byref(new), C.byref(desc), C.sizeof(rt.ModuleDesc))
    new.struct_size = C.sizeof(Newer)
    assert lib.mcx_result_rows(C.cast(C.byref(new), C.POINTER(rt.ModuleDesc))) == rt.E_INVALID
    assert "newer mcx.h" in rt.last_error()
    desc.struct_size = 0
    with pytest.raises(ValueError, match="struct_size is 0"):
        rt.result_rows(desc)
    desc.struct_size = 8                            # smaller than the first release
    with pytest.raises(ValueError, match="smaller than the first release"):
        rt.result_rows(desc)


def test_module_key_names_the_cached_code_object():
    """mcx_module_key: the 32-hex-digit name of the code object in the cache directory (what profiles/ cites)."""
    from wgpu_montecarlo.api import functions_to_hip

    src = functions_to_hip([lambda x: x * 3.25])
    desc = rt.make_desc(rt.KIND_INTEGRATE, 1, rt.DIST_UNIFORM)
    key = rt.module_key(src, desc)
    assert re.fullmatch(r"[0-9a-f]{32}", key)
    rt.precompile(src, desc)
    assert (Path(rt.cache_dir()) / f"{key}.hsaco").exists()
    assert rt.module_key(src, rt.make_desc(rt.KIND_INTEGRATE, 1, rt.DIST_NORMAL)) != key
