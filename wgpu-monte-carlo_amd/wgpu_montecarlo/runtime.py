"""ctypes binding of libmcx.so (C ABI: include/mcx.h).

This is the seam where the reference crosses from Python into its PyO3 module
`wgpu_montecarlo._core` (reference __init__.py:49-57, src/lib.rs:516-520). There is no CPU fallback:
if the shared library is missing the import fails, and if no gfx950 GPU is visible creating an engine
raises RuntimeError("Failed to initialize GPU: ...") exactly where the reference does (src/lib.rs:26-28).
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import threading
import weakref
from pathlib import Path
from typing import Optional

import numpy as np

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MCX_LIBRARY", str(_PKG_DIR / "libmcx.so")))

DIST_UNIFORM, DIST_NORMAL, DIST_EXPONENTIAL, DIST_CUSTOM = 0, 1, 2, 3
TABLE_CDF, TABLE_PDF, TABLE_LOGPDF = 0, 1, 2
KIND_INTEGRATE, KIND_MCMC = 0, 1
RNG_PCG_REF, RNG_PHILOX = 0, 1
WALK_INDEPENDENT, WALK_RANDOM, WALK_RANDOM_SYMMETRIC, WALK_ADAPTIVE = 0, 1, 2, 3
RNG_CODES = {"pcg_ref": RNG_PCG_REF, "philox": RNG_PHILOX}

E_INVALID, E_RUNTIME, E_COMPILE, E_NODEVICE = -1, -2, -3, -4
LDS_PER_CU = 160 * 1024          # gfx950 (csrc/mcx_runtime.cpp: kLdsPerCu)

DIST_CODES = {"uniform": DIST_UNIFORM, "normal": DIST_NORMAL, "exponential": DIST_EXPONENTIAL, "custom": DIST_CUSTOM}


class Dispatch(C.Structure):
    _fields_ = [("workgroup_size", C.c_uint32), ("workgroup_count", C.c_uint32),
                ("loops_per_thread", C.c_uint32), ("total_threads", C.c_uint32)]


class Shard(C.Structure):
    _fields_ = [("idx_begin", C.c_uint32), ("idx_count", C.c_uint32),
                ("unit_begin", C.c_uint32), ("unit_end", C.c_uint32)]


class ModuleDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("kind", C.c_int32), ("k", C.c_int32), ("dist_type", C.c_int32), ("weight", C.c_int32),
                ("p_table", C.c_int32), ("q_table", C.c_int32), ("guard_endpoints", C.c_int32),
                ("precise_sampler", C.c_int32), ("block", C.c_int32), ("tables_lds", C.c_int32),
                ("rng", C.c_int32), ("unit_params", C.c_int32), ("second_moments", C.c_int32),
                ("walk", C.c_int32), ("cell_tables", C.c_int32), ("q_sampler", C.c_int32),
                ("moment_family", C.c_int32), ("user_tables", C.c_int32), ("logpdf_analytic", C.c_int32),
                ("cdf_direct", C.c_int32), ("cell_noclamp", C.c_int32), ("cell_addr16", C.c_int32)]


class IntegrateParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32), ("n_samples", C.c_uint64), ("target_threads", C.c_int64), ("seed", C.c_uint32),
                ("param1", C.c_float), ("param2", C.c_float), ("rank", C.c_uint32), ("world", C.c_uint32),
                ("cdf", C.c_void_p), ("target_pdf", C.c_void_p), ("proposal_pdf", C.c_void_p)]


class McmcParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_steps", C.c_uint32), ("n_chains", C.c_uint32), ("n_burnin", C.c_uint32),
                ("target_threads", C.c_int64), ("seed", C.c_uint32), ("param1", C.c_float),
                ("param2", C.c_float), ("rank", C.c_uint32), ("world", C.c_uint32),
                ("cdf", C.c_void_p), ("target_logpdf", C.c_void_p), ("proposal_logpdf", C.c_void_p),
                ("x0", C.c_float), ("target_accept", C.c_float)]


class TableFacts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n", C.c_uint32), ("has_cells", C.c_uint32), ("direct_bits", C.c_uint32),
                ("guide_bits", C.c_uint32), ("lds_bytes", C.c_uint32), ("inv_dk", C.c_float), ("value_min", C.c_float),
                ("value_max", C.c_float), ("reach_known", C.c_uint32)]


class CallInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_blocks", C.c_uint32), ("block", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("launches", C.c_uint32), ("kernel_ms", C.c_float), ("segments", C.c_uint32)]


_SZ_INTEGRATE, _SZ_MCMC = C.sizeof(IntegrateParams), C.sizeof(McmcParams)
SEGMENTS_AUTO = 0xFFFFFFFF
ABI_VERSION = 4

# every symbol include/mcx.h declares (tests check that the library exports all of them)
EXPORTED_SYMBOLS = [
    "mcx_version", "mcx_last_error", "mcx_hip_runtime", "mcx_dispatch_config", "mcx_mcmc_dispatch_config", "mcx_shard_integrate",
    "mcx_shard_units", "mcx_shard_chains", "mcx_device_count", "mcx_engine_create", "mcx_engine_destroy", "mcx_engine_device",
    "mcx_engine_last_kernel_ms", "mcx_engine_last_launch", "mcx_engine_set_target_threads", "mcx_module_build",
    "mcx_module_precompile", "mcx_result_rows", "mcx_module_source", "mcx_free", "mcx_module_release", "mcx_cache_dir",
    "mcx_table_create", "mcx_table_release", "mcx_table_info", "mcx_table_lds_bytes", "mcx_table_cell_map", "mcx_table_has_cells", "mcx_table_cells", "mcx_integrate", "mcx_integrate_device",
    "mcx_mcmc", "mcx_mcmc_device", "mcx_integrate_multi", "mcx_mcmc_multi",
    "mcx_engine_last_launch_count", "mcx_module_static_lds", "mcx_lds_table_budget", "mcx_rccl_library",
    "mcx_comm_create", "mcx_comm_destroy", "mcx_comm_size", "mcx_integrate_comm", "mcx_mcmc_comm",
    "mcx_selftest_streams", "mcx_set_max_launch_units", "mcx_table_has_direct", "mcx_mcmc_block_hint", "mcx_cell_pads", "mcx_cell_pads_host", "mcx_default_launch_blocks", "mcx_engine_set_mcmc_segments",
    "mcx_abi_version", "mcx_module_key", "mcx_table_analyse", "mcx_table_facts_of", "mcx_engine_last_call", "mcx_module_block",
    "mcx_module_desc_fit", "mcx_module_build_fitted", "mcx_module_desc_fit_host", "mcx_wgsl_translate", "mcx_wgsl_prelude", "mcx_wgsl_plan",
    "mcx_core_create", "mcx_core_destroy", "mcx_core_engine", "mcx_core_integrate", "mcx_core_mcmc",
]

_lib = None
_lib_lock = threading.Lock()


def _share_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels ship their own libamdhip64.so. libmcx binds at run time to the HIP runtime that is
    already in the process; if torch is installed but not imported yet, map ITS runtime first (without
    importing torch) so that a later `import torch` and libmcx use one runtime instance -- otherwise torch's
    stream handles would belong to a different runtime than the one our kernels are launched with."""
    if os.environ.get("MCX_HIP_RUNTIME"):
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    lib_dir = Path(list(spec.submodule_search_locations)[0]) / "lib"
    cand = lib_dir / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
            os.environ["MCX_HIP_RUNTIME"] = str(cand)
        except OSError:
            return
        # the same goes for RCCL (mcx_comm_*): torch's librccl.so is linked against torch's HIP runtime, the system
        # one against the system runtime -- bind the one that matches the runtime just mapped
        rccl = lib_dir / "librccl.so"
        if rccl.exists() and not os.environ.get("MCX_RCCL"):
            os.environ["MCX_RCCL"] = str(rccl)


def load():
    """Load libmcx.so (raises ImportError if it has not been built)."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not LIB_PATH.exists():
            raise ImportError(
                f"libmcx.so not found at {LIB_PATH}. Build it with `make -C wgpu-monte-carlo_amd/csrc` "
                f"(or `python -c 'import __graft_entry__ as g; g.build()'`)."
            )
        _share_torch_hip_runtime()
        try:
            L = C.CDLL(str(LIB_PATH))
        except OSError as exc:
            raise ImportError(f"could not load {LIB_PATH}: {exc}")
        vp, u32, i64, u64 = C.c_void_p, C.c_uint32, C.c_int64, C.c_uint64
        L.mcx_version.restype = C.c_char_p
        L.mcx_last_error.restype = C.c_char_p
        L.mcx_cache_dir.restype = C.c_char_p
        L.mcx_hip_runtime.restype = C.c_char_p
        L.mcx_dispatch_config.argtypes = [u64, i64, C.POINTER(Dispatch)]
        L.mcx_mcmc_dispatch_config.argtypes = [u32, i64, C.POINTER(Dispatch)]
        L.mcx_shard_integrate.argtypes = [C.POINTER(Dispatch), C.c_int, u32, u32, C.POINTER(Shard)]
        L.mcx_shard_units.argtypes = [C.POINTER(Dispatch), u32, u32, u32, C.POINTER(Shard)]
        L.mcx_shard_chains.argtypes = [u32, u32, u32, C.POINTER(u32), C.POINTER(u32)]
        L.mcx_engine_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.mcx_engine_destroy.argtypes = [vp]
        L.mcx_engine_destroy.restype = None
        L.mcx_engine_device.argtypes = [vp]
        L.mcx_engine_last_kernel_ms.argtypes = [vp]
        L.mcx_engine_last_kernel_ms.restype = C.c_float
        L.mcx_engine_last_launch.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
        L.mcx_engine_set_target_threads.argtypes = [vp, u32]
        L.mcx_engine_set_mcmc_segments.argtypes = [vp, u32]
        L.mcx_module_build.argtypes = [vp, C.c_char_p, C.POINTER(ModuleDesc), C.POINTER(vp)]
        L.mcx_module_precompile.argtypes = [C.c_char_p, C.POINTER(ModuleDesc), C.POINTER(C.c_int)]
        L.mcx_wgsl_translate.argtypes = [C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(vp)]
        L.mcx_wgsl_prelude.restype = C.c_char_p
        L.mcx_wgsl_plan.argtypes = [vp, C.POINTER(ModuleDesc), C.POINTER(vp)]
        L.mcx_module_desc_fit.argtypes = [C.POINTER(ModuleDesc), vp, vp, vp, C.c_float, C.c_float, C.POINTER(u32)]
        L.mcx_module_build_fitted.argtypes = [vp, C.c_char_p, C.POINTER(ModuleDesc), vp, vp, vp, u32, C.POINTER(vp)]
        L.mcx_module_desc_fit_host.argtypes = [C.POINTER(ModuleDesc), C.POINTER(TableFacts), C.POINTER(TableFacts), C.POINTER(C.c_float),
                                               C.POINTER(TableFacts), C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(u32)]
        L.mcx_result_rows.argtypes = [C.POINTER(ModuleDesc)]
        L.mcx_module_source.argtypes = [C.c_char_p, C.POINTER(ModuleDesc), C.POINTER(vp)]
        L.mcx_free.argtypes = [vp]
        L.mcx_free.restype = None
        L.mcx_module_release.argtypes = [vp]
        L.mcx_module_release.restype = None
        L.mcx_table_create.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), u32, C.POINTER(vp)]
        L.mcx_table_release.argtypes = [vp]
        L.mcx_table_release.restype = None
        L.mcx_table_info.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_float), C.POINTER(u32)]
        L.mcx_table_cell_map.argtypes = [C.POINTER(C.c_float), u32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.mcx_table_has_cells.argtypes = [vp]
        L.mcx_table_has_direct.argtypes = [vp]
        L.mcx_cell_pads.argtypes = [vp, C.c_int32, C.c_float, C.c_float, vp, C.c_int32, C.POINTER(u32), C.POINTER(u32)]
        L.mcx_default_launch_blocks.argtypes = [C.c_uint64, u32, u32]
        L.mcx_default_launch_blocks.restype = u32
        L.mcx_cell_pads_host.argtypes = [C.POINTER(C.c_float), u32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_float, C.c_float,
                                         C.c_int32, C.POINTER(u32), C.POINTER(u32)]
        L.mcx_mcmc_block_hint.argtypes = [u32]
        L.mcx_mcmc_block_hint.restype = u32
        L.mcx_table_lds_bytes.argtypes = [vp]
        L.mcx_table_lds_bytes.restype = u32
        L.mcx_table_cells.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), u32, C.POINTER(C.c_float)]
        L.mcx_integrate.argtypes = [vp, vp, C.POINTER(IntegrateParams), C.POINTER(C.c_double), C.POINTER(u64)]
        L.mcx_integrate_device.argtypes = [vp, vp, C.POINTER(IntegrateParams), vp, vp, C.POINTER(u64)]
        L.mcx_mcmc.argtypes = [vp, vp, C.POINTER(McmcParams), C.POINTER(C.c_double), C.POINTER(u64)]
        L.mcx_mcmc_device.argtypes = [vp, vp, C.POINTER(McmcParams), vp, vp, C.POINTER(u64)]
        L.mcx_integrate_multi.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(C.POINTER(IntegrateParams)), C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(u64)]
        L.mcx_mcmc_multi.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(C.POINTER(McmcParams)), C.c_int,
                                     C.POINTER(C.c_double), C.POINTER(u64)]
        L.mcx_engine_last_launch_count.argtypes = [vp]
        L.mcx_engine_last_launch_count.restype = u32
        L.mcx_module_static_lds.argtypes = [vp]
        L.mcx_module_static_lds.restype = u32
        L.mcx_lds_table_budget.argtypes = [C.POINTER(ModuleDesc)]
        L.mcx_lds_table_budget.restype = u32
        L.mcx_rccl_library.restype = C.c_char_p
        L.mcx_comm_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp)]
        L.mcx_comm_destroy.argtypes = [vp]
        L.mcx_comm_destroy.restype = None
        L.mcx_comm_size.argtypes = [vp]
        L.mcx_integrate_comm.argtypes = [vp, C.POINTER(vp), C.POINTER(C.POINTER(IntegrateParams)), C.POINTER(C.c_double),
                                         C.POINTER(u64)]
        L.mcx_mcmc_comm.argtypes = [vp, C.POINTER(vp), C.POINTER(C.POINTER(McmcParams)), C.POINTER(C.c_double), C.POINTER(u64)]
        u32p = C.POINTER(u32)
        L.mcx_selftest_streams.argtypes = [vp, u32, u32p, u32p, u32p, u32p, u32p, C.POINTER(C.c_float), u32, u32p, u32p, u32p]
        L.mcx_set_max_launch_units.argtypes = [u64]
        L.mcx_set_max_launch_units.restype = None
        L.mcx_abi_version.restype = u32
        L.mcx_module_key.argtypes = [C.c_char_p, C.POINTER(ModuleDesc), C.c_char_p]
        L.mcx_table_analyse.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), u32, C.POINTER(TableFacts)]
        L.mcx_table_facts_of.argtypes = [vp, C.POINTER(TableFacts)]
        L.mcx_engine_last_call.argtypes = [vp, C.POINTER(CallInfo), C.c_int]
        L.mcx_module_block.argtypes = [vp]
        L.mcx_module_block.restype = u32
        if int(L.mcx_abi_version()) != ABI_VERSION:
            raise ImportError(f"{LIB_PATH} has ABI version {L.mcx_abi_version()}, this binding expects {ABI_VERSION}: rebuild it")
        _lib = L
        return _lib


def last_error() -> str:
    return (load().mcx_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map a C status to the exception type the reference raises for the same condition."""
    if rc == 0:
        return
    msg = last_error()
    if rc == E_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


# ---- planning helpers (no GPU needed) ---------------------------------------------------------------
def dispatch_config(n_samples: int, target_threads: Optional[int] = None) -> Dispatch:
    d = Dispatch()
    check(load().mcx_dispatch_config(int(n_samples), int(target_threads or 0), C.byref(d)))
    return d


def mcmc_dispatch_config(n_chains: int, target_threads: Optional[int] = None) -> Dispatch:
    d = Dispatch()
    check(load().mcx_mcmc_dispatch_config(int(n_chains), int(target_threads or 0), C.byref(d)))
    return d


def shard_integrate(d: Dispatch, dist_type: int, rank: int, world: int) -> Shard:
    s = Shard()
    check(load().mcx_shard_integrate(C.byref(d), int(dist_type), int(rank), int(world), C.byref(s)))
    return s


def shard_units(d: Dispatch, iterations_per_unit: int, rank: int, world: int) -> Shard:
    s = Shard()
    check(load().mcx_shard_units(C.byref(d), int(iterations_per_unit), int(rank), int(world), C.byref(s)))
    return s


def mcmc_block_hint(chains: int) -> int:
    """Workgroup size for an MCMC launch of `chains` chains on one GPU (include/mcx.h: mcx_mcmc_block_hint)."""
    return int(load().mcx_mcmc_block_hint(int(chains)))


def shard_chains(total_chains: int, rank: int, world: int):
    b, n = C.c_uint32(0), C.c_uint32(0)
    check(load().mcx_shard_chains(int(total_chains), int(rank), int(world), C.byref(b), C.byref(n)))
    return int(b.value), int(n.value)


def make_desc(kind: int, k: int, dist_type: int, weight: bool = False, p_table: bool = False,
              q_table: bool = False, guard_endpoints: bool = True, precise_sampler: bool = False,
              block: int = 0, tables_lds: bool = True, rng: int = 0, second_moments: bool = False,
              unit_params: bool = False, walk: int = 0, cell_tables: bool = False,
              q_sampler: bool = False, moment_family: bool = False, user_tables: int = 0,
              logpdf_analytic: int = 0, cdf_direct: bool = False, cell_noclamp: bool = False,
              cell_addr16: bool = False) -> ModuleDesc:
    if not block and os.environ.get("MCX_BLOCK"):        # tuning knob
        block = int(os.environ["MCX_BLOCK"])
    return ModuleDesc(C.sizeof(ModuleDesc), kind, k, dist_type, int(weight), int(p_table), int(q_table), int(guard_endpoints),
                      int(precise_sampler), int(block), int(tables_lds), int(rng), int(unit_params), int(second_moments),
                      int(walk), int(cell_tables), int(q_sampler), int(moment_family), int(user_tables), int(logpdf_analytic),
                      int(cdf_direct), int(cell_noclamp), int(cell_addr16))


class WgslProgram(C.Structure):
    """include/mcx.h: mcx_wgsl_program"""
    _fields_ = [("struct_size", C.c_uint32), ("kind", C.c_int32), ("k", C.c_int32), ("functions", C.POINTER(C.c_char_p)),
                ("dist_type", C.c_int32), ("param1", C.c_float), ("param2", C.c_float), ("math", C.c_int32),
                ("have_target_table", C.c_int32), ("have_proposal_table", C.c_int32), ("target_dist_type", C.c_int32),
                ("target_param1", C.c_float), ("target_param2", C.c_float)]


def wgsl_plan(kind: int, functions, dist_type: int, p1: float, p2: float, math: str, have_target: bool, have_proposal: bool,
              target_dist_type: int = 0, t1: float = 0.0, t2: float = 0.0):
    """libmcx's planning of one payload of the reference's native module (include/mcx.h: mcx_wgsl_plan): K WGSL strings ->
    (HIP text, desc with the structural fields set: weight / p_table / q_table / q_sampler, or user_tables / moment_family, or
    logpdf_analytic). Needs no GPU. Raises TranspilerError for text outside the translator's subset."""
    from .frontend import TranspilerError

    codes = {"precise": 0, "default": 1, "fast": 2}
    if math not in codes:
        raise ValueError(f"math must be one of {tuple(codes)}")
    if not all(isinstance(f, str) for f in functions):
        raise TypeError("WGSL function strings expected")
    texts = (C.c_char_p * max(len(functions), 1))(*[f.encode() for f in functions])
    prog = WgslProgram(C.sizeof(WgslProgram), int(kind), len(functions), C.cast(texts, C.POINTER(C.c_char_p)), int(dist_type), float(p1),
                       float(p2), codes[math], int(bool(have_target)), int(bool(have_proposal)), int(target_dist_type), float(t1), float(t2))
    desc = ModuleDesc()
    out = C.c_void_p()
    rc = load().mcx_wgsl_plan(C.byref(prog), C.byref(desc), C.byref(out))
    if rc == -5:
        raise TranspilerError(last_error())
    check(rc)
    try:
        return C.string_at(out).decode(), desc
    finally:
        load().mcx_free(out)


def module_desc_fit(desc: ModuleDesc, cdf, t0, t1, p1: float, p2: float) -> int:
    """libmcx's performance planning of one call (include/mcx.h: mcx_module_desc_fit / mcx_module_desc_fit_host): which table
    forms, LDS staging and workgroup size the call gets. Fills cell_tables, cell_noclamp, cell_addr16, tables_lds, cdf_direct,
    unit_params and block of `desc` in place; returns the LDS bytes of the sentinel pads. Tables are resident ones (Table) or, for
    a planner without a device, HostTables."""
    pad = C.c_uint32(0)
    live = [t for t in (cdf, t0, t1) if t is not None]
    if live and all(isinstance(t, HostTable) for t in live):
        fp = C.POINTER(C.c_float)
        facts = lambda t: C.byref(t.facts) if t is not None else None
        keys = lambda t: t.keys.ctypes.data_as(fp) if t is not None else None
        check(load().mcx_module_desc_fit_host(C.byref(desc), facts(cdf), facts(t0), keys(t0), facts(t1), keys(t1), float(p1), float(p2),
                                              C.byref(pad)))
    else:
        h = lambda t: t._h if t is not None else None
        check(load().mcx_module_desc_fit(C.byref(desc), h(cdf), h(t0), h(t1), float(p1), float(p2), C.byref(pad)))
    return int(pad.value)


def cell_pads(table: "Table", dist_type: int, p1: float, p2: float, cdf: Optional["Table"] = None, guard: bool = True):
    """(pad_l, pad_r): the sentinel cells a cell_noclamp launch adds either side of `table` when the call samples from
    (dist_type, p1, p2[, cdf]); None when the table has no cell form or the sampler's range is unbounded / too wide."""
    if isinstance(table, HostTable):             # planning without a device: the same arithmetic from the keys alone
        x_range = (cdf.value_min, cdf.value_max) if (cdf is not None and cdf.reach_known) else None
        if dist_type == DIST_CUSTOM and x_range is None:
            return None
        return cell_pads_host(table.keys, dist_type, p1, p2, x_range, guard) if table.has_cells else None
    pl, pr = C.c_uint32(), C.c_uint32()
    ok = load().mcx_cell_pads(table._h, int(dist_type), float(p1), float(p2), cdf._h if cdf is not None else None, int(guard),
                              C.byref(pl), C.byref(pr))
    return (pl.value, pr.value) if ok == 1 else None


def default_launch_blocks(samples: int, lds_bytes: int, block: int) -> int:
    """Workgroups a launch aims for by default (include/mcx.h: mcx_default_launch_blocks)."""
    return int(load().mcx_default_launch_blocks(int(samples), int(lds_bytes), int(block)))


def cell_pads_host(keys, dist_type: int, p1: float, p2: float, x_range=None, guard: bool = True):
    """cell_pads() from the table's keys alone (no GPU): (pad_l, pad_r) or None. x_range = (min, max) of the x column of
    the sampling CDF table for DIST_CUSTOM."""
    keys = np.ascontiguousarray(keys, dtype=np.float32)
    pl, pr = C.c_uint32(), C.c_uint32()
    lo, hi = x_range if x_range is not None else (0.0, 0.0)
    ok = load().mcx_cell_pads_host(keys.ctypes.data_as(C.POINTER(C.c_float)), len(keys), int(dist_type), float(p1), float(p2),
                                   int(x_range is not None), float(lo), float(hi), int(guard), C.byref(pl), C.byref(pr))
    return (pl.value, pr.value) if ok == 1 else None


def table_facts(kind: int, keys, values) -> TableFacts:
    """Everything mcx_table_create derives from a table before uploading it, without a device (include/mcx.h:
    mcx_table_analyse): has_cells, direct_bits, guide_bits, lds_bytes, inv_dk, value range, reach_known."""
    k = np.ascontiguousarray(keys, dtype=np.float32)
    v = np.ascontiguousarray(values, dtype=np.float32)
    f = TableFacts(C.sizeof(TableFacts))
    fp = C.POINTER(C.c_float)
    check(load().mcx_table_analyse(int(kind), k.ctypes.data_as(fp), v.ctypes.data_as(fp), len(k), C.byref(f)))
    return f


def module_key(user_src: str, desc: ModuleDesc) -> str:
    """Cache key of the code object (user_src, desc) compiles to: <cache dir>/<key>.hsaco (include/mcx.h: mcx_module_key)."""
    buf = C.create_string_buffer(33)
    check(load().mcx_module_key(user_src.encode(), C.byref(desc), buf))
    return buf.value.decode()


def cache_dir() -> str:
    return (load().mcx_cache_dir() or b"").decode()


def table_cells(keys, values):
    """Per-cell {intercept, slope} of a strict-grid PDF / log-PDF table (float32 [n-1, 2]) or None (include/mcx.h:
    mcx_table_cells)."""
    k = np.ascontiguousarray(keys, dtype=np.float32)
    v = np.ascontiguousarray(values, dtype=np.float32)
    out = np.zeros((max(len(k) - 1, 0), 2), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    rc = int(load().mcx_table_cells(k.ctypes.data_as(fp), v.ctypes.data_as(fp), len(k), out.ctypes.data_as(fp)))
    if rc < 0:
        check(rc)
    return out if rc == 1 else None


def table_cell_map(keys):
    """(scale, c0) of the cell form's index map idx = floor(x * scale + c0) (include/mcx.h: mcx_table_cell_map)."""
    k = np.ascontiguousarray(keys, dtype=np.float32)
    scale, c0 = C.c_float(0), C.c_float(0)
    check(load().mcx_table_cell_map(k.ctypes.data_as(C.POINTER(C.c_float)), len(k), C.byref(scale), C.byref(c0)))
    return np.float32(scale.value), np.float32(c0.value)


def result_rows(desc: ModuleDesc) -> int:
    """Doubles a call with this module writes (include/mcx.h: mcx_result_rows)."""
    rows = int(load().mcx_result_rows(C.byref(desc)))
    if rows < 0:
        check(rows)
    return rows


def set_max_launch_units(units: int) -> None:
    """Work bound of one main-kernel launch (0 = default); larger calls are split (include/mcx.h)."""
    load().mcx_set_max_launch_units(int(units))


def lds_table_budget(desc: ModuleDesc) -> int:
    """LDS bytes a module built from `desc` leaves for staged tables (include/mcx.h: mcx_lds_table_budget)."""
    return int(load().mcx_lds_table_budget(C.byref(desc)))


def module_source(user_src: str, desc: ModuleDesc) -> str:
    out = C.c_void_p()
    check(load().mcx_module_source(user_src.encode(), C.byref(desc), C.byref(out)))
    try:
        return C.string_at(out).decode()
    finally:
        load().mcx_free(out)


def precompile(user_src: str, desc: ModuleDesc) -> int:
    """hiprtc-compile into the disk cache (no GPU needed). Returns 0 = compiled, 1 = memory hit, 2 = disk hit."""
    hit = C.c_int(0)
    check(load().mcx_module_precompile(user_src.encode(), C.byref(desc), C.byref(hit)))
    return int(hit.value)


STREAM_ENGINE = C.c_void_p(-1)        # MCX_STREAM_ENGINE: the engine's own stream


def _stream_arg(stream: Optional[int]):
    """None -> the engine's stream; an integer is a hipStream_t taken literally (0 = HIP's null stream,
    which is what torch.cuda.current_stream().cuda_stream returns for torch's default stream)."""
    return STREAM_ENGINE if stream is None else C.c_void_p(int(stream))


def device_count() -> int:
    return int(load().mcx_device_count())


def hip_runtime() -> str:
    """Path / soname of the HIP runtime libmcx is bound to."""
    return (load().mcx_hip_runtime() or b"").decode()


# ---- handles ---------------------------------------------------------------------------------------
_live_engines = weakref.WeakSet()
_live_comms = weakref.WeakSet()


class Table:
    def __init__(self, engine: "Engine", kind: int, keys: np.ndarray, values: np.ndarray):
        keys = np.ascontiguousarray(keys, dtype=np.float32)
        values = np.ascontiguousarray(values, dtype=np.float32)
        if keys.shape != values.shape or keys.ndim != 1:
            raise ValueError("table keys and values must be 1D arrays of the same length")
        self._engine = engine
        self._h = C.c_void_p()
        fp = C.POINTER(C.c_float)
        check(load().mcx_table_create(engine._h, kind, keys.ctypes.data_as(fp), values.ctypes.data_as(fp),
                                      len(keys), C.byref(self._h)))
        self.kind, self.n = kind, len(keys)
        self.keys, self.values = keys, values                              # kept: the table is re-created per device
        f = TableFacts(C.sizeof(TableFacts))
        check(load().mcx_table_facts_of(self._h, C.byref(f)))
        self.has_cells = f.has_cells == 1                # slope-intercept cell form (strict grid)
        self.direct_bits = int(f.direct_bits)            # CDF tables: bucket-direct records (0: none)
        self.lds_bytes = int(f.lds_bytes)                # staged per workgroup when tables_lds = 1
        self.reach_known = f.reach_known == 1            # CDF tables: every draw stays inside [value_min, value_max]
        self.value_min, self.value_max = float(f.value_min), float(f.value_max)

    def info(self) -> dict:
        n, inv, bits = C.c_uint32(), C.c_float(), C.c_uint32()
        check(load().mcx_table_info(self._h, C.byref(n), C.byref(inv), C.byref(bits)))
        return dict(n=n.value, inv_dk=inv.value, guide_bits=bits.value)

    def release(self) -> None:
        if self._h:
            load().mcx_table_release(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Module:
    def __init__(self, engine: "Engine", user_src: str, desc: ModuleDesc):
        self._engine = engine
        self.desc = desc
        self.user_src = user_src
        self._h = C.c_void_p()
        check(load().mcx_module_build(engine._h, user_src.encode(), C.byref(desc), C.byref(self._h)))
        self.static_lds = int(load().mcx_module_static_lds(self._h))
        self.block = int(load().mcx_module_block(self._h))

    def release(self) -> None:
        if self._h:
            load().mcx_module_release(self._h)      # waits for the module's last launch before unloading the code
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class HostTable:
    """What a planner needs to know about a table, derived without a device (include/mcx.h: mcx_table_analyse)."""

    _h = None

    def __init__(self, kind: int, keys: np.ndarray, values: np.ndarray):
        keys = np.ascontiguousarray(keys, dtype=np.float32)
        values = np.ascontiguousarray(values, dtype=np.float32)
        if keys.shape != values.shape or keys.ndim != 1:
            raise ValueError("table keys and values must be 1D arrays of the same length")
        f = table_facts(kind, keys, values)
        self.facts = f
        self.kind, self.n, self.keys, self.values = kind, len(keys), keys, values
        self.has_cells, self.direct_bits, self.lds_bytes = f.has_cells == 1, int(f.direct_bits), int(f.lds_bytes)
        self.reach_known = f.reach_known == 1
        self.value_min, self.value_max = float(f.value_min), float(f.value_max)


class HostModule:
    """A module that exists as a cached code object only (hiprtc needs no GPU): what HostEngine.module returns."""

    _h = None

    def __init__(self, user_src: str, desc: ModuleDesc):
        self.user_src, self.desc = user_src, desc
        self.cache_hit = precompile(user_src, desc)              # 0 = compiled now, 1 / 2 = memory / disk hit
        self.key = module_key(user_src, desc)
        self.static_lds = LDS_PER_CU - lds_table_budget(desc)    # the upper bound the planner works with
        self.block = int(desc.block)

    @property
    def code_object(self) -> Path:
        return Path(cache_dir()) / f"{self.key}.hsaco"


class HostEngine:
    """Compile-only stand-in for Engine: tables are analysed and modules compiled into the code-object cache exactly as
    a call on a GPU would build them, but nothing can be launched. For warming a deployment's cache in a GPU-less
    build step (__graft_entry__.build) and for inspecting what a call compiles to (tools/issue_model.py)."""

    device = -1
    _h = None

    def __init__(self):
        self._plans = {}               # api._cached_plan

    def cached_table(self, kind: int, keys, values) -> HostTable:
        return HostTable(kind, keys, values)

    def module(self, user_src: str, desc: ModuleDesc) -> HostModule:
        return HostModule(user_src, desc)

    def _no_gpu(self, *a, **k):
        raise RuntimeError("this integrator was built with MonteCarloIntegrator.planner(): it compiles, it cannot launch")

    integrate = mcmc = last_call = last_launch = last_kernel_ms = set_target_threads = set_mcmc_segments = _no_gpu


class Engine:
    """One GPU + one stream (replaces the reference's ComputeEngine, src/engine.rs:70-131)."""

    _shared = {}
    _shared_lock = threading.Lock()

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        check(load().mcx_engine_create(int(device), C.byref(self._h)))
        self.device = int(device)
        self._modules = {}
        self._tables = {}
        self._plans = {}               # api._cached_plan: compiled plans of repeat calls (hold modules + tables)
        self._cache_lock = threading.Lock()      # module / table caches: host threads may share an engine
        _live_engines.add(self)        # closed at interpreter exit while the HIP runtime is still fully alive

    @classmethod
    def shared(cls, device: int = 0) -> "Engine":
        """The process-wide engine of a device: integrators (and the convenience functions, which build a new
        integrator per call like the reference's) share its stream, loaded modules and resident tables."""
        with cls._shared_lock:
            eng = cls._shared.get(int(device))
            if eng is None or not eng._h:
                eng = cls(device)
                cls._shared[int(device)] = eng
            return eng

    @classmethod
    def close_shared(cls) -> None:
        """Release the shared engines (registered with atexit: device objects are freed while the HIP runtime is
        still fully alive, not in whatever order module teardown destroys Python objects)."""
        with cls._shared_lock:
            engines, cls._shared = list(cls._shared.values()), {}
        for eng in engines:
            try:
                eng.close()
            except Exception:
                pass

    def cached_table(self, kind: int, keys, values) -> "Table":
        keys = np.ascontiguousarray(keys, dtype=np.float32)
        values = np.ascontiguousarray(values, dtype=np.float32)
        key = (kind, keys.tobytes(), values.tobytes())
        with self._cache_lock:
            tb = self._tables.pop(key, None)
            if tb is None:
                while len(self._tables) >= 64:
                    self._tables.pop(next(iter(self._tables)))        # oldest first; in-use tables stay referenced by plans
                tb = Table(self, kind, keys, values)
            self._tables[key] = tb                                    # (re-)insert as the most recent
        return tb

    MAX_MODULES = 256

    def module(self, user_src: str, desc: ModuleDesc) -> Module:
        """Loaded module for (source, desc): least recently used out beyond MAX_MODULES. An evicted module is released
        when its last user drops it (plans / prepared integrands hold references), and mcx_module_release waits for
        the module's last launch before it unloads the code object."""
        key = (user_src, bytes(desc))
        with self._cache_lock:
            mod = self._modules.pop(key, None)
            if mod is None:
                mod = Module(self, user_src, desc)
                while len(self._modules) >= self.MAX_MODULES:
                    self._modules.pop(next(iter(self._modules)))     # dicts keep insertion order: the oldest entry
            self._modules[key] = mod                                  # (re-)insert as the most recent
        return mod

    def table(self, kind: int, keys, values) -> Table:
        return Table(self, kind, keys, values)

    def selftest_streams(self, triples, philox_counters=None, philox_keys=None) -> dict:
        """Raw integer streams from the GPU (include/mcx.h: mcx_selftest_streams). triples: uint32 [n, 3] of
        (seed, idx, iter); philox_counters uint32 [m, 4], philox_keys uint32 [m, 2]."""
        t = np.ascontiguousarray(triples, dtype=np.uint32).reshape(-1, 3)
        n = len(t)
        pc = np.ascontiguousarray(philox_counters if philox_counters is not None else np.zeros((0, 4)), dtype=np.uint32).reshape(-1, 4)
        pk = np.ascontiguousarray(philox_keys if philox_keys is not None else np.zeros((0, 2)), dtype=np.uint32).reshape(-1, 2)
        if len(pc) != len(pk):
            raise ValueError("one key per Philox counter")
        out = {name: np.zeros(max(n, 1), dtype=np.uint32) for name in ("combined", "hash", "stepped", "angle")}
        u = np.zeros(max(n, 1), dtype=np.float32)
        po = np.zeros((max(len(pc), 1), 4), dtype=np.uint32)
        u32p = C.POINTER(C.c_uint32)
        ptr = lambda a: a.ctypes.data_as(u32p)
        check(load().mcx_selftest_streams(self._h, n, ptr(t), ptr(out["combined"]), ptr(out["hash"]), ptr(out["stepped"]),
                                          ptr(out["angle"]), u.ctypes.data_as(C.POINTER(C.c_float)), len(pc), ptr(pc), ptr(pk),
                                          ptr(po)))
        res = {k: v[:n] for k, v in out.items()}
        res["u"] = u[:n]
        res["philox"] = po[:len(pc)]
        return res

    def set_mcmc_segments(self, n: int) -> None:
        """MCMC calls of the independence sampler (normal proposal) as two chain halves on two streams x n step segments
        (include/mcx.h: mcx_engine_set_mcmc_segments): SEGMENTS_AUTO (the default) = 8 (4) for launches of >= 131 072 chains with >= 1.4e9 (7e8) chain-steps of work,
        0 = always one launch per call, 2..64 = that many whenever the call qualifies."""
        check(load().mcx_engine_set_mcmc_segments(self._h, int(n)))

    def set_target_threads(self, n: int) -> None:
        check(load().mcx_engine_set_target_threads(self._h, int(n)))

    def last_kernel_ms(self) -> float:
        return float(load().mcx_engine_last_kernel_ms(self._h))

    def last_call(self, with_timing: bool = True) -> dict:
        """Launch geometry and main-kernel time of the last call, one C call (include/mcx.h: mcx_engine_last_call)."""
        info = CallInfo(C.sizeof(CallInfo))
        check(load().mcx_engine_last_call(self._h, C.byref(info), int(with_timing)))
        d = dict(n_blocks=info.n_blocks, block=info.block, lds_bytes=info.lds_bytes, launches=info.launches,
                 segments=info.segments)
        if with_timing:
            d["kernel_ms"] = float(info.kernel_ms)
        return d

    def last_launch(self) -> dict:
        return self.last_call(with_timing=False)      # no wait: the geometry is known when the call is enqueued

    @staticmethod
    def _ptr(t: Optional[Table]):
        return t._h if t is not None else None

    def integrate(self, mod: Module, n_samples: int, seed: int, param1: float, param2: float,
                  target_threads: Optional[int] = None, cdf: Optional[Table] = None,
                  target_pdf: Optional[Table] = None, proposal_pdf: Optional[Table] = None,
                  rank: int = 0, world: int = 1, d_sums: Optional[int] = None, stream: Optional[int] = None):
        """Returns (sums float64[K] or None when d_sums is given, n_eff)."""
        p = IntegrateParams(_SZ_INTEGRATE, 0, int(n_samples), int(target_threads or 0), int(seed) & 0xFFFFFFFF, float(param1),
                            float(param2), int(rank), int(world), self._ptr(cdf), self._ptr(target_pdf),
                            self._ptr(proposal_pdf))
        n_eff = C.c_uint64(0)
        if d_sums is not None:
            check(load().mcx_integrate_device(self._h, mod._h, C.byref(p), C.c_void_p(d_sums),
                                              _stream_arg(stream), C.byref(n_eff)))
            return None, int(n_eff.value)
        sums = np.zeros(mod.desc.k * (2 if mod.desc.second_moments else 1), dtype=np.float64)
        check(load().mcx_integrate(self._h, mod._h, C.byref(p), sums.ctypes.data_as(C.POINTER(C.c_double)),
                                   C.byref(n_eff)))
        return sums, int(n_eff.value)

    def mcmc(self, mod: Module, n_steps: int, n_chains: int, n_burnin: int, seed: int, param1: float,
             param2: float, target_logpdf: Table, proposal_logpdf: Table, target_threads: Optional[int] = None,
             cdf: Optional[Table] = None, rank: int = 0, world: int = 1, d_sums: Optional[int] = None,
             stream: Optional[int] = None, x0: float = 0.0, target_accept: float = 0.44):
        """Returns (sums float64[result_rows(desc)] (row k, or 2k with second moments = accepted steps) or None,
        n_eff)."""
        p = McmcParams(_SZ_MCMC, int(n_steps), int(n_chains), int(n_burnin), int(target_threads or 0),
                       int(seed) & 0xFFFFFFFF, float(param1), float(param2), int(rank), int(world),
                       self._ptr(cdf), self._ptr(target_logpdf), self._ptr(proposal_logpdf), float(x0), float(target_accept))
        n_eff = C.c_uint64(0)
        if d_sums is not None:
            check(load().mcx_mcmc_device(self._h, mod._h, C.byref(p), C.c_void_p(d_sums),
                                         _stream_arg(stream), C.byref(n_eff)))
            return None, int(n_eff.value)
        sums = np.zeros(result_rows(mod.desc), dtype=np.float64)
        check(load().mcx_mcmc(self._h, mod._h, C.byref(p), sums.ctypes.data_as(C.POINTER(C.c_double)),
                              C.byref(n_eff)))
        return sums, int(n_eff.value)

    def close(self) -> None:
        for comm in list(_live_comms):      # a communicator over this engine goes first
            if any(e is self for e in comm._engines):
                comm.close()
        self._plans.clear()
        for mod in list(self._modules.values()):
            mod.release()
        self._modules.clear()
        for tb in list(self._tables.values()):
            tb.release()
        self._tables.clear()
        if self._h:
            load().mcx_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _multi(fn_name: str, shards, make_params, ptype):
    """shards: [(Engine, Module, tables-dict)], one per device; one host thread enqueues all of them, then the K
    doubles of each are added on the host (include/mcx.h: mcx_integrate_multi / mcx_mcmc_multi)."""
    n = len(shards)
    params = [make_params(r, n, eng, tables) for r, (eng, _, tables) in enumerate(shards)]
    engines = (C.c_void_p * n)(*[eng._h for eng, _, _ in shards])
    modules = (C.c_void_p * n)(*[mod._h for _, mod, _ in shards])
    pptr = (C.POINTER(ptype) * n)(*[C.pointer(p) for p in params])
    sums = np.zeros(result_rows(shards[0][1].desc), dtype=np.float64)
    n_eff = C.c_uint64(0)
    check(getattr(load(), fn_name)(engines, modules, pptr, n, sums.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n_eff)))
    return sums, int(n_eff.value)


def integrate_multi(shards, n_samples: int, seed: int, param1: float, param2: float, target_threads: Optional[int] = None):
    """Whole-grid sums from len(shards) engines driven by this thread. tables-dict keys: cdf, target_pdf, proposal_pdf."""
    def make(r, n, eng, tb):
        return IntegrateParams(_SZ_INTEGRATE, 0, int(n_samples), int(target_threads or 0), int(seed) & 0xFFFFFFFF, float(param1), float(param2),
                               r, n, Engine._ptr(tb.get("cdf")), Engine._ptr(tb.get("target_pdf")), Engine._ptr(tb.get("proposal_pdf")))
    return _multi("mcx_integrate_multi", shards, make, IntegrateParams)


def mcmc_multi(shards, n_steps: int, n_chains: int, n_burnin: int, seed: int, param1: float, param2: float,
               target_threads: Optional[int] = None, x0: float = 0.0, target_accept: float = 0.44):
    """Chain-sharded MH over len(shards) engines. tables-dict keys: cdf, target_logpdf, proposal_logpdf."""
    def make(r, n, eng, tb):
        return McmcParams(_SZ_MCMC, int(n_steps), int(n_chains), int(n_burnin), int(target_threads or 0), int(seed) & 0xFFFFFFFF,
                          float(param1), float(param2), r, n, Engine._ptr(tb.get("cdf")), Engine._ptr(tb.get("target_logpdf")),
                          Engine._ptr(tb.get("proposal_logpdf")), float(x0), float(target_accept))
    return _multi("mcx_mcmc_multi", shards, make, McmcParams)


class Comm:
    """RCCL communicator over the devices of several engines, driven by one host thread (include/mcx.h:
    mcx_comm_create). shards as for integrate_multi: [(Engine, Module, tables-dict)], one per distinct device."""

    def __init__(self, engines):
        # When PyTorch is installed, the RCCL that matches the HIP runtime of this process is PyTorch's bundled copy
        # (_share_torch_hip_runtime). It must be brought in by PyTorch's own loader: dlopen-ing it first and importing
        # torch afterwards ends in a double free at interpreter exit (measured on the GPU box: communicator created,
        # then `import torch` -> abort at exit; torch first -> clean). So a process that uses the communicator imports
        # torch here, before libmcx binds RCCL. Without PyTorch the system RCCL and HIP runtime are used.
        import sys

        if "torch" not in sys.modules and os.environ.get("MCX_RCCL", "").endswith("/torch/lib/librccl.so"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        self._engines = list(engines)
        n = len(self._engines)
        self._h = C.c_void_p()
        arr = (C.c_void_p * n)(*[e._h for e in self._engines])
        check(load().mcx_comm_create(arr, n, C.byref(self._h)))
        _live_comms.add(self)          # destroyed at exit before the engines and before the runtimes unload

    @property
    def size(self) -> int:
        return int(load().mcx_comm_size(self._h))

    def _call(self, fn_name, shards, params, ptype):
        n = len(shards)
        if [eng for eng, _, _ in shards] != self._engines:
            raise ValueError("shards must use the communicator's engines, in order")
        modules = (C.c_void_p * n)(*[mod._h for _, mod, _ in shards])
        pptr = (C.POINTER(ptype) * n)(*[C.pointer(p) for p in params])
        sums = np.zeros(result_rows(shards[0][1].desc), dtype=np.float64)
        n_eff = C.c_uint64(0)
        check(getattr(load(), fn_name)(self._h, modules, pptr, sums.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n_eff)))
        return sums, int(n_eff.value)

    def integrate(self, shards, n_samples: int, seed: int, param1: float, param2: float, target_threads: Optional[int] = None):
        n = len(shards)
        params = [IntegrateParams(_SZ_INTEGRATE, 0, int(n_samples), int(target_threads or 0), int(seed) & 0xFFFFFFFF, float(param1), float(param2),
                                  r, n, Engine._ptr(tb.get("cdf")), Engine._ptr(tb.get("target_pdf")),
                                  Engine._ptr(tb.get("proposal_pdf"))) for r, (_, _, tb) in enumerate(shards)]
        return self._call("mcx_integrate_comm", shards, params, IntegrateParams)

    def mcmc(self, shards, n_steps: int, n_chains: int, n_burnin: int, seed: int, param1: float, param2: float,
             target_threads: Optional[int] = None, x0: float = 0.0, target_accept: float = 0.44):
        n = len(shards)
        params = [McmcParams(_SZ_MCMC, int(n_steps), int(n_chains), int(n_burnin), int(target_threads or 0), int(seed) & 0xFFFFFFFF,
                             float(param1), float(param2), r, n, Engine._ptr(tb.get("cdf")), Engine._ptr(tb.get("target_logpdf")),
                             Engine._ptr(tb.get("proposal_logpdf")), float(x0), float(target_accept))
                  for r, (_, _, tb) in enumerate(shards)]
        return self._call("mcx_mcmc_comm", shards, params, McmcParams)

    def close(self) -> None:
        if self._h:
            load().mcx_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_library() -> str:
    return (load().mcx_rccl_library() or b"").decode()


def _close_comms() -> None:
    """Communicators still alive at interpreter exit are destroyed here, while RCCL and the HIP runtime are fully
    alive (a communicator left to __del__ during module teardown segfaulted on the GPU box)."""
    for comm in list(_live_comms):
        try:
            comm.close()
        except Exception:
            pass
    for eng in list(_live_engines):        # engines that are not in Engine._shared (built directly or per duplicate device)
        try:
            eng.close()
        except Exception:
            pass


atexit.register(Engine.close_shared)
atexit.register(_close_comms)           # atexit runs last-registered first: communicators go before their engines
