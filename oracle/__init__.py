"""ctypes binding of the CPU oracle (oracle/mcx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package. See the header of mcx_oracle.c for what it restates and for the
parity status ("float parity unpinned beyond the reference's statistical tolerances").
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
import os as _os

# MCX_ORACLE_LIBRARY selects another build of the same source (tools/sanitize_cpu.sh: the ASan + UBSan build)
_LIB_PATH = Path(_os.environ["MCX_ORACLE_LIBRARY"]) if _os.environ.get("MCX_ORACLE_LIBRARY") else _HERE / "liboracle.so"

UNIFORM, NORMAL, EXPONENTIAL, CUSTOM = 0, 1, 2, 3
FN_IDENTITY, FN_POW, FN_SIN, FN_COS, FN_EXP, FN_GT, FN_BENCH, FN_ABS, FN_CONST, FN_SQ = range(10)
PDF_NONE, PDF_UNIFORM, PDF_NORMAL, PDF_EXPONENTIAL, PDF_TABLE = range(5)


def build(force: bool = False) -> Path:
    """Compile liboracle.so with gcc if missing (or stale)."""
    src = _HERE / "mcx_oracle.c"
    if _os.environ.get("MCX_ORACLE_LIBRARY"):
        return _LIB_PATH
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


class _Fn(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_float)]


class _Pdf(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_float), ("b", C.c_float), ("c", C.c_float),
                ("table", C.POINTER(C.c_float))]


class _K1Args(C.Structure):
    _fields_ = [("n_samples", C.c_uint64), ("target_threads", C.c_int64), ("seed", C.c_uint32),
                ("dist_type", C.c_int32), ("param1", C.c_float), ("param2", C.c_float),
                ("table_size", C.c_uint32), ("cdf_table", C.POINTER(C.c_float)),
                ("x_table", C.POINTER(C.c_float)), ("guard", C.c_int32), ("weighted", C.c_int32),
                ("p", _Pdf), ("q", _Pdf), ("rng", C.c_int32)]


class _McmcArgs(C.Structure):
    _fields_ = [("n_steps", C.c_uint32), ("n_chains", C.c_uint32), ("n_burnin", C.c_uint32),
                ("target_threads", C.c_int64), ("seed", C.c_uint32), ("proposal_type", C.c_int32),
                ("param1", C.c_float), ("param2", C.c_float), ("table_size", C.c_uint32),
                ("cdf_table", C.POINTER(C.c_float)), ("x_table", C.POINTER(C.c_float)),
                ("target_logpdf", C.POINTER(C.c_float)), ("proposal_logpdf", C.POINTER(C.c_float)),
                ("guard", C.c_int32), ("rng", C.c_int32), ("walk", C.c_int32), ("x0", C.c_float),
                ("target_accept", C.c_float), ("target_type", C.c_int32), ("target_p1", C.c_float), ("target_p2", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        L.orc_pcg_hash.restype = C.c_uint32
        L.orc_pcg_hash.argtypes = [C.c_uint32]
        L.orc_combined.restype = C.c_uint32
        L.orc_combined.argtypes = [C.c_uint32] * 3
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_random_uniform.restype = C.c_float
        L.orc_random_uniform.argtypes = [C.c_uint32] * 3
        L.orc_dispatch_config.argtypes = [C.c_uint64, C.c_int64, C.POINTER(C.c_uint32)]
        L.orc_mcmc_dispatch_config.argtypes = [C.c_uint32, C.c_int64, C.POINTER(C.c_uint32)]
        L.orc_table_lookup_xy.restype = C.c_float
        L.orc_table_lookup_xy.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_float, C.c_float]
        L.orc_sample_cdf.restype = C.c_float
        L.orc_sample_cdf.argtypes = [C.c_float, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_k1.argtypes = [C.POINTER(_K1Args), C.POINTER(_Fn), C.c_int, C.POINTER(C.c_float),
                             C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.orc_k1_samples.argtypes = [C.POINTER(_K1Args), C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        L.orc_mcmc.argtypes = [C.POINTER(_McmcArgs), C.POINTER(_Fn), C.c_int, C.POINTER(C.c_float),
                               C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.c_uint32,
                               C.POINTER(C.c_double)]
        L.orc_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def interleave(x, v):
    """[n, x0, v0, x1, v1, ...] -- engine.rs:544-551"""
    x = _f32(x); v = _f32(v)
    out = np.empty(1 + 2 * len(x), dtype=np.float32)
    out[0] = np.float32(len(x))
    out[1::2] = x
    out[2::2] = v
    return out


def pcg_hash(v: int) -> int:
    return int(lib().orc_pcg_hash(C.c_uint32(v & 0xFFFFFFFF)))


def combined(seed: int, idx: int, it: int) -> int:
    return int(lib().orc_combined(seed & 0xFFFFFFFF, idx & 0xFFFFFFFF, it & 0xFFFFFFFF))


def random_uniform(seed: int, idx: int, it: int) -> float:
    return float(lib().orc_random_uniform(seed & 0xFFFFFFFF, idx & 0xFFFFFFFF, it & 0xFFFFFFFF))


def dispatch_config(n_samples: int, target_threads=None):
    out = (C.c_uint32 * 4)()
    lib().orc_dispatch_config(int(n_samples), int(target_threads or 0), out)
    return dict(workgroup_size=out[0], workgroup_count=out[1], loops_per_thread=out[2], total_threads=out[3])


def mcmc_dispatch_config(n_chains: int, target_threads=None):
    out = (C.c_uint32 * 4)()
    lib().orc_mcmc_dispatch_config(int(n_chains), int(target_threads or 0), out)
    return dict(workgroup_size=out[0], workgroup_count=out[1], loops_per_thread=out[2], total_threads=out[3])


def table_lookup(xs, vs, x, outside):
    xs = _f32(xs); vs = _f32(vs)
    return float(lib().orc_table_lookup_xy(_fp(xs), _fp(vs), len(xs), C.c_float(x), C.c_float(outside)))


def sample_cdf(u, cdf, xt):
    cdf = _f32(cdf); xt = _f32(xt)
    return float(lib().orc_sample_cdf(C.c_float(u), len(cdf), _fp(cdf), _fp(xt)))


def _fns(fns):
    arr = (_Fn * len(fns))()
    for i, f in enumerate(fns):
        if isinstance(f, int):
            f = (f, 0.0)
        arr[i].kind, arr[i].a = int(f[0]), float(f[1])
    return arr


def _pdf(spec, keep):
    p = _Pdf()
    if spec is None:
        p.kind = PDF_NONE
        return p
    kind = spec[0]
    p.kind = kind
    if kind == PDF_TABLE:
        data = interleave(spec[1], spec[2])
        keep.append(data)
        p.table = _fp(data)
    else:
        vals = list(spec[1:]) + [0.0, 0.0, 0.0]
        p.a, p.b, p.c = float(vals[0]), float(vals[1]), float(vals[2])
    return p


def philox4x32_10(counter, key):
    """Philox4x32-10 block function (libmcx's opt-in stream; not part of the reference)."""
    c = (C.c_uint32 * 4)(*[v & 0xFFFFFFFF for v in counter])
    k = (C.c_uint32 * 2)(*[v & 0xFFFFFFFF for v in key])
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return tuple(int(v) for v in o)


def _k1_args(n_samples, seed, dist_type, param1, param2, cdf_table, x_table, target_threads, guard, p, q, keep, rng=0):
    a = _K1Args()
    a.n_samples = int(n_samples)
    a.target_threads = int(target_threads or 0)
    a.seed = seed & 0xFFFFFFFF
    a.dist_type = dist_type
    a.param1, a.param2 = float(param1), float(param2)
    cdf_table = _f32(cdf_table); x_table = _f32(x_table)
    keep += [cdf_table, x_table]
    a.table_size = 0 if cdf_table is None else len(cdf_table)
    a.cdf_table, a.x_table = _fp(cdf_table), _fp(x_table)
    a.guard = int(guard)
    a.weighted = int(p is not None or q is not None)
    a.p, a.q = _pdf(p, keep), _pdf(q, keep)
    a.rng = int(rng)
    return a


def integrate(fns, dist_type, param1=0.0, param2=1.0, n_samples=1_000_000, seed=42, cdf_table=None,
              x_table=None, target_threads=None, guard=0, p=None, q=None, rng=0):
    """Restated K1/K2. fns: list of (FN_*, a). p, q: None or (PDF_*, params...) / (PDF_TABLE, x, pdf).

    Returns dict(ref=float32[K] the reference's f32 result, sums=float64[K], n_eff=int)."""
    keep = []
    a = _k1_args(n_samples, seed, dist_type, param1, param2, cdf_table, x_table, target_threads, guard, p, q, keep, rng)
    K = len(fns)
    ref = np.zeros(K, dtype=np.float32)
    sums = np.zeros(K, dtype=np.float64)
    n_eff = C.c_uint64(0)
    rc = lib().orc_k1(C.byref(a), _fns(fns), K, _fp(ref), sums.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n_eff))
    if rc:
        raise MemoryError("oracle allocation failed")
    return dict(ref=ref, sums=sums, n_eff=int(n_eff.value))


def samples(dist_type, param1=0.0, param2=1.0, n_samples=1_000_000, seed=42, cdf_table=None, x_table=None,
            target_threads=None, guard=0, idx0=0, nidx=None, rng=0):
    """x(idx, i) as float32[nidx, L] for evaluating arbitrary functions in numpy."""
    keep = []
    a = _k1_args(n_samples, seed, dist_type, param1, param2, cdf_table, x_table, target_threads, guard, None, None, keep, rng)
    cfg = dispatch_config(n_samples, target_threads)
    if nidx is None:
        nidx = cfg["total_threads"] - idx0
    out = np.empty((nidx, cfg["loops_per_thread"]), dtype=np.float32)
    lib().orc_k1_samples(C.byref(a), idx0, nidx, _fp(out))
    return out


def mcmc(fns, proposal_type, param1, param2, target_x, target_logpdf, proposal_x, proposal_logpdf,
         n_steps=1000, n_chains=256, n_burnin=100, seed=42, cdf_table=None, x_table=None,
         target_threads=None, guard=0, trace_chains=0, rng=0, walk=0, x0=0.0, target_accept=0.44, target_analytic=None):
    """Restated K3. Returns dict(ref, sums (K+1, last = accepted steps), n_eff, trace, sumsq (K), chain_mean_sq (K)).
    walk: 0 independent proposals (the reference), 1 / 2 / 3 libmcx's random-walk extensions (general / symmetric /
    adaptive symmetric with step-scale tuning towards target_accept during burn-in)."""
    a = _McmcArgs()
    a.n_steps, a.n_chains, a.n_burnin = int(n_steps), int(n_chains), int(n_burnin)
    a.target_threads = int(target_threads or 0)
    a.seed = seed & 0xFFFFFFFF
    a.proposal_type = proposal_type
    a.param1, a.param2 = float(param1), float(param2)
    cdf_table = _f32(cdf_table); x_table = _f32(x_table)
    a.table_size = 0 if cdf_table is None else len(cdf_table)
    a.cdf_table, a.x_table = _fp(cdf_table), _fp(x_table)
    # a table given as None selects the reference's analytic log-density (shader_gen.rs:543-571): the target's from
    # target_analytic = (dist_type, p1, p2), the proposal's from (proposal_type, param1, param2)
    tl = interleave(target_x, target_logpdf) if target_x is not None else None
    pl = interleave(proposal_x, proposal_logpdf) if proposal_x is not None else None
    a.target_logpdf, a.proposal_logpdf = _fp(tl), _fp(pl)
    if tl is None:
        a.target_type, a.target_p1, a.target_p2 = int(target_analytic[0]), float(target_analytic[1]), float(target_analytic[2])
    a.guard = int(guard)
    a.rng = int(rng)
    a.walk, a.x0, a.target_accept = int(walk), float(x0), float(target_accept)
    K = len(fns)
    ref = np.zeros(K, dtype=np.float32)
    sums = np.zeros(K + 1, dtype=np.float64)
    diag = np.zeros(2 * K + 1, dtype=np.float64)
    n_eff = C.c_uint64(0)
    trace = np.zeros((trace_chains, n_steps), dtype=np.float32) if trace_chains else None
    rc = lib().orc_mcmc(C.byref(a), _fns(fns), K, _fp(ref), sums.ctypes.data_as(C.POINTER(C.c_double)),
                        C.byref(n_eff), _fp(trace), int(trace_chains), diag.ctypes.data_as(C.POINTER(C.c_double)))
    if rc:
        raise MemoryError("oracle allocation failed")
    return dict(ref=ref, sums=sums, n_eff=int(n_eff.value), trace=trace, sumsq=diag[:K], chain_mean_sq=diag[K:2 * K],
                scale_sum=float(diag[2 * K]))


def num_threads() -> int:
    return int(lib().orc_num_threads())
