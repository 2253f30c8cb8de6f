"""MonteCarloIntegrator / IntegrationResult / convenience functions on MI355X.

Public surface and behaviour follow the reference's python/wgpu_montecarlo/__init__.py:611-1266
(signatures, defaults, exception types and messages, `n_samples` reporting the request rather than
T*L, MCMC `n_samples = n_chains * n_steps`). Underneath, nothing is shared with it:

  reference                                          here
  -------------------------------------------------  -----------------------------------------------
  callable -> WGSL text (transpiler.py)              callable -> IR -> HIP C++ (frontend.py, emit_hip.py)
  IS = text wrappers f*p/q, K copies of p and q      IS is a kernel mode: weight p/q once per sample
  _core.integrate / integrate_is_tables / _mcmc      libmcx.so C ABI via ctypes (runtime.py)
  one wgpu device, [T][K] f32 readback, CPU f32 mean on-device f64 two-stage reduction, K doubles back
  --                                                 torch.distributed (RCCL) shard + one all-reduce
"""
from __future__ import annotations

import os
import threading
from typing import Callable, List, Optional, Sequence, Union

import numpy as np

from . import distributed, emit_hip, frontend, runtime, wgsl_to_hip
from .distributions import _density_key, Distribution, DistributionType
from .frontend import TranspilerError

FunctionLike = Union[Callable, str]

class IntegrationResult:
    """Expected values, one per function, in the order the functions were given."""

    def __init__(self, values: np.ndarray, n_samples: int, n_functions: int, meta: Optional[dict] = None):
        self.values = np.array(values, dtype=np.float64)
        self.n_samples = n_samples
        self.n_functions = n_functions
        self.meta = meta or {}

    def __repr__(self):
        return f"IntegrationResult(values={self.values}, n_samples={self.n_samples})"

    def __getitem__(self, idx):
        return self.values[idx]

    def __len__(self):
        return self.n_functions


def _check_seed(seed) -> int:
    if isinstance(seed, bool) or not isinstance(seed, (int, np.integer)):
        raise TypeError(f"seed must be an integer, got {type(seed).__name__}")
    if not 0 <= int(seed) <= 0xFFFFFFFF:
        raise OverflowError("seed must fit an unsigned 32-bit integer")
    return int(seed)


def _check_count(value, name: str, bits: int = 64) -> int:
    if isinstance(value, bool) or not isinstance(value, (int, np.integer)):
        raise TypeError(f"{name} must be an integer, got {type(value).__name__}")
    if not 0 <= int(value) < (1 << bits):
        raise OverflowError(f"{name} must fit an unsigned {bits}-bit integer")
    return int(value)


def _function_key(fn):
    """Hashable identity of one integrand for the plan cache: frontend.fingerprint for callables (code object + every
    captured value), the text itself for WGSL strings."""
    if isinstance(fn, str):
        return fn
    return frontend.fingerprint(fn)


def _distribution_key(d: Distribution):
    """Identity of a Distribution for the plan cache: its type, its parameters, the value of its density closure (code object +
    captured values + the globals it names, where all of those are plain values: distributions._density_key -- a few attribute
    reads, no source is parsed) and the identities of its table arrays. By value where a value exists, so that
    `Distribution.normal(0, 1)` written inline in every call (the reference's examples and benchmark do) finds the plan of the
    previous call instead of building and caching a new one each time. A density without such a value (it captures arrays, say:
    from_pdf_table) falls back to the object's identity (the cache entry keeps the object alive, so an id() cannot be reused while
    the entry exists). Writing into a table array IN PLACE after it has been used is not seen: tables are treated as immutable once
    a call has seen them."""
    p = d.params
    try:
        density = _density_key(d._pdf_func)
    except (TypeError, ValueError):
        density = None
    if density is None:
        density = (id(d), id(d._pdf_func))
    return (d.dist_type, tuple(p.items()) if p else None, density, id(d._x_table), id(d._cdf_table), id(d._pdf_table))


_PLAN_CACHE_ENTRIES = 128
_PLAN_CACHE_LOCK = threading.Lock()

# Tuning knobs read while a plan is built (here, in runtime.make_desc and in libmcx's source assembly): part of the plan
# key, so flipping one at run time (A/B scripts, tests) is not answered from the cache.
_PLAN_ENV = ("MCX_BLOCK", "MCX_NO_ADDR16", "MCX_NO_NOCLAMP", "MCX_DIRECT_MAX_ROWS", "MCX_NO_DIRECT", "MCX_NO_CELLS",
             "MCX_EXTRA_DEFINES", "MCX_EXTRA_FLAGS")
_ENV_DATA = getattr(os.environ, "_data", None)            # CPython on POSIX: the bytes-keyed dict behind os.environ
_PLAN_ENV_B = tuple(os.fsencode(k) for k in _PLAN_ENV)


def _env_key():
    d = _ENV_DATA
    if d is not None:
        return tuple([d.get(k) for k in _PLAN_ENV_B])       # ~0.4 us; os.environ.get would encode every name
    return tuple([os.environ.get(k) for k in _PLAN_ENV])

_EMIT_CACHE = {}      # (code object | wgsl text, slot, math, captured constants) -> HIP text


def _emit_cached(key, build):
    text = _EMIT_CACHE.get(key)
    if text is None:
        text = build()
        if len(_EMIT_CACHE) > 4096:
            _EMIT_CACHE.clear()
        _EMIT_CACHE[key] = text
    return text


def functions_to_hip(functions: Sequence[FunctionLike], math="default") -> str:
    """Emit `user_func_0 .. user_func_{K-1}` (HIP C++) for callables and raw WGSL strings."""
    parts = [emit_hip.prelude()]
    for i, fn in enumerate(functions):
        if callable(fn):
            ir_fn = frontend.lower(fn, bind_defaults=True)
            name = f"user_func_{i}"
            key = (getattr(fn, "__code__", None), name, math, tuple(ir_fn.consts.items()))
            parts.append(_emit_cached(key, lambda: emit_hip.emit_function(ir_fn, name, math)))
        elif isinstance(fn, str):
            parts.append(_emit_cached((fn, i, "wgsl", math), lambda: wgsl_to_hip.translate(fn, i, f"user_func_{i}", math)))
        else:
            raise TypeError(f"Function must be callable or WGSL string, got {type(fn)}")
    return "\n\n".join(parts)


def build_module(engine, user_src: str, desc, *tables: Optional[runtime.Table], extra_bytes: int = 0):
    """Build (or fetch) the module for `desc`, then hold the plan-time LDS decisions against the code object's REAL
    static LDS (the launch checks it: csrc/mcx_runtime.cpp integrate_impl / mcmc_impl). mcx_module_desc_fit decided with an
    upper bound; should the real figure ever be larger, a default-on optimisation (staged tables, cell_noclamp,
    cell_addr16) must fall back to the slower form, not turn a valid call into MCX_E_INVALID. Shared with the
    `_core` binding (_core.py)."""
    mod = engine.module(user_src, desc)
    if not desc.tables_lds:
        return mod
    need = sum(tb.lds_bytes for tb in tables if tb is not None) + (extra_bytes if desc.cell_noclamp else 0)
    if desc.cdf_direct:
        need += (mod.block // 64) * 128 * 4             # the per-wave queues of the bucket-direct sampler
    total = need + mod.static_lds
    if need and total > runtime.LDS_PER_CU:
        desc.tables_lds, desc.cdf_direct, desc.cell_noclamp, desc.cell_addr16 = 0, 0, 0, 0
        return engine.module(user_src, desc)
    if desc.cell_addr16 and total > 65536:
        desc.cell_addr16 = 0
        return engine.module(user_src, desc)
    return mod


_MOMENT_FAMILY_MIN_K = 8     # below this the per-sample multiply chain is as cheap (K = 4: 10 ops per pair against 12)


def _moment_family(functions: Sequence[FunctionLike]) -> bool:
    """True when function i is exactly x**(i+1) for every i (x, x**2, ...; `x*x` and captured integer exponents count):
    the fused-moments workload, for which the kernel accumulates two samples at a time through Newton's identity for
    power sums (module desc `moment_family`)."""
    if len(functions) < _MOMENT_FAMILY_MIN_K:
        return False
    for i, fn in enumerate(functions):
        if not callable(fn):
            return False
        try:
            ir_fn = frontend.lower(fn, bind_defaults=True)
        except TranspilerError:
            return False
        if len(ir_fn.params) != 1 or len(ir_fn.body) != 1 or not isinstance(ir_fn.body[0], frontend.Return):
            return False
        ret = ir_fn.body[0]
        x = ir_fn.params[0]
        is_x = lambda node: isinstance(node, frontend.Var) and node.name == x and x not in ir_fn.consts
        node, degree = ret.value, None
        if ret.boolean:
            return False
        if is_x(node):
            degree = 1
        elif isinstance(node, frontend.Bin) and node.op == "*" and is_x(node.left) and is_x(node.right):
            degree = 2
        elif isinstance(node, frontend.Pow) and is_x(node.base):
            e = node.exponent
            if isinstance(e, frontend.Var) and e.name in ir_fn.consts:
                e = frontend.Num(ir_fn.consts[e.name])
            if isinstance(e, frontend.Num) and float(e.value).is_integer():
                degree = int(e.value)
        if degree != i + 1:
            return False
    return True


_PDF_OUTSIDE_SUBSET = set()      # code objects of PDF closures the emitter rejected


def _pdf_to_hip(dist: Distribution, name: str, math="default") -> Optional[str]:
    """HIP text of a distribution's PDF closure, or None when it is outside the emitter's subset
    (the reference's table-vs-analytic decision, __init__.py:825-838)."""
    fn = dist._pdf_func
    code = getattr(fn, "__code__", None)
    if code is not None and code in _PDF_OUTSIDE_SUBSET:     # e.g. the np.interp closure of from_pdf_table: decided once
        return None
    try:
        ir_fn = frontend.lower(fn)
        key = (code, name, math, tuple(ir_fn.consts.items()))
        return _emit_cached(key, lambda: emit_hip.emit_function(ir_fn, name, math))
    except TranspilerError:
        if code is not None:
            _PDF_OUTSIDE_SUBSET.add(code)
        return None


def _dist_params(dist: Distribution):
    """(code, param1, param2): lib.rs:436-502 including its silent defaults."""
    code = runtime.DIST_CODES[dist.dist_type.name.lower()]
    p = dist.params

    def num(key, default):
        v = p.get(key, default)
        try:
            return float(v)
        except (TypeError, ValueError):
            return float(default)

    if code == runtime.DIST_UNIFORM:
        return code, num("min", 0.0), num("max", 1.0)
    if code == runtime.DIST_NORMAL:
        return code, num("mean", 0.0), num("std", 1.0)
    if code == runtime.DIST_EXPONENTIAL:
        return code, num("lambda", 1.0), 0.0
    return code, 0.0, 0.0


_NORMAL_PDF_CODE = Distribution.normal(0.0, 1.0)._pdf_func.__code__


def _is_factory_normal(dist: Distribution, mean: float, std: float) -> bool:
    """True when dist._pdf_func is the closure Distribution.normal builds for exactly (mean, std) -- i.e. the density
    exp(-((x - mean)/std)^2 / 2) / (std sqrt(2 pi)) of the distribution the sampler draws from."""
    fn = dist._pdf_func
    if getattr(fn, "__code__", None) is not _NORMAL_PDF_CODE or not fn.__closure__:
        return False
    try:
        cells = dict(zip(fn.__code__.co_freevars, (c.cell_contents for c in fn.__closure__)))
        return float(cells["mean"]) == mean and float(cells["sigma"]) == std and std > 0.0
    except (KeyError, TypeError, ValueError):
        return False


def _default_device() -> int:
    """HIP device of an integrator built without `device=`: entry LOCAL_RANK of MCX_DEVICES (a comma-separated list
    of device indices, e.g. "4,5,6,7" for a job that owns the upper half of a node) when that is set, else LOCAL_RANK
    itself on a multi-GPU box, else 0."""
    local = int(os.environ.get("LOCAL_RANK", "0"))
    listed = [d for d in os.environ.get("MCX_DEVICES", "").replace(" ", "").split(",") if d]
    if listed:
        try:
            return int(listed[local % len(listed)])
        except ValueError:
            raise ValueError(f"MCX_DEVICES must be a comma-separated list of device indices, got {os.environ['MCX_DEVICES']!r}")
    return local if runtime.device_count() > 1 else 0


class _Plan:
    """One compiled call: the module, its resident tables and the scalar parameters -- everything except the sizes and
    the seed, which are launch-time arguments. integrate* build a plan and launch it once; prepare_* hand it out."""

    __slots__ = ("kind", "module", "desc", "k", "rows", "p1", "p2", "tables", "x0", "target_accept", "proposal_kind", "walk",
                 "replicas")

    def __init__(self, kind, module, desc, k, rows, p1, p2, tables, x0=0.0, target_accept=0.44, proposal_kind="independent",
                 walk=0):
        self.kind, self.module, self.desc, self.k, self.rows = kind, module, desc, k, rows
        self.p1, self.p2, self.tables = p1, p2, tables
        self.x0, self.target_accept, self.proposal_kind, self.walk = x0, target_accept, proposal_kind, walk
        self.replicas = None          # per-device copies of (engine, module, tables), MonteCarloIntegrator(devices=[...])


class MonteCarloIntegrator:
    """Fused multi-function Monte-Carlo integrator on one MI355X (or one rank of N).

    Args:
        target_threads: logical thread count T of the reference's sample grid (default 65536). It
            fixes the sample indexing (T, L = ceil(n/T)), not the physical launch geometry.
        device: HIP device index (default: entry LOCAL_RANK of MCX_DEVICES if that is set, else LOCAL_RANK, else 0).
        process_group: ranks to shard every call over. None (default): this GPU only, like the reference (set
            MCX_DISTRIBUTED=1 to make None mean "world"); "world": the world group of the initialised
            torch.distributed job; or a torch.distributed process group. Sharding is opt-in because every sharded
            call is a collective: all ranks of the group must make it.
        math: "default" (the hardware exp / log / sqrt / rcp, range-reduced hardware sin / cos / tan, pow as exp2(y log2|x|),
            sinh / cosh on the hardware exp: all within the accuracy WGSL itself promises for these builtins, emit_hip.py),
            "fast" (sin / cos / tan as the bare instructions, |x| < ~1600) or "precise" (ocml + IEEE division everywhere).
        strict_reference_uniform: reproduce u = float(hash)*2^-32 on the closed interval [0,1]
            (reference behaviour, can produce log(0)); default False guards the end points.
        rng: "pcg_ref" (default) is the reference's counter hash -- bit-exact sample indexing, but a 32-bit
            counter space that is oversubscribed beyond ~4e9 uniforms per call; "philox" is Philox4x32-10 with a
            128-bit counter (four iterations per call for integrate / importance sampling, one call per two MH steps);
            "auto" uses the reference's stream for every call that stays within its 2^32 counter space -- where it is
            the parity stream and its estimates converge -- and Philox for the calls that draw more (BASELINE's
            1 048 576-chain MCMC: 2.3e10 uniforms; 1e10-sample integrals), where the reference's own estimates sit
            ~2 sigma off on average (profiles/r03_seed_sweep_c2_c5.jsonl). meta["rng"] says which one a call used.
        devices: several HIP device indices driven by THIS process (no torch.distributed): every call is sharded over
            them from one host thread and joined by libmcx's own RCCL communicator (include/mcx.h: mcx_comm_create --
            one grouped ncclAllReduce of the K doubles over xGMI), or by a host-side sum when a device is listed more
            than once or RCCL cannot be loaded. An alternative to one process per GPU; not combinable with
            process_group.
        std_error: also accumulate sum (f_k w)^2 in the same pass; integrate / importance-sampling results then
            carry result.meta["std_error"][k] = sqrt((E[(f w)^2] - E[f w]^2) / N_eff) (extension; K <= 32).
            integrate_mcmc results carry batch-means standard errors over the independent chains plus
            meta["tau_int"] and meta["ess"] (K <= 16).
    """

    @classmethod
    def planner(cls, **kw) -> "MonteCarloIntegrator":
        """An integrator that compiles but cannot launch (no GPU needed): prepare_integrate / prepare_importance_sampling /
        prepare_mcmc analyse the tables and hiprtc-compile the module into the code-object cache exactly as a call on
        a GPU would, so a deployment can warm its cache in a GPU-less build step. Same keyword arguments as the
        constructor (device / devices / process_group excepted)."""
        return cls(_host_engine=True, **kw)

    def __init__(self, target_threads: Optional[int] = None, device: Optional[int] = None, process_group=None,
                 math: str = "default", strict_reference_uniform: bool = False, rng: str = "pcg_ref",
                 std_error: bool = False, devices: Optional[Sequence[int]] = None, _host_engine: bool = False):
        runtime.load()                               # ImportError if libmcx.so has not been built
        if math not in ("default", "fast", "precise"):
            raise ValueError("math must be 'default', 'fast' or 'precise'")
        if rng not in runtime.RNG_CODES and rng != "auto":
            raise ValueError("rng must be 'pcg_ref' (the reference's stream), 'philox' or 'auto'")
        self._rng_auto = rng == "auto"
        self._rng = runtime.RNG_PCG_REF if self._rng_auto else runtime.RNG_CODES[rng]
        self._std_error = bool(std_error)
        if devices is not None:
            devices = [int(d) for d in devices]
            if not devices:
                raise ValueError("devices must list at least one device index")
            if process_group is not None:
                raise ValueError("devices (one process driving several GPUs) and process_group (one process per GPU) are alternatives")
            device = devices[0]
        if device is None and not _host_engine:
            device = _default_device()
        # RuntimeError("Failed to initialize GPU: ...") without a GPU. Engines are shared per device: building an
        # integrator per call (as the convenience functions do) costs no device initialisation after the first.
        self._engine = runtime.HostEngine() if _host_engine else runtime.Engine.shared(device)
        self._engines = [self._engine]
        self._comm = None
        if devices is not None and len(devices) > 1:
            # a device listed twice gets engines of its own (RCCL then refuses: host-side sum)
            seen, self._engines = set(), []
            for d in devices:
                self._engines.append(runtime.Engine.shared(d) if d not in seen else runtime.Engine(d))
                seen.add(d)
            if len(seen) == len(devices):
                try:
                    self._comm = runtime.Comm(self._engines)
                except (RuntimeError, ValueError):
                    self._comm = None
        self._integrator = self._engine             # attribute name the reference uses for its native object
        self._target_threads = target_threads
        self._math = math
        self._precise_sampler = math == "precise"
        self._guard = not strict_reference_uniform
        self._group = distributed.resolve_group(process_group)
        # what a plan depends on besides the functions and distributions (plans are cached per engine, shared by integrators)
        self._mode = (math, self._guard, self._rng, self._std_error)

    def _pick_rng(self, draws: int) -> int:
        """The stream a call that draws `draws` uniforms uses: the integrator's own, or with rng="auto" Philox beyond the
        2^32 counter space of the reference's hash."""
        return runtime.RNG_PHILOX if (self._rng_auto and draws > (1 << 32)) else self._rng

    # ---- plan cache -------------------------------------------------------------------------------
    def _cached_plan(self, kind: str, functions, dists, extra, build, rng: Optional[int] = None):
        """The compiled plan of a repeat call: one dict lookup on (function fingerprints, distribution identities, mode)
        instead of lowering, emission, desc fitting and module lookup (the reference re-transpiles and re-compiles per
        call, src/engine.rs:325-331). Anything unhashable or outside the emitter's subset takes the uncached path, which
        raises what it always raised."""
        cache = getattr(self._engine, "_plans", None)
        if cache is None:
            return build()
        try:
            key = (kind, tuple([_function_key(f) for f in functions]), tuple([_distribution_key(d) for d in dists]),
                   self._mode, self._rng if rng is None else rng, extra, _env_key())
            hit = cache.get(key)
        except (TypeError, TranspilerError, AttributeError):
            return build()
        if hit is not None:
            return hit[0]
        plan = build()
        with _PLAN_CACHE_LOCK:                       # host threads may share an engine: lookups are lock-free, updates are not
            while len(cache) >= _PLAN_CACHE_ENTRIES:
                cache.pop(next(iter(cache)))         # oldest entry (dicts keep insertion order)
            cache[key] = (plan, tuple(functions), tuple(dists))  # the strong references keep the id()s in the key valid
        return plan

    # ---- helpers ---------------------------------------------------------------------------------
    def _table(self, kind: int, keys: np.ndarray, values: np.ndarray) -> runtime.Table:
        return self._engine.cached_table(kind, keys, values)

    def _cdf_table(self, dist: Distribution) -> Optional[runtime.Table]:
        if dist.dist_type != DistributionType.CUSTOM:
            return None
        if dist._x_table is None or dist._cdf_table is None:
            raise ValueError("custom distribution has no CDF table")
        return self._table(runtime.TABLE_CDF, dist._cdf_table, dist._x_table)

    def _build_module(self, user_src: str, desc, *tables: Optional[runtime.Table], extra_bytes: int = 0):
        return build_module(self._engine, user_src, desc, *tables, extra_bytes=extra_bytes)

    def _replicas(self, plan: "_Plan"):
        """[(engine, module, tables)] of a plan on every device of this integrator (built on first use). Plans are cached per
        engine and shared by integrators, which may drive different device sets: the replicas are kept per engine set."""
        key = tuple(id(e) for e in self._engines)
        if plan.replicas is None:
            plan.replicas = {}
        hit = plan.replicas.get(key)
        # the entry keeps the engine objects alive, so an id() cannot be re-used by another engine; an engine that was
        # closed since (its modules and tables are gone) invalidates the entry
        if hit is not None and all(a is b and b._h for a, b in zip(hit[0], self._engines)):
            return hit[1]
        reps = [(self._engine, plan.module, plan.tables)]
        for eng in self._engines[1:]:
            tabs = {k: (eng.cached_table(t.kind, t.keys, t.values) if t is not None else None) for k, t in plan.tables.items()}
            reps.append((eng, eng.module(plan.module.user_src, plan.desc), tabs))
        for stale in [k for k, (engines, _) in plan.replicas.items() if not all(e._h for e in engines)]:
            del plan.replicas[stale]
        plan.replicas[key] = (list(self._engines), reps)
        return reps

    def _run_devices(self, plan: "_Plan", sizes, seed: int):
        """One host thread, several devices: device r runs shard r of len(devices); libmcx joins them."""
        shards = self._replicas(plan)
        if plan.kind == "mcmc":
            n_steps, n_chains, n_burnin = sizes
            fn = self._comm.mcmc if self._comm is not None else runtime.mcmc_multi
            return fn(shards, n_steps, n_chains, n_burnin, seed, plan.p1, plan.p2, target_threads=self._target_threads,
                      x0=plan.x0, target_accept=plan.target_accept)
        fn = self._comm.integrate if self._comm is not None else runtime.integrate_multi
        return fn(shards, sizes, seed, plan.p1, plan.p2, target_threads=self._target_threads)

    def _run(self, rows: int, call, plan: "_Plan" = None, sizes=None, seed: int = 0):
        """Run one sharded launch and combine the ranks with ONE sum all-reduce of `rows` doubles.

        call(d_sums, stream) -> (host sums or None, n_eff). With an RCCL group the partial sums never
        leave the GPU before the collective: the kernels write them to a torch CUDA buffer on torch's
        current stream and the all-reduce runs on that buffer over xGMI."""
        if len(self._engines) > 1 and plan is not None:
            sums, n_eff = self._run_devices(plan, sizes, seed)
            with np.errstate(divide="ignore", invalid="ignore"):
                return sums / float(n_eff), n_eff
        g = self._group
        if g is None or g.world < 2:
            sums, n_eff = call(None, None)
            if n_eff:
                return sums / float(n_eff), n_eff
            with np.errstate(divide="ignore", invalid="ignore"):      # n_samples = 0 -> 0/0 = NaN, like the reference
                return sums / float(n_eff), n_eff
        if g.backend == "nccl":
            import torch

            dev = torch.device("cuda", self._engine.device)
            buf = torch.empty(rows, dtype=torch.float64, device=dev)
            _, n_eff = call(buf.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
            distributed.all_reduce_device(g, buf)
            sums = buf.cpu().numpy()
        else:                                   # gloo: host tensors (CPU rehearsal of the N > 1 path)
            sums, n_eff = call(None, None)
            sums = distributed.all_reduce_host(g, sums)
        return sums / float(n_eff), n_eff

    def _warn_if_oversubscribed(self, n_eff: int, what: str = "n_samples", rng: Optional[int] = None) -> None:
        """The reference's counter hash has 2^32 distinct inputs. Beyond that many samples per call the estimator
        stops converging: the stream is a finite population whose own mean is off by ~3e-5 (measured: E[x] on N(0,1)
        is 10 sigma low at n = 1e11 and 33 sigma low at 1e12, tools/stream_quality.py / DESIGN.md 4.4)."""
        if (self._rng if rng is None else rng) == runtime.RNG_PCG_REF and n_eff > (1 << 32) and not getattr(self, "_warned_stream", False):
            import warnings

            self._warned_stream = True
            warnings.warn(
                f"{what} = {n_eff:.3g} exceeds the 2^32 counter space of the reference's random stream: accuracy "
                f"saturates at about 3e-5 absolute. Pass rng='philox' to MonteCarloIntegrator for a 128-bit counter stream.",
                UserWarning, stacklevel=3)

    def _use_moment_family(self, functions) -> bool:
        """Pairwise power-sum accumulation for x, x**2, ..., x**K (K >= 8); math="precise" and std_error evaluate
        every function per sample."""
        return not self._std_error and self._math != "precise" and _moment_family(functions)

    def _rank_world(self):
        return (self._group.rank, self._group.world) if self._group is not None else (0, 1)

    # ---- plans: emission + compilation + resident tables, no launch -----------------------------------
    def _plan_integrate(self, functions, distribution, rng: Optional[int] = None) -> _Plan:
        rng = self._rng if rng is None else rng
        if len(functions) == 0:
            raise ValueError("At least one function is required")
        user_src = functions_to_hip(functions, self._math)
        code, p1, p2 = _dist_params(distribution)
        cdf = self._cdf_table(distribution)
        k = len(functions)
        desc = runtime.make_desc(runtime.KIND_INTEGRATE, k, code, guard_endpoints=self._guard,
                                 precise_sampler=self._precise_sampler, rng=rng,
                                 second_moments=self._std_error, moment_family=self._use_moment_family(functions))
        runtime.module_desc_fit(desc, cdf, None, None, p1, p2)
        return _Plan("integrate", self._build_module(user_src, desc, cdf), desc, k, runtime.result_rows(desc), p1, p2,
                     dict(cdf=cdf))

    def _plan_importance_sampling(self, functions, target_distribution, proposal_distribution, rng: Optional[int] = None) -> _Plan:
        rng = self._rng if rng is None else rng
        if len(functions) == 0:
            raise ValueError("At least one function is required")
        p_src = _pdf_to_hip(target_distribution, "mcx_pdf_p", self._math)
        q_src = _pdf_to_hip(proposal_distribution, "mcx_pdf_q", self._math)
        user_src = functions_to_hip(functions, self._math)
        code, p1, p2 = _dist_params(proposal_distribution)
        cdf = self._cdf_table(proposal_distribution)
        p_table = q_table = None
        # the samples are drawn from the proposal: for Distribution.normal its density is a function of the deviate the
        # sampler already holds, so the kernel forms 1/q from z instead of evaluating the emitted closure at x
        q_sampler = (q_src is not None and not self._precise_sampler and code == runtime.DIST_NORMAL
                     and _is_factory_normal(proposal_distribution, p1, p2))
        if p_src is None:
            xs, dens = target_distribution.get_or_compute_pdf_table()
            p_table = self._table(runtime.TABLE_PDF, xs, dens)
        else:
            user_src += "\n\n" + p_src
        if q_src is None:
            xs, dens = proposal_distribution.get_or_compute_pdf_table()
            q_table = self._table(runtime.TABLE_PDF, xs, dens)
        elif not q_sampler:
            user_src += "\n\n" + q_src
        k = len(functions)
        desc = runtime.make_desc(runtime.KIND_INTEGRATE, k, code, weight=True,
                                 p_table=p_table is not None, q_table=q_table is not None,
                                 guard_endpoints=self._guard, precise_sampler=self._precise_sampler,
                                 rng=rng, second_moments=self._std_error, q_sampler=q_sampler,
                                 moment_family=self._use_moment_family(functions))
        pad_bytes = runtime.module_desc_fit(desc, cdf, p_table, q_table, p1, p2)       # cell form, pads, LDS staging, block: libmcx decides
        return _Plan("integrate", self._build_module(user_src, desc, cdf, p_table, q_table, extra_bytes=pad_bytes), desc, k,
                     runtime.result_rows(desc), p1, p2, dict(cdf=cdf, target_pdf=p_table, proposal_pdf=q_table))

    def _mcmc_block(self, n_chains: int, parts: Optional[int] = None) -> int:
        """desc.block for an MCMC call of n_chains chains: 0 (libmcx's default, 1024 threads when tables are staged)
        unless this rank's / device's share of the padded chains is too small to give every CU several workgroups of
        that size (mcx_mcmc_block_hint: 131 072 chains -- C4 over 8 GPUs -- run 1.5x faster with 256 threads)."""
        padded = runtime.mcmc_dispatch_config(n_chains, self._target_threads).total_threads
        if parts is None:
            parts = len(self._engines) if len(self._engines) > 1 else self._rank_world()[1]
        hint = runtime.mcmc_block_hint(-(-padded // max(parts, 1)))
        return 0 if hint >= 1024 else hint

    def _plan_mcmc(self, functions, target_distribution, proposal_distribution, proposal_kind="independent",
                   initial_state=0.0, target_accept=0.44, block: int = 0, rng: Optional[int] = None) -> _Plan:
        rng = self._rng if rng is None else rng
        if proposal_kind not in ("independent", "random_walk", "adaptive_random_walk"):
            raise ValueError(f"Unknown proposal_kind: {proposal_kind!r} (expected 'independent', 'random_walk' or "
                             f"'adaptive_random_walk')")
        if proposal_kind == "adaptive_random_walk" and not 0.0 < float(target_accept) < 1.0:
            raise ValueError("target_accept must lie strictly between 0 and 1")
        if len(functions) == 0:
            raise ValueError("At least one function is required")
        user_src = functions_to_hip(functions, self._math)
        code, p1, p2 = _dist_params(proposal_distribution)
        # Distribution.normal proposals: log q(x) = -z^2/2 + const for the deviate z the sampler holds, so no proposal
        # table is interpolated (the reference's 2048-point table of the same function is up to 6e-6 below it)
        q_sampler = (code == runtime.DIST_NORMAL and not self._precise_sampler
                     and _is_factory_normal(proposal_distribution, p1, p2))
        tx, tlog = target_distribution.get_log_pdf_table()
        t_table = self._table(runtime.TABLE_LOGPDF, tx, tlog)
        q_table = None
        if not q_sampler:
            px, plog = proposal_distribution.get_log_pdf_table()
            q_table = self._table(runtime.TABLE_LOGPDF, px, plog)
        cdf = self._cdf_table(proposal_distribution)
        walk = runtime.WALK_INDEPENDENT
        if proposal_kind != "independent":
            symmetric = (code == runtime.DIST_NORMAL and p1 == 0.0) or (code == runtime.DIST_UNIFORM and p1 == -p2)
            walk = runtime.WALK_RANDOM_SYMMETRIC if symmetric else runtime.WALK_RANDOM
            if proposal_kind == "adaptive_random_walk":
                if not symmetric:
                    raise ValueError("adaptive_random_walk needs increments symmetric about 0: normal(0, s) or uniform(-w, w)")
                walk = runtime.WALK_ADAPTIVE
        k = len(functions)
        desc = runtime.make_desc(runtime.KIND_MCMC, k, code, guard_endpoints=self._guard,
                                 precise_sampler=self._precise_sampler, rng=rng, block=block,
                                 second_moments=self._std_error, walk=walk, q_sampler=q_sampler)
        # cell form, pads (independent proposals: every lookup is at a draw of the proposal, whose range is known), 16-bit LDS
        # addresses, LDS staging, workgroup size: libmcx decides (mcx_module_desc_fit)
        pad_bytes = runtime.module_desc_fit(desc, cdf, t_table, q_table, p1, p2)
        return _Plan("mcmc", self._build_module(user_src, desc, cdf, t_table, q_table, extra_bytes=pad_bytes), desc, k,
                     runtime.result_rows(desc), p1, p2,
                     dict(cdf=cdf, target_logpdf=t_table, proposal_logpdf=q_table), x0=float(initial_state),
                     target_accept=float(target_accept), proposal_kind=proposal_kind, walk=walk)

    def _enqueue(self, plan: _Plan, sizes, seed: int, d_sums, stream, rank_world=None):
        """Launch this rank's shard of `plan`: (host sums or None when d_sums is given, n_eff of the WHOLE job)."""
        rank, world = rank_world if rank_world is not None else self._rank_world()
        tb = plan.tables
        if plan.kind == "mcmc":
            n_steps, n_chains, n_burnin = sizes
            return self._engine.mcmc(plan.module, n_steps, n_chains, n_burnin, seed, plan.p1, plan.p2,
                                     tb["target_logpdf"], tb["proposal_logpdf"], target_threads=self._target_threads,
                                     cdf=tb["cdf"], rank=rank, world=world, d_sums=d_sums, stream=stream,
                                     x0=plan.x0, target_accept=plan.target_accept)
        return self._engine.integrate(plan.module, sizes, seed, plan.p1, plan.p2, self._target_threads, cdf=tb.get("cdf"),
                                      target_pdf=tb.get("target_pdf"), proposal_pdf=tb.get("proposal_pdf"),
                                      rank=rank, world=world, d_sums=d_sums, stream=stream)

    # ---- K1 ----------------------------------------------------------------------------------------
    def integrate(self, functions: List[FunctionLike], distribution: Distribution, n_samples: int = 1_000_000,
                  seed: int = 42) -> IntegrationResult:
        """E[f_k(X)], X ~ distribution, for all functions on the same samples."""
        n_samples = _check_count(n_samples, "n_samples")
        seed = _check_seed(seed)
        rng = self._pick_rng(n_samples)
        plan = self._cached_plan("integrate", functions, (distribution,), None,
                                 lambda: self._plan_integrate(functions, distribution, rng), rng)
        values, n_eff = self._run(plan.rows, lambda d_sums, stream: self._enqueue(plan, n_samples, seed, d_sums, stream),
                                  plan, n_samples, seed)
        self._warn_if_oversubscribed(n_eff, rng=rng)
        return IntegrationResult(values[:plan.k], n_samples, plan.k, self._meta(n_eff, values, plan.k, rng))

    # ---- K2 ----------------------------------------------------------------------------------------
    def integrate_importance_sampling(self, functions: List[FunctionLike], target_distribution: Distribution,
                                      proposal_distribution: Distribution, n_samples: int = 1_000_000,
                                      seed: int = 42) -> IntegrationResult:
        """E_p[f_k(X)] ~= mean f_k(x) p(x)/q(x), x ~ q."""
        n_samples = _check_count(n_samples, "n_samples")
        seed = _check_seed(seed)
        rng = self._pick_rng(n_samples)
        plan = self._cached_plan("is", functions, (target_distribution, proposal_distribution), None,
                                 lambda: self._plan_importance_sampling(functions, target_distribution, proposal_distribution, rng), rng)
        values, n_eff = self._run(plan.rows, lambda d_sums, stream: self._enqueue(plan, n_samples, seed, d_sums, stream),
                                  plan, n_samples, seed)
        return IntegrationResult(values[:plan.k], n_samples, plan.k, self._meta(n_eff, values, plan.k, rng))

    # ---- K3 ----------------------------------------------------------------------------------------
    def integrate_mcmc(self, functions: List[FunctionLike], target_distribution: Distribution,
                       proposal_distribution: Distribution, n_steps: int = 10_000, n_chains: int = 1024,
                       n_burnin: int = 1_000, seed: int = 42, proposal_kind: str = "independent",
                       initial_state: float = 0.0, target_accept: float = 0.44) -> IntegrationResult:
        """E_p[f_k(X)] by Metropolis-Hastings, one chain per logical thread.

        proposal_kind="independent" (default) is the reference's sampler: x' ~ proposal_distribution
        (shader_gen.rs:466-539). proposal_kind="random_walk" (extension; the reference leaves it open at
        shader_gen.rs:514) draws the *increment* from proposal_distribution: x' = x + d, chains start at
        initial_state + d_0; the Hastings correction log q(-d) - log q(d) is applied unless the increment
        distribution is symmetric about 0 (normal(0, s), uniform(-w, w)). proposal_kind="adaptive_random_walk"
        (symmetric increments only) additionally tunes a per-chain step scale during burn-in towards `target_accept`
        (x' = x + s d; log s += t^-1/2 (accepted - target_accept) after burn-in step t; frozen while sampling) and
        reports the mean final scale as meta["step_scale"]. With std_error=True the result carries
        meta["std_error"] (batch means over chains), meta["ess"] and meta["tau_int"] per function (K <= 16)."""
        if proposal_kind not in ("independent", "random_walk", "adaptive_random_walk"):
            raise ValueError(f"Unknown proposal_kind: {proposal_kind!r} (expected 'independent', 'random_walk' or "
                             f"'adaptive_random_walk')")
        if proposal_kind == "adaptive_random_walk" and not 0.0 < float(target_accept) < 1.0:
            raise ValueError("target_accept must lie strictly between 0 and 1")
        if len(functions) == 0:
            raise ValueError("At least one function is required")
        n_steps, n_chains, n_burnin = self._check_mcmc_sizes(n_steps, n_chains, n_burnin)
        seed = _check_seed(seed)
        block = self._mcmc_block(n_chains)
        padded = runtime.mcmc_dispatch_config(n_chains, self._target_threads).total_threads
        rng = self._pick_rng(2 * padded * (n_steps + n_burnin))          # one proposal + one accept uniform per step
        plan = self._cached_plan("mcmc", functions, (target_distribution, proposal_distribution),
                                 (proposal_kind, float(initial_state), float(target_accept), block),
                                 lambda: self._plan_mcmc(functions, target_distribution, proposal_distribution, proposal_kind,
                                                         initial_state, target_accept, block=block, rng=rng), rng)
        values, n_eff = self._run(plan.rows, lambda d_sums, stream: self._enqueue(
            plan, (n_steps, n_chains, n_burnin), seed, d_sums, stream), plan, (n_steps, n_chains, n_burnin), seed)
        return self._mcmc_result(plan, values, n_eff, n_steps, n_chains, n_burnin, rng)

    @staticmethod
    def _check_mcmc_sizes(n_steps, n_chains, n_burnin):
        if n_steps <= 0:
            raise ValueError("n_steps must be positive")
        if n_chains <= 0:
            raise ValueError("n_chains must be positive")
        if n_burnin < 0:
            raise ValueError("n_burnin must be non-negative")
        return (_check_count(n_steps, "n_steps", 32), _check_count(n_chains, "n_chains", 32),
                _check_count(n_burnin, "n_burnin", 32))

    def _mcmc_result(self, plan: _Plan, values, n_eff: int, n_steps: int, n_chains: int, n_burnin: int,
                     rng: Optional[int] = None) -> IntegrationResult:
        """values = all-rank sums / n_eff for every result row of an MCMC plan -> the public result + diagnostics."""
        k = plan.k
        meta = self._meta(n_eff, rng=rng)
        total_chains = n_eff // n_steps
        # chains whose counters collide replay each other's random numbers with a time shift: the estimate stays
        # consistent but the chains are no longer independent (measured at C4's size with random-walk proposals:
        # 5-7 batch-means standard errors off with the reference stream, 0.4 with Philox)
        self._warn_if_oversubscribed(2 * total_chains * (n_steps + n_burnin), "uniform draws (chains x steps x 2)", rng=rng)
        row_accept = 2 * k if self._std_error else k
        meta["accept_rate"] = float(values[row_accept]) * n_eff / (float(total_chains) * (n_steps + n_burnin))
        meta["proposal_kind"] = plan.proposal_kind
        if plan.walk == runtime.WALK_ADAPTIVE:
            meta["step_scale"] = float(values[-1]) * n_eff / float(total_chains)      # mean final scale over the chains
        if self._std_error:
            # batch means with one batch per chain: chains are independent, so the spread of their means measures
            # the Monte-Carlo error whatever the autocorrelation inside a chain is
            mean = values[:k]
            with np.errstate(invalid="ignore", divide="ignore"):
                var_f = np.maximum(values[k:2 * k] - mean ** 2, 0.0)
                var_between = np.maximum(values[2 * k + 1:3 * k + 1] * (n_eff / float(total_chains)) - mean ** 2, 0.0)
                meta["std_error"] = np.sqrt(var_between / float(max(total_chains - 1, 1)))
                meta["tau_int"] = n_steps * var_between / var_f
                meta["ess"] = float(n_eff) / meta["tau_int"]
        return IntegrationResult(values[:k], n_chains * n_steps, k, meta)

    # ---- prepared (asynchronous) forms ---------------------------------------------------------------
    def prepare_integrate(self, functions: List[FunctionLike], distribution: Distribution) -> "PreparedIntegrand":
        """Emit + compile once; the returned object launches without host synchronisation.

        Extension over the reference API for serving / benchmarking loops: every `integrate()` call of the
        reference re-transpiles and re-compiles (src/engine.rs:325-331)."""
        return PreparedIntegrand(self, self._plan_integrate(functions, distribution))

    def prepare_importance_sampling(self, functions: List[FunctionLike], target_distribution: Distribution,
                                    proposal_distribution: Distribution) -> "PreparedIntegrand":
        """prepare_integrate for integrate_importance_sampling (same launch interface)."""
        return PreparedIntegrand(self, self._plan_importance_sampling(functions, target_distribution, proposal_distribution))

    def prepare_mcmc(self, functions: List[FunctionLike], target_distribution: Distribution,
                     proposal_distribution: Distribution, proposal_kind: str = "independent", initial_state: float = 0.0,
                     target_accept: float = 0.44) -> "PreparedMcmc":
        """prepare_integrate for integrate_mcmc: launch(n_steps, n_chains, n_burnin, seed, out)."""
        make = lambda block: self._plan_mcmc(functions, target_distribution, proposal_distribution, proposal_kind,
                                             initial_state, target_accept, block=block)
        return PreparedMcmc(self, make(0), make)

    def _meta(self, n_eff: int, values=None, k: int = 0, rng: Optional[int] = None) -> dict:
        meta = self._engine.last_call()               # n_blocks, block, lds_bytes, launches, kernel_ms: one C call
        rank, world = self._rank_world()
        meta["n_eff"], meta["rank"], meta["world"] = n_eff, rank, world
        meta["rng"] = "philox" if (self._rng if rng is None else rng) == runtime.RNG_PHILOX else "pcg_ref"
        meta["devices"] = [e.device for e in self._engines]
        meta["collective"] = ("rccl" if self._comm is not None else "host-sum") if len(self._engines) > 1 else None
        if values is not None and len(values) == 2 * k and k:
            with np.errstate(invalid="ignore", divide="ignore"):
                var = np.maximum(values[k:] - values[:k] ** 2, 0.0)
                meta["std_error"] = np.sqrt(var / float(n_eff))
        return meta


class PreparedIntegrand:
    """A compiled fused integrand (integrate / importance sampling) bound to one engine: see
    MonteCarloIntegrator.prepare_integrate / prepare_importance_sampling."""

    def __init__(self, owner: MonteCarloIntegrator, plan: _Plan):
        self._owner, self._plan = owner, plan
        self.k, self.rows = plan.k, plan.rows

    def _launch(self, sizes, seed: int, out, async_op: bool, reduce: bool, shard=None):
        import torch

        owner = self._owner
        if len(owner._engines) > 1:
            raise RuntimeError("launch() enqueues on one device's stream; an integrator built with devices=[...] is "
                               "driven through the blocking calls (integrate*, run)")
        if out.dtype != torch.float64 or out.numel() < self.rows or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous float64 CUDA tensor with at least {self.rows} elements")
        stream = torch.cuda.current_stream(out.device).cuda_stream
        _, n_eff = owner._enqueue(self._plan, sizes, seed, out.data_ptr(), stream, rank_world=shard)
        work = None
        if reduce and shard is None and owner._rank_world()[1] > 1:
            work = distributed.all_reduce_device(owner._group, out, async_op=async_op)
        return (n_eff, work) if async_op else n_eff

    def launch(self, n_samples: int, seed: int, out, async_op: bool = False, reduce: bool = True, shard=None):
        """Enqueue sampling + reduction (+ one sum all-reduce when sharded) on torch's current stream.

        shard=(rank, world) overrides the integrator's own position in its process group for this launch and implies
        reduce=False: (0, 1) runs the WHOLE grid on this GPU -- what bench.py's sharded-equals-single check compares
        the all-reduced shards with.

        `out` is a float64 CUDA tensor with `rows` (= k; 2k with std_error) elements that receives the SUMS over the
        whole job (all ranks); divide by the returned n_eff for the expected values. No host synchronisation.
        With async_op=True the collective does not block the current stream (the next launch overlaps it);
        returns (n_eff, work) and the caller waits on `work` (None on a single GPU) before reading `out`.
        reduce=False leaves this rank's partial sums in `out` (no collective)."""
        return self._launch(int(n_samples), seed, out, async_op, reduce, shard)

    def run(self, n_samples: int, seed: int = 42) -> IntegrationResult:
        """Blocking form: same result as MonteCarloIntegrator.integrate() / integrate_importance_sampling()."""
        owner, plan = self._owner, self._plan
        values, n_eff = owner._run(plan.rows, lambda d_sums, stream: owner._enqueue(plan, int(n_samples), seed, d_sums, stream),
                                   plan, int(n_samples), seed)
        return IntegrationResult(values[:plan.k], n_samples, plan.k, owner._meta(n_eff, values, plan.k))


class PreparedMcmc(PreparedIntegrand):
    """A compiled Metropolis-Hastings call (MonteCarloIntegrator.prepare_mcmc). `rows` = k + 1 (row k: accepted steps;
    3k + 1 with std_error, one more for the adaptive walk). The workgroup size is a compile-time property of the module
    and the right one depends on how many chains a launch carries (MonteCarloIntegrator._mcmc_block), which is only
    known at launch: variants are built on first use and kept."""

    def __init__(self, owner: MonteCarloIntegrator, plan: _Plan, make_plan):
        super().__init__(owner, plan)
        self._make_plan = make_plan
        self._variants = {0: plan}

    def _select(self, n_chains: int, shard=None) -> None:
        block = self._owner._mcmc_block(n_chains, parts=shard[1] if shard is not None else None)
        if block not in self._variants:
            self._variants[block] = self._make_plan(block)
        self._plan = self._variants[block]

    def launch(self, n_steps: int, n_chains: int, n_burnin: int, seed: int, out, async_op: bool = False,
               reduce: bool = True, shard=None):
        """Enqueue this rank's chains (+ one sum all-reduce when sharded); `out` receives `rows` sums, n_eff =
        padded chains x n_steps is returned. See PreparedIntegrand.launch."""
        sizes = MonteCarloIntegrator._check_mcmc_sizes(n_steps, n_chains, n_burnin)
        self._select(sizes[1], shard)
        return self._launch(sizes, seed, out, async_op, reduce, shard)

    def run(self, n_steps: int = 10_000, n_chains: int = 1024, n_burnin: int = 1_000, seed: int = 42) -> IntegrationResult:
        """Blocking form: same result as MonteCarloIntegrator.integrate_mcmc()."""
        sizes = MonteCarloIntegrator._check_mcmc_sizes(n_steps, n_chains, n_burnin)
        self._select(sizes[1])
        owner, plan = self._owner, self._plan
        values, n_eff = owner._run(plan.rows, lambda d_sums, stream: owner._enqueue(plan, sizes, seed, d_sums, stream), plan, sizes, seed)
        return owner._mcmc_result(plan, values, n_eff, *sizes)


def integrate(functions: List[FunctionLike], distribution: Distribution, n_samples: int = 1_000_000, seed: int = 42,
              target_threads: Optional[int] = None) -> IntegrationResult:
    """Shorthand for MonteCarloIntegrator(target_threads).integrate(...)."""
    return MonteCarloIntegrator(target_threads=target_threads).integrate(functions, distribution, n_samples, seed)


def integrate_importance_sampling(functions: List[FunctionLike], target_distribution: Distribution,
                                  proposal_distribution: Distribution, n_samples: int = 1_000_000, seed: int = 42,
                                  target_threads: Optional[int] = None) -> IntegrationResult:
    """Shorthand for MonteCarloIntegrator(target_threads).integrate_importance_sampling(...)."""
    return MonteCarloIntegrator(target_threads=target_threads).integrate_importance_sampling(
        functions, target_distribution, proposal_distribution, n_samples, seed)


def integrate_mcmc(functions: List[FunctionLike], target_distribution: Distribution,
                   proposal_distribution: Distribution, n_steps: int = 10_000, n_chains: int = 1024,
                   n_burnin: int = 1_000, seed: int = 42, target_threads: Optional[int] = None) -> IntegrationResult:
    """Shorthand for MonteCarloIntegrator(target_threads).integrate_mcmc(...). As in the reference,
    target_threads (if given) overrides n_chains (src/engine.rs:860)."""
    return MonteCarloIntegrator(target_threads=target_threads).integrate_mcmc(
        functions, target_distribution, proposal_distribution, n_steps, n_chains, n_burnin, seed)
