#!/usr/bin/env python3
"""bench.py -- throughput of the fused Monte-Carlo hot path on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5] [--scaling weak|strong]

The default (and the headline, BASELINE.json `metric`) is --config c2: samples/sec of integrate([x, x**2, x**3,
x**4], Normal(0,1)) at 1e9 samples per GPU per step. One step = one pass of the hot path (counter RNG + sampler +
K evaluations + two-stage f64 reduction in one launch); for N > 1 every rank runs its shard of the SAME logical
sample grid / chain range and the partial sums are joined by one RCCL sum all-reduce. Inputs are four scalars
(+ tables resident in HBM): nothing is staged from the host inside the timed region. Rank 0 prints ONE JSON line.

Process model. `python bench.py --gpus N` launches itself: the parent parses the arguments, starts N FRESH child
processes (one rank per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their
environment) BEFORE it makes any GPU call -- it never imports torch or loads libmcx -- and relays rank 0's JSON
line. Started under `python -m torch.distributed.run` (WORLD_SIZE already set) it is one of the ranks. Nothing is
ever exec'ed from a process that has touched the GPU.

Other configs (BASELINE.json configs[2..4]): c3 importance sampling with a 512-point target table, c4
chain-sharded Metropolis-Hastings (1 048 576 chains x 11 000 steps), c5 K = 32 moments of Beta(2,5) at 1e10
samples. Default scaling: c2 / c3 weak (nominal size per GPU), c4 / c5 strong (BASELINE fixes the total). The default
command times ALL of them: c2 is the headline (`value`), c3 / c4 / c5 run through the same timed loop right after it
and are reported under `configs` in the same JSON line (--legs), each with its own kernel time, roofline, accuracy on
both streams and a short CPU-baseline slice. At N > 1 the line also carries `self_check`: an RCCL all-reduce of ones
(= N) and, per config, the all-reduced shard sums against the whole grid run on rank 0 (`sharded_equals_single`).
At N = 1 it also carries `core_binding` (the reference's own `_core` payloads for configs[1..3] replayed through this package's `_core`
binding) and `reference_protocol`: the reference's own benchmark (examples/benchmark.py: wall clock around blocking
integrate() calls of x / (exp(sin x) + cos(exp x)) on N(0,1)) at n = 1e3 / 1e5 / 1e7 / 1e9.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for _p in (ROOT / "wgpu-monte-carlo_amd", ROOT, ROOT / "tools"):
    if str(_p) not in sys.path:
        sys.path.insert(0, str(_p))

DEFAULT_SCALING = {"c1": "weak", "c2": "weak", "c3": "weak", "c4": "strong", "c5": "strong"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4", "c5"],
                    help="BASELINE.json workload (default c2 = the headline metric)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: the config's nominal size PER GPU per step; strong: in total (default per config)")
    ap.add_argument("--scale", type=float, default=1.0, help="multiply the nominal size (quick runs)")
    ap.add_argument("--samples-per-gpu", type=float, default=None, help="c2, weak scaling: samples per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-clock budget of the CPU baseline slice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold", action="store_true",
                    help="skip the cold-start probe (a child process; it is also skipped under rocprofv3, where "
                         "starting a child from a profiled process is not allowed on the GPU pool)")
    ap.add_argument("--no-philox", action="store_true", help="skip the second timed loop on the Philox stream")
    ap.add_argument("--prewarm-ms", type=float, default=80.0,
                    help="untimed device warm-up before the W warm-up steps: the same step is launched back to back for this "
                         "long so that the GPU's clock has left its idle state when the W + K steps run (measured: with "
                         "W = 3 straight from idle 0.452 ms per step, after 40 ms of work 0.395; 0 disables it)")
    ap.add_argument("--target-phys", type=int, default=0, help="physical threads per launch (tuning)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the independent steps are issued on in turn. 2 lets the next step's workgroups fill "
                         "the CUs the previous step's tail leaves idle (measured 0.420 -> 0.397 ms per step); the default "
                         "stays 1 so that a launch's duration in the rocprofv3 trace is the kernel alone, not two "
                         "launches sharing the chip")
    ap.add_argument("--mcmc-segments", type=int, default=None,
                    help="c4: run each MCMC call as two chain halves on two streams x this many step segments "
                         "(mcx_engine_set_mcmc_segments; measured 8.5 -> 8.0 ms at 8). Default: libmcx's own rule (8 segments for "
                         "launches of >= 1 048 576 chains); 0 = one launch per call")
    ap.add_argument("--legs", default="auto",
                    help="other BASELINE configs timed in the same run and reported under `configs` in the one JSON line: "
                         "auto (c3,c4,c5 when --config c2 at full scale and not under rocprofv3), none, or a comma list")
    ap.add_argument("--leg-steps", type=int, default=20, help="timed steps per leg (capped by --steps)")
    ap.add_argument("--leg-cpu-seconds", type=float, default=2.5, help="wall-clock budget of each leg's CPU baseline slice")
    ap.add_argument("--rng", default="pcg_ref", help="stream of the headline loop: pcg_ref (the reference's) or philox")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; the measured configuration) or gloo (rehearsal)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (with --backend gloo) to exercise the N > 1 code path")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="no GPU work at all: spawn, rendezvous (gloo), one host all-reduce, relay -- the launcher's own test")
    ap.add_argument("--cold-probe", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# launcher (no torch, no libmcx in this part)
# ---------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_profiler() -> bool:
    """rocprofv3 preloads its tool library, which initialises the GPU before Python starts: a child process started
    from here would be an exec from a GPU-initialised process."""
    env = os.environ
    return "rocprof" in env.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in env)


def cold_probe(args, popen=subprocess.run):
    """First-call latency with an empty code-object cache, measured in a fresh child process (hiprtc compile of the
    module + code-object load + first launch), BASELINE.md section 4 'cold'. Must run before this process touches the GPU."""
    if args.no_cold or args.rehearse_cpu or under_profiler():
        return None
    env = dict(os.environ, MCX_NO_DISK_CACHE="1", MCX_BENCH_CHILD="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        res = popen([sys.executable, str(ROOT / "bench.py"), "--cold-probe", "--config", args.config],
                    env=env, capture_output=True, text=True, timeout=600)
        for line in reversed(res.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": (res.stderr or res.stdout)[-400:]}
    except Exception as exc:        # the probe must never take the benchmark down
        return {"error": repr(exc)}


def spawn_ranks(args, argv, popen=subprocess.Popen):
    """Start one fresh child per rank and relay rank 0's output. Called before anything in this process has
    imported torch or loaded libmcx (asserted): the children initialise the GPU, the parent never does."""
    assert "torch" not in sys.modules and "wgpu_montecarlo" not in sys.modules, "the launcher must stay GPU-free"
    cold = cold_probe(args)
    port = _free_port()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(0 if args.single_device else rank),
                   WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MCX_BENCH_CHILD="1",
                   MCX_BENCH_COLD=json.dumps(cold))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between the ranks' processes on this driver
        procs.append(popen([sys.executable, str(ROOT / "bench.py")] + list(argv), env=env,
                           stdout=subprocess.PIPE if rank == 0 else None, text=True))
    # supervise: if any rank fails, stop the others (they would otherwise sit in the rendezvous / a collective)
    rc = 0
    while any(p.poll() is None for p in procs):
        failed = [p for p in procs if p.poll() not in (None, 0)]
        if failed:
            rc = failed[0].returncode
            for p in procs:
                if p.poll() is None:
                    p.kill()               # the exact children we started, never a pattern
            break
        time.sleep(0.2)
    out = procs[0].stdout.read() if procs[0].stdout is not None else ""
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out or "")
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (plain-C restatement of the reference kernels, OpenMP over the logical thread index) on
# a bounded slice of the same workload. Checker infrastructure, timed beside the GPU, never part of the product.
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(config: str, target_seconds: float):
    import numpy as np

    import baseline_configs as bc
    import oracle

    def timed(call):
        t0 = time.perf_counter()
        res = call()
        return res, time.perf_counter() - t0

    pows = lambda k: [(oracle.FN_IDENTITY, 0)] + [(oracle.FN_POW, j) for j in range(2, k + 1)]
    if config in ("c1", "c2"):
        k = 4 if config == "c2" else 2
        run = lambda n, seed: oracle.integrate(pows(k), oracle.NORMAL, 0.0, 1.0, n_samples=n, seed=seed, guard=1)
        pilot_n, cap, units = 50_000_000, 4e9, (lambda r: r["n_eff"])          # <= 2^32: inside the counter space
        what, truth = f"K={k} moments on N(0,1)", np.array([0.0, 1.0, 0.0, 3.0])[:k]
    elif config == "c3":
        xs = np.linspace(0, 10, 512).astype(np.float32)
        ps = np.exp(-np.linspace(0, 10, 512)).astype(np.float32)
        run = lambda n, seed: oracle.integrate(pows(4), oracle.NORMAL, 2.0, 3.0, n_samples=n, seed=seed, guard=1,
                                               p=(oracle.PDF_TABLE, xs, ps), q=(oracle.PDF_NORMAL, 2.0, 3.0))
        pilot_n, cap, units = 20_000_000, 4e9, (lambda r: r["n_eff"])
        what, truth = "K=4 importance sampling, 512-point target table, N(2,3) proposal", None
    elif config == "c4":
        from wgpu_montecarlo import Distribution     # host-side table builders only (numpy); no GPU is touched

        tx, tl = Distribution.from_pdf(bc.bimodal, support=(-10, 10)).get_log_pdf_table()
        px, pl = Distribution.normal(0.0, 2.0).get_log_pdf_table()
        run = lambda n, seed: oracle.mcmc(pows(2), oracle.NORMAL, 0.0, 2.0, tx, tl, px, pl, n_steps=10_000,
                                          n_chains=n, n_burnin=1000, seed=seed, guard=1)
        pilot_n, cap, units = 512, 1_048_576, (lambda r: (r["n_eff"] // 10_000) * 11_000)
        what, truth = "bimodal-target MH, N(0,2) proposals, 11 000 steps per chain", np.array([0.0, 5.0])
    else:
        from wgpu_montecarlo import Distribution

        beta = Distribution.beta(2.0, 5.0)
        run = lambda n, seed: oracle.integrate(pows(32), oracle.CUSTOM, 0.0, 0.0, n_samples=n, seed=seed, guard=1,
                                               cdf_table=beta._cdf_table, x_table=beta._x_table)
        pilot_n, cap, units = 10_000_000, 4e9, (lambda r: r["n_eff"])
        what, truth = "K=32 moments of Beta(2,5) via its CDF table", None
    run(max(pilot_n // 50, 256), 1)                                           # warm up the thread pool
    pilot, dt = timed(lambda: run(pilot_n, 7))
    rate = units(pilot) / dt
    per_item = units(pilot) / pilot_n
    n = int(min(max(rate * target_seconds / per_item, pilot_n), cap))
    res, dt = timed(lambda: run(n, 42))
    out = dict(value=units(res) / dt, unit="MH steps/s" if config == "c4" else "samples/s",
               cores=oracle.num_threads(), kind="port",
               sample=f"{what}: {'n_chains' if config == 'c4' else 'n_samples'}={n:.3g} ({units(res):.4g} units) of the "
                      f"{config} workload, oracle/mcx_oracle.c (C restatement of the reference kernel, f32, same counter "
                      f"stream) with OpenMP on {oracle.num_threads()} threads, {dt:.1f} s wall")
    if truth is not None:
        out["mean_error_vs_truth"] = [float(v) for v in (res["sums"][:len(truth)] / res["n_eff"] - truth)]
    try:                                          # name the host CPU next to the core count
        with open("/proc/cpuinfo") as fh:
            models = [line.split(":", 1)[1].strip() for line in fh if line.startswith("model name")]
        out["cpu_model"] = models[0] if models else None
        out["logical_cpus"] = os.cpu_count()
    except OSError:
        pass
    return out


# ---------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------
def run_cold_probe(args):
    """Child of cold_probe(): one blocking call of the config at 1e6 samples with no disk cache."""
    t_import = time.perf_counter()
    import baseline_configs as bc
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    t0 = time.perf_counter()
    mc = MonteCarloIntegrator(device=0)
    wl = bc.get(args.config, Distribution)
    t1 = time.perf_counter()
    small = 4096 if args.config == "c4" else 1_000_000
    wl.blocking(mc, small, 42)
    t2 = time.perf_counter()
    wl.blocking(mc, small, 43)
    t3 = time.perf_counter()
    print(json.dumps(dict(cold_first_call_ms=(t2 - t1) * 1e3, warm_second_call_ms=(t3 - t2) * 1e3,
                          engine_create_ms=(t1 - t0) * 1e3, import_ms=(t0 - t_import) * 1e3,
                          what=f"{args.config} blocking call at {small} {'chains' if args.config == 'c4' else 'samples'}, "
                               f"MCX_NO_DISK_CACHE=1: emission + hiprtc compile + module load + launch + read-back")),
          flush=True)


def run_rehearsal(args, rank, world):
    """--rehearse-cpu: the launcher's plumbing without a GPU (spawn -> gloo rendezvous -> one all-reduce -> relay)."""
    import numpy as np
    import torch.distributed as dist

    from wgpu_montecarlo import distributed as mcd

    if world > 1:
        dist.init_process_group("gloo")
    group = mcd.resolve_group("world") if world > 1 else None
    total = mcd.all_reduce_host(group, np.array([float(rank + 1)]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(dict(rehearsal=True, n_gpus=world, ranks_sum=float(total[0]), expected=world * (world + 1) / 2,
                              cold=json.loads(os.environ.get("MCX_BENCH_COLD", "null")))), flush=True)


def timed_loop(torch, dist, world, wl, prepared, n_step, out, warmup, steps, n_streams, device):
    """W untimed + exactly K timed steps, bracketed by barrier + synchronize on both sides. Returns (seconds, n_eff)."""
    pending = []
    # steps are independent integrals (own seed, own row of `out`): issuing them on alternating streams lets the next
    # step's workgroups start while the previous step's last workgroups drain (one engine, per-stream scratch)
    streams = [torch.cuda.Stream(device=device) for _ in range(max(1, n_streams))] if n_streams > 1 else [None]

    def step(i):
        # the all-reduce of step i overlaps the kernel of step i + 1 (different rows of `out`)
        if streams[0] is None:
            n_eff_, work = wl.launch(prepared, n_step, 42 + i, out[i], async_op=True)
        else:
            with torch.cuda.stream(streams[i % len(streams)]):
                n_eff_, work = wl.launch(prepared, n_step, 42 + i, out[i], async_op=True)
        if work is not None:
            pending.append(work)
        return n_eff_

    def fence():
        while pending:
            pending.pop().wait()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n_eff = 0
    for i in range(warmup):
        n_eff = step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        n_eff = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, n_eff


def accuracy(np, wl, out_rows, n_eff):
    """|mean - truth| against the 3-sigma band for every timed step. out_rows: [steps, rows] all-rank sums."""
    means = out_rows / float(n_eff)
    if wl.n_steps:                                   # MCMC: row k holds the accepted steps (burn-in included)
        total_chains = n_eff // wl.n_steps
        accept = float(np.mean(out_rows[:, wl.k]) / (float(total_chains) * (wl.n_steps + wl.n_burnin)))
        truth, band = wl.band(n_eff, accept)
    else:
        accept = None
        truth, band = wl.band(n_eff)
    abs_err = np.abs(means[:, :wl.k] - truth)
    d = dict(abs_err_vs_truth=abs_err.max(axis=0).tolist(), three_sigma=band.tolist(),
             worst_err_over_3sigma=float((abs_err / band).max()), frac_within_3sigma=float((abs_err <= band).mean()))
    if accept is not None:
        d["accept_rate"] = accept
    return d


class Ctx:
    """What every config leg of one rank shares: the process group, the device, the product's classes."""

    def __init__(self, args, torch, dist, np, world, rank, local_rank, device):
        import baseline_configs as bc
        from wgpu_montecarlo import Distribution, MonteCarloIntegrator
        from wgpu_montecarlo import runtime as rt

        self.args, self.torch, self.dist, self.np = args, torch, dist, np
        self.world, self.rank, self.local_rank, self.device = world, rank, local_rank, device
        self.bc, self.Distribution, self.MonteCarloIntegrator, self.rt = bc, Distribution, MonteCarloIntegrator, rt
        self.group = "world" if world > 1 else None


def uniform_draws_per_step(name, wl, units_per_step):
    """Uniforms one step draws from the counter stream: the reference's hash takes a 32-bit counter,
    so beyond 2^32 draws per call its estimates stop converging and only the Philox stream keeps the 3-sigma claim."""
    return 2 * units_per_step if name == "c4" else units_per_step      # MH step: one proposal + one accept uniform


def sharded_equals_single(ctx, wl, prepared, n_step, n_eff):
    """N > 1 self-check of the sharding itself: the all-reduced sums of the rank shards against the SAME logical grid run
    whole on rank 0 (shard = (0, 1)). The samples are identical by construction (global (seed, idx, iter) counters);
    only the summation order differs, so the means agree to ~1e-9 -- a dropped or double-counted shard boundary
    (one unit of every logical thread) would move E[x^2] by >= 1e-4 relative. MCMC: the accepted-step counts, integers,
    must agree exactly."""
    torch, dist, np = ctx.torch, ctx.dist, ctx.np
    seed = 4242
    sharded = torch.zeros(wl.rows, dtype=torch.float64, device=ctx.device)
    wl.launch(prepared, n_step, seed, sharded)                       # shards + blocking all-reduce
    single = torch.zeros(wl.rows, dtype=torch.float64, device=ctx.device)
    if ctx.rank == 0:
        wl.launch(prepared, n_step, seed, single, shard=(0, 1))      # the whole grid on this GPU, no collective
    torch.cuda.synchronize()
    dist.barrier()
    if ctx.rank != 0:
        return None
    a, b = sharded.cpu().numpy() / float(n_eff), single.cpu().numpy() / float(n_eff)
    truth, band = wl.band(n_eff, 0.66) if wl.n_steps else wl.band(n_eff)
    scale = np.abs(truth) + band * np.sqrt(float(n_eff)) / 3.0      # |mean| + one standard deviation of the integrand
    rel = np.abs(a[:wl.k] - b[:wl.k]) / scale
    ok = bool(np.all(rel <= 1e-7))
    d = dict(ok=ok, max_mean_diff_over_scale=float(rel.max()), tolerance=1e-7, seed=seed,
             what="all-reduced shard sums vs the whole grid on rank 0 (shard=(0,1)), same seed")
    if wl.n_steps:
        d["accepted_steps_equal"] = bool(sharded[wl.k].item() == single[wl.k].item())
        d["ok"] = ok and d["accepted_steps_equal"]
    return d


def shard_timings(ctx, wl, prepared, nominal, scaling, one_gpu_ms, reps=8):
    """N = 1 only: what each rank of a 2 / 4 / 8-GPU run of this config would launch -- the shard (rank 0 and the last rank of
    N) of the step that run takes (weak: N x the nominal size; strong: the nominal size) -- timed on THIS GPU with
    `launch(..., shard=(r, N))`. Not a multi-GPU measurement: it leaves out the all-reduce of K doubles (latency-bound,
    overlapped with the next step) and assumes N GPUs like this one; it is the per-rank kernel time a real run cannot
    beat, in the driver's own record even when no 8-GPU node is available to it."""
    torch = ctx.torch
    out = torch.zeros(wl.rows, dtype=torch.float64, device=ctx.device)
    table = {}
    for n in (2, 4, 8):
        n_step = int(nominal) * (n if scaling == "weak" else 1)
        worst = 0.0
        for r in (0, n - 1):
            wl.launch(prepared, n_step, 7, out, shard=(r, n))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for j in range(reps):
                wl.launch(prepared, n_step, 100 + j, out, shard=(r, n))
            e1.record()
            torch.cuda.synchronize()
            worst = max(worst, e0.elapsed_time(e1) / reps)
        table[str(n)] = dict(slowest_shard_ms=worst,
                             expected_speedup=(n * one_gpu_ms / worst) if scaling == "weak" else (one_gpu_ms / worst))
    table["note"] = ("single-GPU timings of the shards an N-GPU run launches (rank 0 and rank N-1), no collective: an expectation, "
                     "not a multi-GPU measurement")
    return table


def measure_config(ctx, name, steps, warmup, primary, cpu_seconds):
    """One BASELINE config through the same timed loop: reference stream, Philox stream, the dominant kernel's time and
    roofline, the blocking API call, the N > 1 self-check, the CPU baseline. Returns the leg's dict (rank 0) or None."""
    args, torch, dist, np, world, rank = ctx.args, ctx.torch, ctx.dist, ctx.np, ctx.world, ctx.rank
    bc, device = ctx.bc, ctx.device
    wl = bc.get(name, ctx.Distribution)
    scaling = (args.scaling if primary else None) or DEFAULT_SCALING[name]
    nominal = wl.nominal
    if primary and args.samples_per_gpu is not None and name == "c2":
        nominal = int(args.samples_per_gpu)
    n_step = int(nominal * args.scale) * (world if scaling == "weak" else 1)        # size of one step, whole job
    n_streams = args.streams if primary else 1

    def make(rng):
        mc = ctx.MonteCarloIntegrator(device=ctx.local_rank, rng=rng, process_group=ctx.group)
        if primary and args.target_phys:
            mc._engine.set_target_threads(args.target_phys)
        if args.mcmc_segments is not None:
            mc._engine.set_mcmc_segments(args.mcmc_segments)
        return mc, wl.prepare(mc)

    integ, prepared = make(args.rng)
    out = torch.zeros(warmup + steps + 1, wl.rows, dtype=torch.float64, device=device)

    def prewarm(prep):
        """Device warm-up, untimed and outside the W + K steps: after process start-up the GPU sits in its idle clock
        state and needs tens of milliseconds of sustained work to leave it (DVFS) -- a power-management transient,
        not a property of the kernel. Returns the number of steps launched."""
        if args.prewarm_ms <= 0:
            return 0
        scratch0 = torch.zeros(wl.rows, dtype=torch.float64, device=device)
        launched, t_end = 0, time.perf_counter() + args.prewarm_ms * 1e-3
        while launched < 2 or time.perf_counter() < t_end:
            for _ in range(4):
                wl.launch(prep, n_step, 7, scratch0, reduce=False)
            launched += 4
            torch.cuda.synchronize()
        return launched

    prewarm_steps = prewarm(prepared)
    elapsed, n_eff = timed_loop(torch, dist, world, wl, prepared, n_step, out, warmup, steps, n_streams, device)
    acc = accuracy(np, wl, out[warmup:warmup + steps].cpu().numpy(), n_eff)

    # the same loop on the Philox stream: beyond 2^32 uniforms per step (8 ranks x 1e9 samples; C4; C5) the reference's
    # 32-bit counter hash is oversubscribed and its estimates stop converging -- report both
    philox = None
    if not args.no_philox and args.rng != "philox":
        _, prepared_px = make("philox")
        prewarm(prepared_px)
        out_px = torch.zeros_like(out)
        w_px = min(warmup, 3)
        el_px, n_eff_px = timed_loop(torch, dist, world, wl, prepared_px, n_step, out_px, w_px, steps, n_streams, device)
        philox = dict(value=wl.units(n_eff_px) * steps / el_px, ms_per_step=el_px / steps * 1e3,
                      **accuracy(np, wl, out_px[w_px:w_px + steps].cpu().numpy(), n_eff_px))

    # dominant kernel: R calls of this rank's shard back to back on the stream they run on, NO collective, one
    # HIP-event pair around all of them -> per-call time of main (+ side-stream segments) + fold kernel including
    # launch gaps. A time-segmented MCMC call is S x 2 launches on two streams joined before the fold: the event pair
    # on the calling stream brackets all of them.
    reps = max(10, min(steps, 40))
    scratch = torch.zeros(wl.rows, dtype=torch.float64, device=device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    wl.launch(prepared, n_step, 7, scratch, reduce=False)
    torch.cuda.synchronize()
    ev0.record()
    for j in range(reps):
        wl.launch(prepared, n_step, 100 + j, scratch, reduce=False)
    ev1.record()
    torch.cuda.synchronize()
    kernel_ms = ev0.elapsed_time(ev1) / reps
    launch = integ._engine.last_launch()
    if world > 1:
        kms = torch.tensor([kernel_ms], dtype=torch.float64, device=device)
        gathered = [torch.zeros_like(kms) for _ in range(world)]
        dist.all_gather(gathered, kms)
        per_rank_kernel_ms = [float(t.item()) for t in gathered]
    else:
        per_rank_kernel_ms = [kernel_ms]

    units_per_step = wl.units(n_eff)
    units_per_launch = units_per_step / world                   # this rank's shard
    try:
        module_key = ctx.rt.module_key(prepared._plan.module.user_src, prepared._plan.desc)
    except Exception:                                # noqa: BLE001 -- the key only selects which profile entry is quoted
        module_key = None
    ops, ops_source = bc.ops_per_unit(name, module_key)
    valu_achieved = units_per_launch * ops / (kernel_ms * 1e-3)
    # algorithmic HBM bytes of one call: one rows*8-byte record per workgroup per launch; a time-segmented MCMC call (S x 2
    # launches; last_launch reports the workgroups of one segment, both halves) also carries {x, w} = 8 bytes per chain
    # through HBM between segments: S writes + (S - 1) reads
    segments = launch.get("segments", 0)
    records = launch["n_blocks"] * (segments if segments else max(launch["launches"], 1))
    hbm_bytes = records * wl.rows * 8.0
    if segments:
        hbm_bytes += (n_eff // wl.n_steps) / world * 8.0 * (2 * segments - 1)
    hbm_gbps = hbm_bytes / (kernel_ms * 1e-3) / 1e9

    check = sharded_equals_single(ctx, wl, prepared, n_step, n_eff) if world > 1 else None
    shards = shard_timings(ctx, wl, prepared, nominal * args.scale, scaling, kernel_ms) if (world == 1 and not under_profiler()) else None

    # blocking Python-API latency for the same call (plan-cache lookup + launch + the K doubles back), rank-local
    api = None
    if world == 1:
        api_times = []
        for _ in range(8):
            t1 = time.perf_counter()
            res = wl.blocking(integ, n_step, 42)
            api_times.append((time.perf_counter() - t1) * 1e3)
        api = dict(api_first_call_ms=api_times[0], api_call_ms=float(np.median(api_times[2:])), api_values=res.values[:4].tolist())

    if rank != 0:
        return None
    value = units_per_step * steps / elapsed
    backend = "RCCL" if args.backend == "nccl" else args.backend
    draws = uniform_draws_per_step(name, wl, units_per_step)
    # HBM traffic and executed instruction counts of this kernel come from separate rocprofv3 --pmc passes of this same
    # command (tools/profile_all.sh -> profiles/rNN_pmc_summary.txt); quoted only when that profile was taken with the
    # launch geometry of this run, else null
    traffic, pmc = None, pmc_summary(name, launch, world)
    if pmc:
        traffic = pmc.get("hbm_bytes_per_launch_corrected")
    leg = {
        "metric": "samples/sec (whole node), K=4 fused functions on N(0,1)" if name == "c2" else
                  f"{wl.unit} (whole node), BASELINE config {name}",
        "value": value,
        "unit": wl.unit,
        "steps": steps,
        "warmup": warmup,
        "device_prewarm": {"ms": args.prewarm_ms, "steps": prewarm_steps,
                           "note": "untimed launches of the same step before the W warm-up steps, so that the W + K steps "
                                   "do not run on the idle clock state the GPU is in after process start-up"},
        "ms_per_step": elapsed / steps * 1e3,
        "scaling": scaling,
        "config": {
            "workload": wl.title + f"; logical grid T={'1048576 chains' if name == 'c4' else 65536}; "
                        f"{scaling} scaling: {n_step:.4g} {'chains' if name == 'c4' else 'samples'} per step over {world} GPU(s)",
            "name": name,
            "size_per_step": n_step,
            "n_eff_per_step": int(n_eff),
            "units_per_step": int(units_per_step),
            "parallelism": (f"{'chain' if name == 'c4' else 'sample-grid'} shards x{world}, one {backend} "
                            f"sum all-reduce of {wl.rows} f64 per step") if world > 1 else "single GPU",
            "accumulate": "f32 registers per <= 128 units -> f64",
            "rng": args.rng, "streams": n_streams,
            "mcmc_segments": "auto" if args.mcmc_segments is None else args.mcmc_segments,
            "hip_runtime": ctx.rt.hip_runtime(),
        },
        **acc,
        "uniform_draws_per_step": int(draws),
        # which stream the 3-sigma claim holds for at this size, by construction (not by this run's luck): the reference's
        # counter hash has 2^32 distinct inputs, Philox 2^128
        "stream_valid": {"pcg_ref": bool(draws <= 2**32), "philox": True},
        "per_gpu_units_per_s": value / world,
        "philox": philox,
        "roofline": {
            "bound": "valu",
            "achieved": valu_achieved / 1e12,
            "peak": bc.VALU_PEAK_LANEOPS / 1e12,
            "unit": "Tlane-op/s",
            "frac": valu_achieved / bc.VALU_PEAK_LANEOPS,
            "traffic": traffic,
            "kernel": "mcx_mcmc_kernel" if name == "c4" else "mcx_integrate_kernel",
            "kernel_ms": kernel_ms,
            "kernel_ms_method": f"{reps} calls of this rank's shard (main + fold kernel, no collective) back to back "
                                f"on one stream, one HIP-event pair around all of them",
            "per_rank_kernel_ms": per_rank_kernel_ms,
            "ops_per_unit": ops,
            "ops_per_unit_source": ops_source,
            "executed_valu_per_unit": pmc.get("valu_inst_per_unit") if pmc else None,
            "valu_issue_frac_of_peak": pmc.get("valu_issue_frac_of_peak") if pmc else None,
            "pmc_source": f"profiles/{pmc['_file']} (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes; SQ_INSTS_VALU)" if pmc else None,
            "issue_model": issue_model(ctx, module_key, kernel_ms, units_per_launch, pmc),
            "units_per_launch": units_per_launch,
            "launch": launch,
            "note": "the binding resource of these fused kernels is vector-ALU issue (SURVEY.md 8d): ops_per_unit = the hot "
                    "loop of the code object that ran, in lane-op equivalents with the survey's weights (plain op 1, integer "
                    "multiply 4, transcendental 2; tools/issue_model.py), peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz; "
                    "issue_model prices the same instructions with the measured per-class issue costs; the HBM view of the "
                    "same launch is in roofline_hbm",
        },
        # the same kernel against the HBM roofline, in the generic schema: algorithmic bytes = rows * 8 B per workgroup
        "roofline_hbm": {"bound": "hbm", "achieved": hbm_gbps, "peak": bc.HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": hbm_gbps / bc.HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": hbm_bytes,
                         "note": "the kernel writes one rows*8-byte record per workgroup and reads only code, arguments "
                                 "and <= 72 KiB of tables (staged into LDS once per workgroup, L2-resident): HBM is "
                                 "~1e-5 of peak by design, not the bound"},
    }
    if check is not None:
        leg["sharded_equals_single"] = check
    if shards is not None:
        leg["shard_timings"] = shards
    if api:
        leg.update(api)
    if not args.no_cpu_baseline and world == 1:
        leg["cpu_baseline"] = cpu_baseline(name, cpu_seconds)
        if name == "c2":
            leg["cpu_baseline_numpy"] = cpu_baseline_numpy(min(cpu_seconds, 4.0))
    return leg


def issue_model(ctx, key, kernel_ms, units_per_launch, pmc):
    """SIMD-cycles per wave-unit, measured against modelled. measured = kernel time x clock x 1024 SIMDs / wave-units;
    modelled = sum over instruction classes of (count in the hot loop of THIS module's code object) x (issue cost of the
    class, profiles/r02_valu_issue_microbench.txt). The class counts are read from profiles/rNN_issue_model.json, which
    tools/issue_model.py writes from the disassembly of the cached code object and which names that object by its cache
    key: a module whose key is not in the file (kernel changed since) reports modelled = null rather than a stale figure."""
    path = latest_profile("issue_model.json")
    clock = (pmc or {}).get("effective_clock_ghz") or ctx.bc.NOMINAL_CLOCK_GHZ
    wave_units = units_per_launch / 64.0
    measured = kernel_ms * 1e-3 * clock * 1e9 * 1024.0 / wave_units
    out = dict(cycles_per_unit_measured=measured, clock_ghz=clock,
               clock_source="GRBM_GUI_ACTIVE of the committed PMC pass" if (pmc or {}).get("effective_clock_ghz") else "nominal",
               module_key=key, cycles_per_unit_modelled=None, source=None)
    if path is None or key is None:
        return out
    try:
        table = json.loads(path.read_text())
    except ValueError:
        return out
    entry = table.get("modules", {}).get(key)
    if entry:
        out.update(cycles_per_unit_modelled=entry["cycles_per_unit_modelled"], classes_per_unit=entry["classes_per_unit"],
                   class_costs=table.get("class_costs"), source=f"profiles/{path.name}")
    return out


def latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that has one."""
    hits = sorted((ROOT / "profiles").glob(f"r[0-9][0-9]_{suffix}"))
    return hits[-1] if hits else None


def cpu_baseline_numpy(target_seconds):
    """The 'benchmark.py-style' figure of SURVEY.md 8(d)(ii): the reference's own CPU comparison is a numpy / Python loop
    on ONE core (examples/benchmark.py:43-68). Here: oracle/numpy_port.py, a vectorised numpy restatement of the C2
    kernel on the reference's counter stream, single-threaded."""
    import numpy as np

    from oracle import numpy_port as npp

    t0 = time.perf_counter()
    npp.normal_moments(4, 2_000_000, seed=1)
    rate = 2_000_000 / (time.perf_counter() - t0)
    n = int(min(max(rate * target_seconds, 2_000_000), 2e9))
    t0 = time.perf_counter()
    sums, n_eff = npp.normal_moments(4, n, seed=42)
    dt = time.perf_counter() - t0
    return dict(value=n_eff / dt, unit="samples/s", cores=1, kind="port",
                sample=f"K=4 moments on N(0,1): n_samples={n:.3g} of the c2 workload, oracle/numpy_port.py (vectorised numpy "
                       f"restatement of the reference kernel, f32 terms, same counter stream), one core, {dt:.1f} s wall",
                mean_error_vs_truth=[float(v) for v in (sums / n_eff - np.array([0.0, 1.0, 0.0, 3.0]))])


def protocol_integrand(x):
    return x / (math.exp(math.sin(x)) + math.cos(math.exp(x)))


def reference_protocol(ctx):
    """The reference's own benchmark (examples/benchmark.py:8-68) on this GPU, N = 1: wall clock around the blocking
    `integrate([f], Normal(0,1), n)` for f(x) = x / (exp(sin x) + cos(exp x)), one warm-up call at n = 1000, at the smallest,
    a middle and the largest size of its list (1e3, 1e5, 1e7) and at 1e9; beside it the kernel time, and the single-core numpy
    evaluation the reference compares itself with (vectorised here -- the reference's per-element loop is 100x slower), at 1e7.
    The reference's call recompiles its shader every time (src/engine.rs:325-331); here the first call of a process compiles or
    loads the code object (`first_call_ms`) and the others find it cached."""
    np = ctx.np
    f = protocol_integrand
    mc = ctx.MonteCarloIntegrator(device=ctx.local_rank)
    normal = ctx.Distribution.normal                      # built inline in every call, as examples/benchmark.py:33-38 writes it
    t0 = time.perf_counter()
    mc.integrate([f], normal(0.0, 1.0), n_samples=1000)
    out = {"integrand": "x / (exp(sin x) + cos(exp x)) on N(0,1), examples/benchmark.py of the reference", "math": "default",
           "first_call_ms": (time.perf_counter() - t0) * 1e3, "calls": []}
    for n in (1_000, 100_000, 10_000_000, 1_000_000_000):
        times = []
        for _ in range(12):
            t0 = time.perf_counter()
            res = mc.integrate([f], normal(0.0, 1.0), n_samples=n)
            times.append((time.perf_counter() - t0) * 1e3)
        out["calls"].append({"n_samples": n, "n_eff": res.meta["n_eff"], "call_ms": float(np.median(times[2:])), "kernel_ms": res.meta["kernel_ms"],
                             "samples_per_s": res.meta["n_eff"] / (float(np.median(times[2:])) * 1e-3), "value": float(res.values[0])})
    xs = np.random.default_rng(0).standard_normal(10_000_000).astype(np.float32)
    t0 = time.perf_counter()
    float(np.mean(xs / (np.exp(np.sin(xs)) + np.cos(np.exp(xs)))))
    out["numpy_vectorised_1core_ms_at_1e7"] = (time.perf_counter() - t0) * 1e3
    return out


def core_binding_leg(ctx):
    """The drop-in boundary itself, N = 1: the payloads the reference's own Python half hands to its native module for BASELINE
    configs[1..3] (tests/golden/boundary_payloads.*: its transpiler's WGSL, its importance-sampling wrapper text, parameter dicts,
    float32 tables -- recorded data, replayed verbatim at full size) through this package's `_core` binding (wgpu_montecarlo/_core.py,
    the reference's `_core.MonteCarloIntegrator` interface over libmcx), blocking calls, best of 3."""
    sys.path.insert(0, str(ROOT / "tools"))
    import core_payload_bench as cpb
    from wgpu_montecarlo import _core

    calls = cpb.golden_calls()
    core = _core.MonteCarloIntegrator(device=ctx.local_rank)
    out = {"binding": "wgpu_montecarlo._core.MonteCarloIntegrator (WGSL text in, float32[K] out)", "math": core._math, "calls": {}}
    for name, index in (("c2", 1), ("c3", 2), ("c4", 3)):
        method, a = calls[index]
        ms, got = cpb.best(lambda: getattr(core, method)(*a))
        units = (a[3] if method != "integrate_mcmc" else (a[5] + a[7]) * a[6])
        out["calls"][name] = {"method": method, "k": len(a[0]), "call_ms": ms, "units_per_s": units / (ms * 1e-3), "values": [float(v) for v in got[:4]]}
    return out


def leg_names(args):
    """BASELINE configs timed next to the headline in the same line (the driver only ever runs the default command)."""
    if args.legs == "none":
        return []
    if args.legs != "auto":
        return [c for c in args.legs.replace(" ", "").split(",") if c and c != args.config]
    if args.config != "c2" or under_profiler() or args.scale != 1.0:
        return []
    return ["c3", "c4", "c5"]


def run_rank(args):
    import datetime

    import numpy as np
    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before anything initialises the GPU (ranks started by torch.distributed.run)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if args.rehearse_cpu:
        return run_rehearsal(args, rank, world)
    # cold-start probe: a child of rank 0, started before this process touches the GPU (a parent launcher has already
    # run it and handed the result down)
    cold = json.loads(os.environ["MCX_BENCH_COLD"]) if "MCX_BENCH_COLD" in os.environ else (cold_probe(args) if rank == 0 else None)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        # a desynchronised rank must end the job, not hang it: collectives give up after 5 minutes
        timeout = datetime.timedelta(seconds=300)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=timeout)
        else:
            dist.init_process_group(args.backend, timeout=timeout)

    ctx = Ctx(args, torch, dist, np, world, rank, local_rank, device)
    t_start = time.perf_counter()

    # N > 1: is the collective the RCCL one, and does it add up? An all-reduce of 1.0 over the group must give `world`.
    rccl_sum_of_ones = None
    if world > 1:
        ones = torch.ones(1, dtype=torch.float64, device=device)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl_sum_of_ones = float(ones.item())

    line = measure_config(ctx, args.config, args.steps, args.warmup, True, args.cpu_seconds)
    legs = {}
    for name in leg_names(args):
        # legs share the headline's process group: every rank runs the same deterministic code, so a failure (a module
        # that does not compile, an invalid launch) is raised on all ranks at the same point and recorded, not fatal
        try:
            legs[name] = measure_config(ctx, name, min(args.steps, args.leg_steps), min(args.warmup, 3), False, args.leg_cpu_seconds)
        except Exception as exc:                     # noqa: BLE001 -- the headline must survive a broken leg
            legs[name] = {"error": f"{type(exc).__name__}: {exc}"[:600]}
    protocol = None
    if world == 1 and legs and not under_profiler():
        try:
            protocol = reference_protocol(ctx)
        except Exception as exc:                         # noqa: BLE001
            protocol = {"error": f"{type(exc).__name__}: {exc}"[:600]}
    binding = None
    if world == 1 and legs and not under_profiler():
        try:
            binding = core_binding_leg(ctx)
        except Exception as exc:                         # noqa: BLE001
            binding = {"error": f"{type(exc).__name__}: {exc}"[:600]}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        backend = "RCCL" if args.backend == "nccl" else args.backend
        line.update({
            "n_gpus": world,
            "higher_is_better": True,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "rccl_ranks": world if (world > 1 and args.backend == "nccl") else None,
            "collective_backend": args.backend if world > 1 else None,
            "cold": cold,
            "configs": legs,
            "reference_protocol": protocol,
            "core_binding": binding,
            "self_check": None if world == 1 else {
                "rccl_sum_of_ones": rccl_sum_of_ones,
                "rccl_sum_of_ones_ok": rccl_sum_of_ones == float(world),
                "collective": backend,
                "sharded_equals_single": {n: (l or {}).get("sharded_equals_single") for n, l in [(args.config, line)] + list(legs.items())},
            },
            "bench_wall_s": time.perf_counter() - t_start,
        })
        # key order: the contract's keys first
        head = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"]
        ordered = {k: line[k] for k in head if k in line}
        ordered.update({k: v for k, v in line.items() if k not in ordered})
        sys.stdout.flush()
        print(json.dumps(ordered), flush=True)


def pmc_summary(config, launch, world):
    """The committed PMC digest of this config's kernel (profiles/rNN_pmc_summary.txt, latest round), if its launch
    geometry matches this run: same workgroup size, same grid, same number of launches per call."""
    path = latest_profile("pmc_summary.txt")
    if world != 1 or path is None:
        return None
    for text in path.read_text().splitlines():
        try:
            d = json.loads(text)
        except ValueError:
            continue
        if d.get("config") != config or str(d.get("workgroup")) != str(launch["block"]):
            continue
        if int(d.get("launches_per_call", 1)) != max(launch["launches"], 1):
            continue
        grid = launch["n_blocks"] * launch["block"]
        if str(d.get("grid")) == str(grid) or str(d.get("grid_per_segment")) == str(grid):
            d["_file"] = path.name
            return d
    return None


def prewarm_cache():
    """hiprtc-compile the modules bench.py and smoke() launch into the in-tree code-object cache (needs no GPU): every
    BASELINE config on both streams -- through the product's own planner (MonteCarloIntegrator.planner), so the descs
    are the ones a GPU run builds --, the MCMC workgroup-size variants a chain-sharded run selects, and the K = 4 base
    modules of each sampler."""
    import baseline_configs as bc
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    K = 4
    src = functions_to_hip(bc.moment_functions(4))
    for dist in (rt.DIST_UNIFORM, rt.DIST_NORMAL, rt.DIST_EXPONENTIAL, rt.DIST_CUSTOM):
        rt.precompile(src, rt.make_desc(rt.KIND_INTEGRATE, K, dist))
    built = {}
    for rng in ("pcg_ref", "philox"):
        mc = MonteCarloIntegrator.planner(rng=rng)
        for name in ("c2", "c3", "c4", "c5"):
            wl = bc.get(name, Distribution)
            prepared = wl.prepare(mc)
            built[(name, rng)] = prepared._plan.module.key
            if name == "c4":
                for parts in (2, 4, 8):             # a rank's share of 1 048 576 chains picks its workgroup size
                    prepared._select(wl.nominal, shard=(0, parts))
                    built[(name, rng, parts)] = prepared._plan.module.key
    built["reference_protocol"] = MonteCarloIntegrator.planner().prepare_integrate([protocol_integrand], Distribution.normal(0.0, 1.0))._plan.module.key
    return built


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.cold_probe:
        return run_cold_probe(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main() or 0)
