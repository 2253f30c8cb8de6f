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
samples. Default scaling: c2 / c3 weak (nominal size per GPU), c4 / c5 strong (BASELINE fixes the total).
"""
import argparse
import json
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
    ap.add_argument("--mcmc-segments", type=int, default=0,
                    help="c4 only, opt-in: run each MCMC call as two chain halves on two streams x this many step segments "
                         "(mcx_engine_set_mcmc_segments; measured 8.5 -> 8.0 ms at 8). Default 0: one launch per call, so "
                         "that a traced launch is the whole call")
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


def run_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist

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
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    import baseline_configs as bc
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator
    from wgpu_montecarlo import runtime as _rt

    wl = bc.get(args.config, Distribution)
    scaling = args.scaling or DEFAULT_SCALING[args.config]
    nominal = wl.nominal
    if args.samples_per_gpu is not None and args.config == "c2":
        nominal = int(args.samples_per_gpu)
    n_step = int(nominal * args.scale) * (world if scaling == "weak" else 1)        # size of one step, whole job
    group = "world" if world > 1 else None

    def make(rng):
        mc = MonteCarloIntegrator(device=local_rank, rng=rng, process_group=group)
        if args.target_phys:
            mc._engine.set_target_threads(args.target_phys)
        if args.mcmc_segments:
            mc._engine.set_mcmc_segments(args.mcmc_segments)
        return mc, wl.prepare(mc)

    integ, prepared = make(args.rng)
    out = torch.zeros(args.warmup + args.steps + 1, wl.rows, dtype=torch.float64, device=device)

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
    elapsed, n_eff = timed_loop(torch, dist, world, wl, prepared, n_step, out, args.warmup, args.steps, args.streams, device)
    acc = accuracy(np, wl, out[args.warmup:args.warmup + args.steps].cpu().numpy(), n_eff)

    # the same loop on the Philox stream: beyond 2^32 uniforms per step (8 ranks x 1e9 samples; C4; C5) the reference's
    # 32-bit counter hash is oversubscribed and its estimates stop converging -- report both
    philox = None
    if not args.no_philox and args.rng != "philox":
        _, prepared_px = make("philox")
        prewarm(prepared_px)
        out_px = torch.zeros_like(out)
        el_px, n_eff_px = timed_loop(torch, dist, world, wl, prepared_px, n_step, out_px, min(args.warmup, 3),
                                     args.steps, args.streams, device)
        philox = dict(value=wl.units(n_eff_px) * args.steps / el_px, ms_per_step=el_px / args.steps * 1e3,
                      **accuracy(np, wl, out_px[min(args.warmup, 3):min(args.warmup, 3) + args.steps].cpu().numpy(), n_eff_px))

    # dominant kernel: R launches of this rank's shard back to back on the stream they run on, NO collective, one
    # HIP-event pair around all of them -> per-launch time of main + fold kernel including launch gaps
    reps = max(10, min(args.steps, 40))
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
    ops = bc.OPS_PER_UNIT[args.config]
    valu_achieved = units_per_launch * ops / (kernel_ms * 1e-3)
    hbm_bytes = launch["n_blocks"] * launch["launches"] * wl.rows * 8.0       # one rows*8-byte record per workgroup
    hbm_gbps = hbm_bytes / (kernel_ms * 1e-3) / 1e9

    # blocking Python-API latency for the same call (emission + launch + D2H of the K doubles), rank-local
    api = None
    if world == 1:
        api_times = []
        for _ in range(6):
            t1 = time.perf_counter()
            res = wl.blocking(integ, n_step, 42)
            api_times.append((time.perf_counter() - t1) * 1e3)
        api = dict(api_first_call_ms=api_times[0], api_call_ms=float(np.median(api_times[1:])), api_values=res.values[:4].tolist())

    line = None
    if rank == 0:
        value = units_per_step * args.steps / elapsed
        backend = "RCCL" if args.backend == "nccl" else args.backend
        # HBM traffic and executed instruction counts of this kernel come from separate rocprofv3 --pmc passes of this
        # same command (tools/profile_all.sh -> profiles/r02_pmc_summary.txt); quoted only when that profile was taken
        # with the launch geometry of this run, else null
        traffic, pmc = None, pmc_summary(args.config, launch, world)
        if pmc:
            traffic = pmc.get("hbm_bytes_per_launch_corrected")
        line = {
            "metric": "samples/sec (whole node), K=4 fused functions on N(0,1)" if args.config == "c2" else
                      f"{wl.unit} (whole node), BASELINE config {args.config}",
            "value": value,
            "unit": wl.unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "device_prewarm": {"ms": args.prewarm_ms, "steps": prewarm_steps,
                               "note": "untimed launches of the same step before the W warm-up steps, so that the W + K steps "
                                       "do not run on the idle clock state the GPU is in after process start-up"},
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl.title + f"; logical grid T={'1048576 chains' if args.config == 'c4' else 65536}; "
                            f"{scaling} scaling: {n_step:.4g} {'chains' if args.config == 'c4' else 'samples'} per step over {world} GPU(s)",
                "name": args.config,
                "size_per_step": n_step,
                "n_eff_per_step": int(n_eff),
                "units_per_step": int(units_per_step),
                "parallelism": (f"{'chain' if args.config == 'c4' else 'sample-grid'} shards x{world}, one {backend} "
                                f"sum all-reduce of {wl.rows} f64 per step") if world > 1 else "single GPU",
                "accumulate": "f32 registers per <= 128 units -> f64",
                "rng": args.rng, "streams": args.streams, "mcmc_segments": args.mcmc_segments,
                "hip_runtime": _rt.hip_runtime(),
            },
            "rccl_ranks": dist.get_world_size() if (world > 1 and args.backend == "nccl") else None,
            "collective_backend": args.backend if world > 1 else None,
            **acc,
            "per_gpu_units_per_s": value / world,
            "philox": philox,
            "cold": cold,
            "roofline": {
                "bound": "valu",
                "achieved": valu_achieved / 1e12,
                "peak": bc.VALU_PEAK_LANEOPS / 1e12,
                "unit": "Tlane-op/s",
                "frac": valu_achieved / bc.VALU_PEAK_LANEOPS,
                "traffic": traffic,
                "kernel": "mcx_mcmc_kernel" if args.config == "c4" else "mcx_integrate_kernel",
                "kernel_ms": kernel_ms,
                "kernel_ms_method": f"{reps} launches of this rank's shard (main + fold kernel, no collective) back to back "
                                    f"on one stream, one HIP-event pair around all of them",
                "per_rank_kernel_ms": per_rank_kernel_ms,
                "ops_per_unit": ops,
                "executed_valu_per_unit": pmc.get("valu_inst_per_unit") if pmc else None,
                "valu_issue_frac_of_peak": pmc.get("valu_issue_frac_of_peak") if pmc else None,
                "pmc_source": "profiles/r02_pmc_summary.txt (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes; SQ_INSTS_VALU)" if pmc else None,
                "units_per_launch": units_per_launch,
                "launch": launch,
                "note": "the binding resource of these fused kernels is vector-ALU issue (SURVEY.md 8d): ops_per_unit is "
                        "the ALGORITHMIC lane-op count of the config (DESIGN.md 4, tools/baseline_configs.py), peak = "
                        "256 CU x 4 SIMD x 32 lanes x 2.4 GHz; executed-instruction counts are in profiles/r02_*_pmc_*; "
                        "the HBM view of the same launch is in roofline_hbm",
            },
            # the same kernel against the HBM roofline, in the generic schema: algorithmic bytes = rows * 8 B per workgroup
            "roofline_hbm": {"bound": "hbm", "achieved": hbm_gbps, "peak": bc.HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": hbm_gbps / bc.HBM_PEAK_GBPS, "traffic": traffic,
                             "algorithmic_bytes_per_launch": hbm_bytes,
                             "note": "the kernel writes one rows*8-byte record per workgroup and reads only code, arguments "
                                     "and <= 56 KiB of tables (staged into LDS once per workgroup, L2-resident): HBM is "
                                     "~1e-5 of peak by design, not the bound"},
        }
        if api:
            line.update(api)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.config, args.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


def pmc_summary(config, launch, world):
    """The committed PMC digest of this config's kernel (profiles/r02_pmc_summary.txt), if its launch geometry matches."""
    path = ROOT / "profiles" / "r02_pmc_summary.txt"
    if world != 1 or not path.exists():
        return None
    for text in path.read_text().splitlines():
        try:
            d = json.loads(text)
        except ValueError:
            continue
        if d.get("config") == config and str(d.get("workgroup")) == str(launch["block"]) and \
                str(d.get("grid")) == str(launch["n_blocks"] * launch["block"]) and launch["launches"] == 1:
            return d
    return None


def prewarm_cache():
    """hiprtc-compile the bench / smoke modules into the in-tree code-object cache (needs no GPU)."""
    import baseline_configs as bc
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    K = 4
    src = functions_to_hip(bc.moment_functions(4))
    for dist in (rt.DIST_UNIFORM, rt.DIST_NORMAL, rt.DIST_EXPONENTIAL, rt.DIST_CUSTOM):
        rt.precompile(src, rt.make_desc(rt.KIND_INTEGRATE, K, dist))
    rt.precompile(src, rt.make_desc(rt.KIND_INTEGRATE, K, rt.DIST_NORMAL, unit_params=True))      # N(0,1): the headline
    rt.precompile(src, rt.make_desc(rt.KIND_INTEGRATE, K, rt.DIST_NORMAL, unit_params=True, rng=rt.RNG_PHILOX))


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
