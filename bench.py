#!/usr/bin/env python3
"""bench.py -- headline benchmark: samples/sec of the fused K=4 integrate on N(0,1) (BASELINE config C2).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--samples-per-gpu S]

One step = one pass of the hot path: integrate([x, x**2, x**3, x**4], Normal(0,1), n_samples = S*N)
(fused RNG + Box-Muller + 4 evaluations + two-stage f64 reduction; for N > 1 each rank runs its shard
of the SAME logical sample grid and the K partial sums are combined by one RCCL all-reduce). Weak
scaling: S = 1e9 samples per GPU per step. Inputs (four scalars) are kernel arguments; nothing is
staged from the host inside the timed region. Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for _p in (ROOT / "wgpu-monte-carlo_amd", ROOT):
    if str(_p) not in sys.path:
        sys.path.insert(0, str(_p))

K = 4
TRUTH = (0.0, 1.0, 0.0, 3.0)
# variance of x^k under N(0,1): E[x^2k] - E[x^k]^2 = 1, 2, 15, 96
SIGMA = (1.0, 2.0 ** 0.5, 15.0 ** 0.5, 96.0 ** 0.5)

# Algorithmic VALU cost of one sample of this workload, in lane-op equivalents (DESIGN.md "Roofline"):
# per Box-Muller pair 23 plain ops (N(0,1): no affine map) + 2 integer multiplies (x4) + 4 transcendentals (x2),
# per sample 3 multiplies + 4 adds  ->  (23 + 8 + 8) / 2 + 7 = 26.5   (27.5 for a general N(mean, std))
OPS_PER_SAMPLE = 26.5
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9      # 7.86e13: CUs x SIMDs x lanes x clock (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0


def moment_functions():
    f1 = lambda x: x
    f2 = lambda x: x**2
    f3 = lambda x: x**3
    f4 = lambda x: x**4
    return [f1, f2, f3, f4]


def prewarm_cache():
    """hiprtc-compile the bench / smoke modules into the in-tree code-object cache (needs no GPU)."""
    from wgpu_montecarlo import runtime as rt
    from wgpu_montecarlo.api import functions_to_hip

    src = functions_to_hip(moment_functions())
    for dist in (rt.DIST_UNIFORM, rt.DIST_NORMAL, rt.DIST_EXPONENTIAL, rt.DIST_CUSTOM):
        rt.precompile(src, rt.make_desc(rt.KIND_INTEGRATE, K, dist))
    rt.precompile(src, rt.make_desc(rt.KIND_INTEGRATE, K, rt.DIST_NORMAL, unit_params=True))      # N(0,1): the headline
    rt.precompile(src, rt.make_desc(rt.KIND_INTEGRATE, K, rt.DIST_NORMAL, unit_params=True, rng=rt.RNG_PHILOX))


def cpu_baseline(target_seconds: float):
    """Time the CPU oracle (plain-C restatement of the reference kernel, OpenMP over the logical thread index,
    all host cores) on a bounded slice of the same workload: a pilot run sizes the slice to ~target_seconds."""
    import oracle

    fns = [(oracle.FN_IDENTITY, 0), (oracle.FN_POW, 2), (oracle.FN_POW, 3), (oracle.FN_POW, 4)]
    oracle.integrate(fns, oracle.NORMAL, 0.0, 1.0, n_samples=1_000_000, seed=1)     # warm up the thread pool
    t0 = time.perf_counter()
    pilot = oracle.integrate(fns, oracle.NORMAL, 0.0, 1.0, n_samples=50_000_000, seed=7)
    rate = pilot["n_eff"] / (time.perf_counter() - t0)
    # at most 4e9: inside the 2^32 counter space of the reference stream; guard=1 as libmcx runs it (the strict
    # reference takes log(0) for the one hash value 0, which a slice this large does meet)
    n_samples = int(min(max(rate * target_seconds, 1e8), 4e9))
    t0 = time.perf_counter()
    res = oracle.integrate(fns, oracle.NORMAL, 0.0, 1.0, n_samples=n_samples, seed=42, guard=1)
    dt = time.perf_counter() - t0
    return dict(value=res["n_eff"] / dt, unit="samples/s", cores=oracle.num_threads(), kind="port",
                sample=f"K=4 moments on N(0,1), n={n_samples:.2e} (N_eff {res['n_eff']}) of the 1e9-per-GPU-per-step workload, "
                       f"oracle/mcx_oracle.c (C restatement of the reference kernel, f32, same counter stream) with OpenMP on "
                       f"{oracle.num_threads()} threads, {dt:.1f} s wall",
                mean_error_vs_truth=[float(v) for v in (res["sums"] / res["n_eff"] - np.array(TRUTH))])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--samples-per-gpu", type=float, default=1e9)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-clock budget of the CPU baseline slice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--target-phys", type=int, default=0, help="physical threads per launch (tuning)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the independent steps are issued on in turn. 2 lets the next step's workgroups fill "
                         "the CUs the previous step's tail leaves idle (measured 0.420 -> 0.397 ms per step); the default "
                         "stays 1 so that a launch's duration in the rocprofv3 trace is the kernel alone, not two "
                         "launches sharing the chip")
    ap.add_argument("--rng", default="pcg_ref", help="pcg_ref (the reference's stream; the headline) or philox")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; the measured configuration) or gloo (rehearsal)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (with --backend gloo) to exercise the N > 1 code path")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
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

    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    integ = MonteCarloIntegrator(device=local_rank, rng=args.rng)
    from wgpu_montecarlo import runtime as _rt

    integ_runtime = _rt.hip_runtime()
    if args.target_phys:
        integ._engine.set_target_threads(args.target_phys)
    prepared = integ.prepare_integrate(moment_functions(), Distribution.normal(0.0, 1.0))
    n_total = int(args.samples_per_gpu) * world
    out = torch.zeros(args.warmup + args.steps + 1, K, dtype=torch.float64, device=device)

    pending = []
    # steps are independent integrals (own seed, own row of `out`): issuing them on alternating streams lets the next
    # step's workgroups start while the previous step's last workgroups drain (one engine, per-stream scratch)
    streams = [torch.cuda.Stream(device=device) for _ in range(max(1, args.streams))]

    def step(i):
        # the all-reduce of step i overlaps the kernel of step i + 1 (different rows of `out`)
        with torch.cuda.stream(streams[i % len(streams)]):
            n_eff_, work = prepared.launch(n_total, 42 + i, out[i], async_op=True)
        if work is not None:
            pending.append(work)
        return n_eff_

    def fence():
        while pending:
            pending.pop().wait()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        n_eff = step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        n_eff = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # accuracy of every timed step (sums over all ranks are already in `out`)
    means = (out[args.warmup:args.warmup + args.steps] / float(n_eff)).cpu().numpy()
    abs_err = np.abs(means - np.array(TRUTH))
    three_sigma = 3.0 * np.array(SIGMA) / np.sqrt(float(n_eff))
    worst_ratio = float((abs_err / three_sigma).max())

    # dominant kernel: HIP-event duration on the stream it runs on, one launch at a time
    durations = []
    for j in range(10):
        step(args.warmup + args.steps)
        durations.append(integ._engine.last_kernel_ms())
    fence()
    kernel_ms = float(np.mean(durations))
    launch = integ._engine.last_launch()
    samples_per_launch = n_eff / world
    valu_achieved = samples_per_launch * OPS_PER_SAMPLE / (kernel_ms * 1e-3)
    hbm_bytes = launch["n_blocks"] * K * 8.0
    hbm_gbps = hbm_bytes / (kernel_ms * 1e-3) / 1e9

    # blocking Python-API latency for the same call (includes emission + launch + D2H of K doubles)
    api_times = []
    for _ in range(6):
        t1 = time.perf_counter()
        res = integ.integrate(moment_functions(), Distribution.normal(0.0, 1.0), n_samples=n_total, seed=42)
        api_times.append((time.perf_counter() - t1) * 1e3)
    api_first_ms, api_ms = api_times[0], float(np.median(api_times[1:]))
    fence()

    if rank == 0:
        value = n_total * args.steps / elapsed
        line = {
            "metric": "samples/sec (whole node), K=4 fused functions on N(0,1)",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "integrate([x, x**2, x**3, x**4], Normal(0,1)), n_samples=1e9 per GPU per step "
                            "(BASELINE configs[1]); logical grid T=65536",
                "n_samples_per_step": n_total,
                "n_eff_per_step": int(n_eff),
                "parallelism": (f"sample-grid shards x{world}, one {'RCCL' if args.backend == 'nccl' else args.backend} "
                                f"sum all-reduce of {K} f64 per step") if world > 1 else "single GPU",
                "accumulate": "f32 registers per 128 pairs -> f64",
                "rng": args.rng, "streams": len(streams),
                "hip_runtime": integ_runtime,
            },
            "abs_err_vs_truth": abs_err.max(axis=0).tolist(),
            "three_sigma": three_sigma.tolist(),
            "worst_err_over_3sigma": worst_ratio,
            "frac_within_3sigma": float((abs_err <= three_sigma).mean()),
            "per_gpu_samples_per_s": value / world,
            "api_call_ms": api_ms,                 # median of 5 blocking calls after the first
            "api_first_call_ms": api_first_ms,
            "api_values": res.values.tolist(),
            "roofline": {
                "bound": "valu",
                "achieved": valu_achieved / 1e12,
                "peak": VALU_PEAK_LANEOPS / 1e12,
                "unit": "Tlane-op/s",
                "frac": valu_achieved / VALU_PEAK_LANEOPS,
                # HBM bytes per launch from separate rocprofv3 --pmc passes of this same command
                # (profiles/r01_bench_n1_pmc_{fetch,write}_counters.csv): WRITE_SIZE 128 KiB (= the algorithmic
                # n_blocks*K*8 B of partial sums) + FETCH_SIZE 37.1 KiB (code objects + kernel arguments).
                "traffic": 128 * 1024 + 37.1 * 1024 if world == 1 and launch["n_blocks"] == 4096 else None,
                "kernel": "mcx_integrate_kernel",
                "measured_valu_peak": 51.5,   # Tlane-op/s sustained by v_fma_f32 (profiles/r01_valu_rates_microbench.txt)
                "kernel_ms": kernel_ms,
                "ops_per_sample": OPS_PER_SAMPLE,
                "launch": launch,
                "note": "the binding resource of this fused kernel is vector-ALU issue (SURVEY.md 8d); the HBM view of "
                        "the same launch is in roofline_hbm",
            },
            # the same kernel against the HBM roofline, in the generic schema: algorithmic bytes = K*8 B per workgroup
            "roofline_hbm": {"bound": "hbm", "achieved": hbm_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": hbm_gbps / HBM_PEAK_GBPS,
                             "traffic": 128 * 1024 + 37.1 * 1024 if world == 1 and launch["n_blocks"] == 4096 else None,
                             "algorithmic_bytes_per_launch": hbm_bytes,
                             "note": "the kernel writes one K*8-byte record per workgroup and reads only code + arguments: "
                                     "HBM is ~4e-5 of peak by design, not the bound"},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
