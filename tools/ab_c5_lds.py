#!/usr/bin/env python3
"""C5 (K = 32 moments of Beta(2,5), n = 1e10): what the LDS pipe costs the one BASELINE kernel with non-issue stalls
(round 2: SQ_WAIT_ANY / SQ_WAVE_CYCLES 0.39, LDS array 34 % busy, 46 % of those cycles bank conflicts).

Variants (compile-time switches of device/mcx_kernels.hpp, handed to libmcx through MCX_EXTRA_DEFINES):
  bpermute_flush   the wave flush as shipped: 6 ds_bpermute_b32 per row + 32 dependent LDS read-modify-writes by lane 0
  dpp_flush        DPP row reductions on the vector ALU, 16 lanes update the f64 slots at once
  dpp_soa_2xb32    + bucket-direct records as two 4-byte planes, two ds_read_b32 per draw
  dpp_soa_read2    + the same planes read by one ds_read2st64_b32
  dpp_flush512     + f32 accumulators folded every 512 units instead of 256

For each: kernel time of `bench.py --config c5` (main + fold, HIP events, 2 runs interleaved with the others) and two
rocprofv3 --pmc passes over the same command (instruction counts; LDS bank conflicts / busy cycles / waits).
Run on the GPU box from the repo root:   python3 tools/ab_c5_lds.py > gpurun_out/r03_c5_lds_variants.txt
This process never touches the GPU: every measurement is a child process.

The variants lost (profiles/r03_c5_lds_variants.txt) and their code paths (MCX_FLUSH_DPP, MCX_DIRECT_SOA) were removed from
device/mcx_kernels.hpp afterwards: to re-run this comparison check out commit 9b35320, where they exist.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
VARIANTS = [("bpermute_flush", "MCX_FLUSH_DPP=0"), ("dpp_flush", "MCX_FLUSH_DPP=1"), ("dpp_soa_2xb32", "MCX_FLUSH_DPP=1;MCX_DIRECT_SOA=1"),
            ("dpp_soa_read2", "MCX_FLUSH_DPP=1;MCX_DIRECT_SOA=2"), ("dpp_flush512", "MCX_FLUSH_DPP=1;MCX_FLUSH=512")]
BENCH = ["bench.py", "--config", "c5", "--no-cpu-baseline", "--no-cold", "--no-philox", "--legs", "none"]
PASSES = [["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS",
           "SQ_BUSY_CYCLES"],
          ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]]
UNITS = 10_000_007_168


def env_for(defs):
    env = dict(os.environ, TMPDIR="/tmp")
    if defs:
        env["MCX_EXTRA_DEFINES"] = defs
    else:
        env.pop("MCX_EXTRA_DEFINES", None)
    return env


def timed(defs, steps=10):
    res = subprocess.run([sys.executable] + BENCH + ["--steps", str(steps), "--warmup", "3"], cwd=ROOT, env=env_for(defs),
                         capture_output=True, text=True, timeout=600)
    for line in reversed(res.stdout.splitlines()):
        if line.startswith("{"):
            d = json.loads(line)
            return d["roofline"]["kernel_ms"], d["worst_err_over_3sigma"]
    raise SystemExit(f"bench failed for {defs!r}: {res.stderr[-800:]}")


def counters(defs):
    out = {}
    for names in PASSES:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + names + ["--output-format", "csv", "-d", tmp, "--", "python3"] + BENCH +
                           ["--steps", "3", "--warmup", "1"], cwd=ROOT, env=env_for(defs), capture_output=True, text=True, timeout=900)
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            for f in glob.glob(f"{tmp}/**/*_counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Kernel_Name"].startswith("mcx_integrate_kernel"):
                        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            for c in names:
                vals = [d[c] for d in per.values() if c in d]
                if vals:
                    out[c] = sum(vals) / len(vals)
    return out


def main():
    times = collections.defaultdict(list)
    for _ in range(2):                               # interleaved: box drift hits every variant alike
        for name, defs in VARIANTS:
            ms, err = timed(defs)
            times[name].append(ms)
            print(f"# {name:16s} kernel_ms {ms:.3f}  worst_err_over_3sigma {err:.2f}", flush=True)
    for name, defs in VARIANTS:
        c = counters(defs)
        wave_units = UNITS / 64.0
        row = dict(variant=name, defines=defs, kernel_ms=[round(t, 3) for t in times[name]], kernel_ms_best=round(min(times[name]), 3))
        if c.get("SQ_INSTS_VALU"):
            row.update(valu_per_sample=round(c["SQ_INSTS_VALU"] / wave_units, 2), lds_inst_per_sample=round(c.get("SQ_INSTS_LDS", 0) / wave_units, 3),
                       salu_per_sample=round(c.get("SQ_INSTS_SALU", 0) / wave_units, 2),
                       wait_any_over_wave_cycles=round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3),
                       wait_inst_any_over_wave_cycles=round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3))
        if c.get("SQ_LDS_IDX_ACTIVE"):
            row.update(lds_bank_conflict_cycles=c.get("SQ_LDS_BANK_CONFLICT"), lds_idx_active_cycles=c["SQ_LDS_IDX_ACTIVE"],
                       lds_bank_conflict_share=round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 3),
                       lds_array_cycles_per_wave_sample=round(c["SQ_LDS_IDX_ACTIVE"] / wave_units, 2))
            if c.get("GRBM_GUI_ACTIVE"):
                row["lds_array_busy_frac"] = round(c["SQ_LDS_IDX_ACTIVE"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 256.0), 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
