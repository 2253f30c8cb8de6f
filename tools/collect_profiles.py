#!/usr/bin/env python3
"""Copy the summaries tools/profile_all.sh left under gpurun_out/prof_<tag>/ into profiles/ (tracked).

    python tools/collect_profiles.py r01
"""
import csv
import glob
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles"


def one(pattern):
    hits = sorted(glob.glob(str(src / pattern)))
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return Path(hits[-1])


shutil.copy(one("bench_stats/*/*_kernel_stats.csv"), dst / f"{tag}_bench_n1_kernel_stats.csv")
shutil.copy(src / "bench_under_rocprof.json", dst / f"{tag}_bench_n1_under_rocprof.json")
shutil.copy(one("configs_stats/*/*_kernel_stats.csv"), dst / f"{tag}_configs_c1_c5_kernel_stats.csv")
shutil.copy(src / "configs_under_rocprof.jsonl", dst / f"{tag}_configs_c1_c5_under_rocprof.jsonl")
for name in ("sq", "fetch", "write", "grbm"):
    rows = [r for r in csv.DictReader(open(one(f"pmc_{name}/*/*_counter_collection.csv"))) if r["Kernel_Name"].startswith("mcx_")]
    cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value"]
    with open(dst / f"{tag}_bench_n1_pmc_{name}_counters.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=[c for c in cols if c in rows[0]], extrasaction="ignore")
        w.writeheader()
        w.writerows(rows)
bench = ROOT / "gpurun_out" / "bench_n1.json"
if bench.exists():
    shutil.copy(bench, dst / f"{tag}_bench_n1.json")
print("profiles refreshed for", tag)
