#!/usr/bin/env python3
"""Condense what tools/profile_all.sh left under gpurun_out/prof_<tag>/ into small tracked files.

    python tools/collect_profiles.py r02                 # -> profiles/
    python tools/collect_profiles.py r02 --into DIR      # (profile_all.sh uses this on the GPU box)

Per config cX: <tag>_cX_kernel_stats.csv (rocprofv3 --stats summary), <tag>_cX_under_rocprof.json (the bench line of
that profiled run), <tag>_cX_pmc.csv (every counter of every pass, summed over the chip per dispatch and averaged over
the dispatches of the main kernel), and <tag>_pmc_summary.txt: per-unit derived figures -- VALU instructions per
sample / MH step, LDS instructions, bank-conflict share of the LDS-active cycles, busy fractions, HBM bytes per call.

A call may be several launches of the main kernel (a time-segmented MCMC call is S x 2 launches on two streams, folded
once): `launches_per_call` comes from the bench line of the profiled run, counters are summed per CALL (mean per
dispatch x launches per call), and `rocprof_call_span_us` is the wall span of a call's main-kernel dispatches in the
kernel trace (first start to last end between two fold kernels) -- for a one-launch call that is the launch's duration.
"""
import argparse
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ap = argparse.ArgumentParser()
ap.add_argument("tag", nargs="?", default="r03")
ap.add_argument("--into", default=str(ROOT / "profiles"))
args = ap.parse_args()
tag = args.tag
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = Path(args.into)
dst.mkdir(parents=True, exist_ok=True)
MAIN = ("mcx_integrate_kernel", "mcx_mcmc_kernel")


def latest(pattern):
    hits = sorted(glob.glob(str(src / pattern)))
    return Path(hits[-1]) if hits else None


summary = []
for cfg in ("c2", "c3", "c4", "c5"):
    stats = latest(f"{cfg}_stats/*/*_kernel_stats.csv")
    if stats is None:
        continue
    shutil.copy(stats, dst / f"{tag}_{cfg}_kernel_stats.csv")
    line = None
    bench_json = src / f"{cfg}_under_rocprof.json"
    if bench_json.exists():
        shutil.copy(bench_json, dst / f"{tag}_{cfg}_under_rocprof.json")
        for text in bench_json.read_text().splitlines():
            if text.startswith("{"):
                line = json.loads(text)
    avg_ns = None
    for r in csv.DictReader(open(stats)):
        if r["Name"].startswith(MAIN):
            avg_ns = float(r["AverageNs"])
    launches = int(((line or {}).get("roofline", {}).get("launch", {}) or {}).get("launches", 1) or 1)
    # wall span of one call in the kernel trace: the main-kernel dispatches between two fold kernels
    span_ns = None
    trace = latest(f"{cfg}_stats/*/*_kernel_trace.csv")
    if trace is not None:
        rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(trace))
                       if r["Kernel_Name"].startswith(MAIN + ("mcx_fold_kernel",))), key=lambda t: t[1])
        spans, cur = [], []
        for start, end, kname in rows:                      # ordered by END time: a call's fold ends after all of its launches
            if kname.startswith("mcx_fold_kernel"):
                if len(cur) == launches:
                    spans.append(max(e for _, e in cur) - min(b for b, _ in cur))
                cur = []
            else:
                cur.append((start, end))
        if spans:
            spans = spans[len(spans) // 4:]               # drop the warm-up quarter
            span_ns = sum(spans) / len(spans)
    counters = collections.OrderedDict()
    meta = {}
    gui_ns = {}
    for p in ("a", "b", "fetch", "write"):
        f = latest(f"{cfg}_pmc_{p}/*/*_counter_collection.csv")
        if f is None:
            continue
        per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(MAIN):
                per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":           # this pass's own duration of the dispatch (the counter passes
                    gui_ns[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])      # serialise kernels)
                meta = {k: r.get(k, "") for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count")}
        names = sorted({c for d in per_dispatch.values() for c in d})
        for c in names:
            vals = [d[c] for d in per_dispatch.values() if c in d]
            counters[c] = (sum(vals) / len(vals) * launches, len(vals), p)      # per CALL
    with open(dst / f"{tag}_{cfg}_pmc.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["config", "kernel", "grid", "workgroup", "lds_bytes", "vgprs", "counter", "mean_per_call", "dispatches", "pass"])
        for c, (v, n, p) in counters.items():
            w.writerow([cfg, meta.get("Kernel_Name", ""), meta.get("Grid_Size", ""), meta.get("Workgroup_Size", ""),
                        meta.get("LDS_Block_Size", ""), meta.get("VGPR_Count", ""), c, f"{v:.6g}", n, p])
    get = lambda c: counters[c][0] if c in counters else None
    units = line["roofline"]["units_per_launch"] if line else None
    call_ns = span_ns if (span_ns and launches > 1) else avg_ns
    launch_info = ((line or {}).get("roofline", {}).get("launch", {}) or {})
    d = dict(config=cfg, kernel=meta.get("Kernel_Name"), grid=meta.get("Grid_Size"), workgroup=meta.get("Workgroup_Size"),
             lds_bytes=meta.get("LDS_Block_Size"), vgprs=meta.get("VGPR_Count"), units_per_launch=units,
             launches_per_call=launches, segments=launch_info.get("segments", 0),
             grid_per_segment=(launch_info.get("n_blocks", 0) * launch_info.get("block", 0)) if launches > 1 else None,
             rocprof_avg_us=avg_ns / 1e3 if avg_ns else None,
             rocprof_call_span_us=call_ns / 1e3 if call_ns else None,
             bench_kernel_ms=line["roofline"]["kernel_ms"] if line else None,
             bench_roofline_frac=line["roofline"]["frac"] if line else None)
    if units:
        waves_units = units / 64.0
        if get("SQ_INSTS_VALU"):
            d["valu_inst_per_unit"] = get("SQ_INSTS_VALU") / waves_units
        if get("SQ_INSTS_LDS"):
            d["lds_inst_per_unit"] = get("SQ_INSTS_LDS") / waves_units
        if get("SQ_INSTS_SALU"):
            d["salu_inst_per_unit"] = get("SQ_INSTS_SALU") / waves_units
        if call_ns and get("SQ_INSTS_VALU"):
            # raw issue: executed wave-instructions x 64 lanes per second against 256 x 4 x 32 lanes x 2.4 GHz
            d["valu_issue_frac_of_peak"] = get("SQ_INSTS_VALU") * 64 / (call_ns * 1e-9) / (256 * 4 * 32 * 2.4e9)
    if get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share"] = get("SQ_LDS_BANK_CONFLICT") / get("SQ_LDS_IDX_ACTIVE")
    if get("SQ_BUSY_CYCLES") and get("SQ_ACTIVE_INST_LDS") is not None and get("SQ_WAVE_CYCLES"):
        d["active_inst_valu_over_wave_cycles"] = get("SQ_ACTIVE_INST_VALU") / get("SQ_WAVE_CYCLES")
        d["active_inst_lds_over_wave_cycles"] = get("SQ_ACTIVE_INST_LDS") / get("SQ_WAVE_CYCLES")
    if get("SQ_WAVE_CYCLES") and get("SQ_WAIT_INST_ANY") is not None:
        d["wait_inst_any_over_wave_cycles"] = get("SQ_WAIT_INST_ANY") / get("SQ_WAVE_CYCLES")
        d["wait_any_over_wave_cycles"] = (get("SQ_WAIT_ANY") or 0.0) / get("SQ_WAVE_CYCLES")
    if get("SQ_LDS_IDX_ACTIVE") and get("GRBM_GUI_ACTIVE"):
        # LDS-array cycles summed over CUs against the shader cycles of the launch (GRBM_GUI_ACTIVE is summed over 8 XCDs)
        # (two launches of a segmented call share the chip: the busy fraction is per launch-time, summed over what overlaps)
        d["lds_array_busy_frac"] = get("SQ_LDS_IDX_ACTIVE") / (get("GRBM_GUI_ACTIVE") / 8.0 * 256.0)
    if get("GRBM_GUI_ACTIVE") and gui_ns:
        # busy cycles of one launch (per-call counter / launches per call; summed over 8 XCDs) over the duration the SAME
        # dispatch had in that counter pass
        d["effective_clock_ghz"] = get("GRBM_GUI_ACTIVE") / launches / 8.0 / (sum(gui_ns.values()) / len(gui_ns))
    # HBM bytes per launch: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide
    # coalesced reads (MI355X_MICROARCH.md, HBM), so the read side is doubled for the upper estimate
    if get("FETCH_SIZE") is not None:
        d["fetch_kib"] = get("FETCH_SIZE")
    if get("WRITE_SIZE") is not None:
        d["write_kib"] = get("WRITE_SIZE")
    if get("FETCH_SIZE") is not None and get("WRITE_SIZE") is not None:
        d["hbm_bytes_per_launch_corrected"] = (2.0 * get("FETCH_SIZE") + get("WRITE_SIZE")) * 1024.0
    summary.append(d)

with open(dst / f"{tag}_pmc_summary.txt", "w") as fh:
    for d in summary:
        fh.write(json.dumps(d) + "\n")
print("profiles condensed for", tag, "->", dst)
