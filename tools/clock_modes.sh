#!/bin/bash
# Same box, same session: effective shader clock (GRBM_GUI_ACTIVE / 8 / duration) of the C2 kernel when launches are
# queued back to back (bench.py) and when every call is blocking (tools/call_overhead.py).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/clk_async gpurun_out/clk_block
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/clk_async -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --streams 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/clk_block -- python3 tools/call_overhead.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for name in ("clk_async", "clk_block"):
    f = glob.glob(f"gpurun_out/{name}/*/*kernel_trace.csv")[0]
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if r["Kernel_Name"] == "mcx_integrate_kernel"}
    g = glob.glob(f"gpurun_out/{name}/*/*counter_collection.csv")[0]
    clk = []
    for r in csv.DictReader(open(g)):
        if r["Kernel_Name"] == "mcx_integrate_kernel" and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and dur.get(r["Dispatch_Id"], 0) > 300000:
            clk.append((dur[r["Dispatch_Id"]] / 1e3, float(r["Counter_Value"]) / 8 / dur[r["Dispatch_Id"]]))
    clk.sort()
    d = [c[0] for c in clk]; c = [c[1] for c in clk]
    print(name, "n=1e9 launches", len(clk), "duration us min/median/max %.1f %.1f %.1f" % (min(d), d[len(d)//2], max(d)),
          "clock GHz min/median/max %.3f %.3f %.3f" % (min(c), sorted(c)[len(c)//2], max(c)))
PY
