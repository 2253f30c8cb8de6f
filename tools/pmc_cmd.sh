#!/bin/bash
# SQ counters (two passes) of the mcx kernels launched by an arbitrary python command, e.g.
#   bash tools/pmc_cmd.sh beta_k4 tools/ab_block.py 4 2e9
set -u
NAME=$1; shift
OUT=gpurun_out/pmc_$NAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/a" -- python3 "$@" > "$OUT/a.out" 2> "$OUT/a.err"
echo "pass a rc=$?"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/b" -- python3 "$@" > "$OUT/b.out" 2> "$OUT/b.err"
echo "pass b rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("a", "b"):
    for f in glob.glob(f"{out}/{p}/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("mcx_") and "fold" not in r["Kernel_Name"]:
                acc[(r["Dispatch_Id"], r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"], r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in list(acc.items())[-1:]:
            print(p, k, dict(v))
    for f in glob.glob(f"{out}/{p}/*/*kernel_trace.csv"):
        rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("mcx_") and "fold" not in r["Kernel_Name"]]
        if rows:
            r = rows[-1]
            print(p, "last kernel duration us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
