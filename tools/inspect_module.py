#!/usr/bin/env python3
"""Dump the translation unit of a BASELINE-style module, compile it offline with hipcc -S for gfx950 and print
register / LDS / occupancy figures plus the instruction mix of the hottest loop.

    python tools/inspect_module.py c2|c3|c4|c5 [--math fast] [--out /tmp/isa]
"""
import argparse
import collections
import importlib.util
import math
import re
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wgpu-monte-carlo_amd"))
from wgpu_montecarlo import Distribution  # noqa: E402
from wgpu_montecarlo import runtime as rt  # noqa: E402
from wgpu_montecarlo.api import _pdf_to_hip, functions_to_hip  # noqa: E402

f1 = lambda x: x
f2 = lambda x: x**2
f3 = lambda x: x**3
f4 = lambda x: x**4


def k32():
    return [lambda x, k=k: x**k for k in range(1, 33)]


def build(which, fast):
    if which == "c2":
        return functions_to_hip([f1, f2, f3, f4], fast), rt.make_desc(rt.KIND_INTEGRATE, 4, rt.DIST_NORMAL, unit_params=True)
    if which == "c3":
        return functions_to_hip([f1, f2, f3, f4], fast), rt.make_desc(rt.KIND_INTEGRATE, 4, rt.DIST_NORMAL, weight=True,
                                                                      p_table=True, cell_tables=True, q_sampler=True)
    if which == "c4":
        return functions_to_hip([f1, f2], fast), rt.make_desc(rt.KIND_MCMC, 2, rt.DIST_NORMAL, cell_tables=True, q_sampler=True)
    if which == "c5":
        return functions_to_hip(k32(), fast), rt.make_desc(rt.KIND_INTEGRATE, 32, rt.DIST_CUSTOM, moment_family=True)
    raise SystemExit("unknown module " + which)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which")
    ap.add_argument("--math", default="default")
    ap.add_argument("--out", default="/tmp/isa")
    ap.add_argument("--flags", default="")
    args = ap.parse_args()
    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    user_src, desc = build(args.which, args.math)
    text = rt.module_source(user_src, desc)
    hip = out / f"{args.which}.hip"
    hip.write_text("#include <hip/hip_runtime.h>\n" + text)
    asm = out / f"{args.which}.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-fno-slp-vectorize", "--cuda-device-only",
           "-S", str(hip), "-o", str(asm), "-Rpass-analysis=kernel-resource-usage"] + args.flags.split()
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode:
        print(proc.stderr[-3000:])
        raise SystemExit(1)
    kernel = "mcx_mcmc_kernel" if desc.kind == rt.KIND_MCMC else "mcx_integrate_kernel"
    grab = False
    for line in proc.stderr.splitlines():
        if "Function Name" in line:
            grab = kernel in line
        if grab and "remark:" in line and any(k in line for k in ("VGPRs:", "SGPRs:", "Occupancy", "LDS Size", "ScratchSize", "Spill")):
            print(line.split("remark:")[1].strip())
    # instruction mix of the innermost loops of the kernel
    lines = asm.read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    for lo, hi in sorted(loops, key=lambda t: t[1] - t[0])[-4:]:
        ops = collections.Counter(l.split()[0] for l in body[lo:hi + 1] if re.match(r"^\s+[vsdg][a-z_0-9]+", l) and not l.strip().startswith(";"))
        valu = sum(c for o, c in ops.items() if o.startswith("v_"))
        print(f"loop lines {lo}-{hi}: {hi - lo} lines, VALU {valu}, SALU {sum(c for o, c in ops.items() if o.startswith('s_'))}, "
              f"DS {sum(c for o, c in ops.items() if o.startswith('ds_'))}, "
              f"trans {sum(c for o, c in ops.items() if re.match(r'v_(log|exp|sin|cos|sqrt|rcp|rsq)_', o))}, "
              f"div_fixup {ops.get('v_div_fixup_f32', 0)}, mul_lo {ops.get('v_mul_lo_u32', 0)}")
    print("asm:", asm)


if __name__ == "__main__":
    main()
