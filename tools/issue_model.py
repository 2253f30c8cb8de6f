#!/usr/bin/env python3
"""Issue model of the BASELINE kernels from the ISA of the code objects that actually run.

    python tools/issue_model.py [--tag r03] [--configs c2,c3,c4,c5] [--rng pcg_ref,philox]      # needs no GPU

For each config the product's own planner (MonteCarloIntegrator.planner: same tables, same module desc, same hiprtc as
a call on the GPU) names the code object in the cache; it is disassembled with llvm-objdump, the hot loop of the main
kernel is located, its instructions are sorted into issue classes, and the classes are priced with the per-class issue
costs measured IN a kernel on an MI355X (tools/ubench/valu_issue.hip -> profiles/r02_valu_issue_microbench.txt):

    cycles_per_unit_modelled = sum_class count_per_unit(class) x cost(class)       [SIMD-cycles per wave-unit]

bench.py prints it next to cycles_per_unit_measured = kernel_s x clock x 1024 SIMDs / (units / 64) for the module whose
cache key matches (roofline.issue_model); a kernel that changed since this file was written has another key and shows
no modelled figure rather than a stale one. Output: profiles/<tag>_issue_model.json.

Hot loop = the smallest loop (backward branch) of the kernel that holds the trip's hash multiplies (v_mul_lo_u32 for the
reference stream, v_mad_u64_u32 for Philox): table-staging loops hold none, the enclosing flush-block loop holds more
code. Loops nested INSIDE the hot loop (C5: the batched resolve of queued draws, a search that runs once per 64 flagged
draws) are priced separately with their expected frequency.
"""
import argparse
import collections
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "wgpu-monte-carlo_amd", ROOT, ROOT / "tools"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# SIMD-cycles per wave-instruction per SIMD at >= 4 waves per SIMD (profiles/r02_valu_issue_microbench.txt)
CLASS_COSTS = {
    "full": 2.2,        # v_fma/fmac/mul/add/sub_f32, v_mov, v_xor/and/or, v_add/sub_u32, v_lshrrev with VGPR / constant operands
    "half": 4.1,        # any VALU instruction with an SGPR source; v_cmp, v_cndmask, v_med3, v_min/max, v_cvt, v_mul_lo/hi, v_alignbit, ...
    "trans": 10.0,      # v_log/exp/sin/cos/sqrt/rcp/rsq_f32: 8.1 alone, ~10 inside a mixed stream ("1 v_log + 3 v_fma": 16.8 per group)
    "mad64": 4.8,       # v_mad_u64_u32 (Philox)
    "lds": 4.0,         # ds_read / ds_write / ds_bpermute issue -- ASSUMED (not in the microbenchmark): one half-rate slot
}
TRANS = re.compile(r"^v_(log|exp|sin|cos|sqrt|rcp|rsq)_f32")
HALF = re.compile(r"^v_(cmp|cmpx|cndmask|med3|min|max|cvt|mul_lo|mul_hi|mad_u32_u24|mad_i32_i24|lshl_add|add3|and_or|bfi|bfe|alignbit|"
                  r"floor|ceil|trunc|rndne|fract|ldexp|frexp|pk_|readlane|readfirstlane|writelane|mbcnt|lshl_or|xad|perm|sad|"
                  r"add_co|sub_co|addc_co|subb_co|subrev_co|div_|bcnt|ffb|cubeid|permlane)|_f64|_i64|_u64")
FULL = re.compile(r"^v_(fma|fmac|fmamk|fmaak|mul|add|sub|subrev|mac|mad)_f32|^v_mov_b32|^v_(xor|and|or|not)_b32|^v_(add|sub|subrev)_u32|"
                  r"^v_(lshrrev|lshlrev|ashrrev)_b32|^v_(add|sub|subrev)_nc_u32|^v_xnor_b32|^v_(add|sub)_i32")
SGPR_SRC = re.compile(r"^(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?|m0|ttmp\d+)$")

# samples (or MH steps) one trip of the hot loop handles, and the hash multiplies it must hold
UNITS_PER_TRIP = {("c2", "pcg_ref"): (2, 2), ("c3", "pcg_ref"): (2, 2), ("c4", "pcg_ref"): (2, 4), ("c5", "pcg_ref"): (4, 4),
                  # Philox4x32-10 = 20 products per call; those of the counter words that are 0 fold away at compile time
                  ("c2", "philox"): (4, 15), ("c3", "philox"): (4, 15), ("c4", "philox"): (2, 15), ("c5", "philox"): (4, 15)}


def classify(op, operands):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base.startswith("ds_"):
        return "lds"
    if not base.startswith("v_"):
        return None
    if base.startswith("v_mad_u64_u32"):
        return "mad64"
    if TRANS.match(base):
        return "trans"
    if HALF.search(base):
        return "half"
    srcs = operands[1:]
    if any(SGPR_SRC.match(s.strip().lstrip("-|").rstrip("|")) for s in srcs):
        return "half"
    if FULL.match(base):
        return "full"
    return "half?"                       # unknown opcode: priced as half rate and listed


def disassemble(path):
    out = subprocess.run([OBJDUMP, "-d", str(path)], capture_output=True, text=True, check=True).stdout
    kernels, cur = {}, None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
        if m and cur is not None:
            op, rest, addr = m.group(1), m.group(2), int(m.group(3), 16)
            target = None
            t = re.search(r"<\w+\+0x([0-9a-f]+)>", line)
            if t and op.startswith(("s_cbranch", "s_branch")):
                target = int(t.group(1), 16)
            cur.append(dict(op=op, operands=[o for o in rest.split(", ") if o], addr=addr, target=target))
    return kernels


def hot_loop(insts, hash_op, need):
    base = insts[0]["addr"]
    loops = []
    for i, ins in enumerate(insts):
        if ins["target"] is not None and base + ins["target"] <= ins["addr"]:
            start = next(j for j, x in enumerate(insts) if x["addr"] == base + ins["target"])
            loops.append((start, i))
    best = None
    for lo, hi in loops:
        n_hash = sum(1 for x in insts[lo:hi + 1] if x["op"].startswith(hash_op))
        if n_hash >= need and (best is None or hi - lo < best[1] - best[0]):
            best = (lo, hi)
    if best is None:
        raise SystemExit(f"no loop with {need} x {hash_op} found")
    nested = [(lo, hi) for lo, hi in loops if best[0] < lo and hi < best[1]]
    # keep the outermost nested loops only
    nested = [n for n in nested if not any(m[0] <= n[0] and n[1] <= m[1] and m != n for m in nested)]
    # a nested loop sits in a region that a forward branch of the trip skips (the resolve step runs when the queue's
    # write position crosses a block boundary): the whole skipped region is cold, not only the loop inside it
    index_of = {x["addr"]: j for j, x in enumerate(insts)}
    regions = []
    for a, b in nested:
        region = (a, b)
        for i in range(a - 1, best[0] - 1, -1):          # outwards; the wave-uniform (scc) guard is the one that skips it
            t = insts[i]["target"]
            if t is not None and base + t > insts[i]["addr"] and index_of.get(base + t, -1) > b and index_of[base + t] <= best[1]:
                region = (i + 1, index_of[base + t] - 1)
                if insts[i]["op"].startswith("s_cbranch_scc"):
                    break
        regions.append(region)
    regions = [r for r in regions if not any(m[0] <= r[0] and r[1] <= m[1] and m != r for m in regions)]
    return best, sorted(set(regions))


def count(insts, lo, hi, skip):
    classes, opcodes, salu = collections.Counter(), collections.Counter(), 0
    for i in range(lo, hi + 1):
        if any(a <= i <= b for a, b in skip):
            continue
        ins = insts[i]
        c = classify(ins["op"], ins["operands"])
        if c is None:
            if ins["op"].startswith("s_") and not ins["op"].startswith(("s_waitcnt", "s_nop")):
                salu += 1
            continue
        classes[c] += 1
        opcodes[f"{c}:{re.sub(r'_(e32|e64)$', '', ins['op'])}"] += 1
    return classes, opcodes, salu


def analyse(name, rng, module):
    kernels = disassemble(module.code_object)
    kname = "mcx_mcmc_kernel" if name == "c4" else "mcx_integrate_kernel"
    insts = kernels[kname]
    units, need = UNITS_PER_TRIP[(name, rng)]
    (lo, hi), nested = hot_loop(insts, "v_mad_u64_u32" if rng == "philox" else "v_mul_lo_u32", need)
    classes, opcodes, salu = count(insts, lo, hi, nested)
    per_unit = {c: n / units for c, n in classes.items()}
    cold = None
    if nested:
        # C5: the batched resolve. 17 % of the draws are flagged on Beta(2,5) at 8192 buckets; 64 of them
        # are resolved by one pass of the nested code, whose search loop runs ~2 steps (window of a bucket with nodes).
        freq = 0.17 / 64.0 if name == "c5" else 0.0
        cc = collections.Counter()
        for a, b in nested:
            c2, _, _ = count(insts, a, b, [])
            for k, v in c2.items():
                cc[k] += 1.5 * v                    # the region once, its search loop ~2 times
        cold = dict(frequency_per_unit=freq, classes_per_pass={k: v for k, v in cc.items()},
                    note="nested loops inside the hot loop (batched resolve of flagged draws): per-unit cost = frequency x pass")
        for k, v in cc.items():
            per_unit[k] = per_unit.get(k, 0.0) + freq * v
    cost = lambda c: CLASS_COSTS["half" if c == "half?" else c]
    modelled = sum(n * cost(c) for c, n in per_unit.items())
    valu = sum(n for c, n in per_unit.items() if c != "lds")
    # the same instructions with SURVEY.md 8(d)'s weights (lane-op equivalents: plain VALU op 1 -- an FMA is one --, integer
    # multiply 4, transcendental 2; LDS and scalar instructions are not vector-ALU work): what roofline.ops_per_unit is
    int_mul = sum(n for k, n in opcodes.items() if re.search(r":v_(mul_lo|mul_hi|mad_u64)_", k)) / units
    trans_n = per_unit.get("trans", 0.0)
    survey_ops = (valu - int_mul - trans_n) + 4.0 * int_mul + 2.0 * trans_n
    return dict(config=name, rng=rng, kernel=kname, code_object=module.code_object.name,
                hot_loop=dict(first=f"+0x{insts[lo]['addr'] - insts[0]['addr']:x}", last=f"+0x{insts[hi]['addr'] - insts[0]['addr']:x}",
                              instructions=hi - lo + 1, units_per_trip=units),
                classes_per_trip=dict(classes), salu_per_trip=salu, classes_per_unit=per_unit, valu_per_unit=valu,
                cold=cold, unknown_opcodes=sorted(k for k in opcodes if k.startswith("half?")),
                opcodes_per_trip=dict(sorted(opcodes.items())),
                int_mul_per_unit=int_mul, survey_weighted_ops_per_unit=survey_ops,
                cycles_per_unit_modelled=modelled)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--configs", default="c2,c3,c4,c5")
    ap.add_argument("--rng", default="pcg_ref,philox")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import baseline_configs as bc
    from wgpu_montecarlo import Distribution, MonteCarloIntegrator

    modules = {}
    for rng in args.rng.split(","):
        mc = MonteCarloIntegrator.planner(rng=rng)
        for name in args.configs.split(","):
            wl = bc.get(name, Distribution)
            mod = wl.prepare(mc)._plan.module
            modules[mod.key] = analyse(name, rng, mod)
            e = modules[mod.key]
            print(f"{name} {rng:8s} {mod.key}  loop {e['hot_loop']['instructions']:4d} instr / {e['hot_loop']['units_per_trip']} units  "
                  f"VALU/unit {e['valu_per_unit']:.2f}  classes/unit {{{', '.join(f'{k}: {v:.2f}' for k, v in sorted(e['classes_per_unit'].items()))}}}  "
                  f"modelled {e['cycles_per_unit_modelled']:.1f} cycles/unit  survey-weighted ops/unit {e['survey_weighted_ops_per_unit']:.2f}" + (f"  unknown {e['unknown_opcodes']}" if e["unknown_opcodes"] else ""))
    out = Path(args.out) if args.out else ROOT / "profiles" / f"{args.tag}_issue_model.json"
    out.write_text(json.dumps(dict(
        what="instruction classes of the hot loop of each BASELINE module (ISA of the cached code object, llvm-objdump) priced "
             "with the in-kernel issue costs of profiles/r02_valu_issue_microbench.txt; tools/issue_model.py",
        class_costs=CLASS_COSTS, class_costs_source="profiles/r02_valu_issue_microbench.txt ('lds' assumed)",
        unit="SIMD-cycles per wave-unit (unit = sample, or MH step for c4)", modules=modules), indent=1) + "\n")
    print("->", out)


if __name__ == "__main__":
    main()
