#!/usr/bin/env python3
"""One line per config from a tools/run_configs.py JSONL file."""
import json
import sys

for line in open(sys.argv[1]):
    d = json.loads(line)
    if "kernel_ms" in d:
        print("%-34s kernel %8.3f ms  call %8.3f ms  %.3g %s  worst err/3sigma %.2f" % (
            d["config"][:34], d["kernel_ms"], d["call_ms"], d["throughput_kernel"], d["unit"], d["worst_err_over_3sigma"] or 0))
