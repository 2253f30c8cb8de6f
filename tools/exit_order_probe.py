#!/usr/bin/env python3
"""Process-exit probe for the run-time bound libraries (GPU box): create an engine and an RCCL communicator through
libmcx, import torch before / after / not at all, leave the communicator open or not, and exit. Every mode must end
with exit code 0: PyTorch's bundled librccl has to be loaded by PyTorch's own loader (runtime.Comm imports torch first)
and communicators have to be destroyed before their engines and before interpreter teardown (runtime._close_comms).

    python tools/exit_order_probe.py plain | torch_first | torch_after | torch_after_cuda | torch_after_noclose | torch_first_noclose
"""
import sys, os
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT / "wgpu-monte-carlo_amd"), str(ROOT)]
mode = sys.argv[1]
if mode.startswith("torch_first"):
    import torch
from wgpu_montecarlo import runtime as rt
print("rccl env", os.environ.get("MCX_RCCL"), flush=True)
eng = rt.Engine(0)
if "nocomm" not in mode:
    comm = rt.Comm([eng])
    print("rccl lib", rt.rccl_library(), flush=True)
    if "noclose" not in mode:
        comm.close()
if mode.startswith("torch_after"):
    import torch
    if "cuda" in mode:
        torch.zeros(4, device="cuda").sum().item()
eng.close()
print("done", mode, flush=True)
