"""The C ABI used from plain C (no Python, no torch in the process): compile tests/cabi_client.c with gcc and run it."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "wgpu-monte-carlo_amd" / "wgpu_montecarlo" / "libmcx.so"


def _build(tmp_path):
    exe = tmp_path / "cabi_client"
    subprocess.run(["gcc", "-O2", "-I", str(ROOT / "include"), str(ROOT / "tests" / "cabi_client.c"), "-o", str(exe),
                    "-ldl", "-lm"], check=True, capture_output=True, text=True)
    return exe


def test_c_client_planning_without_gpu(tmp_path):
    res = subprocess.run([str(_build(tmp_path)), str(LIB)], capture_output=True, text=True, timeout=120)
    assert res.returncode in (0, 3), res.stdout + res.stderr
    assert "OK planning T=65536 L=16" in res.stdout


@pytest.mark.gpu
def test_c_client_end_to_end(tmp_path):
    res = subprocess.run([str(_build(tmp_path)), str(LIB)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "OK integrate n_eff=100007936" in res.stdout and "OK errors" in res.stdout
