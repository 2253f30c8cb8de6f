"""The C ABI used from plain C (no Python, no torch in the process): compile tests/cabi_client.c with gcc and run it."""
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "wgpu-monte-carlo_amd" / "wgpu_montecarlo" / "libmcx.so"


def _build(tmp_path, old_layout=False):
    exe = tmp_path / ("cabi_client_old" if old_layout else "cabi_client")
    subprocess.run(["gcc", "-O2", "-Wall", "-I", str(ROOT / "include"), str(ROOT / "tests" / "cabi_client.c"), "-o", str(exe),
                    "-ldl", "-lm"] + (["-DMCX_CLIENT_OLD_LAYOUT"] if old_layout else []), check=True, capture_output=True, text=True)
    return exe


@pytest.mark.parametrize("old_layout", [False, True])
def test_c_client_planning_without_gpu(tmp_path, old_layout):
    """Planning + the versioned-struct contract (include/mcx.h "ABI versioning"): the caller's own layout is accepted --
    also the shorter one of a client built against the first release of mcx_module_desc --, a longer or an
    uninitialised one is refused."""
    res = subprocess.run([str(_build(tmp_path, old_layout)), str(LIB)], capture_output=True, text=True, timeout=120)
    assert res.returncode in (0, 3), res.stdout + res.stderr
    assert "OK planning T=65536 L=16" in res.stdout and "OK wgsl translated" in res.stdout
    m = re.search(r"OK abi version 4, desc of (\d+) bytes \(library: (\d+)\)", res.stdout)
    assert m, res.stdout
    mine, libs = int(m.group(1)), int(m.group(2))
    assert (mine == 52 and libs > mine) if old_layout else (mine == libs), res.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("old_layout", [False, True])
def test_c_client_end_to_end(tmp_path, old_layout):
    """The same sums whether the client knows the whole module desc or only its first release (what a binding compiled
    against an older mcx.h hands over): libmcx zero-fills the fields such a caller does not know."""
    res = subprocess.run([str(_build(tmp_path, old_layout)), str(LIB)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "OK integrate n_eff=100007936" in res.stdout and "OK errors" in res.stdout
    assert old_layout or ("OK fitted importance sampling: cells 1" in res.stdout and "OK planned from the reference's wrapper text" in res.stdout
                          and "OK mcx_core: integrate_is_tables" in res.stdout), res.stdout
