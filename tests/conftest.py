"""pytest configuration: `gpu` marker, import paths for the product package and the oracle."""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "wgpu-monte-carlo_amd", ROOT):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def _ensure_native_pieces():
    """A fresh checkout has no libmcx.so / liboracle.so (build artefacts are not in git): build them once."""
    import subprocess

    lib = ROOT / "wgpu-monte-carlo_amd" / "wgpu_montecarlo" / "libmcx.so"
    if not lib.exists():
        subprocess.run(["make", "-C", str(ROOT / "wgpu-monte-carlo_amd" / "csrc")], check=True, capture_output=True)
    if not (ROOT / "oracle" / "liboracle.so").exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "liboracle.so"], check=True, capture_output=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_native_pieces()


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(scope="session")
def integrator():
    """One engine for the whole GPU session (module + table caches stay warm)."""
    from wgpu_montecarlo import MonteCarloIntegrator

    return MonteCarloIntegrator()
