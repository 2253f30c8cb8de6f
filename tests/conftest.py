"""pytest configuration: `gpu` marker, import paths for the product package and the oracle."""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "wgpu-monte-carlo_amd", ROOT):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(scope="session")
def integrator():
    """One engine for the whole GPU session (module + table caches stay warm)."""
    from wgpu_montecarlo import MonteCarloIntegrator

    return MonteCarloIntegrator()
