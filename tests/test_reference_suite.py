"""The reference's OWN test-suite, unmodified and read where it lies (/root/reference/tests), run against this package's
Python layer through the `wgpu_montecarlo` import path -- build container only: /root/reference does not travel to the GPU
box, so the test skips itself there. Nothing of the reference is copied into the repository.

Without a GPU every test that needs the native half must fail in exactly the place the reference would -- constructing
the integrator raises RuntimeError("Failed to initialize GPU: ...") (src/lib.rs:26-28) -- and every test that does not
(the whole transpiler suite, Distribution construction / tables / validation) must pass. With a GPU in the container all
of them would be expected to pass; that run is what tests/test_gpu_*.py restate scenario by scenario for the GPU box."""
import subprocess
import sys
import xml.etree.ElementTree as ET
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF_TESTS = Path("/root/reference/tests")


@pytest.mark.skipif(not REF_TESTS.is_dir(), reason="the reference checkout exists in the build container only")
def test_reference_suite_against_this_package(tmp_path):
    report = tmp_path / "reference_suite.xml"
    env = {"PYTHONPATH": str(ROOT / "wgpu-monte-carlo_amd"), "PATH": "/usr/bin:/bin", "HOME": str(tmp_path)}
    res = subprocess.run([sys.executable, "-m", "pytest", str(REF_TESTS), "-q", "-p", "no:cacheprovider", f"--junitxml={report}",
                          "-W", "ignore"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert report.exists(), res.stdout[-2000:] + res.stderr[-2000:]
    passed, gpu_needed, other = [], [], []
    for case in ET.parse(report).getroot().iter("testcase"):
        name = f"{case.get('classname')}::{case.get('name')}"
        bad = [el for el in case if el.tag in ("failure", "error")]
        if not bad:
            if not any(el.tag == "skipped" for el in case):
                passed.append(name)
        elif all("Failed to initialize GPU" in ((el.get("message") or "") + (el.text or "")) for el in bad):
            gpu_needed.append(name)
        else:
            other.append((name, (bad[0].get("message") or "")[:200]))
    assert not other, other                                       # a failure for any reason but "no GPU here" is a parity bug
    transpiler = [n for n in passed if "test_transpiler" in n]
    assert len(transpiler) == 66, len(transpiler)                 # tests/test_transpiler.py: all of it
    assert len(passed) >= 87 and len(passed) + len(gpu_needed) >= 160, (len(passed), len(gpu_needed))
    import torch

    if not torch.cuda.is_available():
        assert gpu_needed, "without a GPU the integration tests must stop at 'Failed to initialize GPU'"
