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


REF_PKG = Path("/root/reference/python/wgpu_montecarlo")


@pytest.mark.skipif(not REF_PKG.is_dir(), reason="the reference checkout exists in the build container only")
def test_the_references_python_half_runs_on_this_core(tmp_path):
    """the reference's own `__init__.py` and `transpiler.py`, unmodified and read where they lie
    (symlinked into a scratch package, nothing copied into the repository), with this repo's `_core` in place of the compiled
    module. The package must import with HAS_RUST_EXTENSION true, its public integrator must reach libmcx -- on this GPU-less
    box: RuntimeError("Failed to initialize GPU: ..."), what src/lib.rs:26-28 raises -- and its transpiler must emit the text
    this package's transpile_function emits."""
    pkg = tmp_path / "wgpu_montecarlo"
    pkg.mkdir()
    for name in ("__init__.py", "transpiler.py"):
        (pkg / name).symlink_to(REF_PKG / name)
    (pkg / "_core.py").write_text(
        "import sys\n"
        f"sys.path.insert(0, {str(ROOT / 'wgpu-monte-carlo_amd')!r})\n"
        "import importlib\n"
        "_mine = importlib.import_module('mcx_pkg._core')\n"
        "MonteCarloIntegrator = _mine.MonteCarloIntegrator\n")
    # this repo's package under another name, so that `wgpu_montecarlo` is the reference's
    (tmp_path / "mcx_pkg").symlink_to(ROOT / "wgpu-monte-carlo_amd" / "wgpu_montecarlo")
    script = (
        "import sys\n"
        f"sys.path.insert(0, {str(tmp_path)!r})\n"
        "import wgpu_montecarlo as ref\n"
        f"assert ref.__file__.startswith({str(tmp_path)!r}) and ref.HAS_RUST_EXTENSION\n"
        "print('TEXT', ref.transpile_function(lambda x: x**2 + 1.0).replace(chr(10), '|'))\n"
        "try:\n"
        "    ref.MonteCarloIntegrator()\n"
        "    print('CONSTRUCTED')\n"
        "except RuntimeError as exc:\n"
        "    print('RUNTIME', exc)\n")
    (tmp_path / "drive.py").write_text(script)                      # a file: the reference's transpiler reads the lambda's source
    res = subprocess.run([sys.executable, str(tmp_path / "drive.py")], capture_output=True, text=True, timeout=300, cwd=tmp_path,
                         env={"PATH": "/usr/bin:/bin", "HOME": str(tmp_path)})
    assert res.returncode == 0, res.stdout + res.stderr
    assert "pow(x, 2.0)" in res.stdout
    import torch

    if torch.cuda.is_available():
        assert "CONSTRUCTED" in res.stdout
    else:
        assert "RUNTIME Failed to initialize GPU" in res.stdout, res.stdout
