#!/usr/bin/env python3
"""Generate tests/golden/*.json|npz from the reference's pure-Python half.

Runs ONLY in the build container (it imports /root/reference/python, which does not exist on the GPU
box). The reference's native half (_core) is absent, so only the Python layer runs: Distribution table
builders, the transpiler, and the payloads that would cross the Python -> native boundary (captured by
substituting a recorder for `_core.MonteCarloIntegrator`). Outputs are data only.

    python tools/make_golden.py
"""
import json
import math
import re
import sys
import warnings
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
sys.path.insert(0, "/root/reference/python")
sys.path.insert(0, str(GOLDEN))

with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    import wgpu_montecarlo as ref   # the REFERENCE package (pure Python; _core missing)

assert "/root/reference" in ref.__file__, ref.__file__
import corpus  # noqa: E402


def bimodal(x):
    return 0.5 * (math.exp(-0.5 * (x - 2) ** 2) + math.exp(-0.5 * (x + 2) ** 2))


def laplace(x):
    return math.exp(-abs(x)) / 2


def shifted(x):
    return math.exp(-0.5 * ((x - 40.0) / 3.0) ** 2)


def half_open(x):
    return 0.2 if 5 <= x < 10 else 0.0


def tables():
    out = {}
    d = ref.Distribution.beta(2.0, 5.0)
    out["beta25_x"], out["beta25_cdf"] = d._x_table, d._cdf_table
    x, p = d.get_or_compute_pdf_table()
    out["beta25_pdf"] = p
    _, lp = d.get_log_pdf_table()
    out["beta25_logpdf"] = lp
    d = ref.Distribution.beta(0.5, 0.5, table_size=100)       # forced up to 1000 points, inf at the ends
    out["beta_half_x"], out["beta_half_cdf"] = d._x_table, d._cdf_table
    d = ref.Distribution.from_pdf(bimodal, support=(-10, 10))
    out["bimodal_x"], out["bimodal_cdf"] = d._x_table, d._cdf_table
    out["bimodal_logpdf"] = d.get_log_pdf_table()[1]
    d = ref.Distribution.from_pdf(laplace)
    out["laplace_support"] = np.array(d.params["support"], dtype=np.float64)
    out["laplace_x"], out["laplace_cdf"] = d._x_table, d._cdf_table
    d = ref.Distribution.from_pdf(shifted, table_size=3000)
    out["shifted_support"] = np.array(d.params["support"], dtype=np.float64)
    out["shifted_cdf"] = d._cdf_table
    for name, dist in (("normal01", ref.Distribution.normal(0.0, 1.0)), ("normal_m2s3", ref.Distribution.normal(2.0, 3.0)),
                       ("uniform_m1_3", ref.Distribution.uniform(-1.0, 3.0)), ("exp2", ref.Distribution.exponential(2.0))):
        x, lp = dist.get_log_pdf_table()
        out[f"{name}_logx"], out[f"{name}_logpdf"] = x, lp
        out[f"{name}_pdf"] = dist.get_or_compute_pdf_table()[1]
    xs = np.linspace(0, 10, 512)
    d = ref.Distribution.from_pdf_table(xs, np.exp(-xs))
    out["exptable_x"], out["exptable_pdf"], out["exptable_cdf"] = d._x_table, d._pdf_table, d._cdf_table
    out["exptable_pdf_at"] = np.array([d.pdf(v) for v in (-1.0, 0.0, 0.013, 5.0, 9.999, 10.0, 10.5)])
    np.savez_compressed(GOLDEN / "distribution_tables.npz", **out)
    supports = {}
    for name, fn in (("bimodal", bimodal), ("laplace", laplace), ("shifted", shifted)):
        supports[name] = list(ref._find_support(fn))
    try:
        ref._find_support(half_open)
        supports["half_open"] = "found"
    except ValueError as exc:
        supports["half_open"] = "ValueError: " + str(exc)[:60]
    (GOLDEN / "supports.json").write_text(json.dumps(supports, indent=1))


def transpiler():
    result = {}
    for name, fn in corpus.corpus().items():
        try:
            text = ref.transpile_function(fn)
            text = re.sub(r"user_func_[0-9a-f]{8}", "user_func_XXXXXXXX", text)
            result[name] = {"ok": True, "wgsl": sort_const_runs(text)}
        except ref.TranspilerError as exc:
            result[name] = {"ok": False, "error": str(exc)}
    (GOLDEN / "transpiler_corpus.json").write_text(json.dumps(result, indent=1, sort_keys=True))


class Recorder:
    """Stands in for _core.MonteCarloIntegrator: records the positional payload of each call."""

    def __init__(self):
        self.calls = []

    def _rec(self, name, args, k):
        self.calls.append((name, args))
        return np.zeros(k, dtype=np.float32)

    def integrate(self, *a):
        return self._rec("integrate", a, len(a[0]))

    def integrate_is_tables(self, *a):
        return self._rec("integrate_is_tables", a, len(a[0]))

    def integrate_mcmc(self, *a):
        return self._rec("integrate_mcmc", a, len(a[0]))


def normalise_wgsl(text: str) -> str:
    """Make an emitted WGSL string reproducible: the transpiler names lambdas user_func_<uuid8> (transpiler.py:522)
    and emits captured constants in set-iteration order (transpiler.py:259-269). Names are numbered in order of
    first appearance; runs of consecutive `const` lines are sorted. Nothing else is touched."""
    names = {}

    def stable(m):
        return names.setdefault(m.group(0), f"user_func_{len(names):08x}")

    return sort_const_runs(re.sub(r"user_func_[0-9a-f]{8}", stable, text))


def sort_const_runs(text: str) -> str:
    out, run = [], []
    for line in text.split("\n"):
        if line.strip().startswith("const "):
            run.append(line)
            continue
        out += sorted(run)
        run = []
        out.append(line)
    out += sorted(run)
    return "\n".join(out)


def boundary_payloads():
    integ = ref.MonteCarloIntegrator.__new__(ref.MonteCarloIntegrator)
    integ._integrator = Recorder()
    integ._target_threads = None
    f1 = lambda x: x
    f2 = lambda x: x**2
    f3 = lambda x: x**3
    f4 = lambda x: x**4
    integ.integrate([f1, f2], ref.Distribution.normal(0.0, 1.0), n_samples=1_000_000)
    integ.integrate([f1, f2, f3, f4], ref.Distribution.normal(0.0, 1.0), n_samples=10**9)
    xs = np.linspace(0, 10, 512)
    integ.integrate_importance_sampling([f1, f2, f3, f4], ref.Distribution.from_pdf_table(xs, np.exp(-xs)),
                                        ref.Distribution.normal(2.0, 3.0), n_samples=10**9)
    integ.integrate_mcmc([f1, f2], ref.Distribution.from_pdf(bimodal, support=(-10, 10)),
                         ref.Distribution.normal(0.0, 2.0), n_steps=10_000, n_chains=1_048_576, n_burnin=1000)
    integ.integrate([f1, f2], ref.Distribution.beta(2.0, 5.0), n_samples=10**10)
    integ.integrate_importance_sampling([f1], ref.Distribution.normal(0.0, 1.0), ref.Distribution.normal(0.5, 1.5),
                                        n_samples=1000)
    arrays, meta = {}, []
    for ci, (name, args) in enumerate(integ._integrator.calls):
        entry = {"method": name, "args": []}
        for ai, a in enumerate(args):
            if isinstance(a, np.ndarray):
                key = f"call{ci}_arg{ai}"
                arrays[key] = a
                entry["args"].append({"array": key, "dtype": str(a.dtype), "len": int(a.shape[0])})
            elif isinstance(a, list):
                entry["args"].append({"n_functions": len(a),
                                      "uses_table_p": any("pdf_target_from_table" in s for s in a),
                                      "uses_table_q": any("pdf_proposal_from_table" in s for s in a),
                                      # the WGSL strings exactly as the reference's Python half handed them to _core
                                      "wgsl": [normalise_wgsl(s) for s in a]})
            elif isinstance(a, dict):
                entry["args"].append({k: (list(v) if isinstance(v, tuple) else v) for k, v in a.items()})
            else:
                entry["args"].append(a)
        meta.append(entry)
    (GOLDEN / "boundary_payloads.json").write_text(json.dumps(meta, indent=1))
    np.savez_compressed(GOLDEN / "boundary_payloads.npz", **arrays)


if __name__ == "__main__":
    tables()
    transpiler()
    boundary_payloads()
    print("golden fixtures written to", GOLDEN)
