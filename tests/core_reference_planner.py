"""Test infrastructure: the planning of a `_core` payload as wgpu_montecarlo/_core.py did it in Python until it moved into libmcx
(csrc/mcx_wgsl.cpp: mcx_wgsl_plan). tests/test_wgsl_plan.py holds the C++ planner to this restatement -- same HIP text, same desc
fields -- on the reference's recorded payloads and on variants that must be refused. Not imported by the package."""
import re
from typing import Optional

import numpy as np

import wgsl_reference_translator as wgsl_ref
from wgpu_montecarlo import emit_hip, runtime

# The importance-sampling wrapper the reference generates per function i (python/wgpu_montecarlo/__init__.py:893-899, 968-974):
# `fn _is_wrapper_i(x) { let f_val = _is_f_orig_i(x); let p = <P>; let q = <Q>; return f_val * p / q; }` followed by the
# definitions of _is_pdf_p_i / _is_pdf_q_i (when analytic) and of _is_f_orig_i (the user's function, possibly with helpers).
_WRAPPER = re.compile(
    r"^\s*fn\s+_is_wrapper_(\d+)\s*\(\s*x\s*:\s*f32\s*\)\s*->\s*f32\s*\{\s*"
    r"let\s+f_val\s*=\s*_is_f_orig_\1\s*\(\s*x\s*\)\s*;\s*"
    r"let\s+p\s*=\s*(pdf_target_from_table|_is_pdf_p_\1)\s*\(\s*x\s*\)\s*;\s*"
    r"let\s+q\s*=\s*(pdf_proposal_from_table|_is_pdf_q_\1)\s*\(\s*x\s*\)\s*;\s*"
    r"return\s+f_val\s*\*\s*p\s*/\s*q\s*;\s*\}", re.S)
_FN_START = re.compile(r"\bfn\s+([A-Za-z_][A-Za-z_0-9]*)\s*\(")


def _split_weighted(functions):
    """(f_texts, p_text or None, q_text or None) when EVERY string is one of the reference's importance-sampling wrappers
    around the same p and q (None = that density is read from its table); None when any of them is anything else."""
    f_texts, p_seen, q_seen = [], set(), set()
    for text in functions:
        m = _WRAPPER.match(text) if isinstance(text, str) else None
        if m is None:
            return None
        i, p_call, q_call = m.group(1), m.group(2), m.group(3)
        rest = text[m.end():]
        # a definition starts at a `fn name(` outside every brace (as libmcx's planner walks the tokens)
        starts = [(g.start(), g.group(1)) for g in _FN_START.finditer(rest) if rest[:g.start()].count("{") == rest[:g.start()].count("}")]
        parts = {}                                  # the three named definitions; helpers stay with the one they follow
        order = []
        for j, (pos, name) in enumerate(starts):
            end = starts[j + 1][0] if j + 1 < len(starts) else len(rest)
            if name in (f"_is_pdf_p_{i}", f"_is_pdf_q_{i}", f"_is_f_orig_{i}"):
                order.append(name)
                parts[name] = rest[pos:end]
            elif order:
                parts[order[-1]] += rest[pos:end]
            else:
                return None
        if f"_is_f_orig_{i}" not in parts or rest[:starts[0][0] if starts else 0].strip():
            return None
        p_text = parts.get(f"_is_pdf_p_{i}")
        q_text = parts.get(f"_is_pdf_q_{i}")
        if (p_text is None) != (p_call == "pdf_target_from_table") or (q_text is None) != (q_call == "pdf_proposal_from_table"):
            return None
        p_seen.add(None if p_text is None else p_text.replace(f"_is_pdf_p_{i}", "_is_pdf_p").strip())
        q_seen.add(None if q_text is None else q_text.replace(f"_is_pdf_q_{i}", "_is_pdf_q").strip())
        f_texts.append(parts[f"_is_f_orig_{i}"])
    if len(p_seen) != 1 or len(q_seen) != 1:
        return None
    return f_texts, next(iter(p_seen)), next(iter(q_seen))


_MOMENT_X = re.compile(r"^\s*fn\s+\w+\s*\(\s*x\s*:\s*f32\s*\)\s*->\s*f32\s*\{\s*return\s+x\s*;\s*\}\s*$")
_MOMENT_POW = re.compile(r"^\s*fn\s+\w+\s*\(\s*x\s*:\s*f32\s*\)\s*->\s*f32\s*\{\s*return\s+pow\s*\(\s*x\s*,\s*(\d+)(?:\.0*)?\s*\)\s*;\s*\}\s*$")


def _moment_family(functions) -> bool:
    """Are the K >= 8 strings exactly the transpiler's text for x, x**2, .., x**K (`return x;`, `return pow(x, k.0);`)? Then
    the kernel accumulates the power sums of two / four samples at a time (desc.moment_family), as api._moment_family
    decides from the IR for the same workload (BASELINE configs[4])."""
    if len(functions) < 8 or not all(isinstance(t, str) for t in functions) or not _MOMENT_X.match(functions[0]):
        return False
    for i, text in enumerate(functions[1:], start=2):
        m = _MOMENT_POW.match(text)
        if not m or int(m.group(1)) != i:
            return False
    return True


def _is_normal_pdf_text(text: Optional[str], mean: float, std: float) -> bool:
    """Is `text` the closure Distribution.normal(mean, std) hands the transpiler (python/wgpu_montecarlo/__init__.py:343-347:
    exp(-0.5 z z) / (sigma sqrt_2pi), z = (x - mean) / sigma), for exactly the parameters the call samples with? Then 1/q is a
    function of the deviate the sampler already holds (desc.q_sampler), as in api.py."""
    if text is None:
        return False
    if not re.match(r"^\s*fn\s+\w+\s*\(\s*x\s*:\s*f32\s*\)\s*->\s*f32\s*\{", text) or not text.rstrip().endswith("}"):
        return False                                     # the whole definition in the generator's format, not only its body
    consts = dict(re.findall(r"const\s+(\w+)\s*:\s*f32\s*=\s*([-+0-9.eE]+)\s*;", text))
    body = re.sub(r"\s+", "", text)
    try:
        return (set(consts) == {"mean", "sigma", "sqrt_2pi"} and float(consts["mean"]) == float(mean) and float(consts["sigma"]) == float(std)
                and abs(float(consts["sqrt_2pi"]) - 2.5066282746310002) < 1e-12
                and "varz=((x-mean)/sigma);return(exp((((-0.5)*z)*z))/(sigma*sqrt_2pi));" in body)
    except ValueError:
        return False


def _analytic_logpdf(name: str, code: int, p1: float, p2: float) -> str:
    """HIP text of the reference's analytic log-density for one distribution type, generate_log_pdf_code_for_dist
    (src/shader_gen.rs:543-571): what its MH step evaluates when `_core.integrate_mcmc` is given no table. The normal
    case is `pow(z, 2.0)` in the reference's WGSL -- backend-defined for z < 0 -- and is emitted as the intended z * z."""
    a, b = repr(float(np.float32(p1))) + "f", repr(float(np.float32(p2))) + "f"
    if code == runtime.DIST_UNIFORM:
        body = f"(({a} <= x) && (x < {b})) ? -logf({b} - {a}) : -100.0f"
    elif code == runtime.DIST_NORMAL:
        body = f"-0.5f * (((x - {a}) / {b}) * ((x - {a}) / {b})) - logf({b} * 2.50662827463f)"
    elif code == runtime.DIST_EXPONENTIAL:
        body = f"(x >= 0.0f) ? logf({a}) - {a} * x : -100.0f"
    else:
        # a custom distribution without its table: the reference would emit a call to a lookup function that is not
        # bound (shader_gen.rs:566-569 / SURVEY.md App. C-8) and fail at pipeline creation
        raise RuntimeError("Failed to create MCMC pipeline: a custom distribution needs its log-PDF table")
    return f"MCX_DEV float {name}(float x) {{ return {body}; }}"




def plan(kind, functions, code, p1, p2, math, have_target, have_proposal, t_code=0, t1=0.0, t2=0.0):
    """(HIP text, dict of the structural desc fields) for one payload."""
    literal = math == "precise"
    k = len(functions)
    tr = lambda text, slot, name: wgsl_ref.translate(text, slot, name, math)
    parts = [emit_hip.prelude()]
    d = dict(weight=0, p_table=0, q_table=0, q_sampler=0, user_tables=0, moment_family=0, logpdf_analytic=0)
    if kind == runtime.KIND_INTEGRATE:
        split = None if literal else _split_weighted(functions)
        if split is not None and ((split[1] is None) != bool(have_target) or (split[2] is None) != bool(have_proposal)):
            split = None
        if split is not None:
            q_sampler = split[2] is not None and code == runtime.DIST_NORMAL and _is_normal_pdf_text(split[2], p1, p2)
            parts += [tr(t, i, f"user_func_{i}") for i, t in enumerate(split[0])]
            if split[1] is not None:
                parts.append(tr(split[1], k, "mcx_pdf_p"))
            if split[2] is not None and not q_sampler:
                parts.append(tr(split[2], k + 1, "mcx_pdf_q"))
            d.update(weight=1, p_table=int(split[1] is None), q_table=int(split[2] is None), q_sampler=int(q_sampler))
        else:
            parts += [tr(t, i, f"user_func_{i}") for i, t in enumerate(functions)]
            mask = (1 if have_target else 0) | (2 if have_proposal else 0)
            d.update(user_tables=mask, moment_family=int(not literal and mask == 0 and k <= 32 and _moment_family(functions)))
    else:
        parts += [tr(t, i, f"user_func_{i}") for i, t in enumerate(functions)]
        if not have_target:
            parts.append(_analytic_logpdf("mcx_logpdf_p", t_code, t1, t2))
            d["logpdf_analytic"] |= 1
        d["q_sampler"] = int(not literal and code == runtime.DIST_NORMAL)
        if not d["q_sampler"] and not have_proposal:
            parts.append(_analytic_logpdf("mcx_logpdf_q", code, p1, p2))
            d["logpdf_analytic"] |= 2
    return "\n\n".join(parts), d
