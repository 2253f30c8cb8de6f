"""Mutation fuzzing of libmcx's WGSL translator and payload planner (csrc/mcx_wgsl.cpp): they parse text a caller hands over, in C++.
Every WGSL string the suite knows is mutated at the token level (deletions, duplications, swaps, insertions of stray tokens, cuts) and
fed to mcx_wgsl_translate and mcx_wgsl_plan; the C++ translator must agree with the Python restatement
(tests/wgsl_reference_translator.py) on every mutant -- the same text, or a refusal with the same message -- and nothing may crash
(tools/sanitize_cpu.sh runs this file against the ASan + UBSan build of libmcx)."""
import random
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "wgpu-monte-carlo_amd"))
sys.path.insert(0, str(ROOT / "tests"))
import core_reference_planner as planner_ref  # noqa: E402
import test_wgsl_translator as corpus_mod  # noqa: E402
import wgsl_reference_translator as ref  # noqa: E402
from wgpu_montecarlo import TranspilerError, wgsl_to_hip  # noqa: E402
from wgpu_montecarlo import runtime as rt  # noqa: E402

STRAY = ["(", ")", "{", "}", ";", ",", "fn", "let", "var", "return", "x", "f32", "->", ":", "=", "+", "*", "/", "-", "1.0", "2", "0x1F", "pow",
         "select", "if", "else", "for", "while", "loop", "break", "@", ".", "[", "]", "%=", "<<=", "&&", "!", "~", "_is_wrapper_0", "_is_f_orig_0",
         "pdf_target_from_table", "mcx_x", "vec2", "true", "1e", ".5", "9999999999", "/*", "//", "$", "\n", "const", "sigma", "u32", "bool"]
TOKEN = re.compile(r"\s+|[A-Za-z_][A-Za-z_0-9]*|\d[\w.+-]*|->|<<=|>>=|[<>=!+\-*/%&|^]=|&&|\|\||<<|>>|\+\+|--|.", re.S)


def mutants(text, rng, n):
    toks = TOKEN.findall(text)
    for _ in range(n):
        t = list(toks)
        for _ in range(rng.randint(1, 3)):
            if not t:
                break
            kind, at = rng.randint(0, 5), rng.randrange(len(t))
            if kind == 0:
                del t[at]
            elif kind == 1:
                t.insert(at, t[at])
            elif kind == 2:
                other = rng.randrange(len(t))
                t[at], t[other] = t[other], t[at]
            elif kind == 3:
                t.insert(at, rng.choice(STRAY))
            elif kind == 4:
                t[at] = rng.choice(STRAY)
            else:
                t = t[:at]
        yield "".join(t)


def outcome(fn, *args):
    try:
        return ("ok", fn(*args))
    except TranspilerError as exc:
        return ("refused", str(exc))


def test_mutated_strings_translate_alike_and_nothing_crashes():
    rng = random.Random(20251004)
    texts = corpus_mod.corpus()
    checked = refused = 0
    for text in texts:
        for mutant in mutants(text, rng, 12):
            if "\x00" in mutant:
                continue
            math = rng.choice(["precise", "default", "fast"])
            want = outcome(ref.translate, mutant, 2, "user_func_2", math)
            got = outcome(wgsl_to_hip.translate, mutant, 2, "user_func_2", math)
            assert got == want, (mutant, math)
            checked += 1
            refused += got[0] == "refused"
    assert checked >= 1000 and 0.2 * checked < refused < checked       # the mutants exercise both outcomes


def test_mutated_payloads_plan_alike_and_nothing_crashes():
    import json

    rng = random.Random(7)
    payloads = json.loads((ROOT / "tests" / "golden" / "boundary_payloads.json").read_text())
    wrappers = list(payloads[2]["args"][0]["wgsl"])
    plain = list(payloads[1]["args"][0]["wgsl"])
    fields = ("weight", "p_table", "q_table", "q_sampler", "user_tables", "moment_family", "logpdf_analytic")
    checked = recognised = 0
    for base, have_t in ((wrappers, True), (plain, False)):
        for _ in range(150):
            fns = list(base)
            at = rng.randrange(len(fns))
            fns[at] = next(mutants(fns[at], rng, 1))
            if "\x00" in fns[at]:
                continue
            math = rng.choice(["precise", "default"])

            def cxx():
                src, d = rt.wgsl_plan(rt.KIND_INTEGRATE, fns, rt.DIST_NORMAL, 2.0, 3.0, math, have_t, False)
                return src, {f: getattr(d, f) for f in fields}

            want = outcome(planner_ref.plan, rt.KIND_INTEGRATE, fns, rt.DIST_NORMAL, 2.0, 3.0, math, have_t, False)
            got = outcome(cxx)
            if got[0] == want[0] == "refused":
                pass                                     # both refuse; which of a mutant's several faults is reported first may differ
            else:
                assert got == want, (fns[at], math)
            checked += 1
            recognised += got[0] == "ok" and got[1][1]["weight"] == 1
    assert checked >= 250 and recognised >= 1
