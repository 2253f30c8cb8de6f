"""Device-side integer known-answer tests: the GPU computes the raw counter / hash / Philox words with the device
functions the kernels use (mcx_selftest_streams) and they are compared BIT FOR BIT with

  * SURVEY.md App. A.2's vectors for pcg_hash / the combined counter (src/distribution.rs:62-73) -- the same vectors
    tests/test_oracle_pins.py holds the CPU oracle to,
  * a pure-Python restatement of the reference's formula on random triples,
  * the published Random123 known answers for Philox4x32-10,
  * the oracle on the same inputs.

This is what moves "the integer stream is the reference's" from implied-by-float-sums to proved on the device."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

KAT_COUNTERS = [((42, 0, 0), 42, 1223963391), ((42, 0, 1), 15485905, 1068524562), ((42, 1, 0), 7199411, 2858281791),
                ((42, 65535, 30517), 3794890804, 2007115843), ((12345, 255, 1000002), 77753790, 533615206),
                ((1000041, 7, 11000), 2892164080, 130712447), ((4294967295, 1, 1), 22685231, 1222289643)]
KAT_HASH = {0: 129708002, 1: 2831084092, 2: 2055130248, 3: 2131687100, 42: 1223963391, 4294967295: 3861530882,
            449710063: 0}
KAT_PHILOX = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
              ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
              ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
               (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]


def _py_hash(v):
    s = (v * 747796405 + 2891336453) & 0xFFFFFFFF
    w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
    return ((w >> 22) ^ w) & 0xFFFFFFFF


def test_survey_vectors_on_the_device(integrator):
    eng = integrator._engine
    triples = [t for t, _, _ in KAT_COUNTERS] + [(v, 0, 0) for v in KAT_HASH]          # seed = v, idx = iter = 0 -> combined = v
    got = eng.selftest_streams(triples, [c for c, _, _ in KAT_PHILOX], [k for _, k, _ in KAT_PHILOX])
    want_comb = [c for _, c, _ in KAT_COUNTERS] + list(KAT_HASH)
    want_hash = [h for _, _, h in KAT_COUNTERS] + list(KAT_HASH.values())
    assert got["combined"].tolist() == want_comb
    assert got["hash"].tolist() == want_hash
    assert got["stepped"].tolist() == want_hash                       # the strength-reduced (add-only) state stepping
    assert got["philox"].tolist() == [list(o) for _, _, o in KAT_PHILOX]
    # u = float(h) * 2^-32 on the closed interval: 0 only for the hash value 0 (distribution.rs:72)
    assert got["u"].tolist() == [float(np.float32(h) * np.float32(2.0**-32)) for h in want_hash]
    assert got["u"][len(KAT_COUNTERS) + list(KAT_HASH).index(449710063)] == 0.0
    assert got["u"][0] == pytest.approx(0.28497618, abs=1e-8)         # U(42, 0, 0), SURVEY App. A.2


def test_random_triples_match_the_reference_formula_and_the_oracle(integrator):
    rng = np.random.default_rng(7)
    n = 20000
    triples = np.stack([rng.integers(0, 2**32, n), rng.integers(0, 2**20, n), rng.integers(0, 2**22, n)], axis=1).astype(np.uint64)
    triples[:8] = [[0, 0, 0], [2**32 - 1, 2**20 - 1, 2**22 - 1], [42, 65535, 0], [42, 0, 2000000], [1000041, 1048575, 11000],
                   [7, 3, 1023], [7, 3, 1024], [7, 3, 1025]]
    pc = rng.integers(0, 2**32, (n, 4))
    pk = rng.integers(0, 2**32, (n, 2))
    got = integrator._engine.selftest_streams(triples, pc, pk)
    comb = (triples[:, 0] + triples[:, 1] * 7199369 + triples[:, 2] * 15485863) & 0xFFFFFFFF
    assert np.array_equal(got["combined"], comb.astype(np.uint32))
    want = np.array([_py_hash(int(c)) for c in comb], dtype=np.uint32)
    assert np.array_equal(got["hash"], want)
    assert np.array_equal(got["stepped"], want)
    # the Box-Muller angle word is the hash before its last xorshift: hash == angle ^ (angle >> 22)
    assert np.array_equal(got["angle"] ^ (got["angle"] >> np.uint32(22)), want)
    # ... and the oracle agrees on a sample of them (same C restatement the float parity tests use)
    for i in range(0, n, 997):
        s, ix, it = (int(v) for v in triples[i])
        assert oracle.combined(s, ix, it) == int(got["combined"][i]) and oracle.pcg_hash(int(comb[i])) == int(got["hash"][i])
        assert oracle.philox4x32_10(tuple(int(v) for v in pc[i]), tuple(int(v) for v in pk[i])) == tuple(int(v) for v in got["philox"][i])
