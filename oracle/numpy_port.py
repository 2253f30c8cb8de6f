"""numpy restatement of the reference's K1 kernel (normal sampler, monomial integrands), vectorised, one core.

TEST INFRASTRUCTURE ONLY, like everything under oracle/: used by tests/ (a second, independently written restatement
that the C oracle must agree with on the same counter stream) and by bench.py's `cpu_baseline_numpy` leg -- the
"benchmark.py-style" single-core figure of SURVEY.md 8(d)(ii): the reference's own CPU comparison is a Python / numpy
loop on one core (examples/benchmark.py:43-68), not a parallel port.

Follows (file:line into NightingaleCen/wgpu-monte-carlo):
  pcg_hash, random_uniform          src/distribution.rs:62-73
  sample_normal_box_muller + cache  src/distribution.rs:87-114  (even i draws counters 2i, 2i+1; odd i takes the cached z1)
  kernel body, per-thread mean      src/shader_gen.rs:93-117, 293-303
  host mean over threads            src/lib.rs:129-138 (here in f64 over f32 terms: the oracle's `sums`)
  dispatch sizing                   src/engine.rs:157-181
"""
import numpy as np

_IDX_MULT, _ITER_MULT = np.uint32(7199369), np.uint32(15485863)


def pcg_hash(v):
    """distribution.rs:62-66 on a uint32 array (wrapping arithmetic)."""
    v = np.atleast_1d(np.asarray(v, dtype=np.uint32))           # arrays wrap silently; numpy scalars would warn
    state = v * np.uint32(747796405) + np.uint32(2891336453)
    word = ((state >> ((state >> np.uint32(28)) + np.uint32(4))) ^ state) * np.uint32(277803737)
    out = (word >> np.uint32(22)) ^ word
    return out


def uniform(seed, idx, it):
    """random_uniform, distribution.rs:68-73: float(h) / 4294967295.0 in f32 (the literal rounds to 2^32)."""
    h = pcg_hash(np.uint32(seed) + np.asarray(idx, np.uint32) * _IDX_MULT + np.asarray(it, np.uint32) * _ITER_MULT)
    return h.astype(np.float32) * np.float32(2.0 ** -32)


def dispatch(n_samples, target_threads=None):
    """engine.rs:157-181: (T, L)."""
    target = int(target_threads) if target_threads else 65536
    wgc = -(-target // 256)
    t = wgc * 256
    loops = int(np.uint32(-(-int(n_samples) // t)))
    return t, loops


def normal_moments(k, n_samples, seed=42, mean=0.0, std=1.0, target_threads=None, guard=True, idx_block=4096):
    """sums[j] = sum over the whole grid of x^(j+1), x = mean + std z (f32 terms, f64 sums), and N_eff = T L.

    guard: u1 = max(float(h), 0.5) 2^-32 as libmcx does by default (h == 0 would give log(0) in the reference)."""
    t, loops = dispatch(n_samples, target_threads)
    sums = np.zeros(k, dtype=np.float64)
    pairs = (loops + 1) // 2
    j = np.arange(pairs, dtype=np.uint32)
    two_pi = np.float32(6.283185307179586)
    mean32, std32 = np.float32(mean), np.float32(std)
    for i0 in range(0, t, idx_block):
        idx = np.arange(i0, min(i0 + idx_block, t), dtype=np.uint32)[:, None]
        # pair j = iterations (2j, 2j+1): uniforms from counters 4j and 4j + 1 (iter * 2, iter * 2 + 1 with iter = 2j)
        base = np.uint32(seed) + idx * _IDX_MULT
        h1 = pcg_hash(base + (np.uint32(4) * j)[None, :] * _ITER_MULT)
        h2 = pcg_hash(base + (np.uint32(4) * j + np.uint32(1))[None, :] * _ITER_MULT)
        f1 = h1.astype(np.float32)
        if guard:
            f1 = np.maximum(f1, np.float32(0.5))
        with np.errstate(divide="ignore"):
            r = np.sqrt(np.float32(-2.0) * np.log(f1 * np.float32(2.0 ** -32)))
        theta = two_pi * (h2.astype(np.float32) * np.float32(2.0 ** -32))
        z0 = r * np.cos(theta)
        z1 = r * np.sin(theta)
        if loops & 1:
            z1[:, -1] = np.nan                       # L odd: the last pair's second half is never consumed
        for z in (z0, z1):
            x = mean32 + std32 * z
            p = x.copy()
            for q in range(k):
                sums[q] += np.nansum(p, dtype=np.float64)
                if q + 1 < k:
                    p = p * x
    return sums, t * loops
