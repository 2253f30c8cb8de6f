// mcx_plan.cpp -- pure host planning: logical dispatch geometry (bit-exact with the reference),
// multi-GPU shard ranges, physical launch geometry, table analysis. No HIP calls in this file.
//
// Reference (file:line into /root/reference):
//   calculate_dispatch_config        src/engine.rs:157-181
//   calculate_mcmc_dispatch_config   src/engine.rs:821-832,  setup_mcmc chain padding :860-866
#include "mcx_internal.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

extern "C" {

int mcx_dispatch_config(uint64_t n_samples, int64_t target_threads, mcx_dispatch* out) {
    if (!out) return mcx::fail(MCX_E_INVALID, "mcx_dispatch_config: out is null");
    if (target_threads > 0xFFFFFFFFll) return mcx::fail(MCX_E_INVALID, "target_threads does not fit u32");
    // engine.rs:164-168 -- u32 arithmetic (target + 255 may not wrap for sane inputs; keep u32 like the reference)
    uint32_t target = target_threads > 0 ? (uint32_t)target_threads : 65536u;
    const uint32_t wg = 256u;
    uint32_t wgc = (target + wg - 1u) / wg;
    uint32_t total = wgc * wg;
    if (total == 0u) return mcx::fail(MCX_E_INVALID, "target_threads rounds to zero threads");
    // engine.rs:172-173 -- u64 ceiling division, truncating cast to u32
    uint32_t loops = (uint32_t)((n_samples + (uint64_t)total - 1ull) / (uint64_t)total);
    out->workgroup_size = wg;
    out->workgroup_count = wgc;
    out->loops_per_thread = loops;
    out->total_threads = total;
    return MCX_OK;
}

int mcx_mcmc_dispatch_config(uint32_t n_chains, int64_t target_threads, mcx_dispatch* out) {
    if (!out) return mcx::fail(MCX_E_INVALID, "mcx_mcmc_dispatch_config: out is null");
    if (target_threads > 0xFFFFFFFFll) return mcx::fail(MCX_E_INVALID, "target_threads does not fit u32");
    uint32_t chains = target_threads > 0 ? (uint32_t)target_threads : n_chains;   // engine.rs:860
    const uint32_t wg = 256u;
    uint32_t wgc = (chains + wg - 1u) / wg;                                       // engine.rs:823
    out->workgroup_size = wg;
    out->workgroup_count = wgc;
    out->loops_per_thread = 1u;                                                   // engine.rs:829
    out->total_threads = wgc * wg;
    return MCX_OK;
}

int mcx_shard_units(const mcx_dispatch* d, uint32_t iterations_per_unit, uint32_t rank, uint32_t world, mcx_shard* out) {
    if (!d || !out) return mcx::fail(MCX_E_INVALID, "mcx_shard_units: null argument");
    if (world == 0u || rank >= world) return mcx::fail(MCX_E_INVALID, "mcx_shard_units: rank/world out of range");
    if (iterations_per_unit == 0u) return mcx::fail(MCX_E_INVALID, "mcx_shard_units: iterations_per_unit must be positive");
    const uint64_t L = d->loops_per_thread;
    const uint64_t units = (L + iterations_per_unit - 1ull) / iterations_per_unit;
    if (units >= (uint64_t)world) {
        // split the iteration axis: every rank sees every logical idx, a contiguous unit range
        out->idx_begin = 0u;
        out->idx_count = d->total_threads;
        out->unit_begin = (uint32_t)(units * rank / world);
        out->unit_end = (uint32_t)(units * (rank + 1ull) / world);
    } else {
        // fewer units than ranks: split the idx axis in whole reference workgroups of 256
        const uint64_t wgc = d->workgroup_count;
        uint32_t b0 = (uint32_t)(wgc * rank / world), b1 = (uint32_t)(wgc * (rank + 1ull) / world);
        out->idx_begin = b0 * 256u;
        out->idx_count = (b1 - b0) * 256u;
        out->unit_begin = 0u;
        out->unit_end = (uint32_t)units;
    }
    return MCX_OK;
}

int mcx_shard_integrate(const mcx_dispatch* d, int dist_type, uint32_t rank, uint32_t world, mcx_shard* out) {
    // units per logical thread: iterations, or Box-Muller pairs for the normal sampler
    return mcx_shard_units(d, dist_type == MCX_DIST_NORMAL ? 2u : 1u, rank, world, out);
}

int mcx_shard_chains(uint32_t total_chains, uint32_t rank, uint32_t world,
                     uint32_t* chain_begin, uint32_t* chain_count) {
    if (!chain_begin || !chain_count) return mcx::fail(MCX_E_INVALID, "mcx_shard_chains: null argument");
    if (world == 0u || rank >= world) return mcx::fail(MCX_E_INVALID, "mcx_shard_chains: rank/world out of range");
    if (total_chains % 256u) return mcx::fail(MCX_E_INVALID, "mcx_shard_chains: total_chains must be a multiple of 256");
    const uint64_t wgc = total_chains / 256u;
    uint32_t b0 = (uint32_t)(wgc * rank / world), b1 = (uint32_t)(wgc * (rank + 1ull) / world);
    *chain_begin = b0 * 256u;
    *chain_count = (b1 - b0) * 256u;
    return MCX_OK;
}

// Workgroup size for an MCMC launch of `chains` chains on one GPU (one chain per thread, so the chain count IS the
// parallelism). 1024-thread workgroups share one LDS copy of the tables between 16 waves and are best when every CU
// gets several of them; a rank's share of a chain-sharded run (BASELINE C4 over 8 GPUs: 131 072 chains) would leave
// half of the 256 CUs without any workgroup at that size. Measured on C4's kernel (profiles/
// r02_mcmc_block_size_vs_chain_count.txt, ms at 256 / 512 / 1024 threads): 1 048 576 chains 9.40 / 9.33 / 9.29;
// 524 288: 5.27 / 5.23 / 5.31; 262 144: 2.93 / 2.95 / 2.90; 131 072: 1.84 / 1.88 / 2.78; 65 536: 1.42 / 1.78 / 2.71.
// Rule: the largest size that still yields >= 4 workgroups per CU, else 256.
uint32_t mcx_mcmc_block_hint(uint32_t chains) {
    for (uint32_t block : {1024u, 512u}) {
        if ((uint64_t)chains / block >= 4ull * 256ull) return block;
    }
    return 256u;
}

}  // extern "C"

namespace mcx {

// Physical geometry for K1/K2: cut each logical thread's unit range into n_chunks so that about
// `target_phys` physical threads exist (>= 16 waves per CU on 256 CUs), never more chunks than units.
LaunchPlan plan_integrate(const mcx_shard& s, uint32_t target_phys, uint32_t block) {
    LaunchPlan p{};
    const uint64_t units = s.unit_end > s.unit_begin ? (uint64_t)(s.unit_end - s.unit_begin) : 0ull;
    if (units == 0ull || s.idx_count == 0u) return p;    // empty shard
    uint64_t want = ((uint64_t)target_phys + s.idx_count - 1ull) / s.idx_count;
    if (want < 1ull) want = 1ull;
    if (want > units) want = units;
    uint64_t upc = (units + want - 1ull) / want;
    uint64_t n_chunks = (units + upc - 1ull) / upc;
    p.units_per_chunk = (uint32_t)upc;
    p.n_chunks = (uint32_t)n_chunks;
    uint64_t threads = (uint64_t)s.idx_count * n_chunks;
    p.n_blocks = (uint32_t)((threads + block - 1ull) / block);
    return p;
}

// Analyse a table on the host: uniform-grid scale for {x, value} tables, guide table for CDFs.
void analyse_table(int kind, const float* keys, uint32_t n, float* inv_dk, std::vector<uint32_t>* guide,
                   uint32_t* guide_bits) {
    *inv_dk = 0.0f;
    *guide_bits = 0u;
    guide->clear();
    if (n < 2u) return;
    if (kind == MCX_TABLE_PDF || kind == MCX_TABLE_LOGPDF) {
        const double k0 = keys[0], span = (double)keys[n - 1u] - k0;
        if (!(span > 0.0) || !std::isfinite(span)) return;
        const double dk = span / (double)(n - 1u);
        bool uniform = true;
        for (uint32_t i = 0; i < n && uniform; ++i)
            uniform = std::fabs((double)keys[i] - (k0 + dk * (double)i)) <= 0.25 * dk;
        if (uniform) *inv_dk = (float)(1.0 / dk);
        return;
    }
    // CDF: the guide is only valid where the reference's 12-step search is an exact lower bound
    // (n <= 4096, distribution.rs:133) and the keys are non-decreasing.
    if (n > 4096u) return;
    for (uint32_t i = 1; i < n; ++i)
        if (!(keys[i] >= keys[i - 1u])) return;
    // G = 4 * pow2ceil(n) buckets (<= 8192: 32 KiB of LDS): on a smooth CDF most buckets hold <= 1 key,
    // and the longest window inside a wave shrinks (Beta(2,5), n = 2048: mean max-over-64-lanes search
    // length 3.6 steps at G = n, 1.9 at G = 4n).
    uint32_t bits = 0u;
    while ((1u << bits) < n) ++bits;
    uint32_t extra = 2u;
    if (const char* env = getenv("MCX_GUIDE_EXTRA_BITS")) extra = (uint32_t)atoi(env);      // tuning knob
    bits += extra;
    if (bits > 13u) bits = 13u;
    const uint32_t G = 1u << bits;
    std::vector<uint32_t> bound(G + 1u);
    uint32_t i = 0u;
    for (uint32_t b = 0; b <= G; ++b) {
        const float q = (float)b / (float)G;
        // first i in [0, n-2] with key[i] >= q, else n-1 (the reference never tests index n-1)
        while (i < n - 1u && keys[i] < q) ++i;
        bound[b] = i;
    }
    guide->resize(G);
    for (uint32_t b = 0; b < G; ++b) (*guide)[b] = bound[b] | (bound[b + 1u] << 16);
    *guide_bits = bits;
}

// Bucket-direct inverse CDF. The inverse of a piecewise-linear CDF is piecewise linear in u with kinks at the table's
// cdf values; cut [0, 1) into G = 2^bits equal buckets (bucket of a draw = the top `bits` bits of its hash word).
// A bucket with NO cdf node inside lies in ONE cell c of the table -- the very cell the reference's lower-bound search
// selects for every u of the bucket (distribution.rs:128-158) -- so there x(u) = x_b + s * (u - b/G) with
// x_b = x[c] + s (b/G - cdf[c]), s = (x[c+1] - x[c]) / (cdf[c+1] - cdf[c]): the record {x_b, s * 2^-32} turns the
// sample into one 8-byte read and one FMA on the low hash bits. A bucket WITH nodes stores its search window
// {lo | hi << 16, -0.0f}: lo / hi = the lower bounds of b/G and (b+1)/G, between which the reference's search result
// must lie; the sign bit of the second word is the flag (a non-negative slope never sets it). With G >= 4 n most of
// the probability mass falls into node-free buckets (Beta(2,5), n = 2048, G = 8192: 83 %).
// Not built (direct stays empty) for tables the guide is not valid for, or with a decreasing x column.
void build_cdf_direct(const float* cdf, const float* x, uint32_t n, std::vector<float>* direct, uint32_t* direct_bits) {
    direct->clear();
    *direct_bits = 0u;
    if (n < 2u || n > 4096u) return;                     // windows are packed in 16 bits; 12-step search exact for n <= 4096
    for (uint32_t i = 1; i < n; ++i)
        if (!(cdf[i] >= cdf[i - 1u]) || !(x[i] >= x[i - 1u])) return;
    if (!(cdf[0] >= 0.0f) || !(cdf[n - 1u] <= 1.0f)) return;
    uint32_t bits = 0u;
    while ((1u << bits) < n) ++bits;
    uint32_t extra = 2u;
    if (const char* env = getenv("MCX_DIRECT_EXTRA_BITS")) extra = (uint32_t)atoi(env);      // tuning knob
    bits += extra;
    if (bits > 13u) bits = 13u;                          // 8192 records = 64 KiB of LDS
    if (bits < 4u) bits = 4u;
    const uint32_t G = 1u << bits;
    std::vector<uint32_t> bound(G + 1u);
    uint32_t i = 0u;
    for (uint32_t b = 0; b <= G; ++b) {
        const float q = (float)b / (float)G;
        while (i < n - 1u && cdf[i] < q) ++i;            // first i in [0, n-2] with cdf[i] >= q, else n-1
        bound[b] = i;
    }
    direct->resize(2ull * G);
    for (uint32_t b = 0; b < G; ++b) {
        const uint32_t lo = bound[b], hi = bound[b + 1u];
        float w0, w1;
        if (lo != hi) {                                  // nodes inside: search window, flagged by the sign of -0.0f
            const uint32_t packed = lo | (hi << 16);
            memcpy(&w0, &packed, 4);
            w1 = -0.0f;
        } else if (lo == 0u) {                           // the whole bucket is at or below cdf[0]: the lookup returns x[0]
            w0 = x[0];
            w1 = 0.0f;
        } else {
            const uint32_t c = lo - 1u;                  // cdf[c] < b/G and (b+1)/G <= cdf[c+1]
            const double dc = (double)cdf[c + 1u] - (double)cdf[c];
            const double s = dc > 0.0 ? ((double)x[c + 1u] - (double)x[c]) / dc : 0.0;
            w0 = (float)((double)x[c] + s * ((double)b / (double)G - (double)cdf[c]));
            w1 = (float)(s * 0x1.0p-32);
            if (!(w1 >= 0.0f) || !std::isfinite(w0) || !std::isfinite(w1)) { direct->clear(); return; }
        }
        (*direct)[2ull * b] = w0;
        (*direct)[2ull * b + 1u] = w1;
    }
    *direct_bits = bits;
}

void build_cdf_slopes(const float* cdf, const float* x, uint32_t n, std::vector<float>* slopes) {
    slopes->assign(n, 0.0f);
    for (uint32_t c = 0; c + 1u < n; ++c) {
        const float dc = cdf[c + 1u] - cdf[c];                     // f32, like the lookup it replaces
        if (dc < 1.0e-10f || !std::isfinite(dc)) continue;
        const double s = ((double)x[c + 1u] - (double)x[c]) / (double)dc;
        (*slopes)[c] = std::isfinite(s) ? (float)s : 0.0f;
    }
}

// Index map of the sentinel-padded cell array: idx = floor(x * scale + c0) clamped to [0, n]; 0 and n are the
// {outside, 0} sentinels, 1 + c is cell c. Both table ends are INSIDE (the reference interpolates at x == key[0] and
// x == key[n-1]; a CDF-sampled x lands exactly on key[n-1] about 3e-8 of the time), so the grid is shrunk by eps cells
// at either end: key[0] maps to 1 + eps and key[n-1] to n - eps. eps also has to cover the rounding of the f32 FMA
// (about n * 2^-23 cells); a lane within eps cells outside the table extends the end cell's line that far.
void cell_map(const float* keys, uint32_t n, float* scale, float* c0) {
    const double k0 = keys[0], span = (double)keys[n - 1u] - k0;
    const double eps = std::fmax(1.0 / 512.0, (double)n / 1048576.0);
    const double s = ((double)(n - 1u) - 2.0 * eps) / span;
    *scale = (float)s;
    *c0 = (float)(1.0 + eps - k0 * (double)*scale);
}

void build_cells(const float* keys, const float* values, uint32_t n, std::vector<float>* cells) {
    cells->clear();
    if (n < 2u) return;
    const double k0 = keys[0], span = (double)keys[n - 1u] - k0;
    if (!(span > 0.0) || !std::isfinite(span)) return;
    const double dk = span / (double)(n - 1u);
    // strict grid: every key within max(1e-3 cell, 1 f32 ulp of the largest key) of k0 + i*dk. The arithmetic cell
    // guess is not verified on this path, so a merely "nearly uniform" table (analyse_table's 1/4-cell test) is
    // not enough.
    const double maxabs = std::fmax(std::fabs(k0), std::fabs((double)keys[n - 1u]));
    const double tol = std::fmax(1.0e-3 * dk, maxabs * 1.1920929e-7);
    if (tol > 0.05 * dk) return;                       // cells narrower than f32 can resolve
    if (dk < 2.0e-10) return;                          // the reference's flat-segment guard (dx < 1e-10 -> left value) could fire
    for (uint32_t i = 0; i < n; ++i)
        if (!(std::fabs((double)keys[i] - (k0 + dk * (double)i)) <= tol) || !std::isfinite((double)values[i])) return;
    cells->resize(2ull * (n - 1u));
    for (uint32_t c = 0; c + 1u < n; ++c) {
        const double x0 = keys[c], x1 = keys[c + 1u], v0 = values[c], v1 = values[c + 1u];
        const double s = (v1 - v0) / (x1 - x0);
        (*cells)[2ull * c] = (float)(v0 - s * x0);
        (*cells)[2ull * c + 1u] = (float)s;
    }
}

}  // namespace mcx
