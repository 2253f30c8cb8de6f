// mcx_device.hpp -- gfx950 (CDNA4, wave64) device library for the fused Monte-Carlo kernels.
//
// This text is compiled at run time by hiprtc (and ahead of time by hipcc for the build check);
// it is prepended to the emitted user functions and followed by mcx_kernels.hpp.
//
// Semantics restated from the reference's WGSL device library (file:line are into /root/reference):
//   pcg_hash / random_uniform          src/distribution.rs:62-73
//   sample_uniform                     src/distribution.rs:80-82
//   sample_normal_box_muller           src/distribution.rs:87-114
//   sample_exponential                 src/distribution.rs:120-124
//   sample_from_cdf_table              src/distribution.rs:128-158
//   pdf_*_from_table / log_pdf_*       src/distribution.rs:181-223, 375-417
// What is NOT kept: the WGSL text, the bind-group layout, the one-thread-per-logical-index mapping.
// The u32 counter stream (seed + idx*7199369 + iter*15485863 -> PCG-RXS-M-XS) is bit-exact.
#pragma once

typedef unsigned int u32;
typedef unsigned long long u64;

#ifndef MCX_GUARD_ENDPOINTS
#define MCX_GUARD_ENDPOINTS 1      // 1: u in (0,1] for log(), [0,1) for affine maps; 0: strict reference float(h)*2^-32
#endif
#ifndef MCX_UNIFORM_TABLES
#define MCX_UNIFORM_TABLES 0       // 1: every PDF / log-PDF table of the module is a uniform grid (host-checked)
#endif
#ifndef MCX_PRECISE_SAMPLER
#define MCX_PRECISE_SAMPLER 0      // 1: ocml logf/sinf/cosf in the samplers instead of v_log/v_sin/v_cos
#endif

#define MCX_PCG_MULT   747796405u
#define MCX_PCG_INC    2891336453u
#define MCX_PCG_OUTMUL 277803737u
#define MCX_IDX_MULT   7199369u
#define MCX_ITER_MULT  15485863u
// The LCG stage of pcg_hash is affine in the counter, so stepping `iter` by d steps the LCG
// state by d * MCX_STATE_STEP (mod 2^32): the first of the two multiplies of every hash is
// strength-reduced to one add in the hot loops.
#define MCX_STATE_STEP (MCX_ITER_MULT * MCX_PCG_MULT)

#define MCX_DIST_UNIFORM     0
#define MCX_DIST_NORMAL      1
#define MCX_DIST_EXPONENTIAL 2
#define MCX_DIST_CUSTOM      3

#define MCX_DEV __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// counter hash
// ---------------------------------------------------------------------------------------------

// LCG state of pcg_hash(seed + idx*7199369 + iter*15485863); all arithmetic wraps mod 2^32.
MCX_DEV u32 mcx_state(u32 seed, u32 idx, u32 iter) {
    u32 v = seed + idx * MCX_IDX_MULT + iter * MCX_ITER_MULT;
    return v * MCX_PCG_MULT + MCX_PCG_INC;
}

// RXS-M-XS output permutation applied to an LCG state (distribution.rs:64-65).
MCX_DEV u32 mcx_pcg_out(u32 state) {
    u32 word = ((state >> ((state >> 28u) + 4u)) ^ state) * MCX_PCG_OUTMUL;
    return (word >> 22u) ^ word;
}

// The hash output that only feeds the Box-Muller angle. With MCX_THETA_FROM_BITS the angle is the top 23 bits of
// the word (see mcx_box_muller); the final xorshift `^ (word >> 22)` changes only the lowest of those 23 bits
// (bit 9 ^= bit 31), i.e. the angle by at most 2^-23 rev, the same size as the truncation itself -- so it is not
// computed: two VALU instructions fewer per pair. MCX_PRECISE_SAMPLER / MCX_THETA_FROM_BITS=0 use the full hash.
#ifndef MCX_THETA_FROM_BITS
#define MCX_THETA_FROM_BITS 1
#endif
MCX_DEV u32 mcx_pcg_angle(u32 state) {
#if MCX_PRECISE_SAMPLER || !MCX_THETA_FROM_BITS
    return mcx_pcg_out(state);
#else
    return ((state >> ((state >> 28u) + 4u)) ^ state) * MCX_PCG_OUTMUL;
#endif
}

// float(h) / 4294967295.0 in f32: the divisor literal rounds to 2^32, so this is an exact scale.
// Closed interval: 0 iff h == 0, 1.0 iff h >= 0xFFFFFF80 (distribution.rs:72).
MCX_DEV float mcx_u01_closed(u32 h) { return (float)h * 0x1.0p-32f; }

// u for affine / table maps: [0,1) when guarded.
MCX_DEV float mcx_u01(u32 h) {
    float u = (float)h * 0x1.0p-32f;
#if MCX_GUARD_ENDPOINTS
    u = fminf(u, 0x1.fffffep-1f);
#endif
    return u;
}

// ---------------------------------------------------------------------------------------------
// opt-in stream: Philox4x32-10 (Salmon et al., SC'11; Random123). Not in the reference -- its counter
// hash has a 32-bit input space, so distinct (idx, iter) collide once a call draws more than ~2^32
// uniforms. counter = (idx, unit, stream, 0), key = (seed, 0x4d435831): 128-bit
// counter space, four u32 outputs per call = four samples (two Box-Muller pairs).
// ---------------------------------------------------------------------------------------------
#ifndef MCX_RNG
#define MCX_RNG 0                  // 0: the reference's PCG counter hash (parity stream), 1: Philox4x32-10
#endif
#define MCX_PHILOX_KEY1 0x4d435831u

struct McxU4 { u32 x, y, z, w; };

MCX_DEV McxU4 mcx_philox4x32_10(McxU4 c, u32 k0, u32 k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32 -> 64 multiply per product (v_mad_u64_u32: 5.2 cycles) instead of v_mul_hi + v_mul_lo (2 x 4.9)
        const u64 p0 = (u64)0xD2511F53u * (u64)c.x;
        const u64 p1 = (u64)0xCD9E8D57u * (u64)c.z;
        const u32 hi0 = (u32)(p0 >> 32), lo0 = (u32)p0, hi1 = (u32)(p1 >> 32), lo1 = (u32)p1;
        c = McxU4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// ---------------------------------------------------------------------------------------------
// samplers
// ---------------------------------------------------------------------------------------------

MCX_DEV float mcx_sample_uniform(float u, float lo, float hi) { return lo + u * (hi - lo); }

MCX_DEV float mcx_native_ln(float x)   { return __builtin_amdgcn_logf(x) * 0x1.62e43p-1f; }   // v_log_f32 * ln2

MCX_DEV float mcx_sample_exponential(float u, float lambda) {
    float v = fmaxf(u, 1.0e-7f);
#if MCX_PRECISE_SAMPLER
    return -logf(v) / lambda;
#else
    // lambda is wave-uniform: its reciprocal is hoisted out of the sampling loop, the sample costs one multiply.
    // (As an IEEE division this was 1.41 ms per 2e9 samples for lambda = 2.5 against 0.78 ms for lambda = 1.)
    return -mcx_native_ln(v) * __builtin_amdgcn_rcpf(lambda);
#endif
}

// One Box-Muller pair from two hash outputs. z0 = r cos(2 pi u2), z1 = r sin(2 pi u2), r = sqrt(-2 ln u1).
// v_sin_f32 / v_cos_f32 take their argument in revolutions, so u2 feeds them directly.
//
// MCX_THETA_FROM_BITS (default, not with MCX_PRECISE_SAMPLER): the angle is handed to v_sin/v_cos as
// 1 + (h2 >> 9) * 2^-23 revolutions in [1, 2) -- the top 23 bits of h2 dropped into the mantissa of 1.0f by ONE
// v_alignbit_b32 (the hardware reduces the integer part itself) instead of v_cvt_f32_u32 (half rate) + v_mul_f32.
// The angle is truncated to 2^-23 rev instead of rounded to 24 significant bits: a per-sample difference of
// <= 7.5e-7 rad, below the error of v_sin_f32 itself, and a fixed rotation of the (z0, z1) pair on average.
// h2 may be the angle word of mcx_pcg_angle.
MCX_DEV void mcx_box_muller(u32 h1, u32 h2, float& z0, float& z1) {
    float f1 = (float)h1;
#if MCX_GUARD_ENDPOINTS
    f1 = fmaxf(f1, 0.5f);                         // h == 0 -> u1 = 2^-33 instead of log(0)
#endif
#if MCX_PRECISE_SAMPLER || !MCX_THETA_FROM_BITS
    float u2 = (float)h2 * 0x1.0p-32f;
#else
    float u2 = __builtin_bit_cast(float, __builtin_amdgcn_alignbit(0x7Fu, h2, 9u));
#endif
#if MCX_PRECISE_SAMPLER
    float u1 = f1 * 0x1.0p-32f;
    float r = sqrtf(-2.0f * logf(u1));
    float theta = 6.283185307179586f * u2;
    z0 = r * cosf(theta);
    z1 = r * sinf(theta);
#else
    // -2 ln(f1 * 2^-32) = (32 - log2 f1) * 2 ln 2 >= 0. Only a 1-ulp overshoot of v_log_f32 at f1 ~ 2^32 could make it
    // -5e-6; |.| is an input modifier of v_sqrt_f32 (no instruction) and leaves every r2 >= 0 untouched.
    float l2 = __builtin_amdgcn_logf(f1);
    // (Tried: the guard as the clamp modifier of an FMA forming (32 - log2 f1) / 33 in [0, 1] instead of the v_max_f32 on f1,
    // one full-rate multiply more -- C2 0.396 against 0.394 ms, C3 0.667 / 0.660, C4 8.55 / 8.65: nothing in it.)
    float r2 = fmaf(l2, -0x1.62e43p+0f, 32.0f * 0x1.62e43p+0f);
    float r = __builtin_amdgcn_sqrtf(__builtin_fabsf(r2));
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
#endif
}

// ---------------------------------------------------------------------------------------------
// tables
// ---------------------------------------------------------------------------------------------
// A table is n interleaved {key, value} pairs (float2). For CDF sampling key = cdf, value = x;
// for PDF / log-PDF lookup key = x, value = pdf. The reference's buffers hold the same numbers as
// [n, x0, v0, x1, v1, ...] (engine.rs:533-564) resp. two separate arrays (engine.rs:235-295).

#ifndef MCX_CELL_TABLES
#define MCX_CELL_TABLES 0
#endif

// Pointers into the staged tables carry their address space explicitly: LDS (address space 3) when the module stages
// its tables (MCX_TABLES_LDS), global otherwise. Left generic, hiprtc's optimiser (ROCm 7.2) fails to infer LDS for
// pointers that travel through this struct and emits flat_load + 64-bit address arithmetic for every lookup -- which
// is what the round-1 builds of the table kernels ran (hipcc's newer clang did infer it, so offline ISA looked fine).
#ifndef MCX_TABLES_LDS
#define MCX_TABLES_LDS 1
#endif
#ifndef MCX_TBL                      // (MCX_EXTRA_DEFINES="MCX_TBL=" rebuilds the round-1 generic-pointer code for A/B runs)
#if MCX_TABLES_LDS
#define MCX_TBL __attribute__((address_space(3)))
#else
#define MCX_TBL
#endif
#endif
struct McxTable {
    const MCX_TBL float2* kv;     // LDS or global
    u32   n;
    float k0, k1;         // kv[0].x, kv[n-1].x
    float inv_dk;         // (n-1)/(k[n-1]-k[0]) when the keys are a uniform grid, else 0
    const MCX_TBL u32* guide;     // CDF only: guide[b] = lo | hi << 16, the search window of bucket b, or null
    u32   guide_bits;     // number of buckets G = 1 << guide_bits
    const MCX_TBL float2* cells;  // PDF / log-PDF on a strict grid: cells[1 + c] = {intercept, slope} of cell c, cells[0] and
                          // cells[n] = {outside, 0} for x left / right of the table; else null (then kv is set)
    float cell_scale, cell_c0;   // padded cell index = floor(x * cell_scale + cell_c0) (host: cell_map)
    float cell_s8, cell_c8;      // staged cells: LDS BYTE address of the cell = trunc(x * cell_s8 + cell_c8) & ~7
    float cell_lo8, cell_hi8;    //   clamped to [lo8, hi8] = the two sentinels (not with MCX_CELL_NOCLAMP)
    const MCX_TBL float* slopes;  // CDF: slopes[c] = dx/dcdf of cell c (0 for cells narrower than 1e-10), or null
};

// A wave-uniform value as a per-lane (VGPR) copy the compiler cannot fold back into an SGPR: on gfx950 a VALU
// instruction with an SGPR source operand issues at half rate (tools/ubench/valu_issue.hip).
MCX_DEV float mcx_in_vgpr(float s) {
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}
MCX_DEV u32 mcx_in_vgpr_u32(u32 s) {
    u32 v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}

// a / b with v_rcp_f32 (<= 1.5 ulp): used for the interpolation weights and the importance ratio.
// WGSL only promises 2.5 ulp for f32 division, so this stays inside the reference's own contract.
MCX_DEV float mcx_div(float a, float b) {
#if MCX_PRECISE_SAMPLER
    return a / b;
#else
    return a * __builtin_amdgcn_rcpf(b);
#endif
}

// First index i in [0, n-1] with key[i] >= q, searching exactly like the reference's capped loop
// (it never tests index n-1; `cap` = 12 for CDF sampling, 16 for PDF lookup).
template <int CAP>
MCX_DEV u32 mcx_lower_bound_capped(const MCX_TBL float2* kv, u32 n, float q) {
    u32 low = 0u, high = n - 1u;
#pragma unroll 1
    for (int j = 0; j < CAP; ++j) {
        if (low >= high) break;
        u32 mid = (low + high) >> 1;
        if (kv[mid].x < q) low = mid + 1u; else high = mid;
    }
    return low;
}

// WGSL mix(): "the linear blend e1 * (1 - e3) + e2 * e3" -- the form the reference's lookups use. The single-fma form
// a + t * (b - a) (MCX_MIX_TWO_PRODUCTS=0) saves two VALU instructions per lookup but measured < 1 % on C3/C4/C5.
#ifndef MCX_MIX_TWO_PRODUCTS
#define MCX_MIX_TWO_PRODUCTS 1
#endif
MCX_DEV float mcx_mix(float a, float b, float t) {
#if MCX_MIX_TWO_PRODUCTS
    return a * (1.0f - t) + b * t;
#else
    return fmaf(t, b - a, a);
#endif
}

#ifndef MCX_CDF_DIRECT
#define MCX_CDF_DIRECT 0
#endif
#if MCX_CDF_DIRECT
// Bucket-direct inverse CDF (host: mcx_plan.cpp build_cdf_direct; integrate kernel, reference stream). The record of
// the draw's bucket either IS the answer -- the line of the one table cell the whole bucket lies in, the cell the
// reference's search would select -- or it is flagged (sign bit of .y) and carries the search window [lo, hi] of a
// bucket that holds cdf nodes. Only the records live in LDS (8 bytes x 8192 buckets for a 2048-point table: two
// 1024-thread workgroups per CU); the {cdf, x} pairs and slopes the flagged draws need stay in global memory, where
// 24 KiB of them sit in the CU's L1 / the XCD's L2 -- they are read only by the batched resolve step.
struct McxCdfDirect {
    const __attribute__((address_space(3))) float2* rec;   // LDS (global when !MCX_TABLES_LDS: see MCX_REC)
    const float2* kv;          // global {cdf, x}
    const float* slopes;       // global dx/dcdf per cell
    u32 shift, mask;           // per-lane copies (mcx_in_vgpr_u32): bucket = h >> shift, low bits = h & mask
};
// the record of the draw's bucket: one 8-byte LDS read. (Measured and not kept, round 3: the records as two 4-byte planes --
// 1.7x the bank-conflict cycles, slower or equal; profiles/r03_c5_lds_variants.txt, code in commit 9b35320.)
MCX_DEV float2 mcx_cdf_rec(const McxCdfDirect& cd, u32 h) { return cd.rec[h >> cd.shift]; }
// The flag test as ONE v_cmp whose SGPR-pair result is the wave's ballot: bit l = lane l's record is flagged. The
// per-lane predicate is recovered with inverse_ballot (the mask itself becomes the exec mask: no second compare, no
// v_cndmask / v_cmp_ne round trip). volatile: the compare reads EXEC implicitly and must stay where it is written.
// hiprtc's clang (ROCm 7.2) has the LLVM intrinsic but not yet the __builtin_amdgcn_inverse_ballot_w64 spelling that
// hipcc's clang offers: bind the intrinsic by its IR name.
extern "C" __device__ bool mcx_inverse_ballot(u64 mask) __asm("llvm.amdgcn.inverse.ballot.i64");
MCX_DEV u64 mcx_cdf_flag_mask(float2 r) {
    u64 m;
    asm volatile("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(m) : "v"(__builtin_bit_cast(int, r.y)));
    return m;
}
// x on the record's line from the low hash bits: (u - b/G) = (h & mask) * 2^-32 exactly. For a flagged record this is
// ~1e-34 (the packed window read as a float, times -0.0): finite and harmless; the resolve step supplies the sample.
MCX_DEV float mcx_cdf_line(const McxCdfDirect& cd, float2 r, u32 h) { return fmaf(r.y, (float)(h & cd.mask), r.x); }
// The reference's lower bound inside the window of a flagged record, then its interpolant in slope form.
MCX_DEV float mcx_cdf_search_window(const McxCdfDirect& cd, u32 window, float u) {
    u32 lo = window & 0xFFFFu, hi = window >> 16;
    while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        if (cd.kv[mid].x < u) lo = mid + 1u; else hi = mid;
    }
    const u32 il = __builtin_elementwise_sub_sat(lo, 1u);
    const float2 a = cd.kv[il];
    const float d = fminf(fmaxf(u - a.x, 0.0f), 1.0f);
    return fmaf(cd.slopes[il], d, a.y);
}
#endif

// sample_from_cdf_table (distribution.rs:128-158). key = cdf, value = x.
// h is the hash u = float(h) * 2^-32 was made from.
MCX_DEV float mcx_sample_cdf(const McxTable& tb, float u, u32 h) {
    const u32 n = tb.n;
    u32 low;
    if (tb.guide != nullptr) {
        // bucket of u, then a short search inside the bucket's window [lo, hi]: same index as the full
        // lower bound because the table is non-decreasing (checked on the host) and n <= 4096.
        // The bucket comes from the integer: b = h >> (32 - bits). Rounding h to 24 bits moves u by less than a
        // bucket boundary (b/G and (b+1)/G are representable), so b/G <= u <= (b+1)/G still holds and the window --
        // whose upper end is the lower bound of (b+1)/G itself -- still contains the answer.
        const u32 w = tb.guide[h >> (32u - tb.guide_bits)];
        u32 lo = w & 0xFFFFu, hi = w >> 16;
        while (lo < hi) {
            u32 mid = (lo + hi) >> 1;
            if (tb.kv[mid].x < u) lo = mid + 1u; else hi = mid;
        }
        low = lo;
    } else {
        low = mcx_lower_bound_capped<12>(tb.kv, n, u);
    }
    const u32 il = __builtin_elementwise_sub_sat(low, 1u);       // max(low, 1) - 1: one saturating subtract
#if !MCX_PRECISE_SAMPLER
    // x = x[il] + slope[il] * (u - cdf[il]): the same interpolant with the division done once on the host (every CDF
    // table carries its slopes). low == 0 (u <= cdf[0]) must return x[0]: the difference is clamped at 0 -- an output
    // modifier of the subtraction, not an instruction.
    const float2 a = tb.kv[il];
    const float d = fminf(fmaxf(u - a.x, 0.0f), 1.0f);
    return fmaf(tb.slopes[il], d, a.y);
#else
    const u32 ih = low < n - 1u ? low : n - 1u;
    const float2 a = tb.kv[il], b2 = tb.kv[ih];
    const float dc = b2.x - a.x;
    const float t = mcx_div(u - a.x, dc);
    return dc < 1.0e-10f ? a.y : mcx_mix(a.y, b2.y, t);
#endif
}

// pdf_*_from_table / log_pdf_*_from_table (distribution.rs:181-223, 375-417). key = x, value = pdf.
// `outside` is 0.0 for PDF tables and -100.0 for log-PDF tables.
//
// The reference's search + clamps select the cell c with key[c] < x <= key[c+1] (c = 0 when x == key[0]).
// On a uniform grid that cell is guessed arithmetically and VERIFIED with the two keys the interpolation
// needs anyway. A lane whose guess fails (x within float rounding of a grid key: ~5e-4 of the lanes) takes
// the cold path: the neighbouring cells, then the reference's capped binary search (always the path on
// non-uniform grids, inv_dk == 0). Out-of-range lanes take no branch; they are masked at the end.
MCX_DEV float mcx_lerp_cell(float2 a, float2 b, float x) {
    const float dx = b.x - a.x;
    const float t = mcx_div(x - a.x, dx);
    return dx < 1.0e-10f ? a.y : mcx_mix(a.y, b.y, t);
}

// Cold path: cell by the reference's search (guess g < 0: no guess available); it runs for ~5e-4 of the
// lookups on uniform grids. (Out-of-line `__device__ __attribute__((noinline))` and a compile-time
// MCX_UNIFORM_TABLES specialisation were measured on C3/C4: both within +-3 % noise.)
#ifndef MCX_COLD
#define MCX_COLD MCX_DEV
#endif
MCX_COLD float mcx_table_lookup_cold(const MCX_TBL float2* kv, u32 n, float x, int g) {
    u32 low = 0xFFFFFFFFu;
    if (g >= 0) {
        if (g > 0 && kv[g - 1].x < x && x <= kv[g].x) low = (u32)g - 1u;
        else if ((u32)g + 2u < n && kv[g + 1].x < x && x <= kv[g + 2].x) low = (u32)g + 1u;
    }
    if (low == 0xFFFFFFFFu) {
        low = mcx_lower_bound_capped<16>(kv, n, x);
        low = (low > 1u ? low : 1u) - 1u;
        low = low < n - 2u ? low : n - 2u;
    }
    return mcx_lerp_cell(kv[low], kv[low + 1u], x);
}

//
// Strict f32-linspace grids (the tables Distribution builds) take the cell form instead: the host stores the
// interpolant of cell c as value(x) = s_c * x + a_c (coefficients from f64), the lookup is one 8-byte LDS read and
// one FMA, and nothing needs verifying -- a guess that lands in the neighbouring cell within float rounding of a
// node extends that cell's line by <= 1e-3 cell. Error against the exact piecewise-linear interpolant: 3.6e-7 mean
// / 5.7e-6 max on C4's log-PDF table, the same as the reference's own f32 evaluation (2.7e-7 / 3.4e-6). Half the
// LDS bytes and bank conflicts of the key/value form, ~10 VALU instead of ~24. Compiled in when the module was
// built with cell_tables (every PDF / log-PDF table of the call has cells; the host layer decides per call).
#ifndef MCX_CELL_NOCLAMP
#define MCX_CELL_NOCLAMP 0
#endif
#ifndef MCX_CELL_ADDR16
#define MCX_CELL_ADDR16 0
#endif
#if MCX_CELL_TABLES
MCX_DEV float2 mcx_cell_fetch(const McxTable& tb, float x) {
#if MCX_TABLES_LDS
    // The cell's LDS byte address straight from the FMA: x * (8 scale) + (8 c0 + base), truncated, low three bits
    // cleared -- v_fma, v_cvt_u32, v_and, against v_fma, v_cvt, v_lshl_add (half rate) for an index. The clamp to the
    // two {outside, 0} sentinels is one v_med3_f32 (half rate; lanes outside the table need no compare / select
    // afterwards, both table ends map inside); MCX_CELL_NOCLAMP launches stage sentinels over the sampler's whole
    // range instead (host: cell_pads) and need none.
    float t = fmaf(x, tb.cell_s8, tb.cell_c8);
#if !MCX_CELL_NOCLAMP
    t = __builtin_amdgcn_fmed3f(t, tb.cell_lo8, tb.cell_hi8);
#endif
#if MCX_CELL_ADDR16
    // the constants carry + 2^16 (mcx_stage_table): t lies in [2^16, 2^17), its mantissa is the byte address times 2^7 --
    // a shift and an AND (full rate) instead of the half-rate v_cvt_u32_f32
    return *(const __attribute__((address_space(3))) float2*)((__builtin_bit_cast(u32, t) >> 7) & 0xFFF8u);
#else
    return *(const __attribute__((address_space(3))) float2*)((u32)t & ~7u);
#endif
#else
    const float gf = __builtin_amdgcn_fmed3f(fmaf(x, tb.cell_scale, tb.cell_c0), 0.0f, (float)tb.n);
    return tb.cells[(u32)gf];
#endif
}
#endif

MCX_DEV float mcx_table_lookup(const McxTable& tb, float x, float outside) {
    const MCX_TBL float2* kv = tb.kv;
    const u32 n = tb.n;
    const bool out_of_range = (x < tb.k0) || (x > tb.k1);
#if MCX_CELL_TABLES
    (void)kv; (void)n; (void)out_of_range; (void)outside;
    const float2 c = mcx_cell_fetch(tb, x);
    return fmaf(c.y, x, c.x);
#else
    float v;
#if MCX_UNIFORM_TABLES
    {                                                         // every table of this module is a uniform grid
#else
    if (tb.inv_dk != 0.0f) {                                  // wave-uniform
#endif
        const float gf = fminf(fmaxf((x - tb.k0) * tb.inv_dk, 0.0f), (float)(n - 2u));
        const u32 g = (u32)gf;
        const float2 a = kv[g], b = kv[g + 1u];
        v = mcx_lerp_cell(a, b, x);
        const bool verified = (a.x < x) && (x <= b.x);
        if (__builtin_expect(!verified && !out_of_range, 0)) v = mcx_table_lookup_cold(kv, n, x, (int)g);
    }
#if !MCX_UNIFORM_TABLES
    else {
        v = outside;
        if (!out_of_range) v = mcx_table_lookup_cold(kv, n, x, -1);
    }
#endif
    return out_of_range ? outside : v;
#endif
}

// The importance-sampling tables of a launch. With MCX_USER_TABLES they are also visible to user functions: the
// reference's IS wrappers are WGSL text that calls pdf_target_from_table(x) / pdf_proposal_from_table(x)
// (python/wgpu_montecarlo/__init__.py:968-974), which the WGSL translator maps to the two accessors below.
struct McxIsTables { McxTable p, q; };
#ifndef MCX_USER_TABLES
#define MCX_USER_TABLES 0
#endif
#if MCX_USER_TABLES
__shared__ McxIsTables mcx_user_tables;
MCX_DEV float mcx_user_pdf_target(float x)   { return mcx_table_lookup(mcx_user_tables.p, x, 0.0f); }
MCX_DEV float mcx_user_pdf_proposal(float x) { return mcx_table_lookup(mcx_user_tables.q, x, 0.0f); }
#endif

// ---------------------------------------------------------------------------------------------
// WGSL builtins the emitter may call (semantics: W3C WGSL; reference FUNC_MAP transpiler.py:82-112)
// ---------------------------------------------------------------------------------------------
MCX_DEV float mcx_sign(float x)  { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
MCX_DEV float mcx_fract(float x) { return x - floorf(x); }
MCX_DEV float mcx_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
MCX_DEV float mcx_step(float edge, float x) { return x >= edge ? 1.0f : 0.0f; }
MCX_DEV float mcx_smoothstep(float lo, float hi, float x) {
    float t = mcx_clamp((x - lo) / (hi - lo), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
MCX_DEV float mcx_select(float f, float t, bool c) { return c ? t : f; }

// sin / cos / tan / pow of math="default" (emit_hip.py): the hardware instructions behind an argument reduction
// that keeps their error flat. v_sin_f32 / v_cos_f32 take revolutions; x / 2pi is formed as n + phase with
// n = rint(x * c_hi) and phase = fma(x, c_lo, fma(x, c_hi, -n)) (c_hi + c_lo = 1/2pi to 2^-52, the inner fma is the
// exact product less n), so the phase lies in [-0.5, 0.5] and is good to 2^-25 revolutions whatever |x| is, and
// small arguments keep their relative accuracy. Measured over 2^24 points per range (tools/ubench/trig_accuracy.hip,
// profiles/r03_trig_pow_accuracy.txt): absolute error <= 2.6e-7 for |x| <= 128 and <= 4e-7 up to 1e6, against 7e-8 for
// ocml and the 4.9e-4 WGSL promises on [-pi, pi]; 61 against 146 cycles per wave for a (sin, cos) pair, the range check
// included. From 1e6 on (where consecutive floats are 0.06 rad apart and more) the ocml routine runs.
MCX_DEV float mcx_trig_phase(float x) {
    const float c_hi = 0x1.45f306p-3f, c_lo = 0x1.b9391p-28f;
    const float n = __builtin_rintf(x * c_hi);
    return __builtin_fmaf(x, c_lo, __builtin_fmaf(x, c_hi, -n));
}
#define MCX_TRIG_HW_BOUND 1.0e6f
// The range check is a wave-level one: every lane takes the hardware path, and only a wave that holds an argument beyond
// the bound branches (a scalar branch on the ballot) to patch those lanes with the ocml value -- 61 cycles per (sin, cos)
// pair against 79 for a per-lane if / else around the two paths and 45 with no check at all.
#define MCX_RARE_LANES(cond) (__builtin_expect(__builtin_amdgcn_ballot_w64(cond) != 0ull, 0) && (cond))
MCX_DEV float mcx_sin(float x) {
    float r = __builtin_amdgcn_sinf(mcx_trig_phase(x));
    if (MCX_RARE_LANES(!(fabsf(x) < MCX_TRIG_HW_BOUND))) r = sinf(x);
    return r;
}
MCX_DEV float mcx_cos(float x) {
    float r = __builtin_amdgcn_cosf(mcx_trig_phase(x));
    if (MCX_RARE_LANES(!(fabsf(x) < MCX_TRIG_HW_BOUND))) r = cosf(x);
    return r;
}
MCX_DEV float mcx_tan(float x) {
    const float ph = mcx_trig_phase(x);
    float r = __builtin_amdgcn_sinf(ph) * __builtin_amdgcn_rcpf(__builtin_amdgcn_cosf(ph));
    if (MCX_RARE_LANES(!(fabsf(x) < MCX_TRIG_HW_BOUND))) r = tanf(x);
    return r;
}
// sinh / cosh as WGSL defines their accuracy, (exp(x) -+ exp(-x)) / 2 on v_exp_f32 / v_rcp_f32: e2 = exp(|x|) / 2 is formed as
// exp2(|x| log2 e - 1), so nothing overflows before the result does; sinh switches to its odd series below 0.5, where the
// difference of the two exponentials cancels. Relative error <= about 1e-7 * (3 + |x|) (profiles/r03_trig_pow_accuracy.txt);
// 12 and 6 instructions against 120 and 116 for the ocml routines.
MCX_DEV float mcx_cosh(float x) {
    const float e2 = __builtin_amdgcn_exp2f(__builtin_fmaf(fabsf(x), 0x1.715476p+0f, -1.0f));
    return __builtin_fmaf(0.25f, __builtin_amdgcn_rcpf(e2), e2);
}
MCX_DEV float mcx_sinh(float x) {
    const float a = fabsf(x), a2 = a * a;
    const float e2 = __builtin_amdgcn_exp2f(__builtin_fmaf(a, 0x1.715476p+0f, -1.0f));
    const float big = __builtin_fmaf(-0.25f, __builtin_amdgcn_rcpf(e2), e2);
    const float small = a * __builtin_fmaf(a2, __builtin_fmaf(a2, __builtin_fmaf(a2, 0x1.a01a02p-13f, 0x1.111112p-7f), 0x1.555556p-3f), 1.0f);
    return __builtin_copysignf(a < 0.5f ? small : big, x);
}
// pow as WGSL defines its accuracy, exp2(y * log2(x)) on v_log_f32 / v_exp_f32 (relative error about
// 1.2e-7 * (1 + |y * log2 x|); 40 against 600 cycles per wave for ocml powf), with powf's results for a negative
// base: an integral exponent carries the sign of its parity, any other gives NaN. Zeros, denormals, infinities
// and NaNs in either argument take the ocml routine.
MCX_DEV float mcx_pow(float x, float y) {
    const float ax = fabsf(x);
    float r = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(ax));
    if (x < 0.0f) {
        const float h = 0.5f * y;
        r = (truncf(y) != y) ? __builtin_nanf("") : ((truncf(h) != h) ? -r : r);
    }
    if (MCX_RARE_LANES(!(ax >= 0x1p-126f && ax < __builtin_inff() && fabsf(y) < __builtin_inff()))) r = powf(x, y);
    return r;
}
MCX_DEV float mcx_degrees(float r) { return r * 57.29577951308232f; }
MCX_DEV float mcx_radians(float d) { return d * 0.017453292519943295f; }
// WGSL `%`: truncated remainder for floats (fmodf), the C operator for integers
template <class A, class B>
MCX_DEV auto mcx_mod(A a, B b) {
    if constexpr (__is_integral(A) && __is_integral(B)) return a % b;
    else return fmodf((float)a, (float)b);
}
MCX_DEV float mcx_b2f(bool b) { return b ? 1.0f : 0.0f; }
MCX_DEV float mcx_b2f(float v) { return v; }
MCX_DEV float mcx_b2f(int v) { return (float)v; }
MCX_DEV float mcx_b2f(u32 v) { return (float)v; }

// ---------------------------------------------------------------------------------------------
// wave64 / workgroup reductions (fixed order => bit-reproducible for a fixed launch geometry)
// ---------------------------------------------------------------------------------------------
MCX_DEV double mcx_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
MCX_DEV float mcx_wave_sum_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
MCX_DEV u32 mcx_wave_sum_u32(u32 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
