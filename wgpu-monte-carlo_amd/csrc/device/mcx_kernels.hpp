// mcx_kernels.hpp -- the fused Monte-Carlo kernels for gfx950 (wave64, 256 CUs).
//
// Translation unit layout (assembled by mcx_runtime.cpp, compiled by hiprtc):
//     #define MCX_K .. / MCX_DIST .. / MCX_BLOCK .. / MCX_WEIGHT .. / MCX_P_TABLE .. / MCX_Q_TABLE .. / MCX_CELL_TABLES ..
//             / MCX_Q_SAMPLER .. / MCX_WALK .. / MCX_SECOND_MOMENTS .. / MCX_RNG .. / MCX_UNIT_PARAMS ..
//     mcx_args.h, mcx_device.hpp
//     <emitted user functions: user_func_0 .. user_func_{K-1}, optional mcx_pdf_p / mcx_pdf_q>
//     <generated mcx_eval_all>
//     mcx_kernels.hpp  (this file)
//
// Reference kernels these replace (file:line into /root/reference):
//   K1 integrate             src/shader_gen.rs:58-118   (loop :105-112, epilogue :293-303)
//   K2 integrate + PDF tables src/shader_gen.rs:149-213 (+ IS wrapper text python/wgpu_montecarlo/__init__.py:893-899, 968-974)
//   K3 mcmc                  src/shader_gen.rs:347-429, step :511-537, epilogue :574-579
//   host mean over threads   src/lib.rs:129-138
//
// Design differences (MI355X-first):
//   * logical (idx, i) sample grid is decoupled from physical threads: each logical thread's loop
//     is cut into chunks so that up to 4096 workgroups (16 per CU) exist whatever T is; the multiset of samples
//     is unchanged, only the summation order differs.
//   * per-thread sums go f32 registers -> f64 registers every MCX_FLUSH units -> wave64 xor-shuffle
//     -> LDS across waves -> one contiguous record `partials[workgroup][k]`; a second tiny kernel
//     folds the records in a fixed order. No [T][K] output buffer, no host-side reduction.
//   * lookup tables are staged once per workgroup into LDS: PDF / log-PDF tables on a strict grid as per-cell
//     {intercept, slope} (one read + one FMA per lookup), others as interleaved {key, value}; CDF tables with their
//     inverse-CDF slopes and a guide table that bounds the search.
//   * the importance weight p/q is computed once per sample, not once per function.
#pragma once

#ifndef MCX_K
#error "MCX_K must be defined"
#endif
#ifndef MCX_DIST
#error "MCX_DIST must be defined"
#endif
#ifndef MCX_BLOCK
#define MCX_BLOCK 256
#endif
#ifndef MCX_WEIGHT
#define MCX_WEIGHT 0
#endif
#ifndef MCX_P_TABLE
#define MCX_P_TABLE 0
#endif
#ifndef MCX_Q_TABLE
#define MCX_Q_TABLE 0
#endif
#ifndef MCX_TABLES_LDS
#define MCX_TABLES_LDS 1
#endif
#ifndef MCX_FLUSH
// units accumulated in f32 before folding into the f64 sums. Many rows fold through the wave reduction + LDS
// (MCX_WAVE_FLUSH), which costs ~8 instructions per row: half as often there (C5: 12.97 -> 12.75 ms; 64: 13.5 ms).
// K = 4 is indifferent between 128 and 256 and loses below (C2: 0.393 / 0.393 / 0.403 / 0.409 ms at 256 / 128 / 64 / 32).
#define MCX_FLUSH (MCX_K > 8 ? 256 : 128)
#endif
#ifndef MCX_UNROLL
#define MCX_UNROLL 1             // unroll factor of the sampling loops (tuning knob)
#endif
static constexpr int mcx_unroll = MCX_UNROLL;   // a constant, not the macro, in the pragmas: they survive -save-temps

#define MCX_WAVES (MCX_BLOCK / 64)

// Unit parameters -- normal(0,1), uniform(0,1), exponential(1) -- are specialised at JIT time: the affine map
// mean + sigma * z (resp. min + u * (max - min), -ln(u) / lambda) is the identity and is not emitted. Bit-identical
// results (0 + 1 * z == z); for the headline N(0,1) workload it removes 2 FMAs and a re-materialised move per pair.
#ifndef MCX_UNIT_PARAMS
#define MCX_UNIT_PARAMS 0
#endif
#if MCX_UNIT_PARAMS
#define MCX_AFFINE(z) (z)
#else
// mean + std * z as ONE full-rate v_fma_f32. On gfx950 a VALU instruction that reads an SGPR operand issues at HALF
// rate (4.1 instead of 2.2 cycles per wave-instruction per SIMD; inline constants and literals are free --
// tools/ubench/valu_issue.hip, profiles/r02_valu_issue_microbench.txt), and an FMA may read only one scalar operand
// anyway, which the compiler otherwise re-materialises (v_mov_b32 from the SGPR) in every iteration. So every
// wave-uniform float the hot loops multiply or add with lives in a VGPR: opaque copies made once per kernel (pv).
#define MCX_AFFINE(z) (pv.p1 + pv.p2 * (z))
#endif
// The call's distribution parameters (and what the loops derive from them) as per-lane copies of wave-uniform values.
struct McxParamsV {
    float p1, p2;        // min / mean / lambda, max / std
    float inv_p1;        // exponential: 1 / lambda
    float span;          // uniform: max - min
    float q_norm;        // q_sampler weights: std * sqrt(2 pi)
};
template <class Args>
MCX_DEV McxParamsV mcx_params_v(const Args& a) {
    McxParamsV pv;
    pv.p1 = mcx_in_vgpr(a.param1);
    pv.p2 = mcx_in_vgpr(a.param2);
    pv.inv_p1 = mcx_in_vgpr(__builtin_amdgcn_rcpf(a.param1));
    pv.span = mcx_in_vgpr(a.param2 - a.param1);
    pv.q_norm = mcx_in_vgpr(a.param2 * 2.5066282746310002f);
    return pv;
}

// Large K: per-thread f64 sums would need 2K VGPRs (K = 32 spills). Instead every MCX_FLUSH units each
// wave reduces its f32 accumulators with xor-shuffles and lane 0 adds the K wave totals into f64 slots in
// LDS ("LDS-staged" reduction); the registers hold f32 accumulators only.
#ifndef MCX_WAVE_FLUSH
#define MCX_WAVE_FLUSH (MCX_K > 8)
#endif
// Two accumulator sets (one per sample of a pair) give two independent accumulation chains; beyond K = 16 the
// second set only costs registers.
#ifndef MCX_PAIR_LANES
#define MCX_PAIR_LANES (MCX_K <= 16)
#endif

extern __shared__ __attribute__((aligned(16))) unsigned char mcx_lds_raw[];

// Copy one table into LDS at byte offset `off` (advanced), or leave it in HBM.
MCX_DEV McxTable mcx_stage_table(const McxTableDesc& d, u32& off) {
    McxTable t;
    t.n = d.n;
    t.inv_dk = d.inv_dk;
    t.guide_bits = d.guide_bits;
    t.guide = nullptr;
    t.kv = nullptr;
    t.cells = nullptr;
    t.cell_c0 = mcx_in_vgpr(d.cell_c0);           // VGPR copies: an SGPR operand would halve the rate of the index FMA
    t.cell_scale = mcx_in_vgpr(d.cell_scale);
    t.slopes = nullptr;
    t.k0 = 0.0f;
    t.k1 = 0.0f;
    t.cell_s8 = t.cell_c8 = t.cell_lo8 = t.cell_hi8 = 0.0f;
    if (d.n == 0u) return t;
#if MCX_CELL_TABLES
    const bool cell_form = d.cells != nullptr;          // PDF / log-PDF table: stage the n-1 cells, not kv (CDF tables: kv)
#else
    const bool cell_form = false;
#endif
#if MCX_TABLES_LDS
    MCX_TBL float2* dst = (MCX_TBL float2*)(mcx_lds_raw + off);
    const float2* src = (const float2*)(cell_form ? d.cells : d.kv);
    if (cell_form) {
        // two sentinels + n - 1 cells, and pad_l / pad_r more copies of the sentinels either side (MCX_CELL_NOCLAMP)
        const u32 count = d.n + 1u + d.pad_l + d.pad_r;
        for (u32 i = threadIdx.x; i < count; i += MCX_BLOCK) {
            const u32 j = i < d.pad_l ? 0u : i - d.pad_l;
            dst[i] = src[j < d.n ? j : d.n];
        }
        off += count * 8u;
        t.cells = dst + d.pad_l;
        // byte address of sentinel 0: exact (< 2^24); MCX_CELL_ADDR16: + 2^16, the address is then read out of the mantissa
        const float base = (float)((u32)(__UINTPTR_TYPE__)dst + 8u * d.pad_l + (MCX_CELL_ADDR16 ? 65536u : 0u));
        t.cell_s8 = mcx_in_vgpr(8.0f * d.cell_scale);
        t.cell_c8 = mcx_in_vgpr(fmaf(8.0f, d.cell_c0, base));
        t.cell_lo8 = mcx_in_vgpr(base);
        t.cell_hi8 = mcx_in_vgpr(base + 8.0f * (float)d.n);
    } else {
        for (u32 i = threadIdx.x; i < d.n; i += MCX_BLOCK) dst[i] = src[i];
        off += d.n * 8u;
        t.kv = dst;
    }
    if (d.slopes != nullptr) {
        MCX_TBL float* sdst = (MCX_TBL float*)(mcx_lds_raw + off);
        for (u32 i = threadIdx.x; i < d.n; i += MCX_BLOCK) sdst[i] = d.slopes[i];
        off += (d.n * 4u + 7u) & ~7u;      // keep the next table's float2 8-byte aligned (same rounding on the host)
        t.slopes = sdst;
    }
    if (d.guide != nullptr) {
        MCX_TBL u32* gdst = (MCX_TBL u32*)(mcx_lds_raw + off);
        u32 gn = 1u << d.guide_bits;
        for (u32 i = threadIdx.x; i < gn; i += MCX_BLOCK) gdst[i] = d.guide[i];
        off += gn * 4u;
        t.guide = gdst;
    }
#else
    if (cell_form) t.cells = (const float2*)d.cells; else t.kv = (const float2*)d.kv;
    t.slopes = d.slopes;
    t.guide = d.guide;
#endif
    t.k0 = d.kv[0];
    t.k1 = d.kv[2u * (d.n - 1u)];
    return t;
}

#if MCX_CDF_DIRECT
// Stage the bucket-direct records of the sampling CDF table (the launch guarantees the table has them); its {cdf, x}
// pairs and slopes are NOT staged (McxCdfDirect). Unconditional code: a pointer that is LDS on every path keeps its
// address space (ds_read_b64, not flat_load) through hiprtc's optimiser.
MCX_DEV McxCdfDirect mcx_stage_cdf_direct(const McxTableDesc& d, u32& off) {
    McxCdfDirect cd;
    cd.shift = mcx_in_vgpr_u32(32u - d.direct_bits);
    cd.mask = mcx_in_vgpr_u32((1u << (32u - d.direct_bits)) - 1u);
    cd.kv = (const float2*)d.kv;
    cd.slopes = d.slopes;
    __attribute__((address_space(3))) float2* dst = (__attribute__((address_space(3))) float2*)(mcx_lds_raw + off);
    const u32 dn = 1u << d.direct_bits;
    for (u32 i = threadIdx.x; i < dn; i += MCX_BLOCK) dst[i] = ((const float2*)d.direct)[i];
    off += dn * 8u;
    cd.rec = dst;
    return cd;
}
#endif

// Fold the per-thread f64 sums of a workgroup and store them as one contiguous record
// partials[blockIdx.x * N + k] (a single N*8-byte store per workgroup).
template <int N>
MCX_DEV void mcx_block_reduce_store(double (&v)[N], double* partials) {
    __shared__ double red[MCX_WAVES][N];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double s = mcx_wave_sum(v[k]);
        if (lane == 0u) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < (u32)N) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < MCX_WAVES; ++w) s += red[w][threadIdx.x];
        partials[(u64)blockIdx.x * N + threadIdx.x] = s;
    }
}

// =============================================================================================
// K1 / K2: fused sample + (weight) + K evaluations + reduction
// =============================================================================================
#ifndef MCX_Q_SAMPLER
#define MCX_Q_SAMPLER 0
#endif
#ifndef MCX_MOMENT_FAMILY
#define MCX_MOMENT_FAMILY 0
#endif

// Importance weight p(x) / q(x) of one sample, computed once per sample and shared by the K functions (the
// reference re-evaluates p and q inside each of its K wrapper functions); 1 for plain integration.
MCX_DEV float mcx_weight(float x, const McxIsTables& tb) {
#if MCX_WEIGHT
#if MCX_P_TABLE
    const float p = mcx_table_lookup(tb.p, x, 0.0f);
#else
    const float p = mcx_b2f(mcx_pdf_p(x));
#endif
#if MCX_Q_TABLE
    const float q = mcx_table_lookup(tb.q, x, 0.0f);
#elif MCX_Q_SAMPLER
    const float q = 1.0f;                 // not reached: normal samples take mcx_weight_z
#else
    const float q = mcx_b2f(mcx_pdf_q(x));
#endif
    return mcx_div(p, q);
#else
    (void)x; (void)tb;
    return 1.0f;
#endif
}

// Normal sampler: the sample is x = mean + std * z. With MCX_Q_SAMPLER the importance weight uses
// 1/q(x) = std * sqrt(2 pi) * exp(z^2 / 2) from the deviate itself -- one v_exp_f32 and three multiplies instead of
// evaluating the emitted N(mean, std) density at x (subtract, divide, square, exp, scale) and taking its reciprocal.
MCX_DEV float mcx_weight_z(float z, float x, const McxParamsV& pv, const McxIsTables& tb) {
#if MCX_WEIGHT && MCX_Q_SAMPLER
#if MCX_P_TABLE
    const float p = mcx_table_lookup(tb.p, x, 0.0f);
#else
    const float p = mcx_b2f(mcx_pdf_p(x));
#endif
#if MCX_UNIT_PARAMS
    return p * (2.5066282746310002f * __builtin_amdgcn_exp2f((z * z) * 0.72134752044448170f));
#else
    return p * (pv.q_norm * __builtin_amdgcn_exp2f((z * z) * 0.72134752044448170f));
#endif
#else
    (void)z; (void)pv;
    return mcx_weight(x, tb);
#endif
}

// acc[k * S] is the accumulator of function k (S = 2: the A/B sample lanes are interleaved, see below).
template <int S>
MCX_DEV void mcx_accumulate(float x, const McxIsTables& tb, float* acc) {
    mcx_eval_all<S>(x, mcx_weight(x, tb), acc);
}
template <int S>
MCX_DEV void mcx_accumulate_z(float z, float x, const McxParamsV& pv, const McxIsTables& tb, float* acc) {
    mcx_eval_all<S>(x, mcx_weight_z(z, x, pv, tb), acc);
}

// MCX_MOMENT_QUAD: the moment family takes four samples per trip through mcx_eval_quad (1.25 VALU per power per sample
// against 1.5 for pairs); what is left of a flush block (at most one pair and one single) goes the pair way.
// Measured (2e9 samples, warm): reference stream, Beta(2,5) 16 / 32 rows 1.86 -> 1.77 / 2.70 -> 2.48 ms, N(0,1) 16 / 32 rows
// 1.46 -> 1.45 / 2.31 -> 2.25 ms, but 8 rows 0.99 -> 1.11 ms (set-up outweighs the saving): from 12 rows. The Philox loop
// has the four outputs of one call at hand and otherwise evaluates per sample: always.
#ifndef MCX_MOMENT_QUAD
#define MCX_MOMENT_QUAD (MCX_MOMENT_FAMILY && (MCX_K >= 12 || MCX_RNG == 1))
#endif

// Two samples at once. MCX_MOMENT_FAMILY (user_func_i(x) = x^(i+1), promised by the caller): the generated
// mcx_eval_pair accumulates the weighted power sums wa a^k + wb b^k through Newton's identity into the first
// accumulator set; otherwise each sample goes through mcx_eval_all into its own set.
template <int S>
MCX_DEV void mcx_accumulate_pair(float xa, float xb, float wa, float wb, float* acc) {
#if MCX_MOMENT_FAMILY
    mcx_eval_pair<S>(xa, xb, wa, wb, acc);
#else
    mcx_eval_all<S>(xa, wa, acc);
    mcx_eval_all<S>(xb, wb, acc + (S - 1));
#endif
}

// One sample of a non-normal distribution from one hash output.
MCX_DEV float mcx_draw(u32 h, const McxParamsV& pv, const McxTable& cdf_tb) {
#if MCX_DIST == MCX_DIST_UNIFORM
    (void)cdf_tb;
    return MCX_UNIT_PARAMS ? mcx_u01(h) : fmaf(mcx_u01(h), pv.span, pv.p1);          // min + u (max - min)
#elif MCX_DIST == MCX_DIST_EXPONENTIAL
    (void)cdf_tb;
#if MCX_PRECISE_SAMPLER
    return mcx_sample_exponential(mcx_u01_closed(h), MCX_UNIT_PARAMS ? 1.0f : pv.p1);
#else
    // -ln(max(u, 1e-7)) / lambda with the reciprocal of the wave-uniform lambda taken once per kernel
    const float e = -mcx_native_ln(fmaxf(mcx_u01_closed(h), 1.0e-7f));
    return MCX_UNIT_PARAMS ? e : e * pv.inv_p1;
#endif
#else
    (void)pv;
    return mcx_sample_cdf(cdf_tb, mcx_u01_closed(h), h);
#endif
}

// 1024-thread workgroups are chosen so that more waves share one staged copy of the tables; two of them fit a CU only
// with <= 64 VGPRs per lane (8 waves per SIMD). Hold the register allocator to that (K = 32 moments on a CDF table with
// the Philox stream: 66 VGPRs and 3.53 ms per 2e9 samples without, 2.87 ms with).
#ifndef MCX_INTEGRATE_ATTR
#if MCX_BLOCK == 1024 && MCX_TABLES_LDS
#define MCX_INTEGRATE_ATTR __attribute__((amdgpu_waves_per_eu(8)))
#else
#define MCX_INTEGRATE_ATTR
#endif
#endif
#if !defined(MCX_KIND) || MCX_KIND == 0          // a module holds the kernel of its kind (+ the fold kernel)
extern "C" __global__ void __launch_bounds__(MCX_BLOCK) MCX_INTEGRATE_ATTR
mcx_integrate_kernel(McxIntegrateArgs a) {
    const McxParamsV pv = mcx_params_v(a);
    (void)pv;
    u32 lds_off = 0u;
#if MCX_CDF_DIRECT
    const McxCdfDirect cd = mcx_stage_cdf_direct(a.cdf, lds_off);
    McxTable cdf_tb = mcx_stage_table(McxTableDesc{}, lds_off);          // nothing else of the CDF table is staged
#else
    McxTable cdf_tb = mcx_stage_table(a.cdf, lds_off);
#endif
    McxIsTables is_tb;
    is_tb.p = mcx_stage_table(a.target_pdf, lds_off);
    is_tb.q = mcx_stage_table(a.proposal_pdf, lds_off);
    (void)cdf_tb;
#if MCX_USER_TABLES
    if (threadIdx.x == 0u) mcx_user_tables = is_tb;          // user functions read them through mcx_user_pdf_*()
#endif
    __syncthreads();

    // idx_count is a multiple of 64 (whole reference workgroups; checked on the host), so the 64 lanes of
    // a wave share one chunk: the unit range [u0, u1) and every loop bound below are wave-uniform and
    // live in SGPRs (readfirstlane makes that visible to the compiler); only `idx` differs per lane.
    const u32 g = blockIdx.x * MCX_BLOCK + threadIdx.x;
    const u32 g_wave = __builtin_amdgcn_readfirstlane(g);
    const u32 total = a.idx_count * a.n_chunks;
    const bool active = g_wave < total;
    const u32 chunk = active ? g_wave / a.idx_count : 0u;
    const u32 idx = a.idx_begin + (active ? g - chunk * a.idx_count : 0u);
    u32 u0 = a.unit_begin + chunk * a.units_per_chunk;
    u32 u1 = u0 + a.units_per_chunk;
    u1 = u1 < a.unit_end ? u1 : a.unit_end;
    if (!active) { u0 = 0u; u1 = 0u; }

    // Two accumulator sets: lane A takes the first sample of every pair, lane B the second (two independent add chains),
    // interleaved in one array [A0, B0, A1, B1, ...]. The modules are compiled with -fno-slp-vectorize: letting the
    // compiler pack (A_k, B_k) into v_pk_* was measured slower on gfx950 than scalar code.
#if MCX_PAIR_LANES
#define MCX_ACC_S 2
#else
#define MCX_ACC_S 1
#endif
    float acc[MCX_K * MCX_ACC_S];
#define MCX_ACC_A(k) acc[(k) * MCX_ACC_S]
#define MCX_ACC_B(k) acc[(k) * MCX_ACC_S + (MCX_ACC_S - 1)]

#if MCX_WAVE_FLUSH
    __shared__ double wave_sums[MCX_WAVES][MCX_K];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane < (u32)MCX_K) wave_sums[wave][lane] = 0.0;
    // (Measured and not kept, round 3: the K wave totals by DPP adds on the vector ALU instead of ds_bpermute_b32 on the LDS
    // pipe, 16 lanes updating the f64 slots at once -- 20 % fewer LDS instructions, no faster: the kernel is bound by VALU
    // issue and the LDS pipe has slack. profiles/r03_c5_lds_variants.txt; the variant's code is in commit 9b35320.)
#define MCX_FLUSH_ACC()                                                                     \
    do {                                                                                    \
        _Pragma("unroll") for (int k = 0; k < MCX_K; ++k) {                                 \
            float s_ = mcx_wave_sum_f32(MCX_PAIR_LANES ? MCX_ACC_A(k) + MCX_ACC_B(k) : MCX_ACC_A(k)); \
            if (lane == 0u) wave_sums[wave][k] += (double)s_;                               \
        }                                                                                   \
    } while (0)
#else
    double sum[MCX_K];
#pragma unroll
    for (int k = 0; k < MCX_K; ++k) sum[k] = 0.0;
#define MCX_FLUSH_ACC()                                                                     \
    do {                                                                                    \
        _Pragma("unroll") for (int k = 0; k < MCX_K; ++k)                                   \
            sum[k] += MCX_PAIR_LANES ? (double)MCX_ACC_A(k) + (double)MCX_ACC_B(k) : (double)MCX_ACC_A(k); \
    } while (0)
#endif
#define MCX_ZERO_ACC()                                                                      \
    do {                                                                                    \
        _Pragma("unroll") for (int k = 0; k < MCX_K * MCX_ACC_S; ++k) acc[k] = 0.0f;                 \
    } while (0)

#if MCX_CDF_DIRECT
    // Bucket-direct sampling: a draw whose bucket holds no cdf node is one 8-byte LDS read + one FMA. The others (17 % on
    // Beta(2,5) at 8192 buckets) would make EVERY wave walk the search path for a few lanes each iteration; instead their
    // hash words go to a per-wave LDS queue (one v_cmp per sample gives the ballot, mbcnt the slots) and are resolved 64
    // at a time with all lanes busy. The sum does not care which lane, or which flush block, a sample is added in; every
    // draw of the grid is still evaluated exactly once. (Measured and not kept: requesting the records of pair p + 1
    // before evaluating pair p, and one combined append per pair -- 1.61 ms against 1.59 ms per 2e9 samples at K = 4, and
    // 4 KiB more LDS.) Interface to the loops below, both streams:
    //   direct_pair(hA, hB, xA, xB, lA, lB) / direct_one(h, x, l): x = the sample to evaluate in this lane now, l = there
    //   is one (false: evaluate nothing here); direct_finish(): evaluates what the queue still holds.
#ifndef MCX_QCAP
#define MCX_QCAP 128u                 // <= 63 left over + 64 appended
#endif
#ifndef MCX_DEFER_TEST
#define MCX_DEFER_TEST 0
#endif
    __attribute__((address_space(3))) u32* const queue =
        (__attribute__((address_space(3))) u32*)(mcx_lds_raw + lds_off) + (threadIdx.x >> 6) * MCX_QCAP;
    const u32 lane_id = threadIdx.x & 63u;
#ifndef MCX_DIRECT_SWAP
#define MCX_DIRECT_SWAP (MCX_K >= 12)        // measured: 8 rows neutral, 16 rows + 4 %, 32 rows + 9 %
#endif
#if MCX_DIRECT_SWAP
    // Many rows: the evaluation is the expensive part, and a flagged lane sitting it out (and being evaluated again in the
    // resolve step) wastes it. Here the queue is a ring of 128 words that holds hash words on their way in and resolved
    // samples on their way out, never both in one slot: a flagged lane EXCHANGES its hash word for the sample resolved
    // in that slot one lap (128 flagged draws) earlier and evaluates that instead -- every lane of every trip evaluates
    // a real sample, the resolve step only searches. Whenever the write position crosses a 64-slot boundary the block
    // just completed is resolved in place by all 64 lanes. Slots that were never written hold MCX_RING_EMPTY; at the
    // end the partial block is resolved and all 128 slots are evaluated.
#if MCX_MOMENT_FAMILY && !MCX_WEIGHT
#define MCX_RING_EMPTY 0x07000000u       // 9.6e-35f: its powers add nothing to the power sums, no mask needed
#define MCX_RING_LIVE(bits) true
#else
#define MCX_RING_EMPTY 0xFFFFFFFFu       // a NaN pattern no resolved sample has
#define MCX_RING_LIVE(bits) ((bits) != MCX_RING_EMPTY)
#endif
    queue[lane_id] = MCX_RING_EMPTY;
    queue[lane_id + 64u] = MCX_RING_EMPTY;
    u32 r_pos = 0u;                   // wave-uniform write position, 0..127
    auto resolve = [&](u32 base, u32 take) {
        __builtin_amdgcn_wave_barrier();
        if (lane_id < take) {
            const u32 h = queue[base + lane_id];
            const float2 r = mcx_cdf_rec(cd, h);
            queue[base + lane_id] =
                __builtin_bit_cast(u32, mcx_cdf_search_window(cd, __builtin_bit_cast(u32, r.x), (float)h * 0x1.0p-32f));
        }
        __builtin_amdgcn_wave_barrier();
    };
    // flagged lanes (ballot m): hash word in, an earlier resolved sample out (x, live = it is one)
    auto swap_in = [&](u64 m, bool flagged, u32 h, float& x, bool& live) {
        const u32 pos = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, r_pos));
        if (flagged) {
            const u32 old = __hip_atomic_exchange(&queue[pos & 127u], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            live = MCX_RING_LIVE(old);
            x = live ? __builtin_bit_cast(float, old) : 0.0f;          // never hand the NaN pattern on (0 * NaN in a weighted sum)
        }
        const u32 np = r_pos + (u32)__builtin_popcountll(m);
        if ((np ^ r_pos) & 64u) resolve(r_pos & 64u, 64u);
        r_pos = np & 127u;
    };
    auto direct_one = [&](u32 h, float& x, bool& l) {
        const float2 r = mcx_cdf_rec(cd, h);
        const u64 m = mcx_cdf_flag_mask(r);
        const bool f = mcx_inverse_ballot(m);
        x = mcx_cdf_line(cd, r, h);
        l = true;
        swap_in(m, f, h, x, l);
    };
    auto direct_pair = [&](u32 hA, u32 hB, float& xA, float& xB, bool& lA, bool& lB) {
        const float2 rA = mcx_cdf_rec(cd, hA);
        const float2 rB = mcx_cdf_rec(cd, hB);
        const u64 mA = mcx_cdf_flag_mask(rA), mB = mcx_cdf_flag_mask(rB);
        const bool fA = mcx_inverse_ballot(mA), fB = mcx_inverse_ballot(mB);
        xA = mcx_cdf_line(cd, rA, hA), xB = mcx_cdf_line(cd, rB, hB);
        lA = true, lB = true;
        swap_in(mA, fA, hA, xA, lA);
        swap_in(mB, fB, hB, xB, lB);
    };
    auto direct_finish = [&]() {                             // the partial block, then everything still in the ring
        if (r_pos & 63u) resolve(r_pos & 64u, r_pos & 63u);
        const u32 bA = queue[lane_id], bB = queue[lane_id + 64u];
        MCX_ZERO_ACC();
        if (MCX_RING_LIVE(bA)) mcx_accumulate<MCX_ACC_S>(__builtin_bit_cast(float, bA), is_tb, acc);
        if (MCX_RING_LIVE(bB)) mcx_accumulate<MCX_ACC_S>(__builtin_bit_cast(float, bB), is_tb, acc + (MCX_ACC_S - 1));
        MCX_FLUSH_ACC();
    };
#else
    u32 q_count = 0u;                 // wave-uniform
    // resolve the newest `take` (<= 64) queued draws, one per lane: the reference's search inside the bucket's window
    auto resolve = [&](u32 take) {
        __builtin_amdgcn_wave_barrier();
        if (lane_id < take) {
            const u32 h = queue[q_count - take + lane_id];
            const float2 r = mcx_cdf_rec(cd, h);
            const float x = mcx_cdf_search_window(cd, __builtin_bit_cast(u32, r.x), (float)h * 0x1.0p-32f);
            mcx_accumulate<MCX_ACC_S>(x, is_tb, acc);
        }
        q_count -= take;
        __builtin_amdgcn_wave_barrier();
    };
    // append the flagged lanes' hash words (m = their ballot); resolve a full batch as soon as there is one
    auto defer = [&](u64 m, bool flagged, u32 h) {
#if MCX_DEFER_TEST
        if (m != 0ull)                // wave-uniform; almost always true (1 - 0.83^64), so by default not tested
#endif
        {
            const u32 pos = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, q_count));
            if (flagged) queue[pos] = h;
            q_count += (u32)__builtin_popcountll(m);
            if (q_count >= 64u) resolve(64u);
        }
    };
    // a flagged lane sits the evaluation out (l = false; its x is ~1e-34, mcx_cdf_line); its draw is evaluated in resolve
    auto direct_one = [&](u32 h, float& x, bool& l) {
        const float2 r = mcx_cdf_rec(cd, h);
        const u64 m = mcx_cdf_flag_mask(r);
        const bool f = mcx_inverse_ballot(m);
        x = mcx_cdf_line(cd, r, h);
        l = !f;
        defer(m, f, h);
    };
    auto direct_pair = [&](u32 hA, u32 hB, float& xA, float& xB, bool& lA, bool& lB) {
        const float2 rA = mcx_cdf_rec(cd, hA);
        const float2 rB = mcx_cdf_rec(cd, hB);
        const u64 mA = mcx_cdf_flag_mask(rA), mB = mcx_cdf_flag_mask(rB);
        const bool fA = mcx_inverse_ballot(mA), fB = mcx_inverse_ballot(mB);
        xA = mcx_cdf_line(cd, rA, hA), xB = mcx_cdf_line(cd, rB, hB);
        lA = !fA, lB = !fB;
        defer(mA, fA, hA);
        defer(mB, fB, hB);
    };
    auto direct_finish = [&]() {                             // what is left in the queue (< 64 draws)
        if (q_count != 0u) {
            MCX_ZERO_ACC();
            resolve(q_count);
            MCX_FLUSH_ACC();
        }
    };
#endif
    // evaluation of what direct_pair / direct_one handed out. Moment family without weights: a lane without a sample
    // carries x ~ 1e-34, whose powers add nothing -- no masking; with weights its weight is forced to 0.
#define MCX_LIVE_WEIGHT(l, x) (MCX_WEIGHT ? ((l) ? mcx_weight(x, is_tb) : 0.0f) : 1.0f)
    auto direct_eval_pair = [&](float xA, float xB, bool lA, bool lB) {
#if MCX_MOMENT_FAMILY
        mcx_accumulate_pair<MCX_ACC_S>(xA, xB, MCX_LIVE_WEIGHT(lA, xA), MCX_LIVE_WEIGHT(lB, xB), acc);
#else
        if (lA) mcx_accumulate<MCX_ACC_S>(xA, is_tb, acc);
        if (lB) mcx_accumulate<MCX_ACC_S>(xB, is_tb, acc + (MCX_ACC_S - 1));
#endif
    };
#if MCX_MOMENT_QUAD
    auto direct_eval_quad = [&](float xA, float xB, float xC, float xD, bool lA, bool lB, bool lC, bool lD) {
        mcx_eval_quad<MCX_ACC_S>(xA, xB, xC, xD, MCX_LIVE_WEIGHT(lA, xA), MCX_LIVE_WEIGHT(lB, xB), MCX_LIVE_WEIGHT(lC, xC),
                                 MCX_LIVE_WEIGHT(lD, xD), acc);
    };
#endif
#endif          // MCX_CDF_DIRECT

#if MCX_RNG == 1
    // Philox stream: unit = one Philox call j = iterations 4j .. 4j+3 (two Box-Muller pairs for the normal)
    auto philox_sample = [&](u32 h_first, u32 h_second, u32 n_valid) {
        // consumes two outputs = two iterations; n_valid in {1, 2} of them exist (i < L)
#if MCX_DIST == MCX_DIST_NORMAL
        float z0, z1;
        mcx_box_muller(h_first, h_second, z0, z1);
        mcx_accumulate_z<MCX_ACC_S>(z0, MCX_AFFINE(z0), pv, is_tb, acc);
        if (n_valid > 1u) mcx_accumulate_z<MCX_ACC_S>(z1, MCX_AFFINE(z1), pv, is_tb, acc + (MCX_ACC_S - 1));
#elif MCX_CDF_DIRECT
        float x;
        bool l;
        direct_one(h_first, x, l);
        if (l) mcx_accumulate<MCX_ACC_S>(x, is_tb, acc);
        if (n_valid > 1u) {                                   // wave-uniform
            direct_one(h_second, x, l);
            if (l) mcx_accumulate<MCX_ACC_S>(x, is_tb, acc + (MCX_ACC_S - 1));
        }
#else
        mcx_accumulate<MCX_ACC_S>(mcx_draw(h_first, pv, cdf_tb), is_tb, acc);
        if (n_valid > 1u) mcx_accumulate<MCX_ACC_S>(mcx_draw(h_second, pv, cdf_tb), is_tb, acc + (MCX_ACC_S - 1));
#endif
    };
    const u32 full_quads = a.loops_per_thread >> 2;          // calls whose four iterations all exist
    const u32 e_full = u1 < full_quads ? u1 : full_quads;
    u32 j = u0;
    while (j < e_full) {
        u32 blk_end = j + MCX_FLUSH / 2u;
        blk_end = blk_end < e_full ? blk_end : e_full;
        MCX_ZERO_ACC();
        for (; j < blk_end; ++j) {
            const McxU4 o = mcx_philox4x32_10(McxU4{idx, j, 0u, 0u}, a.seed, MCX_PHILOX_KEY1);
#if MCX_CDF_DIRECT
            float xa, xb, xc, xd;
            bool la, lb, lc, ld;
            direct_pair(o.x, o.y, xa, xb, la, lb);
            direct_pair(o.z, o.w, xc, xd, lc, ld);
#if MCX_MOMENT_QUAD
            direct_eval_quad(xa, xb, xc, xd, la, lb, lc, ld);
#else
            direct_eval_pair(xa, xb, la, lb);
            direct_eval_pair(xc, xd, lc, ld);
#endif
#elif MCX_MOMENT_QUAD
            // the four outputs of one call are one quad of the moment family
#if MCX_DIST == MCX_DIST_NORMAL
            float z0, z1, z2, z3;
            mcx_box_muller(o.x, o.y, z0, z1);
            mcx_box_muller(o.z, o.w, z2, z3);
            const float xa = MCX_AFFINE(z0), xb = MCX_AFFINE(z1), xc = MCX_AFFINE(z2), xd = MCX_AFFINE(z3);
            mcx_eval_quad<MCX_ACC_S>(xa, xb, xc, xd, mcx_weight_z(z0, xa, pv, is_tb), mcx_weight_z(z1, xb, pv, is_tb),
                                     mcx_weight_z(z2, xc, pv, is_tb), mcx_weight_z(z3, xd, pv, is_tb), acc);
#else
            const float xa = mcx_draw(o.x, pv, cdf_tb), xb = mcx_draw(o.y, pv, cdf_tb);
            const float xc = mcx_draw(o.z, pv, cdf_tb), xd = mcx_draw(o.w, pv, cdf_tb);
            mcx_eval_quad<MCX_ACC_S>(xa, xb, xc, xd, mcx_weight(xa, is_tb), mcx_weight(xb, is_tb), mcx_weight(xc, is_tb),
                                     mcx_weight(xd, is_tb), acc);
#endif
#else
            philox_sample(o.x, o.y, 2u);
            philox_sample(o.z, o.w, 2u);
#endif
        }
        MCX_FLUSH_ACC();
    }
    if (active && u1 > full_quads) {                          // the last, partial call: 1..3 iterations left
        const u32 rem = a.loops_per_thread - 4u * full_quads;
        const McxU4 o = mcx_philox4x32_10(McxU4{idx, full_quads, 0u, 0u}, a.seed, MCX_PHILOX_KEY1);
        MCX_ZERO_ACC();
        philox_sample(o.x, o.y, rem >= 2u ? 2u : 1u);
        if (rem == 3u) philox_sample(o.z, o.w, 1u);
        MCX_FLUSH_ACC();
    }
#if MCX_CDF_DIRECT
    direct_finish();
#endif
#elif MCX_DIST == MCX_DIST_NORMAL
    // unit = Box-Muller pair j: iterations (2j, 2j+1), counters (4j, 4j+1) (distribution.rs:97-98)
    const u32 full_pairs = a.loops_per_thread >> 1;          // pairs whose second half is used
    const u32 e_full = u1 < full_pairs ? u1 : full_pairs;
    u32 st = mcx_state(a.seed, idx, 4u * u0);
    u32 j = u0;
    while (j < e_full) {
        u32 blk_end = j + MCX_FLUSH;
        blk_end = blk_end < e_full ? blk_end : e_full;
        MCX_ZERO_ACC();
        auto draw_pair = [&](float& xa, float& xb, float& wa, float& wb) {
            u32 h1 = mcx_pcg_out(st);
            u32 h2 = mcx_pcg_angle(st + MCX_STATE_STEP);
            st += 4u * MCX_STATE_STEP;
            float z0, z1;
            mcx_box_muller(h1, h2, z0, z1);
            xa = MCX_AFFINE(z0), xb = MCX_AFFINE(z1);
            wa = mcx_weight_z(z0, xa, pv, is_tb), wb = mcx_weight_z(z1, xb, pv, is_tb);
        };
#if MCX_MOMENT_QUAD
        for (; j + 1u < blk_end; j += 2u) {
            float xa, xb, xc, xd, wa, wb, wc, wd;
            draw_pair(xa, xb, wa, wb);
            draw_pair(xc, xd, wc, wd);
            mcx_eval_quad<MCX_ACC_S>(xa, xb, xc, xd, wa, wb, wc, wd, acc);
        }
#endif
#pragma unroll mcx_unroll
        for (; j < blk_end; ++j) {
            float xa, xb, wa, wb;
            draw_pair(xa, xb, wa, wb);
            mcx_accumulate_pair<MCX_ACC_S>(xa, xb, wa, wb, acc);
        }
        MCX_FLUSH_ACC();
    }
    if (active && u1 > full_pairs) {
        // L odd: the last pair contributes z0 only, z1 is discarded (shader_gen.rs:105-112)
        st = mcx_state(a.seed, idx, 4u * full_pairs);
        u32 h1 = mcx_pcg_out(st);
        u32 h2 = mcx_pcg_angle(st + MCX_STATE_STEP);
        float z0, z1;
        mcx_box_muller(h1, h2, z0, z1);
        MCX_ZERO_ACC();
        mcx_accumulate_z<MCX_ACC_S>(z0, MCX_AFFINE(z0), pv, is_tb, acc);
        MCX_FLUSH_ACC();
    }
#elif MCX_CDF_DIRECT
    // unit = iteration i, counter i (distribution.rs:333)
    u32 st = mcx_state(a.seed, idx, u0);
    u32 i = u0;
    while (i < u1) {
        u32 blk_end = i + 2u * MCX_FLUSH;
        blk_end = blk_end < u1 ? blk_end : u1;
        MCX_ZERO_ACC();
        auto draw_pair = [&](float& xA, float& xB, bool& lA, bool& lB) {
            const u32 hA = mcx_pcg_out(st);
            const u32 hB = mcx_pcg_out(st + MCX_STATE_STEP);
            st += 2u * MCX_STATE_STEP;
            direct_pair(hA, hB, xA, xB, lA, lB);
        };
#if MCX_MOMENT_QUAD
        for (; i + 3u < blk_end; i += 4u) {
            float xA, xB, xC, xD;
            bool lA, lB, lC, lD;
            draw_pair(xA, xB, lA, lB);
            draw_pair(xC, xD, lC, lD);
            direct_eval_quad(xA, xB, xC, xD, lA, lB, lC, lD);
        }
#endif
#pragma unroll mcx_unroll
        for (; i + 1u < blk_end; i += 2u) {
            float xA, xB;
            bool lA, lB;
            draw_pair(xA, xB, lA, lB);
            direct_eval_pair(xA, xB, lA, lB);
        }
        if (i < blk_end) {                                   // odd tail of the block
            float x;
            bool l;
            direct_one(mcx_pcg_out(st), x, l);
            st += MCX_STATE_STEP;
            ++i;
            if (l) mcx_accumulate<MCX_ACC_S>(x, is_tb, acc);
        }
        MCX_FLUSH_ACC();
    }
    direct_finish();
#else
    // unit = iteration i, counter i (distribution.rs:333); two iterations per trip for the A/B lanes
    u32 st = mcx_state(a.seed, idx, u0);
    u32 i = u0;
    while (i < u1) {
        u32 blk_end = i + 2u * MCX_FLUSH;
        blk_end = blk_end < u1 ? blk_end : u1;
        MCX_ZERO_ACC();
        auto draw_pair = [&](float& xA, float& xB) {
            u32 hA = mcx_pcg_out(st);
            u32 hB = mcx_pcg_out(st + MCX_STATE_STEP);
            st += 2u * MCX_STATE_STEP;
            xA = mcx_draw(hA, pv, cdf_tb);
            xB = mcx_draw(hB, pv, cdf_tb);
        };
#if MCX_MOMENT_QUAD
        for (; i + 3u < blk_end; i += 4u) {
            float xA, xB, xC, xD;
            draw_pair(xA, xB);
            draw_pair(xC, xD);
            mcx_eval_quad<MCX_ACC_S>(xA, xB, xC, xD, mcx_weight(xA, is_tb), mcx_weight(xB, is_tb), mcx_weight(xC, is_tb),
                                     mcx_weight(xD, is_tb), acc);
        }
#endif
#pragma unroll mcx_unroll
        for (; i + 1u < blk_end; i += 2u) {
            float xA, xB;
            draw_pair(xA, xB);
            mcx_accumulate_pair<MCX_ACC_S>(xA, xB, mcx_weight(xA, is_tb), mcx_weight(xB, is_tb), acc);
        }
        if (i < blk_end) {                                   // odd tail of the block
            mcx_accumulate<MCX_ACC_S>(mcx_draw(mcx_pcg_out(st), pv, cdf_tb), is_tb, acc);
            st += MCX_STATE_STEP;
            ++i;
        }
        MCX_FLUSH_ACC();
    }
#endif

#if MCX_WAVE_FLUSH
    __syncthreads();
    if (threadIdx.x < (u32)MCX_K) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < MCX_WAVES; ++w) s += wave_sums[w][threadIdx.x];
        a.partials[(u64)blockIdx.x * MCX_K + threadIdx.x] = s;
    }
#else
    mcx_block_reduce_store<MCX_K>(sum, a.partials);
#endif
#undef MCX_FLUSH_ACC
#undef MCX_ZERO_ACC
#undef MCX_ACC_A
#undef MCX_ACC_B
#undef MCX_ACC_S
}

#endif          // MCX_KIND == 0

template <int V> struct McxPhase { static constexpr int value = V; };

// =============================================================================================
// K3: Metropolis-Hastings, one chain per thread. MCX_WALK 0: independent proposals x' ~ q (the reference);
// 1: random walk x' = x + d, d ~ q, with the Hastings correction log q(-d) - log q(d); 2: random walk with a
// symmetric q (no correction); 3: symmetric random walk x' = x + s d whose per-chain scale s is tuned during burn-in
// (log s += t^-1/2 (accepted - target) after step t) and frozen afterwards. MCX_SECOND_MOMENTS: accumulators also carry f^2, and every chain adds the
// square of its own mean to rows MCX_K+1.. (batch means over chains -> standard error, effective sample size).
// =============================================================================================
#ifndef MCX_WALK
#define MCX_WALK 0
#endif
#ifndef MCX_SECOND_MOMENTS
#define MCX_SECOND_MOMENTS 0
#endif
#if MCX_SECOND_MOMENTS
#define MCX_MCMC_ROWS0 (MCX_K + 1 + MCX_NF)
#else
#define MCX_MCMC_ROWS0 (MCX_K + 1)
#endif
#define MCX_MCMC_ROWS (MCX_MCMC_ROWS0 + (MCX_WALK == 3 ? 1 : 0))     // adaptive walk: + the sum of the final step scales
#ifndef MCX_LOGPDF_ANALYTIC
#define MCX_LOGPDF_ANALYTIC 0          // bit 0 / 1: target / proposal log-density from mcx_logpdf_p / mcx_logpdf_q (user_src)
#endif
// log p / log q of the MH step (shader_gen.rs:327-339, 496-509): from the staged table, or the emitted analytic form
#if MCX_LOGPDF_ANALYTIC & 1
#define MCX_LOGP(tb, x) mcx_b2f(mcx_logpdf_p(x))
#else
#define MCX_LOGP(tb, x) mcx_table_lookup(tb, x, -100.0f)
#endif
#if MCX_LOGPDF_ANALYTIC & 2
#define MCX_LOGQ(tb, x) mcx_b2f(mcx_logpdf_q(x))
#else
#define MCX_LOGQ(tb, x) mcx_table_lookup(tb, x, -100.0f)
#endif
#ifndef MCX_PROP_ITER_OFFSET
#define MCX_PROP_ITER_OFFSET 1000000u      // shader_gen.rs:477-489
#endif
#ifndef MCX_ACCEPT_SEED_OFFSET
#define MCX_ACCEPT_SEED_OFFSET 999999u     // shader_gen.rs:529
#endif

// One proposal of a non-normal family from one hash output.
MCX_DEV float mcx_draw_proposal(u32 h, const McxParamsV& pv, const McxTable& cdf_tb) { return mcx_draw(h, pv, cdf_tb); }

#if !defined(MCX_KIND) || MCX_KIND == 1
extern "C" __global__ void __launch_bounds__(MCX_BLOCK)
mcx_mcmc_kernel(McxMcmcArgs a) {
    const McxParamsV pv = mcx_params_v(a);
    (void)pv;
    u32 lds_off = 0u;
    McxTable cdf_tb = mcx_stage_table(a.cdf, lds_off);
    McxTable lp_tb = mcx_stage_table(a.target_logpdf, lds_off);
    McxTable lq_tb = mcx_stage_table(a.proposal_logpdf, lds_off);
    (void)cdf_tb; (void)lq_tb; (void)lp_tb;
    __syncthreads();

    // chain_count is a multiple of 256, so a wave is entirely inside or outside the launch's chain range
    const u32 g = blockIdx.x * MCX_BLOCK + threadIdx.x;
    const bool active = __builtin_amdgcn_readfirstlane(g) < a.chain_count;
    const u32 idx = a.chain_begin + (active ? g : 0u);
    // the last step this launch runs (a time segment ends earlier: McxMcmcArgs.it_end); wave-uniform
    const u32 total_steps = active ? (a.it_end ? a.it_end : a.n_burnin + a.n_steps) : 0u;

    double sum[MCX_MCMC_ROWS];
#pragma unroll
    for (int k = 0; k < MCX_MCMC_ROWS; ++k) sum[k] = 0.0;
    float acc[MCX_K];
#pragma unroll
    for (int k = 0; k < MCX_K; ++k) acc[k] = 0.0f;
    u64 n_accept = 0u;            // accepted steps of the whole wave (scalar: ballot + popcount, no VALU)
    u32 since_flush = 0u;

    // ---- initial state ~ proposal, counter iter = 0 (shader_gen.rs:445-463) ----
    float cur_x;
    float z_init = 0.0f;          // normal proposal: the standard deviate behind cur_x (MCX_Q_SAMPLER)
    (void)z_init;
#if MCX_RNG == 1
    // Philox stream (opt-in): one call per TWO steps, counter (idx, it >> 1, 1, 0). Step `it` takes half it & 1 of
    // the call: the normal proposal is z0 (even) / z1 (odd) of the Box-Muller pair from outputs (x, y), any other
    // proposal draws from output x (even) / y (odd); the accept uniform is output z (even) / w (odd). it = 0 is the
    // initial state (even half of call 0, no accept test).
    float ph_odd_draw;             // the odd half of call 0, consumed by step 1
    u32 ph_odd_accept;
    {
        const McxU4 o = mcx_philox4x32_10(McxU4{idx, 0u, 1u, 0u}, a.seed, MCX_PHILOX_KEY1);
#if MCX_DIST == MCX_DIST_NORMAL
        float z0, z1;
        mcx_box_muller(o.x, o.y, z0, z1);
        cur_x = MCX_AFFINE(z0);
        z_init = z0;
        ph_odd_draw = z1;          // normal: the deviate; mh_step_h applies the affine map
#else
        cur_x = mcx_draw_proposal(o.x, pv, cdf_tb);
        ph_odd_draw = mcx_draw_proposal(o.y, pv, cdf_tb);
#endif
        ph_odd_accept = o.w;
    }
#elif MCX_DIST == MCX_DIST_NORMAL
    float z_cached;
    {
        u32 s0 = mcx_state(a.seed, idx, 0u);
        float z0;
        mcx_box_muller(mcx_pcg_out(s0), mcx_pcg_angle(s0 + MCX_STATE_STEP), z0, z_cached);
        cur_x = MCX_AFFINE(z0);       // z1 stays cached for step it = 1
        z_init = z0;
    }
    // proposal state for even `it`: counters 2*(it+OFFSET), 2*(it+OFFSET)+1
    u32 st_prop = mcx_state(a.seed, idx, 2u * (2u + MCX_PROP_ITER_OFFSET));
#else
    cur_x = mcx_draw_proposal(mcx_pcg_out(mcx_state(a.seed, idx, 0u)), pv, cdf_tb);
    u32 st_prop = mcx_state(a.seed, idx, 1u + MCX_PROP_ITER_OFFSET);
#endif
#if MCX_WALK
    cur_x += a.x0;                                            // chains start at x0 + d_0 (once per chain)
#endif
    float cur_lp = MCX_LOGP(lp_tb, cur_x);
#if MCX_WALK == 0
#if MCX_Q_SAMPLER
    // normal proposal: log q(x) = -z^2/2 - log(std sqrt(2 pi)) for the deviate z behind x; the constant cancels in
    // log alpha. (The reference interpolates a 2048-point table of the same function: up to 6e-6 below it, and -100
    // beyond 7 std, which a draw reaches with probability 2.6e-12.)
    float cur_lq = -0.5f * z_init * z_init;
#else
    float cur_lq = MCX_LOGQ(lq_tb, cur_x);   // pure function of cur_x: cached
#endif
#endif
#if MCX_RNG == 0
    u32 st_acc = mcx_state(a.seed + MCX_ACCEPT_SEED_OFFSET, idx, 1u);
#endif
    // Independence sampler, default math: log alpha = log p(x') + log q(x) - log p(x) - log q(x') (shader_gen.rs:526)
    // = w(x') - w(x) with w = log p - log q, so the chain carries (x, w): one state select and two additions fewer per
    // step than carrying log p and log q separately (v_cndmask_b32 is a half-rate instruction on gfx950). Same value up
    // to the rounding of the regrouped sum; math="precise" keeps the reference's left-to-right form.
#define MCX_W_STATE (MCX_WALK == 0 && !MCX_PRECISE_SAMPLER)
#if MCX_W_STATE
    cur_lp = cur_lp - cur_lq;                                 // from here on cur_lp holds w(current)
#endif

#if MCX_WALK == 1 && MCX_Q_SAMPLER
    const float rw_ms = MCX_UNIT_PARAMS ? 0.0f : mcx_in_vgpr(a.param1 / a.param2);  // mean / std of the increments
#endif
    // Second half of a Metropolis-Hastings step (shader_gen.rs:527-537): accept test, state update, accumulation.
    // `it` is wave-uniform. prop_lq is used by the independent sampler only (it becomes the cached log q(current)).
#if MCX_WALK == 3
    float ad_log_s = 0.0f, ad_scale = 1.0f;      // per-chain step scale, adapted during burn-in only
    const float ad_target = mcx_in_vgpr(a.target_accept);
#endif
    // `phase`: McxPhase<0> decides per step whether it is a sampling step and when to fold the f32 accumulators
    // (wave-uniform compares and a counter: ~5 scalar instructions and 2 branches per step); <1> = a burn-in step,
    // <2> = a sampling step inside a block whose caller folds -- the batched loop below knows which it is in.
    u32 n_accept_blk = 0u;        // accepted steps of the current block (phases 1, 2): one 32-bit scalar add per step
    auto mh_finish = [&](auto phase, u32 it, float prop_x, float prop_lp, float prop_lq, float log_alpha, u32 ha) {
        constexpr int PHASE = decltype(phase)::value;
#if MCX_PRECISE_SAMPLER
        float ln_u = logf(mcx_u01_closed(ha));
#else
        float ln_u = fmaf(__builtin_amdgcn_logf((float)ha), 0x1.62e43p-1f, -32.0f * 0x1.62e43p-1f);   // h = 0 -> -inf: accept
#endif
#if MCX_WALK == 3
        const bool was_inside = cur_lp > -100.0f;
#endif
#if MCX_WALK
        // a proposal outside the target table has density 0, not e^-100: without this a chain that starts outside
        // would see a flat landscape, accept every move and diffuse away instead of waiting for a move back inside
        const bool take = (ln_u < log_alpha) & (prop_lp > -100.0f);
#else
        const bool take = ln_u < log_alpha;
#endif
#ifndef MCX_MASKED_STATE_UPDATE
#define MCX_MASKED_STATE_UPDATE 0      // 1: exec-masked v_mov (full rate) instead of v_cndmask (half rate). Measured on C4:
                                       // slower, 9.57 against 9.21 ms (131 072 chains: 1.98 / 1.74): the scalar mask handling and
                                       // the branch cost more than the two half-rate selects save
#endif
#if MCX_MASKED_STATE_UPDATE && MCX_W_STATE
        if (take) asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "+v"(cur_x), "+v"(cur_lp) : "v"(prop_x), "v"(prop_lp));
#else
        cur_x = take ? prop_x : cur_x;
        cur_lp = take ? prop_lp : cur_lp;
#endif
#if MCX_WALK == 0 && !MCX_W_STATE
        cur_lq = take ? prop_lq : cur_lq;
#else
        (void)prop_lq;
#endif
        if constexpr (PHASE == 0) n_accept += (u64)__builtin_popcountll(__builtin_amdgcn_ballot_w64(take));
        else n_accept_blk += (u32)__builtin_popcountll(__builtin_amdgcn_ballot_w64(take));
#if MCX_WALK == 3
        if (it <= a.n_burnin) {                  // wave-uniform; diminishing adaptation, none while sampling
            // a chain that is still outside the target table rejects every proposal that does not land inside: that
            // says nothing about the scale, and shrinking it there would strand the chain
            const float g = was_inside ? __builtin_amdgcn_rsqf((float)it) : 0.0f;
            ad_log_s = fmaf(g, (take ? 1.0f : 0.0f) - ad_target, ad_log_s);
            ad_scale = __builtin_amdgcn_exp2f(ad_log_s * 1.4426950408889634f);
        }
#endif
        if constexpr (PHASE == 2) {
            mcx_eval_all<1>(cur_x, 1.0f, acc);
        } else if constexpr (PHASE == 0) {
            if (it > a.n_burnin) {               // accumulate after every sampling step (shader_gen.rs:417-423)
                mcx_eval_all<1>(cur_x, 1.0f, acc);
                if (++since_flush == 2u * MCX_FLUSH) {
#pragma unroll
                    for (int k = 0; k < MCX_K; ++k) { sum[k] += (double)acc[k]; acc[k] = 0.0f; }
                    since_flush = 0u;
                }
            }
        }
    };
    // One Metropolis-Hastings step with the proposal draw and accept hash ha (shader_gen.rs:511-537).
    // `zd`: the standard normal deviate for a normal proposal (the affine map is applied here), else the draw itself.
    auto mh_step_h = [&](u32 it, float zd, u32 ha) {
#if MCX_DIST == MCX_DIST_NORMAL
        const float draw = MCX_AFFINE(zd);
#else
        const float draw = zd;
#endif
#if MCX_WALK == 0
        const float prop_x = draw;
        float prop_lp = MCX_LOGP(lp_tb, prop_x);
#if MCX_W_STATE
#if MCX_Q_SAMPLER
        const float prop_w = fmaf(0.5f * zd, zd, prop_lp);       // log p(x') - log q(x'), log q = -z^2 / 2 (+ const)
#else
        const float prop_w = prop_lp - MCX_LOGQ(lq_tb, prop_x);
#endif
        mh_finish(McxPhase<0>{}, it, prop_x, prop_w, 0.0f, prop_w - cur_lp, ha);
#else
#if MCX_Q_SAMPLER
        float prop_lq = -0.5f * zd * zd;
#else
        float prop_lq = MCX_LOGQ(lq_tb, prop_x);
#endif
        mh_finish(McxPhase<0>{}, it, prop_x, prop_lp, prop_lq, prop_lp + cur_lq - cur_lp - prop_lq, ha);   // shader_gen.rs:526
#endif
#elif MCX_WALK == 1
        const float prop_x = cur_x + draw;
        float prop_lp = MCX_LOGP(lp_tb, prop_x);
#if MCX_Q_SAMPLER
        // d = m + s z: log q(-d) - log q(d) = (z^2 - (z + 2m/s)^2) / 2 = -(2m/s) (z + m/s)
        mh_finish(McxPhase<0>{}, it, prop_x, prop_lp, 0.0f, prop_lp - cur_lp - (2.0f * rw_ms) * (zd + rw_ms), ha);
#else
        float lq_fwd = MCX_LOGQ(lq_tb, draw);                     // q(x' | x) = q(d)
        float lq_back = MCX_LOGQ(lq_tb, -draw);                   // q(x | x') = q(-d)
        mh_finish(McxPhase<0>{}, it, prop_x, prop_lp, 0.0f, prop_lp + lq_back - cur_lp - lq_fwd, ha);
#endif
#elif MCX_WALK == 2
        const float prop_x = cur_x + draw;
        float prop_lp = MCX_LOGP(lp_tb, prop_x);
        mh_finish(McxPhase<0>{}, it, prop_x, prop_lp, 0.0f, prop_lp - cur_lp, ha);
#else
        const float prop_x = fmaf(ad_scale, draw, cur_x);
        float prop_lp = MCX_LOGP(lp_tb, prop_x);
        mh_finish(McxPhase<0>{}, it, prop_x, prop_lp, 0.0f, prop_lp - cur_lp, ha);
#endif
    };
#if MCX_RNG == 0
    // reference stream: the accept uniform is U(seed + 999999, idx, it) (shader_gen.rs:529)
    auto mh_step = [&](u32 it, float prop_x) {
        const u32 ha = mcx_pcg_out(st_acc);
        st_acc += MCX_STATE_STEP;
        mh_step_h(it, prop_x, ha);
    };
#endif

    // For the independence sampler everything about a proposal except the compare and the selects is independent of
    // the chain's state, so both proposals of a trip (points, table reads, w) are produced before the first accept
    // test: two LDS reads in flight instead of one read per uniform branch of mh_finish. Measured on C4's kernel (ms
    // without / with): 1 048 576 chains 9.43 / 9.40, 524 288: 5.33 / 5.18, 131 072: 1.85 / 1.78, 65 536: 1.43 / 1.33 --
    // it pays most where a small shard leaves 2-4 waves per SIMD and the step's dependent chain is exposed. (Round 1
    // measured the same re-ordering slower, 13.76 against 13.4 ms: that was with flat_load lookups.) Not for the Philox
    // stream: there the trip already carries a 20-multiply Philox call and the batched form measured 12.8 against
    // 11.9 ms on C4.
#ifndef MCX_MH_BATCH
#define MCX_MH_BATCH (MCX_W_STATE && MCX_DIST == MCX_DIST_NORMAL && MCX_RNG == 0)
#endif
#if MCX_MH_BATCH
    auto mh_two_steps = [&](auto phase, u32 it, float z0, float z1, u32 ha0, u32 ha1) {
        const float x0 = MCX_AFFINE(z0), x1 = MCX_AFFINE(z1);
        const float lp0 = MCX_LOGP(lp_tb, x0), lp1 = MCX_LOGP(lp_tb, x1);
#if MCX_Q_SAMPLER
        const float w0 = fmaf(0.5f * z0, z0, lp0), w1 = fmaf(0.5f * z1, z1, lp1);
#else
        const float w0 = lp0 - MCX_LOGQ(lq_tb, x0), w1 = lp1 - MCX_LOGQ(lq_tb, x1);
#endif
        mh_finish(phase, it, x0, w0, 0.0f, w0 - cur_lp, ha0);
        mh_finish(phase, it + 1u, x1, w1, 0.0f, w1 - cur_lp, ha1);
    };
#endif
#if MCX_RNG == 1
    // time segments (host: mcmc_impl) for the Philox stream as well: a later segment resumes (x, w) at an even step; the
    // call index it >> 1 and the halves of its four outputs follow from `it` alone
#define MCX_MH_SEG_PHILOX (MCX_W_STATE && MCX_DIST == MCX_DIST_NORMAL)
    u32 it_first = 2u;
#if MCX_MH_SEG_PHILOX
    if (a.it_begin > 1u) {
        it_first = a.it_begin;
        if (active) {
            const float2 st = ((const float2*)a.seg_state)[g];
            cur_x = st.x;
            cur_lp = st.y;
        }
    } else
#endif
    if (total_steps >= 1u) mh_step_h(1u, ph_odd_draw, ph_odd_accept);
    for (u32 it = it_first; it <= total_steps; it += 2u) {
        const McxU4 o = mcx_philox4x32_10(McxU4{idx, it >> 1, 1u, 0u}, a.seed, MCX_PHILOX_KEY1);
#if MCX_DIST == MCX_DIST_NORMAL
        float z0, z1;
        mcx_box_muller(o.x, o.y, z0, z1);
        mh_step_h(it, z0, o.z);
        if (it + 1u <= total_steps) mh_step_h(it + 1u, z1, o.w);      // wave-uniform
#else
        mh_step_h(it, mcx_draw_proposal(o.x, pv, cdf_tb), o.z);
        if (it + 1u <= total_steps) mh_step_h(it + 1u, mcx_draw_proposal(o.y, pv, cdf_tb), o.w);
#endif
    }
#if MCX_MH_SEG_PHILOX
    if (a.seg_state != nullptr && active) {
        float2 st;
        st.x = cur_x;
        st.y = cur_lp;
        ((float2*)a.seg_state)[g] = st;
    }
#endif
#elif MCX_DIST == MCX_DIST_NORMAL
    // odd `it` consumes the z1 cached by the previous draw (it = 1: the initial draw's), even `it` draws a
    // new pair from counters 2*(it+OFFSET), +1 (distribution.rs:90-114 through shader_gen.rs:481)
    u32 it = 1u;
#if MCX_MH_BATCH
    if (a.it_begin > 1u) {                        // a later time segment: resume (x, w); the streams follow from `it`
        it = a.it_begin;                          // even (host: segments end on odd steps)
        if (active) {
            const float2 st = ((const float2*)a.seg_state)[g];
            cur_x = st.x;
            cur_lp = st.y;
        }
        st_prop = mcx_state(a.seed, idx, 2u * (it + MCX_PROP_ITER_OFFSET));
        st_acc = mcx_state(a.seed + MCX_ACCEPT_SEED_OFFSET, idx, it);
    } else
#endif
    if (total_steps >= 1u) { mh_step(1u, z_cached); it = 2u; }
#if MCX_MH_BATCH
    auto trip = [&](auto phase) {                 // steps it, it + 1
        float z0, z1;
        mcx_box_muller(mcx_pcg_out(st_prop), mcx_pcg_angle(st_prop + MCX_STATE_STEP), z0, z1);
        st_prop += 4u * MCX_STATE_STEP;
        const u32 ha0 = mcx_pcg_out(st_acc), ha1 = mcx_pcg_out(st_acc + MCX_STATE_STEP);
        st_acc += 2u * MCX_STATE_STEP;
        mh_two_steps(phase, it, z0, z1, ha0, ha1);
        it += 2u;
    };
    // burn-in trips, at most one trip that straddles the end of the burn-in, then sampling trips in blocks of MCX_FLUSH
    // (= 2 MCX_FLUSH steps, the cadence of the per-step counter) with the fold after each block: the step itself carries
    // no phase test, no flush counter and a 32-bit accept count
    // Measured on C4's kernel (ms per call, per-step tests / phased loops; profiles/r02b_mh_phased_loop_ab.txt):
    // 1 048 576 chains 8.57 / 8.97, 524 288: 4.97 / 5.07, 262 144: 2.79 / 2.79, 131 072: 1.71 / 1.59, 65 536: 1.36 / 1.15.
    // Fewer scalar instructions win where a shard leaves 2-4 waves per SIMD and each wave's own instruction stream is
    // exposed; with 8 waves per SIMD the tighter loop is SLOWER -- the waves run it in step and meet at the transcendental
    // unit and the LDS together, where the per-step scalar work of the other form keeps them apart. The host layer picks
    // 256-thread workgroups exactly for those small shards (mcx_mcmc_block_hint), so the workgroup size selects the form.
    // (Starting the waves of a SIMD 128 .. 6400 cycles apart does not substitute for it: phased 8.97 -> 8.86 ms, and it
    // costs C2 and C3 6 %.)
#ifndef MCX_MH_PHASED
#define MCX_MH_PHASED (MCX_BLOCK <= 256)
#endif
    // MCX_MH_AHEAD = 4 or 8 (phased loops only; 2 = the plain trip): that many proposals (Box-Muller pairs, table reads,
    // accept hashes and their logs) ahead of as many accept tests. The small shards that select the phased form run
    // 2-4 waves per SIMD, where one wave's dependent chain hash -> log/sqrt/sin/cos -> table read -> log -> compare is
    // exposed: more independent work per trip fills it. The accept tests themselves are unchanged (same expression,
    // same order): the chains are the same chains, bit for bit. (At full size, 8 waves per SIMD, four ahead measured
    // slower: 8.79 against 8.54 ms.) Measured on C4's shards: profiles/r03_mh_ahead.txt.
#ifndef MCX_MH_AHEAD
#define MCX_MH_AHEAD (MCX_MH_PHASED ? 4 : 2)
#endif
#if MCX_MH_AHEAD > 2
    auto trip_n = [&](auto phase) {               // steps it .. it + MCX_MH_AHEAD - 1
        constexpr int NP = MCX_MH_AHEAD / 2;
        float z[2 * NP], xs[2 * NP], ws[2 * NP];
        u32 ha[2 * NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            mcx_box_muller(mcx_pcg_out(st_prop + (4u * q) * MCX_STATE_STEP), mcx_pcg_angle(st_prop + (4u * q + 1u) * MCX_STATE_STEP),
                           z[2 * q], z[2 * q + 1]);
            ha[2 * q] = mcx_pcg_out(st_acc + (2u * q) * MCX_STATE_STEP);
            ha[2 * q + 1] = mcx_pcg_out(st_acc + (2u * q + 1u) * MCX_STATE_STEP);
        }
        st_prop += (4u * NP) * MCX_STATE_STEP;
        st_acc += (2u * NP) * MCX_STATE_STEP;
#pragma unroll
        for (int q = 0; q < 2 * NP; ++q) {
            xs[q] = MCX_AFFINE(z[q]);
            const float lp = MCX_LOGP(lp_tb, xs[q]);
#if MCX_Q_SAMPLER
            ws[q] = fmaf(0.5f * z[q], z[q], lp);
#else
            ws[q] = lp - MCX_LOGQ(lq_tb, xs[q]);
#endif
        }
#pragma unroll
        for (int q = 0; q < 2 * NP; ++q) mh_finish(phase, it + (u32)q, xs[q], ws[q], 0.0f, ws[q] - cur_lp, ha[q]);
        it += 2u * NP;
    };
#endif
#if !MCX_MH_PHASED
    // (Four proposals ahead of four accept tests instead of two: 8.79 against 8.54 ms. Not kept.)
    while (it + 1u <= total_steps) trip(McxPhase<0>{});
#endif
    while (it + 1u <= total_steps && it + 1u <= a.n_burnin) {
        u32 blk_end = it + (1u << 20);            // the 32-bit accept count of a block cannot overflow
        blk_end = blk_end < a.n_burnin ? blk_end : a.n_burnin;
        blk_end = blk_end < total_steps ? blk_end : total_steps;
#if MCX_MH_AHEAD > 2
        while (it + (MCX_MH_AHEAD - 1u) <= blk_end) trip_n(McxPhase<1>{});
#endif
        while (it + 1u <= blk_end) trip(McxPhase<1>{});
        n_accept += (u64)n_accept_blk;
        n_accept_blk = 0u;
    }
    if (it + 1u <= total_steps && it <= a.n_burnin) trip(McxPhase<0>{});
    while (it + 1u <= total_steps) {
        u32 blk_end = it + 2u * MCX_FLUSH - 1u - since_flush;         // last step of this block
        blk_end = blk_end < total_steps ? blk_end : total_steps;
#if MCX_MH_AHEAD > 2
        while (it + (MCX_MH_AHEAD - 1u) <= blk_end) trip_n(McxPhase<2>{});
#endif
        while (it + 1u <= blk_end) trip(McxPhase<2>{});
        n_accept += (u64)n_accept_blk;
        n_accept_blk = 0u;
        since_flush = 0u;
#pragma unroll
        for (int k = 0; k < MCX_K; ++k) { sum[k] += (double)acc[k]; acc[k] = 0.0f; }
    }
#else
    for (; it + 1u <= total_steps; it += 2u) {
        float z0, z1;
        mcx_box_muller(mcx_pcg_out(st_prop), mcx_pcg_angle(st_prop + MCX_STATE_STEP), z0, z1);
        st_prop += 4u * MCX_STATE_STEP;
        mh_step(it, z0);
        mh_step(it + 1u, z1);
    }
#endif
    if (it <= total_steps && it >= 2u) {
        float z0, z1;
        mcx_box_muller(mcx_pcg_out(st_prop), mcx_pcg_angle(st_prop + MCX_STATE_STEP), z0, z1);
        mh_step(it, z0);
    }
#if MCX_MH_BATCH
    if (a.seg_state != nullptr && active) {
        float2 st;
        st.x = cur_x;
        st.y = cur_lp;
        ((float2*)a.seg_state)[g] = st;
    }
#endif
#else
    for (u32 it = 1u; it <= total_steps; ++it) {
        u32 h = mcx_pcg_out(st_prop);
        st_prop += MCX_STATE_STEP;
        mh_step(it, mcx_draw_proposal(h, pv, cdf_tb));
    }
#endif
#pragma unroll
    for (int k = 0; k < MCX_K; ++k) sum[k] += (double)acc[k];
    sum[MCX_K] = (threadIdx.x & 63u) == 0u ? (double)n_accept : 0.0;
#if MCX_SECOND_MOMENTS
    {
        const double inv_steps = 1.0 / (double)a.n_steps;    // inactive waves: sums are zero
#pragma unroll
        for (int k = 0; k < MCX_NF; ++k) { const double m = sum[k] * inv_steps; sum[MCX_K + 1 + k] = m * m; }
    }
#endif

#if MCX_WALK == 3
    sum[MCX_MCMC_ROWS - 1] = active ? (double)ad_scale : 0.0;
#endif

    mcx_block_reduce_store<MCX_MCMC_ROWS>(sum, a.partials);
}

#endif          // MCX_KIND == 1

// =============================================================================================
// stage 2: fold partials[n_blocks][rows] -> out[rows] in a fixed order (one workgroup per row)
// =============================================================================================
extern "C" __global__ void __launch_bounds__(256)
mcx_fold_kernel(const double* partials, u32 n_blocks, double* out, u32* done, u32 ticket) {
    __shared__ double red[4];
    const u32 rows = gridDim.x;
    double s = 0.0;
    for (u32 i = threadIdx.x; i < n_blocks; i += 256u) s += partials[(u64)i * rows + blockIdx.x];
    s = mcx_wave_sum(s);
    if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0u) {
        out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        // Blocking calls: `out` and `done` are pinned host memory. The row's ticket is stored after its sum, ordered by a
        // system-scope fence; the host polls the `rows` tickets instead of waiting for the stream's completion signal.
        if (done != nullptr) {
            __threadfence_system();
            __hip_atomic_store(&done[blockIdx.x], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
