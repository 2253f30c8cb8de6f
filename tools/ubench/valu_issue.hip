// valu_issue.hip -- what one SIMD of gfx950 sustains in vector-ALU issue, measured INSIDE the kernel.
//
// Round 1's valu_rates.hip timed ~0.3 ms launches with HIP events at the nominal 2.4 GHz and read 51.5 T lane-ops/s
// for plain v_fma_f32 (65 % of the 78.6 T spec figure) -- while real kernels of this repo issue at up to 70 % of spec
// (C5: 83 VALU per sample at 6.7e11 samples/s). A bound that a real kernel beats is not a bound; this version removes
// what distorted it:
//   * 128 VALU instructions per loop trip (the round-1 loop had 8 VALU + 3 scalar loop instructions per trip),
//   * cycles from s_memtime inside each wave (shader clock) and the clock itself from s_memrealtime (100 MHz), so
//     neither launch ramp / tail nor the DVFS clock enters the cycles-per-instruction figure,
//   * multi-millisecond launches, and an explicit sweep of resident waves per SIMD (1, 2, 4, 8) and of the number
//     of independent dependency chains per wave (1, 2, 4, 8).
// Output: real shader cycles per wave-instruction per SIMD, the clock held, and the resulting lane-op rate.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

enum { OP_FMA = 0, OP_FMA_SGPR, OP_ADD_F32, OP_XOR, OP_ADD_U32, OP_LSHR, OP_MIX, OP_MUL_LO, OP_CVT, OP_LOG, OP_PK_FMA, OP_MOV,
       OP_FMA_INLINE, OP_MUL_SGPR, OP_MUL_VGPR, OP_ADD_SGPR, OP_FMAC, OP_FMA_DISTINCT, OP_FMA_SGPR_MID, OP_XOR_SGPR, OP_MUL_LIT, OP_FMA_SELF,
       OP_MUL_LO_SGPR, OP_CNDMASK_VCC, OP_CNDMASK_SPAIR, OP_CMP_VCC, OP_CMP_SPAIR, OP_LSHL_ADD_S, OP_LSHL_ADD_V, OP_ALIGNBIT_S, OP_ALIGNBIT_V, OP_ADD_LIT, OP_MED3, OP_MAX_INLINE, OP_SQRT_ABS,
       OP_MAX_V, OP_MIN_V, OP_SUB_F32, OP_AND, OP_OR, OP_LSHL_V, OP_BFE, OP_CVT_U32_F32, OP_FLOOR, OP_EXP, OP_RCP, OP_LOG_FMA3, OP_LOG_FMA1, OP_C2MIX, OP_MAX_U32, OP_SUB_U32, OP_CVT_I32, OP_MUL_HI, OP_MAD_U32_U24, OP_AND_OR, OP_BFI, OP_ADD3, OP_LDEXP, OP_CMP_E64_VCC };

template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) issue_kernel(unsigned long long* stamps, int iters, float seedf, float* sink) {
    float a[8];
    unsigned u[8];
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        a[c] = seedf + threadIdx.x * 1e-3f + c;
        u[c] = (unsigned)(threadIdx.x * 2654435761u + c * 40503u) | 1u;
        p[c] = v2{a[c], a[c] + 0.5f};
    }
    float vs = seedf * 0.999f;      // a second VGPR operand
    unsigned long long mask64 = 0x5555555555555555ull;
    asm volatile("s_mov_b64 %0, %0" : "+s"(mask64));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#define ONE(c)                                                                                                   \
    if constexpr (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(c) % CHAINS]) : "v"(vs), "v"(a[((c) + 1) % 8])); \
    else if constexpr (OP == OP_FMA_SGPR) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[(c) % CHAINS]) : "s"(seedf));        \
    else if constexpr (OP == OP_ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[(c) % CHAINS]) : "v"(vs));               \
    else if constexpr (OP == OP_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]));     \
    else if constexpr (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8])); \
    else if constexpr (OP == OP_LSHR) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(u[(c) % CHAINS]));                          \
    else if constexpr (OP == OP_MIX) { if ((c) & 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8])); \
                                       else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[(c) % CHAINS]) : "v"(vs)); }        \
    else if constexpr (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8])); \
    else if constexpr (OP == OP_CVT) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[(c) % CHAINS]) : "v"(u[(c) % CHAINS]));      \
    else if constexpr (OP == OP_LOG) asm volatile("v_log_f32 %0, %0" : "+v"(a[(c) % CHAINS]));                                 \
    else if constexpr (OP == OP_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[(c) % CHAINS]) : "v"(p[((c) + 1) % 8])); \
    else if constexpr (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[(c) % CHAINS]) : "v"(vs));                         \
    else if constexpr (OP == OP_FMA_INLINE) asm volatile("v_fma_f32 %0, %0, 0.5, %0" : "+v"(a[(c) % CHAINS]));                   \
    else if constexpr (OP == OP_MUL_SGPR) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[(c) % CHAINS]) : "s"(seedf));            \
    else if constexpr (OP == OP_MUL_VGPR) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[(c) % CHAINS]) : "v"(vs));               \
    else if constexpr (OP == OP_ADD_SGPR) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[(c) % CHAINS]) : "s"(seedf));            \
    else if constexpr (OP == OP_FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[(c) % CHAINS]) : "v"(vs), "v"(a[((c) + 1) % 8])); \
    else if constexpr (OP == OP_FMA_DISTINCT) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[(c) % CHAINS]) : "v"(vs), "v"(a[((c) + 1) % 8]), "v"(a[((c) + 2) % 8])); \
    else if constexpr (OP == OP_FMA_SGPR_MID) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[(c) % CHAINS]) : "v"(vs), "s"(seedf), "v"(a[((c) + 2) % 8])); \
    else if constexpr (OP == OP_XOR_SGPR) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[(c) % CHAINS]) : "s"(iters));           \
    else if constexpr (OP == OP_MUL_LIT) asm volatile("v_mul_f32 %0, 0x3f317218, %0" : "+v"(a[(c) % CHAINS]));                   \
    else if constexpr (OP == OP_FMA_SELF) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[(c) % CHAINS]) : "v"(vs));         \
    else if constexpr (OP == OP_MUL_LO_SGPR) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[(c) % CHAINS]) : "s"(iters));      \
    else if constexpr (OP == OP_CNDMASK_VCC) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]) : "vcc"); \
    else if constexpr (OP == OP_CNDMASK_SPAIR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]), "s"(mask64)); \
    else if constexpr (OP == OP_CMP_VCC) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(a[(c) % CHAINS]), "v"(a[((c) + 1) % 8]) : "vcc"); \
    else if constexpr (OP == OP_CMP_SPAIR) asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(mask64) : "v"(a[(c) % CHAINS]), "v"(a[((c) + 1) % 8])); \
    else if constexpr (OP == OP_LSHL_ADD_S) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[(c) % CHAINS]) : "s"(iters)); \
    else if constexpr (OP == OP_LSHL_ADD_V) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8])); \
    else if constexpr (OP == OP_ALIGNBIT_S) asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(u[(c) % CHAINS]) : "s"(iters));  \
    else if constexpr (OP == OP_ALIGNBIT_V) asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8])); \
    else if constexpr (OP == OP_ADD_LIT) asm volatile("v_add_u32 %0, 0x577a1e13, %0" : "+v"(u[(c) % CHAINS]));                   \
    else if constexpr (OP == OP_MED3) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(a[(c) % CHAINS]) : "v"(vs));                \
    else if constexpr (OP == OP_MAX_INLINE) asm volatile("v_max_f32 %0, 0.5, %0" : "+v"(a[(c) % CHAINS]));                       \
    else if constexpr (OP == OP_SQRT_ABS) asm volatile("v_sqrt_f32 %0, |%0|" : "+v"(a[(c) % CHAINS]));                          \
    else if constexpr (OP == OP_MAX_V) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[(c) % CHAINS]) : "v"(vs));                   \
    else if constexpr (OP == OP_MIN_V) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[(c) % CHAINS]) : "v"(vs));                   \
    else if constexpr (OP == OP_SUB_F32) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[(c) % CHAINS]) : "v"(vs));                 \
    else if constexpr (OP == OP_AND) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]));       \
    else if constexpr (OP == OP_OR) asm volatile("v_or_b32 %0, %1, %0" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]));         \
    else if constexpr (OP == OP_LSHL_V) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8])); \
    else if constexpr (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(u[(c) % CHAINS]));                              \
    else if constexpr (OP == OP_CVT_U32_F32) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[(c) % CHAINS]) : "v"(a[(c) % CHAINS])); \
    else if constexpr (OP == OP_FLOOR) asm volatile("v_floor_f32 %0, %0" : "+v"(a[(c) % CHAINS]));                                \
    else if constexpr (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[(c) % CHAINS]));                                    \
    else if constexpr (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[(c) % CHAINS]));                                    \
    else if constexpr (OP == OP_LOG_FMA3) { if (((c) & 3) == 0) asm volatile("v_log_f32 %0, %0" : "+v"(a[(c) % CHAINS]));         \
                                            else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[(c) % CHAINS]) : "v"(vs)); }   \
    else if constexpr (OP == OP_LOG_FMA1) { if (((c) & 1) == 0) asm volatile("v_log_f32 %0, %0" : "+v"(a[(c) % CHAINS]));         \
                                            else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[(c) % CHAINS]) : "v"(vs)); }   \
    else if constexpr (OP == OP_C2MIX) {                                                                                         \
        /* the instruction mix of the C2 loop (34 per pair) in 32 slots: 4 transcendental, 2 mul_lo, 1 cvt, 1 alignbit, 1 max, 23 full-rate */ \
        constexpr int k = (c) % 32;                                                                                              \
        if (k == 5 || k == 13 || k == 21 || k == 29) asm volatile("v_log_f32 %0, %0" : "+v"(a[(c) % CHAINS]));                    \
        else if (k == 3 || k == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]));      \
        else if (k == 7) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[(c) % CHAINS]) : "v"(u[(c) % CHAINS]));                    \
        else if (k == 15) asm volatile("v_alignbit_b32 %0, %1, %0, 9" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]));          \
        else if (k == 17) asm volatile("v_max_f32 %0, 0.5, %0" : "+v"(a[(c) % CHAINS]));                                         \
        else if (k & 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]));                    \
        else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[(c) % CHAINS]) : "v"(vs)); }                                       \
    else if constexpr (OP == OP_MAX_U32) asm volatile("v_max_u32 %0, 1, %0" : "+v"(u[(c) % CHAINS]));                             \
    else if constexpr (OP == OP_SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]));    \
    else if constexpr (OP == OP_CVT_I32) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[(c) % CHAINS]) : "v"(u[(c) % CHAINS]));    \
    else if constexpr (OP == OP_MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]));  \
    else if constexpr (OP == OP_MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8])); \
    else if constexpr (OP == OP_AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]), "v"(u[((c) + 2) % 8])); \
    else if constexpr (OP == OP_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]), "v"(u[((c) + 2) % 8])); \
    else if constexpr (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]), "v"(u[((c) + 2) % 8])); \
    else if constexpr (OP == OP_LDEXP) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[(c) % CHAINS]) : "v"(u[((c) + 1) % 8]));    \
    else if constexpr (OP == OP_CMP_E64_VCC) asm volatile("v_cmp_gt_f32_e64 vcc, %0, %1" : : "v"(a[(c) % CHAINS]), "v"(a[((c) + 1) % 8]) : "vcc");
#define EIGHT(b) ONE(b) ONE(b + 1) ONE(b + 2) ONE(b + 3) ONE(b + 4) ONE(b + 5) ONE(b + 6) ONE(b + 7)
        EIGHT(0) EIGHT(8) EIGHT(16) EIGHT(24) EIGHT(32) EIGHT(40) EIGHT(48) EIGHT(56)
        EIGHT(64) EIGHT(72) EIGHT(80) EIGHT(88) EIGHT(96) EIGHT(104) EIGHT(112) EIGHT(120)
#undef EIGHT
#undef ONE
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += a[c] + (float)u[c] + p[c].x + p[c].y;
    if (s == 123.456f) sink[0] = s;
}

template <int OP, int CHAINS>
static void run(const char* name, int waves_per_simd, unsigned long long* d_stamps, float* d_sink) {
    const int blocks = 256 * waves_per_simd, iters = (OP == OP_LOG || OP == OP_EXP || OP == OP_RCP || OP == OP_SQRT_ABS) ? 6000 : 12000;
    hipLaunchKernelGGL((issue_kernel<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, d_stamps, 200, 1.0001f, d_sink);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((issue_kernel<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, d_stamps, iters, 1.0001f, d_sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st(2 * blocks * 4);
    hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)st[2 * w]); clk.push_back((double)st[2 * w] / (double)st[2 * w + 1] * 0.1); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double med_cyc = cyc[cyc.size() / 2], ghz = clk[clk.size() / 2];
    const double instr_per_wave = (double)iters * 128.0;
    // throughput from the wall clock of the launch (20-40 ms: ramp and tail are negligible) and the clock the waves saw;
    // a wave's own elapsed cycles tell how many of the launch's waves were resident together
    const double total_instr = instr_per_wave * blocks * 4.0;
    const double lanes_per_s = total_instr * 64.0 / (ms * 1e-3);
    const double cyc_per_instr = 1024.0 * ghz * 1e9 * (ms * 1e-3) / total_instr;
    const double resident = waves_per_simd * med_cyc / (ghz * 1e9 * ms * 1e-3);
    printf("%-24s chains %d  waves/SIMD launched %d (resident together %.1f)  %7.3f ms  %5.2f cycles per wave-instruction per SIMD  clock %.3f GHz  %6.2f T lane-ops/s (%.0f %% of 78.6)\n",
           name, CHAINS, waves_per_simd, resident, ms, cyc_per_instr, ghz, lanes_per_s / 1e12, lanes_per_s / 78.6e12 * 100);
    fflush(stdout);
}

int main() {
    unsigned long long* d_stamps; float* d_sink;
    hipMalloc(&d_stamps, 2 * 256 * 8 * 4 * 8);
    hipMalloc(&d_sink, 64);
    for (int w : {1, 2, 4, 8}) run<OP_FMA, 8>("v_fma_f32 3 VGPR src", w, d_stamps, d_sink);
    for (int w : {1, 2, 4, 8}) run<OP_FMA_SGPR, 8>("v_fma_f32 a,a,s,a", w, d_stamps, d_sink);
    run<OP_FMA_SGPR, 1>("v_fma_f32 a,a,s,a", 8, d_stamps, d_sink);
    run<OP_FMA_SGPR, 2>("v_fma_f32 a,a,s,a", 8, d_stamps, d_sink);
    run<OP_FMA_SGPR, 4>("v_fma_f32 a,a,s,a", 8, d_stamps, d_sink);
    for (int w : {8}) run<OP_FMA_SELF, 8>("v_fma a,a,v,a", w, d_stamps, d_sink);
    for (int w : {8}) run<OP_FMA_DISTINCT, 8>("v_fma d,v,v,v distinct", w, d_stamps, d_sink);
    for (int w : {8}) run<OP_FMA_SGPR_MID, 8>("v_fma d,v,s,v", w, d_stamps, d_sink);
    for (int w : {8}) run<OP_FMA_INLINE, 8>("v_fma a,a,0.5,a", w, d_stamps, d_sink);
    for (int w : {8}) run<OP_FMAC, 8>("v_fmac_f32 (VOP2)", w, d_stamps, d_sink);
    for (int w : {8}) run<OP_MUL_VGPR, 8>("v_mul_f32 a,v,a", w, d_stamps, d_sink);
    for (int w : {8}) run<OP_MUL_SGPR, 8>("v_mul_f32 a,s,a", w, d_stamps, d_sink);
    for (int w : {8}) run<OP_MUL_LIT, 8>("v_mul_f32 a,literal,a", w, d_stamps, d_sink);
    for (int w : {8}) run<OP_ADD_SGPR, 8>("v_add_f32 a,s,a", w, d_stamps, d_sink);
    for (int w : {8}) run<OP_XOR_SGPR, 8>("v_xor_b32 a,s,a", w, d_stamps, d_sink);
    run<OP_MUL_LO_SGPR, 8>("v_mul_lo_u32 a,a,s", 8, d_stamps, d_sink);
    run<OP_CNDMASK_VCC, 8>("v_cndmask e32 (vcc)", 8, d_stamps, d_sink);
    run<OP_CNDMASK_SPAIR, 8>("v_cndmask e64 s[pair]", 8, d_stamps, d_sink);
    run<OP_CMP_VCC, 8>("v_cmp_gt_f32 -> vcc", 8, d_stamps, d_sink);
    run<OP_CMP_SPAIR, 8>("v_cmp_gt_f32 -> s[pair]", 8, d_stamps, d_sink);
    run<OP_LSHL_ADD_S, 8>("v_lshl_add_u32 a,a,3,s", 8, d_stamps, d_sink);
    run<OP_LSHL_ADD_V, 8>("v_lshl_add_u32 a,a,3,v", 8, d_stamps, d_sink);
    run<OP_ALIGNBIT_S, 8>("v_alignbit a,s,a,9", 8, d_stamps, d_sink);
    run<OP_ALIGNBIT_V, 8>("v_alignbit a,v,a,9", 8, d_stamps, d_sink);
    run<OP_ADD_LIT, 8>("v_add_u32 a,literal,a", 8, d_stamps, d_sink);
    run<OP_MED3, 8>("v_med3_f32 a,a,0,v", 8, d_stamps, d_sink);
    run<OP_MAX_INLINE, 8>("v_max_f32 a,0.5,a", 8, d_stamps, d_sink);
    run<OP_SQRT_ABS, 8>("v_sqrt_f32 a,|a|", 8, d_stamps, d_sink);
    run<OP_MAX_V, 8>("v_max_f32 a,v,a", 8, d_stamps, d_sink);
    run<OP_MIN_V, 8>("v_min_f32 a,v,a", 8, d_stamps, d_sink);
    run<OP_MAX_U32, 8>("v_max_u32 a,1,a", 8, d_stamps, d_sink);
    run<OP_SUB_F32, 8>("v_sub_f32 a,v,a", 8, d_stamps, d_sink);
    run<OP_SUB_U32, 8>("v_sub_u32", 8, d_stamps, d_sink);
    run<OP_AND, 8>("v_and_b32", 8, d_stamps, d_sink);
    run<OP_OR, 8>("v_or_b32", 8, d_stamps, d_sink);
    run<OP_LSHL_V, 8>("v_lshrrev_b32 a,v,a", 8, d_stamps, d_sink);
    run<OP_BFE, 8>("v_bfe_u32 a,a,3,9", 8, d_stamps, d_sink);
    run<OP_AND_OR, 8>("v_and_or_b32", 8, d_stamps, d_sink);
    run<OP_BFI, 8>("v_bfi_b32", 8, d_stamps, d_sink);
    run<OP_ADD3, 8>("v_add3_u32", 8, d_stamps, d_sink);
    run<OP_MAD_U32_U24, 8>("v_mad_u32_u24", 8, d_stamps, d_sink);
    run<OP_MUL_HI, 8>("v_mul_hi_u32", 8, d_stamps, d_sink);
    run<OP_CVT_U32_F32, 8>("v_cvt_u32_f32", 8, d_stamps, d_sink);
    run<OP_CVT_I32, 8>("v_cvt_f32_i32", 8, d_stamps, d_sink);
    run<OP_FLOOR, 8>("v_floor_f32", 8, d_stamps, d_sink);
    run<OP_LDEXP, 8>("v_ldexp_f32", 8, d_stamps, d_sink);
    run<OP_CMP_E64_VCC, 8>("v_cmp_gt_f32_e64 vcc", 8, d_stamps, d_sink);
    run<OP_EXP, 8>("v_exp_f32", 8, d_stamps, d_sink);
    run<OP_RCP, 8>("v_rcp_f32", 8, d_stamps, d_sink);
    run<OP_LOG_FMA3, 8>("1 v_log + 3 v_fma", 8, d_stamps, d_sink);
    run<OP_LOG_FMA1, 8>("1 v_log + 1 v_fma", 8, d_stamps, d_sink);
    run<OP_C2MIX, 8>("C2 loop mix (x4 per trip)", 8, d_stamps, d_sink);
    run<OP_C2MIX, 8>("C2 loop mix (x4 per trip)", 4, d_stamps, d_sink);
    for (int w : {4, 8}) run<OP_ADD_F32, 8>("v_add_f32", w, d_stamps, d_sink);
    for (int w : {4, 8}) run<OP_MOV, 8>("v_mov_b32", w, d_stamps, d_sink);
    for (int w : {4, 8}) run<OP_XOR, 8>("v_xor_b32", w, d_stamps, d_sink);
    for (int w : {4, 8}) run<OP_ADD_U32, 8>("v_add_u32", w, d_stamps, d_sink);
    for (int w : {4, 8}) run<OP_LSHR, 8>("v_lshrrev_b32 imm", w, d_stamps, d_sink);
    for (int w : {4, 8}) run<OP_MIX, 8>("fma / xor alternating", w, d_stamps, d_sink);
    for (int w : {4, 8}) run<OP_MUL_LO, 8>("v_mul_lo_u32", w, d_stamps, d_sink);
    for (int w : {4, 8}) run<OP_CVT, 8>("v_cvt_f32_u32", w, d_stamps, d_sink);
    for (int w : {4, 8}) run<OP_LOG, 8>("v_log_f32", w, d_stamps, d_sink);
    for (int w : {4, 8}) run<OP_PK_FMA, 8>("v_pk_fma_f32", w, d_stamps, d_sink);
    return 0;
}
