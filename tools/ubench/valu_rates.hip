// valu_rates.hip -- measures the issue cost of the VALU instructions the Monte-Carlo kernels are made
// of, on gfx950, with every SIMD busy (8 waves/SIMD, 8 independent chains per lane). Output: cycles
// per wave-instruction per SIMD, relative to the 2.4 GHz nominal clock and relative to v_fma_f32.
// Used to weight the algorithmic lane-op count in DESIGN.md "Roofline" (SURVEY.md 8(d) asks for the
// v_mul_lo_u32 cost to be microbenchmarked).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHAINS 8
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters, float seedf) {
    float a[CHAINS];
    unsigned u[CHAINS];
    double d[CHAINS];
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p[CHAINS];
    for (int c = 0; c < CHAINS; ++c) {
        a[c] = seedf + threadIdx.x * 1e-3f + c;
        u[c] = (unsigned)(threadIdx.x * 2654435761u + c * 40503u) | 1u;
        d[c] = a[c];
        p[c] = v2{a[c], a[c] + 0.5f};
    }
    for (int i = 0; i < iters; ++i) {
#define STEP(c)                                                                                         \
    if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[c]) : "v"(seedf));          \
    else if constexpr (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS])); \
    else if constexpr (OP == 2) asm volatile("v_log_f32 %0, %0" : "+v"(a[c]));                          \
    else if constexpr (OP == 3) asm volatile("v_sin_f32 %0, %0" : "+v"(a[c]));                          \
    else if constexpr (OP == 4) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[c]));                         \
    else if constexpr (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[c]));               \
    else if constexpr (OP == 6) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[c]));                      \
    else if constexpr (OP == 7) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[c]) : "v"(u[c]));         \
    else if constexpr (OP == 8) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS])); \
    else if constexpr (OP == 9) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]));     \
    else if constexpr (OP == 10) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[c]));                         \
    else if constexpr (OP == 11) asm volatile("v_exp_f32 %0, %0" : "+v"(a[c]));                         \
    else if constexpr (OP == 12) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS])); \
    else if constexpr (OP == 13) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[c]) : "v"(u[c]), "v"(u[(c + 1) % CHAINS]) : "vcc"); \
    else if constexpr (OP == 14) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[c]));                 \
    else if constexpr (OP == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]));  \
    else if constexpr (OP == 16) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[c]) : "v"(p[(c + 1) % CHAINS]), "v"(p[(c + 2) % CHAINS])); \
    else if constexpr (OP == 17) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[c]) : "v"(p[(c + 1) % CHAINS]), "v"(p[(c + 2) % CHAINS])); \
    else if constexpr (OP == 18) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(p[(c + 1) % CHAINS])); \
    else if constexpr (OP == 19) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[c]) : "v"(a[(c + 1) % CHAINS]), "v"(a[(c + 2) % CHAINS])); \
    else if constexpr (OP == 20) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[c]) : "v"(p[(c + 1) % CHAINS]), "v"(p[(c + 2) % CHAINS]));
        REP8(STEP)
#undef STEP
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += a[c] + (float)u[c] + (float)d[c] + p[c].x + p[c].y;
    if (s == 123.456f) out[0] = s;
}

template <int OP>
static void run(const char* name, float* d_out, double* base) {
    const int blocks = 256 * 8, threads = 256, iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<OP><<<blocks, threads>>>(d_out, 64, 1.0001f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        rate_kernel<OP><<<blocks, threads>>>(d_out, iters, 1.0001f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double wave_instrs = (double)blocks * (threads / 64) * iters * CHAINS;
    double simd_cycles = 256.0 * 4.0 * 2.4e9 * (best * 1e-3);
    double cyc = simd_cycles / wave_instrs;      // SIMD cycles (at 2.4 GHz nominal) per wave-instruction
    if (OP == 0) *base = cyc;
    printf("%-16s %8.3f ms  %6.2f cyc/wave-instr/SIMD @2.4GHz   x%.2f of v_fma_f32   %.2f Tlane-instr/s\n", name, best, cyc,
           cyc / *base, wave_instrs * 64 / (best * 1e-3) / 1e12);
}

int main() {
    float* d_out; hipMalloc(&d_out, 64);
    double base = 1;
    run<0>("v_fma_f32", d_out, &base);
    run<15>("v_add_u32", d_out, &base);
    run<9>("v_xor_b32", d_out, &base);
    run<8>("v_lshrrev_b32", d_out, &base);
    run<7>("v_cvt_f32_u32", d_out, &base);
    run<1>("v_mul_lo_u32", d_out, &base);
    run<12>("v_mul_u32_u24", d_out, &base);
    run<13>("v_mad_u64_u32", d_out, &base);
    run<2>("v_log_f32", d_out, &base);
    run<11>("v_exp_f32", d_out, &base);
    run<3>("v_sin_f32", d_out, &base);
    run<4>("v_sqrt_f32", d_out, &base);
    run<10>("v_rcp_f32", d_out, &base);
    run<5>("v_pk_fma_f32", d_out, &base);
    run<6>("v_add_f64", d_out, &base);
    run<14>("v_fma_f64", d_out, &base);
    run<19>("v_fma_f32 3src", d_out, &base);
    run<16>("v_pk_fma_f32 3src", d_out, &base);
    run<20>("v_pk_fma opsel", d_out, &base);
    run<17>("v_pk_mul_f32 2src", d_out, &base);
    run<18>("v_pk_add_f32 2src", d_out, &base);
    return 0;
}
