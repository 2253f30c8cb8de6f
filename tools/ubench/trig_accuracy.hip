// trig_accuracy.hip -- pointwise error of the sin / cos forms an integrand can be compiled with, on gfx950:
//   0 ocml sinf / cosf                      (math="precise", and "default" outside the guarded range)
//   1 v_sin_f32(x * 1/2pi)                  (__sinf / __cosf, math="fast")
//   2 v_sin_f32 of a compensated x / 2pi    (two-constant product, the phase kept to ~2^-25 revolutions)
// (and, further down, tan, pow, sinh and cosh of the default mode against their ocml routines)
// against sin / cos evaluated in f64 on the device, over 2^24 evenly spaced points of [-B, B] for a list of B.
// Prints the largest absolute error and the time per wave-call of each form.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/trig_accuracy tools/ubench/trig_accuracy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#if defined(__HIP_DEVICE_COMPILE__)                           // the device library is device-only text (hiprtc compiles it at run time)
#include "../../wgpu-monte-carlo_amd/csrc/device/mcx_args.h"
#include "../../wgpu-monte-carlo_amd/csrc/device/mcx_device.hpp"
#else
__device__ float mcx_sin(float); __device__ float mcx_cos(float); __device__ float mcx_tan(float); __device__ float mcx_pow(float, float);
__device__ float mcx_sinh(float); __device__ float mcx_cosh(float);
#endif

__device__ __forceinline__ float phase(float x) {
    const float c_hi = 0x1.45f306p-3f;                       // 1/2pi rounded to f32
    const float c_lo = 0x1.b9391p-28f;                       // 1/2pi - c_hi
    float p = x * c_hi;
    float r = __builtin_fmaf(x, c_hi, -p);                   // exact low part of the product
    r = __builtin_fmaf(x, c_lo, r);
    return __builtin_amdgcn_fractf(p) + r;                   // whole revolutions dropped before the low part is added
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float g_expect(float x, int c) {          // guard variant: the slow path marked unlikely
    if (__builtin_expect(fabsf(x) < MCX_TRIG_HW_BOUND, 1)) return c ? __builtin_amdgcn_cosf(mcx_trig_phase(x)) : __builtin_amdgcn_sinf(mcx_trig_phase(x));
    return c ? cosf(x) : sinf(x);
}
__device__ __forceinline__ float g_ballot(float x, int c) {          // guard variant: unconditional hardware path, wave-level test for the slow one
    float r = c ? __builtin_amdgcn_cosf(mcx_trig_phase(x)) : __builtin_amdgcn_sinf(mcx_trig_phase(x));
    const bool far = !(fabsf(x) < MCX_TRIG_HW_BOUND);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(far) != 0ull, 0)) { if (far) r = c ? cosf(x) : sinf(x); }
    return r;
}
__device__ __forceinline__ float g_none(float x, int c) { return c ? __builtin_amdgcn_cosf(mcx_trig_phase(x)) : __builtin_amdgcn_sinf(mcx_trig_phase(x)); }
#else
__device__ float g_expect(float, int); __device__ float g_ballot(float, int); __device__ float g_none(float, int);
#endif

template <int FORM, int COS>
__device__ __forceinline__ float eval(float x) {
    if constexpr (FORM == 4) return g_expect(x, COS);
    if constexpr (FORM == 5) return g_ballot(x, COS);
    if constexpr (FORM == 6) return g_none(x, COS);
    if constexpr (FORM == 0) return COS ? cosf(x) : sinf(x);
    else if constexpr (FORM == 1) return COS ? __cosf(x) : __sinf(x);
    else if constexpr (FORM == 2) return COS ? __builtin_amdgcn_cosf(phase(x)) : __builtin_amdgcn_sinf(phase(x));
    else return COS ? mcx_cos(x) : mcx_sin(x);                 // what math="default" compiles (device/mcx_device.hpp)
}

template <int FORM, int COS>
__global__ void __launch_bounds__(256) err_kernel(double* worst, float bound, unsigned n) {
    double w = 0.0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float x = -bound + 2.0f * bound * ((float)i / (float)(n - 1));
        double want = COS ? cos((double)x) : sin((double)x);
        double e = fabs((double)eval<FORM, COS>(x) - want);
        w = e > w ? e : w;
    }
    for (int o = 32; o; o >>= 1) { double v = __shfl_xor(w, o); w = v > w ? v : w; }
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)worst, (unsigned long long)__double_as_longlong(w));
}

template <int FORM>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters, float seed) {
    float a = seed + threadIdx.x * 1e-3f, s = 0.f;
    for (int i = 0; i < iters; ++i) { s += eval<FORM, 0>(a) + eval<FORM, 1>(a); a += 0.37f; if (a > 30.f) a -= 60.f; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FORM, int COS>
double worst_of(double* d_w, float bound) {
    hipMemset(d_w, 0, sizeof(double));
    err_kernel<FORM, COS><<<2048, 256>>>(d_w, bound, 1u << 24);
    double w; hipMemcpy(&w, d_w, sizeof(double), hipMemcpyDeviceToHost);
    return w;
}

template <int FORM>
double rate(float* d_out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    rate_kernel<FORM><<<256 * 8, 256>>>(d_out, 200, 0.1f);
    hipEventRecord(a); rate_kernel<FORM><<<256 * 8, 256>>>(d_out, 4000, 0.1f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 * 2.4e9 / (4000.0 * 8.0);      // nominal-clock cycles per (sin, cos) pair of one wave: 2048 blocks x 4 waves = 8 waves per SIMD
}

// pow(x, y) for x > 0: ocml powf against v_exp_f32(y * v_log_f32(x)), relative error over x in (0, 16], y in [-8, 8]
template <int FORM>
__global__ void __launch_bounds__(256) pow_err_kernel(double* worst, unsigned n) {
    double w = 0.0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float x = 16.0f * ((float)(i & 4095u) + 1.0f) / 4096.0f, y = -8.0f + 16.0f * (float)(i >> 12) / (float)((n >> 12) - 1);
        double want = pow((double)x, (double)y);
        float got = FORM == 0 ? powf(x, y) : mcx_pow(x, y);
        double e = fabs((double)got - want) / want;
        w = e > w ? e : w;
    }
    for (int o = 32; o; o >>= 1) { double v = __shfl_xor(w, o); w = v > w ? v : w; }
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)worst, (unsigned long long)__double_as_longlong(w));
}

template <int FORM>
__global__ void __launch_bounds__(256) pow_rate_kernel(float* out, int iters, float seed) {
    float a = seed + threadIdx.x * 1e-3f, s = 0.f;
    for (int i = 0; i < iters; ++i) { s += FORM == 0 ? powf(a, 1.5f + s * 1e-9f) : mcx_pow(a, 1.5f + s * 1e-9f); a += 0.37f; if (a > 30.f) a -= 29.5f; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// special values and negative bases: mcx_pow against powf, pair by pair (both NaN, or equal to 2e-5 relative)
__global__ void pow_edges_kernel(int* bad, float* got2) {
    const float v[] = {0.0f, -0.0f, 1.0f, -1.0f, 0.5f, -0.5f, 2.0f, -2.0f, 3.0f, -3.0f, 7.5f, -7.5f, 1e-40f, -1e-40f, 1e30f, -1e30f, 16777216.0f, -16777217.0f,
                       __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), 1.5f, -1.5f, 4.0f, -4.0f, 5.0f, -5.0f, 1e-3f, -1e-3f, 33.0f, -33.0f, 100.0f, -101.0f};
    const int n = sizeof(v) / sizeof(v[0]);
    const int i = threadIdx.x / n, j = threadIdx.x % n;
    if (i >= n) return;
    const float want = powf(v[i], v[j]), got = mcx_pow(v[i], v[j]);
    bool ok = (want != want) ? (got != got) : (want == got || fabsf(got - want) <= 2e-5f * fabsf(want));
    if (!ok) { int k = atomicAdd(bad, 1); if (k < 8) { got2[4 * k] = v[i]; got2[4 * k + 1] = v[j]; got2[4 * k + 2] = want; got2[4 * k + 3] = got; } }
}

template <int FORM>
__global__ void __launch_bounds__(256) tan_err_kernel(double* worst, float bound, unsigned n) {
    double w = 0.0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float x = -bound + 2.0f * bound * ((float)i / (float)(n - 1));
        double want = tan((double)x);
        double e = fabs((double)(FORM == 0 ? tanf(x) : mcx_tan(x)) - want) / (1.0 + want * want);      // error as an angle: d(tan) = (1 + tan^2) d(x)
        w = e > w ? e : w;
    }
    for (int o = 32; o; o >>= 1) { double v = __shfl_xor(w, o); w = v > w ? v : w; }
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)worst, (unsigned long long)__double_as_longlong(w));
}

// sinh / cosh: relative error over [-bound, bound] (sinh: relative to max(|sinh x|, 1e-30))
template <int FORM, int COSH>
__global__ void __launch_bounds__(256) hyp_err_kernel(double* worst, float bound, unsigned n) {
    double w = 0.0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float x = -bound + 2.0f * bound * ((float)i / (float)(n - 1));
        double want = COSH ? cosh((double)x) : sinh((double)x);
        float got = COSH ? (FORM == 0 ? coshf(x) : mcx_cosh(x)) : (FORM == 0 ? sinhf(x) : mcx_sinh(x));
        double e = fabs((double)got - want) / fmax(fabs(want), 1e-30);
        if (want > 3.4e38 || want < -3.4e38) e = (fabsf(got) > 3.39e38f && (got > 0) == (want > 0)) ? 0.0 : 1.0;     // overflow: the infinity, or the last finite floats at its edge
        w = e > w ? e : w;
    }
    for (int o = 32; o; o >>= 1) { double v = __shfl_xor(w, o); w = v > w ? v : w; }
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)worst, (unsigned long long)__double_as_longlong(w));
}

template <int FORM>
__global__ void __launch_bounds__(256) hyp_rate_kernel(float* out, int iters, float seed) {
    float a = seed + threadIdx.x * 1e-3f, s = 0.f;
    for (int i = 0; i < iters; ++i) { s += FORM == 0 ? sinhf(a) + coshf(a) : mcx_sinh(a) + mcx_cosh(a); a += 0.37f; if (a > 30.f) a -= 60.f; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

void hyp_report(double* d_w, float* d_out) {
    for (float b : {1e-3f, 0.5f, 4.f, 30.f, 88.f, 95.f}) {
        double w[4];
        hipMemset(d_w, 0, 8); hyp_err_kernel<0, 0><<<2048, 256>>>(d_w, b, 1u << 24); hipMemcpy(&w[0], d_w, 8, hipMemcpyDeviceToHost);
        hipMemset(d_w, 0, 8); hyp_err_kernel<1, 0><<<2048, 256>>>(d_w, b, 1u << 24); hipMemcpy(&w[1], d_w, 8, hipMemcpyDeviceToHost);
        hipMemset(d_w, 0, 8); hyp_err_kernel<0, 1><<<2048, 256>>>(d_w, b, 1u << 24); hipMemcpy(&w[2], d_w, 8, hipMemcpyDeviceToHost);
        hipMemset(d_w, 0, 8); hyp_err_kernel<1, 1><<<2048, 256>>>(d_w, b, 1u << 24); hipMemcpy(&w[3], d_w, 8, hipMemcpyDeviceToHost);
        printf("[-%g, %g] worst relative error: sinh ocml %.3e mcx_sinh %.3e   cosh ocml %.3e mcx_cosh %.3e\n", b, b, w[0], w[1], w[2], w[3]);
    }
    float ms[2]; hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hyp_rate_kernel<0><<<2048, 256>>>(d_out, 100, 0.1f);
    hipEventRecord(a); hyp_rate_kernel<0><<<2048, 256>>>(d_out, 4000, 0.1f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms[0], a, b);
    hipEventRecord(a); hyp_rate_kernel<1><<<2048, 256>>>(d_out, 4000, 0.1f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms[1], a, b);
    printf("nominal cycles per wave per (sinh, cosh) pair: ocml %.1f  mcx %.1f\n", ms[0] * 1e-3 * 2.4e9 / 32000.0, ms[1] * 1e-3 * 2.4e9 / 32000.0);
}

void pow_report(double* d_w, float* d_out) {
    {
        int* d_bad; hipMalloc(&d_bad, 4); hipMemset(d_bad, 0, 4);
        pow_edges_kernel<<<1, 1024>>>(d_bad, d_out);
        int bad; float ex[32]; hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost); hipMemcpy(ex, d_out, sizeof(ex), hipMemcpyDeviceToHost);
        printf("pow special values and negative bases: %d of 1024 (x, y) pairs differ from powf\n", bad);
        for (int k = 0; k < bad && k < 8; ++k) printf("   x %g  y %g  powf %g  mcx_pow %g\n", ex[4 * k], ex[4 * k + 1], ex[4 * k + 2], ex[4 * k + 3]);
        for (float b : {1.5f, 100.f, 9.9e5f}) {
            double w[2];
            hipMemset(d_w, 0, 8); tan_err_kernel<0><<<2048, 256>>>(d_w, b, 1u << 24); hipMemcpy(&w[0], d_w, 8, hipMemcpyDeviceToHost);
            hipMemset(d_w, 0, 8); tan_err_kernel<1><<<2048, 256>>>(d_w, b, 1u << 24); hipMemcpy(&w[1], d_w, 8, hipMemcpyDeviceToHost);
            printf("tan on [-%g, %g], worst error as an angle: ocml %.3e  mcx_tan %.3e\n", b, b, w[0], w[1]);
        }
    }
    double w[2]; float ms[2];
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipMemset(d_w, 0, 8); pow_err_kernel<0><<<2048, 256>>>(d_w, 1u << 24); hipMemcpy(&w[0], d_w, 8, hipMemcpyDeviceToHost);
    hipMemset(d_w, 0, 8); pow_err_kernel<1><<<2048, 256>>>(d_w, 1u << 24); hipMemcpy(&w[1], d_w, 8, hipMemcpyDeviceToHost);
    pow_rate_kernel<0><<<2048, 256>>>(d_out, 100, 0.6f);
    hipEventRecord(a); pow_rate_kernel<0><<<2048, 256>>>(d_out, 4000, 0.6f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms[0], a, b);
    hipEventRecord(a); pow_rate_kernel<1><<<2048, 256>>>(d_out, 4000, 0.6f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms[1], a, b);
    printf("pow(x, y), x in (0, 16], y in [-8, 8]: worst relative error ocml %.3e  mcx_pow %.3e;  nominal cycles per wave-call ocml %.1f  mcx_pow %.1f\n", w[0], w[1],
           ms[0] * 1e-3 * 2.4e9 / 32000.0, ms[1] * 1e-3 * 2.4e9 / 32000.0);
}

int main() {
    double* d_w; float* d_out;
    hipMalloc(&d_w, sizeof(double)); hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
    const float bounds[] = {0.01f, 1.f, 3.14159274f, 8.f, 32.f, 128.f, 512.f, 1600.f, 1e5f, 9.9e5f, 1e7f, 1e9f};
    printf("%-10s %-5s %14s %14s %14s %14s\n", "bound", "fn", "ocml", "v_sin(x/2pi)", "comp. fract", "mcx_sin/cos");
    for (float b : bounds) {
        printf("%-10.4g %-5s %14.3e %14.3e %14.3e %14.3e\n", b, "sin", worst_of<0, 0>(d_w, b), worst_of<1, 0>(d_w, b), worst_of<2, 0>(d_w, b), worst_of<3, 0>(d_w, b));
        printf("%-10.4g %-5s %14.3e %14.3e %14.3e %14.3e\n", b, "cos", worst_of<0, 1>(d_w, b), worst_of<1, 1>(d_w, b), worst_of<2, 1>(d_w, b), worst_of<3, 1>(d_w, b));
    }
    printf("nominal cycles per wave per (sin, cos) pair, loop overhead included: ocml %.1f  v_sin %.1f  comp. fract %.1f  mcx_sin/cos %.1f\n", rate<0>(d_out), rate<1>(d_out),
           rate<2>(d_out), rate<3>(d_out));
    printf("guard variants, same units: unlikely-marked %.1f  wave-level test %.1f  no guard %.1f\n", rate<4>(d_out), rate<5>(d_out), rate<6>(d_out));
    pow_report(d_w, d_out);
    hyp_report(d_w, d_out);
    return 0;
}
