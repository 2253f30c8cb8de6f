// aot_default.hip -- ahead-of-time build check of the kernel skeletons (hipcc --offload-arch=gfx950).
// Same translation-unit layout that mcx_runtime.cpp assembles for hiprtc, with the default
// moment functions x, x^2, x^3, x^4 standing in for the emitted user functions.
#include <hip/hip_runtime.h>
#define MCX_K 4
#ifndef MCX_DIST
#define MCX_DIST 1
#endif
#ifndef MCX_AOT_KIND
#define MCX_AOT_KIND 0
#endif
#if MCX_AOT_KIND == 1 || MCX_DIST == 3
#define MCX_BLOCK 1024
#else
#define MCX_BLOCK 256
#endif
#define MCX_WEIGHT 0
#include "device/mcx_args.h"
#include "device/mcx_device.hpp"
MCX_DEV float user_func_0(float x) { return x; }
MCX_DEV float user_func_1(float x) { return x * x; }
MCX_DEV float user_func_2(float x) { return x * x * x; }
MCX_DEV float user_func_3(float x) { return (x * x) * (x * x); }
template <int S> MCX_DEV void mcx_eval_all(float x, float w, float* acc) {
    acc[0 * S] += mcx_b2f(user_func_0(x)) * w;
    acc[1 * S] += mcx_b2f(user_func_1(x)) * w;
    acc[2 * S] += mcx_b2f(user_func_2(x)) * w;
    acc[3 * S] += mcx_b2f(user_func_3(x)) * w;
}
#include "device/mcx_kernels.hpp"
