/* cabi_client.c -- a plain-C client of libmcx.so (no Python, no torch): what a foreign-language binding of the
 * reference (cgo / JNI / Rust FFI) would do. Built and run by tests/test_gpu_cabi_client.py.
 *
 *   gcc -O2 -I include tests/cabi_client.c -o cabi_client -ldl
 *   ./cabi_client path/to/libmcx.so
 *
 * Prints "OK ..." lines; exits non-zero on any mismatch. Without a GPU it stops after the planning checks
 * with exit code 3.
 *
 * -DMCX_CLIENT_OLD_LAYOUT builds the same client as a caller compiled against an OLDER mcx.h would look to the
 * library: its module desc ends with `unit_params` (the first release of the struct) and says so in struct_size.
 * libmcx must zero-fill what such a caller does not know and give the same sums (include/mcx.h, "ABI versioning").
 */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "mcx.h"

#ifdef MCX_CLIENT_OLD_LAYOUT
typedef struct client_module_desc {            /* mcx_module_desc as of its first release */
    uint32_t struct_size;
    int32_t kind, k, dist_type, weight, p_table, q_table, guard_endpoints, precise_sampler, block, tables_lds, rng, unit_params;
} client_module_desc;
#define DESC_PTR(d) ((const mcx_module_desc*)(d))
static void client_desc_init(client_module_desc* d) { memset(d, 0, sizeof *d); d->struct_size = (uint32_t)sizeof *d; d->guard_endpoints = 1; d->tables_lds = 1; }
#else
typedef mcx_module_desc client_module_desc;
#define DESC_PTR(d) (d)
static void client_desc_init(client_module_desc* d) { mcx_module_desc_init(d); }
#endif

#define LOAD(name) __typeof__(name)* p_##name = (__typeof__(name)*)dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); return 2; }

static const char* USER_SRC =
    "MCX_DEV float user_func_0(float x) { return x; }\n"
    "MCX_DEV float user_func_1(float x) { return x * x; }\n"
    "MCX_DEV float user_func_2(float x) { return x > 1.0f ? 1.0f : 0.0f; }\n";

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s libmcx.so\n", argv[0]); return 2; }
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(mcx_version) LOAD(mcx_last_error) LOAD(mcx_dispatch_config) LOAD(mcx_mcmc_dispatch_config)
    LOAD(mcx_shard_integrate) LOAD(mcx_device_count) LOAD(mcx_engine_create) LOAD(mcx_engine_destroy)
    LOAD(mcx_module_build) LOAD(mcx_module_release) LOAD(mcx_integrate) LOAD(mcx_engine_last_kernel_ms)
    LOAD(mcx_hip_runtime) LOAD(mcx_comm_create) LOAD(mcx_comm_destroy) LOAD(mcx_comm_size) LOAD(mcx_integrate_comm)
    LOAD(mcx_rccl_library) LOAD(mcx_lds_table_budget) LOAD(mcx_module_static_lds) LOAD(mcx_engine_last_launch_count)
    LOAD(mcx_abi_version) LOAD(mcx_result_rows) LOAD(mcx_table_create) LOAD(mcx_table_release) LOAD(mcx_module_desc_fit)
    LOAD(mcx_module_build_fitted) LOAD(mcx_module_block) LOAD(mcx_wgsl_translate) LOAD(mcx_wgsl_prelude) LOAD(mcx_free)
    LOAD(mcx_wgsl_plan) LOAD(mcx_core_create) LOAD(mcx_core_destroy) LOAD(mcx_core_integrate) LOAD(mcx_core_mcmc) LOAD(mcx_core_engine)

    printf("OK version %s\n", p_mcx_version());
    mcx_dispatch d;
    if (p_mcx_dispatch_config(1000000000ull, 0, &d) || d.total_threads != 65536u || d.loops_per_thread != 15259u) return 1;
    if (p_mcx_mcmc_dispatch_config(1u, 0, &d) || d.total_threads != 256u) return 1;
    if (p_mcx_dispatch_config(1000000ull, 0, &d)) return 1;
    mcx_shard s0, s1;
    if (p_mcx_shard_integrate(&d, MCX_DIST_NORMAL, 0, 2, &s0) || p_mcx_shard_integrate(&d, MCX_DIST_NORMAL, 1, 2, &s1)) return 1;
    if (s0.unit_begin != 0u || s0.unit_end != s1.unit_begin || s1.unit_end != 8u) return 1;
    printf("OK planning T=%u L=%u\n", d.total_threads, d.loops_per_thread);

    /* versioned structs: the caller's layout is accepted, an uninitialised or a longer one is refused */
    if (p_mcx_abi_version() != MCX_ABI_VERSION) { fprintf(stderr, "abi version %u\n", p_mcx_abi_version()); return 1; }
    client_module_desc desc;
    client_desc_init(&desc);
    desc.kind = MCX_KIND_INTEGRATE; desc.k = 3; desc.dist_type = MCX_DIST_NORMAL;
    if (p_mcx_result_rows(DESC_PTR(&desc)) != 3) { fprintf(stderr, "result_rows: %s\n", p_mcx_last_error()); return 1; }
    {
        struct { mcx_module_desc d; int32_t from_the_future[4]; } longer;
        memset(&longer, 0, sizeof longer);
        memcpy(&longer.d, &desc, sizeof desc);
        longer.d.struct_size = (uint32_t)sizeof longer;
        if (p_mcx_result_rows(&longer.d) != MCX_E_INVALID || !strstr(p_mcx_last_error(), "newer mcx.h")) return 1;
        longer.d.struct_size = 0u;
        if (p_mcx_result_rows(&longer.d) != MCX_E_INVALID || !strstr(p_mcx_last_error(), "struct_size is 0")) return 1;
    }
    {   /* the reference's own payload format: WGSL text in (what its transpiler writes for x**2 and a user string with a helper) */
        char *f0 = NULL, *f1 = NULL;
        if (p_mcx_wgsl_translate("fn user_func_9b8538e9(x: f32) -> f32 {\n    return pow(x, 2.0);\n}", 0, "user_func_0", 1, &f0) ||
            p_mcx_wgsl_translate("fn g(x: f32) -> f32 { return twice(sin(x)) ; }\nfn twice(y: f32) -> f32 { return 2.0 * y; }", 1, "user_func_1", 1, &f1)) {
            fprintf(stderr, "wgsl: %s\n", p_mcx_last_error()); return 1;
        }
        if (!strstr(f0, "McxPowI<2>::of(x)") || !strstr(f1, "mcx_uf1_twice(mcx_sin(x))") || !strstr(p_mcx_wgsl_prelude(), "struct McxPowI")) {
            fprintf(stderr, "wgsl text:\n%s\n%s\n", f0, f1); return 1;
        }
        char* none = NULL;
        if (p_mcx_wgsl_translate("fn f(x: vec2<f32>) -> f32 { return 1.0; }", 0, "user_func_0", 0, &none) != MCX_E_TRANSLATE ||
            !strstr(p_mcx_last_error(), "unsupported type 'vec2'")) return 1;
        printf("OK wgsl translated: %u + %u bytes of HIP\n", (unsigned)strlen(f0), (unsigned)strlen(f1));
        p_mcx_free(f0); p_mcx_free(f1);
    }
    printf("OK abi version %u, desc of %u bytes (library: %u)\n", p_mcx_abi_version(), (unsigned)sizeof desc, (unsigned)sizeof(mcx_module_desc));

    if (p_mcx_device_count() < 1) { printf("no GPU visible: stopping after the planning checks\n"); return 3; }
    mcx_engine* e = NULL;
    if (p_mcx_engine_create(0, &e)) { fprintf(stderr, "engine: %s\n", p_mcx_last_error()); return 1; }
    printf("OK engine on %s\n", p_mcx_hip_runtime());
    mcx_module* m = NULL;
    if (p_mcx_module_build(e, USER_SRC, DESC_PTR(&desc), &m)) { fprintf(stderr, "module: %s\n", p_mcx_last_error()); return 1; }
    mcx_integrate_params p;
    mcx_integrate_params_init(&p);
    p.n_samples = 100000000ull; p.seed = 42u; p.param1 = 0.0f; p.param2 = 1.0f;
    double sums[3]; uint64_t n_eff = 0;
    if (p_mcx_integrate(e, m, &p, sums, &n_eff)) { fprintf(stderr, "integrate: %s\n", p_mcx_last_error()); return 1; }
    double mean = sums[0] / (double)n_eff, second = sums[1] / (double)n_eff, tail = sums[2] / (double)n_eff;
    printf("OK integrate n_eff=%llu E[x]=%.6f E[x^2]=%.6f P(x>1)=%.6f kernel_ms=%.3f\n", (unsigned long long)n_eff, mean, second,
           tail, p_mcx_engine_last_kernel_ms(e));
    if (n_eff != 65536ull * 1526ull) return 1;
    if (fabs(mean) > 5e-4 || fabs(second - 1.0) > 1e-3 || fabs(tail - 0.158655) > 3e-4) return 1;
    if (p_mcx_engine_last_launch_count(e) != 1u) return 1;
    {   /* an uninitialised parameter block is refused, not read */
        mcx_integrate_params raw;
        memset(&raw, 0, sizeof raw);
        raw.n_samples = 1000u; raw.world = 1u;
        double junk[3]; uint64_t jn = 0;
        if (p_mcx_integrate(e, m, &raw, junk, &jn) != MCX_E_INVALID || !strstr(p_mcx_last_error(), "struct_size is 0")) return 1;
    }
    if (p_mcx_lds_table_budget(DESC_PTR(&desc)) + p_mcx_module_static_lds(m) > 160u * 1024u || p_mcx_lds_table_budget(DESC_PTR(&desc)) < 150u * 1024u) return 1;
    /* single-process RCCL path: a communicator over this process's engines (here: the one GPU of the box); the call
     * ends with ncclAllReduce(K doubles) on the engine's stream instead of a host-side sum */
    {
        mcx_comm* c = NULL;
        mcx_engine* engines[1] = {e};
        if (p_mcx_comm_create(engines, 1, &c)) { fprintf(stderr, "comm: %s\n", p_mcx_last_error()); return 1; }
        if (p_mcx_comm_size(c) != 1) return 1;
        mcx_module* mods[1] = {m};
        const mcx_integrate_params* pp[1] = {&p};
        double csums[3]; uint64_t cn = 0;
        if (p_mcx_integrate_comm(c, mods, pp, csums, &cn)) { fprintf(stderr, "integrate_comm: %s\n", p_mcx_last_error()); return 1; }
        if (cn != n_eff || csums[0] != sums[0] || csums[1] != sums[1] || csums[2] != sums[2]) {
            fprintf(stderr, "comm sums differ: %.17g %.17g\n", csums[0], sums[0]); return 1;
        }
        p_mcx_comm_destroy(c);
        printf("OK rccl communicator (1 rank) on %s: same sums\n", p_mcx_rccl_library());
    }
#ifndef MCX_CLIENT_OLD_LAYOUT
    {   /* importance sampling with a target table, planned by libmcx itself: E_p[x], E_p[x^2] for p = N(0,1) tabulated on a
         * 512-point grid, draws from N(0.5, 1.5). mcx_module_desc_fit sets the fields that are guarantees about the table. */
        enum { N = 512 };
        static float xs[N], ps[N];
        for (int i = 0; i < N; ++i) {
            xs[i] = (float)(-7.0 + 14.0 * i / (N - 1));
            ps[i] = (float)(exp(-0.5 * (double)xs[i] * (double)xs[i]) / 2.5066282746310002);
        }
        mcx_table* t = NULL;
        if (p_mcx_table_create(e, MCX_TABLE_PDF, xs, ps, N, &t)) { fprintf(stderr, "table: %s\n", p_mcx_last_error()); return 1; }
        mcx_module_desc wd;
        mcx_module_desc_init(&wd);
        wd.kind = MCX_KIND_INTEGRATE; wd.k = 2; wd.dist_type = MCX_DIST_NORMAL; wd.weight = 1; wd.p_table = 1; wd.q_sampler = 1;
        uint32_t pad_bytes = 0u;
        if (p_mcx_module_desc_fit(&wd, NULL, t, NULL, 0.5f, 1.5f, &pad_bytes)) { fprintf(stderr, "fit: %s\n", p_mcx_last_error()); return 1; }
        if (!wd.cell_tables || !wd.cell_noclamp || !wd.tables_lds || wd.unit_params || wd.block != 512 || pad_bytes == 0u) {
            fprintf(stderr, "fit: cells %d noclamp %d lds %d unit %d block %d pads %u\n", wd.cell_tables, wd.cell_noclamp, wd.tables_lds,
                    wd.unit_params, wd.block, pad_bytes);
            return 1;
        }
        mcx_module* wm = NULL;
        if (p_mcx_module_build_fitted(e, USER_SRC, &wd, NULL, t, NULL, pad_bytes, &wm)) { fprintf(stderr, "build_fitted: %s\n", p_mcx_last_error()); return 1; }
        mcx_integrate_params wp;
        mcx_integrate_params_init(&wp);
        wp.n_samples = 50000000ull; wp.seed = 7u; wp.param1 = 0.5f; wp.param2 = 1.5f; wp.target_pdf = t;
        double ws[2]; uint64_t wn = 0;
        if (p_mcx_integrate(e, wm, &wp, ws, &wn)) { fprintf(stderr, "weighted integrate: %s\n", p_mcx_last_error()); return 1; }
        printf("OK fitted importance sampling: cells %d, %u pad bytes, block %u, E_p[x]=%.6f E_p[x^2]=%.6f kernel_ms=%.3f\n", wd.cell_tables,
               pad_bytes, p_mcx_module_block(wm), ws[0] / (double)wn, ws[1] / (double)wn, p_mcx_engine_last_kernel_ms(e));
        if (fabs(ws[0] / (double)wn) > 1.5e-3 || fabs(ws[1] / (double)wn - 1.0) > 3e-3) return 1;
        /* the same call as the reference's Python half phrases it: two importance-sampling wrappers (python/wgpu_montecarlo/
         * __init__.py:968-980) around pdf_target_from_table and Distribution.normal(0.5, 1.5)'s closure. mcx_wgsl_plan recognises them:
         * K integrands + one weight, 1/q from the deviate -- the module above -- and the sums agree to rounding. */
        static const char* W0 =
            "\nfn _is_wrapper_0(x: f32) -> f32 {\n    let f_val = _is_f_orig_0(x);\n    let p = pdf_target_from_table(x);\n    let q = _is_pdf_q_0(x);\n"
            "    return f_val * p / q;\n}\n\n\nfn _is_pdf_q_0(x: f32) -> f32 {\n    const mean: f32 = 0.5;\n    const sigma: f32 = 1.5;\n"
            "    const sqrt_2pi: f32 = 2.5066282746310002;\n    var z = ((x - mean) / sigma);\n    return (exp((((-0.5) * z) * z)) / (sigma * sqrt_2pi));\n}\n"
            "fn _is_f_orig_0(x: f32) -> f32 {\n    return x;\n}\n";
        static const char* W1 =
            "\nfn _is_wrapper_1(x: f32) -> f32 {\n    let f_val = _is_f_orig_1(x);\n    let p = pdf_target_from_table(x);\n    let q = _is_pdf_q_1(x);\n"
            "    return f_val * p / q;\n}\n\n\nfn _is_pdf_q_1(x: f32) -> f32 {\n    const mean: f32 = 0.5;\n    const sigma: f32 = 1.5;\n"
            "    const sqrt_2pi: f32 = 2.5066282746310002;\n    var z = ((x - mean) / sigma);\n    return (exp((((-0.5) * z) * z)) / (sigma * sqrt_2pi));\n}\n"
            "fn _is_f_orig_1(x: f32) -> f32 {\n    return pow(x, 2.0);\n}\n";
        const char* payload[2] = {W0, W1};
        mcx_wgsl_program prog;
        mcx_wgsl_program_init(&prog);
        prog.kind = MCX_KIND_INTEGRATE; prog.k = 2; prog.functions = payload; prog.dist_type = MCX_DIST_NORMAL;
        prog.param1 = 0.5f; prog.param2 = 1.5f; prog.have_target_table = 1;
        mcx_module_desc pd;
        char* planned = NULL;
        if (p_mcx_wgsl_plan(&prog, &pd, &planned)) { fprintf(stderr, "plan: %s\n", p_mcx_last_error()); return 1; }
        if (!pd.weight || !pd.p_table || pd.q_table || !pd.q_sampler || pd.user_tables || strstr(planned, "mcx_pdf_q")) {
            fprintf(stderr, "plan: weight %d p_table %d q_sampler %d\n%s\n", pd.weight, pd.p_table, pd.q_sampler, planned); return 1;
        }
        uint32_t ppad = 0u;
        mcx_module* pm = NULL;
        if (p_mcx_module_desc_fit(&pd, NULL, t, NULL, 0.5f, 1.5f, &ppad) || p_mcx_module_build_fitted(e, planned, &pd, NULL, t, NULL, ppad, &pm)) {
            fprintf(stderr, "planned module: %s\n", p_mcx_last_error()); return 1;
        }
        double ps2[2]; uint64_t pn = 0;
        if (p_mcx_integrate(e, pm, &wp, ps2, &pn)) { fprintf(stderr, "planned integrate: %s\n", p_mcx_last_error()); return 1; }
        printf("OK planned from the reference's wrapper text: E_p[x]=%.6f E_p[x^2]=%.6f (hand-built module: %.6f %.6f)\n", ps2[0] / (double)pn,
               ps2[1] / (double)pn, ws[0] / (double)wn, ws[1] / (double)wn);
        if (pn != wn || fabs(ps2[0] - ws[0]) > 1e-6 * (double)wn || fabs(ps2[1] - ws[1]) > 1e-6 * (double)wn) return 1;
        {   /* and as src/lib.rs would forward it: the reference's native object, call for call (strings + float32 tables in, K float32
             * means out). The second call finds its tables by content and its module by payload. */
            mcx_core* core = NULL;
            if (p_mcx_core_create(0, 1, &core)) { fprintf(stderr, "core: %s\n", p_mcx_last_error()); return 1; }
            mcx_core_tables ct;
            mcx_core_tables_init(&ct);
            ct.target_x = xs; ct.target_v = ps; ct.n_target = N;
            float v1[2], v2[2];
            if (p_mcx_core_integrate(core, payload, 2, MCX_DIST_NORMAL, 0.5f, 1.5f, 50000000ull, 7u, &ct, 0, v1) ||
                p_mcx_core_integrate(core, payload, 2, MCX_DIST_NORMAL, 0.5f, 1.5f, 50000000ull, 7u, &ct, 0, v2)) {
                fprintf(stderr, "core integrate: %s\n", p_mcx_last_error()); return 1;
            }
            if (v1[0] != v2[0] || v1[1] != v2[1] || v1[0] != (float)(ps2[0] / (double)pn) || v1[1] != (float)(ps2[1] / (double)pn)) {
                fprintf(stderr, "core values %.9g %.9g vs %.9g %.9g\n", v1[0], v1[1], ps2[0] / (double)pn, ps2[1] / (double)pn); return 1;
            }
            /* integrate_mcmc with no tables at all: both log-densities analytic (src/shader_gen.rs:543-571), N(0.5, 1) target */
            const char* fm[2] = {"fn user_func_0a(x: f32) -> f32 {\n    return x;\n}", "fn user_func_0b(x: f32) -> f32 {\n    return pow(x, 2.0);\n}"};
            float vm[2];
            if (p_mcx_core_mcmc(core, fm, 2, MCX_DIST_NORMAL, 0.0f, 2.0f, MCX_DIST_NORMAL, 0.5f, 1.0f, 2000u, 4096u, 200u, 42u, NULL, 0, vm)) {
                fprintf(stderr, "core mcmc: %s\n", p_mcx_last_error()); return 1;
            }
            if (fabs(vm[0] - 0.5f) > 0.02f || fabs(vm[1] - 1.25f) > 0.05f) { fprintf(stderr, "core mcmc values %g %g\n", vm[0], vm[1]); return 1; }
            if (p_mcx_core_integrate(core, payload, 0, MCX_DIST_NORMAL, 0.f, 1.f, 1000ull, 1u, NULL, 0, v1) != MCX_E_INVALID ||
                !strstr(p_mcx_last_error(), "At least one function")) return 1;
            if (p_mcx_core_mcmc(core, fm, 2, MCX_DIST_NORMAL, 0.0f, 2.0f, MCX_DIST_NORMAL, 0.5f, 1.0f, 0u, 4096u, 200u, 42u, NULL, 0, vm) != MCX_E_INVALID ||
                !strstr(p_mcx_last_error(), "n_steps must be positive")) return 1;
            printf("OK mcx_core: integrate_is_tables %.6f %.6f (twice, identical), integrate_mcmc %.4f %.4f\n", v2[0], v2[1], vm[0], vm[1]);
            p_mcx_core_destroy(core);
        }
        prog.math = 0;                                  /* literal: two functions that read the table themselves */
        char* literal = NULL;
        if (p_mcx_wgsl_plan(&prog, &pd, &literal) || pd.weight || pd.user_tables != 1 || !strstr(literal, "mcx_user_pdf_target(x)")) return 1;
        p_mcx_free(planned); p_mcx_free(literal);
        p_mcx_module_release(pm);
        p_mcx_module_release(wm);
        p_mcx_table_release(t);
    }
#endif
    desc.k = 0;                                                     /* src/lib.rs:61-65 */
    mcx_module* bad = NULL;
    if (p_mcx_module_build(e, USER_SRC, DESC_PTR(&desc), &bad) != MCX_E_INVALID || !strstr(p_mcx_last_error(), "At least one function")) return 1;
    printf("OK errors: %s\n", p_mcx_last_error());
    p_mcx_module_release(m);
    p_mcx_engine_destroy(e);
    return 0;
}
