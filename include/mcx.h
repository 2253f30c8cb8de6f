/* mcx.h -- C ABI of libmcx.so, the MI355X (gfx950) runtime behind
 * MonteCarloIntegrator.{integrate, integrate_importance_sampling, integrate_mcmc}.
 *
 * It replaces the reference's PyO3 module `wgpu_montecarlo._core` (src/lib.rs) together with the
 * Rust layers under it (src/engine.rs device runtime, src/shader_gen.rs kernel assembler,
 * src/distribution.rs device library). Every entry point cites the reference interface it
 * replaces (file:line into NightingaleCen/wgpu-monte-carlo). Plain pointers and sizes only.
 *
 * Conventions: every function returns 0 on success or a negative MCX_E_* code; the text of the
 * last error of the calling thread is available from mcx_last_error(). Host pointers passed in
 * are only read during the call. Handles are owned by the caller and released with the matching
 * *_destroy / *_release call. One engine drives one GPU; calls on one engine are serialised by
 * an internal mutex.
 */
#ifndef MCX_H
#define MCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCX_OK              0
#define MCX_E_INVALID      -1   /* bad argument (Python side raises ValueError)                 */
#define MCX_E_RUNTIME      -2   /* HIP runtime failure (RuntimeError)                           */
#define MCX_E_COMPILE      -3   /* hiprtc failure, log in mcx_last_error() (RuntimeError)       */
#define MCX_E_NODEVICE     -4   /* no usable GPU (RuntimeError "Failed to initialize GPU: ..")  */
#define MCX_E_TRANSLATE    -5   /* a WGSL function string outside the translator's subset (TranspilerError) */

/* distribution type codes: DistributionParamsBuffer.dist_type, src/engine.rs:32-37, src/lib.rs:436-502 */
#define MCX_DIST_UNIFORM     0
#define MCX_DIST_NORMAL      1
#define MCX_DIST_EXPONENTIAL 2
#define MCX_DIST_CUSTOM      3

/* table kinds */
#define MCX_TABLE_CDF     0   /* keys = cdf, values = x  : sample_from_cdf_table, distribution.rs:128-158 */
#define MCX_TABLE_PDF     1   /* keys = x, values = pdf  : pdf_*_from_table, distribution.rs:181-275      */
#define MCX_TABLE_LOGPDF  2   /* keys = x, values = log pdf : log_pdf_*_from_table, distribution.rs:375-469 */

typedef struct mcx_engine mcx_engine;
typedef struct mcx_module mcx_module;
typedef struct mcx_table  mcx_table;

const char* mcx_version(void);
const char* mcx_last_error(void);

/* ------------------------------------------------------------------------------------------
 * ABI versioning. What this header replaces -- _core.MonteCarloIntegrator.integrate / integrate_is_tables /
 * integrate_mcmc, src/lib.rs:47-59, 158-180, 296-324 -- is a stable positional signature; the structs below carry the
 * same arguments by name and have grown from round to round. They only ever grow AT THE END, and each starts with
 * `struct_size` = sizeof(the struct) as the CALLER compiled it (set by the mcx_*_init helpers below):
 *   - a caller built against an older, shorter layout is accepted; the fields it does not know read as 0 (every
 *     field added after the first release has 0 = "off / reference behaviour" as its default);
 *   - a caller built against a newer, longer layout than this library is refused with MCX_E_INVALID;
 *   - struct_size = 0 (a struct that was never initialised) is refused with MCX_E_INVALID.
 * MCX_ABI_VERSION counts layouts: 1 = round 1 (module desc up to `unit_params`), 2 = round 2 (up to `cell_addr16`,
 * unversioned), 3 = struct_size first, 4 = the versioned structs of 3 unchanged, plus mcx_wgsl_program / mcx_core_tables and the
 * entry points that take them (mcx_wgsl_*, mcx_module_desc_fit*, mcx_core_*): a host checks >= 4 before it looks them up.
 * ------------------------------------------------------------------------------------------ */
#define MCX_ABI_VERSION 4
uint32_t mcx_abi_version(void);        /* MCX_ABI_VERSION the library was built with */

/* ------------------------------------------------------------------------------------------
 * Planning -- pure host arithmetic, usable without a GPU.
 * ------------------------------------------------------------------------------------------ */

/* DispatchConfig, src/engine.rs:39-46 */
typedef struct mcx_dispatch {
    uint32_t workgroup_size;     /* always 256 */
    uint32_t workgroup_count;
    uint32_t loops_per_thread;   /* L */
    uint32_t total_threads;      /* T */
} mcx_dispatch;

/* ComputeEngine::calculate_dispatch_config, src/engine.rs:157-181.
 * target_threads <= 0 selects the reference default 65536. L is the truncating u32 cast of
 * ceil(n_samples / T), as in the reference. */
int mcx_dispatch_config(uint64_t n_samples, int64_t target_threads, mcx_dispatch* out);

/* ComputeEngine::calculate_mcmc_dispatch_config + setup_mcmc, src/engine.rs:821-832, 860-866:
 * chains = target_threads if > 0 else n_chains, padded up to a multiple of 256; the padded
 * count is what runs and what is averaged. */
int mcx_mcmc_dispatch_config(uint32_t n_chains, int64_t target_threads, mcx_dispatch* out);

/* One rank's share of the logical sample grid {(idx, i): idx < T, i < L} (new; the reference is
 * single-device). Units are loop iterations i, or Box-Muller pairs (2j, 2j+1) for the normal
 * distribution, so that a shard boundary never splits a pair. The union over ranks is exactly the
 * single-GPU grid. */
typedef struct mcx_shard {
    uint32_t idx_begin, idx_count;
    uint32_t unit_begin, unit_end;
} mcx_shard;
int mcx_shard_integrate(const mcx_dispatch* d, int dist_type, uint32_t rank, uint32_t world, mcx_shard* out);
/* Same with an explicit unit size: iterations per unit = 1 (uniform / exponential / custom), 2 (normal: a
 * Box-Muller pair), 4 (Philox stream: one call). */
int mcx_shard_units(const mcx_dispatch* d, uint32_t iterations_per_unit, uint32_t rank, uint32_t world, mcx_shard* out);

/* Recommended mcx_module_desc.block for an MCMC module whose launches carry `chains` chains on one GPU (a rank's share
 * of a chain-sharded run): the largest of 1024 / 512 / 256 threads that still gives every CU >= 4 workgroups, else 256.
 * One chain per thread: with 1024-thread workgroups a 131 072-chain shard would occupy half the CUs. */
uint32_t mcx_mcmc_block_hint(uint32_t chains);

/* One rank's contiguous share of the padded chain range [0, T), in multiples of 256 chains. */
int mcx_shard_chains(uint32_t total_chains, uint32_t rank, uint32_t world,
                     uint32_t* chain_begin, uint32_t* chain_count);

/* ------------------------------------------------------------------------------------------
 * Engine -- replaces ComputeEngine::new, src/engine.rs:91-131 / _core.MonteCarloIntegrator(), src/lib.rs:24-31
 * ------------------------------------------------------------------------------------------ */
int  mcx_device_count(void);                         /* 0 when no GPU is visible */
/* Which HIP runtime libmcx bound to at run time. libmcx links none: it shares the libamdhip64 instance that
 * is already mapped in the process (PyTorch-ROCm bundles its own), else MCX_HIP_RUNTIME, else the system one --
 * stream handles passed to *_device calls must come from that same runtime. */
const char* mcx_hip_runtime(void);
int  mcx_engine_create(int device, mcx_engine** out);
void mcx_engine_destroy(mcx_engine* e);
int  mcx_engine_device(const mcx_engine* e);
/* Duration in ms of the main (sampling) kernel of the last call on this engine, measured with
 * HIP events on the stream it was launched on; < 0 if nothing was launched. */
float mcx_engine_last_kernel_ms(mcx_engine* e);
/* Launch geometry of the last call: physical workgroups, threads per workgroup, dynamic LDS bytes. */
int  mcx_engine_last_launch(mcx_engine* e, uint32_t* n_blocks, uint32_t* block, uint32_t* lds_bytes);
/* Everything the three getters above and below report, in one call (what a binding reads after every blocking call). */
typedef struct mcx_call_info {
    uint32_t struct_size;      /* sizeof(mcx_call_info) in the caller's build */
    uint32_t n_blocks, block, lds_bytes, launches;
    float    kernel_ms;        /* < 0 if nothing was launched */
    uint32_t segments;         /* MCMC: time segments the call ran as (launches = 2 x segments, n_blocks = the workgroups of
                                * one segment, both halves), 0 for an unsegmented call */
} mcx_call_info;
/* with_timing = 0 leaves kernel_ms at -1 and does not wait; 1 waits for the call's timing event like mcx_engine_last_kernel_ms */
int  mcx_engine_last_call(mcx_engine* e, mcx_call_info* out, int with_timing);
/* Main-kernel launches the last call was split into (1 unless the call exceeded the per-launch work bound of
 * ~1e11 samples / chain-steps, MCX_MAX_LAUNCH_UNITS; the reference always issues one dispatch, src/engine.rs:468-525). */
uint32_t mcx_engine_last_launch_count(const mcx_engine* e);
/* Tuning knob: physical threads a launch aims for; 0 = the default, 4096 workgroups of the module's size (16 per CU). */
int  mcx_engine_set_target_threads(mcx_engine* e, uint32_t physical_threads);
/* Time segments of an MCMC call of the independence sampler with a normal proposal (either stream; no
 * precise_sampler / second_moments / walk): the call runs as two halves of the chains on two streams, each cut into
 * `segments` launches over consecutive step ranges. A launch that fills the chip a small whole number of times --
 * 1 048 576 chains = exactly twice -- leaves CUs idle while its last workgroups finish; the other half's next segment
 * covers that (C4: 8.5 -> 8.0 ms with 8 segments). Same chains, same draws, same accept decisions: the streams are
 * functions of (seed, chain, step), the chain state {x, w} travels through a device buffer, every launch adds its own
 * partial sums (the reference issues one dispatch, src/engine.rs:468-525).
 *   MCX_SEGMENTS_AUTO (the default; MCX_MCMC_SEGMENTS overrides it): launches of >= 131 072 chains (two waves per SIMD
 *   and more) run in 8 segments from 1.4e9 chain-steps (~1 ms of work), in 4 from 7e8, else in one launch -- each segment
 *   costs ~10 us of launches. At 11 000 steps per chain: 1 048 576 chains 8.55 -> 8.12 ms, 524 288: 4.60 -> 4.13, 131 072:
 *   1.43 -> 1.36; 65 536 chains lose (0.95 -> 1.22 ms), short calls lose (1 048 576 chains x 400 steps: 0.34 -> 0.38 ms).
 *   0 or 1: always one launch; 2..64: that many whenever the call qualifies.
 * The side stream, its events and the state buffer are kept PER CALLER STREAM (like the per-workgroup partial sums), so
 * segmented calls in flight on different streams of one engine do not share them. */
#define MCX_SEGMENTS_AUTO 0xFFFFFFFFu
int  mcx_engine_set_mcmc_segments(mcx_engine* e, uint32_t segments);
/* The default: workgroups a launch of `samples` samples aims for when each workgroup stages `lds_bytes` of tables --
 * 4096 (16 per CU) once every workgroup samples at least 6 samples per staged byte, never fewer than 2^20 / block, and in
 * whole rounds of what the chip holds at once (256 CUs x min(2048 / block, 160 KiB / staged bytes) workgroups). */
uint32_t mcx_default_launch_blocks(uint64_t samples, uint32_t lds_bytes, uint32_t block);

/* ------------------------------------------------------------------------------------------
 * Modules -- replaces generate_*_shader (src/shader_gen.rs:45, 134, 312) +
 * create_*_pipeline (src/engine.rs:325, 621, 1022): the emitted user functions are fused into
 * the hand-written kernel skeleton and compiled for gfx950 with hiprtc. Unlike the reference,
 * which recompiles on every call, code objects are cached by source hash (memory + disk).
 * ------------------------------------------------------------------------------------------ */
#define MCX_KIND_INTEGRATE 0   /* K1 / K2 */
#define MCX_KIND_MCMC      1   /* K3 */

typedef struct mcx_module_desc {
    uint32_t struct_size;      /* sizeof(mcx_module_desc) in the caller's build: mcx_module_desc_init() */
    int32_t kind;              /* MCX_KIND_* */
    int32_t k;                 /* number of fused user functions user_func_0 .. user_func_{k-1} */
    int32_t dist_type;         /* sampling (K1/K2) or proposal (K3) distribution */
    int32_t weight;            /* 1: importance weight p/q applied (K2) */
    int32_t p_table;           /* 1: target pdf from table, 0: analytic mcx_pdf_p in user_src */
    int32_t q_table;           /* 1: proposal pdf from table, 0: analytic mcx_pdf_q in user_src */
    int32_t guard_endpoints;   /* 1 (default): u in (0,1] / [0,1); 0: strict reference u = float(h)*2^-32 */
    int32_t precise_sampler;   /* 1: ocml log/sin/cos in the samplers; 0: v_log/v_sin/v_cos */
    int32_t block;             /* threads per workgroup: 0 = auto (256, or 1024 when tables are staged) */
    int32_t tables_lds;        /* 1 (default): tables staged in LDS; 0: read from HBM/L2 */
    int32_t rng;               /* 0 (default): the reference's PCG counter hash (parity stream);
                                * 1: Philox4x32-10, key (seed, 'MCX1'); counter (idx, i/4, 0, 0) for K1/K2 (four iterations per
                                * call), (idx, it / 2, 1, 0) for K3 (one call per two MH steps) -- opt-in for runs that draw more
                                * than ~2^32 uniforms */
    int32_t unit_params;       /* 1: the caller guarantees param1/param2 are the identity -- normal(0,1), uniform(0,1),
                                * exponential(1) -- so the affine map of the sampler is not emitted (bit-identical results) */
    int32_t second_moments;    /* 1: rows k..2k-1 of the result hold the sums of (f_i * w)^2 (standard errors). MCMC
                                * (k <= 16): additionally row 2k = accepted steps and rows 2k+1..3k hold the sums over
                                * chains of (per-chain mean of f_i)^2 -- batch means for standard errors and effective
                                * sample sizes. See mcx_result_rows. */
    int32_t walk;              /* MCMC only. MCX_WALK_INDEPENDENT (0, the reference: x' ~ q, shader_gen.rs:466-539),
                                * MCX_WALK_RANDOM (1): x' = x + d, d ~ q, log alpha = log p(x') - log p(x) + log q(-d) - log q(d),
                                * MCX_WALK_RANDOM_SYMMETRIC (2): same with the q terms dropped (caller guarantees q(d) = q(-d)).
                                * MCX_WALK_ADAPTIVE (3): symmetric random walk x' = x + s d whose per-chain scale s adapts during
                                * burn-in only (log s += t^-1/2 (accepted - target_accept) after step t, s = 1 at the start) and is
                                * frozen for the sampling steps; one more result row holds the sum of the final scales.
                                * Random-walk proposals outside the target table (log p <= -100) are always rejected.
                                * The reference leaves this open ("For now, we use independent proposal", shader_gen.rs:514). */
    int32_t cell_tables;       /* 1: the caller guarantees every PDF / log-PDF table bound to this module has the
                                * slope-intercept cell form (mcx_table_has_cells); the lookup is then compiled as one
                                * 8-byte read + one FMA with no search path (checked at launch; not with precise_sampler) */
    int32_t q_sampler;         /* 1 (normal sampling / proposal distribution only): the proposal density is the sampler's own
                                * N(param1, param2) and is formed from the standard-normal deviate z the sampler already holds.
                                * Importance sampling: 1/q(x) = param2 * sqrt(2 pi) * exp(z^2 / 2); mcx_pdf_q is not called. Same
                                * value as the reference's f * p / q with q from the Distribution.normal closure
                                * (python/wgpu_montecarlo/__init__.py:893-899), without the second exp and the reciprocal.
                                * MCMC: log q = -z^2 / 2 (+ a constant that cancels); no proposal_logpdf table is bound. The
                                * reference interpolates a 2048-point table of that function (src/shader_gen.rs:521-526):
                                * <= 6e-6 below it inside +-7 std, -100 outside (probability 2.6e-12 per draw). */
    int32_t moment_family;     /* 1 (integrate / importance sampling, no second_moments): the caller guarantees
                                * user_func_i(x) = x^(i+1) for every i -- the fused-moments workload. Two samples a, b are then
                                * accumulated together through Newton's identity for (weighted) power sums,
                                * s_k = (a + b) s_{k-1} - a b s_{k-2}: 3 operations per power per pair instead of 4. */
    int32_t user_tables;       /* integrate modules: bit 0 / bit 1 = the target / proposal PDF table of the call is staged and
                                * visible to the user functions through the device functions mcx_user_pdf_target and mcx_user_pdf_proposal -- what the
                                * reference's own importance-sampling wrappers call as pdf_target_from_table(x) /
                                * pdf_proposal_from_table(x) (src/distribution.rs:181-223, python/wgpu_montecarlo/
                                * __init__.py:968-974). For callers that hand over the reference's WGSL text unchanged
                                * (wgpu_montecarlo/_core.py); the package's own API uses the weight mode instead. */
    int32_t logpdf_analytic;   /* MCMC modules: bit 0 / bit 1 = the target / proposal log-density is the device function
                                * `mcx_logpdf_p` / `mcx_logpdf_q` (float -> float) of user_src instead of a table -- the reference's fallback
                                * when `_core.integrate_mcmc` gets no table (src/shader_gen.rs:327-339, 496-509, 543-571).
                                * The matching mcx_mcmc_params table pointer is then ignored. Not with q_sampler. */
    int32_t cdf_direct;        /* 1 (custom sampling / proposal distribution, not with precise_sampler): the caller guarantees the
                                * CDF table of every call has the bucket-direct form (mcx_table_has_direct). A draw whose
                                * bucket (the top bits of its hash word) holds no cdf node is then ONE 8-byte read and one FMA --
                                * the same cell the reference's lower-bound search selects (src/distribution.rs:128-158), its
                                * line evaluated on the unrounded low hash bits (<= 2.4e-7 * slope from the blend on the
                                * rounded u); other draws run the search inside the bucket's window. The integrate kernel
                                * additionally defers those draws to a per-wave LDS queue and resolves them 64 at a time. */
    int32_t cell_noclamp;      /* 1 (with cell_tables and tables_lds; not with user_tables or walk): the cell lookup drops its index
                                * clamp -- one half-rate v_med3_f32 per lookup. Every x a call looks up is then one of the
                                * sampler's draws (the weight p(x) / q(x); the independence sampler's proposals and initial
                                * states), whose range the launch knows from the call's parameters; the kernel stages that many
                                * more {outside, 0} sentinel cells either side of each table, so a lookup outside the table
                                * still reads what the reference returns there (0 / -100: src/distribution.rs:190-195,
                                * 384-389). A call whose range needs more than 4096 extra cells on a side is refused: ask
                                * mcx_cell_pads before setting this. */
    int32_t cell_addr16;       /* 1 (with cell_tables and tables_lds): the caller guarantees that static + dynamic LDS of every launch stay
                                * within 64 KiB, so a cell's LDS byte address fits 16 bits and is read straight out of the mantissa
                                * of the index FMA (offset by 2^16: ulp 2^-7 byte) -- a shift and an AND instead of a half-rate
                                * v_cvt_u32_f32. Checked at launch. */
} mcx_module_desc;
/* Zero-fill, set struct_size, and the two defaults that are not 0: guard_endpoints = 1, tables_lds = 1. */
static inline void mcx_module_desc_init(mcx_module_desc* d) {
    uint32_t i;
    for (i = 0; i < sizeof(*d); ++i) ((unsigned char*)d)[i] = 0;
    d->struct_size = (uint32_t)sizeof(*d);
    d->guard_endpoints = 1;
    d->tables_lds = 1;
}

#define MCX_RNG_PCG_REF 0
#define MCX_RNG_PHILOX  1
#define MCX_WALK_INDEPENDENT      0
#define MCX_WALK_RANDOM           1
#define MCX_WALK_RANDOM_SYMMETRIC 2
#define MCX_WALK_ADAPTIVE         3

/* Number of doubles a call with this module writes to sums_out / d_sums (<= 65), or a negative error:
 * integrate: k (2k with second_moments); MCMC: k + 1 (3k + 1 with second_moments), one more with MCX_WALK_ADAPTIVE. */
int  mcx_result_rows(const mcx_module_desc* desc);

/* user_src: HIP C++ text defining `__device__ float user_func_i(float x)` for i < k (and
 * mcx_pdf_p / mcx_pdf_q when weight && !p_table / !q_table). */
int  mcx_module_build(mcx_engine* e, const char* user_src, const mcx_module_desc* desc, mcx_module** out);
/* hiprtc compile into the on-disk cache only: needs no GPU. cache_hit may be NULL. */
int  mcx_module_precompile(const char* user_src, const mcx_module_desc* desc, int* cache_hit);
/* Cache key of the code object (user_src, desc) compiles to: 32 hex digits + NUL into key_out[33]. The code object is
 * <mcx_cache_dir()>/<key>.hsaco; profiles/ identifies disassembled modules by it. Needs no GPU. */
int  mcx_module_key(const char* user_src, const mcx_module_desc* desc, char* key_out);
/* Full translation unit that would be compiled (for inspection / offline hipcc). Caller frees with mcx_free. */
int  mcx_module_source(const char* user_src, const mcx_module_desc* desc, char** out_text);
void mcx_free(void* p);
void mcx_module_release(mcx_module* m);
/* Threads per workgroup the module was compiled for (desc.block, or the default libmcx chose for block = 0). */
uint32_t mcx_module_block(const mcx_module* m);
/* Static LDS bytes of the module's main kernel (its cross-wave reduction scratch), read from the code object. */
uint32_t mcx_module_static_lds(const mcx_module* m);
/* LDS bytes a module built from `desc` leaves for staged tables: 160 KiB per CU minus (an upper bound of) the static
 * scratch its kernel declares. A call whose tables (sum of mcx_table_lds_bytes) exceed it must build the module with
 * tables_lds = 0; launches check the exact figure. 0 for an invalid desc. */
uint32_t mcx_lds_table_budget(const mcx_module_desc* desc);
/* ------------------------------------------------------------------------------------------
 * WGSL function strings -- what the reference's native half is handed (src/lib.rs:47-59: `functions: Vec<String>`, the output of its
 * Python transpiler, user-written strings and its importance-sampling wrappers, python/wgpu_montecarlo/__init__.py:740-742, 893-905,
 * 968-980) and splices into its shader (src/shader_gen.rs:45-128). libmcx's kernels are HIP: mcx_wgsl_translate turns one such string
 * (entry function first, helpers after it; the scalar subset: f32 / i32 / u32 / bool, let / var / const, if / else, for / while / loop,
 * the WGSL builtins, calls to pdf_target_from_table / pdf_proposal_from_table -> desc.user_tables) into the `MCX_DEV` functions
 * mcx_module_build takes as user_src: the entry is named `entry_name` (user_func_<i>, mcx_pdf_p, ...), helpers are prefixed
 * mcx_uf<slot>_. math: 0 = ocml builtins, 1 = the hardware forms within WGSL's own accuracy bounds (device/mcx_device.hpp: mcx_sin ..),
 * 2 = additionally the bare v_sin / v_cos. user_src = mcx_wgsl_prelude() + the translations. *out_text: mcx_free. Needs no GPU.
 * ------------------------------------------------------------------------------------------ */
int  mcx_wgsl_translate(const char* wgsl, int32_t slot, const char* entry_name, int32_t math, char** out_text);
const char* mcx_wgsl_prelude(void);

/* A whole payload of the reference's native module -- the K strings of one call -- planned: what the strings ARE is visible in their
 * text, and libmcx builds the better module for it (math != 0; math = 0 compiles every string literally as user_func_i):
 *   - every string one of the importance-sampling wrappers the reference's Python half generates (`_is_wrapper_i`: f_val * p / q around
 *     _is_f_orig_i, _is_pdf_p_i | pdf_target_from_table, _is_pdf_q_i | pdf_proposal_from_table; python/wgpu_montecarlo/__init__.py:
 *     893-905, 968-980) around the same p and q: K integrands + ONE weight per sample (desc.weight, p_table / q_table, mcx_pdf_p /
 *     mcx_pdf_q in the text); a q that is Distribution.normal's closure for exactly (param1, param2): desc.q_sampler;
 *   - the transpiler's text for x, x**2, .., x**K (8 <= K <= 32): desc.moment_family;
 *   - MCMC: a log-density without its table becomes the analytic one of its distribution type (desc.logpdf_analytic,
 *     src/shader_gen.rs:543-571); a normal proposal takes log q from its own deviate (desc.q_sampler; its table is then not bound).
 * Fills desc_out (initialised here; set desc.block / call mcx_module_desc_fit with the call's tables afterwards) and *user_src_out
 * (mcx_free). Bind the target / proposal PDF tables as target_pdf / proposal_pdf in either case (desc.weight or desc.user_tables). */
typedef struct mcx_wgsl_program {
    uint32_t struct_size;          /* mcx_wgsl_program_init() */
    int32_t  kind;                 /* MCX_KIND_INTEGRATE (integrate / integrate_is_tables) or MCX_KIND_MCMC */
    int32_t  k;
    const char* const* functions;  /* k WGSL strings */
    int32_t  dist_type;            /* sampling / proposal distribution and its parameters */
    float    param1, param2;
    int32_t  math;                 /* 0 precise (literal), 1 default, 2 fast */
    int32_t  have_target_table;    /* the call brings a target PDF (integrate) / log-PDF (MCMC) table */
    int32_t  have_proposal_table;  /* likewise for the proposal */
    int32_t  target_dist_type;     /* MCMC without a target table: the distribution whose analytic log-density is evaluated */
    float    target_param1, target_param2;
} mcx_wgsl_program;
static inline void mcx_wgsl_program_init(mcx_wgsl_program* g) {
    uint32_t i;
    for (i = 0; i < sizeof(*g); ++i) ((unsigned char*)g)[i] = 0;
    g->struct_size = (uint32_t)sizeof(*g);
    g->math = 1;
}
int  mcx_wgsl_plan(const mcx_wgsl_program* prog, mcx_module_desc* desc_out, char** user_src_out);

/* ------------------------------------------------------------------------------------------
 * The reference's native module, call for call. `_core.MonteCarloIntegrator` (src/lib.rs:17-431) owns a ComputeEngine and has three
 * methods that take WGSL strings, a distribution, float32 tables and sizes and return K float32 means. mcx_core is that object: an
 * engine plus what the reference rebuilds on every call and libmcx keeps -- resident tables found again by content (the reference
 * re-uploads them per call, src/engine.rs:235-295), planned + compiled modules found again by payload (it recompiles its shader per
 * call, :325-331). A Rust / C host of src/lib.rs forwards its arguments unchanged. math: 0 literal / 1 default / 2 fast (mcx_wgsl_plan).
 * Errors as the reference raises them: MCX_E_INVALID = ValueError ("At least one function is required", "n_steps must be positive"),
 * MCX_E_RUNTIME / _COMPILE / _NODEVICE = RuntimeError, MCX_E_TRANSLATE = TranspilerError.
 * ------------------------------------------------------------------------------------------ */
typedef struct mcx_core mcx_core;
int  mcx_core_create(int device, int32_t math, mcx_core** out);               /* MonteCarloIntegrator::new, src/lib.rs:24-31 */
void mcx_core_destroy(mcx_core* c);
mcx_engine* mcx_core_engine(mcx_core* c);                                     /* for mcx_engine_last_call etc.; owned by the core */
/* The optional table arguments of the three methods (NULL / 0 = not given, as Python's None). */
typedef struct mcx_core_tables {
    uint32_t struct_size;                              /* mcx_core_tables_init() */
    uint32_t n_cdf, n_target, n_proposal;
    const float* x_table;   const float* cdf_table;    /* custom sampling / proposal distribution (src/lib.rs:71-77) */
    const float* target_x;  const float* target_v;     /* integrate_is_tables: target PDF table; integrate_mcmc: target log-PDF table */
    const float* proposal_x; const float* proposal_v;  /* likewise for the proposal */
} mcx_core_tables;
static inline void mcx_core_tables_init(mcx_core_tables* t) {
    uint32_t i;
    for (i = 0; i < sizeof(*t); ++i) ((unsigned char*)t)[i] = 0;
    t->struct_size = (uint32_t)sizeof(*t);
}
/* integrate (src/lib.rs:47-141) and integrate_is_tables (:158-275; the PDF tables in `tables`): k WGSL strings -> values_out[k]. */
int  mcx_core_integrate(mcx_core* c, const char* const* functions, int32_t k, int32_t dist_type, float param1, float param2,
                        uint64_t n_samples, uint32_t seed, const mcx_core_tables* tables, int64_t target_threads, float* values_out);
/* integrate_mcmc (src/lib.rs:296-431). A log-PDF table that is not given becomes the analytic log-density of that distribution type. */
int  mcx_core_mcmc(mcx_core* c, const char* const* functions, int32_t k, int32_t proposal_dist_type, float param1, float param2,
                   int32_t target_dist_type, float target_param1, float target_param2, uint32_t n_steps, uint32_t n_chains,
                   uint32_t n_burnin, uint32_t seed, const mcx_core_tables* tables, int64_t target_threads, float* values_out);

/* Where code objects are cached (default: <dir of libmcx.so>/jit_cache, override MCX_CACHE_DIR). The in-memory copy
 * is an LRU of MCX_CODE_CACHE_ENTRIES (default 256) code objects. */
const char* mcx_cache_dir(void);

/* ------------------------------------------------------------------------------------------
 * Tables -- replaces the storage buffers of setup_integration / create_interleaved_pdf_buffer /
 * setup_mcmc (src/engine.rs:235-295, 533-564, 963-989). Uploaded once, resident in HBM.
 * ------------------------------------------------------------------------------------------ */
int  mcx_table_create(mcx_engine* e, int kind, const float* keys, const float* values, uint32_t n, mcx_table** out);
void mcx_table_release(mcx_table* t);
/* Host-side analysis results (also usable in tests): uniform-grid flag and guide-table bits. */
int  mcx_table_info(const mcx_table* t, uint32_t* n, float* inv_dk, uint32_t* guide_bits);
/* LDS bytes a launch stages for this table (key/value pairs or cells, CDF slopes, guide): what a module built with
 * tables_lds = 1 needs per workgroup for it. */
uint32_t mcx_table_lds_bytes(const mcx_table* t);
/* Host-side: the index map of the cell form, idx = floor(x * scale + c0) clamped to [0, n]: 1 + c is cell c, 0 and n
 * are sentinel cells {outside value, slope 0} for x left / right of the table (both table ends map inside). */
int  mcx_table_cell_map(const float* keys, uint32_t n, float* scale_out, float* c0_out);
/* CDF tables: log2 of the number of bucket-direct records the table was stored with, 0 if it has none (non-monotone
 * cdf or x column, n > 4096). */
int  mcx_table_has_direct(const mcx_table* t);
/* 1 if the table was stored with slope-intercept cells (PDF / log-PDF kinds on a strict f32-linspace grid), else 0. */
int  mcx_table_has_cells(const mcx_table* t);
/* 1 and the sentinel cells a cell_noclamp launch would add either side of table `t` when the call samples from
 * (dist_type, param1, param2[, cdf]) -- 0 when the table has no cell form or the range is unbounded / too wide. */
int  mcx_cell_pads(const mcx_table* t, int32_t dist_type, float param1, float param2, const mcx_table* cdf, int32_t guard_endpoints,
                   uint32_t* pad_l, uint32_t* pad_r);
/* The same from the table's keys alone, no GPU needed (the strict-grid index map of mcx_table_cell_map; for a custom
 * sampling distribution pass the range of its CDF table's x column with have_x = 1 -- only when that table's
 * mcx_table_facts.reach_known is 1). */
int  mcx_cell_pads_host(const float* keys, uint32_t n, int32_t dist_type, float param1, float param2, int32_t have_x, float x_min,
                        float x_max, int32_t guard_endpoints, uint32_t* pad_l, uint32_t* pad_r);
/* Host-side, no GPU needed: the per-cell line coefficients a PDF / log-PDF table on a strict f32-linspace grid is
 * stored with (value(x) = slope * x + intercept on cell c: the interpolant of src/distribution.rs:181-223 / 375-417 in
 * slope-intercept form, coefficients from f64). Returns 1 and fills cells_out[2 * (n - 1)] = {intercept, slope} per
 * cell (cells_out may be NULL), or 0 when the keys are not such a grid -- lookups then run the verified / searched
 * key-value path. */
int  mcx_table_cells(const float* keys, const float* values, uint32_t n, float* cells_out);
/* Host-side, no GPU needed: everything mcx_table_create derives from a table before it uploads it -- what a planner
 * needs to choose a module desc (tables_lds, cell_tables, cdf_direct, cell_noclamp) without a device. */
typedef struct mcx_table_facts {
    uint32_t struct_size;      /* sizeof(mcx_table_facts) in the caller's build */
    uint32_t n;
    uint32_t has_cells;        /* mcx_table_has_cells */
    uint32_t direct_bits;      /* mcx_table_has_direct */
    uint32_t guide_bits;       /* 0: no guide table (non-monotone keys or n > 4096) */
    uint32_t lds_bytes;        /* mcx_table_lds_bytes */
    float    inv_dk;           /* uniform-grid scale of the keys, 0 if they are not a uniform grid */
    float    value_min, value_max;
    uint32_t reach_known;      /* CDF tables: 1 if every draw stays within [value_min, value_max] -- the table has a guide
                                * or bucket-direct form (monotone, n <= 4096: the search ends in the right cell) and
                                * cdf[n-1] >= 1 (no u beyond the last node, where the cell's line is extrapolated) */
} mcx_table_facts;
int  mcx_table_analyse(int kind, const float* keys, const float* values, uint32_t n, mcx_table_facts* out);
int  mcx_table_facts_of(const mcx_table* t, mcx_table_facts* out);       /* the same of a resident table */

/* Performance planning of one call (the host layer of this repo calls it for its own plans). Given the
 * tables the call will bind (cdf: the custom sampling / proposal table or NULL; t0, t1: its PDF tables -- target, proposal -- for
 * an integrate module, its log-PDF tables for an MCMC module, NULL where none) and the sampler's parameters, fills the
 * fields of `d` that are guarantees about them: cell_tables (every table a strict grid), cell_noclamp (+ the LDS bytes of the
 * sentinel cells in *pad_bytes), cell_addr16, tables_lds (do they fit next to the kernel's scratch), cdf_direct, unit_params and,
 * when block is 0 and small tables favour it, block = 512. Set everything else (kind, k, dist_type, weight, p_table, q_table,
 * precise_sampler, rng, second_moments, walk, q_sampler, moment_family, user_tables, logpdf_analytic) before the call. The tuning
 * switches of the host layer apply (MCX_NO_NOCLAMP, MCX_NO_ADDR16, MCX_NO_DIRECT, MCX_DIRECT_MAX_ROWS, MCX_BLOCK). */
int  mcx_module_desc_fit(mcx_module_desc* d, const mcx_table* cdf, const mcx_table* t0, const mcx_table* t1, float param1, float param2,
                         uint32_t* pad_bytes);
/* The same without a device: the tables as mcx_table_analyse describes them, plus the keys (x column) of t0 / t1 for the pads.
 * What a GPU-less build step uses to compile exactly the modules a call on the GPU will build. */
int  mcx_module_desc_fit_host(mcx_module_desc* d, const mcx_table_facts* cdf, const mcx_table_facts* t0, const float* keys0,
                              const mcx_table_facts* t1, const float* keys1, float param1, float param2, uint32_t* pad_bytes);
/* mcx_module_build of a fitted desc, then the plan-time LDS decisions held against the code object's real static LDS: should the
 * staged tables (or cell_addr16's 64 KiB) not fit after all, the flags that need them are cleared in `d` and the module is rebuilt
 * -- a default-on optimisation falls back, it does not fail the launch. */
int  mcx_module_build_fitted(mcx_engine* e, const char* user_src, mcx_module_desc* d, const mcx_table* cdf, const mcx_table* t0,
                             const mcx_table* t1, uint32_t pad_bytes, mcx_module** out);

/* ------------------------------------------------------------------------------------------
 * Integration -- replaces _core.MonteCarloIntegrator.integrate (src/lib.rs:47-141) and
 * .integrate_is_tables (src/lib.rs:158-275): setup -> execute -> reduce.
 * ------------------------------------------------------------------------------------------ */
typedef struct mcx_integrate_params {
    uint32_t struct_size;        /* sizeof(mcx_integrate_params) in the caller's build: mcx_integrate_params_init() */
    uint32_t reserved0;          /* 0 */
    uint64_t n_samples;          /* requested; rounded UP to T*L like the reference */
    int64_t  target_threads;     /* <= 0: default 65536 */
    uint32_t seed;
    float    param1, param2;     /* uniform (min,max) / normal (mean,std) / exponential (lambda,-) */
    uint32_t rank, world;        /* shard selector; world = 1 for the whole grid */
    const mcx_table* cdf;        /* MCX_TABLE_CDF, custom distribution only */
    const mcx_table* target_pdf;   /* MCX_TABLE_PDF when desc.p_table */
    const mcx_table* proposal_pdf; /* MCX_TABLE_PDF when desc.q_table */
} mcx_integrate_params;
static inline void mcx_integrate_params_init(mcx_integrate_params* p) {      /* zero-fill, struct_size, world = 1 */
    uint32_t i;
    for (i = 0; i < sizeof(*p); ++i) ((unsigned char*)p)[i] = 0;
    p->struct_size = (uint32_t)sizeof(*p);
    p->world = 1;
}

/* sums_out[k] = sum over this rank's shard of f_k(x) (*p/q); n_eff_out = T*L of the WHOLE grid.
 * The reference's return value is sums / n_eff (mean of per-thread means of equal length). */
int mcx_integrate(mcx_engine* e, mcx_module* m, const mcx_integrate_params* p,
                  double* sums_out, uint64_t* n_eff_out);
/* Same, but the K sums are left in device memory at d_sums (K doubles) and the work is enqueued
 * on `stream` without a host sync: the caller runs the collective (RCCL all-reduce) ordered after it.
 * `stream` is a hipStream_t taken literally -- NULL is HIP's null stream (what torch calls its default
 * stream) -- or MCX_STREAM_ENGINE for the engine's own non-blocking stream. */
#define MCX_STREAM_ENGINE ((void*)(intptr_t)-1)
int mcx_integrate_device(mcx_engine* e, mcx_module* m, const mcx_integrate_params* p,
                         void* d_sums, void* stream, uint64_t* n_eff_out);

/* ------------------------------------------------------------------------------------------
 * MCMC -- replaces _core.MonteCarloIntegrator.integrate_mcmc (src/lib.rs:296-431).
 * ------------------------------------------------------------------------------------------ */
typedef struct mcx_mcmc_params {
    uint32_t struct_size;        /* sizeof(mcx_mcmc_params) in the caller's build: mcx_mcmc_params_init() */
    uint32_t n_steps, n_chains, n_burnin;
    int64_t  target_threads;     /* > 0 overrides n_chains (src/engine.rs:860) */
    uint32_t seed;
    float    param1, param2;     /* proposal parameters */
    uint32_t rank, world;
    const mcx_table* cdf;              /* custom proposal */
    const mcx_table* target_logpdf;    /* MCX_TABLE_LOGPDF, required unless desc.logpdf_analytic & 1 */
    const mcx_table* proposal_logpdf;  /* MCX_TABLE_LOGPDF, required unless desc.q_sampler or desc.logpdf_analytic & 2 */
    float    x0;                 /* random-walk modules: chains start at x0 + d_0 (d_0 = the iter-0 draw); else ignored */
    float    target_accept;      /* MCX_WALK_ADAPTIVE: acceptance rate the step scale is tuned towards (0 < a < 1) */
} mcx_mcmc_params;
static inline void mcx_mcmc_params_init(mcx_mcmc_params* p) {                /* zero-fill, struct_size, world = 1 */
    uint32_t i;
    for (i = 0; i < sizeof(*p); ++i) ((unsigned char*)p)[i] = 0;
    p->struct_size = (uint32_t)sizeof(*p);
    p->world = 1;
    p->target_accept = 0.44f;
}

/* sums_out[0..k) = sum over this rank's chains and all sampling steps of f_k(x_t);
 * sums_out[k] = number of accepted steps (burn-in included); n_eff_out = padded chains * n_steps.
 * With desc.second_moments: [0..k) sums of f, [k..2k) sums of f^2, [2k] accepted steps,
 * [2k+1..3k+1) sums over chains of (chain mean of f_i)^2. MCX_WALK_ADAPTIVE appends the sum of the chains' final
 * step scales as the last row. */
int mcx_mcmc(mcx_engine* e, mcx_module* m, const mcx_mcmc_params* p,
             double* sums_out, uint64_t* n_eff_out);
int mcx_mcmc_device(mcx_engine* e, mcx_module* m, const mcx_mcmc_params* p,
                    void* d_sums, void* stream, uint64_t* n_eff_out);

/* ------------------------------------------------------------------------------------------
 * One host thread, several devices. The reference is single-device (ComputeEngine::new picks one adapter,
 * src/engine.rs:91-131); this is the C-level equivalent of the one-process-per-GPU path the Python layer runs over
 * torch.distributed. The shards params[r] (rank = r, world = n) are enqueued on engines[r] without waiting in
 * between. modules[r] must have been built on engines[r] from the same source and desc; tables in params[r] belong
 * to engines[r].
 *
 * mcx_*_multi (no RCCL): the n results of mcx_result_rows doubles each are copied back (all copies in flight, then
 * one wait per device) and added on the host in rank order. Several engines may share a device (how the tests
 * exercise it on one GPU).
 *
 * mcx_*_comm (RCCL over xGMI): mcx_comm_create runs ncclCommInitAll over the engines' devices (one engine per
 * distinct device); a call ends with ONE grouped ncclAllReduce (sum of mcx_result_rows doubles, in place in each
 * engine's result buffer, on each engine's stream) and one copy of K doubles from rank 0. librccl is bound at run
 * time like the HIP runtime (the copy already mapped in the process, else MCX_RCCL, else the system's).
 * ------------------------------------------------------------------------------------------ */
int mcx_integrate_multi(mcx_engine* const* engines, mcx_module* const* modules, const mcx_integrate_params* const* params,
                        int n, double* sums_out, uint64_t* n_eff_out);
int mcx_mcmc_multi(mcx_engine* const* engines, mcx_module* const* modules, const mcx_mcmc_params* const* params,
                   int n, double* sums_out, uint64_t* n_eff_out);


/* ------------------------------------------------------------------------------------------
 * Self-test: the integer side of the random streams computed on the GPU by the device functions the kernels use,
 * returned raw for bit-exact comparison (tests/test_gpu_kat.py). For each of the n triples (seed, idx, iter):
 * combined = seed + idx*7199369 + iter*15485863 and hash = pcg_hash(combined) (src/distribution.rs:62-73); stepped =
 * the same hash reached by stepping the LCG state with adds as the hot loops do; angle = the hash word that feeds the
 * Box-Muller angle (hash without its final xorshift); u = float(hash) * 2^-32. For each of the n_philox
 * (counter[4], key[2]) pairs: the Philox4x32-10 block (Random123).
 * ------------------------------------------------------------------------------------------ */
int mcx_selftest_streams(mcx_engine* e, uint32_t n, const uint32_t* seed_idx_iter, uint32_t* combined_out,
                         uint32_t* hash_out, uint32_t* stepped_out, uint32_t* angle_out, float* u_out,
                         uint32_t n_philox, const uint32_t* counters, const uint32_t* keys, uint32_t* philox_out);
/* Work bound of one main-kernel launch in samples / chain-steps (0 restores the default, ~1e11 or
 * MCX_MAX_LAUNCH_UNITS): larger calls are split into several launches folded together (process-wide). */
void mcx_set_max_launch_units(uint64_t units);

typedef struct mcx_comm mcx_comm;
const char* mcx_rccl_library(void);                 /* which librccl was bound ("" if none could be) */
int  mcx_comm_create(mcx_engine* const* engines, int n, mcx_comm** out);
void mcx_comm_destroy(mcx_comm* c);
int  mcx_comm_size(const mcx_comm* c);
/* modules[r] / params[r] belong to the r-th engine the communicator was created with (rank = r, world = size). */
int mcx_integrate_comm(mcx_comm* c, mcx_module* const* modules, const mcx_integrate_params* const* params,
                       double* sums_out, uint64_t* n_eff_out);
int mcx_mcmc_comm(mcx_comm* c, mcx_module* const* modules, const mcx_mcmc_params* const* params,
                  double* sums_out, uint64_t* n_eff_out);

#ifdef __cplusplus
}
#endif
#endif /* MCX_H */
