// mcx_args.h -- kernel argument blocks shared by the host runtime (mcx_runtime.cpp) and the
// device kernels (mcx_kernels.hpp). Plain C layout, passed to the kernels by value.
//
// They replace the reference's uniform buffers IntegrationParams / McmcParams /
// DistributionParamsBuffer (src/engine.rs:7-37) and its storage-buffer bindings
// (src/engine.rs:334-390, 688-717, 1038-1136).
#pragma once

#ifdef __HIPCC_RTC__
typedef unsigned int       mcx_u32;
typedef unsigned long long mcx_u64;
#else
#include <stdint.h>
typedef uint32_t mcx_u32;
typedef uint64_t mcx_u64;
#endif

// One lookup table resident in HBM: n interleaved {key,value} float pairs.
typedef struct McxTableDesc {
    const float* kv;        // device pointer, 2*n floats ({cdf,x} or {x,pdf})
    const mcx_u32* guide;   // device pointer, 1<<guide_bits packed windows (lo | hi<<16), or null
    mcx_u32 n;
    mcx_u32 guide_bits;
    float   inv_dk;         // (n-1)/(key[n-1]-key[0]) if the keys form a uniform grid, else 0
    float   cell_c0;        // cell index (with the leading sentinel) = floor(x * cell_scale + cell_c0), see cell_map()
    const float* cells;     // device pointer, n+1 {intercept, slope} pairs: {outside, 0}, the n-1 cells, {outside, 0}
                            // (PDF / log-PDF tables on a strict grid), or null
    const float* slopes;    // device pointer, n inverse-CDF slopes dx/dcdf per cell (CDF tables), or null
    float   cell_scale;
    mcx_u32 direct_bits;    // CDF tables: log2 of the number of bucket-direct records, 0 = none
    const float* direct;    // device pointer, 2 floats per bucket: {x_b, slope * 2^-32} or {lo | hi << 16, -0.0f}
    mcx_u32 pad_l, pad_r;   // cell form, MCX_CELL_NOCLAMP launches: extra {outside, 0} sentinels staged either side (host: cell_pads)
} McxTableDesc;

// K1 / K2: plain and importance-sampling integration.
//
// Logical grid (reference): idx in [0,T), i in [0,L). A launch covers idx in
// [idx_begin, idx_begin+idx_count) x units in [unit_begin, unit_end), where a unit is one loop
// iteration i (uniform / exponential / custom) or one Box-Muller pair j <-> iterations (2j, 2j+1)
// (normal). Physical thread g handles logical index idx_begin + g % idx_count and the chunk
// g / idx_count of `units_per_chunk` consecutive units.
typedef struct McxIntegrateArgs {
    mcx_u32 seed;
    mcx_u32 idx_begin;
    mcx_u32 idx_count;
    mcx_u32 unit_begin;
    mcx_u32 unit_end;
    mcx_u32 units_per_chunk;
    mcx_u32 n_chunks;
    mcx_u32 loops_per_thread;   // L: a normal pair's second half is used only if 2j+1 < L
    float   param1;             // min / mean / lambda
    float   param2;             // max / std
    mcx_u32 _pad0, _pad1;
    McxTableDesc cdf;           // custom sampling distribution  {cdf, x}
    McxTableDesc target_pdf;    // IS: target  {x, pdf}   (n == 0: analytic mcx_pdf_p)
    McxTableDesc proposal_pdf;  // IS: proposal {x, pdf}  (n == 0: analytic mcx_pdf_q)
    double* partials;           // [gridDim.x][K] per-workgroup partial sums
} McxIntegrateArgs;

// K3: independent-proposal Metropolis-Hastings, one chain per logical thread.
typedef struct McxMcmcArgs {
    mcx_u32 seed;
    mcx_u32 chain_begin;        // first logical chain index of this launch
    mcx_u32 chain_count;        // chains in this launch
    mcx_u32 n_steps;
    mcx_u32 n_burnin;
    float   x0;                 // random-walk start (chains start at x0 + first draw)
    float   param1;             // proposal min / mean / lambda
    float   param2;             // proposal max / std
    float   target_accept;      // MCX_WALK 3: acceptance rate the per-chain step scale adapts towards during burn-in
    mcx_u32 _pad0;
    McxTableDesc cdf;           // custom proposal sampling {cdf, x}
    McxTableDesc target_logpdf;   // {x, log p}
    McxTableDesc proposal_logpdf; // {x, log q}
    double* partials;           // [gridDim.x][rows]; column MCX_K = accepted-step count
    // Time segments (host: mcmc_impl; independence sampler with a normal proposal, either stream): this launch runs steps it_begin .. it_end of
    // every chain (1-based, burn-in included; 0 / 0 = all of them). A launch with it_begin > 1 resumes each chain from
    // seg_state[chain - chain_begin] = {x, w}; one with seg_state != null leaves it there at its end. The random streams
    // are functions of (seed, chain, it), so nothing else carries over.
    mcx_u32 it_begin, it_end;
    float*  seg_state;
} McxMcmcArgs;
