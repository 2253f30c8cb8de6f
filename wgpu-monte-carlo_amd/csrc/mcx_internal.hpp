// mcx_internal.hpp -- declarations shared by mcx_plan.cpp and mcx_runtime.cpp
#pragma once
#include "../../include/mcx.h"

#include <string>
#include <vector>

namespace mcx {

// Record an error message for mcx_last_error() and return `code`.
int fail(int code, const std::string& msg);

struct LaunchPlan {
    uint32_t units_per_chunk;
    uint32_t n_chunks;
    uint32_t n_blocks;      // 0: nothing to launch (empty shard)
};

LaunchPlan plan_integrate(const mcx_shard& s, uint32_t target_phys, uint32_t block);

void analyse_table(int kind, const float* keys, uint32_t n, float* inv_dk, std::vector<uint32_t>* guide,
                   uint32_t* guide_bits);

// PDF / log-PDF tables whose keys are an f32 linspace: per-cell line coefficients {a_c, s_c}, c < n-1, with
// value(x) = s_c * x + a_c on cell c (the reference's interpolant, distribution.rs:181-223, in slope-intercept form,
// coefficients computed in f64). Empty when the keys are not such a grid (then the verified / searched path runs).
// CDF tables: slope[c] = (x[c+1] - x[c]) / (cdf[c+1] - cdf[c]) of cell c < n-1 (0 where the reference's lookup
// returns x[c] because the cell is narrower than 1e-10, distribution.rs:152-155), slope[n-1] = 0.
void build_cdf_slopes(const float* cdf, const float* x, uint32_t n, std::vector<float>* slopes);

// CDF tables: the bucket-direct inverse (see mcx_plan.cpp). direct = 2^bits records {x_b, slope * 2^-32} or
// {lo | hi << 16, -0.0f}; empty when the table does not qualify.
void build_cdf_direct(const float* cdf, const float* x, uint32_t n, std::vector<float>* direct, uint32_t* direct_bits);

void cell_map(const float* keys, uint32_t n, float* scale, float* c0);
void build_cells(const float* keys, const float* values, uint32_t n, std::vector<float>* cells);

}  // namespace mcx
