// mcx_core.cpp -- the reference's native module, call for call (include/mcx.h: mcx_core_*).
//
// The reference's PyO3 class `_core.MonteCarloIntegrator` (src/lib.rs:17-431) owns a ComputeEngine and has three methods --
// integrate, integrate_is_tables, integrate_mcmc -- that take WGSL strings, a distribution, float32 tables and sizes, and
// return K float32 means. mcx_core is that object over libmcx: an engine plus what the reference rebuilds on every call and
// libmcx keeps (src/engine.rs:325-331 recompiles the shader per call; :235-295 re-uploads the tables): resident tables by
// content, planned + compiled modules by payload. A Rust / C / Go host of src/lib.rs forwards its arguments unchanged;
// wgpu_montecarlo/_core.py is this file's ctypes binding.
//
// Written against the public C ABI only (mcx_wgsl_plan, mcx_module_desc_fit, mcx_module_build_fitted, mcx_table_create,
// mcx_integrate, mcx_mcmc): it is a client of libmcx that happens to live inside it.
#include <cstdlib>
#include <cstring>
#include <list>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mcx.h"
#include "mcx_internal.hpp"

namespace {

// 128 bits of content hash, eight bytes at a time (two independent multiply-xorshift lanes; not cryptographic: it names a cache entry)
struct Key { uint64_t a, b; bool operator==(const Key& o) const { return a == o.a && b == o.b; } };
struct KeyHash { size_t operator()(const Key& k) const { return (size_t)(k.a ^ (k.b * 0x9E3779B97F4A7C15ull)); } };
struct Hasher {
    uint64_t a = 14695981039346656037ull, b = 0x9E3779B97F4A7C15ull;
    void word(uint64_t w) {
        a = (a ^ w) * 0x100000001B3ull;            a ^= a >> 32;
        b = (b + w) * 0xD6E8FEB86659FD93ull;       b ^= b >> 29;
    }
    void add(const void* p, size_t n) {
        const unsigned char* c = (const unsigned char*)p;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, c + i, 8); word(w); }
        uint64_t tail = 0;
        if (i < n) memcpy(&tail, c + i, n - i);
        word(tail ^ ((uint64_t)n << 56));          // the length closes the run: "ab" + "c" and "a" + "bc" differ
    }
    template <class T> void pod(const T& v) { add(&v, sizeof v); }
    Key key() const { return {a, b}; }
};

// least-recently-used map of owned objects
template <class V, void (*Release)(V)>
struct Lru {
    size_t cap;
    std::list<std::pair<Key, V>> order;                                   // front = most recent
    std::unordered_map<Key, typename std::list<std::pair<Key, V>>::iterator, KeyHash> index;
    explicit Lru(size_t c) : cap(c) {}
    bool get(const Key& k, V* out) {
        auto it = index.find(k);
        if (it == index.end()) return false;
        order.splice(order.begin(), order, it->second);
        *out = it->second->second;
        return true;
    }
    void put(const Key& k, V v) {
        order.emplace_front(k, v);
        index[k] = order.begin();
        while (order.size() > cap) {
            Release(order.back().second);
            index.erase(order.back().first);
            order.pop_back();
        }
    }
    void clear() { for (auto& kv : order) Release(kv.second); order.clear(); index.clear(); }
};

struct Planned { mcx_module* module; mcx_module_desc desc; };
void release_table(mcx_table* t) { mcx_table_release(t); }
void release_planned(Planned p) { mcx_module_release(p.module); }

}  // namespace

struct mcx_core {
    mcx_engine* engine = nullptr;
    int32_t math = 1;
    std::mutex mu;
    Lru<mcx_table*, release_table> tables{64};
    Lru<Planned, release_planned> modules{128};
};

namespace {

// a resident table for these contents: uploaded once, found again by content (the reference re-creates its storage buffers per call)
int table_for(mcx_core* c, int kind, const float* keys, const float* values, uint32_t n, mcx_table** out, Key* content) {
    *out = nullptr;
    *content = {0, 0};
    if (!keys || !values || n == 0u) return MCX_OK;
    Hasher h;
    h.pod(kind); h.pod(n); h.add(keys, n * sizeof(float)); h.add(values, n * sizeof(float));
    *content = h.key();
    if (c->tables.get(h.key(), out)) return MCX_OK;
    if (int rc = mcx_table_create(c->engine, kind, keys, values, n, out)) return rc;
    c->tables.put(h.key(), *out);
    return MCX_OK;
}

// the planned, fitted, compiled module of one payload shape
int module_for(mcx_core* c, const mcx_wgsl_program& prog, int32_t block_hint, const mcx_table* cdf, const mcx_table* t0, const mcx_table* t1,
               const Key (&contents)[3], Planned* out) {
    Hasher h;
    h.pod(prog.kind); h.pod(prog.k); h.pod(prog.dist_type); h.pod(prog.param1); h.pod(prog.param2); h.pod(prog.math);
    h.pod(prog.have_target_table); h.pod(prog.have_proposal_table); h.pod(prog.target_dist_type); h.pod(prog.target_param1);
    h.pod(prog.target_param2); h.pod(block_hint); h.pod(contents);          // the tables by content: the plan depends on what they are, not where
    for (int i = 0; i < prog.k; ++i) {
        const char* f = prog.functions[i] ? prog.functions[i] : "";
        const size_t n = strlen(f);
        h.pod(n); h.add(f, n);
    }
    if (c->modules.get(h.key(), out)) return MCX_OK;
    mcx_module_desc desc;
    char* src = nullptr;
    if (int rc = mcx_wgsl_plan(&prog, &desc, &src)) return rc;
    int rc = MCX_OK;
    mcx_module* m = nullptr;
    if (prog.math == 0) rc = mcx_module_build(c->engine, src, &desc, &m);              // literal: the module as planned
    else {
        if (block_hint) desc.block = block_hint;
        uint32_t pad_bytes = 0u;
        // the proposal's log-PDF table is not bound when log q comes from the deviate
        const mcx_table* q = (prog.kind == MCX_KIND_MCMC && desc.q_sampler) ? nullptr : t1;
        rc = mcx_module_desc_fit(&desc, cdf, t0, q, prog.param1, prog.param2, &pad_bytes);
        if (!rc) rc = mcx_module_build_fitted(c->engine, src, &desc, cdf, t0, q, pad_bytes, &m);
    }
    mcx_free(src);
    if (rc) return rc;
    *out = {m, desc};
    c->modules.put(h.key(), *out);
    return MCX_OK;
}

int check_core(const mcx_core* c, const mcx_core_tables* t, const float* values_out, const char* who) {
    if (!c || !values_out) return mcx::fail(MCX_E_INVALID, std::string(who) + ": null argument");
    if (t && t->struct_size != sizeof(mcx_core_tables)) return mcx::fail(MCX_E_INVALID, std::string(who) + ": tables of another ABI version (mcx_core_tables_init)");
    return MCX_OK;
}

}  // namespace

extern "C" {

int mcx_core_create(int device, int32_t math, mcx_core** out) {
    if (!out) return mcx::fail(MCX_E_INVALID, "mcx_core_create: out is null");
    if (math < 0 || math > 2) return mcx::fail(MCX_E_INVALID, "math must be one of ('precise', 'default', 'fast')");
    mcx_core* c = new mcx_core();
    c->math = math;
    if (int rc = mcx_engine_create(device, &c->engine)) { delete c; return rc; }
    *out = c;
    return MCX_OK;
}

void mcx_core_destroy(mcx_core* c) {
    if (!c) return;
    c->modules.clear();
    c->tables.clear();
    mcx_engine_destroy(c->engine);
    delete c;
}

mcx_engine* mcx_core_engine(mcx_core* c) { return c ? c->engine : nullptr; }

int mcx_core_integrate(mcx_core* c, const char* const* functions, int32_t k, int32_t dist_type, float param1, float param2,
                       uint64_t n_samples, uint32_t seed, const mcx_core_tables* tb, int64_t target_threads, float* values_out) {
    if (int rc = check_core(c, tb, values_out, "mcx_core_integrate")) return rc;
    if (k <= 0 || !functions) return mcx::fail(MCX_E_INVALID, "At least one function is required");                 // src/lib.rs:61-65
    std::lock_guard<std::mutex> lk(c->mu);
    mcx_table *cdf = nullptr, *target = nullptr, *proposal = nullptr;
    Key contents[3] = {{0, 0}, {0, 0}, {0, 0}};
    if (dist_type == MCX_DIST_CUSTOM) {
        if (!tb || !tb->x_table || !tb->cdf_table || tb->n_cdf == 0u)
            return mcx::fail(MCX_E_RUNTIME, "Failed to setup integration: custom distribution requires x_table and cdf_table");
        if (int rc = table_for(c, MCX_TABLE_CDF, tb->cdf_table, tb->x_table, tb->n_cdf, &cdf, &contents[0])) return rc;
    }
    if (tb) {
        if (int rc = table_for(c, MCX_TABLE_PDF, tb->target_x, tb->target_v, tb->n_target, &target, &contents[1])) return rc;
        if (int rc = table_for(c, MCX_TABLE_PDF, tb->proposal_x, tb->proposal_v, tb->n_proposal, &proposal, &contents[2])) return rc;
    }
    mcx_wgsl_program prog;
    mcx_wgsl_program_init(&prog);
    prog.kind = MCX_KIND_INTEGRATE; prog.k = k; prog.functions = functions; prog.dist_type = dist_type;
    prog.param1 = param1; prog.param2 = param2; prog.math = c->math;
    prog.have_target_table = target ? 1 : 0; prog.have_proposal_table = proposal ? 1 : 0;
    Planned pl;
    if (int rc = module_for(c, prog, 0, cdf, target, proposal, contents, &pl)) return rc;
    mcx_integrate_params p;
    mcx_integrate_params_init(&p);
    p.n_samples = n_samples; p.target_threads = target_threads; p.seed = seed; p.param1 = param1; p.param2 = param2;
    p.cdf = cdf; p.target_pdf = target; p.proposal_pdf = proposal;       // bound alike whether the module weights with them or its functions read them
    std::vector<double> sums((size_t)mcx_result_rows(&pl.desc));
    uint64_t n_eff = 0;
    if (int rc = mcx_integrate(c->engine, pl.module, &p, sums.data(), &n_eff)) return rc;
    for (int i = 0; i < k; ++i) values_out[i] = (float)(sums[(size_t)i] / (double)n_eff);          // the reference's CPU mean, src/lib.rs:129-138
    return MCX_OK;
}

int mcx_core_mcmc(mcx_core* c, const char* const* functions, int32_t k, int32_t proposal_dist_type, float param1, float param2,
                  int32_t target_dist_type, float target_param1, float target_param2, uint32_t n_steps, uint32_t n_chains, uint32_t n_burnin,
                  uint32_t seed, const mcx_core_tables* tb, int64_t target_threads, float* values_out) {
    if (int rc = check_core(c, tb, values_out, "mcx_core_mcmc")) return rc;
    if (k <= 0 || !functions) return mcx::fail(MCX_E_INVALID, "At least one function is required");
    if (n_steps == 0u) return mcx::fail(MCX_E_INVALID, "n_steps must be positive");                                  // src/lib.rs:332-336
    if (n_chains == 0u) return mcx::fail(MCX_E_INVALID, "n_chains must be positive");                                // src/lib.rs:338-342
    std::lock_guard<std::mutex> lk(c->mu);
    mcx_table *cdf = nullptr, *target = nullptr, *proposal = nullptr;
    Key contents[3] = {{0, 0}, {0, 0}, {0, 0}};
    if (proposal_dist_type == MCX_DIST_CUSTOM) {
        if (!tb || !tb->x_table || !tb->cdf_table || tb->n_cdf == 0u)
            return mcx::fail(MCX_E_RUNTIME, "Failed to setup integration: custom distribution requires x_table and cdf_table");
        if (int rc = table_for(c, MCX_TABLE_CDF, tb->cdf_table, tb->x_table, tb->n_cdf, &cdf, &contents[0])) return rc;
    }
    mcx_wgsl_program prog;
    mcx_wgsl_program_init(&prog);
    prog.kind = MCX_KIND_MCMC; prog.k = k; prog.functions = functions; prog.dist_type = proposal_dist_type;
    prog.param1 = param1; prog.param2 = param2; prog.math = c->math;
    prog.have_target_table = (tb && tb->target_x && tb->target_v && tb->n_target) ? 1 : 0;
    prog.have_proposal_table = (tb && tb->proposal_x && tb->proposal_v && tb->n_proposal) ? 1 : 0;
    prog.target_dist_type = target_dist_type; prog.target_param1 = target_param1; prog.target_param2 = target_param2;
    if (prog.have_target_table) if (int rc = table_for(c, MCX_TABLE_LOGPDF, tb->target_x, tb->target_v, tb->n_target, &target, &contents[1])) return rc;
    // a normal proposal's log q comes from its own deviate unless math is precise: its table is then neither uploaded nor bound
    const bool q_from_deviate = c->math != 0 && proposal_dist_type == MCX_DIST_NORMAL;
    if (prog.have_proposal_table && !q_from_deviate)
        if (int rc = table_for(c, MCX_TABLE_LOGPDF, tb->proposal_x, tb->proposal_v, tb->n_proposal, &proposal, &contents[2])) return rc;
    int32_t block_hint = 0;
    if (c->math != 0) {                                           // the workgroup size a small chain count wants (one chain per thread)
        mcx_dispatch d;
        if (int rc = mcx_mcmc_dispatch_config(n_chains, target_threads, &d)) return rc;
        const uint32_t hint = mcx_mcmc_block_hint(d.total_threads);
        block_hint = hint >= 1024u ? 0 : (int32_t)hint;
    }
    Planned pl;
    if (int rc = module_for(c, prog, block_hint, cdf, target, proposal, contents, &pl)) return rc;
    mcx_mcmc_params p;
    mcx_mcmc_params_init(&p);
    p.n_steps = n_steps; p.n_chains = n_chains; p.n_burnin = n_burnin; p.target_threads = target_threads; p.seed = seed;
    p.param1 = param1; p.param2 = param2; p.cdf = cdf; p.target_logpdf = target; p.proposal_logpdf = proposal;
    std::vector<double> sums((size_t)mcx_result_rows(&pl.desc));
    uint64_t n_eff = 0;
    if (int rc = mcx_mcmc(c->engine, pl.module, &p, sums.data(), &n_eff)) return rc;
    for (int i = 0; i < k; ++i) values_out[i] = (float)(sums[(size_t)i] / (double)n_eff);
    return MCX_OK;
}

}  // extern "C"
