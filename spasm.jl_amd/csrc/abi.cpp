// abi.cpp -- host-side part of the C ABI: containers, ownership, options, logging, synthetic inputs.
//
// Mirrors the libspasm entry points SpaSM.jl binds (reference src/SpaSM.jl, line cited per function).
// No arithmetic of the hot path lives here: echelonize / kernel / transpose are in engine.hip.
#include "common.hpp"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>
#include <algorithm>
#include <cmath>

extern "C" {

// data symbol poked by SpaSM.log(), reference src/SpaSM.jl:34-46
SPASM_API int (*logcallback)(char *) = nullptr;

} // extern "C"

static thread_local std::string g_last_error;

void spasm_set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    spasm_logf("[spasm_amd] ERROR: %s\n", buf);
}

void spasm_clear_error() { g_last_error.clear(); }

// Progress text goes to the callback when SpaSM.log() installed one, else to fd 2 -- which Julia
// redirects around the ccall unless verbose (reference src/SpaSM.jl:838-858).
void spasm_logf(const char *fmt, ...)
{
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (logcallback) logcallback(buf);
    else fputs(buf, stderr);
}

extern "C" {

SPASM_API const char *spasm_amd_last_error(void) { return g_last_error.c_str(); }

SPASM_API double spasm_wtime(void) // reference src/SpaSM.jl:430
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

SPASM_API void spasm_field_init(i64 p, spasm_field F) // commented binding at reference src/SpaSM.jl:65; values :73-76
{
    F->p = p;
    F->halfp = p / 2;
    F->mhalfp = p / 2 - p + 1;
    F->dinvp = 1.0 / (double)p;
}

SPASM_API i64 spasm_nnz(const struct spasm_csr *A) { return A->p[A->n]; } // reference src/SpaSM.jl:432

// reference src/SpaSM.jl:441.  libc allocator so that spasm_csr_free / realloc compose (":434-439").
SPASM_API struct spasm_csr *spasm_csr_alloc(int n, int m, i64 nzmax, i64 prime, bool with_values)
{
    if (n < 0 || m < 0 || nzmax < 0) { spasm_set_error("spasm_csr_alloc: negative size"); return nullptr; }
    struct spasm_csr *A = (struct spasm_csr *)malloc(sizeof *A);
    if (!A) return nullptr;
    i64 cap = nzmax > 0 ? nzmax : 1;
    A->nzmax = nzmax;
    A->n = n;
    A->m = m;
    A->p = (i64 *)malloc(sizeof(i64) * ((size_t)n + 1));
    A->j = (int *)malloc(sizeof(int) * (size_t)cap);
    A->x = with_values ? (spasm_ZZp *)malloc(sizeof(spasm_ZZp) * (size_t)cap) : nullptr;
    if (!A->p || !A->j || (with_values && !A->x)) {
        free(A->p); free(A->j); free(A->x); free(A);
        spasm_set_error("spasm_csr_alloc: out of memory (n=%d nzmax=%lld)", n, (long long)nzmax);
        return nullptr;
    }
    A->p[0] = 0;
    spasm_field_init(prime, A->field);
    return A;
}

SPASM_API void spasm_csr_realloc(struct spasm_csr *A, i64 nzmax) // reference src/SpaSM.jl:447
{
    if (nzmax < 0) nzmax = spasm_nnz(A);
    i64 cap = nzmax > 0 ? nzmax : 1;
    A->j = (int *)realloc(A->j, sizeof(int) * (size_t)cap);
    if (A->x) A->x = (spasm_ZZp *)realloc(A->x, sizeof(spasm_ZZp) * (size_t)cap);
    A->nzmax = nzmax;
}

SPASM_API void spasm_csr_resize(struct spasm_csr *A, int n, int m) // reference src/SpaSM.jl:449
{
    A->m = m;
    if (n != A->n) {
        i64 last = A->p[n < A->n ? n : A->n];
        A->p = (i64 *)realloc(A->p, sizeof(i64) * ((size_t)n + 1));
        for (int i = A->n + 1; i <= n; i++) A->p[i] = last; // new rows are empty
        A->n = n;
    }
}

SPASM_API void spasm_csr_free(struct spasm_csr *A) // reference src/SpaSM.jl:451 (called from a GC finalizer)
{
    if (!A) return;
    free(A->p); free(A->j); free(A->x); free(A);
}

SPASM_API void spasm_lu_free(struct spasm_lu *N) // reference src/SpaSM.jl:463; U/L are wrapped own=false (:289,:292)
{
    if (!N) return;
    spasm_csr_free(N->U);
    spasm_csr_free(N->L);
    free(N->qinv);
    free(N->p);
    free(N);
}

SPASM_API int spasm_get_num_threads(void) { return 1; } // reference src/SpaSM.jl:470 (the engine's parallelism is on the device)
SPASM_API int spasm_get_thread_num(void) { return 0; }  // reference src/SpaSM.jl:475

// reference src/SpaSM.jl:817; field list :325-343.  Values are libspasm's defaults as recalled
// (SURVEY 8a row a3); the Julia struct is zero-filled before this call.
SPASM_API void spasm_echelonize_init_opts(struct echelonize_opts *o)
{
    memset(o, 0, sizeof *o);
    o->enable_greedy_pivot_search = true;
    o->enable_tall_and_skinny = true;
    o->enable_dense = true;
    o->enable_GPLU = true;
    o->L = false;
    o->complete = false;
    o->min_pivot_proportion = 0.1;
    o->max_round = 3;
    o->sparsity_threshold = 0.05;
    o->dense_block_size = 1000;
    o->low_rank_ratio = 0.5;
    o->tall_and_skinny_ratio = 5;
    o->low_rank_start_weight = -1;
}

// ------------------------------------------------------------------------------------------------
// Synthetic inputs: splitmix64-seeded xoshiro256**, one independent stream per row so
// the bytes do not depend on how many threads generate them.
// ------------------------------------------------------------------------------------------------
} // extern "C"

namespace {
struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x)
    {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed)
    {
        for (int i = 0; i < 4; i++) s[i] = splitmix(seed);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    // unbiased integer in [0, n)
    uint64_t below(uint64_t n)
    {
        uint64_t lim = UINT64_MAX - UINT64_MAX % n;
        uint64_t r;
        do r = next(); while (r >= lim);
        return r % n;
    }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

inline uint64_t row_seed(uint64_t seed, uint64_t row)
{
    uint64_t x = seed ^ (row * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull);
    return Rng::splitmix(x);
}

// columns of one row, in generation order (unsorted)
void gen_row_cols(int kind, int m, double density, int row_nnz, Rng &rng, std::vector<int> &cols)
{
    cols.clear();
    if (m <= 0) return;
    if (kind == 0) {
        if (density >= 1.0) {
            for (int j = 0; j < m; j++) cols.push_back(j);
        } else if (density > 0.0) {
            const double l1d = log1p(-density);
            double pos = -1.0;
            for (;;) {
                double u = rng.unit();
                if (u <= 0.0) u = 1e-300;
                pos += floor(log(u) / l1d) + 1.0; // geometric gap
                if (pos >= (double)m) break;
                cols.push_back((int)pos);
            }
        }
        for (size_t k = cols.size(); k > 1; k--) std::swap(cols[k - 1], cols[rng.below(k)]); // unsorted
    } else {
        int k = row_nnz < m ? row_nnz : m;
        while ((int)cols.size() < k) {
            int c = (int)rng.below((uint64_t)m);
            bool dup = false;
            for (int d : cols) if (d == c) { dup = true; break; }
            if (!dup) cols.push_back(c);
        }
    }
}
} // namespace

extern "C" SPASM_API struct spasm_csr *spasm_amd_synth_csr(int kind, int n, int m, double density, int row_nnz,
                                                            i64 prime, uint64_t seed)
{
    spasm_clear_error();
    if (n < 0 || m < 0 || prime <= 2 || prime > 0xfffffffbLL || (kind != 0 && kind != 1)) {
        spasm_set_error("spasm_amd_synth_csr: bad arguments");
        return nullptr;
    }
    std::vector<i64> rp((size_t)n + 1, 0);
    // pass 1: row lengths
#pragma omp parallel
    {
        std::vector<int> cols;
#pragma omp for schedule(static)
        for (int i = 0; i < n; i++) {
            Rng rng(row_seed(seed, (uint64_t)i));
            gen_row_cols(kind, m, density, row_nnz, rng, cols);
            rp[(size_t)i + 1] = (i64)cols.size();
        }
    }
    for (int i = 0; i < n; i++) rp[(size_t)i + 1] += rp[(size_t)i];
    struct spasm_csr *A = spasm_csr_alloc(n, m, rp[(size_t)n], prime, true);
    if (!A) return nullptr;
    memcpy(A->p, rp.data(), sizeof(i64) * ((size_t)n + 1));
    const i64 halfp = prime / 2;
    // pass 2: same streams again, now with values drawn after the columns
#pragma omp parallel
    {
        std::vector<int> cols;
#pragma omp for schedule(static)
        for (int i = 0; i < n; i++) {
            Rng rng(row_seed(seed, (uint64_t)i));
            gen_row_cols(kind, m, density, row_nnz, rng, cols);
            i64 base = A->p[i];
            for (size_t k = 0; k < cols.size(); k++) {
                i64 v = 1 + (i64)rng.below((uint64_t)(prime - 1)); // nonzero residue in [1, p-1]
                if (v > halfp) v -= prime;                         // balanced (reference src/SpaSM.jl:955-958)
                A->j[base + (i64)k] = cols[k];
                A->x[base + (i64)k] = (spasm_ZZp)v;
            }
        }
    }
    return A;
}
