// abi.cpp -- host-side part of the C ABI: containers, ownership, options, logging, synthetic inputs.
//
// Mirrors the libspasm entry points SpaSM.jl binds (reference src/SpaSM.jl, line cited per function).
// No arithmetic of the hot path lives here: echelonize / kernel / transpose are in engine.hip.
#include "common.hpp"
#include <hip/hip_runtime.h>
#include "zp.hpp"
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

// ------------------------------------------------------------------------------------------------
// triplets and the SMS text format (reference src/SpaSM.jl:453-529; format :1029-1042, :1063-1086)
// ------------------------------------------------------------------------------------------------
SPASM_API struct spasm_triplet *spasm_triplet_alloc(int n, int m, i64 nzmax, i64 prime, bool with_values)
{
    struct spasm_triplet *T = (struct spasm_triplet *)malloc(sizeof *T);
    if (!T) return nullptr;
    const i64 cap = nzmax > 0 ? nzmax : 1;
    T->nzmax = nzmax;
    T->nz = 0;
    T->n = n;
    T->m = m;
    T->i = (int *)malloc(sizeof(int) * (size_t)cap);
    T->j = (int *)malloc(sizeof(int) * (size_t)cap);
    T->x = with_values ? (spasm_ZZp *)malloc(sizeof(spasm_ZZp) * (size_t)cap) : nullptr;
    spasm_field_init(prime, T->field);
    return T;
}

SPASM_API void spasm_triplet_realloc(struct spasm_triplet *T, i64 nzmax)
{
    if (nzmax < 0) nzmax = T->nz;
    const i64 cap = nzmax > 0 ? nzmax : 1;
    T->i = (int *)realloc(T->i, sizeof(int) * (size_t)cap);
    T->j = (int *)realloc(T->j, sizeof(int) * (size_t)cap);
    if (T->x) T->x = (spasm_ZZp *)realloc(T->x, sizeof(spasm_ZZp) * (size_t)cap);
    T->nzmax = nzmax;
}

SPASM_API void spasm_triplet_free(struct spasm_triplet *T)
{
    if (!T) return;
    free(T->i); free(T->j); free(T->x); free(T);
}

// value reduced to its balanced representative (reference src/SpaSM.jl:955-958); zeros are not stored
SPASM_API void spasm_add_entry(struct spasm_triplet *T, int i, int j, i64 x)
{
    const i64 p = T->field->p;
    i64 v = x % p;
    if (v < 0) v += p;
    if (v > T->field->halfp) v -= p;
    if (T->x && v == 0) return;
    if (T->nz == T->nzmax) spasm_triplet_realloc(T, 2 * T->nzmax + 1);
    T->i[T->nz] = i;
    T->j[T->nz] = j;
    if (T->x) T->x[T->nz] = (spasm_ZZp)v;
    T->nz++;
    if (i + 1 > T->n) T->n = i + 1;
    if (j + 1 > T->m) T->m = j + 1;
}

SPASM_API void spasm_triplet_transpose(struct spasm_triplet *T)
{
    std::swap(T->i, T->j);
    std::swap(T->n, T->m);
}

// triplet -> CSR: stable counting sort by row; entries repeating a position are summed, zero sums dropped
SPASM_API struct spasm_csr *spasm_compress(const struct spasm_triplet *T)
{
    const int n = T->n, m = T->m;
    const i64 nz = T->nz, p = T->field->p;
    struct spasm_csr *A = spasm_csr_alloc(n, m, nz, p, T->x != nullptr);
    if (!A) return nullptr;
    std::vector<i64> w((size_t)n + 1, 0);
    for (i64 k = 0; k < nz; k++) w[(size_t)T->i[k] + 1]++;
    for (int i = 0; i < n; i++) w[(size_t)i + 1] += w[(size_t)i];
    std::vector<i64> start(w.begin(), w.end());
    for (i64 k = 0; k < nz; k++) {
        const i64 q = w[(size_t)T->i[k]]++;
        A->j[q] = T->j[k];
        if (A->x) A->x[q] = T->x[k];
    }
    // merge duplicates inside each row (rows of a triplet file are short; a mark array keeps it linear)
    std::vector<i64> mark((size_t)(m > 0 ? m : 1), -1);
    i64 out = 0;
    for (int i = 0; i < n; i++) {
        const i64 lo = start[(size_t)i], hi = start[(size_t)i + 1], row_out = out;
        for (i64 k = lo; k < hi; k++) {
            const int c = A->j[k];
            if (mark[(size_t)c] >= row_out) {
                if (A->x) {
                    i64 v = ((i64)A->x[mark[(size_t)c]] + (i64)A->x[k]) % p;
                    if (v > T->field->halfp) v -= p; else if (v < T->field->mhalfp) v += p;
                    A->x[mark[(size_t)c]] = (spasm_ZZp)v;
                }
            } else {
                mark[(size_t)c] = out;
                A->j[out] = c;
                if (A->x) A->x[out] = A->x[k];
                out++;
            }
        }
        if (A->x) { // drop entries that cancelled
            i64 keep = row_out;
            for (i64 k = row_out; k < out; k++) {
                if (A->x[k] != 0) { A->j[keep] = A->j[k]; A->x[keep] = A->x[k]; mark[(size_t)A->j[keep]] = keep; keep++; }
                else mark[(size_t)A->j[k]] = -1;
            }
            out = keep;
        }
        A->p[i] = row_out;
    }
    A->p[n] = out;
    return A;
}

namespace {
// SHA-256 (FIPS 180-4), for the optional hash of spasm_triplet_load
struct Sha256 {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t buf[64];
    size_t fill = 0;
    uint64_t total = 0;
    static uint32_t rotr(uint32_t x, int k) { return (x >> k) | (x << (32 - k)); }
    void block(const uint8_t *b)
    {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
            0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
            0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
            0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
            0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
            0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)b[4 * i] << 24 | (uint32_t)b[4 * i + 1] << 16 | (uint32_t)b[4 * i + 2] << 8 | b[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
            const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & bb) ^ (a & c) ^ (bb & c), t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const uint8_t *p, size_t n)
    {
        total += n;
        while (n) {
            const size_t k = std::min(n, 64 - fill);
            memcpy(buf + fill, p, k);
            fill += k; p += k; n -= k;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void final(uint8_t *out)
    {
        const uint64_t bits = total * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t len[8];
        for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(len, 8);
        for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
    }
};
} // namespace

// ---- rank certificates (reference src/SpaSM.jl:345-353, :928-933; include/spasm_amd.h) ---------------------------------------
void spasm_cert_challenge(const uint8_t *hash, i64 prime, int r, const int *ri, const int *cj, spasm_ZZp *x)
{
    // seed = SHA-256(hash || prime || r || rows || columns); block k of the stream = SHA-256(seed || k), cut into 32-bit words, each
    // masked to the bits of the prime and rejected when >= prime (uniform residues)
    uint8_t seed[32];
    {
        Sha256 h;
        h.update(hash, 32);
        const int64_t hdr[2] = {prime, (int64_t)r};
        h.update((const uint8_t *)hdr, sizeof hdr);
        if (r > 0) { h.update((const uint8_t *)ri, (size_t)r * sizeof(int)); h.update((const uint8_t *)cj, (size_t)r * sizeof(int)); }
        h.final(seed);
    }
    uint64_t mask = 1;
    while (mask <= (uint64_t)prime) mask <<= 1;
    mask -= 1;
    const ZpField F = zp_field_make(prime);
    uint64_t ctr = 0;
    int have = 0;
    uint8_t blk[32];
    for (int k = 0; k < r;) {
        if (have == 0) {
            Sha256 h;
            h.update(seed, 32);
            h.update((const uint8_t *)&ctr, sizeof ctr);
            h.final(blk);
            ctr++;
            have = 8;
        }
        const uint8_t *w = blk + 4 * (8 - have);
        have--;
        const uint64_t v = ((uint64_t)w[0] << 24 | (uint64_t)w[1] << 16 | (uint64_t)w[2] << 8 | (uint64_t)w[3]) & mask;
        if (v >= (uint64_t)prime) continue;
        x[k++] = zp_reduce(F, (int64_t)v);
    }
}

extern "C" SPASM_API bool spasm_certificate_rank_verify(const struct spasm_csr *A, const uint8_t *hash, const struct spasm_rank_certificate *proof)
{
    spasm_clear_error();
    if (!A || !hash || !proof) { spasm_set_error("spasm_certificate_rank_verify: null argument"); return false; }
    const int r = proof->r, n = A->n, m = A->m;
    if (r < 0 || r > n || r > m || proof->prime != A->field->p || memcmp(proof->hash, hash, 32) != 0) return false;
    if (r == 0) return true;
    if (!proof->i || !proof->j || !proof->x || !proof->y) return false;
    const ZpField F = zp_field_make(A->field->p);
    // the rows and the columns are distinct and in range
    std::vector<int> pos((size_t)m, -1);
    std::vector<char> seen((size_t)n, 0);
    for (int k = 0; k < r; k++) {
        const int i = proof->i[k], j = proof->j[k];
        if (i < 0 || i >= n || j < 0 || j >= m || seen[(size_t)i] || pos[(size_t)j] >= 0) return false;
        seen[(size_t)i] = 1;
        pos[(size_t)j] = k;
    }
    // the challenge is the one the certificate's own commitment yields
    std::vector<spasm_ZZp> x((size_t)r);
    spasm_cert_challenge(hash, proof->prime, r, proof->i, proof->j, x.data());
    for (int k = 0; k < r; k++) if (zp_reduce(F, (int64_t)proof->x[k]) != x[(size_t)k]) return false;
    // y * A[i, j] == x, exactly
    std::vector<spasm_ZZp> acc((size_t)r, 0);
    for (int k = 0; k < r; k++) {
        const int yk = zp_reduce(F, (int64_t)proof->y[k]);
        if (yk == 0) continue;
        const int i = proof->i[k];
        for (i64 t = A->p[i]; t < A->p[i + 1]; t++) {
            const int c = pos[(size_t)A->j[t]];
            if (c >= 0) acc[(size_t)c] = zp_axpy(F, yk, zp_reduce(F, (int64_t)A->x[t]), acc[(size_t)c]);
        }
    }
    for (int k = 0; k < r; k++) if (acc[(size_t)k] != x[(size_t)k]) return false;
    return true;
}

extern "C" SPASM_API void spasm_amd_certificate_challenge(const uint8_t *hash, i64 prime, int r, const int *i, const int *j, spasm_ZZp *x)
{
    if (hash && x && r >= 0 && (r == 0 || (i && j))) spasm_cert_challenge(hash, prime, r, i, j, x);
}

extern "C" SPASM_API void spasm_rank_certificate_free(struct spasm_rank_certificate *proof)
{
    if (!proof) return;
    free(proof->i); free(proof->j); free(proof->x); free(proof->y);
    free(proof);
}

extern "C" SPASM_API void spasm_rank_certificate_save(const struct spasm_rank_certificate *proof, void *file)
{
    FILE *f = (FILE *)file;
    if (!proof || !f) return;
    fprintf(f, "spasm-amd rank certificate v1\n%d %lld\n", proof->r, (long long)proof->prime);
    for (int k = 0; k < 32; k++) fprintf(f, "%02x", proof->hash[k]);
    fprintf(f, "\n");
    for (int k = 0; k < proof->r; k++) fprintf(f, "%d %d %d %d\n", proof->i[k], proof->j[k], proof->x[k], proof->y[k]);
}

extern "C" SPASM_API bool spasm_rank_certificate_load(void *file, struct spasm_rank_certificate *proof)
{
    FILE *f = (FILE *)file;
    if (!proof || !f) return false;
    char line[128];
    if (!fgets(line, sizeof line, f) || strncmp(line, "spasm-amd rank certificate v1", 29) != 0) return false;
    int r = 0;
    long long prime = 0;
    if (fscanf(f, "%d %lld", &r, &prime) != 2 || r < 0) return false;
    {
        // r comes from the file: every one of its r lines takes 8 bytes at least ("i j x y\n"), so a count the rest of the file
        // cannot hold is refused before anything is allocated for it
        const long at = ftell(f);
        if (at >= 0 && fseek(f, 0, SEEK_END) == 0) {
            const long end = ftell(f);
            if (fseek(f, at, SEEK_SET) != 0) return false;
            if (end >= at && (long long)r * 8 > (long long)(end - at) + 8) return false;
        }
    }
    char hex[80];
    if (fscanf(f, "%79s", hex) != 1 || strlen(hex) != 64) return false;
    for (int k = 0; k < 32; k++) {
        unsigned v = 0;
        if (sscanf(hex + 2 * k, "%2x", &v) != 1) return false;
        proof->hash[k] = (uint8_t)v;
    }
    int *ri = (int *)malloc(sizeof(int) * (size_t)std::max(r, 1)), *cj = (int *)malloc(sizeof(int) * (size_t)std::max(r, 1));
    spasm_ZZp *x = (spasm_ZZp *)malloc(sizeof(spasm_ZZp) * (size_t)std::max(r, 1)), *y = (spasm_ZZp *)malloc(sizeof(spasm_ZZp) * (size_t)std::max(r, 1));
    bool ok = ri && cj && x && y;
    for (int k = 0; ok && k < r; k++) ok = fscanf(f, "%d %d %d %d", &ri[k], &cj[k], &x[k], &y[k]) == 4;
    if (!ok) { free(ri); free(cj); free(x); free(y); return false; }
    proof->r = r;
    proof->prime = prime;
    proof->i = ri; proof->j = cj; proof->x = x; proof->y = y;
    return true;
}

// SMS reader: header "n m M" (the letter is skipped, reference src/SpaSM.jl:1070), then "i j v" 1-based until "0 0 0"
SPASM_API struct spasm_triplet *spasm_triplet_load(void *file, i64 prime, uint8_t *hash)
{
    spasm_clear_error();
    FILE *f = (FILE *)file;
    if (!f) { spasm_set_error("spasm_triplet_load: NULL file"); return nullptr; }
    std::string text;
    char chunk[1 << 16];
    size_t got;
    while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) text.append(chunk, got);
    if (hash) { Sha256 s; s.update((const uint8_t *)text.data(), text.size()); s.final(hash); }
    size_t pos = 0;
    auto read_int = [&](i64 &v) -> bool { // reference read_Int (:1044-1061): skip to the next digit, honour '-'
        bool neg = false;
        while (pos < text.size() && !(text[pos] >= '0' && text[pos] <= '9')) { if (text[pos] == '-') neg = !neg; pos++; }
        if (pos >= text.size()) return false;
        v = 0;
        while (pos < text.size() && text[pos] >= '0' && text[pos] <= '9') { v = 10 * v + (text[pos] - '0'); pos++; }
        if (neg) v = -v;
        return true;
    };
    i64 n, m;
    if (!read_int(n) || !read_int(m) || n < 0 || m < 0 || n > 0x7fffffff || m > 0x7fffffff) { spasm_set_error("spasm_triplet_load: bad SMS header"); return nullptr; }
    struct spasm_triplet *T = spasm_triplet_alloc((int)n, (int)m, 1024, prime, true);
    if (!T) return nullptr;
    for (;;) {
        i64 i, j, v;
        if (!read_int(i) || !read_int(j) || !read_int(v)) { spasm_set_error("spasm_triplet_load: missing \"0 0 0\" terminator"); spasm_triplet_free(T); return nullptr; }
        if (i == 0) break;
        if (i < 1 || j < 1 || i > n || j > m) { spasm_set_error("spasm_triplet_load: entry (%lld,%lld) outside %lld x %lld", (long long)i, (long long)j, (long long)n, (long long)m); spasm_triplet_free(T); return nullptr; }
        spasm_add_entry(T, (int)(i - 1), (int)(j - 1), v);
    }
    T->n = (int)n; // the header rules even when trailing rows / columns are empty
    T->m = (int)m;
    return T;
}

SPASM_API void spasm_triplet_save(const struct spasm_triplet *T, void *file)
{
    FILE *f = (FILE *)file;
    fprintf(f, "%d %d M\n", T->n, T->m);
    for (i64 k = 0; k < T->nz; k++) fprintf(f, "%d %d %d\n", T->i[k] + 1, T->j[k] + 1, T->x ? T->x[k] : 1);
    fprintf(f, "0 0 0\n");
}

SPASM_API void spasm_csr_save(const struct spasm_csr *A, void *file)
{
    FILE *f = (FILE *)file;
    fprintf(f, "%d %d M\n", A->n, A->m);
    for (int i = 0; i < A->n; i++)
        for (i64 k = A->p[i]; k < A->p[i + 1]; k++) fprintf(f, "%d %d %d\n", i + 1, A->j[k] + 1, A->x ? A->x[k] : 1);
    fprintf(f, "0 0 0\n");
}

// reference src/SpaSM.jl:619-620 (spasm_scatter.c): x += beta * A[i] on a dense vector of balanced residues.  Host-side, one row:
// the innermost loop of libspasm's Schur complement, kept as a callable for SpaSM.jl's `scatter` (the engine's own scatter is the
// device kernels of csrc/stream.hpp / kernels.hpp).
SPASM_API void spasm_scatter(const struct spasm_csr *A, int i, spasm_ZZp beta, spasm_ZZp *x)
{
    if (!A || !x || i < 0 || i >= A->n) return;
    const ZpField F = zp_field_make(A->field->p);
    for (i64 k = A->p[i]; k < A->p[i + 1]; k++) {
        const int j = A->j[k];
        x[j] = zp_axpy(F, beta, A->x[k], x[j]);
    }
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

} // namespace

// ------------------------------------------------------------------------------------------------
// spasm_factorization_verify (reference src/SpaSM.jl:934): see include/spasm_amd.h for what is and is not checked
// ------------------------------------------------------------------------------------------------
extern "C" SPASM_API bool spasm_factorization_verify(const struct spasm_csr *A, const struct spasm_lu *fact, uint64_t seed)
{
    if (!A || !fact || !fact->U || !fact->qinv) { spasm_set_error("spasm_factorization_verify: null argument"); return false; }
    const struct spasm_csr *U = fact->U;
    const int n = A->n, m = A->m, r = U->n;
    const uint64_t p = (uint64_t)A->field->p;
    if (U->m != m || fact->r != r || (uint64_t)U->field->p != p) return false;
    auto res = [p](int v) -> uint64_t { return v < 0 ? (uint64_t)((int64_t)v + (int64_t)p) : (uint64_t)v; };
    // (a) echelon shape: qinv is a bijection pivot column -> row, every row holds a 1 on its pivot column, and U is (permuted)
    // upper triangular: the relation "row a has an entry on the pivot column of row b" has no cycle.  Pivots need not be the
    // leftmost entries of their rows (the "FL on columns" search takes others).
    std::vector<int> pivcol((size_t)std::max(r, 1), -1);
    for (int j = 0; j < m; j++) {
        const int k = fact->qinv[j];
        if (k < -1 || k >= r) return false;
        if (k >= 0) {
            if (pivcol[(size_t)k] != -1) return false;
            pivcol[(size_t)k] = j;
        }
    }
    std::vector<int> indeg((size_t)std::max(r, 1), 0), order;
    order.reserve((size_t)r);
    for (int k = 0; k < r; k++) {
        if (pivcol[(size_t)k] < 0) return false;
        bool has_pivot = false;
        for (i64 q = U->p[k]; q < U->p[k + 1]; q++) {
            if (U->j[q] < 0 || U->j[q] >= m) return false;
            if (U->j[q] == pivcol[(size_t)k]) { if (res(U->x[q]) != 1 || has_pivot) return false; has_pivot = true; }
            else if (fact->qinv[U->j[q]] >= 0) indeg[(size_t)fact->qinv[U->j[q]]]++;
        }
        if (!has_pivot) return false;
    }
    for (int k = 0; k < r; k++) if (indeg[(size_t)k] == 0) order.push_back(k);
    for (size_t h = 0; h < order.size(); h++) {
        const int k = order[h];
        for (i64 q = U->p[k]; q < U->p[k + 1]; q++) {
            const int b = fact->qinv[U->j[q]];
            if (b >= 0 && b != k && --indeg[(size_t)b] == 0) order.push_back(b);
        }
    }
    if ((int)order.size() != r) return false; // a cycle: not an echelon form under any permutation
    // (b) random combinations of the rows of A must reduce to zero
    const int trials = p < 65536 ? 8 : 2;
    // (the trials are independent: one host thread each, every trial with a generator of its own derived from the seed)
    int failed = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : failed)
    for (int t = 0; t < trials; t++) {
        std::vector<uint64_t> y((size_t)std::max(m, 1), 0);
        Rng rng(seed ^ 0x5350415346564552ull ^ ((uint64_t)(t + 1) * 0x9E3779B97F4A7C15ull));
        bool bad = false;
        for (int i = 0; i < n && !bad; i++) {
            const uint64_t xi = rng.below(p);
            if (xi == 0) continue;
            for (i64 q = A->p[i]; q < A->p[i + 1]; q++) {
                const int c = A->j[q];
                if (c < 0 || c >= m) { bad = true; break; }
                y[(size_t)c] = (y[(size_t)c] + xi * res(A->x ? A->x[q] : 1) % p) % p; // xi, residue < 2^32: the product fits 64 bits
            }
        }
        // rows in topological order: row k touches, among pivot columns, only those of rows after it, so an eliminated column
        // stays zero
        if (!bad)
            for (int k : order) {
                const int j = pivcol[(size_t)k];
                if (y[(size_t)j] == 0) continue;
                const uint64_t c = y[(size_t)j];
                for (i64 q = U->p[k]; q < U->p[k + 1]; q++) {
                    const size_t col = (size_t)U->j[q];
                    y[col] = (y[col] + (p - c * res(U->x[q]) % p)) % p;
                }
            }
        for (int j = 0; j < m && !bad; j++) if (y[(size_t)j] != 0) bad = true;
        if (bad) failed = 1;
    }
    if (failed) return false;
    // (c), (d): with L the check is two-sided
    if (fact->L) {
        const struct spasm_csr *L = fact->L;
        if (L->n != n || L->m != r || !fact->p || (uint64_t)L->field->p != p) return false;
        // (d) the pivotal rows of L: entries only on columns <= their own, non-zero on it
        std::vector<char> seen((size_t)std::max(n, 1), 0);
        for (int k = 0; k < r; k++) {
            const int i = fact->p[k];
            if (i < 0 || i >= n || seen[(size_t)i]) return false;
            seen[(size_t)i] = 1;
            bool diag = false;
            for (i64 q = L->p[i]; q < L->p[i + 1]; q++) {
                if (L->j[q] < 0 || L->j[q] > k) return false;
                if (L->j[q] == k) { if (res(L->x[q]) == 0 || diag) return false; diag = true; }
            }
            if (!diag) return false;
        }
        // (c) x * L * U == x * A
        std::vector<uint64_t> yl((size_t)std::max(r, 1)), z((size_t)std::max(m, 1)), wv((size_t)std::max(m, 1));
        Rng rng2(seed ^ 0x4C55564552494659ull);
        for (int t = 0; t < trials; t++) {
            std::fill(yl.begin(), yl.end(), 0);
            std::fill(z.begin(), z.end(), 0);
            std::fill(wv.begin(), wv.end(), 0);
            for (int i = 0; i < n; i++) {
                const uint64_t xi = rng2.below(p);
                if (xi == 0) continue;
                for (i64 q = A->p[i]; q < A->p[i + 1]; q++) wv[(size_t)A->j[q]] = (wv[(size_t)A->j[q]] + xi * res(A->x ? A->x[q] : 1) % p) % p;
                for (i64 q = L->p[i]; q < L->p[i + 1]; q++) {
                    if (L->j[q] < 0 || L->j[q] >= r) return false;
                    yl[(size_t)L->j[q]] = (yl[(size_t)L->j[q]] + xi * res(L->x[q]) % p) % p;
                }
            }
            for (int k = 0; k < r; k++) {
                const uint64_t c = yl[(size_t)k];
                if (c == 0) continue;
                for (i64 q = U->p[k]; q < U->p[k + 1]; q++) z[(size_t)U->j[q]] = (z[(size_t)U->j[q]] + c * res(U->x[q]) % p) % p;
            }
            for (int j = 0; j < m; j++) if (z[(size_t)j] != wv[(size_t)j]) return false;
        }
    }
    return true;
}

namespace {
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

// kind 2: Macaulay-like.  `nbase` = max(2, n / 2500) base polynomials with 10..row_nnz terms (column pattern
// inside a window of m/8 columns, fixed coefficients); row i is base (i mod nbase) translated by a seeded offset, so most rows have
// distinct leading columns (many Faugere-Lachartre pivots) and the Schur complement collapses to a small dense tail.
static struct spasm_csr *synth_macaulay(int n, int m, int row_nnz, i64 prime, uint64_t seed)
{
    const int nbase = std::max(2, n / 2500);
    const int window = std::max(row_nnz + 1, m / 8);
    const int maxterms = std::max(10, row_nnz);
    std::vector<std::vector<int>> bcols((size_t)nbase), bvals((size_t)nbase);
    const i64 halfp = prime / 2;
    for (int b = 0; b < nbase; b++) {
        Rng rng(row_seed(seed ^ 0xBA5Eull, (uint64_t)b));
        const int terms = 10 + (int)rng.below((uint64_t)(maxterms - 10 + 1));
        std::vector<int> &c = bcols[(size_t)b];
        while ((int)c.size() < std::min(terms, window)) {
            const int x = (int)rng.below((uint64_t)window);
            if (std::find(c.begin(), c.end(), x) == c.end()) c.push_back(x);
        }
        for (size_t k = 0; k < c.size(); k++) {
            i64 v = 1 + (i64)rng.below((uint64_t)(prime - 1));
            if (v > halfp) v -= prime;
            bvals[(size_t)b].push_back((int)v);
        }
    }
    std::vector<i64> rp((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) rp[(size_t)i + 1] = rp[(size_t)i] + (i64)bcols[(size_t)(i % nbase)].size();
    struct spasm_csr *A = spasm_csr_alloc(n, m, rp[(size_t)n], prime, true);
    if (!A) return nullptr;
    memcpy(A->p, rp.data(), sizeof(i64) * ((size_t)n + 1));
    const int span = std::max(1, m - window);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
        Rng rng(row_seed(seed, (uint64_t)i));
        const int off = (int)rng.below((uint64_t)span);
        const std::vector<int> &c = bcols[(size_t)(i % nbase)];
        const std::vector<int> &v = bvals[(size_t)(i % nbase)];
        for (size_t k = 0; k < c.size(); k++) {
            A->j[A->p[i] + (i64)k] = c[k] + off;
            A->x[A->p[i] + (i64)k] = v[k];
        }
    }
    return A;
}

extern "C" SPASM_API struct spasm_csr *spasm_amd_synth_csr(int kind, int n, int m, double density, int row_nnz,
                                                            i64 prime, uint64_t seed)
{
    spasm_clear_error();
    if (n < 0 || m < 0 || prime <= 2 || prime > 0xfffffffbLL || kind < 0 || kind > 2) {
        spasm_set_error("spasm_amd_synth_csr: bad arguments");
        return nullptr;
    }
    if (kind == 2) return synth_macaulay(n, m, row_nnz, prime, seed);
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
