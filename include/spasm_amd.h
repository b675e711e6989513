/*
 * spasm_amd.h -- C ABI of the MI355X-native sparse GF(p) echelonization engine.
 *
 * This is the drop-in boundary: every `spasm_*` symbol below has the name, argument
 * order and struct layout that SpaSM.jl binds with `@ccall spasm_lib.<sym>` (reference
 * file `src/SpaSM.jl`; the line of each binding is cited next to the declaration).
 * Pointing `spasm_lib` at `libspasm_amd.so` re-routes the echelonize / kernel hot path
 * to the HIP engine without touching the Julia side.
 *
 * Conventions (all from the reference wrapper):
 *   - indices are 0-based on this side of the boundary      (src/SpaSM.jl:486,597,961)
 *   - values are balanced residues in [mhalfp, halfp], i32  (src/SpaSM.jl:79-88)
 *   - rows of a CSR need not be sorted by column            (src/SpaSM.jl:1017-1020)
 *   - returned objects are owned by the caller and released with spasm_csr_free /
 *     spasm_lu_free from an arbitrary thread                (src/SpaSM.jl:146-150,273-277)
 *
 * The `spasm_amd_*` symbols are engine extensions (device-resident handles for
 * benchmarking and multi-GPU sharding); the reference has no counterpart for them.
 */
#ifndef SPASM_AMD_H
#define SPASM_AMD_H

#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t i64;
typedef int32_t spasm_ZZp; /* balanced representative, src/SpaSM.jl:79-81 */

/* struct Field, src/SpaSM.jl:51-56 (32 bytes, embedded by value) */
struct spasm_field_struct {
    i64 p;
    i64 halfp;
    i64 mhalfp;
    double dinvp;
};
typedef struct spasm_field_struct spasm_field[1];

/* struct _CSR, src/SpaSM.jl:126-134 (72 bytes) */
struct spasm_csr {
    i64 nzmax;
    int n;          /* rows */
    int m;          /* columns */
    i64 *p;         /* n+1 row starts */
    int *j;         /* column indices */
    spasm_ZZp *x;   /* values (may be NULL when allocated with_values = false) */
    spasm_field field;
};

/* struct _Triplet, src/SpaSM.jl:234-243 (80 bytes) */
struct spasm_triplet {
    i64 nzmax;
    i64 nz;
    int n;
    int m;
    int *i;
    int *j;
    spasm_ZZp *x;
    spasm_field field;
};

/* struct _LU, src/SpaSM.jl:262-270 (48 bytes) */
struct spasm_lu {
    int r;                    /* rank */
    bool complete;            /* L present and complete */
    struct spasm_csr *L;      /* NULL unless opts->L */
    struct spasm_csr *U;      /* r x m, unit pivots, pivot not necessarily first in row */
    int *qinv;                /* m entries: row of U holding the pivot of column j, or -1 */
    int *p;                   /* >= m entries (Julia views it with length U->m, :300) */
    struct spasm_triplet *Ltmp;
};

/* struct EchelonizeOpts, src/SpaSM.jl:325-343 (64 bytes) */
struct echelonize_opts {
    bool enable_greedy_pivot_search;
    bool enable_tall_and_skinny;
    bool enable_dense;
    bool enable_GPLU;
    bool L;
    bool complete;
    double min_pivot_proportion;
    int max_round;
    double sparsity_threshold;
    int dense_block_size;     /* Julia declares Int (8 bytes) at :339; low 32 bits are read */
    double low_rank_ratio;
    double tall_and_skinny_ratio;
    double low_rank_start_weight;
};

/* The layouts SpaSM.jl's mirrors imply (x86-64, little-endian), checked where the header is compiled: a field that
 * moves is a build error here, not a wrong pointer in Julia. */
#if defined(__cplusplus)
#define SPASM_LAYOUT_ASSERT(cond, msg) static_assert(cond, msg)
#else
#define SPASM_LAYOUT_ASSERT(cond, msg) _Static_assert(cond, msg)
#endif
SPASM_LAYOUT_ASSERT(sizeof(struct spasm_field_struct) == 32 && offsetof(struct spasm_field_struct, p) == 0 && offsetof(struct spasm_field_struct, halfp) == 8 &&
                        offsetof(struct spasm_field_struct, mhalfp) == 16 && offsetof(struct spasm_field_struct, dinvp) == 24,
                    "spasm_field: src/SpaSM.jl:51-56");
SPASM_LAYOUT_ASSERT(sizeof(struct spasm_csr) == 72 && offsetof(struct spasm_csr, nzmax) == 0 && offsetof(struct spasm_csr, n) == 8 && offsetof(struct spasm_csr, m) == 12 &&
                        offsetof(struct spasm_csr, p) == 16 && offsetof(struct spasm_csr, j) == 24 && offsetof(struct spasm_csr, x) == 32 &&
                        offsetof(struct spasm_csr, field) == 40,
                    "spasm_csr: src/SpaSM.jl:126-134");
SPASM_LAYOUT_ASSERT(sizeof(struct spasm_triplet) == 80 && offsetof(struct spasm_triplet, nz) == 8 && offsetof(struct spasm_triplet, n) == 16 &&
                        offsetof(struct spasm_triplet, m) == 20 && offsetof(struct spasm_triplet, i) == 24 && offsetof(struct spasm_triplet, j) == 32 &&
                        offsetof(struct spasm_triplet, x) == 40 && offsetof(struct spasm_triplet, field) == 48,
                    "spasm_triplet: src/SpaSM.jl:234-243");
SPASM_LAYOUT_ASSERT(sizeof(struct spasm_lu) == 48 && offsetof(struct spasm_lu, r) == 0 && offsetof(struct spasm_lu, complete) == 4 && offsetof(struct spasm_lu, L) == 8 &&
                        offsetof(struct spasm_lu, U) == 16 && offsetof(struct spasm_lu, qinv) == 24 && offsetof(struct spasm_lu, p) == 32 &&
                        offsetof(struct spasm_lu, Ltmp) == 40,
                    "spasm_lu: src/SpaSM.jl:262-270");
SPASM_LAYOUT_ASSERT(sizeof(struct echelonize_opts) == 64 && offsetof(struct echelonize_opts, enable_greedy_pivot_search) == 0 &&
                        offsetof(struct echelonize_opts, enable_tall_and_skinny) == 1 && offsetof(struct echelonize_opts, enable_dense) == 2 &&
                        offsetof(struct echelonize_opts, enable_GPLU) == 3 && offsetof(struct echelonize_opts, L) == 4 && offsetof(struct echelonize_opts, complete) == 5 &&
                        offsetof(struct echelonize_opts, min_pivot_proportion) == 8 && offsetof(struct echelonize_opts, max_round) == 16 &&
                        offsetof(struct echelonize_opts, sparsity_threshold) == 24 && offsetof(struct echelonize_opts, dense_block_size) == 32 &&
                        offsetof(struct echelonize_opts, low_rank_ratio) == 40 && offsetof(struct echelonize_opts, tall_and_skinny_ratio) == 48 &&
                        offsetof(struct echelonize_opts, low_rank_start_weight) == 56,
                    "echelonize_opts: src/SpaSM.jl:325-343");

/* ---- data symbol: SpaSM.log() stores a C callback here, src/SpaSM.jl:34-46 ---- */
extern int (*logcallback)(char *);

/* ---- spasm_util.c surface ---- */
double spasm_wtime(void);                                            /* src/SpaSM.jl:430 */
i64 spasm_nnz(const struct spasm_csr *A);                            /* src/SpaSM.jl:432 */
struct spasm_csr *spasm_csr_alloc(int n, int m, i64 nzmax, i64 prime, bool with_values); /* :441 */
void spasm_csr_realloc(struct spasm_csr *A, i64 nzmax);              /* src/SpaSM.jl:447 */
void spasm_csr_resize(struct spasm_csr *A, int n, int m);            /* src/SpaSM.jl:449 */
void spasm_csr_free(struct spasm_csr *A);                            /* src/SpaSM.jl:451 */
void spasm_lu_free(struct spasm_lu *N);                              /* src/SpaSM.jl:463 */
int spasm_get_num_threads(void);                                     /* src/SpaSM.jl:470 */
int spasm_get_thread_num(void);                                      /* src/SpaSM.jl:475 */

/* ---- spasm_triplet.c / spasm_io.c surface: the SMS wire format ("n m M" header, 1-based "i j v" lines,
 * "0 0 0" terminator; reference src/SpaSM.jl:1029-1042, :1063-1086).  Host-side I/O, no device work. ---- */
struct spasm_triplet *spasm_triplet_alloc(int n, int m, i64 nzmax, i64 prime, bool with_values); /* src/SpaSM.jl:453 */
void spasm_triplet_realloc(struct spasm_triplet *T, i64 nzmax);      /* src/SpaSM.jl:455 */
void spasm_triplet_free(struct spasm_triplet *T);                    /* src/SpaSM.jl:457 */
void spasm_add_entry(struct spasm_triplet *T, int i, int j, i64 x);  /* src/SpaSM.jl:486 */
void spasm_triplet_transpose(struct spasm_triplet *T);               /* src/SpaSM.jl:491 */
struct spasm_csr *spasm_compress(const struct spasm_triplet *T);     /* src/SpaSM.jl:493 */
struct spasm_triplet *spasm_triplet_load(void *file, i64 prime, uint8_t *hash); /* FILE*, src/SpaSM.jl:501; hash = SHA-256 of the stream or NULL */
void spasm_triplet_save(const struct spasm_triplet *T, void *file);  /* FILE*, src/SpaSM.jl:514 */
void spasm_csr_save(const struct spasm_csr *A, void *file);          /* FILE*, src/SpaSM.jl:523 */

/* ---- spasm_certificate.c: the probabilistic self-check of a factorization (src/SpaSM.jl:934).  Host-side, O(nnz).
 * Checks (a) the shape of the echelon form: every row of U holds a 1 on its pivot column qinv^-1(k) and U is (permuted)
 * triangular, so its rows are independent; (b) for random vectors x drawn from `seed`, that x*A reduces to zero modulo the rows
 * of U, i.e. that the row space of A lies in that of U (a wrong U escapes with probability <= 1/p per trial; 2 trials, 8 for
 * p < 2^16).  Together: rank(A) <= r and U spans at least the rows of A.
 * With fact->L (echelonize_opts.L) the check is two-sided, as libspasm's: (c) x*L*U == x*A for random x, i.e. A == L*U, and
 * (d) the rows p[0 .. r) of L form a triangular matrix with a non-zero diagonal, so U = L_P^-1 A_P: every row of U lies in the
 * row space of A and rank(A) == r.  Without L, (c) and (d) are skipped: a U with rows outside the row space of A passes. ---- */
bool spasm_factorization_verify(const struct spasm_csr *A, const struct spasm_lu *fact, uint64_t seed);

/* ---- spasm_certificate.c: rank certificates (src/SpaSM.jl:345-353, :928-933).  A certificate proves rank(A) >= r to somebody who
 * only has A: r rows i[] and r columns j[] whose r x r submatrix C is non-singular, shown by a vector y with y * C == x for a
 * challenge x the prover cannot choose -- x is drawn from SHA-256(hash, prime, r, i, j) (Fiat-Shamir; `hash` is the 32-byte digest
 * of the matrix file as spasm_triplet_load computes it): if C were singular a random x would lie in its row space with probability
 * <= 1/p -- PER CHALLENGE.  The struct SpaSM.jl mirrors holds one (x, y) pair, so that is all one certificate carries: a prover who
 * may retry (other rows, other columns, another order -- each choice hashes to a new x) gets a singular C accepted after about p
 * attempts, which is cheap for p = 127 or 65521.  The certificate is therefore evidence against ERRORS (a wrong rank out of a faulty
 * run), not against an adversarial prover with a small prime; that needs k challenges with p^k >= 2^64, i.e. another struct.
 * Verification is host-side and O(nnz(A)); together with spasm_factorization_verify (rank(A) <= r) it pins the rank.
 * Creation takes the pivotal rows and the pivot columns of `fact` and solves y * C == x on the device (C is echelonized with L,
 * then spasm_solve).  libspasm's on-disk format is not in the reference tree: save/load use a text format of their own
 * ("spasm-amd rank certificate v1", then r, prime, the hash in hex and r lines "i j x y"). ---- */
struct spasm_rank_certificate {      /* src/SpaSM.jl:345-353 */
    int r;
    i64 prime;
    uint8_t hash[32];
    int *i;                          /* r rows of A */
    int *j;                          /* r columns of A */
    spasm_ZZp *x;                    /* the challenge */
    spasm_ZZp *y;                    /* the response: y * A[i, j] == x */
};
struct spasm_rank_certificate *spasm_certificate_rank_create(const struct spasm_csr *A, const uint8_t *hash, const struct spasm_lu *fact); /* :928 */
bool spasm_certificate_rank_verify(const struct spasm_csr *A, const uint8_t *hash, const struct spasm_rank_certificate *proof);           /* :930 */
void spasm_rank_certificate_save(const struct spasm_rank_certificate *proof, void *file);  /* FILE*, :931 */
bool spasm_rank_certificate_load(void *file, struct spasm_rank_certificate *proof);        /* FILE*, :933; fills a caller-owned struct */
void spasm_rank_certificate_free(struct spasm_rank_certificate *proof);                    /* engine extension: the arrays and the struct */
/* the challenge a certificate with these rows and columns must answer (r balanced residues); engine extension, for verifiers and tests */
void spasm_amd_certificate_challenge(const uint8_t *hash, i64 prime, int r, const int *i, const int *j, spasm_ZZp *x);

/* ---- spasm_solve.c (src/SpaSM.jl:895-923), for a factorization that carries L (echelonize with opts->L; the sparse rounds then
 * keep their multiplier lists and the dense finish is not used).  L is n x r with A[i] == sum_k L[i][k] U[k]; the entry of row
 * p[k] on column k is the pivot U's row k was divided by.
 * spasm_gesv: X (rows of B x rows of A, zero outside the pivotal rows) with X*A == B; ok[k] tells whether row k of B is in the
 *             row space.  Two batched triangular solves on the device (Y*U == B, then X_P*L_P == Y).
 * spasm_solve: the same for one dense vector b (m entries) -> x (n = rows of A entries); false when there is no solution.
 *             NOTE the prototype wrapper allocates x with fact.U.n entries (src/SpaSM.jl:898): bind it with size(fact.L, 1). ---- */
struct spasm_csr *spasm_gesv(const struct spasm_lu *fact, const struct spasm_csr *B, bool *ok);
bool spasm_solve(const struct spasm_lu *fact, const spasm_ZZp *b, spasm_ZZp *x);

/* ---- spasm_ZZp.c surface (commented-out binding at src/SpaSM.jl:65; arithmetic restated :73-88,:383-390) ---- */
void spasm_field_init(i64 p, spasm_field F);

/* ---- spasm_scatter.c / spasm_triangular.c as SpaSM.jl binds them ---- */
void spasm_scatter(const struct spasm_csr *A, int i, spasm_ZZp beta, spasm_ZZp *x);   /* src/SpaSM.jl:620: x += beta * A[i] (host) */
/* src/SpaSM.jl:721, semantics :694-713: solve x * U = B[k]; x (m entries, need not be initialised) receives the solution
 * scattered over the columns, xj[top .. m) its pattern (xj has 3 m entries, zero on entry; only xj[top .. m) is written);
 * with x_b on the pivot columns (qinv[j] >= 0) and x_a on the others, x_b * U + x_a == B[k].  Pivots of U must be 1; they need
 * not be the first entries of their rows.  Returns top (-1 on failure).  One row through the batched device solve
 * (spasm_amd_triangular_solve does all rows of B in one pass). */
int spasm_sparse_triangular_solve(const struct spasm_csr *U, const struct spasm_csr *B, int k, int *xj, spasm_ZZp *x, const int *qinv);

/* ---- spasm_transpose.c ---- */
struct spasm_csr *spasm_transpose(const struct spasm_csr *A);        /* src/SpaSM.jl:589 (one-argument form) */

/* ---- spasm_echelonize.c / spasm_kernel.c : THE hot path ---- */
void spasm_echelonize_init_opts(struct echelonize_opts *opts);       /* src/SpaSM.jl:817 */
struct spasm_lu *spasm_echelonize(const struct spasm_csr *A, struct echelonize_opts *opts); /* :863 */
struct spasm_csr *spasm_kernel(const struct spasm_lu *fact);         /* src/SpaSM.jl:879 */
/* spasm_rref.c (src/SpaSM.jl:871): the reduced row echelon form of fact->U (r x m; row k = row k of U with its entries on the
 * other pivot columns eliminated, pivot 1 first); Rqinv (m entries, may be NULL) receives the pivot column -> row map.
 * Pivots must be the leftmost entries of their rows (this engine's LUs). */
struct spasm_csr *spasm_rref(const struct spasm_lu *fact, int *Rqinv);

/* ====================================================================================
 * Engine extensions (no reference counterpart): device-resident handles.
 * ==================================================================================== */

/* Per-round record written by the engine (what libspasm prints per round, README.md:19-38). */
struct spasm_amd_round_stats {
    int round;
    int rows_in;          /* rows of the matrix entering the round */
    i64 nnz_in;
    int npiv;             /* structural pivots elected this round */
    int rows_out;         /* non-empty rows of the Schur complement */
    i64 nnz_out;
    i64 nnz_reduced;      /* reference scatter trip count: sum nnz(A_i) + sum over applications nnz(U_r) */
    i64 applications;     /* (row, pivot-row) eliminations performed */
    i64 read_bytes;       /* algorithmic read bytes: 8*nnz_reduced + 16*segments + 4*m */
    double ms_pivots;     /* device time, pivot election + U build */
    double ms_solve;      /* device time, triangular-solve kernel (multipliers) */
    double ms_scatter;    /* device time, scatter/accumulate kernel (Schur rows) */
    double ms_total;
    /* scatter kernels, per size class (one launch each): device time, rows, entries streamed (the row's own
     * entries + the non-pivot part of every applied pivot row) and row segments visited.  Classes 0..6: LDS hash
     * tables of 256..16384 slots, 7: the global-memory last resort, 8..14: the streaming twins of 0..6 (rows whose
     * entries are written out directly; a row with too many duplicate columns is redone by its hash class) */
    double ms_class[16];
    int rows_class[16];
    i64 ent_class[16];
    i64 seg_class[16];
    i64 stream_fix;       /* duplicate columns the streaming kernels merged after the fact */
    i64 stream_redo;      /* rows the streaming kernels handed back to the hash-table kernels */
    /* per-round setup, wall time with its host synchronisations: ms_uinv = Uinv = (I + U_PP)^-1 (rounds that keep to the
     * multiplier lists; 0 when the round goes along W); ms_w = the levels of the pivot graph + sizing + the first build of W
     * (included in ms_pivots of an echelonize round; a plan pays them when it is created, and rebuilds W in every run: ms_wbuild) */
    double ms_uinv;
    double ms_w;
    i64 npiv_open;        /* of npiv: pivots the "Faugere-Lachartre on columns" search added to the leftmost-entry ones
                           * (echelonize rounds with enable_greedy_pivot_search; 0 for plans, which keep to leftmost entries) */
    /* W = -(I + U_PP)^-1 U_PN as the Schur step builds it, level by level of the pivot graph (csrc/wlevel.hpp): device time of
     * the build inside the last plan run (it is part of every run), levels, entries of W, rows left to the workgroup kernel */
    double ms_wbuild;
    i64 w_levels;
    i64 w_entries;
    i64 w_long_rows;
    i64 npiv_greedy;      /* of npiv: pivots the greedy cycle-free search added (reference README.md:23; csrc/greedy.hpp) */
    /* the fused Schur kernel (csrc/fused.hpp: plan + stream of a row in one kernel; round 4): device time of its launch and of the
     * fix-up launch behind it (plans with class timing on; 0 otherwise), rows it took, entries it streamed, row segments it visited
     * (1 per row + 1 per run of W), rows it left to the general path (whose launches are the classes above), entries of S its waves
     * took from the cursor */
    double ms_fused;
    double ms_fused_fix;
    i64 rows_fused;
    i64 ent_fused;
    i64 seg_fused;
    i64 rows_rejected;
    i64 s_entries_used;
    double ms_levels;     /* the levels of the pivot graph (relaxation + sort), once per round: part of ms_w */
    double ms_w_sizing;   /* of ms_w: the device-memory query and the allocation of the buffers of the W build (wall time) */
};

typedef struct spasm_amd_schur_plan spasm_amd_schur_plan;

/* Last error text of the calling thread ("" if none). Engine calls return NULL / nonzero on failure. */
const char *spasm_amd_last_error(void);

/* Number of HIP devices visible; <= 0 when there is none (the hot path then fails loudly). */
int spasm_amd_device_count(void);
int spasm_amd_set_device(int dev);

/* Deterministic synthetic CSR (host memory, caller frees with spasm_csr_free):
 *   kind 0: every entry present with probability `density` (BASELINE config 2)
 *   kind 1: exactly `row_nnz` distinct uniform columns per row (BASELINE configs 3/4)
 *   kind 2: Macaulay-like, rows are translates of n/2500 base patterns of 10..row_nnz terms (BASELINE config 5)
 * values uniform on the nonzero balanced residues; columns unsorted. */
struct spasm_csr *spasm_amd_synth_csr(int kind, int n, int m, double density, int row_nnz,
                                      i64 prime, uint64_t seed);

/* Build a device-resident plan for ONE Schur round of A (BASELINE config 3):
 * uploads rows [row_lo,row_hi) of A as this device's shard, elects the Faugere-Lachartre
 * pivots of the WHOLE matrix (so that every shard sees the same U), builds U on the device.
 * Returns NULL on failure. */
spasm_amd_schur_plan *spasm_amd_schur_plan_create(const struct spasm_csr *A, int row_lo, int row_hi);
/* Same with the plan's rows taken as row_lo, row_lo + stride, ... < row_hi.  Strided shards (rank r of G: row_lo = r,
 * stride = G) are balanced; contiguous blocks are not when the election's tie-break puts the pivots in the first rows. */
spasm_amd_schur_plan *spasm_amd_schur_plan_create_strided(const struct spasm_csr *A, int row_lo, int row_hi, int stride);
/* Run the round once on `stream` (a hipStream_t, NULL = default): solve + scatter kernels.
 * Returns 0 on success. Safe to call repeatedly (outputs are overwritten). */
int spasm_amd_schur_plan_run(spasm_amd_schur_plan *plan, void *stream);
/* Per-class event pairs around the scatter launches (stats->ms_class) are recorded when `on` (default); they cost a
 * few microseconds of launch gap each, so throughput runs switch them off and profile one extra run with them on. */
void spasm_amd_schur_plan_class_timing(spasm_amd_schur_plan *plan, int on);
/* Block until the plan's last run has finished and fill `stats` (counters + event timings). */
int spasm_amd_schur_plan_stats(spasm_amd_schur_plan *plan, struct spasm_amd_round_stats *stats);
/* Copy the Schur complement of the last run back to host CSR (n_shard_nonpivot x m). p_out (may be
 * NULL) receives, per output row, the index of the originating row of A. */
struct spasm_csr *spasm_amd_schur_plan_fetch(spasm_amd_schur_plan *plan, int *p_out);
/* The pivot rows of the plan's round as they enter U (scaled to a unit pivot, reference src/SpaSM.jl:712), in pivot-index order
 * (= ascending pivot column for a plan); pivcol_out / row_out (npiv ints each, may be NULL) receive the pivot column and the originating
 * row of A of each.  What a row-sharded echelonization appends to U after every exchange, without rebuilding it on the host. */
struct spasm_csr *spasm_amd_schur_plan_fetch_U(spasm_amd_schur_plan *plan, int *pivcol_out, int *row_out);
void spasm_amd_schur_plan_free(spasm_amd_schur_plan *plan);

/* ---- row-sharded round with an exchange of the pivot rows (one process per GPU; the collectives are the
 * caller's: torch.distributed / RCCL all-reduce(MIN) on the keys, all-gather on the exported rows) ----
 * All `*_dev` arguments are DEVICE pointers owned by the caller.
 *   1. shard_create   uploads rows [row_lo,row_hi) of A only
 *   2. shard_elect    writes this shard's election keys, one i64 per column: (row length << 32 | global row),
 *                     INT64_MAX where the shard proposes nothing            -> caller: all-reduce(MIN)
 *   3. shard_set_keys takes the reduced keys, numbers the pivots; returns npiv (< 0 on error) and reports
 *                     how many pivot rows / entries this shard owns         -> caller: all-gather of the counts
 *   4. shard_export   packs the owned pivot rows: hdr_dev[2*k] = pivot index, hdr_dev[2*k+1] = length;
 *                     ent_dev = their {col,val} pairs back to back          -> caller: all-gather (variable length)
 *   5. shard_import   takes the concatenation over ranks (any order of parts), builds U; then plan_run /
 *                     plan_stats / plan_fetch work on the returned plan for this shard's non-pivot rows. */
typedef struct spasm_amd_shard spasm_amd_shard;
spasm_amd_shard *spasm_amd_shard_create(const struct spasm_csr *A, int row_lo, int row_hi);
spasm_amd_shard *spasm_amd_shard_create_strided(const struct spasm_csr *A, int row_lo, int row_hi, int stride);
int spasm_amd_shard_elect(spasm_amd_shard *sh, int64_t *keys_dev);
int spasm_amd_shard_set_keys(spasm_amd_shard *sh, const int64_t *keys_dev, int *n_owned, i64 *nnz_owned);
/* r04: shard_set_keys in two halves with "Faugere-Lachartre on columns" (enable_greedy_pivot_search, src/SpaSM.jl:326; README.md:23)
 * between them -- the search of the single-device round over ROW SHARDS: shard_assign numbers the leftmost pivots (returns how many);
 * shard_open_step runs one step of the search on this shard's rows: `in_dev` is the array the step before left, REDUCED over the
 * shards by the caller, `out_dev` receives the array this step leaves (m elements) for the caller to reduce --
 *   step 0 BEGIN -> closed (int32, MAX) | 1 HIST (in: closed) -> colcnt (int32, SUM) | 2 PROPOSE (in: colcnt) -> best2 (int64, MIN)
 *   3 ACCEPT (in: best2) -> newflag (int32, MAX) | 4 RECORD (in: newflag; pass = 1..4) -> closed (int32, MAX), returns the pivots the
 *   pass accepted (0: the search is over) | 5 FINISH: renumbers, returns the pivots the search added
 * (steps 1-4 once per pass, at most four passes: four m-word reductions per pass); shard_finish_keys then does the rest of
 * shard_set_keys (the shard's non-pivot rows, the pivot rows it owns).  < 0 on error. */
int spasm_amd_shard_assign(spasm_amd_shard *sh, const int64_t *keys_dev);
int spasm_amd_shard_open_step(spasm_amd_shard *sh, int step, int pass, const void *in_dev, void *out_dev);
int spasm_amd_shard_finish_keys(spasm_amd_shard *sh, int *n_owned, i64 *nnz_owned);
int spasm_amd_shard_export(spasm_amd_shard *sh, int *hdr_dev, int *ent_dev);
spasm_amd_schur_plan *spasm_amd_shard_import(spasm_amd_shard *sh, int n_rows, i64 n_entries, const int *hdr_dev, const int *ent_dev);
/* X * U = B for every row of B in one device pass: what SpaSM.jl's sparse_triangular_solve(U, B, qinv) / `B / LU`
 * (src/SpaSM.jl:733-755) obtains by looping spasm_sparse_triangular_solve over the rows of B.  Semantics of :694-713: with x_b on
 * the pivot columns and x_a on the others, x_b * U + x_a == B[k]; X (rows of B x rows of U) holds x_b indexed by the row of U;
 * ok[k] = 1 when x_a is empty, i.e. X[k] * U == B[k].  U needs unit pivots that are the leftmost entries of their rows. */
struct spasm_csr *spasm_amd_triangular_solve(const struct spasm_csr *U, const int *qinv, const struct spasm_csr *B, unsigned char *ok);
/* The kernel step of a multi-GPU run: the kernel vectors of the free columns number first, first + step, ... only (free
 * columns counted in ascending order; vector f of spasm_kernel(fact) is row (f - first) / step here). */
struct spasm_csr *spasm_amd_kernel_strided(const struct spasm_lu *fact, int first, int step);
/* Multi-round sharded echelonization (spasm.jl_amd/sharded.py: echelonize_sharded): the Schur rows of a sharded plan become the
 * shard's matrix of the next round, on the device, under the same numbering (local row i = original row lo + i * stride; this
 * round's pivot rows and empty rows are empty rows).  Runs the plan if it has not run; CONSUMES the plan (also on failure the
 * caller must not use it again); rows_out / nnz_out: non-empty rows and entries of the new matrix. */
spasm_amd_shard *spasm_amd_schur_plan_advance(spasm_amd_schur_plan *plan, int *rows_out, i64 *nnz_out);
/* the shard's current rows as a host CSR with one row per local row (empty ones included); free with spasm_csr_free */
struct spasm_csr *spasm_amd_shard_fetch(spasm_amd_shard *sh);
void spasm_amd_shard_free(spasm_amd_shard *sh);

/* spasm_echelonize over several devices of THIS process: `nshards` row shards (shard s: rows s, s + nshards, ...; device s modulo
 * the number of visible devices, so nshards = 8 on an 8-GPU node puts one shard on each), per round an election (per-column minimum
 * of the shards' keys), an exchange of the elected pivot rows (peer copies over xGMI) and the local Schur complement of every shard's
 * rows.  A remainder that is dense (or a round whose Schur complement is estimated dense: spasm_schur_estimate_density /
 * spasm_schur_dense, reference src/SpaSM.jl:763-766) is finished by ALL shards together for primes below 2^16 (csrc/dense_multi.hpp:
 * the rows stay where they are, per panel of 64 columns the candidates' panel entries go to shard 0 and the elected pivot rows to every
 * shard); a small or sparse remainder, and larger primes, by the single-device engine on device 0.  Leftmost-entry pivots throughout, so rank, pivot columns and kernel
 * equal those of spasm_echelonize with enable_greedy_pivot_search = 0 whatever nshards is.  What spasm.jl_amd/sharded.py does with
 * one process per GPU and RCCL, behind one call for hosts without torch.distributed (the Julia side: one more @ccall). */
struct spasm_lu *spasm_amd_echelonize_multi(const struct spasm_csr *A, struct echelonize_opts *opts, int nshards);
/* ---- the dense finish over row shards with ONE PROCESS PER SHARD (spasm.jl_amd/sharded.py; csrc/dense_multi.hpp states the
 * protocol).  The steps are the engine's, the exchanges between them the caller's collectives; `*_dev` are DEVICE pointers of the caller.
 *   shard_import_U      as spasm_amd_shard_import, but stops after U is built (no W / Uinv, no dry run)
 *   schur_plan_prepare  .. which this catches up on when the round stays sparse after all
 *   dshard_open         the columns this shard's Schur rows of the round can touch, as flags          -> dshard_flags -> all-reduce(MAX)
 *   dshard_density      takes the reduced flags (all shards then number the columns alike), estimates the density of the round's
 *                       Schur complement on 64 columns (spasm_schur_estimate_density); C_out = columns of the dense matrix
 *   dshard_build        this shard's Schur rows straight into its dense matrix (spasm_schur_dense), guest rows behind them
 *   per block of KB columns: dshard_block_begin; per panel of 64 (q-th of the block, columns c0 .. c0 + w):
 *     dshard_candidates   the rows this shard's own elimination of the panel elects, cand_bytes bytes   -> all-gather
 *     dshard_elect        the panel's pivots among the candidates of all shards (every rank runs it: same result); npp pivots,
 *                         cnt[k] / first[k]: how many shard k owns and the guest slot of its first
 *     dshard_pack         this shard's winners: cnt rows of (ldc - c0) elements, then nd planes of cnt x KB bytes -> one broadcast per owner
 *     dshard_unpack       an owner's buffer into the guest rows (also the owner's own)
 *     dshard_apply        the panel: elimination by the now known pivots, triangular solve of the pivot rows, update inside the block
 *   dshard_block_end(b0, b1, panels); after the last block dshard_finish = pivots found by all shards together (< 0: error), and
 *   dshard_fetch_U = the rows of U this shard owns (pivcol_out / row_out: C ints each, n_out of them used). */
typedef struct spasm_amd_dshard spasm_amd_dshard;
spasm_amd_schur_plan *spasm_amd_shard_import_U(spasm_amd_shard *sh, int n_rows, i64 n_entries, const int *hdr_dev, const int *ent_dev);
int spasm_amd_schur_plan_prepare(spasm_amd_schur_plan *plan);
spasm_amd_dshard *spasm_amd_dshard_open(spasm_amd_schur_plan *plan, int me, int nshards);
/* r04: the same finish over the shard's CURRENT rows (a remainder that is dense already; no round, no U): _flags / _density (returns 1.0)
 * / _build and the panel steps as after spasm_amd_dshard_open.  The shard stays the caller's. */
spasm_amd_dshard *spasm_amd_dshard_open_rows(spasm_amd_shard *sh, int me, int nshards);
int spasm_amd_dshard_flags(spasm_amd_dshard *ds, int *flags_dev);
double spasm_amd_dshard_density(spasm_amd_dshard *ds, const int *flags_dev, int free_cols, int *C_out);
int spasm_amd_dshard_build(spasm_amd_dshard *ds);
int spasm_amd_dshard_info(spasm_amd_dshard *ds, int *C_out, int *KB, i64 *ldc, int *elem, int *nd, int *cand_bytes);
int spasm_amd_dshard_block_begin(spasm_amd_dshard *ds);
int spasm_amd_dshard_candidates(spasm_amd_dshard *ds, int c0, int w, void *cand_dev);
int spasm_amd_dshard_elect(spasm_amd_dshard *ds, const void *stack_dev, int w, int *npp, int *cnt, int *first);
i64 spasm_amd_dshard_pack(spasm_amd_dshard *ds, int c0, void *buf_dev);
int spasm_amd_dshard_unpack(spasm_amd_dshard *ds, int q, int c0, int owner, const void *buf_dev);
int spasm_amd_dshard_apply(spasm_amd_dshard *ds, int q, int c0, int w, int b1);
int spasm_amd_dshard_block_end(spasm_amd_dshard *ds, int b0, int b1, int npan);
int spasm_amd_dshard_finish(spasm_amd_dshard *ds);
struct spasm_csr *spasm_amd_dshard_fetch_U(spasm_amd_dshard *ds, int *pivcol_out, int *row_out, int *n_out);
void spasm_amd_dshard_close(spasm_amd_dshard *ds);

/* How the most recent spasm_amd_echelonize_multi of this thread finished: 0 = remainder gathered to device 0 (or nothing left),
 * 1 = a round's Schur complement straight to dense on all shards, 2 = the dense remainder on all shards. */
int spasm_amd_multi_last_finish(void);

/* rank(A) = rank(echelonize(A)) (reference src/SpaSM.jl:1149) without materialising U on the host: the pivot rows are counted where
 * they are found.  For matrices whose U outgrows the host (BASELINE config 5: 10^10 entries and more above 1/3 scale).  Same
 * options as spasm_echelonize (L excluded); -1 on error. */
i64 spasm_amd_rank(const struct spasm_csr *A, struct echelonize_opts *opts);

/* Per-round records of the most recent spasm_echelonize call on this thread. */
int spasm_amd_last_rounds(struct spasm_amd_round_stats *out, int max_rounds);

/* The device's field arithmetic (csrc/zp.hpp, the restatement of spasm_ZZp.c as SpaSM.jl gives it, src/SpaSM.jl:383-390) on n
 * test vectors: a, b, c are balanced residues; out receives 8 ints per vector: a*b, a*b+c, a+b, a-b, -a, a^-1 (0 for a = 0),
 * a*b through the scatter kernels' lazy product + short reduction, 64*a*b through a lazy accumulator + full reduction.
 * Returns 0 on success. */
int spasm_amd_zp_probe(i64 prime, int n, const int *a, const int *b, const int *c, int *out);

#ifdef __cplusplus
}
#endif
#endif /* SPASM_AMD_H */
