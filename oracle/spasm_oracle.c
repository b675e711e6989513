/*
 * spasm_oracle.c -- CPU ORACLE for the echelonize / kernel hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product (spasm.jl_amd/, libspasm_amd.so) links, imports or calls this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the
 * checker / reported baseline.
 *
 * What it restates.  SpaSM.jl (reference, /root/reference/src/SpaSM.jl) is a ccall wrapper; the
 * arithmetic of the path lives in the third-party C library libspasm (github cbouilla/spasm),
 * shipped as the binary artifact Spasm_jll, version UNPINNED (Project.toml:15, no [compat] entry
 * Project.toml:19-24, Manifest.toml ignored .gitignore:28) and absent from /root/reference.  This
 * file therefore restates libspasm's published algorithm (Faugere-Lachartre structural pivots ->
 * sparse Schur complement by symbolic reach + numeric scatter -> GPLU finish -> kernel by solves
 * against the transpose of U) and anchors it on what the reference tree does hold:
 *   - field arithmetic            src/SpaSM.jl:73-76 (Field), :83-88 (normalize), :383-390 (+,-,*,inv,axpy)
 *   - CSR / LU / opts layouts     src/SpaSM.jl:126-134, :262-270, :325-343
 *   - triangular-solve semantics  src/SpaSM.jl:694-713 (x_b*U + x_a == B[k], unit pivots, xj = 3m ints)
 *   - scatter                     src/SpaSM.jl:619-620 (x += beta*A[i])
 *   - reach / dfs prototypes      src/SpaSM.jl:627-628
 *   - schur / pivots prototypes   src/SpaSM.jl:761-770, :776-778
 *   - qinv[j] == -1 <=> free col  src/SpaSM.jl:1152
 *   - phase order                 README.md:19-41
 * PINNING: checked against every known-answer vector the reference holds for this path
 * (test/runtests.jl:7-24, README.md:10-47) in tests/test_oracle_golden.py, plus an independent
 * dense elimination mod p (numpy) on random matrices.  Beyond those vectors parity is established
 * by uniqueness: every pivot chosen here is the LEFTMOST entry of its (reduced) row, hence the set
 * of pivot columns is the set of leading columns of the row space, an invariant of the matrix; the
 * kernel basis in libspasm's normal form (K[j] = -1 on its free column) is then unique.
 *
 * Deviation recorded: a pivot column whose accumulated value cancelled to exactly 0 is skipped
 * (no scatter, not counted); results are unaffected.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include <math.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/spasm_amd.h"

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ field, src/SpaSM.jl:73-88,383-390 */

ORC_API void orc_field_init(i64 p, struct spasm_field_struct *F)
{
    /* Field(p) = (p, p/2, p/2 - p + 1, 1/p), src/SpaSM.jl:73-76 */
    F->p = p;
    F->halfp = p / 2;
    F->mhalfp = p / 2 - p + 1;
    F->dinvp = 1.0 / (double)p;
}

static inline spasm_ZZp zp_normalize(const struct spasm_field_struct *F, i64 x)
{
    /* src/SpaSM.jl:83-88 */
    if (x < F->mhalfp) x += F->p;
    else if (x > F->halfp) x -= F->p;
    return (spasm_ZZp)x;
}

ORC_API spasm_ZZp orc_zp_init(const struct spasm_field_struct *F, i64 x)
{
    /* ZZp(F,x) = normalize(mod(x,p)), src/SpaSM.jl:96 (mod is the non-negative remainder) */
    i64 r = x % F->p;
    if (r < 0) r += F->p;
    return zp_normalize(F, r);
}

ORC_API spasm_ZZp orc_zp_add(const struct spasm_field_struct *F, spasm_ZZp a, spasm_ZZp b)
{
    return zp_normalize(F, (i64)a + (i64)b); /* :383 */
}

ORC_API spasm_ZZp orc_zp_sub(const struct spasm_field_struct *F, spasm_ZZp a, spasm_ZZp b)
{
    return zp_normalize(F, (i64)a - (i64)b); /* :384 */
}

ORC_API spasm_ZZp orc_zp_mul(const struct spasm_field_struct *F, spasm_ZZp a, spasm_ZZp b)
{
    /* q = round(a*b*dinvp); normalize(a*b - q*p), src/SpaSM.jl:385 */
    i64 q = (i64)nearbyint((double)a * (double)b * F->dinvp);
    return zp_normalize(F, (i64)a * (i64)b - q * F->p);
}

ORC_API spasm_ZZp orc_zp_axpy(const struct spasm_field_struct *F, spasm_ZZp a, spasm_ZZp x, spasm_ZZp y)
{
    /* src/SpaSM.jl:387-390 */
    i64 q = (i64)nearbyint(((double)a * (double)x + (double)y) * F->dinvp);
    return zp_normalize(F, (i64)a * (i64)x + (i64)y - q * F->p);
}

ORC_API spasm_ZZp orc_zp_inverse(const struct spasm_field_struct *F, spasm_ZZp a)
{
    /* inv(a) = normalize(gcdx(a mod p, p)[2]), src/SpaSM.jl:386: extended Euclid */
    i64 r0 = a < 0 ? a + F->p : a, r1 = F->p;
    i64 s0 = 1, s1 = 0;
    while (r1 != 0) {
        i64 q = r0 / r1;
        i64 t = r0 - q * r1; r0 = r1; r1 = t;
        t = s0 - q * s1; s0 = s1; s1 = t;
    }
    assert(r0 == 1);
    return orc_zp_init(F, s0);
}

/* ------------------------------------------------------------------ CSR container, src/SpaSM.jl:126-134,441-451 */

ORC_API struct spasm_csr *orc_csr_alloc(int n, int m, i64 nzmax, i64 prime, int with_values)
{
    struct spasm_csr *A = malloc(sizeof(*A));
    A->nzmax = nzmax;
    A->n = n;
    A->m = m;
    A->p = malloc(sizeof(i64) * ((size_t)n + 1));
    A->j = malloc(sizeof(int) * (size_t)(nzmax > 0 ? nzmax : 1));
    A->x = with_values ? malloc(sizeof(spasm_ZZp) * (size_t)(nzmax > 0 ? nzmax : 1)) : NULL;
    A->p[0] = 0;
    orc_field_init(prime, A->field);
    return A;
}

ORC_API void orc_csr_free(struct spasm_csr *A)
{
    if (!A) return;
    free(A->p); free(A->j); free(A->x); free(A);
}

static void csr_realloc(struct spasm_csr *A, i64 nzmax)
{
    A->j = realloc(A->j, sizeof(int) * (size_t)(nzmax > 0 ? nzmax : 1));
    if (A->x) A->x = realloc(A->x, sizeof(spasm_ZZp) * (size_t)(nzmax > 0 ? nzmax : 1));
    A->nzmax = nzmax;
}

ORC_API i64 orc_nnz(const struct spasm_csr *A) { return A->p[A->n]; } /* src/SpaSM.jl:432,1013 */

ORC_API void orc_lu_free(struct spasm_lu *N)
{
    if (!N) return;
    orc_csr_free(N->U); orc_csr_free(N->L); free(N->qinv); free(N->p); free(N);
}

/* ------------------------------------------------------------------ transpose (counting sort), src/SpaSM.jl:589 */

ORC_API struct spasm_csr *orc_transpose(const struct spasm_csr *A)
{
    int n = A->n, m = A->m;
    i64 nz = A->p[n];
    struct spasm_csr *T = orc_csr_alloc(m, n, nz, A->field->p, A->x != NULL);
    i64 *w = calloc((size_t)m + 1, sizeof(i64));
    for (i64 k = 0; k < nz; k++) w[A->j[k] + 1]++;
    for (int j = 0; j < m; j++) w[j + 1] += w[j];
    memcpy(T->p, w, sizeof(i64) * ((size_t)m + 1));
    for (int i = 0; i < n; i++)
        for (i64 k = A->p[i]; k < A->p[i + 1]; k++) {
            i64 q = w[A->j[k]]++;
            T->j[q] = i;
            if (T->x) T->x[q] = A->x[k];
        }
    free(w);
    return T;
}

/* ------------------------------------------------------------------ scatter, src/SpaSM.jl:619-620 */

static inline void scatter(const struct spasm_csr *A, int i, spasm_ZZp beta, spasm_ZZp *x)
{
    const struct spasm_field_struct *F = A->field;
    for (i64 k = A->p[i]; k < A->p[i + 1]; k++) {
        int j = A->j[k];
        x[j] = orc_zp_axpy(F, beta, A->x[k], x[j]);
    }
}

/* ------------------------------------------------------------------ reach, src/SpaSM.jl:627-628
 * Non-recursive DFS over the graph "column j -> columns of the pivot row U[qinv[j]]".
 * xj has 3*m ints: [0,m) output stack filled from the top, [m,2m) recursion stack, [2m,3m) marks
 * (doc-comment src/SpaSM.jl:699-700).  Marks are cleared on exit ("it remains OK"). */

static int dfs(int jstart, const struct spasm_csr *U, int top, int *xj, int *pstack, int *marks, const int *qinv)
{
    int head = 0;
    int *rstack = xj + U->m; /* recursion stack lives in xj[m..2m) */
    rstack[0] = jstart;
    while (head >= 0) {
        int j = rstack[head];
        int i = qinv[j];
        if (!marks[j]) {
            marks[j] = 1;
            pstack[head] = 0; /* next offset to visit inside the pivot row of column j */
        }
        int done = 1;
        if (i >= 0) {
            i64 base = U->p[i], len = U->p[i + 1] - base;
            for (int px = pstack[head]; px < len; px++) {
                int jj = U->j[base + px];
                if (marks[jj]) continue;
                pstack[head] = px + 1;
                rstack[++head] = jj;
                done = 0;
                break;
            }
        }
        if (done) {
            head--;
            xj[--top] = j;
        }
    }
    return top;
}

static int reach(const struct spasm_csr *U, const struct spasm_csr *B, int k, int *xj, int *pstack, const int *qinv)
{
    int m = U->m;
    int *marks = xj + 2 * m;
    int top = m;
    for (i64 px = B->p[k]; px < B->p[k + 1]; px++) {
        int j = B->j[px];
        if (!marks[j]) top = dfs(j, U, top, xj, pstack, marks, qinv);
    }
    for (int px = top; px < m; px++) marks[xj[px]] = 0;
    return top;
}

/* spasm_sparse_triangular_solve, semantics src/SpaSM.jl:694-713:
 * solve x*U = B[k]; pattern in xj[top:m); x_b*U + x_a == B[k]; pivots of U are 1.
 * `work` (i64[2], may be NULL) accumulates {applications, nnz scattered} = the scatter trip count. */
ORC_API int orc_sparse_triangular_solve(const struct spasm_csr *U, const struct spasm_csr *B, int k,
                                        int *xj, spasm_ZZp *x, const int *qinv, int *pstack, i64 *work)
{
    const struct spasm_field_struct *F = B->field;
    int m = U->m;
    int top = reach(U, B, k, xj, pstack, qinv);
    for (int px = top; px < m; px++) x[xj[px]] = 0;
    for (i64 px = B->p[k]; px < B->p[k + 1]; px++) x[B->j[px]] = B->x[px];
    for (int px = top; px < m; px++) {
        int j = xj[px];
        int i = qinv[j];
        if (i < 0) continue;
        spasm_ZZp xjv = x[j];
        if (xjv == 0) continue; /* cancelled: nothing to eliminate (deviation noted in header) */
        scatter(U, i, orc_zp_sub(F, 0, xjv), x); /* pivot is 1: multiply row by -x[j] */
        assert(x[j] == 0);
        x[j] = xjv; /* keep the coefficient: x_b is the row of L */
        if (work) { work[0] += 1; work[1] += U->p[i + 1] - U->p[i]; }
    }
    return top;
}

/* ------------------------------------------------------------------ Faugere-Lachartre pivots, proto src/SpaSM.jl:776-778
 * Candidate of a row = its leftmost entry; per column keep the sparsest candidate row (ties: lowest
 * row).  Pivot rows are copied into U scaled by pivot^-1 ("pivots in U are all equal to 1", :712).
 * Returns the number of new pivots; is_piv[i] set.
 *
 * on_columns != 0 adds the second search libspasm's log names (README.md:22 "Faugere-Lachartre on columns"; the option is
 * enable_greedy_pivot_search, src/SpaSM.jl:326).  libspasm's source is not in the reference tree, so its exact visiting order
 * cannot be followed; this is the order-free rule the MI355X engine uses (kernels.hpp, k_close_cols ..), restated so that
 * both take the SAME pivots and the parity tests stay bit-exact:
 *   - a column is closed when a leftmost-pivot row holds it, open otherwise;
 *   - a non-pivot row proposes its open column of smallest occupancy among the non-pivot rows (ties: leftmost);
 *   - per column the sparsest proposing row wins (ties: lowest row);
 *   - a winner is accepted when no other column of its row received a proposal;
 *   - the accepted rows become pivot rows, their columns are closed, and the rows left propose again (up to 4 passes).
 * Accepted rows enter U BEFORE the leftmost-pivot rows of the round (they may hold leftmost-pivot columns, never the other
 * way round), so U stays in topological order.  n_open, when given, receives the number of pivots this search added. */

static void emit_pivot_row(const struct spasm_csr *A, int i, int j, struct spasm_csr *U, int *qinv, char *is_piv, int *Uorig, const int *orig)
{
    const struct spasm_field_struct *F = A->field;
    i64 lo = A->p[i], hi = A->p[i + 1];
    i64 unz = U->p[U->n];
    if (unz + (hi - lo) > U->nzmax) csr_realloc(U, 2 * U->nzmax + (hi - lo));
    spasm_ZZp piv = 0;
    for (i64 k = lo; k < hi; k++) if (A->j[k] == j) piv = A->x[k];
    assert(piv != 0);
    spasm_ZZp inv = orc_zp_inverse(F, piv);
    for (i64 k = lo; k < hi; k++) {
        U->j[unz] = A->j[k];
        U->x[unz] = orc_zp_mul(F, inv, A->x[k]);
        unz++;
    }
    qinv[j] = U->n;
    if (Uorig) Uorig[U->n] = orig ? orig[i] : i;
    U->n++;
    U->p[U->n] = unz;
    is_piv[i] = 1;
}

/* The third search of the [pivots] log, "greedy alternating cycle-free search" (README.md:23; the option is
 * enable_greedy_pivot_search, src/SpaSM.jl:326).  As for "FL on columns", libspasm's source is not in the reference tree: its search
 * (a breadth-first search per row, rows of different threads checking each other's new pivots in a critical section) depends on the
 * order the rows are visited in, so this is the order-free rule of the MI355X engine (csrc/greedy.hpp), restated so that both take
 * the SAME pivots; tests/fl_columns_ref.py states it a third time, independently, in Python.
 *   pass (up to 3, until one accepts nothing), for every non-empty row i that is no pivot row and has at most 256 entries:
 *     reach(i)   = pivots reachable from row i through pivot columns; more than 1024 of them: the row sits the pass out;
 *     touched(i) = pivot-free columns held by the rows of reach(i);
 *     proposal   = the leftmost pivot-free column of row i that is not touched; per column the smallest (length, row) wins;
 *     a winner is accepted unless another column of (own pivot-free columns + touched) has a winner with a smaller key.
 * pc / pr: the pivots (column, row), np of them on entry; the new ones are appended.  Returns how many were added. */
#define GREEDY_REACH_MAX_DEFAULT 2
#define GREEDY_OCC_MAX_DEFAULT 1

/* breadth-first search from row i through the pivot rows: stamp[r] = mark for the rows reached (queue[0 .. *ntail) lists them).
 * Returns 0 when more than `budget` pivots are reached. */
static int greedy_search(const struct spasm_csr *A, int i, int mark, const int *prow_of_col, int *stamp, int *queue, int budget, int *ntail)
{
    int head = 0, tail = 0;
    for (i64 k = A->p[i]; k < A->p[i + 1]; k++) {
        int r = prow_of_col[A->j[k]];
        if (r >= 0 && stamp[r] != mark) {
            stamp[r] = mark;
            if (tail >= budget) return 0;
            queue[tail++] = r;
        }
    }
    while (head < tail) {
        int r = queue[head++];
        for (i64 k = A->p[r]; k < A->p[r + 1]; k++) {
            int r2 = prow_of_col[A->j[k]];
            if (r2 >= 0 && stamp[r2] != mark) {
                stamp[r2] = mark;
                if (tail >= budget) return 0;
                queue[tail++] = r2;
            }
        }
    }
    *ntail = tail;
    return 1;
}

static int greedy_extend(const struct spasm_csr *A, int *pc, int *pr, int np)
{
    enum { GREEDY_PASSES = 3, GR_MAXLEN = 256, GR_BUDGET = 1024 };
    /* the two limits of csrc/greedy.hpp (same defaults, same environment variables): rows that reach more pivots sit the pass out,
     * columns more rows without pivot hold are no candidates */
    int reach_max = GREEDY_REACH_MAX_DEFAULT, occ_max = GREEDY_OCC_MAX_DEFAULT;
    { const char *e = getenv("SPASM_AMD_GREEDY_REACH_MAX"); if (e) { reach_max = atoi(e); if (reach_max > GR_BUDGET) reach_max = GR_BUDGET; if (reach_max < 0) reach_max = 0; } }
    { const char *e = getenv("SPASM_AMD_GREEDY_OCC_MAX"); if (e) { occ_max = atoi(e); if (occ_max < 1) occ_max = 1; } }
    const int n = A->n, m = A->m;
    int *occ = malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    int *prow_of_col = malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    char *taken = calloc((size_t)(n > 0 ? n : 1), 1);
    int *stamp = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));   /* row reached by the search numbered stamp[.] */
    int *cstamp = malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));  /* column touched by the search numbered cstamp[.] */
    int *queue = malloc(sizeof(int) * (size_t)(GR_BUDGET + 2));
    int *prop = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int *acc = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    i64 *winner = malloc(sizeof(i64) * (size_t)(m > 0 ? m : 1));
    int added = 0, mark = 0;
    for (int j = 0; j < m; j++) { prow_of_col[j] = -1; cstamp[j] = -1; }
    for (int i = 0; i < n; i++) stamp[i] = -1;
    for (int t = 0; t < np; t++) { prow_of_col[pc[t]] = pr[t]; taken[pr[t]] = 1; }
    for (int pass = 1; pass <= GREEDY_PASSES; pass++) {
        for (int j = 0; j < m; j++) { winner[j] = -1; occ[j] = 0; }
        for (int i = 0; i < n; i++)
            if (!taken[i]) for (i64 k = A->p[i]; k < A->p[i + 1]; k++) occ[A->j[k]]++;
        for (int i = 0; i < n; i++) {
            prop[i] = -1;
            i64 len = A->p[i + 1] - A->p[i];
            if (taken[i] || len == 0 || len > GR_MAXLEN) continue;
            int tail = 0;
            mark++;
            if (!greedy_search(A, i, mark, prow_of_col, stamp, queue, reach_max, &tail)) continue;
            for (int t = 0; t < tail; t++)
                for (i64 k = A->p[queue[t]]; k < A->p[queue[t] + 1]; k++) cstamp[A->j[k]] = mark; /* (pivot columns too: harmless) */
            int best = -1;
            for (i64 k = A->p[i]; k < A->p[i + 1]; k++) {
                int c = A->j[k];
                if (prow_of_col[c] >= 0 || cstamp[c] == mark || occ[c] > occ_max) continue;
                if (best < 0 || occ[c] < occ[best] || (occ[c] == occ[best] && c < best)) best = c;
            }
            if (best < 0) continue;
            prop[i] = best;
            i64 key = (len << 32) | (i64)i;
            if (winner[best] < 0 || key < winner[best]) winner[best] = key;
        }
        int nacc = 0;
        for (int i = 0; i < n; i++) {
            int j = prop[i];
            if (j < 0) continue;
            i64 len = A->p[i + 1] - A->p[i];
            i64 key = (len << 32) | (i64)i;
            if (winner[j] != key) continue;
            int tail = 0;
            mark++;
            if (!greedy_search(A, i, mark, prow_of_col, stamp, queue, reach_max, &tail)) continue;
            int rejected = 0;
            /* full(i) = the pivot-free columns of row i and of the rows it reaches */
            for (int t = -1; t < tail && !rejected; t++) {
                int r = t < 0 ? i : queue[t];
                for (i64 k = A->p[r]; k < A->p[r + 1]; k++) {
                    int c = A->j[k];
                    if (prow_of_col[c] < 0 && c != j && winner[c] >= 0 && winner[c] < key) { rejected = 1; break; }
                }
            }
            if (!rejected) acc[nacc++] = i;
        }
        for (int t = 0; t < nacc; t++) {
            int i = acc[t];
            pc[np + added] = prop[i];
            pr[np + added] = i;
            added++;
            prow_of_col[prop[i]] = i;
            taken[i] = 1;
        }
        if (nacc == 0) break;
    }
    free(prow_of_col); free(taken); free(stamp); free(cstamp); free(queue); free(prop); free(acc); free(winner); free(occ);
    return added;
}

/* all pivots of a round in a topological order when the greedy search has added some: descending level (0 = the row holds no other
 * pivot column, else 1 + the deepest level among the pivot columns it holds), ascending column inside a level */
static void topological_numbering(const struct spasm_csr *A, int *pc, int *pr, int np)
{
    const int m = A->m;
    int *idx_of_col = malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    int *lev = calloc((size_t)(np > 0 ? np : 1), sizeof(int));
    for (int j = 0; j < m; j++) idx_of_col[j] = -1;
    for (int t = 0; t < np; t++) idx_of_col[pc[t]] = t;
    for (int sweep = 0, changed = 1; changed; sweep++) {
        assert(sweep <= np + 1);
        changed = 0;
        for (int t = 0; t < np; t++) {
            int l = 0;
            for (i64 k = A->p[pr[t]]; k < A->p[pr[t] + 1]; k++) {
                int c = A->j[k];
                if (c != pc[t] && idx_of_col[c] >= 0 && lev[idx_of_col[c]] + 1 > l) l = lev[idx_of_col[c]] + 1;
            }
            if (l > lev[t]) { lev[t] = l; changed = 1; }
        }
    }
    /* sort by (-level, column): counting on the columns, then stable on the levels */
    int *bycol = malloc(sizeof(int) * (size_t)(np > 0 ? np : 1));
    int w = 0, maxlev = 0;
    for (int j = 0; j < m; j++) if (idx_of_col[j] >= 0) bycol[w++] = idx_of_col[j];
    for (int t = 0; t < np; t++) if (lev[t] > maxlev) maxlev = lev[t];
    int *npc = malloc(sizeof(int) * (size_t)(np > 0 ? np : 1)), *npr = malloc(sizeof(int) * (size_t)(np > 0 ? np : 1));
    w = 0;
    for (int l = maxlev; l >= 0; l--)
        for (int t = 0; t < np; t++)
            if (lev[bycol[t]] == l) { npc[w] = pc[bycol[t]]; npr[w] = pr[bycol[t]]; w++; }
    assert(w == np);
    memcpy(pc, npc, sizeof(int) * (size_t)np);
    memcpy(pr, npr, sizeof(int) * (size_t)np);
    free(idx_of_col); free(lev); free(bycol); free(npc); free(npr);
}

static int fl_pivots_ex(const struct spasm_csr *A, struct spasm_csr *U, int *qinv, char *is_piv, int *Uorig, const int *orig,
                        int on_columns, int *n_open, int *n_greedy)
{
    int emitted = 0, ngreedy = 0;
    int n = A->n, m = A->m;
    int *best = malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    for (int j = 0; j < m; j++) best[j] = -1;
    for (int i = 0; i < n; i++) {
        is_piv[i] = 0;
        i64 lo = A->p[i], hi = A->p[i + 1];
        if (lo == hi) continue;
        int jmin = A->j[lo];
        for (i64 k = lo + 1; k < hi; k++) if (A->j[k] < jmin) jmin = A->j[k];
        assert(qinv[jmin] < 0);
        int b = best[jmin];
        if (b < 0 || (hi - lo) < (A->p[b + 1] - A->p[b])) best[jmin] = i;
    }
    int npiv = 0, nopen = 0;
    if (on_columns) {
        /* up to OPEN_PASSES passes (kernels.hpp): the rows a pass accepts become pivot rows, their columns are closed, and the
         * rows left propose again; the pivots of a LATER pass enter U BEFORE those of an earlier one */
        enum { OPEN_PASSES = 4 };
        char *taken = calloc((size_t)(n > 0 ? n : 1), 1);   /* row is a pivot row (leftmost, or of an earlier pass) */
        char *closed = calloc((size_t)(m > 0 ? m : 1), 1);
        int *cnt = malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
        int *best2 = malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
        int *pcol = malloc(sizeof(int) * (size_t)(m > 0 ? m : 1)); /* accepted (column, row) pairs, pass after pass */
        int *prow = malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
        int pstart[OPEN_PASSES + 2];
        int npass = 0, any = 0;
        pstart[0] = 0;
        for (int j = 0; j < m; j++) if (best[j] >= 0) { taken[best[j]] = 1; any = 1; }
        if (any) {
            for (int i = 0; i < n; i++)
                if (taken[i]) for (i64 k = A->p[i]; k < A->p[i + 1]; k++) closed[A->j[k]] = 1;
            for (int pass = 1; pass <= OPEN_PASSES; pass++) {
                for (int j = 0; j < m; j++) { cnt[j] = 0; best2[j] = -1; }
                for (int i = 0; i < n; i++)
                    if (!taken[i]) for (i64 k = A->p[i]; k < A->p[i + 1]; k++) cnt[A->j[k]]++;
                for (int i = 0; i < n; i++) {
                    if (taken[i]) continue;
                    i64 lo = A->p[i], hi = A->p[i + 1];
                    int c = -1;
                    for (i64 k = lo; k < hi; k++) {
                        int j = A->j[k];
                        if (closed[j]) continue;
                        if (c < 0 || cnt[j] < cnt[c] || (cnt[j] == cnt[c] && j < c)) c = j;
                    }
                    if (c < 0) continue;
                    int b = best2[c];
                    if (b < 0 || (hi - lo) < (A->p[b + 1] - A->p[b])) best2[c] = i;
                }
                int at = pstart[pass - 1];
                for (int j = 0; j < m; j++) {
                    int i = best2[j];
                    if (i < 0) continue;
                    int clash = 0;
                    for (i64 k = A->p[i]; k < A->p[i + 1]; k++) if (A->j[k] != j && best2[A->j[k]] >= 0) clash = 1;
                    if (clash) continue;
                    pcol[at] = j; prow[at] = i; at++;
                }
                pstart[pass] = at;
                if (at == pstart[pass - 1]) break;
                npass = pass;
                for (int t = pstart[pass - 1]; t < at; t++) {
                    taken[prow[t]] = 1;
                    for (i64 k = A->p[prow[t]]; k < A->p[prow[t] + 1]; k++) closed[A->j[k]] = 1;
                }
            }
            /* all pivots of the round in the numbering of the first two searches: later passes first, then the leftmost ones */
            int *ac = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)), *ar = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
            int na = 0;
            for (int pass = npass; pass >= 1; pass--)
                for (int t = pstart[pass - 1]; t < pstart[pass]; t++) { ac[na] = pcol[t]; ar[na] = prow[t]; na++; nopen++; }
            for (int j = 0; j < m; j++) if (best[j] >= 0) { ac[na] = j; ar[na] = best[j]; na++; npiv++; }
            /* the third search (on_columns >= 2); when it finds anything all pivots are renumbered topologically */
            if (on_columns >= 2) {
                ngreedy = greedy_extend(A, ac, ar, na);
                if (ngreedy > 0) { na += ngreedy; topological_numbering(A, ac, ar, na); }
            }
            for (int t = 0; t < na; t++) emit_pivot_row(A, ar[t], ac[t], U, qinv, is_piv, Uorig, orig);
            free(ac); free(ar);
            emitted = 1;
        }
        free(taken); free(closed); free(cnt); free(best2); free(pcol); free(prow);
    }
    if (!emitted)
        for (int j = 0; j < m; j++) {
            int i = best[j];
            if (i < 0) continue;
            emit_pivot_row(A, i, j, U, qinv, is_piv, Uorig, orig);
            npiv++;
        }
    free(best);
    if (n_open) *n_open = nopen;
    if (n_greedy) *n_greedy = ngreedy;
    return npiv + nopen + ngreedy;
}

static int fl_pivots(const struct spasm_csr *A, struct spasm_csr *U, int *qinv, int *Urow_of, char *is_piv, int *Uorig, const int *orig)
{
    (void)Urow_of;
    return fl_pivots_ex(A, U, qinv, is_piv, Uorig, orig, 0, NULL, NULL);
}

/* ------------------------------------------------------------------ Schur complement, proto src/SpaSM.jl:761-762
 * For each non-pivot row: solve against U, keep the entries on non-pivot columns.  OpenMP over rows
 * with a per-thread dense x (4m bytes) and xj (12m bytes) as in libspasm (src/SpaSM.jl:699-700).
 * Unlike libspasm the output keeps the input row order (deterministic), built in two passes.
 * stats[0] += applications, stats[1] += nnz_reduced (= sum nnz(A_i) + sum nnz(U_r) per application). */

static struct spasm_csr *schur_range(const struct spasm_csr *A, const char *is_piv, const struct spasm_csr *U,
                                     const int *qinv, int *p_out, i64 *stats, int keep_empty, int row_lo, int row_hi)
{
    int n = A->n, m = A->m;
    int *rows = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int nr = 0;
    if (row_lo < 0) row_lo = 0;
    if (row_hi > n) row_hi = n;
    for (int i = row_lo; i < row_hi; i++) if (!is_piv[i] && (keep_empty || A->p[i + 1] > A->p[i])) rows[nr++] = i;
    /* every thread appends its Schur rows to an arena of its own; the rows are copied to their place in S by a second
     * parallel loop (a serial gather of the 4 GB of config 3 took three times as long as the elimination itself) */
    int nth = omp_get_max_threads();
    int **aj = calloc((size_t)nth, sizeof(*aj));
    spasm_ZZp **ax = calloc((size_t)nth, sizeof(*ax));
    int *rth = malloc(sizeof(int) * (size_t)(nr > 0 ? nr : 1));
    i64 *roff = malloc(sizeof(i64) * (size_t)(nr > 0 ? nr : 1));
    i64 *rn = malloc(sizeof(i64) * (size_t)(nr + 1));
    i64 tot_app = 0, tot_red = 0;
    double tpar0 = omp_get_wtime();
#pragma omp parallel reduction(+ : tot_app, tot_red)
    {
        const int tid = omp_get_thread_num();
        spasm_ZZp *x = malloc(sizeof(spasm_ZZp) * (size_t)m);
        int *xj = calloc(3 * (size_t)m, sizeof(int));
        int *pstack = malloc(sizeof(int) * (size_t)m);
        i64 cap = 1 << 16, len = 0;
        int *lj = malloc(sizeof(int) * (size_t)cap);
        spasm_ZZp *lx = malloc(sizeof(spasm_ZZp) * (size_t)cap);
#pragma omp for schedule(dynamic, 64)
        for (int t = 0; t < nr; t++) {
            int i = rows[t];
            i64 work[2] = {0, 0};
            int top = orc_sparse_triangular_solve(U, A, i, xj, x, qinv, pstack, work);
            tot_app += work[0];
            tot_red += work[1] + (A->p[i + 1] - A->p[i]);
            if (len + (m - top) > cap) {
                while (len + (m - top) > cap) cap *= 2;
                lj = realloc(lj, sizeof(int) * (size_t)cap);
                lx = realloc(lx, sizeof(spasm_ZZp) * (size_t)cap);
            }
            i64 cnt = 0;
            for (int px = top; px < m; px++) {
                int j = xj[px];
                if (qinv[j] < 0 && x[j] != 0) { lj[len + cnt] = j; lx[len + cnt] = x[j]; cnt++; }
            }
            rn[t] = cnt;
            rth[t] = tid;
            roff[t] = len;
            len += cnt;
        }
        aj[tid] = lj;
        ax[tid] = lx;
        free(x); free(xj); free(pstack);
    }
    double tpar1 = omp_get_wtime();
    i64 tot = 0;
    int nout = 0;
    i64 *rdst = malloc(sizeof(i64) * (size_t)(nr + 1)); /* where row t starts in S, -1 when it is dropped */
    int *rrow = malloc(sizeof(int) * (size_t)(nr + 1));
    for (int t = 0; t < nr; t++) {
        if (keep_empty || rn[t] > 0) { rdst[t] = tot; rrow[t] = nout; tot += rn[t]; nout++; }
        else { rdst[t] = -1; rrow[t] = -1; }
    }
    struct spasm_csr *S = orc_csr_alloc(nout, m, tot, A->field->p, 1);
    S->p[0] = 0;
#pragma omp parallel for schedule(static)
    for (int t = 0; t < nr; t++) {
        if (rdst[t] < 0) continue;
        memcpy(S->j + rdst[t], aj[rth[t]] + roff[t], sizeof(int) * (size_t)rn[t]);
        memcpy(S->x + rdst[t], ax[rth[t]] + roff[t], sizeof(spasm_ZZp) * (size_t)rn[t]);
        S->p[rrow[t] + 1] = rdst[t] + rn[t];
        if (p_out) p_out[rrow[t]] = rows[t];
    }
    if (getenv("ORC_TIMING")) fprintf(stderr, "[oracle] schur rows: %.3f s elimination, %.3f s gather\n", tpar1 - tpar0, omp_get_wtime() - tpar1);
    for (int k = 0; k < nth; k++) { free(aj[k]); free(ax[k]); }
    free(aj); free(ax); free(rth); free(roff); free(rdst); free(rrow);
    if (stats) { stats[0] += tot_app; stats[1] += tot_red; }
    free(rows); free(rn);
    return S;
}

ORC_API struct spasm_csr *orc_schur(const struct spasm_csr *A, const char *is_piv, const struct spasm_csr *U,
                                    const int *qinv, int *p_out, i64 *stats, int keep_empty)
{
    return schur_range(A, is_piv, U, qinv, p_out, stats, keep_empty, 0, A->n);
}

/* One Schur round of A on its own (BASELINE config 3 unit of work): elect the FL pivots of A,
 * build U, reduce every non-pivot row.  out[0]=npiv out[1]=applications out[2]=nnz_reduced
 * out[3]=nnz(S) out[4]=rows(S non-empty) out[5]=nnz(U); seconds[0]=pivots seconds[1]=schur. */
ORC_API struct spasm_csr *orc_schur_round_range(const struct spasm_csr *A, int row_lo, int row_hi, i64 *out, double *seconds,
                                                struct spasm_csr **U_out, int *qinv_out);

ORC_API struct spasm_csr *orc_schur_round(const struct spasm_csr *A, i64 *out, double *seconds,
                                          struct spasm_csr **U_out, int *qinv_out)
{
    return orc_schur_round_range(A, 0, A->n, out, seconds, U_out, qinv_out);
}

/* Same, but only the non-pivot rows inside [row_lo,row_hi) are reduced (the pivots are still those of
 * the whole matrix): a bounded sample of the round for the timed CPU baseline, and a row shard. */
ORC_API struct spasm_csr *orc_schur_round_range(const struct spasm_csr *A, int row_lo, int row_hi, i64 *out, double *seconds,
                                                struct spasm_csr **U_out, int *qinv_out)
{
    int n = A->n, m = A->m;
    struct timespec t0, t1, t2;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    struct spasm_csr *U = orc_csr_alloc(n < m ? n : m, m, orc_nnz(A) / 4 + 16, A->field->p, 1);
    U->n = 0;
    int *qinv = malloc(sizeof(int) * (size_t)m);
    for (int j = 0; j < m; j++) qinv[j] = -1;
    char *is_piv = malloc((size_t)(n > 0 ? n : 1));
    int npiv = fl_pivots(A, U, qinv, NULL, is_piv, NULL, NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    i64 stats[2] = {0, 0};
    struct spasm_csr *S = schur_range(A, is_piv, U, qinv, NULL, stats, 1, row_lo, row_hi);
    clock_gettime(CLOCK_MONOTONIC, &t2);
    int nonempty = 0;
    for (int i = 0; i < S->n; i++) nonempty += S->p[i + 1] > S->p[i];
    if (out) { out[0] = npiv; out[1] = stats[0]; out[2] = stats[1]; out[3] = orc_nnz(S); out[4] = nonempty; out[5] = orc_nnz(U); }
    if (seconds) {
        seconds[0] = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
        seconds[1] = (t2.tv_sec - t1.tv_sec) + 1e-9 * (t2.tv_nsec - t1.tv_nsec);
    }
    if (qinv_out) memcpy(qinv_out, qinv, sizeof(int) * (size_t)m);
    if (U_out) *U_out = U; else orc_csr_free(U);
    free(qinv); free(is_piv);
    return S;
}

/* ------------------------------------------------------------------ echelonize, call site src/SpaSM.jl:863; log README.md:19-38
 * round loop: FL pivots -> stop test -> Schur -> repeat (<= max_round); finish with GPLU:
 * each remaining row is solved against U and its LEFTMOST surviving entry becomes a pivot. */

ORC_API void orc_echelonize_init_opts(struct echelonize_opts *o)
{
    /* field list src/SpaSM.jl:325-343; default values are libspasm's (recalled, not in tree) */
    memset(o, 0, sizeof(*o));
    o->enable_greedy_pivot_search = 1;
    o->enable_tall_and_skinny = 1;
    o->enable_dense = 1;
    o->enable_GPLU = 1;
    o->L = 0;
    o->complete = 0;
    o->min_pivot_proportion = 0.1;
    o->max_round = 3;
    o->sparsity_threshold = 0.05;
    o->dense_block_size = 1000;
    o->low_rank_ratio = 0.5;
    o->tall_and_skinny_ratio = 5;
    o->low_rank_start_weight = -1;
}

ORC_API struct spasm_lu *orc_echelonize(const struct spasm_csr *A0, const struct echelonize_opts *opts_in, i64 *stats)
{
    struct echelonize_opts dflt;
    if (!opts_in) { orc_echelonize_init_opts(&dflt); opts_in = &dflt; }
    int n = A0->n, m = A0->m;
    i64 prime = A0->field->p;
    const struct spasm_field_struct *F = A0->field;
    int maxr = n < m ? n : m;
    struct spasm_csr *U = orc_csr_alloc(maxr, m, orc_nnz(A0) + 16, prime, 1);
    U->n = 0;
    int *qinv = malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    for (int j = 0; j < m; j++) qinv[j] = -1;
    int plen = n > m ? n : m;
    int *Uorig = malloc(sizeof(int) * (size_t)(plen > 0 ? plen : 1));

    /* live matrix: starts as a borrowed view of A0 */
    const struct spasm_csr *A = A0;
    struct spasm_csr *owned = NULL;
    int *orig = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)); /* original row of each live row */
    for (int i = 0; i < n; i++) orig[i] = i;

    for (int round = 0; round < opts_in->max_round; round++) {
        if (orc_nnz(A) == 0 || U->n == maxr) break;
        char *is_piv = malloc((size_t)(A->n > 0 ? A->n : 1));
        int rank_before = U->n;
        /* enable_greedy_pivot_search: "FL on columns" and the greedy cycle-free search (SPASM_AMD_NO_CYCLE_FREE_SEARCH=1, read by
         * the engine as well: without the third search, for A/B tests) */
        const char *no3 = getenv("SPASM_AMD_NO_CYCLE_FREE_SEARCH");
        int searches = opts_in->enable_greedy_pivot_search ? ((no3 && atoi(no3)) ? 1 : 2) : 0;
        int npiv = fl_pivots_ex(A, U, qinv, is_piv, Uorig, orig, searches, NULL, NULL);
        int avail = A->n < m - rank_before ? A->n : m - rank_before;
        int *p_out = malloc(sizeof(int) * (size_t)(A->n > 0 ? A->n : 1));
        struct spasm_csr *S = orc_schur(A, is_piv, U, qinv, p_out, stats, 0);
        int *norig = malloc(sizeof(int) * (size_t)(S->n > 0 ? S->n : 1));
        for (int i = 0; i < S->n; i++) norig[i] = orig[p_out[i]];
        free(orig); orig = norig;
        free(p_out); free(is_piv);
        orc_csr_free(owned);
        owned = S; A = S;
        if (npiv < opts_in->min_pivot_proportion * avail) break; /* README.md:32 "not enough pivots found" */
    }

    /* GPLU finish (README.md:34): sequential, leftmost surviving entry is the new pivot */
    {
        spasm_ZZp *x = malloc(sizeof(spasm_ZZp) * (size_t)(m > 0 ? m : 1));
        int *xj = calloc(3 * (size_t)(m > 0 ? m : 1), sizeof(int));
        int *pstack = malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
        for (int i = 0; i < A->n && U->n < maxr; i++) {
            if (A->p[i + 1] == A->p[i]) continue;
            i64 work[2] = {0, 0};
            int top = orc_sparse_triangular_solve(U, A, i, xj, x, qinv, pstack, work);
            if (stats) { stats[0] += work[0]; stats[1] += work[1] + (A->p[i + 1] - A->p[i]); }
            int jpiv = -1;
            i64 cnt = 0;
            for (int px = top; px < m; px++) {
                int j = xj[px];
                if (qinv[j] < 0 && x[j] != 0) { cnt++; if (jpiv < 0 || j < jpiv) jpiv = j; }
            }
            if (jpiv < 0) continue;
            i64 unz = U->p[U->n];
            if (unz + cnt > U->nzmax) csr_realloc(U, 2 * U->nzmax + cnt);
            spasm_ZZp inv = orc_zp_inverse(F, x[jpiv]);
            for (int px = top; px < m; px++) {
                int j = xj[px];
                if (qinv[j] < 0 && x[j] != 0) { U->j[unz] = j; U->x[unz] = orc_zp_mul(F, inv, x[j]); unz++; }
            }
            qinv[jpiv] = U->n;
            Uorig[U->n] = orig[i];
            U->n++;
            U->p[U->n] = unz;
        }
        free(x); free(xj); free(pstack);
    }
    orc_csr_free(owned);
    free(orig);

    struct spasm_lu *N = malloc(sizeof(*N));
    N->r = U->n;
    N->complete = 0;
    N->L = NULL;
    N->U = U;
    N->qinv = qinv;
    for (int i = U->n; i < plen; i++) Uorig[i] = -1;
    N->p = Uorig;
    N->Ltmp = NULL;
    return N;
}

/* ------------------------------------------------------------------ kernel, call site src/SpaSM.jl:879; README.md:39-41
 * Right-kernel basis: transpose U; for each free column j (qinv[j] < 0, :1152), ascending, solve
 * y * Ut = Ut[j] (every column of Ut carries a pivot: the pivot of U-row r sits on Ut-row c_r);
 * emit the row {(j,-1)} U {(c_r, y_r)}.  Known answers: test/runtests.jl:20-23, README.md:44-47. */

ORC_API struct spasm_csr *orc_kernel(const struct spasm_lu *fact)
{
    const struct spasm_csr *U = fact->U;
    int r = U->n, m = U->m;
    const int *qinv = fact->qinv;
    struct spasm_csr *Ut = orc_transpose(U);           /* m x r */
    int *q = malloc(sizeof(int) * (size_t)(r > 0 ? r : 1)); /* q[r'] = pivot column of U-row r' = pivot ROW of Ut-column r' */
    for (int j = 0; j < m; j++) if (qinv[j] >= 0) q[qinv[j]] = j;
    /* view Ut restricted to its pivotal rows as a triangular matrix T (r x r): T-row r' = Ut-row q[r'] */
    struct spasm_csr T = *Ut;
    T.n = r; T.m = r;
    i64 *Tp = malloc(sizeof(i64) * ((size_t)r + 1));
    i64 tnz = 0;
    for (int a = 0; a < r; a++) tnz += Ut->p[q[a] + 1] - Ut->p[q[a]];
    int *Tj = malloc(sizeof(int) * (size_t)(tnz > 0 ? tnz : 1));
    spasm_ZZp *Tx = malloc(sizeof(spasm_ZZp) * (size_t)(tnz > 0 ? tnz : 1));
    tnz = 0;
    for (int a = 0; a < r; a++) {
        Tp[a] = tnz;
        for (i64 k = Ut->p[q[a]]; k < Ut->p[q[a] + 1]; k++) { Tj[tnz] = Ut->j[k]; Tx[tnz] = Ut->x[k]; tnz++; }
    }
    Tp[r] = tnz;
    T.p = Tp; T.j = Tj; T.x = Tx;
    int *ident = malloc(sizeof(int) * (size_t)(r > 0 ? r : 1)); /* pivot of T-column a is on T-row a */
    for (int a = 0; a < r; a++) ident[a] = a;
    /* B = the free rows of Ut seen as rows over the r columns */
    struct spasm_csr B = *Ut;
    B.m = r;

    int nfree = m - r;
    struct spasm_csr *K = orc_csr_alloc(nfree, m, 16 + 2 * (i64)nfree, U->field->p, 1);
    K->n = 0;
    spasm_ZZp *x = malloc(sizeof(spasm_ZZp) * (size_t)(r > 0 ? r : 1));
    int *xj = calloc(3 * (size_t)(r > 0 ? r : 1), sizeof(int));
    int *pstack = malloc(sizeof(int) * (size_t)(r > 0 ? r : 1));
    i64 knz = 0;
    for (int j = 0; j < m; j++) {
        if (qinv[j] >= 0) continue;
        int top = r;
        if (r > 0) top = orc_sparse_triangular_solve(&T, &B, j, xj, x, ident, pstack, NULL);
        i64 need = knz + (r - top) + 1;
        if (need > K->nzmax) csr_realloc(K, 2 * need);
        K->j[knz] = j; K->x[knz] = -1; knz++;           /* K[j] = -1 (test/runtests.jl:21: 42012 == -1) */
        for (int px = top; px < r; px++) {
            int a = xj[px];
            if (x[a] != 0) { K->j[knz] = q[a]; K->x[knz] = x[a]; knz++; }
        }
        K->n++;
        K->p[K->n] = knz;
    }
    free(x); free(xj); free(pstack); free(ident); free(Tp); free(Tj); free(Tx); free(q);
    orc_csr_free(Ut);
    return K;
}

/* n > 0: use n OpenMP threads from now on; n <= 0: back to all of them (the 1-thread CPU baseline of bench.py) */
ORC_API void orc_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
#else
    (void)n;
#endif
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
