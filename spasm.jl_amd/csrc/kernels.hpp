// kernels.hpp -- hand-written gfx950 kernels of the echelonization round.
//
// One round = (1) Faugere-Lachartre pivot election, (2) U build (pivot rows scaled to unit pivot and
// split by column class), (3) SOLVE: per non-pivot row, the multipliers of the pivot rows that reach
// it (sparse triangular solve restricted to pivot columns), (4) SCATTER: per non-pivot row, the Schur
// row = its non-pivot entries minus sum(multiplier * pivot row) accumulated in an LDS hash table.
//
// Replaces libspasm's spasm_pivots_extract_structural / spasm_schur / spasm_sparse_triangular_solve /
// spasm_scatter as SpaSM.jl documents them (reference src/SpaSM.jl:776-778, :761-762, :694-713, :619-620).
//
// Why the split into SOLVE and SCATTER (instead of libspasm's per-row reach + scatter over a dense x):
// x_b * U + x_a = B[k]  (reference src/SpaSM.jl:704-707) separates into
//        x_b * U_PP = B[k]_P          (dependent chain, but U_PP holds only ~14 % of U's entries)
//        x_a = B[k]_N - x_b * U_PN    (no ordering constraint: every pivot row applied concurrently)
// with U_PP in PIVOT-INDEX space (strictly upper triangular: pivots are numbered in a topological
// order) and U_PN in column space.  The scatter trip count is the same as the reference's.
//
// Layout in HBM: rows are (start:i64, len:i32) slices of an array of {col:i32, val:i32} pairs
// ("CSR with slack": a Schur row is written compactly at the start of a slot sized by its bound),
// so one 8-byte load fetches an entry and a 16-lane group reads a 128-byte line of a pivot row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "zp.hpp"

typedef long long i64d;
typedef unsigned long long u64d;

#define EMPTY_KEY (-1)
#define NO_BEST 0x7fffffffffffffffull   // larger than any (len << 32 | row) key, also as a SIGNED 64-bit value (all-reduce MIN)

// header of a pivot row: off = offset of its slot in UPP/UPN/Ufull, npp/npn = entries on pivot /
// non-pivot columns (pivot entry itself excluded), len = full length (npp + npn + 1)
struct __attribute__((aligned(16))) UHdr {
    unsigned off;
    int npp;
    int npn;
    int len;
};

// The multiplier pool is cut into NPOOL regions with one bump counter each (own cache line): a single
// counter would serialise ~10^6 returning atomics per round (~4 ns each on one address).
#define NPOOL 1024
#define POOL_STRIDE 16   // u64 words between two counters (128 bytes)

// Statistics are accumulated in NCTR copies of RoundCounters (one per workgroup id mod NCTR, summed on the host):
// tens of thousands of waves adding to ONE address at the end of a kernel serialise at ~5 ns each (0.3-0.5 ms per
// kernel, measured).  The *_overflow counters that hand out list positions stay in copy 0.
#define NCTR 256
struct RoundCounters;
__device__ __forceinline__ RoundCounters *ctr_shard(RoundCounters *base);

struct RoundCounters {
    u64d applications;   // (row, pivot row) eliminations with nonzero multiplier
    u64d nnz_reduced;    // reference scatter trip count: sum nnz(A_i) + sum nnz(U_r) over applications
    u64d segments;       // row segments visited (1 per row + 1 per application)
    u64d lpool_used;     // (unsharded users only) entries of the pool handed out
    int combine_overflow;// rows with more distinct pivots than the first combine class accepts
    int solve_overflow;  // rows whose reach did not fit the LDS list of the first solve class
    int solve_failed;    // rows whose reach did not fit the largest solve class
    int lpool_overflow;  // multiplier pool exhausted
    int wplan_reject;    // rows the W plan leaves to the multiplier-list path
    int scatter_overflow;// rows whose bound exceeds the largest hash table class
    int nonempty_out;    // non-empty Schur rows
    u64d nnz_out;        // entries of the Schur complement
    u64d class_ent[16];  // scatter kernels, per class (0..7 hash tables + last resort, 8..15 streaming): entries streamed
    u64d class_seg[16];  // scatter kernels, per class: row segments visited
    int stream_redo;     // rows the streaming kernel handed back to the hash-table kernel (too many duplicate columns)
    int stream_fix;      // duplicate columns the streaming kernel merged after the fact
};

// bump allocation of n entries from the region of this workgroup; returns the absolute offset in the pool,
// or ~0 when the region is full (the host then grows the pool and reruns)
__device__ __forceinline__ u64d pool_alloc(u64d *counters, u64d region_cap, u64d n, int npool)
{
    const unsigned region = blockIdx.x % (unsigned)npool; // npool <= NPOOL regions are in use (fewer when there are few rows)
    const u64d pos = atomicAdd(&counters[(size_t)region * POOL_STRIDE], n);
    if (pos + n > region_cap) return ~0ull;
    return (u64d)region * region_cap + pos;
}

__device__ __forceinline__ RoundCounters *ctr_shard(RoundCounters *base) { return base + (blockIdx.x & (NCTR - 1)); }

template <int TEAM> __device__ __forceinline__ u64d team_ballot(bool pred)
{
    u64d b = __ballot(pred);
    if (TEAM == 64) return b;
    const int base = (threadIdx.x & 63) & ~(TEAM - 1);
    return (b >> base) & ((1ull << (TEAM & 63)) - 1ull);
}

// workgroup barrier that orders LDS traffic only: outstanding global loads / stores stay in flight
// (__syncthreads() would drain them: s_waitcnt vmcnt(0) before every s_barrier)
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// min over the 64 lanes without touching LDS (ds_bpermute shuffles cost an LDS round trip each):
// DPP row shifts inside the 16-lane rows, then the row_bcast steps of the GFX9 reduction idiom
__device__ __forceinline__ int wave_min_i32(int x)
{
    x = min(x, __builtin_amdgcn_update_dpp(INT_MAX, x, 0x111, 0xf, 0xf, false)); // row_shr:1
    x = min(x, __builtin_amdgcn_update_dpp(INT_MAX, x, 0x112, 0xf, 0xf, false)); // row_shr:2
    x = min(x, __builtin_amdgcn_update_dpp(INT_MAX, x, 0x114, 0xf, 0xf, false)); // row_shr:4
    x = min(x, __builtin_amdgcn_update_dpp(INT_MAX, x, 0x118, 0xf, 0xf, false)); // row_shr:8
    x = min(x, __builtin_amdgcn_update_dpp(INT_MAX, x, 0x142, 0xa, 0xf, false)); // row_bcast:15 into rows 1,3
    x = min(x, __builtin_amdgcn_update_dpp(INT_MAX, x, 0x143, 0xc, 0xf, false)); // row_bcast:31 into rows 2,3
    return __builtin_amdgcn_readlane(x, 63);
}

__device__ __forceinline__ u64d lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// inclusive prefix sum over the TEAM lanes of a team (TEAM = 16: one DPP row; TEAM = 64: the wave) and the team's total in every
// lane, without LDS: DPP row shifts (lanes shifted in from outside the row read 0), rotations for the total
template <int TEAM> __device__ __forceinline__ int team_incl_scan(int x, int &total);
template <> __device__ __forceinline__ int team_incl_scan<16>(int x, int &total)
{
    int t = x;
    t += __builtin_amdgcn_update_dpp(0, t, 0x128, 0xf, 0xf, false); // row_ror:8
    t += __builtin_amdgcn_update_dpp(0, t, 0x124, 0xf, 0xf, false); // row_ror:4
    t += __builtin_amdgcn_update_dpp(0, t, 0x122, 0xf, 0xf, false); // row_ror:2
    t += __builtin_amdgcn_update_dpp(0, t, 0x121, 0xf, 0xf, false); // row_ror:1
    total = t;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false); // row_shr:8
    return x;
}
template <> __device__ __forceinline__ int team_incl_scan<64>(int x, int &total)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1,3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2,3
    total = __builtin_amdgcn_readlane(x, 63);
    return x;
}

// ------------------------------------------------------------------------------------------------
// ingest: host-style CSR arrays (p,j,x) -> (start,len,ent)
// ------------------------------------------------------------------------------------------------
// values are brought to the canonical balanced residue here, once: the kernels downstream rely on it (the streaming scatter
// writes an own entry as it stands; reference src/SpaSM.jl:955-958 builds CSRs that way, but a C caller may not)
__global__ void k_pack_entries(i64d nnz, ZpField F, const int *__restrict__ j, const int *__restrict__ x, int2 *__restrict__ ent)
{
    i64d k = (i64d)blockIdx.x * blockDim.x + threadIdx.x;
    const i64d stride = (i64d)gridDim.x * blockDim.x;
    for (; k < nnz; k += stride) ent[k] = make_int2(j[k], x ? zp_reduce(F, (int64_t)x[k]) : 1);
}

// Explicit zeros (an entry whose value is 0 mod p; reference src/SpaSM.jl:959, :979 drops them when a CSR is built, but a C caller may
// pass them) leave the rows at ingest, in place: "CSR with slack" lets a row shrink where it lies.  A zero that stayed could be
// elected as a pivot (its "inverse" is 0: a silently wrong U).  One team per row, stable.
template <int TEAM>
__global__ void k_drop_zeros(int n, const i64d *__restrict__ start, int *__restrict__ len, int2 *__restrict__ ent)
{
    const int tl = threadIdx.x % TEAM;
    const int i = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (i >= n) return;
    const i64d st = start[i];
    const int ln = len[i];
    int out = 0;
    for (int k0 = 0; k0 < ln; k0 += TEAM) {
        const int k = k0 + tl;
        int2 e = make_int2(0, 0);
        if (k < ln) e = ent[st + k];
        const bool keep = k < ln && e.y != 0;
        const u64d mk = team_ballot<TEAM>(keep);
        if (keep && out + __popcll(mk & ((1ull << tl) - 1ull)) != k) ent[st + out + __popcll(mk & ((1ull << tl) - 1ull))] = e; // (writes land at or before what was read)
        out += __popcll(mk);
    }
    if (tl == 0 && out != ln) len[i] = out;
}

// local row i of a shard is global row row_lo + i * row_stride (stride 1: a contiguous block; stride G: every G-th row,
// which balances the shards when the pivots concentrate at low row indices)
__global__ void k_pack_rows(int n, int row_lo, int row_stride, const i64d *__restrict__ p, i64d *__restrict__ start, int *__restrict__ len, int *__restrict__ orig)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    start[i] = p[i];
    len[i] = (int)(p[i + 1] - p[i]);
    orig[i] = row_lo + i * row_stride;
}

// leftmost column of every row (INT_MAX for an empty row); TEAM lanes per row
template <int TEAM>
__global__ void k_row_lead(int n, const i64d *__restrict__ start, const int *__restrict__ len, const int2 *__restrict__ ent, int *__restrict__ lead)
{
    const int tl = threadIdx.x % TEAM;
    const int i = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (i >= n) return;
    const i64d st = start[i];
    const int ln = len[i];
    int mn = INT_MAX;
    for (int k = tl; k < ln; k += TEAM) mn = min(mn, ent[st + k].x);
    for (int o = TEAM / 2; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o, TEAM));
    if (tl == 0) lead[i] = mn;
}

// ------------------------------------------------------------------------------------------------
// Faugere-Lachartre election: candidate of a row = its leftmost entry; per column the sparsest
// candidate wins, ties to the lowest row: atomicMin on (len << 32 | row).
// `row_base` makes the row id global when rows are sharded over devices.
// ------------------------------------------------------------------------------------------------
__global__ void k_elect(int n, int row_base, int row_stride, const int *__restrict__ len, const int *__restrict__ lead, u64d *__restrict__ best)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ln = len[i];
    if (ln <= 0) return;
    atomicMin(&best[lead[i]], ((u64d)(unsigned)ln << 32) | (u64d)(unsigned)(row_base + i * row_stride));
}

// election keys of two shards merged: per column the smaller (row length << 32 | row) key wins (what all-reduce(MIN) does between processes)
__global__ void k_min_u64(i64d n, u64d *__restrict__ dst, const u64d *__restrict__ src)
{
    const i64d i = (i64d)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const u64d a = dst[i], b = src[i]; dst[i] = b < a ? b : a; }
}

// "FL on columns" over the shards of one process: the m-word arrays of a step merged pairwise (what all-reduce MAX / SUM does between processes)
__global__ void k_max_i32(i64d n, int *__restrict__ dst, const int *__restrict__ src)
{
    const i64d i = (i64d)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int a = dst[i], b = src[i]; dst[i] = b > a ? b : a; }
}
__global__ void k_add_i32(i64d n, int *__restrict__ dst, const int *__restrict__ src)
{
    const i64d i = (i64d)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

__global__ void k_fill_u64(i64d n, u64d v, u64d *__restrict__ out)
{
    i64d i = (i64d)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

__global__ void k_col_flags(int m, const u64d *__restrict__ best, int *__restrict__ flag)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) flag[j] = best[j] != NO_BEST;
    if (j == m) flag[j] = 0;
}

// pivots are numbered by ascending pivot column: with leftmost pivots this is a topological order
// (a pivot row only has entries to the right of its pivot)
__global__ void k_col_assign(int m, const u64d *__restrict__ best, const int *__restrict__ scan, int *__restrict__ qinv_r,
                             int *__restrict__ pivrow, int *__restrict__ pivcol)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const u64d b = best[j];
    if (b != NO_BEST) {
        const int idx = scan[j];
        qinv_r[j] = idx;
        pivrow[idx] = (int)(unsigned)(b & 0xffffffffull);
        pivcol[idx] = j;
    } else {
        qinv_r[j] = -1;
    }
}

// ------------------------------------------------------------------------------------------------
// "FL on columns" (reference README.md:22, the second of the three searches spasm_pivots_extract_structural names, prototype
// src/SpaSM.jl:776-778): beyond the leftmost-entry pivots, a non-pivot row may take a pivot on a column that NO pivot row
// touches -- such a column is "open" -- because no elimination by the pivots found so far can reach that entry.
// libspasm walks the rows one after the other and closes the columns of each row it accepts; here all rows propose at once:
//   1. closed[c] = 1 for every column of every leftmost-pivot row;
//   2. occupancy histogram cnt[c] = rows (pivot rows excluded) holding column c, with a wave-level reduction: the lanes of a
//      wave that hold the same column are counted with one atomic (BASELINE north_star: "column-occupancy histograms via
//      wave-level reductions");
//   3. a non-pivot row proposes its open column of smallest occupancy (ties: leftmost) -- the pivot that will cause the least
//      fill-in; per column the sparsest proposing row wins (ties: lowest row), as in the leftmost election;
//   4. a winner is accepted when none of its OTHER columns was won by anybody: accepted rows then do not contain each other's
//      pivot columns, leftmost-pivot rows do not contain them either (they are open), so U stays triangular when the new
//      pivots are numbered before the leftmost ones;
//   5. the accepted rows become pivot rows, their columns are closed, and 2 - 4 run again (up to OPEN_PASSES times, until a pass
//      accepts nothing): rows that lost their column, or clashed, get another chance on what is still open.  Pivots of a later pass
//      are numbered before those of an earlier one (their rows may hold earlier pivot columns, never the other way round).
// Deterministic and order-free, so the CPU oracle can take the same pivots (oracle/spasm_oracle.c: fl_on_columns).
// ------------------------------------------------------------------------------------------------
#define OPEN_PASSES 4
__device__ __forceinline__ int shard_local(int g, int row_lo, int row_stride, int n);
// (rowsrc holds GLOBAL rows; a shard (row_lo, row_stride, n local rows) closes the columns of the pivot rows it holds, the
// reduction over the shards -- a maximum -- gives everybody all of them; one device: (0, 1, n))
template <int TEAM>
__global__ void k_close_cols(int npiv, const int *__restrict__ rowsrc, int row_lo, int row_stride, int n, const i64d *__restrict__ start, const int *__restrict__ len,
                             const int2 *__restrict__ ent, int *__restrict__ closed)
{
    const int tl = threadIdx.x % TEAM;
    const int idx = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (idx >= npiv) return;
    const int row = shard_local(rowsrc[idx], row_lo, row_stride, n);
    if (row < 0) return;
    const i64d st = start[row];
    const int ln = len[row];
    for (int k = tl; k < ln; k += TEAM) closed[ent[st + k].x] = 1;
}

// cnt[c] += number of (non-pivot) rows holding column c; one wave = 64 consecutive entries of the flattened row list, lanes with
// equal columns are merged before the atomic (readfirstlane "waterfall" over the distinct columns of the wave)
__global__ void k_col_histogram(int n, const int *__restrict__ is_piv, const i64d *__restrict__ start, const int *__restrict__ len,
                                const int2 *__restrict__ ent, int *__restrict__ cnt)
{
    constexpr int TEAM = 8; // lanes per row: rows of ~20 entries take three steps
    const int tl = threadIdx.x % TEAM;
    const int i = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    const bool live = i < n && !is_piv[i];
    const i64d st = live ? start[i] : 0;
    const int ln = live ? len[i] : 0;
    int steps = (ln + TEAM - 1) / TEAM;
    for (int o = 32; o > 0; o >>= 1) steps = max(steps, __shfl_xor(steps, o)); // (the wave walks together: ballots below)
    for (int s = 0; s < steps; s++) {
        const int k = s * TEAM + tl;
        const bool have = k < ln;
        const int c = have ? ent[st + k].x : -1;
        u64d todo = __ballot(have);
        while (todo) {
            const int c0 = __builtin_amdgcn_readlane(c, __ffsll((long long)todo) - 1);
            const u64d same = __ballot(have && c == c0);
            if ((threadIdx.x & 63) == __ffsll((long long)same) - 1) atomicAdd(&cnt[c0], __popcll(same));
            todo &= ~same;
        }
    }
}

// a non-pivot row proposes its open column of smallest occupancy (ties: leftmost): best2[c] = min (len << 32 | row)
template <int TEAM>
__global__ void k_propose_open(int n, int row_base, int row_stride, const int *__restrict__ is_piv, const i64d *__restrict__ start,
                               const int *__restrict__ len, const int2 *__restrict__ ent, const int *__restrict__ closed, const int *__restrict__ cnt,
                               int *__restrict__ prop, u64d *__restrict__ best2)
{
    const int tl = threadIdx.x % TEAM;
    const int i = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (i >= n) return;
    int choice = -1;
    if (!is_piv[i] && len[i] > 0) {
        const i64d st = start[i];
        const int ln = len[i];
        u64d key = ~0ull; // (occupancy << 32 | column)
        for (int k = tl; k < ln; k += TEAM) {
            const int c = ent[st + k].x;
            if (!closed[c]) key = min(key, ((u64d)(unsigned)cnt[c] << 32) | (u64d)(unsigned)c);
        }
        for (int o = TEAM / 2; o > 0; o >>= 1) key = min(key, (u64d)__shfl_xor((long long)key, o, TEAM));
        if (key != ~0ull) {
            choice = (int)(unsigned)(key & 0xffffffffull);
            if (tl == 0) atomicMin(&best2[choice], ((u64d)(unsigned)ln << 32) | (u64d)(unsigned)(row_base + i * row_stride));
        }
    }
    if (tl == 0) prop[i] = choice;
}

// the winner of column c is accepted (best[c] takes it over) when no other column of its row was won by anybody
template <int TEAM>
__global__ void k_accept_open(int n, int row_base, int row_stride, const i64d *__restrict__ start, const int *__restrict__ len,
                              const int2 *__restrict__ ent, const int *__restrict__ prop, const u64d *__restrict__ best2, int *__restrict__ newflag)
{
    const int tl = threadIdx.x % TEAM;
    const int i = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (i >= n) return;
    const int c = prop[i];
    if (c < 0) return;
    if ((int)(unsigned)(best2[c] & 0xffffffffull) != row_base + i * row_stride) return; // another row won the column
    const i64d st = start[i];
    const int ln = len[i];
    bool clash = false;
    for (int k = tl; k < ln; k += TEAM) {
        const int c2 = ent[st + k].x;
        clash |= c2 != c && best2[c2] != NO_BEST;
    }
    if (team_ballot<TEAM>(clash) == 0 && tl == 0) newflag[c] = 1;
}

// numbering with the open-column pivots first (ascending column), then the leftmost pivots (ascending column)
__global__ void k_col_flags2(int m, const u64d *__restrict__ best, const int *__restrict__ newflag, int *__restrict__ flflag)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) flflag[j] = best[j] != NO_BEST;
    if (j == m) flflag[j] = 0;
    (void)newflag;
}
// the pivots a pass of the search accepted: their pass, their rank inside it (ascending column), their row; the rows become pivot
// rows for the next pass (is_piv) and are listed (newrows) so that their columns can be closed
__global__ void k_record_open(int m, int pass, const int *__restrict__ newflag, const int *__restrict__ newscan, const u64d *__restrict__ best2,
                              int *__restrict__ newpass, int *__restrict__ newidx, int *__restrict__ newrow_of_col, int *__restrict__ newrows,
                              int *__restrict__ is_piv, int row_lo, int row_stride, int n)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m || !newflag[j]) return;
    const int row = (int)(unsigned)(best2[j] & 0xffffffffull); // (a GLOBAL row)
    newpass[j] = pass;
    newidx[j] = newscan[j];
    newrow_of_col[j] = row;
    newrows[newscan[j]] = row;
    const int loc = shard_local(row, row_lo, row_stride, n);
    if (loc >= 0) is_piv[loc] = 1;
}

// numbering: the open-column pivots of the LAST pass first, then the earlier passes, each by ascending column, then the leftmost
// pivots by ascending column.  base.x/y/z/w = first index of pass 1 / 2 / 3 / 4; nnew = all open-column pivots.
__global__ void k_col_assign2(int m, int nnew, int4 base, const u64d *__restrict__ best, const int *__restrict__ newpass, const int *__restrict__ newidx,
                              const int *__restrict__ newrow_of_col, const int *__restrict__ flscan, int *__restrict__ qinv_r, int *__restrict__ pivrow,
                              int *__restrict__ pivcol)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    int idx = -1, row = -1;
    const int ps = newpass[j];
    if (ps > 0) {
        idx = (ps == 1 ? base.x : ps == 2 ? base.y : ps == 3 ? base.z : base.w) + newidx[j];
        row = newrow_of_col[j];
    } else if (best[j] != NO_BEST) {
        idx = nnew + flscan[j];
        row = (int)(unsigned)(best[j] & 0xffffffffull);
    }
    qinv_r[j] = idx;
    if (idx >= 0) {
        pivrow[idx] = row;
        pivcol[idx] = j;
    }
}

// local index of global row g in the shard (row_lo, row_stride, n rows), or -1
__device__ __forceinline__ int shard_local(int g, int row_lo, int row_stride, int n)
{
    const int r = g - row_lo;
    if (r < 0 || r % row_stride != 0) return -1;
    const int q = r / row_stride;
    return q < n ? q : -1;
}

__global__ void k_mark_rows(int npiv, int row_lo, int row_stride, int n, const int *__restrict__ pivrow, int *__restrict__ is_piv)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npiv) return;
    const int r = shard_local(pivrow[idx], row_lo, row_stride, n);
    if (r >= 0) is_piv[r] = 1;
}

__global__ void k_row_flags(int n, int lo, int hi, int step, const int *__restrict__ is_piv, const int *__restrict__ len, int *__restrict__ flag)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i >= lo && i < hi && (i - lo) % step == 0 && !is_piv[i] && len[i] > 0) ? 1 : 0;
    if (i == n) flag[i] = 0;
}

__global__ void k_compact(int n, const int *__restrict__ flag, const int *__restrict__ scan, int *__restrict__ list)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) list[scan[i]] = i;
}

// length of each pivot row (i64 for the scan); rows are addressed through `rowsrc` = a local row
__global__ void k_gather_len(int npiv, const int *__restrict__ rowsrc, const int *__restrict__ len, i64d *__restrict__ out)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < npiv) out[idx] = len[rowsrc[idx]];
    if (idx == npiv) out[idx] = 0;
}

// ------------------------------------------------------------------------------------------------
// U build: one TEAM per pivot.  Scales the row by pivot^-1 ("pivots in U are all equal to 1",
// reference src/SpaSM.jl:712), writes the full row (column space, for the returned U) and the split
// copies: UPP = entries on OTHER pivot columns as {pivot index, val}; UPN = entries on non-pivot
// columns as {col, val}.
// ------------------------------------------------------------------------------------------------
template <int TEAM>
__global__ void k_build_U(int npiv, ZpField F, const int *__restrict__ rowsrc, const int *__restrict__ pivcol,
                          const i64d *__restrict__ start, const int *__restrict__ len, const int2 *__restrict__ ent,
                          const int *__restrict__ qinv_r, const i64d *__restrict__ uoff,
                          int2 *__restrict__ Ufull, int2 *__restrict__ UPP, int2 *__restrict__ UPN, UHdr *__restrict__ uhdr,
                          int *__restrict__ pivval)
{
    const int tl = threadIdx.x % TEAM;
    const int idx = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (idx >= npiv) return;
    const int row = rowsrc[idx];
    const int pc = pivcol[idx];
    const i64d st = start[row];
    const int ln = len[row];
    const i64d uo = uoff[idx];
    int pv = 0;
    for (int k = tl; k < ln; k += TEAM) {
        const int2 e = ent[st + k];
        if (e.x == pc) pv = e.y;
    }
    for (int o = TEAM / 2; o > 0; o >>= 1) pv += __shfl_xor(pv, o, TEAM); // exactly one lane holds it
    const int inv = zp_inverse(F, pv);
    int npp = 0, npn = 0;
    const u64d below = (1ull << tl) - 1ull;
    for (int k0 = 0; k0 < ln; k0 += TEAM) {
        const int k = k0 + tl;
        const bool valid = k < ln;
        int2 e = make_int2(-1, 0);
        int v = 0, q = -1;
        if (valid) {
            e = ent[st + k];
            v = (e.x == pc) ? 1 : zp_mul(F, inv, e.y);
            Ufull[uo + k] = make_int2(e.x, v);
            if (e.x != pc) q = qinv_r[e.x];
        }
        const bool isPP = valid && e.x != pc && q >= 0;
        const bool isPN = valid && e.x != pc && q < 0;
        const u64d mPP = team_ballot<TEAM>(isPP), mPN = team_ballot<TEAM>(isPN);
        if (isPP) UPP[uo + npp + __popcll(mPP & below)] = make_int2(q, v);
        if (isPN) UPN[uo + npn + __popcll(mPN & below)] = make_int2(e.x, v);
        npp += __popcll(mPP);
        npn += __popcll(mPN);
    }
    if (tl == 0) {
        UHdr h;
        h.off = (unsigned)uo;
        h.npp = npp;
        h.npn = npn;
        h.len = ln;
        uhdr[idx] = h;
        if (pivval) pivval[idx] = pv; // the entry of L on the diagonal: row = pv * (row of U) (reference src/SpaSM.jl:705-712)
    }
}

// ------------------------------------------------------------------------------------------------
// SOLVE.  One TEAM of lanes per non-pivot row.  The team keeps, in LDS, a list of (pivot index,
// value) sorted by pivot index: the prefix [0,head) is final (the multipliers x_b, i.e. the row of
// L, reference src/SpaSM.jl:705-707), the suffix is pending.  Pivot indices are a topological
// order, so popping the smallest pending index is always legal and new indices land in the suffix.
// Output per row: the list in a bump-allocated pool, its length, and the bound of the Schur row
// (its non-pivot entries + sum npn over applications), capped by the number of free columns.
// ------------------------------------------------------------------------------------------------
struct SolveArgs {
    int nrows;                 // rows to process
    const int *rows;           // local row of each (NULL: identity)
    const int *self_idx;       // per processed row: pivot index to ignore (kernel basis), or NULL
    const int *retry;          // when non-NULL: list of row slots to process instead of 0..nrows-1
    const int *retry_count;
    const i64d *start;
    const int *len;
    const int2 *ent;
    const int *qinv_r;
    const UHdr *uhdr;
    const int2 *UPP;
    int4 *Lpool;               // multiplier records {stream position of the pivot row's entries (combine kernel only), multiplier,
                               // offset of the pivot row in UPN, npn}
    int *Lidx;                 // when non-NULL: pivot index of every record (kernel basis, triangular solve, L factor)
    int2 *Lpool2;              // when non-NULL the list is published as {pivot index, value} instead (rows of Uinv)
    u64d lpool_cap;            // entries per pool region
    u64d *pool_ctr;            // NPOOL bump counters, POOL_STRIDE words apart
    int npool;                 // regions in use: min(NPOOL, rows / 4), so that a handful of rows can use the whole pool
    i64d *Lstart;
    int *Llen;
    i64d *bound;
    int free_cols;             // columns without a pivot after this round: cap of any Schur row
    int *overflow_list;        // row slots whose reach overflowed the LDS list (NULL: only count)
    int *overflow_count;
    RoundCounters *ctr;
    ZpField F;
};

// LDS-qualified volatile pointer: the team's lanes exchange data through these arrays inside one wave,
// so every access must be a real ds_read/ds_write in program order (and must stay in address space 3).
typedef __attribute__((address_space(3))) volatile int lds_vint;

template <int TEAM, int CAP>
__device__ __forceinline__ bool sorted_insert(lds_vint *key, lds_vint *val, int head, int &cnt, int q, int v,
                                              const ZpField &F, int tl)
{
    // rank of q among the pending keys [head,cnt)
    int pos = head, found = -1;
    for (int b = head; b < cnt; b += TEAM) {
        const int i = b + tl;
        const int kk = (i < cnt) ? key[i] : INT_MAX;
        const u64d mlt = team_ballot<TEAM>(kk < q);
        const u64d meq = team_ballot<TEAM>(kk == q);
        pos += __popcll(mlt);
        if (meq) found = b + __ffsll((long long)meq) - 1;
    }
    if (found >= 0) {
        if (tl == 0) val[found] = zp_add(F, val[found], v);
        return true;
    }
    if (cnt >= CAP) return false;
    // shift [pos,cnt) up by one, highest chunk first
    for (int hi = cnt; hi > pos; hi -= TEAM) {
        const int i = hi - 1 - tl;
        int kk = 0, vv = 0;
        if (i >= pos) { kk = key[i]; vv = val[i]; }
        if (i >= pos) { key[i + 1] = kk; val[i + 1] = vv; }
    }
    if (tl == 0) { key[pos] = q; val[pos] = v; }
    cnt++;
    return true;
}

template <int TEAM, int CAP, int TPB>
__global__ __launch_bounds__(TPB) void k_solve(SolveArgs a)
{
    constexpr int TEAMS = TPB / TEAM;
    __shared__ int s_key[TEAMS * CAP];
    __shared__ int s_val[TEAMS * CAP];
    const int team = threadIdx.x / TEAM;
    const int tl = threadIdx.x % TEAM;
    lds_vint *key = (lds_vint *)(s_key + team * CAP);
    lds_vint *val = (lds_vint *)(s_val + team * CAP);
    const ZpField F = a.F;

    int total = a.nrows;
    if (a.retry) total = *a.retry_count;
    u64d c_app = 0, c_red = 0, c_seg = 0;
    for (i64d gteam = (i64d)blockIdx.x * TEAMS + team; gteam < total; gteam += (i64d)gridDim.x * TEAMS) {
        const int t = a.retry ? a.retry[gteam] : (int)gteam;
        const int row = a.rows ? a.rows[t] : t;
        const int self = a.self_idx ? a.self_idx[t] : -1;
        const i64d st = a.start[row];
        const int ln = a.len[row];
        int cnt = 0, nN = 0;
        bool ok = true;
        // ---- the row's own entries: pivot columns into the list, the others only counted
        for (int k0 = 0; k0 < ln && ok; k0 += TEAM) {
            const int k = k0 + tl;
            const bool valid = k < ln;
            int q = -1, v = 0;
            if (valid) {
                const int2 e = a.ent[st + k];
                q = a.qinv_r[e.x];
                v = e.y;
            }
            const bool isP = valid && q >= 0 && q != self;
            const bool isN = valid && q < 0;
            u64d mP = team_ballot<TEAM>(isP);
            nN += __popcll(team_ballot<TEAM>(isN));
            while (mP) {
                const int src = __ffsll((long long)mP) - 1;
                mP &= mP - 1;
                const int qq = __shfl(q, src, TEAM);
                const int vv = __shfl(v, src, TEAM);
                if (!sorted_insert<TEAM, CAP>(key, val, 0, cnt, qq, vv, F, tl)) { ok = false; break; }
            }
        }
        // ---- eliminate in pivot-index order
        int head = 0;
        i64d bound = nN;
        u64d r_app = 0, r_red = (u64d)ln, r_seg = 1;
        while (ok && head < cnt) {
            const int idx = key[head];
            const int mult = val[head];
            head++;
            if (mult == 0) continue; // cancelled: nothing to eliminate
            const UHdr h = a.uhdr[idx];
            r_app += 1;
            r_red += (u64d)h.len;
            r_seg += 1;
            bound += h.npn;
            const int nm = zp_neg(F, mult);
            for (int k0 = 0; k0 < h.npp && ok; k0 += TEAM) {
                const int k = k0 + tl;
                int2 e = make_int2(0, 0);
                if (k < h.npp) e = a.UPP[(i64d)h.off + k];
                const int nin = min(TEAM, h.npp - k0);
                for (int s = 0; s < nin; s++) {
                    const int qq = __shfl(e.x, s, TEAM);
                    const int vv = __shfl(e.y, s, TEAM);
                    if (!sorted_insert<TEAM, CAP>(key, val, head, cnt, qq, zp_mul(F, nm, vv), F, tl)) { ok = false; break; }
                }
            }
        }
        if (!ok) {
            // reach too large for this class: hand the row to the next class
            if (tl == 0) {
                a.Llen[t] = -1;
                a.Lstart[t] = 0;
                a.bound[t] = 0;
                const int pos = atomicAdd(a.overflow_count, 1);
                if (a.overflow_list) a.overflow_list[pos] = t;
            }
        } else {
            // ---- publish the multipliers
            u64d base = 0;
            if (tl == 0) base = pool_alloc(a.pool_ctr, a.lpool_cap, (u64d)cnt, a.npool);
            base = __shfl(base, 0, TEAM);
            if (base == ~0ull) {
                if (tl == 0) { atomicAdd(&ctr_shard(a.ctr)->lpool_overflow, 1); a.Llen[t] = 0; a.Lstart[t] = 0; a.bound[t] = 0; }
            } else {
                if (a.Lpool2) {
                    for (int i = tl; i < cnt; i += TEAM) a.Lpool2[base + i] = make_int2(key[i], val[i]);
                } else {
                    for (int i = tl; i < cnt; i += TEAM) {
                        const int kk = key[i], vv = val[i];
                        UHdr h; h.off = 0; h.npn = 0;
                        if (vv != 0) h = a.uhdr[kk];
                        a.Lpool[base + i] = make_int4(0, vv, (int)h.off, h.npn);
                        if (a.Lidx) a.Lidx[base + i] = kk;
                    }
                }
                if (tl == 0) {
                    a.Lstart[t] = (i64d)base;
                    a.Llen[t] = cnt;
                    a.bound[t] = bound < (i64d)a.free_cols ? bound : (i64d)a.free_cols;
                    c_app += r_app; c_red += r_red; c_seg += r_seg;
                }
            }
        }
    }
    // one atomic per wave per counter
    for (int o = 32; o > 0; o >>= 1) {
        c_app += __shfl_xor(c_app, o);
        c_red += __shfl_xor(c_red, o);
        c_seg += __shfl_xor(c_seg, o);
    }
    if ((threadIdx.x & 63) == 0 && (c_app | c_red | c_seg)) {
        atomicAdd(&ctr_shard(a.ctr)->applications, c_app);
        atomicAdd(&ctr_shard(a.ctr)->nnz_reduced, c_red);
        atomicAdd(&ctr_shard(a.ctr)->segments, c_seg);
    }
}

// ------------------------------------------------------------------------------------------------
// binning of rows by the size of the hash table their Schur row needs
// ------------------------------------------------------------------------------------------------
#define NCLASS 16
#define NHASHMAX 8     // classes 0..6: LDS hash tables, 7: the global-memory last resort
#define NSTREAM0 8     // classes 8..14: the streaming twins of classes 0..6
// everything the scatter kernel needs to know about one row, gathered once by the binning pass so that the
// kernel's per-row dependent-load chain is: descriptor (prefetched) -> multiplier records -> pivot row entries
struct __attribute__((aligned(16))) RowDesc {
    i64d ent_start;   // the row's own entries
    i64d l_start;     // its multiplier records
    i64d s_start;     // slot of its Schur row
    int len;
    int llen;
    int t;            // row slot (index of the Schur row)
    int bound;        // bound of the Schur row = length of the row's entry stream (streaming kernel: exactly nN + sum npn)
    long long pmask;  // bit k set <=> own entry k sits on a pivot column (from the solve kernel); < 0: not available
};

// A descriptor fetched with VECTOR loads (every lane reads the same 48 bytes): a scalar load would sit on
// lgkmcnt, and every LDS wait of the kernel is lgkmcnt(0) -- the prefetch would be waited for at once.
struct DescRegs { int4 a, b, c; };
__device__ __forceinline__ DescRegs desc_load(const RowDesc *p)
{
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z)); // opaque zero: keeps the address in a VGPR
    const int4 *q = (const int4 *)((const char *)p + z);
    DescRegs r;
    r.a = q[0];
    r.b = q[1];
    r.c = q[2];
    return r;
}
__device__ __forceinline__ RowDesc desc_unpack(const DescRegs &r)
{
    RowDesc d;
    d.ent_start = (i64d)(((u64d)(unsigned)__builtin_amdgcn_readfirstlane(r.a.y) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(r.a.x));
    d.l_start = (i64d)(((u64d)(unsigned)__builtin_amdgcn_readfirstlane(r.a.w) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(r.a.z));
    d.s_start = (i64d)(((u64d)(unsigned)__builtin_amdgcn_readfirstlane(r.b.y) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(r.b.x));
    d.len = __builtin_amdgcn_readfirstlane(r.b.z);
    d.llen = __builtin_amdgcn_readfirstlane(r.b.w);
    d.t = __builtin_amdgcn_readfirstlane(r.c.x);
    d.bound = __builtin_amdgcn_readfirstlane(r.c.y);
    d.pmask = (long long)(((u64d)(unsigned)__builtin_amdgcn_readfirstlane(r.c.w) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(r.c.z));
    return d;
}

struct BinArgs {
    int nrows;
    const i64d *bound;
    const int *Llen;
    const int *rows;         // local row of each slot (NULL: identity)
    const i64d *start;
    const int *len;
    const int *orig;
    const i64d *Lstart;
    const i64d *sstart;
    const long long *pmask;  // per row slot, from the solve kernel (NULL: none)
    const int *sflag;        // per row slot: 1 = the records carry stream positions and the bound is exact (streaming kernel)
    int stream_classes;      // hash classes 0..stream_classes-1 have a streaming twin (class + NSTREAM0); 0 = streaming off
    int *Sorig;              // out: originating row of every row slot
    int *fixcnt;             // out: 0 for every row slot (duplicates the streaming kernels leave for k_stream_fix)
    RowDesc *desc;           // [NCLASS][nrows]
    i64d cap[NCLASS];        // class c takes rows with bound <= cap[c]; the last class takes the rest
    int *class_count;        // [NCLASS]
    int *class_list;         // [NCLASS][nrows]
};

__global__ __launch_bounds__(256) void k_bin(BinArgs a)
{
    constexpr int PER = 4; // rows per thread
    __shared__ int s_cnt[NCLASS];
    __shared__ int s_base[NCLASS];
    if (threadIdx.x < NCLASS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int cls[PER], pos[PER];
    const int base = blockIdx.x * (256 * PER);
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int t = base + i * 256 + threadIdx.x;
        cls[i] = -1;
        pos[i] = 0;
        if (t < a.nrows && a.Llen[t] >= 0) {
            const i64d b = a.bound[t];
            int c = NHASHMAX - 1;
            for (int k = NHASHMAX - 2; k >= 0; k--) if (b <= a.cap[k]) c = k;
            if (c < a.stream_classes && a.sflag && a.sflag[t]) c += NSTREAM0;
            cls[i] = c;
            pos[i] = atomicAdd(&s_cnt[c], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < NCLASS && s_cnt[threadIdx.x] > 0) s_base[threadIdx.x] = atomicAdd(&a.class_count[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; i++)
        if (cls[i] >= 0) {
            const int t = base + i * 256 + threadIdx.x;
            const size_t slot = (size_t)cls[i] * a.nrows + s_base[cls[i]] + pos[i];
            a.class_list[slot] = t;
            const int row = a.rows ? a.rows[t] : t;
            RowDesc d;
            d.ent_start = a.start[row];
            d.l_start = a.Lstart[t];
            d.s_start = a.sstart[t];
            // (a row planned along W has its own entries among its chunks: the scatter kernels must not take them from the matrix again)
            d.len = (a.sflag && a.sflag[t]) ? 0 : a.len[row];
            d.llen = a.Llen[t];
            d.t = t;
            d.bound = (int)(a.bound[t] < (i64d)INT_MAX ? a.bound[t] : (i64d)INT_MAX);
            d.pmask = a.pmask ? a.pmask[t] : -1;
            a.desc[slot] = d;
        }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int t = base + i * 256 + threadIdx.x;
        if (t < a.nrows) {
            a.Sorig[t] = a.orig[a.rows ? a.rows[t] : t];
            a.fixcnt[t] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// SCATTER.  One workgroup per non-pivot row (grid-stride over the rows of one size class).
// An open-addressing hash table in LDS maps column -> lazily reduced accumulator.  The row's own
// non-pivot entries are inserted, then every pivot row of its multiplier list is streamed from
// UPN by a 16-lane group (one 128-byte line per step) and accumulated with LDS atomics; finally the
// table is swept, reduced to balanced residues, compacted and written as the Schur row.
// ------------------------------------------------------------------------------------------------
struct ScatterArgs {
    const int *class_count;    // rows in this class
    const RowDesc *desc;       // their descriptors
    const int2 *ent;
    const int *qinv_r;
    const UHdr *uhdr;
    const int2 *UPN;
    const int4 *Lpool;
    int2 *Sent;
    int *Slen;
    int *Slead;
    int *Sorig;
    RoundCounters *ctr;
    int cls;                   // index of this size class (for the per-class counters)
    u64d *stamps;              // diagnostic build only: [class][NSTAMP] cycle sums + [class][NSTAMP] wave counts
    int dbg;                   // TIMING ABLATIONS ONLY (env SPASM_DBG, results are wrong when non-zero):
                               // 1 = no Schur stores, 2 = no accumulation of pivot rows, 4 = sweep only resets, 8 = no pivot-row loads
    ZpField F;
};

// ------------------------------------------------------------------------------------------------
// LDS hash table of one Schur row: open addressing with double hashing (the probe step is odd, so
// it visits every slot of the power-of-two table).  Column array + accumulator array (i32 lazy for
// small primes, i64 otherwise).
// ------------------------------------------------------------------------------------------------
template <int LOGT, bool SMALL> struct RowTable;

template <int LOGT> struct RowTable<LOGT, true> {
    // keys and accumulators in SEPARATE arrays: with interleaved {key,val} slots every CAS lands on an even bank and
    // every add on an odd one, i.e. half the banks per instruction (tools/lds_bank_bench.hip: 2.7 vs 3.8 inserts/clk/CU)
    static constexpr int T = 1 << LOGT;
    static constexpr size_t BYTES = (size_t)T * 8;
    int *key;
    int *val;
    __device__ __forceinline__ void bind(unsigned char *base) { key = (int *)base; val = key + T; }
    __device__ __forceinline__ int *keyp(unsigned h) { return &key[h]; }
    __device__ __forceinline__ void add(unsigned h, int v) { atomicAdd(&val[h], v); }
    __device__ __forceinline__ void clear(int s) { key[s] = EMPTY_KEY; val[s] = 0; }
    __device__ __forceinline__ void read(int s, int &k, int &acc) { k = key[s]; acc = val[s]; }
};

template <int LOGT> struct RowTable<LOGT, false> {
    static constexpr int T = 1 << LOGT;
    static constexpr size_t BYTES = (size_t)T * 12;
    long long *val;
    int *key;
    __device__ __forceinline__ void bind(unsigned char *base) { val = (long long *)base; key = (int *)(base + (size_t)T * 8); }
    __device__ __forceinline__ int *keyp(unsigned h) { return &key[h]; }
    __device__ __forceinline__ void add(unsigned h, long long v) { atomicAdd((u64d *)&val[h], (u64d)v); }
    __device__ __forceinline__ void clear(int s) { key[s] = EMPTY_KEY; val[s] = 0; }
    __device__ __forceinline__ void read(int s, int &k, long long &acc) { k = key[s]; acc = val[s]; }
};

template <int LOGT> __device__ __forceinline__ void hash2(int c, unsigned &h, unsigned &step)
{
    const unsigned x = (unsigned)c * 0x9E3779B1u;
    h = x >> (32 - LOGT);
    step = ((x >> 7) & ((1u << LOGT) - 1)) | 1u;
}

// N entries of a lane inserted together: all pending CAS of a round are in flight at once, so the lane
// pays about max(probe length) LDS round trips instead of their sum
template <int LOGT, int N, bool SMALL>
__device__ __forceinline__ void table_add_n(RowTable<LOGT, SMALL> &tab, const int (&c)[N], const typename ZpAcc<SMALL>::type (&v)[N],
                                            unsigned valid, RoundCounters *ctr)
{
    constexpr unsigned T = 1u << LOGT;
    unsigned h[N], st[N];
#pragma unroll
    for (int j = 0; j < N; j++) hash2<LOGT>(c[j], h[j], st[j]);
    unsigned pending = valid;
    for (unsigned round = 0; pending != 0 && round < T; round++) {
        int k[N];
#pragma unroll
        for (int j = 0; j < N; j++) {
            k[j] = 0;
            if (pending & (1u << j)) k[j] = atomicCAS(tab.keyp(h[j]), EMPTY_KEY, c[j]);
        }
#pragma unroll
        for (int j = 0; j < N; j++) {
            if (pending & (1u << j)) {
                if (k[j] == EMPTY_KEY || k[j] == c[j]) {
                    tab.add(h[j], v[j]);
                    pending &= ~(1u << j);
                } else {
                    h[j] = (h[j] + st[j]) & (T - 1);
                }
            }
        }
    }
    if (pending) atomicAdd(&ctr_shard(ctr)->scatter_overflow, 1);
}

template <int LOGT, bool SMALL>
__device__ __forceinline__ void table_add(RowTable<LOGT, SMALL> &tab, int c, typename ZpAcc<SMALL>::type v, RoundCounters *ctr)
{
    const int cc[1] = {c};
    const typename ZpAcc<SMALL>::type vv[1] = {v};
    table_add_n<LOGT, 1, SMALL>(tab, cc, vv, 1u, ctr);
}

// ------------------------------------------------------------------------------------------------
// First probe for everyone, retries together.  A loop "probe until all 64 lanes x N entries are in" runs as long as the
// unluckiest entry and does so with a handful of live lanes after the first trip (about 3/4 of the entries land on
// their first probe).  Instead every entry gets ONE probe, and the losers are appended to a list of the wave in LDS;
// the list is drained 64 entries at a time by a one-entry-per-lane probe loop whose lanes all have work.
// ------------------------------------------------------------------------------------------------
constexpr int RCAP = 192; // entries of a wave's retry list = the worst case of one table_try_n<3>

template <bool SMALL> struct RetryList;
template <> struct RetryList<true> {
    static constexpr size_t BYTES = (size_t)RCAP * 8;
    int2 *buf;
    int cnt; // wave-uniform
    __device__ __forceinline__ void bind(unsigned char *p) { buf = (int2 *)p; cnt = 0; }
    // |v| <= 0.51 p < 2^15.1 and h < 2^14 (largest table) share a word
    __device__ __forceinline__ void put(int i, int c, int v, unsigned h) { buf[i] = make_int2(c, (int)(((unsigned)v << 14) | h)); }
    __device__ __forceinline__ void get(int i, int &c, int &v, unsigned &h) const { const int2 e = buf[i]; c = e.x; v = e.y >> 14; h = (unsigned)e.y & 0x3fffu; }
};
template <> struct RetryList<false> {
    static constexpr size_t BYTES = (size_t)RCAP * 16;
    int4 *buf;
    int cnt;
    __device__ __forceinline__ void bind(unsigned char *p) { buf = (int4 *)p; cnt = 0; }
    __device__ __forceinline__ void put(int i, int c, long long v, unsigned h) { buf[i] = make_int4(c, (int)h, (int)(unsigned)(v & 0xffffffffll), (int)(v >> 32)); }
    __device__ __forceinline__ void get(int i, int &c, long long &v, unsigned &h) const
    {
        const int4 e = buf[i];
        c = e.x; h = (unsigned)e.y; v = ((long long)e.w << 32) | (long long)(unsigned)e.z;
    }
};

// empty the list: one entry per lane, probing on from where its first probe left off
template <int LOGT, bool SMALL>
__device__ __forceinline__ void retry_drain(RowTable<LOGT, SMALL> &tab, RetryList<SMALL> &rl, RoundCounters *ctr)
{
    typedef typename ZpAcc<SMALL>::type Acc;
    constexpr unsigned T = 1u << LOGT;
    const int lane = threadIdx.x & 63;
    for (int b = 0; b < rl.cnt; b += 64) {
        bool pending = b + lane < rl.cnt;
        int c = 0;
        Acc v = 0;
        unsigned h = 0, st = 1;
        if (pending) {
            rl.get(b + lane, c, v, h);
            unsigned h0;
            hash2<LOGT>(c, h0, st);
        }
        for (unsigned round = 0; round < T && __ballot(pending) != 0; round++) {
            if (pending) {
                const int k = atomicCAS(tab.keyp(h), EMPTY_KEY, c);
                if (k == EMPTY_KEY || k == c) { tab.add(h, v); pending = false; }
                else h = (h + st) & (T - 1);
            }
        }
        if (pending) atomicAdd(&ctr_shard(ctr)->scatter_overflow, 1);
    }
    rl.cnt = 0;
}

// one probe for each of the N entries of this lane; entries that met a foreign key go to the retry list
template <int LOGT, int N, bool SMALL>
__device__ __forceinline__ void table_try_n(RowTable<LOGT, SMALL> &tab, RetryList<SMALL> &rl, const int (&c)[N],
                                            const typename ZpAcc<SMALL>::type (&v)[N], unsigned valid, RoundCounters *ctr)
{
    static_assert(N * 64 <= RCAP, "the retry list must hold one batch");
    constexpr unsigned T = 1u << LOGT;
    unsigned h[N], st[N];
    int k[N];
#pragma unroll
    for (int j = 0; j < N; j++) hash2<LOGT>(c[j], h[j], st[j]);
#pragma unroll
    for (int j = 0; j < N; j++) {
        k[j] = c[j];
        if (valid & (1u << j)) k[j] = atomicCAS(tab.keyp(h[j]), EMPTY_KEY, c[j]);
    }
    u64d fm[N];
    bool fail[N];
    int nfail = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
        const bool ok = (k[j] == EMPTY_KEY) || (k[j] == c[j]); // lanes without an entry read back c[j]: "ok", nothing to add
        if (ok && (valid & (1u << j))) tab.add(h[j], v[j]);
        fail[j] = !ok;
        fm[j] = __ballot(fail[j]);
        nfail += __popcll(fm[j]);
    }
    if (nfail == 0) return;
    if (rl.cnt + nfail > RCAP) retry_drain<LOGT, SMALL>(tab, rl, ctr);
    int pos = rl.cnt;
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (fail[j]) rl.put(pos + __popcll(fm[j] & lanemask_lt()), c[j], v[j], (h[j] + st[j]) & (T - 1));
        pos += __popcll(fm[j]);
    }
    rl.cnt = pos;
}

// final reduction of a lazy accumulator to the balanced residue
template <bool SMALL> __device__ __forceinline__ int acc_reduce(const ZpField &F, typename ZpAcc<SMALL>::type acc);
template <> __device__ __forceinline__ int acc_reduce<true>(const ZpField &F, int acc)
{
    // |acc| < 2^31 and p < 2^16: the float quotient is within 1 of the exact one, two corrections finish
    const int p = (int)F.p, hp = (int)F.halfp, mhp = (int)F.mhalfp;
    int r = acc - __float2int_rn((float)acc * F.finvp) * p;
    if (r > hp) r -= p; else if (r < mhp) r += p;
    if (r > hp) r -= p; else if (r < mhp) r += p;
    return r;
}
template <> __device__ __forceinline__ int acc_reduce<false>(const ZpField &F, long long acc) { return zp_reduce(F, acc); }

// the same for an accumulator that summed at most 2^20 lazy terms (|term| <= 0.51 p): the quotient is then below 2^20,
// a 24-bit operand, its float estimate is off by < 3 * 2^-24 * 2^20 < 0.2 before rounding, so |r| < 0.7 p and ONE
// correction lands in the balanced range.  Every row of the LDS classes qualifies (their multiplier lists are far shorter).
template <bool SMALL> __device__ __forceinline__ int acc_reduce_short(const ZpField &F, typename ZpAcc<SMALL>::type acc);
template <> __device__ __forceinline__ int acc_reduce_short<true>(const ZpField &F, int acc)
{
    // u = r - mhalfp lies in [0, p) when r is balanced; of u - p, u, u + p exactly one does, and it is the unsigned minimum
    const int p = (int)F.p, mhp = (int)F.mhalfp;
    const unsigned u = (unsigned)(zp_small_lazy(acc, -F.finvp, p) - mhp);
    return (int)min(min(u, u - (unsigned)p), u + (unsigned)p) + mhp;
}
template <> __device__ __forceinline__ int acc_reduce_short<false>(const ZpField &F, long long acc) { return zp_reduce(F, acc); }

// Diagnostic build only (-DSPASM_STAMPS): per-phase cycle sums of the scatter kernel, written to a buffer of
// their own (ScatterArgs::stamps) that nothing else reads.  Never compiled into the shipped library.
#if defined(SPASM_STAMPS) || defined(SPASM_ABLATE)
#define SCATTER_DBG(a, bit) ((a).dbg & (bit)) // timing ablations (env SPASM_DBG), diagnostic builds only
#else
#define SCATTER_DBG(a, bit) false
#endif
#ifdef SPASM_STAMPS
#define NSTAMP 8
__device__ __forceinline__ u64d stamp_now()
{
    u64d t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP(i) do { const u64d _n = stamp_now(); st_sum[i] += _n - st_last; st_last = _n; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// TPR = threads cooperating on one row: 64 (a wave per row, WPB independent rows per workgroup, no
//       barriers) or WPB*64 (the whole workgroup on one row, LDS-only barriers).
// MAXR = rounds of pivot rows whose loads are all issued before any accumulation (registers);
//       round r hands pivot row gg + r*NG of the row's multiplier list to the 8-lane group gg.
// Software pipeline across the rows of a team: descriptor two rows ahead, multiplier records and own
// entries one row ahead (issued before the sweep), so a row exposes one global-load latency.
// MINW = waves per SIMD the register allocator must leave room for (occupancy cliffs at 64/72/80/96/128 VGPRs)
template <int LOGT, int TPR, int WPB, int MAXR, bool SMALL, int MINW>
__global__ __launch_bounds__(WPB * 64, MINW) void k_scatter(ScatterArgs a)
{
    typedef typename ZpAcc<SMALL>::type Acc;
    typedef RowTable<LOGT, SMALL> Tab;
    constexpr bool WAVE_ROW = (TPR == 64);
    static_assert(WAVE_ROW || TPR == WPB * 64, "a row is owned by one wave or by the whole workgroup");
    constexpr int T = 1 << LOGT;
    constexpr int G = 8;            // lanes streaming one pivot row: 64 contiguous bytes per step
    constexpr int NG = TPR / G;     // pivot rows per round
    constexpr size_t RB = RetryList<SMALL>::BYTES;
    constexpr size_t SLOT = Tab::BYTES + 32 + RB; // wave-per-row: table, 32 bytes of s_misc, retry list of the wave
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform: row descriptors then live in SGPRs
    const int rtid = WAVE_ROW ? lane : tid;             // thread index inside the row team
    unsigned char *base = s_raw + (WAVE_ROW ? (size_t)wave * SLOT : 0);
    Tab tab;
    tab.bind(base);
    int *s_misc = (int *)(base + Tab::BYTES);           // block-per-row only: [0..3] entries written (per wave, or [0] shared), [4..7] leftmost columns
    RetryList<SMALL> rl;                                // block-per-row: the lists of the waves follow the shared table
    rl.bind(base + Tab::BYTES + 32 + (WAVE_ROW ? 0 : (size_t)wave * RB));
    const int gg = rtid / G, gl = rtid % G;
    const ZpField F = a.F;

    const int count = *a.class_count;
    if ((WAVE_ROW ? (int)blockIdx.x * WPB : (int)blockIdx.x) >= count) return; // nothing for this workgroup: skip even the table reset
    for (int s = rtid; s < T; s += TPR) tab.clear(s);
    if (rtid == 0) { s_misc[0] = 0; s_misc[4] = INT_MAX; }
    __syncthreads();

    const int first = WAVE_ROW ? (int)blockIdx.x * WPB + wave : (int)blockIdx.x;
    const int stride = WAVE_ROW ? (int)gridDim.x * WPB : (int)gridDim.x;
    u64d c_nnz = 0, c_ent = 0, c_seg = 0;
    int c_rows = 0;

#ifdef SPASM_STAMPS
    u64d st_sum[NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64d st_last = stamp_now();
#endif
    // ---- pipeline registers
    RowDesc d, dn;
    d.ent_start = d.l_start = d.s_start = 0; d.len = d.llen = d.t = d.bound = 0; d.pmask = -1;
    dn = d;
    int2 own = make_int2(0, 0);
    int4 rec[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; r++) rec[r] = make_int4(0, 0, 0, 0);
    if (first < count) {
        d = desc_unpack(desc_load(&a.desc[first]));
        if (rtid < d.len) own = a.ent[d.ent_start + rtid];
#pragma unroll
        for (int r = 0; r < MAXR; r++)
            if (gg + r * NG < d.llen) rec[r] = a.Lpool[d.l_start + gg + r * NG];
    }
    if (first + stride < count) dn = desc_unpack(desc_load(&a.desc[first + stride]));

    STAMP(0); // prologue
    for (int w = first; w < count; w += stride) {
        // (A) prefetch the descriptor two rows ahead (vector load; unpacked to SGPRs at the end of the iteration)
        // unconditional, clamped to the last descriptor: a load under `if` made the compiler wait for it on the spot and copy
        // its 12 registers into the merged variable -- a full global-load latency at the top of every row
        const DescRegs dnn_regs = desc_load(&a.desc[min(w + 2 * stride, count - 1)]);
        // (B) every load of the current row: qinv of the own entry, entries of the pivot rows
        const int ln = d.len, ll = d.llen;
        int q_own = 0;
        if (d.pmask >= 0) q_own = ((d.pmask >> rtid) & 1) ? 0 : -1; // the solve kernel already classified the own entries (rows of <= 32)
        else if (rtid < ln) q_own = a.qinv_r[own.x];
        int2 u[MAXR][3];
        int npn[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            npn[r] = (gg + r * NG < ll && rec[r].y != 0) ? rec[r].w : 0;
            // unconditional loads, clamped into the pivot row (lanes past its end re-read its last entry, groups without a
            // row read UPN[0]; the insert masks them out): a predicated load costs a compare, two exec updates and a branch
            const int2 *up = a.UPN + (unsigned)rec[r].z;
            const int last = max(npn[r] - 1, 0);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                u[r][j] = make_int2(0, 0);
                if (!SCATTER_DBG(a, 8)) u[r][j] = up[min(gl + j * G, last)];
            }
        }
#ifdef SPASM_STAMPS
        STAMP(1); // issue of the row's loads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(2); // waiting for them
#endif
        int r_ent = 0, r_seg = 0; // this row's contribution to the counters, 32-bit until the row is done
        bool anylong = false;
        // (C) accumulate: the own entry, then per round the (up to) 3 entries of this lane as one batch
        {
            const int oc[1] = {own.x};
            const Acc ov[1] = {(Acc)own.y};
            table_try_n<LOGT, 1, SMALL>(tab, rl, oc, ov, (rtid < ln && q_own < 0) ? 1u : 0u, a.ctr);
        }
        STAMP(3); // own entries
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            if (__ballot(npn[r] > 0) != 0) { // wave-uniform: the retry list is bookkept per wave
                if (gl == 0) { r_ent += npn[r]; r_seg += npn[r] > 0; }
                const int nm = -rec[r].y; // congruent to the canonical negative; the lazy product does not need more
                int bc[3];
                Acc bv[3];
                unsigned valid = 0;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    bc[j] = u[r][j].x;
                    bv[j] = ZpAcc<SMALL>::mul_lazy(F, nm, u[r][j].y);
                    if (gl + j * G < npn[r]) valid |= 1u << j;
                }
                if (SCATTER_DBG(a, 2)) asm volatile("" ::"v"(bc[0]), "v"(bv[0]));
                else table_try_n<LOGT, 3, SMALL>(tab, rl, bc, bv, valid, a.ctr);
                anylong |= npn[r] > 3 * G;
            }
        }
        // runs longer than 24 entries (rows of W: a row the streaming kernel handed back has a few runs of ~170): the rest of a run is
        // taken by all 64 lanes of the wave whose group had its head, through the retry list like everything else
        if (__ballot(anylong) != 0) {
            const int nrec = min(ll, NG * MAXR);
            for (int e = 0; e < nrec; e++) { // (wave-uniform)
                if (((e % NG) * G) / 64 != (WAVE_ROW ? 0 : wave)) continue; // the head went to a group of another wave
                const int4 le = a.Lpool[d.l_start + e];
                if (le.y == 0 || le.w <= 3 * G) continue;
                const int nm2 = -le.y;
                const int2 *up2 = a.UPN + (unsigned)le.z;
                for (int k0 = 3 * G; k0 < le.w; k0 += 64) {
                    const int k = k0 + lane;
                    const int2 uu = up2[min(k, le.w - 1)];
                    const int oc[1] = {uu.x};
                    const Acc ov[1] = {ZpAcc<SMALL>::mul_lazy(F, nm2, uu.y)};
                    table_try_n<LOGT, 1, SMALL>(tab, rl, oc, ov, k < le.w ? 1u : 0u, a.ctr);
                }
            }
        }
        STAMP(4); // unrolled rounds of pivot rows
        // (D) what the unrolled part did not cover: more pivot rows than NG*MAXR, own rows longer than TPR
        for (int e = gg + MAXR * NG; e < ll; e += NG) {
            const int4 le = a.Lpool[d.l_start + e];
            if (le.y == 0) continue;
            if (gl == 0) { r_ent += le.w; r_seg += 1; }
            const int nm = -le.y;
            const int2 *up = a.UPN + (unsigned)le.z;
            for (int k = gl; k < le.w; k += 3 * G) {
                int bc[3];
                Acc bv[3];
                unsigned valid = 0;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    int2 uu = make_int2(0, 0);
                    if (k + j * G < le.w) { uu = up[k + j * G]; valid |= 1u << j; }
                    bc[j] = uu.x;
                    bv[j] = ZpAcc<SMALL>::mul_lazy(F, nm, uu.y);
                }
                table_add_n<LOGT, 3, SMALL>(tab, bc, bv, valid, a.ctr);
            }
        }
        for (int k = rtid + TPR; k < ln; k += TPR) {
            const int2 e = a.ent[d.ent_start + k];
            if (a.qinv_r[e.x] < 0) table_add<LOGT, SMALL>(tab, e.x, (Acc)e.y, a.ctr);
        }
        retry_drain<LOGT, SMALL>(tab, rl, a.ctr);
        c_ent += (u64d)(unsigned)r_ent;
        c_seg += (u64d)(unsigned)r_seg;
        STAMP(5); // remainder loops + retries
        // (A') stage-1 data of the NEXT row (its own entries and multiplier records) straight into the pipeline
        // registers, which are dead from here on: these loads fly during the sweep
        const int ln_cur = ln;
        {
            // UNCONDITIONAL and clamped (lanes past the end of the row / groups past the end of the list re-read the last
            // entry / record and are masked out where the data is used; after the last row `dn` is the last descriptor
            // again).  Under an `if` the compiler loaded into temporaries, waited for them on the spot and copied them into
            // the pipeline registers: the latency these prefetches are meant to hide was paid in full before the sweep.
            own = a.ent[dn.ent_start + min(rtid, max(dn.len - 1, 0))];
            const int lastrec = max(dn.llen - 1, 0);
#pragma unroll
            for (int r = 0; r < MAXR; r++) rec[r] = a.Lpool[dn.l_start + min(gg + r * NG, lastrec)];
        }
        if (WAVE_ROW) __builtin_amdgcn_wave_barrier(); else lds_barrier();
        // (E) sweep: reduce, compact, write; reset the table on the way.  Branch-free per slot: an empty slot holds
        // accumulator 0, hence reduces to 0.  Output addresses are row base (scalar) + 32-bit lane offset.
        unsigned char *const rowp = (unsigned char *)(a.Sent + d.s_start);
        int mylead = INT_MAX;
        int n_out, lead_out;
        constexpr int NIT = T / TPR;
        if constexpr (NIT <= 16) {
            // every slot group of this wave is held in registers: count first, so that a workgroup sharing the row
            // needs ONE exchange (counts + leftmost columns through s_misc, no atomics) before the stores
            int cc[NIT], vv[NIT];
            u64d mm[NIT];
            int tot = 0;
#pragma unroll
            for (int q0 = 0; q0 < NIT; q0 += 4) {
                Acc aa[4];
#pragma unroll
                for (int q = q0; q < q0 + 4 && q < NIT; q++) tab.read(q * TPR + rtid, cc[q], aa[q - q0]);
#pragma unroll
                for (int q = q0; q < q0 + 4 && q < NIT; q++) tab.clear(q * TPR + rtid);
#pragma unroll
                for (int q = q0; q < q0 + 4 && q < NIT; q++) {
                    vv[q] = acc_reduce_short<SMALL>(F, aa[q - q0]);
                    const bool nz = vv[q] != 0 && !SCATTER_DBG(a, 4);
                    mm[q] = __ballot(nz);
                    tot += __popcll(mm[q]);
                    if (nz) mylead = min(mylead, cc[q]);
                }
            }
            mylead = wave_min_i32(mylead);
            int pos = 0;
            n_out = tot;
            lead_out = mylead;
            if (!WAVE_ROW) {
                if (lane == 0) { s_misc[wave] = tot; s_misc[4 + wave] = mylead; }
                lds_barrier();
                n_out = 0;
                lead_out = INT_MAX;
#pragma unroll
                for (int w2 = 0; w2 < WPB; w2++) {
                    const int cw = s_misc[w2];
                    n_out += cw;
                    if (w2 < wave) pos += cw;
                    lead_out = min(lead_out, s_misc[4 + w2]);
                }
            }
            if (tot != 0) {
#pragma unroll
                for (int q = 0; q < NIT; q++) {
                    if (vv[q] != 0 && !SCATTER_DBG(a, 4) && !SCATTER_DBG(a, 1)) { // streamed out, never re-read here: keep it from evicting the pivot rows
                        const unsigned off = (unsigned)(pos + __popcll(mm[q] & lanemask_lt())) << 3;
                        const long long pk = ((long long)(unsigned)vv[q] << 32) | (unsigned)cc[q];
                        __builtin_nontemporal_store(pk, (long long *)(rowp + off));
                    }
                    pos += __popcll(mm[q]);
                }
            }
        } else {
            // tables of more than 16 slot groups per wave: 4 groups at a time, one LDS reservation per batch
            int wbase = 0;
            for (int it0 = 0; it0 < NIT; it0 += 4) {
                int cc[4], vv[4];
                Acc aa[4];
                u64d mm[4];
                int tot = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) tab.read((it0 + q) * TPR + rtid, cc[q], aa[q]);
#pragma unroll
                for (int q = 0; q < 4; q++) tab.clear((it0 + q) * TPR + rtid);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    vv[q] = acc_reduce_short<SMALL>(F, aa[q]);
                    mm[q] = __ballot(vv[q] != 0 && !SCATTER_DBG(a, 4));
                    tot += __popcll(mm[q]);
                }
                if (tot == 0) continue;
                int pos = wbase;
                if (!WAVE_ROW) {
                    if (lane == 0) pos = atomicAdd(&s_misc[0], tot);
                    pos = __builtin_amdgcn_readfirstlane(pos);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (vv[q] != 0 && !SCATTER_DBG(a, 4)) {
                        if (!SCATTER_DBG(a, 1)) {
                            const unsigned off = (unsigned)(pos + __popcll(mm[q] & lanemask_lt())) << 3;
                            const long long pk = ((long long)(unsigned)vv[q] << 32) | (unsigned)cc[q];
                            __builtin_nontemporal_store(pk, (long long *)(rowp + off));
                        }
                        mylead = min(mylead, cc[q]);
                    }
                    pos += __popcll(mm[q]);
                }
                wbase += tot;
            }
            mylead = wave_min_i32(mylead);
            n_out = wbase;
            lead_out = mylead;
            if (!WAVE_ROW) {
                if (lane == 0 && mylead != INT_MAX) atomicMin(&s_misc[4], mylead);
                lds_barrier();
                n_out = s_misc[0];
                lead_out = s_misc[4];
                lds_barrier();
                if (rtid == 0) { s_misc[0] = 0; s_misc[4] = INT_MAX; }
            }
        }
        if (rtid == 0) {
            a.Slen[d.t] = n_out;
            a.Slead[d.t] = lead_out;
            c_nnz += (u64d)n_out;
            c_rows += n_out > 0;
            c_ent += (u64d)ln_cur;
            c_seg += 1;
        }
        if (WAVE_ROW) __builtin_amdgcn_wave_barrier(); else lds_barrier();
        STAMP(6); // prefetch issue + sweep + stores
        // rotate the pipeline
        d = dn;
        dn = desc_unpack(dnn_regs); // (beyond the last row this is the last descriptor again: never used)
    }
#ifdef SPASM_STAMPS
    if (lane == 0 && a.stamps) {
        for (int i = 0; i < NSTAMP; i++) atomicAdd(&a.stamps[(size_t)a.cls * 2 * NSTAMP + i], st_sum[i]);
        atomicAdd(&a.stamps[(size_t)a.cls * 2 * NSTAMP + NSTAMP], 1ull);
    }
#endif
    // counters: one atomic per wave
    for (int o = 32; o > 0; o >>= 1) {
        c_ent += __shfl_xor(c_ent, o);
        c_seg += __shfl_xor(c_seg, o);
        c_nnz += __shfl_xor(c_nnz, o);
        c_rows += __shfl_xor(c_rows, o);
    }
    if (lane == 0) {
        if (c_nnz) atomicAdd(&ctr_shard(a.ctr)->nnz_out, c_nnz);
        if (c_rows) atomicAdd(&ctr_shard(a.ctr)->nonempty_out, c_rows);
        if (c_ent | c_seg) {
            atomicAdd(&ctr_shard(a.ctr)->class_ent[a.cls], c_ent);
            atomicAdd(&ctr_shard(a.ctr)->class_seg[a.cls], c_seg);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// SCATTER, unbounded rows (last resort): one workgroup per row, a dense i64 accumulator over the
// columns and a touched-bitmap in global memory (both all-zero between rows), the list of touched
// columns beside them.  Only rows whose bound exceeds the largest LDS table come here.
// ------------------------------------------------------------------------------------------------
struct BigScatterArgs {
    ScatterArgs s;
    int m;
    int nwords;            // (m + 31) / 32
    long long *xdense;     // [gridDim.x][m]
    unsigned *bitmap;      // [gridDim.x][nwords]
    int *touched;          // [gridDim.x][m]
};

__global__ __launch_bounds__(256) void k_scatter_big(BigScatterArgs b)
{
    const ScatterArgs &a = b.s;
    __shared__ int s_ntouch, s_nout, s_lead;
    const int tid = threadIdx.x, lane = tid & 63;
    const ZpField F = a.F;
    long long *x = b.xdense + (size_t)blockIdx.x * b.m;
    unsigned *bm = b.bitmap + (size_t)blockIdx.x * b.nwords;
    int *tl = b.touched + (size_t)blockIdx.x * b.m;
    const int count = *a.class_count;
    u64d c_ent = 0, c_seg = 0;
    auto add = [&](int c, int v) {
        atomicAdd((u64d *)&x[c], (u64d)(long long)v);
        const unsigned bit = 1u << (c & 31);
        const unsigned old = atomicOr(&bm[c >> 5], bit);
        if (!(old & bit)) tl[atomicAdd(&s_ntouch, 1)] = c;
    };
    for (int w = blockIdx.x; w < count; w += gridDim.x) {
        const RowDesc d = a.desc[w];
        if (tid == 0) { s_ntouch = 0; s_nout = 0; s_lead = INT_MAX; }
        __syncthreads();
        for (int k = tid; k < d.len; k += 256) {
            const int2 e = a.ent[d.ent_start + k];
            if (a.qinv_r[e.x] < 0) add(e.x, e.y);
        }
        // 8-lane groups stream the pivot rows, as in the LDS kernel
        const int gg = tid >> 3, gl = tid & 7;
        for (int e = gg; e < d.llen; e += 32) {
            const int4 le = a.Lpool[d.l_start + e];
            if (le.y == 0) continue;
            if (gl == 0) { c_ent += (u64d)le.w; c_seg += 1; }
            const int nm = zp_neg(F, le.y);
            const int2 *up = a.UPN + (unsigned)le.z;
            for (int k = gl; k < le.w; k += 8) { const int2 u = up[k]; add(u.x, zp_mul(F, nm, u.y)); }
        }
        __syncthreads();
        const int nt = s_ntouch;
        for (int i0 = 0; i0 < nt; i0 += 256) {
            const int i = i0 + tid;
            int c = 0, v = 0;
            if (i < nt) {
                c = tl[i];
                v = zp_reduce(F, x[c]);
                x[c] = 0;
                bm[c >> 5] = 0;
            }
            const u64d m = __ballot(v != 0);
            if (m) {
                int pos = 0;
                if (lane == 0) pos = atomicAdd(&s_nout, __popcll(m));
                pos = __builtin_amdgcn_readfirstlane(pos);
                if (v != 0) {
                    a.Sent[d.s_start + pos + __popcll(m & lanemask_lt())] = make_int2(c, v);
                    atomicMin(&s_lead, c);
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            a.Slen[d.t] = s_nout;
            a.Slead[d.t] = s_lead;
            atomicAdd(&ctr_shard(a.ctr)->nnz_out, (u64d)s_nout);
            if (s_nout > 0) atomicAdd(&ctr_shard(a.ctr)->nonempty_out, 1);
            c_ent += (u64d)d.len;
            c_seg += 1;
        }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) { c_ent += __shfl_xor(c_ent, o); c_seg += __shfl_xor(c_seg, o); }
    if (lane == 0 && (c_ent | c_seg)) {
        atomicAdd(&ctr_shard(a.ctr)->class_ent[a.cls & 7], c_ent);
        atomicAdd(&ctr_shard(a.ctr)->class_seg[a.cls & 7], c_seg);
    }
}

// rows that were not processed (no class): publish empty rows so that the output is well defined
__global__ void k_scatter_mark_failed(int nrows, const int *__restrict__ Llen, int *__restrict__ Slen, int *__restrict__ Slead)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nrows && Llen[t] < 0) { Slen[t] = 0; Slead[t] = INT_MAX; }
}

// ------------------------------------------------------------------------------------------------
// transpose (counting sort by column) and kernel-basis assembly
// ------------------------------------------------------------------------------------------------
__global__ void k_iota(int n, int *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// cnt[c] += 1 for every entry of the listed rows; TEAM lanes per row
template <int TEAM>
__global__ void k_count_cols(int n, const i64d *__restrict__ start, const int *__restrict__ len, const int2 *__restrict__ ent, int *__restrict__ cnt)
{
    const int tl = threadIdx.x % TEAM;
    const int i = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (i >= n) return;
    const i64d st = start[i];
    const int ln = len[i];
    for (int k = tl; k < ln; k += TEAM) atomicAdd(&cnt[ent[st + k].x], 1);
}

// length of every output row: column j of the input becomes a row when keep[j] < 0 (or keep == NULL);
// `diag` adds one leading entry (the -1 of a kernel vector).  flag[j] = 1 for rows that exist.
__global__ void k_trow_len(int m, const int *__restrict__ keep, const int *__restrict__ cnt, int diag, i64d *__restrict__ tlen, int *__restrict__ flag)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) {
        const bool is_row = keep ? keep[j] < 0 : true;
        tlen[j] = is_row ? (i64d)cnt[j] + diag : 0;
        flag[j] = is_row ? 1 : 0;
    }
    if (j == m) { tlen[j] = 0; flag[j] = 0; }
}

// row pointers of the output (rows = kept columns, ascending) and the diagonal entries
__global__ void k_trow_ptr(int m, const int *__restrict__ keep, const int *__restrict__ rowidx, const i64d *__restrict__ tstart, int diag,
                           i64d *__restrict__ Tp, int2 *__restrict__ Tent, int *__restrict__ cursor)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) {
        const bool is_row = keep ? keep[j] < 0 : true;
        if (is_row) {
            Tp[rowidx[j]] = tstart[j];
            if (diag) Tent[tstart[j]] = make_int2(j, -1); // K[j] = -1 (reference test/runtests.jl:21: 42012 == -1)
        }
        cursor[j] = diag;
    }
    if (j == m) Tp[rowidx[m]] = tstart[m];
}

// entry (c, v) of input row a lands in output row c as (label[a], v)
template <int TEAM>
__global__ void k_tfill(int n, const i64d *__restrict__ start, const int *__restrict__ len, const int2 *__restrict__ ent,
                        const int *__restrict__ label, const i64d *__restrict__ tstart, int *__restrict__ cursor, int2 *__restrict__ Tent)
{
    const int tl = threadIdx.x % TEAM;
    const int i = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (i >= n) return;
    const i64d st = start[i];
    const int ln = len[i];
    const int lab = label ? label[i] : i;
    for (int k = tl; k < ln; k += TEAM) {
        const int2 e = ent[st + k];
        const int pos = atomicAdd(&cursor[e.x], 1);
        Tent[tstart[e.x] + pos] = make_int2(lab, e.y);
    }
}

// compact a "CSR with slack" matrix into tight arrays (row order kept)
__global__ void k_copy_len64(int n, const int *__restrict__ len, i64d *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = len[i];
    if (i == n) out[i] = 0;
}

// rows of a batch of the round appended compactly to the matrix of the next round: entries to dst + ostart[i],
// row i of the batch becomes row row_off + i with start dst_off + ostart[i]
// A shard's next-round matrix from its Schur rows, in place: local row (orig - lo) / stride of the new matrix is the slice of
// the Schur row with that original number; every other row (this round's pivots, rows that were or became empty) is empty.
// No entry moves: the new matrix takes over the Schur entry buffer.
__global__ void k_advance_init(int n, int row_lo, int row_stride, i64d *__restrict__ start, int *__restrict__ len, int *__restrict__ lead,
                               int *__restrict__ orig)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    start[i] = 0;
    len[i] = 0;
    lead[i] = INT_MAX;
    orig[i] = row_lo + i * row_stride;
}

__global__ void k_advance_rows(int nrows, int n, int row_lo, int row_stride, const i64d *__restrict__ sstart, const int *__restrict__ slen,
                               const int *__restrict__ slead, const int *__restrict__ sorig, i64d *__restrict__ start, int *__restrict__ len,
                               int *__restrict__ lead)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nrows) return;
    const int li = shard_local(sorig[t], row_lo, row_stride, n);
    if (li < 0) return;
    start[li] = sstart[t];
    len[li] = slen[t];
    lead[li] = slead[t];
}

template <int TEAM>
__global__ void k_append_rows(int n, const i64d *__restrict__ start, const int *__restrict__ len, const int *__restrict__ lead,
                              const int *__restrict__ orig, const int2 *__restrict__ ent, const i64d *__restrict__ ostart, i64d dst_off,
                              int row_off, int2 *__restrict__ dst, i64d *__restrict__ dstart, int *__restrict__ dlen, int *__restrict__ dlead,
                              int *__restrict__ dorig)
{
    const int tl = threadIdx.x % TEAM;
    const int i = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (i >= n) return;
    const i64d st = start[i], os = dst_off + ostart[i];
    const int ln = len[i];
    for (int k = tl; k < ln; k += TEAM) dst[os + k] = ent[st + k];
    if (tl == 0) {
        dstart[row_off + i] = os;
        dlen[row_off + i] = ln;
        dlead[row_off + i] = lead[i];
        dorig[row_off + i] = orig[i];
    }
}

// {col, val} pairs -> separate column / value arrays (what the host CSR holds), so that the download lands in place
__global__ void k_split_ent(i64d n, const int2 *__restrict__ ent, int *__restrict__ oj, int *__restrict__ ox)
{
    i64d i = (i64d)blockIdx.x * blockDim.x + threadIdx.x;
    const i64d stride = (i64d)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int2 e = ent[i];
        oj[i] = e.x;
        ox[i] = e.y;
    }
}

template <int TEAM>
__global__ void k_compact_rows(int n, const i64d *__restrict__ start, const int *__restrict__ len, const int2 *__restrict__ ent,
                               const i64d *__restrict__ ostart, int *__restrict__ oj, int *__restrict__ ox)
{
    const int tl = threadIdx.x % TEAM;
    const int i = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (i >= n) return;
    const i64d st = start[i], os = ostart[i];
    const int ln = len[i];
    for (int k = tl; k < ln; k += TEAM) {
        const int2 e = ent[st + k];
        oj[os + k] = e.x;
        ox[os + k] = e.y;
    }
}

// ------------------------------------------------------------------------------------------------
// SOLVE without dependent chains.  Per round the rows of Uinv = (I + N)^-1 (N = U_PP, strictly upper
// triangular in pivot-index space) are computed once by the chain solve above applied to the unit
// rows e_r (only npiv rows, short reach).  The multipliers of a non-pivot row are then
//        x_b = a_P * Uinv = sum over its entries (c, a_c) on pivot columns of a_c * Uinv[qinv(c)]
// a dependency-free sparse combine: each TEAM accumulates the lists into a small LDS hash table
// keyed by pivot index.  Same x_b (the triangular system has one solution), hence the same counters.
// ------------------------------------------------------------------------------------------------
__global__ void k_unit_rows(int npiv, const int *__restrict__ pivcol, i64d *__restrict__ start, int *__restrict__ len, int2 *__restrict__ ent)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npiv) return;
    start[t] = t;
    len[t] = 1;
    ent[t] = make_int2(pivcol[t], 1);
}

// per-column record for the combine: pivot index of the column (or -1) and where its row of Uinv lives
// (len < 0: that row of Uinv is not available)
__global__ void k_colinfo(int m, const int *__restrict__ qinv_r, const i64d *__restrict__ UinvStart, const int *__restrict__ UinvLen,
                          int4 *__restrict__ colinfo, unsigned *__restrict__ pbits)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x; // blockDim.x is a multiple of 64: whole waves, whole bitmap words
    const int q = j < m ? qinv_r[j] : -1;
    int4 r = make_int4(q, 0, 0, 0);
    if (q >= 0) { r.y = (int)(unsigned)UinvStart[q]; r.z = UinvLen[q]; }
    if (j < m) colinfo[j] = r;
    // one bit per column: is it a pivot column?  m/8 bytes stay in L2 (the 16-byte records do not: 16 m bytes), and five of
    // six entries of a row sit on non-pivot columns and need nothing else
    const u64d b = __ballot(q >= 0);
    const int lane = threadIdx.x & 63;
    if ((lane & 31) == 0) pbits[j >> 5] = (unsigned)(b >> lane);
}

// what a solve starts from, in one launch instead of five memsets (a launch costs ~4.5 us, a third of a 1/8 shard's fixed
// cost): statistics and pool counters zero, class counts zero, every row's pivot-column mask "none", bound[n] = 0
__global__ void k_solve_reset(int n, unsigned *__restrict__ ctr_words, int nctr_words, u64d *__restrict__ pool_ctr, int npool_words,
                              int *__restrict__ class_count, int nclass, i64d *__restrict__ bound, long long *__restrict__ pmask,
                              int *__restrict__ sflag, u64d *__restrict__ own_ctr)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nctr_words) ctr_words[t] = 0;
    if (t < npool_words) { pool_ctr[t] = 0; if (own_ctr) own_ctr[t] = 0; } // (the plan kernel's own-entry regions: same layout)
    if (t < nclass) class_count[t] = 0;
    if (t <= n) { pmask[t] = -1; sflag[t] = 0; }
    if (t == n) bound[n] = 0;
}

// (start,len) of the listed rows, gathered once so that a row team reads them with one load
__global__ void k_gather_rows(int n, const int *__restrict__ rows, const i64d *__restrict__ start, const int *__restrict__ len,
                              i64d *__restrict__ ostart, int *__restrict__ olen)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int row = rows ? rows[t] : t;
    ostart[t] = start[row];
    olen[t] = len[row];
}

struct CombineArgs {
    int nrows;
    const int *retry;          // when non-NULL: row slots to process (those that overflowed the previous class)
    const int *retry_count;
    const int *self_idx;
    const unsigned *pbits;     // bit j: column j is a pivot column of this round
    int dbg;                   // timing ablations, diagnostic builds only: 16 = no Uinv loads, 32 = no inserts, 64 = no header gathers, 128 = no record stores
    const i64d *rstart;        // per row slot: start / length of the row's own entries
    const int *rlen;
    const int2 *ent;
    const int4 *colinfo;       // per column: {pivot index or -1, offset of its Uinv row, its length, -}
    const UHdr *uhdr;
    const int2 *UinvPool;
    int4 *Lpool;               // records {stream position of the pivot row's first entry << 16 | npn, multiplier, offset in UPN, npn}
    int *Lidx;                 // when non-NULL: the pivot index of every record
    int *sflag;                // out: per row slot, 1 when the streaming scatter may take the row (exact bound within the free
                               // columns, no zero-valued own entry)
    u64d lpool_cap;            // entries per pool region
    u64d *pool_ctr;
    int npool;
    i64d *Lstart;
    int *Llen;
    i64d *bound;
    long long *pmask;          // out: per row slot, bit k <=> own entry k is on a pivot column (rows of <= 32 entries), else -1
    int free_cols;
    int *overflow_list;
    int *overflow_count;
    RoundCounters *ctr;
    ZpField F;
};

template <int LOGC, bool SMALL>
__device__ __forceinline__ bool team_table_add(int *key, typename ZpAcc<SMALL>::type *val, int idx, typename ZpAcc<SMALL>::type prod, int &fresh_slot)
{
    constexpr int CAPS = 1 << LOGC;
    unsigned h = ((unsigned)idx * 0x9E3779B1u) >> (32 - LOGC);
    for (int probes = 0; probes < CAPS; probes++) {
        const int kk = atomicCAS(&key[h], EMPTY_KEY, idx);
        if (kk == EMPTY_KEY || kk == idx) {
            if (kk == EMPTY_KEY) fresh_slot = (int)h; // a new pivot index: the caller lists its slot
            if (SMALL) atomicAdd((int *)&val[h], (int)prod);
            else atomicAdd((u64d *)&val[h], (u64d)prod);
            return true;
        }
        h = (h + 1) & (CAPS - 1);
    }
    return false;
}

template <int TEAM, int LOGC, int TPB, bool SMALL>
__global__ __launch_bounds__(TPB) void k_combine(CombineArgs a)
{
    typedef typename ZpAcc<SMALL>::type Acc;
    constexpr int CAPS = 1 << LOGC;      // slots per team
    constexpr int MAXD = CAPS / 2;       // distinct pivot indices a team accepts
    constexpr int TEAMS = TPB / TEAM;
    constexpr int MAXP = 4;              // Uinv rows whose loads are issued together
    __shared__ Acc s_val[TEAMS * CAPS];
    __shared__ int s_key[TEAMS * CAPS];
    __shared__ unsigned short s_slot[TEAMS * MAXD]; // slots of the distinct pivot indices, in order of arrival
    const int team = threadIdx.x / TEAM;
    const int tl = threadIdx.x % TEAM;
    int *key = s_key + team * CAPS;
    Acc *val = s_val + team * CAPS;
    unsigned short *slots = s_slot + team * MAXD;
    const ZpField F = a.F;
    for (int s = tl; s < CAPS; s += TEAM) { key[s] = EMPTY_KEY; val[s] = 0; }
    __syncthreads();

    u64d c_app = 0, c_red = 0, c_seg = 0;
    const int total = a.retry ? *a.retry_count : a.nrows;
    const i64d first = (i64d)blockIdx.x * TEAMS + team, stride = (i64d)gridDim.x * TEAMS;
    // pipeline: (slot, start, len) and the first 2*TEAM own entries of the next row are loaded one row ahead
    int t_n = 0, ln_n = 0;
    i64d st_n = 0;
    int2 own_na = make_int2(0, 0), own_nb = make_int2(0, 0);
    if (first < total) {
        t_n = a.retry ? a.retry[first] : (int)first;
        st_n = a.rstart[t_n];
        ln_n = a.rlen[t_n];
        if (tl < ln_n) own_na = a.ent[st_n + tl];
        if (tl + TEAM < ln_n) own_nb = a.ent[st_n + tl + TEAM];
    }
    for (i64d gteam = first; gteam < total; gteam += stride) {
        const int t = t_n, ln = ln_n;
        const i64d st = st_n;
        const int2 own_a = own_na, own_b = own_nb;
        if (gteam + stride < total) {
            t_n = a.retry ? a.retry[gteam + stride] : (int)(gteam + stride);
            st_n = a.rstart[t_n];
            ln_n = a.rlen[t_n];
            own_na = make_int2(0, 0);
            own_nb = make_int2(0, 0);
            if (tl < ln_n) own_na = a.ent[st_n + tl];
            if (tl + TEAM < ln_n) own_nb = a.ent[st_n + tl + TEAM];
        }
        const int self = a.self_idx ? a.self_idx[t] : -1;
        int nN = 0;
        bool ok = true;
        bool zero_own = false; // an own entry whose value is 0 mod p: the streaming scatter would write it out as it stands
        // distinct pivot indices met so far: the same value in every lane of the team (the team's inserts are sequential, so
        // no LDS counter is needed: a counter hit by 16 lanes at once serialises 16-fold)
        int dcnt = 0;
        auto note_fresh = [&](const int fs) {
            const u64d m = team_ballot<TEAM>(fs >= 0);
            if (fs >= 0) {
                const int p = dcnt + __popcll(m & ((1ull << tl) - 1ull));
                if (p < MAXD) slots[p] = (unsigned short)fs;
            }
            dcnt += __popcll(m);
        };
        // one batch of TEAM own entries: entries on pivot columns pull their row of Uinv into the table
        auto process = [&](const int2 own, const int4 ci, const bool valid) {
            const bool isP = valid && ci.x >= 0 && ci.x != self;
            nN += __popcll(team_ballot<TEAM>(valid && ci.x < 0));
            zero_own |= valid && own.y == 0;
            u64d mP = team_ballot<TEAM>(isP);
            while (mP && ok) {
                // up to MAXP pivot entries: all their Uinv rows are requested before any accumulation
                int av[MAXP], ul[MAXP];
                unsigned uo[MAXP];
                int2 wv[MAXP];
                bool fail = false;
#pragma unroll
                for (int j = 0; j < MAXP; j++) {
                    av[j] = 0; ul[j] = 0; uo[j] = 0;
                    wv[j] = make_int2(0, 0);
                    if (mP) {
                        const int src = __ffsll((long long)mP) - 1;
                        mP &= mP - 1;
                        av[j] = __shfl(own.y, src, TEAM);
                        uo[j] = (unsigned)__shfl(ci.y, src, TEAM);
                        ul[j] = __shfl(ci.z, src, TEAM);
                        if (ul[j] < 0) fail = true;
                        if (tl < ul[j] && !SCATTER_DBG(a, 16)) wv[j] = a.UinvPool[(i64d)uo[j] + tl];
                    }
                }
#pragma unroll
                for (int j = 0; j < MAXP; j++) {
                    int fs = -1;
                    if (tl < ul[j] && !SCATTER_DBG(a, 32) && !team_table_add<LOGC, SMALL>(key, val, wv[j].x, ZpAcc<SMALL>::mul_lazy(F, av[j], wv[j].y), fs)) fail = true;
                    note_fresh(fs);
                    for (int i0 = TEAM; i0 < ul[j]; i0 += TEAM) { // rows of Uinv longer than the team (trip count uniform in the team)
                        const int i = i0 + tl;
                        fs = -1;
                        if (i < ul[j]) {
                            const int2 w2 = a.UinvPool[(i64d)uo[j] + i];
                            if (!team_table_add<LOGC, SMALL>(key, val, w2.x, ZpAcc<SMALL>::mul_lazy(F, av[j], w2.y), fs)) fail = true;
                        }
                        note_fresh(fs);
                    }
                }
                if (team_ballot<TEAM>(fail) != 0) ok = false;
                if (dcnt > MAXD) ok = false;
            }
        };
        {
            // the two prefetched batches: both column-record gathers in flight together
            const bool va = tl < ln, vb = tl + TEAM < ln;
            int4 ci_a = make_int4(-1, 0, 0, 0), ci_b = ci_a;
            // both bitmap words requested before either is used (lanes without an entry hold column 0: word 0 is always there);
            // under separate `if`s the second load was issued only after the first had returned
            const unsigned wa = a.pbits[(unsigned)own_a.x >> 5], wb = a.pbits[(unsigned)own_b.x >> 5];
            const bool pa = va && ((wa >> (own_a.x & 31)) & 1u), pb = vb && ((wb >> (own_b.x & 31)) & 1u);
            if (pa) ci_a = a.colinfo[own_a.x];
            if (pb) ci_b = a.colinfo[own_b.x];
            {
                const u64d mA = team_ballot<TEAM>(va && ci_a.x >= 0), mB = team_ballot<TEAM>(vb && ci_b.x >= 0);
                if (a.pmask && tl == 0) a.pmask[t] = (ln <= 32 && 2 * TEAM >= ln) ? (long long)(mA | (TEAM < 32 ? (mB << (TEAM & 31)) : 0ull)) : -1;
            }
            process(own_a, ci_a, va);
            if (ln > TEAM && ok) process(own_b, ci_b, vb);
        }
        for (int k0 = 2 * TEAM; k0 < ln && ok; k0 += TEAM) {
            const int k = k0 + tl;
            const bool valid = k < ln;
            int2 own = make_int2(0, 0);
            int4 ci = make_int4(-1, 0, 0, 0);
            if (valid) {
                own = a.ent[st + k];
                if ((a.pbits[(unsigned)own.x >> 5] >> (own.x & 31)) & 1u) ci = a.colinfo[own.x];
            }
            process(own, ci, valid);
        }
        const int dcount = dcnt;
        if (!ok) {
            for (int s = tl; s < CAPS; s += TEAM) { key[s] = EMPTY_KEY; val[s] = 0; }
            if (tl == 0) {
                a.Llen[t] = -1;
                a.Lstart[t] = 0;
                a.bound[t] = 0;
                const int pos = atomicAdd(a.overflow_count, 1);
                if (a.overflow_list) a.overflow_list[pos] = t;
            }
            continue;
        }
        u64d base = 0;
        if (tl == 0) base = pool_alloc(a.pool_ctr, a.lpool_cap, (u64d)dcount, a.npool);
        base = __shfl(base, 0, TEAM);
        const bool room = base != ~0ull;
        // ---- one pass over the occupied slots (dcount <= MAXD of them): reduce, drop the zero multipliers, fetch the
        // headers of the applied pivot rows (gathers of a batch in flight together), records out
        int nout = 0;
        i64d bound = 0; // the row's entry stream: the pivot rows in record order; its nN own entries on non-pivot columns fill it from the end
        u64d r_red = 0;
        for (int i0 = 0; i0 < dcount; i0 += 4 * TEAM) {
            int kk[4], vv[4], pos[4];
            UHdr hh[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = i0 + j * TEAM + tl;
                kk[j] = 0; vv[j] = 0;
                hh[j].off = 0; hh[j].npp = 0; hh[j].npn = 0; hh[j].len = 0;
                if (i < dcount) {
                    const int s = slots[i];
                    kk[j] = key[s];
                    vv[j] = acc_reduce_short<SMALL>(F, val[s]);
                    key[s] = EMPTY_KEY;
                    val[s] = 0;
                }
                const u64d m = team_ballot<TEAM>(vv[j] != 0);
                pos[j] = nout + __popcll(m & ((1ull << tl) - 1ull));
                nout += __popcll(m);
                if (vv[j] != 0 && !SCATTER_DBG(a, 64)) hh[j] = a.uhdr[kk[j]];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // stream position of the record's pivot row: running total + exclusive prefix over the team (records without a
                // multiplier carry npn = 0)
                int tot;
                const int incl = team_incl_scan<TEAM>(hh[j].npn, tot);
                if (vv[j] != 0) {
                    const int pre = (int)(bound < (i64d)INT_MAX ? bound : (i64d)INT_MAX) + incl - hh[j].npn;
                    if (room && !SCATTER_DBG(a, 128)) {
                        // .x: what the streaming scatter reads in one word (its rows have bounds, hence positions and npn, below 2^14)
                        a.Lpool[base + pos[j]] = make_int4((int)(((unsigned)pre << 16) | (unsigned)min(hh[j].npn, 0xffff)), vv[j], (int)hh[j].off, hh[j].npn);
                        if (a.Lidx) a.Lidx[base + pos[j]] = kk[j];
                    }
                    r_red += (u64d)hh[j].len;
                }
                bound += tot;
            }
        }
        for (int o = TEAM / 2; o > 0; o >>= 1) r_red += __shfl_xor(r_red, o, TEAM);
        r_red += (u64d)ln;
        bound += nN;
        const bool any_zero_own = team_ballot<TEAM>(zero_own) != 0;
        if (tl == 0) {
            if (!room) {
                atomicAdd(&ctr_shard(a.ctr)->lpool_overflow, 1);
                a.Llen[t] = 0; a.Lstart[t] = 0; a.bound[t] = 0;
            } else {
                a.Lstart[t] = (i64d)base;
                a.Llen[t] = SCATTER_DBG(a, 128) ? 0 : nout; // ablation: the scatter must not read records that were never written
                a.bound[t] = bound < (i64d)a.free_cols ? bound : (i64d)a.free_cols;
                if (a.sflag) a.sflag[t] = (bound <= (i64d)a.free_cols && !any_zero_own) ? 1 : 0;
                c_app += (u64d)nout;
                c_red += r_red;
                c_seg += 1 + (u64d)nout;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        c_app += __shfl_xor(c_app, o);
        c_red += __shfl_xor(c_red, o);
        c_seg += __shfl_xor(c_seg, o);
    }
    if ((threadIdx.x & 63) == 0 && (c_app | c_red | c_seg)) {
        atomicAdd(&ctr_shard(a.ctr)->applications, c_app);
        atomicAdd(&ctr_shard(a.ctr)->nnz_reduced, c_red);
        atomicAdd(&ctr_shard(a.ctr)->segments, c_seg);
    }
}

// ------------------------------------------------------------------------------------------------
// SOLVE, unbounded reach (last resort): one workgroup per row, a dense value vector and a pending
// bitmap over the pivot indices in global memory (both all-zero between rows), pivots popped in
// increasing index order.  Serial in the reach; only rows that overflow every LDS class come here.
// ------------------------------------------------------------------------------------------------
struct BigSolveArgs {
    SolveArgs s;
    int npiv;
    int nwords;              // (npiv + 31) / 32
    int *xdense;             // [gridDim.x][npiv]
    unsigned *bitmap;        // [gridDim.x][nwords]
    int4 *scratch;           // [gridDim.x][npiv]
};

// LDSX = true: the value vector and the bitmap live in LDS (dynamic shared memory: 4*npiv + 4*nwords bytes, so only
// for rounds with up to ~30000 pivots): the same algorithm at LDS latency, the class that takes over from the
// small sorted lists when it fits.
// NT = threads per row: 256 for the LDS variant (one or two rows per CU anyway); 64 for the global one, where the work per
// popped pivot is a handful of entries and the rows in flight (x4 with one wave per row) are the throughput
template <bool LDSX, int NT>
__global__ __launch_bounds__(NT) void k_solve_big(BigSolveArgs b)
{
    const SolveArgs &a = b.s;
    __shared__ int s_word;       // index of the word holding the next pending pivot, INT_MAX if none in the window
    __shared__ int s_nN;
    __shared__ u64d s_base;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const int tid = threadIdx.x;
    const ZpField F = a.F;
    int *x = LDSX ? (int *)s_dyn : b.xdense + (size_t)blockIdx.x * b.npiv;
    unsigned *bm = LDSX ? (unsigned *)(s_dyn + (size_t)b.npiv * 4) : b.bitmap + (size_t)blockIdx.x * b.nwords;
    int4 *out = b.scratch + (size_t)blockIdx.x * b.npiv;
    if (LDSX) {
        for (int i = tid; i < b.npiv; i += NT) x[i] = 0;
        for (int i = tid; i < b.nwords; i += NT) bm[i] = 0;
        __syncthreads();
    }
    const int total = *a.retry_count;
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        const int t = a.retry[w];
        const int row = a.rows ? a.rows[t] : t;
        const int self = a.self_idx ? a.self_idx[t] : -1;
        const i64d st = a.start[row];
        const int ln = a.len[row];
        if (tid == 0) s_nN = 0;
        __syncthreads();
        int myN = 0;
        for (int k = tid; k < ln; k += NT) {
            const int2 e = a.ent[st + k];
            const int q = a.qinv_r[e.x];
            if (q >= 0 && q != self) {
                x[q] = e.y; // columns of a row are distinct
                atomicOr(&bm[q >> 5], 1u << (q & 31));
            } else if (q < 0) myN++;
        }
        if (myN) atomicAdd(&s_nN, myN);
        __syncthreads();
        int cnt = 0, wbase = 0;
        i64d bound = 0;
        u64d r_app = 0, r_red = (u64d)ln;
        while (wbase < b.nwords) {
            // first non-zero word in [wbase, wbase + NT): lowest lane per wave, lowest wave through LDS
            const int wi = wbase + tid;
            const unsigned wv = wi < b.nwords ? bm[wi] : 0u;
            if (tid == 0) s_word = INT_MAX;
            __syncthreads();
            const u64d mb = __ballot(wv != 0);
            if (mb && (tid & 63) == 0) atomicMin(&s_word, wbase + (tid & ~63) + (__ffsll((long long)mb) - 1));
            __syncthreads();
            const int found = s_word;
            __syncthreads(); // s_word is reset by the next iteration
            if (found == INT_MAX) { wbase += NT; continue; }
            const unsigned word = bm[found];
            const int bit = __ffs((int)word) - 1;
            const int idx = found * 32 + bit;
            const int mult = x[idx];
            __syncthreads();
            UHdr h; h.off = 0; h.npp = 0; h.npn = 0; h.len = 0;
            if (mult != 0) h = a.uhdr[idx];
            if (tid == 0) {
                atomicAnd(&bm[found], ~(1u << bit)); // other lanes may be setting bits of the same word
                x[idx] = 0;
                out[cnt] = make_int4(idx, mult, (int)h.off, h.npn);
            }
            cnt++;
            if (mult != 0) {
                r_app += 1;
                r_red += (u64d)h.len;
                bound += h.npn;
                const int nm = zp_neg(F, mult);
                for (int k = tid; k < h.npp; k += NT) {
                    const int2 e = a.UPP[(i64d)h.off + k];
                    x[e.x] = zp_axpy(F, nm, e.y, x[e.x]); // targets of one pivot row are distinct
                    atomicOr(&bm[e.x >> 5], 1u << (e.x & 31));
                }
            }
            __syncthreads();
            wbase = found; // the same word may hold further pivots
        }
        // publish
        if (tid == 0) s_base = pool_alloc(a.pool_ctr, a.lpool_cap, (u64d)cnt, a.npool);
        __syncthreads();
        const u64d base = s_base;
        if (base == ~0ull) {
            if (tid == 0) { atomicAdd(&ctr_shard(a.ctr)->lpool_overflow, 1); a.Llen[t] = 0; a.Lstart[t] = 0; a.bound[t] = 0; }
        } else {
            for (int i = tid; i < cnt; i += NT) {
                const int4 r = out[i];
                if (a.Lpool2) a.Lpool2[base + i] = make_int2(r.x, r.y);
                else {
                    a.Lpool[base + i] = make_int4(0, r.y, r.z, r.w);
                    if (a.Lidx) a.Lidx[base + i] = r.x;
                }
            }
            if (tid == 0) {
                bound += s_nN;
                a.Lstart[t] = (i64d)base;
                a.Llen[t] = cnt;
                a.bound[t] = bound < (i64d)a.free_cols ? bound : (i64d)a.free_cols;
                atomicAdd(&ctr_shard(a.ctr)->applications, r_app);
                atomicAdd(&ctr_shard(a.ctr)->nnz_reduced, r_red);
                atomicAdd(&ctr_shard(a.ctr)->segments, 1 + r_app);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// DENSE TAIL (replaces libspasm's spasm_schur_dense + spasm_ffpack_rref finish, prototypes reference
// src/SpaSM.jl:765-769, :805-806).  The live rows x live columns of the Schur complement are gathered
// into a dense row-major i32 matrix and eliminated column by column with the LEFTMOST-pivot rule
// (first non-pivotal row holding a non-zero in the column), so the pivot columns stay the leading
// columns of the row space.  Pivot decisions live on the device: the host only enqueues kernels.
// This first version is a right-looking rank-1 update per column (HBM-bound); the blocked variant
// with an MFMA trailing update is the planned replacement.
// ------------------------------------------------------------------------------------------------
struct DenseState {
    int cur;       // pivot row of the column being eliminated, -1 if the column has none
    int npiv;      // pivots found so far
    int npp;       // pivots found in the current panel (blocked variant)
    int pad;
};

__global__ void k_flag_cols(int n, const i64d *__restrict__ start, const int *__restrict__ len, const int2 *__restrict__ ent, int *__restrict__ flag)
{
    // one wave-sized team per row keeps it simple: rows of the tail are long
    const int tl = threadIdx.x & 63;
    const int i = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (i >= n) return;
    const i64d st = start[i];
    const int ln = len[i];
    for (int k = tl; k < ln; k += 64) flag[ent[st + k].x] = 1;
}

__global__ void k_col_map(int m, const int *__restrict__ flag, const int *__restrict__ scan, int *__restrict__ cmap, int *__restrict__ clist)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    if (flag[j]) { cmap[j] = scan[j]; clist[scan[j]] = j; }
    else cmap[j] = -1;
}

// DT: the element type of the dense matrix -- int, or (dense.hpp path, primes below 2^16 / 2^8) short / signed char: the block
// updates of the dense finish are bound by reading and writing D, and a balanced residue of such a prime fits
template <typename DT>
__global__ void k_dense_fill(int R, const int *__restrict__ rows, const i64d *__restrict__ start, const int *__restrict__ len,
                             const int2 *__restrict__ ent, const int *__restrict__ cmap, DT *__restrict__ D, i64d ldc)
{
    const int tl = threadIdx.x & 63;
    const int r = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (r >= R) return;
    const int row = rows[r];
    const i64d st = start[row];
    const int ln = len[row];
    for (int k = tl; k < ln; k += 64) {
        const int2 e = ent[st + k];
        D[(i64d)r * ldc + cmap[e.x]] = (DT)e.y;
    }
}

// first non-pivotal row with a non-zero in column c (one workgroup)
__global__ __launch_bounds__(1024) void k_dense_find(int c, int R, const int *__restrict__ D, i64d ldc, int *__restrict__ is_piv,
                                                     int *__restrict__ pivrow_of_col, DenseState *__restrict__ st)
{
    __shared__ int s_best;
    if (threadIdx.x == 0) s_best = INT_MAX;
    __syncthreads();
    int best = INT_MAX;
    for (int i = threadIdx.x; i < R && i < best; i += 1024)
        if (!is_piv[i] && D[(i64d)i * ldc + c] != 0) { best = i; break; }
    best = wave_min_i32(best);
    if ((threadIdx.x & 63) == 0 && best != INT_MAX) atomicMin(&s_best, best);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int b = s_best;
        if (b != INT_MAX) {
            st->cur = b;
            st->npiv += 1;
            st->npp += 1;
            is_piv[b] = 1;
            pivrow_of_col[c] = b;
        } else {
            st->cur = -1;
            pivrow_of_col[c] = -1;
        }
    }
}

// normalise the pivot row (unit pivot) into prow[c..C), keep it in D as the U row, and copy column c of the
// non-pivotal rows into fcol (the elimination factors) so that the update kernel has no read/write race on it
__global__ void k_dense_scale(int c, int R, int C, ZpField F, int *__restrict__ D, i64d ldc, const int *__restrict__ is_piv,
                              int *__restrict__ prow, int *__restrict__ fcol, const DenseState *__restrict__ st)
{
    const int p = st->cur;
    if (p < 0) return;
    const int inv = zp_inverse(F, D[(i64d)p * ldc + c]);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c && i < C) {
        const int v = zp_mul(F, inv, D[(i64d)p * ldc + i]);
        prow[i] = v;
    }
    if (i < R) fcol[i] = is_piv[i] ? 0 : D[(i64d)i * ldc + c];
}

__global__ void k_dense_store_prow(int c, int C, int *__restrict__ D, i64d ldc, const int *__restrict__ prow, const DenseState *__restrict__ st)
{
    const int p = st->cur;
    if (p < 0) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c && i < C) D[(i64d)p * ldc + i] = prow[i];
}

// D[i][c..C) -= fcol[i] * prow[c..C) for every non-pivotal row i; blockIdx.y = row, blockIdx.x tiles the columns
__global__ __launch_bounds__(256) void k_dense_elim(int c, int R, int C, ZpField F, int *__restrict__ D, i64d ldc, const int *__restrict__ prow,
                                                    const int *__restrict__ fcol, const DenseState *__restrict__ st)
{
    if (st->cur < 0) return;
    const int i = blockIdx.y;
    const int f = fcol[i];
    if (f == 0) return;
    const int nf = zp_neg(F, f);
    const int j = c + blockIdx.x * 256 + threadIdx.x;
    if (j < C) {
        int *d = D + (i64d)i * ldc + j;
        *d = zp_axpy(F, nf, prow[j], *d);
    }
}

// U rows out of the eliminated dense matrix: pivot column c (dense index) -> row pivrow_of_col[c], entries at columns >= c
// (c0: first column of the range the launch covers -- pivrow_of_col and pscan are indexed from there)
template <typename DT>
__global__ void k_dense_count(int C, const DT *__restrict__ D, i64d ldc, const int *__restrict__ pivrow_of_col, const int *__restrict__ pscan,
                              i64d *__restrict__ ulen, int c0 = 0)
{
    // one workgroup per dense column; pscan = exclusive scan of (pivrow_of_col >= 0)
    const int c = c0 + blockIdx.x;
    const int p = pivrow_of_col[c];
    if (p < 0) return;
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int cnt = 0;
    for (int j = c + threadIdx.x; j < C; j += blockDim.x) cnt += D[(i64d)p * ldc + j] != 0;
    if (cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (threadIdx.x == 0) ulen[pscan[c - c0]] = s_cnt;
}

template <typename DT>
__global__ void k_dense_emit(int C, const DT *__restrict__ D, i64d ldc, const int *__restrict__ pivrow_of_col, const int *__restrict__ pscan,
                             const i64d *__restrict__ uoff, const int *__restrict__ clist, const int *__restrict__ row_orig,
                             int2 *__restrict__ Ufull, int *__restrict__ pivcol, int *__restrict__ piv_orig, int c0 = 0)
{
    // one wave per pivot column keeps the entries of a U row in ascending column order
    const int c = c0 + blockIdx.x;
    const int p = pivrow_of_col[c];
    if (p < 0) return;
    const int k = pscan[c - c0];
    const int lane = threadIdx.x & 63;
    i64d pos = uoff[k];
    for (int j0 = c; j0 < C; j0 += 64) {
        const int j = j0 + lane;
        const int v = j < C ? (int)D[(i64d)p * ldc + j] : 0;
        const u64d m = __ballot(v != 0);
        if (v != 0) Ufull[pos + __popcll(m & lanemask_lt())] = make_int2(clist[j], v);
        pos += __popcll(m);
    }
    if (lane == 0) { pivcol[k] = clist[c]; piv_orig[k] = row_orig[p]; }
}

// ------------------------------------------------------------------------------------------------
// Schur complement through a DENSE W (spasm_schur_dense, prototype reference src/SpaSM.jl:765-766, for rounds whose rows reach
// thousands of pivots -- Macaulay-like matrices).  The multiplier solve walks the reach of every row, one dependent round trip
// per pivot (k_solve / k_solve_big: 8.7 of 11.5 s of kernel time on the 200k x 80k case); a dense image of
//     W = -(I + U_PP)^-1 U_PN          (one row per pivot, one column per column of the dense matrix D)
// costs the same whatever the reach: row q of W is  -U_PN[q] - sum over the entries (c, v) of U_PP[q] of v * W[c], c > q, so the
// rows are computed level by level of the pivot graph (levels from the host, one launch each), and then
//     D[row] = row_N + sum over the row's entries (col, a) on pivot columns of a * W[q(col)].
// Every step is a combination of dense rows with a handful of coefficients: HBM-bound streaming, 16 bytes per lane.  The columns
// are independent, so W is built for a slab of columns at a time when it does not fit whole, and a sample of 64 columns gives
// the density estimate (spasm_schur_estimate_density) without any row solve.
// cmap_s: column of the matrix -> column of the slab, -1 outside.
// ------------------------------------------------------------------------------------------------
__global__ void k_slab_cmap(int m, const int *__restrict__ cmap, int s0, int Cs, int stride, int *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int idx = cmap[j];
    int o = -1;
    if (idx >= 0) {
        if (stride <= 1) { if (idx >= s0 && idx - s0 < Cs) o = idx - s0; }
        else if (idx % stride == 0 && idx / stride < Cs) o = idx / stride;
    }
    out[j] = o;
}

// WT: element type of the dense W (int, or short / signed char for primes below 2^16 / 2^8, like D)
template <int TEAM, typename WT>
__global__ void k_wd_seed(int npiv, ZpField F, const UHdr *__restrict__ uhdr, const int2 *__restrict__ UPN, const int *__restrict__ cmap_s,
                          WT *__restrict__ Wd, i64d ldw)
{
    const int tl = threadIdx.x % TEAM;
    const int q = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (q >= npiv) return;
    const UHdr h = uhdr[q];
    for (int k = tl; k < h.npn; k += TEAM) {
        const int2 e = UPN[(i64d)h.off + k];
        const int j = cmap_s[e.x];
        if (j >= 0) Wd[(i64d)q * ldw + j] = (WT)zp_neg(F, e.y);
    }
}

typedef int v4i32 __attribute__((ext_vector_type(4)));

template <bool SMALL> struct DenseAcc;
template <> struct DenseAcc<true> { // p < 2^16: |a * w| < 2^30, an i64 takes 2^33 terms
    long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    __device__ __forceinline__ void fma(const ZpField &, int c, v4i32 w)
    {
        a0 += (long long)c * w.x; a1 += (long long)c * w.y; a2 += (long long)c * w.z; a3 += (long long)c * w.w;
    }
    __device__ __forceinline__ v4i32 finish(const ZpField &F, v4i32 base) const
    {
        return (v4i32){zp_reduce(F, a0 + base.x), zp_reduce(F, a1 + base.y), zp_reduce(F, a2 + base.z), zp_reduce(F, a3 + base.w)};
    }
};
template <> struct DenseAcc<false> { // lazy products |r| < 2^31.1 (ZpAcc<false>), an i64 takes 2^31 of them
    long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    __device__ __forceinline__ void fma(const ZpField &F, int c, v4i32 w)
    {
        a0 += ZpAcc<false>::mul_lazy(F, c, w.x); a1 += ZpAcc<false>::mul_lazy(F, c, w.y);
        a2 += ZpAcc<false>::mul_lazy(F, c, w.z); a3 += ZpAcc<false>::mul_lazy(F, c, w.w);
    }
    __device__ __forceinline__ v4i32 finish(const ZpField &F, v4i32 base) const
    {
        return (v4i32){zp_reduce(F, a0 + base.x), zp_reduce(F, a1 + base.y), zp_reduce(F, a2 + base.z), zp_reduce(F, a3 + base.w)};
    }
};

// four consecutive elements of a dense row (the address is a multiple of four elements)
__device__ __forceinline__ v4i32 dense_load4(const int *p) { return *(const v4i32 *)p; }
__device__ __forceinline__ v4i32 dense_load4(const short *p)
{
    const int2 w = *(const int2 *)p;
    return (v4i32){(int)(short)(w.x & 0xffff), w.x >> 16, (int)(short)(w.y & 0xffff), w.y >> 16};
}
__device__ __forceinline__ v4i32 dense_load4(const signed char *p)
{
    const int w = *(const int *)p;
    return (v4i32){(int)(signed char)(w & 0xff), (int)(signed char)((w >> 8) & 0xff), (int)(signed char)((w >> 16) & 0xff), w >> 24};
}
__device__ __forceinline__ void dense_store4(int *p, v4i32 v) { *(v4i32 *)p = v; }
__device__ __forceinline__ void dense_store4(short *p, v4i32 v)
{
    *(int2 *)p = make_int2((v.x & 0xffff) | (v.y << 16), (v.z & 0xffff) | (v.w << 16));
}
__device__ __forceinline__ void dense_store4(signed char *p, v4i32 v)
{
    *(int *)p = (v.x & 0xff) | ((v.y & 0xff) << 8) | ((v.z & 0xff) << 16) | (v.w << 24);
}

// rows order[0 .. nrows) of W (one level of the pivot graph): W[q] -= sum v * W[c].  blockIdx.x = row, blockIdx.y = chunk of
// 4 * blockDim.x columns; ldw and the slab width are multiples of 4
template <bool SMALL, typename WT>
__global__ __launch_bounds__(256) void k_wd_level(int nrows, const int *__restrict__ order, ZpField F, const UHdr *__restrict__ uhdr,
                                                  const int2 *__restrict__ UPP, WT *__restrict__ Wd, i64d ldw, int Cs)
{
    const int q = order[blockIdx.x];
    const int j = (blockIdx.y * blockDim.x + threadIdx.x) * 4;
    if (j >= Cs) return;
    const UHdr h = uhdr[q];
    DenseAcc<SMALL> acc;
    const int2 *up = UPP + (i64d)h.off;
    for (int k = 0; k < h.npp; k++) {
        const int2 e = up[k]; // (uniform over the workgroup)
        const v4i32 w = dense_load4(Wd + (i64d)e.x * ldw + j);
        acc.fma(F, zp_neg(F, e.y), w);
    }
    WT *dst = Wd + (i64d)q * ldw + j;
    dense_store4(dst, acc.finish(F, dense_load4(dst)));
    (void)nrows;
}

// ---- the same two combinations of dense rows for narrow residues (bytes for p < 2^8, shorts for p < 2^16), 16 BYTES per lane and term
// instead of four elements: with byte-wide W a four-element load moves 256 bytes per wave instruction, and the build of the dense
// rows of config 5 ran at half a TB/s (k_wd_rows 9.2 of 58.7 s of kernel time at full size, k_wd_level 6.0).
template <typename WT> struct WideRow {
    static constexpr int EPT = 16 / (int)sizeof(WT);
    static __device__ __forceinline__ void load(const WT *p, int (&w)[EPT])
    {
        const v4i32 raw = *(const v4i32 *)p;
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            if (sizeof(WT) == 1) w[i] = (int)(signed char)((raw[i >> 2] >> (8 * (i & 3))) & 255);
            else w[i] = (int)(short)((raw[i >> 1] >> (16 * (i & 1))) & 65535);
        }
    }
    static __device__ __forceinline__ void store(WT *p, const int (&w)[EPT])
    {
        v4i32 raw = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            if (sizeof(WT) == 1) raw[i >> 2] |= (w[i] & 255) << (8 * (i & 3));
            else raw[i >> 1] |= (w[i] & 65535) << (16 * (i & 1));
        }
        *(v4i32 *)p = raw;
    }
};

// rows order[0 .. nrows) of W (one level): W[q] -= sum v * W[c]; a lane owns WideRow<WT>::EPT consecutive columns
template <typename WT>
__global__ __launch_bounds__(256) void k_wd_level_wide(int nrows, const int *__restrict__ order, ZpField F, const UHdr *__restrict__ uhdr,
                                                       const int2 *__restrict__ UPP, WT *__restrict__ Wd, i64d ldw, int Cs)
{
    constexpr int EPT = WideRow<WT>::EPT;
    const int q = order[blockIdx.x];
    const int j = (blockIdx.y * blockDim.x + threadIdx.x) * EPT;
    if (j >= Cs) return;
    const UHdr h = uhdr[q];
    long long acc[EPT];
#pragma unroll
    for (int i = 0; i < EPT; i++) acc[i] = 0;
    const int2 *up = UPP + (i64d)h.off;
    for (int k = 0; k < h.npp; k++) {
        const int2 e = up[k]; // (uniform over the workgroup)
        int w[EPT];
        WideRow<WT>::load(Wd + (i64d)e.x * ldw + j, w);
        const int c = zp_neg(F, e.y);
#pragma unroll
        for (int i = 0; i < EPT; i++) acc[i] += (long long)c * w[i]; // |c w| < 2^30: an i64 takes 2^33 terms
    }
    WT *dst = Wd + (i64d)q * ldw + j;
    int b[EPT];
    WideRow<WT>::load(dst, b);
#pragma unroll
    for (int i = 0; i < EPT; i++) b[i] = zp_reduce(F, acc[i] + b[i]);
    WideRow<WT>::store(dst, b);
    (void)nrows;
}

// dense Schur rows: D[t][dcol0 + j] += sum a * W[q][j] over the row's list (q, a); D and W hold residues of the same width
template <typename WT>
__global__ __launch_bounds__(256) void k_wd_rows_wide(int nrows, ZpField F, const i64d *__restrict__ poff, const int2 *__restrict__ plist,
                                                      const WT *__restrict__ Wd, i64d ldw, int Cs, WT *__restrict__ D, i64d ldc, int dcol0)
{
    constexpr int EPT = WideRow<WT>::EPT;
    const int t = blockIdx.x;
    const int j = (blockIdx.y * blockDim.x + threadIdx.x) * EPT;
    if (j >= Cs) return;
    const i64d lo = poff[t], hi = poff[t + 1];
    if (lo == hi) return; // (the own entries are there already)
    long long acc[EPT];
#pragma unroll
    for (int i = 0; i < EPT; i++) acc[i] = 0;
    for (i64d k = lo; k < hi; k++) {
        const int2 e = plist[k]; // (uniform over the workgroup)
        int w[EPT];
        WideRow<WT>::load(Wd + (i64d)e.x * ldw + j, w);
#pragma unroll
        for (int i = 0; i < EPT; i++) acc[i] += (long long)e.y * w[i];
    }
    WT *dst = D + (i64d)t * ldc + dcol0 + j;
    int b[EPT];
    WideRow<WT>::load(dst, b);
#pragma unroll
    for (int i = 0; i < EPT; i++) b[i] = zp_reduce(F, acc[i] + b[i]);
    WideRow<WT>::store(dst, b);
    (void)nrows;
}

// The entries of the non-pivot rows on pivot columns, as (pivot index, value) lists (once per round: every slab of columns and the
// density estimate go along them), and the scatter of the other entries into the dense rows.
template <int TEAM>
__global__ void k_wd_pcount(int nrows, const int *__restrict__ rows, const i64d *__restrict__ start, const int *__restrict__ len,
                            const int2 *__restrict__ ent, const int *__restrict__ qinv_r, i64d *__restrict__ cnt)
{
    const int tl = threadIdx.x % TEAM;
    const int t = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (t > nrows) return;
    int c = 0;
    if (t < nrows) {
        const int row = rows ? rows[t] : t;
        const i64d st = start[row];
        const int ln = len[row];
        for (int k = tl; k < ln; k += TEAM) c += qinv_r[ent[st + k].x] >= 0;
        for (int o = TEAM / 2; o > 0; o >>= 1) c += __shfl_xor(c, o, TEAM);
    }
    if (tl == 0) cnt[t] = c;
}

template <int TEAM>
__global__ void k_wd_pfill(int nrows, const int *__restrict__ rows, const i64d *__restrict__ start, const int *__restrict__ len,
                           const int2 *__restrict__ ent, const int *__restrict__ qinv_r, const i64d *__restrict__ off, int2 *__restrict__ plist)
{
    const int tl = threadIdx.x % TEAM;
    const int t = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (t >= nrows) return;
    const int row = rows ? rows[t] : t;
    const i64d st = start[row];
    const int ln = len[row];
    i64d pos = off[t];
    const u64d below = (1ull << tl) - 1ull;
    for (int k0 = 0; k0 < ln; k0 += TEAM) {
        const int k = k0 + tl;
        int2 e = make_int2(0, 0);
        int q = -1;
        if (k < ln) { e = ent[st + k]; q = qinv_r[e.x]; }
        const u64d m = team_ballot<TEAM>(q >= 0);
        if (q >= 0) plist[pos + __popcll(m & below)] = make_int2(q, e.y);
        pos += __popcll(m);
    }
}

// D[t][dcol0 + cmap_s[col]] = value for the entries on columns of the slab (D zero before; the columns of a row are distinct)
template <int TEAM, typename DT>
__global__ void k_wd_own(int nrows, const int *__restrict__ rows, const i64d *__restrict__ start, const int *__restrict__ len,
                         const int2 *__restrict__ ent, const int *__restrict__ qinv_r, const int *__restrict__ cmap_s, DT *__restrict__ D, i64d ldc,
                         int dcol0)
{
    const int tl = threadIdx.x % TEAM;
    const int t = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (t >= nrows) return;
    const int row = rows ? rows[t] : t;
    const i64d st = start[row];
    const int ln = len[row];
    for (int k = tl; k < ln; k += TEAM) {
        const int2 e = ent[st + k];
        if (qinv_r[e.x] >= 0) continue;
        const int j = cmap_s[e.x];
        if (j >= 0) D[(i64d)t * ldc + dcol0 + j] = (DT)e.y;
    }
}

// dense Schur rows: D[t][dcol0 + j] += sum a * W[q][j] over the row's list (q, a).  blockIdx.x = row slot t
template <bool SMALL, typename DT, typename WT>
__global__ __launch_bounds__(256) void k_wd_rows(int nrows, ZpField F, const i64d *__restrict__ poff, const int2 *__restrict__ plist,
                                                 const WT *__restrict__ Wd, i64d ldw, int Cs, DT *__restrict__ D, i64d ldc, int dcol0)
{
    const int t = blockIdx.x;
    const int j = (blockIdx.y * blockDim.x + threadIdx.x) * 4;
    if (j >= Cs) return;
    const i64d lo = poff[t], hi = poff[t + 1];
    if (lo == hi) return; // (the own entries are there already)
    DenseAcc<SMALL> acc;
    for (i64d k = lo; k < hi; k++) {
        const int2 e = plist[k]; // (uniform over the workgroup)
        acc.fma(F, e.y, dense_load4(Wd + (i64d)e.x * ldw + j));
    }
    DT *dst = D + (i64d)t * ldc + dcol0 + j;
    dense_store4(dst, acc.finish(F, dense_load4(dst)));
    (void)nrows;
}

__global__ void k_count_nonzero(i64d n, const int *__restrict__ v, u64d *__restrict__ out)
{
    u64d c = 0;
    for (i64d i = (i64d)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64d)gridDim.x * blockDim.x) c += v[i] != 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor((long long)c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// ------------------------------------------------------------------------------------------------
// The L factor (echelonize_opts.L, reference src/SpaSM.jl:331; struct spasm_lu.L :266): the multipliers x_b of every reduced row
// ARE its row of L (A[i] = sum_k L[i][k] U[k], :705-707), so a round only has to hand out the lists the solve kernels leave in
// the record pool: per row slot the non-zero (row of U, multiplier) pairs, compacted.
// ------------------------------------------------------------------------------------------------
__global__ void k_l_count(int nrows, const i64d *__restrict__ Lstart, const int *__restrict__ Llen, const int4 *__restrict__ Lpool, i64d *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    const int t = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (t > nrows) return;
    int c = 0;
    if (t < nrows) {
        const i64d ls = Lstart[t];
        const int ll = Llen[t];
        for (int i = lane; i < ll; i += 64) c += Lpool[ls + i].y != 0;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    }
    if (lane == 0) cnt[t] = c;
}

__global__ void k_l_fill(int nrows, int ubase, const i64d *__restrict__ Lstart, const int *__restrict__ Llen, const int4 *__restrict__ Lpool,
                         const int *__restrict__ Lidx, const i64d *__restrict__ off, int *__restrict__ oj, int *__restrict__ ox)
{
    const int lane = threadIdx.x & 63;
    const int t = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (t >= nrows) return;
    const i64d ls = Lstart[t];
    const int ll = Llen[t];
    i64d pos = off[t];
    const u64d below = (1ull << lane) - 1ull;
    for (int i0 = 0; i0 < ll; i0 += 64) {
        const int i = i0 + lane;
        int v = 0, q = 0;
        if (i < ll) { v = Lpool[ls + i].y; q = Lidx[ls + i]; }
        const u64d m = __ballot(v != 0);
        if (v != 0) {
            const i64d w = pos + __popcll(m & below);
            oj[w] = ubase + q;
            ox[w] = v;
        }
        pos += __popcll(m);
    }
}

// flag[j] = 1 for a column that holds entries (flag comes in from k_flag_cols) and carries no pivot of this round
__global__ void k_mask_pivot_cols(int m, const int *__restrict__ qinv_r, int *__restrict__ flag)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m && qinv_r[j] >= 0) flag[j] = 0;
    if (j == m) flag[j] = 0;
}

// every step-th element of a list
__global__ void k_pick_stride(int n, int step, const int *__restrict__ src, int *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[(i64d)i * step];
}

// out[i] = src[idx[i]]
__global__ void k_gather_int(int n, const int *__restrict__ idx, const int *__restrict__ src, int *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

__global__ void k_flag_nonneg(int n, const int *__restrict__ v, int *__restrict__ flag)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = v[i] >= 0;
    if (i == n) flag[i] = 0;
}

__global__ void k_flag_live(int n, const int *__restrict__ len, int *__restrict__ flag)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = len[i] > 0;
    if (i == n) flag[i] = 0;
}

// ------------------------------------------------------------------------------------------------
// kernel basis assembly (reference call site src/SpaSM.jl:879)
// ------------------------------------------------------------------------------------------------
__global__ void k_rows_from_ptr(int n, const i64d *__restrict__ p, i64d *__restrict__ start, int *__restrict__ len, int *__restrict__ orig)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    start[i] = p[i];
    len[i] = (int)(p[i + 1] - p[i]);
    orig[i] = i;
}

// entries of kernel vector f: the -1 on its free column plus one per non-zero multiplier
__global__ void k_kcount(int nfree, const i64d *__restrict__ Lstart, const int *__restrict__ Llen, const int4 *__restrict__ Lpool, i64d *__restrict__ klen)
{
    const int lane = threadIdx.x & 63;
    const int f = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (f > nfree) return;
    if (f == nfree) { if (lane == 0) klen[f] = 0; return; }
    const i64d ls = Lstart[f];
    const int ll = Llen[f];
    int cnt = 0;
    for (int i = lane; i < ll; i += 64) cnt += Lpool[ls + i].y != 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) klen[f] = (i64d)cnt + 1;
}

__global__ void k_kfill(int nfree, const int *__restrict__ freecol, const int *__restrict__ colof, const i64d *__restrict__ Lstart,
                        const int *__restrict__ Llen, const int4 *__restrict__ Lpool, const int *__restrict__ Lidx,
                        const i64d *__restrict__ kstart, int2 *__restrict__ Kent)
{
    const int lane = threadIdx.x & 63;
    const int f = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (f >= nfree) return;
    const i64d ls = Lstart[f];
    const int ll = Llen[f];
    i64d pos = kstart[f];
    if (lane == 0) Kent[pos] = make_int2(freecol[f], -1); // K[j] = -1 (reference test/runtests.jl:21: 42012 == -1 mod 42013)
    pos += 1;
    for (int i0 = 0; i0 < ll; i0 += 64) {
        const int i = i0 + lane;
        int4 r = make_int4(0, 0, 0, 0);
        if (i < ll) r = Lpool[ls + i];
        const u64d m = __ballot(r.y != 0);
        if (r.y != 0) Kent[pos + __popcll(m & lanemask_lt())] = make_int2(colof[Lidx[ls + i]], r.y);
        pos += __popcll(m);
    }
}

// X of the batched triangular solve X * U = B: row f = {(row of U of pivot idx, multiplier)}, zero multipliers dropped
__global__ void k_xcount(int n, const i64d *__restrict__ Lstart, const int *__restrict__ Llen, const int4 *__restrict__ Lpool, i64d *__restrict__ xlen)
{
    const int lane = threadIdx.x & 63;
    const int f = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (f > n) return;
    if (f == n) { if (lane == 0) xlen[f] = 0; return; }
    const i64d ls = Lstart[f];
    const int ll = Llen[f];
    int cnt = 0;
    for (int i = lane; i < ll; i += 64) cnt += Lpool[ls + i].y != 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) xlen[f] = (i64d)cnt;
}

__global__ void k_xfill(int n, const int *__restrict__ rowof, const i64d *__restrict__ Lstart, const int *__restrict__ Llen,
                        const int4 *__restrict__ Lpool, const int *__restrict__ Lidx, const i64d *__restrict__ xstart, int *__restrict__ oj,
                        int *__restrict__ ox)
{
    const int lane = threadIdx.x & 63;
    const int f = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (f >= n) return;
    const i64d ls = Lstart[f];
    const int ll = Llen[f];
    i64d pos = xstart[f];
    for (int i0 = 0; i0 < ll; i0 += 64) {
        const int i = i0 + lane;
        int4 r = make_int4(0, 0, 0, 0);
        if (i < ll) r = Lpool[ls + i];
        const u64d m = __ballot(r.y != 0);
        if (r.y != 0) {
            const i64d at = pos + __popcll(m & lanemask_lt());
            oj[at] = rowof[Lidx[ls + i]];
            ox[at] = r.y;
        }
        pos += __popcll(m);
    }
}

// ------------------------------------------------------------------------------------------------
// multi-GPU exchange of the elected pivot rows: a rank exports the pivot rows it owns,
// every rank imports the concatenation (rank-major, each part in ascending pivot index)
// ------------------------------------------------------------------------------------------------
// owned[idx] = 1 when pivot idx is a local row
__global__ void k_owned_flags(int npiv, int row_lo, int row_stride, int n, const int *__restrict__ pivrow, int *__restrict__ flag)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < npiv) flag[idx] = shard_local(pivrow[idx], row_lo, row_stride, n) >= 0;
    if (idx == npiv) flag[idx] = 0;
}

// header (pivot index, length) of each owned pivot row, in ascending pivot index
__global__ void k_export_hdr(int npiv, int row_lo, int row_stride, const int *__restrict__ flag, const int *__restrict__ scan, const int *__restrict__ pivrow,
                             const int *__restrict__ len, int2 *__restrict__ hdr, i64d *__restrict__ olen)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npiv || !flag[idx]) return;
    const int l = len[(pivrow[idx] - row_lo) / row_stride];
    hdr[scan[idx]] = make_int2(idx, l);
    olen[scan[idx]] = l;
}

template <int TEAM>
__global__ void k_export_rows(int nown, int row_lo, int row_stride, const int2 *__restrict__ hdr, const i64d *__restrict__ ooff, const int *__restrict__ pivrow,
                              const i64d *__restrict__ start, const int2 *__restrict__ ent, int2 *__restrict__ out)
{
    const int tl = threadIdx.x % TEAM;
    const int k = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (k >= nown) return;
    const int2 h = hdr[k];
    const i64d st = start[(pivrow[h.x] - row_lo) / row_stride];
    const i64d os = ooff[k];
    for (int i = tl; i < h.y; i += TEAM) out[os + i] = ent[st + i];
}

// imported headers -> row table of the pivot-row matrix PM and rowsrc[pivot index] = row of PM
__global__ void k_import_rows(int ntot, const int2 *__restrict__ hdr, const i64d *__restrict__ off, i64d *__restrict__ start, int *__restrict__ len,
                              int *__restrict__ orig, int *__restrict__ rowsrc)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ntot) return;
    const int2 h = hdr[k];
    start[k] = off[k];
    len[k] = h.y;
    orig[k] = k;
    rowsrc[h.x] = k;
}

__global__ void k_hdr_len64(int n, const int2 *__restrict__ hdr, i64d *__restrict__ out)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = hdr[k].y;
    if (k == n) out[k] = 0;
}

// ------------------------------------------------------------------------------------------------
// DENSE TAIL, blocked (p <= 2^24).  Panels of DPB columns are eliminated column by column exactly as
// above but only inside the panel, recording the elimination factors Lm[row][t] of the panel's t-th
// pivot.  The columns to the right are then brought up to date in two steps:
//   TRSM  u_t = inv_t * (row_{p_t} - sum_{s<t} Lm[p_t][s] u_s)        (the panel's pivot rows, sequential in t)
//   GEMM  row_i -= sum_t Lm[i][t] u_t  for every non-pivotal row i    (f64 MFMA: integers below 2^53 are exact)
// ------------------------------------------------------------------------------------------------
#define DPB 64
typedef double v4f64 __attribute__((ext_vector_type(4)));




// Two launches per column of a panel (were five: a launch costs ~4.5 us and a 6091-column tail has 6091 of each).
// (1) one workgroup: first live row with a non-zero in column c (leftmost-pivot rule, row order; the scan stops at the
//     first hit), then the panel part of that row scaled to a unit pivot -- 64 entries, the only part of the old
//     "scale" pass that does not need the whole column;
// (2) one workgroup per row: its multiplier (recorded in fcol / Lm for the TRSM + GEMM) and the elimination of the panel
//     part of the row.  Pivot rows are skipped, so nobody reads the row that (1) rewrote except through prow.
__global__ __launch_bounds__(1024) void k_panel_find(int c, int c0, int c1, int R, ZpField F, int *__restrict__ D, i64d ldc, int *__restrict__ is_piv,
                                                     int *__restrict__ pivrow_of_col, int *__restrict__ prow, int *__restrict__ pan_row,
                                                     int *__restrict__ pan_inv, DenseState *__restrict__ st)
{
    __shared__ int s_best;
    if (threadIdx.x == 0) s_best = INT_MAX;
    __syncthreads();
    int best = INT_MAX;
    for (int i = threadIdx.x; i < R && i < best; i += 1024)
        if (!is_piv[i] && D[(i64d)i * ldc + c] != 0) { best = i; break; }
    best = wave_min_i32(best);
    if ((threadIdx.x & 63) == 0 && best != INT_MAX) atomicMin(&s_best, best);
    __syncthreads();
    const int p = s_best;
    const int npp = (c == c0 ? 0 : st->npp) + (p != INT_MAX ? 1 : 0); // the panel's pivot count restarts with its first column
    int inv = 0, scaled = 0;
    const int j = c + (int)threadIdx.x;
    if (p != INT_MAX) {
        inv = zp_inverse(F, D[(i64d)p * ldc + c]);
        if (j < c1) scaled = zp_mul(F, inv, D[(i64d)p * ldc + j]);
    }
    __syncthreads(); // every thread has read st->npp and the unscaled pivot entry
    if (p != INT_MAX && j < c1) { prow[j] = scaled; D[(i64d)p * ldc + j] = scaled; }
    if (threadIdx.x == 0) {
        st->npp = npp;
        if (p != INT_MAX) {
            st->cur = p;
            st->npiv += 1;
            is_piv[p] = 1;
            pivrow_of_col[c] = p;
            pan_row[npp - 1] = p;
            pan_inv[npp - 1] = inv;
        } else {
            st->cur = -1;
            pivrow_of_col[c] = -1;
        }
    }
}

__global__ __launch_bounds__(64) void k_panel_elim2(int c, int c1, ZpField F, int *__restrict__ D, i64d ldc, const int *__restrict__ is_piv,
                                                   const int *__restrict__ prow, int *__restrict__ fcol, double *__restrict__ Lm,
                                                   const DenseState *__restrict__ st)
{
    if (st->cur < 0) return;
    const int i = blockIdx.x;
    const int f = is_piv[i] ? 0 : D[(i64d)i * ldc + c]; // read by every lane before lane 0 overwrites it below (same wave, in order)
    if (threadIdx.x == 0) {
        fcol[i] = f;
        Lm[(size_t)i * DPB + (st->npp - 1)] = (double)f;
    }
    if (f == 0) return;
    const int nf = zp_neg(F, f);
    const int j = c + threadIdx.x;
    if (j < c1) {
        int *d = D + (i64d)i * ldc + j;
        *d = zp_axpy(F, nf, prow[j], *d);
    }
}

// TRSM of a panel: Upan[t][j] = inv_t * (D[p_t][j] - sum_{s<t} Lm[p_t][s] * Upan[s][j]) for the columns j right of the panel.
// One thread per column; the panel's 64 x 64 multipliers sit in LDS, the column's 64 results stay in registers, the sums are
// exact in f64 (64 (p/2)^2 < 2^53 for p <= 2^24).  The former version re-read Upan from global memory 2016 times per thread.
__global__ __launch_bounds__(64) void k_panel_trsm2(int c1, int C, ZpField F, int *__restrict__ D, i64d ldc, const double *__restrict__ Lm,
                                                   const int *__restrict__ pan_row, const int *__restrict__ pan_inv, double *__restrict__ Upan, i64d ldu,
                                                   const DenseState *__restrict__ st)
{
    __shared__ double s_l[DPB * DPB];
    __shared__ int s_row[DPB], s_inv[DPB];
    const int npp = st->npp;
    for (int e = threadIdx.x; e < npp * DPB; e += 64) {
        const int t = e / DPB, s = e % DPB;
        s_l[e] = Lm[(size_t)pan_row[t] * DPB + s];
    }
    if ((int)threadIdx.x < npp) { s_row[threadIdx.x] = pan_row[threadIdx.x]; s_inv[threadIdx.x] = pan_inv[threadIdx.x]; }
    __syncthreads();
    const int j = c1 + blockIdx.x * 64 + threadIdx.x;
    if (j >= C) return;
    double u[DPB];
#pragma unroll
    for (int t = 0; t < DPB; t++) {
        u[t] = 0.0;
        if (t < npp) { // uniform
            double acc = (double)D[(i64d)s_row[t] * ldc + j];
#pragma unroll
            for (int s = 0; s < t; s++) acc = fma(-s_l[t * DPB + s], u[s], acc);
            const int v = zp_mul(F, s_inv[t], zp_reduce(F, (long long)acc));
            u[t] = (double)v;
            Upan[(i64d)t * ldu + j] = u[t];
            D[(i64d)s_row[t] * ldc + j] = v;
        }
    }
}


// one thread per column to the right of the panel; Upan[t][j] (f64) and D[p_t][j] receive the normalised pivot rows

// D[i][j] -= sum_t Lm[i][t] * Upan[t][j] for the non-pivotal rows; workgroup tile 64 x 64, each wave 16 rows x 64 columns,
// v_mfma_f64_16x16x4_f64: A[l&15][k = l>>4], B[k = l>>4][l&15], C/D col = l&15, row = (l>>4) + 4*reg
__global__ __launch_bounds__(256) void k_dense_gemm(int c1, int R, int C, ZpField F, int *__restrict__ D, i64d ldc, const double *__restrict__ Lm,
                                                   const double *__restrict__ Upan, i64d ldu, const int *__restrict__ is_piv,
                                                   const DenseState *__restrict__ st)
{
    if (st->npp == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 64 + wave * 16; // rows on x: gridDim.y stops at 65535, a tail can have millions of rows
    const int cb = c1 + blockIdx.y * 64;
    v4f64 acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const double *la = Lm + (size_t)(r0 + (lane & 15)) * DPB + (lane >> 4);   // Lm is padded to a multiple of 64 rows
    const double *ub = Upan + (i64d)(lane >> 4) * ldu + cb + (lane & 15);      // Upan is padded to a multiple of 64 columns
    for (int k0 = 0; k0 < DPB; k0 += 4) {
        const double a = la[k0];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const double b = ub[(i64d)k0 * ldu + q * 16];
            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int col = cb + q * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = r0 + (lane >> 4) + 4 * r;
            if (row < R && col < C && !is_piv[row]) {
                int *d = D + (i64d)row * ldc + col;
                *d = zp_reduce(F, (long long)*d - (long long)acc[q][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Field arithmetic as the kernels use it, one lane per test vector (tests/test_gpu_zp.py compares with Python integers;
// reference src/SpaSM.jl:383-390).  out[8 * i ..] = mul, axpy, add, sub, neg, inverse, lazy product reduced
// (mul_lazy + acc_reduce_short: the scatter kernels' path), sum of 64 lazy products reduced (acc_reduce: the hash tables' path).
// ------------------------------------------------------------------------------------------------
template <bool SMALL>
__global__ void k_zp_probe(ZpField F, int n, const int *__restrict__ a, const int *__restrict__ b, const int *__restrict__ c, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = a[i], y = b[i], z = c[i];
    int *o = out + (size_t)8 * i;
    o[0] = zp_mul(F, x, y);
    o[1] = zp_axpy(F, x, y, z);
    o[2] = zp_add(F, x, y);
    o[3] = zp_sub(F, x, y);
    o[4] = zp_neg(F, x);
    o[5] = x != 0 ? zp_inverse(F, x) : 0;
    o[6] = acc_reduce_short<SMALL>(F, ZpAcc<SMALL>::mul_lazy(F, x, y));
    typename ZpAcc<SMALL>::type acc = 0;
    for (int k = 0; k < 64; k++) acc += ZpAcc<SMALL>::mul_lazy(F, x, y);
    o[7] = acc_reduce<SMALL>(F, acc);
}

