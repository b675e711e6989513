// The third structural pivot search: "greedy alternating cycle-free search" (reference README.md:23, the third line of the
// [pivots] log of spasm_pivots_extract_structural, prototype src/SpaSM.jl:776-778; switch enable_greedy_pivot_search, :326).
//
// libspasm's source is not in the reference tree.  What its log names is the search of Bouillaguet, Delaplace, Voge, "Parallel
// sparse PLUQ factorization modulo p" (PASCO 2017): a non-pivot row i may take a pivot on one of its columns j that carries no
// pivot, as long as the pivots stay permutable to a triangular matrix -- i.e. as long as no pivot row REACHABLE from row i (through
// the pivot columns row i holds, then the pivot columns those rows hold, ...) has an entry in column j; libspasm runs a breadth-
// first search per row and lets rows of different threads check each other's new pivots in a critical section.  Here all rows
// search at once and the result does not depend on any order (so the CPU oracle and the plain-Python rule of tests/ take the same
// pivots):
//   pass (up to GREEDY_PASSES, until one accepts nothing), for every non-empty row i that is no pivot row and has at most GR_MAXLEN
//   entries:
//     reach(i)    the pivots reachable from row i; a row that reaches more than GR_BUDGET pivots sits the pass out;
//     touched(i)  the columns without pivot that the rows of reach(i) hold;
//     candidates  the columns of row i without pivot that are not in touched(i) and that at most GREEDY_OCC_MAX rows without pivot
//                 hold (occ[c], counted over the rows that are no pivot rows at the start of the pass); the row PROPOSES the one of
//                 smallest occupancy (ties: leftmost), j*(i);
//     winner[c]   per proposed column the smallest key (row length, row) among its proposers;
//     a winner x is ACCEPTED unless some column c != j*(x) of  full(x) = (columns of row x without pivot) + touched(x)  has a
//     winner with a SMALLER key: among the accepted rows an edge x -> y (column j*(y) in full(x)) then always goes to a larger key,
//     so they cannot close a cycle, whatever paths through the old pivots connect them; a path from x back to x through old
//     pivots alone would put j*(x) in touched(x).
//   The accepted rows are pivot rows on their columns from the next pass on.
// When the search finds anything ALL pivots of the round are renumbered in a topological order: level 0 = rows that hold no other
// pivot column, level l = 1 + the deepest level among the pivot columns the row holds; descending level, ascending column inside a
// level (a row only holds pivot columns of lower levels: they come after it, which is what U_PP's consumers expect).
//
// One wave per row; visited set, queue and the row's own columns in LDS.  Latency-bound pointer chasing like libspasm's, but
// hundreds of thousands of rows at once; rows whose candidates are all dead stop early (most do).
#pragma once

#include "kernels.hpp"

#define GREEDY_PASSES 3
#define GR_MAXLEN 256     // longest row that searches (its columns live in LDS)
#define GR_BUDGET 1024    // most pivots a searching row may reach: capacity of the queue (the limit in force is GREEDY_REACH_MAX)
// The two limits that keep the search from buying pivots with fill-in (measured, DESIGN.md section 2: without them the Schur
// complement of config 3 at 1/4 grows 13-fold for 806 more pivots): a pivot on column c must be applied to every row that holds c,
// directly or through the pivot rows that hold it, and it drags the whole reach of its row along.  Defaults: free pivots only.
//   GREEDY_REACH_MAX  a row that reaches more pivots than this sits the pass out   (env SPASM_AMD_GREEDY_REACH_MAX, <= GR_BUDGET)
//   GREEDY_OCC_MAX    a column more rows without pivot than this hold is no candidate  (env SPASM_AMD_GREEDY_OCC_MAX)
#define GREEDY_REACH_MAX_DEFAULT 2
#define GREEDY_OCC_MAX_DEFAULT 1
#define GR_VIS 2048       // slots of the visited set (load <= 1/2)
#define GR_OWN 512        // slots of the table of the row's own pivot-free columns
#define GR_WPB 4

struct GreedyWave {
    int vis[GR_VIS];
    int queue[GR_BUDGET];
    int own[GR_OWN];      // column, -1: empty
    int alive[GR_OWN];
    int tail, nalive, overflow, reject;
};

__device__ __forceinline__ unsigned gr_hash(int x) { return (unsigned)x * 2654435761u; }

// MODE 1: proposals (prop[i] = j* or -1; best2[j*] = min key).  MODE 2: acceptance of the winners (accept[i] = 1).
template <int MODE>
__global__ __launch_bounds__(64 * GR_WPB) void k_greedy(int n, const int *__restrict__ is_piv, const i64d *__restrict__ start, const int *__restrict__ len,
                                                        const int2 *__restrict__ ent, const int *__restrict__ qinv_r, const int *__restrict__ pivrow,
                                                        u64d *__restrict__ best2, int *__restrict__ prop, int *__restrict__ accept,
                                                        const int *__restrict__ colcnt, int occ_max, int reach_max)
{
    __shared__ GreedyWave s_w[GR_WPB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * GR_WPB + wave;
    if (i >= n) return;
    GreedyWave &W = s_w[wave];
    // (what one lane writes and another reads between two wave barriers goes through volatile accesses: LDS operations of a wave
    // complete in order, the compiler must only be kept from holding the values in registers)
    volatile int *v_queue = W.queue, *v_own = W.own, *v_alive = W.alive;
    volatile int *v_tail = &W.tail, *v_nalive = &W.nalive, *v_overflow = &W.overflow, *v_reject = &W.reject;
    const int ln = len[i];
    const bool searches = !is_piv[i] && ln > 0 && ln <= GR_MAXLEN;
    int jstar = -1;
    u64d mykey = 0;
    if (MODE == 1) {
        if (!searches) { if (lane == 0) prop[i] = -1; return; }
    } else {
        if (lane == 0) accept[i] = 0;
        if (!searches) return;
        jstar = prop[i];
        if (jstar < 0) return;
        mykey = ((u64d)(unsigned)ln << 32) | (u64d)(unsigned)i;
        if (best2[jstar] != mykey) return; // another row won the column
    }
    if (MODE == 1) mykey = ((u64d)(unsigned)ln << 32) | (u64d)(unsigned)i;
    for (int k = lane; k < GR_VIS; k += 64) W.vis[k] = -1;
    for (int k = lane; k < GR_OWN; k += 64) { W.own[k] = -1; W.alive[k] = 0; }
    if (lane == 0) { W.tail = 0; W.nalive = 0; W.overflow = 0; W.reject = 0; }
    __builtin_amdgcn_wave_barrier();
    // a pivot enters the visited set and, when it is new, the queue
    auto visit = [&](int q) {
        if (*v_overflow) return; // (so that at most GR_BUDGET + 64 pivots ever enter the set of GR_VIS slots)
        unsigned h = gr_hash(q) & (GR_VIS - 1);
        for (;;) {
            const int old = atomicCAS(&W.vis[h], -1, q);
            if (old == q) return;
            if (old == -1) {
                const int pos = atomicAdd(&W.tail, 1);
                if (pos < reach_max) v_queue[pos] = q;
                else *v_overflow = 1;
                return;
            }
            h = (h + 1) & (GR_VIS - 1);
        }
    };
    // the row itself
    const i64d st = start[i];
    for (int k = lane; k < ln; k += 64) {
        const int c = ent[st + k].x;
        const int q = qinv_r[c];
        if (q >= 0) visit(q);
        else if (MODE == 1) {
            unsigned h = gr_hash(c) & (GR_OWN - 1);
            while (atomicCAS(&W.own[h], -1, c) != -1) h = (h + 1) & (GR_OWN - 1); // (the columns of a row are distinct)
            // (a column more than occ_max rows without pivot hold is no candidate: its pivot would have to be applied to all of them)
            if (colcnt[c] <= occ_max) {
                v_alive[h] = 1;
                atomicAdd(&W.nalive, 1);
            }
        } else if (c != jstar && best2[c] < mykey) *v_reject = 1;
    }
    __builtin_amdgcn_wave_barrier();
    // breadth first: four pivot rows at a time, 16 lanes each
    const int team = lane >> 4, tl = lane & 15;
    int head = 0;
    for (;;) {
        const int tail = min(*v_tail, reach_max);
        if (head >= tail || *v_overflow) break;
        if (MODE == 1 && *v_nalive <= 0) break;
        if (MODE == 2 && *v_reject) break;
        const int mine = head + team;
        if (mine < tail) {
            const int r = pivrow[v_queue[mine]];
            const i64d rs = start[r];
            const int rl = len[r];
            for (int k = tl; k < rl; k += 16) {
                if (*v_overflow) break;
                const int c = ent[rs + k].x;
                const int q = qinv_r[c];
                if (q >= 0) visit(q);
                else if (MODE == 1) {
                    unsigned h = gr_hash(c) & (GR_OWN - 1);
                    for (;;) {
                        const int o = v_own[h];
                        if (o == -1) break;
                        if (o == c) { if (atomicExch(&W.alive[h], 0) == 1) atomicSub(&W.nalive, 1); break; }
                        h = (h + 1) & (GR_OWN - 1);
                    }
                } else if (c != jstar && best2[c] < mykey) *v_reject = 1;
            }
        }
        head = min(head + 4, tail);
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
    if (MODE == 1) {
        // the surviving candidate of smallest occupancy (ties: leftmost)
        u64d bestk = ~0ull;
        if (!*v_overflow && *v_nalive > 0)
            for (int k = lane; k < GR_OWN; k += 64)
                if (v_alive[k]) bestk = min(bestk, ((u64d)(unsigned)colcnt[v_own[k]] << 32) | (u64d)(unsigned)v_own[k]);
        for (int o = 32; o > 0; o >>= 1) bestk = min(bestk, (u64d)__shfl_xor((long long)bestk, o));
        if (lane == 0) {
            const int choice = bestk == ~0ull ? -1 : (int)(unsigned)(bestk & 0xffffffffull);
            prop[i] = choice;
            if (choice >= 0) atomicMin(&best2[choice], mykey);
        }
    } else {
        if (lane == 0) accept[i] = (!*v_overflow && !*v_reject) ? 1 : 0;
    }
}

// the rows a pass accepted become pivots behind the ones there are: pivrow / pivcol / qinv_r / is_piv
__global__ void k_greedy_record(int n, int npiv0, const int *__restrict__ accept, const int *__restrict__ ascan, const int *__restrict__ prop, int *__restrict__ pivrow,
                                int *__restrict__ pivcol, int *__restrict__ qinv_r, int *__restrict__ is_piv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !accept[i]) return;
    const int idx = npiv0 + ascan[i];
    pivrow[idx] = i;
    pivcol[idx] = prop[i];
    qinv_r[prop[i]] = idx;
    is_piv[i] = 1;
}

// level of a pivot = 0 when its row holds no other pivot column, else 1 + the deepest among them
template <int TEAM>
__global__ void k_piv_relax(int npiv, const int *__restrict__ pivrow, const int *__restrict__ pivcol, const i64d *__restrict__ start, const int *__restrict__ len,
                            const int2 *__restrict__ ent, const int *__restrict__ qinv_r, int *__restrict__ lev, int *__restrict__ changed)
{
    const int tl = threadIdx.x % TEAM;
    const int idx = (int)(((i64d)blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (idx >= npiv) return;
    const int row = pivrow[idx], pc = pivcol[idx];
    const i64d st = start[row];
    const int ln = len[row];
    int l = 0;
    for (int k = tl; k < ln; k += TEAM) {
        const int c = ent[st + k].x;
        if (c == pc) continue;
        const int q = qinv_r[c];
        if (q >= 0) l = max(l, __hip_atomic_load(&lev[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1);
    }
    for (int o = TEAM / 2; o > 0; o >>= 1) l = max(l, __shfl_xor(l, o, TEAM));
    if (tl == 0 && l > lev[idx]) {
        __hip_atomic_store(&lev[idx], l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (changed) *changed = 1;
    }
}

__global__ void k_piv_keys(int npiv, const int *__restrict__ lev, const int *__restrict__ pivcol, u64d *__restrict__ keys, int *__restrict__ iota)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npiv) return;
    keys[idx] = ((u64d)(unsigned)(0x7fffffff - lev[idx]) << 32) | (u64d)(unsigned)pivcol[idx];
    iota[idx] = idx;
}

__global__ void k_piv_permute(int npiv, const int *__restrict__ order, const int *__restrict__ pivrow_in, const int *__restrict__ pivcol_in, int *__restrict__ pivrow,
                              int *__restrict__ pivcol, int *__restrict__ qinv_r)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= npiv) return;
    const int o = order[k];
    pivrow[k] = pivrow_in[o];
    pivcol[k] = pivcol_in[o];
    qinv_r[pivcol_in[o]] = k;
}
