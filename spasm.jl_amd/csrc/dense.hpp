// Dense finish for primes below 2^16 (BASELINE configs 2, 3, 5: p = 65521 and p = 127) -- device code.
//
// Replaces, for those primes, the f64 panels of kernels.hpp (libspasm's spasm_ffpack_rref / spasm_ffpack_LU finish over the
// datatype spasm_datatype_choose picks, prototypes reference src/SpaSM.jl:805-812, enum :373).  Same elimination -- columns left
// to right, the pivot of a column is the FIRST row, not yet a pivot, that holds a non-zero in it -- organised in two levels:
//
//   panel (64 columns)   copied transposed into P[64][rows] (same element type as D); ONE persistent cooperative kernel eliminates it column by column
//                        with the rows of every workgroup resident in LDS (as bytes / shorts) and one grid barrier per column
//                        (k_panel_lu): every workgroup publishes its bid and its candidate row in a 320-byte record of
//                        write-through stores, arrives on one of 64 counters, and reads the winner's record behind the barrier.
//                        The multipliers stay in P where they were read, as in an in-place LU.
//   block (KB columns)   the panels of a block update only the columns of the block (K = 64); the columns right of the block
//                        are updated once per block with K = KB.
//   updates              rows that became pivots: triangular solve among the 64 of a panel (k_trsm_i8); everybody else: an int8
//                        MFMA GEMM (k_gemm_i8, v_mfma_i32_32x32x32_i8, operands staged in LDS).  Residues are split in signed
//                        base-256 digits -- one for p < 2^8, two for p < 2^16 (three accumulators: d0*d0, d0*d1 + d1*d0, d1*d1)
//                        -- and recombined in 64 bits, so the result is exact.
// The multipliers F[row][slot] and the normalised pivot rows Ut[column][slot] are kept as digit planes with the K index
// contiguous, the layout both MFMA operands want (16 consecutive k per lane).  The dense matrix D itself is kept as bytes
// (p < 2^8) or shorts (template parameter DT): its block updates are bound by reading and writing it.
#pragma once

#include "kernels.hpp"

#define DP_W 64            // columns of a panel
#define DP_REC 80          // ints of a candidate record: the row (64), the inverse of its leading entry, the bid (row index)
#define DP_NONE 0x7fffffff

typedef int v16i32 __attribute__((ext_vector_type(16)));

struct PanelInfo {          // one per panel of the current block; written by k_panel_lu
    int npp;                // pivots found in the panel
    int gbase;              // pivots found before it (sequence number of its first pivot)
    int pad0, pad1;
    int row[DP_W];          // pivot rows in election order, -1 beyond npp
    int col[DP_W];          // their columns inside the panel
    int inv[DP_W];          // inverse of the pivot value
    int mtri[DP_W * DP_W];  // mtri[t * 64 + s] = multiplier of pivot row t at pivot s < t (0 elsewhere)
};

#define DP_NCTR 64           // arrival counters of the grid barrier (workgroup b adds to counter b % 64), 128 bytes apart
struct PanelSync {          // reset by k_panel_load before every k_panel_lu launch
    unsigned arrive[DP_NCTR * 32];
    unsigned timeout;       // set when a bounded spin gave up (the launch is then reported as failed)
    unsigned pad[3];
};

// ---- P[j][i] = D[i][c0 + j] (j < w; 0 beyond), i < R; rows R .. Rp-1 are zero
template <typename DT>
__global__ __launch_bounds__(256) void k_panel_load(int R, int Rp, int c0, int w, const DT *__restrict__ D, i64d ldc, DT *__restrict__ P, PanelSync *sy)
{
    __shared__ int tile[DP_W][DP_W + 1];
    const int i0 = blockIdx.x * 64;
    if (blockIdx.x == 0) { // the words the panel kernel polls, reset in stream order before it
        if (threadIdx.x < DP_NCTR) sy->arrive[threadIdx.x * 32] = 0;
        if (threadIdx.x == 0) sy->timeout = 0;
    }
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int r = idx >> 6, j = idx & 63;
        const int i = i0 + r;
        tile[j][r] = (i < R && j < w) ? (int)D[(i64d)i * ldc + c0 + j] : 0;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int j = idx >> 6, r = idx & 63;
        if (i0 + r < Rp) P[(i64d)j * Rp + i0 + r] = (DT)tile[j][r];
    }
}

// grid barrier number `epoch` (1, 2, ...) of a launch whose `nblocks` workgroups are all resident (cooperative launch).  No fences:
// everything one workgroup hands to another in k_panel_lu is written with agent-scope atomic stores (write-through) and read with
// agent-scope atomic loads (guide, guideline 16, forms R1 / R2); every storing wave drains its stores before the workgroup's barrier,
// one lane then arrives and polls.  A fence pair (release: L2 write-back, acquire: invalidate) cost ~30 us per column instead.
// Every workgroup's arrival used to be an atomic add on ONE word, and its bid an atomicMin on another: 2 x 256 read-modify-writes
// on two addresses serialise at the memory side, 7-9 us per column.  Now: 64 arrival counters (at most 4 adds each), polled by
// the 64 lanes of wave 0 with one load per poll; the bids are plain write-through stores next to the candidate rows
// (record word DP_W + 1), read once after the barrier, four per lane, and reduced in the wave.
// Returns the winning bid (DP_NONE: no pivot in the column), or -1 when a spin timed out.  The bids are read AFTER the full
// counts have been seen (dependent, later loads): a workgroup's record is complete before its arrival (vmcnt(0), barrier, then
// the add), so they are final then.  (Loading bids and counters in one poll is WRONG: the loads can be served out of order,
// workgroups then disagree on the winner.)
__device__ __forceinline__ int panel_grid_barrier(PanelSync *sy, unsigned epoch, unsigned nblocks, const int *recs)
{
    __shared__ int s_win;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) __hip_atomic_fetch_add(&sy->arrive[(blockIdx.x % DP_NCTR) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // counter `lane` takes the workgroups lane, lane + 64, ...
        const unsigned mine = lane < (int)nblocks ? (nblocks - lane + DP_NCTR - 1) / DP_NCTR : 0;
        const unsigned target = epoch * mine;
        int win = -1;
        unsigned spins = 0;
        for (;;) {
            const unsigned a = mine ? __hip_atomic_load(&sy->arrive[lane * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (__all(a >= target)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                int best = DP_NONE;
                for (unsigned b = lane; b < nblocks; b += 64)
                    best = min(best, __hip_atomic_load(&recs[(size_t)b * DP_REC + DP_W + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                win = wave_min_i32(best);
                break;
            }
            if (++spins > (1u << 24) || __hip_atomic_load(&sy->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { // (~ seconds)
                __hip_atomic_store(&sy->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (lane == 0) s_win = win;
    }
    __syncthreads();
    return s_win;
}

// inverses of all residues of a prime < 2^16 (one table per dense elimination): invtab[a], 0 < a < p
__device__ __forceinline__ int zp_inverse_small(int p, int a);
__global__ void k_inv_table(int p, int *__restrict__ invtab);

// a^-1 mod p for p < 2^16 in 32-bit arithmetic (the f64 / i64 Euclid of zp_inverse costs ~10 us on one lane, once per column of a
// panel, on the critical path of every workgroup); a is a non-zero balanced residue
__device__ __forceinline__ int zp_inverse_small(int p, int a)
{
    unsigned r0 = (unsigned)p, r1 = (unsigned)(a < 0 ? a + p : a);
    int s0 = 0, s1 = 1;
    while (r1 != 0) {
        const unsigned q = r0 / r1;
        const unsigned t = r0 - q * r1; r0 = r1; r1 = t;
        const int u = s0 - (int)q * s1; s0 = s1; s1 = u; // |s| <= p throughout
    }
    int r = s0 % p;
    if (r < 0) r += p;
    return r > p / 2 ? r - p : r;
}

__global__ void k_inv_table(int p, int *__restrict__ invtab)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < p) invtab[a] = a ? zp_inverse_small(p, a) : 0;
}

// x + a * b for balanced residues of a prime < 2^16, result balanced: one lazy reduction and a correction either way
__device__ __forceinline__ int zp_axpy_small(const ZpField &F, int a, int b, int x)
{
    int r = zp_small_lazy(__mul24(a, b) + x, -F.finvp, (int)F.p);
    if (r > (int)F.halfp) r -= (int)F.p;
    else if (r < (int)F.mhalfp) r += (int)F.p;
    return r;
}

// ---- the panel, one cooperative launch of NT-thread workgroups.  Workgroup b owns rows [b * chunk, (b + 1) * chunk) of P; with
// INLDS they live in LDS as X[j * chunk + r] for the whole launch (chunk * 256 bytes), otherwise the kernel works on P in place
// (L2 / MALL resident; only the owner of a row ever reads or writes it).  seq[i] = sequence number of the pivot row i became,
// -1 while it is none.  candrow: 2 x gridDim.x records of DP_REC ints (the candidate row, the inverse of its entry in the current
// column, the bid).
// XT: how a residue is kept, in P and in LDS -- signed char for p < 2^8, short for p < 2^16 (balanced residues fit), so that
// 2304 / 1152 rows per workgroup stay resident in LDS instead of 576, and the in-place variant (more rows than that: it is bound by
// streaming P through L2 / HBM once per column) moves a quarter / half of the bytes.
template <bool INLDS, int NT, typename XT>
__global__ __launch_bounds__(NT) void k_panel_lu(int Rp, int chunk, int w, int c0, ZpField F, XT *__restrict__ P, int *__restrict__ seq,
                                                 int *__restrict__ pivrow_of_col, PanelInfo *__restrict__ info, PanelSync *sy, int *candrow,
                                                 DenseState *st, const int *__restrict__ invtab, unsigned long long *stamps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dynx[]; // INLDS: chunk * 64 residues
    XT *s_x = (XT *)s_dynx;
#define PSTAMP(k) do { if (stamps && blockIdx.x == 0 && tid == 0) stamps[c * 8 + (k)] = wall_clock64(); } while (0)
    __shared__ int s_prow[DP_W];
    __shared__ int s_cand;
    const int tid = threadIdx.x;
    const int base = blockIdx.x * chunk;
    const i64d xs = INLDS ? (i64d)chunk : (i64d)Rp; // stride between the columns of the row storage
    XT *X = INLDS ? s_x : P + base;
    if (INLDS) {
        for (int j = 0; j < DP_W; j++)
            for (int r = tid; r < chunk; r += NT) s_x[j * chunk + r] = P[(i64d)j * Rp + base + r];
    }
    // the status of the rows this thread owns (r = tid + NT k): bit k set = not a pivot (yet)
    unsigned long long live = 0;
    const int nmine = tid < chunk ? (chunk - tid + NT - 1) / NT : 0;
    for (int k = 0; k < nmine; k++)
        if (seq[base + tid + NT * k] < 0) live |= 1ull << k;
    const int gbase = st->npiv;
    int npp = 0;
    __syncthreads();
    bool alive = true;
    for (int c = 0; c < w && alive; c++) {
        // ---- election: my first live row with a non-zero in column c
        PSTAMP(0);
        if (tid == 0) s_cand = DP_NONE;
        __syncthreads();
        int mine = DP_NONE;
        {
            unsigned long long m = live;
            while (m) {
                const int k = __ffsll((long long)m) - 1;
                m &= m - 1;
                if (X[(i64d)c * xs + tid + NT * k] != 0) { mine = base + tid + NT * k; break; }
            }
        }
        mine = wave_min_i32(mine);
        if ((tid & 63) == 0 && mine != DP_NONE) atomicMin(&s_cand, mine);
        __syncthreads();
        const int wg_cand = s_cand;
        PSTAMP(1);
        int *rec = candrow + (size_t)((c & 1) * gridDim.x + blockIdx.x) * DP_REC;
        if (tid == 2 * DP_W) __hip_atomic_store(&rec[DP_W + 1], wg_cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // the bid, always
        if (wg_cand != DP_NONE) {
            // the candidate's row (unscaled) and the inverse of its leading entry where every workgroup can read them, then the bid
            // (two sets of records, by column parity: a workgroup may bid for column c + 1 while another still reads column c's)
            if (tid < DP_W) __hip_atomic_store(&rec[tid], (int)X[(i64d)tid * xs + (wg_cand - base)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == DP_W) {
                const int lead = (int)X[(i64d)c * xs + (wg_cand - base)];
                __hip_atomic_store(&rec[DP_W], invtab[lead < 0 ? lead + (int)F.p : lead], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        PSTAMP(2);
        const int p = panel_grid_barrier(sy, (unsigned)c + 1, gridDim.x, candrow + (size_t)((c & 1) * gridDim.x) * DP_REC);
        PSTAMP(3);
        if (p < 0) { alive = false; continue; }
        if (p == DP_NONE) continue; // no pivot in this column
        const int owner = p / chunk;
        int *orec = candrow + (size_t)((c & 1) * gridDim.x + owner) * DP_REC;
        const int inv = __hip_atomic_load(&orec[DP_W], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < DP_W) {
            const int raw = __hip_atomic_load(&orec[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_prow[tid] = tid > c && tid < w ? zp_axpy_small(F, inv, raw, 0) : 0;
        }
        __syncthreads();
        PSTAMP(4);
        // ---- elimination of my live rows; the multiplier stays in column c
        {
            unsigned long long m = live;
            while (m) {
                const int k = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int r = tid + NT * k;
                if (base + r == p) { live &= ~(1ull << k); continue; } // the new pivot row: frozen from here on
                const int f = (int)X[(i64d)c * xs + r];
                if (f == 0) continue;
                const int nf = -f;
                for (int j = c + 1; j < w; j++) {
                    XT *x = &X[(i64d)j * xs + r];
                    *x = (XT)zp_axpy_small(F, nf, s_prow[j], (int)*x);
                }
            }
        }
        if (blockIdx.x == owner && tid == 0) {
            seq[p] = gbase + npp;
            pivrow_of_col[c0 + c] = p;
            info->row[npp] = p;
            info->col[npp] = c;
            info->inv[npp] = inv;
        }
        npp++;
        __syncthreads(); // s_prow is rewritten by the next column
        PSTAMP(5);
    }
    if (INLDS) {
        __syncthreads();
        for (int j = 0; j < DP_W; j++)
            for (int r = tid; r < chunk; r += NT) P[(i64d)j * Rp + base + r] = s_x[j * chunk + r];
    }
    if (blockIdx.x == 0 && tid == 0) {
        info->npp = npp;
        info->gbase = gbase;
        for (int t = npp; t < DP_W; t++) { info->row[t] = -1; info->col[t] = -1; info->inv[t] = 0; }
        st->npp = npp;
        st->npiv = gbase + npp;
        if (!alive) st->pad = 1; // a grid barrier timed out: the host reports the elimination as failed
    }
}

// ---- tall matrices: more rows than the LDS of the chip holds (config 5 at 1/5: 666 k rows against 590 k).  The in-place variant of
// k_panel_lu streams every row's 64 panel entries through L2 once per COLUMN (2 KB per row and panel).  Instead the pivots of a
// panel are elected among the rows that DO fit (the first ones: the leftmost rule prefers them anyway) by the LDS-resident
// kernel, and the rows beyond them FOLLOW: with the panel's pivots known, a row's elimination needs no election and no barrier --
// it is loaded once, eliminated against the normalised pivot rows column by column exactly as a resident row is, stored once
// (128 bytes per row and panel).  The one thing a follower cannot do is become a pivot: when a column that found no pivot among
// the resident rows holds a non-zero in a follower, *flag is set and the host redoes the panel in place over all rows
// (never seen on a matrix tall enough to have followers; tested with a planted column).
template <int NT, typename XT>
__global__ __launch_bounds__(NT) void k_panel_follow(int Rp, int row0, int chunk, int w, ZpField F, XT *__restrict__ P, const int *__restrict__ seq,
                                                     const PanelInfo *__restrict__ info, int *__restrict__ flag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dynf[]; // chunk * 64 residues
    XT *X = (XT *)s_dynf;
    __shared__ int s_T[DP_W][DP_W]; // s_T[c][j]: the normalised pivot row of column c at column j > c (0 where the column has no pivot)
    __shared__ int s_has[DP_W];
    const int tid = threadIdx.x;
    const int base = row0 + blockIdx.x * chunk;
    const int nloc = min(chunk, Rp - base);
    if (nloc <= 0) return;
    for (int idx = tid; idx < DP_W * DP_W; idx += NT) s_T[idx >> 6][idx & 63] = 0;
    if (tid < DP_W) s_has[tid] = 0;
    __syncthreads();
    const int npp = info->npp;
    for (int idx = tid; idx < npp * DP_W; idx += NT) {
        const int t = idx >> 6, j = idx & 63;
        const int c = info->col[t], p = info->row[t], inv = info->inv[t];
        if (j > c && j < w) s_T[c][j] = zp_axpy_small(F, inv, (int)P[(i64d)j * Rp + p], 0);
        if (j == 0) s_has[c] = 1;
    }
    for (int j = 0; j < DP_W; j++)
        for (int r = tid; r < nloc; r += NT) X[j * chunk + r] = P[(i64d)j * Rp + base + r];
    unsigned long long live = 0;
    const int nmine = tid < nloc ? (nloc - tid + NT - 1) / NT : 0;
    for (int k = 0; k < nmine; k++)
        if (seq[base + tid + NT * k] < 0) live |= 1ull << k;
    __syncthreads();
    bool bad = false;
    for (int c = 0; c < w; c++) {
        const bool has = s_has[c] != 0;
        unsigned long long m = live;
        while (m) {
            const int k = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int r = tid + NT * k;
            const int f = (int)X[c * chunk + r];
            if (f == 0) continue;
            if (!has) { bad = true; continue; } // the pivot of this column would have to be a follower
            const int nf = -f;
            for (int j = c + 1; j < w; j++) {
                XT *x = &X[j * chunk + r];
                *x = (XT)zp_axpy_small(F, nf, s_T[c][j], (int)*x);
            }
        }
    }
    if (bad) *flag = 1;
    __syncthreads();
    for (int j = 0; j < DP_W; j++)
        for (int r = tid; r < nloc; r += NT) P[(i64d)j * Rp + base + r] = X[j * chunk + r];
}

// what k_panel_lu recorded for a panel, taken back (the host redoes the panel): its pivot rows are ordinary rows again
__global__ void k_panel_undo(int c0, PanelInfo *__restrict__ info, int *__restrict__ seq, int *__restrict__ pivrow_of_col, DenseState *st)
{
    const int t = threadIdx.x;
    const int npp = info->npp;
    if (t < npp) {
        seq[info->row[t]] = -1;
        pivrow_of_col[c0 + info->col[t]] = -1;
    }
    if (t == 0) { st->npiv = info->gbase; st->npp = 0; }
}

// signed base-256 digits of the residue v of a prime < 2^16: a representative of v's class in [-32896, 32639]
__device__ __forceinline__ void zp_digits(const ZpField &F, int v, int &d0, int &d1)
{
    if (v > 32639) v -= (int)F.p;
    d0 = ((v + 128) & 255) - 128;
    d1 = (v - d0) >> 8;
}

// ---- after k_panel_lu: the panel's columns of D (non-pivot rows: zero; the panel's pivot rows: normalised), the multipliers as
// digit planes F[d][i][slot0 + s] (0 where row i was a pivot already when pivot s was elected), and mtri.
template <int ND, typename DT>
__global__ __launch_bounds__(256) void k_panel_store(int R, int Rp, int c0, int w, ZpField F, const DT *__restrict__ P, const int *__restrict__ seq,
                                                    DT *__restrict__ D, i64d ldc, PanelInfo *__restrict__ info, signed char *__restrict__ Fd, i64d fplane,
                                                    int KB, int slot0)
{
    __shared__ int tile[DP_W][DP_W + 1];
    __shared__ int s_col[DP_W], s_inv[DP_W], s_row[DP_W];
    const int i0 = blockIdx.x * 64;
    const int npp = info->npp, gbase = info->gbase;
    if (threadIdx.x < DP_W) { s_col[threadIdx.x] = info->col[threadIdx.x]; s_inv[threadIdx.x] = info->inv[threadIdx.x]; s_row[threadIdx.x] = info->row[threadIdx.x]; }
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int j = idx >> 6, r = idx & 63;
        tile[j][r] = i0 + r < Rp ? (int)P[(i64d)j * Rp + i0 + r] : 0;
    }
    __syncthreads();
    // multipliers: thread r (< 64) of each group of 64 writes the 64 slots of its row, 16 at a time
    for (int idx = threadIdx.x; idx < 64 * 4; idx += 256) {
        const int r = idx >> 2, part = idx & 3;
        const int i = i0 + r;
        if (i >= Rp) continue;
        const int sq = i < R ? seq[i] : 0; // padding rows count as pivots of old: all zero
        int w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0};
        for (int b = 0; b < 16; b++) {
            const int s = part * 16 + b;
            int v = 0;
            if (s < npp && (sq < 0 || sq > gbase + s)) v = tile[s_col[s]][r];
            int d0, d1;
            zp_digits(F, v, d0, d1);
            w0[b >> 2] |= (d0 & 255) << (8 * (b & 3));
            w1[b >> 2] |= (d1 & 255) << (8 * (b & 3));
        }
        *(int4 *)(Fd + (i64d)i * KB + slot0 + part * 16) = make_int4(w0[0], w0[1], w0[2], w0[3]);
        if (ND == 2) *(int4 *)(Fd + fplane + (i64d)i * KB + slot0 + part * 16) = make_int4(w1[0], w1[1], w1[2], w1[3]);
    }
    // mtri (one workgroup does it: its rows are anywhere, so it reads P, not the tile)
    if (blockIdx.x == 0)
        for (int idx = threadIdx.x; idx < DP_W * DP_W; idx += 256) {
            const int t = idx >> 6, s = idx & 63;
            info->mtri[idx] = (s < t && t < npp) ? (int)P[(i64d)s_col[s] * Rp + s_row[t]] : 0;
        }
    __syncthreads();
    // D: rows that are no pivots get zeros; the panel's own pivot rows their normalised entries; older pivot rows stay as they are
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int r = idx >> 6, j = idx & 63;
        const int i = i0 + r;
        if (i >= R || j >= w) continue;
        const int sq = seq[i];
        if (sq >= 0 && sq < gbase) continue;
        int v = 0;
        if (sq >= gbase) {
            const int t = sq - gbase, ct = s_col[t];
            if (j == ct) v = 1;
            else if (j > ct) v = zp_mul(F, s_inv[t], tile[j][r]);
        }
        D[(i64d)i * ldc + c0 + j] = (DT)v;
    }
}

// ---- the pivot rows of one panel on the columns [ja, jb): u_t = inv_t (D[p_t][j] - sum_{s<t} mtri[t][s] u_s); D[p_t][j] = u_t and
// the digit planes Ut[d][j][slot0 + t].  One thread per column.
template <int ND, typename DT>
__global__ __launch_bounds__(64) void k_trsm_i8(int ja, int jb, ZpField F, DT *__restrict__ D, i64d ldc, const PanelInfo *__restrict__ info,
                                                signed char *__restrict__ Ut, i64d uplane, int KB, int slot0)
{
    __shared__ int s_m[DP_W * DP_W];
    __shared__ int s_row[DP_W], s_inv[DP_W];
    const int npp = info->npp;
    for (int e = threadIdx.x; e < DP_W * DP_W; e += 64) s_m[e] = info->mtri[e];
    s_row[threadIdx.x] = info->row[threadIdx.x];
    s_inv[threadIdx.x] = info->inv[threadIdx.x];
    __syncthreads();
    const int j = ja + blockIdx.x * 64 + threadIdx.x;
    if (j >= jb) return;
    int u[DP_W];
    int w0[16], w1[16];
#pragma unroll
    for (int b = 0; b < 16; b++) { w0[b] = 0; w1[b] = 0; }
#pragma unroll
    for (int t = 0; t < DP_W; t++) {
        u[t] = 0;
        if (t < npp) { // uniform
            long long acc = (long long)D[(i64d)s_row[t] * ldc + j];
#pragma unroll
            for (int s = 0; s < t; s++) acc -= (long long)s_m[t * DP_W + s] * (long long)u[s]; // |term| < 2^30, 64 terms
            const int v = zp_mul(F, s_inv[t], zp_reduce(F, acc));
            u[t] = v;
            D[(i64d)s_row[t] * ldc + j] = (DT)v;
            int d0, d1;
            zp_digits(F, v, d0, d1);
            w0[t >> 2] |= (d0 & 255) << (8 * (t & 3));
            w1[t >> 2] |= (d1 & 255) << (8 * (t & 3));
        }
    }
    int4 *o0 = (int4 *)(Ut + (i64d)j * KB + slot0);
#pragma unroll
    for (int q = 0; q < 4; q++) o0[q] = make_int4(w0[4 * q], w0[4 * q + 1], w0[4 * q + 2], w0[4 * q + 3]);
    if (ND == 2) {
        int4 *o1 = (int4 *)(Ut + uplane + (i64d)j * KB + slot0);
#pragma unroll
        for (int q = 0; q < 4; q++) o1[q] = make_int4(w1[4 * q], w1[4 * q + 1], w1[4 * q + 2], w1[4 * q + 3]);
    }
}

// ---- D[i][j] -= sum_{k < K} F[i][k0 + k] * Ut[j][k0 + k]  (mod p), j in [ja, jb).
// rows == NULL: all rows i < R that are no pivots (seq[i] < 0); otherwise the rows rows[0 .. nrows) (entries < 0 skipped).
// Four waves as WM x WN, each with TM x TN tiles of 32 x 32 (v_mfma_i32_32x32x32_i8): the workgroup covers BM = 32 WM TM rows
// and BN = 32 WN TN columns.  One digit: 2 x 2 waves of 2 x 2 tiles, 128 x 128 (a fragment read from LDS feeds two MFMAs);
// two digits (three accumulators per tile): 4 x 1 waves of 1 x 2 tiles, 128 x 64.  K in stages of 64 bytes staged through LDS
// (rows of 64 bytes padded to 80: conflict-free 16-byte reads).  Lane l holds A[row l & 31][k = 16 (l >> 5) + 0..15] and
// B[same k][column l & 31] (both operands take the same k, so any permutation of k inside the instruction cancels);
// C/D: column l & 31, row (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
#define GI_LDS_STRIDE 80
#ifndef GI_BAND
#define GI_BAND 32
#endif
// stages of K in flight in registers (one digit / two digits).  Four (168 VGPRs, no scratch) changed nothing: 0.420 s against 0.413 s
// for the GEMMs of config 5 at 1/5 -- the kernel does not wait for the latency of its loads, it is bound by what LDS and L1 carry
#ifndef GI_NPF
#define GI_NPF 1
#endif
#ifndef GI_NPF2
#define GI_NPF2 1
#endif
template <int ND, int APT, int BPT>
__device__ __forceinline__ void gemm_i8_fetch(v4i32 (&ra)[ND][APT], v4i32 (&rb)[ND][BPT], const signed char *__restrict__ Fd, i64d fplane,
                                              const signed char *__restrict__ Ut, i64d uplane, const i64d (&aoff)[APT], i64d boff, int KB, int ks)
{
#pragma unroll
    for (int d = 0; d < ND; d++) {
#pragma unroll
        for (int u = 0; u < APT; u++) ra[d][u] = *(const v4i32 *)(Fd + (i64d)d * fplane + aoff[u] + ks);
#pragma unroll
        for (int u = 0; u < BPT; u++) rb[d][u] = *(const v4i32 *)(Ut + (i64d)d * uplane + boff + (i64d)(64 * u) * KB + ks);
    }
}
template <int ND, int WM, int WN, int TM, int TN, typename DT, int MINB = 2, int NPFX = 0>
__global__ __launch_bounds__(256, MINB) void k_gemm_i8(int R, int ja, int jb, int k0, int K, ZpField F, DT *__restrict__ D, i64d ldc, const int *__restrict__ seq,
                                                    const int *__restrict__ rows, int nrows, const signed char *__restrict__ Fd, i64d fplane,
                                                    const signed char *__restrict__ Ut, i64d uplane, int KB, int ntm, int ntn)
{
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int APT = BM * 4 / 256, BPT = BN * 4 / 256; // 16-byte pieces per thread and stage
    static_assert(APT >= 1 && BPT >= 1 && BM * 4 % 256 == 0 && BN * 4 % 256 == 0, "tile vs staging");
    __shared__ __attribute__((aligned(16))) signed char s_a[ND][BM * GI_LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) signed char s_b[ND][BN * GI_LDS_STRIDE];
    __shared__ int s_gi[BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // Tile order (1-D grid of ntm x ntn tiles): bands of GI_BAND row tiles; inside a band the column tiles one after the other, each
    // with the band's row tiles.  The workgroups in flight then share the band's rows of F (4096 x K bytes: L2 / MALL) and a few
    // column tiles of Ut; row tiles fastest over the whole grid re-read all of F from HBM once per column tile (85 GB per update
    // of the 333k x 32k tail of config 5 at 1/10).
    int tm, tn;
    {
        const int per_band = GI_BAND * ntn;
        const int band = blockIdx.x / per_band, within = blockIdx.x % per_band;
        const int rows_in_band = min(GI_BAND, ntm - band * GI_BAND);
        tn = within / rows_in_band;
        tm = band * GI_BAND + within % rows_in_band;
        if (tn >= ntn) return; // (cannot happen for full bands; the last band is launched with its own count)
    }
    const int m0 = tm * BM, j0 = ja + tn * BN;
    if (tid < BM) {
        const int mi = m0 + tid;
        int gi = -1;
        if (rows) { if (mi < nrows) gi = rows[mi]; }
        else if (mi < R && seq[mi] < 0) gi = mi;
        s_gi[tid] = gi;
    }
    __syncthreads();
    constexpr int NACC = ND == 1 ? 1 : 3, A1 = ND == 1 ? 0 : 1, A2 = ND == 1 ? 0 : 2, D1 = ND - 1;
    v16i32 acc[NACC][TM][TN];
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
        for (int m = 0; m < TM; m++)
#pragma unroll
            for (int n = 0; n < TN; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[a][m][n][r] = 0;
    // staging: piece x (0 .. BM * 4) = row x / 4, 16-byte segment x % 4; thread tid takes pieces tid, tid + 256, ...
    const int seg = tid & 3, prow = tid >> 2;
    int gia[APT];
#pragma unroll
    for (int u = 0; u < APT; u++) gia[u] = s_gi[prow + 64 * u];
    // software pipeline: the global loads of stage s + 1 are in flight while stage s is multiplied out of LDS.  (Check the
    // ISA after touching this: when the staging registers end up in scratch memory every "prefetch" is waited for and spilled at once.)
    // Rows that are skipped load row 0 instead of branching around the load; their products are never stored.
    // GI_NPF stages of 64 bytes of K are in flight in registers (see GI_NPF: deeper than one bought nothing)
    constexpr int NPF = NPFX ? NPFX : (ND == 1 ? GI_NPF : GI_NPF2);
    v4i32 ra[NPF][ND][APT], rb[NPF][ND][BPT]; // (native vectors: arrays of HIP's int4 struct stayed in scratch memory)
    i64d aoff[APT];
#pragma unroll
    for (int u = 0; u < APT; u++) aoff[u] = (i64d)(gia[u] >= 0 ? gia[u] : 0) * KB + k0 + seg * 16;
    const i64d boff = (i64d)(j0 + prow) * KB + k0 + seg * 16;
#pragma unroll
    for (int p = 0; p < NPF; p++)
        if (64 * p < K) gemm_i8_fetch<ND, APT, BPT>(ra[p], rb[p], Fd, fplane, Ut, uplane, aoff, boff, KB, 64 * p);
    for (int ks0 = 0; ks0 < K; ks0 += 64 * NPF) {
#pragma unroll
      for (int p = 0; p < NPF; p++) {
        const int ks = ks0 + 64 * p;
        if (ks >= K) break; // (uniform)
        __syncthreads(); // the previous stage has been consumed
#pragma unroll
        for (int d = 0; d < ND; d++) {
#pragma unroll
            for (int u = 0; u < APT; u++) *(v4i32 *)(&s_a[d][(prow + 64 * u) * GI_LDS_STRIDE + seg * 16]) = ra[p][d][u];
#pragma unroll
            for (int u = 0; u < BPT; u++) *(v4i32 *)(&s_b[d][(prow + 64 * u) * GI_LDS_STRIDE + seg * 16]) = rb[p][d][u];
        }
        __syncthreads();
        if (ks + 64 * NPF < K) gemm_i8_fetch<ND, APT, BPT>(ra[p], rb[p], Fd, fplane, Ut, uplane, aoff, boff, KB, ks + 64 * NPF);
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int ko = kk * 32 + 16 * (lane >> 5);
            v4i32 fa[ND][TM], fb[ND][TN];
#pragma unroll
            for (int d = 0; d < ND; d++) {
#pragma unroll
                for (int m = 0; m < TM; m++) fa[d][m] = *(const v4i32 *)(&s_a[d][((wm * TM + m) * 32 + (lane & 31)) * GI_LDS_STRIDE + ko]);
#pragma unroll
                for (int n = 0; n < TN; n++) fb[d][n] = *(const v4i32 *)(&s_b[d][((wn * TN + n) * 32 + (lane & 31)) * GI_LDS_STRIDE + ko]);
            }
#pragma unroll
            for (int m = 0; m < TM; m++)
#pragma unroll
                for (int n = 0; n < TN; n++) {
                    acc[0][m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[0][m], fb[0][n], acc[0][m][n], 0, 0, 0);
                    if (ND == 2) {
                        acc[A1][m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[0][m], fb[D1][n], acc[A1][m][n], 0, 0, 0);
                        acc[A1][m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[D1][m], fb[0][n], acc[A1][m][n], 0, 0, 0);
                        acc[A2][m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[D1][m], fb[D1][n], acc[A2][m][n], 0, 0, 0);
                    }
                }
        }
      }
    }
    // epilogue.  A lane holds 16 ROWS of one column of every 32 x 32 tile (the MFMA's C layout): written to D as it is that is 16
    // narrow loads and 16 narrow stores per tile and lane -- as much time in the address path as the tile's MFMAs take (the LDS-DMA
    // kernel below gained 15 % from this change alone).  Each tile goes through LDS instead (the staging buffers are free now) and
    // comes back row-wise: a lane then owns 16 consecutive elements of one row of D -- 16-byte loads and stores.
    static_assert(sizeof(DT) <= 2, "D is kept as bytes or shorts");
    __syncthreads();
    // 32 rows of 36 ints per wave: waves 0, 1 in s_a, waves 2, 3 in s_b (each at least 10 KB)
    static_assert(sizeof(s_a) >= 2 * 32 * 36 * 4 && sizeof(s_b) >= 2 * 32 * 36 * 4, "epilogue scratch");
    int *S = (int *)(wave < 2 ? &s_a[0][0] : &s_b[0][0]) + (wave & 1) * (32 * 36);
    const int erow = lane >> 1, ecol = (lane & 1) * 16;
#pragma unroll
    for (int m = 0; m < TM; m++)
#pragma unroll
        for (int n = 0; n < TN; n++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int w;
                if (ND == 1) w = acc[0][m][n][r];
                else w = zp_reduce(F, (long long)acc[0][m][n][r] + (long long)acc[A1][m][n][r] * 256 + (long long)acc[A2][m][n][r] * 65536);
                S[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            int av[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const v4i32 t = *(const v4i32 *)(S + erow * 36 + ecol + 4 * q);
                av[4 * q] = t[0]; av[4 * q + 1] = t[1]; av[4 * q + 2] = t[2]; av[4 * q + 3] = t[3];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int gi = s_gi[(wm * TM + m) * 32 + erow];
            const int col0 = j0 + (wn * TN + n) * 32 + ecol;
            if (gi >= 0 && col0 < jb) {
                DT *dp = D + (i64d)gi * ldc + col0;
                constexpr int NV = (int)sizeof(DT); // 16-byte pieces of the 16 elements
                v4i32 in[NV], out[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) in[v] = ((const v4i32 *)dp)[v];
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    int dvv;
                    if (NV == 1) dvv = (int)(signed char)((in[0][e >> 2] >> (8 * (e & 3))) & 255);
                    else dvv = (int)(short)((in[e >> 3][(e >> 1) & 3] >> (16 * (e & 1))) & 65535);
                    int x;
                    if (ND == 1) {
                        // |acc| <= K * 127^2 < 2^25 for K <= 2048: 32-bit lazy reduction (|x / p| < 2^22) and one correction
                        x = zp_small_lazy(dvv - av[e], -F.finvp, (int)F.p);
                        if (x > (int)F.halfp) x -= (int)F.p;
                        else if (x < (int)F.mhalfp) x += (int)F.p;
                    } else {
                        x = dvv - av[e]; // both balanced residues
                        if (x > (int)F.halfp) x -= (int)F.p;
                        else if (x < (int)F.mhalfp) x += (int)F.p;
                    }
                    if (col0 + e >= jb) x = dvv;
                    if (NV == 1) {
                        if ((e & 3) == 0) out[0][e >> 2] = 0;
                        out[0][e >> 2] |= (x & 255) << (8 * (e & 3));
                    } else {
                        if ((e & 1) == 0) out[e >> 3][(e >> 1) & 3] = 0;
                        out[e >> 3][(e >> 1) & 3] |= (x & 65535) << (16 * (e & 1));
                    }
                }
#pragma unroll
                for (int v = 0; v < NV; v++) ((v4i32 *)dp)[v] = out[v];
            }
        }
}


// ================================================================================================
// Dense finish over ROW SHARDS (engine.hip: dense_finish_multi).  Rows never move: every shard keeps its rows of D, eliminates
// them against the panel's pivots, updates them with its own GEMMs.  What crosses shards per panel of 64 columns:
//   candidates  every shard eliminates its own rows' panel (k_panel_lu on a scratch copy of the bookkeeping) and names the <= 64
//               rows its elimination elected: they span the panel part of all its rows.  Their 64 panel entries each, as they
//               are in D, go to the root shard (CandRec, 16.7 KB per shard).
//   election    the root eliminates the stacked candidates (64 x shards rows: one workgroup of k_panel_lu) -- the pivots of the
//               panel are the rows this elects, the leading columns of the span of ALL rows: the pivot COLUMNS are those of the
//               single-device finish.  PanelGlob (winners and the normalised 64 x 64 panel part T of the pivot rows) goes to all.
//   guests      every winner's row of D (columns from the panel on) and of F (its multipliers of the block so far) is copied into
//               the same GUEST row of every shard (rows R .. R + KB of D: 64 per panel of the block), so that every shard runs the
//               pivot rows' triangular solves and its own GEMMs without further traffic: (C - c0) bytes per pivot and shard.
//   apply       with T known a shard's rows need no election and no barrier (k_panel_apply, the follower of k_panel_follow);
//               the winners -- guest copies, and the original on its owner -- freeze at their columns.
// At the end of a block the owners copy the finished guest rows over the originals.
// ================================================================================================
#define DM_MAXSHARDS 64

struct CandRec {
    int n;                 // candidates (<= 64)
    int pad[3];
    int row[DP_W];         // local rows, -1 beyond n
    int val[DP_W][DP_W];   // val[t][j] = D[row[t]][c0 + j] before the panel (0 beyond the panel's width)
};

struct PanelGlob {
    int npp, pad[3];
    int col[DP_W], inv[DP_W];     // pivot t: its column inside the panel, the inverse of its entry there
    int shard[DP_W], row[DP_W];   // its owner and the row there
    int pos[DP_W];                // its guest slot among the 64 of the panel (owner-major)
    int piv_of_col[DP_W];         // column -> pivot index, -1: no pivot
    int cnt[DM_MAXSHARDS], first[DM_MAXSHARDS]; // winners of every shard, and the guest slot of its first
    int T[DP_W][DP_W];            // T[c][j], j > c: the normalised pivot row of column c (0 where the column has none)
};

template <typename DT>
__global__ __launch_bounds__(256) void k_cand_gather(int c0, int w, const DT *__restrict__ D, i64d ldc, const PanelInfo *__restrict__ info, CandRec *__restrict__ out,
                                                     const DenseState *__restrict__ st_tmp, int *__restrict__ flag)
{
    const int n = info->npp;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out->n = n;
        if (st_tmp->pad) atomicOr(flag, 2); // a grid barrier of the candidates' elimination timed out
    }
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < DP_W * DP_W; idx += gridDim.x * 256) {
        const int t = idx >> 6, j = idx & 63;
        const int r = t < n ? info->row[t] : -1;
        out->val[t][j] = (r >= 0 && j < w) ? (int)D[(i64d)r * ldc + c0 + j] : 0;
        if (j == 0) out->row[t] = r;
    }
}

// the stacked candidates of G shards as the panel of a one-workgroup k_panel_lu: P[j][64 k + t], seq = -1 for real candidates
template <typename DT>
__global__ __launch_bounds__(256) void k_stack_load(int G, int Rs, const CandRec *__restrict__ stack, DT *__restrict__ P, int *__restrict__ seq, PanelSync *sy,
                                                    DenseState *st)
{
    if (blockIdx.x == 0) {
        if (threadIdx.x < DP_NCTR) sy->arrive[threadIdx.x * 32] = 0;
        if (threadIdx.x == 0) { sy->timeout = 0; st->npiv = 0; st->npp = 0; st->pad = 0; }
    }
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < Rs * DP_W; idx += gridDim.x * 256) {
        const int j = idx / Rs, r = idx % Rs;
        const int k = r >> 6, t = r & 63;
        const bool real = k < G && t < stack[k].n;
        P[(i64d)j * Rs + r] = real ? (DT)stack[k].val[t][j] : (DT)0;
        if (j == 0) seq[r] = real ? -1 : 0;
    }
}

// what the election among the stacked candidates found, as every shard needs it
template <typename DT>
__global__ __launch_bounds__(1024) void k_make_glob(int G, int Rs, int w, ZpField F, const CandRec *__restrict__ stack, const PanelInfo *__restrict__ info,
                                                    const DT *__restrict__ P, PanelGlob *__restrict__ g)
{
    const int tid = threadIdx.x;
    const int npp = info->npp;
    __shared__ int s_sh[DP_W];
    for (int idx = tid; idx < DP_W * DP_W; idx += 1024) g->T[idx >> 6][idx & 63] = 0;
    if (tid < DP_W) {
        g->piv_of_col[tid] = -1;
        const int t = tid;
        int c = -1, inv = 0, sh = -1, row = -1;
        if (t < npp) {
            c = info->col[t];
            inv = info->inv[t];
            const int sr = info->row[t];
            sh = sr >> 6;
            row = stack[sh].row[sr & 63];
        }
        g->col[t] = c; g->inv[t] = inv; g->shard[t] = sh; g->row[t] = row;
        s_sh[t] = sh;
    }
    __syncthreads();
    if (tid < DP_W && tid < npp) g->piv_of_col[g->col[tid]] = tid;
    if (tid == 0) {
        g->npp = npp;
        int at = 0;
        for (int k = 0; k < DM_MAXSHARDS; k++) {
            g->first[k] = at;
            int c = 0;
            if (k < G)
                for (int t = 0; t < npp; t++)
                    if (s_sh[t] == k) g->pos[t] = at + c++;
            g->cnt[k] = c;
            at += c;
        }
        for (int t = npp; t < DP_W; t++) g->pos[t] = -1;
    }
    __syncthreads();
    for (int idx = tid; idx < npp * DP_W; idx += 1024) {
        const int t = idx >> 6, j = idx & 63;
        const int c = info->col[t], sr = info->row[t], inv = info->inv[t];
        if (j > c && j < w) g->T[c][j] = zp_axpy_small(F, inv, (int)P[(i64d)j * Rs + sr], 0);
    }
}

// the rows a shard won, packed in winner order: their entries of D from column c0 on, their rows of F
template <typename DT>
__global__ __launch_bounds__(256) void k_export_pack(int me, int c0, i64d ldc, const PanelGlob *__restrict__ g, const DT *__restrict__ D,
                                                     const signed char *__restrict__ Fd, i64d fplane, int KB, int ND, DT *__restrict__ expD,
                                                     signed char *__restrict__ expF)
{
    const int t = blockIdx.x;
    if (t >= g->npp || g->shard[t] != me) return;
    const int slot = g->pos[t] - g->first[me];
    const int cnt = g->cnt[me];
    const i64d src = (i64d)g->row[t] * ldc, dst = (i64d)slot * ldc;
    for (i64d j = c0 + blockIdx.y * 256 + threadIdx.x; j < ldc; j += (i64d)gridDim.y * 256) expD[dst + j] = D[src + j];
    if (blockIdx.y == 0)
        for (int d = 0; d < ND; d++)
            for (int k = threadIdx.x; k < KB; k += 256) expF[((i64d)d * cnt + slot) * KB + k] = Fd[(i64d)d * fplane + (i64d)g->row[t] * KB + k];
}

// the guest rows of the panel go live; own_map[guest] = the row they came from when it is this shard's
__global__ void k_apply_prep(int me, int guest0, int q, const PanelGlob *__restrict__ g, int *__restrict__ seq, int *__restrict__ own_map)
{
    const int t = threadIdx.x;
    if (t < g->npp) {
        const int gi = q * DP_W + g->pos[t];
        seq[guest0 + gi] = -1;
        own_map[gi] = g->shard[t] == me ? g->row[t] : -1;
    }
}

// the elimination of a shard's rows by the panel's pivots (T of the PanelGlob): no election, no barrier.  The winners -- the guest
// rows of the panel, and on its owner the row a guest was copied from -- take part until their own column and are pivots from there.
// *flag is set when a live row holds a non-zero in a column without a pivot (the candidates did not span the shard's rows: a bug).
template <int NT, typename XT>
__global__ __launch_bounds__(NT) void k_panel_apply(int Rp, int chunk, int w, ZpField F, XT *__restrict__ P, int *__restrict__ seq, const PanelGlob *__restrict__ g,
                                                    int me, int guest0, const DenseState *__restrict__ st, int *__restrict__ flag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyna[]; // chunk * 64 residues
    XT *X = (XT *)s_dyna;
    __shared__ int s_T[DP_W][DP_W];
    __shared__ int s_t[DP_W], s_g[DP_W], s_o[DP_W];
    const int tid = threadIdx.x;
    const int base = blockIdx.x * chunk;
    const int nloc = min(chunk, Rp - base);
    if (nloc <= 0) return;
    for (int idx = tid; idx < DP_W * DP_W; idx += NT) s_T[idx >> 6][idx & 63] = g->T[idx >> 6][idx & 63];
    if (tid < DP_W) {
        const int t = g->piv_of_col[tid];
        s_t[tid] = t;
        s_g[tid] = t >= 0 ? guest0 + g->pos[t] : -1;
        s_o[tid] = (t >= 0 && g->shard[t] == me) ? g->row[t] : -1;
    }
    for (int j = 0; j < DP_W; j++)
        for (int r = tid; r < nloc; r += NT) X[j * chunk + r] = P[(i64d)j * Rp + base + r];
    unsigned long long live = 0;
    const int nmine = tid < nloc ? (nloc - tid + NT - 1) / NT : 0;
    for (int k = 0; k < nmine; k++)
        if (seq[base + tid + NT * k] < 0) live |= 1ull << k;
    const int gbase = st->npiv;
    __syncthreads();
    bool bad = false;
    for (int c = 0; c < w; c++) {
        const int t = s_t[c];
        const int rg = s_g[c] - base, ro = s_o[c] - base;
        unsigned long long m = live;
        while (m) {
            const int k = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int r = tid + NT * k;
            if (t >= 0 && (r == rg || r == ro)) { seq[base + r] = gbase + t; live &= ~(1ull << k); continue; }
            const int f = (int)X[c * chunk + r];
            if (f == 0) continue;
            if (t < 0) { bad = true; continue; }
            const int nf = -f;
            for (int j = c + 1; j < w; j++) {
                XT *x = &X[j * chunk + r];
                *x = (XT)zp_axpy_small(F, nf, s_T[c][j], (int)*x);
            }
        }
    }
    if (bad) atomicOr(flag, 1);
    __syncthreads();
    for (int j = 0; j < DP_W; j++)
        for (int r = tid; r < nloc; r += NT) P[(i64d)j * Rp + base + r] = X[j * chunk + r];
}

// the panel's record as k_panel_store / k_trsm_i8 / the pivot-row GEMMs read it: the pivot rows are the GUEST rows
__global__ void k_apply_info(int me, int guest0, int c0, const PanelGlob *__restrict__ g, PanelInfo *__restrict__ info, DenseState *st, int *__restrict__ own_pivrow_of_col)
{
    const int t = threadIdx.x;
    const int npp = g->npp;
    if (t < DP_W) {
        info->row[t] = t < npp ? guest0 + g->pos[t] : -1;
        info->col[t] = t < npp ? g->col[t] : -1;
        info->inv[t] = t < npp ? g->inv[t] : 0;
        if (t < npp && g->shard[t] == me) own_pivrow_of_col[c0 + g->col[t]] = g->row[t];
    }
    if (t == 0) {
        info->npp = npp;
        info->gbase = st->npiv;
        st->npp = npp;
        st->npiv += npp;
    }
}

// end of a block: the finished guest rows over the rows they came from (columns from the block's first on)
template <typename DT>
__global__ __launch_bounds__(256) void k_guest_copyback(int guest0, int b0, i64d ldc, const int *__restrict__ own_map, DT *__restrict__ D)
{
    const int o = own_map[blockIdx.x];
    if (o < 0) return;
    const i64d src = (i64d)(guest0 + blockIdx.x) * ldc, dst = (i64d)o * ldc;
    for (i64d j = b0 + blockIdx.y * 256 + threadIdx.x; j < ldc; j += (i64d)gridDim.y * 256) D[dst + j] = D[src + j];
}

__global__ void k_or_int(int n, int *__restrict__ a, const int *__restrict__ b)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] |= b[i];
}

// ================================================================================================
// Tall-and-skinny finish (engine.hip: dense_finish_tall; libspasm's enable_tall_and_skinny / tall_and_skinny_ratio, reference
// src/SpaSM.jl:327, :341).  A remainder with many more rows than columns has at most C pivots: eliminating ALL its rows costs
// R C^2 / 2, although the rank is settled by few of them.  Instead
//   1. a first slab of R1 = C + C/8 rows is eliminated as usual (R1 C^2 / 2): r1 pivots, echelon rows u_t;
//   2. the reduced form of those rows on the f = C - r1 columns WITHOUT pivot, Z_t = u_t,N - sum_{s > t} u_t[pcol_s] Z_s (blocked back
//      substitution on the int8 GEMM: r1^2 f / 2);
//   3. every other row d is reduced in one step, t = d_N - d_P Z (R2 r1 f): zero on all pivot columns by construction;
//   4. the residuals (R2 x f, small) are eliminated like any dense matrix: the pivots they still hold.
// Exact, not probabilistic (libspasm's low-rank mode draws random combinations and stops when a batch finds nothing): every row is
// reduced, and the pivot columns are the leading columns of the row space whatever the slab was (a basis with distinct leading
// columns -- the u_t and the echelon form of the residuals -- shows them all).
// ================================================================================================

// F[d][i][k] = digit d of D[rowidx ? rowidx[i] : i][cols[k]], k < K; 0 for K <= k < Kpad
// (col_off: D holds a slab of the columns, starting at column col_off of the numbering `cols` uses)
template <int ND, typename DT>
__global__ __launch_bounds__(256) void k_tall_gather_F(int nrows, const int *__restrict__ rowidx, const DT *__restrict__ D, i64d ldc, const int *__restrict__ cols, int K,
                                                       int Kpad, ZpField F, signed char *__restrict__ Fd, i64d fplane, int KB, int col_off = 0)
{
    const int i = blockIdx.x;
    if (i >= nrows) return;
    const i64d src = (i64d)(rowidx ? rowidx[i] : i) * ldc - col_off;
    for (int k = threadIdx.x; k < Kpad; k += 256) {
        int v = k < K ? (int)D[src + cols[k]] : 0;
        int d0, d1;
        zp_digits(F, v, d0, d1);
        Fd[(i64d)i * KB + k] = (signed char)d0;
        if (ND == 2) Fd[fplane + (i64d)i * KB + k] = (signed char)d1;
    }
}

// Ut[d][j][k] = digit d of Z[k][j], k < K (rows of Z), j < ncols; 0 for K <= k < Kpad and for ncols <= j < ncols_pad
template <int ND, typename DT>
__global__ __launch_bounds__(256) void k_tall_Ut(const DT *__restrict__ Z, i64d ldz, int K, int Kpad, int ncols, int ncols_pad, ZpField F, signed char *__restrict__ Ut, i64d uplane,
                                                 int KB)
{
    const int j = blockIdx.x;
    if (j >= ncols_pad) return;
    for (int k = threadIdx.x; k < Kpad; k += 256) {
        int v = (k < K && j < ncols) ? (int)Z[(i64d)k * ldz + j] : 0;
        int d0, d1;
        zp_digits(F, v, d0, d1);
        Ut[(i64d)j * KB + k] = (signed char)d0;
        if (ND == 2) Ut[uplane + (i64d)j * KB + k] = (signed char)d1;
    }
}

// out[i][j] = D[rowidx ? rowidx[i] : i][cols[j]], j < ncols (0 up to ldo)
template <typename DT>
__global__ __launch_bounds__(256) void k_tall_gather_cols(int nrows, const int *__restrict__ rowidx, const DT *__restrict__ D, i64d ldc, const int *__restrict__ cols, int ncols,
                                                          DT *__restrict__ out, i64d ldo)
{
    const int i = blockIdx.x;
    if (i >= nrows) return;
    const i64d src = (i64d)(rowidx ? rowidx[i] : i) * ldc;
    for (int j = threadIdx.x; j < (int)ldo; j += 256) out[(i64d)i * ldo + j] = j < ncols ? D[src + cols[j]] : (DT)0;
}

// out[i][j0 + j] += D[i][cols[j] - col_off] (mod p), j < ncols: the columns cols[0 .. ncols) of a slab of D that starts at column
// col_off.  ADDED, not stored: with the source in several column slabs the GEMMs of the slabs before this one have already
// subtracted their share of d_P Z from these columns of the residual (storing dropped it: the residuals then kept parts of the
// row space they should have lost, and the rank came out too high -- found with a matrix of planted rank, tools/planted_rank.py).
template <typename DT>
__global__ __launch_bounds__(256) void k_tall_gather_slab(int nrows, const DT *__restrict__ D, i64d ldc, const int *__restrict__ cols, int ncols, int col_off, ZpField F,
                                                          DT *__restrict__ out, i64d ldo, int j0)
{
    const int i = blockIdx.x;
    if (i >= nrows) return;
    const i64d src = (i64d)i * ldc - col_off;
    for (int j = threadIdx.x; j < ncols; j += 256) {
        int v = (int)out[(i64d)i * ldo + j0 + j] + (int)D[src + cols[j]];
        if (v > (int)F.halfp) v -= (int)F.p;
        else if (v < (int)F.mhalfp) v += (int)F.p;
        out[(i64d)i * ldo + j0 + j] = (DT)v;
    }
}

// the pivot rows of an eliminated residual matrix T, packed in column order: E[slot] = T[pc[c]], slot = scan[c] (the number of pivot
// columns before c); their origins and the pivot row of every column in the packed numbering.  One workgroup per column.
template <typename DT>
__global__ __launch_bounds__(256) void k_tall_compact(int f, i64d ldz, const int *__restrict__ pc, const int *__restrict__ scan, const DT *__restrict__ T,
                                                      const int *__restrict__ origT, DT *__restrict__ E, int *__restrict__ origE, int *__restrict__ pcE)
{
    const int c = blockIdx.x;
    if (c >= f) return;
    const int r = pc[c];
    if (r < 0) { if (threadIdx.x == 0) pcE[c] = -1; return; }
    const int slot = scan[c];
    if (threadIdx.x == 0) { pcE[c] = slot; origE[slot] = origT[r]; }
    for (i64d j = threadIdx.x; j < ldz; j += 256) E[(i64d)slot * ldz + j] = T[(i64d)r * ldz + j];
}

// back substitution inside a block of nb <= 64 pivots (rows t0 .. t0 + nb of Z, pivot rows prow[t], pivot columns pcol[t]):
// z_t -= sum_{t < s < t0 + nb} D[prow[t]][pcol[s]] z_s, t descending.  One workgroup per 16 columns of Z: thread (tx, ty) owns column
// tx and the rows 4 ty .. 4 ty + 3; the 64 x 64 coefficients and the tile of Z live in LDS.  Right-looking: once z_t is final every
// row above it takes its term at once (one barrier per t; the sums stay in 64 bits: at most 63 terms below 2^30).
template <typename DT>
__global__ __launch_bounds__(256) void k_tall_backsub(int t0, int nb, int ncols, ZpField F, const DT *__restrict__ D, i64d ldc, const int *__restrict__ prow,
                                                      const int *__restrict__ pcol, DT *__restrict__ Z, i64d ldz)
{
    __shared__ int s_u[64][65];
    __shared__ int s_z[64][16];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int j = blockIdx.x * 16 + tx;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int a = idx >> 6, b = idx & 63;
        s_u[a][b] = (a < nb && b < nb && b > a) ? (int)D[(i64d)prow[t0 + a] * ldc + pcol[t0 + b]] : 0;
    }
    long long acc[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int a = 4 * ty + r;
        acc[r] = (a < nb && j < ncols) ? (long long)Z[(i64d)(t0 + a) * ldz + j] : 0;
    }
    __syncthreads();
    for (int t = nb - 1; t >= 0; t--) {
        if ((t >> 2) == ty) s_z[t][tx] = zp_reduce(F, acc[t & 3]); // (uniform per thread row group: the owner publishes z_t)
        __syncthreads();
        const int zt = s_z[t][tx];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int a = 4 * ty + r;
            if (a < t) acc[r] -= (long long)s_u[a][t] * (long long)zt;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int a = 4 * ty + r;
        if (a < nb && j < ncols) Z[(i64d)(t0 + a) * ldz + j] = (DT)s_z[a][tx];
    }
}

// lists of the columns with and without pivot, both ascending: flag = pivrow_of_col >= 0, scans of the two flags
__global__ void k_tall_split_cols(int C, const int *__restrict__ pivrow_of_col, const int *__restrict__ pscan, int *__restrict__ pcol, int *__restrict__ prow,
                                  int *__restrict__ fcol)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int p = pivrow_of_col[c];
    if (p >= 0) { pcol[pscan[c]] = c; prow[pscan[c]] = p; }
    else fcol[c - pscan[c]] = c;
}

__global__ void k_gather_int2(int n, const int *__restrict__ idx, const int *__restrict__ src, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

// ================================================================================================
// The large updates of the one-digit finish (p < 2^8) on a 256 x 256 tile: the structure the CDNA guide names for getting past the
// ceiling of the 128 x 128, two-barriers-per-stage kernel above -- one workgroup of 8 waves per CU, operands by LDS-DMA
// (global_load_lds, 16 bytes per lane: no staging registers), four LDS buffers, the loads of two K-tiles in flight ACROSS raw
// barriers (counted vmcnt, never 0 in the loop).  LDS image of a K-tile (64 bytes of K): A[row][64 bytes] and B[row][64 bytes] -- a wave
// instruction of the DMA fills 16 rows, four lanes per row (whole 64-byte runs of global memory) --, the four 16-byte pieces of a row
// permuted by (row >> 1) & 3 on the SOURCE side so that the fragment reads (32 rows, one piece) fall on all banks.
// Waves 2 (rows) x 4 (columns), 128 x 64 each: 4 x 2 tiles of 32 x 32.
// ================================================================================================
#define GL_BM 256
#define GL_BN 256
#define GL_NBUF 4
#define GL_TILE_BYTES 32768 // A 16 KB + B 16 KB per K-tile
#define GL_LDS_BYTES (GL_NBUF * GL_TILE_BYTES + GL_BM * 4)

__device__ __forceinline__ void gl_dma16(const signed char *g, unsigned char *l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

template <typename DT>
__global__ __launch_bounds__(512, 1) void k_gemm_i8_glds(int R, int ja, int jb, int k0, int K, ZpField F, DT *__restrict__ D, i64d ldc, const int *__restrict__ seq,
                                                         const signed char *__restrict__ Fd, const signed char *__restrict__ Ut, int KB, int ntm, int ntn)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_gl[]; // [GL_NBUF][A | B], then the 256 row numbers of the tile
    int *s_gi = (int *)(s_gl + GL_NBUF * GL_TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    int tm, tn;
    {
        const int per_band = GI_BAND * ntn;
        const int band = blockIdx.x / per_band, within = blockIdx.x % per_band;
        const int rows_in_band = min(GI_BAND, ntm - band * GI_BAND);
        tn = within / rows_in_band;
        tm = band * GI_BAND + within % rows_in_band;
        if (tn >= ntn) return;
    }
    const int m0 = tm * GL_BM, j0 = ja + tn * GL_BN;
    if (tid < GL_BM) {
        const int mi = m0 + tid;
        s_gi[tid] = (mi < R && seq[mi] < 0) ? mi : -1;
    }
    __syncthreads();
    // the two A and two B pieces this wave brings in per K-tile: piece id = 2 wave + i = 16 rows, all 64 bytes of the K slice (four
    // lanes per row: whole 64-byte runs of global memory); the 16-byte slot a lane fills holds seg = slot ^ ((row >> 1) & 3)
    const signed char *pa[2], *pb[2];
    int lo[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int id = 2 * wave + i;
        const int row = id * 16 + (lane >> 2), seg = (lane & 3) ^ ((row >> 1) & 3);
        const int gi = s_gi[row];
        pa[i] = Fd + (i64d)(gi >= 0 ? gi : 0) * KB + k0 + seg * 16;
        pb[i] = Ut + (i64d)(j0 + row) * KB + k0 + seg * 16;
        lo[i] = id * 1024;
    }
    auto issue = [&](int t, int b) {
        unsigned char *base = s_gl + b * GL_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            gl_dma16(pa[i] + (i64d)t * 64, base + lo[i]);
            gl_dma16(pb[i] + (i64d)t * 64, base + 16384 + lo[i]);
        }
    };
    v16i32 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0;
    const int nt = K >> 6;
    // (Reading the fragments of the next half K-tile while the MFMAs of this one run -- explicit software pipelining inside a wave --
    // changed nothing: 0.243 s against 0.234 s on config 5 at 1/5; the second wave of the SIMD already fills those gaps.)
    const int r31 = lane & 31;
    issue(0, 0);
    if (nt > 1) issue(1, 1);
    for (int t = 0; t < nt; t++) {
        const int b = t & 3;
        if (t + 2 < nt) {
            issue(t + 2, (t + 2) & 3);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (t + 1 < nt) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ONE barrier per K-tile: with four buffers the DMAs issued above, K-tile t + 2, go to the buffer of K-tile t - 2, which every
        // wave had finished reading before it arrived at the barrier of iteration t - 1
        __builtin_amdgcn_s_barrier();
        const unsigned char *A = s_gl + b * GL_TILE_BYTES, *B = A + 16384;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            // row 32 rb + (lane & 31), seg 2 kk + (lane >> 5): slot = seg ^ ((row >> 1) & 3) (32 rb is a multiple of 8: the lane decides)
            const int so = r31 * 64 + (((2 * kk + (lane >> 5)) ^ ((r31 >> 1) & 3)) << 4);
            v4i32 fa[4], fb[2];
#pragma unroll
            for (int m = 0; m < 4; m++) fa[m] = *(const v4i32 *)(A + so + (wm * 4 + m) * 2048);
#pragma unroll
            for (int n = 0; n < 2; n++) fb[n] = *(const v4i32 *)(B + so + (wn * 2 + n) * 2048);
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[m], fb[n], acc[m][n], 0, 0, 0);
        }
    }
    // epilogue.  A lane holds 16 ROWS of one column of every 32 x 32 tile (the MFMA's C layout): written to D as it is that is 32
    // byte-wide loads and stores per tile and lane, as much time in the address path as the tile's MFMAs take.  Each tile goes
    // through LDS instead (the K-tile buffers are free now) and comes back row-wise: a lane then owns 16 consecutive bytes of one
    // row of D -- one 16-byte load, one 16-byte store.
    static_assert(sizeof(DT) == 1, "the one-digit finish keeps D as bytes");
    __syncthreads();
    int *S = (int *)(s_gl + wave * (32 * 36 * 4)); // 32 rows of 36 ints (16-byte aligned rows, odd multiple of 4 banks)
    const int erow = lane >> 1, ecol = (lane & 1) * 16;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
#pragma unroll
            for (int r = 0; r < 16; r++) S[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[m][n][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            int av[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const v4i32 t = *(const v4i32 *)(S + erow * 36 + ecol + 4 * q);
                av[4 * q] = t[0]; av[4 * q + 1] = t[1]; av[4 * q + 2] = t[2]; av[4 * q + 3] = t[3];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int gi = s_gi[(wm * 4 + m) * 32 + erow];
            const int col0 = j0 + (wn * 2 + n) * 32 + ecol;
            if (gi >= 0 && col0 < jb) {
                signed char *dp = (signed char *)D + (i64d)gi * ldc + col0;
                const v4i32 dv4 = *(const v4i32 *)dp;
                v4i32 out;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    int word = 0;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int dvv = (int)(signed char)((dv4[q] >> (8 * e)) & 255);
                        int x = zp_small_lazy(dvv - av[4 * q + e], -F.finvp, (int)F.p);
                        if (x > (int)F.halfp) x -= (int)F.p;
                        else if (x < (int)F.mhalfp) x += (int)F.p;
                        if (col0 + 4 * q + e >= jb) x = dvv;
                        word |= (x & 255) << (8 * e);
                    }
                    out[q] = word;
                }
                *(v4i32 *)dp = out;
            }
        }
}
