// wlevel.hpp -- W = -(I + U_PP)^-1 * U_PN built level by level of the pivot graph (no Uinv, no host synchronisation).
//
// W is what lets a non-pivot row's Schur row be written as  x_a = B[k]_N + sum_c a_c * W[qinv(c)]  (stream.hpp; the solve
// x_b * U + x_a = B[k] of reference src/SpaSM.jl:694-713 re-associated).  Row q of W satisfies
//        W[q] = -U_PN[q] - sum over the entries (c, v) of U_PP[q] of v * W[c]
// and U_PP only refers to pivots that come later in the topological order, so with
//        level(q) = 0 when U_PP[q] is empty, 1 + max level(c) otherwise
// all rows of one level are independent: one launch per level, every row a sparse combination of finished rows.  (The reference
// reaches the same pivot rows by a depth-first search per row, spasm_reach, src/SpaSM.jl:627-628; the levels are the breadth-first
// image of that search, computed once per round for all rows.)
//
// A row is merged in an LDS hash table (column -> lazy accumulator) sized by its bound npn + sum len(W[c]), swept, compacted and
// stored behind the rows before it: a wave per row up to 512 entries (k_wlevel_wave), the workgroup for longer ones
// (k_wlevel_wg).  Output space comes from one cursor; a wave takes 1024 entries at a time, so the cursor sees one atomic per
// ~6 rows, and keeps what is left of its block for the next level.  A row that cannot be built (longer than the largest table,
// or the region is full) is published with length -1: the plan kernel sends the rows that would need it to the multiplier lists.
#pragma once
#include "kernels.hpp"

constexpr int WMAXLEV = 254;                    // deeper pivot graphs keep to the multiplier lists
constexpr int WL_TMAX = 1024;                   // table slots of a wave
constexpr int WL_WAVE_BOUND = WL_TMAX / 2;      // longest bound a wave takes (tables at most half full)
constexpr int WL_NCD = 64;                      // chunk descriptors of a wave
constexpr u64d WL_BLK = 4096;                   // entries a wave / workgroup takes from the cursor at a time (one returning atomic on
                                                // one word costs ~11 ns of that word's time: 170 000 rows must not queue there)
constexpr int WL_NSUB = 64;                     // the lists of rows left to the workgroup kernels are kept in 64 parts with a counter each
constexpr int WL_SUBSTRIDE = 32;                // (ints between two counters: a line of their own; thousands of appends to ONE word
                                                // serialise at ~11 ns each -- they were most of the wave kernel's time on the middle levels)
// wstate words
// (the cursor on a line of its own; the statistics are only kept by the first build of a U: tens of thousands of waves adding to
// one word at the end of a kernel serialise)
constexpr int WS_CURSOR = 0, WS_UNAVAIL = 16, WS_ERROR = 17, WS_ENTRIES = 18, WS_BIGROWS = 19, WS_WORDS = 32;

// one bit per column: is it a pivot column of this round?  (blockDim.x is a multiple of 64: whole waves, whole bitmap words)
__global__ void k_pbits(int m, const int *__restrict__ qinv_r, unsigned *__restrict__ pbits)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = j < m ? qinv_r[j] : -1;
    const u64d b = __ballot(q >= 0);
    const int lane = threadIdx.x & 63;
    if ((lane & 31) == 0) pbits[j >> 5] = (unsigned)(b >> lane);
}

// ------------------------------------------------------------------------------------------------
// levels of the pivot graph by relaxation: lev[q] = max(lev[q], 1 + max lev[c]).  Every launch sees at least the values the
// launch before it left, so `depth + 1` launches reach the fixed point; *changed (when given) tells whether this one moved
// anything.  The loads and stores race with those of other threads on purpose (monotone: a stale value only delays).
// ------------------------------------------------------------------------------------------------
__global__ void k_lev_relax(int npiv, const UHdr *__restrict__ uhdr, const int2 *__restrict__ UPP, int *lev, int *changed, int sweeps)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= npiv) return;
    const UHdr h = uhdr[q];
    if (h.npp == 0) return;
    int cur = __hip_atomic_load(&lev[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ch = false;
    for (int s = 0; s < sweeps; s++) {
        int l = 0;
        for (int k = 0; k < h.npp; k++) {
            const int c = UPP[(size_t)h.off + k].x;
            l = max(l, __hip_atomic_load(&lev[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1);
        }
        if (l > cur) {
            cur = l;
            ch = true;
            __hip_atomic_store(&lev[q], l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (ch && changed) *changed = 1;
}

// keys = the levels in ascending order: lev_start[L] = first position of level L (levels are contiguous: a row of level L
// depends on one of level L - 1); lev_start[] is prefilled with npiv
__global__ void k_lev_starts(int npiv, const int *__restrict__ keys, int *__restrict__ lev_start, int maxlev)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npiv) return;
    const int k = keys[i];
    const int kp = i ? keys[i - 1] : -1;
    for (int L = kp + 1; L <= k && L <= maxlev; L++) lev_start[L] = i;
}

__global__ void k_fill_int(int n, int v, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

// what a build starts from: cursor and statistics zero, every wave without a block, no long rows at any level
__global__ void k_wbuild_reset(u64d *__restrict__ wstate, u64d *__restrict__ wblk, int nblk_words, int *__restrict__ big_count, int nlev, u64d cursor0)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < WS_WORDS) wstate[t] = t == WS_CURSOR ? cursor0 : 0;
    if (t < nblk_words) wblk[t] = 0;
    if (t < nlev) big_count[t] = 0;
}

// statistics of a finished build, from the rows' {offset, length} records: entries of W, rows that could not be built
__global__ void k_wstats(int npiv, const int2 *__restrict__ wrow, u64d *__restrict__ wstate)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    long long len = q < npiv ? (long long)wrow[q].y : 0;
    const u64d un = __ballot(len < 0);
    len = len < 0 ? 0 : len;
    for (int o = 32; o > 0; o >>= 1) len += __shfl_xor(len, o);
    if ((threadIdx.x & 63) == 0) {
        if (len) atomicAdd(wstate + WS_ENTRIES, (u64d)len);
        if (un) atomicAdd(wstate + WS_UNAVAIL, (u64d)__popcll(un));
    }
}

// a pivot row as the level kernels want it, in level order: one 32-byte record instead of order[] -> uhdr[] -> pivcol[]
struct __attribute__((aligned(32))) WLevRec {
    int q;        // pivot index
    unsigned off; // of its entries in UPP / UPN
    int npp, npn;
    int pivcol;
    int pad[3];
};

__global__ void k_lev_recs(int npiv, const int *__restrict__ order, const UHdr *__restrict__ uhdr, const int *__restrict__ pivcol, WLevRec *__restrict__ recs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npiv) return;
    const int q = order[i];
    const UHdr h = uhdr[q];
    WLevRec r;
    r.q = q;
    r.off = h.off;
    r.npp = h.npp;
    r.npn = h.npn;
    r.pivcol = pivcol[q];
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    recs[i] = r;
}

struct WLevelArgs {
    int cnt;                   // rows of this level (wave kernel)
    const WLevRec *recs;       // wave kernel: the level's rows; workgroup kernel: ALL rows in level order (its lists index them)
    const int2 *UPP;           // {pivot index, value} of the entries on other pivot columns
    int2 *buf;                 // [U_PN | own entries of the plan | W]: rows of W are read and written here
    int2 *wrow;                // per pivot index: {offset in buf, length} of its row of W (length -1: not available)
    int4 *wcol;                // per pivot COLUMN: {pivot index, length, offset, 0} -- what the plan kernel reads
    u64d *wstate;              // WS_* words
    unsigned wbase;            // where W starts in buf
    u64d wcap;                 // entries of W's region
    u64d *wblk;                // per wave slot of the wave kernel: {next free entry, end} of its block (relative to wbase)
    int rec_base;              // index of the level's first row among all rows
    // lists of rows (indices into all rows) left to a workgroup kernel: WL_NSUB parts of list_stride ints, part s with its counter
    // at count[s * WL_SUBSTRIDE]; a workgroup appends to part blockIdx.x % WL_NSUB
    int list_stride;
    int *mid_list;             // wave kernel: rows it leaves to the workgroup kernel with the medium table ..
    int *mid_count;
    int *big_list;             // .. and to the one with the largest table
    int *big_count;
    const int *list;           // workgroup kernel: its rows
    const int *count;
    int tslots;                // workgroup kernel: slots of its table (a power of two)
    int blk_base;              // workgroup kernel: its first slot in wblk
    ZpField F;
};

// `need` entries (rounded up to 16: rows start on 128-byte lines) from the wave's block, or from the cursor when the block is
// used up / the row is long; ~0 when the region is full.  Wave-uniform.
__device__ __forceinline__ u64d wl_take(u64d &pos, u64d &end, int need, u64d *cursor, u64d cap)
{
    const int lane = threadIdx.x & 63;
    const u64d n = ((u64d)need + 15ull) & ~15ull;
    if (n == 0) return 0;
    if (n > WL_BLK / 2) {
        u64d b = 0;
        if (lane == 0) b = atomicAdd(cursor, n);
        b = __shfl(b, 0);
        return b + n <= cap ? b : ~0ull;
    }
    if (pos + n > end) {
        u64d b = 0;
        if (lane == 0) b = atomicAdd(cursor, WL_BLK);
        b = __shfl(b, 0);
        if (b + WL_BLK > cap) return ~0ull;
        pos = b;
        end = b + WL_BLK;
    }
    const u64d r = pos;
    pos += n;
    return r;
}

// N entries of a lane into a table of mask + 1 slots (a power of two, at most half full): all pending CAS of a round are in
// flight at once (as table_add_n, with the table size at run time).  Returns false when an entry found no slot.
template <bool SMALL, int N>
__device__ __forceinline__ bool wl_insert_n(int *key, typename ZpAcc<SMALL>::type *val, unsigned mask, int shift, const int (&c)[N],
                                            const typename ZpAcc<SMALL>::type (&v)[N], unsigned pending)
{
    unsigned h[N], st[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        const unsigned x = (unsigned)c[j] * 0x9E3779B1u;
        h[j] = x >> shift;
        st[j] = ((x >> 7) & mask) | 1u;
    }
    for (unsigned round = 0; pending != 0 && round <= mask; round++) {
        int k[N];
#pragma unroll
        for (int j = 0; j < N; j++) {
            k[j] = 0;
            if (pending & (1u << j)) k[j] = atomicCAS(&key[h[j]], EMPTY_KEY, c[j]);
        }
#pragma unroll
        for (int j = 0; j < N; j++) {
            if (pending & (1u << j)) {
                if (k[j] == EMPTY_KEY || k[j] == c[j]) {
                    if (SMALL) atomicAdd((int *)&val[h[j]], (int)v[j]);
                    else atomicAdd((u64d *)&val[h[j]], (u64d)v[j]);
                    pending &= ~(1u << j);
                } else {
                    h[j] = (h[j] + st[j]) & mask;
                }
            }
        }
    }
    return pending == 0;
}

// the chunks [t0, t0 + G), [t0 + tstep, ...) .. of a descriptor list in LDS: up to 64 consecutive entries each, multiplied and
// accumulated; the loads of a group of G = 8 are in flight together (a row of a few hundred entries costs ONE round trip here)
template <bool SMALL, int G = 8>
__device__ __forceinline__ bool wl_consume(lds_vint *cd_off, lds_vint *cd_len, lds_vint *cd_mul, int C, int t0, int tstep, const int2 *buf, int *key,
                                           typename ZpAcc<SMALL>::type *val, unsigned mask, int shift, const ZpField &F)
{
    typedef typename ZpAcc<SMALL>::type Acc;
    const int lane = threadIdx.x & 63;
    bool ok = true;
    for (int g0 = t0; g0 < C; g0 += tstep) {
        int2 e[G];
        int cl[G], cm[G];
#pragma unroll
        for (int u = 0; u < G; u++) {
            const int t = min(g0 + u, C - 1);
            const unsigned co = (unsigned)cd_off[t];
            cl[u] = g0 + u < C ? cd_len[t] : 0;
            cm[u] = cd_mul[t];
            e[u] = buf[(size_t)co + (unsigned)min(lane, max(cl[u] - 1, 0))]; // unconditional, clamped into the chunk
        }
#pragma unroll
        for (int h = 0; h < G; h += 4) {
            if (g0 + h >= C) break; // (uniform)
            int cc[4];
            Acc vv[4];
            unsigned pend = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                cc[u] = e[h + u].x;
                vv[u] = ZpAcc<SMALL>::mul_lazy(F, cm[h + u], e[h + u].y);
                if (lane < cl[h + u]) pend |= 1u << u;
            }
            ok &= wl_insert_n<SMALL, 4>(key, val, mask, shift, cc, vv, pend);
        }
    }
    return ok;
}

// ------------------------------------------------------------------------------------------------
// level 0: rows without entries on other pivot columns, W[q] = -U_PN[q].  Their places in W are a prefix sum over their lengths
// (rounded up to 16), computed once per U (k_lev0_sizes + scan): the copy needs no allocation and no table, a team of 16 lanes
// per row.  The cursor of a build starts behind them.
// ------------------------------------------------------------------------------------------------
__global__ void k_lev0_sizes(int npiv, const int *__restrict__ cnt_dev, const WLevRec *__restrict__ recs, unsigned *__restrict__ sz, u64d *__restrict__ entries)
{
    // (over all npiv rows in level order: those behind level 0 take no room; the level's size is still on the device here)
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int cnt = *cnt_dev;
    int npn = 0;
    if (r < cnt) { npn = recs[r].npn; sz[r] = (unsigned)((npn + 15) & ~15); }
    else if (r <= npiv) sz[r] = 0;
    for (int o = 32; o > 0; o >>= 1) npn += __shfl_xor(npn, o);
    if ((threadIdx.x & 63) == 0 && npn) atomicAdd(entries, (u64d)npn);
}

__global__ __launch_bounds__(256) void k_wlevel0(int cnt, const WLevRec *__restrict__ recs, const unsigned *__restrict__ off0, int2 *buf, int2 *__restrict__ wrow,
                                                 int4 *__restrict__ wcol, unsigned wbase, ZpField F)
{
    const int tl = threadIdx.x & 15;
    const int team = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4), nteams = (int)((gridDim.x * blockDim.x) >> 4);
    for (int r = team; r < cnt; r += nteams) {
        const WLevRec rc = recs[r];
        const unsigned mine = wbase + off0[r];
        for (int k = tl; k < rc.npn; k += 16) {
            const int2 e = buf[(size_t)rc.off + k];
            buf[(size_t)mine + k] = make_int2(e.x, zp_neg(F, e.y));
        }
        if (tl == 0) {
            wrow[rc.q] = make_int2((int)mine, rc.npn);
            wcol[rc.pivcol] = make_int4(rc.q, rc.npn, (int)mine, 0);
        }
    }
}

__device__ __forceinline__ void wl_list_append(int *list, int *count, int stride, int row)
{
    const int sub = (int)(blockIdx.x % WL_NSUB);
    list[(size_t)sub * stride + atomicAdd(count + sub * WL_SUBSTRIDE, 1)] = row;
}

// a record as two 16-byte loads from one address for all lanes; unpacked to scalars where it is used
struct WlRecRegs { int4 a; int b; };
__device__ __forceinline__ WlRecRegs wl_rec_load(const WLevRec *p)
{
    WlRecRegs r;
    r.a = *(const int4 *)p;
    r.b = ((const int *)p)[4];
    return r;
}

constexpr int WL_MID_BOUND = 2048;   // rows up to this bound go to the workgroup kernel with 4096 slots (four workgroups per CU)
template <bool SMALL> constexpr int wl_big_slots() { return SMALL ? 16384 : 8192; }
constexpr int wl_mid_slots() { return 2 * WL_MID_BOUND; }
// chunk descriptors of a workgroup batch: NT dependencies at a time + the longest bound / 64
constexpr int wl_wg_ncd(int nt) { return nt + 256; }
template <bool SMALL> constexpr size_t wl_wg_lds_bytes(int slots, int nt)
{
    return (size_t)slots * (sizeof(typename ZpAcc<SMALL>::type) + 4) + (size_t)3 * wl_wg_ncd(nt) * 4 + 256;
}

// ------------------------------------------------------------------------------------------------
// one level, a wave per row.  Rows of a level are independent, so everything a row needs before its first dependent load is
// fetched while the rows before it are worked on: its record three rows ahead, its entries on pivot columns two rows ahead,
// the {offset, length} of the rows of W they name and its first 64 own entries one row ahead (unconditional loads, clamped:
// past the end of the level the last row again).  What is left on the row's own critical path is one round trip for the runs
// of W it combines.
// ------------------------------------------------------------------------------------------------
template <bool SMALL>
__global__ __launch_bounds__(256) void k_wlevel_wave(WLevelArgs a)
{
    typedef typename ZpAcc<SMALL>::type Acc;
    __shared__ int s_key[4][WL_TMAX];
    __shared__ Acc s_val[4][WL_TMAX];
    __shared__ int s_cd[4][3][WL_NCD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int *key = s_key[wave];
    Acc *val = s_val[wave];
    lds_vint *cd_off = (lds_vint *)s_cd[wave][0], *cd_len = (lds_vint *)s_cd[wave][1], *cd_mul = (lds_vint *)s_cd[wave][2];
    const ZpField F = a.F;
    const int wslot = (int)blockIdx.x * 4 + wave;
    const int stride = (int)gridDim.x * 4;
    if (wslot >= a.cnt) return;
    u64d bpos = a.wblk[2 * wslot], bend = a.wblk[2 * wslot + 1];
    bpos = __shfl(bpos, 0);
    bend = __shfl(bend, 0);
    u64d c_ent = 0;
    int c_unavail = 0, c_err = 0, c_big = 0;
    const int last = a.cnt - 1;
    auto dep_load = [&](const WlRecRegs &r) { return a.UPP[(size_t)(unsigned)r.a.y + (unsigned)min(lane, max(min(r.a.z, 64) - 1, 0))]; };
    auto wr_load = [&](const WlRecRegs &r, const int2 &d) { return a.wrow[lane < min(r.a.z, 64) ? d.x : 0]; };
    auto own_load = [&](const WlRecRegs &r) { return a.buf[(size_t)(unsigned)r.a.y + (unsigned)min(lane, max(min(r.a.w, 64) - 1, 0))]; };

    WlRecRegs R0 = wl_rec_load(a.recs + min(wslot, last)), R1 = wl_rec_load(a.recs + min(wslot + stride, last)),
              R2 = wl_rec_load(a.recs + min(wslot + 2 * stride, last));
    int2 D0 = dep_load(R0), D1 = dep_load(R1);
    int2 W0 = wr_load(R0, D0), O0 = own_load(R0);
    for (int i = wslot; i < a.cnt; i += stride) {
        const WlRecRegs R3 = wl_rec_load(a.recs + min(i + 3 * stride, last));
        const int2 D2 = dep_load(R2);
        const int2 W1 = wr_load(R1, D1);
        const int2 O1 = own_load(R1);
        const int q = __builtin_amdgcn_readfirstlane(R0.a.x);
        const unsigned uo = (unsigned)__builtin_amdgcn_readfirstlane(R0.a.y);
        const int npp = __builtin_amdgcn_readfirstlane(R0.a.z), npn = __builtin_amdgcn_readfirstlane(R0.a.w);
        const int pc = __builtin_amdgcn_readfirstlane(R0.b);
        unsigned out_off = 0;
        int n_out = 0;
        bool avail = true, deferred = false;
        if (npp == 0) {
            // level 0: W[q] = -U_PN[q]
            const u64d r = wl_take(bpos, bend, npn, a.wstate + WS_CURSOR, a.wcap);
            if (r == ~0ull) avail = false;
            else {
                out_off = a.wbase + (unsigned)r;
                if (lane < npn) a.buf[(size_t)out_off + lane] = make_int2(O0.x, zp_neg(F, O0.y));
                for (int k = 64 + lane; k < npn; k += 64) {
                    const int2 e = a.buf[(size_t)uo + k];
                    a.buf[(size_t)out_off + k] = make_int2(e.x, zp_neg(F, e.y));
                }
                n_out = npn;
            }
        } else {
            const bool mine = lane < min(npp, 64);
            const int wl = mine ? W0.y : 0;
            if (__ballot(wl < 0) != 0) avail = false; // a row this one needs could not be built (the rows that need THIS one learn it the same way)
            else {
                const int nch = (wl + 63) >> 6;
                int tot_len, tot_ch;
                (void)team_incl_scan<64>(wl, tot_len);
                const int incl_ch = team_incl_scan<64>(nch, tot_ch);
                const int bound = npn + tot_len, own_rest = max(((npn + 63) >> 6) - 1, 0), C = own_rest + tot_ch;
                if (npp > 64 || bound > WL_WAVE_BOUND || C > WL_NCD) {
                    // (with more than 64 dependencies the bound above is a lower bound: the large table then; its kernel checks)
                    const bool tomid = npp <= 64 && bound <= WL_MID_BOUND;
                    if (lane == 0) {
                        if (tomid) wl_list_append(a.mid_list, a.mid_count, a.list_stride, a.rec_base + i);
                        else wl_list_append(a.big_list, a.big_count, a.list_stride, a.rec_base + i);
                    }
                    deferred = true;
                    c_big++;
                } else {
                    // chunk descriptors: what is left of the row's own entries (multiplier -1; the first 64 are in registers), then the
                    // runs of the dependencies
                    const int nm = zp_neg(F, D0.y);
                    const int cbase = own_rest + incl_ch - nch;
                    for (int r = 0; r < nch; r++) {
                        cd_off[cbase + r] = (int)((unsigned)W0.x + 64u * (unsigned)r);
                        cd_len[cbase + r] = min(64, wl - 64 * r);
                        cd_mul[cbase + r] = nm;
                    }
                    for (int r = lane; r < own_rest; r += 64) {
                        cd_off[r] = (int)(uo + 64u * (unsigned)(r + 1));
                        cd_len[r] = min(64, npn - 64 * (r + 1));
                        cd_mul[r] = -1;
                    }
                    int logt = 7;
                    while ((1 << logt) < 2 * bound) logt++;
                    const int T = 1 << logt;
                    const unsigned mask = (unsigned)T - 1u;
                    for (int s = lane; s < T; s += 64) { key[s] = EMPTY_KEY; val[s] = 0; }
                    __builtin_amdgcn_wave_barrier();
                    bool ok = true;
                    {
                        const int cc[1] = {O0.x};
                        const Acc vv[1] = {(Acc)(-O0.y)};
                        ok &= wl_insert_n<SMALL, 1>(key, val, mask, 32 - logt, cc, vv, lane < npn ? 1u : 0u);
                    }
                    ok &= wl_consume<SMALL>(cd_off, cd_len, cd_mul, C, 0, 8, a.buf, key, val, mask, 32 - logt, F);
                    if (__ballot(!ok) != 0) c_err++;
                    __builtin_amdgcn_wave_barrier();
                    // sweep: count, take the space, write
                    int tot = 0;
                    for (int s0 = 0; s0 < T; s0 += 64) {
                        const int k = key[s0 + lane];
                        const int r = k == EMPTY_KEY ? 0 : acc_reduce_short<SMALL>(F, val[s0 + lane]);
                        tot += __popcll(__ballot(r != 0));
                    }
                    const u64d r0 = wl_take(bpos, bend, tot, a.wstate + WS_CURSOR, a.wcap);
                    if (r0 == ~0ull) avail = false;
                    else {
                        out_off = a.wbase + (unsigned)r0;
                        int pos = 0;
                        for (int s0 = 0; s0 < T; s0 += 64) {
                            const int k = key[s0 + lane];
                            const int r = k == EMPTY_KEY ? 0 : acc_reduce_short<SMALL>(F, val[s0 + lane]);
                            const u64d mm = __ballot(r != 0);
                            if (r != 0) a.buf[(size_t)out_off + pos + __popcll(mm & lanemask_lt())] = make_int2(k, r);
                            pos += __popcll(mm);
                        }
                        n_out = tot;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (!deferred) {
            if (lane == 0) {
                const int len = avail ? n_out : -1;
                a.wrow[q] = make_int2((int)out_off, len);
                a.wcol[pc] = make_int4(q, len, (int)out_off, 0);
            }
            if (avail) c_ent += (u64d)n_out; else c_unavail++;
        }
        R0 = R1; R1 = R2; R2 = R3;
        D0 = D1; D1 = D2;
        W0 = W1;
        O0 = O1;
    }
    if (lane == 0) {
        a.wblk[2 * wslot] = bpos;
        a.wblk[2 * wslot + 1] = bend;
        (void)c_ent; (void)c_unavail; (void)c_big; // (statistics: k_wstats works them out of wrow[] after the build -- tens of thousands of
                                                   // waves adding to three words at the end of every level kernel cost 0.7 ms of a 2 ms build)
        if (c_err) atomicAdd(a.wstate + WS_ERROR, (u64d)c_err);
    }
}

// ------------------------------------------------------------------------------------------------
// the long rows of a level, a workgroup per row: a table of a.tslots slots in dynamic LDS (4096: four workgroups per CU; or all
// a CU has).  The next row's record and its first 256 dependencies are fetched while this one is worked on.
// ------------------------------------------------------------------------------------------------
// NT threads per workgroup: 1024 with the largest table (one workgroup per CU whatever its size), 512 with the medium one (four
// per CU): clearing, filling and sweeping a table of thousands of slots is most of a long row's time, and it divides by the waves
template <bool SMALL, int NT>
__global__ __launch_bounds__(NT) void k_wlevel_wg(WLevelArgs a)
{
    constexpr int NW = NT / 64, WL_WG_NCD = wl_wg_ncd(NT);
    typedef typename ZpAcc<SMALL>::type Acc;
    const int TB = a.tslots;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    Acc *val = (Acc *)s_raw;
    int *key = (int *)(s_raw + sizeof(Acc) * (size_t)TB);
    lds_vint *cd_off = (lds_vint *)(key + TB), *cd_len = cd_off + WL_WG_NCD, *cd_mul = cd_len + WL_WG_NCD;
    lds_vint *s_misc = cd_mul + WL_WG_NCD; // 64 words: [0, 16) per-wave sums, [16, 32) per-wave flags, [32, 48) per-wave chunk counts, [48, 50) the allocation
    // its rows: a list the wave kernel (or this kernel with the medium table) left, or -- levels of few rows, which skip the wave
    // kernel -- all cnt rows of the level
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ int s_pre[WL_NSUB + 1]; // list mode: rows in the parts before part s
    int nrows = a.cnt;
    if (a.list) {
        if (tid < WL_NSUB) {
            const int c = a.count[tid * WL_SUBSTRIDE];
            int tot;
            const int incl = team_incl_scan<64>(c, tot);
            s_pre[tid + 1] = incl;
            if (tid == 0) s_pre[0] = 0;
        }
        __syncthreads();
        nrows = s_pre[WL_NSUB];
    }
    if ((int)blockIdx.x >= nrows) return;
    const ZpField F = a.F;
    const int last = nrows - 1, stride = (int)gridDim.x;
    // row i of this kernel: in list mode the (i - pre[s])-th row of the part s that holds position i
    auto row_of = [&](int i) {
        i = min(i, last);
        if (!a.list) return a.rec_base + i;
        int s0 = 0;
#pragma unroll
        for (int step = WL_NSUB / 2; step > 0; step >>= 1)
            if (s_pre[s0 + step] <= i) s0 += step;
        return a.list[(size_t)s0 * a.list_stride + (i - s_pre[s0])];
    };
    auto rec_of = [&](int i) { return wl_rec_load(a.recs + row_of(i)); };
    auto dep_load = [&](const WlRecRegs &r) { return a.UPP[(size_t)(unsigned)r.a.y + (unsigned)min(tid, max(min(r.a.z, NT) - 1, 0))]; };
    auto wr_load = [&](const WlRecRegs &r, const int2 &d) { return a.wrow[tid < min(r.a.z, NT) ? d.x : 0]; };
    auto own_load = [&](const WlRecRegs &r) { return a.buf[(size_t)(unsigned)r.a.y + (unsigned)min(tid, max(min(r.a.w, NT) - 1, 0))]; };
    u64d bpos = a.wblk[2 * (a.blk_base + (int)blockIdx.x)], bend = a.wblk[2 * (a.blk_base + (int)blockIdx.x) + 1]; // (used by thread 0)
    WlRecRegs R0 = rec_of(blockIdx.x), R1 = rec_of(blockIdx.x + stride);
    int2 D0 = dep_load(R0), O0 = own_load(R0);
    int2 W0 = wr_load(R0, D0);
    for (int i = blockIdx.x; i < nrows; i += stride) {
        const WlRecRegs R2 = rec_of(i + 2 * stride);
        const int2 D1 = dep_load(R1), O1 = own_load(R1);
        const int q = __builtin_amdgcn_readfirstlane(R0.a.x);
        const unsigned uo = (unsigned)__builtin_amdgcn_readfirstlane(R0.a.y);
        const int npp = __builtin_amdgcn_readfirstlane(R0.a.z), npn = __builtin_amdgcn_readfirstlane(R0.a.w);
        const int pc = __builtin_amdgcn_readfirstlane(R0.b);
        // ---- bound of the row, and whether every row it needs is there
        long long mylen = 0;
        bool un = false;
        if (tid < min(npp, NT)) { un = W0.y < 0; mylen = max(W0.y, 0); }
        for (int k = NT + tid; k < npp; k += NT) {
            const int2 d = a.UPP[(size_t)uo + k];
            const int2 wr = a.wrow[d.x];
            un |= wr.y < 0;
            mylen += max(wr.y, 0);
        }
        for (int o = 32; o > 0; o >>= 1) mylen += __shfl_xor(mylen, o);
        const bool wun = __ballot(un) != 0;
        if (lane == 0) { s_misc[wave] = (int)min(mylen, (long long)(INT_MAX / 32)); s_misc[16 + wave] = wun ? 1 : 0; }
        __syncthreads();
        long long bound = npn;
        bool unavail = false;
        for (int w2 = 0; w2 < NW; w2++) { bound += s_misc[w2]; unavail |= s_misc[16 + w2] != 0; }
        bool avail = !unavail && bound <= TB / 2;
        unsigned out_off = 0;
        int n_out = 0;
        if (!avail && !unavail && !a.list && a.big_list) {
            // range mode with the medium table: the row goes to the kernel with the largest one (which publishes it)
            if (tid == 0) wl_list_append(a.big_list, a.big_count, a.list_stride, a.rec_base + i);
            __syncthreads();
            R0 = R1; R1 = R2;
            W0 = wr_load(R0, D1);
            D0 = D1;
            O0 = O1;
            continue;
        }
        if (avail) { // (uniform over the workgroup)
            int logt = 10;
            while ((1ll << logt) < 2 * bound) logt++;
            const int T = 1 << logt;
            const unsigned mask = (unsigned)T - 1u;
            const int shift = 32 - logt;
            for (int s = tid; s < T; s += NT) { key[s] = EMPTY_KEY; val[s] = 0; }
            __syncthreads();
            bool ok = true;
            // the row's own entries on non-pivot columns (the first NT were fetched with the record)
            for (int k = tid; k < npn; k += NT) {
                const int2 e = k < NT ? O0 : a.buf[(size_t)uo + k];
                const int cc[1] = {e.x};
                const Acc vv[1] = {(Acc)(-e.y)};
                ok &= wl_insert_n<SMALL, 1>(key, val, mask, shift, cc, vv, 1u);
            }
            // its dependencies, NT at a time: their runs become chunks, the waves take groups of 8 chunks in turn
            for (int b0 = 0; b0 < npp; b0 += NT) {
                int2 d = make_int2(0, 0), wr = make_int2(0, 0);
                if (b0 == 0) { if (tid < npp) { d = D0; wr = W0; } }
                else if (b0 + tid < npp) {
                    d = a.UPP[(size_t)uo + b0 + tid];
                    wr = a.wrow[d.x];
                }
                const int wl = wr.y, nch = (wl + 63) >> 6;
                int wtot;
                const int incl = team_incl_scan<64>(nch, wtot);
                __syncthreads(); // (the chunks of the batch before are consumed)
                if (lane == 0) s_misc[32 + wave] = wtot;
                __syncthreads();
                int cbase = 0, Cb = 0;
                for (int w2 = 0; w2 < NW; w2++) {
                    const int cw = s_misc[32 + w2];
                    if (w2 < wave) cbase += cw;
                    Cb += cw;
                }
                const int nm = zp_neg(F, d.y);
                const int at = cbase + incl - nch;
                for (int r = 0; r < nch; r++) {
                    cd_off[at + r] = (int)((unsigned)wr.x + 64u * (unsigned)r);
                    cd_len[at + r] = min(64, wl - 64 * r);
                    cd_mul[at + r] = nm;
                }
                __syncthreads();
                ok &= wl_consume<SMALL>(cd_off, cd_len, cd_mul, Cb, wave * 8, NW * 8, a.buf, key, val, mask, shift, F);
            }
            __syncthreads();
            // sweep: wave w owns the slots it * NT + w * 64 + lane
            int tot = 0;
            for (int s0 = wave * 64; s0 < T; s0 += NT) {
                const int k = key[s0 + lane];
                const int r = k == EMPTY_KEY ? 0 : acc_reduce_short<SMALL>(F, val[s0 + lane]);
                tot += __popcll(__ballot(r != 0));
            }
            const bool wok = __ballot(!ok) == 0;
            if (lane == 0) { s_misc[wave] = tot; if (!wok) atomicAdd(a.wstate + WS_ERROR, 1ull); }
            __syncthreads();
            int pre = 0;
            for (int w2 = 0; w2 < NW; w2++) {
                const int cw = s_misc[w2];
                if (w2 < wave) pre += cw;
                n_out += cw;
            }
            if (tid == 0) {
                // (as wl_take, by one thread: from the workgroup's block, or from the cursor when that is used up / the row is long)
                const u64d n = ((u64d)n_out + 15ull) & ~15ull;
                u64d b = 0;
                bool fits = true;
                if (n > WL_BLK / 2) {
                    b = atomicAdd(a.wstate + WS_CURSOR, n);
                    fits = b + n <= a.wcap;
                } else if (n > 0) {
                    if (bpos + n > bend) {
                        const u64d nb = atomicAdd(a.wstate + WS_CURSOR, WL_BLK);
                        fits = nb + WL_BLK <= a.wcap;
                        if (fits) { bpos = nb; bend = nb + WL_BLK; }
                    }
                    if (fits) { b = bpos; bpos += n; }
                }
                s_misc[48] = fits ? 1 : 0;
                s_misc[49] = (int)(unsigned)b;
            }
            __syncthreads();
            avail = s_misc[48] != 0;
            if (avail) {
                out_off = a.wbase + (unsigned)s_misc[49];
                int pos = pre;
                for (int s0 = wave * 64; s0 < T; s0 += NT) {
                    const int k = key[s0 + lane];
                    const int r = k == EMPTY_KEY ? 0 : acc_reduce_short<SMALL>(F, val[s0 + lane]);
                    const u64d mm = __ballot(r != 0);
                    if (r != 0) a.buf[(size_t)out_off + pos + __popcll(mm & lanemask_lt())] = make_int2(k, r);
                    pos += __popcll(mm);
                }
            }
        }
        if (tid == 0) {
            const int len = avail ? n_out : -1;
            a.wrow[q] = make_int2((int)out_off, len);
            a.wcol[pc] = make_int4(q, len, (int)out_off, 0);
        }
        __syncthreads();
        R0 = R1; R1 = R2;
        W0 = wr_load(R0, D1);
        D0 = D1;
        O0 = O1;
    }
    if (tid == 0) {
        a.wblk[2 * (a.blk_base + (int)blockIdx.x)] = bpos;
        a.wblk[2 * (a.blk_base + (int)blockIdx.x) + 1] = bend;
    }
}
