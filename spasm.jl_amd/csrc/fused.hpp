// fused.hpp -- the Schur row of a non-pivot row in ONE kernel: the plan of the row in the prologue, then its stream.
//
// Same contract as k_wplan + k_bin + k_wstream (stream.hpp): x_a = B[k]_N + sum over the entries (c, a_c) of B[k] on pivot
// columns of a_c * W[qinv(c)], the non-pivot part of the solution of x * U = B[k] (reference src/SpaSM.jl:694-713, :761-762),
// every entry stored at a fixed position of the row's stream while it goes by, LDS tag tables for the duplicate columns, the
// duplicates merged afterwards by k_stream_fix.  What is gone is everything BETWEEN the rows of W and the stream: no chunk
// records (16 bytes per 64 entries written and read back), no copy of the row's own entries, no bound array + scan, no
// binning pass, no descriptors.  A wave
//   * takes blocks of consecutive row slots from eight counters (its own first, the others when that one is used up: the
//     rows of a round differ by a factor of a hundred in length, and a static deal leaves the chip waiting for its slowest wave);
//   * keeps five rows in flight, one stage each: {start, length} of a row (a block at a time, a lane per row), its entries
//     (a lane per entry), the pivot-column bit of every entry, {length, offset} of the row of W of every entry on a pivot
//     column, the first D chunks of those runs -- a row's own critical path is the last of these round trips;
//   * plans the row in registers: a prefix sum over the lengths of its runs gives every run its place in the stream, the
//     entries on non-pivot columns take the front; the space comes from ONE cursor, 16384 entries at a time per wave (rows
//     start on 128-byte lines), so the Schur complement is written compactly, without a slot sized by a bound per row;
//   * sizes the three tag tables by the row (the first one holds 8/5 of the stream, a power of two) inside a fixed LDS share, and
//     clears what the row used.
// Rows whose stream is longer than the tables of this launch hold go on a list for a launch with larger tables (fewer waves
// per CU); rows it cannot take at all (more than 64 own entries, a row of W that could not be built, a zero among its entries,
// no space left) go on the list of the general path (k_wplan .. of stream.hpp with the hash-table kernels behind it).
#pragma once
#include "stream.hpp"

constexpr int FZ_NWORK = 8;        // block counters
constexpr int FZ_WSTRIDE = 32;     // words between two of them (a line each)
constexpr u64d FZ_SBLK = 16384;    // entries of S a wave takes from the cursor at a time
constexpr int FZ_B = 8;            // row slots per block
constexpr int FZ_MINLOGT = 8;     // (the third table then has 2^6 slots: slot + the 18 bits of the word still name the column)

struct FusedArgs {
    int nrows;                 // row slots of the round
    const int *slots;          // NULL: the slots 0 .. nrows-1; else the slots to take (a list another launch left) ..
    const int *slot_count;     // .. and how many (device)
    const int4 *rinfo;         // per slot: {start (low, high), length, originating row} of the row's own entries
    const int2 *ent;
    const unsigned *pbits;     // bit j: column j is a pivot column of this round
    const int4 *wcol;          // per pivot column: {pivot index, length of its row of W (-1: not available), offset, -}
    const int2 *buf;           // the U_PN + own + W buffer
    int2 *Sent;
    u64d scap;                 // entries Sent holds
    u64d *scursor;
    i64d *Sstart;
    int *Slen;
    int *Slead;
    int *Sorig;
    int2 *fixbuf;              // [slot][SFIX] duplicates found in the row, merged afterwards by k_stream_fix
    int *fixcnt;
    int *long_list;            // slots whose stream is beyond this launch's tables but within long_bound
    int *long_count;
    int long_bound;
    int *rej_list;             // slots left to the general path
    int *rej_count;
    unsigned *work;            // FZ_NWORK block counters
    RoundCounters *ctr;
    int cls;                   // index for the per-class counters
    int free_cols;
    ZpField F;
};

// LDS of one wave: three tag tables of 2^LOGTMAX, half and a quarter of that, 64 B of counters, fix-up list, loser list
template <int LOGTMAX> struct FzLds {
    static constexpr int WORDS = (1 << LOGTMAX) + (1 << (LOGTMAX - 1)) + (1 << (LOGTMAX - 2));
    static constexpr size_t TABB = (size_t)4 * WORDS, MISCB = 64, FIXB = (size_t)SFIX * 8, LSTB = (size_t)SLCAP * 16;
    static constexpr size_t SLOT = TABB + MISCB + FIXB + LSTB;
};
__host__ __device__ constexpr int fz_cap(int logt) { return 5 << (logt - 3); } // a first table of 2^logt words at most 5/8 full
template <int LOGTMAX> __host__ __device__ constexpr size_t fused_lds_bytes(int wpb) { return FzLds<LOGTMAX>::SLOT * (size_t)wpb; }

// the three tables of the current row: LDS pointers of their first words, log2 of the first one's size (all wave-uniform)
struct FzTab {
    unsigned *t[3];
    int logt;
};

template <int LV> __device__ __forceinline__ unsigned fz_want(const FzTab &tb, int c, int pos1, unsigned *&slotp)
{
    constexpr unsigned K = LV == 0 ? STREAM_K1 : (LV == 1 ? STREAM_K2 : STREAM_K3);
    const int L = tb.logt - LV;
    const unsigned x = stream_mul24(c, K);
    slotp = tb.t[LV] + __builtin_amdgcn_ubfe(x, 24 - L, L);
    return (x << 14) | (unsigned)pos1;
}

// what is left of an insertion after the first table (stream_insert_n of stream.hpp, tables sized at run time): entries that
// met ANOTHER column (left >= 2^14) try the second table, then the third.  The compare-and-swaps of a level are all issued before
// any answer is looked at: N round trips of LDS latency per level would otherwise be the longest part of a group.
template <int LV, int N>
__device__ __forceinline__ void fz_insert_level(const FzTab &tb, const int (&c)[N], const int (&pos1)[N], unsigned (&old)[N], unsigned (&left)[N])
{
    unsigned w[N], o[N];
    bool need[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        need[j] = left[j] >= 0x4000u;
        o[j] = 0;
        w[j] = 0;
        if (need[j]) {
            unsigned *sp;
            w[j] = fz_want<LV>(tb, c[j], pos1[j], sp);
            o[j] = atomicCAS(sp, 0u, w[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (need[j]) {
            old[j] = o[j];
            left[j] = o[j] == 0 ? 0u : o[j] ^ w[j];
        }
    }
}
template <int N>
__device__ __forceinline__ void fz_insert_rest(const FzTab &tb, const int (&c)[N], const int (&pos1)[N], unsigned (&old)[N], unsigned (&left)[N])
{
    fz_insert_level<1, N>(tb, c, pos1, old, left);
    fz_insert_level<2, N>(tb, c, pos1, old, left);
}

// a value every lane holds, as a scalar
__device__ __forceinline__ int fz_sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u64d fz_sgpr64(u64d v)
{
    return ((u64d)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// keeps the use of a value where the source has it (the compiler otherwise takes the answer of an atomic right behind it)
__device__ __forceinline__ u64d fz_pin(u64d v)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi)::"memory");
    return ((u64d)hi << 32) | lo;
}
__device__ __forceinline__ int fz_lane_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ i64d fz_lane_i64(i64d v, int l)
{
    return (i64d)(((u64d)(unsigned)__builtin_amdgcn_readlane((int)((u64d)v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)v, l));
}

// the plan of a row (all wave-uniform).  Its chunks -- at most 64 consecutive entries of a run of W each -- sit in a table of 64
// lanes shared by the rows in flight: chunk j of the row in lane (cbase + j) & 63.
constexpr int FZ_MAXC = 32;        // chunks a row of the wave kernel may have: the double groups of two rows fit the 64 lanes
struct FzPlan {
    int t;          // row slot; < 0: nothing to do (no row, or the row went on a list)
    int nN;         // own entries on non-pivot columns (the front of the stream)
    int bound;      // length of the stream
    int logt;
    int C;          // chunks
    int cbase;      // lane of its first chunk
    int nruns;
    u64d mN;        // lanes whose own entry sits on a non-pivot column
    i64d sbase;     // where the row starts in S
};

// LIST: the launch takes the slots of a list (a.slots, a.slot_count) instead of all of them
template <int LOGTMAX, int WPB, bool SMALL, bool LIST>
__global__ __launch_bounds__(WPB * 64) void k_schur_fused(FusedArgs a)
{
    typedef FzLds<LOGTMAX> L;
    constexpr int FCAP = SFIX;
    constexpr int Q = 4;      // chunks whose loads / table traffic are in flight together
    constexpr int DG = 2 * Q; // a double group: group A in ringA, group B in ringB
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char *base = s_raw + (size_t)wave * L::SLOT;
    unsigned *t1 = (unsigned *)base;
    int *misc = (int *)(base + L::TABB);
    int2 *fix = (int2 *)(base + L::TABB + L::MISCB);
    int4 *lst = (int4 *)(base + L::TABB + L::MISCB + L::FIXB);
    const ZpField F = a.F;
    const int N = LIST ? *a.slot_count : a.nrows;
    const int nblk = (N + FZ_B - 1) / FZ_B;

    for (int s = lane * 4; s < L::WORDS; s += 256) *(int4 *)(t1 + s) = make_int4(0, 0, 0, 0);
    if (lane < 16) misc[lane] = 0;
    __builtin_amdgcn_wave_barrier();

    // ---- blocks of FZ_B consecutive row slots, from the counters: the lb-th block of counter x is block lb * FZ_NWORK + x
    int home = (int)(blockIdx.x & (FZ_NWORK - 1)); // the counter this wave takes blocks from
    int tried = 0;                                 // counters found used up
    // (atomicInc, not atomicAdd: the compiler rewrites a uniform add into one add per wave + a broadcast, and waits for the answer
    // on the spot; the answer is wanted a whole row later)
    auto ask = [&]() -> unsigned { // (lane 0's value counts)
        unsigned r = 0;
        if (lane == 0) r = atomicInc(a.work + (size_t)home * FZ_WSTRIDE, 0xffffffffu);
        return r;
    };
    // turns the answer into a block, going round the counters when this one is used up; -1: no work left
    auto steal = [&]() -> int { // (the end of the launch: every answer is waited for)
        for (;;) {
            tried++;
            if (tried >= FZ_NWORK) return -1;
            home = (home + 1) & (FZ_NWORK - 1);
            const u64d lb = (unsigned)fz_sgpr((int)ask());
            const u64d g = lb * FZ_NWORK + (u64d)home;
            if (g < (u64d)nblk) return (int)g;
        }
    };
    auto resolve = [&](unsigned r) -> int {
        const u64d lb = (unsigned)fz_sgpr((int)r);
        const u64d g = lb * FZ_NWORK + (u64d)home;
        if (g < (u64d)nblk) return (int)g;
        return steal();
    };
    // row number s of this wave is position (s % FZ_B) of its block number s / FZ_B: g_cur, g_nxt = the blocks of the current row
    // and the one after it, g_new = the one after that, asked for when the current block starts
    int s_limit = INT_MAX; // first row number beyond the wave's last block
    int g_cur = resolve(ask()), g_nxt = -1, g_new = -1;
    if (g_cur < 0) s_limit = 0;
    else {
        g_nxt = resolve(ask());
        if (g_nxt < 0) s_limit = FZ_B;
    }
    // index (into the slots of this launch) of row number s, s within the current block or the next; >= N: no such row
    auto index_of = [&](int s, int s_blk0) -> int {
        if (s < 0 || s >= s_limit) return N;
        const int g = s < s_blk0 + FZ_B ? g_cur : g_nxt;
        return g < 0 ? N : min(g * FZ_B + (s & (FZ_B - 1)), N);
    };

    // ---- the wave's block of S
    u64d sb_pos = 0, sb_end = 0;
    auto take_s = [&](int need) -> i64d { // `need` entries, rounded up to 16; -1: S is full
        const u64d n = ((u64d)need + 15ull) & ~15ull;
        if (n == 0) return 0;
        if (n > FZ_SBLK / 2) {
            u64d b = 0;
            if (lane == 0) b = atomicAdd(a.scursor, n);
            b = fz_sgpr64(b);
            return b + n <= a.scap ? (i64d)b : -1;
        }
        if (sb_pos + n > sb_end) {
            u64d b = 0;
            if (lane == 0) b = atomicAdd(a.scursor, FZ_SBLK);
            b = fz_sgpr64(b);
            if (b + FZ_SBLK > a.scap) return -1;
            sb_pos = b;
            sb_end = b + FZ_SBLK;
        }
        const u64d r = sb_pos;
        sb_pos += n;
        return (i64d)r;
    };

    // ---- the chunk table (a lane per chunk): where the chunk starts in buf, the multiplier of its run, its stream position << 8 | entries
    int T_off = 0, T_mul = 0, T_pn = 1;

    // ---- the plan of row number s (its entries e, the lanes mP on pivot columns, their {-, length, offset, -} of W); its chunks go
    // to the lanes from cbase on
    auto make_plan = [&](int t, int ln, const int2 &e, u64d mP, const int4 &wc, int cbase) -> FzPlan {
        FzPlan pl;
        pl.t = -1;
        pl.nN = 0; pl.bound = 0; pl.logt = FZ_MINLOGT; pl.C = 0; pl.cbase = cbase & 63; pl.nruns = 0; pl.mN = 0; pl.sbase = 0;
        if (t < 0) return pl;
        const bool valid = lane < min(ln, 64);
        const bool isP = (mP >> lane) & 1ull;
        const int wl = isP ? wc.y : 0;
        // what this kernel does not take: more than 64 own entries, a zero among them, a row of W that could not be built
        const bool bad = ln > 64 || __ballot((valid && e.y == 0) || wl < 0) != 0;
        const int nch = (max(wl, 0) + 63) >> 6;
        int tot = 0, ctot = 0;
        const int incl = team_incl_scan<64>(max(wl, 0), tot);
        const int cincl = team_incl_scan<64>(nch, ctot);
        const u64d mN = __ballot(valid && !isP);
        const int nN = __popcll(mN);
        const i64d bound = (i64d)nN + (i64d)tot;
        bool tolong = false, rej = bad || bound > (i64d)a.free_cols;
        if (!rej && (bound > (i64d)fz_cap(LOGTMAX) || ctot > FZ_MAXC)) {
            if (a.long_list && bound <= (i64d)a.long_bound) tolong = true;
            else rej = true;
        }
        i64d sb = 0;
        if (!rej && !tolong) {
            sb = take_s((int)bound);
            if (sb < 0) rej = true;
        }
        if (rej || tolong) {
            if (lane == 0) {
                if (tolong) a.long_list[atomicAdd(a.long_count, 1)] = t;
                else a.rej_list[atomicAdd(a.rej_count, 1)] = t;
            }
            return pl;
        }
        int logt = FZ_MINLOGT;
        while (fz_cap(logt) < (int)bound) logt++;
        pl.t = t;
        pl.nN = nN;
        pl.bound = (int)bound;
        pl.logt = logt;
        pl.C = ctot;
        pl.nruns = __popcll(mP);
        pl.mN = mN;
        pl.sbase = sb;
        // the chunks: run by run, the lane of chunk j takes what chunk j is
        const int cb0 = cincl - nch, rpos = nN + incl - max(wl, 0);
        const int jl = (lane - cbase) & 63;
        for (u64d m = mP; m != 0; m &= m - 1) {
            const int l = __ffsll((long long)m) - 1;
            const int cb = fz_lane_i32(cb0, l), nc = fz_lane_i32(nch, l), wo = fz_lane_i32(wc.z, l), w_l = fz_lane_i32(wl, l), mu = fz_lane_i32(e.y, l),
                      rp = fz_lane_i32(rpos, l);
            const int j = jl - cb;
            if (j >= 0 && j < nc) {
                T_off = wo + 64 * j;
                T_mul = mu;
                T_pn = ((rp + 64 * j) << 8) | min(64, w_l - 64 * j);
            }
        }
        if (lane == 0) a.Sstart[t] = sb;
        return pl;
    };
    auto padded = [&](int C) { return max(DG, (C + DG - 1) & ~(DG - 1)); }; // every row takes a whole number of double groups, at least one

    // ---- chunk loads.  The chunks of the rows form ONE stream: group A of a double group lives in ringA, group B in ringB, and
    // while double group d is worked on the loads of double group d + 1 are issued -- of the same row, or the first one of the next
    // row (whose plan is made before the current row starts).  Always Q loads per request, with clamped chunk numbers: a number of
    // loads that depends on the row would make every wait for an older load a wait for all of them.
    int2 ringA[Q], ringB[Q];
    auto refill = [&](const FzPlan &cur, int cp, const FzPlan &nxt, int j0, int2 (&r)[Q]) {
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int j = j0 + q;
            // chunk j of the current row, or chunk j - cp of the next one; beyond the last chunk: the last chunk again (a row without
            // chunks: whatever its first lane holds -- offset 0, one entry, at the start)
            const int g = j < cp ? cur.cbase + min(j, max(cur.C - 1, 0)) : nxt.cbase + min(j - cp, max(nxt.C - 1, 0));
            const unsigned off = (unsigned)fz_lane_i32(T_off, g & 63);
            const int n = fz_lane_i32(T_pn, g & 63) & 0xff;
            r[q] = a.buf[(size_t)off + (unsigned)min(lane, max(n - 1, 0))];
        }
    };

    // ---- the stages.  I: {start, length, slot} of a row (slot < 0: no row) + its originating row; E: its entries, a lane each;
    // PB: the word of pbits of every entry; WC: {-, length, offset, -} of the row of W of every entry on a pivot column
    int4 I5 = make_int4(0, 0, 0, 0), I4 = I5;
    int T5 = -1, T4 = -1, T3 = -1, T2 = -1, T1 = -1; // slots (scalar)
    int SL6 = 0, SL5 = 0;                            // LIST: the word of the list of row s + 6 / s + 5
    int L3 = 0, L2 = 0, L1 = 0;                      // lengths (scalar)
    int2 E4 = make_int2(0, 0), E3 = E4, E2 = E4, E1 = E4, E0 = E4;
    unsigned PB3 = 0, PB2 = 0;
    int4 WC2 = make_int4(0, 0, 0, 0), WC1 = WC2;
    u64d mP2 = 0, mP1 = 0;
    FzPlan P0, P1;
    P0.t = -1; P0.nN = 0; P0.bound = 0; P0.logt = FZ_MINLOGT; P0.C = 0; P0.cbase = 0; P0.nruns = 0; P0.mN = 0; P0.sbase = 0;
    u64d c_nnz = 0, c_seg = 0;
    int c_rows = 0, c_redo = 0;
    int s_blk0 = 0; // first row number of the current block

    // (the first five passes only fill the stages: row 0 is the current row at s = 0)
    for (int s = LIST ? -6 : -5; s < s_limit; s++) {
        // ---- blocks: when a block starts the one after the next is asked for (the answer is taken at the end of this pass)
        unsigned asked = 0;
        const bool block_start = s > 0 && (s & (FZ_B - 1)) == 0;
        if (block_start) {
            s_blk0 = s;
            g_cur = g_nxt;
            g_nxt = g_new;
        }
        const bool asking = (s == -5 || block_start) && s_limit == INT_MAX; // (at s = -5: for block 2)
        if (asking) asked = ask();
        // ---- the stages of the rows ahead: row s + 5 (who), s + 4 (entries), s + 3 (pivot bits), s + 2 (rows of W) -- four loads,
        // always (lanes without an entry repeat the row's last one, rows that do not exist read the first words of the arrays)
        {
            const int i5 = index_of(s + 5, s_blk0);
            if (LIST) {
                // (a list of slots: one stage more, all lanes the same word; row s + 6 is in this block or the next)
                const int i6 = s + 6 < s_limit ? min((s + 6 < s_blk0 + FZ_B ? g_cur : g_nxt) * FZ_B + ((s + 6) & (FZ_B - 1)), N - 1) : 0;
                SL6 = a.slots[max(i6, 0)];
                T5 = i5 < N ? fz_sgpr(SL5) : -1;
            } else T5 = i5 < N ? i5 : -1;
            I5 = a.rinfo[max(T5, 0)];
        }
        int L4 = 0;
        {
            const i64d st4 = (i64d)(((u64d)(unsigned)fz_sgpr(I4.y) << 32) | (unsigned)fz_sgpr(I4.x));
            L4 = T4 < 0 ? 0 : fz_sgpr(I4.z);
            E4 = a.ent[(T4 < 0 ? 0 : st4) + min(lane, max(min(L4, 64) - 1, 0))];
            if (lane == 0 && T4 >= 0) a.Sorig[T4] = I4.w;
        }
        PB3 = a.pbits[(unsigned)E3.x >> 5];
        {
            const int l2 = L2;
            const bool p2 = lane < min(l2, 64) && ((PB2 >> (E2.x & 31)) & 1u);
            mP2 = __ballot(p2);
            WC2 = a.wcol[p2 ? E2.x : 0];
        }
        // ---- the plan of row s + 1; its chunks go behind those of the current row
        const int cp = padded(P0.C);
        P1 = make_plan(T1, L1, E1, mP1, WC1, P0.cbase + cp);

        // ---- the current row
        const bool live = P0.t >= 0;
        FzTab tb;
        tb.logt = P0.logt;
        tb.t[0] = t1;
        tb.t[1] = t1 + (1 << P0.logt);
        tb.t[2] = t1 + (1 << P0.logt) + (1 << (P0.logt - 1));
        unsigned char *const rowp = (unsigned char *)(a.Sent + P0.sbase);
        int mylead = INT_MAX;
        if (live) {
            // own entries on non-pivot columns: the front of the stream
            const bool isN = (P0.mN >> lane) & 1ull;
            if (isN) {
                const int pos = __popcll(P0.mN & lanemask_lt());
                __builtin_nontemporal_store(((long long)(unsigned)E0.y << 32) | (unsigned)E0.x, (long long *)(rowp + ((unsigned)pos << 3)));
                mylead = E0.x;
                const int cc[1] = {E0.x}, pp[1] = {pos + 1};
                unsigned oo[1], ll[1];
                unsigned *sp;
                const unsigned w = fz_want<0>(tb, cc[0], pp[0], sp);
                oo[0] = atomicCAS(sp, 0u, w);
                ll[0] = oo[0] == 0 ? 0u : oo[0] ^ w;
                fz_insert_rest<1>(tb, cc, pp, oo, ll);
                // (the own entries of a row have distinct columns and come first: no duplicate here, but an entry may lose in all tables)
                stream_report(stream_outcome(ll[0]), oo[0], true, E0.x, E0.y, pos + 1, misc, fix, FCAP, lst);
            }
        }
        // the runs, a double group at a time (a row without chunks still takes one: it carries the requests of the next row)
        {
            auto do_group = [&](const int2 (&ring)[Q], int g0) {
                int cc[Q], vv[Q], pp[Q];
                bool act[Q];
                unsigned oo[Q], left[Q], w1[Q];
                unsigned *s1[Q];
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const int gq = (P0.cbase + min(g0 + q, max(P0.C - 1, 0))) & 63;
                    const int pn = fz_lane_i32(T_pn, gq), mul = fz_lane_i32(T_mul, gq);
                    act[q] = g0 + q < P0.C && lane < (pn & 0xff);
                    cc[q] = ring[q].x;
                    vv[q] = stream_mul<SMALL>(F, mul, ring[q].y);
                    pp[q] = (int)((unsigned)pn >> 8) + lane + 1;
                    w1[q] = fz_want<0>(tb, cc[q], pp[q], s1[q]);
                    oo[q] = 0;
                }
                // store + first-table CAS of the Q chunks, all in flight together
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    if (act[q]) {
                        __builtin_nontemporal_store(((long long)(unsigned)vv[q] << 32) | (unsigned)cc[q], (long long *)(rowp + ((unsigned)(pp[q] - 1) << 3)));
                        mylead = min(mylead, cc[q]);
                        oo[q] = atomicCAS(s1[q], 0u, w1[q]);
                    }
                }
#pragma unroll
                for (int q = 0; q < Q; q++) left[q] = oo[q] == 0 ? 0u : oo[q] ^ w1[q];
                fz_insert_rest<Q>(tb, cc, pp, oo, left);
                unsigned any = 0;
#pragma unroll
                for (int q = 0; q < Q; q++) any |= left[q];
                if (__ballot(any != 0) != 0) { // rare: duplicates, triple losers
#pragma unroll
                    for (int q = 0; q < Q; q++) stream_report(stream_outcome(left[q]), oo[q], act[q], cc[q], vv[q], pp[q], misc, fix, FCAP, lst);
                }
            };
            for (int g0 = 0; g0 < cp; g0 += DG) {
                if (g0 < P0.C) do_group(ringA, g0);
                refill(P0, cp, P1, g0 + DG, ringA);
                if (g0 + Q < P0.C) do_group(ringB, g0 + Q);
                refill(P0, cp, P1, g0 + DG + Q, ringB);
            }
        }

        // ---- end of the row: leftmost column, losers of all tables, duplicates
        if (live) {
            const int lead_out = wave_min_i32(mylead);
            const int t_cur = P0.t, E = P0.bound;
            bool redo = false;
            const int nlist = __builtin_amdgcn_readfirstlane(((lds_vint *)misc)[1]);
            redo = nlist > SLCAP;
            if (nlist != 0 && !redo) {
                // entries that lost in all tables are in none: compare them among themselves (a handful)
                for (int b = 0; b < nlist; b += 64) {
                    const int i = b + lane;
                    int4 me = make_int4(-1, 0, -1, 0);
                    if (i < nlist) me = lst[i];
                    int owner = -1;
                    for (int j = 0; j < nlist; j++) {
                        const int4 o = lst[j];
                        if (j < i && owner < 0 && o.x == me.x && o.z != me.z) owner = o.z;
                    }
                    if (owner >= 0) stream_fix_push(misc, fix, FCAP, owner, me.z, me.y);
                }
            }
            const int nfix = __builtin_amdgcn_readfirstlane(((lds_vint *)misc)[0]);
            redo = redo || nfix > FCAP;
            if (redo) {
                // too many duplicate columns for the lists: the general path takes the row (its space in S stays unused)
                if (lane == 0) a.rej_list[atomicAdd(a.rej_count, 1)] = t_cur;
                c_redo += 1;
            } else {
                if (lane < nfix) a.fixbuf[(size_t)t_cur * SFIX + lane] = fix[lane];
                if (lane == 0) {
                    a.Slen[t_cur] = E; // duplicates are merged by k_stream_fix, which corrects the length then
                    a.Slead[t_cur] = E > 0 ? lead_out : INT_MAX;
                    a.fixcnt[t_cur] = nfix;
                }
                c_nnz += (u64d)E;
                c_rows += E > 0;
                c_seg += 1 + (u64d)P0.nruns;
            }
            // reset what the row used of the tables, and the counters
            const int words = 7 << (P0.logt - 2);
            for (int sidx = lane * 4; sidx < words; sidx += 256) *(int4 *)(t1 + sidx) = make_int4(0, 0, 0, 0);
            if (lane < 2) misc[lane] = 0;
            __builtin_amdgcn_wave_barrier();
        }

        // ---- the block asked for at the top of this pass
        if (asking) {
            g_new = resolve(asked);
            if (s == -5) { // (block 2 was asked for early: the first block starts at row 0)
                if (g_new < 0) s_limit = min(s_limit, 2 * FZ_B);
            } else if (g_new < 0) s_limit = min(s_limit, s_blk0 + 2 * FZ_B);
        }
        // ---- rotate
        P0 = P1;
        I4 = I5;
        SL5 = SL6;
        T1 = T2; T2 = T3; T3 = T4; T4 = T5;
        L1 = L2; L2 = L3; L3 = L4;
        E0 = E1; E1 = E2; E2 = E3; E3 = E4;
        PB2 = PB3;
        WC1 = WC2;
        mP1 = mP2;
    }
    if (lane == 0) {
        if (c_nnz) atomicAdd(&ctr_shard(a.ctr)->nnz_out, c_nnz);
        if (c_rows) atomicAdd(&ctr_shard(a.ctr)->nonempty_out, c_rows);
        if (c_redo) atomicAdd(&ctr_shard(a.ctr)->stream_redo, c_redo);
        if (c_nnz | c_seg) {
            atomicAdd(&ctr_shard(a.ctr)->class_ent[a.cls], c_nnz);
            atomicAdd(&ctr_shard(a.ctr)->class_seg[a.cls], c_seg);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// small kernels around the fused step
// ------------------------------------------------------------------------------------------------
// per row slot: where the row's own entries are, how many, and the row of the input matrix it comes from -- one 16-byte record,
// so that a wave learns a row with one load
__global__ void k_gather_info(int n, const int *__restrict__ rows, const i64d *__restrict__ start, const int *__restrict__ len, const int *__restrict__ orig,
                              int4 *__restrict__ rinfo)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int row = rows ? rows[t] : t;
    const u64d st = (u64d)start[row];
    rinfo[t] = make_int4((int)(unsigned)st, (int)(unsigned)(st >> 32), len[row], orig[row]);
}

// what a fused step starts from: block counters, the cursor of S, the list counts, the statistics, no duplicates in any row
__global__ void k_fused_reset(int nrows, unsigned *__restrict__ work, int nwork_words, u64d *__restrict__ cursor, int *__restrict__ counts, int ncounts,
                              unsigned *__restrict__ ctr_words, int nctr_words, int *__restrict__ fixcnt)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nwork_words) work[t] = 0;
    if (t == 0) *cursor = 0;
    if (t < ncounts) counts[t] = 0;
    if (t < nctr_words) ctr_words[t] = 0;
    if (t <= nrows) fixcnt[t] = 0;
}

// the rows the general path took for the fused step, back under their slots (their entries lie behind those of the fused rows)
__global__ void k_merge_rej(int nrej, const int *__restrict__ rej, const i64d *__restrict__ fstart, const int *__restrict__ flen, const int *__restrict__ flead,
                            const int *__restrict__ forig, i64d base, i64d *__restrict__ Sstart, int *__restrict__ Slen, int *__restrict__ Slead, int *__restrict__ Sorig)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrej) return;
    const int t = rej[i];
    Sstart[t] = fstart[i] + base;
    Slen[t] = flen[i];
    Slead[t] = flead[i];
    Sorig[t] = forig[i];
}

// rows[rej[i]] for the general path
__global__ void k_rej_rows(int nrej, const int *__restrict__ rej, const int *__restrict__ rows, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrej) out[i] = rows ? rows[rej[i]] : rej[i];
}
