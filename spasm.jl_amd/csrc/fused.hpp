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

// what the kernel needs now and then (kept in device memory: the arguments it holds in registers all the time are few)
// (pointers into GLOBAL memory, said so: through a pointer that was itself loaded from memory the compiler otherwise emits flat
// instructions, and one flat access in flight makes every later wait for a load a wait for all loads)
typedef __attribute__((address_space(1))) int fz_gint;
struct FusedRare {
    fz_gint *long_list;        // slots whose stream is beyond this launch's tables but within FusedArgs::long_bound
    fz_gint *long_count;
    fz_gint *rej_list;         // slots left to the general path
    fz_gint *rej_count;
};
__device__ __forceinline__ void fz_list_append(fz_gint *list, fz_gint *count, int t)
{
    list[__hip_atomic_fetch_add(count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = t;
}

struct FusedArgs {
    int nrows;                 // row slots of the round
    const int *slots;          // LIST launches: the slots to take (a list another launch left) ..
    const int *slot_count;     // .. and how many (device)
    const int4 *rinfo;         // per slot: {start (low, high), length, originating row} of the row's own entries
    const int2 *ent;
    const unsigned *pbits;     // bit j: column j is a pivot column of this round
    const int4 *wcol;          // per pivot column: {pivot index, length of its row of W (-1: not available), offset, -}
    const int2 *buf;           // the U_PN + own + W buffer
    int2 *Sent;
    u64d scap;                 // entries Sent holds
    u64d *scursor;
    int4 *rec;                 // per slot two records: {start (low, high), length, leftmost column}, {duplicates (-1: the row was not taken), -, -, -}
    int2 *fixbuf;              // [slot][SFIX] duplicates found in the row, merged afterwards by k_fused_finish
    const FusedRare *rare;
    unsigned *work;            // FZ_NWORK block counters
    RoundCounters *ctr;
    int cls;                   // index for the per-class counters
    int free_cols;
    int long_bound;            // rows with a stream beyond this launch's tables but within this go on rare->long_list (0: no such list)
    unsigned buf_bytes, pbits_bytes, wcol_bytes; // sizes of buf / pbits / wcol (the kernel reaches them through 32-bit offsets)
    u64d *stamps;              // diagnostic build only: NSTAMP cycle sums + the wave count
    ZpField F;
};

// LDS of one wave: three tag tables of 2^LOGTMAX, half and a quarter of that, 64 B of counters, fix-up list, loser list
template <int LOGTMAX> struct FzLds {
    static constexpr int WORDS = (1 << LOGTMAX) + (1 << (LOGTMAX - 1)) + (1 << (LOGTMAX - 2));
    static constexpr size_t TABB = (size_t)4 * WORDS, MISCB = 64, FIXB = (size_t)SFIX * 8, LSTB = (size_t)SLCAP * 16, MARKB = 256;
    static constexpr size_t SLOT = TABB + MISCB + FIXB + LSTB + MARKB;
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
// an 8-byte load from a wave-uniform base + a 32-bit byte offset per lane (the form that needs no 64-bit address per lane)
__device__ __forceinline__ int2 fz_load8(const int2 *ubase, unsigned byteoff) { return *(const int2 *)((const char *)ubase + byteoff); }
// lane l of v becomes x (x, l scalar)
__device__ __forceinline__ void fz_set_lane(int &v, int x, int l)
{
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(x), "s"(l) : "m0");
}
__device__ __forceinline__ int fz_lane_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ i64d fz_lane_i64(i64d v, int l)
{
    return (i64d)(((u64d)(unsigned)__builtin_amdgcn_readlane((int)((u64d)v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)v, l));
}

// the plan of a row (all wave-uniform).  Its chunks -- at most 64 consecutive entries of a run of W each -- sit in a table of 64
// lanes shared by the rows in flight: chunk j of the row in lane (cb + j) & 63, padded with empty chunks to whole double groups.
constexpr int FZ_MAXC = 32;        // chunks a row of the wave kernel may have: the double groups of two rows fit the 64 lanes
struct FzPlan {
    int t;          // row slot; < 0: nothing to do (no row, or the row went on a list)
    int nN;         // own entries on non-pivot columns (the front of the stream)
    int bound;      // length of the stream
    int logt;
    int C;          // chunks
    int cp;         // .. padded to whole double groups (at least one)
    int cb;         // lane of its first chunk
    int nr;         // runs
    u64d mN;        // lanes whose own entry sits on a non-pivot column
    i64d sbase;     // where the row starts in S
};

typedef int fz_v2i __attribute__((ext_vector_type(2)));
typedef int fz_v4i __attribute__((ext_vector_type(4)));
constexpr int FZ_RSRC = 0x00020000; // dword 3 of a raw buffer resource (gfx9 family)
// a raw buffer over `bytes` bytes at p: accesses are base + a 32-bit scalar offset + a 32-bit lane offset, and what falls outside is
// dropped (stores) / read as zero (loads)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fz_rsrc(const void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, FZ_RSRC); }

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
    const int lane8 = lane << 3;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char *base = s_raw + (size_t)wave * L::SLOT;
    unsigned *t1 = (unsigned *)base;
    int *misc = (int *)(base + L::TABB);
    int2 *fix = (int2 *)(base + L::TABB + L::MISCB);
    int4 *lst = (int4 *)(base + L::TABB + L::MISCB + L::FIXB);
    int *mark = (int *)(base + L::TABB + L::MISCB + L::FIXB + L::LSTB); // 64 words: which run starts at a chunk
    const ZpField F = a.F;
    const int N = LIST ? *a.slot_count : a.nrows;
    const int nblk = (N + FZ_B - 1) / FZ_B;
    const __amdgpu_buffer_rsrc_t r_buf = fz_rsrc(a.buf, a.buf_bytes), r_info = fz_rsrc(a.rinfo, (unsigned)a.nrows * 16u + 16u),
                                 r_pb = fz_rsrc(a.pbits, a.pbits_bytes), r_wc = fz_rsrc(a.wcol, a.wcol_bytes),
                                 r_rec = fz_rsrc(a.rec, (unsigned)a.nrows * 32u + 32u);

    for (int s = lane * 4; s < L::WORDS; s += 256) *(int4 *)(t1 + s) = make_int4(0, 0, 0, 0);
    if (lane < 16) misc[lane] = 0;
    mark[lane] = -1;
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

    // ---- rows this launch does not take: their slots are collected 64 at a time (a lane each) and go on their list with ONE
    // atomic per 64 -- a round leaves up to a third of its rows, and that many returning atomics on one word would take
    // milliseconds (11 ns each, one after the other)
    int rejq = 0, longq = 0, nrejq = 0, nlongq = 0;
    auto flush = [&](int &q, int &n, fz_gint *list, fz_gint *count) {
        if (n == 0) return;
        int at = 0;
        if (lane == 0) at = __hip_atomic_fetch_add(count, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        at = fz_sgpr(at);
        if (lane < n) list[at + lane] = q;
        n = 0;
    };
    auto push_rej = [&](int t) {
        fz_set_lane(rejq, t, nrejq);
        if (++nrejq == 64) flush(rejq, nrejq, a.rare->rej_list, a.rare->rej_count);
    };
    auto push_long = [&](int t) {
        fz_set_lane(longq, t, nlongq);
        if (++nlongq == 64) flush(longq, nlongq, a.rare->long_list, a.rare->long_count);
    };

    // ---- the chunk table (a lane per chunk): byte offset of the chunk in buf; byte offset of its first entry in the row's stream;
    // (stream position + 1) << 8 | entries (0: an empty chunk of the padding); the multiplier of its run
    int T_ld = 0, T_st = 0, T_pn = 0, T_mul = 0;

    // ---- the plan of a row: slot t, ln own entries e, the lanes mP on pivot columns with {length, offset} wc of their rows of W;
    // its chunks go to the lanes from cb on
    auto make_plan = [&](int t, int ln, const int2 &e, u64d mP, const fz_v2i &wc, int cb) -> FzPlan {
        FzPlan pl;
        pl.t = -1;
        pl.nN = 0; pl.bound = 0; pl.logt = FZ_MINLOGT; pl.C = 0; pl.cp = DG; pl.cb = cb & 63; pl.nr = 0; pl.mN = 0; pl.sbase = 0;
        const int jl = (lane - cb) & 63; // the chunk this lane would hold
        bool ok = t >= 0;
        int nN = 0, bound = 0, ctot = 0, nch = 0, incl = 0, cincl = 0, wl = 0;
        u64d mN = 0;
        if (ok) {
            const bool valid = lane < min(ln, 64);
            const bool isP = (mP >> lane) & 1ull;
            wl = isP ? wc.x : 0;
            // what this kernel does not take: more than 64 own entries, a zero among them, a row of W that could not be built
            const bool bad = ln > 64 || __ballot((valid && e.y == 0) || wl < 0) != 0;
            wl = max(wl, 0);
            nch = (wl + 63) >> 6;
            int tot = 0;
            incl = team_incl_scan<64>(wl, tot);
            cincl = team_incl_scan<64>(nch, ctot);
            mN = __ballot(valid && !isP);
            nN = __popcll(mN);
            const i64d b64 = (i64d)nN + (i64d)tot;
            bool tolong = false, rej = bad || b64 > (i64d)a.free_cols;
            if (!rej && (b64 > (i64d)fz_cap(LOGTMAX) || ctot > FZ_MAXC)) {
                if (b64 <= (i64d)a.long_bound) tolong = true;
                else rej = true;
            }
            i64d sb = 0;
            if (!rej && !tolong) {
                sb = take_s((int)b64);
                if (sb < 0) rej = true;
            }
            if (rej || tolong) {
                if (tolong) push_long(t);
                else push_rej(t);
                ok = false;
            } else {
                bound = (int)b64;
                pl.sbase = sb;
            }
        }
        if (!ok) {
            // nothing to do for this row: one double group of empty chunks (it carries the requests of the row behind it)
            if (jl < DG) { T_ld = 0; T_pn = 0; }
            return pl;
        }
        int logt = FZ_MINLOGT;
        while (fz_cap(logt) < bound) logt++;
        pl.t = t;
        pl.nN = nN;
        pl.bound = bound;
        pl.logt = logt;
        pl.C = ctot;
        pl.cp = max(DG, (ctot + DG - 1) & ~(DG - 1));
        pl.nr = __popcll(mP);
        pl.mN = mN;
        // the chunks.  Every run names itself at its first chunk (LDS), a prefix maximum tells every chunk its run, and the run's
        // {offset, length, stream position, multiplier} come over by lane permutation.
        const int cb0 = cincl - nch, rpos = nN + incl - wl;
        if (nch > 0) mark[cb0] = lane;
        __builtin_amdgcn_wave_barrier();
        int run = mark[jl]; // (-1 where no run starts; chunk lanes beyond the row read marks of the padding: never set)
        __builtin_amdgcn_wave_barrier();
        if (nch > 0) mark[cb0] = -1;
        if (jl >= ctot) run = -1;
        // inclusive prefix maximum over jl = 0 .. 63: the lanes are rotated by cb, so go through the lane order of jl -- a rotation
        // brings chunk jl to lane jl, the scan runs there, and the result is rotated back
        {
            const int src = (lane + cb) & 63; // lane holding chunk `lane`
            int r = __builtin_amdgcn_ds_bpermute(src << 2, run);
            r = max(r, __builtin_amdgcn_update_dpp(-1, r, 0x111, 0xf, 0xf, false));
            r = max(r, __builtin_amdgcn_update_dpp(-1, r, 0x112, 0xf, 0xf, false));
            r = max(r, __builtin_amdgcn_update_dpp(-1, r, 0x114, 0xf, 0xf, false));
            r = max(r, __builtin_amdgcn_update_dpp(-1, r, 0x118, 0xf, 0xf, false));
            r = max(r, __builtin_amdgcn_update_dpp(-1, r, 0x142, 0xa, 0xf, false));
            r = max(r, __builtin_amdgcn_update_dpp(-1, r, 0x143, 0xc, 0xf, false));
            run = __builtin_amdgcn_ds_bpermute(jl << 2, r);
        }
        {
            const int ra = max(run, 0) << 2;
            const int r_cb = __builtin_amdgcn_ds_bpermute(ra, cb0), r_wl = __builtin_amdgcn_ds_bpermute(ra, wl), r_wo = __builtin_amdgcn_ds_bpermute(ra, wc.y),
                      r_rp = __builtin_amdgcn_ds_bpermute(ra, rpos), r_mu = __builtin_amdgcn_ds_bpermute(ra, e.y);
            if (jl < pl.cp) {
                const int jj = jl - r_cb;
                const bool real = run >= 0 && jl < ctot; // (the prefix maximum ran on into the padding)
                T_ld = real ? (r_wo + 64 * jj) << 3 : 0;
                T_st = real ? (r_rp + 64 * jj) << 3 : 0;
                T_pn = real ? ((r_rp + 64 * jj + 1) << 8) | min(64, r_wl - 64 * jj) : 0;
                T_mul = r_mu;
            }
        }
        return pl;
    };

    // ---- chunk loads.  The chunks of the rows form ONE stream: every row takes a whole number of double groups (at least one),
    // group A of a double group lives in ringA, group B in ringB, and while double group d is worked on the loads of double group
    // d + 1 are issued -- of the same row, or the first one of the next row (whose plan is made before the current row starts, and
    // whose chunks sit behind those of the current row in the table).  Always Q loads per request (an empty chunk reads the first
    // entries of the buffer): a number of loads that depends on the row would make every wait for an older load a wait for all.
    fz_v2i ringA[Q], ringB[Q];
    auto refill = [&](int c0, fz_v2i (&r)[Q]) {
#pragma unroll
        for (int q = 0; q < Q; q++) r[q] = __builtin_amdgcn_raw_buffer_load_b64(r_buf, lane8, fz_lane_i32(T_ld, (c0 + q) & 63), 0);
    };

    // ---- the stages.  I: {start, length, originating row} of a row, T its slot (< 0: no row); E: its entries, a lane each;
    // PB: the word of pbits of every entry; WC: {length, offset} of the row of W of every entry on a pivot column
    fz_v4i I5 = {0, 0, 0, 0}, I4 = I5;
    int T5 = -1, T4 = -1, T3 = -1, T2 = -1, T1 = -1; // slots (scalar)
    int SL6 = 0, SL5 = 0;                            // LIST: the word of the list of row s + 6 / s + 5
    int L3 = 0, L2 = 0, L1 = 0;                      // lengths (scalar)
    int2 E4 = make_int2(0, 0), E3 = E4, E2 = E4, E1 = E4, E0 = E4;
    unsigned PB3 = 0, PB2 = 0;
    fz_v2i WC2 = {0, 0}, WC1 = WC2;
    u64d mP2 = 0, mP1 = 0;
    FzPlan P0, P1;
    P0.t = -1; P0.nN = 0; P0.bound = 0; P0.logt = FZ_MINLOGT; P0.C = 0; P0.cp = DG; P0.cb = 0; P0.nr = 0; P0.mN = 0; P0.sbase = 0;
    u64d c_nnz = 0, c_seg = 0;
    int c_rows = 0, c_redo = 0;
    int s_blk0 = 0; // first row number of the current block

#ifdef SPASM_STAMPS
    u64d st_sum[NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64d st_last = stamp_now();
#endif
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
            I5 = __builtin_amdgcn_raw_buffer_load_b128(r_info, 0, max(T5, 0) << 4, 0);
        }
        int L4 = 0;
        {
            const u64d st4 = ((u64d)(unsigned)fz_sgpr(I4.y) << 32) | (unsigned)fz_sgpr(I4.x);
            L4 = T4 < 0 ? 0 : fz_sgpr(I4.z);
            E4 = fz_load8(a.ent + (T4 < 0 ? 0 : st4), (unsigned)min(lane, max(min(L4, 64) - 1, 0)) << 3);
        }
        PB3 = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r_pb, (int)(((unsigned)E3.x >> 5) << 2), 0, 0);
        {
            const bool p2 = lane < min(L2, 64) && ((PB2 >> (E2.x & 31)) & 1u);
            mP2 = __ballot(p2);
            WC2 = __builtin_amdgcn_raw_buffer_load_b64(r_wc, p2 ? (E2.x << 4) : 0, 4, 0); // {length, offset} of the column's record
        }
        STAMP(0); // blocks, stage loads
        // ---- the plan of row s + 1; its chunks go behind those of the current row
        P1 = make_plan(T1, L1, E1, mP1, WC1, P0.cb + P0.cp);

        STAMP(1); // plan
        // ---- the current row
        const bool live = P0.t >= 0;
        FzTab tb;
        tb.logt = P0.logt;
        tb.t[0] = t1;
        tb.t[1] = t1 + (1 << P0.logt);
        tb.t[2] = t1 + (1 << P0.logt) + (1 << (P0.logt - 1));
        const __amdgpu_buffer_rsrc_t r_row = fz_rsrc(a.Sent + P0.sbase, (unsigned)P0.bound << 3);
        int mylead = INT_MAX;
        if (live) {
            // own entries on non-pivot columns: the front of the stream
            const bool isN = (P0.mN >> lane) & 1ull;
            if (isN) {
                const int pos = __popcll(P0.mN & lanemask_lt());
                const fz_v2i ev = {E0.x, E0.y};
                __builtin_amdgcn_raw_buffer_store_b64(ev, r_row, pos << 3, 0, 2);
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
        STAMP(2); // own entries
        // the runs, a double group at a time (a row without chunks still takes one: it carries the requests of the next row)
        {
            auto do_group = [&](const fz_v2i (&ring)[Q], int c0) {
                int cc[Q], vv[Q], pp[Q], st[Q];
                bool act[Q];
                unsigned oo[Q], left[Q], w1[Q];
                unsigned *s1[Q];
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const int g = (c0 + q) & 63;
                    const int pn = fz_lane_i32(T_pn, g), mul = fz_lane_i32(T_mul, g);
                    st[q] = fz_lane_i32(T_st, g);
                    act[q] = lane < (pn & 0xff);
                    cc[q] = ring[q].x;
                    vv[q] = stream_mul<SMALL>(F, mul, ring[q].y);
                    pp[q] = (int)((unsigned)pn >> 8) + lane;
                    w1[q] = fz_want<0>(tb, cc[q], pp[q], s1[q]);
                    oo[q] = 0;
                }
                // store + first-table CAS of the Q chunks, all in flight together
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    if (act[q]) {
                        const fz_v2i ev = {cc[q], vv[q]};
                        __builtin_amdgcn_raw_buffer_store_b64(ev, r_row, lane8, st[q], 2);
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
            // (a group that is all padding still "uses" its loads: the registers are about to be loaded again, and the compiler
            // then waits for exactly these loads instead of for everything in flight)
            auto touch = [&](const fz_v2i (&ring)[Q]) {
#pragma unroll
                for (int q = 0; q < Q; q++) asm volatile("" ::"v"(ring[q].x), "v"(ring[q].y));
            };
            for (int g0 = 0; g0 < P0.cp; g0 += DG) {
                const int c0 = P0.cb + g0;
                if (g0 < P0.C) do_group(ringA, c0);
                else touch(ringA);
                refill(c0 + DG, ringA);
                if (g0 + Q < P0.C) do_group(ringB, c0 + Q);
                else touch(ringB);
                refill(c0 + DG + Q, ringB);
            }
        }

        STAMP(3); // groups + requests
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
                push_rej(t_cur);
                c_redo += 1;
            } else {
                if (lane < nfix) a.fixbuf[(size_t)t_cur * SFIX + lane] = fix[lane];
                // the row's record: {where it starts (low, high), its length (duplicates still in), leftmost column}, {duplicates}
                if (lane == 0) {
                    const fz_v4i r0 = {(int)(unsigned)(u64d)P0.sbase, (int)(unsigned)((u64d)P0.sbase >> 32), E, E > 0 ? lead_out : INT_MAX};
                    __builtin_amdgcn_raw_buffer_store_b128(r0, r_rec, 0, t_cur << 5, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(nfix, r_rec, 16, t_cur << 5, 0);
                }
                c_nnz += (u64d)E;
                c_rows += E > 0;
                c_seg += 1 + (u64d)P0.nr;
            }
            // reset what the row used of the tables, and the counters
            const int words = 7 << (P0.logt - 2);
            for (int sidx = lane * 4; sidx < words; sidx += 256) *(int4 *)(t1 + sidx) = make_int4(0, 0, 0, 0);
            if (lane < 2) misc[lane] = 0;
            __builtin_amdgcn_wave_barrier();
        }

        STAMP(4); // end of the row
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
        STAMP(5); // block answer, rotation (waits for the stage loads)
    }
#ifdef SPASM_STAMPS
    if (lane == 0 && a.stamps) {
        for (int i = 0; i < NSTAMP; i++) atomicAdd(&a.stamps[i], st_sum[i]);
        atomicAdd(&a.stamps[NSTAMP], 1ull);
    }
#endif
    flush(rejq, nrejq, a.rare->rej_list, a.rare->rej_count);
    flush(longq, nlongq, a.rare->long_list, a.rare->long_count);
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

// what a fused step starts from: block counters, the cursor of S, the list counts, the statistics, no row taken yet
__global__ void k_fused_reset(int nrows, unsigned *__restrict__ work, int nwork_words, u64d *__restrict__ cursor, int *__restrict__ counts, int ncounts,
                              unsigned *__restrict__ ctr_words, int nctr_words, int4 *__restrict__ rec)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nwork_words) work[t] = 0;
    if (t == 0) { cursor[0] = 0; cursor[1] = 0; }
    if (t < ncounts) counts[t] = 0;
    if (t < nctr_words) ctr_words[t] = 0;
    if (t < nrows) rec[2 * (size_t)t + 1] = make_int4(-1, 0, 0, 0);
}

// Behind the fused kernels: the duplicates they found are merged row by row (stream_fixup of stream.hpp: a wave looks at 16 row
// slots and works through those that have any), and the rows' records become the arrays the rest of the engine reads
// (start / length / leftmost column / originating row per slot).  Slots no fused kernel took are left alone: the general path
// fills them in.
struct FusedFinishArgs {
    int nrows;
    const int4 *rec;
    const int4 *rinfo;
    const int2 *fixbuf;
    int2 *Sent;
    i64d *Sstart;
    int *Slen;
    int *Slead;
    int *Sorig;
    RoundCounters *ctr;
    ZpField F;
};
__global__ __launch_bounds__(256) void k_fused_finish(FusedFinishArgs a)
{
    __shared__ int s_scratch[4][8 * SFIX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t0 = (blockIdx.x * 4 + wave) * 16;
    if (t0 >= a.nrows) return;
    const bool mine = lane < 16 && t0 + lane < a.nrows;
    int4 r0 = make_int4(0, 0, 0, INT_MAX);
    int nfix = -1;
    if (mine) {
        nfix = a.rec[2 * (size_t)(t0 + lane) + 1].x;
        if (nfix >= 0) r0 = a.rec[2 * (size_t)(t0 + lane)];
    }
    u64d m = __ballot(nfix > 0);
    u64d holes = 0;
    int emptied = 0, merged = 0;
    while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int t = t0 + l;
        const int n = __builtin_amdgcn_readlane(nfix, l);
        const int E = __builtin_amdgcn_readlane(r0.z, l);
        int lead = __builtin_amdgcn_readlane(r0.w, l);
        const i64d st = (i64d)(((u64d)(unsigned)__builtin_amdgcn_readlane(r0.y, l) << 32) | (unsigned)__builtin_amdgcn_readlane(r0.x, l));
        const int n_out = stream_fixup<SFIX>(a.F, a.fixbuf + (size_t)t * SFIX, n, s_scratch[wave], (u64d *)(a.Sent + st), E, &lead);
        if (lane == l) {
            r0.z = n_out;
            r0.w = n_out > 0 ? lead : INT_MAX;
        }
        holes += (u64d)(E - n_out);
        emptied += (n_out == 0 && E > 0);
        merged += n;
    }
    if (mine && nfix >= 0) {
        const int t = t0 + lane;
        a.Sstart[t] = (i64d)(((u64d)(unsigned)r0.y << 32) | (unsigned)r0.x);
        a.Slen[t] = r0.z;
        a.Slead[t] = r0.w;
        a.Sorig[t] = a.rinfo[t].w;
    }
    if (lane == 0 && merged) {
        atomicAdd(&ctr_shard(a.ctr)->nnz_out, (u64d)0 - holes); // (mod 2^64: the fused kernels counted the stream lengths)
        if (emptied) atomicAdd(&ctr_shard(a.ctr)->nonempty_out, -emptied);
        atomicAdd(&ctr_shard(a.ctr)->stream_fix, merged);
    }
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

// ------------------------------------------------------------------------------------------------
// Which rows the fused kernel takes, decided before it starts: a team of 16 lanes reads a row's entries, their pivot bits and the
// lengths of the rows of W they name -- the length of the row's stream and its chunks follow -- and the slot goes on the list of
// the fused kernel or on the list of the general path.  With the lists known up front the two paths run side by side (the general
// path's rows are few and long: behind the fused kernel they would be a tail of their own), and the fused kernel meets no row it
// has to turn away.
// ------------------------------------------------------------------------------------------------
struct ClassifyArgs {
    int nrows;
    const int4 *rinfo;
    const int2 *ent;
    const unsigned *pbits;
    const int4 *wcol;
    int *flag;                 // out, per slot (+ one 0 behind the last): 1 = the fused kernel takes the row
    u64d *general_bound;       // sum of the stream lengths of the general rows (those it could work out: rows of up to 64 entries)
    int cap;                   // longest stream the fused kernel takes
    int free_cols;
};
__global__ __launch_bounds__(256) void k_fused_classify(ClassifyArgs a)
{
    constexpr int TEAM = 16;
    __shared__ u64d s_gb;
    if (threadIdx.x == 0) s_gb = 0;
    __syncthreads();
    const int tl = threadIdx.x % TEAM;
    const int t = (int)((blockIdx.x * blockDim.x + threadIdx.x) / TEAM);
    if (t < a.nrows) {
        const int4 inf = a.rinfo[t];
        const i64d st = (i64d)(((u64d)(unsigned)inf.y << 32) | (unsigned)inf.x);
        const int ln = inf.z;
        long long tot = 0;
        int C = 0, nN = 0;
        bool bad = ln > 64;
        for (int k0 = 0; k0 < ln && !bad; k0 += TEAM) {
            const int k = k0 + tl;
            const bool valid = k < ln;
            int2 e = make_int2(0, 1);
            if (valid) e = a.ent[st + k];
            const bool isP = valid && ((a.pbits[(unsigned)e.x >> 5] >> (e.x & 31)) & 1u);
            int wl = 0;
            if (isP) wl = a.wcol[e.x].y;
            bad |= team_ballot<TEAM>((valid && e.y == 0) || wl < 0) != 0;
            int s1, s2;
            (void)team_incl_scan<TEAM>(max(wl, 0), s1);
            (void)team_incl_scan<TEAM>((max(wl, 0) + 63) >> 6, s2);
            tot += s1;
            C += s2;
            nN += __popcll(team_ballot<TEAM>(valid && !isP));
        }
        const long long bound = (long long)nN + tot;
        const bool fused = !bad && bound <= (long long)a.cap && bound <= (long long)a.free_cols && C <= FZ_MAXC;
        if (tl == 0) {
            a.flag[t] = fused ? 1 : 0;
            if (!fused) atomicAdd(&s_gb, (u64d)bound);
        }
    } else if (t == a.nrows && tl == 0) a.flag[t] = 0;
    __syncthreads();
    if (threadIdx.x == 0 && s_gb) atomicAdd(a.general_bound, s_gb);
}
// the two lists, in slot order, from the flags and their exclusive prefix sums (pos[nrows] = rows of the fused kernel)
__global__ void k_fused_lists(int nrows, const int *__restrict__ flag, const int *__restrict__ pos, int *__restrict__ fused_list, int *__restrict__ general_list,
                              int *__restrict__ counts)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nrows) {
        if (flag[t]) fused_list[pos[t]] = t;
        else general_list[t - pos[t]] = t;
    }
    if (t == nrows) { counts[2] = pos[nrows]; counts[3] = nrows - pos[nrows]; }
}
