// stream.hpp -- SCATTER without accumulators: the Schur row is written while its entries stream by.
//
// Same contract as k_scatter (kernels.hpp): x_a = B[k]_N - x_b * U_PN, the non-pivot part of the solution of
// x * U = B[k] (reference src/SpaSM.jl:694-713), trip count of the reference's scatter loop (:619-620).
//
// On sparse matrices nearly every entry of a Schur row lands on a column of its own: with ~600 entries thrown on ~10^6
// columns a row sees 0.2 collisions.  The hash-table kernel pays for the general case on every entry (CAS + atomic add in
// LDS, probing, then a sweep over all slots that reduces, compacts and stores).  Here an entry's value IS its final value
// unless a second entry shows up on the same column, so:
//   * every entry of the row's stream (its own entries on non-pivot columns, then the non-pivot parts of the applied pivot
//     rows, in record order) has a fixed POSITION known before anything is loaded: the combine kernel stores the running
//     total of npn in each multiplier record;
//   * the entry is multiplied, reduced to its balanced residue and stored straight to S[row][position] from registers;
//   * LDS only detects duplicates, with DIRECT-MAPPED tables of 32-bit words and no probing: x = column * K mod 2^24 is a
//     bijection of the column (K odd, columns below 2^24), the slot is its top bits, and the word stored is
//     (x << 14) | (position + 1) -- the slot index and the 18 low bits of x in the word identify the column exactly.  One
//     CAS per entry; an entry that finds ANOTHER column in its slot (the table is at most 5/16 full) gets one more CAS in a
//     second table a quarter of the size under another K; what loses there too (about 1 %) goes to a short list that is
//     compared pairwise at the end of the row.  No loops, no retry lists;
//   * an entry that finds its own column under another position is a duplicate: {owner's position, own position, value}
//     goes to the row's fix-up list.  After the last entry the (rare) fix-ups are applied in global memory: the value is
//     added to the owner's entry with a 64-bit compare-and-swap, the duplicate's position becomes a hole, and holes are
//     filled with the entries at the end of the row (the order of a row's entries carries no meaning, reference
//     src/SpaSM.jl:1017-1020);
//   * no per-entry predicates: a lane past the end of its pivot row holds a copy of the row's last entry (clamped index) and
//     repeats that entry's store and CAS -- same bytes to the same address, and a CAS that finds the very word it wanted to
//     write counts as "in";
//   * the loads of a row's pivot-row entries are issued one row ahead (its records two rows ahead, its descriptor three).
// Rows with more duplicates than the fix-up list holds (structured matrices) are handed back to the hash-table kernel
// of the same size class through its row list.
#pragma once
#include "kernels.hpp"

struct StreamArgs {
    const int *class_count;    // rows in this class
    const RowDesc *desc;       // their descriptors
    const int2 *ent;
    const int *qinv_r;
    const int2 *UPN;
    const int4 *Lpool;         // {stream position << 16 | npn, multiplier, offset in UPN, npn}
    int2 *Sent;
    int *Slen;
    int *Slead;
    RoundCounters *ctr;
    int cls;                   // index of this class for the per-class counters (NSTREAM0 + size class)
    int *redo_count;           // the hash-table class of the same size: rows this kernel gives up on are appended there
    RowDesc *redo_desc;
    int2 *fixbuf;              // [row slot][SFIX] duplicates found in the row, merged afterwards by k_stream_fix
    int *fixcnt;               // [row slot] how many
    u64d *stamps;              // diagnostic build only: [class][NSTAMP] cycle sums + [class][NSTAMP] wave counts
    int dbg;                   // TIMING ABLATIONS ONLY (diagnostic builds, env SPASM_DBG; results are wrong when non-zero):
                               // 1 = no Schur stores, 2 = no duplicate check (no LDS traffic), 4 = fix-ups ignored, 8 = no pivot-row loads
    ZpField F;
};

constexpr int SLCAP = 64; // entries of a row's list of second-table losers
constexpr int SFIX = 16;  // duplicates a row may collect before it is handed to the hash-table kernel (a row of 2560 random columns
                          // out of 10^6 expects 3)

// LDS of one row: first table (4 B x 2^logt), second table (a quarter), 64 B of counters, fix-up list, loser list
__host__ __device__ constexpr size_t stream_row_bytes(int logt) { return ((size_t)5 << logt) + 64 + (size_t)SFIX * 8 + (size_t)SLCAP * 16; }
__host__ __device__ constexpr size_t stream_lds_bytes(int logt, int tpr, int wpb) { return tpr == 64 ? stream_row_bytes(logt) * (size_t)wpb : stream_row_bytes(logt); }

// the row's fix-up list: {owner position << 14 | own position, value}; positions are below 2^14
__device__ __forceinline__ void stream_fix_push(int *s_nfix, int2 *fix, int fcap, int owner_pos, int pos, int v)
{
    const int i = atomicAdd(s_nfix, 1);
    if (i < fcap) fix[i] = make_int2((int)(((unsigned)owner_pos << 14) | (unsigned)pos), v);
}

constexpr unsigned STREAM_K1 = 0x9E3779u, STREAM_K2 = 0x85EBCBu; // odd: c -> c * K mod 2^24 is a bijection

// multiplier * entry as THE balanced residue (it is stored as it stands)
template <bool SMALL> __device__ __forceinline__ int stream_mul(const ZpField &F, int nm, int y);
template <> __device__ __forceinline__ int stream_mul<true>(const ZpField &F, int nm, int y) { return acc_reduce_short<true>(F, __mul24(nm, y)); }
template <> __device__ __forceinline__ int stream_mul<false>(const ZpField &F, int nm, int y) { return zp_mul(F, nm, y); }

__device__ __forceinline__ u64d stream_load_fresh(const u64d *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ------------------------------------------------------------------------------------------------
// Merging the duplicates of a row, by ONE wave, after the fact (k_stream_fix below: a kernel of its own, so that the streaming
// kernel never waits for its stores): the duplicate's value is added to the owner's entry, then the holes are filled from
// the end of the row.  `scratch`: 4 arrays of 2 * FCAP ints in LDS.  Returns the row's length; `lead` is recomputed when an
// entry cancelled to zero (its column leaves the row).
// ------------------------------------------------------------------------------------------------
template <int FCAP>
__device__ __forceinline__ int stream_fixup(const ZpField F, const int2 *fix, int nfix, int *scratch, u64d *row, int E, int *lead)
{
    constexpr int HC = 2 * FCAP;
    const int lane = threadIdx.x & 63;
    int *holes = scratch, *tailflag = scratch + HC, *lowh = scratch + 2 * HC, *livet = scratch + 3 * HC;
    // (A) owner += value, by compare-and-swap on the 8-byte entry (several duplicates of one column retry each other)
    for (int b = 0; b < nfix; b += 64) {
        const int f = b + lane;
        if (f < nfix) {
            const int2 e = fix[f];
            const int po = (int)((unsigned)e.x >> 14);
            holes[f] = e.x & 0x3fff;
            u64d *p = row + po;
            u64d cur = stream_load_fresh(p);
            for (;;) {
                const int nv = zp_add(F, (int)(cur >> 32), e.y);
                const u64d want = ((u64d)(unsigned)nv << 32) | (cur & 0xffffffffull);
                const u64d old = atomicCAS(p, cur, want);
                if (old == cur) break;
                cur = old;
            }
        }
    }
    int nh = nfix, nz = 0;
    // (B) owners that cancelled to zero are holes too (listed once)
    for (int b = 0; b < nfix; b += 64) {
        const int f = b + lane;
        int po = -1;
        bool z = false;
        if (f < nfix) {
            po = (int)((unsigned)fix[f].x >> 14);
            z = (int)(stream_load_fresh(row + po) >> 32) == 0;
        }
        u64d mz = __ballot(z);
        while (mz) {
            const int l = __ffsll((long long)mz) - 1;
            mz &= mz - 1;
            const int pz = __shfl(po, l);
            bool dup = false;
            for (int i = nfix + lane; i < nh; i += 64) dup |= holes[i] == pz;
            if (__ballot(dup) == 0) {
                if (lane == 0) holes[nh] = pz;
                nh++;
                nz++;
            }
        }
    }
    const int n_out = E - nh;
    // (C) the last nh positions of the stream: those that are not holes move into the holes below n_out
    for (int i = lane; i < nh; i += 64) tailflag[i] = 0;
    for (int i = lane; i < nh; i += 64) {
        const int hp = holes[i];
        if (hp >= n_out) tailflag[hp - n_out] = 1;
    }
    int nlow = 0, nlive = 0;
    for (int b = 0; b < nh; b += 64) {
        const int i = b + lane;
        const bool in = i < nh;
        const int hp = in ? holes[i] : INT_MAX;
        const bool low = in && hp < n_out;
        const u64d ml = __ballot(low);
        if (low) lowh[nlow + __popcll(ml & lanemask_lt())] = hp;
        nlow += __popcll(ml);
        const bool live = in && tailflag[i] == 0;
        const u64d mv = __ballot(live);
        if (live) livet[nlive + __popcll(mv & lanemask_lt())] = n_out + i;
        nlive += __popcll(mv);
    }
    for (int i = lane; i < nlow; i += 64) row[lowh[i]] = stream_load_fresh(row + livet[i]); // nlow == nlive
    if (nz > 0) {
        // a column left the row: the leftmost column is whatever the final row says
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int mn = INT_MAX;
        for (int i = lane; i < n_out; i += 64) mn = min(mn, (int)(unsigned)stream_load_fresh(row + i));
        *lead = wave_min_i32(mn);
    }
    return n_out;
}


// A row descriptor (12 dwords) fetched as ONE dword per lane and unpacked with v_readlane when it is needed: a scalar load
// would sit on lgkmcnt, which every LDS wait of the kernel drains (its whole miss latency was exposed, measured), and twelve
// registers per descriptor in flight is what the k_scatter form costs.
__device__ __forceinline__ int stream_desc_load(const RowDesc *p) { return ((const int *)p)[(threadIdx.x & 63) % 12]; }
__device__ __forceinline__ RowDesc stream_desc_unpack(int v)
{
    RowDesc d;
    const unsigned w0 = __builtin_amdgcn_readlane(v, 0), w1 = __builtin_amdgcn_readlane(v, 1), w2 = __builtin_amdgcn_readlane(v, 2),
                   w3 = __builtin_amdgcn_readlane(v, 3), w4 = __builtin_amdgcn_readlane(v, 4), w5 = __builtin_amdgcn_readlane(v, 5),
                   w10 = __builtin_amdgcn_readlane(v, 10), w11 = __builtin_amdgcn_readlane(v, 11);
    d.ent_start = (i64d)(((u64d)w1 << 32) | w0);
    d.l_start = (i64d)(((u64d)w3 << 32) | w2);
    d.s_start = (i64d)(((u64d)w5 << 32) | w4);
    d.len = __builtin_amdgcn_readlane(v, 6);
    d.llen = __builtin_amdgcn_readlane(v, 7);
    d.t = __builtin_amdgcn_readlane(v, 8);
    d.bound = __builtin_amdgcn_readlane(v, 9);
    d.pmask = (long long)(((u64d)w11 << 32) | w10);
    return d;
}

// c * K mod 2^32 on the full-rate 24-bit multiplier.  As inline assembly because the compiler folds the shift that follows
// ((x << 14) | pos) into the constant and then needs v_mul_lo_u32 (quarter rate) for the 38-bit K << 14.
__device__ __forceinline__ unsigned stream_mul24(int c, unsigned k)
{
    unsigned x;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(x) : "v"(c), "s"(k));
    return x;
}

// the two direct-mapped tables of a row
template <int LOGT> struct StreamTabs {
    static constexpr int L2 = LOGT - 2;
    unsigned *t1, *t2;
    __device__ __forceinline__ void bind(unsigned char *p) { t1 = (unsigned *)p; t2 = t1 + (1 << LOGT); }
    // first table: the word wanted for (column, position + 1) and its slot
    __device__ __forceinline__ unsigned want1(int c, int pos1, unsigned &slot) const
    {
        const unsigned x = stream_mul24(c, STREAM_K1);
        slot = __builtin_amdgcn_ubfe(x, 24 - LOGT, LOGT);
        return (x << 14) | (unsigned)pos1;
    }
    __device__ __forceinline__ unsigned want2(int c, int pos1, unsigned &slot) const
    {
        const unsigned x = stream_mul24(c, STREAM_K2);
        slot = __builtin_amdgcn_ubfe(x, 24 - L2, L2);
        return (x << 14) | (unsigned)pos1;
    }
};

// Insertion of N entries of a lane: all first-table CAS are in flight at once, then the second-table CAS of those that met
// another column -- two LDS round trips per batch instead of 2 N.  Outcome per entry, as ONE word and without branches:
//   t = 0 if the CAS found the slot empty, else (word found) ^ (word wanted):  t = 0  the entry is in (new, or it met itself: a
//   lane past the end of a pivot row repeats its last entry);  0 < t < 2^14  same column under another position: a duplicate;
//   t >= 2^14  another column.
// left[j] = 0 when entry j needs nothing more; otherwise the rare side decodes it with stream_outcome().
template <int LOGT, int N>
__device__ __forceinline__ void stream_insert_n(const StreamTabs<LOGT> &tb, const int (&c)[N], const int (&pos1)[N], unsigned (&old)[N], unsigned (&left)[N])
{
    unsigned w1[N], s1[N];
#pragma unroll
    for (int j = 0; j < N; j++) w1[j] = tb.want1(c[j], pos1[j], s1[j]);
#pragma unroll
    for (int j = 0; j < N; j++) old[j] = atomicCAS(&tb.t1[s1[j]], 0u, w1[j]);
    unsigned t1[N];
#pragma unroll
    for (int j = 0; j < N; j++) t1[j] = old[j] == 0 ? 0u : old[j] ^ w1[j];
#pragma unroll
    for (int j = 0; j < N; j++) {
        left[j] = t1[j];
        if (t1[j] >= 0x4000u) { // another column in the slot: second table
            unsigned s2;
            const unsigned w2 = tb.want2(c[j], pos1[j], s2);
            const unsigned o2 = atomicCAS(&tb.t2[s2], 0u, w2);
            old[j] = o2;
            left[j] = o2 == 0 ? 0u : o2 ^ w2;
        }
    }
}
// what is left of an insertion: 0 nothing, 1 duplicate of the entry at position (old & 0x3fff) - 1, 2 lost in both tables
__device__ __forceinline__ unsigned stream_outcome(unsigned left) { return left == 0 ? 0u : (left < 0x4000u ? 1u : 2u); }
template <int LOGT>
__device__ __forceinline__ unsigned stream_insert(const StreamTabs<LOGT> &tb, int c, int pos1, unsigned &old)
{
    const int cc[1] = {c}, pp[1] = {pos1};
    unsigned oo[1], rr[1];
    stream_insert_n<LOGT, 1>(tb, cc, pp, oo, rr);
    old = oo[0];
    return stream_outcome(rr[0]);
}

// the slow side of an insertion (rare): duplicates to the fix-up list, double losers to the row's list.  prim = false for a
// clamped copy of an entry: it reports nothing (the entry itself does).
__device__ __forceinline__ void stream_report(unsigned res, unsigned old, bool prim, int c, int v, int pos1, int *misc, int2 *fix, int fcap, int4 *lst)
{
    if (!prim || res == 0) return;
    if (res == 1) stream_fix_push(misc, fix, fcap, (int)(old & 0x3fffu) - 1, pos1 - 1, v);
    else {
        const int i = atomicAdd(misc + 1, 1);
        if (i < SLCAP) lst[i] = make_int4(c, v, pos1 - 1, 0);
    }
}

// TPR = threads cooperating on one row: 64 (a wave per row, WPB independent rows per workgroup, no barriers) or WPB * 64
// MAXR = rounds of pivot rows kept in registers per row; round r hands pivot row gg + r * NG of the row's record list to
//        the 8-lane group gg.  Rows with more records finish in a loop that loads as it goes.
template <int LOGT, int TPR, int WPB, int MAXR, bool SMALL, int MINW>
__global__ __launch_bounds__(WPB * 64, MINW) void k_stream(StreamArgs a)
{
    constexpr bool WAVE_ROW = (TPR == 64);
    static_assert(WAVE_ROW || TPR == WPB * 64, "a row is owned by one wave or by the whole workgroup");
    constexpr int T1 = 1 << LOGT;
    constexpr int G = 8;
    constexpr int NG = TPR / G;
    constexpr int FCAP = SFIX;
    constexpr size_t TABB = (size_t)5 * T1, MISCB = 64, FIXB = (size_t)FCAP * 8;
    constexpr size_t SLOT = stream_row_bytes(LOGT);
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rtid = WAVE_ROW ? lane : tid;
    unsigned char *base = s_raw + (WAVE_ROW ? (size_t)wave * SLOT : 0);
    StreamTabs<LOGT> tb;
    tb.bind(base);
    // misc[par * 8 + 0] = fix-ups pushed, [+ 1] = losers listed, [+ 4 + w] = leftmost column seen by wave w (block-per-row);
    // two parities so that a row's words can be reset while the waves are still reading the previous row's
    int *misc = (int *)(base + TABB);
    int2 *fix = (int2 *)(base + TABB + MISCB);
    int4 *lst = (int4 *)(base + TABB + MISCB + FIXB);
    const int gg = rtid / G, gl = rtid % G;
    const ZpField F = a.F;

    const int count = *a.class_count;
    if ((WAVE_ROW ? (int)blockIdx.x * WPB : (int)blockIdx.x) >= count) return;
    for (int s = rtid * 4; s < T1 + T1 / 4; s += TPR * 4) *(int4 *)(tb.t1 + s) = make_int4(0, 0, 0, 0);
    if (rtid < 16) misc[rtid] = (rtid & 7) >= 4 ? INT_MAX : 0;
    __syncthreads();

    const int first = WAVE_ROW ? (int)blockIdx.x * WPB + wave : (int)blockIdx.x;
    const int stride = WAVE_ROW ? (int)gridDim.x * WPB : (int)gridDim.x;
    // statistics: wave-uniform values only (they live in SGPRs)
    u64d c_nnz = 0, c_ent = 0, c_seg = 0;
    int c_rows = 0, c_redo = 0;
    int par = 0;

    // ---- pipeline registers: descriptors of this row and the next two (SGPRs; a fourth in flight), records of this row and
    // the next, pivot-row entries and first own entries of this row.  Everything is loaded unconditionally with clamped
    // indices (beyond the last row: the last descriptor again), see k_scatter.
    RowDesc d, dn, dnn;
    int3 rec[MAXR], rec_n[MAXR]; // {position << 16 | npn, multiplier, offset in UPN} (the 4th word of a record is the hash kernel's)
    int2 u[MAXR][3];
    int2 own;
    auto load_rec = [&](const RowDesc &dd, int r) { return *(const int3 *)&a.Lpool[dd.l_start + min(gg + r * NG, max(dd.llen - 1, 0))]; };
    auto load_u = [&](const int3 &rc, int j) {
        const int last = max((rc.x & 0xffff) - 1, 0);
        return a.UPN[(size_t)(unsigned)rc.z + (unsigned)min(gl + j * G, last)];
    };
    {
        d = stream_desc_unpack(stream_desc_load(a.desc + min(first, count - 1)));
        dn = stream_desc_unpack(stream_desc_load(a.desc + min(first + stride, count - 1)));
        dnn = stream_desc_unpack(stream_desc_load(a.desc + min(first + 2 * stride, count - 1)));
#pragma unroll
        for (int r = 0; r < MAXR; r++) { rec[r] = load_rec(d, r); rec_n[r] = load_rec(dn, r); }
        own = a.ent[d.ent_start + min(lane, max(d.len - 1, 0))];
#pragma unroll
        for (int r = 0; r < MAXR; r++)
#pragma unroll
            for (int j = 0; j < 3; j++) u[r][j] = load_u(rec[r], j);
    }

#ifdef SPASM_STAMPS
    u64d st_sum[NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64d st_last = stamp_now();
#endif
    for (int w = first; w < count; w += stride) {
        // ---- (1) the loads of the rows ahead
        const int d3_reg = stream_desc_load(a.desc + min(w + 3 * stride, count - 1));
        int3 rec_nn[MAXR];
        int2 u_n[MAXR][3];
#pragma unroll
        for (int r = 0; r < MAXR; r++) rec_nn[r] = load_rec(dnn, r);
#pragma unroll
        for (int r = 0; r < MAXR; r++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                u_n[r][j] = make_int2(gl + j * G + r * 1024 + gg * 32, 1);
                if (!SCATTER_DBG(a, 8)) u_n[r][j] = load_u(rec_n[r], j);
            }
        const int2 own_n = a.ent[dn.ent_start + min(lane, max(dn.len - 1, 0))];
        const int ln = d.len, ll = d.llen;
        int *const mrow = misc + par * 8;
        int q_own = 0;
        if (d.pmask >= 0) q_own = ((d.pmask >> lane) & 1) ? 0 : -1;
        else if (lane < ln) q_own = a.qinv_r[own.x];
        unsigned char *const rowp = (unsigned char *)(a.Sent + d.s_start);
        int mylead = INT_MAX;
        int nN = 0;
        unsigned rare = 0; // any lane, any entry: something to report
        STAMP(0); // issue of the loads of the rows ahead
        // ---- (2) the row's own entries on non-pivot columns: stream positions 0 .. nN-1, one wave (their rank is a ballot)
        if (WAVE_ROW || wave == 0) {
            {
                const bool nonpiv = lane < ln && q_own < 0;
                const u64d m = __ballot(nonpiv);
                const int pos1 = __popcll(m & lanemask_lt()) + 1;
                if (nonpiv) {
                    if (!SCATTER_DBG(a, 1)) __builtin_nontemporal_store(((long long)(unsigned)own.y << 32) | (unsigned)own.x, (long long *)(rowp + ((unsigned)(pos1 - 1) << 3)));
                    mylead = min(mylead, own.x);
                    if (!SCATTER_DBG(a, 2)) {
                        unsigned old;
                        const unsigned res = stream_insert<LOGT>(tb, own.x, pos1, old);
                        if (res) stream_report(res, old, true, own.x, own.y, pos1, mrow, fix, FCAP, lst);
                    }
                }
                nN = __popcll(m);
            }
            for (int k0 = 64; k0 < ln; k0 += 64) { // rows longer than a wave (wave-uniform trip count)
                const int k = k0 + lane;
                int2 e = make_int2(0, 0);
                bool nonpiv = false;
                if (k < ln) {
                    e = a.ent[d.ent_start + k];
                    nonpiv = a.qinv_r[e.x] < 0;
                }
                const u64d m = __ballot(nonpiv);
                const int pos1 = nN + __popcll(m & lanemask_lt()) + 1;
                if (nonpiv) {
                    __builtin_nontemporal_store(((long long)(unsigned)e.y << 32) | (unsigned)e.x, (long long *)(rowp + ((unsigned)(pos1 - 1) << 3)));
                    mylead = min(mylead, e.x);
                    unsigned old;
                    const unsigned res = stream_insert<LOGT>(tb, e.x, pos1, old);
                    if (res) stream_report(res, old, true, e.x, e.y, pos1, mrow, fix, FCAP, lst);
                }
                nN += __popcll(m);
            }
        }
        STAMP(1); // own entries
        // ---- (3) the pivot rows: multiply, store at the entry's stream position, CAS for the duplicate check
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            if (r * NG < ll) { // scalar: there are records for this round
                const int np = rec[r].x & 0xffff;
                const int nm = -rec[r].y;
                const int last = max(np - 1, 0);
                const int pre1 = (int)((unsigned)rec[r].x >> 16) + 1;
                // a pivot row may consist of its pivot alone: such a group (and a group that repeats such a record) sits out
                const bool gok = np > 0;
                int bc[3], bv[3], bp[3];
                unsigned old[3], res[3];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    bc[j] = u[r][j].x;
                    bv[j] = stream_mul<SMALL>(F, nm, u[r][j].y);
                    bp[j] = pre1 + min(gl + j * G, last);
                    res[j] = 0;
                    old[j] = 0;
                }
                if (gok) {
#pragma unroll
                    for (int j = 0; j < 3; j++)
                        if (!SCATTER_DBG(a, 1)) __builtin_nontemporal_store(((long long)(unsigned)bv[j] << 32) | (unsigned)bc[j], (long long *)(rowp + ((unsigned)(bp[j] - 1) << 3)));
                    if (!SCATTER_DBG(a, 2)) stream_insert_n<LOGT, 3>(tb, bc, bp, old, res);
                    mylead = min(mylead, min(bc[0], min(bc[1], bc[2])));
                }
                if (__ballot((res[0] | res[1] | res[2]) != 0) != 0) { // rare: duplicates, double losers
                    const bool gprim = gg + r * NG < ll;
#pragma unroll
                    for (int j = 0; j < 3; j++) stream_report(stream_outcome(res[j]), old[j], gprim && gl + j * G <= last, bc[j], bv[j], bp[j], mrow, fix, FCAP, lst);
                }
                if (__ballot(np > 3 * G) != 0) { // pivot rows longer than 24 entries
                    const int2 *up = a.UPN + (unsigned)rec[r].z;
                    const bool gprim = gg + r * NG < ll;
                    for (int k = gl + 3 * G; k < np && gprim; k += G) {
                        const int2 uu = up[k];
                        const int vv = stream_mul<SMALL>(F, nm, uu.y);
                        const int pp1 = pre1 + k;
                        __builtin_nontemporal_store(((long long)(unsigned)vv << 32) | (unsigned)uu.x, (long long *)(rowp + ((unsigned)(pp1 - 1) << 3)));
                        mylead = min(mylead, uu.x);
                        unsigned o2;
                        const unsigned r2 = stream_insert<LOGT>(tb, uu.x, pp1, o2);
                        if (r2) stream_report(r2, o2, true, uu.x, vv, pp1, mrow, fix, FCAP, lst);
                    }
                }
            }
        }
        // ---- more pivot rows than NG * MAXR: one round at a time, loaded as it goes (wave-uniform trip count)
        for (int e0 = MAXR * NG; e0 < ll; e0 += NG) {
            const int e = e0 + gg;
            int4 le = make_int4(0, 0, 0, 0);
            if (e < ll) le = a.Lpool[d.l_start + e];
            const int np = le.x & 0xffff;
            const int nm = -le.y;
            const int pre1 = (int)((unsigned)le.x >> 16) + 1;
            const int2 *up = a.UPN + (unsigned)le.z;
            for (int k = gl; k < np; k += G) {
                const int2 uu = up[k];
                const int vv = stream_mul<SMALL>(F, nm, uu.y);
                const int pp1 = pre1 + k;
                __builtin_nontemporal_store(((long long)(unsigned)vv << 32) | (unsigned)uu.x, (long long *)(rowp + ((unsigned)(pp1 - 1) << 3)));
                mylead = min(mylead, uu.x);
                unsigned o2;
                const unsigned r2 = stream_insert<LOGT>(tb, uu.x, pp1, o2);
                if (r2) stream_report(r2, o2, true, uu.x, vv, pp1, mrow, fix, FCAP, lst);
            }
        }
        STAMP(2); // rounds of pivot rows
        const int t_cur = d.t, E = d.bound, ln_cur = ln;
        // ---- (4) end of the row: leftmost column, losers of both tables, duplicates
        mylead = wave_min_i32(mylead);
        int lead_out = mylead;
        if (!WAVE_ROW) {
            if (lane == 0) mrow[4 + wave] = mylead;
            lds_barrier();
#pragma unroll
            for (int w2 = 0; w2 < WPB; w2++) lead_out = min(lead_out, mrow[4 + w2]);
        }
        // the lists are wave 0's business from here on (the other waves go on to reset the tables: nothing below touches those)
        bool redo = false;
        int nfix = 0;
        const int n_out = E; // duplicates are merged by k_stream_fix, which corrects the length then
        if (WAVE_ROW || wave == 0) {
            const int nlist = __builtin_amdgcn_readfirstlane(*(volatile int *)(mrow + 1));
            redo = nlist > SLCAP;
            if (nlist != 0 && !redo) {
                // entries that lost in both tables are in neither: compare them among themselves (a handful)
                for (int b = 0; b < nlist; b += 64) {
                    const int i = b + lane;
                    int4 me = make_int4(-1, 0, -1, 0);
                    if (i < nlist) me = lst[i];
                    int owner = -1;
                    for (int j = 0; j < nlist; j++) {
                        const int4 o = lst[j];
                        if (j < i && owner < 0 && o.x == me.x && o.z != me.z) owner = o.z;
                    }
                    if (owner >= 0) stream_fix_push(mrow, fix, FCAP, owner, me.z, me.y);
                }
            }
            nfix = SCATTER_DBG(a, 4) ? 0 : __builtin_amdgcn_readfirstlane(*(volatile int *)mrow);
            redo = redo || nfix > FCAP;
        }
        if (WAVE_ROW || wave == 0) {
            if (redo) {
                // too many duplicate columns for the lists: the hash-table kernel of this size class takes the row
                if (lane == 0) {
                    const int at = atomicAdd(a.redo_count, 1);
                    a.redo_desc[at] = d;
                }
                c_redo += 1;
            } else {
                if (lane < nfix) a.fixbuf[(size_t)t_cur * SFIX + lane] = fix[lane];
                if (lane == 0) {
                    a.Slen[t_cur] = n_out;
                    a.Slead[t_cur] = n_out > 0 ? lead_out : INT_MAX;
                    if (nfix) a.fixcnt[t_cur] = nfix; // (zero from the binning pass otherwise)
                }
                c_nnz += (u64d)n_out;
                c_rows += n_out > 0;
                // entries streamed: the own entries + the non-pivot parts of the applied pivot rows (= E - nN, the records of the
                // combine kernel all carry a multiplier); segments: the row + one per record
                c_ent += (u64d)ln_cur + (u64d)(E - nN);
                c_seg += 1 + (u64d)ll;
            }
        }
        STAMP(3); // end of the row: lead, lists, fix-ups
        // ---- (5) reset: the tables, and the other parity's words (nobody reads them any more: every wave is past its barrier)
        for (int s = rtid * 4; s < T1 + T1 / 4; s += TPR * 4) *(int4 *)(tb.t1 + s) = make_int4(0, 0, 0, 0);
        if (WAVE_ROW) {
            if (lane < 2) mrow[lane] = 0;
            __builtin_amdgcn_wave_barrier();
        } else {
            par ^= 1;
            if (rtid < 8) misc[par * 8 + rtid] = rtid >= 4 ? INT_MAX : 0;
            lds_barrier();
        }
        STAMP(4); // table reset + barrier
        // ---- (6) rotate the pipeline
        d = dn;
        dn = dnn;
        dnn = stream_desc_unpack(d3_reg);
        own = own_n;
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            rec[r] = rec_n[r];
            rec_n[r] = rec_nn[r];
#pragma unroll
            for (int j = 0; j < 3; j++) u[r][j] = u_n[r][j];
        }
#ifdef SPASM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        STAMP(5); // rotation: waits for the loads issued at the top
    }
#ifdef SPASM_STAMPS
    if (lane == 0 && a.stamps) {
        for (int i = 0; i < NSTAMP; i++) atomicAdd(&a.stamps[(size_t)a.cls * 2 * NSTAMP + i], st_sum[i]);
        atomicAdd(&a.stamps[(size_t)a.cls * 2 * NSTAMP + NSTAMP], 1ull);
    }
#endif
    if (lane == 0 && (WAVE_ROW || wave == 0)) {
        if (c_nnz) atomicAdd(&ctr_shard(a.ctr)->nnz_out, c_nnz);
        if (c_rows) atomicAdd(&ctr_shard(a.ctr)->nonempty_out, c_rows);
        if (c_redo) atomicAdd(&ctr_shard(a.ctr)->stream_redo, c_redo);
        if (c_ent | c_seg) {
            atomicAdd(&ctr_shard(a.ctr)->class_ent[a.cls], c_ent);
            atomicAdd(&ctr_shard(a.ctr)->class_seg[a.cls], c_seg);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The duplicates the streaming kernels found, merged row by row: a wave looks at 64 row slots and works through those that
// have any (about a third of the rows of 1000 entries, a few percent of the short ones).
// ------------------------------------------------------------------------------------------------
struct StreamFixArgs {
    int nrows;
    const int *fixcnt;
    const int2 *fixbuf;
    const i64d *sstart;
    int2 *Sent;
    int *Slen;
    int *Slead;
    RoundCounters *ctr;
    ZpField F;
};

__global__ __launch_bounds__(256) void k_stream_fix(StreamFixArgs a)
{
    __shared__ int s_scratch[4][8 * SFIX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 16 row slots per wave: the rows with duplicates are worked through one after the other, each a chain of global round trips
    const int t0 = (blockIdx.x * 4 + wave) * 16;
    if (t0 >= a.nrows) return;
    const int mine = (lane < 16 && t0 + lane < a.nrows) ? a.fixcnt[t0 + lane] : 0;
    u64d m = __ballot(mine > 0);
    u64d holes = 0;
    int emptied = 0, merged = 0;
    while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int t = t0 + l;
        const int n = __builtin_amdgcn_readlane(mine, l);
        const int E = a.Slen[t];
        int lead = a.Slead[t];
        const int n_out = stream_fixup<SFIX>(a.F, a.fixbuf + (size_t)t * SFIX, n, s_scratch[wave], (u64d *)(a.Sent + a.sstart[t]), E, &lead);
        if (lane == 0) {
            a.Slen[t] = n_out;
            a.Slead[t] = n_out > 0 ? lead : INT_MAX;
        }
        holes += (u64d)(E - n_out);
        emptied += (n_out == 0 && E > 0);
        merged += n;
    }
    if (lane == 0 && merged) {
        atomicAdd(&ctr_shard(a.ctr)->nnz_out, (u64d)0 - holes); // (mod 2^64: the streaming kernel counted the stream lengths)
        if (emptied) atomicAdd(&ctr_shard(a.ctr)->nonempty_out, -emptied);
        atomicAdd(&ctr_shard(a.ctr)->stream_fix, merged);
    }
}
