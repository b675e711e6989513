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
constexpr int SFIX = 64;  // duplicates a row may collect before it is handed to the hash-table kernel: random columns collide rarely (a row
                          // of 2560 out of 10^6 expects 3), but two runs of W that share a pivot row share its ~17 columns

// LDS of one row: three tables of 2^logt, 2^(logt-1) and 2^(logt-2) words, 64 B of counters, fix-up list, loser list
__host__ __device__ constexpr size_t stream_row_bytes(int logt) { return ((size_t)7 << logt) + 64 + (size_t)SFIX * 8 + (size_t)SLCAP * 16; }
__host__ __device__ constexpr size_t stream_lds_bytes(int logt, int tpr, int wpb) { return tpr == 64 ? stream_row_bytes(logt) * (size_t)wpb : stream_row_bytes(logt); }

// the row's fix-up list: {owner position << 14 | own position, value}; positions are below 2^14
__device__ __forceinline__ void stream_fix_push(int *s_nfix, int2 *fix, int fcap, int owner_pos, int pos, int v)
{
    const int i = atomicAdd(s_nfix, 1);
    if (i < fcap) fix[i] = make_int2((int)(((unsigned)owner_pos << 14) | (unsigned)pos), v);
}

constexpr unsigned STREAM_K1 = 0x9E3779u, STREAM_K2 = 0x85EBCBu, STREAM_K3 = 0xC2B2AFu; // odd: c -> c * K mod 2^24 is a bijection

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

// The three direct-mapped tables of a row: 2^LOGT words at most 5/8 full, then a half and a quarter of that for the entries that
// met another column in the table before (about 30 %, then 6 %, then 0.4 % of the entries: those go to the row's list).
template <int LOGT> struct StreamTabs {
    static constexpr int WORDS = (1 << LOGT) + (1 << (LOGT - 1)) + (1 << (LOGT - 2));
    unsigned *t1;
    __device__ __forceinline__ void bind(unsigned char *p) { t1 = (unsigned *)p; }
    // level lv (0, 1, 2): the word wanted for (column, position + 1) and its slot (index into t1[], the tables lie back to back)
    template <int LV> __device__ __forceinline__ unsigned want(int c, int pos1, unsigned &slot) const
    {
        constexpr int L = LOGT - LV;
        constexpr unsigned K = LV == 0 ? STREAM_K1 : (LV == 1 ? STREAM_K2 : STREAM_K3);
        constexpr unsigned BASE = LV == 0 ? 0u : (LV == 1 ? (1u << LOGT) : (1u << LOGT) + (1u << (LOGT - 1)));
        const unsigned x = stream_mul24(c, K);
        slot = BASE + __builtin_amdgcn_ubfe(x, 24 - L, L);
        return (x << 14) | (unsigned)pos1;
    }
};

// Insertion of N entries of a lane: all first-table CAS are in flight at once, then the second-table CAS of those that met
// another column, then the third -- three LDS round trips per batch instead of up to 3 N.  Outcome per entry, as ONE word and
// without branches:
//   t = 0 if the CAS found the slot empty, else (word found) ^ (word wanted):  t = 0  the entry is in (new, or it met itself: a
//   lane past the end of a run repeats its last entry);  0 < t < 2^14  same column under another position: a duplicate;
//   t >= 2^14  another column.
// left[j] = 0 when entry j needs nothing more; otherwise the rare side decodes it with stream_outcome().
template <int LOGT, int N>
__device__ __forceinline__ void stream_insert_n(const StreamTabs<LOGT> &tb, const int (&c)[N], const int (&pos1)[N], unsigned (&old)[N], unsigned (&left)[N])
{
    unsigned w1[N], s1[N];
#pragma unroll
    for (int j = 0; j < N; j++) w1[j] = tb.template want<0>(c[j], pos1[j], s1[j]);
#pragma unroll
    for (int j = 0; j < N; j++) old[j] = atomicCAS(&tb.t1[s1[j]], 0u, w1[j]);
#pragma unroll
    for (int j = 0; j < N; j++) left[j] = old[j] == 0 ? 0u : old[j] ^ w1[j];
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (left[j] >= 0x4000u) { // another column in the slot: second table
            unsigned s2;
            const unsigned w2 = tb.template want<1>(c[j], pos1[j], s2);
            const unsigned o2 = atomicCAS(&tb.t1[s2], 0u, w2);
            old[j] = o2;
            left[j] = o2 == 0 ? 0u : o2 ^ w2;
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (left[j] >= 0x4000u) { // and again: third table
            unsigned s3;
            const unsigned w3 = tb.template want<2>(c[j], pos1[j], s3);
            const unsigned o3 = atomicCAS(&tb.t1[s3], 0u, w3);
            old[j] = o3;
            left[j] = o3 == 0 ? 0u : o3 ^ w3;
        }
    }
}
// what is left of an insertion: 0 nothing, 1 duplicate of the entry at position (old & 0x3fff) - 1, 2 lost in all three tables
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

// ------------------------------------------------------------------------------------------------
// W = -(I + U_PP)^-1 * U_PN, one row per pivot: with it the Schur row of B[k] needs no chain of eliminations,
//        x_a = B[k]_N + sum over the entries (c, a_c) of B[k] on pivot columns of a_c * W[qinv(c)]
// (x_b * U_PN with x_b = B[k]_P * Uinv, re-associated; same x_a, reference src/SpaSM.jl:704-707).  A row of config 3 applies
// ~34 pivot rows of ~17 entries through its multiplier list; it has ~3.4 entries on pivot columns, and their rows of W hold the
// same ~580 entries in 3.4 contiguous runs: no multiplier list to build, and a wave streams a run 64 entries at a time with
// every lane busy.  W is built by every Schur step, level by level of the pivot graph (wlevel.hpp); its rows live in one buffer
// with U_PN and the rows' own entries, so a record {position << 16 | len, -a_c, offset, len} reads the same for every scatter
// kernel.
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// The plan of a row's Schur row under W: the run of every entry on a pivot column, cut into chunks of 64 entries with one record
// each, their stream positions (runs first, in record order; the row's own non-pivot entries fill the stream from its end), the
// bound = length of the stream.  Takes the place of
// the combine kernel: no Uinv rows to merge, no headers to gather.  TEAM lanes per row, two passes over the row's entries
// (the first counts, so that the records can be allocated; the entries come back from L2).
// ------------------------------------------------------------------------------------------------
struct WPlanArgs {
    int nrows;
    const i64d *rstart;        // per row slot: start / length of the row's own entries
    const int *rlen;
    const int2 *ent;
    const unsigned *pbits;     // bit j: column j is a pivot column of this round
    const int4 *wcol;          // per pivot column: {pivot index, length of its row of W (-1: not available), offset, -}
    int4 *Lpool;               // records {position << 16 | entries, -multiplier, offset in the U_PN + own + W buffer, entries}
    u64d lpool_cap;            // records per pool region
    u64d *pool_ctr;
    int npool;
    int2 *upn;                 // the U_PN + own + W buffer: the rows' own entries on non-pivot columns are copied in front of W
    unsigned own_base;         // where that part starts
    u64d own_cap;              // entries per region of it
    u64d *own_ctr;             // its bump counters (NPOOL, POOL_STRIDE apart)
    i64d *Lstart;
    int *Llen;
    i64d *bound;
    long long *pmask;
    int *sflag;
    int free_cols;
    int max_bound;             // rows with a longer stream are left to the multiplier-list path
    int wave_row_bound;        // streams up to this length are handled by one wave
    int *overflow_list;        // rows this kernel leaves to the combine kernel
    int *overflow_count;
    RoundCounters *ctr;
    ZpField F;
    int chunk_log;             // 6: chunks of 64 entries (a lane of the streaming kernel takes one), 7: of 128 (a lane takes two)
};

// The plan of a row's Schur row under W: its stream = the row's own entries on non-pivot columns (copied, compacted, behind W: they
// are then chunks like the others, with multiplier 1) followed by the run of every entry on a pivot column; one record per chunk
// of 64 consecutive entries; the bound = length of the stream.  Takes the place of the combine kernel: no Uinv rows to merge, no
// headers to gather.  TEAM lanes per row; rows of up to 2 TEAM entries (all of config 3) stay in registers between the
// counting and the writing pass, longer ones are read twice.  Records and own entries are carved from blocks a team takes from
// the pools now and then (one returning atomic per ~16 rows instead of two per row).
template <int TEAM, int TPB>
__global__ __launch_bounds__(TPB) void k_wplan(WPlanArgs a)
{
    constexpr int TEAMS = TPB / TEAM;
    constexpr u64d RBLK = 64, OBLK = 256;
    const int tl = threadIdx.x % TEAM;
    const int team = threadIdx.x / TEAM;
    const ZpField F = a.F;
    const int CL = a.chunk_log, CH = 1 << CL;
    u64d rpos = 0, rend = 0, opos = 0, oend = 0; // the team's current blocks (uniform in the team)
    auto take = [&](u64d &pos, u64d &end, u64d need, u64d blk, u64d *ctr, u64d cap) -> u64d {
        if (pos + need > end) {
            const u64d n = need > blk ? need : blk;
            u64d b = 0;
            if (tl == 0) b = pool_alloc(ctr, cap, n, a.npool);
            b = __shfl(b, 0, TEAM);
            if (b == ~0ull) return ~0ull;
            pos = b;
            end = b + n;
        }
        const u64d r = pos;
        pos += need;
        return r;
    };
    auto is_piv = [&](int c) -> bool { return (a.pbits[(unsigned)c >> 5] >> (c & 31)) & 1u; };
    // pipeline: (start, length) and the first 2 TEAM entries of the next row are loaded one row ahead
    const i64d first = (i64d)blockIdx.x * TEAMS + team, stride = (i64d)gridDim.x * TEAMS;
    i64d st_n = 0;
    int ln_n = 0;
    int2 ea_n = make_int2(0, 0), eb_n = make_int2(0, 0);
    if (first < a.nrows) {
        st_n = a.rstart[first];
        ln_n = a.rlen[first];
        if (tl < ln_n) ea_n = a.ent[st_n + tl];
        if (tl + TEAM < ln_n) eb_n = a.ent[st_n + tl + TEAM];
    }
    for (i64d t64 = first; t64 < a.nrows; t64 += stride) {
        const int t = (int)t64;
        const i64d st = st_n;
        const int ln = ln_n;
        const int2 ea = ea_n, eb = eb_n;
        if (t64 + stride < a.nrows) {
            st_n = a.rstart[t64 + stride];
            ln_n = a.rlen[t64 + stride];
            ea_n = make_int2(0, 0);
            eb_n = make_int2(0, 0);
            if (tl < ln_n) ea_n = a.ent[st_n + tl];
            if (tl + TEAM < ln_n) eb_n = a.ent[st_n + tl + TEAM];
        }
        // ---- pass 1: count.  C chunks of runs, nN entries on non-pivot columns
        int C = 0, nN = 0;
        bool zero_own = false;
        u64d pm = 0;
        int4 ci_a = make_int4(-1, 0, 0, 0), ci_b = ci_a;
        bool pa = false, pb = false;
        for (int k0 = 0; k0 < ln; k0 += TEAM) {
            const int k = k0 + tl;
            const bool valid = k < ln;
            int2 e = k0 == 0 ? ea : (k0 == TEAM ? eb : make_int2(0, 0));
            if (k0 >= 2 * TEAM && valid) e = a.ent[st + k];
            const bool isP = valid && is_piv(e.x);
            int4 ci = make_int4(-1, 0, 0, 0);
            if (isP) ci = a.wcol[e.x];
            if (k0 == 0) { ci_a = ci; pa = isP; }
            if (k0 == TEAM) { ci_b = ci; pb = isP; }
            zero_own |= valid && (e.y == 0 || (isP && ci.y < 0)); // (a row of W that could not be built: the lists take the row)
            const u64d mP = team_ballot<TEAM>(isP);
            if (k0 < 64) pm |= mP << (k0 & 63);
            int tot;
            (void)team_incl_scan<TEAM>(isP ? (ci.y + CH - 1) >> CL : 0, tot);
            C += tot;
            nN += __popcll(team_ballot<TEAM>(valid && !isP));
        }
        const bool anyzero = team_ballot<TEAM>(zero_own) != 0;
        const int own_chunks = (nN + CH - 1) >> CL;
        const int R = C + own_chunks;
        const u64d base = take(rpos, rend, (u64d)R, RBLK, a.pool_ctr, a.lpool_cap);
        const u64d obase = base == ~0ull ? ~0ull : take(opos, oend, (u64d)nN, OBLK, a.own_ctr, a.own_cap);
        // ---- pass 2: write.  The stream: own entries first (positions 0 .. nN-1), then the runs in entry order
        i64d run = nN;
        int w = own_chunks, no = 0;
        const bool room = base != ~0ull && obase != ~0ull;
        for (int k0 = 0; k0 < ln && room; k0 += TEAM) {
            const int k = k0 + tl;
            const bool valid = k < ln;
            int2 e = k0 == 0 ? ea : (k0 == TEAM ? eb : make_int2(0, 0));
            if (k0 >= 2 * TEAM && valid) e = a.ent[st + k];
            bool isP = k0 == 0 ? pa : (k0 == TEAM ? pb : false);
            int4 ci = k0 == 0 ? ci_a : (k0 == TEAM ? ci_b : make_int4(-1, 0, 0, 0));
            if (k0 >= 2 * TEAM) {
                isP = valid && is_piv(e.x);
                if (isP) ci = a.wcol[e.x];
            }
            // the entry itself, when it sits on a non-pivot column
            const u64d mN = team_ballot<TEAM>(valid && !isP);
            if (valid && !isP) a.upn[(size_t)a.own_base + obase + no + __popcll(mN & ((1ull << tl) - 1ull))] = e;
            no += __popcll(mN);
            // its run, when it sits on a pivot column
            const int len = isP ? ci.y : 0;
            const int nch = (len + CH - 1) >> CL;
            int tot, ctot;
            const int incl = team_incl_scan<TEAM>(len, tot);
            const int cincl = team_incl_scan<TEAM>(nch, ctot);
            const int nm = zp_neg(F, e.y);
            i64d pre = run + incl - len;
            int4 *out = a.Lpool + base + w + (cincl - nch);
            for (int q = 0; q < nch; q++) { // (the lanes of a team have runs of different lengths)
                const int clen = min(CH, len - CH * q);
                const unsigned px = (pre < 0x8000 ? (unsigned)pre : 0x7fffu) << 16 | (unsigned)clen;
                out[q] = make_int4((int)px, nm, (int)((unsigned)ci.z + (unsigned)CH * (unsigned)q), clen);
                pre += clen;
            }
            w += ctot;
            run += tot;
        }
        if (room) // the own entries as chunks with multiplier 1 (the scatter kernels subtract record.y times the entry)
            for (int q = tl; q < own_chunks; q += TEAM) {
                const int clen = min(CH, nN - CH * q);
                a.Lpool[base + q] = make_int4((int)(((unsigned)(CH * q) << 16) | (unsigned)clen), -1, (int)(a.own_base + (unsigned)obase + (unsigned)CH * (unsigned)q), clen);
            }
        if (tl == 0) {
            if (base == ~0ull) { // the record pool is full: the host grows it and runs the plan again
                atomicAdd(&ctr_shard(a.ctr)->lpool_overflow, 1);
                a.Llen[t] = 0; a.Lstart[t] = 0; a.bound[t] = 0;
            } else {
                const i64d bound = run;
                // a wave holds 64 chunk records: one wave per row up to wave_row_bound entries, four beyond.  (No room for the own
                // entries -- the regions of that buffer fill unevenly -- also sends the row to the lists.)
                const bool ok = room && bound <= (i64d)a.free_cols && bound <= (i64d)a.max_bound && !anyzero && R <= (bound <= (i64d)a.wave_row_bound ? 64 : 256);
                if (ok) {
                    a.Lstart[t] = (i64d)base;
                    a.Llen[t] = R;
                    a.bound[t] = bound;
                    a.pmask[t] = ln <= 32 ? (long long)pm : -1;
                    a.sflag[t] = 1;
                } else {
                    // too long, too many duplicates to expect, or a zero among its entries: the multiplier-list path takes the row
                    a.Llen[t] = -1; a.Lstart[t] = 0; a.bound[t] = 0;
                    a.overflow_list[atomicAdd(a.overflow_count, 1)] = t;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// SCATTER along the rows of W.  A row's record list names its chunks (up to 64 consecutive entries of a run, their multiplier
// and stream position); record w + g * NW is the g-th chunk of wave w of the NW waves that share the row, and lane g of that wave
// holds it.  Every entry is treated as k_stream treated it: multiply, store at its stream position, one CAS in the
// direct-mapped tables for the duplicate check.  D chunk loads per wave are in flight: those of a row are requested while the
// row before it is being finished.
// TPR = threads per row (64: a wave per row, WPB rows per workgroup; else the workgroup).
// ------------------------------------------------------------------------------------------------
// EPL = entries per lane and chunk: 1 (chunks of up to 64 entries, an 8-byte load and store per lane) or 2 (chunks of up to 128: a lane
// takes the entries 2 l and 2 l + 1 with ONE 16-byte load, and one 16-byte store where the chunk is full; the per-chunk work --
// the record's fields by v_readlane, the ring slot, the group bookkeeping -- is paid once per 128 entries).
typedef int v4i32s __attribute__((ext_vector_type(4)));
// BLOOM: the duplicate check as a FILTER instead of the three exact tag tables.  A row's duplicates are rare (config 3: 0.35 per row of
// 630 entries) but the tables make every entry pay for them -- hash, CAS, compare, and, because SOME lane of 64 always meets another
// column, the second- and third-table code of every chunk: 33 of a chunk's 59 VALU.  Here an entry sets two bits of ONE word of a
// 2^LOGT-word bitmap with one returning ds_or (one word: the later of two equal columns sees both bits set whatever the interleaving
// of lanes and waves); an entry whose bits were set already is a SUSPECT (a duplicate, or ~0.1 % of the entries by chance), goes on
// the row's list, and is resolved at the end of the row against the row itself, read back from where it was just stored.
template <int LOGT, int TPR, int WPB, int D, bool SMALL, int MINW, int EPL = 1, int QX = 0, bool BLOOM = false>
__global__ __launch_bounds__(WPB * 64, MINW) void k_wstream(StreamArgs a)
{
    static_assert(EPL == 1 || EPL == 2, "one or two entries per lane");
    static_assert(!(BLOOM && EPL == 2), "the filter is built for one entry per lane");
    constexpr bool WAVE_ROW = (TPR == 64);
    static_assert(WAVE_ROW || TPR == WPB * 64, "a row is owned by one wave or by the whole workgroup");
    constexpr int NW = WAVE_ROW ? 1 : WPB; // waves sharing a row
    constexpr int FCAP = SFIX;
    constexpr size_t TABB = (size_t)4 * StreamTabs<LOGT>::WORDS, MISCB = 64, FIXB = (size_t)FCAP * 8;
    constexpr size_t SLOT = stream_row_bytes(LOGT);
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rtid = WAVE_ROW ? lane : tid;
    const int rw = WAVE_ROW ? 0 : wave; // this wave's number among the waves of its row
    unsigned char *base = s_raw + (WAVE_ROW ? (size_t)wave * SLOT : 0);
    StreamTabs<LOGT> tb;
    tb.bind(base);
    int *misc = (int *)(base + TABB);
    int2 *fix = (int2 *)(base + TABB + MISCB);
    int4 *lst = (int4 *)(base + TABB + MISCB + FIXB);
    const ZpField F = a.F;

    const int count = *a.class_count;
    if ((WAVE_ROW ? (int)blockIdx.x * WPB : (int)blockIdx.x) >= count) return;
    for (int s = rtid * 4; s < StreamTabs<LOGT>::WORDS; s += TPR * 4) *(int4 *)(tb.t1 + s) = make_int4(0, 0, 0, 0);
    if (rtid < 16) misc[rtid] = (rtid & 7) >= 4 ? INT_MAX : 0;
    __syncthreads();

    const int first = WAVE_ROW ? (int)blockIdx.x * WPB + wave : (int)blockIdx.x;
    const int stride = WAVE_ROW ? (int)gridDim.x * WPB : (int)gridDim.x;
    u64d c_nnz = 0, c_ent = 0, c_seg = 0; // (wave-uniform: they live in SGPRs)
    int c_rows = 0, c_redo = 0;
    int par = 0;

    // ---- pipeline: descriptors of this row and the next two (a fourth in flight), this wave's chunk records of this row and the
    // next.  Unconditional loads with clamped indices (beyond the last row: the last descriptor again).
    RowDesc d, dn, dnn;
    int4 rec, rec_n;
    auto load_rec = [&](const RowDesc &dd) { return a.Lpool[dd.l_start + min(rw + lane * NW, max(dd.llen - 1, 0))]; };
    d = stream_desc_unpack(stream_desc_load(a.desc + min(first, count - 1)));
    dn = stream_desc_unpack(stream_desc_load(a.desc + min(first + stride, count - 1)));
    dnn = stream_desc_unpack(stream_desc_load(a.desc + min(first + 2 * stride, count - 1)));
    rec = load_rec(d);
    rec_n = load_rec(dn);
    // the ring of chunk loads.  Always D requests per row (slots past the wave's last chunk repeat it, a wave without chunks reads
    // entry 0): with a number of loads that depends on the row the compiler can only wait for ALL loads in flight when an older
    // one is needed.
    // (EPL = 2: a slot holds the entries 2 l and 2 l + 1 of its chunk -- the load is clamped to the chunk's last entry, the entry
    // behind that one belongs to somebody else and is replaced by a copy of the first where it is used)
    typedef typename std::conditional<EPL == 2, int4, int2>::type ring_t;
    ring_t ring[D];
    auto ring_load = [&](unsigned off, int clen) -> ring_t {
        if constexpr (EPL == 2) {
            const v4i32s v = *(const v4i32s *)(a.UPN + (size_t)off + (unsigned)min(2 * lane, clen - 1));
            return make_int4(v.x, v.y, v.z, v.w);
        } else {
            return a.UPN[(size_t)off + (unsigned)min(lane, clen - 1)];
        }
    };
    // (Refilling a group's slots with the next row's chunks as soon as the group has used them -- in front of most of the row's stores
    // instead of behind all of them; the stamps put 15 - 20 % of a row's cycles into the issue of these loads -- was tried in r04:
    // class 1 0.186 ms against 0.200, classes 3 and 4 0.82 - 0.86 / 0.45 - 0.48 against 0.80 / 0.44: 3.41 ms per step against 3.31.)
    auto request = [&](const int4 &rc, int nmine, bool live) {
#pragma unroll
        for (int j = 0; j < D; j++) {
            const int g = min(j, max(nmine - 1, 0));
            const bool have = live && nmine > 0;
            const unsigned off = have ? (unsigned)__builtin_amdgcn_readlane(rc.z, g) : 0u;
            const int clen = have ? __builtin_amdgcn_readlane(rc.w, g) : 1;
            if constexpr (EPL == 1) {
                if (SCATTER_DBG(a, 8)) { ring[j] = make_int2(lane + 4096 * j, 1); continue; }
            }
            ring[j] = ring_load(off, clen);
        }
    };
    auto chunks_of = [&](int ll) { return ll > rw ? (ll - rw + NW - 1) / NW : 0; };
    request(rec, chunks_of(d.llen), first < count);

#ifdef SPASM_STAMPS
    u64d st_sum[NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64d st_last = stamp_now();
#endif
    for (int w = first; w < count; w += stride) {
        const int d3_reg = stream_desc_load(a.desc + min(w + 3 * stride, count - 1));
        const int4 rec_nn = load_rec(dnn);
        const int ll = d.llen, E = d.bound, t_cur = d.t;
        int *const mrow = misc + par * 8;
        unsigned char *const rowp = (unsigned char *)(a.Sent + d.s_start);
        int mylead = INT_MAX;
        const int my_chunks = chunks_of(ll);
        STAMP(0); // requests for the rows ahead
        // ---- the chunks, Q at a time: the first-table CAS of all Q are in flight together, then the second- and third-table ones.
        // The first D come from the ring, in straight-line code (a loop header would cost a vmcnt(0)); what a wave has beyond them
        // (few rows) is loaded and used group by group in a loop.
        // Q chunks = NE entries of a lane per group
        constexpr int Q = QX ? QX : (EPL == 2 ? (D >= 2 ? 2 : 1) : (D >= 4 ? 4 : D)); // (QX: chunks per group chosen by the caller)
        constexpr int NE = Q * EPL;
        static_assert(D % Q == 0, "whole groups of slots");
        auto do_group = [&](const ring_t (&e)[Q], int g0, int nthere) {
            int cc[NE], vv[NE], pp[NE], cl[Q];
            bool there[NE]; // the entry is the lane's own (not a clamped copy)
            unsigned oo[NE], left[NE];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int g = min(g0 + q, max(my_chunks - 1, 0));
                const int rx = __builtin_amdgcn_readlane(rec.x, g);
                const int nmul = -__builtin_amdgcn_readlane(rec.y, g);
                cl[q] = rx & 0xffff;
                if constexpr (EPL == 2) {
                    const int i0 = min(2 * lane, cl[q] - 1);
                    const bool second = 2 * lane + 1 < cl[q]; // (else: a copy of the first, same position: storing and inserting it again changes nothing)
                    cc[2 * q] = e[q].x;
                    vv[2 * q] = stream_mul<SMALL>(F, nmul, e[q].y);
                    pp[2 * q] = (int)((unsigned)rx >> 16) + i0 + 1;
                    cc[2 * q + 1] = second ? e[q].z : e[q].x;
                    vv[2 * q + 1] = second ? stream_mul<SMALL>(F, nmul, e[q].w) : vv[2 * q];
                    pp[2 * q + 1] = pp[2 * q] + (second ? 1 : 0);
                    there[2 * q] = 2 * lane < cl[q];
                    there[2 * q + 1] = second;
                } else {
                    cc[q] = e[q].x;
                    vv[q] = stream_mul<SMALL>(F, nmul, e[q].y);
                    pp[q] = (int)((unsigned)rx >> 16) + min(lane, cl[q] - 1) + 1;
                    there[q] = lane < cl[q];
                }
            }
#pragma unroll
            for (int k = 0; k < NE; k++) { oo[k] = 0; left[k] = 0; }
            auto store_chunk = [&](int q) {
                if constexpr (EPL == 2) {
                    if (cl[q] == 128) { // (wave-uniform) a full chunk: both entries of every lane in one 16-byte store
                        v4i32s v;
                        v.x = cc[2 * q]; v.y = vv[2 * q]; v.z = cc[2 * q + 1]; v.w = vv[2 * q + 1];
                        __builtin_nontemporal_store(v, (v4i32s *)(rowp + ((unsigned)(pp[2 * q] - 1) << 3)));
                    } else {
                        __builtin_nontemporal_store(((long long)(unsigned)vv[2 * q] << 32) | (unsigned)cc[2 * q], (long long *)(rowp + ((unsigned)(pp[2 * q] - 1) << 3)));
                        __builtin_nontemporal_store(((long long)(unsigned)vv[2 * q + 1] << 32) | (unsigned)cc[2 * q + 1], (long long *)(rowp + ((unsigned)(pp[2 * q + 1] - 1) << 3)));
                    }
                    mylead = min(mylead, min(cc[2 * q], cc[2 * q + 1]));
                } else {
                    __builtin_nontemporal_store(((long long)(unsigned)vv[q] << 32) | (unsigned)cc[q], (long long *)(rowp + ((unsigned)(pp[q] - 1) << 3)));
                    mylead = min(mylead, cc[q]);
                }
            };
            if constexpr (BLOOM) {
#pragma unroll
                for (int q = 0; q < Q; q++)
                    if (q < nthere) store_chunk(q); // (wave-uniform)
                unsigned msk[NE], wd[NE], ow[NE];
#pragma unroll
                for (int k = 0; k < NE; k++) {
                    const unsigned h = stream_mul24(cc[k], STREAM_K1), h2 = stream_mul24(cc[k], STREAM_K2);
                    // (all 1.75 * 2^LOGT words of the row's table area; three bits: a chance hit needs another entry in the word
                    // AND its bits to cover these three -- about 0.005 % of the entries)
                    wd[k] = (__builtin_amdgcn_ubfe(h, 8, 16) * (unsigned)StreamTabs<LOGT>::WORDS) >> 16;
                    const unsigned m2 = (1u << __builtin_amdgcn_ubfe(h2, 19, 5)) | (1u << __builtin_amdgcn_ubfe(h2, 14, 5)) | (1u << __builtin_amdgcn_ubfe(h2, 9, 5));
                    msk[k] = (k / EPL < nthere && there[k]) ? m2 : 0u; // (a clamped copy sets nothing and is never a suspect)
                }
#pragma unroll
                for (int k = 0; k < NE; k++) ow[k] = atomicOr(&tb.t1[wd[k]], msk[k]);
                bool sus = false;
#pragma unroll
                for (int k = 0; k < NE; k++) sus |= msk[k] != 0u && (ow[k] & msk[k]) == msk[k];
                if (__ballot(sus) != 0) { // rare
#pragma unroll
                    for (int k = 0; k < NE; k++)
                        if (msk[k] != 0u && (ow[k] & msk[k]) == msk[k]) {
                            const int i = atomicAdd(mrow + 1, 1);
                            if (i < SLCAP) lst[i] = make_int4(cc[k], vv[k], pp[k] - 1, INT_MAX); // (.w: the smallest OTHER position on the column, found below)
                        }
                }
                return;
            }
            if (nthere >= Q) { // a full group (wave-uniform)
#pragma unroll
                for (int q = 0; q < Q; q++)
                    if (!SCATTER_DBG(a, 1)) store_chunk(q);
                    else if constexpr (EPL == 2) mylead = min(mylead, min(cc[2 * q], cc[2 * q + 1]));
                    else mylead = min(mylead, cc[q]);
                if (!SCATTER_DBG(a, 2)) stream_insert_n<LOGT, NE>(tb, cc, pp, oo, left);
            } else {
#pragma unroll
                for (int q = 0; q < Q; q++)
                    if (q < nthere) {
                        store_chunk(q);
                        int c1[EPL], p1[EPL];
                        unsigned o1[EPL], l1[EPL];
#pragma unroll
                        for (int u = 0; u < EPL; u++) { c1[u] = cc[EPL * q + u]; p1[u] = pp[EPL * q + u]; }
                        stream_insert_n<LOGT, EPL>(tb, c1, p1, o1, l1);
#pragma unroll
                        for (int u = 0; u < EPL; u++) { oo[EPL * q + u] = o1[u]; left[EPL * q + u] = l1[u]; }
                    }
            }
            unsigned any = 0;
#pragma unroll
            for (int k = 0; k < NE; k++) any |= left[k];
            if (__ballot(any != 0) != 0) { // rare: duplicates, triple losers; a lane past the end of its chunk holds a copy
#pragma unroll
                for (int k = 0; k < NE; k++)
                    stream_report(stream_outcome(left[k]), oo[k], k / EPL < nthere && there[k], cc[k], vv[k], pp[k], mrow, fix, FCAP, lst);
            }
        };
        const int next_chunks = chunks_of(dn.llen);
        const bool next_live = w + stride < count;
#pragma unroll
        for (int j0 = 0; j0 < D; j0 += Q) {
            if (j0 < my_chunks) { // (wave-uniform)
                ring_t e[Q];
#pragma unroll
                for (int q = 0; q < Q; q++) e[q] = ring[j0 + q];
                do_group(e, j0, my_chunks - j0);
            }
        }
        if (my_chunks > D) {
            for (int g0 = D; g0 < my_chunks; g0 += Q) {
                // (more than 64 chunks per wave do not occur: the plan kernel leaves such rows to the lists)
                ring_t e[Q];
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    const int g = min(g0 + q, my_chunks - 1);
                    e[q] = ring_load((unsigned)__builtin_amdgcn_readlane(rec.z, g), __builtin_amdgcn_readlane(rec.w, g));
                }
                do_group(e, g0, my_chunks - g0);
            }
        }
        STAMP(1); // the chunks
        // ---- the first D chunks of the NEXT row are requested now: they fly while this row is finished
        request(rec_n, next_chunks, next_live);
        STAMP(2); // requests for the next row
        // ---- end of the row: leftmost column, losers of all tables, duplicates
        mylead = wave_min_i32(mylead);
        int lead_out = mylead;
        if (!WAVE_ROW) {
            if (lane == 0) mrow[4 + wave] = mylead;
            lds_barrier();
#pragma unroll
            for (int w2 = 0; w2 < WPB; w2++) lead_out = min(lead_out, mrow[4 + w2]);
        }
        if constexpr (BLOOM) {
            // Suspects are resolved against the row's INPUT, read again (the runs of W are read-only and L2-resident; reading the
            // stored row back would wait for its stores to be acknowledged -- tens of microseconds while the chip streams writes).
            // Every wave of the row scans its own chunks.  Every column has at most one entry that is no suspect (the first to reach
            // its word).  A suspect s with other entries on its column: o = the smallest position of the others; o < s: s merges
            // into o (said once, by the row's first wave); else s owns the column and the others that are no suspects themselves
            // merge into s (a second scan, rare; the suspects among them say so on their own turn).  Nobody else: chance.
            const int nsus = __builtin_amdgcn_readfirstlane(((lds_vint *)mrow)[1]);
            if (nsus != 0 && nsus <= SLCAP) {
                auto scan = [&](bool owners) {
                    for (int g0 = 0; g0 < my_chunks; g0 += 8) {
                        int2 e[8];
                        int ps[8], nm[8];
                        bool ok[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const int g = min(g0 + u, my_chunks - 1);
                            const int clen = __builtin_amdgcn_readlane(rec.w, g);
                            e[u] = a.UPN[(size_t)(unsigned)__builtin_amdgcn_readlane(rec.z, g) + (unsigned)min(lane, clen - 1)];
                            ps[u] = (int)((unsigned)__builtin_amdgcn_readlane(rec.x, g) >> 16) + lane;
                            nm[u] = -__builtin_amdgcn_readlane(rec.y, g);
                            ok[u] = g0 + u < my_chunks && lane < clen;
                        }
                        for (int i = 0; i < nsus; i++) {
                            const int4 sp = lst[i];
                            if (owners && !(sp.w != INT_MAX && sp.w > sp.z)) continue; // (uniform)
#pragma unroll
                            for (int u = 0; u < 8; u++) {
                                if (ok[u] && e[u].x == sp.x && ps[u] != sp.z) {
                                    if (!owners) atomicMin(&lst[i].w, ps[u]);
                                    else {
                                        bool plain = true; // (not a suspect itself)
                                        for (int j = 0; j < nsus; j++) plain = plain && lst[j].z != ps[u];
                                        if (plain) stream_fix_push(mrow, fix, FCAP, sp.z, ps[u], stream_mul<SMALL>(F, nm[u], e[u].y));
                                    }
                                }
                            }
                        }
                    }
                };
                scan(false);
                if (!WAVE_ROW) lds_barrier();
                bool any_owner = false;
                for (int i = 0; i < nsus; i++) {
                    const int4 sp = lst[i];
                    if (sp.w == INT_MAX) continue;
                    if (sp.w < sp.z) { if ((WAVE_ROW || wave == 0) && lane == 0) stream_fix_push(mrow, fix, FCAP, sp.w, sp.z, sp.y); }
                    else any_owner = true;
                }
                if (any_owner) { // (the same on every wave of the row: they read the same words)
                    scan(true);
                    if (!WAVE_ROW) lds_barrier();
                }
            }
        }
        // the lists are wave 0's business from here on (the other waves go on to reset the tables: nothing below touches those)
        if (WAVE_ROW || wave == 0) {
            bool redo = false;
            const int nlist = __builtin_amdgcn_readfirstlane(((lds_vint *)mrow)[1]); // (an LDS-qualified read: a generic one drains vmcnt too)
            redo = nlist > SLCAP;
            if (!BLOOM && nlist != 0 && !redo) {
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
                    if (owner >= 0) stream_fix_push(mrow, fix, FCAP, owner, me.z, me.y);
                }
            }
            const int nfix = SCATTER_DBG(a, 4) ? 0 : __builtin_amdgcn_readfirstlane(((lds_vint *)mrow)[0]);
            redo = redo || nfix > FCAP;
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
                    a.Slen[t_cur] = E; // duplicates are merged by k_stream_fix, which corrects the length then
                    a.Slead[t_cur] = E > 0 ? lead_out : INT_MAX;
                    if (nfix) a.fixcnt[t_cur] = nfix; // (zero from the binning pass otherwise)
                }
                c_nnz += (u64d)E;
                c_rows += E > 0;
                c_ent += (u64d)E; // entries streamed: the own entries and the runs
                c_seg += 1 + (u64d)ll;
            }
        }
        STAMP(3); // end of the row
        // ---- reset: the tables, and the other parity's words (nobody reads them any more: every wave is past its barrier)
        for (int s = rtid * 4; s < StreamTabs<LOGT>::WORDS; s += TPR * 4) *(int4 *)(tb.t1 + s) = make_int4(0, 0, 0, 0);
        if (WAVE_ROW) {
            if (lane < 2) mrow[lane] = 0;
            __builtin_amdgcn_wave_barrier();
        } else {
            par ^= 1;
            if (rtid < 8) misc[par * 8 + rtid] = rtid >= 4 ? INT_MAX : 0;
            lds_barrier();
        }
        STAMP(4); // table reset + barrier
        // ---- rotate the pipeline
        d = dn;
        dn = dnn;
        dnn = stream_desc_unpack(d3_reg);
        rec = rec_n; rec_n = rec_nn;
        STAMP(5); // rotation: waits for the descriptor / records / own entries requested at the top
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
