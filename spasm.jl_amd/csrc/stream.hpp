// stream.hpp -- SCATTER without accumulators: the Schur row is written while its entries stream by.
//
// Same contract as k_scatter (kernels.hpp): x_a = B[k]_N - x_b * U_PN, the non-pivot part of the solution of
// x * U = B[k] (reference src/SpaSM.jl:694-713), trip count of the reference's scatter loop (:619-620).
//
// On sparse matrices nearly every entry of a Schur row lands on a column of its own: with ~600 entries thrown on ~10^6
// columns a row sees 0.2 collisions.  The hash-table kernel pays for the general case on every entry (CAS + atomic add in
// LDS, then a sweep over all slots that reduces, compacts and stores).  Here an entry's value IS its final value unless a
// second entry shows up on the same column, so:
//   * every entry of the row's stream (its own entries on non-pivot columns, then the non-pivot parts of the applied pivot
//     rows, in record order) has a fixed POSITION known before anything is loaded: the combine kernel stores the running
//     total of npn in each multiplier record;
//   * the entry is multiplied, reduced to its balanced residue and stored straight to S[row][position] from registers;
//   * the LDS table only detects duplicates: one 64-bit CAS per entry on {column, position}.  An entry that finds its
//     column already there goes to the row's fix-up list {owner's position, own position, value};
//   * after the last entry the (rare) fix-ups are applied in global memory: the value is added to the owner's entry with a
//     64-bit compare-and-swap, the duplicate's position becomes a hole, and holes are filled with the entries at the end
//     of the row (the order of a row's entries carries no meaning, reference src/SpaSM.jl:1017-1020).
// No accumulate, no sweep: LDS traffic per entry is one CAS, VALU work about a third of the hash kernel's.
// Rows with more duplicates than the fix-up list holds (structured matrices) are handed back to the hash-table kernel
// of the same size class through its row list.
#pragma once
#include "kernels.hpp"

#define EMPTY64 (~0ull)

struct StreamArgs {
    const int *class_count;    // rows in this class
    const RowDesc *desc;       // their descriptors
    const int2 *ent;
    const int *qinv_r;
    const int2 *UPN;
    const int4 *Lpool;         // {stream position, multiplier, offset in UPN, npn}
    int2 *Sent;
    int *Slen;
    int *Slead;
    RoundCounters *ctr;
    int cls;                   // index of this class for the per-class counters (NSTREAM0 + size class)
    int *redo_count;           // the hash-table class of the same size: rows this kernel gives up on are appended there
    RowDesc *redo_desc;
    int dbg;                   // TIMING ABLATIONS ONLY (diagnostic builds, env SPASM_DBG; results are wrong when non-zero):
                               // 1 = no Schur stores, 2 = no duplicate check (no LDS traffic), 4 = fix-ups ignored, 8 = no pivot-row loads,
                               // 32 = 32-bit CAS on the column only
    ZpField F;
};

constexpr int SRCAP = 192; // entries of a wave's retry list = the worst case of one batch of 3 entries per lane

// fix-up entries a row may collect before it is handed to the hash-table kernel
__host__ __device__ constexpr int stream_fcap(int logt) { return logt <= 11 ? 64 : (1 << logt) / 32; }

__host__ __device__ constexpr size_t stream_lds_bytes(int logt, int tpr, int wpb)
{
    const size_t row = ((size_t)8 << logt) + 64 + (size_t)stream_fcap(logt) * 8; // table, misc, fix-up list
    const size_t retry = (size_t)SRCAP * 16;                                      // per wave
    return tpr == 64 ? (row + retry) * (size_t)wpb : row + retry * (size_t)wpb;
}

struct StreamRetry {
    int4 *buf;
    int cnt; // wave-uniform
    __device__ __forceinline__ void bind(unsigned char *p) { buf = (int4 *)p; cnt = 0; }
    // prim = 0: a clamped copy of an entry (it must find the entry in the table, or put it there, but never report a duplicate)
    __device__ __forceinline__ void put(int i, int c, int v, int pos, unsigned h, int prim) { buf[i] = make_int4(c, v, (int)(((unsigned)pos << 14) | h), prim); }
    __device__ __forceinline__ void get(int i, int &c, int &v, int &pos, unsigned &h, int &prim) const
    {
        const int4 e = buf[i];
        c = e.x; v = e.y; pos = (int)((unsigned)e.z >> 14); h = (unsigned)e.z & 0x3fffu; prim = e.w;
    }
};

// the row's fix-up list: {owner position << 14 | own position, value}; positions are below 2^14 (largest class: 10240)
__device__ __forceinline__ void stream_fix_push(int *s_nfix, int2 *fix, int fcap, int owner_pos, int pos, int v)
{
    const int i = atomicAdd(s_nfix, 1);
    if (i < fcap) fix[i] = make_int2((int)(((unsigned)owner_pos << 14) | (unsigned)pos), v);
}

// a table slot: {column, byte offset of the entry in the Schur row}
__device__ __forceinline__ u64d stream_pack(int c, int off) { return ((u64d)(unsigned)off << 32) | (u64d)(unsigned)c; }

// slot of a column: multiplicative hash on full-rate 24-bit multiplies (v_mul_lo_u32 is quarter rate); columns that differ
// only above bit 23 share a slot, which costs probes, never correctness
template <int LOGT> __device__ __forceinline__ unsigned stream_hash(int c) { return __umul24((unsigned)c, 0x9E3779u); }
template <int LOGT> __device__ __forceinline__ unsigned stream_slot(unsigned x) { return x >> (32 - LOGT); }
template <int LOGT> __device__ __forceinline__ unsigned stream_step(unsigned x) { return ((x >> 5) & ((1u << LOGT) - 1)) | 1u; }

// drain the wave's retry list: one entry per lane, probing on from where its first probe left off
template <int LOGT>
__device__ __forceinline__ void stream_drain(u64d *tab, StreamRetry &rl, int *s_nfix, int2 *fix, RoundCounters *ctr)
{
    constexpr unsigned T = 1u << LOGT;
    constexpr int FCAP = stream_fcap(LOGT);
    const int lane = threadIdx.x & 63;
    for (int b = 0; b < rl.cnt; b += 64) {
        bool pending = b + lane < rl.cnt;
        int c = 0, v = 0, pos = 0, prim = 0;
        unsigned h = 0, st = 1;
        if (pending) {
            rl.get(b + lane, c, v, pos, h, prim);
            st = stream_step<LOGT>(stream_hash<LOGT>(c));
        }
        const u64d want = stream_pack(c, pos << 3);
        for (unsigned round = 0; round < T && __ballot(pending) != 0; round++) {
            if (pending) {
                const u64d old = atomicCAS(&tab[h], EMPTY64, want);
                if (old == EMPTY64 || old == want) pending = false; // (met itself: a lane past the end of a pivot row repeats its last entry)
                else if ((int)(unsigned)old == c) {
                    if (prim) stream_fix_push(s_nfix, fix, FCAP, (int)(old >> 35), pos, v);
                    pending = false;
                } else h = (h + st) & (T - 1);
            }
        }
        if (pending) atomicAdd(&ctr_shard(ctr)->scatter_overflow, 1);
    }
    rl.cnt = 0;
}

// one entry of one lane, probing until it is in (safe under divergence: no wave-level bookkeeping)
template <int LOGT>
__device__ __forceinline__ void stream_add_1(u64d *tab, int *s_nfix, int2 *fix, int c, int v, int pos, RoundCounters *ctr)
{
    constexpr unsigned T = 1u << LOGT;
    constexpr int FCAP = stream_fcap(LOGT);
    const unsigned x = stream_hash<LOGT>(c);
    unsigned h = stream_slot<LOGT>(x);
    const unsigned st = stream_step<LOGT>(x);
    const u64d want = stream_pack(c, pos << 3);
    for (unsigned round = 0; round < T; round++) {
        const u64d old = atomicCAS(&tab[h], EMPTY64, want);
        if (old == EMPTY64 || old == want) return;
        if ((int)(unsigned)old == c) { stream_fix_push(s_nfix, fix, FCAP, (int)(old >> 35), pos, v); return; }
        h = (h + st) & (T - 1);
    }
    atomicAdd(&ctr_shard(ctr)->scatter_overflow, 1);
}

// multiplier * entry as THE balanced residue (it is stored as it stands)
template <bool SMALL> __device__ __forceinline__ int stream_mul(const ZpField &F, int nm, int y);
template <> __device__ __forceinline__ int stream_mul<true>(const ZpField &F, int nm, int y) { return acc_reduce_short<true>(F, __mul24(nm, y)); }
template <> __device__ __forceinline__ int stream_mul<false>(const ZpField &F, int nm, int y) { return zp_mul(F, nm, y); }

__device__ __forceinline__ u64d stream_load_fresh(const u64d *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ------------------------------------------------------------------------------------------------
// The rare path, run by ONE wave once every store of the row has left its wave (s_waitcnt vmcnt(0) + barrier): apply the
// fix-ups in global memory, then fill the holes from the end of the row.  `scratch` is the row's table memory (its
// contents are dead by now): 4 arrays of 2 * FCAP ints.  Returns the row's length; `lead` is recomputed when an entry
// cancelled to zero (its column leaves the row).
// ------------------------------------------------------------------------------------------------
template <int FCAP>
__device__ __noinline__ int stream_fixup(const ZpField F, const int2 *fix, int nfix, int *scratch, u64d *row, int E, int *lead)
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

// What a batch of N probes left unresolved: an entry that met ANOTHER column goes to the wave's retry list, an entry that met
// its own column under another position is a duplicate and goes to the row's fix-up list.  (An entry that met itself --
// same column, same position: lanes past the end of a pivot row repeat its last entry -- is in.)  Wave-uniform call.
template <int LOGT, int N>
__device__ __forceinline__ void stream_resolve(u64d *tab, StreamRetry &rl, int *s_nfix, int2 *fix, const int (&c)[N], const int (&v)[N],
                                               const unsigned (&off)[N], const u64d (&old)[N], const bool (&bad)[N], const bool (&prim)[N],
                                               RoundCounters *ctr)
{
    // prim[j]: this lane holds the entry itself, not a clamped copy of it -- a duplicate column is reported once
    constexpr int FCAP = stream_fcap(LOGT);
    constexpr unsigned T = 1u << LOGT;
    u64d fm[N];
    int nfail = 0;
    bool anysame = false;
#pragma unroll
    for (int j = 0; j < N; j++) {
        const bool same = bad[j] && (int)(unsigned)old[j] == c[j];
        anysame |= same && prim[j];
        fm[j] = __ballot(bad[j] && !same);
        nfail += __popcll(fm[j]);
    }
    if (__ballot(anysame) != 0) {
#pragma unroll
        for (int j = 0; j < N; j++)
            if (prim[j] && bad[j] && (int)(unsigned)old[j] == c[j]) stream_fix_push(s_nfix, fix, FCAP, (int)(old[j] >> 35), (int)(off[j] >> 3), v[j]);
    }
    if (nfail == 0) return;
    if (rl.cnt + nfail > SRCAP) stream_drain<LOGT>(tab, rl, s_nfix, fix, ctr);
    int at = rl.cnt;
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (fm[j] & (1ull << (threadIdx.x & 63))) {
            const unsigned x = stream_hash<LOGT>(c[j]);
            rl.put(at + __popcll(fm[j] & lanemask_lt()), c[j], v[j], (int)(off[j] >> 3), (stream_slot<LOGT>(x) + stream_step<LOGT>(x)) & (T - 1), prim[j] ? 1 : 0);
        }
        at += __popcll(fm[j]);
    }
    rl.cnt = at;
}

// TPR = threads cooperating on one row: 64 (a wave per row, WPB independent rows per workgroup, no barriers) or WPB * 64
// MAXR = rounds of pivot rows whose loads are all issued before any of them is used; round r hands pivot row gg + r * NG
//        of the row's record list to the 8-lane group gg
// The common path carries NO per-entry predicate: a lane past the end of its pivot row holds a copy of the row's last entry
// (clamped index) and simply repeats that entry's store and its CAS -- same bytes to the same address, and a CAS that finds
// {same column, same position} counts as "in".  Groups past the end of the record list repeat the last record the same way.
template <int LOGT, int TPR, int WPB, int MAXR, bool SMALL, int MINW>
__global__ __launch_bounds__(WPB * 64, MINW) void k_stream(StreamArgs a)
{
    constexpr bool WAVE_ROW = (TPR == 64);
    static_assert(WAVE_ROW || TPR == WPB * 64, "a row is owned by one wave or by the whole workgroup");
    constexpr int T = 1 << LOGT;
    constexpr int G = 8;
    constexpr int NG = TPR / G;
    constexpr int FCAP = stream_fcap(LOGT);
    constexpr size_t TABB = (size_t)T * 8, MISCB = 64, FIXB = (size_t)FCAP * 8, RB = (size_t)SRCAP * 16;
    constexpr size_t SLOT = TABB + MISCB + FIXB + RB; // wave-per-row: everything of a row
    static_assert((size_t)8 * FCAP * 4 <= TABB, "the fix-up scratch lives in the table");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rtid = WAVE_ROW ? lane : tid;
    unsigned char *base = s_raw + (WAVE_ROW ? (size_t)wave * SLOT : 0);
    u64d *tab = (u64d *)base;
    // misc[par * 8 + 0] = fix-ups pushed, [par * 8 + 4 + w] = leftmost column seen by wave w (block-per-row); two parities so
    // that a row's words can be reset while the waves are still reading the previous row's
    int *misc = (int *)(base + TABB);
    int2 *fix = (int2 *)(base + TABB + MISCB);
    StreamRetry rl;
    rl.bind(base + TABB + MISCB + FIXB + (WAVE_ROW ? 0 : (size_t)wave * RB));
    const int gg = rtid / G, gl = rtid % G;
    const ZpField F = a.F;

    const int count = *a.class_count;
    if ((WAVE_ROW ? (int)blockIdx.x * WPB : (int)blockIdx.x) >= count) return;
    for (int s = rtid * 2; s < T; s += TPR * 2) *(int4 *)(tab + s) = make_int4(-1, -1, -1, -1);
    if (rtid < 16) misc[rtid] = (rtid & 7) >= 4 ? INT_MAX : 0;
    __syncthreads();

    const int first = WAVE_ROW ? (int)blockIdx.x * WPB + wave : (int)blockIdx.x;
    const int stride = WAVE_ROW ? (int)gridDim.x * WPB : (int)gridDim.x;
    u64d c_nnz = 0, c_ent = 0, c_seg = 0;
    int c_rows = 0, c_fix = 0, c_redo = 0;
    int par = 0;

    // ---- pipeline registers: descriptor two rows ahead, first own entries + records one row ahead
    RowDesc d, dn;
    d.ent_start = d.l_start = d.s_start = 0; d.len = d.llen = d.t = d.bound = 0; d.pmask = -1;
    dn = d;
    int2 own = make_int2(0, 0);
    int4 rec[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; r++) rec[r] = make_int4(0, 0, 0, 0);
    if (first < count) {
        d = desc_unpack(desc_load(&a.desc[first]));
        own = a.ent[d.ent_start + min(lane, max(d.len - 1, 0))];
        const int lastrec = max(d.llen - 1, 0);
#pragma unroll
        for (int r = 0; r < MAXR; r++) rec[r] = a.Lpool[d.l_start + min(gg + r * NG, lastrec)];
    }
    if (first + stride < count) dn = desc_unpack(desc_load(&a.desc[first + stride]));

    for (int w = first; w < count; w += stride) {
        const DescRegs dnn_regs = desc_load(&a.desc[min(w + 2 * stride, count - 1)]);
        const int ln = d.len, ll = d.llen;
        int *const s_nfix = misc + par * 8;
        // ---- every load of the row: qinv of the own entry (first 64, wave 0), entries of the pivot rows (clamped indices)
        int q_own = 0;
        if (d.pmask >= 0) q_own = ((d.pmask >> lane) & 1) ? 0 : -1;
        else if (lane < ln) q_own = a.qinv_r[own.x];
        int2 u[MAXR][3];
        bool gok[MAXR];         // the group's record has entries (a pivot row may consist of its pivot alone)
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            const int np = rec[r].w;
            gok[r] = np > 0 && rec[r].y != 0;
            const int2 *up = a.UPN + (unsigned)rec[r].z;
            const int last = max(np - 1, 0);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int idx = min(gl + j * G, last);
                u[r][j] = make_int2(idx + r * 1024 + gg * 32, 1);
                if (!SCATTER_DBG(a, 8)) u[r][j] = up[idx];
            }
        }
        // ---- stage-1 data of the NEXT row (its first own entries, its records), behind this row's loads: they have long
        // arrived when this row is done.  Unconditional and clamped (see k_scatter): after the last row `dn` is the last
        // descriptor again.
        int2 own_n;
        int4 rec_n[MAXR];
        {
            own_n = a.ent[dn.ent_start + min(lane, max(dn.len - 1, 0))];
            const int lastrec = max(dn.llen - 1, 0);
#pragma unroll
            for (int r = 0; r < MAXR; r++) rec_n[r] = a.Lpool[dn.l_start + min(gg + r * NG, lastrec)];
        }
        unsigned char *const rowp = (unsigned char *)(a.Sent + d.s_start);
        int mylead = INT_MAX;
        int nN = 0;
        // ---- the row's own entries on non-pivot columns: stream positions 0 .. nN-1, one wave (their rank is a ballot)
        if (WAVE_ROW || wave == 0) {
            {
                const bool nonpiv = lane < ln && q_own < 0;
                const u64d m = __ballot(nonpiv);
                const unsigned off = (unsigned)__popcll(m & lanemask_lt()) << 3;
                u64d old = EMPTY64;
                const u64d want = stream_pack(own.x, (int)off);
                if (nonpiv) {
                    if (!SCATTER_DBG(a, 1)) __builtin_nontemporal_store(((long long)(unsigned)own.y << 32) | (unsigned)own.x, (long long *)(rowp + off));
                    mylead = min(mylead, own.x);
                    if (!SCATTER_DBG(a, 2)) old = atomicCAS(&tab[stream_slot<LOGT>(stream_hash<LOGT>(own.x))], EMPTY64, want);
                }
                const bool bad[1] = {old != EMPTY64 && old != want};
                if (__ballot(bad[0]) != 0) {
                    const int oc[1] = {own.x}, ov[1] = {own.y};
                    const unsigned oo[1] = {off};
                    const u64d ol[1] = {old};
                    const bool pr[1] = {true};
                    stream_resolve<LOGT, 1>(tab, rl, s_nfix, fix, oc, ov, oo, ol, bad, pr, a.ctr);
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
                const unsigned off = (unsigned)(nN + __popcll(m & lanemask_lt())) << 3;
                u64d old = EMPTY64;
                const u64d want = stream_pack(e.x, (int)off);
                if (nonpiv) {
                    if (!SCATTER_DBG(a, 1)) __builtin_nontemporal_store(((long long)(unsigned)e.y << 32) | (unsigned)e.x, (long long *)(rowp + off));
                    mylead = min(mylead, e.x);
                    if (!SCATTER_DBG(a, 2)) old = atomicCAS(&tab[stream_slot<LOGT>(stream_hash<LOGT>(e.x))], EMPTY64, want);
                }
                const bool bad[1] = {old != EMPTY64 && old != want};
                if (__ballot(bad[0]) != 0) {
                    const int oc[1] = {e.x}, ov[1] = {e.y};
                    const unsigned oo[1] = {off};
                    const u64d ol[1] = {old};
                    const bool pr[1] = {true};
                    stream_resolve<LOGT, 1>(tab, rl, s_nfix, fix, oc, ov, oo, ol, bad, pr, a.ctr);
                }
                nN += __popcll(m);
            }
        }
        // ---- the pivot rows: multiply, store at the entry's stream position, one CAS for the duplicate check
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            if (r * NG < ll) { // scalar: there are records for this round
                const int nm = -rec[r].y;
                const bool allok = __ballot(!gok[r]) == 0;
                int bc[3], bv[3];
                unsigned uoff[3]; // byte offset of the entry in the Schur row
                u64d old[3], want[3];
                bool bad[3];
                const int last = max(rec[r].w - 1, 0);
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    bc[j] = u[r][j].x;
                    bv[j] = stream_mul<SMALL>(F, nm, u[r][j].y);
                    uoff[j] = (unsigned)(rec[r].x + min(gl + j * G, last)) << 3;
                    want[j] = stream_pack(bc[j], (int)uoff[j]);
                    old[j] = EMPTY64;
                }
                if (allok) {
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        if (!SCATTER_DBG(a, 2)) old[j] = atomicCAS(&tab[stream_slot<LOGT>(stream_hash<LOGT>(bc[j]))], EMPTY64, want[j]);
                        if (!SCATTER_DBG(a, 1)) __builtin_nontemporal_store(((long long)(unsigned)bv[j] << 32) | (unsigned)bc[j], (long long *)(rowp + uoff[j]));
                    }
                    mylead = min(mylead, min(bc[0], min(bc[1], bc[2])));
                } else if (gok[r]) { // a group without entries in this round: the others go on under their own predicate
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        old[j] = atomicCAS(&tab[stream_slot<LOGT>(stream_hash<LOGT>(bc[j]))], EMPTY64, want[j]);
                        __builtin_nontemporal_store(((long long)(unsigned)bv[j] << 32) | (unsigned)bc[j], (long long *)(rowp + uoff[j]));
                    }
                    mylead = min(mylead, min(bc[0], min(bc[1], bc[2])));
                }
#pragma unroll
                for (int j = 0; j < 3; j++) bad[j] = old[j] != EMPTY64 && old[j] != want[j];
                if (__ballot(bad[0] | bad[1] | bad[2]) != 0) {
                    const bool gprim = gg + r * NG < ll;
                    const bool pr[3] = {gprim && gl <= last, gprim && gl + G <= last, gprim && gl + 2 * G <= last};
                    stream_resolve<LOGT, 3>(tab, rl, s_nfix, fix, bc, bv, uoff, old, bad, pr, a.ctr);
                }
                if (__ballot(gok[r] && rec[r].w > 3 * G) != 0) { // pivot rows longer than 24 entries
                    const int2 *up = a.UPN + (unsigned)rec[r].z;
                    const int np = gok[r] ? rec[r].w : 0;
                    for (int k = gl + 3 * G; k < np; k += G) {
                        const int2 uu = up[k];
                        const int vv = stream_mul<SMALL>(F, nm, uu.y);
                        const int pp = rec[r].x + k;
                        __builtin_nontemporal_store(((long long)(unsigned)vv << 32) | (unsigned)uu.x, (long long *)(rowp + ((unsigned)pp << 3)));
                        mylead = min(mylead, uu.x);
                        stream_add_1<LOGT>(tab, s_nfix, fix, uu.x, vv, pp, a.ctr);
                    }
                }
            }
        }
        // ---- more pivot rows than NG * MAXR: one round at a time, per-lane probing (wave-uniform trip count)
        for (int e0 = MAXR * NG; e0 < ll; e0 += NG) {
            const int e = e0 + gg;
            int4 le = make_int4(0, 0, 0, 0);
            if (e < ll) le = a.Lpool[d.l_start + e];
            const int np = le.y != 0 ? le.w : 0;
            const int nm = -le.y;
            const int2 *up = a.UPN + (unsigned)le.z;
            for (int k = gl; k < np; k += G) {
                const int2 uu = up[k];
                const int vv = stream_mul<SMALL>(F, nm, uu.y);
                const int pp = le.x + k;
                __builtin_nontemporal_store(((long long)(unsigned)vv << 32) | (unsigned)uu.x, (long long *)(rowp + ((unsigned)pp << 3)));
                mylead = min(mylead, uu.x);
                stream_add_1<LOGT>(tab, s_nfix, fix, uu.x, vv, pp, a.ctr);
            }
        }
        stream_drain<LOGT>(tab, rl, s_nfix, fix, a.ctr);
        const int t_cur = d.t, E = d.bound, ln_cur = ln;
        // ---- end of the row: leftmost column, duplicates
        mylead = wave_min_i32(mylead);
        int lead_out = mylead;
        if (!WAVE_ROW) {
            if (lane == 0) misc[par * 8 + 4 + wave] = mylead;
            lds_barrier();
#pragma unroll
            for (int w2 = 0; w2 < WPB; w2++) lead_out = min(lead_out, misc[par * 8 + 4 + w2]);
        }
        const int nfix = SCATTER_DBG(a, 4) ? 0 : *(volatile int *)s_nfix;
        int n_out = E;
        bool redo = false;
        if (nfix != 0) { // uniform over the row's team
            redo = nfix > FCAP;
            if (!redo) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's stores of the row have left
                if (!WAVE_ROW) lds_barrier();
                if (WAVE_ROW || wave == 0) n_out = stream_fixup<FCAP>(F, fix, nfix, (int *)tab, (u64d *)rowp, E, &lead_out);
                if (!WAVE_ROW) lds_barrier(); // the scratch in the table is dead: it may be cleared
            }
        }
        if (rtid == 0) {
            if (redo) {
                // too many duplicate columns for the fix-up list: the hash-table kernel of this size class takes the row
                const int at = atomicAdd(a.redo_count, 1);
                a.redo_desc[at] = d;
                c_redo += 1;
            } else {
                a.Slen[t_cur] = n_out;
                a.Slead[t_cur] = n_out > 0 ? lead_out : INT_MAX;
                c_nnz += (u64d)n_out;
                c_rows += n_out > 0;
                // entries streamed: the own entries + the non-pivot parts of the applied pivot rows (= E - nN, the records of the
                // combine kernel all carry a multiplier); segments: the row + one per record
                c_ent += (u64d)ln_cur + (u64d)(E - nN);
                c_seg += 1 + (u64d)ll;
                c_fix += nfix;
            }
        }
        // ---- reset: the table, and the other parity's words (nobody reads them any more: every wave is past its barrier)
        for (int s = rtid * 2; s < T; s += TPR * 2) *(int4 *)(tab + s) = make_int4(-1, -1, -1, -1);
        if (WAVE_ROW) {
            if (lane == 0) *s_nfix = 0;
            __builtin_amdgcn_wave_barrier();
        } else {
            par ^= 1;
            if (rtid < 8) misc[par * 8 + rtid] = rtid >= 4 ? INT_MAX : 0;
            lds_barrier();
        }
        d = dn;
        dn = desc_unpack(dnn_regs);
        own = own_n;
#pragma unroll
        for (int r = 0; r < MAXR; r++) rec[r] = rec_n[r];
    }
    for (int o = 32; o > 0; o >>= 1) {
        c_ent += __shfl_xor(c_ent, o);
        c_seg += __shfl_xor(c_seg, o);
        c_nnz += __shfl_xor(c_nnz, o);
        c_rows += __shfl_xor(c_rows, o);
        c_fix += __shfl_xor(c_fix, o);
        c_redo += __shfl_xor(c_redo, o);
    }
    if (lane == 0) {
        if (c_nnz) atomicAdd(&ctr_shard(a.ctr)->nnz_out, c_nnz);
        if (c_rows) atomicAdd(&ctr_shard(a.ctr)->nonempty_out, c_rows);
        if (c_fix) atomicAdd(&ctr_shard(a.ctr)->stream_fix, c_fix);
        if (c_redo) atomicAdd(&ctr_shard(a.ctr)->stream_redo, c_redo);
        if (c_ent | c_seg) {
            atomicAdd(&ctr_shard(a.ctr)->class_ent[a.cls], c_ent);
            atomicAdd(&ctr_shard(a.ctr)->class_seg[a.cls], c_seg);
        }
    }
}
