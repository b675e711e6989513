// engine.hip -- round driver of the MI355X echelonization engine and its C ABI entry points.
//
// spasm_echelonize / spasm_kernel / spasm_transpose keep the signatures SpaSM.jl binds (reference
// src/SpaSM.jl:863, :879, :589): host CSR in, host LU / CSR out, caller owns the result.  Inside,
// the matrix is uploaded once, every round runs on the device (kernels.hpp), and only U, qinv and
// the rank come back.  There is NO CPU fallback: without a HIP device these entry points fail.
#include "common.hpp"
#include "kernels.hpp"
#include "stream.hpp"
#include "wlevel.hpp"
#include "fused.hpp"
#include "greedy.hpp"
#include "dense.hpp"
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <vector>
#include <memory>
#include <stdexcept>
#include <type_traits>
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <atomic>
#include <string>

namespace {

struct EngineError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            char _b[512];                                                                             \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            throw EngineError(_b);                                                                    \
        }                                                                                             \
    } while (0)

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept
    {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
    void alloc(size_t count)
    {
        release();
        if (count == 0) count = 1;
        const hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            size_t fr = 0, tot = 0;
            (void)hipMemGetInfo(&fr, &tot);
            (void)hipGetLastError();
            char b[256];
            snprintf(b, sizeof b, "device allocation of %.2f GiB failed: %s (%.1f of %.1f GiB free)", (double)(count * sizeof(T)) / 1073741824.0,
                     hipGetErrorString(e), (double)fr / 1073741824.0, (double)tot / 1073741824.0);
            throw EngineError(b);
        }
        n = count;
    }
    void ensure(size_t count) { if (count > n) alloc(count); }
    void zero(hipStream_t s) { if (p) HIPCHK(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
};

inline int cdiv(i64 a, i64 b) { return (int)((a + b - 1) / b); }

// device matrix: rows are slices (start,len) of ent; lead = leftmost column; orig = row of the input matrix
struct DevMat {
    int n = 0, m = 0;
    DevBuf<i64d> start;
    DevBuf<int> len, lead, orig;
    DevBuf<int2> ent;
};

struct Scanner {
    DevBuf<unsigned char> tmp;
    template <class T> void exclusive(const T *in, T *out, size_t n, hipStream_t s)
    {
        size_t bytes = 0;
        HIPCHK(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), s));
        tmp.ensure(bytes);
        HIPCHK(rocprim::exclusive_scan(tmp.p, bytes, in, out, T(0), n, rocprim::plus<T>(), s));
    }
};

// hash-table classes of the scatter kernel: log2(slots), threads per row, waves per workgroup,
// unrolled rounds of pivot rows, largest row bound (load factor <= 5/8)
struct ScatterClass { int logt, tpr, wpb, maxr; i64 cap; };
const ScatterClass kClasses[7] = {{8, 64, 4, 1, 160},  {9, 64, 4, 2, 320},   {10, 64, 4, 4, 640},  {11, 128, 2, 4, 1280},
                                  {12, 256, 4, 4, 2560}, {13, 256, 4, 5, 5120}, {14, 256, 4, 5, 10240}};
const int kNumHashClasses = 7;

inline size_t scatter_lds_bytes(const ScatterClass &c, bool small)
{
    const size_t retry = small ? RetryList<true>::BYTES : RetryList<false>::BYTES; // one list per wave
    const size_t table = ((size_t)1 << c.logt) * (small ? 8 : 12) + 32; // + s_misc
    return c.tpr == 64 ? (table + retry) * (size_t)c.wpb : table + retry * (size_t)c.wpb;
}

// function attributes (the opt-in to more than 64 KB of dynamic LDS) are per DEVICE: a process that drives several devices
// (spasm_amd_echelonize_multi) sets them on each
constexpr int kMaxDev = 64;
inline int current_device()
{
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    return dev & (kMaxDev - 1);
}

template <int LOGT, int TPR, int WPB, int MAXR, bool SMALL, int MINW = 1> void init_scatter(size_t lds)
{
    static bool attr_done[kMaxDev] = {false};
    bool &done = attr_done[current_device()];
    if (!done) {
        HIPCHK(hipFuncSetAttribute((const void *)k_scatter<LOGT, TPR, WPB, MAXR, SMALL, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        done = true;
    }
}
template <int LOGT, int TPR, int WPB, int MAXR, bool SMALL, int MINW = 1> void launch_scatter(const ScatterArgs &a, int grid, size_t lds, hipStream_t s)
{
    init_scatter<LOGT, TPR, WPB, MAXR, SMALL, MINW>(lds);
    if (grid <= 0) return; // (warm-up call: attributes only)
    hipLaunchKernelGGL((k_scatter<LOGT, TPR, WPB, MAXR, SMALL, MINW>), dim3(grid), dim3(WPB * 64), lds, s, a);
    HIPCHK(hipGetLastError());
}

template <bool SMALL> void launch_scatter_class(int cls, const ScatterArgs &a, int grid, size_t lds, hipStream_t s)
{
    switch (cls) {
    case 0: launch_scatter<8, 64, 4, 1, SMALL>(a, grid, lds, s); break;
    case 1: launch_scatter<9, 64, 4, 2, SMALL>(a, grid, lds, s); break;
    case 2: launch_scatter<10, 64, 4, 4, SMALL, 4>(a, grid, lds, s); break;   // 8 KiB tables: 19 rows per CU fit, keep the registers under 96
    case 3: launch_scatter<11, 128, 2, 4, SMALL, 4>(a, grid, lds, s); break;  // 16 KiB tables: 10 workgroups of 2 waves per CU
    case 4: launch_scatter<12, 256, 4, 4, SMALL, 4>(a, grid, lds, s); break;
    case 5: launch_scatter<13, 256, 4, 5, SMALL>(a, grid, lds, s); break;
    case 6: if (SMALL) launch_scatter<14, 256, 4, 5, true>(a, grid, lds, s); break;
    default: break;
    }
}

// grid = min(rows, resident workgroups): the kernels walk their row list with a grid stride, and a workgroup that is not resident
// from the start would do its share after everybody else (the LDS bound alone can be above what the registers admit)
template <int LOGT, int TPR, int WPB, int B, bool SMALL, int MINW = 1, int EPL = 1, int QX = 0, bool BLOOM = false> void launch_wstream(const StreamArgs &a, int nrows, int num_cu, size_t lds, hipStream_t s)
{
    static int per_cu_dev[kMaxDev] = {0};
    int &per_cu = per_cu_dev[current_device()];
    if (!per_cu) {
        HIPCHK(hipFuncSetAttribute((const void *)k_wstream<LOGT, TPR, WPB, B, SMALL, MINW, EPL, QX, BLOOM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nb = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_wstream<LOGT, TPR, WPB, B, SMALL, MINW, EPL, QX, BLOOM>, WPB * 64, lds));
        per_cu = std::max(nb, 1);
    }
    if (nrows <= 0) return; // (warm-up call: attributes and occupancy only)
    const int rows_per_block = TPR == 64 ? WPB : 1;
    const int grid = std::max(1, std::min((nrows + rows_per_block - 1) / rows_per_block, num_cu * per_cu));
    hipLaunchKernelGGL((k_wstream<LOGT, TPR, WPB, B, SMALL, MINW, EPL, QX, BLOOM>), dim3(grid), dim3(WPB * 64), lds, s, a);
    HIPCHK(hipGetLastError());
}

// chunks of the plan along W: 2^6 entries (a lane of the streaming kernel takes one entry per chunk) or 2^7 (two: one 16-byte load and,
// where the chunk is full, one 16-byte store per lane; half the per-chunk work).  SPASM_AMD_CHUNK = 64 | 128.
// Measured on config 3 (the rows of W average 170 entries): chunks of 128 are SLOWER, 3.75 ms per step against 3.33 -- a run of 170
// entries is three chunks of 64 (89 % of the lanes busy) or two of 128 (66 %), and a chunk of the two-entry kernel costs 1.7 x the
// instructions of a one-entry chunk: nothing is saved below runs of ~500 entries.  Default 64; the other stays for such matrices.
inline bool stream_bloom()
{
    const char *e = getenv("SPASM_AMD_BLOOM");
    return e && atoi(e) != 0;
}
inline int stream_chunk_log()
{
    const char *e = getenv("SPASM_AMD_CHUNK"); // (read per launch: a test sets it for one case)
    return e && atoi(e) == 128 ? 7 : 6;
}

// the streaming twins of the hash-table classes (row bounds up to 160 << c): first table of 256 << c words (at most 5/8 full); a
// wave per row up to 640 entries, then 2 and 4 waves
const int kNumStreamClasses = 7;
inline int stream_logt(int c) { return 8 + c; }
inline int stream_tpr(int c) { return c <= 2 ? 64 : (c == 3 ? 128 : 256); }
inline int stream_wpb(int c) { return c == 3 ? 2 : 4; }
template <bool SMALL> void launch_stream_class(int cls, const StreamArgs &a, int nrows, int num_cu, size_t lds, hipStream_t s, int chunk_log = 6)
{
    if (chunk_log == 7) {
        switch (cls) { // (ring slots D: chunks of 128 in flight per wave)
        case 0: launch_wstream<8, 64, 4, 2, SMALL, 1, 2>(a, nrows, num_cu, lds, s); break;
        case 1: launch_wstream<9, 64, 4, 4, SMALL, 1, 2>(a, nrows, num_cu, lds, s); break;
        case 2: launch_wstream<10, 64, 4, 4, SMALL, 1, 2>(a, nrows, num_cu, lds, s); break;
        case 3: launch_wstream<11, 128, 2, 4, SMALL, 1, 2>(a, nrows, num_cu, lds, s); break;
        case 4: launch_wstream<12, 256, 4, 4, SMALL, 1, 2>(a, nrows, num_cu, lds, s); break;
        case 5: launch_wstream<13, 256, 4, 4, SMALL, 1, 2>(a, nrows, num_cu, lds, s); break;
        case 6: launch_wstream<14, 256, 4, 4, SMALL, 1, 2>(a, nrows, num_cu, lds, s); break;
        default: break;
        }
        return;
    }
    if (stream_bloom()) { // the duplicate check as a filter (stream.hpp: BLOOM)
        switch (cls) {
        case 0: launch_wstream<8, 64, 4, 4, SMALL, 1, 1, 0, true>(a, nrows, num_cu, lds, s); break;
        case 1: launch_wstream<9, 64, 4, 8, SMALL, 1, 1, 0, true>(a, nrows, num_cu, lds, s); break;
        case 2: launch_wstream<10, 64, 4, 8, SMALL, 1, 1, 0, true>(a, nrows, num_cu, lds, s); break;
        case 3: launch_wstream<11, 128, 2, 8, SMALL, 1, 1, 0, true>(a, nrows, num_cu, lds, s); break;
        case 4: launch_wstream<12, 256, 4, 8, SMALL, 1, 1, 0, true>(a, nrows, num_cu, lds, s); break;
        case 5: launch_wstream<13, 256, 4, 8, SMALL, 1, 1, 0, true>(a, nrows, num_cu, lds, s); break;
        case 6: launch_wstream<14, 256, 4, 8, SMALL, 1, 1, 0, true>(a, nrows, num_cu, lds, s); break;
        default: break;
        }
        return;
    }
    // (groups of two chunks instead of four for the classes whose rows have three to six chunks, so that fewer of them take the
    // one-at-a-time path of a group that is not full: no difference, 0.1085 / 0.208 ms against 0.1083 / 0.2043; template parameter QX)
    switch (cls) {
    case 0: launch_wstream<8, 64, 4, 4, SMALL>(a, nrows, num_cu, lds, s); break;
    case 1: launch_wstream<9, 64, 4, 8, SMALL>(a, nrows, num_cu, lds, s); break;
    case 2: launch_wstream<10, 64, 4, 8, SMALL>(a, nrows, num_cu, lds, s); break;
    case 3: launch_wstream<11, 128, 2, 8, SMALL>(a, nrows, num_cu, lds, s); break;
    case 4: launch_wstream<12, 256, 4, 8, SMALL>(a, nrows, num_cu, lds, s); break;
    case 5: launch_wstream<13, 256, 4, 8, SMALL>(a, nrows, num_cu, lds, s); break;
    case 6: launch_wstream<14, 256, 4, 8, SMALL>(a, nrows, num_cu, lds, s); break;
    default: break;
    }
}

// ------------------------------------------------------------------------------------------------
// One echelonization round on one device.
// ------------------------------------------------------------------------------------------------
struct Round {
    ZpField F;
    int m = 0;
    int num_cu = 256;
    hipStream_t stream = nullptr;
    Scanner scan;

    // election
    DevBuf<u64d> best;
    DevBuf<int> colflag, colscan, qinv_r, pivrow, pivcol, is_piv, rowflag, rowscan, np_rows;
    int npiv = 0, nnp = 0;
    // U of this round
    DevBuf<i64d> ulen, uoff;
    i64 utotal = 0;
    DevBuf<int2> Ufull, UPP, UPN;
    DevBuf<UHdr> uhdr;
    // Uinv = (I + U_PP)^-1, one row per pivot (built once per round when the reach is short)
    bool use_uinv = false;
    DevMat E;                       // the unit rows e_r
    DevBuf<int2> UinvPool;
    DevBuf<i64d> UinvStart, ubound;
    DevBuf<int> UinvLen;
    DevBuf<int4> colinfo;           // per column: pivot index + location of its Uinv row
    DevBuf<unsigned> pbits;         // per column: one bit, set for pivot columns
    DevBuf<i64d> rstart;            // per processed row slot: start / length of its own entries
    DevBuf<int> rlen;
    i64 uinv_nnz = 0;
    // solve
    DevBuf<int4> Lpool;
    DevBuf<int> Lidx;               // pivot index of every record (only when want_idx: kernel basis, triangular solve)
    bool want_idx = false;
    DevBuf<int> sflag;              // per row slot: the streaming scatter may take the row
    DevBuf<int2> fixbuf;            // per row slot: SFIX duplicates found by the streaming kernels, merged by k_stream_fix
    DevBuf<int> fixcnt;
    bool use_stream = true;         // SPASM_AMD_STREAM=0 turns W and the streaming scatter off
    int chunk_log = 6;              // entries per chunk record of the last plan along W (2^6 or 2^7: stream_chunk_log)
    // W = -(I + U_PP)^-1 U_PN (stream.hpp), built level by level of the pivot graph (wlevel.hpp).  One buffer, addressed by 32-bit
    // offsets: [U_PN | the own non-pivot entries of the rows the plan kernel takes | the rows of W]
    bool use_w = false;
    bool force_lists = false;       // keep to the multiplier lists even when W is there (exact trip counters: they count list entries)
    DevBuf<int4> wcol;              // per pivot column: {pivot index, length, offset} of its row of W
    DevBuf<int2> wrow;              // the same per pivot index: {offset, length}
    DevBuf<int> wreject_list;
    i64 wtotal = 0;                 // entries of W's region handed out by the last build
    i64 wcap = 0;                   // entries of W's region
    i64 w_entries = 0, w_long_rows = 0; // entries of W, rows the workgroup kernel built (statistics of the first build)
    i64 own_base = 0, wbase = 0;    // where the own entries / the rows of W start in the buffer
    i64 own_total = 0;              // entries of the part of the buffer that takes the rows' own non-pivot entries
    DevBuf<u64d> own_ctr;           // its bump counters
    DevBuf<u64d> wstate, wblk;      // cursor + statistics of the build; the blocks its waves carve rows from
    DevBuf<int> wmid_list, wbig_list, wbig_count; // rows the wave kernel leaves to the workgroup kernels; counts [2 * level + {0: medium, 1: large}]
    std::vector<int> wbig_at;       // rows the workgroup kernels built per level and tier (statistics: SPASM_AMD_WDEBUG, plans)
    bool want_w_row_stats = false;  // plans read them (spasm_amd_round_stats::w_long_rows)
    DevBuf<WLevRec> lev_recs;       // the pivot rows in level order, as the level kernels read them
    DevBuf<unsigned> lev0_sz, lev0_off; // level 0: lengths rounded up to 16, and their prefix sums = the rows' places in W
    DevBuf<u64d> lev0_ent;
    i64 lev0_total = 0, lev0_entries = 0;
    int wave_per_cu_blocks = 0;     // resident workgroups of k_wlevel_wave per CU
    // levels of the pivot graph: level L = lev_order[lev_start[L] .. lev_start[L + 1])
    DevBuf<int> lev, lev_keys, lev_iota, lev_order, lev_start_d, lev_flag;
    DevBuf<unsigned char> sort_tmp;
    std::vector<int> lev_start;
    int depth = -1;                 // deepest level; -1: no levels (not computed, or the graph is deeper than WMAXLEV)
    bool quiet_rejects = false;     // a plan whose dry run saw the plan kernel reject no row: later runs skip the chain-solve launches
    DevBuf<u64d> pool_ctr;          // NPOOL sharded bump counters
    int npool_active = NPOOL;       // regions in use by the current solve
    u64d region_cap = 0;
    DevBuf<i64d> Lstart, bound, sstart;
    DevBuf<int> Llen, overflow_list, overflow2_list, fail_list;
    DevBuf<long long> pmask;        // per row slot: which own entries sit on pivot columns (solve -> scatter)
    // last-resort solve (dense vector + bitmap over the pivot indices, per workgroup)
    int big_blocks = 0, big_npiv = -1;
    DevBuf<int> xdense;
    DevBuf<unsigned> bitmap;
    DevBuf<int4> bigscratch;
    DevBuf<RoundCounters> ctr;
    RoundCounters hctr;
    // scatter
    DevBuf<int> class_count, class_list;
    DevBuf<RowDesc> class_desc;
    // last-resort scatter (dense accumulator + bitmap + touched list over the columns, per workgroup)
    int bigsc_blocks = 0, bigsc_m = -1;
    DevBuf<long long> sc_xdense;
    DevBuf<unsigned> sc_bitmap;
    DevBuf<int> sc_touched;
    DevBuf<u64d> stamps;            // diagnostic build only
    DevMat S;
    i64 s_capacity = 0;      // entries S.ent can hold (sum of bounds at the time it was sized)
    i64 s_ent_base = 0;      // where the rows of run_scatter start in S.ent (behind the rows of the fused step, when it left any to the general path)
    // ---- the fused Schur step (fused.hpp): plan + stream of a row in one kernel, S written compactly from one cursor
    bool use_fused = false;  // SPASM_AMD_FUSED=1: on.  Off by default: as measured in round 4 it does not beat k_wplan + k_bin + the streaming classes
    DevBuf<int4> rinfo;      // per row slot: {start, length, originating row}
    DevBuf<unsigned> fz_work;
    DevBuf<u64d> fz_cursor;
    DevBuf<int> fz_counts;   // [0] rows left to the launch with large tables, [1] rows left to the general path
    DevBuf<RoundCounters> fz_ctr; // statistics of the fused kernels (the general path clears its own when it runs behind them)
    RoundCounters hfz;       // .. on the host
    bool last_fused = false; // the last Schur step went through run_fused
    DevBuf<int> fz_long_list, fz_gen_list, fz_rej_list, fz_rej_rows, fz_flag, fz_pos;
    Scanner fz_scan;         // (its own scratch: the general path scans on another stream meanwhile)
    DevBuf<int4> fz_rec;     // per row slot: the two records of the fused kernels
    DevBuf<FusedRare> fz_rare;
    const int *fz_rare_lists[2] = {nullptr, nullptr}; // the list buffers fz_rare names (they may be reallocated for a larger round)
    DevBuf<i64d> fb_start;   // the general path's own row arrays while it works for the fused step
    DevBuf<int> fb_len, fb_lead, fb_orig;
    i64 fz_used = 0;         // entries of S handed out by the last fused step (incl. what its waves left of their blocks)
    int fz_nrej = 0, fz_nlong = 0;
    int fz_rows = 0;         // rows the fused kernels took
    hipEvent_t ev_fz[3] = {nullptr, nullptr, nullptr}; // around the fused kernels (class_timing)
    i64 s_total_bound = 0;
    int free_cols = 0;
    // timing
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t side = nullptr;     // the few rows of the largest table classes run beside the others
    hipStream_t twin_s[3] = {nullptr, nullptr, nullptr}; // with `side`: the hash-table twins of the streaming classes, four abreast
    hipEvent_t twin_ev[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t lane_s[4] = {nullptr, nullptr, nullptr, nullptr}; // further lanes of the streaming classes (lane 0 = the round's stream)
    hipEvent_t lane_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_cls_done[NHASHMAX] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_cls[NCLASS + 1];  // one in front of every scatter launch (class_timing) and one behind the last
    int launch_cls[NCLASS];         // class id of launch i
    int nlaunch = 0;
    int hclass_count[NCLASS];
    int nhash_used = 0;
    bool class_timing = true;       // record an event pair around every scatter class (costs a few microseconds of gap each)

    Round()
    {
        memset(&hctr, 0, sizeof hctr);
        for (auto &e : ev) HIPCHK(hipEventCreate(&e));
        for (auto &e : ev_cls) { e = nullptr; HIPCHK(hipEventCreate(&e)); }
        memset(hclass_count, 0, sizeof hclass_count);
        memset(launch_cls, 0, sizeof launch_cls);
        if (const char *e = getenv("SPASM_AMD_STREAM")) use_stream = atoi(e) != 0;
        if (const char *e = getenv("SPASM_AMD_FUSED")) use_fused = atoi(e) != 0;
        for (auto &e : ev_fz) HIPCHK(hipEventCreate(&e));
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cu = prop.multiProcessorCount;
    }
    ~Round()
    {
        for (auto &e : ev) if (e) (void)hipEventDestroy(e);
        for (auto &e : ev_cls) if (e) (void)hipEventDestroy(e);
        for (auto &e : ev_fz) if (e) (void)hipEventDestroy(e);
        for (auto &e : ev_w) if (e) (void)hipEventDestroy(e);
        if (fb_stream) (void)hipStreamDestroy(fb_stream);
        if (ev_fb_fork) (void)hipEventDestroy(ev_fb_fork);
        if (ev_fb_join) (void)hipEventDestroy(ev_fb_join);
        if (ev_classified) (void)hipEventDestroy(ev_classified);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        for (auto &e : lane_ev) if (e) (void)hipEventDestroy(e);
        for (auto &l : lane_s) if (l) (void)hipStreamDestroy(l);
        for (auto &e : ev_cls_done) if (e) (void)hipEventDestroy(e);
        if (side) (void)hipStreamDestroy(side);
        for (auto &e : twin_ev) if (e) (void)hipEventDestroy(e);
        for (auto &l : twin_s) if (l) (void)hipStreamDestroy(l);
    }

    // ---- (1a) local candidates: best[j] = min over local rows with leftmost column j of (len, global row)
    void elect_local(const DevMat &A, int row_base, int row_stride = 1)
    {
        m = A.m;
        best.ensure((size_t)m + 1);
        hipLaunchKernelGGL(k_fill_u64, dim3(cdiv((i64)m + 1, 256)), dim3(256), 0, stream, (i64d)m + 1, (u64d)NO_BEST, best.p);
        HIPCHK(hipGetLastError());
        if (A.n > 0) {
            hipLaunchKernelGGL(k_elect, dim3(cdiv(A.n, 256)), dim3(256), 0, stream, A.n, row_base, row_stride, A.len.p, A.lead.p, best.p);
            HIPCHK(hipGetLastError());
        }
    }

    // ---- (1b) number the pivots by ascending column; pivrow holds GLOBAL row ids
    void assign_pivots()
    {
        colflag.ensure((size_t)m + 1);
        colscan.ensure((size_t)m + 1);
        qinv_r.ensure((size_t)m + 1);
        hipLaunchKernelGGL(k_col_flags, dim3(cdiv((i64)m + 1, 256)), dim3(256), 0, stream, m, best.p, colflag.p);
        HIPCHK(hipGetLastError());
        scan.exclusive(colflag.p, colscan.p, (size_t)m + 1, stream);
        HIPCHK(hipMemcpyAsync(&npiv, colscan.p + m, sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        pivrow.ensure((size_t)npiv + 1);
        pivcol.ensure((size_t)npiv + 1);
        if (m > 0) {
            hipLaunchKernelGGL(k_col_assign, dim3(cdiv(m, 256)), dim3(256), 0, stream, m, best.p, colscan.p, qinv_r.p, pivrow.p, pivcol.p);
            HIPCHK(hipGetLastError());
        }
    }

    // ---- (1b') "FL on columns": after the leftmost election (best[] filled, pivots numbered), non-pivot rows take pivots on
    // columns no pivot row touches (kernels.hpp, k_close_cols ..).  Renumbers the pivots: the new ones first.  Returns how many
    // were added.  Single device only (the proposals would need a second all-reduce in a sharded round).
    DevBuf<int> closed, colcnt, prop, newflag, newscan;
    bool quiet_fallbacks = false; // a plan whose dry run sent no row past the first combine class: later runs skip those launches
    bool quiet_known = false;     // set once a plan's dry run is over: the rows and the matrix of its solves do not change any more
    int gathered_n = -1;
    DevBuf<int> pivval; // the pivot entries before scaling (the diagonal of L), filled by build_U when want_pivval
    bool want_pivval = false;
    DevBuf<u64d> best2;
    int n_leftmost = 0, n_open = 0;
    DevBuf<int> newpass, newidx, newrow_of_col, newrows;
    // The search in steps, cut where an array that describes ALL rows must be reduced over the row shards (one device: no cut).
    // (row_base, row_stride): local row i is global row row_base + i * row_stride; pivrow / newrows hold global rows.
    int open_count[OPEN_PASSES + 1] = {0};
    int open_nnew = 0, open_npass = 0;
    // -> closed[] (reduce: MAX): the columns the pivot rows THIS shard holds touch
    void open_begin(const DevMat &A, int row_base, int row_stride)
    {
        n_leftmost = npiv;
        n_open = 0;
        open_nnew = open_npass = 0;
        for (int &c : open_count) c = 0;
        closed.ensure((size_t)m + 1); colcnt.ensure((size_t)m + 1); newflag.ensure((size_t)m + 1); newscan.ensure((size_t)m + 1);
        newpass.ensure((size_t)m + 1); newidx.ensure((size_t)m + 1); newrow_of_col.ensure((size_t)m + 1); newrows.ensure((size_t)m + 1);
        best2.ensure((size_t)m + 1);
        prop.ensure((size_t)A.n + 1);
        is_piv.ensure((size_t)A.n + 1);
        HIPCHK(hipMemsetAsync(closed.p, 0, ((size_t)m + 1) * sizeof(int), stream));
        HIPCHK(hipMemsetAsync(newpass.p, 0, ((size_t)m + 1) * sizeof(int), stream));
        HIPCHK(hipMemsetAsync(is_piv.p, 0, ((size_t)A.n + 1) * sizeof(int), stream));
        if (npiv > 0) {
            hipLaunchKernelGGL(k_mark_rows, dim3(cdiv(npiv, 256)), dim3(256), 0, stream, npiv, row_base, row_stride, A.n, pivrow.p, is_piv.p);
            constexpr int TEAM = 8;
            hipLaunchKernelGGL((k_close_cols<TEAM>), dim3(cdiv((i64)npiv * TEAM, 256)), dim3(256), 0, stream, npiv, pivrow.p, row_base, row_stride, A.n, A.start.p, A.len.p,
                               A.ent.p, closed.p);
            HIPCHK(hipGetLastError());
        }
    }
    // -> colcnt[] (reduce: SUM): rows that are no pivots per column
    void open_hist(const DevMat &A)
    {
        HIPCHK(hipMemsetAsync(colcnt.p, 0, ((size_t)m + 1) * sizeof(int), stream));
        HIPCHK(hipMemsetAsync(newflag.p, 0, ((size_t)m + 1) * sizeof(int), stream));
        hipLaunchKernelGGL(k_fill_u64, dim3(cdiv((i64)m + 1, 256)), dim3(256), 0, stream, (i64d)m + 1, (u64d)NO_BEST, best2.p);
        if (A.n > 0) hipLaunchKernelGGL(k_col_histogram, dim3(cdiv((i64)A.n * 8, 256)), dim3(256), 0, stream, A.n, is_piv.p, A.start.p, A.len.p, A.ent.p, colcnt.p);
        HIPCHK(hipGetLastError());
    }
    // -> best2[] (reduce: MIN): the best proposal per column
    void open_propose(const DevMat &A, int row_base, int row_stride)
    {
        constexpr int TEAM = 8;
        if (A.n > 0)
            hipLaunchKernelGGL((k_propose_open<TEAM>), dim3(cdiv((i64)A.n * TEAM, 256)), dim3(256), 0, stream, A.n, row_base, row_stride, is_piv.p, A.start.p, A.len.p,
                               A.ent.p, closed.p, colcnt.p, prop.p, best2.p);
        HIPCHK(hipGetLastError());
    }
    // -> newflag[] (reduce: MAX): the columns whose winner holds no other proposed column
    void open_accept(const DevMat &A, int row_base, int row_stride)
    {
        constexpr int TEAM = 8;
        if (A.n > 0)
            hipLaunchKernelGGL((k_accept_open<TEAM>), dim3(cdiv((i64)A.n * TEAM, 256)), dim3(256), 0, stream, A.n, row_base, row_stride, A.start.p, A.len.p, A.ent.p,
                               prop.p, best2.p, newflag.p);
        HIPCHK(hipGetLastError());
    }
    // the accepted pivots of the pass (the same on every shard) -> closed[] (reduce: MAX); returns how many
    int open_record(const DevMat &A, int pass, int row_base, int row_stride)
    {
        scan.exclusive(newflag.p, newscan.p, (size_t)m + 1, stream);
        int nacc = 0;
        HIPCHK(hipMemcpyAsync(&nacc, newscan.p + m, sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (nacc == 0) return 0;
        constexpr int TEAM = 8;
        hipLaunchKernelGGL(k_record_open, dim3(cdiv(m, 256)), dim3(256), 0, stream, m, pass, newflag.p, newscan.p, best2.p, newpass.p, newidx.p,
                           newrow_of_col.p, newrows.p, is_piv.p, row_base, row_stride, A.n);
        hipLaunchKernelGGL((k_close_cols<TEAM>), dim3(cdiv((i64)nacc * TEAM, 256)), dim3(256), 0, stream, nacc, newrows.p, row_base, row_stride, A.n, A.start.p, A.len.p,
                           A.ent.p, closed.p);
        HIPCHK(hipGetLastError());
        open_count[pass] = nacc;
        open_nnew += nacc;
        open_npass = pass;
        return nacc;
    }
    // renumber: the open-column pivots first (last pass first), the leftmost pivots behind them (colscan still holds their ascending
    // numbering).  Returns how many pivots the search added.
    int open_finish()
    {
        const int nnew = open_nnew;
        if (nnew == 0) return 0;
        int base[OPEN_PASSES + 2] = {0};
        for (int pass = open_npass, at = 0; pass >= 1; pass--) { base[pass] = at; at += open_count[pass]; }
        static_assert(OPEN_PASSES == 4, "k_col_assign2 takes the bases of four passes");
        pivrow.ensure((size_t)npiv + nnew + 1);
        pivcol.ensure((size_t)npiv + nnew + 1);
        hipLaunchKernelGGL(k_col_assign2, dim3(cdiv(m, 256)), dim3(256), 0, stream, m, nnew, make_int4(base[1], base[2], base[3], base[4]), best.p, newpass.p,
                           newidx.p, newrow_of_col.p, colscan.p, qinv_r.p, pivrow.p, pivcol.p);
        HIPCHK(hipGetLastError());
        npiv += nnew;
        n_open = nnew;
        return nnew;
    }
    int extend_pivots_on_open_columns(const DevMat &A)
    {
        n_leftmost = npiv;
        n_open = 0;
        if (npiv == 0 || A.n == 0) return 0;
        open_begin(A, 0, 1);
        for (int pass = 1; pass <= OPEN_PASSES; pass++) {
            open_hist(A);
            open_propose(A, 0, 1);
            open_accept(A, 0, 1);
            if (open_record(A, pass, 0, 1) == 0) break;
        }
        return open_finish();
    }

    // ---- (1b'') the greedy cycle-free search (greedy.hpp; reference README.md:23): after the leftmost election and "FL on columns",
    // non-pivot rows take pivots on columns that carry none as long as the pivots stay permutable to a triangle.  Appends the new
    // pivots, then renumbers ALL pivots of the round topologically (descending level, ascending column).  Returns how many were
    // added.  Single device only, like "FL on columns".
    DevBuf<int> gr_accept, gr_ascan, gr_lev, gr_iota, gr_order, gr_prow, gr_pcol, gr_flag;
    DevBuf<u64d> gr_keys, gr_keys2;
    int n_greedy = 0;
    int extend_pivots_cycle_free(const DevMat &A)
    {
        n_greedy = 0;
        if (npiv == 0 || A.n == 0) return 0;
        const int n = A.n;
        best2.ensure((size_t)m + 1);
        prop.ensure((size_t)n + 1);
        gr_accept.ensure((size_t)n + 1);
        gr_ascan.ensure((size_t)n + 1);
        is_piv.ensure((size_t)n + 1);
        HIPCHK(hipMemsetAsync(is_piv.p, 0, ((size_t)n + 1) * sizeof(int), stream));
        hipLaunchKernelGGL(k_mark_rows, dim3(cdiv(npiv, 256)), dim3(256), 0, stream, npiv, 0, 1, n, pivrow.p, is_piv.p);
        HIPCHK(hipGetLastError());
        int nnew = 0;
        int reach_max = GREEDY_REACH_MAX_DEFAULT, occ_max = GREEDY_OCC_MAX_DEFAULT;
        if (const char *e = getenv("SPASM_AMD_GREEDY_REACH_MAX")) reach_max = std::min(GR_BUDGET, std::max(0, atoi(e)));
        if (const char *e = getenv("SPASM_AMD_GREEDY_OCC_MAX")) occ_max = std::max(1, atoi(e));
        colcnt.ensure((size_t)m + 1);
        for (int pass = 1; pass <= GREEDY_PASSES; pass++) {
            HIPCHK(hipMemsetAsync(colcnt.p, 0, ((size_t)m + 1) * sizeof(int), stream));
            hipLaunchKernelGGL(k_col_histogram, dim3(cdiv((i64)n * 8, 256)), dim3(256), 0, stream, n, is_piv.p, A.start.p, A.len.p, A.ent.p, colcnt.p);
            hipLaunchKernelGGL(k_fill_u64, dim3(cdiv((i64)m + 1, 256)), dim3(256), 0, stream, (i64d)m + 1, (u64d)NO_BEST, best2.p);
            hipLaunchKernelGGL((k_greedy<1>), dim3(cdiv(n, GR_WPB)), dim3(64 * GR_WPB), 0, stream, n, is_piv.p, A.start.p, A.len.p, A.ent.p, qinv_r.p, pivrow.p, best2.p,
                               prop.p, gr_accept.p, colcnt.p, occ_max, reach_max);
            hipLaunchKernelGGL((k_greedy<2>), dim3(cdiv(n, GR_WPB)), dim3(64 * GR_WPB), 0, stream, n, is_piv.p, A.start.p, A.len.p, A.ent.p, qinv_r.p, pivrow.p, best2.p,
                               prop.p, gr_accept.p, colcnt.p, occ_max, reach_max);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemsetAsync(gr_accept.p + n, 0, sizeof(int), stream));
            scan.exclusive(gr_accept.p, gr_ascan.p, (size_t)n + 1, stream);
            int nacc = 0;
            HIPCHK(hipMemcpyAsync(&nacc, gr_ascan.p + n, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            if (nacc == 0) break;
            // (pivrow / pivcol grow: keep what is there)
            if ((size_t)(npiv + nnew + nacc + 1) > pivrow.n) {
                DevBuf<int> r2, c2;
                r2.alloc((size_t)(npiv + nnew + nacc) + (size_t)n / 8 + 1024);
                c2.alloc(r2.n);
                HIPCHK(hipMemcpyAsync(r2.p, pivrow.p, (size_t)(npiv + nnew) * sizeof(int), hipMemcpyDeviceToDevice, stream));
                HIPCHK(hipMemcpyAsync(c2.p, pivcol.p, (size_t)(npiv + nnew) * sizeof(int), hipMemcpyDeviceToDevice, stream));
                HIPCHK(hipStreamSynchronize(stream));
                pivrow = std::move(r2);
                pivcol = std::move(c2);
            }
            hipLaunchKernelGGL(k_greedy_record, dim3(cdiv(n, 256)), dim3(256), 0, stream, n, npiv + nnew, gr_accept.p, gr_ascan.p, prop.p, pivrow.p, pivcol.p, qinv_r.p,
                               is_piv.p);
            HIPCHK(hipGetLastError());
            nnew += nacc;
        }
        if (nnew == 0) return 0;
        npiv += nnew;
        n_greedy = nnew;
        // ---- topological renumbering of all pivots of the round
        gr_lev.ensure((size_t)npiv + 1); gr_iota.ensure((size_t)npiv + 1); gr_order.ensure((size_t)npiv + 1);
        gr_prow.ensure((size_t)npiv + 1); gr_pcol.ensure((size_t)npiv + 1); gr_keys.ensure((size_t)npiv + 1); gr_keys2.ensure((size_t)npiv + 1);
        gr_flag.ensure(4);
        HIPCHK(hipMemsetAsync(gr_lev.p, 0, ((size_t)npiv + 1) * sizeof(int), stream));
        constexpr int TEAM = 8;
        for (int batch = 0;; batch++) {
            if (batch * 8 > npiv + 8) throw EngineError("greedy pivot search: the pivots do not form an acyclic graph");
            for (int it = 0; it < 8; it++) {
                if (it == 7) HIPCHK(hipMemsetAsync(gr_flag.p, 0, sizeof(int), stream));
                hipLaunchKernelGGL((k_piv_relax<TEAM>), dim3(cdiv((i64)npiv * TEAM, 256)), dim3(256), 0, stream, npiv, pivrow.p, pivcol.p, A.start.p, A.len.p, A.ent.p,
                                   qinv_r.p, gr_lev.p, it == 7 ? gr_flag.p : (int *)nullptr);
            }
            HIPCHK(hipGetLastError());
            int ch = 1;
            HIPCHK(hipMemcpyAsync(&ch, gr_flag.p, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            if (ch == 0) break;
        }
        hipLaunchKernelGGL(k_piv_keys, dim3(cdiv(npiv, 256)), dim3(256), 0, stream, npiv, gr_lev.p, pivcol.p, gr_keys.p, gr_iota.p);
        HIPCHK(hipGetLastError());
        {
            size_t bytes = 0;
            HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, gr_keys.p, gr_keys2.p, gr_iota.p, gr_order.p, (size_t)npiv, 0, 64, stream));
            sort_tmp.ensure(bytes);
            HIPCHK(rocprim::radix_sort_pairs(sort_tmp.p, bytes, gr_keys.p, gr_keys2.p, gr_iota.p, gr_order.p, (size_t)npiv, 0, 64, stream));
        }
        HIPCHK(hipMemcpyAsync(gr_prow.p, pivrow.p, (size_t)npiv * sizeof(int), hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipMemcpyAsync(gr_pcol.p, pivcol.p, (size_t)npiv * sizeof(int), hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(k_piv_permute, dim3(cdiv(npiv, 256)), dim3(256), 0, stream, npiv, gr_order.p, gr_prow.p, gr_pcol.p, pivrow.p, pivcol.p, qinv_r.p);
        HIPCHK(hipGetLastError());
        return nnew;
    }

    // ---- (1c) local non-pivot, non-empty rows
    // row_stride: local row i is global row row_base + i * row_stride; (lo, hi, step): which LOCAL rows this plan reduces
    void mark_local(const DevMat &A, int row_base, int lo = 0, int hi = INT_MAX, int row_stride = 1, int step = 1)
    {
        is_piv.ensure((size_t)A.n + 1);
        rowflag.ensure((size_t)A.n + 1);
        rowscan.ensure((size_t)A.n + 1);
        np_rows.ensure((size_t)A.n + 1);
        HIPCHK(hipMemsetAsync(is_piv.p, 0, ((size_t)A.n + 1) * sizeof(int), stream));
        if (npiv > 0) {
            hipLaunchKernelGGL(k_mark_rows, dim3(cdiv(npiv, 256)), dim3(256), 0, stream, npiv, row_base, row_stride, A.n, pivrow.p, is_piv.p);
            HIPCHK(hipGetLastError());
        }
        hipLaunchKernelGGL(k_row_flags, dim3(cdiv((i64)A.n + 1, 256)), dim3(256), 0, stream, A.n, lo, hi, step, is_piv.p, A.len.p, rowflag.p);
        HIPCHK(hipGetLastError());
        scan.exclusive(rowflag.p, rowscan.p, (size_t)A.n + 1, stream);
        HIPCHK(hipMemcpyAsync(&nnp, rowscan.p + A.n, sizeof(int), hipMemcpyDeviceToHost, stream));
        if (A.n > 0) {
            hipLaunchKernelGGL(k_compact, dim3(cdiv(A.n, 256)), dim3(256), 0, stream, A.n, rowflag.p, rowscan.p, np_rows.p);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(stream));
    }

    // ---- (2) U from the pivot rows; PM holds them, rowsrc[idx] = row of PM (device array)
    void build_U(const DevMat &PM, const int *rowsrc)
    {
        ulen.ensure((size_t)npiv + 1);
        uoff.ensure((size_t)npiv + 1);
        uhdr.ensure((size_t)npiv + 1);
        hipLaunchKernelGGL(k_gather_len, dim3(cdiv((i64)npiv + 1, 256)), dim3(256), 0, stream, npiv, rowsrc, PM.len.p, ulen.p);
        HIPCHK(hipGetLastError());
        scan.exclusive(ulen.p, uoff.p, (size_t)npiv + 1, stream);
        i64d tot = 0;
        HIPCHK(hipMemcpyAsync(&tot, uoff.p + npiv, sizeof(i64d), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        utotal = tot;
        if (utotal >= (i64)0xffffffffLL) throw EngineError("pivot rows of one round exceed 2^32 entries");
        Ufull.ensure((size_t)utotal + 1);
        UPP.ensure((size_t)utotal + 1);
        UPN.ensure((size_t)utotal + 1);
        if (npiv > 0) {
            constexpr int TEAM = 16;
            if (want_pivval) pivval.ensure((size_t)npiv + 1);
            hipLaunchKernelGGL((k_build_U<TEAM>), dim3(cdiv((i64)npiv * TEAM, 256)), dim3(256), 0, stream, npiv, F, rowsrc, pivcol.p,
                               PM.start.p, PM.len.p, PM.ent.p, qinv_r.p, uoff.p, Ufull.p, UPP.p, UPN.p, uhdr.p, want_pivval ? pivval.p : nullptr);
            HIPCHK(hipGetLastError());
        }
        free_cols = m - npiv;
    }

    // ---- (3) SOLVE for `nrows` rows of M (rows = list of local rows or NULL)
    void alloc_solve(int nrows, i64 lpool_entries)
    {
        Lstart.ensure((size_t)nrows + 1);
        Llen.ensure((size_t)nrows + 1);
        bound.ensure((size_t)nrows + 1);
        sstart.ensure((size_t)nrows + 1);
        overflow_list.ensure((size_t)nrows + 1);
        fail_list.ensure((size_t)nrows + 1);
        overflow2_list.ensure((size_t)nrows + 1);
        pmask.ensure((size_t)nrows + 1);
        sflag.ensure((size_t)nrows + 1);
        fixcnt.ensure((size_t)nrows + 1);
        if (use_stream) fixbuf.ensure(((size_t)nrows + 1) * SFIX);
        npool_active = std::min(NPOOL, std::max(1, nrows / 4));
        region_cap = ((u64d)lpool_entries + npool_active - 1) / npool_active;
        Lpool.ensure((size_t)(region_cap * npool_active) + 1);
        if (want_idx) Lidx.ensure((size_t)(region_cap * npool_active) + 1);
        pool_ctr.ensure((size_t)NPOOL * POOL_STRIDE);
        alloc_big();
        ctr.ensure(NCTR);
        class_count.ensure(NCLASS);
        class_list.ensure((size_t)NCLASS * (size_t)(nrows > 0 ? nrows : 1));
        class_desc.ensure((size_t)NCLASS * (size_t)(nrows > 0 ? nrows : 1));
        S.start.ensure((size_t)nrows + 1);
        S.len.ensure((size_t)nrows + 1);
        S.lead.ensure((size_t)nrows + 1);
        S.orig.ensure((size_t)nrows + 1);
        // (what run_solve would otherwise allocate on its first use, between two launches of a round's timed step)
        rstart.ensure((size_t)nrows + 1);
        rlen.ensure((size_t)nrows + 1);
        wreject_list.ensure((size_t)nrows + 1);
        own_ctr.ensure((size_t)NPOOL * POOL_STRIDE);
    }

    void alloc_big()
    {
        if (big_npiv == npiv && big_blocks > 0) return;
        // the dense fallback keeps npiv * 20 bytes per workgroup; stay under ~2 GB
        const i64 per = std::max<i64>((i64)npiv, 1) * 20;
        // the kernel is a chain of dependent round trips per popped pivot: rows in flight are its throughput, so as many
        // one-wave workgroups as the chip holds (32 per CU), within 16 GiB of scratch
        big_blocks = (int)std::max<i64>(1, std::min<i64>((i64)num_cu * 32, ((i64)16 << 30) / per));
        big_npiv = npiv;
        const size_t nw = ((size_t)std::max(npiv, 1) + 31) / 32;
        xdense.alloc((size_t)big_blocks * (size_t)std::max(npiv, 1));
        bitmap.alloc((size_t)big_blocks * nw);
        bigscratch.alloc((size_t)big_blocks * (size_t)std::max(npiv, 1));
        xdense.zero(stream);
        bitmap.zero(stream);
    }

    // chain solve: small teams first, then one wave per row, then the unbounded fallback
    // (only, only_count: when given, the row slots to solve -- what the plan kernel left -- instead of all of them)
    void launch_chain(SolveArgs a, int nrows, bool first_class, const int *only = nullptr, const int *only_count = nullptr)
    {
        if (first_class) {
            a.retry = only;
            a.retry_count = only_count;
            a.overflow_list = overflow_list.p;
            a.overflow_count = &ctr.p->solve_overflow;
            constexpr int TEAM = 8, CAP = 128, TPB = 256;
            const int grid = only ? std::min(cdiv((i64)nrows * TEAM, TPB), num_cu * 8) : cdiv((i64)nrows * TEAM, TPB);
            hipLaunchKernelGGL((k_solve<TEAM, CAP, TPB>), dim3(grid), dim3(TPB), 0, stream, a);
            HIPCHK(hipGetLastError());
        }
        BigSolveArgs b;
        b.npiv = std::max(npiv, 1);
        b.nwords = (std::max(npiv, 1) + 31) / 32;
        b.xdense = xdense.p;
        b.bitmap = bitmap.p;
        b.scratch = bigscratch.p;
        const size_t lds_dense = (size_t)b.npiv * 4 + (size_t)b.nwords * 4;
        if (lds_dense <= 150 * 1024) {
            // second class: one workgroup per row, dense multiplier vector + pending bitmap in LDS (unbounded reach)
            b.s = a;
            b.s.retry = overflow_list.p;
            b.s.retry_count = &ctr.p->solve_overflow;
            b.s.overflow_list = nullptr;
            b.s.overflow_count = nullptr;
            static bool attr_done[kMaxDev] = {false};
            bool &done = attr_done[current_device()];
            if (!done) {
                HIPCHK(hipFuncSetAttribute((const void *)k_solve_big<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                done = true;
            }
            const int per_cu = (int)std::max<size_t>(1, (size_t)(160 * 1024) / (lds_dense + 64));
            hipLaunchKernelGGL((k_solve_big<true, 256>), dim3(std::min(big_blocks, num_cu * per_cu)), dim3(256), lds_dense, stream, b);
            HIPCHK(hipGetLastError());
            return;
        }
        a.retry = overflow_list.p;
        a.retry_count = &ctr.p->solve_overflow;
        a.overflow_list = fail_list.p;
        a.overflow_count = &ctr.p->solve_failed;
        {
            constexpr int TEAM = 64, CAP = 4096, TPB = 64;
            hipLaunchKernelGGL((k_solve<TEAM, CAP, TPB>), dim3(std::min(nrows, 5 * num_cu)), dim3(TPB), 0, stream, a);
            HIPCHK(hipGetLastError());
        }
        b.s = a;
        b.s.retry = fail_list.p;
        b.s.retry_count = &ctr.p->solve_failed;
        b.s.overflow_list = nullptr;
        b.s.overflow_count = nullptr;
        hipLaunchKernelGGL((k_solve_big<false, 64>), dim3(big_blocks), dim3(64), 0, stream, b);
        HIPCHK(hipGetLastError());
    }

    // rows of Uinv: the chain solve applied to the unit rows e_r (once per round, after build_U)
    // expected_rows = rows the solve will process: building Uinv costs npiv chain solves, so it pays only for more rows than that
    double ms_uinv = 0, ms_w = 0;   // wall time of the last prepare_uinv / prepare_w, host synchronisations included
    double ms_levels = 0;           // of ms_w: the levels of the pivot graph (build_levels)
    double ms_w_sizing = 0;         // of ms_w: sizing the [U_PN | own | W] buffer and allocating what the build needs

    void prepare_uinv(i64 expected_rows = ((i64)1 << 62))
    {
        HIPCHK(hipStreamSynchronize(stream));
        const double t0 = spasm_wtime();
        prepare_uinv_(expected_rows);
        HIPCHK(hipStreamSynchronize(stream));
        ms_uinv = 1e3 * (spasm_wtime() - t0);
    }
    void prepare_w(i64 expected_rows, i64 own_entries)
    {
        HIPCHK(hipStreamSynchronize(stream));
        const double t0 = spasm_wtime();
        prepare_w_(expected_rows, own_entries);
        HIPCHK(hipStreamSynchronize(stream));
        ms_w = 1e3 * (spasm_wtime() - t0);
    }

    void prepare_uinv_(i64 expected_rows)
    {
        use_uinv = false;
        uinv_nnz = 0;
        if (npiv == 0 || expected_rows < (i64)npiv) return;
        E.n = npiv;
        E.m = m;
        E.start.ensure((size_t)npiv + 1);
        E.len.ensure((size_t)npiv + 1);
        E.ent.ensure((size_t)npiv + 1);
        UinvStart.ensure((size_t)npiv + 1);
        UinvLen.ensure((size_t)npiv + 1);
        ubound.ensure((size_t)npiv + 1);
        overflow_list.ensure((size_t)npiv + 1);
        fail_list.ensure((size_t)npiv + 1);
        ctr.ensure(NCTR);
        alloc_big();
        hipLaunchKernelGGL(k_unit_rows, dim3(cdiv(npiv, 256)), dim3(256), 0, stream, npiv, pivcol.p, E.start.p, E.len.p, E.ent.p);
        HIPCHK(hipGetLastError());
        i64 pool = std::max<i64>(24 * (i64)npiv, 1 << 16);
        const i64 limit = 96 * (i64)npiv + (1 << 16); // beyond this average reach the combine would cost more than the chains
        for (;;) {
            npool_active = std::min(NPOOL, std::max(1, npiv / 4));
            const u64d ucap = ((u64d)pool + npool_active - 1) / npool_active;
            UinvPool.ensure((size_t)(ucap * npool_active) + 1);
            pool_ctr.ensure((size_t)NPOOL * POOL_STRIDE);
            HIPCHK(hipMemsetAsync(ctr.p, 0, NCTR * sizeof(RoundCounters), stream));
            HIPCHK(hipMemsetAsync(pool_ctr.p, 0, (size_t)NPOOL * POOL_STRIDE * sizeof(u64d), stream));
            SolveArgs a;
            a.nrows = npiv;
            a.rows = nullptr;
            a.self_idx = nullptr;
            a.start = E.start.p;
            a.len = E.len.p;
            a.ent = E.ent.p;
            a.qinv_r = qinv_r.p;
            a.uhdr = uhdr.p;
            a.UPP = UPP.p;
            a.Lpool = nullptr;
            a.Lidx = nullptr;
            a.Lpool2 = UinvPool.p;
            a.lpool_cap = ucap;
            a.pool_ctr = pool_ctr.p;
            a.npool = npool_active;
            a.Lstart = UinvStart.p;
            a.Llen = UinvLen.p;
            a.bound = ubound.p;
            a.free_cols = free_cols;
            a.ctr = ctr.p;
            a.F = F;
            // only the LDS classes: a pivot whose own reach is beyond them makes Uinv too large anyway
            a.retry = nullptr;
            a.retry_count = nullptr;
            a.overflow_list = overflow_list.p;
            a.overflow_count = &ctr.p->solve_overflow;
            {
                constexpr int TEAM = 8, CAP = 128, TPB = 256;
                hipLaunchKernelGGL((k_solve<TEAM, CAP, TPB>), dim3(cdiv((i64)npiv * TEAM, TPB)), dim3(TPB), 0, stream, a);
                HIPCHK(hipGetLastError());
            }
            {
                // pivots whose own reach exceeds the first class (128) already account for more than the budget of Uinv:
                // do not spend seconds in the second class to find that out (Macaulay-like rounds)
                int novf = 0;
                HIPCHK(hipMemcpyAsync(&novf, &ctr.p->solve_overflow, sizeof(int), hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                if ((i64)novf * 128 > limit) return;
            }
            a.retry = overflow_list.p;
            a.retry_count = &ctr.p->solve_overflow;
            a.overflow_list = nullptr;
            a.overflow_count = &ctr.p->solve_failed;
            {
                constexpr int TEAM = 64, CAP = 4096, TPB = 64;
                hipLaunchKernelGGL((k_solve<TEAM, CAP, TPB>), dim3(std::min(npiv, 5 * num_cu)), dim3(TPB), 0, stream, a);
                HIPCHK(hipGetLastError());
            }
            RoundCounters c = read_counters();
            const u64d used = pool_used(); // synchronises
            if (c.solve_failed) return;                       // some pivot reaches > 4096 others: chains it is
            if (c.lpool_overflow) {
                if (pool >= NPOOL * (limit / NPOOL + 1)) return; // Uinv too dense
                pool = std::min<i64>(std::max<i64>(pool * 4, (i64)used + 1024), NPOOL * (limit / NPOOL + 1));
                continue;
            }
            c.lpool_used = used;
            uinv_nnz = (i64)c.lpool_used;
            use_uinv = uinv_nnz <= limit && uinv_nnz < (i64)0x7fffffff;
            if (use_uinv) {
                colinfo.ensure((size_t)m + 1);
                pbits.ensure((size_t)cdiv(m, 256) * 8 + 2);
                hipLaunchKernelGGL(k_colinfo, dim3(cdiv(m, 256)), dim3(256), 0, stream, m, qinv_r.p, UinvStart.p, UinvLen.p, colinfo.p, pbits.p);
                HIPCHK(hipGetLastError());
            }
            return;
        }
    }

    // ---- levels of the pivot graph (wlevel.hpp): relaxation on the device, rows ordered by level with a stable radix sort (a
    // level's rows in ascending pivot index: the build is the same every time).  Part of the pivot bookkeeping of a round, like the
    // reference's topological reordering of the pivots (README.md:21-24); the Schur step proper (build_w_levels ..) has no
    // host synchronisation.
    void build_levels()
    {
        depth = -1;
        if (npiv == 0) return;
        lev.ensure((size_t)npiv + 1);
        lev_keys.ensure((size_t)npiv + 1);
        lev_iota.ensure((size_t)npiv + 1);
        lev_order.ensure((size_t)npiv + 1);
        lev_start_d.ensure(WMAXLEV + 4);
        lev_flag.ensure(4);
        HIPCHK(hipMemsetAsync(lev.p, 0, ((size_t)npiv + 1) * sizeof(int), stream));
        bool converged = false;
        // a launch moves every row at least one level on: WMAXLEV + 2 launches settle any graph that is not too deep
        for (int batch = 0; batch * 8 < WMAXLEV + 2 && !converged; batch++) {
            for (int it = 0; it < 8; it++) {
                if (it == 7) HIPCHK(hipMemsetAsync(lev_flag.p, 0, sizeof(int), stream));
                hipLaunchKernelGGL(k_lev_relax, dim3(cdiv(npiv, 256)), dim3(256), 0, stream, npiv, uhdr.p, UPP.p, lev.p, it == 7 ? lev_flag.p : (int *)nullptr, 2);
            }
            HIPCHK(hipGetLastError());
            int ch = 1;
            HIPCHK(hipMemcpyAsync(&ch, lev_flag.p, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            converged = ch == 0;
        }
        if (!converged) return;
        hipLaunchKernelGGL(k_iota, dim3(cdiv(npiv, 256)), dim3(256), 0, stream, npiv, lev_iota.p);
        HIPCHK(hipGetLastError());
        {
            size_t bytes = 0;
            HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, lev.p, lev_keys.p, lev_iota.p, lev_order.p, (size_t)npiv, 0, 10, stream));
            sort_tmp.ensure(bytes);
            HIPCHK(rocprim::radix_sort_pairs(sort_tmp.p, bytes, lev.p, lev_keys.p, lev_iota.p, lev_order.p, (size_t)npiv, 0, 10, stream));
        }
        hipLaunchKernelGGL(k_fill_int, dim3(1), dim3(WMAXLEV + 4 <= 256 ? 256 : 512), 0, stream, WMAXLEV + 3, npiv, lev_start_d.p);
        hipLaunchKernelGGL(k_lev_starts, dim3(cdiv(npiv, 256)), dim3(256), 0, stream, npiv, lev_keys.p, lev_start_d.p, WMAXLEV + 1);
        HIPCHK(hipGetLastError());
        // what follows needs nothing from the host: the records in level order, the places of level 0 (rows that are copied) in W --
        // ONE read-back at the end
        lev_recs.ensure((size_t)npiv + 1);
        hipLaunchKernelGGL(k_lev_recs, dim3(cdiv(npiv, 256)), dim3(256), 0, stream, npiv, lev_order.p, uhdr.p, pivcol.p, lev_recs.p);
        lev0_sz.ensure((size_t)npiv + 2);
        lev0_off.ensure((size_t)npiv + 2);
        lev0_ent.ensure(1);
        HIPCHK(hipMemsetAsync(lev0_ent.p, 0, sizeof(u64d), stream));
        hipLaunchKernelGGL(k_lev0_sizes, dim3(cdiv((i64)npiv + 1, 256)), dim3(256), 0, stream, npiv, lev_start_d.p + 1, lev_recs.p, lev0_sz.p, lev0_ent.p);
        HIPCHK(hipGetLastError());
        scan.exclusive(lev0_sz.p, lev0_off.p, (size_t)npiv + 1, stream);
        lev_start.assign(WMAXLEV + 3, npiv);
        int deepest = 0;
        unsigned tot0 = 0;
        u64d ent0 = 0;
        HIPCHK(hipMemcpyAsync(lev_start.data(), lev_start_d.p, (WMAXLEV + 3) * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(&deepest, lev_keys.p + (npiv - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(&tot0, lev0_off.p + npiv, sizeof tot0, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(&ent0, lev0_ent.p, sizeof ent0, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (deepest > WMAXLEV) return;
        depth = deepest;
        lev_start[(size_t)depth + 1] = npiv;
        lev0_total = tot0;
        lev0_entries = (i64)ent0;
    }

    // ---- W, level by level (kernels only: part of every Schur step).  first: the build that follows build_U -- the workgroup kernel
    // runs behind every level and the caller then notes where it had work (wbig_at); later builds of the same U skip the others.
    int wblk_slots() const { return num_cu * (16 * 4 + 8 + 1); } // waves of the wave kernel, workgroups of the medium and of the large table
    // ---- W, level by level (kernels only: part of every Schur step).  Level 0 is a copy (k_wlevel0).  A level of many rows: the
    // wave kernel, then the workgroup kernel with the medium table for the rows it left, then the one with the largest table; a level
    // of at most 2048 rows skips the wave kernel (a launch costs ~10 us whatever it does): all its rows go to the workgroup kernel
    // with the medium table -- with the largest one when the level has no more rows than the chip has CUs.
    // first: the build that follows build_U, every launch is made and the caller notes the list counts (wbig_at); later builds of
    // the same U skip the launches whose lists were empty.
    hipEvent_t ev_w[2] = {nullptr, nullptr}; // around the level launches of the last W build
    void build_w_levels()
    {
        if (!ev_w[0]) for (auto &e : ev_w) HIPCHK(hipEventCreate(&e));
        HIPCHK(hipEventRecord(ev_w[0], stream));
        const int nblk_words = 2 * wblk_slots();
        const int ncount_words = 2 * (depth + 2) * WL_NSUB * WL_SUBSTRIDE;
        hipLaunchKernelGGL(k_wbuild_reset, dim3(cdiv(std::max(nblk_words, ncount_words), 256)), dim3(256), 0, stream, wstate.p, wblk.p, nblk_words,
                           wbig_count.p, ncount_words, (u64d)lev0_total);
        HIPCHK(hipGetLastError());
        constexpr int NT_MID = 256, NT_BIG = 1024; // (512 threads for the medium table were slower: its rows wait on memory, and fewer workgroups fit a CU)
        const size_t lds_mid = F.small ? wl_wg_lds_bytes<true>(wl_mid_slots(), NT_MID) : wl_wg_lds_bytes<false>(wl_mid_slots(), NT_MID);
        const int big_slots = F.small ? wl_big_slots<true>() : wl_big_slots<false>();
        const size_t lds_big = F.small ? wl_wg_lds_bytes<true>(big_slots, NT_BIG) : wl_wg_lds_bytes<false>(big_slots, NT_BIG);
        if (!wave_per_cu_blocks) {
            int nb = 0;
            if (F.small) HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_wlevel_wave<true>, 256, 0));
            else HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_wlevel_wave<false>, 256, 0));
            wave_per_cu_blocks = std::min(std::max(nb, 1), 16);
            if (F.small) {
                HIPCHK(hipFuncSetAttribute((const void *)k_wlevel_wg<true, NT_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
                HIPCHK(hipFuncSetAttribute((const void *)k_wlevel_wg<true, NT_MID>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mid));
            } else {
                HIPCHK(hipFuncSetAttribute((const void *)k_wlevel_wg<false, NT_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
                HIPCHK(hipFuncSetAttribute((const void *)k_wlevel_wg<false, NT_MID>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mid));
            }
        }
        const int mid_per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) / (lds_mid + 256)));
        WLevelArgs a;
        a.UPP = UPP.p;
        a.buf = UPN.p;
        a.wrow = wrow.p;
        a.wcol = wcol.p;
        a.wstate = wstate.p;
        a.wbase = (unsigned)wbase;
        a.wcap = (u64d)wcap;
        a.wblk = wblk.p;
        a.F = F;
        a.list_stride = npiv;
        auto launch_wg = [&](int tier, int grid) {
            a.tslots = tier ? big_slots : wl_mid_slots();
            a.blk_base = num_cu * 16 * 4 + (tier ? num_cu * 8 : 0);
            if (tier) {
                if (F.small) hipLaunchKernelGGL((k_wlevel_wg<true, NT_BIG>), dim3(grid), dim3(NT_BIG), lds_big, stream, a);
                else hipLaunchKernelGGL((k_wlevel_wg<false, NT_BIG>), dim3(grid), dim3(NT_BIG), lds_big, stream, a);
            } else {
                if (F.small) hipLaunchKernelGGL((k_wlevel_wg<true, NT_MID>), dim3(grid), dim3(NT_MID), lds_mid, stream, a);
                else hipLaunchKernelGGL((k_wlevel_wg<false, NT_MID>), dim3(grid), dim3(NT_MID), lds_mid, stream, a);
            }
        };
        {
            const int cnt0 = lev_start[1];
            hipLaunchKernelGGL(k_wlevel0, dim3(std::max(1, std::min(cdiv((i64)cnt0 * 16, 256), num_cu * 8))), dim3(256), 0, stream, cnt0, lev_recs.p, lev0_off.p,
                               UPN.p, wrow.p, wcol.p, (unsigned)wbase, F);
        }
        for (int L = 1; L <= depth; L++) {
            const int cnt = lev_start[(size_t)L + 1] - lev_start[(size_t)L];
            if (cnt <= 0) continue;
            int *const mid_count = wbig_count.p + (size_t)(2 * L) * WL_NSUB * WL_SUBSTRIDE, *const big_count = wbig_count.p + (size_t)(2 * L + 1) * WL_NSUB * WL_SUBSTRIDE;
            // (every tier launch is made, with the grid of a full list: how many rows a list holds is only known on the device, and the
            // build runs once per round -- a launch whose list is empty costs a few microseconds)
            const int known_mid = -1, known_big = -1;
            a.cnt = cnt;
            a.rec_base = lev_start[(size_t)L];
            a.mid_list = wmid_list.p;
            a.mid_count = mid_count;
            a.big_list = wbig_list.p;
            a.big_count = big_count;
            a.list = nullptr;
            a.count = nullptr;
            if (cnt > 2048) {
                a.recs = lev_recs.p + lev_start[(size_t)L];
                a.tslots = 0;
                a.blk_base = 0;
                const int grid = std::min(cdiv(cnt, 4), num_cu * wave_per_cu_blocks);
                if (F.small) hipLaunchKernelGGL(k_wlevel_wave<true>, dim3(grid), dim3(256), 0, stream, a);
                else hipLaunchKernelGGL(k_wlevel_wave<false>, dim3(grid), dim3(256), 0, stream, a);
                a.recs = lev_recs.p;
                if (known_mid != 0) {
                    a.list = wmid_list.p;
                    a.count = mid_count;
                    a.big_list = nullptr; // (the wave kernel sorted the rows: what is on this list fits the medium table)
                    launch_wg(0, known_mid < 0 ? num_cu * mid_per_cu : std::min(num_cu * mid_per_cu, known_mid));
                }
            } else {
                a.recs = lev_recs.p;
                if (cnt <= num_cu) {
                    a.big_list = nullptr; // all rows with the largest table: what does not fit is not available
                    launch_wg(1, cnt);
                    continue;
                }
                launch_wg(0, std::min(num_cu * mid_per_cu, cnt)); // rows beyond the medium table go on the list of the largest
            }
            if (known_big != 0) {
                a.list = wbig_list.p;
                a.count = big_count;
                a.big_list = nullptr;
                launch_wg(1, known_big < 0 ? num_cu : std::min(num_cu, known_big));
            }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ev_w[1], stream));
    }

    // Decides whether this round goes along W, sizes the [U_PN | own | W] buffer and builds W once (synchronises: the outcome of
    // the build is read back).  Worth it when many more rows than pivots are reduced; a W much larger than U means long chains,
    // which the multiplier lists serve better.
    // own_entries: upper bound of the entries of the rows that will be reduced along W (their non-pivot entries are copied into
    // the buffer by the plan kernel)
    void prepare_w_(i64 expected_rows, i64 own_entries)
    {
        use_w = false;
        wtotal = 0;
        ms_levels = 0;
        ms_w_sizing = 0;
        if (!use_stream || force_lists || want_idx || m >= (1 << 24) || npiv == 0 || expected_rows < 2 * (i64)npiv) return;
        {
            HIPCHK(hipStreamSynchronize(stream));
            const double t0 = spasm_wtime();
            build_levels();
            HIPCHK(hipStreamSynchronize(stream));
            ms_levels = 1e3 * (spasm_wtime() - t0);
        }
        if (depth < 0) return;
        const double t_sizing0 = spasm_wtime();
        // room for the rows' own entries: what they have, half as much again for uneven regions, and a block per team of the plan kernel
        const i64 own_room = ((own_entries + own_entries / 2 + (i64)num_cu * 16 * 16 * 256 + NPOOL) + 15) & ~(i64)15;
        own_base = (utotal + 15) & ~(i64)15;
        wbase = own_base + own_room;
        // offsets into the buffer are 32-bit
        i64 cap = std::min<i64>(64 * std::max<i64>(utotal, 1 << 16), (i64)0xffffffffLL - wbase - 64);
        {
            // memory only vetoes (the route changes, not the result): half of what is free beside the buffer already there
            size_t fr = 0, totmem = 0;
            HIPCHK(hipMemGetInfo(&fr, &totmem));
            const i64 have = (i64)UPN.n;
            const i64 room = have + (i64)(fr / 2 / sizeof(int2));
            if (wbase + cap + 1 > room) cap = room - wbase - 1;
        }
        cap &= ~(i64)15;
        if (cap < 4 * std::max<i64>(utotal, 1 << 12) || cap < 2 * lev0_total) return;
        if ((i64)UPN.n < wbase + cap + 1) {
            DevBuf<int2> both;
            both.alloc((size_t)(wbase + cap) + 1);
            if (utotal > 0) HIPCHK(hipMemcpyAsync(both.p, UPN.p, (size_t)utotal * sizeof(int2), hipMemcpyDeviceToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));
            UPN = std::move(both);
        }
        wcap = cap;
        wrow.ensure((size_t)npiv + 1);
        wcol.ensure((size_t)m + 1);
        wmid_list.ensure((size_t)WL_NSUB * (size_t)npiv + 1);
        wbig_list.ensure((size_t)WL_NSUB * (size_t)npiv + 1);
        wbig_count.ensure((size_t)2 * (WMAXLEV + 2) * WL_NSUB * WL_SUBSTRIDE);
        wstate.ensure(WS_WORDS);
        wblk.ensure((size_t)2 * wblk_slots());
        own_ctr.ensure((size_t)NPOOL * POOL_STRIDE);
        pbits.ensure((size_t)cdiv(m, 256) * 8 + 2);
        hipLaunchKernelGGL(k_pbits, dim3(cdiv(m, 256)), dim3(256), 0, stream, m, qinv_r.p, pbits.p);
        HIPCHK(hipGetLastError());
        wbig_at.assign((size_t)2 * (depth + 2), 0);
        {
            // (wall time of the sizing: the device-memory query and whatever had to be allocated -- once per Round for rounds of one size)
            HIPCHK(hipStreamSynchronize(stream));
            ms_w_sizing = 1e3 * (spasm_wtime() - t_sizing0);
        }
        build_w_levels();
        hipLaunchKernelGGL(k_wstats, dim3(cdiv(npiv, 256)), dim3(256), 0, stream, npiv, wrow.p, wstate.p);
        HIPCHK(hipGetLastError());
        u64d ws[WS_WORDS];
        HIPCHK(hipMemcpyAsync(ws, wstate.p, sizeof ws, hipMemcpyDeviceToHost, stream));
        const bool wdebug = getenv("SPASM_AMD_WDEBUG") != nullptr || want_w_row_stats;
        std::vector<int> parts;
        if (wdebug) { // (how many rows the workgroup kernels built, level by level: a statistic, 280 KB over PCIe)
            parts.resize((size_t)2 * (depth + 1) * WL_NSUB * WL_SUBSTRIDE);
            HIPCHK(hipMemcpyAsync(parts.data(), wbig_count.p, parts.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
        }
        HIPCHK(hipStreamSynchronize(stream));
        if (wdebug)
            for (int k = 0; k < 2 * (depth + 1); k++)
                for (int sb = 0; sb < WL_NSUB; sb++) wbig_at[(size_t)k] += parts[((size_t)k * WL_NSUB + sb) * WL_SUBSTRIDE];
        if (ws[WS_ERROR]) throw EngineError("a hash table of the W build filled up (internal bound violated)");
        if (getenv("SPASM_AMD_WDEBUG")) {
            fprintf(stderr, "[wlevel] npiv %d depth %d entries %llu cursor %llu unavailable %llu long rows %llu; rows per level (medium, large):", npiv, depth,
                    (unsigned long long)ws[WS_ENTRIES], (unsigned long long)ws[WS_CURSOR], (unsigned long long)ws[WS_UNAVAIL], (unsigned long long)ws[WS_BIGROWS]);
            for (int L = 0; L <= depth; L++)
                fprintf(stderr, " %d (%d, %d)", lev_start[(size_t)L + 1] - lev_start[(size_t)L], wbig_at[(size_t)2 * L], wbig_at[(size_t)2 * L + 1]);
            fprintf(stderr, "\n");
        }
        // rows that could not be built send the rows that need them to the multiplier lists: a few are fine, many mean W does not pay
        if (ws[WS_UNAVAIL] * 64 > (u64d)npiv) return;
        wtotal = (i64)std::min<u64d>(ws[WS_CURSOR], (u64d)cap);
        w_entries = (i64)ws[WS_ENTRIES] + lev0_entries;
        w_long_rows = 0;
        for (int k = 0; k < 2 * (depth + 1); k++) w_long_rows += wbig_at[(size_t)k];
        own_total = own_room;
        use_w = true;
    }

    void run_solve(const DevMat &M, const int *rows, const int *self_idx, int nrows)
    {
        // the plan along the rows of W; what it leaves (streams beyond the streaming classes, rows that are bound to be full of
        // duplicate columns, rows that need a row of W that could not be built) goes through the multiplier lists
        const bool wmode = use_w && !force_lists && !self_idx && !want_idx;
        {
            static_assert(sizeof(RoundCounters) % 4 == 0, "cleared word by word");
            const int nctr_words = (int)(NCTR * sizeof(RoundCounters) / 4), npool_words = NPOOL * POOL_STRIDE;
            const int span = std::max(std::max(nctr_words, npool_words), nrows + 1);
            // (the counters of the plan kernel's own-entry regions are cleared here as well: one launch less per step)
            if (wmode) own_ctr.ensure((size_t)NPOOL * POOL_STRIDE);
            hipLaunchKernelGGL(k_solve_reset, dim3(cdiv(span, 256)), dim3(256), 0, stream, nrows, (unsigned *)ctr.p, nctr_words, pool_ctr.p,
                               npool_words, class_count.p, (int)NCLASS, bound.p, pmask.p, sflag.p, wmode ? own_ctr.p : (u64d *)nullptr);
            HIPCHK(hipGetLastError());
        }
        if (nrows == 0) return;
        SolveArgs a;
        a.nrows = nrows;
        a.rows = rows;
        a.self_idx = self_idx;
        a.retry = nullptr;
        a.retry_count = nullptr;
        a.start = M.start.p;
        a.len = M.len.p;
        a.ent = M.ent.p;
        a.qinv_r = qinv_r.p;
        a.uhdr = uhdr.p;
        a.UPP = UPP.p;
        a.Lpool = Lpool.p;
        a.Lidx = want_idx ? Lidx.p : nullptr;
        a.Lpool2 = nullptr;
        a.lpool_cap = region_cap;
        a.pool_ctr = pool_ctr.p;
        a.npool = npool_active;
        a.Lstart = Lstart.p;
        a.Llen = Llen.p;
        a.bound = bound.p;
        a.free_cols = free_cols;
        a.overflow_list = overflow_list.p;
        a.overflow_count = &ctr.p->solve_overflow;
        a.ctr = ctr.p;
        a.F = F;
        if (!wmode && !use_uinv) {
            launch_chain(a, nrows, true);
            return;
        }
        rstart.ensure((size_t)nrows + 1);
        rlen.ensure((size_t)nrows + 1);
        // (every step: a round gathers them once, and a plan's step is a round's)
        hipLaunchKernelGGL(k_gather_rows, dim3(cdiv(nrows, 256)), dim3(256), 0, stream, nrows, rows, M.start.p, M.len.p, rstart.p, rlen.p);
        HIPCHK(hipGetLastError());
        gathered_n = nrows;
        if (wmode) {
            wreject_list.ensure((size_t)nrows + 1);
            WPlanArgs wp;
            wp.nrows = nrows;
            wp.rstart = rstart.p;
            wp.rlen = rlen.p;
            wp.ent = M.ent.p;
            wp.pbits = pbits.p;
            wp.wcol = wcol.p;
            wp.Lpool = Lpool.p;
            wp.lpool_cap = region_cap;
            wp.pool_ctr = pool_ctr.p;
            wp.npool = npool_active;
            wp.upn = UPN.p;
            wp.own_base = (unsigned)own_base;
            wp.own_cap = (u64d)own_total / (u64d)npool_active;
            wp.own_ctr = own_ctr.p;
            wp.Lstart = Lstart.p;
            wp.Llen = Llen.p;
            wp.bound = bound.p;
            wp.pmask = pmask.p;
            wp.sflag = sflag.p;
            wp.free_cols = free_cols;
            wp.max_bound = (int)kClasses[kNumStreamClasses - 1].cap;
            wp.wave_row_bound = (int)kClasses[2].cap;
            wp.overflow_list = wreject_list.p;
            wp.overflow_count = &ctr.p->wplan_reject;
            wp.ctr = ctr.p;
            wp.F = F;
            chunk_log = stream_chunk_log(); // (the streaming kernels of this solve's scatter take what the plan cut)
            wp.chunk_log = chunk_log;
            constexpr int TEAM = 16, TPB = 256;
            hipLaunchKernelGGL((k_wplan<TEAM, TPB>), dim3(std::min(cdiv((i64)nrows * TEAM, TPB), num_cu * 16)), dim3(TPB), 0, stream, wp);
            HIPCHK(hipGetLastError());
            if (!use_uinv) {
                // no Uinv beside W: the rows the plan kernel left are solved by elimination chains
                // (a plan that has seen its rows once knows whether anybody gets this far)
                if (!quiet_rejects) launch_chain(a, nrows, true, wreject_list.p, &ctr.p->wplan_reject);
                return;
            }
        }
        CombineArgs c;
        c.nrows = nrows;
        c.self_idx = self_idx;
        c.rstart = rstart.p;
        c.rlen = rlen.p;
        c.ent = M.ent.p;
        c.colinfo = colinfo.p;
        c.pbits = pbits.p;
        c.uhdr = uhdr.p;
        c.UinvPool = UinvPool.p;
        c.Lpool = Lpool.p;
        c.Lidx = want_idx ? Lidx.p : nullptr;
        c.sflag = nullptr; // (only the plan along W marks rows for the streaming kernel)
        c.lpool_cap = region_cap;
        c.pool_ctr = pool_ctr.p;
        c.npool = npool_active;
        c.Lstart = Lstart.p;
        c.Llen = Llen.p;
        c.bound = bound.p;
        c.pmask = pmask.p;
        c.free_cols = free_cols;
        c.retry = wmode ? wreject_list.p : nullptr;
        c.retry_count = wmode ? &ctr.p->wplan_reject : nullptr;
        c.overflow_list = overflow2_list.p;
        c.overflow_count = &ctr.p->combine_overflow;
        c.ctr = ctr.p;
        {
            const char *dbg = getenv("SPASM_DBG"); // timing ablations (diagnostic builds only)
            c.dbg = dbg ? atoi(dbg) : 0;
        }
        c.F = F;
        {
            constexpr int TEAM = 16, LOGC = 8, TPB = 256; // up to 128 distinct pivots per row
            const int grid = std::min(cdiv((i64)nrows * TEAM, TPB), num_cu * 8);
            if (F.small) hipLaunchKernelGGL((k_combine<TEAM, LOGC, TPB, true>), dim3(grid), dim3(TPB), 0, stream, c);
            else hipLaunchKernelGGL((k_combine<TEAM, LOGC, TPB, false>), dim3(grid), dim3(TPB), 0, stream, c);
            HIPCHK(hipGetLastError());
        }
        // (a plan that has seen its rows once knows whether anybody gets this far: three or four launches of ~8 us each, in
        // order, are a tenth of the step of a 1/8 shard)
        if (!quiet_fallbacks) {
            c.retry = overflow2_list.p;
            c.retry_count = &ctr.p->combine_overflow;
            c.overflow_list = overflow_list.p;
            c.overflow_count = &ctr.p->solve_overflow;
            {
                constexpr int TEAM = 64, LOGC = 12, TPB = 64; // one wave per row, up to 2048 distinct pivots
                const int grid = std::min(nrows, num_cu * 4);
                if (F.small) hipLaunchKernelGGL((k_combine<TEAM, LOGC, TPB, true>), dim3(grid), dim3(TPB), 0, stream, c);
                else hipLaunchKernelGGL((k_combine<TEAM, LOGC, TPB, false>), dim3(grid), dim3(TPB), 0, stream, c);
                HIPCHK(hipGetLastError());
            }
            launch_chain(a, nrows, false); // what is left: chain classes
        }
    }

    // sum of the NCTR statistic copies (synchronises)
    // (total_bound_of: also fetch the total of the bounds of that many rows, s_total_bound, in the same round trip)
    RoundCounters read_counters(const RoundCounters *from = nullptr, int total_bound_of = -1)
    {
        std::vector<RoundCounters> h(NCTR);
        i64d tot = 0;
        HIPCHK(hipMemcpyAsync(h.data(), from ? from : ctr.p, NCTR * sizeof(RoundCounters), hipMemcpyDeviceToHost, stream));
        if (total_bound_of >= 0) HIPCHK(hipMemcpyAsync(&tot, sstart.p + total_bound_of, sizeof(i64d), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (total_bound_of >= 0) s_total_bound = tot;
        RoundCounters c = h[0];
        for (int i = 1; i < NCTR; i++) {
            c.applications += h[i].applications;
            c.nnz_reduced += h[i].nnz_reduced;
            c.segments += h[i].segments;
            c.lpool_overflow += h[i].lpool_overflow;
            c.scatter_overflow += h[i].scatter_overflow;
            c.nonempty_out += h[i].nonempty_out;
            c.nnz_out += h[i].nnz_out;
            c.stream_redo += h[i].stream_redo;
            c.stream_fix += h[i].stream_fix;
            for (int k = 0; k < 16; k++) { c.class_ent[k] += h[i].class_ent[k]; c.class_seg[k] += h[i].class_seg[k]; }
        }
        return c;
    }

    // entries handed out by the fullest region times NPOOL (what a balanced pool would need); synchronises
    u64d pool_used()
    {
        std::vector<u64d> h((size_t)NPOOL * POOL_STRIDE);
        HIPCHK(hipMemcpyAsync(h.data(), pool_ctr.p, h.size() * sizeof(u64d), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        u64d mx = 0;
        for (int r = 0; r < npool_active; r++) mx = std::max(mx, h[(size_t)r * POOL_STRIDE]);
        return mx * (u64d)npool_active;
    }

    // solve + bounds with automatic growth of the multiplier pool; returns the total bound of S
    // returns -1 when the multiplier records of these rows would need more than max_pool entries (the caller then
    // processes fewer rows at a time)
    i64 solve_phase(const DevMat &M, const int *rows, const int *self_idx, int nrows, i64 lpool_guess, i64 max_pool = ((i64)1 << 62))
    {
        i64 pool = std::min<i64>(std::max<i64>(lpool_guess, 1 << 16), max_pool);
        for (;;) {
            alloc_solve(nrows, pool);
            run_solve(M, rows, self_idx, nrows);
            run_bounds(nrows);
            const RoundCounters c = read_counters(nullptr, nrows); // the counters and the total of the bounds: one round trip
            const i64 tot = s_total_bound;
            if (c.lpool_overflow) {
                if (pool >= max_pool) return -1;
                pool = std::min<i64>(std::max<i64>(pool * 2, (i64)(pool_used() * 5 / 4) + 1024), max_pool);
                continue;
            }
            return tot;
        }
    }

    // ---- (4) SCATTER.  S.ent must hold s_total_bound entries (size_S() after a solve).
    void run_bounds(int nrows)
    {
        scan.exclusive(bound.p, sstart.p, (size_t)nrows + 1, stream);
    }

    i64 fetch_total_bound(int nrows)
    {
        i64d tot = 0;
        HIPCHK(hipMemcpyAsync(&tot, sstart.p + nrows, sizeof(i64d), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        s_total_bound = tot;
        return tot;
    }

    // ---- what the scatter needs once per Round: its streams and events, the opt-in of its kernels to large LDS, the scratch of the
    // last-resort kernel.  Done on the first scatter -- or, by a round, in front of its timed step (warm_scatter): a hipMalloc of
    // gigabytes or the creation of six streams between two launches of the step is milliseconds of an idle device.
    static int n_lanes() { static const int v = [] { const char *e = getenv("SPASM_AMD_STREAM_LANES"); return std::min(4, std::max(1, e ? atoi(e) : 2)); }(); return v; }
    static int n_twins() { static const int v = [] { const char *e = getenv("SPASM_AMD_TWIN_STREAMS"); return std::min(4, std::max(1, e ? atoi(e) : 2)); }(); return v; }
    void ensure_streams()
    {
        if (!side) {
            HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        }
        if (!ev_cls_done[0]) for (auto &e : ev_cls_done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (int l = 1; l < n_lanes(); l++)
            if (!lane_s[l]) {
                HIPCHK(hipStreamCreateWithFlags(&lane_s[l], hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&lane_ev[l], hipEventDisableTiming));
            }
        for (int t = 0; t < n_twins() - 1; t++)
            if (!twin_s[t]) {
                HIPCHK(hipStreamCreateWithFlags(&twin_s[t], hipStreamNonBlocking));
                HIPCHK(hipEventCreateWithFlags(&twin_ev[t], hipEventDisableTiming));
            }
    }
    void ensure_big_scatter(hipStream_t s)
    {
        if (bigsc_m == m) return;
        const i64 per = std::max<i64>((i64)m, 1) * 13; // bytes per workgroup: 8 accumulator + 4 list + bitmap
        // rows in flight are its throughput; its rows are the handful beyond every LDS table: 2 GiB of scratch at most
        bigsc_blocks = (int)std::max<i64>(1, std::min<i64>((i64)num_cu * 8, ((i64)2 << 30) / per));
        bigsc_m = m;
        const size_t nw = ((size_t)std::max(m, 1) + 31) / 32;
        sc_xdense.alloc((size_t)bigsc_blocks * (size_t)std::max(m, 1));
        sc_bitmap.alloc((size_t)bigsc_blocks * nw);
        sc_touched.alloc((size_t)bigsc_blocks * (size_t)std::max(m, 1));
        sc_xdense.zero(s);
        sc_bitmap.zero(s);
    }
    void warm_scatter()
    {
        ensure_streams();
        ensure_big_scatter(stream);
        ScatterArgs a;
        memset(&a, 0, sizeof a);
        StreamArgs sa;
        memset(&sa, 0, sizeof sa);
        const int nhash = F.small ? kNumHashClasses : kNumHashClasses - 1;
        for (int c = 0; c < nhash; c++) {
            const size_t lds = scatter_lds_bytes(kClasses[c], F.small);
            if (F.small) launch_scatter_class<true>(c, a, 0, lds, stream);
            else launch_scatter_class<false>(c, a, 0, lds, stream);
        }
        for (int c = 0; c < std::min(nhash, kNumStreamClasses); c++) {
            const size_t lds = stream_lds_bytes(stream_logt(c), stream_tpr(c), stream_wpb(c));
            if (F.small) launch_stream_class<true>(c, sa, 0, num_cu, lds, stream, stream_chunk_log());
            else launch_stream_class<false>(c, sa, 0, num_cu, lds, stream, stream_chunk_log());
        }
    }

    void run_scatter(const DevMat &M, const int *rows, int nrows)
    {
        const int4 *recs = Lpool.p;
        S.n = nrows;
        S.m = m;
        nlaunch = 0;
        last_fused = false;
        if (nrows == 0) return;
        // (class_count was cleared by the reset kernel of the solve that precedes every scatter)
        int nhash = F.small ? kNumHashClasses : kNumHashClasses - 1; // 12-byte slots: the 2^14 table exceeds LDS
        // the streaming kernel goes along the rows of W (the plan kernel marks the rows it can take)
        const bool streaming = use_w && !force_lists && !want_idx;
        BinArgs b;
        b.nrows = nrows;
        b.bound = bound.p;
        b.Llen = Llen.p;
        for (int c = 0; c < NCLASS; c++) b.cap[c] = -1;
        for (int c = 0; c < nhash; c++) b.cap[c] = kClasses[c].cap;
        // classes nhash..NHASHMAX-2 are unused (cap -1 never matches); class NHASHMAX-1 collects what fits nowhere
        b.class_count = class_count.p;
        b.class_list = class_list.p;
        b.rows = rows;
        b.start = M.start.p;
        b.len = M.len.p;
        b.orig = M.orig.p;
        b.Lstart = Lstart.p;
        b.sstart = sstart.p;
        b.pmask = pmask.p;
        b.sflag = sflag.p;
        b.stream_classes = streaming ? std::min(nhash, kNumStreamClasses) : 0;
        b.Sorig = S.orig.p;
        b.fixcnt = fixcnt.p;
        b.desc = class_desc.p;
        hipLaunchKernelGGL(k_bin, dim3(cdiv(nrows, 1024)), dim3(256), 0, stream, b);
        HIPCHK(hipGetLastError());

        ScatterArgs a;
        a.ent = M.ent.p;
        a.qinv_r = qinv_r.p;
        a.uhdr = uhdr.p;
        a.UPN = UPN.p;
        a.Lpool = recs;
        a.Sent = S.ent.p + s_ent_base;
        a.Slen = S.len.p;
        a.Slead = S.lead.p;
        a.ctr = ctr.p;
        a.F = F;
        {
            const char *dbg = getenv("SPASM_DBG"); // timing ablations of the scatter kernel (wrong results when set)
            a.dbg = dbg ? atoi(dbg) : 0;
        }
        a.stamps = nullptr;
#ifdef SPASM_STAMPS
        stamps.ensure(NCLASS * 2 * NSTAMP);
        HIPCHK(hipMemsetAsync(stamps.p, 0, NCLASS * 2 * NSTAMP * sizeof(u64d), stream));
        a.stamps = stamps.p;
#endif
        StreamArgs sa;
        sa.ent = M.ent.p;
        sa.qinv_r = qinv_r.p;
        sa.UPN = UPN.p;
        sa.Lpool = recs;
        sa.Sent = S.ent.p + s_ent_base;
        sa.Slen = S.len.p;
        sa.Slead = S.lead.p;
        sa.ctr = ctr.p;
        sa.F = F;
        sa.dbg = a.dbg;
        sa.stamps = a.stamps;
        sa.fixbuf = fixbuf.p;
        sa.fixcnt = fixcnt.p;
        nhash_used = nhash;
        // The classes with the largest tables hold few rows (config 3: 7829 and 51 of 830527), each a long serial job of one
        // workgroup: launched after the others they are a tail on a nearly empty chip (0.27 of 4.1 ms; 74 of 760 us for a
        // 1/8 shard).  They go first, on a side stream, and the wide classes fill the rest of the chip meanwhile.
        const int first_side = 6;
        const bool use_side = !class_timing && (nhash > first_side || streaming);
        if (use_side) {
            ensure_streams();
            HIPCHK(hipEventRecord(ev_fork, stream));
            HIPCHK(hipStreamWaitEvent(side, ev_fork, 0));
        }
        auto mark = [&](int cls_id, hipStream_t s) { // event in front of a launch, for the per-class times
            if (!class_timing || nlaunch >= NCLASS) return;
            HIPCHK(hipEventRecord(ev_cls[nlaunch], s));
            launch_cls[nlaunch++] = cls_id;
        };
        auto launch_class = [&](int c, hipStream_t s) {
            mark(c, s);
            a.cls = c;
            a.class_count = class_count.p + c;
            a.desc = class_desc.p + (size_t)c * nrows;
            const size_t lds = scatter_lds_bytes(kClasses[c], F.small);
            int per_cu = (int)std::min<size_t>(32 / kClasses[c].wpb, (160 * 1024) / lds);
            if (per_cu < 1) per_cu = 1;
            const int rows_per_block = kClasses[c].tpr == 64 ? kClasses[c].wpb : 1;
            const int grid = std::max(1, std::min(cdiv(nrows, rows_per_block), num_cu * per_cu));
            if (F.small) launch_scatter_class<true>(c, a, grid, lds, s);
            else launch_scatter_class<false>(c, a, grid, lds, s);
        };
        // the streaming twin of hash class c; rows it gives up on land in class c's list, which therefore runs after it
        auto launch_stream_cls = [&](int c, hipStream_t s) {
            if (!streaming || c >= kNumStreamClasses) return;
            mark(NSTREAM0 + c, s);
            sa.cls = NSTREAM0 + c;
            sa.class_count = class_count.p + NSTREAM0 + c;
            sa.desc = class_desc.p + (size_t)(NSTREAM0 + c) * nrows;
            sa.redo_count = class_count.p + c;
            sa.redo_desc = class_desc.p + (size_t)c * nrows;
            const size_t lds = stream_lds_bytes(stream_logt(c), stream_tpr(c), stream_wpb(c));
            if (F.small) launch_stream_class<true>(c, sa, nrows, num_cu, lds, s, chunk_log);
            else launch_stream_class<false>(c, sa, nrows, num_cu, lds, s, chunk_log);
        };
        // the duplicates the streaming kernels found are merged afterwards, a wave per row that has any
        auto launch_stream_fix = [&](hipStream_t s) {
            if (!streaming) return;
            mark(NCLASS - 1, s); // (reported as class 15)
            StreamFixArgs fa;
            fa.nrows = nrows;
            fa.fixcnt = fixcnt.p;
            fa.fixbuf = fixbuf.p;
            fa.sstart = sstart.p;
            fa.Sent = S.ent.p + s_ent_base;
            fa.Slen = S.len.p;
            fa.Slead = S.lead.p;
            fa.ctr = ctr.p;
            fa.F = F;
            hipLaunchKernelGGL(k_stream_fix, dim3(cdiv(nrows, 64)), dim3(256), 0, s, fa);
            HIPCHK(hipGetLastError());
        };
        auto launch_big = [&](hipStream_t s) {
            // rows that fit no LDS table: the last class, through the global-memory kernel
            ensure_big_scatter(s);
            mark(NHASHMAX - 1, s);
            BigScatterArgs bb;
            bb.s = a;
            bb.s.cls = NHASHMAX - 1;
            bb.s.class_count = class_count.p + (NHASHMAX - 1);
            bb.s.desc = class_desc.p + (size_t)(NHASHMAX - 1) * nrows;
            bb.m = std::max(m, 1);
            bb.nwords = (std::max(m, 1) + 31) / 32;
            bb.xdense = sc_xdense.p;
            bb.bitmap = sc_bitmap.p;
            bb.touched = sc_touched.p;
            hipLaunchKernelGGL(k_scatter_big, dim3(bigsc_blocks), dim3(256), 0, s, bb);
            HIPCHK(hipGetLastError());
        };
        if (use_side && streaming) {
            // the streaming kernels on the main stream, longest rows first; the hash-table kernel of a class -- it only gets what its
            // streaming twin hands back -- follows that twin on the side stream, beside the streaming kernels of the shorter rows
            // several lanes: the streaming kernels go round-robin over a few streams, so that the workgroups of the next classes
            // fill the chip while the last ones of a class drain (config 3: 2.68 ms per step on one lane, 2.38 on two)
            const int lanes = n_lanes();
            for (int l = 1; l < lanes; l++) HIPCHK(hipStreamWaitEvent(lane_s[l], ev_fork, 0));
            // the hash-table twins (a handful of rows each: 20 - 50 us of one or two workgroups whatever the shard size) run two
            // abreast on streams of their own: chained on one stream they were 0.15 ms of a 0.6 ms step of a 1/8 shard
            // (HIP multiplexes streams over four hardware queues: with more than four streams in play a twin waiting for its event
            // blocks the streaming kernels queued behind it -- 2.4 -> 3.0 ms per step with six streams; four it is)
            const int ntwin = n_twins();
            for (int c = nhash - 1; c >= 0; c--) {
                const int l = (nhash - 1 - c) % lanes;
                hipStream_t lane = l ? lane_s[l] : stream;
                launch_stream_cls(c, lane);
                HIPCHK(hipEventRecord(ev_cls_done[c], lane));
                const int t = (nhash - 1 - c) % ntwin;
                hipStream_t tw = t ? twin_s[t - 1] : side;
                HIPCHK(hipStreamWaitEvent(tw, ev_cls_done[c], 0));
                launch_class(c, tw);
            }
            for (int l = 1; l < lanes; l++) {
                HIPCHK(hipEventRecord(lane_ev[l], lane_s[l]));
                HIPCHK(hipStreamWaitEvent(stream, lane_ev[l], 0));
            }
            for (int t = 0; t < ntwin - 1; t++) {
                HIPCHK(hipEventRecord(twin_ev[t], twin_s[t]));
                HIPCHK(hipStreamWaitEvent(stream, twin_ev[t], 0));
            }
            launch_stream_fix(stream);
            launch_big(side);
            HIPCHK(hipEventRecord(ev_join, side));
            HIPCHK(hipStreamWaitEvent(stream, ev_join, 0));
        } else if (use_side) {
            for (int c = nhash - 1; c >= first_side; c--) launch_class(c, side); // longest rows first
            launch_big(side);
            HIPCHK(hipEventRecord(ev_join, side));
            for (int c = 0; c < first_side; c++) launch_class(c, stream);
            HIPCHK(hipStreamWaitEvent(stream, ev_join, 0));
        } else {
            for (int c = nhash - 1; c >= 0; c--) launch_stream_cls(c, stream);
            launch_stream_fix(stream);
            for (int c = 0; c < nhash; c++) launch_class(c, stream);
            launch_big(stream);
            if (class_timing) HIPCHK(hipEventRecord(ev_cls[nlaunch], stream));
        }
        hipLaunchKernelGGL(k_scatter_mark_failed, dim3(cdiv(nrows, 256)), dim3(256), 0, stream, nrows, Llen.p, S.len.p, S.lead.p);
        HIPCHK(hipGetLastError());
        // the Schur rows start where their slots start
        HIPCHK(hipMemcpyAsync(S.start.p, sstart.p, ((size_t)nrows + 1) * sizeof(i64d), hipMemcpyDeviceToDevice, stream));
    }

    // ---- (3 + 4, fused) the Schur rows of `nrows` rows of M in one pass over the rows (fused.hpp).  S.ent must hold `scap` entries.
    // No host synchronisation; afterwards fetch_fused() tells how many rows were left to the general path (fused_fallback()).
    // (the kernel reaches W, wcol and its row records through 32-bit byte offsets)
    bool fused_ok() const
    {
        return use_fused && use_w && !force_lists && !want_idx && m < (1 << 24) && (u64d)(wbase + wcap + 64) * sizeof(int2) < 0xffffffffull &&
               (u64d)m * sizeof(int4) < 0xf0000000ull;
    }
    static constexpr int FZ_LOGT = 11, FZ_WPB = 5;
    // capacity of S the fused step wants for an estimated `entries` of Schur rows: every wave may leave a block unfinished
    i64 fused_capacity(i64 entries) const { return entries + entries / 50 + (i64)num_cu * 16 * (i64)FZ_SBLK + 64; }
    // The step: (1) every row is classified (k_fused_classify: the fused kernel's, or the general path's); (2) the fused kernel
    // takes its list on the round's stream while (3) the general path -- the plan along W / the multiplier lists, the hash-table
    // and streaming classes -- takes the other list on a stream of its own, its Schur rows behind `scap` in S.ent; (4) the finish
    // kernel merges the duplicates of the fused rows and writes the row arrays, the general rows are put under their slots.
    // known = false: the list sizes are read back after (1) (one host synchronisation) and the general path sizes its pools as it
    // goes (more of them, on its own stream: the fused kernel runs meanwhile); S.ent grows when the general rows need more than the
    // caller gave.  known = true (a plan that has run the step before): no host synchronisation at all.
    // Returns false when the round is not one for the fused kernel (it would take less than half of the rows): nothing was done.
    int fz_nfused = 0, fz_ngeneral = 0;
    int fz_nleft = 0;        // known = true: rows the fused kernel takes up and leaves (too many duplicate columns), as an earlier run of the step saw
    i64 fb_pool_left = 1 << 16;
    i64 fz_general_bound = 0;
    hipStream_t fb_stream = nullptr;
    hipEvent_t ev_fb_fork = nullptr, ev_fb_join = nullptr;
    DevBuf<int2> fb_fixbuf;
    DevBuf<int> fb_fixcnt;
    bool run_fused(const DevMat &M, const int *rows, int nrows, i64 scap, bool known)
    {
        S.n = nrows;
        S.m = m;
        nlaunch = 0;
        fb_ran = false;
        left_ran = false;
        rinfo.ensure((size_t)nrows + 1);
        S.start.ensure((size_t)nrows + 1);
        S.len.ensure((size_t)nrows + 1);
        S.lead.ensure((size_t)nrows + 1);
        S.orig.ensure((size_t)nrows + 1);
        fixbuf.ensure(((size_t)nrows + 1) * SFIX);
        fz_rec.ensure(2 * ((size_t)nrows + 1));
        fz_long_list.ensure((size_t)nrows + 1); // (the fused kernel's list)
        fz_gen_list.ensure((size_t)nrows + 1);
        fz_flag.ensure((size_t)nrows + 2);
        fz_pos.ensure((size_t)nrows + 2);
        fz_rej_list.ensure((size_t)nrows + 1);
        fz_work.ensure((size_t)2 * FZ_NWORK * FZ_WSTRIDE);
        fz_cursor.ensure(2);
        fz_counts.ensure(4);
        if (!fz_rare.p || fz_rare_lists[0] != fz_long_list.p || fz_rare_lists[1] != fz_rej_list.p) {
            fz_rare.ensure(1);
            FusedRare h;
            h.long_list = nullptr; // (no launch with larger tables: rows beyond the fused kernel's go to the general path)
            h.long_count = (fz_gint *)fz_counts.p;
            h.rej_list = (fz_gint *)fz_rej_list.p;
            h.rej_count = (fz_gint *)(fz_counts.p + 1);
            HIPCHK(hipMemcpyAsync(fz_rare.p, &h, sizeof h, hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));
            fz_rare_lists[0] = fz_long_list.p;
            fz_rare_lists[1] = fz_rej_list.p;
        }
        fz_ctr.ensure(NCTR);
        ctr.ensure(NCTR);
        class_count.ensure(NCLASS);
        if (!fb_stream) {
            HIPCHK(hipStreamCreateWithFlags(&fb_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ev_fb_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ev_fb_join, hipEventDisableTiming));
        }
        if (nrows > 0) {
            hipLaunchKernelGGL(k_gather_info, dim3(cdiv(nrows, 256)), dim3(256), 0, stream, nrows, rows, M.start.p, M.len.p, M.orig.p, rinfo.p);
            HIPCHK(hipGetLastError());
        }
        {
            static_assert(sizeof(RoundCounters) % 4 == 0, "cleared word by word");
            const int nctr_words = (int)(NCTR * sizeof(RoundCounters) / 4), nwork = 2 * FZ_NWORK * FZ_WSTRIDE;
            const int span = std::max(std::max(nctr_words, nwork), nrows + 1);
            hipLaunchKernelGGL(k_fused_reset, dim3(cdiv(span, 256)), dim3(256), 0, stream, nrows, fz_work.p, nwork, fz_cursor.p, fz_counts.p, 4, (unsigned *)fz_ctr.p,
                               nctr_words, fz_rec.p);
            HIPCHK(hipGetLastError());
        }
        if (nrows > 0) {
            ClassifyArgs c;
            c.nrows = nrows;
            c.rinfo = rinfo.p;
            c.ent = M.ent.p;
            c.pbits = pbits.p;
            c.wcol = wcol.p;
            c.flag = fz_flag.p;
            c.general_bound = fz_cursor.p + 1;
            c.cap = fz_cap(FZ_LOGT);
            c.free_cols = free_cols;
            hipLaunchKernelGGL(k_fused_classify, dim3(cdiv(((i64)nrows + 1) * 16, 256)), dim3(256), 0, stream, c);
            HIPCHK(hipGetLastError());
            // (lists in slot order: the general path's pools are cut into regions by workgroup, and a plan sizes them on one run for all)
            fz_scan.exclusive(fz_flag.p, fz_pos.p, (size_t)nrows + 1, stream);
            hipLaunchKernelGGL(k_fused_lists, dim3(cdiv((i64)nrows + 1, 256)), dim3(256), 0, stream, nrows, fz_flag.p, fz_pos.p, fz_long_list.p, fz_gen_list.p, fz_counts.p);
            HIPCHK(hipGetLastError());
        }
        if (!ev_classified) HIPCHK(hipEventCreateWithFlags(&ev_classified, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev_classified, stream));
        if (!known) {
            int cnt[4] = {0, 0, 0, 0};
            u64d cur[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(cnt, fz_counts.p, sizeof cnt, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(cur, fz_cursor.p, sizeof cur, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            fz_nfused = cnt[2];
            fz_ngeneral = cnt[3];
            fz_general_bound = (i64)cur[1];
            if (fz_nfused < nrows / 2) return false; // (rows of hundreds of entries, later rounds: the general path is the path)
            // room for the general rows behind the fused ones, before anything runs (S.ent must not move under a running kernel)
            S.ent.ensure((size_t)(scap + fz_general_bound + fz_general_bound / 8 + 16 * (i64)fz_ngeneral + 1024));
        }
        last_fused = true;
        fz_scap = scap;
        if (class_timing) HIPCHK(hipEventRecord(ev_fz[0], stream));
        if (fz_nfused > 0) {
            FusedArgs a;
            a.nrows = nrows;
            a.slots = fz_long_list.p;
            a.slot_count = fz_counts.p + 2;
            a.rinfo = rinfo.p;
            a.ent = M.ent.p;
            a.pbits = pbits.p;
            a.wcol = wcol.p;
            a.buf = UPN.p;
            a.Sent = S.ent.p;
            a.scap = (u64d)scap;
            a.scursor = fz_cursor.p;
            a.rec = fz_rec.p;
            a.fixbuf = fixbuf.p;
            a.rare = fz_rare.p;
            a.long_bound = 0;
            a.stamps = nullptr;
#ifdef SPASM_STAMPS
            stamps.ensure(NCLASS * 2 * NSTAMP);
            HIPCHK(hipMemsetAsync(stamps.p, 0, NCLASS * 2 * NSTAMP * sizeof(u64d), stream));
            a.stamps = stamps.p;
#endif
            a.buf_bytes = (unsigned)std::min<u64d>((u64d)UPN.n * sizeof(int2), 0xffffffffull);
            a.pbits_bytes = (unsigned)std::min<u64d>((u64d)pbits.n * sizeof(unsigned), 0xffffffffull);
            a.wcol_bytes = (unsigned)std::min<u64d>((u64d)wcol.n * sizeof(int4), 0xffffffffull);
            a.work = fz_work.p;
            a.ctr = fz_ctr.p;
            a.cls = NCLASS - 1; // (the statistics slot of the fix-up launch, which counts nothing there)
            a.free_cols = free_cols;
            a.F = F;
            const size_t lds = fused_lds_bytes<FZ_LOGT>(FZ_WPB);
            static int per_cu[16] = {0};
            int dev = 0;
            HIPCHK(hipGetDevice(&dev));
            int &pc = per_cu[(dev & 7) * 2 + (F.small ? 1 : 0)];
            if (!pc) {
                int nb = 0;
                if (F.small) {
                    HIPCHK(hipFuncSetAttribute((const void *)k_schur_fused<FZ_LOGT, FZ_WPB, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_schur_fused<FZ_LOGT, FZ_WPB, true, true>, FZ_WPB * 64, lds));
                } else {
                    HIPCHK(hipFuncSetAttribute((const void *)k_schur_fused<FZ_LOGT, FZ_WPB, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_schur_fused<FZ_LOGT, FZ_WPB, false, true>, FZ_WPB * 64, lds));
                }
                pc = std::max(nb, 1);
            }
            // every wave takes blocks of FZ_B rows until none is left: as many workgroups as are resident, fewer for few rows
            const int grid = std::max(1, std::min(cdiv(cdiv(fz_nfused, FZ_B), FZ_WPB), num_cu * pc));
            if (F.small) hipLaunchKernelGGL((k_schur_fused<FZ_LOGT, FZ_WPB, true, true>), dim3(grid), dim3(FZ_WPB * 64), lds, stream, a);
            else hipLaunchKernelGGL((k_schur_fused<FZ_LOGT, FZ_WPB, false, true>), dim3(grid), dim3(FZ_WPB * 64), lds, stream, a);
            HIPCHK(hipGetLastError());
        }
        if (class_timing) HIPCHK(hipEventRecord(ev_fz[1], stream));
        if (nrows > 0) {
            FusedFinishArgs fa;
            fa.nrows = nrows;
            fa.rec = fz_rec.p;
            fa.rinfo = rinfo.p;
            fa.fixbuf = fixbuf.p;
            fa.Sent = S.ent.p;
            fa.Sstart = S.start.p;
            fa.Slen = S.len.p;
            fa.Slead = S.lead.p;
            fa.Sorig = S.orig.p;
            fa.ctr = fz_ctr.p;
            fa.F = F;
            hipLaunchKernelGGL(k_fused_finish, dim3(cdiv(nrows, 64)), dim3(256), 0, stream, fa);
            HIPCHK(hipGetLastError());
        }
        if (class_timing) HIPCHK(hipEventRecord(ev_fz[2], stream));
        // the general path beside them (its launches were not queued before the fused kernel's: the host may have to wait for its
        // sizes, and the fused kernel runs meanwhile)
        if (fz_ngeneral > 0) {
            // (it only needs the lists: it waits for the classification, not for the launches queued behind it)
            hipStream_t keep = stream;
            HIPCHK(hipStreamWaitEvent(fb_stream, ev_classified, 0));
            stream = fb_stream;
            try {
                general_rows(M, rows, fz_gen_list.p, fz_ngeneral, scap, !known);
            } catch (...) {
                stream = keep;
                throw;
            }
            stream = keep;
            HIPCHK(hipEventRecord(ev_fb_join, fb_stream));
            HIPCHK(hipStreamWaitEvent(stream, ev_fb_join, 0));
            hipLaunchKernelGGL(k_merge_rej, dim3(cdiv(fz_ngeneral, 256)), dim3(256), 0, stream, fz_ngeneral, fz_gen_list.p, fb_start.p, fb_len.p, fb_lead.p, fb_orig.p,
                               (i64d)fb_base, S.start.p, S.len.p, S.lead.p, S.orig.p);
            HIPCHK(hipGetLastError());
            fb_ran = true;
        }
        if (known && fz_nleft > 0) {
            // the rows the fused kernel leaves (the same every time, in any order): through the general path behind everything else
            const i64 base = fz_ngeneral > 0 ? fb_base + fb_tot + 16 : scap;
            const i64 keep_pool = fb_pool, keep_tot = fb_tot, keep_base = fb_base;
            fb_pool = fb_pool_left;
            ctr_left.ensure(NCTR);
            class_count_left.ensure(NCLASS);
            std::swap(ctr, ctr_left); std::swap(class_count, class_count_left);
            try {
                general_rows(M, rows, fz_rej_list.p, fz_nleft, base, false);
            } catch (...) {
                std::swap(ctr, ctr_left); std::swap(class_count, class_count_left);
                fb_pool = keep_pool;
                throw;
            }
            std::swap(ctr, ctr_left); std::swap(class_count, class_count_left);
            fb_pool = keep_pool;
            left_ran = true;
            hipLaunchKernelGGL(k_merge_rej, dim3(cdiv(fz_nleft, 256)), dim3(256), 0, stream, fz_nleft, fz_rej_list.p, fb_start.p, fb_len.p, fb_lead.p, fb_orig.p, (i64d)fb_base,
                               S.start.p, S.len.p, S.lead.p, S.orig.p);
            HIPCHK(hipGetLastError());
            fb_tot = keep_tot;
            fb_base = keep_base;
        }
        return true;
    }

    // what the fused kernels did (synchronises): their statistics, rows left to the general path, space of S in use
    void fetch_fused()
    {
        hfz = read_counters(fz_ctr.p);
        int cnt[4] = {0, 0, 0, 0};
        u64d cur = 0;
        HIPCHK(hipMemcpyAsync(cnt, fz_counts.p, sizeof cnt, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(&cur, fz_cursor.p, sizeof cur, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        fz_nlong = cnt[0];
        fz_nrej = cnt[1]; // rows the fused kernel took up and had to leave (too many duplicate columns, no room in S)
        fz_used = (i64)cur;
        fz_rows = fz_nfused - fz_nrej;
        hctr = hfz;
        memset(hclass_count, 0, sizeof hclass_count);
#ifdef SPASM_STAMPS
        {
            std::vector<u64d> h(2 * NSTAMP);
            HIPCHK(hipMemcpy(h.data(), stamps.p, h.size() * sizeof(u64d), hipMemcpyDeviceToHost));
            static const char *names[6] = {"stages", "plan", "own", "groups", "rowend", "rotate"};
            const double waves = (double)h[NSTAMP], nr = (double)std::max(fz_rows, 1);
            fprintf(stderr, "[stamps] fused: rows %d waves %.0f: cycles per row:", fz_rows, waves);
            for (int i = 0; i < 6; i++) fprintf(stderr, " %s=%.0f", names[i], (double)h[i] / nr);
            fprintf(stderr, "\n");
        }
#endif
    }
    // the statistics of a pass of the general path (its counters `from`, its class counts) on top of what hctr holds (synchronises)
    void add_general(const RoundCounters *from, const int *class_count_dev)
    {
        const RoundCounters g = read_counters(from);
        int cc[NCLASS];
        HIPCHK(hipMemcpyAsync(cc, class_count_dev, NCLASS * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (g.lpool_overflow) throw EngineError("multiplier pool exhausted");
        if (g.scatter_overflow) throw EngineError("a hash table of the scatter kernel filled up (internal bound violated)");
        hctr.applications += g.applications;
        hctr.nnz_reduced += g.nnz_reduced;
        hctr.segments += g.segments;
        hctr.nonempty_out += g.nonempty_out;
        hctr.nnz_out += g.nnz_out;
        hctr.stream_redo += g.stream_redo;
        hctr.stream_fix += g.stream_fix;
        for (int c = 0; c < NCLASS - 1; c++) { hctr.class_ent[c] += g.class_ent[c]; hctr.class_seg[c] += g.class_seg[c]; }
        for (int c = 0; c < NCLASS; c++) hclass_count[c] += cc[c];
    }
    // after any Schur step (synchronises)
    bool fb_ran = false, left_ran = false;
    DevBuf<RoundCounters> ctr_left;  // the counters / class counts of the pass over the rows the fused kernel left
    DevBuf<int> class_count_left;
    void fetch_step()
    {
        if (!last_fused) { fetch_counters(); return; }
        fetch_fused();
        if (fb_ran) add_general(ctr.p, class_count.p);
        if (left_ran) add_general(ctr_left.p, class_count_left.p);
    }

    // `n` row slots (list `slots`) through the general path on the current `stream`: the plan along W / the multiplier lists, the
    // hash-table and streaming classes; their Schur rows go into S.ent from `base` on and, for now, under the general path's own
    // slot numbers 0 .. n-1 in fb_start / fb_len / fb_lead / fb_orig (k_merge_rej puts them under their slots).  sync = false: pools
    // and S are known to be large enough (a plan that has run this before), no host synchronisation.
    i64 fb_pool = 1 << 16, fb_base = 0, fb_tot = 0, fb_left_tot = 0;
    i64 fz_scap = 0;
    hipEvent_t ev_classified = nullptr;
    void general_rows(const DevMat &M, const int *rows, const int *slots, int n, i64 base, bool sync)
    {
        fz_rej_rows.ensure((size_t)n + 1);
        hipLaunchKernelGGL(k_rej_rows, dim3(cdiv(n, 256)), dim3(256), 0, stream, n, slots, rows, fz_rej_rows.p);
        HIPCHK(hipGetLastError());
        // the general path writes S.start / len / lead / orig, fixbuf / fixcnt by ITS slots 0 .. n-1: arrays of its own
        auto swap_rows = [&]() {
            std::swap(S.start, fb_start); std::swap(S.len, fb_len); std::swap(S.lead, fb_lead); std::swap(S.orig, fb_orig);
            std::swap(fixbuf, fb_fixbuf); std::swap(fixcnt, fb_fixcnt);
        };
        swap_rows();
        const int keep_n = S.n;
        const bool qk = quiet_known, lf = last_fused;
        quiet_known = false;
        gathered_n = -1;
        base = (base + 15) & ~(i64)15;
        s_ent_base = base;
        fb_base = base;
        try {
            if (sync) {
                const i64 tot = solve_phase(M, fz_rej_rows.p, nullptr, n, std::max<i64>(fb_pool, 64 * (i64)n));
                if (tot < 0) throw EngineError("the rows left to the general path do not fit the device memory");
                fb_pool = std::max<i64>(fb_pool, (i64)(pool_used() * 5 / 4) + 1024);
                fb_tot = tot;
                if (getenv("SPASM_AMD_FZDEBUG")) fprintf(stderr, "[fused] general rows %d: pool %lld records (used %llu, regions %d), %lld entries of S from %lld, S holds %zu\n", n, (long long)fb_pool, (unsigned long long)pool_used(), npool_active, (long long)tot, (long long)base, S.ent.n);
                if ((i64)S.ent.n < base + tot + 1) {
                    // (rare: the classification could not work out the streams of all rows) a larger S, once nothing writes the old one
                    HIPCHK(hipDeviceSynchronize());
                    DevBuf<int2> bigger;
                    bigger.alloc((size_t)(base + tot + tot / 8 + 1));
                    if (base > 0) HIPCHK(hipMemcpy(bigger.p, S.ent.p, (size_t)std::min<i64>(base, (i64)S.ent.n) * sizeof(int2), hipMemcpyDeviceToDevice));
                    S.ent = std::move(bigger); // (the rows already written keep their places: S.start holds offsets)
                }
            } else {
                alloc_solve(n, fb_pool);
                run_solve(M, fz_rej_rows.p, nullptr, n);
                run_bounds(n);
                if (getenv("SPASM_AMD_FZDEBUG")) { const RoundCounters c = read_counters(); fprintf(stderr, "[fused] general rows %d (no sync): pool %lld, used %llu, regions %d, region cap %llu, overflow %d, rejects %d\n", n, (long long)fb_pool, (unsigned long long)pool_used(), npool_active, (unsigned long long)region_cap, c.lpool_overflow, c.wplan_reject); }
            }
            run_scatter(M, fz_rej_rows.p, n);
        } catch (...) {
            s_ent_base = 0;
            quiet_known = qk;
            last_fused = lf;
            swap_rows();
            throw;
        }
        s_ent_base = 0;
        quiet_known = qk;
        last_fused = lf;
        gathered_n = -1;
        swap_rows();
        S.n = keep_n;
    }

    // rows the fused kernel took up and had to leave (fetch_fused counted them: too many duplicate columns for its lists, or no room
    // left in S): through the general path, behind everything else in S.ent.  Synchronises; their statistics are added to hctr.
    void fused_leftovers(const DevMat &M, const int *rows)
    {
        if (fz_nrej <= 0) return;
        const i64 base = fb_ran ? fb_base + fb_tot + 16 : fz_scap;
        HIPCHK(hipStreamSynchronize(stream));
        const i64 keep_pool = fb_pool, keep_tot = fb_tot, keep_base = fb_base;
        fb_pool = 1 << 16;
        ctr_left.ensure(NCTR);
        class_count_left.ensure(NCLASS);
        std::swap(ctr, ctr_left); std::swap(class_count, class_count_left);
        try {
            general_rows(M, rows, fz_rej_list.p, fz_nrej, base, true);
        } catch (...) {
            std::swap(ctr, ctr_left); std::swap(class_count, class_count_left);
            fb_pool = keep_pool;
            throw;
        }
        std::swap(ctr, ctr_left); std::swap(class_count, class_count_left);
        fb_pool_left = 4 * fb_pool; // (a plan runs these rows again, in another order: room to spare)
        fb_left_tot = fb_tot;
        fb_pool = keep_pool;
        hipLaunchKernelGGL(k_merge_rej, dim3(cdiv(fz_nrej, 256)), dim3(256), 0, stream, fz_nrej, fz_rej_list.p, fb_start.p, fb_len.p, fb_lead.p, fb_orig.p, (i64d)fb_base,
                           S.start.p, S.len.p, S.lead.p, S.orig.p);
        HIPCHK(hipGetLastError());
        fb_tot = keep_tot;
        fb_base = keep_base;
        left_ran = true;
        add_general(ctr_left.p, class_count_left.p);
        last_fused = true;
    }

    void fetch_counters()
    {
        hctr = read_counters();
        int *cc = hclass_count;
        HIPCHK(hipMemcpyAsync(cc, class_count.p, NCLASS * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
#ifdef SPASM_STAMPS
        {
            std::vector<u64d> h(NCLASS * 2 * NSTAMP);
            HIPCHK(hipMemcpy(h.data(), stamps.p, h.size() * sizeof(u64d), hipMemcpyDeviceToHost));
            static const char *names[NSTAMP] = {"prologue", "issue", "wait-loads", "own", "rounds", "remainder", "sweep", "-"};
            static const char *snames[NSTAMP] = {"top+own", "chunks", "next-req", "rowend", "reset", "rotate", "-", "-"};
            for (int c = NSTREAM0; c < NCLASS; c++) {
                const double waves = (double)h[(size_t)c * 2 * NSTAMP + NSTAMP];
                if (waves == 0) continue;
                fprintf(stderr, "[stamps] stream class %d rows %d waves %.0f: cycles per wave:", c - NSTREAM0, cc[c], waves);
                for (int i = 0; i < 6; i++) fprintf(stderr, " %s=%.0f", snames[i], (double)h[(size_t)c * 2 * NSTAMP + i] / waves);
                fprintf(stderr, "\n");
            }
            for (int c = 0; c < NHASHMAX; c++) {
                const double waves = (double)h[(size_t)c * 2 * NSTAMP + NSTAMP];
                if (waves == 0) continue;
                fprintf(stderr, "[stamps] class %d rows %d waves %.0f: cycles per wave:", c, cc[c], waves);
                for (int i = 0; i < 7; i++) fprintf(stderr, " %s=%.0f", names[i], (double)h[(size_t)c * 2 * NSTAMP + i] / waves);
                fprintf(stderr, "\n");
            }
        }
#endif
        if (hctr.lpool_overflow) throw EngineError("multiplier pool exhausted");
        if (hctr.scatter_overflow) throw EngineError("a hash table of the scatter kernel filled up (internal bound violated)");
    }
};

// ------------------------------------------------------------------------------------------------
// host CSR -> device matrix
// ------------------------------------------------------------------------------------------------
void upload_csr_strided(const struct spasm_csr *A, int row_lo, int row_hi, int stride, DevMat &M, hipStream_t s);

void upload_csr(const struct spasm_csr *A, int row_lo, int row_hi, DevMat &M, hipStream_t s)
{
    const int n = row_hi - row_lo;
    const i64 base = A->p[row_lo];
    const i64 nnz = A->p[row_hi] - base;
    M.n = n;
    M.m = A->m;
    M.start.ensure((size_t)n + 1);
    M.len.ensure((size_t)n + 1);
    M.lead.ensure((size_t)n + 1);
    M.orig.ensure((size_t)n + 1);
    M.ent.ensure((size_t)nnz + 1);
    DevBuf<i64d> dp;
    DevBuf<int> dj, dx;
    dp.alloc((size_t)n + 1);
    dj.alloc((size_t)nnz + 1);
    dx.alloc((size_t)nnz + 1);
    // row pointers relative to the shard
    std::vector<i64d> hp((size_t)n + 1);
    for (int i = 0; i <= n; i++) hp[(size_t)i] = A->p[row_lo + i] - base;
    HIPCHK(hipMemcpyAsync(dp.p, hp.data(), ((size_t)n + 1) * sizeof(i64d), hipMemcpyHostToDevice, s));
    if (nnz > 0) {
        HIPCHK(hipMemcpyAsync(dj.p, A->j + base, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, s));
        if (A->x) HIPCHK(hipMemcpyAsync(dx.p, A->x + base, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pack_entries, dim3(std::min<i64>(cdiv(nnz, 256), 65536)), dim3(256), 0, s, (i64d)nnz, zp_field_make(A->field->p), dj.p,
                           A->x ? dx.p : nullptr, M.ent.p);
        HIPCHK(hipGetLastError());
    }
    if (n > 0) {
        hipLaunchKernelGGL(k_pack_rows, dim3(cdiv(n, 256)), dim3(256), 0, s, n, row_lo, 1, dp.p, M.start.p, M.len.p, M.orig.p);
        HIPCHK(hipGetLastError());
        constexpr int TEAM = 8;
        if (A->x) hipLaunchKernelGGL((k_drop_zeros<TEAM>), dim3(cdiv((i64)n * TEAM, 256)), dim3(256), 0, s, n, M.start.p, M.len.p, M.ent.p);
        hipLaunchKernelGGL((k_row_lead<TEAM>), dim3(cdiv((i64)n * TEAM, 256)), dim3(256), 0, s, n, M.start.p, M.len.p, M.ent.p, M.lead.p);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(s)); // hp and the staging buffers go out of scope
}

// rows row_lo, row_lo + stride, ... < row_hi of A, gathered on the host and uploaded as one matrix
void upload_csr_strided(const struct spasm_csr *A, int row_lo, int row_hi, int stride, DevMat &M, hipStream_t s)
{
    if (stride <= 1) { upload_csr(A, row_lo, row_hi, M, s); return; }
    const int n = row_hi > row_lo ? (row_hi - row_lo + stride - 1) / stride : 0;
    std::vector<i64d> hp((size_t)n + 1, 0);
    for (int i = 0; i < n; i++) { const int g = row_lo + i * stride; hp[(size_t)i + 1] = hp[(size_t)i] + (A->p[g + 1] - A->p[g]); }
    const i64 nnz = hp[(size_t)n];
    std::vector<int> hj((size_t)nnz + 1), hx((size_t)nnz + 1);
    for (int i = 0; i < n; i++) {
        const int g = row_lo + i * stride;
        const i64 len = A->p[g + 1] - A->p[g];
        memcpy(hj.data() + hp[(size_t)i], A->j + A->p[g], sizeof(int) * (size_t)len);
        if (A->x) memcpy(hx.data() + hp[(size_t)i], A->x + A->p[g], sizeof(int) * (size_t)len);
    }
    M.n = n;
    M.m = A->m;
    M.start.ensure((size_t)n + 1);
    M.len.ensure((size_t)n + 1);
    M.lead.ensure((size_t)n + 1);
    M.orig.ensure((size_t)n + 1);
    M.ent.ensure((size_t)nnz + 1);
    DevBuf<i64d> dp;
    DevBuf<int> dj, dx;
    dp.alloc((size_t)n + 1);
    dj.alloc((size_t)nnz + 1);
    dx.alloc((size_t)nnz + 1);
    HIPCHK(hipMemcpyAsync(dp.p, hp.data(), ((size_t)n + 1) * sizeof(i64d), hipMemcpyHostToDevice, s));
    if (nnz > 0) {
        HIPCHK(hipMemcpyAsync(dj.p, hj.data(), (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, s));
        if (A->x) HIPCHK(hipMemcpyAsync(dx.p, hx.data(), (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pack_entries, dim3(std::min<i64>(cdiv(nnz, 256), 65536)), dim3(256), 0, s, (i64d)nnz, zp_field_make(A->field->p), dj.p,
                           A->x ? dx.p : nullptr, M.ent.p);
        HIPCHK(hipGetLastError());
    }
    if (n > 0) {
        hipLaunchKernelGGL(k_pack_rows, dim3(cdiv(n, 256)), dim3(256), 0, s, n, row_lo, stride, dp.p, M.start.p, M.len.p, M.orig.p);
        HIPCHK(hipGetLastError());
        constexpr int TEAM = 8;
        if (A->x) hipLaunchKernelGGL((k_drop_zeros<TEAM>), dim3(cdiv((i64)n * TEAM, 256)), dim3(256), 0, s, n, M.start.p, M.len.p, M.ent.p);
        hipLaunchKernelGGL((k_row_lead<TEAM>), dim3(cdiv((i64)n * TEAM, 256)), dim3(256), 0, s, n, M.start.p, M.len.p, M.ent.p, M.lead.p);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(s));
}

void require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) throw EngineError("no HIP device: the MI355X engine has no CPU fallback");
}

void check_input(const struct spasm_csr *A, const char *who)
{
    if (!A) throw EngineError(std::string(who) + ": NULL matrix");
    const i64 p = A->field->p;
    if (p <= 2 || p > 0xfffffffbLL) throw EngineError(std::string(who) + ": prime out of range (2 < p <= 0xfffffffb)");
    if (!A->x && A->p[A->n] > 0) throw EngineError(std::string(who) + ": matrix without values");
}

thread_local std::vector<spasm_amd_round_stats> g_last_rounds;
thread_local int g_multi_finish = 0; // how the last spasm_amd_echelonize_multi finished (spasm_amd_multi_last_finish)

i64 read_bytes_of(const RoundCounters &c, int m) { return 8 * (i64)c.nnz_reduced + 16 * (i64)c.segments + 4 * (i64)m; }

void fill_stats(spasm_amd_round_stats &st, const Round &R, int round, int rows_in, i64 nnz_in)
{
    memset(&st, 0, sizeof st);
    st.round = round;
    st.rows_in = rows_in;
    st.nnz_in = nnz_in;
    st.npiv = R.npiv;
    st.npiv_open = R.n_open;
    st.npiv_greedy = R.n_greedy;
    st.rows_out = R.hctr.nonempty_out;
    st.nnz_out = (i64)R.hctr.nnz_out;
    st.nnz_reduced = (i64)R.hctr.nnz_reduced;
    st.applications = (i64)R.hctr.applications;
    st.read_bytes = read_bytes_of(R.hctr, R.m);
    float ms = 0;
    if (hipEventElapsedTime(&ms, R.ev[0], R.ev[1]) == hipSuccess) st.ms_pivots = ms;
    if (hipEventElapsedTime(&ms, R.ev[1], R.ev[2]) == hipSuccess) st.ms_solve = ms;
    if (hipEventElapsedTime(&ms, R.ev[2], R.ev[3]) == hipSuccess) st.ms_scatter = ms;
    if (hipEventElapsedTime(&ms, R.ev[0], R.ev[3]) == hipSuccess) st.ms_total = ms;
    for (int i = 0; i < R.nlaunch; i++)
        if (R.class_timing && hipEventElapsedTime(&ms, R.ev_cls[i], R.ev_cls[i + 1]) == hipSuccess) st.ms_class[R.launch_cls[i]] = ms;
    for (int c = 0; c < NCLASS; c++) {
        st.rows_class[c] = R.hclass_count[c];
        st.ent_class[c] = (i64)R.hctr.class_ent[c];
        st.seg_class[c] = (i64)R.hctr.class_seg[c];
    }
    st.stream_fix = R.hctr.stream_fix;
    st.stream_redo = R.hctr.stream_redo;
    st.ms_uinv = R.ms_uinv;
    st.ms_w = R.ms_w;
    if (R.last_fused) {
        if (R.class_timing && hipEventElapsedTime(&ms, R.ev_fz[0], R.ev_fz[1]) == hipSuccess) st.ms_fused = ms;
        if (R.class_timing && hipEventElapsedTime(&ms, R.ev_fz[1], R.ev_fz[2]) == hipSuccess) st.ms_fused_fix = ms;
        st.rows_fused = R.fz_rows;
        st.ent_fused = (i64)R.hfz.class_ent[NCLASS - 1];
        st.seg_fused = (i64)R.hfz.class_seg[NCLASS - 1];
        st.rows_rejected = R.fz_nrej;
        st.s_entries_used = R.fz_used;
    }
    st.ms_levels = R.ms_levels;
    st.ms_w_sizing = R.ms_w_sizing;
    if (R.use_w) {
        if (R.ev_w[0] && hipEventElapsedTime(&ms, R.ev_w[0], R.ev_w[1]) == hipSuccess) st.ms_wbuild = ms;
        st.w_levels = R.depth + 1;
        st.w_entries = R.w_entries;
        st.w_long_rows = R.w_long_rows;
    }
    (void)hipGetLastError(); // (an event that was never recorded makes hipEventElapsedTime fail: not an error of the engine)
}

// malloc'ed, uninitialised, geometrically growing int array whose buffer ends up in the spasm_csr the caller frees: the entries
// of U go device -> this buffer -> caller, without a value-initialising resize (a memset of gigabytes) or a final copy
struct HostInts {
    int *ptr = nullptr;
    size_t n = 0, cap = 0;
    HostInts() {}
    HostInts(const HostInts &) = delete;
    HostInts &operator=(const HostInts &) = delete;
    ~HostInts() { free(ptr); }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    int *data() { return ptr; }
    int *grow(size_t count) // room for `count` more; returns where they go
    {
        if (n + count > cap) {
            const size_t want = std::max(n + count, cap + cap / 2 + 1024);
            int *q = (int *)realloc(ptr, want * sizeof(int));
            if (!q) throw EngineError("out of host memory for U");
            ptr = q;
            cap = want;
        }
        int *at = ptr + n;
        n += count;
        return at;
    }
    int *steal() { int *q = ptr; ptr = nullptr; n = cap = 0; return q; }
};

struct HostU {
    std::vector<i64> p;      // row pointers
    HostInts j, x;
    std::vector<int> pivcol; // pivot column of each row
    std::vector<int> orig;   // originating row of the input
    // rank only (spasm_amd_rank): the rows of U are counted, their entries never leave the device -- at config 5's scale they are
    // what outgrows the host (10^10 entries and more), not anything the device holds
    bool rank_only = false;
    i64 uncollected = 0;     // pivots counted without a row of U to show for them
};

// entries of pivot rows (device, {col,val} pairs) appended to the host arrays of U: split on the device, downloaded straight
// into the tails of U.j / U.x (was: a host copy of the pairs and 2 x nnz push_backs -- a third of config 2's echelonize)
void append_entries(HostU &U, const int2 *dent, i64 count, hipStream_t s)
{
    if (count <= 0 || U.rank_only) return;
    DevBuf<int> dj, dx;
    dj.alloc((size_t)count);
    dx.alloc((size_t)count);
    hipLaunchKernelGGL(k_split_ent, dim3((unsigned)std::min<i64>((count + 255) / 256, 65536)), dim3(256), 0, s, (i64d)count, dent, dj.p, dx.p);
    HIPCHK(hipGetLastError());
    int *hj = U.j.grow((size_t)count), *hx = U.x.grow((size_t)count);
    HIPCHK(hipMemcpyAsync(hj, dj.p, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hx, dx.p, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
}

// append the pivot rows of a round (device) to the host copy of U
void append_round_U(HostU &U, const Round &R, const DevMat &A, hipStream_t s)
{
    const int np = R.npiv;
    if (np == 0) return;
    std::vector<i64d> off((size_t)np + 1);
    std::vector<int> pc((size_t)np), pr((size_t)np);
    std::vector<int> orig((size_t)A.n);
    HIPCHK(hipMemcpyAsync(off.data(), R.uoff.p, ((size_t)np + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(pc.data(), R.pivcol.p, (size_t)np * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(pr.data(), R.pivrow.p, (size_t)np * sizeof(int), hipMemcpyDeviceToHost, s));
    if (A.n > 0) HIPCHK(hipMemcpyAsync(orig.data(), A.orig.p, (size_t)A.n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const i64 base = U.p.back();
    for (int k = 0; k < np; k++) {
        U.p.push_back(base + off[(size_t)k + 1]);
        U.pivcol.push_back(pc[(size_t)k]);
        U.orig.push_back(orig[(size_t)pr[(size_t)k]]);
    }
    append_entries(U, R.Ufull.p, R.utotal, s);
}

// ------------------------------------------------------------------------------------------------
// dense tail: leftmost-pivot elimination of the live part of `M`; its pivot rows are appended to U
// ------------------------------------------------------------------------------------------------
// cells the dense finish may hold: a third of the free device memory in bytes (beside it: the dense W of a slab of columns, the
// digit planes of a block, the rows of U it emits), at least 2^31 cells; elem = bytes per cell (dense_elem_bytes)
// The hand-off to the dense finish is decided by the SHAPE: at most 2^36 bytes of dense matrix (64 GiB: a quarter of an MI355X,
// beside the dense W of a slab of columns, the digit planes of a block and the rows of U it emits).  Free memory only vetoes
// (a shared or smaller device): the caller then stays with sparse rounds, and the veto is logged, since with "FL on columns" the
// pivot columns depend on where the sparse rounds stop.
i64 dense_max_entries(int elem = 4)
{
    const i64 by_shape = ((i64)1 << 36) / std::max(elem, 1);
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return by_shape;
    const i64 by_memory = (i64)(fr / 3) / std::max(elem, 1);
    if (by_memory < by_shape) {
        static bool said = false;
        if (!said) spasm_logf("[echelonize] dense finish capped by free device memory (%.1f GiB free): %lld cells instead of %lld\n", (double)fr / 1073741824.0,
                              (long long)by_memory, (long long)by_shape);
        said = true;
        return by_memory;
    }
    return by_shape;
}

// which element type the dense matrix of a finish gets: bytes / shorts when the int8 path of dense.hpp will take it (its block
// updates are bound by reading and writing D), ints otherwise (f64 panels, rank-1 updates)
int dense_elem_bytes(const ZpField &F, i64 R)
{
    const char *force64 = getenv("SPASM_AMD_DENSE_F64"); // diagnostics: the f64 panels for every prime
    if (!F.small || (force64 && atoi(force64))) return 4;
    if ((R + 255) / 256 > 65536) return 4; // more rows than the panel kernel takes (64 per thread of 256 workgroups)
    return F.p <= 255 ? 1 : 2;
}

// The pivot rows of the dense columns [c_lo, c_hi) of an eliminated (or partly eliminated: the pivot rows of those columns are final)
// dense matrix, appended to U.  Returns how many.
// (device buffers of the extraction, kept between calls: hipFree waits for the whole device, which would end the overlap)
struct ExtractWork {
    Scanner scan;
    DevBuf<int> pflag, pscan, pivcol, porig, dj, dx;
    DevBuf<i64d> ulen, uoff;
    DevBuf<int2> Ufull;
    // Device -> pageable host memory runs at ~7 GB/s (the runtime stages it through its own pinned chunks, one host thread copying):
    // the 21 GB of U of a 73k x 73k finish took 3.1 s, longer than the elimination.  Two pinned buffers of our own instead: the DMA of
    // piece i + 1 runs while piece i is copied out by all host threads.
    static constexpr size_t PIN_BYTES = (size_t)64 << 20;
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t pev[2] = {nullptr, nullptr};
    ExtractWork() {}
    ExtractWork(const ExtractWork &) = delete;
    ~ExtractWork()
    {
        for (int b = 0; b < 2; b++) {
            if (pin[b]) (void)hipHostFree(pin[b]);
            if (pev[b]) (void)hipEventDestroy(pev[b]);
        }
    }
    void d2h(void *dst, const void *src, size_t bytes, hipStream_t s)
    {
        if (bytes < ((size_t)8 << 20)) {
            HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            return;
        }
        for (int b = 0; b < 2; b++) {
            if (!pin[b]) HIPCHK(hipHostMalloc(&pin[b], PIN_BYTES, hipHostMallocDefault));
            if (!pev[b]) HIPCHK(hipEventCreateWithFlags(&pev[b], hipEventDisableTiming));
        }
        const size_t npieces = (bytes + PIN_BYTES - 1) / PIN_BYTES;
        auto piece = [&](size_t i) { return std::min(PIN_BYTES, bytes - i * PIN_BYTES); };
        HIPCHK(hipMemcpyAsync(pin[0], src, piece(0), hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(pev[0], s));
        for (size_t i = 0; i < npieces; i++) {
            const int b = (int)(i & 1);
            if (i + 1 < npieces) {
                HIPCHK(hipMemcpyAsync(pin[b ^ 1], (const char *)src + (i + 1) * PIN_BYTES, piece(i + 1), hipMemcpyDeviceToHost, s));
                HIPCHK(hipEventRecord(pev[b ^ 1], s));
            }
            HIPCHK(hipEventSynchronize(pev[b]));
            const size_t n = piece(i);
            char *out = (char *)dst + i * PIN_BYTES;
            const char *in = (const char *)pin[b];
            const size_t part = (size_t)4 << 20;
            const long long nparts = (long long)((n + part - 1) / part);
#pragma omp parallel for schedule(static) num_threads(8)
            for (long long q = 0; q < nparts; q++) memcpy(out + (size_t)q * part, in + (size_t)q * part, std::min(part, n - (size_t)q * part));
        }
    }
};

template <typename DT>
int dense_extract_range(const DT *Dp, int C, i64 ldc, const int *pivrow_of_col, int c_lo, int c_hi, const int *clist, const int *row_orig, HostU &U, ExtractWork &W,
                        hipStream_t s)
{
    const int nc = c_hi - c_lo;
    if (nc <= 0) return 0;
    Scanner &scan = W.scan;
    DevBuf<int> &pflag = W.pflag, &pscan = W.pscan;
    pflag.ensure((size_t)nc + 1); pscan.ensure((size_t)nc + 1);
    hipLaunchKernelGGL(k_flag_nonneg, dim3(cdiv((i64)nc + 1, 256)), dim3(256), 0, s, nc, pivrow_of_col + c_lo, pflag.p);
    HIPCHK(hipGetLastError());
    scan.exclusive(pflag.p, pscan.p, (size_t)nc + 1, s);
    int npd = 0;
    HIPCHK(hipMemcpyAsync(&npd, pscan.p + nc, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (npd == 0) return 0;
    if (U.rank_only) { U.uncollected += npd; return npd; }
    DevBuf<i64d> &ulen = W.ulen, &uoff = W.uoff;
    ulen.ensure((size_t)npd + 1024); uoff.ensure((size_t)npd + 1024);
    HIPCHK(hipMemsetAsync(ulen.p, 0, ((size_t)npd + 1) * sizeof(i64d), s));
    hipLaunchKernelGGL((k_dense_count<DT>), dim3(nc), dim3(256), 0, s, C, Dp, (i64d)ldc, pivrow_of_col, pscan.p, ulen.p, c_lo);
    HIPCHK(hipGetLastError());
    scan.exclusive(ulen.p, uoff.p, (size_t)npd + 1, s);
    i64d tot = 0;
    HIPCHK(hipMemcpyAsync(&tot, uoff.p + npd, sizeof tot, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    DevBuf<int2> &Ufull = W.Ufull;
    DevBuf<int> &pivcol = W.pivcol, &porig = W.porig;
    // (a quarter more than asked for: a later call that needs a little more must not free -- hipFree waits for the whole device)
    if ((size_t)tot + 1 > Ufull.n) Ufull.alloc((size_t)tot + (size_t)tot / 4 + 1024);
    pivcol.ensure((size_t)npd + 1024); porig.ensure((size_t)npd + 1024);
    hipLaunchKernelGGL((k_dense_emit<DT>), dim3(nc), dim3(64), 0, s, C, Dp, (i64d)ldc, pivrow_of_col, pscan.p, uoff.p, clist, row_orig, Ufull.p, pivcol.p, porig.p, c_lo);
    HIPCHK(hipGetLastError());
    std::vector<i64d> off((size_t)npd + 1);
    std::vector<int> pc((size_t)npd), po((size_t)npd);
    HIPCHK(hipMemcpyAsync(off.data(), uoff.p, ((size_t)npd + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(pc.data(), pivcol.p, (size_t)npd * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(po.data(), porig.p, (size_t)npd * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const i64 base = U.p.back();
    for (int k = 0; k < npd; k++) {
        U.p.push_back(base + off[(size_t)k + 1]);
        U.pivcol.push_back(pc[(size_t)k]);
        U.orig.push_back(po[(size_t)k]);
    }
    if (tot > 0) {
        if ((size_t)tot > W.dj.n) { W.dj.alloc((size_t)tot + (size_t)tot / 4 + 1024); W.dx.alloc(W.dj.n); }
        hipLaunchKernelGGL(k_split_ent, dim3((unsigned)std::min<i64>(((i64)tot + 255) / 256, 65536)), dim3(256), 0, s, (i64d)tot, Ufull.p, W.dj.p, W.dx.p);
        HIPCHK(hipGetLastError());
        int *hj = U.j.grow((size_t)tot), *hx = U.x.grow((size_t)tot);
        W.d2h(hj, W.dj.p, (size_t)tot * sizeof(int), s);
        W.d2h(hx, W.dx.p, (size_t)tot * sizeof(int), s);
    }
    return npd;
}

// The rows of U of a dense elimination on their way to the host WHILE it runs: the pivot rows of a block of columns are final when
// the block is done (its pivot rows were updated right of the block last), so the block before the one the device works on is
// extracted and copied on a stream of its own (a 10^9-entry U took as long to download as the elimination took: 1.0 of 1.9 s for
// config 5 at 1/5).  The host blocks in the copies; the device has the current block's launches queued by then.
struct UStreamer {
    const void *Dp = nullptr;
    int C = 0;
    i64 ldc = 0;
    const int *pivrow_of_col = nullptr, *clist = nullptr, *row_orig = nullptr;
    HostU *U = nullptr;
    hipStream_t s2 = nullptr;
    hipEvent_t ev = nullptr;
    ExtractWork work;
    int done_to = 0, found = 0;
    double seconds = 0;
    UStreamer() {}
    UStreamer(const UStreamer &) = delete;
    ~UStreamer()
    {
        if (ev) (void)hipEventDestroy(ev);
        if (s2) (void)hipStreamDestroy(s2);
    }
    void init(const void *Dp_, int C_, i64 ldc_, const int *pc, const int *clist_, const int *row_orig_, HostU &U_)
    {
        Dp = Dp_; C = C_; ldc = ldc_; pivrow_of_col = pc; clist = clist_; row_orig = row_orig_; U = &U_;
        HIPCHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    // everything queued on `main` so far finishes the columns below c_hi: remember the point ..
    void mark(hipStream_t main) { HIPCHK(hipEventRecord(ev, main)); }
    // .. and (later, with more work queued on the main stream) take the pivot rows of [done_to, c_hi)
    template <typename DT> void take(int c_hi)
    {
        if (c_hi <= done_to) return;
        const double t0 = spasm_wtime();
        HIPCHK(hipStreamWaitEvent(s2, ev, 0));
        found += dense_extract_range((const DT *)Dp, C, ldc, pivrow_of_col, done_to, c_hi, clist, row_orig, *U, work, s2);
        done_to = c_hi;
        seconds += spasm_wtime() - t0;
    }
};

template <typename DT>
bool dense_eliminate_i8(DevBuf<DT> &D, int R, int C, i64 ldc, const ZpField &F, DevBuf<int> &pivrow_of_col, hipStream_t s, UStreamer *us = nullptr)
{
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        HIPCHK(hipGetDevice(&dev));
        HIPCHK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int ND = F.p <= 255 ? 1 : 2;
    int KB = 1024;
    if (const char *e = getenv("SPASM_AMD_DENSE_KB")) KB = std::min(2048, std::max(64, atoi(e) / 64 * 64)); // tests: several blocks on small matrices
    // rows per workgroup of the panel kernel: a multiple of 64; resident in LDS when they fit (144 KB: 2304 rows of bytes for
    // p < 2^8, 1152 rows of shorts otherwise)
    int G = num_cu;
    int chunk = (int)((((i64)R + G - 1) / G + 63) / 64 * 64);
    if (chunk > 65536) return false; // (64 rows per thread at most)
    G = (int)(((i64)R + chunk - 1) / chunk);
    const int Rp = G * chunk;
    const int xbytes = ND == 1 ? 1 : 2;
    const int lds_rows = 147456 / (DP_W * xbytes);
    const char *force_global = getenv("SPASM_AMD_PANEL_GLOBAL"); // tests: the in-place variant on small matrices
    const bool inlds = chunk <= lds_rows && !(force_global && atoi(force_global));
    // more rows than fit: the first num_cu * res_chunk rows stay LDS-resident and elect the pivots, the others follow (dense.hpp,
    // k_panel_follow).  SPASM_AMD_PANEL_RES_ROWS (tests): rows per workgroup that count as fitting.
    int res_chunk = lds_rows / 64 * 64;
    if (const char *e = getenv("SPASM_AMD_PANEL_RES_ROWS")) res_chunk = std::min(res_chunk, std::max(64, atoi(e) / 64 * 64));
    const char *no_follow = getenv("SPASM_AMD_PANEL_NO_FOLLOW"); // A/B: the in-place variant for tall matrices, as before
    int G_res = num_cu;
    if (const char *e = getenv("SPASM_AMD_PANEL_RES_WGS")) G_res = std::min(num_cu, std::max(1, atoi(e))); // tests: resident workgroups
    const i64 R_res = (i64)G_res * res_chunk; // (< Rp when tall)
    const bool tall = R_res < (i64)R && !(force_global && atoi(force_global)) && !(no_follow && atoi(no_follow));
    const int fol_chunk = std::min(res_chunk, (int)((147456 - (int)sizeof(int) * (DP_W * DP_W + DP_W) - 1024) / (DP_W * xbytes) / 64 * 64));
    const int G_fol = tall ? (int)((Rp - R_res + fol_chunk - 1) / fol_chunk) : 0;
    DevBuf<int> follow_flag;
    if (tall) follow_flag.alloc(1);
    int redone = 0, npanels_done = 0;
    const int Cp = (int)ldc + 256; // (the GEMM stages whole tiles of 128 or 256 columns of Ut, starting at any multiple of 64)
    DevBuf<DT> P;
    DevBuf<int> seq, candrow, invtab;
    DevBuf<signed char> Fd, Ut;
    DevBuf<PanelInfo> info;
    DevBuf<PanelSync> sync;
    DevBuf<DenseState> st;
    const int npanel = KB / DP_W;
    const i64 fplane = (i64)Rp * KB, uplane = (i64)Cp * KB;
    P.alloc((size_t)DP_W * (size_t)Rp);
    seq.alloc((size_t)Rp);
    candrow.alloc((size_t)2 * std::max(G, num_cu) * DP_REC);
    Fd.alloc((size_t)ND * (size_t)fplane);
    Ut.alloc((size_t)ND * (size_t)uplane);
    info.alloc((size_t)npanel);
    sync.alloc(1);
    st.alloc(1);
    st.zero(s);
    DevBuf<unsigned long long> stamps; // diagnostics: phase times of the panel kernel's columns (workgroup 0), 100 MHz ticks
    if (const char *e = getenv("SPASM_AMD_PANEL_STAMPS")) if (atoi(e)) { stamps.alloc(DP_W * 8); stamps.zero(s); }
    invtab.alloc((size_t)F.p);
    hipLaunchKernelGGL(k_inv_table, dim3(cdiv(F.p, 256)), dim3(256), 0, s, (int)F.p, invtab.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(seq.p, 0xff, (size_t)Rp * sizeof(int), s));
    HIPCHK(hipMemsetAsync(pivrow_of_col.p, 0xff, ((size_t)C + 1) * sizeof(int), s));
    const size_t lds = inlds ? (size_t)chunk * DP_W * (size_t)xbytes : 0;
    static bool attr_done_dev[kMaxDev] = {false}; // (per device, and per element type: this function is a template)
    bool &attr_done = attr_done_dev[current_device()];
    if (!attr_done) {
        HIPCHK(hipFuncSetAttribute((const void *)k_panel_lu<true, 1024, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456));
        HIPCHK(hipFuncSetAttribute((const void *)k_panel_follow<1024, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456 - (int)sizeof(int) * (DP_W * DP_W + DP_W) - 1024));
        attr_done = true;
    }
    // The panel kernel's own grid barrier needs every workgroup of a launch resident at once (a plain launch does not check it, as
    // the cooperative launch did): ask the runtime how many fit and refuse cleanly instead of spinning into the barrier's timeout.
    {
        auto fits = [&](const void *fn, int wgs, size_t dyn, const char *what) {
            int per_cu = 0;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 1024, dyn));
            if ((i64)wgs > (i64)per_cu * num_cu)
                throw EngineError(std::string("dense finish: the ") + what + " panel kernel needs " + std::to_string(wgs) + " resident workgroups, the device holds " +
                                  std::to_string((i64)per_cu * num_cu));
        };
        const void *fn_lds = (const void *)k_panel_lu<true, 1024, DT>, *fn_glb = (const void *)k_panel_lu<false, 1024, DT>;
        if (tall) {
            fits(fn_lds, G_res, (size_t)res_chunk * DP_W * (size_t)xbytes, "LDS-resident");
            fits(fn_glb, G, 0, "in-place");
        } else
            fits(inlds ? fn_lds : fn_glb, G, lds, inlds ? "LDS-resident" : "in-place");
    }
    const char *glds_env = getenv("SPASM_AMD_GEMM_GLDS"); // A/B: the 256 x 256 LDS-DMA kernel for the large one-digit updates
    const bool glds_tile = glds_env && atoi(glds_env) != 0;
    if constexpr (std::is_same<DT, signed char>::value) {
        if (glds_tile) HIPCHK(hipFuncSetAttribute((const void *)k_gemm_i8_glds<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, GL_LDS_BYTES));
    }
    auto gemm = [&](int ja, int jb, int k0, int K, const int *rows, int nrows) {
        if (jb <= ja || K <= 0) return;
        // (1-D grid: the kernel orders the tiles itself, in bands of row tiles; a partial last band has fewer row tiles, and its
        // tiles still number rows_in_band * ntn, so the total is simply ntm * ntn)
        const int ntm = cdiv(rows ? nrows : R, 128);
        if (ND == 1 && glds_tile && !rows && K >= 256 && jb - ja >= 1024 && R >= 4096 && std::is_same<DT, signed char>::value) {
            // the large updates of the one-digit finish: 256 x 256 tile, operands by LDS-DMA, loads in flight across barriers
            if constexpr (std::is_same<DT, signed char>::value) {
                const int ntm2 = cdiv(R, GL_BM), ntn2 = cdiv(jb - ja, GL_BN);
                hipLaunchKernelGGL((k_gemm_i8_glds<DT>), dim3((unsigned)((i64)ntm2 * ntn2)), dim3(512), GL_LDS_BYTES, s, R, ja, jb, k0, K, F, D.p, (i64d)ldc, seq.p, Fd.p, Ut.p,
                                   KB, ntm2, ntn2);
            }
        } else if (ND == 1) {
            const int ntn = cdiv(jb - ja, 128);
            hipLaunchKernelGGL((k_gemm_i8<1, 2, 2, 2, 2, DT>), dim3((unsigned)((i64)ntm * ntn)), dim3(256), 0, s, R, ja, jb, k0, K, F, D.p, (i64d)ldc, seq.p, rows, nrows,
                               Fd.p, (i64d)fplane, Ut.p, (i64d)uplane, KB, ntm, ntn);
        } else {
            const int ntn = cdiv(jb - ja, 64);
            hipLaunchKernelGGL((k_gemm_i8<2, 4, 1, 1, 2, DT>), dim3((unsigned)((i64)ntm * ntn)), dim3(256), 0, s, R, ja, jb, k0, K, F, D.p, (i64d)ldc, seq.p, rows, nrows,
                               Fd.p, (i64d)fplane, Ut.p, (i64d)uplane, KB, ntm, ntn);
        }
    };
    auto trsm = [&](int q, int ja, int jb) {
        if (jb <= ja) return;
        if (ND == 1) hipLaunchKernelGGL((k_trsm_i8<1, DT>), dim3(cdiv(jb - ja, 64)), dim3(64), 0, s, ja, jb, F, D.p, (i64d)ldc, info.p + q, Ut.p, (i64d)uplane, KB, q * DP_W);
        else hipLaunchKernelGGL((k_trsm_i8<2, DT>), dim3(cdiv(jb - ja, 64)), dim3(64), 0, s, ja, jb, F, D.p, (i64d)ldc, info.p + q, Ut.p, (i64d)uplane, KB, q * DP_W);
    };
    int marked_to = 0; // columns whose pivot rows are final once the event of the streamer has passed
    for (int b0 = 0; b0 < C; b0 += KB) {
        const int b1 = std::min(b0 + KB, C);
        HIPCHK(hipMemsetAsync(Fd.p, 0, (size_t)ND * (size_t)fplane, s));
        HIPCHK(hipMemsetAsync(Ut.p, 0, (size_t)ND * (size_t)uplane, s));
        int q = 0;
        for (int c0 = b0; c0 < b1; c0 += DP_W, q++) {
            const int c1 = std::min(c0 + DP_W, b1), w = c1 - c0;
            hipLaunchKernelGGL((k_panel_load<DT>), dim3(Rp / 64), dim3(256), 0, s, R, Rp, c0, w, D.p, (i64d)ldc, P.p, sync.p);
            {
                int a_Rp = Rp, a_chunk = tall ? res_chunk : chunk, a_w = w, a_c0 = c0;
                ZpField a_F = F;
                DT *a_P = P.p;
                int *a_seq = seq.p, *a_pc = pivrow_of_col.p, *a_cand = candrow.p;
                PanelInfo *a_info = info.p + q;
                PanelSync *a_sy = sync.p;
                DenseState *a_st = st.p;
                const int *a_inv = invtab.p;
                unsigned long long *a_stamps = stamps.p;
                void *args[] = {&a_Rp, &a_chunk, &a_w, &a_c0, &a_F, &a_P, &a_seq, &a_pc, &a_info, &a_sy, &a_cand, &a_st, &a_inv, &a_stamps};
                const void *fn_lds = (const void *)k_panel_lu<true, 1024, DT>, *fn_glb = (const void *)k_panel_lu<false, 1024, DT>;
                // A PLAIN launch: G <= one workgroup per CU (checked against the runtime's occupancy figure above), so the grid is
                // resident as a whole once it has started and the kernel's own barrier (dense.hpp: panel_grid_barrier, bounded
                // spins) is enough.  The U streamer's kernels (scan, k_dense_count, k_dense_emit, k_split_ent on its stream s2) do
                // run beside it: they are short, use no LDS to speak of and end on their own, so they can delay the start of a
                // panel workgroup, never hold one out for good; two ELIMINATIONS on one device at once can (each takes a share of
                // the CUs and waits for the rest): ranks that share a device take turns (sharded.py: the replicated finish and dshard_candidates).
                // hipLaunchCooperativeKernel bought nothing but its launch-time size check, cost ~17 us per panel, and its dedicated
                // HSA queue made every rocprofv3-profiled process die in exit(): libamdhip64's exit handler tears that queue down
                // inside libhsa-runtime64 after rocprofiler-sdk has finalised its queue interception (tools/segv_probe.sh,
                // profiles/r03_exit_sigsegv_backtrace.txt -- no frame of this library in the trace).
                // (the follow path gives up for the rest of this elimination once it has misfired four times and more often than
                // not: every misfire costs the resident kernel, the followers' pass and a read-back on top of the in-place panel --
                // residuals of a weak first slab, whose pivots sit anywhere in 2.7M rows: 639 of 648 panels redone)
                const bool follow_now = tall && !(redone >= 4 && 2 * redone > npanels_done);
                if (tall && !follow_now) a_chunk = chunk;
                if (!follow_now) {
                    HIPCHK(hipLaunchKernel(inlds ? fn_lds : fn_glb, dim3(G), dim3(1024), args, (size_t)lds, s));
                } else {
                    // the pivots among the resident rows, then the followers; a follower that should have been a pivot: redo in place
                    HIPCHK(hipMemsetAsync(follow_flag.p, 0, sizeof(int), s));
                    HIPCHK(hipLaunchKernel(fn_lds, dim3(G_res), dim3(1024), args, (size_t)res_chunk * DP_W * (size_t)xbytes, s));
                    hipLaunchKernelGGL((k_panel_follow<1024, DT>), dim3(G_fol), dim3(1024), (size_t)fol_chunk * DP_W * (size_t)xbytes, s, Rp, (int)R_res, fol_chunk, w, F, P.p,
                                       seq.p, info.p + q, follow_flag.p);
                    HIPCHK(hipGetLastError());
                    int fl = 0;
                    HIPCHK(hipMemcpyAsync(&fl, follow_flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
                    HIPCHK(hipStreamSynchronize(s));
                    npanels_done++;
                    if (fl) {
                        redone++;
                        hipLaunchKernelGGL(k_panel_undo, dim3(1), dim3(DP_W), 0, s, c0, info.p + q, seq.p, pivrow_of_col.p, st.p);
                        hipLaunchKernelGGL((k_panel_load<DT>), dim3(Rp / 64), dim3(256), 0, s, R, Rp, c0, w, D.p, (i64d)ldc, P.p, sync.p);
                        HIPCHK(hipGetLastError());
                        a_chunk = chunk;
                        HIPCHK(hipLaunchKernel(fn_glb, dim3(G), dim3(1024), args, 0, s));
                    }
                }
            }
            if (ND == 1) hipLaunchKernelGGL((k_panel_store<1, DT>), dim3(Rp / 64), dim3(256), 0, s, R, Rp, c0, w, F, P.p, seq.p, D.p, (i64d)ldc, info.p + q, Fd.p, (i64d)fplane, KB, q * DP_W);
            else hipLaunchKernelGGL((k_panel_store<2, DT>), dim3(Rp / 64), dim3(256), 0, s, R, Rp, c0, w, F, P.p, seq.p, D.p, (i64d)ldc, info.p + q, Fd.p, (i64d)fplane, KB, q * DP_W);
            // inside the block: the panel's pivot rows, then everybody else, K = 64
            trsm(q, c1, b1);
            gemm(c1, b1, q * DP_W, DP_W, nullptr, 0);
            HIPCHK(hipGetLastError());
        }
        // right of the block: the pivot rows panel by panel (those of earlier panels first applied through the GEMM), then all others
        const int npan = q;
        for (int t = 0; t < npan; t++) {
            if (t > 0) gemm(b1, C, 0, t * DP_W, (const int *)((const char *)(info.p + t) + offsetof(PanelInfo, row)), DP_W);
            trsm(t, b1, C);
        }
        gemm(b1, C, 0, npan * DP_W, nullptr, 0);
        HIPCHK(hipGetLastError());
        if (us) {
            // this block is queued as a whole: the rows of U of the block BEFORE it go to the host now (the host blocks in the copy,
            // the device works through this block meanwhile); then the event that says this block is done
            if (marked_to > us->done_to) us->template take<DT>(marked_to);
            us->mark(s);
            marked_to = b1;
        }
    }
    DenseState hst;
    HIPCHK(hipMemcpyAsync(&hst, st.p, sizeof hst, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (hst.pad) throw EngineError("dense finish: a grid barrier of the panel kernel timed out (the device is shared with another process?)");
    if (tall) spasm_logf("[echelonize/dense] tall panels: %lld resident rows elect, %lld follow; %d of %d panels redone in place%s\n", (long long)R_res,
                         (long long)R - (long long)R_res, redone, npanels_done,
                         redone >= 4 && 2 * redone > npanels_done ? ", the others in place from the start" : "");
    if (stamps.p) {
        std::vector<unsigned long long> h(DP_W * 8);
        HIPCHK(hipMemcpy(h.data(), stamps.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double acc[6] = {0, 0, 0, 0, 0, 0};
        int n = 0;
        for (int c = 0; c + 1 < DP_W; c++) {
            if (!h[(size_t)c * 8 + 5] || !h[(size_t)(c + 1) * 8]) continue;
            for (int k = 0; k < 5; k++) acc[k] += (double)(h[(size_t)c * 8 + k + 1] - h[(size_t)c * 8 + k]);
            acc[5] += (double)(h[(size_t)(c + 1) * 8] - h[(size_t)c * 8 + 5]);
            n++;
        }
        if (n) fprintf(stderr, "[panel stamps] last panel, %d columns, us per column: scan %.2f  record %.2f  barrier %.2f  pivot row %.2f  eliminate %.2f  loop %.2f\n", n,
                       acc[0] / n / 100, acc[1] / n / 100, acc[2] / n / 100, acc[3] / n / 100, acc[4] / n / 100, acc[5] / n / 100);
    }
    return true;
}

// The pivot rows of an eliminated dense matrix (pivrow_of_col[c] = row of D that is the pivot row of its column c, -1: none), as
// sparse rows appended to U: column c of D is column clist[c] of the matrix, row r comes from row row_orig[r] of the input.
// *seconds (if given) receives the time spent.  Returns the number of rows appended.
template <typename DT>
int dense_extract_U(const DT *Dp, int C, i64 ldc, const int *pivrow_of_col, const int *clist, const int *row_orig, HostU &U, hipStream_t s)
{
    Scanner scan;
    DevBuf<int> pflag, pscan;
    pflag.alloc((size_t)C + 1); pscan.alloc((size_t)C + 1);
    hipLaunchKernelGGL(k_flag_nonneg, dim3(cdiv((i64)C + 1, 256)), dim3(256), 0, s, C, pivrow_of_col, pflag.p);
    HIPCHK(hipGetLastError());
    scan.exclusive(pflag.p, pscan.p, (size_t)C + 1, s);
    int npd = 0;
    HIPCHK(hipMemcpyAsync(&npd, pscan.p + C, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (npd == 0) return 0;
    if (U.rank_only) { U.uncollected += npd; return npd; }
    DevBuf<i64d> ulen, uoff;
    ulen.alloc((size_t)npd + 1); uoff.alloc((size_t)npd + 1);
    ulen.zero(s);
    hipLaunchKernelGGL((k_dense_count<DT>), dim3(C), dim3(256), 0, s, C, Dp, (i64d)ldc, pivrow_of_col, pscan.p, ulen.p);
    HIPCHK(hipGetLastError());
    scan.exclusive(ulen.p, uoff.p, (size_t)npd + 1, s);
    i64d tot = 0;
    HIPCHK(hipMemcpyAsync(&tot, uoff.p + npd, sizeof tot, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    DevBuf<int2> Ufull;
    DevBuf<int> pivcol, porig;
    Ufull.alloc((size_t)tot + 1); pivcol.alloc((size_t)npd + 1); porig.alloc((size_t)npd + 1);
    hipLaunchKernelGGL((k_dense_emit<DT>), dim3(C), dim3(64), 0, s, C, Dp, (i64d)ldc, pivrow_of_col, pscan.p, uoff.p, clist, row_orig,
                       Ufull.p, pivcol.p, porig.p);
    HIPCHK(hipGetLastError());
    std::vector<i64d> off((size_t)npd + 1);
    std::vector<int> pc((size_t)npd), po((size_t)npd);
    HIPCHK(hipMemcpyAsync(off.data(), uoff.p, ((size_t)npd + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(pc.data(), pivcol.p, (size_t)npd * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(po.data(), porig.p, (size_t)npd * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const i64 base = U.p.back();
    for (int k = 0; k < npd; k++) {
        U.p.push_back(base + off[(size_t)k + 1]);
        U.pivcol.push_back(pc[(size_t)k]);
        U.orig.push_back(po[(size_t)k]);
    }
    append_entries(U, Ufull.p, (i64)tot, s);
    return npd;
}

// Leftmost-pivot elimination of a dense R x C matrix D (row-major, leading dimension ldc, residues mod p) whose column c is
// column clist[c] of the matrix and whose row r comes from row row_orig[r] of the input; the pivot rows found are appended to U.
template <typename DT>
int dense_eliminate(DevBuf<DT> &D, int R, int C, i64 ldc, const int *clist, const int *row_orig, const ZpField &F, HostU &U, hipStream_t s)
{
    const double te0 = spasm_wtime();
    double te1 = te0;
    Scanner scan;
    DevBuf<int> is_piv, pivrow_of_col, prow, fcol;
    DevBuf<DenseState> st;
    is_piv.alloc((size_t)R + 1);
    pivrow_of_col.alloc((size_t)C + 1);
    prow.alloc((size_t)ldc);
    fcol.alloc((size_t)R + 1);
    st.alloc(1);
    is_piv.zero(s); st.zero(s);
    const int rc = std::max(R, C);
    UStreamer us;
    bool streamed = false;
    if constexpr (!std::is_same<DT, int>::value) {
        // narrow D: the int8 path (dense_elem_bytes has checked that it applies); pivrow_of_col is filled, D holds the echelon form.
        // The rows of U leave for the host block by block while the elimination goes on (UStreamer).
        us.init(D.p, C, ldc, pivrow_of_col.p, clist, row_orig, U);
        streamed = true;
        if (!dense_eliminate_i8(D, R, C, ldc, F, pivrow_of_col, s, &us)) throw EngineError("dense finish: shape outside the panel kernel's range");
    } else {
        if (F.p <= ((i64)1 << 24)) {
            // blocked: panels of DPB columns, f64-MFMA trailing update (exact: 64 * (p/2)^2 < 2^53)
            const i64 R64 = ((i64)R + 63) / 64 * 64, ldu = ldc;
            DevBuf<double> Lm, Upan;
            DevBuf<int> pan_row, pan_inv;
            Lm.alloc((size_t)R64 * DPB);
            Upan.alloc((size_t)DPB * (size_t)ldu);
            pan_row.alloc(DPB);
            pan_inv.alloc(DPB);
            for (int c0 = 0; c0 < C; c0 += DPB) {
                const int c1 = std::min(c0 + DPB, C);
                Lm.zero(s);
                Upan.zero(s);
                for (int c = c0; c < c1; c++) {
                    hipLaunchKernelGGL(k_panel_find, dim3(1), dim3(1024), 0, s, c, c0, c1, R, F, D.p, (i64d)ldc, is_piv.p, pivrow_of_col.p, prow.p,
                                       pan_row.p, pan_inv.p, st.p);
                    hipLaunchKernelGGL(k_panel_elim2, dim3(R), dim3(64), 0, s, c, c1, F, D.p, (i64d)ldc, is_piv.p, prow.p, fcol.p, Lm.p, st.p);
                }
                HIPCHK(hipGetLastError());
                if (c1 < C) {
                    hipLaunchKernelGGL(k_panel_trsm2, dim3(cdiv(C - c1, 64)), dim3(64), 0, s, c1, C, F, D.p, (i64d)ldc, Lm.p, pan_row.p, pan_inv.p, Upan.p,
                                       (i64d)ldu, st.p);
                    hipLaunchKernelGGL(k_dense_gemm, dim3((unsigned)(R64 / 64), cdiv(C - c1, 64)), dim3(256), 0, s, c1, R, C, F, D.p, (i64d)ldc, Lm.p,
                                       Upan.p, (i64d)ldu, is_piv.p, st.p);
                    HIPCHK(hipGetLastError());
                }
            }
            HIPCHK(hipStreamSynchronize(s)); // Lm / Upan go out of scope
        } else {
            for (int c = 0; c < C; c++) {
                hipLaunchKernelGGL(k_dense_find, dim3(1), dim3(1024), 0, s, c, R, D.p, (i64d)ldc, is_piv.p, pivrow_of_col.p, st.p);
                hipLaunchKernelGGL(k_dense_scale, dim3(cdiv(rc, 256)), dim3(256), 0, s, c, R, C, F, D.p, (i64d)ldc, is_piv.p, prow.p, fcol.p, st.p);
                hipLaunchKernelGGL(k_dense_store_prow, dim3(cdiv(C, 256)), dim3(256), 0, s, c, C, D.p, (i64d)ldc, prow.p, st.p);
                hipLaunchKernelGGL(k_dense_elim, dim3(cdiv(C - c, 256), R), dim3(256), 0, s, c, R, C, F, D.p, (i64d)ldc, prow.p, fcol.p, st.p);
                if ((c & 255) == 255) HIPCHK(hipGetLastError());
            }
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    te1 = spasm_wtime();
    int npd = 0;
    if constexpr (!std::is_same<DT, int>::value) {
        if (streamed) {
            us.template take<DT>(C);
            npd = us.found;
        }
    } else {
        npd = dense_extract_U(D.p, C, ldc, pivrow_of_col.p, clist, row_orig, U, s);
    }
    if (streamed)
        spasm_logf("[echelonize/dense] %d x %d dense tail: %d pivots [elimination with the rows of U leaving block by block %.2fs, the last block's rows %.2fs; %.2fs of copies in all]\n",
                   R, C, npd, te1 - te0, spasm_wtime() - te1, us.seconds);
    else
        spasm_logf("[echelonize/dense] %d x %d dense tail: %d pivots [elimination %.2fs, rows of U to the host %.2fs]\n", R, C, npd, te1 - te0, spasm_wtime() - te1);
    return npd;
}


#include "dense_tall.hpp"

// cells of dense matrix a finish of `rows` x `cols` keeps resident: all of it, or -- tall and skinny -- the first slab and a batch
double dense_cells_resident(const struct echelonize_opts *opts, const ZpField &F, i64 rows, i64 cols)
{
    if (rows <= INT_MAX && cols <= INT_MAX && tall_applies(opts, F, rows, cols)) {
        // (the slab, and the smallest batch of other rows: batches and the dense W adapt to what memory is left)
        const i64 slab = tall_first_slab((int)rows, (int)cols);
        return ((double)slab + (double)std::min<i64>(rows - slab, 4096)) * (double)cols;
    }
    return (double)rows * (double)cols;
}

// cells the finish of `rows` x `cols` may keep resident: the shape cap of dense_max_entries, or -- tall and skinny, where the resident
// part is the slab that carries the pivots and cannot be smaller -- what 55 % of the free memory holds (config 5 at full size:
// a slab of 365k x 324k bytes = 118 GB)
double dense_cells_allowed(const struct echelonize_opts *opts, const ZpField &F, i64 rows, i64 cols, int elem)
{
    if (rows <= INT_MAX && cols <= INT_MAX && tall_applies(opts, F, rows, cols)) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) return 0.55 * (double)fr / (double)std::max(elem, 1);
    }
    return (double)dense_max_entries(elem);
}

// the live rows of a sparse matrix, a range at a time, as dense rows (RowSource of dense_tall.hpp)
template <typename DT> struct FillRowSource {
    const DevMat &M;
    const int *rows, *cmap;
    i64 ldc;
    hipStream_t s;
    void fill(int off, int cnt, DT *Dp)
    {
        if (cnt <= 0) return;
        hipLaunchKernelGGL((k_dense_fill<DT>), dim3(cdiv((i64)cnt * 64, 256)), dim3(256), 0, s, cnt, rows + off, M.start.p, M.len.p, M.ent.p, cmap, Dp, (i64d)ldc);
        HIPCHK(hipGetLastError());
    }
    int nslabs() const { return 1; }
    void slab_range(int, i64 &s0, int &w) const { s0 = 0; w = (int)ldc; }
    void prepare_slab(int) {}
    void fill_slab(int, int off, int cnt, DT *Dp, i64) { fill(off, cnt, Dp); }
    void done() {}
};

int run_dense_tail(const DevMat &M, const ZpField &F, HostU &U, hipStream_t s, const struct echelonize_opts *opts = nullptr)
{
    const int n = M.n, m = M.m;
    Scanner scan;
    // live rows and live columns
    DevBuf<int> rflag, rscan, rows, cflag, cscan, cmap, clist;
    rflag.alloc((size_t)n + 1); rscan.alloc((size_t)n + 1); rows.alloc((size_t)n + 1);
    cflag.alloc((size_t)m + 1); cscan.alloc((size_t)m + 1); cmap.alloc((size_t)m + 1); clist.alloc((size_t)m + 1);
    hipLaunchKernelGGL(k_flag_live, dim3(cdiv((i64)n + 1, 256)), dim3(256), 0, s, n, M.len.p, rflag.p);
    HIPCHK(hipGetLastError());
    scan.exclusive(rflag.p, rscan.p, (size_t)n + 1, s);
    hipLaunchKernelGGL(k_compact, dim3(cdiv(std::max(n, 1), 256)), dim3(256), 0, s, n, rflag.p, rscan.p, rows.p);
    HIPCHK(hipGetLastError());
    cflag.zero(s);
    if (n > 0) {
        hipLaunchKernelGGL(k_flag_cols, dim3(cdiv((i64)n * 64, 256)), dim3(256), 0, s, n, M.start.p, M.len.p, M.ent.p, cflag.p);
        HIPCHK(hipGetLastError());
    }
    scan.exclusive(cflag.p, cscan.p, (size_t)m + 1, s);
    int R = 0, C = 0;
    HIPCHK(hipMemcpyAsync(&R, rscan.p + n, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&C, cscan.p + m, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (R == 0 || C == 0) return 0;
    hipLaunchKernelGGL(k_col_map, dim3(cdiv(m, 256)), dim3(256), 0, s, m, cflag.p, cscan.p, cmap.p, clist.p);
    HIPCHK(hipGetLastError());
    const i64 ldc = ((i64)C + 63) / 64 * 64;
    DevBuf<int> row_orig;
    row_orig.alloc((size_t)R + 1);
    hipLaunchKernelGGL(k_gather_int, dim3(cdiv(R, 256)), dim3(256), 0, s, R, rows.p, M.orig.p, row_orig.p);
    HIPCHK(hipGetLastError());
    auto go = [&](auto tag) {
        using DT = decltype(tag);
        if constexpr (!std::is_same<DT, int>::value) {
            if (tall_applies(opts, F, R, C)) {
                FillRowSource<DT> src{M, rows.p, cmap.p, ldc, s};
                return dense_finish_tall<DT>(src, R, C, ldc, clist.p, row_orig.p, F, U, s);
            }
        }
        DevBuf<DT> D;
        D.alloc((size_t)R * (size_t)ldc);
        D.zero(s);
        hipLaunchKernelGGL((k_dense_fill<DT>), dim3(cdiv((i64)R * 64, 256)), dim3(256), 0, s, R, rows.p, M.start.p, M.len.p, M.ent.p, cmap.p, D.p, (i64d)ldc);
        HIPCHK(hipGetLastError());
        return dense_eliminate(D, R, C, ldc, clist.p, row_orig.p, F, U, s);
    };
    switch (dense_elem_bytes(F, R)) {
    case 1: return go((signed char)0);
    case 2: return go((short)0);
    default: return go((int)0);
    }
}

// ------------------------------------------------------------------------------------------------
// The Schur complement of a round straight into a dense matrix (spasm_schur_dense, prototype reference src/SpaSM.jl:765-766),
// through a dense image of W = -(I + U_PP)^-1 U_PN (kernels.hpp, k_wd_level ..): no multiplier solve, so the cost does not
// depend on how many pivots a row reaches.
// ------------------------------------------------------------------------------------------------
struct DenseW {
    Round &R;
    const DevMat &cur;
    hipStream_t s;
    Scanner scan;
    DevBuf<int> cflag, cscan, cmap, clist, cmap_s, order;
    DevBuf<unsigned char> Wd; // the dense W of the current slab, wbytes per element
    int wbytes = 4;
    std::vector<int> lvl_off; // rows order[lvl_off[l-1] .. lvl_off[l]) make level l >= 1 (level 0 needs no combination)
    int C = 0, depth = 0;
    bool have_levels = false;

    DenseW(Round &R_, const DevMat &cur_, hipStream_t s_) : R(R_), cur(cur_), s(s_) {}

    // columns of the dense matrix: those that hold entries of the current matrix and carry no pivot of this round.  In two halves:
    // row shards OR their flags between flag_columns() and finish_columns(), so that all of them number the columns alike.
    void flag_columns()
    {
        const int m = cur.m;
        cflag.alloc((size_t)m + 1); cscan.alloc((size_t)m + 1); cmap.alloc((size_t)m + 1); clist.alloc((size_t)m + 1); cmap_s.alloc((size_t)m + 1);
        cflag.zero(s);
        if (cur.n > 0) {
            hipLaunchKernelGGL(k_flag_cols, dim3(cdiv((i64)cur.n * 64, 256)), dim3(256), 0, s, cur.n, cur.start.p, cur.len.p, cur.ent.p, cflag.p);
            HIPCHK(hipGetLastError());
        }
        hipLaunchKernelGGL(k_mask_pivot_cols, dim3(cdiv((i64)m + 1, 256)), dim3(256), 0, s, m, R.qinv_r.p, cflag.p);
        HIPCHK(hipGetLastError());
    }
    int finish_columns()
    {
        const int m = cur.m;
        scan.exclusive(cflag.p, cscan.p, (size_t)m + 1, s);
        HIPCHK(hipMemcpyAsync(&C, cscan.p + m, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (C > 0) {
            hipLaunchKernelGGL(k_col_map, dim3(cdiv(m, 256)), dim3(256), 0, s, m, cflag.p, cscan.p, cmap.p, clist.p);
            HIPCHK(hipGetLastError());
        }
        return C;
    }
    int map_columns()
    {
        flag_columns();
        return finish_columns();
    }

    // levels of the pivot graph (pivot indices are a topological order: row q of U_PP only holds indices > q)
    void levels()
    {
        if (have_levels) return;
        have_levels = true;
        const int npiv = R.npiv;
        std::vector<UHdr> hdr((size_t)std::max(npiv, 1));
        std::vector<int2> upp((size_t)std::max<i64>(R.utotal, 1));
        HIPCHK(hipMemcpyAsync(hdr.data(), R.uhdr.p, (size_t)npiv * sizeof(UHdr), hipMemcpyDeviceToHost, s));
        if (R.utotal > 0) HIPCHK(hipMemcpyAsync(upp.data(), R.UPP.p, (size_t)R.utotal * sizeof(int2), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::vector<int> lev((size_t)std::max(npiv, 1), 0);
        depth = 0;
        for (int q = npiv - 1; q >= 0; q--) {
            const UHdr &h = hdr[(size_t)q];
            int l = 0;
            for (int k = 0; k < h.npp; k++) l = std::max(l, lev[(size_t)upp[(size_t)h.off + (size_t)k].x] + 1);
            lev[(size_t)q] = l;
            depth = std::max(depth, l);
        }
        lvl_off.assign((size_t)depth + 1, 0);
        for (int q = 0; q < npiv; q++) if (lev[(size_t)q] > 0) lvl_off[(size_t)lev[(size_t)q]]++;
        // lvl_off[l] = end of level l after the prefix sum; level l starts at lvl_off[l - 1]
        for (int l = 1; l <= depth; l++) lvl_off[(size_t)l] += lvl_off[(size_t)l - 1];
        std::vector<int> h_order((size_t)std::max(lvl_off[(size_t)depth], 1)), cursor(lvl_off.begin(), lvl_off.end());
        for (int q = 0; q < npiv; q++) {
            const int l = lev[(size_t)q];
            if (l > 0) h_order[(size_t)cursor[(size_t)l - 1]++] = q;
        }
        order.alloc(h_order.size());
        HIPCHK(hipMemcpyAsync(order.p, h_order.data(), h_order.size() * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
    }

    static int threads_for(int Cs) { return Cs >= 1024 ? 256 : std::max(16, Cs / 4); }

    // W on the columns cmap_s maps to 0 .. Cs-1 (Cs a multiple of 64)
    void build_w(int Cs)
    {
        levels();
        const int npiv = R.npiv;
        const i64 ldw = Cs;
        Wd.ensure((size_t)std::max(npiv, 1) * (size_t)ldw * (size_t)wbytes);
        HIPCHK(hipMemsetAsync(Wd.p, 0, (size_t)npiv * (size_t)ldw * (size_t)wbytes, s));
        const int bt = threads_for(Cs);
        const unsigned gy = (unsigned)cdiv(Cs, 4 * bt);
        auto go = [&](auto tag) {
            using WT = decltype(tag);
            WT *wd = (WT *)Wd.p;
            constexpr int TEAM = 16;
            hipLaunchKernelGGL((k_wd_seed<TEAM, WT>), dim3(cdiv((i64)npiv * TEAM, 256)), dim3(256), 0, s, npiv, R.F, R.uhdr.p, R.UPN.p, cmap_s.p, wd, (i64d)ldw);
            HIPCHK(hipGetLastError());
            // (narrow residues: 16 bytes per lane and term -- kernels.hpp, WideRow)
            constexpr bool wide = sizeof(WT) < 4;
            const int ept = 16 / (int)sizeof(WT);
            const int btw = Cs / ept >= 256 ? 256 : std::max(16, Cs / ept);
            const unsigned gyw = (unsigned)cdiv(Cs, ept * btw);
            for (int l = 1; l <= depth; l++) {
                const int lo = lvl_off[(size_t)l - 1], cnt = lvl_off[(size_t)l] - lo;
                if (cnt == 0) continue;
                if constexpr (wide) {
                    if (R.F.small) hipLaunchKernelGGL((k_wd_level_wide<WT>), dim3((unsigned)cnt, gyw), dim3(btw), 0, s, cnt, order.p + lo, R.F, R.uhdr.p, R.UPP.p, wd, (i64d)ldw, Cs);
                    else hipLaunchKernelGGL((k_wd_level<false, WT>), dim3((unsigned)cnt, gy), dim3(bt), 0, s, cnt, order.p + lo, R.F, R.uhdr.p, R.UPP.p, wd, (i64d)ldw, Cs);
                } else {
                    if (R.F.small) hipLaunchKernelGGL((k_wd_level<true, WT>), dim3((unsigned)cnt, gy), dim3(bt), 0, s, cnt, order.p + lo, R.F, R.uhdr.p, R.UPP.p, wd, (i64d)ldw, Cs);
                    else hipLaunchKernelGGL((k_wd_level<false, WT>), dim3((unsigned)cnt, gy), dim3(bt), 0, s, cnt, order.p + lo, R.F, R.uhdr.p, R.UPP.p, wd, (i64d)ldw, Cs);
                }
                if ((l & 1023) == 0) HIPCHK(hipGetLastError());
            }
        };
        switch (wbytes) {
        case 1: go((signed char)0); break;
        case 2: go((short)0); break;
        default: go((int)0); break;
        }
        HIPCHK(hipGetLastError());
    }

    // the entries of the rows on pivot columns as (pivot index, value) lists
    DevBuf<i64d> pcnt, poff;
    DevBuf<int2> plist;
    const int *prows = nullptr;
    int pn = -1;
    void row_lists(const int *rows, int nrows)
    {
        if (prows == rows && pn == nrows) return;
        prows = rows; pn = nrows;
        constexpr int TEAM = 16;
        pcnt.alloc((size_t)nrows + 1); poff.alloc((size_t)nrows + 1);
        hipLaunchKernelGGL((k_wd_pcount<TEAM>), dim3(cdiv(((i64)nrows + 1) * TEAM, 256)), dim3(256), 0, s, nrows, rows, cur.start.p, cur.len.p, cur.ent.p,
                           R.qinv_r.p, pcnt.p);
        HIPCHK(hipGetLastError());
        scan.exclusive(pcnt.p, poff.p, (size_t)nrows + 1, s);
        i64d tot = 0;
        HIPCHK(hipMemcpyAsync(&tot, poff.p + nrows, sizeof tot, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        plist.alloc((size_t)tot + 1);
        hipLaunchKernelGGL((k_wd_pfill<TEAM>), dim3(cdiv((i64)nrows * TEAM, 256)), dim3(256), 0, s, nrows, rows, cur.start.p, cur.len.p, cur.ent.p,
                           R.qinv_r.p, poff.p, plist.p);
        HIPCHK(hipGetLastError());
    }

    // the Schur rows of `rows` (local rows, device list) on the slab's columns: Dp[t][dcol0 + j]; Dp is zero there on entry
    template <typename DT> void rows_into(const int *rows, int nrows, int Cs, DT *Dp, i64 ldc, int dcol0)
    {
        if (nrows == 0) return;
        row_lists(rows, nrows);
        constexpr int TEAM = 16;
        hipLaunchKernelGGL((k_wd_own<TEAM, DT>), dim3(cdiv((i64)nrows * TEAM, 256)), dim3(256), 0, s, nrows, rows, cur.start.p, cur.len.p, cur.ent.p, R.qinv_r.p,
                           cmap_s.p, Dp, (i64d)ldc, dcol0);
        const int bt = threads_for(Cs);
        const unsigned gy = (unsigned)cdiv(Cs, 4 * bt);
        auto go = [&](auto tag) {
            using WT = decltype(tag);
            const WT *wd = (const WT *)Wd.p;
            if constexpr (sizeof(WT) < 4 && std::is_same<WT, DT>::value) {
                if (R.F.small) { // narrow residues: 16 bytes per lane and term
                    const int ept = 16 / (int)sizeof(WT);
                    const int btw = Cs / ept >= 256 ? 256 : std::max(16, Cs / ept);
                    hipLaunchKernelGGL((k_wd_rows_wide<WT>), dim3((unsigned)nrows, (unsigned)cdiv(Cs, ept * btw)), dim3(btw), 0, s, nrows, R.F, poff.p, plist.p, wd, (i64d)Cs, Cs, Dp,
                                       (i64d)ldc, dcol0);
                    return;
                }
            }
            if (R.F.small) hipLaunchKernelGGL((k_wd_rows<true, DT, WT>), dim3((unsigned)nrows, gy), dim3(bt), 0, s, nrows, R.F, poff.p, plist.p, wd, (i64d)Cs, Cs, Dp,
                                              (i64d)ldc, dcol0);
            else hipLaunchKernelGGL((k_wd_rows<false, DT, WT>), dim3((unsigned)nrows, gy), dim3(bt), 0, s, nrows, R.F, poff.p, plist.p, wd, (i64d)Cs, Cs, Dp, (i64d)ldc,
                                    dcol0);
        };
        switch (wbytes) {
        case 1: go((signed char)0); break;
        case 2: go((short)0); break;
        default: go((int)0); break;
        }
        HIPCHK(hipGetLastError());
    }

    void slab(int s0, int Cs, int stride)
    {
        hipLaunchKernelGGL(k_slab_cmap, dim3(cdiv(cur.m, 256)), dim3(256), 0, s, cur.m, cmap.p, s0, Cs, stride, cmap_s.p);
        HIPCHK(hipGetLastError());
    }

    // spasm_schur_estimate_density (prototype reference src/SpaSM.jl:763-764) without solving a single row: the Schur complement
    // restricted to 64 of its columns, evenly spread, for ALL the non-pivot rows; density relative to `free_cols` columns
    double estimate_density(const int *rows, int nrows, int free_cols)
    {
        if (C == 0 || nrows == 0 || free_cols <= 0) return 0.0;
        const int Cs = 64, stride = std::max(1, C / Cs);
        const int sampled = std::min(Cs, (C + stride - 1) / stride);
        slab(0, Cs, stride > 1 ? stride : 0); // (stride 1: the first 64 columns, as a slab)
        build_w(Cs);
        DevBuf<int> Dsm;
        DevBuf<u64d> cnt;
        Dsm.alloc((size_t)nrows * Cs);
        Dsm.zero(s);
        cnt.alloc(1);
        cnt.zero(s);
        rows_into(rows, nrows, Cs, Dsm.p, Cs, 0);
        hipLaunchKernelGGL(k_count_nonzero, dim3(std::min(cdiv((i64)nrows * Cs, 256), R.num_cu * 16)), dim3(256), 0, s, (i64d)nrows * Cs, Dsm.p, cnt.p);
        HIPCHK(hipGetLastError());
        u64d nz = 0;
        HIPCHK(hipMemcpyAsync(&nz, cnt.p, sizeof nz, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        // nz / (rows * sampled) is the density over the C live columns; the other free columns are empty
        return (double)nz / ((double)nrows * (double)sampled) * ((double)C / (double)free_cols);
    }
};

// The Schur complement of the round R has prepared (pivots elected, U built), straight into a dense matrix over the columns W has
// mapped: D gets nnp + extra_rows rows (the extra ones zero: the guest rows of the distributed finish), row_orig the input rows.
template <typename DT>
void schur_dense_build(Round &R, const DevMat &cur, int nnp, DenseW &W, int extra_rows, DevBuf<DT> &D, DevBuf<int> &row_orig, hipStream_t s)
{
    const int C = W.C;
    const double tw0 = spasm_wtime();
    const i64 ldc = ((i64)C + 63) / 64 * 64;
    D.alloc(((size_t)nnp + (size_t)extra_rows) * (size_t)ldc);
    D.zero(s);
    row_orig.alloc((size_t)nnp + 1);
    if (nnp > 0) {
        hipLaunchKernelGGL(k_gather_int, dim3(cdiv(nnp, 256)), dim3(256), 0, s, nnp, R.np_rows.p, cur.orig.p, row_orig.p);
        HIPCHK(hipGetLastError());
    }
    // W for as many columns at a time as a third of the free memory holds
    size_t fr = 0, tot = 0;
    HIPCHK(hipMemGetInfo(&fr, &tot));
    W.wbytes = (int)sizeof(DT); // (W holds residues like D does)
    i64 budget = (i64)(fr / 3) + (i64)W.Wd.n;
    if (const char *mb = getenv("SPASM_AMD_MEM_BUDGET_MB")) budget = std::max<i64>(atoll(mb), 1) << 18; // tests: several slabs
    i64 Cs = std::min<i64>(ldc, std::max<i64>(64, budget / ((i64)std::max(R.npiv, 1) * W.wbytes) / 64 * 64));
    int nslab = 0;
    for (i64 s0 = 0; s0 < ldc && nnp > 0; s0 += Cs, nslab++) {
        const int w = (int)std::min<i64>(Cs, ldc - s0);
        W.slab((int)s0, w, 0);
        W.build_w(w);
        W.rows_into(R.np_rows.p, nnp, w, D.p, ldc, (int)s0);
    }
    HIPCHK(hipStreamSynchronize(s));
    spasm_logf("[echelonize/dense] Schur complement %d x %d through a dense W (%d pivots, %d levels, %d slab%s of columns) [%.2fs]\n", nnp, C, R.npiv,
               W.depth, nslab, nslab == 1 ? "" : "s", spasm_wtime() - tw0);
    W.Wd.release();
}

// the Schur rows of a round, a range of the non-pivot rows at a time, through the dense W (RowSource of dense_tall.hpp)
template <typename DT> struct SchurRowSource {
    Round &R;
    const DevMat &cur;
    DenseW &W;
    i64 ldc;
    hipStream_t s;
    i64 Cs = 0;
    bool built = false;
    SchurRowSource(Round &R_, const DevMat &cur_, DenseW &W_, i64 ldc_, hipStream_t s_) : R(R_), cur(cur_), W(W_), ldc(ldc_), s(s_)
    {
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        W.wbytes = (int)sizeof(DT);
        i64 budget = (i64)(fr / 3) + (i64)W.Wd.n;
        if (const char *mb = getenv("SPASM_AMD_MEM_BUDGET_MB")) budget = std::max<i64>(atoll(mb), 1) << 18;
        Cs = std::min<i64>(ldc, std::max<i64>(64, budget / ((i64)std::max(R.npiv, 1) * W.wbytes) / 64 * 64));
    }
    void fill(int off, int cnt, DT *Dp)
    {
        const bool one = Cs >= ldc; // W for all columns at once: built once, kept until done()
        for (i64 s0 = 0; s0 < ldc && cnt > 0; s0 += Cs) {
            const int w = (int)std::min<i64>(Cs, ldc - s0);
            if (!(one && built)) {
                W.slab((int)s0, w, 0);
                W.build_w(w);
                built = true;
            }
            W.rows_into(R.np_rows.p + off, cnt, w, Dp, ldc, (int)s0);
        }
    }
    int nslabs() const { return (int)((ldc + Cs - 1) / Cs); }
    void slab_range(int k, i64 &s0, int &w) const { s0 = (i64)k * Cs; w = (int)std::min<i64>(Cs, ldc - s0); }
    void prepare_slab(int k)
    {
        if (Cs >= ldc && built) return;
        i64 s0;
        int w;
        slab_range(k, s0, w);
        W.slab((int)s0, w, 0);
        W.build_w(w);
        built = true;
    }
    // rows off .. off + cnt on the columns of the slab prepare_slab(k) has built: Dp[i][j], j < w, leading dimension ldw
    void fill_slab(int k, int off, int cnt, DT *Dp, i64 ldw)
    {
        i64 s0;
        int w;
        slab_range(k, s0, w);
        if (cnt > 0) W.rows_into(R.np_rows.p + off, cnt, w, Dp, ldw, 0);
    }
    void done() { W.Wd.release(); built = false; }
};

// The finish of an echelonization whose remainder is dense: the Schur complement of the round R has prepared (pivots elected, U
// built) goes straight into a dense matrix over the columns that are left, and is eliminated there.
void schur_dense_finish(Round &R, const DevMat &cur, int nnp, HostU &U, hipStream_t s, DenseW *prepared = nullptr, const struct echelonize_opts *opts = nullptr)
{
    std::unique_ptr<DenseW> own;
    if (!prepared) {
        own.reset(new DenseW(R, cur, s));
        own->map_columns();
        prepared = own.get();
    }
    DenseW &W = *prepared;
    const int C = W.C;
    if (C == 0 || nnp == 0) return;
    const i64 ldc = ((i64)C + 63) / 64 * 64;
    DevBuf<int> row_orig;
    // (element type of D: dense_elem_bytes)
    auto go = [&](auto tag) {
        using DT = decltype(tag);
        if constexpr (!std::is_same<DT, int>::value) {
            if (tall_applies(opts, R.F, nnp, C)) {
                // many more rows than columns: one slab carries the pivots, the other rows are reduced in one step (dense_tall.hpp)
                row_orig.alloc((size_t)nnp + 1);
                hipLaunchKernelGGL(k_gather_int, dim3(cdiv(nnp, 256)), dim3(256), 0, s, nnp, R.np_rows.p, cur.orig.p, row_orig.p);
                HIPCHK(hipGetLastError());
                SchurRowSource<DT> src(R, cur, W, ldc, s);
                dense_finish_tall<DT>(src, nnp, C, ldc, W.clist.p, row_orig.p, R.F, U, s);
                return;
            }
        }
        DevBuf<DT> D;
        schur_dense_build(R, cur, nnp, W, 0, D, row_orig, s);
        dense_eliminate(D, nnp, C, ldc, W.clist.p, row_orig.p, R.F, U, s);
    };
    switch (dense_elem_bytes(R.F, nnp)) {
    case 1: go((signed char)0); break;
    case 2: go((short)0); break;
    default: go((int)0); break;
    }
}

#include "dense_multi.hpp"

// the L factor on the host, as it is collected: per chunk the rows it belongs to (original row of A per slot, entries per slot),
// then the (row of U, value) pairs
struct HostL {
    std::vector<int> row;     // original row of every entry
    std::vector<int> j, x;    // row of U, value
};

// the multiplier lists of the `cnt` row slots the last solve_phase of R filled (rows np_rows[off ..]) -> HostL; ubase = rows of U
// that existed before this round
void collect_L_lists(HostL &L, Round &R, const DevMat &cur, int off, int cnt, int ubase, hipStream_t s)
{
    if (cnt == 0) return;
    DevBuf<i64d> lcnt, loff;
    lcnt.alloc((size_t)cnt + 1); loff.alloc((size_t)cnt + 1);
    hipLaunchKernelGGL(k_l_count, dim3(cdiv(((i64)cnt + 1) * 64, 256)), dim3(256), 0, s, cnt, R.Lstart.p, R.Llen.p, R.Lpool.p, lcnt.p);
    HIPCHK(hipGetLastError());
    R.scan.exclusive(lcnt.p, loff.p, (size_t)cnt + 1, s);
    std::vector<i64d> h_off((size_t)cnt + 1);
    std::vector<int> h_rows((size_t)cnt), h_orig((size_t)cur.n);
    HIPCHK(hipMemcpyAsync(h_off.data(), loff.p, ((size_t)cnt + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_rows.data(), R.np_rows.p + off, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_orig.data(), cur.orig.p, (size_t)cur.n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const i64 tot = h_off[(size_t)cnt];
    if (tot == 0) return;
    DevBuf<int> oj, ox;
    oj.alloc((size_t)tot); ox.alloc((size_t)tot);
    hipLaunchKernelGGL(k_l_fill, dim3(cdiv((i64)cnt * 64, 256)), dim3(256), 0, s, cnt, ubase, R.Lstart.p, R.Llen.p, R.Lpool.p, R.Lidx.p, loff.p, oj.p, ox.p);
    HIPCHK(hipGetLastError());
    const size_t old = L.j.size();
    L.j.resize(old + (size_t)tot); L.x.resize(old + (size_t)tot); L.row.resize(old + (size_t)tot);
    HIPCHK(hipMemcpyAsync(L.j.data() + old, oj.p, (size_t)tot * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(L.x.data() + old, ox.p, (size_t)tot * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int t = 0; t < cnt; t++) {
        const int o = h_orig[(size_t)h_rows[(size_t)t]];
        for (i64 k = h_off[(size_t)t]; k < h_off[(size_t)t + 1]; k++) L.row[old + (size_t)k] = o;
    }
}

// the diagonal: row U.orig[ubase + k] of A is pivval[k] times row ubase + k of U (plus what its own list said in earlier rounds)
void collect_L_pivots(HostL &L, Round &R, const HostU &U, int ubase, hipStream_t s)
{
    const int np = R.npiv;
    if (np == 0) return;
    std::vector<int> pv((size_t)np);
    HIPCHK(hipMemcpyAsync(pv.data(), R.pivval.p, (size_t)np * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int k = 0; k < np; k++) {
        L.row.push_back(U.orig[(size_t)ubase + (size_t)k]);
        L.j.push_back(ubase + k);
        L.x.push_back(pv[(size_t)k]);
    }
}

struct HostL;
struct spasm_lu *assemble_lu(HostU &U, int n, int m, i64 prime, HostL *HLp);

struct spasm_lu *do_echelonize(const struct spasm_csr *A, struct echelonize_opts *opts, i64 *rank_only = nullptr)
{
    // max_round, min_pivot_proportion, enable_dense and sparsity_threshold (reference src/SpaSM.jl:329-337) decide how far the
    // sparse rounds go and what finishes.  With enable_greedy_pivot_search off every pivot is a leftmost entry, and rank, pivot
    // columns and kernel do not depend on those tunables; with it on (the reference's default) the pivot columns depend on
    // where the sparse rounds stop, the rank of course does not.
    struct echelonize_opts dflt;
    if (!opts) { spasm_echelonize_init_opts(&dflt); opts = &dflt; }
    require_device();
    check_input(A, "spasm_echelonize");
    const int n = A->n, m = A->m;
    const i64 prime = A->field->p;
    const double t0 = spasm_wtime();
    spasm_logf("[echelonize] Start on %d x %d matrix with %lld nnz\n", n, m, (long long)spasm_nnz(A));
    g_last_rounds.clear();
    // Fields of echelonize_opts (reference src/SpaSM.jl:332, :339, :340, :342) this engine has no use for say so when they are set
    // away from spasm_echelonize_init_opts' values: they tune libspasm's dense strategies (rows per dense block of its FFPACK
    // calls; when its randomized low-rank mode starts and with which row weights) and the completion of L -- the dense finish
    // here is one exact blocked elimination (panels of 64 columns, blocks of 1024), which has no such knobs, and L holds the
    // multipliers of the pivotal and the eliminated rows as the rounds produced them.  The result does not depend on them.
    {
        struct echelonize_opts d0;
        spasm_echelonize_init_opts(&d0);
        if (opts->dense_block_size != d0.dense_block_size)
            spasm_logf("[echelonize] option dense_block_size = %d ignored (the dense finish works in panels of 64 columns and blocks of 1024)\n", opts->dense_block_size);
        if (opts->low_rank_ratio != d0.low_rank_ratio || opts->low_rank_start_weight != d0.low_rank_start_weight)
            spasm_logf("[echelonize] options low_rank_ratio = %g / low_rank_start_weight = %g ignored (no randomized low-rank mode: the "
                       "tall-and-skinny finish reduces every row exactly)\n", opts->low_rank_ratio, opts->low_rank_start_weight);
        if (opts->complete != d0.complete) spasm_logf("[echelonize] option complete = %d ignored (L is returned as the rounds produced it; LU.complete stays false)\n", (int)opts->complete);
    }

    hipStream_t stream = nullptr;
    HostU U;
    U.p.push_back(0);
    U.rank_only = rank_only != nullptr;
    std::unique_ptr<DevMat> cur(new DevMat());
    upload_csr(A, 0, n, *cur, stream);
    i64 cur_nnz = spasm_nnz(A);

    std::unique_ptr<Round> R(new Round());
    R->F = zp_field_make(prime);
    R->stream = stream;
    // the trip counters of the round statistics (applications, nnz_reduced, read_bytes) count multiplier-list entries: they are
    // exact only when the rounds keep to the lists, which costs about a third more time per round
    if (const char *e = getenv("SPASM_AMD_ROUND_STATS")) R->force_lists = atoi(e) != 0;
    // (no event pair around every scatter class: with them the classes run one after the other on one stream -- the two lanes and
    // the twins' side streams are what the benchmark's step has, and a round is to cost what that step costs.  The per-class times
    // of the round statistics are then zero; SPASM_AMD_CLASS_TIMING=1 brings them back.)
    R->class_timing = false;
    if (const char *e = getenv("SPASM_AMD_CLASS_TIMING")) R->class_timing = atoi(e) != 0;
    // echelonize_opts.L (reference src/SpaSM.jl:331): keep the multipliers.  They only exist as lists on the sparse path, so the
    // rounds keep to the lists (with the pivot index of every record) and the dense finish is not used.
    const bool want_L = opts->L;
    const bool use_dense = opts->enable_dense && !want_L;
    HostL HL;
    if (want_L) { R->force_lists = true; R->want_idx = true; R->want_pivval = true; }
    int round = 0;
    i64 cur_live = n;
    bool gplu_finish = false; // the sparse rounds are over (max_round / min_pivot_proportion) and the dense finish is not an option
    bool partial = false;     // enable_GPLU = 0 and no dense finish: the factorization stops short (the reference's behaviour)
    while (cur->n > 0 && m > 0) {
        {
            const i64 cfree = (i64)m - (i64)U.pivcol.size();
            const double cells = (double)cur_live * (double)cfree;
            if (use_dense && cur_nnz > 0 && cells > 0 && dense_cells_resident(opts, R->F, cur_live, cfree) <= dense_cells_allowed(opts, R->F, cur_live, cfree, dense_elem_bytes(R->F, cur_live)) &&
                (double)cur_nnz > opts->sparsity_threshold * cells) {
                spasm_logf("[echelonize] finishing; density = %.3f; aspect ratio = %.1f\n", (double)cur_nnz / cells,
                           cfree > 0 ? (double)cur_live / (double)cfree : 0.0);
                run_dense_tail(*cur, R->F, U, stream, opts);
                break;
            }
        }
        HIPCHK(hipEventRecord(R->ev[0], stream));
        R->elect_local(*cur, 0);
        R->assign_pivots();
        if (R->npiv == 0) break; // no non-empty row left
        // enable_greedy_pivot_search (reference src/SpaSM.jl:326): off = leftmost-entry pivots only; on = also "FL on columns"
        // in the max_round sparse rounds (libspasm's third search, the cycle-free greedy one, is not reproduced).  The rounds of the
        // finish keep to leftmost entries, like the reference's GPLU and dense finishes.
        R->n_leftmost = R->npiv;
        R->n_open = 0;
        R->n_greedy = 0;
        if (opts->enable_greedy_pivot_search && !gplu_finish && round < opts->max_round) {
            R->extend_pivots_on_open_columns(*cur);
            const char *no_third = getenv("SPASM_AMD_NO_CYCLE_FREE_SEARCH"); // A/B: the round without the third search
            if (!(no_third && atoi(no_third))) R->extend_pivots_cycle_free(*cur);
        }
        if (!gplu_finish) {
            // The reference's round loop (tunables src/SpaSM.jl:333-337): at most max_round sparse rounds, and none that finds
            // fewer than min_pivot_proportion * min(rows, free columns) pivots ("not enough pivots found; stopping", README.md:32).
            // What is left then goes to the dense finish when it fits the device and may be used, else to the GPLU-style
            // finish: more leftmost-pivot rounds until nothing is left.
            const i64 cfree = (i64)m - (i64)U.pivcol.size();
            const double need = opts->min_pivot_proportion * (double)std::min<i64>(cur_live, cfree);
            const bool few = (double)R->npiv < need;
            const bool spent = round >= opts->max_round;
            if (few || spent) {
                if (few) spasm_logf("[echelonize] not enough pivots found; stopping (%d < %.0f)\n", R->npiv, need);
                const double cells = (double)cur_live * (double)cfree;
                // dense when the remainder is dense enough (the reference's rule) or small enough for the cubic work not to matter
                // (2^28 cells: a 16384 x 16384 remainder); a large sparse remainder is better served by more sparse rounds
                const bool worth = (double)cur_nnz > opts->sparsity_threshold * cells || cells <= (double)((i64)1 << 28);
                if (use_dense && cells > 0 && dense_cells_resident(opts, R->F, cur_live, cfree) <= dense_cells_allowed(opts, R->F, cur_live, cfree, dense_elem_bytes(R->F, cur_live)) && worth) {
                    spasm_logf("[echelonize] finishing; density = %.3f; aspect ratio = %.1f\n", (double)cur_nnz / cells,
                               cfree > 0 ? (double)cur_live / (double)cfree : 0.0);
                    run_dense_tail(*cur, R->F, U, stream, opts);
                    break;
                }
                if (!opts->enable_GPLU) {
                    // enable_GPLU = 0 (reference src/SpaSM.jl:330) with the dense finish off, not applicable or vetoed: no method is left
                    // to finish with.  libspasm says so and returns what the sparse rounds found ([UPSTREAM-RECALL] "Cannot finish
                    // (no valid method; enable either GPLU or dense)"): U is then a partial echelon form and r a LOWER bound of the rank.
                    spasm_logf("[echelonize] Cannot finish (no valid method; enable either GPLU or dense): stopping with %d pivots, a lower bound of the rank\n",
                               (int)U.pivcol.size());
                    partial = true;
                    break;
                }
                spasm_logf("[echelonize] finishing with GPLU-style rounds\n");
                gplu_finish = true;
            }
        }
        R->mark_local(*cur, 0);
        R->build_U(*cur, R->pivrow.p);
        // W (level by level, no Uinv) when many more rows than pivots are reduced and the pivot graph is shallow; otherwise Uinv
        // for the multiplier lists, or elimination chains when that is too dense as well
        R->prepare_w(R->nnp, cur_nnz);
        if (R->use_w) { R->use_uinv = false; R->ms_uinv = 0; }
        else R->prepare_uinv(R->nnp);
        const int nnp = R->nnp;
        // The multiplier records and the slots of the Schur rows of ALL non-pivot rows normally fit (config 3: 0.5 + 4.6 GB).
        // Rounds whose rows reach tens of thousands of pivots (Macaulay-like) can need more than the device has: the rows
        // are then reduced in batches, each appended compactly to the matrix of the next round.
        size_t mem_free = 0, mem_total = 0;
        HIPCHK(hipMemGetInfo(&mem_free, &mem_total));
        i64 avail = (i64)mem_free + (i64)(R->Lpool.n * sizeof(int4)) + (i64)(R->S.ent.n * sizeof(int2));
        i64 floor_entries = 1 << 20;
        if (const char *mb = getenv("SPASM_AMD_MEM_BUDGET_MB")) { // tests: pretend the device is this small
            avail = std::max<i64>(atoll(mb), 1) << 20;
            floor_entries = 1 << 10;
        }
        const i64 max_pool = std::max<i64>((i64)(0.40 * (double)avail) / (i64)sizeof(int4), floor_entries);
        const i64 max_slots = std::max<i64>((i64)(0.30 * (double)avail) / (i64)sizeof(int2), floor_entries);
        std::unique_ptr<DevMat> next(new DevMat());
        next->n = nnp;
        next->m = m;
        RoundCounters acc;
        memset(&acc, 0, sizeof acc);
        int acc_class[NCLASS] = {0};
        float ms_solve = 0, ms_scatter = 0;
        i64 appended = 0; // entries of next->ent in use (batched rounds)
        int off = 0, chunk = nnp, nbatch = 0;
        float ms_pivots = 0;
        DevBuf<i64d> app_scan;
        // ---- the density of the Schur complement, estimated before it is built (reference spasm_schur_estimate_density, prototype
        // src/SpaSM.jl:763-764, log line README.md:25): the Schur rows of a sample of the non-pivot rows (every step-th, at most
        // 2048; libspasm takes 100 at random) are computed for real.  The same sample sizes the record pool and the slots.
        double rec_per_row = 4.0 * (double)cur_nnz / (double)std::max(nnp, 1), slots_per_row = 0;
        double est_density = -1;
        const int free_now = m - (int)U.pivcol.size() - R->npiv;
        const bool dense_possible = use_dense && nnp > 64 && dense_cells_resident(opts, R->F, nnp, free_now) <= dense_cells_allowed(opts, R->F, nnp, free_now, dense_elem_bytes(R->F, nnp));
        std::unique_ptr<DenseW> dw;
        if (nnp > 0 && dense_possible && !R->use_uinv && !R->use_w) {
            // No Uinv: the rows of this round reach many pivots (or there are few rows), and the row sample below would walk those
            // reaches one dependent step at a time (1.7 s for 2048 rows of the 200k x 80k Macaulay-like case).  Sample COLUMNS
            // instead: the Schur complement on 64 of its columns, all rows, through a dense W.
            dw.reset(new DenseW(*R, *cur, stream));
            dw->map_columns();
            est_density = dw->estimate_density(R->np_rows.p, nnp, free_now);
            spasm_logf("Schur complement is %d x %d, estimated density : %.2f (64 columns sampled, %d levels)\n", nnp, free_now, est_density, dw->depth);
        }
        const bool go_dense = dense_possible && est_density > opts->sparsity_threshold;
        if (nnp > 0 && !go_dense) {
            const int probe = std::min(nnp, 2048);
            const int step = nnp / probe;
            DevBuf<int> probe_rows;
            probe_rows.alloc((size_t)probe + 1);
            hipLaunchKernelGGL(k_pick_stride, dim3(cdiv(probe, 256)), dim3(256), 0, stream, probe, step, R->np_rows.p, probe_rows.p);
            HIPCHK(hipGetLastError());
            const i64 tp = R->solve_phase(*cur, probe_rows.p, nullptr, probe, 1 << 20, max_pool);
            if (tp >= 0) {
                rec_per_row = std::max(rec_per_row, 1.25 * (double)R->pool_used() / probe);
                slots_per_row = 1.25 * (double)tp / probe;
            }
            // (the sample's Schur rows are only computed where the answer can matter: the dense finish allowed and within reach)
            if (tp >= 0 && tp <= max_slots && dense_possible && est_density < 0) {
                R->S.ent.ensure((size_t)tp + 1);
                R->run_scatter(*cur, probe_rows.p, probe);
                R->fetch_counters();
                if (free_now > 0) est_density = (double)R->hctr.nnz_out / ((double)probe * (double)free_now);
                spasm_logf("Schur complement is %d x %d, estimated density : %.2f (%d rows sampled)\n", nnp, free_now, est_density, probe);
            } else if (tp < 0) {
                rec_per_row = (double)max_pool / probe;
            }
            if (nnp > 16384) {
                const double fit = std::min((double)max_pool / std::max(rec_per_row, 1.0), slots_per_row > 0 ? (double)max_slots / slots_per_row : 1e18);
                if (fit < (double)nnp) chunk = std::max(1, (int)(0.8 * fit));
            }
        }
        // ---- dense already?  Then the Schur complement goes straight into the dense matrix of the finish (spasm_schur_dense,
        // prototype src/SpaSM.jl:765-766) and is never materialised sparse.
        if (dense_possible && est_density > opts->sparsity_threshold) {
            spasm_logf("[echelonize] round %d\n[pivots] Faugère-Lachartre: %d pivots found\n", round, R->n_leftmost);
            if (R->n_open) spasm_logf("[pivots] ``Faugère-Lachartre on columns'': %d pivots found\n", R->n_open);
            if (R->n_greedy) spasm_logf("[pivots] greedy alternating cycle-free search: %d pivots found\n", R->n_greedy);
            spasm_logf("[echelonize] finishing; density = %.3f (estimated); aspect ratio = %.1f; Schur complement straight to dense\n", est_density,
                       free_now > 0 ? (double)nnp / (double)free_now : 0.0);
            append_round_U(U, *R, *cur, stream);
            schur_dense_finish(*R, *cur, nnp, U, stream, dw.get(), opts);
            spasm_amd_round_stats st;
            fill_stats(st, *R, round, cur->n, cur_nnz);
            st.nnz_out = -1; // (never counted: the rows went dense)
            g_last_rounds.push_back(st);
            break;
        }
        dw.reset();
        while (off < nnp || nnp == 0) {
            const int cnt = std::min(chunk, nnp - off);
            // (what the step allocates is allocated here, in front of the timed step: the record pool, the per-row arrays, and S as the
            // sample sized it -- a 4 GB hipMalloc between two kernels of the step is milliseconds of an idle device)
            R->alloc_solve(cnt, std::min<i64>(std::max<i64>((i64)(rec_per_row * (double)cnt), 1 << 16), max_pool));
            R->warm_scatter();
            if (slots_per_row > 0) R->S.ent.ensure((size_t)std::min<double>((double)max_slots, 1.05 * slots_per_row * (double)cnt) + 1024);
            HIPCHK(hipEventRecord(R->ev[1], stream));
            // The fused step (fused.hpp: plan + stream of a row in one kernel, S written compactly) when the round goes along W and the
            // sample told how long the Schur rows are; the rows it leaves, and every round without W, take the general path: the
            // solve (plan or multiplier lists, bounds), then the scatter classes.
            const i64 fused_est = (R->fused_ok() && slots_per_row > 0) ? (i64)(slots_per_row * (double)cnt) + 16 * (i64)cnt : 0;
            if (fused_est > 0 && fused_est <= max_slots) {
                const i64 scap = R->fused_capacity(fused_est);
                R->S.ent.ensure((size_t)scap + 1);
                HIPCHK(hipEventRecord(R->ev[1], stream));
                HIPCHK(hipEventRecord(R->ev[2], stream));
                if (R->run_fused(*cur, R->np_rows.p + off, cnt, scap, false)) {
                    HIPCHK(hipEventRecord(R->ev[3], stream));
                    R->fetch_step();
                    R->fused_leftovers(*cur, R->np_rows.p + off);
                    goto schur_done;
                }
            }
            {
            const i64 tot = R->solve_phase(*cur, R->np_rows.p + off, nullptr, cnt, std::max<i64>((i64)(rec_per_row * (double)cnt), 1 << 16), max_pool);
            if (tot < 0 || tot > max_slots) {
                if (cnt <= 1) throw EngineError("one row of the Schur complement does not fit the device memory");
                chunk = std::max(1, cnt / 2);
                spasm_logf("[echelonize] round %d: %d rows at a time do not fit the device memory, trying %d\n", round, cnt, chunk);
                continue;
            }
            HIPCHK(hipEventRecord(R->ev[2], stream));
            R->S.ent.ensure((size_t)tot + 1);
            R->run_scatter(*cur, R->np_rows.p + off, cnt);
            HIPCHK(hipEventRecord(R->ev[3], stream));
            R->fetch_counters();
            }
        schur_done:
            if (want_L) collect_L_lists(HL, *R, *cur, off, cnt, (int)U.pivcol.size(), stream);
            {
                float ms = 0;
                if (nbatch == 0 && hipEventElapsedTime(&ms, R->ev[0], R->ev[1]) == hipSuccess) ms_pivots = ms; // incl. the probe
                if (hipEventElapsedTime(&ms, R->ev[1], R->ev[2]) == hipSuccess) ms_solve += ms;
                if (hipEventElapsedTime(&ms, R->ev[2], R->ev[3]) == hipSuccess) ms_scatter += ms;
            }
            acc.applications += R->hctr.applications;
            acc.nnz_reduced += R->hctr.nnz_reduced;
            acc.segments += R->hctr.segments;
            acc.nonempty_out += R->hctr.nonempty_out;
            acc.nnz_out += R->hctr.nnz_out;
            acc.stream_fix += R->hctr.stream_fix;
            acc.stream_redo += R->hctr.stream_redo;
            for (int c = 0; c < 16; c++) { acc.class_ent[c] += R->hctr.class_ent[c]; acc.class_seg[c] += R->hctr.class_seg[c]; }
            for (int c = 0; c < NCLASS; c++) acc_class[c] += R->hclass_count[c];
            nbatch++;
            if (off == 0 && cnt == nnp) {
                // the whole round at once: its slots become the next matrix as they are
                next->start = std::move(R->S.start);
                next->len = std::move(R->S.len);
                next->lead = std::move(R->S.lead);
                next->orig = std::move(R->S.orig);
                next->ent = std::move(R->S.ent);
            } else {
                if (off == 0) {
                    next->start.ensure((size_t)nnp + 1);
                    next->len.ensure((size_t)nnp + 1);
                    next->lead.ensure((size_t)nnp + 1);
                    next->orig.ensure((size_t)nnp + 1);
                }
                const i64 add = (i64)R->hctr.nnz_out;
                if ((size_t)(appended + add + 1) > next->ent.n) { // grow geometrically, keep what is there
                    DevBuf<int2> bigger;
                    bigger.alloc((size_t)std::max<i64>(appended + add + 1, (i64)((double)next->ent.n * 1.5)));
                    if (appended > 0) HIPCHK(hipMemcpyAsync(bigger.p, next->ent.p, (size_t)appended * sizeof(int2), hipMemcpyDeviceToDevice, stream));
                    HIPCHK(hipStreamSynchronize(stream));
                    next->ent = std::move(bigger);
                }
                app_scan.ensure((size_t)cnt + 2);
                hipLaunchKernelGGL(k_copy_len64, dim3(cdiv((i64)cnt + 1, 256)), dim3(256), 0, stream, cnt, R->S.len.p, app_scan.p);
                HIPCHK(hipGetLastError());
                R->scan.exclusive(app_scan.p, app_scan.p, (size_t)cnt + 1, stream);
                constexpr int TEAM = 16;
                hipLaunchKernelGGL((k_append_rows<TEAM>), dim3(cdiv((i64)cnt * TEAM, 256)), dim3(256), 0, stream, cnt, R->S.start.p, R->S.len.p,
                                   R->S.lead.p, R->S.orig.p, R->S.ent.p, app_scan.p, (i64d)appended, off, next->ent.p, next->start.p,
                                   next->len.p, next->lead.p, next->orig.p);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(stream));
                appended += add;
            }
            off += cnt;
            if (nnp == 0) break;
        }
        if (nbatch > 1) {
            // the pools of a batched round are as large as the device allows: give them back before the next round
            R->Lpool.release();
            R->S.ent.release();
        }
        R->hctr.applications = acc.applications;
        R->hctr.nnz_reduced = acc.nnz_reduced;
        R->hctr.segments = acc.segments;
        R->hctr.nonempty_out = acc.nonempty_out;
        R->hctr.nnz_out = acc.nnz_out;
        R->hctr.stream_fix = acc.stream_fix;
        R->hctr.stream_redo = acc.stream_redo;
        for (int c = 0; c < 16; c++) { R->hctr.class_ent[c] = acc.class_ent[c]; R->hctr.class_seg[c] = acc.class_seg[c]; }
        for (int c = 0; c < NCLASS; c++) R->hclass_count[c] = acc_class[c];
        {
            const int ubase = (int)U.pivcol.size();
            append_round_U(U, *R, *cur, stream);
            if (want_L) collect_L_pivots(HL, *R, U, ubase, stream);
        }

        spasm_amd_round_stats st;
        fill_stats(st, *R, round, cur->n, cur_nnz);
        st.ms_pivots = ms_pivots;
        st.ms_solve = ms_solve;
        st.ms_scatter = ms_scatter;
        st.ms_total = ms_pivots + ms_solve + ms_scatter;
        g_last_rounds.push_back(st);
        spasm_logf("[echelonize] round %d\n[pivots] Faugère-Lachartre: %d pivots found [%.1fs]\n", round, R->n_leftmost, st.ms_pivots * 1e-3);
        if (R->n_open) spasm_logf("[pivots] ``Faugère-Lachartre on columns'': %d pivots found\n", R->n_open);
        if (R->n_greedy) spasm_logf("[pivots] greedy alternating cycle-free search: %d pivots found\n", R->n_greedy);
        spasm_logf("Schur complement: %d * %d [%lld nz / density= %.3f], %.1fs%s\n", nnp, m - (int)U.pivcol.size(),
                   (long long)st.nnz_out, nnp > 0 && m > 0 ? (double)st.nnz_out / ((double)nnp * (double)m) : 0.0,
                   (st.ms_solve + st.ms_scatter) * 1e-3, nbatch > 1 ? " (in batches of rows)" : "");

        // the Schur complement becomes the matrix of the next round
        cur = std::move(next);
        cur_nnz = st.nnz_out;
        cur_live = st.rows_out;
        round++;
        if (cur_nnz == 0) break;
    }

    if (rank_only) {
        *rank_only = (i64)U.pivcol.size() + U.uncollected;
        spasm_logf("[echelonize] Done in %.1fs. Rank %lld%s (rank only: the rows of U stayed on the device)\n", spasm_wtime() - t0, (long long)*rank_only,
                   partial ? " (not finished: a lower bound)" : "");
        return nullptr;
    }
    struct spasm_lu *N = assemble_lu(U, n, m, prime, want_L ? &HL : nullptr);
    spasm_logf("[echelonize] Done in %.1fs. Rank %d%s, %lld nz in basis\n", spasm_wtime() - t0, N->r, partial ? " (not finished: a lower bound)" : "", (long long)U.p.back());
    return N;
}

// ---- the host LU from the rows of U collected round by round (layout reference src/SpaSM.jl:262-270; ownership :273-277)
struct spasm_lu *assemble_lu(HostU &U, int n, int m, i64 prime, HostL *HLp)
{
    const bool want_L = HLp != nullptr;
    const int r = (int)U.pivcol.size();
    struct spasm_csr *Uc = spasm_csr_alloc(r, m, 0, prime, true);
    if (!Uc) throw EngineError("out of host memory for U");
    memcpy(Uc->p, U.p.data(), sizeof(i64) * ((size_t)r + 1));
    if (!U.j.empty()) { // the buffers the entries were downloaded into become the arrays of U (malloc'ed, like spasm_csr_alloc's)
        free(Uc->j);
        free(Uc->x);
        Uc->nzmax = (i64)U.j.size();
        Uc->j = U.j.steal();
        Uc->x = U.x.steal();
    }
    struct spasm_lu *N = (struct spasm_lu *)malloc(sizeof *N);
    const int plen = std::max(std::max(n, m), 1);
    int *qinv = (int *)malloc(sizeof(int) * (size_t)std::max(m, 1));
    int *p = (int *)malloc(sizeof(int) * (size_t)plen);
    for (int j = 0; j < m; j++) qinv[j] = -1;
    for (int k = 0; k < r; k++) qinv[U.pivcol[(size_t)k]] = k;
    {
        // pivotal rows first, then the others in ascending order
        std::vector<char> used((size_t)std::max(n, 1), 0);
        int w = 0;
        for (int k = 0; k < r; k++) { p[w++] = U.orig[(size_t)k]; used[(size_t)U.orig[(size_t)k]] = 1; }
        for (int i = 0; i < n; i++) if (!used[(size_t)i]) p[w++] = i;
        for (; w < plen; w++) p[w] = -1;
    }
    struct spasm_csr *Lc = nullptr;
    if (want_L) {
        // rows of L in the order of the rows of A; inside a row the entries arrive round by round, i.e. by ascending row of U
        HostL &HL = *HLp;
        const i64 lnz = (i64)HL.j.size();
        Lc = spasm_csr_alloc(n, r, lnz, prime, true);
        if (!Lc) throw EngineError("out of host memory for L");
        std::vector<i64> cntr((size_t)n + 1, 0);
        for (i64 k = 0; k < lnz; k++) cntr[(size_t)HL.row[(size_t)k] + 1]++;
        for (int i = 0; i < n; i++) cntr[(size_t)i + 1] += cntr[(size_t)i];
        memcpy(Lc->p, cntr.data(), sizeof(i64) * ((size_t)n + 1));
        for (i64 k = 0; k < lnz; k++) {
            const i64 w = cntr[(size_t)HL.row[(size_t)k]]++;
            Lc->j[w] = HL.j[(size_t)k];
            Lc->x[w] = HL.x[(size_t)k];
        }
    }
    N->r = r;
    N->complete = false;
    N->L = Lc;
    N->U = Uc;
    N->qinv = qinv;
    N->p = p;
    N->Ltmp = nullptr;
    return N;
}


// ------------------------------------------------------------------------------------------------
// device transpose: column j of M becomes an output row when keep[j] < 0 (keep == NULL: every column);
// entries are labelled label[row] (NULL: the row index); diag = 1 prepends (j,-1) to every output row
// ------------------------------------------------------------------------------------------------
struct TransposeOut {
    int nrows = 0;
    i64 nnz = 0;
    DevBuf<i64d> Tp;
    DevBuf<int2> Tent;
};

void device_transpose(const DevMat &M, int nrowsM, const int *keep, const int *label, int diag, Scanner &scan, hipStream_t s, TransposeOut &out)
{
    const int m = M.m;
    DevBuf<int> cnt, flag, rowidx, cursor;
    DevBuf<i64d> tlen, tstart;
    cnt.alloc((size_t)m + 1);
    flag.alloc((size_t)m + 1);
    rowidx.alloc((size_t)m + 1);
    cursor.alloc((size_t)m + 1);
    tlen.alloc((size_t)m + 1);
    tstart.alloc((size_t)m + 1);
    cnt.zero(s);
    constexpr int TEAM = 8;
    if (nrowsM > 0) {
        hipLaunchKernelGGL((k_count_cols<TEAM>), dim3(cdiv((i64)nrowsM * TEAM, 256)), dim3(256), 0, s, nrowsM, M.start.p, M.len.p, M.ent.p, cnt.p);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_trow_len, dim3(cdiv((i64)m + 1, 256)), dim3(256), 0, s, m, keep, cnt.p, diag, tlen.p, flag.p);
    HIPCHK(hipGetLastError());
    scan.exclusive(tlen.p, tstart.p, (size_t)m + 1, s);
    scan.exclusive(flag.p, rowidx.p, (size_t)m + 1, s);
    i64d tot = 0;
    int nrows = 0;
    HIPCHK(hipMemcpyAsync(&tot, tstart.p + m, sizeof tot, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&nrows, rowidx.p + m, sizeof nrows, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    out.nrows = nrows;
    out.nnz = tot;
    out.Tp.alloc((size_t)nrows + 1);
    out.Tent.alloc((size_t)tot + 1);
    hipLaunchKernelGGL(k_trow_ptr, dim3(cdiv((i64)m + 1, 256)), dim3(256), 0, s, m, keep, rowidx.p, tstart.p, diag, out.Tp.p, out.Tent.p, cursor.p);
    HIPCHK(hipGetLastError());
    if (nrowsM > 0) {
        hipLaunchKernelGGL((k_tfill<TEAM>), dim3(cdiv((i64)nrowsM * TEAM, 256)), dim3(256), 0, s, nrowsM, M.start.p, M.len.p, M.ent.p, label, tstart.p, cursor.p, out.Tent.p);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(s));
}

struct spasm_csr *transpose_out_to_host(const TransposeOut &T, int ncols, i64 prime, hipStream_t s)
{
    struct spasm_csr *K = spasm_csr_alloc(T.nrows, ncols, T.nnz, prime, true);
    if (!K) throw EngineError("out of host memory");
    std::vector<int2> ent((size_t)T.nnz);
    HIPCHK(hipMemcpyAsync(K->p, T.Tp.p, ((size_t)T.nrows + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
    if (T.nnz > 0) HIPCHK(hipMemcpyAsync(ent.data(), T.Tent.p, (size_t)T.nnz * sizeof(int2), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (i64 k = 0; k < T.nnz; k++) { K->j[k] = ent[(size_t)k].x; K->x[k] = ent[(size_t)k].y; }
    return K;
}

struct spasm_csr *do_transpose(const struct spasm_csr *A)
{
    require_device();
    if (!A) throw EngineError("spasm_transpose: NULL matrix");
    hipStream_t s = nullptr;
    DevMat M;
    upload_csr(A, 0, A->n, M, s);
    Scanner scan;
    TransposeOut T;
    device_transpose(M, A->n, nullptr, nullptr, 0, scan, s, T);
    struct spasm_csr *R = transpose_out_to_host(T, A->n, A->field->p, s);
    if (!A->x) { free(R->x); R->x = nullptr; }
    return R;
}

// ------------------------------------------------------------------------------------------------
// Pivot numbering for the consumers of a factorization (kernel, rref, triangular solve): a topological order of
// "row a of U has an entry on the pivot column of row b  =>  a before b", which makes U_PP strictly upper triangular in
// pivot-index space -- what the solve kernels assume.  U produced by this engine is stored in such an order already
// (rounds in sequence; inside a round the open-column pivots, then the leftmost ones by ascending column), so the check
// is one pass; any other (permuted) triangular U gets a depth-first numbering.  Pivots need not be leftmost entries
// (reference src/SpaSM.jl:712 only promises unit pivots).  Returns perm (perm[t] = row of U with pivot index t) and fills
// pc (pivot column of each row).
// ------------------------------------------------------------------------------------------------
static std::vector<int> pivot_topological_order(const struct spasm_csr *U, const int *qinv, const char *who, std::vector<int> &pc)
{
    const int r = U->n, m = U->m;
    const std::string w(who);
    pc.assign((size_t)std::max(r, 1), -1);
    int named = 0;
    for (int j = 0; j < m; j++) {
        const int a = qinv[j];
        if (a >= r) throw EngineError(w + ": qinv points outside U");
        if (a >= 0) { if (pc[(size_t)a] < 0) named++; pc[(size_t)a] = j; }
    }
    if (named != r) throw EngineError(w + ": qinv does not name one pivot column per row of U");
    // (one pass over the entries of U, on all host threads: U has 10^9 entries on the large cases)
    int bad_unit = 0, unordered = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(| : bad_unit, unordered)
    for (int a = 0; a < r; a++) {
        bool unit = false;
        const int pca = pc[(size_t)a];
        for (i64 k = U->p[a]; k < U->p[a + 1]; k++) {
            const int j = U->j[k];
            if (j == pca) { if (U->x[k] == 1) unit = true; continue; }
            const int b = qinv[j];
            if (b >= 0 && b < a) unordered = 1;
        }
        if (!unit) bad_unit = 1;
    }
    if (bad_unit) throw EngineError(w + ": pivots of U must be 1");
    std::vector<int> perm((size_t)std::max(r, 1));
    const bool ordered = !unordered;
    if (ordered) {
        for (int a = 0; a < r; a++) perm[(size_t)a] = a;
        return perm;
    }
    std::vector<char> state((size_t)r, 0);
    std::vector<int> stack, post;
    std::vector<i64> pos((size_t)r, 0);
    post.reserve((size_t)r);
    for (int root = 0; root < r; root++) {
        if (state[(size_t)root]) continue;
        stack.push_back(root);
        state[(size_t)root] = 1;
        pos[(size_t)root] = U->p[root];
        while (!stack.empty()) {
            const int a = stack.back();
            bool pushed = false;
            while (pos[(size_t)a] < U->p[a + 1]) {
                const int b = qinv[U->j[pos[(size_t)a]++]];
                if (b < 0 || b == a) continue;
                if (state[(size_t)b] == 1) throw EngineError(w + ": U is not (permuted) triangular");
                if (state[(size_t)b] == 0) {
                    state[(size_t)b] = 1;
                    pos[(size_t)b] = U->p[b];
                    stack.push_back(b);
                    pushed = true;
                    break;
                }
            }
            if (!pushed) { state[(size_t)a] = 2; post.push_back(a); stack.pop_back(); }
        }
    }
    for (int t = 0; t < r; t++) perm[(size_t)t] = post[(size_t)(r - 1 - t)];
    return perm;
}

// ------------------------------------------------------------------------------------------------
// kernel basis from an echelonized U (reference call site src/SpaSM.jl:879).
// The reduced row echelon form R of U is computed with the SAME solve + scatter kernels as a Schur
// round (each row of U is reduced by all the others); the kernel vector of free column j is then
// -e_j + sum_a R[a][j] e_{pivcol(a)}: a transpose of R's free part (known answers test/runtests.jl:20-23).
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// spasm_rref (reference src/SpaSM.jl:871): the reduced row echelon form of U.  Row k of R is row k of U minus the
// combination of the OTHER rows that clears its entries on their pivot columns -- one Schur "round" in which every row
// of U is a pivot row and, at the same time, a row to reduce with its own pivot excluded (self_idx).  The pivots are
// numbered in a topological order (pivot_topological_order), so they need not be leftmost entries.
// ------------------------------------------------------------------------------------------------
struct spasm_csr *do_rref(const struct spasm_lu *fact, int *Rqinv)
{
    require_device();
    if (!fact || !fact->U || !fact->qinv) throw EngineError("spasm_rref: incomplete factorization (U / qinv missing)");
    const struct spasm_csr *U = fact->U;
    check_input(U, "spasm_rref");
    const int r = U->n, m = U->m;
    const int *qinv = fact->qinv;
    const i64 prime = U->field->p;
    // pivots numbered in topological order: idx -> (column, row of U)
    std::vector<int> pc;
    const std::vector<int> perm = pivot_topological_order(U, qinv, "spasm_rref", pc);
    std::vector<int> h_qinv_r((size_t)std::max(m, 1), -1), h_pivcol((size_t)r), h_pivrow((size_t)r), h_self((size_t)std::max(r, 1), -1);
    for (int t = 0; t < r; t++) {
        const int k = perm[(size_t)t];
        h_qinv_r[(size_t)pc[(size_t)k]] = t;
        h_self[(size_t)k] = t;
        h_pivcol[(size_t)t] = pc[(size_t)k];
        h_pivrow[(size_t)t] = k;
    }
    struct spasm_csr *Rm = nullptr;
    std::vector<i64> sp((size_t)r + 1, 0);
    std::vector<int> sj, sx;
    if (r > 0) {
        hipStream_t s = nullptr;
        DevMat PM;
        upload_csr(U, 0, r, PM, s);
        std::unique_ptr<Round> R(new Round());
        R->F = zp_field_make(prime);
        R->stream = s;
        R->m = m;
        R->npiv = r;
        R->nnp = r;
        R->qinv_r.alloc((size_t)m + 1);
        R->pivcol.alloc((size_t)r + 1);
        R->pivrow.alloc((size_t)r + 1);
        R->np_rows.alloc((size_t)r + 1);
        DevBuf<int> self;
        self.alloc((size_t)r + 1);
        HIPCHK(hipMemcpyAsync(R->qinv_r.p, h_qinv_r.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(R->pivcol.p, h_pivcol.data(), (size_t)r * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(R->pivrow.p, h_pivrow.data(), (size_t)r * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(self.p, h_self.data(), (size_t)r * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_iota, dim3(cdiv(r, 256)), dim3(256), 0, s, r, R->np_rows.p);
        HIPCHK(hipGetLastError());
        R->build_U(PM, R->pivrow.p);
        R->prepare_uinv(r);
        const i64 tot = R->solve_phase(PM, R->np_rows.p, self.p, r, 4 * spasm_nnz(U));
        R->S.ent.ensure((size_t)tot + 1);
        R->run_scatter(PM, R->np_rows.p, r);
        R->fetch_counters();
        // the non-pivot parts of the reduced rows, compacted on the device
        DevBuf<i64d> len64, ostart;
        len64.alloc((size_t)r + 1);
        ostart.alloc((size_t)r + 1);
        hipLaunchKernelGGL(k_copy_len64, dim3(cdiv((i64)r + 1, 256)), dim3(256), 0, s, r, R->S.len.p, len64.p);
        HIPCHK(hipGetLastError());
        R->scan.exclusive(len64.p, ostart.p, (size_t)r + 1, s);
        HIPCHK(hipMemcpyAsync(sp.data(), ostart.p, ((size_t)r + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const i64 nz = sp[(size_t)r];
        DevBuf<int> oj, ox;
        oj.alloc((size_t)nz + 1);
        ox.alloc((size_t)nz + 1);
        constexpr int TEAM = 16;
        hipLaunchKernelGGL((k_compact_rows<TEAM>), dim3(cdiv((i64)r * TEAM, 256)), dim3(256), 0, s, r, R->S.start.p, R->S.len.p, R->S.ent.p, ostart.p, oj.p, ox.p);
        HIPCHK(hipGetLastError());
        sj.resize((size_t)nz);
        sx.resize((size_t)nz);
        if (nz > 0) {
            HIPCHK(hipMemcpyAsync(sj.data(), oj.p, (size_t)nz * sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(sx.data(), ox.p, (size_t)nz * sizeof(int), hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipStreamSynchronize(s));
    }
    const i64 total = (r > 0 ? sp[(size_t)r] : 0) + r;
    Rm = spasm_csr_alloc(r, m, total, prime, true);
    if (!Rm) throw EngineError("out of host memory for R");
    i64 w = 0;
    for (int k = 0; k < r; k++) {
        Rm->p[k] = w;
        Rm->j[w] = h_pivcol[(size_t)h_self[(size_t)k]];
        Rm->x[w] = 1;
        w++;
        const i64 a = sp[(size_t)k], b = sp[(size_t)k + 1];
        if (b > a) {
            memcpy(Rm->j + w, sj.data() + a, (size_t)(b - a) * sizeof(int));
            memcpy(Rm->x + w, sx.data() + a, (size_t)(b - a) * sizeof(int));
            w += b - a;
        }
    }
    Rm->p[r] = w;
    if (Rqinv) for (int j = 0; j < m; j++) Rqinv[j] = qinv[j];
    return Rm;
}

// ------------------------------------------------------------------------------------------------
// X * U = B for every row of B at once (the reference loops spasm_sparse_triangular_solve over the rows of B,
// src/SpaSM.jl:733-755): with x_b on the pivot columns and x_a on the others, x_b * U + x_a == B[k] (:694-713).  One Schur
// "round" with U as the pivot rows and B as the rows to reduce: the multipliers ARE x_b, the Schur row IS x_a.
// Returns X (rows of B x rows of U); ok[k] = 1 when x_a is empty, i.e. row k has a solution.  Same pivot numbering as rref.
// ------------------------------------------------------------------------------------------------
// resid (may be NULL): receives x_a, the part of B[k] - x_b * U on the columns without a pivot, one row per row of B
struct spasm_csr *do_trisolve(const struct spasm_csr *U, const int *qinv, const struct spasm_csr *B, unsigned char *ok, struct spasm_csr **resid)
{
    require_device();
    if (!U || !qinv || !B) throw EngineError("spasm_amd_triangular_solve: null argument");
    check_input(U, "spasm_amd_triangular_solve");
    check_input(B, "spasm_amd_triangular_solve");
    const int r = U->n, m = U->m, nb = B->n;
    const i64 prime = U->field->p;
    if (B->m != m || B->field->p != prime) throw EngineError("spasm_amd_triangular_solve: B and U differ in columns or field");
    std::vector<int> pc;
    const std::vector<int> perm = pivot_topological_order(U, qinv, "spasm_amd_triangular_solve", pc);
    std::vector<int> h_qinv_r((size_t)std::max(m, 1), -1), h_pivcol((size_t)r), h_pivrow((size_t)r);
    for (int t = 0; t < r; t++) {
        const int k = perm[(size_t)t];
        h_qinv_r[(size_t)pc[(size_t)k]] = t;
        h_pivcol[(size_t)t] = pc[(size_t)k];
        h_pivrow[(size_t)t] = k;
    }
    std::vector<i64> xp((size_t)nb + 1, 0);
    std::vector<int> slen((size_t)std::max(nb, 1), 0);
    struct spasm_csr *X = nullptr;
    if (nb > 0 && r > 0) {
        hipStream_t s = nullptr;
        DevMat PM, BM;
        upload_csr(U, 0, r, PM, s);
        upload_csr(B, 0, nb, BM, s);
        std::unique_ptr<Round> R(new Round());
        R->F = zp_field_make(prime);
        R->stream = s;
        R->m = m;
        R->npiv = r;
        R->nnp = nb;
        R->qinv_r.alloc((size_t)m + 1);
        R->pivcol.alloc((size_t)r + 1);
        R->pivrow.alloc((size_t)r + 1);
        R->np_rows.alloc((size_t)nb + 1);
        HIPCHK(hipMemcpyAsync(R->qinv_r.p, h_qinv_r.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(R->pivcol.p, h_pivcol.data(), (size_t)r * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(R->pivrow.p, h_pivrow.data(), (size_t)r * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_iota, dim3(cdiv(nb, 256)), dim3(256), 0, s, nb, R->np_rows.p);
        HIPCHK(hipGetLastError());
        R->want_idx = true; // X is assembled from (pivot index, multiplier)
        R->build_U(PM, R->pivrow.p);
        R->prepare_uinv(nb);
        const i64 tot = R->solve_phase(BM, R->np_rows.p, nullptr, nb, 4 * (spasm_nnz(B) + spasm_nnz(U)));
        R->S.ent.ensure((size_t)tot + 1);
        R->run_scatter(BM, R->np_rows.p, nb);
        R->fetch_counters();
        DevBuf<i64d> xlen, xstart;
        xlen.alloc((size_t)nb + 1);
        xstart.alloc((size_t)nb + 1);
        hipLaunchKernelGGL(k_xcount, dim3(cdiv(((i64)nb + 1) * 64, 256)), dim3(256), 0, s, nb, R->Lstart.p, R->Llen.p, R->Lpool.p, xlen.p);
        HIPCHK(hipGetLastError());
        R->scan.exclusive(xlen.p, xstart.p, (size_t)nb + 1, s);
        HIPCHK(hipMemcpyAsync(xp.data(), xstart.p, ((size_t)nb + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(slen.data(), R->S.len.p, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const i64 nz = xp[(size_t)nb];
        X = spasm_csr_alloc(nb, r, nz, prime, true);
        if (!X) throw EngineError("out of host memory for X");
        if (nz > 0) {
            DevBuf<int> oj, ox;
            oj.alloc((size_t)nz);
            ox.alloc((size_t)nz);
            hipLaunchKernelGGL(k_xfill, dim3(cdiv((i64)nb * 64, 256)), dim3(256), 0, s, nb, R->pivrow.p, R->Lstart.p, R->Llen.p, R->Lpool.p, R->Lidx.p, xstart.p,
                               oj.p, ox.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(X->j, oj.p, (size_t)nz * sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(X->x, ox.p, (size_t)nz * sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        }
        for (int k = 0; k <= nb; k++) X->p[k] = xp[(size_t)k];
        if (resid) {
            // x_a: the Schur rows of the rows of B, compacted (device) and copied out
            DevBuf<i64d> l64, rstart2;
            l64.alloc((size_t)nb + 1);
            rstart2.alloc((size_t)nb + 1);
            hipLaunchKernelGGL(k_copy_len64, dim3(cdiv((i64)nb + 1, 256)), dim3(256), 0, s, nb, R->S.len.p, l64.p);
            HIPCHK(hipGetLastError());
            R->scan.exclusive(l64.p, rstart2.p, (size_t)nb + 1, s);
            std::vector<i64d> hp2((size_t)nb + 1);
            HIPCHK(hipMemcpyAsync(hp2.data(), rstart2.p, ((size_t)nb + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            const i64 rz = hp2[(size_t)nb];
            struct spasm_csr *Rs = spasm_csr_alloc(nb, m, rz, prime, true);
            if (!Rs) { spasm_csr_free(X); throw EngineError("out of host memory for the residual rows"); }
            for (int k = 0; k <= nb; k++) Rs->p[k] = hp2[(size_t)k];
            if (rz > 0) {
                DevBuf<int> oj2, ox2;
                oj2.alloc((size_t)rz);
                ox2.alloc((size_t)rz);
                constexpr int TEAM = 16;
                hipLaunchKernelGGL((k_compact_rows<TEAM>), dim3(cdiv((i64)nb * TEAM, 256)), dim3(256), 0, s, nb, R->S.start.p, R->S.len.p, R->S.ent.p, rstart2.p, oj2.p, ox2.p);
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpyAsync(Rs->j, oj2.p, (size_t)rz * sizeof(int), hipMemcpyDeviceToHost, s));
                HIPCHK(hipMemcpyAsync(Rs->x, ox2.p, (size_t)rz * sizeof(int), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
            }
            *resid = Rs;
        }
    } else {
        X = spasm_csr_alloc(nb, r, 0, prime, true);
        if (!X) throw EngineError("out of host memory for X");
        for (int k = 0; k <= nb; k++) X->p[k] = 0;
        for (int k = 0; k < nb; k++) slen[(size_t)k] = (int)(B->p[k + 1] - B->p[k]); // without pivots a row is solvable iff it is zero
        if (resid) {
            // no pivots (or no rows): x_a is the row itself
            const i64 bz = B->p[nb];
            struct spasm_csr *Rs = spasm_csr_alloc(nb, m, bz, prime, true);
            if (!Rs) { spasm_csr_free(X); throw EngineError("out of host memory for the residual rows"); }
            const ZpField F = zp_field_make(prime);
            for (int k = 0; k <= nb; k++) Rs->p[k] = B->p[k];
            for (i64 t = 0; t < bz; t++) { Rs->j[t] = B->j[t]; Rs->x[t] = zp_reduce(F, B->x[t]); }
            *resid = Rs;
        }
    }
    if (ok) for (int k = 0; k < nb; k++) ok[k] = slen[(size_t)k] == 0;
    return X;
}

// ------------------------------------------------------------------------------------------------
// spasm_gesv / spasm_solve (reference src/SpaSM.jl:895-923): X * A == B from a factorization that carries L (echelonize_opts.L).
// With A[i] = sum_k L[i][k] U[k]:  X * A = (X * L) * U, so   Y * U = B  (the batched triangular solve on the device, Y is
// nb x r), then  X_P * L_P = Y  on the pivotal rows p[0 .. r) of A, whose rows of L form a lower triangular r x r matrix with the
// pivots d_k = L[p[k]][k] on the diagonal (a row only holds multipliers of pivots elected before it).  Scaled to a unit diagonal
// this is a second triangular solve of the same kind; X is zero outside the pivotal rows.  ok[b] = 1 when row b has a solution.
// ------------------------------------------------------------------------------------------------
struct spasm_csr *do_trisolve(const struct spasm_csr *U, const int *qinv, const struct spasm_csr *B, unsigned char *ok, struct spasm_csr **resid = nullptr);

struct spasm_csr *do_gesv(const struct spasm_lu *fact, const struct spasm_csr *B, unsigned char *ok)
{
    if (!fact || !fact->U || !fact->qinv || !fact->p) throw EngineError("incomplete factorization (U / qinv / p missing)");
    if (!fact->L) throw EngineError("the factorization has no L: echelonize with L = true (reference src/SpaSM.jl:331)");
    if (!B) throw EngineError("null right-hand side");
    const struct spasm_csr *U = fact->U, *L = fact->L;
    const int r = U->n, n = L->n, nb = B->n;
    const i64 prime = U->field->p;
    if (L->m != r) throw EngineError("L must have one column per row of U");
    const ZpField F = zp_field_make(prime);
    // the pivotal rows of L, scaled to a unit diagonal
    std::vector<int> diag((size_t)std::max(r, 1), 0);
    i64 mnz = 0;
    for (int k = 0; k < r; k++) {
        const int i = fact->p[k];
        if (i < 0 || i >= n) throw EngineError("p does not name a row of A for every row of U");
        mnz += L->p[i + 1] - L->p[i];
    }
    struct spasm_csr *M = spasm_csr_alloc(r, r, mnz, prime, true);
    if (!M) throw EngineError("out of host memory");
    std::unique_ptr<struct spasm_csr, void (*)(struct spasm_csr *)> Mguard(M, spasm_csr_free);
    i64 w = 0;
    for (int k = 0; k < r; k++) {
        const int i = fact->p[k];
        M->p[k] = w;
        int d = 0;
        for (i64 q = L->p[i]; q < L->p[i + 1]; q++) {
            if (L->j[q] > k) throw EngineError("L is not triangular on the pivotal rows");
            if (L->j[q] == k) d = L->x[q];
        }
        if (d == 0) throw EngineError("L has a zero on its diagonal");
        diag[(size_t)k] = d;
        const int dinv = zp_inverse(F, d);
        for (i64 q = L->p[i]; q < L->p[i + 1]; q++) {
            M->j[w] = L->j[q];
            M->x[w] = L->j[q] == k ? 1 : zp_mul(F, dinv, L->x[q]);
            w++;
        }
    }
    M->p[r] = w;
    std::vector<int> ident((size_t)std::max(r, 1));
    for (int k = 0; k < r; k++) ident[(size_t)k] = k;
    std::vector<unsigned char> ok1((size_t)std::max(nb, 1), 0), ok2((size_t)std::max(nb, 1), 0);
    struct spasm_csr *Y = do_trisolve(U, fact->qinv, B, ok1.data());
    std::unique_ptr<struct spasm_csr, void (*)(struct spasm_csr *)> Yguard(Y, spasm_csr_free);
    struct spasm_csr *Xs = do_trisolve(M, ident.data(), Y, ok2.data());
    std::unique_ptr<struct spasm_csr, void (*)(struct spasm_csr *)> Xguard(Xs, spasm_csr_free);
    // X[b][p[k]] = Xs[b][k] / d_k
    struct spasm_csr *X = spasm_csr_alloc(nb, n, spasm_nnz(Xs), prime, true);
    if (!X) throw EngineError("out of host memory for X");
    std::vector<int> dinv((size_t)std::max(r, 1));
    for (int k = 0; k < r; k++) dinv[(size_t)k] = zp_inverse(F, diag[(size_t)k]);
    for (int b = 0; b <= nb; b++) X->p[b] = Xs->p[b];
    for (i64 q = 0; q < Xs->p[nb]; q++) {
        const int k = Xs->j[q];
        X->j[q] = fact->p[k];
        X->x[q] = zp_mul(F, Xs->x[q], dinv[(size_t)k]);
    }
    if (ok) for (int b = 0; b < nb; b++) ok[b] = ok1[(size_t)b] && ok2[(size_t)b];
    return X;
}

// first / step: only the free columns number first, first + step, ... (in ascending column order) get their kernel vector:
// the unit of the multi-GPU kernel step (free columns are independent)
struct spasm_csr *do_kernel_core(const struct spasm_csr *U, const int *qinv, int first, int step);

// The part of U a kernel basis needs.  The kernel vector of free column j is -e_j + sum_a y_a e_{pivcol(a)} with
//        y_a = U[a][j] - sum_{b != a} U[a][pivcol(b)] y_b,
// so y_a can only be non-zero when row a holds an entry on a (selected) free column or on the pivot column of a row b that can:
// the rows outside that closure have y = 0 for every free column and drop out, with the pivot columns they own (qinv = -2 below:
// neither free nor a pivot of what is left).  For the factorizations whose U does not fit the device as one round (config 5 at
// 1/3: 5.5e9 entries; the round's U has 32-bit offsets) the closure is what is solved instead -- a few rows when the kernel
// vectors are sparse, all of U when they are not (then the limit stands).  Host side, all threads: one pass for the free columns,
// then sweeps over the rows not yet in the closure until nothing changes.
struct KernelReduced {
    struct spasm_csr *U2 = nullptr;
    std::vector<int> qinv2;
    ~KernelReduced() { if (U2) spasm_csr_free(U2); }
};

void kernel_closure(const struct spasm_csr *U, const int *qinv, int first, int step, KernelReduced &out)
{
    const int r = U->n, m = U->m;
    std::vector<unsigned char> sel((size_t)std::max(m, 1), 0);
    {
        int f = 0;
        for (int j = 0; j < m; j++)
            if (qinv[j] < 0 && qinv[j] != -2) {
                if (f >= first && (f - first) % step == 0) sel[(size_t)j] = 1;
                f++;
            }
    }
    std::vector<unsigned char> need((size_t)std::max(r, 1), 0);
    unsigned char *nd = need.data();
    int any = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(| : any)
    for (int a = 0; a < r; a++)
        for (i64 k = U->p[a]; k < U->p[a + 1]; k++)
            if (sel[(size_t)U->j[k]] && U->x[k] != 0) { __atomic_store_n(&nd[a], 1, __ATOMIC_RELAXED); any = 1; break; }
    int sweeps = 0;
    for (int changed = any; changed; sweeps++) {
        changed = 0;
        // (descending: in a U whose rows come in elimination order a row refers to later rows, and a chain closes in one sweep)
#pragma omp parallel for schedule(static, 4096) reduction(| : changed)
        for (int i = 0; i < r; i++) {
            const int a = r - 1 - i;
            if (__atomic_load_n(&nd[a], __ATOMIC_RELAXED)) continue;
            for (i64 k = U->p[a]; k < U->p[a + 1]; k++) {
                const int b = qinv[U->j[k]];
                if (b >= 0 && b != a && U->x[k] != 0 && __atomic_load_n(&nd[b], __ATOMIC_RELAXED)) {
                    __atomic_store_n(&nd[a], 1, __ATOMIC_RELAXED);
                    changed = 1;
                    break;
                }
            }
        }
    }
    std::vector<int> newidx((size_t)std::max(r, 1), -1);
    int r2 = 0;
    for (int a = 0; a < r; a++) if (need[(size_t)a]) newidx[(size_t)a] = r2++;
    out.qinv2.assign((size_t)std::max(m, 1), -1);
    for (int j = 0; j < m; j++) {
        const int a = qinv[j];
        out.qinv2[(size_t)j] = a < 0 ? a : (newidx[(size_t)a] >= 0 ? newidx[(size_t)a] : -2);
    }
    std::vector<i64> cnt((size_t)r2 + 1, 0);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int a = 0; a < r; a++) {
        if (!need[(size_t)a]) continue;
        i64 c = 0;
        for (i64 k = U->p[a]; k < U->p[a + 1]; k++) c += out.qinv2[(size_t)U->j[k]] != -2;
        cnt[(size_t)newidx[(size_t)a] + 1] = c;
    }
    for (int t = 0; t < r2; t++) cnt[(size_t)t + 1] += cnt[(size_t)t];
    out.U2 = spasm_csr_alloc(r2, m, std::max<i64>(cnt[(size_t)r2], 1), U->field->p, true);
    if (!out.U2) throw EngineError("out of host memory");
    for (int t = 0; t <= r2; t++) out.U2->p[t] = cnt[(size_t)t];
#pragma omp parallel for schedule(dynamic, 1024)
    for (int a = 0; a < r; a++) {
        if (!need[(size_t)a]) continue;
        i64 w = cnt[(size_t)newidx[(size_t)a]];
        for (i64 k = U->p[a]; k < U->p[a + 1]; k++)
            if (out.qinv2[(size_t)U->j[k]] != -2) { out.U2->j[w] = U->j[k]; out.U2->x[w] = U->x[k]; w++; }
    }
    spasm_logf("[kernel] closure of the free columns: %d of %d rows of U, %lld of %lld entries (%d sweeps)\n", r2, r, (long long)cnt[(size_t)r2], (long long)spasm_nnz(U), sweeps);
}

#include "kernel_dense.hpp"

// spasm_kernel through a dense right-hand side (kernel_dense.hpp): every free column at once, the dense tail of U on the int8 GEMM,
// the sparse rows level by level.  For primes below 2^16 (the element types of the dense finish).  dense_tail_rows < 0: chosen by
// density.  Returns NULL when Y (r x free columns) or the dense tail do not fit the device: the caller falls back.
template <typename DT>
struct spasm_csr *kernel_dense_rhs(const struct spasm_csr *U, const int *qinv, int first, int step, i64 dense_tail_rows)
{
    const int r = U->n, m = U->m;
    const i64 prime = U->field->p;
    const ZpField F = zp_field_make(prime);
    const double t_start = spasm_wtime();
    spasm_logf("[kernel] start. U is %d x %d (%lld nnz). All free columns at once: a dense right-hand side\n", r, m, (long long)spasm_nnz(U));
    std::vector<int> pc;
    const std::vector<int> perm = pivot_topological_order(U, qinv, "spasm_kernel", pc);
    std::vector<int> pos_of_row((size_t)std::max(r, 1), 0), pivcol_of_pos((size_t)std::max(r, 1), 0);
    for (int t = 0; t < r; t++) { pos_of_row[(size_t)perm[(size_t)t]] = t; pivcol_of_pos[(size_t)t] = pc[(size_t)perm[(size_t)t]]; }
    std::vector<int> h_free, fidx((size_t)std::max(m, 1), -1);
    {
        int f = 0;
        for (int j = 0; j < m; j++)
            if (qinv[j] < 0 && qinv[j] != -2) {
                if (f >= first && (f - first) % step == 0) { fidx[(size_t)j] = (int)h_free.size(); h_free.push_back(j); }
                f++;
            }
    }
    const int nf = (int)h_free.size();
    if (nf == 0) {
        struct spasm_csr *K = spasm_csr_alloc(0, m, 1, prime, true);
        if (!K) throw EngineError("out of host memory");
        K->p[0] = 0;
        return K;
    }
    const i64 ldz = ((i64)nf + 63) / 64 * 64;
    size_t fr = 0, tot_mem = 0;
    HIPCHK(hipMemGetInfo(&fr, &tot_mem));
    if ((double)std::max(r, 1) * (double)ldz * sizeof(DT) > 0.4 * (double)fr) return nullptr;
    // ---- the dense tail: the longest run of last positions that is dense enough to be worth a dense block (any run is CORRECT: a
    // row only refers to pivots behind it, so the rows of a tail refer to the tail and to free columns only)
    int t0 = r;
    {
        const double budget = 0.35 * (double)fr;
        if (dense_tail_rows >= 0) t0 = r - (int)std::min<i64>(dense_tail_rows, r);
        else {
            // The rows of a dense finish come last in U and hold nearly all of its entries: take the shortest tail that holds 99 %
            // (then 90, 75, 50 %) of the entries of U, provided it IS dense (a twentieth of its triangle + free columns) and fits.
            // (Per-row rules fail on the residual rows of a tall finish, which are short and still belong to the tail.)
            std::vector<double> cum((size_t)r + 1, 0.0); // cum[t] = entries of the positions t .. r-1
            for (int t = r - 1; t >= 0; t--) { const int a = perm[(size_t)t]; cum[(size_t)t] = cum[(size_t)t + 1] + (double)(U->p[a + 1] - U->p[a]); }
            const double cover[4] = {0.99, 0.90, 0.75, 0.50};
            for (int q = 0; q < 4 && t0 == r; q++) {
                int lo = 0, hi = r; // the largest t with cum[t] >= cover * total (cum is non-increasing in t)
                while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (cum[(size_t)mid] >= cover[q] * cum[0]) lo = mid; else hi = mid - 1; }
                const double rows = (double)(r - lo);
                if (rows >= 256 && cum[(size_t)lo] >= 0.05 * rows * (0.5 * rows + nf) && rows * (rows + nf + 64) * sizeof(DT) <= budget) t0 = lo;
            }
        }
        const double rows = (double)(r - t0);
        if (rows * (rows + nf + 64) * sizeof(DT) > budget) return nullptr;
    }
    const int rd = r - t0;
    hipStream_t s = nullptr;
    DevBuf<DT> Y;
    Y.alloc((size_t)std::max(r, 1) * (size_t)ldz);
    Y.zero(s);
    DevBuf<int> d_qinv, d_pos, d_fidx;
    d_qinv.alloc((size_t)m + 1); d_pos.alloc((size_t)r + 1); d_fidx.alloc((size_t)m + 1);
    HIPCHK(hipMemcpyAsync(d_qinv.p, qinv, (size_t)m * sizeof(int), hipMemcpyHostToDevice, s));
    if (r > 0) HIPCHK(hipMemcpyAsync(d_pos.p, pos_of_row.data(), (size_t)r * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_fidx.p, fidx.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, s));
    const double t_prep = spasm_wtime();
    double t_dense_rows = t_prep, t_z = t_prep;
    if (rd > 0) {
        // D1[k][0 .. rd) = the pivot columns of the tail in position order (the pivot of row k on column k), [rd, rd + nf) = the free columns
        const i64 ldc = ((i64)rd + nf + 63) / 64 * 64;
        DevBuf<DT> D1;
        D1.alloc((size_t)rd * (size_t)ldc);
        D1.zero(s);
        const i64 CH = (i64)1 << 28; // entries per chunk of rows on their way to the device
        std::vector<i64> hptr;
        std::vector<int> hj, hx;
        DevBuf<i64d> dptr;
        DevBuf<int> dj, dx;
        // (rows that lie one behind the other in U -- the usual case: U in elimination order -- go to the device from where they are)
        bool contig = true;
        for (int t = t0; t < r && contig; t++) contig = perm[(size_t)t] == perm[(size_t)t0] + (t - t0);
        for (int ta = t0; ta < r;) {
            int tb = ta;
            i64 ne = 0;
            while (tb < r) {
                const int a = perm[(size_t)tb];
                const i64 len = U->p[a + 1] - U->p[a];
                if (tb > ta && ne + len > CH) break;
                ne += len;
                tb++;
            }
            const int nk = tb - ta;
            hptr.assign((size_t)nk + 1, 0);
            for (int k = 0; k < nk; k++) { const int a = perm[(size_t)(ta + k)]; hptr[(size_t)k + 1] = hptr[(size_t)k] + (U->p[a + 1] - U->p[a]); }
            const int *src_j = nullptr, *src_x = nullptr;
            if (contig) {
                const i64 base = U->p[perm[(size_t)ta]];
                src_j = U->j + base; src_x = U->x + base;
            } else {
                hj.resize((size_t)std::max<i64>(ne, 1)); hx.resize((size_t)std::max<i64>(ne, 1));
#pragma omp parallel for schedule(dynamic, 64)
                for (int k = 0; k < nk; k++) {
                    const int a = perm[(size_t)(ta + k)];
                    const i64 len = U->p[a + 1] - U->p[a];
                    memcpy(hj.data() + hptr[(size_t)k], U->j + U->p[a], (size_t)len * sizeof(int));
                    memcpy(hx.data() + hptr[(size_t)k], U->x + U->p[a], (size_t)len * sizeof(int));
                }
                src_j = hj.data(); src_x = hx.data();
            }
            dptr.ensure((size_t)nk + 1); dj.ensure((size_t)std::max<i64>(ne, CH) + 1); dx.ensure((size_t)std::max<i64>(ne, CH) + 1);
            HIPCHK(hipMemcpyAsync(dptr.p, hptr.data(), ((size_t)nk + 1) * sizeof(i64), hipMemcpyHostToDevice, s));
            if (ne > 0) {
                HIPCHK(hipMemcpyAsync(dj.p, src_j, (size_t)ne * sizeof(int), hipMemcpyHostToDevice, s));
                HIPCHK(hipMemcpyAsync(dx.p, src_x, (size_t)ne * sizeof(int), hipMemcpyHostToDevice, s));
            }
            hipLaunchKernelGGL((k_kd_dense_rows<DT>), dim3(nk), dim3(256), 0, s, nk, ta - t0, dptr.p, dj.p, dx.p, d_qinv.p, d_pos.p, d_fidx.p, t0, rd, D1.p, (i64d)ldc);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(s)); // (the staging vectors are reused)
            ta = tb;
        }
        t_dense_rows = spasm_wtime();
        DevBuf<int> prow, fcol; // (prow = pcol = 0 .. rd-1)
        prow.alloc((size_t)rd + 1); fcol.alloc((size_t)nf + 1);
        hipLaunchKernelGGL(k_iota, dim3(cdiv(rd, 256)), dim3(256), 0, s, rd, prow.p);
        hipLaunchKernelGGL(k_iota_from, dim3(cdiv(nf, 256)), dim3(256), 0, s, nf, rd, fcol.p);
        HIPCHK(hipGetLastError());
        TallWork<DT> W(F, s);
        tall_reduced_form<DT>(W, D1.p, ldc, rd, prow.p, prow.p, fcol.p, nf, Y.p + (size_t)t0 * (size_t)ldz, ldz, F, s);
        HIPCHK(hipStreamSynchronize(s));
        t_z = spasm_wtime();
    }
    // ---- the sparse rows: positions [0, t0), level by level (level 0: rows that refer to no other sparse row)
    int nlev = 0;
    if (t0 > 0) {
        std::vector<int> lev((size_t)t0, 0);
        std::vector<i64> eptr((size_t)t0 + 1, 0);
        for (int t = t0 - 1; t >= 0; t--) { // (descending: a row refers to positions behind it)
            const int a = perm[(size_t)t];
            int lv = 0;
            i64 n = 0;
            for (i64 k = U->p[a]; k < U->p[a + 1]; k++) {
                const int c = U->j[k];
                const int b = qinv[c];
                if (b == a || U->x[k] == 0) continue;
                if (b >= 0) { const int pos = pos_of_row[(size_t)b]; if (pos < t0) lv = std::max(lv, lev[(size_t)pos] + 1); n++; }
                else if (fidx[(size_t)c] >= 0) n++;
            }
            lev[(size_t)t] = lv;
            eptr[(size_t)t + 1] = n;
            nlev = std::max(nlev, lv + 1);
        }
        for (int t = 0; t < t0; t++) eptr[(size_t)t + 1] += eptr[(size_t)t];
        const i64 ne = eptr[(size_t)t0];
        // (a U whose dense rows did not end up in the tail would be solved row after row here: hours.  Not this path's case.)
        if ((double)ne * (double)nf > 4e14 || (nlev > (1 << 20) && (double)ne > 1e9)) return nullptr;
        std::vector<int2> hent((size_t)std::max<i64>(ne, 1));
#pragma omp parallel for schedule(dynamic, 1024)
        for (int t = 0; t < t0; t++) {
            const int a = perm[(size_t)t];
            i64 w = eptr[(size_t)t];
            for (i64 k = U->p[a]; k < U->p[a + 1]; k++) {
                const int c = U->j[k];
                const int b = qinv[c];
                if (b == a || U->x[k] == 0) continue;
                if (b >= 0) hent[(size_t)w++] = make_int2(pos_of_row[(size_t)b], U->x[k]);
                else if (fidx[(size_t)c] >= 0) hent[(size_t)w++] = make_int2(-1 - fidx[(size_t)c], U->x[k]);
            }
        }
        std::vector<int> lstart((size_t)nlev + 1, 0), order((size_t)t0);
        for (int t = 0; t < t0; t++) lstart[(size_t)lev[(size_t)t] + 1]++;
        for (int l = 0; l < nlev; l++) lstart[(size_t)l + 1] += lstart[(size_t)l];
        {
            std::vector<int> at(lstart.begin(), lstart.end() - 1);
            for (int t = 0; t < t0; t++) order[(size_t)at[(size_t)lev[(size_t)t]]++] = t;
        }
        DevBuf<i64d> dptr;
        DevBuf<int2> dent;
        DevBuf<int> dord;
        dptr.alloc((size_t)t0 + 1); dent.alloc((size_t)ne + 1); dord.alloc((size_t)t0 + 1);
        HIPCHK(hipMemcpyAsync(dptr.p, eptr.data(), ((size_t)t0 + 1) * sizeof(i64), hipMemcpyHostToDevice, s));
        if (ne > 0) HIPCHK(hipMemcpyAsync(dent.p, hent.data(), (size_t)ne * sizeof(int2), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dord.p, order.data(), (size_t)t0 * sizeof(int), hipMemcpyHostToDevice, s));
        const int ctiles = cdiv(nf, 1024);
        for (int l = 0; l < nlev; l++) {
            const int cnt = lstart[(size_t)l + 1] - lstart[(size_t)l];
            if (cnt == 0) continue;
            hipLaunchKernelGGL((k_kd_level<DT>), dim3(cnt, ctiles), dim3(256), 0, s, cnt, dord.p + lstart[(size_t)l], dptr.p, dent.p, F, Y.p, (i64d)ldz, nf);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
    }
    const double t_sparse = spasm_wtime();
    // ---- K = Y transposed, compacted
    const int nchunks = std::max(1, cdiv(r, KD_CHUNK));
    DevBuf<i64d> cnt, off;
    DevBuf<int> d_pcp, d_free;
    const size_t ncnt = (size_t)nf * (size_t)nchunks;
    cnt.alloc(ncnt + 1); off.alloc(ncnt + 1); d_pcp.alloc((size_t)r + 1); d_free.alloc((size_t)nf + 1);
    HIPCHK(hipMemsetAsync(cnt.p + ncnt, 0, sizeof(i64d), s));
    if (r > 0) HIPCHK(hipMemcpyAsync(d_pcp.p, pivcol_of_pos.data(), (size_t)r * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(d_free.p, h_free.data(), (size_t)nf * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL((k_kd_count<DT>), dim3(cdiv(nf, 256), nchunks), dim3(256), 0, s, r, nf, Y.p, (i64d)ldz, cnt.p, nchunks);
    HIPCHK(hipGetLastError());
    Scanner scan;
    scan.exclusive(cnt.p, off.p, ncnt + 1, s);
    std::vector<i64d> h_off(ncnt + 1);
    HIPCHK(hipMemcpyAsync(h_off.data(), off.p, (ncnt + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const i64 ktot = h_off[ncnt];
    DevBuf<int> Kj, Kx;
    Kj.alloc((size_t)ktot + 1); Kx.alloc((size_t)ktot + 1);
    hipLaunchKernelGGL((k_kd_fill<DT>), dim3(cdiv(nf, 256), nchunks), dim3(256), 0, s, r, nf, Y.p, (i64d)ldz, off.p, nchunks, d_pcp.p, d_free.p, Kj.p, Kx.p);
    HIPCHK(hipGetLastError());
    struct spasm_csr *K = spasm_csr_alloc(nf, m, std::max<i64>(ktot, 1), prime, true);
    if (!K) throw EngineError("out of host memory");
    for (int i = 0; i <= nf; i++) K->p[i] = i < nf ? h_off[(size_t)i * (size_t)nchunks] : ktot;
    if (ktot > 0) {
        HIPCHK(hipMemcpyAsync(K->j, Kj.p, (size_t)ktot * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(K->x, Kx.p, (size_t)ktot * sizeof(int), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    spasm_logf("[kernel] pivot order and free columns %.2fs; dense tail of %d rows: to the device %.2fs, reduced form %.2fs; %d sparse rows in %d levels %.2fs; K %.2fs\n",
               t_prep - t_start, rd, t_dense_rows - t_prep, t_z - t_dense_rows, t0, nlev, t_sparse - t_z, spasm_wtime() - t_sparse);
    spasm_logf("[kernel] done in %.1fs. NNZ(K) = %lld\n", spasm_wtime() - t_start, (long long)spasm_nnz(K));
    return K;
}

struct spasm_csr *do_kernel(const struct spasm_lu *fact, int first = 0, int step = 1)
{
    require_device();
    if (!fact || !fact->U || !fact->qinv) throw EngineError("spasm_kernel: incomplete factorization (U / qinv missing)");
    const struct spasm_csr *U = fact->U;
    check_input(U, "spasm_kernel");
    if (first < 0 || step < 1) throw EngineError("spasm_amd_kernel_strided: bad (first, step)");
    // a U too large for the device as one round goes through its closure (SPASM_AMD_KERNEL_REDUCE_NNZ, tests: the size from which)
    i64 reduce_at = (i64)1 << 31;
    if (const char *e = getenv("SPASM_AMD_KERNEL_REDUCE_NNZ")) reduce_at = atoll(e);
    // all free columns at once through a dense right-hand side (kernel_dense.hpp): primes below 2^16; SPASM_AMD_KERNEL_DENSE_RHS = 1
    // always (tests), 0 never; SPASM_AMD_KERNEL_DENSE_TAIL (tests): rows of the dense tail instead of the choice by density
    {
        const ZpField F = zp_field_make(U->field->p);
        const char *e = getenv("SPASM_AMD_KERNEL_DENSE_RHS");
        // (by default from 2^26 entries of U on: config 5 at 1/5, 2e9 entries, takes 1.3 s this way and 2.3 s with one sparse solve
        // per free column; below that the difference is noise and the sparse solves also serve the large primes)
        const bool want = e ? atoi(e) != 0 : spasm_nnz(U) >= std::min<i64>(reduce_at, (i64)1 << 26);
        if (want && F.small && dense_elem_bytes(F, 1) <= 2) {
            i64 tail = -1;
            if (const char *t = getenv("SPASM_AMD_KERNEL_DENSE_TAIL")) tail = atoll(t);
            struct spasm_csr *K = dense_elem_bytes(F, 1) == 1 ? kernel_dense_rhs<signed char>(U, fact->qinv, first, step, tail)
                                                              : kernel_dense_rhs<short>(U, fact->qinv, first, step, tail);
            if (K) return K;
            spasm_logf("[kernel] the dense right-hand side does not fit the device: one sparse solve per free column\n");
        }
    }
    if (spasm_nnz(U) >= reduce_at) {
        const double t0 = spasm_wtime();
        KernelReduced red;
        kernel_closure(U, fact->qinv, first, step, red);
        spasm_logf("[kernel] closure found in %.2fs\n", spasm_wtime() - t0);
        return do_kernel_core(red.U2, red.qinv2.data(), first, step);
    }
    return do_kernel_core(U, fact->qinv, first, step);
}

// qinv[j] = row of U with its pivot on column j, -1: free column, -2: a pivot column of a row that is not part of U (kernel_closure)
struct spasm_csr *do_kernel_core(const struct spasm_csr *U, const int *qinv, int first, int step)
{
    const int r = U->n, m = U->m;
    const i64 prime = U->field->p;
    const double t0 = spasm_wtime();
    spasm_logf("[kernel] start. U is %d x %d (%lld nnz). Transposing U\n", r, m, (long long)spasm_nnz(U));

    // pivot column of each row, unit pivots checked (reference src/SpaSM.jl:712), topological numbering
    std::vector<int> pc;
    const std::vector<int> perm = pivot_topological_order(U, qinv, "spasm_kernel", pc);
    const double t_order = spasm_wtime();
    std::vector<int> idx_of((size_t)std::max(r, 1));
    for (int t = 0; t < r; t++) idx_of[(size_t)perm[(size_t)t]] = t;

    // Transposed triangular system (what libspasm solves per free column, README.md:39 "Transposing U"):
    // unknowns y_a = k[pivcol(a)], equations y_a + sum_{b != a} U[a][pivcol(b)] y_b = U[a][j].  Numbering the
    // pivots in REVERSE topological order (ridx) makes the system strictly upper triangular for the solve kernels:
    // "pivot row" ridx(b) = column pivcol(b) of U, "right-hand side" of free column j = column j of U.
    std::vector<int> h_lab((size_t)std::max(r, 1)), h_rowsrc((size_t)std::max(r, 1)), h_free;
    for (int a = 0; a < r; a++) h_lab[(size_t)a] = r - 1 - idx_of[(size_t)a];
    for (int t = 0; t < r; t++) h_rowsrc[(size_t)(r - 1 - t)] = pc[(size_t)perm[(size_t)t]]; // column of pivot ridx
    {
        int f = 0;
        for (int j = 0; j < m; j++)
            if (qinv[j] < 0 && qinv[j] != -2) {
                if (f >= first && (f - first) % step == 0) h_free.push_back(j);
                f++;
            }
    }
    const int nfree = (int)h_free.size();

    hipStream_t s = nullptr;
    DevMat PM;
    upload_csr(U, 0, r, PM, s);
    HIPCHK(hipStreamSynchronize(s));
    const double t_upload = spasm_wtime();
    std::unique_ptr<Round> R(new Round());
    R->F = zp_field_make(prime);
    R->stream = s;
    DevBuf<int> lab, rowsrc, freecol;
    lab.alloc((size_t)r + 1);
    rowsrc.alloc((size_t)r + 1);
    freecol.alloc((size_t)nfree + 1);
    if (r > 0) {
        HIPCHK(hipMemcpyAsync(lab.p, h_lab.data(), (size_t)r * sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(rowsrc.p, h_rowsrc.data(), (size_t)r * sizeof(int), hipMemcpyHostToDevice, s));
    }
    if (nfree > 0) HIPCHK(hipMemcpyAsync(freecol.p, h_free.data(), (size_t)nfree * sizeof(int), hipMemcpyHostToDevice, s));
    // Ut: row j = column j of U with entries labelled ridx(a)
    TransposeOut Ut;
    device_transpose(PM, r, nullptr, lab.p, 0, R->scan, s, Ut);
    HIPCHK(hipStreamSynchronize(s));
    const double t_transpose = spasm_wtime();
    DevMat T;
    T.n = m;
    T.m = r;
    T.start.alloc((size_t)m + 1);
    T.len.alloc((size_t)m + 1);
    T.lead.alloc(1);
    T.orig.alloc((size_t)m + 1);
    T.ent = std::move(Ut.Tent);
    if (m > 0) {
        hipLaunchKernelGGL(k_rows_from_ptr, dim3(cdiv(m, 256)), dim3(256), 0, s, m, Ut.Tp.p, T.start.p, T.len.p, T.orig.p);
        HIPCHK(hipGetLastError());
    }
    // the solve structures over the ridx space: every "column" 0..r-1 carries a pivot
    R->m = r;
    R->npiv = r;
    R->qinv_r.alloc((size_t)r + 1);
    R->pivcol.alloc((size_t)r + 1);
    if (r > 0) {
        hipLaunchKernelGGL(k_iota, dim3(cdiv(r, 256)), dim3(256), 0, s, r, R->qinv_r.p);
        hipLaunchKernelGGL(k_iota, dim3(cdiv(r, 256)), dim3(256), 0, s, r, R->pivcol.p);
        HIPCHK(hipGetLastError());
    }
    R->want_idx = true; // the kernel vectors are assembled from (pivot index, multiplier)
    HIPCHK(hipStreamSynchronize(s));
    const double t_rows = spasm_wtime();
    R->build_U(T, rowsrc.p);
    HIPCHK(hipStreamSynchronize(s));
    const double t_buildu = spasm_wtime();
    R->prepare_uinv(nfree);
    // (the record pool: a guess that grows when a solve overflows it.  4 nnz(U) was the guess -- 160 GB of pool for the U of config 5
    // at 1/5, 3.4 of the step's 4.5 s spent allocating it for kernel vectors of one or two entries)
    R->solve_phase(T, freecol.p, nullptr, nfree, std::min<i64>(4 * spasm_nnz(U), std::max<i64>((i64)1 << 24, 1024 * (i64)nfree)));
    HIPCHK(hipStreamSynchronize(s));
    const double t_solve = spasm_wtime();
    // assemble K: row f = {(free column, -1)} U {(column of pivot ridx, y_ridx)}
    DevBuf<i64d> klen, kstart;
    klen.alloc((size_t)nfree + 1);
    kstart.alloc((size_t)nfree + 1);
    hipLaunchKernelGGL(k_kcount, dim3(cdiv(((i64)nfree + 1) * 64, 256)), dim3(256), 0, s, nfree, R->Lstart.p, R->Llen.p, R->Lpool.p, klen.p);
    HIPCHK(hipGetLastError());
    R->scan.exclusive(klen.p, kstart.p, (size_t)nfree + 1, s);
    i64d ktot = 0;
    HIPCHK(hipMemcpyAsync(&ktot, kstart.p + nfree, sizeof ktot, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    DevBuf<int2> Kent;
    Kent.alloc((size_t)ktot + 1);
    if (nfree > 0) {
        hipLaunchKernelGGL(k_kfill, dim3(cdiv((i64)nfree * 64, 256)), dim3(256), 0, s, nfree, freecol.p, rowsrc.p, R->Lstart.p, R->Llen.p, R->Lpool.p, R->Lidx.p,
                           kstart.p, Kent.p);
        HIPCHK(hipGetLastError());
    }
    struct spasm_csr *K = spasm_csr_alloc(nfree, m, ktot, prime, true);
    if (!K) throw EngineError("out of host memory");
    {
        std::vector<int2> ent((size_t)ktot);
        HIPCHK(hipMemcpyAsync(K->p, kstart.p, ((size_t)nfree + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
        if (ktot > 0) HIPCHK(hipMemcpyAsync(ent.data(), Kent.p, (size_t)ktot * sizeof(int2), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (i64 q = 0; q < ktot; q++) { K->j[q] = ent[(size_t)q].x; K->x[q] = ent[(size_t)q].y; }
    }
    spasm_logf("[kernel] pivot order %.2fs, U to the device %.2fs, transpose %.2fs, rows of the transposed system %.2fs, its U %.2fs, solves %.2fs, K %.2fs\n", t_order - t0,
               t_upload - t_order, t_transpose - t_upload, t_rows - t_transpose, t_buildu - t_rows, t_solve - t_buildu, spasm_wtime() - t_solve);
    spasm_logf("[kernel] done in %.1fs. NNZ(K) = %lld\n", spasm_wtime() - t0, (long long)spasm_nnz(K));
    return K;
}

} // namespace

// ------------------------------------------------------------------------------------------------
// device-resident plan for ONE Schur round (bench / multi-GPU shard)
// ------------------------------------------------------------------------------------------------
struct spasm_amd_schur_plan {
    DevMat A;
    DevMat PM;          // sharded runs: the imported pivot rows
    DevBuf<int> rowsrc; // sharded runs: pivot index -> row of PM
    Round R;
    int lo = 0, hi = 0, stride = 1;  // the plan's rows: lo, lo + stride, ... < hi (global row ids)
    i64 nnz_in = 0;
    i64 prime = 0;
    bool ran = false;
    // the reference's trip counters of this round, counted once on the multiplier lists when the plan is made (the runs go
    // along the rows of W, where no list exists to count)
    u64d exact_applications = 0, exact_nnz_reduced = 0, exact_segments = 0;
    // the fused step of this plan: capacity of S it is given, and how many rows it leaves to the general path (every run leaves the
    // same rows: the dry run found out)
    i64 fused_scap = 0;
    int fused_nrej = 0;
};

namespace {

} // namespace

struct spasm_amd_shard {
    spasm_amd_schur_plan *plan = nullptr; // owns the shard's rows and the round; handed to the caller by shard_import
    int n_total = 0;
    int npiv = 0, nown = 0;
    i64 nnz_own = 0;
    DevBuf<int> oflag, oscan;
    DevBuf<int2> ohdr;
    DevBuf<i64d> olen, ooff;
    bool imported = false;
};

namespace {

// what a plan's round needs beside U: W when the round can go along it (rebuilt by every run: it is part of the Schur step),
// else Uinv for the multiplier lists
void plan_prepare(spasm_amd_schur_plan *P)
{
    Round &R = P->R;
    R.want_w_row_stats = true;
    R.prepare_w((i64)1 << 62, P->nnz_in);
    if (R.use_w) { R.use_uinv = false; R.ms_uinv = 0; }
    else R.prepare_uinv();
}

// dry run of the solve on the multiplier lists: sizes the record pool and the Schur slots (the bounds along W are no larger)
// and counts the reference's trip counters once
void plan_dry_run(spasm_amd_schur_plan *P, i64 pool_guess)
{
    Round &R = P->R;
    const bool fl = R.force_lists;
    R.force_lists = true;
    const i64 tot = R.solve_phase(P->A, R.np_rows.p, nullptr, R.nnp, pool_guess);
    const RoundCounters c = R.read_counters();
    P->exact_applications = c.applications;
    P->exact_nnz_reduced = c.nnz_reduced;
    P->exact_segments = c.segments;
    R.force_lists = fl;
    i64 tot2 = 0;
    // along W a row's stream is the sum of the runs of its entries on pivot columns: two runs that share pivot rows count them
    // twice, so these bounds can exceed the lists' -- size for both
    if (R.use_w && !fl) {
        tot2 = R.solve_phase(P->A, R.np_rows.p, nullptr, R.nnp, pool_guess);
        const RoundCounters c2 = R.read_counters(); // the plan runs exactly this again: the same rows take the same classes
        R.quiet_fallbacks = c2.combine_overflow == 0 && c2.solve_overflow == 0 && c2.solve_failed == 0;
        R.quiet_rejects = c2.wplan_reject == 0;
        R.quiet_known = true;
    }
    R.S.ent.ensure((size_t)std::max(tot, tot2) + 1);
    if (R.fused_ok()) {
        // the fused step once, with room to spare: which rows it takes, what the general path needs for the others (its pools are
        // sized by running it for them), whether every row the fused kernel takes up also comes out of it
        P->fused_scap = R.fused_capacity(std::max(tot, tot2) + 16 * (i64)R.nnp);
        R.S.ent.ensure((size_t)P->fused_scap + 1);
        if (!R.run_fused(P->A, R.np_rows.p, R.nnp, P->fused_scap, false)) P->fused_scap = 0;
        else {
            R.fetch_step();
            R.fz_nleft = R.fz_nrej;
            R.fused_leftovers(P->A, R.np_rows.p);
            // (S must hold them too, every time)
            const i64 need = (R.fz_ngeneral > 0 ? R.fb_base + R.fb_tot + 16 : P->fused_scap) + 2 * R.fb_left_tot + 1024;
            if (R.fz_nleft > 0 && (i64)R.S.ent.n < need) R.S.ent.ensure((size_t)need);
        }
        if (P->fused_scap == 0) {
            // (not a round for the fused kernel: the general path for all rows, its pools sized again -- the attempt left them sized for few)
            R.last_fused = false;
            (void)R.solve_phase(P->A, R.np_rows.p, nullptr, R.nnp, pool_guess);
        }
    }
}

spasm_amd_shard *shard_create(const struct spasm_csr *A, int lo, int hi, int stride = 1)
{
    require_device();
    check_input(A, "spasm_amd_shard_create");
    if (lo < 0 || hi > A->n || lo > hi || stride < 1) throw EngineError("spasm_amd_shard_create: bad row range");
    std::unique_ptr<spasm_amd_shard> S(new spasm_amd_shard());
    std::unique_ptr<spasm_amd_schur_plan> P(new spasm_amd_schur_plan());
    P->lo = lo;
    P->hi = hi;
    P->stride = stride;
    P->prime = A->field->p;
    P->nnz_in = 0;
    for (int g = lo; g < hi; g += stride) P->nnz_in += A->p[g + 1] - A->p[g];
    hipStream_t s = nullptr;
    upload_csr_strided(A, lo, hi, stride, P->A, s); // only this shard's rows; orig = global row ids
    P->R.F = zp_field_make(P->prime);
    P->R.stream = s;
    S->n_total = A->n;
    S->plan = P.release();
    return S.release();
}

void shard_elect(spasm_amd_shard *S, int64_t *keys_dev)
{
    Round &R = S->plan->R;
    R.elect_local(S->plan->A, S->plan->lo, S->plan->stride);
    HIPCHK(hipMemcpyAsync(keys_dev, R.best.p, (size_t)S->plan->A.m * sizeof(u64d), hipMemcpyDeviceToDevice, R.stream));
    HIPCHK(hipStreamSynchronize(R.stream));
}

// the reduced election keys -> the leftmost pivots, numbered (the same on every shard); returns how many
int shard_assign(spasm_amd_shard *S, const int64_t *keys_dev)
{
    spasm_amd_schur_plan *P = S->plan;
    Round &R = P->R;
    const int m = P->A.m;
    R.m = m;
    R.best.ensure((size_t)m + 1);
    HIPCHK(hipMemcpyAsync(R.best.p, keys_dev, (size_t)m * sizeof(u64d), hipMemcpyDeviceToDevice, R.stream));
    R.assign_pivots();
    R.n_leftmost = R.npiv;
    R.n_open = 0;
    return R.npiv;
}

// "FL on columns" over row shards (Round::open_*): one step of the search on this shard's rows.  `in_dev`: the array the step before
// left, REDUCED over the shards by the caller (copied in first); `out_dev` receives the array this step leaves, m elements, for the
// caller to reduce: BEGIN -> closed (int32, MAX); HIST (in: closed) -> colcnt (int32, SUM); PROPOSE (in: colcnt) -> best2 (int64,
// MIN); ACCEPT (in: best2) -> newflag (int32, MAX); RECORD (in: newflag) -> closed (int32, MAX), returns the pivots the pass
// accepted (0: the search is over, nothing written); FINISH renumbers and returns the pivots the search added.
enum { OPEN_BEGIN = 0, OPEN_HIST = 1, OPEN_PROPOSE = 2, OPEN_ACCEPT = 3, OPEN_RECORD = 4, OPEN_FINISH = 5 };
int shard_open_step(spasm_amd_shard *S, int step, int pass, const void *in_dev, void *out_dev)
{
    spasm_amd_schur_plan *P = S->plan;
    Round &R = P->R;
    hipStream_t s = R.stream;
    const size_t m = (size_t)P->A.m;
    int ret = 0;
    auto take = [&](void *dst, size_t elem) { if (in_dev) HIPCHK(hipMemcpyAsync(dst, in_dev, m * elem, hipMemcpyDeviceToDevice, s)); };
    auto give = [&](const void *src, size_t elem) { if (out_dev) HIPCHK(hipMemcpyAsync(out_dev, src, m * elem, hipMemcpyDeviceToDevice, s)); };
    switch (step) {
    case OPEN_BEGIN:
        R.open_begin(P->A, P->lo, P->stride);
        give(R.closed.p, sizeof(int));
        break;
    case OPEN_HIST:
        take(R.closed.p, sizeof(int));
        R.open_hist(P->A);
        give(R.colcnt.p, sizeof(int));
        break;
    case OPEN_PROPOSE:
        take(R.colcnt.p, sizeof(int));
        R.open_propose(P->A, P->lo, P->stride);
        give(R.best2.p, sizeof(u64d));
        break;
    case OPEN_ACCEPT:
        take(R.best2.p, sizeof(u64d));
        R.open_accept(P->A, P->lo, P->stride);
        give(R.newflag.p, sizeof(int));
        break;
    case OPEN_RECORD:
        take(R.newflag.p, sizeof(int));
        ret = R.open_record(P->A, pass, P->lo, P->stride);
        if (ret > 0) give(R.closed.p, sizeof(int));
        break;
    case OPEN_FINISH:
        ret = R.open_finish();
        break;
    default:
        throw EngineError("spasm_amd_shard_open_step: unknown step");
    }
    HIPCHK(hipStreamSynchronize(s));
    return ret;
}

// the pivots are final: this shard's non-pivot rows, the pivot rows it owns (what shard_export sends); returns npiv
int shard_finish_keys(spasm_amd_shard *S, int *n_owned, i64 *nnz_owned);

int shard_set_keys(spasm_amd_shard *S, const int64_t *keys_dev, int *n_owned, i64 *nnz_owned)
{
    shard_assign(S, keys_dev);
    return shard_finish_keys(S, n_owned, nnz_owned);
}

int shard_finish_keys(spasm_amd_shard *S, int *n_owned, i64 *nnz_owned)
{
    spasm_amd_schur_plan *P = S->plan;
    Round &R = P->R;
    hipStream_t s = R.stream;
    R.mark_local(P->A, P->lo, 0, INT_MAX, P->stride, 1); // local non-pivot rows; pivrow holds global ids
    S->npiv = R.npiv;
    // owned pivot rows, in ascending pivot index
    const int np = R.npiv;
    S->oflag.alloc((size_t)np + 1);
    S->oscan.alloc((size_t)np + 1);
    hipLaunchKernelGGL(k_owned_flags, dim3(cdiv((i64)np + 1, 256)), dim3(256), 0, s, np, P->lo, P->stride, P->A.n, R.pivrow.p, S->oflag.p);
    HIPCHK(hipGetLastError());
    R.scan.exclusive(S->oflag.p, S->oscan.p, (size_t)np + 1, s);
    HIPCHK(hipMemcpyAsync(&S->nown, S->oscan.p + np, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    S->ohdr.alloc((size_t)S->nown + 1);
    S->olen.alloc((size_t)S->nown + 1);
    S->ooff.alloc((size_t)S->nown + 1);
    HIPCHK(hipMemsetAsync(S->olen.p + S->nown, 0, sizeof(i64d), s));
    if (np > 0) {
        hipLaunchKernelGGL(k_export_hdr, dim3(cdiv(np, 256)), dim3(256), 0, s, np, P->lo, P->stride, S->oflag.p, S->oscan.p, R.pivrow.p, P->A.len.p, S->ohdr.p, S->olen.p);
        HIPCHK(hipGetLastError());
    }
    R.scan.exclusive(S->olen.p, S->ooff.p, (size_t)S->nown + 1, s);
    i64d tot = 0;
    HIPCHK(hipMemcpyAsync(&tot, S->ooff.p + S->nown, sizeof tot, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    S->nnz_own = tot;
    if (n_owned) *n_owned = S->nown;
    if (nnz_owned) *nnz_owned = tot;
    return np;
}

void shard_export(spasm_amd_shard *S, int *hdr_dev, int *ent_dev)
{
    spasm_amd_schur_plan *P = S->plan;
    Round &R = P->R;
    hipStream_t s = R.stream;
    if (S->nown > 0) {
        HIPCHK(hipMemcpyAsync(hdr_dev, S->ohdr.p, (size_t)S->nown * sizeof(int2), hipMemcpyDeviceToDevice, s));
        constexpr int TEAM = 16;
        hipLaunchKernelGGL((k_export_rows<TEAM>), dim3(cdiv((i64)S->nown * TEAM, 256)), dim3(256), 0, s, S->nown, P->lo, P->stride, S->ohdr.p, S->ooff.p, R.pivrow.p,
                           P->A.start.p, P->A.ent.p, (int2 *)ent_dev);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(s));
}

// what a sharded round needs after its U: W / Uinv and the dry run that sizes the pools (the second half of shard_import)
void shard_import_finish(spasm_amd_schur_plan *P)
{
    Round &R = P->R;
    plan_prepare(P);
    HIPCHK(hipEventRecord(R.ev[1], R.stream));
    plan_dry_run(P, 4 * std::max<i64>(P->nnz_in, 1 << 14));
}

// prepare = false: stop after U (the caller decides first whether the round's Schur complement goes dense; shard_import_finish otherwise)
spasm_amd_schur_plan *shard_import(spasm_amd_shard *S, int n_rows, i64 n_entries, const int *hdr_dev, const int *ent_dev, bool prepare = true)
{
    spasm_amd_schur_plan *P = S->plan;
    if (!P) throw EngineError("spasm_amd_shard_import: already imported");
    Round &R = P->R;
    hipStream_t s = R.stream;
    if (n_rows != R.npiv) throw EngineError("spasm_amd_shard_import: the parts do not add up to the elected pivots");
    DevMat &PM = P->PM;
    PM.n = n_rows;
    PM.m = P->A.m;
    PM.start.alloc((size_t)n_rows + 1);
    PM.len.alloc((size_t)n_rows + 1);
    PM.lead.alloc(1);
    PM.orig.alloc((size_t)n_rows + 1);
    PM.ent.alloc((size_t)n_entries + 1);
    P->rowsrc.alloc((size_t)n_rows + 1);
    DevBuf<i64d> l64, off;
    l64.alloc((size_t)n_rows + 1);
    off.alloc((size_t)n_rows + 1);
    hipLaunchKernelGGL(k_hdr_len64, dim3(cdiv((i64)n_rows + 1, 256)), dim3(256), 0, s, n_rows, (const int2 *)hdr_dev, l64.p);
    HIPCHK(hipGetLastError());
    R.scan.exclusive(l64.p, off.p, (size_t)n_rows + 1, s);
    i64d tot = 0;
    HIPCHK(hipMemcpyAsync(&tot, off.p + n_rows, sizeof tot, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (tot != n_entries) throw EngineError("spasm_amd_shard_import: header lengths do not match the entry count");
    if (n_rows > 0) {
        hipLaunchKernelGGL(k_import_rows, dim3(cdiv(n_rows, 256)), dim3(256), 0, s, n_rows, (const int2 *)hdr_dev, off.p, PM.start.p, PM.len.p, PM.orig.p, P->rowsrc.p);
        HIPCHK(hipGetLastError());
    }
    if (n_entries > 0) HIPCHK(hipMemcpyAsync(PM.ent.p, ent_dev, (size_t)n_entries * sizeof(int2), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipEventRecord(R.ev[0], s));
    R.build_U(PM, P->rowsrc.p);
    if (prepare) shard_import_finish(P);
    S->plan = nullptr; // ownership passes to the caller
    S->imported = true;
    return P;
}

void plan_run(spasm_amd_schur_plan *P, hipStream_t s);

// The round of a sharded plan becomes history: its Schur rows become the shard's matrix of the next round, on the device and
// under the same local numbering (local row i = original row lo + i * stride).  Consumes the plan.
spasm_amd_shard *plan_advance(spasm_amd_schur_plan *P, int *rows_out, i64 *nnz_out)
{
    if (!P) throw EngineError("spasm_amd_schur_plan_advance: null plan");
    if (!P->ran) plan_run(P, P->R.stream);
    Round &R = P->R;
    hipStream_t s = R.stream;
    R.fetch_step(); // synchronises; throws on an exhausted pool / table
    const int n = P->A.n, nnp = R.nnp;
    std::unique_ptr<spasm_amd_shard> S2(new spasm_amd_shard());
    std::unique_ptr<spasm_amd_schur_plan> P2(new spasm_amd_schur_plan());
    P2->lo = P->lo;
    P2->hi = P->hi;
    P2->stride = P->stride;
    P2->prime = P->prime;
    P2->nnz_in = (i64)R.hctr.nnz_out;
    DevMat &A2 = P2->A;
    A2.n = n;
    A2.m = P->A.m;
    A2.start.alloc((size_t)n + 1);
    A2.len.alloc((size_t)n + 1);
    A2.lead.alloc((size_t)n + 1);
    A2.orig.alloc((size_t)n + 1);
    if (n > 0) {
        hipLaunchKernelGGL(k_advance_init, dim3(cdiv(n, 256)), dim3(256), 0, s, n, P->lo, P->stride, A2.start.p, A2.len.p, A2.lead.p, A2.orig.p);
        HIPCHK(hipGetLastError());
    }
    if (nnp > 0) {
        hipLaunchKernelGGL(k_advance_rows, dim3(cdiv(nnp, 256)), dim3(256), 0, s, nnp, n, P->lo, P->stride, R.S.start.p, R.S.len.p, R.S.lead.p,
                           R.S.orig.p, A2.start.p, A2.len.p, A2.lead.p);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(s));
    A2.ent = std::move(R.S.ent);
    P2->R.F = zp_field_make(P2->prime);
    P2->R.stream = s;
    S2->n_total = P->hi;
    if (rows_out) *rows_out = R.hctr.nonempty_out;
    if (nnz_out) *nnz_out = (i64)R.hctr.nnz_out;
    S2->plan = P2.release();
    delete P;
    return S2.release();
}

// the shard's rows as a host CSR with one row per LOCAL row (empty rows included; local row i = original row lo + i * stride)
struct spasm_csr *shard_fetch(spasm_amd_shard *S)
{
    if (!S || !S->plan) throw EngineError("spasm_amd_shard_fetch: the shard has no matrix (already imported?)");
    spasm_amd_schur_plan *P = S->plan;
    Round &R = P->R;
    hipStream_t s = R.stream;
    const int n = P->A.n;
    DevBuf<i64d> len64, ostart;
    len64.alloc((size_t)n + 1);
    ostart.alloc((size_t)n + 1);
    hipLaunchKernelGGL(k_copy_len64, dim3(cdiv((i64)n + 1, 256)), dim3(256), 0, s, n, P->A.len.p, len64.p);
    HIPCHK(hipGetLastError());
    R.scan.exclusive(len64.p, ostart.p, (size_t)n + 1, s);
    std::vector<i64d> hp((size_t)n + 1);
    HIPCHK(hipMemcpyAsync(hp.data(), ostart.p, ((size_t)n + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const i64 tot = hp[(size_t)n];
    struct spasm_csr *out = spasm_csr_alloc(n, P->A.m, tot, P->prime, true);
    if (!out) throw EngineError("out of host memory for the shard's rows");
    for (int i = 0; i <= n; i++) out->p[i] = hp[(size_t)i];
    if (tot > 0) {
        DevBuf<int> oj, ox;
        oj.alloc((size_t)tot);
        ox.alloc((size_t)tot);
        constexpr int TEAM = 16;
        hipLaunchKernelGGL((k_compact_rows<TEAM>), dim3(cdiv((i64)n * TEAM, 256)), dim3(256), 0, s, n, P->A.start.p, P->A.len.p, P->A.ent.p, ostart.p, oj.p, ox.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out->j, oj.p, (size_t)tot * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(out->x, ox.p, (size_t)tot * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    return out;
}

spasm_amd_schur_plan *plan_create(const struct spasm_csr *A, int lo, int hi, int stride = 1)
{
    require_device();
    check_input(A, "spasm_amd_schur_plan_create");
    if (lo < 0 || hi > A->n || lo > hi || stride < 1) throw EngineError("spasm_amd_schur_plan_create: bad row range");
    std::unique_ptr<spasm_amd_schur_plan> P(new spasm_amd_schur_plan());
    P->lo = lo;
    P->hi = hi;
    P->stride = stride;
    P->prime = A->field->p;
    P->nnz_in = 0;
    for (int g = lo; g < hi; g += stride) P->nnz_in += A->p[g + 1] - A->p[g];
    hipStream_t s = nullptr;
    // the whole matrix is resident on every device: the election sees all rows, so every shard builds the same U
    upload_csr(A, 0, A->n, P->A, s);
    Round &R = P->R;
    R.F = zp_field_make(P->prime);
    R.stream = s;
    HIPCHK(hipEventRecord(R.ev[0], s));
    R.elect_local(P->A, 0);
    R.assign_pivots();
    R.mark_local(P->A, 0, lo, hi, 1, stride);
    R.build_U(P->A, R.pivrow.p);
    plan_prepare(P.get());
    HIPCHK(hipEventRecord(R.ev[1], s));
    plan_dry_run(P.get(), 4 * spasm_nnz(A));
    return P.release();
}

void plan_run(spasm_amd_schur_plan *P, hipStream_t s)
{
    Round &R = P->R;
    R.stream = s;
    // the whole Schur step of the round (reference spasm_schur, src/SpaSM.jl:761-762: the per-row solve is inside it): W from
    // U, the plan of every row, the scatter.  ev[4] .. ev[1] = the W build.
    HIPCHK(hipEventRecord(R.ev[4], s));
    if (R.use_w) R.build_w_levels();
    else if (R.use_uinv) R.prepare_uinv();
    HIPCHK(hipEventRecord(R.ev[1], s));
    if (R.fused_ok() && P->fused_scap > 0) {
        // plan + stream of every row in one kernel (fused.hpp); the rows it leaves go through the general path behind it
        HIPCHK(hipEventRecord(R.ev[2], s));
        (void)R.run_fused(P->A, R.np_rows.p, R.nnp, P->fused_scap, true);
        HIPCHK(hipEventRecord(R.ev[3], s));
    } else {
        R.run_solve(P->A, R.np_rows.p, nullptr, R.nnp);
        R.run_bounds(R.nnp);
        HIPCHK(hipEventRecord(R.ev[2], s));
        R.run_scatter(P->A, R.np_rows.p, R.nnp);
        HIPCHK(hipEventRecord(R.ev[3], s));
    }
    P->ran = true;
}

// the pivot rows of the plan's round as they enter U: scaled to a unit pivot, in pivot-index order (= ascending pivot column);
// pivcol_out / row_out (npiv ints each, may be NULL): pivot column and originating row (global row number) of each
struct spasm_csr *plan_fetch_U(spasm_amd_schur_plan *P, int *pivcol_out, int *row_out)
{
    Round &R = P->R;
    hipStream_t s = R.stream;
    const int np = R.npiv;
    struct spasm_csr *Uc = spasm_csr_alloc(np, R.m, R.utotal, P->prime, true);
    if (!Uc) throw EngineError("out of host memory for the round's U rows");
    std::unique_ptr<struct spasm_csr, void (*)(struct spasm_csr *)> guard(Uc, spasm_csr_free); // (a HIPCHK below may throw)
    if (np > 0) HIPCHK(hipMemcpyAsync(Uc->p, R.uoff.p, ((size_t)np + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
    else Uc->p[0] = 0;
    if (R.utotal > 0) {
        DevBuf<int> dj, dx;
        dj.alloc((size_t)R.utotal);
        dx.alloc((size_t)R.utotal);
        hipLaunchKernelGGL(k_split_ent, dim3((unsigned)std::min<i64>((R.utotal + 255) / 256, 65536)), dim3(256), 0, s, (i64d)R.utotal, R.Ufull.p, dj.p, dx.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(Uc->j, dj.p, (size_t)R.utotal * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(Uc->x, dx.p, (size_t)R.utotal * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    if (np > 0 && pivcol_out) HIPCHK(hipMemcpyAsync(pivcol_out, R.pivcol.p, (size_t)np * sizeof(int), hipMemcpyDeviceToHost, s));
    if (np > 0 && row_out) HIPCHK(hipMemcpyAsync(row_out, R.pivrow.p, (size_t)np * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return guard.release();
}

struct spasm_csr *plan_fetch(spasm_amd_schur_plan *P, int *p_out)
{
    Round &R = P->R;
    if (!P->ran) throw EngineError("spasm_amd_schur_plan_fetch: run the plan first");
    hipStream_t s = R.stream;
    const int n = R.nnp;
    R.fetch_step();
    DevBuf<i64d> len64, ostart;
    len64.alloc((size_t)n + 1);
    ostart.alloc((size_t)n + 1);
    hipLaunchKernelGGL(k_copy_len64, dim3(cdiv((i64)n + 1, 256)), dim3(256), 0, s, n, R.S.len.p, len64.p);
    HIPCHK(hipGetLastError());
    R.scan.exclusive(len64.p, ostart.p, (size_t)n + 1, s);
    i64d tot = 0;
    HIPCHK(hipMemcpyAsync(&tot, ostart.p + n, sizeof tot, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    DevBuf<int> oj, ox;
    oj.alloc((size_t)tot + 1);
    ox.alloc((size_t)tot + 1);
    if (n > 0) {
        constexpr int TEAM = 16;
        hipLaunchKernelGGL((k_compact_rows<TEAM>), dim3(cdiv((i64)n * TEAM, 256)), dim3(256), 0, s, n, R.S.start.p, R.S.len.p, R.S.ent.p, ostart.p, oj.p, ox.p);
        HIPCHK(hipGetLastError());
    }
    struct spasm_csr *S = spasm_csr_alloc(n, R.m, tot, P->prime, true);
    if (!S) throw EngineError("out of host memory");
    HIPCHK(hipMemcpyAsync(S->p, ostart.p, ((size_t)n + 1) * sizeof(i64d), hipMemcpyDeviceToHost, s));
    if (tot > 0) {
        HIPCHK(hipMemcpyAsync(S->j, oj.p, (size_t)tot * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(S->x, ox.p, (size_t)tot * sizeof(int), hipMemcpyDeviceToHost, s));
    }
    if (p_out && n > 0) HIPCHK(hipMemcpyAsync(p_out, R.S.orig.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return S;
}

// ------------------------------------------------------------------------------------------------
// The whole echelonization, row-sharded over the devices of ONE process (spasm_amd_echelonize_multi): what spasm.jl_amd/sharded.py
// does between processes with torch.distributed, here with the shards of one host thread and peer copies over xGMI in place of the
// collectives -- so that a Julia host reaches all GPUs of a node through one ccall.  Shard s keeps rows s, s + nshards, ... on
// device s % (devices): with fewer devices than shards several shards share a device (how the protocol is tested on one GPU).
// Per round: every shard proposes its election keys, device 0 takes the per-column minimum (all-reduce(MIN)), every shard numbers
// the pivots and exports the pivot rows it owns, all shards receive the concatenation (all-gather) and build the same U, the Schur
// rows stay where they are.  Leftmost-entry pivots only in the sharded rounds and in the finish, so the result does not depend on
// the number of shards.  When little is left, or the remainder is dense enough for the dense finish, the rows are gathered and
// the single-device engine finishes them on device 0.
// ------------------------------------------------------------------------------------------------
struct DeviceGuard {
    int prev = 0;
    DeviceGuard() { (void)hipGetDevice(&prev); }
    ~DeviceGuard() { (void)hipSetDevice(prev); }
};

struct spasm_lu *do_echelonize_multi(const struct spasm_csr *A, struct echelonize_opts *opts, int nshards)
{
    struct echelonize_opts dflt;
    if (!opts) { spasm_echelonize_init_opts(&dflt); opts = &dflt; }
    require_device();
    check_input(A, "spasm_amd_echelonize_multi");
    if (nshards < 1) throw EngineError("spasm_amd_echelonize_multi: at least one shard");
    if (opts->L) throw EngineError("spasm_amd_echelonize_multi: the L factor is only kept by the single-device rounds");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    const int n = A->n, m = A->m;
    const i64 prime = A->field->p;
    const double t0 = spasm_wtime();
    spasm_logf("[echelonize] Start on %d x %d matrix with %lld nnz, %d row shards on %d device(s)\n", n, m, (long long)spasm_nnz(A), nshards, std::min(ndev, nshards));
    DeviceGuard guard;
    auto dev_of = [&](int sh) { return sh % ndev; };
    // the exchange copies device to device: directly over xGMI where the devices can see each other
    for (int a = 0; a < std::min(ndev, nshards); a++) {
        if (hipSetDevice(a) != hipSuccess) continue;
        for (int b = 0; b < std::min(ndev, nshards); b++) {
            int can = 0;
            if (a != b && hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(b, 0);
        }
    }
    (void)hipGetLastError(); // ("already enabled" is not an error of ours)
    struct ShardState {
        spasm_amd_shard *sh = nullptr;
        spasm_amd_schur_plan *plan = nullptr;
        int rows = 0;
        i64 nnz = 0;
        int n_own = 0;
        i64 nnz_own = 0;
        DevBuf<u64d> keys;
        DevBuf<int> hdr, ent, hdr_all, ent_all;
        DevBuf<int> oc_i;   // "FL on columns": the int32 array of the current step
        DevBuf<u64d> oc_q;  // .. and the proposals
    };
    std::vector<std::unique_ptr<ShardState>> st((size_t)nshards);
    auto cleanup = [&]() {
        for (size_t k = 0; k < st.size(); k++) {
            ShardState *q = st[k].get();
            if (!q) continue;
            (void)hipSetDevice(dev_of((int)k));
            if (q->plan) delete q->plan;
            if (q->sh) { delete q->sh->plan; delete q->sh; }
            q->plan = nullptr;
            q->sh = nullptr;
            st[k].reset(); // (its device buffers are released on their device)
        }
    };
    HostU U;
    U.p.push_back(0);
    try {
        for (int k = 0; k < nshards; k++) {
            HIPCHK(hipSetDevice(dev_of(k)));
            st[(size_t)k].reset(new ShardState());
            ShardState &q = *st[(size_t)k];
            q.sh = shard_create(A, k, n, nshards);
            for (int g = k; g < n; g += nshards) { const i64 l = A->p[g + 1] - A->p[g]; q.rows += l > 0; q.nnz += l; }
        }
        DevBuf<u64d> stage; // on device 0: another shard's keys, on their way into the minimum
        int round = 0;
        bool extra_round_done = false;
        i64 last_nnz = -1;
        i64 finish_nnz = (i64)1 << 22;
        if (const char *e = getenv("SPASM_AMD_MULTI_FINISH_NNZ")) finish_nnz = std::max<i64>(atoll(e), 0); // tests: small remainders sharded too
        g_multi_finish = 0;
        // ---- the dense finish over the shards (dense_multi.hpp): primes below 2^16 (the int8 path), as many shards as the election
        // workgroup holds candidates for.  SPASM_AMD_MULTI_GATHER=1 (A/B, tests): the old hand-off, everything to device 0.
        const ZpField F0 = zp_field_make(prime);
        const int elem = F0.p <= 255 ? 1 : 2;
        const char *gather_env = getenv("SPASM_AMD_MULTI_GATHER");
        const bool ddf_ok = opts->enable_dense && F0.small && nshards * DP_W <= 147456 / (DP_W * elem) && nshards <= DM_MAXSHARDS && !(gather_env && atoi(gather_env));
        // .. and remainders worth it: below 16 GiB of dense matrix one device finishes faster than the shards exchange panels
        // (a 44k x 44k remainder over two ranks: 25 s of collectives against 1 s).  SPASM_AMD_MULTI_DENSE_MIN_BYTES (tests): 0.
        double dense_min_bytes = 17179869184.0;
        if (const char *e = getenv("SPASM_AMD_MULTI_DENSE_MIN_BYTES")) dense_min_bytes = atof(e);
        std::vector<int> devs((size_t)nshards);
        for (int k = 0; k < nshards; k++) devs[(size_t)k] = dev_of(k);
        // does a dense matrix of `rows` local rows and `cols` columns fit a shard's device beside what the elimination needs?
        // (no cap by shape here: the shards finish together BECAUSE one device does not hold the remainder -- config 3 / 4 leave
        // 95k x 760k shorts = 144 GB per device of an 8-GPU node; 60 % of the free memory, beside the digit planes and the dense W)
        auto dense_fits = [&](i64 rows, i64 cols) {
            const double bytes = ((double)rows + 2048.0) * (double)((cols + 63) / 64 * 64) * (double)elem;
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
            return bytes <= 0.6 * (double)fr;
        };
        auto run_typed = [&](int C, const std::vector<const int *> &clists, auto &&build) {
            if (elem == 1) return dense_multi_run<signed char>(nshards, devs, F0, C, clists, build, U);
            return dense_multi_run<short>(nshards, devs, F0, C, clists, build, U);
        };
        // The round whose pivots have just been exchanged (every shard holds U): estimate the density of its Schur complement on 64
        // columns (spasm_schur_estimate_density, as the single-device engine does for rounds without W or Uinv); dense -> the Schur
        // complement of every shard's rows goes straight into its dense matrix and the shards eliminate them together.
        auto dense_round = [&]() -> bool {
            const int free_now = m - (int)U.pivcol.size();
            if (free_now <= 0) return false;
            std::vector<std::unique_ptr<DenseW>> dws((size_t)nshards);
            std::vector<int *> flags((size_t)nshards);
            i64 nnp_tot = 0, nnp_max = 0;
            for (int k = 0; k < nshards; k++) {
                HIPCHK(hipSetDevice(dev_of(k)));
                spasm_amd_schur_plan *P = st[(size_t)k]->plan;
                dws[(size_t)k].reset(new DenseW(P->R, P->A, P->R.stream));
                dws[(size_t)k]->flag_columns();
                HIPCHK(hipStreamSynchronize(P->R.stream));
                flags[(size_t)k] = dws[(size_t)k]->cflag.p;
                nnp_tot += P->R.nnp;
                nnp_max = std::max<i64>(nnp_max, P->R.nnp);
            }
            if (nnp_tot <= 64) return false;
            or_flags_across(devs, flags, m);
            int C = 0;
            for (int k = 0; k < nshards; k++) {
                HIPCHK(hipSetDevice(dev_of(k)));
                C = dws[(size_t)k]->finish_columns();
            }
            if (C == 0 || !dense_fits(nnp_max, C)) return false;
            HIPCHK(hipSetDevice(dev_of(0)));
            Round &R0 = st[0]->plan->R;
            const double est = dws[0]->estimate_density(R0.np_rows.p, R0.nnp, free_now);
            spasm_logf("Schur complement is %lld x %d, estimated density : %.2f (64 columns sampled on shard 0, %d levels)\n", (long long)nnp_tot, free_now, est, dws[0]->depth);
            if (!(est > opts->sparsity_threshold)) return false;
            spasm_logf("[echelonize] finishing; density = %.3f (estimated); aspect ratio = %.1f; Schur complement straight to dense on %d shards\n", est,
                       (double)nnp_tot / (double)free_now, nshards);
            std::vector<const int *> clists((size_t)nshards);
            for (int k = 0; k < nshards; k++) clists[(size_t)k] = dws[(size_t)k]->clist.p;
            run_typed(C, clists, [&](int k, int extra, auto &D, DevBuf<int> &row_orig, int &Rk) {
                spasm_amd_schur_plan *P = st[(size_t)k]->plan;
                Rk = P->R.nnp;
                schur_dense_build(P->R, P->A, Rk, *dws[(size_t)k], extra, D, row_orig, P->R.stream);
            });
            return true;
        };
        // the remaining rows are dense already: every shard's live rows as a dense matrix, eliminated together
        auto dense_now = [&]() -> bool {
            std::vector<std::unique_ptr<DenseFill>> fl((size_t)nshards);
            std::vector<int *> flags((size_t)nshards);
            i64 rmax = 0;
            for (int k = 0; k < nshards; k++) {
                HIPCHK(hipSetDevice(dev_of(k)));
                spasm_amd_schur_plan *P = st[(size_t)k]->sh->plan;
                fl[(size_t)k].reset(new DenseFill(P->A, P->R.stream));
                fl[(size_t)k]->flag_columns();
                flags[(size_t)k] = fl[(size_t)k]->cflag.p;
                rmax = std::max<i64>(rmax, fl[(size_t)k]->R);
            }
            or_flags_across(devs, flags, m);
            int C = 0;
            for (int k = 0; k < nshards; k++) {
                HIPCHK(hipSetDevice(dev_of(k)));
                C = fl[(size_t)k]->finish_columns();
            }
            if (C == 0 || !dense_fits(rmax, C)) return false;
            std::vector<const int *> clists((size_t)nshards);
            for (int k = 0; k < nshards; k++) clists[(size_t)k] = fl[(size_t)k]->clist.p;
            run_typed(C, clists, [&](int k, int extra, auto &D, DevBuf<int> &row_orig, int &Rk) {
                Rk = fl[(size_t)k]->R;
                fl[(size_t)k]->build(extra, D, row_orig);
            });
            return true;
        };
        for (;;) {
            i64 rows_left = 0, nnz_left = 0;
            for (auto &q : st) { rows_left += q->rows; nnz_left += q->nnz; }
            if (nnz_left == 0) break;
            const i64 free_cols = (i64)m - (i64)U.pivcol.size();
            const double cells = (double)rows_left * (double)std::max<i64>(free_cols, 1);
            bool dense_enough = opts->enable_dense && (double)nnz_left > opts->sparsity_threshold * cells;
            // (one round ahead as well: when the fill keeps growing at the rate of the last round the next Schur complement would
            // be dense -- it is then never built sparse; the single-device density estimate plays this role there)
            // (with the distributed finish the estimate after the election decides instead: dense_round)
            const bool ddf_now = ddf_ok && cells * (double)elem >= dense_min_bytes;
            if (!ddf_now && last_nnz > 0 && nnz_left > last_nnz)
                dense_enough = dense_enough || (opts->enable_dense && (double)nnz_left * ((double)nnz_left / (double)last_nnz) > opts->sparsity_threshold * cells);
            if (ddf_now && dense_enough && nnz_left > finish_nnz) {
                spasm_logf("[echelonize] finishing; density = %.3f; aspect ratio = %.1f; dense finish on %d shards\n", (double)nnz_left / cells,
                           free_cols > 0 ? (double)rows_left / (double)free_cols : 0.0, nshards);
                if (dense_now()) { g_multi_finish = 2; break; }
            }
            // max_round sparse rounds are over and the remainder is still sparse: the single-device engine would elect once more and
            // estimate the density of that Schur complement before choosing its finish -- so do the shards, once
            const bool spent = round >= opts->max_round;
            const bool one_more = ddf_now && spent && !extra_round_done && !dense_enough && nnz_left > finish_nnz;
            if (one_more) extra_round_done = true;
            if (!one_more && (nnz_left <= finish_nnz || spent || dense_enough)) {
                // ---- hand-off: the remaining rows, under their original numbers, to the single-device engine on device 0
                std::vector<struct spasm_csr *> parts((size_t)nshards, nullptr);
                struct spasm_csr *rest = nullptr;
                struct spasm_lu *fact = nullptr;
                try {
                    i64 tot = 0;
                    for (int k = 0; k < nshards; k++) {
                        HIPCHK(hipSetDevice(dev_of(k)));
                        parts[(size_t)k] = shard_fetch(st[(size_t)k]->sh); // one row per local row: local row i = original row k + i * nshards
                        tot += parts[(size_t)k]->p[parts[(size_t)k]->n];
                    }
                    rest = spasm_csr_alloc(n, m, tot, prime, true);
                    if (!rest) throw EngineError("out of host memory for the remaining rows");
                    i64 w = 0;
                    for (int g = 0; g < n; g++) {
                        const struct spasm_csr *P = parts[(size_t)(g % nshards)];
                        const int i = g / nshards;
                        rest->p[g] = w;
                        const i64 lo = P->p[i], len = P->p[i + 1] - lo;
                        if (len > 0) {
                            memcpy(rest->j + w, P->j + lo, sizeof(int) * (size_t)len);
                            memcpy(rest->x + w, P->x + lo, sizeof(int) * (size_t)len);
                        }
                        w += len;
                    }
                    rest->p[n] = w;
                    for (auto &pp : parts) { spasm_csr_free(pp); pp = nullptr; }
                    HIPCHK(hipSetDevice(dev_of(0)));
                    struct echelonize_opts fo = *opts;
                    fo.enable_greedy_pivot_search = false;
                    fact = do_echelonize(rest, &fo);
                    const int fr = fact->r;
                    std::vector<int> colof((size_t)std::max(fr, 1), -1);
                    for (int j = 0; j < m; j++) if (fact->qinv[j] >= 0) colof[(size_t)fact->qinv[j]] = j;
                    const i64 base = U.p.back();
                    for (int k = 0; k < fr; k++) {
                        U.p.push_back(base + fact->U->p[k + 1]);
                        U.pivcol.push_back(colof[(size_t)k]);
                        U.orig.push_back(fact->p[k]);
                    }
                    const i64 uz = fact->U->p[fr];
                    if (uz > 0) {
                        memcpy(U.j.grow((size_t)uz), fact->U->j, sizeof(int) * (size_t)uz);
                        memcpy(U.x.grow((size_t)uz), fact->U->x, sizeof(int) * (size_t)uz);
                    }
                    spasm_lu_free(fact);
                    spasm_csr_free(rest);
                } catch (...) {
                    for (auto pp : parts) if (pp) spasm_csr_free(pp);
                    if (rest) spasm_csr_free(rest);
                    if (fact) spasm_lu_free(fact);
                    throw;
                }
                break;
            }
            // ---- election: keys of every shard, minimum on device 0, back to every shard
            for (int k = 0; k < nshards; k++) {
                HIPCHK(hipSetDevice(dev_of(k)));
                st[(size_t)k]->keys.ensure((size_t)m + 1);
                shard_elect(st[(size_t)k]->sh, (int64_t *)st[(size_t)k]->keys.p);
            }
            HIPCHK(hipSetDevice(dev_of(0)));
            stage.ensure((size_t)m + 1);
            for (int k = 1; k < nshards; k++) {
                HIPCHK(hipMemcpy(stage.p, st[(size_t)k]->keys.p, (size_t)m * sizeof(u64d), hipMemcpyDeviceToDevice));
                hipLaunchKernelGGL(k_min_u64, dim3(cdiv(m, 256)), dim3(256), 0, nullptr, (i64d)m, st[0]->keys.p, stage.p);
                HIPCHK(hipGetLastError());
                HIPCHK(hipDeviceSynchronize());
            }
            int npiv = 0, n_open = 0;
            i64 tot_rows = 0, tot_ent = 0;
            for (int k = 0; k < nshards; k++) {
                HIPCHK(hipSetDevice(dev_of(k)));
                if (k > 0) HIPCHK(hipMemcpy(st[(size_t)k]->keys.p, st[0]->keys.p, (size_t)m * sizeof(u64d), hipMemcpyDeviceToDevice));
                npiv = shard_assign(st[(size_t)k]->sh, (const int64_t *)st[(size_t)k]->keys.p);
            }
            // "Faugere-Lachartre on columns" over the shards (enable_greedy_pivot_search, in the max_round sparse rounds like the
            // single-device loop): the steps of shard_open_step, the array each leaves merged on device 0 and handed back -- what the
            // four all-reduces per pass do between processes (sharded.py: set_keys_open).  The third, cycle-free search is not sharded.
            if (opts->enable_greedy_pivot_search && round < opts->max_round && npiv > 0 && !getenv("SPASM_AMD_MULTI_NO_OPEN_COLUMNS")) {
                for (int k = 0; k < nshards; k++) {
                    HIPCHK(hipSetDevice(dev_of(k)));
                    st[(size_t)k]->oc_i.ensure((size_t)m + 1);
                    st[(size_t)k]->oc_q.ensure((size_t)m + 1);
                }
                HIPCHK(hipSetDevice(dev_of(0)));
                DevBuf<int> stage_i;
                stage_i.alloc((size_t)m + 1);
                stage.ensure((size_t)m + 1);
                // kind 0: int32 MAX, 1: int32 SUM, 2: u64 MIN; the merged array ends up in every shard's buffer
                auto merge = [&](int kind) {
                    HIPCHK(hipSetDevice(dev_of(0)));
                    for (int k = 1; k < nshards; k++) {
                        if (kind == 2) {
                            HIPCHK(hipMemcpy(stage.p, st[(size_t)k]->oc_q.p, (size_t)m * sizeof(u64d), hipMemcpyDeviceToDevice));
                            hipLaunchKernelGGL(k_min_u64, dim3(cdiv(m, 256)), dim3(256), 0, nullptr, (i64d)m, st[0]->oc_q.p, stage.p);
                        } else {
                            HIPCHK(hipMemcpy(stage_i.p, st[(size_t)k]->oc_i.p, (size_t)m * sizeof(int), hipMemcpyDeviceToDevice));
                            if (kind == 0) hipLaunchKernelGGL(k_max_i32, dim3(cdiv(m, 256)), dim3(256), 0, nullptr, (i64d)m, st[0]->oc_i.p, stage_i.p);
                            else hipLaunchKernelGGL(k_add_i32, dim3(cdiv(m, 256)), dim3(256), 0, nullptr, (i64d)m, st[0]->oc_i.p, stage_i.p);
                        }
                        HIPCHK(hipGetLastError());
                        HIPCHK(hipDeviceSynchronize());
                    }
                    for (int k = 1; k < nshards; k++) {
                        HIPCHK(hipSetDevice(dev_of(k)));
                        if (kind == 2) HIPCHK(hipMemcpy(st[(size_t)k]->oc_q.p, st[0]->oc_q.p, (size_t)m * sizeof(u64d), hipMemcpyDeviceToDevice));
                        else HIPCHK(hipMemcpy(st[(size_t)k]->oc_i.p, st[0]->oc_i.p, (size_t)m * sizeof(int), hipMemcpyDeviceToDevice));
                    }
                };
                // one step on every shard: in / out = the shard's own buffer of the kind (0: none, 1: oc_i, 2: oc_q); returns shard 0's count
                auto step_all = [&](int step, int pass, int in_kind, int out_kind) {
                    int ret = 0;
                    for (int k = 0; k < nshards; k++) {
                        HIPCHK(hipSetDevice(dev_of(k)));
                        ShardState &q = *st[(size_t)k];
                        const void *in = in_kind == 1 ? (const void *)q.oc_i.p : in_kind == 2 ? (const void *)q.oc_q.p : nullptr;
                        void *out = out_kind == 1 ? (void *)q.oc_i.p : out_kind == 2 ? (void *)q.oc_q.p : nullptr;
                        const int rc = shard_open_step(q.sh, step, pass, in, out);
                        if (k == 0) ret = rc;
                        else if (rc != ret) throw EngineError("spasm_amd_echelonize_multi: the shards disagree on the pivots of a pass");
                    }
                    return ret;
                };
                step_all(OPEN_BEGIN, 0, 0, 1);
                merge(0);
                for (int pass = 1; pass <= OPEN_PASSES; pass++) {
                    step_all(OPEN_HIST, pass, 1, 1);
                    merge(1);
                    step_all(OPEN_PROPOSE, pass, 1, 2);
                    merge(2);
                    step_all(OPEN_ACCEPT, pass, 2, 1);
                    merge(0);
                    if (step_all(OPEN_RECORD, pass, 1, 1) == 0) break;
                    merge(0);
                }
                n_open = step_all(OPEN_FINISH, 0, 0, 0);
                if (n_open) spasm_logf("[pivots] ``Faugère-Lachartre on columns'' over %d shards: %d pivots found\n", nshards, n_open);
            }
            for (int k = 0; k < nshards; k++) {
                HIPCHK(hipSetDevice(dev_of(k)));
                ShardState &q = *st[(size_t)k];
                npiv = shard_finish_keys(q.sh, &q.n_own, &q.nnz_own);
                tot_rows += q.n_own;
                tot_ent += q.nnz_own;
            }
            if (npiv == 0) break;
            if (tot_rows != npiv) throw EngineError("spasm_amd_echelonize_multi: the shards do not own the elected pivot rows between them");
            // ---- exchange: every shard exports what it owns; every shard receives the concatenation
            for (int k = 0; k < nshards; k++) {
                HIPCHK(hipSetDevice(dev_of(k)));
                ShardState &q = *st[(size_t)k];
                q.hdr.ensure((size_t)2 * (size_t)std::max(q.n_own, 1));
                q.ent.ensure((size_t)2 * (size_t)std::max<i64>(q.nnz_own, 1));
                shard_export(q.sh, q.hdr.p, q.ent.p);
            }
            for (int k = 0; k < nshards; k++) {
                HIPCHK(hipSetDevice(dev_of(k)));
                ShardState &q = *st[(size_t)k];
                q.hdr_all.ensure((size_t)2 * (size_t)std::max(npiv, 1));
                q.ent_all.ensure((size_t)2 * (size_t)std::max<i64>(tot_ent, 1));
                i64 hr = 0, he = 0;
                for (int src = 0; src < nshards; src++) {
                    const ShardState &o = *st[(size_t)src];
                    if (o.n_own > 0) HIPCHK(hipMemcpy(q.hdr_all.p + 2 * hr, o.hdr.p, (size_t)2 * (size_t)o.n_own * sizeof(int), hipMemcpyDeviceToDevice));
                    if (o.nnz_own > 0) HIPCHK(hipMemcpy(q.ent_all.p + 2 * he, o.ent.p, (size_t)2 * (size_t)o.nnz_own * sizeof(int), hipMemcpyDeviceToDevice));
                    hr += o.n_own;
                    he += o.nnz_own;
                }
            }
            for (int k = 0; k < nshards; k++) {
                HIPCHK(hipSetDevice(dev_of(k)));
                ShardState &q = *st[(size_t)k];
                spasm_amd_shard *sh = q.sh;
                q.plan = shard_import(sh, npiv, tot_ent, q.hdr_all.p, q.ent_all.p, !ddf_now); // (the shard hands its matrix to the plan)
                q.sh = nullptr;
                delete sh;
            }
            // ---- the round's rows of U, from shard 0 (identical on all)
            {
                HIPCHK(hipSetDevice(dev_of(0)));
                std::vector<int> pc((size_t)npiv), ro((size_t)npiv);
                struct spasm_csr *Uc = plan_fetch_U(st[0]->plan, pc.data(), ro.data());
                const i64 base = U.p.back();
                for (int k = 0; k < npiv; k++) {
                    U.p.push_back(base + Uc->p[k + 1]);
                    U.pivcol.push_back(pc[(size_t)k]);
                    U.orig.push_back(ro[(size_t)k]);
                }
                const i64 uz = Uc->p[npiv];
                if (uz > 0) {
                    memcpy(U.j.grow((size_t)uz), Uc->j, sizeof(int) * (size_t)uz);
                    memcpy(U.x.grow((size_t)uz), Uc->x, sizeof(int) * (size_t)uz);
                }
                spasm_csr_free(Uc);
            }
            if (ddf_now) {
                if (dense_round()) {
                    g_multi_finish = 1;
                    spasm_logf("[echelonize] round %d (sharded): %d pivots, %lld rows / %lld entries before it; its Schur complement went dense\n", round, npiv,
                               (long long)rows_left, (long long)nnz_left);
                    break;
                }
                for (int k = 0; k < nshards; k++) {
                    HIPCHK(hipSetDevice(dev_of(k)));
                    shard_import_finish(st[(size_t)k]->plan);
                }
            }
            // ---- the Schur complement of every shard's rows becomes its matrix of the next round, on its device
            last_nnz = nnz_left;
            for (int k = 0; k < nshards; k++) {
                HIPCHK(hipSetDevice(dev_of(k)));
                ShardState &q = *st[(size_t)k];
                spasm_amd_schur_plan *P = q.plan;
                q.plan = nullptr; // consumed by the call, whatever happens
                q.sh = plan_advance(P, &q.rows, &q.nnz);
            }
            spasm_logf("[echelonize] round %d (sharded): %d pivots, %lld rows / %lld entries before it\n", round, npiv, (long long)rows_left, (long long)nnz_left);
            round++;
        }
    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
    struct spasm_lu *N = assemble_lu(U, n, m, prime, nullptr);
    spasm_logf("[echelonize] Done in %.1fs. Rank %d, %lld nz in basis\n", spasm_wtime() - t0, N->r, (long long)U.p.back());
    return N;
}

// ------------------------------------------------------------------------------------------------
// The dense finish over row shards with ONE PROCESS PER SHARD (spasm.jl_amd/sharded.py: torch.distributed, RCCL): the steps of
// dense_finish_multi behind the C ABI, the exchanges between them are the caller's collectives -- an all-gather of the candidate
// records (every rank then runs the election itself: deterministic, nothing is sent back) and one broadcast per owner of the
// packed winner rows.
// ------------------------------------------------------------------------------------------------
struct DShardBase {
    virtual ~DShardBase() {}
    virtual void build(spasm_amd_schur_plan *P, DenseW &W, int me, int nshards, int C) = 0;
    virtual void build_rows(spasm_amd_schur_plan *P, DenseFill &fl, int me, int nshards, int C) = 0; // the live rows of P->A instead of a round's Schur rows
    virtual void block_begin() = 0;
    virtual void candidates(int c0, int w, void *cand_dev) = 0;
    virtual void elect(const void *stack_dev, int w) = 0;
    virtual i64 pack(int c0, void *buf) = 0;
    virtual void unpack(int q, int c0, int owner, const void *buf) = 0;
    virtual void apply(int q, int c0, int w, int b1) = 0;
    virtual void block_end(int b0, int b1, int npan) = 0;
    virtual int finish() = 0;
    virtual int extract(const int *clist, HostU &U) = 0;
    virtual void info(int *KB, i64 *ldc, int *elem, int *nd) = 0;
    PanelGlob hglob;
    int nshards = 1;
};

template <typename DT> struct DShardT : DShardBase {
    DenseShard<DT> sh;
    void build(spasm_amd_schur_plan *P, DenseW &W, int me, int nsh, int C) override
    {
        nshards = nsh;
        memset(&hglob, 0, sizeof hglob);
        sh.me = me;
        sh.C = C;
        sh.ldc = ((i64)C + 63) / 64 * 64;
        sh.F = P->R.F;
        sh.KB = dense_kb();
        sh.R = P->R.nnp;
        sh.s = P->R.stream;
        schur_dense_build(P->R, P->A, sh.R, W, sh.KB, sh.D, sh.row_orig, P->R.stream);
        sh.setup(nsh, true);
    }
    void build_rows(spasm_amd_schur_plan *P, DenseFill &fl, int me, int nsh, int C) override
    {
        nshards = nsh;
        memset(&hglob, 0, sizeof hglob);
        sh.me = me;
        sh.C = C;
        sh.ldc = ((i64)C + 63) / 64 * 64;
        sh.F = P->R.F;
        sh.KB = dense_kb();
        sh.R = fl.R;
        sh.s = P->R.stream;
        fl.build(sh.KB, sh.D, sh.row_orig); // (KB guest rows behind the shard's own, as schur_dense_build leaves them)
        sh.setup(nsh, true);
    }
    void block_begin() override { sh.block_begin(); }
    void candidates(int c0, int w, void *cand_dev) override
    {
        sh.candidates(c0, w);
        HIPCHK(hipMemcpyAsync(cand_dev, sh.cand.p, sizeof(CandRec), hipMemcpyDeviceToDevice, sh.s));
        HIPCHK(hipStreamSynchronize(sh.s));
    }
    void elect(const void *stack_dev, int w) override { sh.elect_from(stack_dev, nshards, w, &hglob); }
    i64 pack(int c0, void *buf) override
    {
        sh.pack(c0, &hglob, true);
        return sh.export_to(c0, &hglob, buf);
    }
    void unpack(int q, int c0, int owner, const void *buf) override { sh.import_from(q, c0, owner, &hglob, buf); }
    void apply(int q, int c0, int w, int b1) override { sh.apply_panel(q, c0, w, b1); }
    void block_end(int b0, int b1, int npan) override { sh.block_end(b0, b1, npan); }
    int finish() override { return sh.finish(); }
    int extract(const int *clist, HostU &U) override { return dense_extract_U(sh.D.p, sh.C, sh.ldc, sh.own_pc.p, clist, sh.row_orig.p, U, sh.s); }
    void info(int *KB, i64 *ldc, int *elem, int *nd) override
    {
        if (KB) *KB = sh.KB;
        if (ldc) *ldc = sh.ldc;
        if (elem) *elem = (int)sizeof(DT);
        if (nd) *nd = sh.ND;
    }
};

} // namespace

struct spasm_amd_dshard {
    spasm_amd_schur_plan *plan = nullptr; // not owned
    std::unique_ptr<DenseW> dw;           // the Schur rows of the plan's round through the dense W ..
    std::unique_ptr<DenseFill> fl;        // .. or (spasm_amd_dshard_open_rows) the live rows of the shard's matrix as they are
    const int *cflag() const { return fl ? fl->cflag.p : dw->cflag.p; }
    const int *clist() const { return fl ? fl->clist.p : dw->clist.p; }
    std::unique_ptr<DShardBase> impl;
    int me = 0, nshards = 1, C = 0;
};

namespace {

spasm_amd_dshard *dshard_open(spasm_amd_schur_plan *P, int me, int nshards)
{
    if (!P) throw EngineError("spasm_amd_dshard_open: null plan");
    if (!P->R.F.small) throw EngineError("spasm_amd_dshard_open: the dense finish over shards takes primes below 2^16");
    const int elem = P->R.F.p <= 255 ? 1 : 2;
    if (nshards < 1 || nshards > DM_MAXSHARDS || nshards * DP_W > 147456 / (DP_W * elem)) throw EngineError("spasm_amd_dshard_open: too many shards for the election workgroup");
    if (me < 0 || me >= nshards) throw EngineError("spasm_amd_dshard_open: shard number out of range");
    std::unique_ptr<spasm_amd_dshard> ds(new spasm_amd_dshard());
    ds->plan = P;
    ds->me = me;
    ds->nshards = nshards;
    ds->dw.reset(new DenseW(P->R, P->A, P->R.stream));
    ds->dw->flag_columns();
    HIPCHK(hipStreamSynchronize(P->R.stream));
    return ds.release();
}

// the same over the shard's CURRENT rows (no round, no U): a remainder that is dense already is eliminated by all ranks together
// where it is (the one-process path's dense_now)
spasm_amd_dshard *dshard_open_rows(spasm_amd_shard *S, int me, int nshards)
{
    if (!S || !S->plan) throw EngineError("spasm_amd_dshard_open_rows: null shard");
    spasm_amd_schur_plan *P = S->plan;
    if (!P->R.F.small) throw EngineError("spasm_amd_dshard_open_rows: the dense finish over shards takes primes below 2^16");
    const int elem = P->R.F.p <= 255 ? 1 : 2;
    if (nshards < 1 || nshards > DM_MAXSHARDS || nshards * DP_W > 147456 / (DP_W * elem)) throw EngineError("spasm_amd_dshard_open_rows: too many shards for the election workgroup");
    if (me < 0 || me >= nshards) throw EngineError("spasm_amd_dshard_open_rows: shard number out of range");
    std::unique_ptr<spasm_amd_dshard> ds(new spasm_amd_dshard());
    ds->plan = P;
    ds->me = me;
    ds->nshards = nshards;
    ds->fl.reset(new DenseFill(P->A, P->R.stream));
    ds->fl->flag_columns();
    return ds.release();
}

// the rows of U this shard owns after the finish, as a host CSR; pivcol_out / row_out: their pivot columns and originating rows
struct spasm_csr *dshard_fetch_U(spasm_amd_dshard *ds, int *pivcol_out, int *row_out, int *n_out)
{
    HostU U;
    U.p.push_back(0);
    const int got = ds->impl->extract(ds->clist(), U);
    const i64 nz = U.p.back();
    struct spasm_csr *Uc = spasm_csr_alloc(got, ds->plan->A.m, nz, ds->plan->prime, true);
    if (!Uc) throw EngineError("out of host memory for the shard's rows of U");
    for (int k = 0; k <= got; k++) Uc->p[k] = U.p[(size_t)k];
    if (nz > 0) {
        memcpy(Uc->j, U.j.data(), sizeof(int) * (size_t)nz);
        memcpy(Uc->x, U.x.data(), sizeof(int) * (size_t)nz);
    }
    for (int k = 0; k < got; k++) {
        if (pivcol_out) pivcol_out[k] = U.pivcol[(size_t)k];
        if (row_out) row_out[k] = U.orig[(size_t)k];
    }
    if (n_out) *n_out = got;
    return Uc;
}

} // namespace

// ================================================================================================
// C ABI
// ================================================================================================

extern "C" {

SPASM_API int spasm_amd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

SPASM_API int spasm_amd_set_device(int dev)
{
    if (hipSetDevice(dev) != hipSuccess) { spasm_set_error("hipSetDevice(%d) failed", dev); return 1; }
    return 0;
}

// engine extension of spasm_echelonize (reference src/SpaSM.jl:863): the same result from `nshards` row shards on the devices of
// this process (include/spasm_amd.h)
SPASM_API struct spasm_lu *spasm_amd_echelonize_multi(const struct spasm_csr *A, struct echelonize_opts *opts, int nshards)
{
    spasm_clear_error();
    try {
        return do_echelonize_multi(A, opts, nshards);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_echelonize_multi: %s", e.what());
        return nullptr;
    }
}

// reference src/SpaSM.jl:863
SPASM_API struct spasm_lu *spasm_echelonize(const struct spasm_csr *A, struct echelonize_opts *opts)
{
    spasm_clear_error();
    try {
        return do_echelonize(A, opts);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_echelonize: %s", e.what());
        return nullptr;
    }
}

// rank(A) = rank(echelonize(A)) (reference src/SpaSM.jl:1149) without the rows of U ever leaving the device; -1 on error
SPASM_API i64 spasm_amd_rank(const struct spasm_csr *A, struct echelonize_opts *opts)
{
    spasm_clear_error();
    try {
        struct echelonize_opts o;
        if (opts) o = *opts;
        else spasm_echelonize_init_opts(&o);
        if (o.L) throw EngineError("the L factor needs the rows of U");
        i64 r = -1;
        (void)do_echelonize(A, &o, &r);
        return r;
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_rank: %s", e.what());
        return -1;
    }
}

// reference src/SpaSM.jl:879
SPASM_API struct spasm_csr *spasm_kernel(const struct spasm_lu *fact)
{
    spasm_clear_error();
    try {
        return do_kernel(fact);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_kernel: %s", e.what());
        return nullptr;
    }
}

SPASM_API struct spasm_csr *spasm_amd_kernel_strided(const struct spasm_lu *fact, int first, int step)
{
    spasm_clear_error();
    try {
        return do_kernel(fact, first, step);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_kernel_strided: %s", e.what());
        return nullptr;
    }
}

// reference src/SpaSM.jl:694-722: x * U = B[k] for ONE row.  The row goes through the batched device solve (do_trisolve: the Schur
// machinery with U as the pivot rows), then the solution is laid out as libspasm does: x scattered over the m columns (x_b on the
// pivot columns, x_a on the others; only the entries of the pattern are written), the pattern in xj[top .. m), top returned.
// Pattern order: the pivot columns reached, then the columns of x_a, each ascending.  xj[0 .. top) and xj[m .. 3m) (libspasm's
// DFS stacks and marks, "it remains OK", :700) are left as they are.  -1 on failure.
SPASM_API int spasm_sparse_triangular_solve(const struct spasm_csr *U, const struct spasm_csr *B, int k, int *xj, spasm_ZZp *x, const int *qinv)
{
    spasm_clear_error();
    struct spasm_csr *X = nullptr, *Rs = nullptr;
    struct spasm_csr one;
    try {
        if (!U || !B || !xj || !x || !qinv) throw EngineError("spasm_sparse_triangular_solve: null argument");
        if (k < 0 || k >= B->n) throw EngineError("spasm_sparse_triangular_solve: row out of range");
        const int m = U->m;
        // row k of B as a matrix of its own (borrowed arrays)
        i64 rp[2] = {0, B->p[k + 1] - B->p[k]};
        one = *B;
        one.n = 1;
        one.nzmax = rp[1];
        one.p = rp;
        one.j = B->j + B->p[k];
        one.x = B->x + B->p[k];
        X = do_trisolve(U, qinv, &one, nullptr, &Rs);
        std::vector<int> colof((size_t)std::max(U->n, 1), -1);
        for (int j = 0; j < m; j++) if (qinv[j] >= 0 && qinv[j] < U->n) colof[(size_t)qinv[j]] = j;
        std::vector<int> pat;
        for (i64 t = X->p[0]; t < X->p[1]; t++) { const int j = colof[(size_t)X->j[t]]; x[j] = X->x[t]; pat.push_back(j); }
        std::sort(pat.begin(), pat.end());
        const size_t npiv_pat = pat.size();
        for (i64 t = Rs->p[0]; t < Rs->p[1]; t++) { x[Rs->j[t]] = Rs->x[t]; pat.push_back(Rs->j[t]); }
        std::sort(pat.begin() + (long)npiv_pat, pat.end());
        const int top = m - (int)pat.size();
        for (size_t t = 0; t < pat.size(); t++) xj[(size_t)top + t] = pat[t];
        spasm_csr_free(X);
        spasm_csr_free(Rs);
        return top;
    } catch (const std::exception &e) {
        if (X) spasm_csr_free(X);
        if (Rs) spasm_csr_free(Rs);
        spasm_set_error("%s", e.what());
        return -1;
    }
}

SPASM_API struct spasm_csr *spasm_amd_triangular_solve(const struct spasm_csr *U, const int *qinv, const struct spasm_csr *B, unsigned char *ok)
{
    spasm_clear_error();
    try {
        return do_trisolve(U, qinv, B, ok);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_triangular_solve: %s", e.what());
        return nullptr;
    }
}

// reference src/SpaSM.jl:915-923
SPASM_API struct spasm_csr *spasm_gesv(const struct spasm_lu *fact, const struct spasm_csr *B, bool *ok)
{
    spasm_clear_error();
    try {
        static_assert(sizeof(bool) == 1, "ok is an array of bytes");
        return do_gesv(fact, B, (unsigned char *)ok);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_gesv: %s", e.what());
        return nullptr;
    }
}

// reference src/SpaSM.jl:895-905: x (n = rows of A entries) with x * A == b (m entries); false when there is no solution
// reference src/SpaSM.jl:928: the certificate of rank(A) >= fact->r (include/spasm_amd.h): rows = the pivotal rows, columns = the
// pivot columns, x = the Fiat-Shamir challenge, y with y * A[rows, columns] == x solved on the device (the r x r submatrix is
// echelonized with L, then the two triangular solves of spasm_solve).  NULL on failure (e.g. when U is not of full rank r on A).
SPASM_API struct spasm_rank_certificate *spasm_certificate_rank_create(const struct spasm_csr *A, const uint8_t *hash, const struct spasm_lu *fact)
{
    spasm_clear_error();
    struct spasm_rank_certificate *P = nullptr;
    struct spasm_csr *Cm = nullptr;
    struct spasm_lu *fc = nullptr;
    try {
        if (!A || !hash || !fact || !fact->U || !fact->qinv || !fact->p) throw EngineError("null argument");
        const int r = fact->r, n = A->n, m = A->m;
        const i64 prime = A->field->p;
        if (fact->U->m != m || r > n || r > m) throw EngineError("the factorization does not belong to this matrix");
        P = (struct spasm_rank_certificate *)calloc(1, sizeof *P);
        if (!P) throw EngineError("out of host memory");
        P->r = r;
        P->prime = prime;
        memcpy(P->hash, hash, 32);
        P->i = (int *)malloc(sizeof(int) * (size_t)std::max(r, 1));
        P->j = (int *)malloc(sizeof(int) * (size_t)std::max(r, 1));
        P->x = (spasm_ZZp *)calloc((size_t)std::max(r, 1), sizeof(spasm_ZZp));
        P->y = (spasm_ZZp *)calloc((size_t)std::max(r, 1), sizeof(spasm_ZZp));
        if (!P->i || !P->j || !P->x || !P->y) throw EngineError("out of host memory");
        std::vector<int> pos((size_t)std::max(m, 1), -1);
        for (int j = 0; j < m; j++) {
            const int k = fact->qinv[j];
            if (k >= 0) { if (k >= r) throw EngineError("qinv names a row beyond the rank"); P->j[k] = j; pos[(size_t)j] = k; }
        }
        i64 cz = 0;
        for (int k = 0; k < r; k++) {
            const int i = fact->p[k];
            if (i < 0 || i >= n) throw EngineError("fact->p does not list the pivotal rows");
            P->i[k] = i;
            for (i64 t = A->p[i]; t < A->p[i + 1]; t++) cz += pos[(size_t)A->j[t]] >= 0;
        }
        spasm_cert_challenge(hash, prime, r, P->i, P->j, P->x);
        if (r == 0) return P;
        // C = A[rows, columns], r x r
        Cm = spasm_csr_alloc(r, r, cz, prime, true);
        if (!Cm) throw EngineError("out of host memory");
        i64 w = 0;
        for (int k = 0; k < r; k++) {
            Cm->p[k] = w;
            const int i = P->i[k];
            for (i64 t = A->p[i]; t < A->p[i + 1]; t++) {
                const int c = pos[(size_t)A->j[t]];
                if (c >= 0) { Cm->j[w] = c; Cm->x[w] = A->x[t]; w++; }
            }
        }
        Cm->p[r] = w;
        struct echelonize_opts o;
        spasm_echelonize_init_opts(&o);
        o.L = true;
        o.enable_greedy_pivot_search = false;
        fc = do_echelonize(Cm, &o);
        if (!fc || fc->r != r) throw EngineError("the pivotal rows and pivot columns do not span a non-singular submatrix");
        if (!spasm_solve(fc, P->x, P->y)) throw EngineError("the challenge has no solution (singular submatrix)");
        spasm_lu_free(fc);
        spasm_csr_free(Cm);
        return P;
    } catch (const std::exception &e) {
        std::string msg = e.what(); // (spasm_solve / spasm_lu_free below reset the error text)
        if (fc) spasm_lu_free(fc);
        if (Cm) spasm_csr_free(Cm);
        spasm_rank_certificate_free(P);
        spasm_set_error("spasm_certificate_rank_create: %s", msg.c_str());
        return nullptr;
    }
}

SPASM_API bool spasm_solve(const struct spasm_lu *fact, const spasm_ZZp *b, spasm_ZZp *x)
{
    spasm_clear_error();
    try {
        if (!fact || !fact->U || !fact->L || !b || !x) throw EngineError("null argument (a factorization with L is needed)");
        const int m = fact->U->m, n = fact->L->n;
        const i64 prime = fact->U->field->p;
        const ZpField F = zp_field_make(prime);
        i64 nz = 0;
        for (int j = 0; j < m; j++) nz += zp_reduce(F, (int64_t)b[j]) != 0;
        struct spasm_csr *B = spasm_csr_alloc(1, m, nz, prime, true);
        if (!B) throw EngineError("out of host memory");
        std::unique_ptr<struct spasm_csr, void (*)(struct spasm_csr *)> Bguard(B, spasm_csr_free);
        i64 w = 0;
        B->p[0] = 0;
        for (int j = 0; j < m; j++) {
            const int v = zp_reduce(F, (int64_t)b[j]);
            if (v != 0) { B->j[w] = j; B->x[w] = v; w++; }
        }
        B->p[1] = w;
        unsigned char ok = 0;
        struct spasm_csr *X = do_gesv(fact, B, &ok);
        for (int i = 0; i < n; i++) x[i] = 0;
        for (i64 q = X->p[0]; q < X->p[1]; q++) x[X->j[q]] = X->x[q];
        spasm_csr_free(X);
        return ok != 0;
    } catch (const std::exception &e) {
        spasm_set_error("spasm_solve: %s", e.what());
        return false;
    }
}

SPASM_API struct spasm_csr *spasm_rref(const struct spasm_lu *fact, int *Rqinv)
{
    spasm_clear_error();
    try {
        return do_rref(fact, Rqinv);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_rref: %s", e.what());
        return nullptr;
    }
}

// reference src/SpaSM.jl:589
SPASM_API struct spasm_csr *spasm_transpose(const struct spasm_csr *A)
{
    spasm_clear_error();
    try {
        return do_transpose(A);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_transpose: %s", e.what());
        return nullptr;
    }
}

SPASM_API spasm_amd_schur_plan *spasm_amd_schur_plan_create(const struct spasm_csr *A, int row_lo, int row_hi)
{
    spasm_clear_error();
    try {
        return plan_create(A, row_lo, row_hi);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_schur_plan_create: %s", e.what());
        return nullptr;
    }
}

SPASM_API spasm_amd_schur_plan *spasm_amd_schur_plan_create_strided(const struct spasm_csr *A, int row_lo, int row_hi, int stride)
{
    spasm_clear_error();
    try {
        return plan_create(A, row_lo, row_hi, stride);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_schur_plan_create_strided: %s", e.what());
        return nullptr;
    }
}

SPASM_API spasm_amd_shard *spasm_amd_shard_create_strided(const struct spasm_csr *A, int row_lo, int row_hi, int stride)
{
    spasm_clear_error();
    try {
        return shard_create(A, row_lo, row_hi, stride);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_shard_create_strided: %s", e.what());
        return nullptr;
    }
}

SPASM_API int spasm_amd_schur_plan_run(spasm_amd_schur_plan *plan, void *stream)
{
    try {
        plan_run(plan, (hipStream_t)stream);
        return 0;
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_schur_plan_run: %s", e.what());
        return 1;
    }
}

SPASM_API void spasm_amd_schur_plan_class_timing(spasm_amd_schur_plan *plan, int on)
{
    if (plan) plan->R.class_timing = on != 0;
}

SPASM_API int spasm_amd_schur_plan_stats(spasm_amd_schur_plan *plan, struct spasm_amd_round_stats *stats)
{
    try {
        if (!plan->ran) throw EngineError("run the plan first");
        plan->R.fetch_step();
        plan->R.hctr.applications = plan->exact_applications;
        plan->R.hctr.nnz_reduced = plan->exact_nnz_reduced;
        plan->R.hctr.segments = plan->exact_segments;
        fill_stats(*stats, plan->R, 0, plan->hi > plan->lo ? (plan->hi - plan->lo + plan->stride - 1) / plan->stride : 0, plan->nnz_in);
        return 0;
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_schur_plan_stats: %s", e.what());
        return 1;
    }
}

SPASM_API struct spasm_csr *spasm_amd_schur_plan_fetch(spasm_amd_schur_plan *plan, int *p_out)
{
    spasm_clear_error();
    try {
        return plan_fetch(plan, p_out);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_schur_plan_fetch: %s", e.what());
        return nullptr;
    }
}

SPASM_API struct spasm_csr *spasm_amd_schur_plan_fetch_U(spasm_amd_schur_plan *plan, int *pivcol_out, int *row_out)
{
    spasm_clear_error();
    try {
        if (!plan) throw EngineError("null plan");
        return plan_fetch_U(plan, pivcol_out, row_out);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_schur_plan_fetch_U: %s", e.what());
        return nullptr;
    }
}

SPASM_API void spasm_amd_schur_plan_free(spasm_amd_schur_plan *plan) { delete plan; }

SPASM_API spasm_amd_shard *spasm_amd_shard_create(const struct spasm_csr *A, int row_lo, int row_hi)
{
    spasm_clear_error();
    try {
        return shard_create(A, row_lo, row_hi);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_shard_create: %s", e.what());
        return nullptr;
    }
}

SPASM_API int spasm_amd_shard_elect(spasm_amd_shard *sh, int64_t *keys_dev)
{
    try {
        shard_elect(sh, keys_dev);
        return 0;
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_shard_elect: %s", e.what());
        return 1;
    }
}

SPASM_API int spasm_amd_shard_set_keys(spasm_amd_shard *sh, const int64_t *keys_dev, int *n_owned, i64 *nnz_owned)
{
    try {
        return shard_set_keys(sh, keys_dev, n_owned, nnz_owned);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_shard_set_keys: %s", e.what());
        return -1;
    }
}

SPASM_API int spasm_amd_shard_assign(spasm_amd_shard *sh, const int64_t *keys_dev)
{
    try {
        return shard_assign(sh, keys_dev);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_shard_assign: %s", e.what());
        return -1;
    }
}

SPASM_API int spasm_amd_shard_open_step(spasm_amd_shard *sh, int step, int pass, const void *in_dev, void *out_dev)
{
    try {
        return shard_open_step(sh, step, pass, in_dev, out_dev);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_shard_open_step: %s", e.what());
        return -1;
    }
}

SPASM_API int spasm_amd_shard_finish_keys(spasm_amd_shard *sh, int *n_owned, i64 *nnz_owned)
{
    try {
        return shard_finish_keys(sh, n_owned, nnz_owned);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_shard_finish_keys: %s", e.what());
        return -1;
    }
}

SPASM_API int spasm_amd_shard_export(spasm_amd_shard *sh, int *hdr_dev, int *ent_dev)
{
    try {
        shard_export(sh, hdr_dev, ent_dev);
        return 0;
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_shard_export: %s", e.what());
        return 1;
    }
}

SPASM_API spasm_amd_schur_plan *spasm_amd_shard_import(spasm_amd_shard *sh, int n_rows, i64 n_entries, const int *hdr_dev, const int *ent_dev)
{
    spasm_clear_error();
    try {
        return shard_import(sh, n_rows, n_entries, hdr_dev, ent_dev);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_shard_import: %s", e.what());
        return nullptr;
    }
}

SPASM_API spasm_amd_shard *spasm_amd_schur_plan_advance(spasm_amd_schur_plan *plan, int *rows_out, i64 *nnz_out)
{
    spasm_clear_error();
    try {
        return plan_advance(plan, rows_out, nnz_out);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_schur_plan_advance: %s", e.what());
        return nullptr;
    }
}

SPASM_API struct spasm_csr *spasm_amd_shard_fetch(spasm_amd_shard *sh)
{
    spasm_clear_error();
    try {
        return shard_fetch(sh);
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_shard_fetch: %s", e.what());
        return nullptr;
    }
}

SPASM_API void spasm_amd_shard_free(spasm_amd_shard *sh)
{
    if (!sh) return;
    delete sh->plan;
    delete sh;
}

// the device's field arithmetic on n test vectors (canonical residues a, b, c in; 8 results per vector out): see k_zp_probe
SPASM_API int spasm_amd_zp_probe(i64 prime, int n, const int *a, const int *b, const int *c, int *out)
{
    spasm_clear_error();
    try {
        require_device();
        if (prime <= 2 || prime > 0xfffffffbLL || n < 0) throw EngineError("bad arguments");
        const ZpField F = zp_field_make(prime);
        DevBuf<int> da, db, dc, dout;
        da.alloc((size_t)n + 1); db.alloc((size_t)n + 1); dc.alloc((size_t)n + 1); dout.alloc((size_t)8 * n + 1);
        hipStream_t s = nullptr;
        if (n > 0) {
            HIPCHK(hipMemcpyAsync(da.p, a, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(db.p, b, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(dc.p, c, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
            if (F.small) hipLaunchKernelGGL(k_zp_probe<true>, dim3(cdiv(n, 256)), dim3(256), 0, s, F, n, da.p, db.p, dc.p, dout.p);
            else hipLaunchKernelGGL(k_zp_probe<false>, dim3(cdiv(n, 256)), dim3(256), 0, s, F, n, da.p, db.p, dc.p, dout.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(out, dout.p, (size_t)8 * n * sizeof(int), hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipStreamSynchronize(s));
        return 0;
    } catch (const std::exception &e) {
        spasm_set_error("spasm_amd_zp_probe: %s", e.what());
        return 1;
    }
}

SPASM_API int spasm_amd_multi_last_finish(void) { return g_multi_finish; }

// ---- the dense finish over row shards, one process per shard (include/spasm_amd.h)
#define DSHARD_TRY(name, body, fail)                                   \
    spasm_clear_error();                                               \
    try { body }                                                       \
    catch (const std::exception &e) { spasm_set_error(name ": %s", e.what()); return fail; }
// (a step called out of order -- no handle, or before dshard_build -- is an error, not a crash)
#define DSHARD_BUILT(ds) do { if (!(ds) || !(ds)->impl) throw EngineError("no dense shard (spasm_amd_dshard_open / _density / _build come first)"); } while (0)
#define DSHARD_OPEN(ds) do { if (!(ds) || (!(ds)->dw && !(ds)->fl)) throw EngineError("no dense shard (spasm_amd_dshard_open comes first)"); } while (0)

SPASM_API spasm_amd_schur_plan *spasm_amd_shard_import_U(spasm_amd_shard *sh, int n_rows, i64 n_entries, const int *hdr_dev, const int *ent_dev)
{
    DSHARD_TRY("spasm_amd_shard_import_U", return shard_import(sh, n_rows, n_entries, hdr_dev, ent_dev, false);, nullptr)
}
SPASM_API int spasm_amd_schur_plan_prepare(spasm_amd_schur_plan *plan)
{
    DSHARD_TRY("spasm_amd_schur_plan_prepare", if (!plan) throw EngineError("null plan"); shard_import_finish(plan); return 0;, -1)
}
SPASM_API spasm_amd_dshard *spasm_amd_dshard_open(spasm_amd_schur_plan *plan, int me, int nshards)
{
    DSHARD_TRY("spasm_amd_dshard_open", return dshard_open(plan, me, nshards);, nullptr)
}
SPASM_API spasm_amd_dshard *spasm_amd_dshard_open_rows(spasm_amd_shard *sh, int me, int nshards)
{
    DSHARD_TRY("spasm_amd_dshard_open_rows", return dshard_open_rows(sh, me, nshards);, nullptr)
}
SPASM_API int spasm_amd_dshard_flags(spasm_amd_dshard *ds, int *flags_dev)
{
    DSHARD_TRY("spasm_amd_dshard_flags", DSHARD_OPEN(ds);
               HIPCHK(hipMemcpy(flags_dev, ds->cflag(), ((size_t)ds->plan->A.m + 1) * sizeof(int), hipMemcpyDeviceToDevice)); return 0;, -1)
}
SPASM_API double spasm_amd_dshard_density(spasm_amd_dshard *ds, const int *flags_dev, int free_cols, int *C_out)
{
    DSHARD_TRY("spasm_amd_dshard_density", DSHARD_OPEN(ds);
               if (ds->fl) { // the rows as they are: the density is the matrix's own, which the caller knows; 1.0 = "dense, as you said"
                   HIPCHK(hipMemcpy(ds->fl->cflag.p, flags_dev, ((size_t)ds->plan->A.m + 1) * sizeof(int), hipMemcpyDeviceToDevice));
                   ds->C = ds->fl->finish_columns();
                   if (C_out) *C_out = ds->C;
                   return 1.0;
               }
               HIPCHK(hipMemcpy(ds->dw->cflag.p, flags_dev, ((size_t)ds->plan->A.m + 1) * sizeof(int), hipMemcpyDeviceToDevice));
               ds->C = ds->dw->finish_columns();
               if (C_out) *C_out = ds->C;
               Round &R = ds->plan->R;
               return ds->dw->estimate_density(R.np_rows.p, R.nnp, free_cols);, -1.0)
}
SPASM_API int spasm_amd_dshard_build(spasm_amd_dshard *ds)
{
    DSHARD_TRY("spasm_amd_dshard_build", DSHARD_OPEN(ds);
               if (ds->C <= 0) throw EngineError("no column left (call spasm_amd_dshard_density first)");
               if (ds->plan->R.F.p <= 255) ds->impl.reset(new DShardT<signed char>());
               else ds->impl.reset(new DShardT<short>());
               if (ds->fl) ds->impl->build_rows(ds->plan, *ds->fl, ds->me, ds->nshards, ds->C);
               else ds->impl->build(ds->plan, *ds->dw, ds->me, ds->nshards, ds->C);
               return 0;, -1)
}
SPASM_API int spasm_amd_dshard_info(spasm_amd_dshard *ds, int *C_out, int *KB, i64 *ldc, int *elem, int *nd, int *cand_bytes)
{
    DSHARD_TRY("spasm_amd_dshard_info", DSHARD_BUILT(ds); if (C_out) *C_out = ds->C; ds->impl->info(KB, ldc, elem, nd);
               if (cand_bytes) *cand_bytes = (int)sizeof(CandRec); return 0;, -1)
}
SPASM_API int spasm_amd_dshard_block_begin(spasm_amd_dshard *ds) { DSHARD_TRY("spasm_amd_dshard_block_begin", DSHARD_BUILT(ds); ds->impl->block_begin(); return 0;, -1) }
SPASM_API int spasm_amd_dshard_candidates(spasm_amd_dshard *ds, int c0, int w, void *cand_dev)
{
    DSHARD_TRY("spasm_amd_dshard_candidates", DSHARD_BUILT(ds); ds->impl->candidates(c0, w, cand_dev); return 0;, -1)
}
SPASM_API int spasm_amd_dshard_elect(spasm_amd_dshard *ds, const void *stack_dev, int w, int *npp, int *cnt, int *first)
{
    DSHARD_TRY("spasm_amd_dshard_elect", DSHARD_BUILT(ds); ds->impl->elect(stack_dev, w); if (npp) *npp = ds->impl->hglob.npp;
               for (int k = 0; k < ds->nshards; k++) { if (cnt) cnt[k] = ds->impl->hglob.cnt[k]; if (first) first[k] = ds->impl->hglob.first[k]; }
               return 0;, -1)
}
SPASM_API i64 spasm_amd_dshard_pack(spasm_amd_dshard *ds, int c0, void *buf_dev) { DSHARD_TRY("spasm_amd_dshard_pack", DSHARD_BUILT(ds); return ds->impl->pack(c0, buf_dev);, -1) }
SPASM_API int spasm_amd_dshard_unpack(spasm_amd_dshard *ds, int q, int c0, int owner, const void *buf_dev)
{
    DSHARD_TRY("spasm_amd_dshard_unpack", DSHARD_BUILT(ds); ds->impl->unpack(q, c0, owner, buf_dev); return 0;, -1)
}
SPASM_API int spasm_amd_dshard_apply(spasm_amd_dshard *ds, int q, int c0, int w, int b1) { DSHARD_TRY("spasm_amd_dshard_apply", DSHARD_BUILT(ds); ds->impl->apply(q, c0, w, b1); return 0;, -1) }
SPASM_API int spasm_amd_dshard_block_end(spasm_amd_dshard *ds, int b0, int b1, int npan)
{
    DSHARD_TRY("spasm_amd_dshard_block_end", DSHARD_BUILT(ds); ds->impl->block_end(b0, b1, npan); return 0;, -1)
}
SPASM_API int spasm_amd_dshard_finish(spasm_amd_dshard *ds) { DSHARD_TRY("spasm_amd_dshard_finish", DSHARD_BUILT(ds); return ds->impl->finish();, -1) }
SPASM_API struct spasm_csr *spasm_amd_dshard_fetch_U(spasm_amd_dshard *ds, int *pivcol_out, int *row_out, int *n_out)
{
    DSHARD_TRY("spasm_amd_dshard_fetch_U", DSHARD_BUILT(ds); return dshard_fetch_U(ds, pivcol_out, row_out, n_out);, nullptr)
}
SPASM_API void spasm_amd_dshard_close(spasm_amd_dshard *ds) { delete ds; }

SPASM_API int spasm_amd_last_rounds(struct spasm_amd_round_stats *out, int max_rounds)
{
    const int n = (int)g_last_rounds.size();
    for (int i = 0; i < n && i < max_rounds; i++) out[i] = g_last_rounds[(size_t)i];
    return n;
}

} // extern "C"
