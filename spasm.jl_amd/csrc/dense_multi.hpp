// Dense finish over row shards -- host side (included by engine.hip inside its anonymous namespace; device side: dense.hpp, the
// section "Dense finish over ROW SHARDS", which also states what crosses shards).
//
// libspasm finishes a dense remainder with spasm_schur_dense + spasm_ffpack_LU on one host (prototypes reference
// src/SpaSM.jl:765-766, :805-806).  BASELINE config 5 asks for that tail on 8 GPUs, and the 760k x 760k remainder of config 3 / 4
// does not fit one: here the rows of the remainder stay on the shard (device) that holds them -- like the Schur rows of the sparse
// rounds -- and a panel of 64 columns costs one gather of candidates (16.7 KB per shard), one PanelGlob back (18 KB) and the
// pivot rows themselves: 64 x (C - c0) bytes to every shard, C^2 / 2 bytes per shard over the whole elimination (config 5:
// 324k columns as bytes = 52 GB per GPU, a third of a second of xGMI against tens of seconds of GEMM).  A 2-D block-cyclic
// layout would divide that by the process-grid width at the price of moving the rows of D; with rows >> columns (Macaulay-like)
// or rows = columns and 8 shards the row layout already keeps the exchange well under the arithmetic.
//
// One host thread per shard drives its device; the threads meet at barriers where data changes hands (peer copies issued by the
// receiving side).  On a one-GPU box all shards share the device and its null stream serialises them: the protocol is the same.
#pragma once

// (engine.hip includes <condition_variable>, <mutex>, <thread>, <atomic> at its top: this file sits inside its namespace)

struct TeamBarrier {
    std::mutex m;
    std::condition_variable cv;
    const int n;
    int count = 0, gen = 0;
    explicit TeamBarrier(int n_) : n(n_) {}
    void wait()
    {
        std::unique_lock<std::mutex> l(m);
        const int g = gen;
        if (++count == n) { count = 0; gen++; cv.notify_all(); }
        else cv.wait(l, [&] { return gen != g; });
    }
};

// first error of any shard's thread; the others keep meeting at the barriers and skip their work
struct TeamError {
    std::mutex m;
    std::atomic<bool> failed{false};
    std::string what;
    template <class Fn> void guard(Fn &&f)
    {
        if (failed.load()) return;
        try { f(); }
        catch (const std::exception &e) {
            std::lock_guard<std::mutex> l(m);
            if (!failed.exchange(true)) what = e.what();
        }
    }
};

// fn(k) for every shard on a host thread of its own (fn sets its device); the first exception is rethrown
template <class Fn> void parallel_shards(int nshards, Fn &&fn)
{
    TeamError err;
    std::vector<std::thread> th;
    for (int k = 1; k < nshards; k++) th.emplace_back([&, k] { err.guard([&] { fn(k); }); });
    err.guard([&] { fn(0); });
    for (auto &t : th) t.join();
    if (err.failed.load()) throw EngineError(err.what);
}

inline int dense_kb()
{
    int KB = 1024;
    if (const char *e = getenv("SPASM_AMD_DENSE_KB")) KB = std::min(2048, std::max(64, atoi(e) / 64 * 64)); // tests: several blocks on small matrices
    return KB;
}

#define DM_RETIRED 0x3fffffff // seq of a guest row that holds nothing (yet): no pivot of any panel, not live

template <typename DT> struct DenseShard {
    int me = 0, dev = 0;
    int R = 0, C = 0;        // local rows, columns
    i64 ldc = 0;
    ZpField F;
    int KB = 1024, ND = 1, npanel = 16, xbytes = 1;
    int Rext = 0, G = 0, chunk = 0, Rp = 0, Cp = 0;
    bool inlds = false;
    int app_chunk = 0, G_app = 0;
    int Rs = 0;              // rows of the stacked candidates (root)
    int num_cu = 0;
    i64 fplane = 0, uplane = 0;
    hipStream_t s = nullptr;
    DevBuf<DT> D, P, Ps, expD;
    DevBuf<int> row_orig, seq, seq_tmp, candrow, invtab, pc_dummy, own_pc, own_map, seq_s, candrow_s, flag;
    DevBuf<signed char> Fd, Ut, expF;
    DevBuf<PanelInfo> info, info_tmp, info_s;
    DevBuf<PanelSync> sync, sync_s;
    DevBuf<DenseState> st, st_tmp, st_s;
    DevBuf<CandRec> cand, stack;
    DevBuf<PanelGlob> glob;

    // D (with KB zero rows behind its R rows) and row_orig have been moved in.  elect_here: this shard runs the election among the
    // candidates of all shards itself (one process per shard: every rank elects, nothing is sent back) -- else only shard 0 does
    void setup(int nshards, bool elect_here = false)
    {
        HIPCHK(hipGetDevice(&dev));
        HIPCHK(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
        ND = F.p <= 255 ? 1 : 2;
        xbytes = ND == 1 ? 1 : 2;
        npanel = KB / DP_W;
        Rext = R + KB;
        G = num_cu;
        chunk = (int)((((i64)Rext + G - 1) / G + 63) / 64 * 64);
        if (chunk > 65536) throw EngineError("distributed dense finish: more rows on a shard than the panel kernel takes");
        G = (int)(((i64)Rext + chunk - 1) / chunk);
        Rp = G * chunk;
        const int lds_rows = 147456 / (DP_W * xbytes);
        const char *force_global = getenv("SPASM_AMD_PANEL_GLOBAL");
        inlds = chunk <= lds_rows && !(force_global && atoi(force_global));
        const int app_lds = 147456 - (int)sizeof(int) * (DP_W * DP_W + 3 * DP_W) - 1024;
        app_chunk = std::min((chunk + 63) / 64 * 64, app_lds / (DP_W * xbytes) / 64 * 64);
        G_app = (Rp + app_chunk - 1) / app_chunk;
        Rs = nshards * DP_W;
        if (Rs > lds_rows) throw EngineError("distributed dense finish: more shards than the election workgroup holds candidates for");
        Cp = (int)ldc + 128;
        fplane = (i64)Rp * KB;
        uplane = (i64)Cp * KB;
        P.alloc((size_t)DP_W * (size_t)Rp);
        seq.alloc((size_t)Rp);
        seq_tmp.alloc((size_t)Rp);
        candrow.alloc((size_t)2 * std::max(G, 1) * DP_REC);
        Fd.alloc((size_t)ND * (size_t)fplane);
        Ut.alloc((size_t)ND * (size_t)uplane);
        info.alloc((size_t)npanel);
        info_tmp.alloc(1);
        sync.alloc(1);
        st.alloc(1);
        st_tmp.alloc(1);
        st.zero(s);
        invtab.alloc((size_t)F.p);
        pc_dummy.alloc(DP_W);
        own_pc.alloc((size_t)C + 1);
        own_map.alloc((size_t)KB);
        flag.alloc(1);
        flag.zero(s);
        cand.alloc(1);
        glob.alloc(1);
        expD.alloc((size_t)DP_W * (size_t)ldc);
        expF.alloc((size_t)ND * DP_W * (size_t)KB);
        if (me == 0 || elect_here) {
            stack.alloc((size_t)nshards);
            Ps.alloc((size_t)DP_W * (size_t)Rs);
            seq_s.alloc((size_t)Rs);
            candrow_s.alloc((size_t)2 * DP_REC);
            info_s.alloc(1);
            sync_s.alloc(1);
            st_s.alloc(1);
        }
        hipLaunchKernelGGL(k_inv_table, dim3(cdiv(F.p, 256)), dim3(256), 0, s, (int)F.p, invtab.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemsetAsync(seq.p, 0xff, (size_t)Rp * sizeof(int), s));
        HIPCHK(hipMemsetAsync(own_pc.p, 0xff, ((size_t)C + 1) * sizeof(int), s));
        // (function attributes are per device: every shard sets them on its own)
        HIPCHK(hipFuncSetAttribute((const void *)k_panel_lu<true, 1024, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456));
        HIPCHK(hipFuncSetAttribute((const void *)k_panel_apply<1024, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, app_lds));
    }

    void gemm(int ja, int jb, int k0, int K, const int *rows, int nrows)
    {
        if (jb <= ja || K <= 0) return;
        const int ntm = cdiv(rows ? nrows : Rext, 128);
        if (ND == 1) {
            const int ntn = cdiv(jb - ja, 128);
            hipLaunchKernelGGL((k_gemm_i8<1, 2, 2, 2, 2, DT>), dim3((unsigned)((i64)ntm * ntn)), dim3(256), 0, s, Rext, ja, jb, k0, K, F, D.p, (i64d)ldc, seq.p, rows, nrows,
                               Fd.p, (i64d)fplane, Ut.p, (i64d)uplane, KB, ntm, ntn);
        } else {
            const int ntn = cdiv(jb - ja, 64);
            hipLaunchKernelGGL((k_gemm_i8<2, 4, 1, 1, 2, DT>), dim3((unsigned)((i64)ntm * ntn)), dim3(256), 0, s, Rext, ja, jb, k0, K, F, D.p, (i64d)ldc, seq.p, rows, nrows,
                               Fd.p, (i64d)fplane, Ut.p, (i64d)uplane, KB, ntm, ntn);
        }
    }
    void trsm(int q, int ja, int jb)
    {
        if (jb <= ja) return;
        if (ND == 1) hipLaunchKernelGGL((k_trsm_i8<1, DT>), dim3(cdiv(jb - ja, 64)), dim3(64), 0, s, ja, jb, F, D.p, (i64d)ldc, info.p + q, Ut.p, (i64d)uplane, KB, q * DP_W);
        else hipLaunchKernelGGL((k_trsm_i8<2, DT>), dim3(cdiv(jb - ja, 64)), dim3(64), 0, s, ja, jb, F, D.p, (i64d)ldc, info.p + q, Ut.p, (i64d)uplane, KB, q * DP_W);
    }

    void launch_panel_lu(bool lds_variant, int grid, size_t lds_bytes, int a_Rp, int a_chunk, int a_w, DT *a_P, int *a_seq, int *a_pc, PanelInfo *a_info, PanelSync *a_sy,
                         int *a_cand, DenseState *a_st)
    {
        int a_c0 = 0; // (the column offset only addresses pivrow_of_col, which is a scratch array here)
        ZpField a_F = F;
        const int *a_inv = invtab.p;
        unsigned long long *a_stamps = nullptr;
        void *args[] = {&a_Rp, &a_chunk, &a_w, &a_c0, &a_F, &a_P, &a_seq, &a_pc, &a_info, &a_sy, &a_cand, &a_st, &a_inv, &a_stamps};
        const void *fn = lds_variant ? (const void *)k_panel_lu<true, 1024, DT> : (const void *)k_panel_lu<false, 1024, DT>;
        HIPCHK(hipLaunchKernel(fn, dim3(grid), dim3(1024), args, lds_bytes, s));
    }

    void block_begin()
    {
        HIPCHK(hipMemsetAsync(Fd.p, 0, (size_t)ND * (size_t)fplane, s));
        HIPCHK(hipMemsetAsync(Ut.p, 0, (size_t)ND * (size_t)uplane, s));
        HIPCHK(hipMemsetAsync(D.p + (size_t)R * (size_t)ldc, 0, (size_t)KB * (size_t)ldc * sizeof(DT), s));
        hipLaunchKernelGGL(k_fill_int, dim3(cdiv(KB, 256)), dim3(256), 0, s, KB, DM_RETIRED, seq.p + R);
        hipLaunchKernelGGL(k_fill_int, dim3(cdiv(KB, 256)), dim3(256), 0, s, KB, -1, own_map.p);
        HIPCHK(hipGetLastError());
    }

    // the rows this shard's own elimination of the panel elects, with their panel entries as they are in D -> cand (synchronises)
    void candidates(int c0, int w)
    {
        hipLaunchKernelGGL((k_panel_load<DT>), dim3(Rp / 64), dim3(256), 0, s, Rext, Rp, c0, w, D.p, (i64d)ldc, P.p, sync.p);
        HIPCHK(hipMemcpyAsync(seq_tmp.p, seq.p, (size_t)Rp * sizeof(int), hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(st_tmp.p, st.p, sizeof(DenseState), hipMemcpyDeviceToDevice, s));
        launch_panel_lu(inlds, G, inlds ? (size_t)chunk * DP_W * (size_t)xbytes : 0, Rp, chunk, w, P.p, seq_tmp.p, pc_dummy.p, info_tmp.p, sync.p, candrow.p, st_tmp.p);
        hipLaunchKernelGGL((k_cand_gather<DT>), dim3(16), dim3(256), 0, s, c0, w, D.p, (i64d)ldc, info_tmp.p, cand.p, st_tmp.p, flag.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
    }

    // root: the pivots of the panel among the candidates of all shards -> glob, and its host copy (synchronises)
    void elect(const std::vector<DenseShard<DT> *> &all, int w, PanelGlob *host_glob)
    {
        const int nsh = (int)all.size();
        for (int k = 0; k < nsh; k++) HIPCHK(hipMemcpyAsync(stack.p + k, all[(size_t)k]->cand.p, sizeof(CandRec), hipMemcpyDeviceToDevice, s));
        elect_stacked(nsh, w, host_glob);
    }
    // the same with the candidates of all shards gathered by the caller (a collective): stack_dev holds nsh records
    void elect_from(const void *stack_dev, int nsh, int w, PanelGlob *host_glob)
    {
        HIPCHK(hipMemcpyAsync(stack.p, stack_dev, (size_t)nsh * sizeof(CandRec), hipMemcpyDeviceToDevice, s));
        elect_stacked(nsh, w, host_glob);
    }
    void elect_stacked(int nsh, int w, PanelGlob *host_glob)
    {
        hipLaunchKernelGGL((k_stack_load<DT>), dim3(cdiv((i64)Rs * DP_W, 256)), dim3(256), 0, s, nsh, Rs, stack.p, Ps.p, seq_s.p, sync_s.p, st_s.p);
        HIPCHK(hipGetLastError());
        launch_panel_lu(true, 1, (size_t)Rs * DP_W * (size_t)xbytes, Rs, Rs, w, Ps.p, seq_s.p, pc_dummy.p, info_s.p, sync_s.p, candrow_s.p, st_s.p);
        hipLaunchKernelGGL((k_make_glob<DT>), dim3(1), dim3(1024), 0, s, nsh, Rs, w, F, stack.p, info_s.p, Ps.p, glob.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(host_glob, glob.p, sizeof(PanelGlob), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }

    // the rows this shard won, packed for the others to fetch (synchronises)
    void pack(int c0, const PanelGlob *host_glob, bool glob_is_here = false)
    {
        if (me != 0 && !glob_is_here) HIPCHK(hipMemcpyAsync(glob.p, host_glob, sizeof(PanelGlob), hipMemcpyHostToDevice, s));
        if (host_glob->cnt[me] > 0) {
            hipLaunchKernelGGL((k_export_pack<DT>), dim3(DP_W, (unsigned)std::max(1, std::min(64, cdiv(ldc - c0, 1024)))), dim3(256), 0, s, me, c0, (i64d)ldc, glob.p, D.p, Fd.p,
                               (i64d)fplane, KB, ND, expD.p, expF.p);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(s));
    }

    // the winners of all shards into this shard's guest rows of panel q, then the panel as the single-device finish does it
    void apply(const std::vector<DenseShard<DT> *> &all, int q, int c0, int w, int b1, const PanelGlob *host_glob)
    {
        const int guest0 = R + q * DP_W;
        for (size_t k = 0; k < all.size(); k++) {
            const int cnt = host_glob->cnt[k], first = host_glob->first[k];
            if (cnt <= 0) continue;
            const DenseShard<DT> &o = *all[k];
            HIPCHK(hipMemcpy2DAsync(D.p + (size_t)(guest0 + first) * (size_t)ldc + c0, (size_t)ldc * sizeof(DT), o.expD.p + c0, (size_t)o.ldc * sizeof(DT),
                                    (size_t)(ldc - c0) * sizeof(DT), (size_t)cnt, hipMemcpyDeviceToDevice, s));
            for (int d = 0; d < ND; d++)
                HIPCHK(hipMemcpyAsync(Fd.p + (size_t)d * (size_t)fplane + (size_t)(guest0 + first) * (size_t)KB, o.expF.p + (size_t)d * (size_t)cnt * (size_t)KB,
                                      (size_t)cnt * (size_t)KB, hipMemcpyDeviceToDevice, s));
        }
        apply_panel(q, c0, w, b1);
    }

    // one process per shard: this shard's packed winners (pack) in the caller's buffer -- cnt rows of ldc - c0 elements, then ND planes of
    // cnt rows of KB multiplier digits.  Returns the bytes written.
    i64 export_to(int c0, const PanelGlob *host_glob, void *buf)
    {
        const int cnt = host_glob->cnt[me];
        if (cnt <= 0) return 0;
        const size_t wbytes = (size_t)(ldc - c0) * sizeof(DT);
        HIPCHK(hipMemcpy2DAsync(buf, wbytes, expD.p + c0, (size_t)ldc * sizeof(DT), wbytes, (size_t)cnt, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync((char *)buf + wbytes * (size_t)cnt, expF.p, (size_t)ND * (size_t)cnt * (size_t)KB, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        return (i64)(wbytes * (size_t)cnt + (size_t)ND * (size_t)cnt * (size_t)KB);
    }
    // .. and an owner's buffer (after the broadcast) into this shard's guest rows of panel q
    void import_from(int q, int c0, int owner, const PanelGlob *host_glob, const void *buf)
    {
        const int cnt = host_glob->cnt[owner], first = host_glob->first[owner];
        if (cnt <= 0) return;
        const int guest0 = R + q * DP_W;
        const size_t wbytes = (size_t)(ldc - c0) * sizeof(DT);
        HIPCHK(hipMemcpy2DAsync(D.p + (size_t)(guest0 + first) * (size_t)ldc + c0, (size_t)ldc * sizeof(DT), buf, wbytes, wbytes, (size_t)cnt, hipMemcpyDeviceToDevice, s));
        for (int d = 0; d < ND; d++)
            HIPCHK(hipMemcpyAsync(Fd.p + (size_t)d * (size_t)fplane + (size_t)(guest0 + first) * (size_t)KB, (const char *)buf + wbytes * (size_t)cnt + (size_t)d * (size_t)cnt * (size_t)KB,
                                  (size_t)cnt * (size_t)KB, hipMemcpyDeviceToDevice, s));
    }

    // the panel once the winners sit in the guest rows: apply, store, the pivot rows' triangular solve, the update inside the block
    void apply_panel(int q, int c0, int w, int b1)
    {
        const int c1 = c0 + w;
        const int guest0 = R + q * DP_W;
        hipLaunchKernelGGL(k_apply_prep, dim3(1), dim3(DP_W), 0, s, me, R, q, glob.p, seq.p, own_map.p);
        hipLaunchKernelGGL((k_panel_load<DT>), dim3(Rp / 64), dim3(256), 0, s, Rext, Rp, c0, w, D.p, (i64d)ldc, P.p, sync.p);
        hipLaunchKernelGGL((k_panel_apply<1024, DT>), dim3(G_app), dim3(1024), (size_t)app_chunk * DP_W * (size_t)xbytes, s, Rp, app_chunk, w, F, P.p, seq.p, glob.p, me,
                           guest0, st.p, flag.p);
        hipLaunchKernelGGL(k_apply_info, dim3(1), dim3(DP_W), 0, s, me, guest0, c0, glob.p, info.p + q, st.p, own_pc.p);
        HIPCHK(hipGetLastError());
        if (ND == 1) hipLaunchKernelGGL((k_panel_store<1, DT>), dim3(Rp / 64), dim3(256), 0, s, Rext, Rp, c0, w, F, P.p, seq.p, D.p, (i64d)ldc, info.p + q, Fd.p, (i64d)fplane, KB, q * DP_W);
        else hipLaunchKernelGGL((k_panel_store<2, DT>), dim3(Rp / 64), dim3(256), 0, s, Rext, Rp, c0, w, F, P.p, seq.p, D.p, (i64d)ldc, info.p + q, Fd.p, (i64d)fplane, KB, q * DP_W);
        trsm(q, c1, b1);
        gemm(c1, b1, q * DP_W, DP_W, nullptr, 0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s)); // (the owners' export buffers are rewritten by the next panel)
    }

    void block_end(int b0, int b1, int npan)
    {
        for (int t = 0; t < npan; t++) {
            if (t > 0) gemm(b1, C, 0, t * DP_W, (const int *)((const char *)(info.p + t) + offsetof(PanelInfo, row)), DP_W);
            trsm(t, b1, C);
        }
        gemm(b1, C, 0, npan * DP_W, nullptr, 0);
        hipLaunchKernelGGL((k_guest_copyback<DT>), dim3((unsigned)KB, (unsigned)std::max(1, std::min(64, cdiv(ldc - b0, 1024)))), dim3(256), 0, s, R, b0, (i64d)ldc, own_map.p, D.p);
        HIPCHK(hipGetLastError());
    }

    // after the last block: did anything go wrong on the device?  Returns the number of pivots (all shards hold the same count)
    int finish()
    {
        DenseState hst;
        int hflag = 0;
        HIPCHK(hipMemcpyAsync(&hst, st.p, sizeof hst, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(&hflag, flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (hflag & 2) throw EngineError("distributed dense finish: a grid barrier of the panel kernel timed out (the device is shared with another process?)");
        if (hflag & 1) throw EngineError("distributed dense finish: a row held an entry in a column no candidate covered");
        return hst.npiv;
    }
};

// the distributed elimination of the shards' dense matrices; every shard's pivot rows (its own rows of D) are appended to U.
// clist (device, on every shard's device): column of D -> column of the matrix.  Returns the pivots found.
template <typename DT>
int dense_finish_multi(std::vector<std::unique_ptr<DenseShard<DT>>> &sh, const std::vector<int> &dev, const std::vector<const int *> &clist, HostU &U)
{
    const int nsh = (int)sh.size();
    const double t0 = spasm_wtime();
    std::vector<DenseShard<DT> *> all;
    for (auto &q : sh) all.push_back(q.get());
    const int C = sh[0]->C;
    const int KB = sh[0]->KB;
    TeamBarrier bar(nsh);
    TeamError err;
    PanelGlob host_glob;
    memset(&host_glob, 0, sizeof host_glob);
    auto body = [&](int k) {
        DenseShard<DT> &me = *all[(size_t)k];
        err.guard([&] {
            HIPCHK(hipSetDevice(dev[(size_t)k]));
            me.setup(nsh);
        });
        bar.wait();
        for (int b0 = 0; b0 < C; b0 += KB) {
            const int b1 = std::min(b0 + KB, C);
            err.guard([&] { me.block_begin(); });
            int q = 0;
            for (int c0 = b0; c0 < b1; c0 += DP_W, q++) {
                const int w = std::min(c0 + DP_W, b1) - c0;
                err.guard([&] { me.candidates(c0, w); });
                bar.wait();
                if (k == 0) err.guard([&] { me.elect(all, w, &host_glob); });
                bar.wait();
                err.guard([&] { me.pack(c0, &host_glob); });
                bar.wait();
                err.guard([&] { me.apply(all, q, c0, w, b1, &host_glob); });
                bar.wait();
            }
            err.guard([&] { me.block_end(b0, b1, q); });
        }
        err.guard([&] { HIPCHK(hipStreamSynchronize(me.s)); });
        bar.wait();
    };
    {
        std::vector<std::thread> th;
        for (int k = 1; k < nsh; k++) th.emplace_back(body, k);
        body(0);
        for (auto &t : th) t.join();
    }
    if (err.failed.load()) throw EngineError(err.what);
    const double t1 = spasm_wtime();
    int npiv = -1, got = 0;
    i64 rows = 0;
    for (int k = 0; k < nsh; k++) {
        HIPCHK(hipSetDevice(dev[(size_t)k]));
        DenseShard<DT> &q = *all[(size_t)k];
        const int np = q.finish();
        if (npiv >= 0 && np != npiv) throw EngineError("distributed dense finish: the shards disagree on the number of pivots");
        npiv = np;
        rows += q.R;
        got += dense_extract_U(q.D.p, C, q.ldc, q.own_pc.p, clist[(size_t)k], q.row_orig.p, U, q.s);
    }
    if (got != npiv) throw EngineError("distributed dense finish: the shards do not own the pivot rows between them");
    spasm_logf("[echelonize/dense] %lld x %d dense tail over %d row shards: %d pivots [elimination %.2fs, rows of U to the host %.2fs]\n", (long long)rows, C, nsh, npiv,
               t1 - t0, spasm_wtime() - t1);
    return npiv;
}

// the live part of a shard's sparse matrix as a dense matrix (what run_dense_tail does on one device), the columns numbered alike
// on all shards: flag_columns(), OR of the flags across the shards, finish_columns(), build()
struct DenseFill {
    const DevMat &M;
    hipStream_t s;
    Scanner scan;
    DevBuf<int> rflag, rscan, rows, cflag, cscan, cmap, clist;
    int R = 0, C = 0;
    DenseFill(const DevMat &M_, hipStream_t s_) : M(M_), s(s_) {}
    void flag_columns()
    {
        const int n = M.n, m = M.m;
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
        HIPCHK(hipMemcpyAsync(&R, rscan.p + n, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    int finish_columns()
    {
        const int m = M.m;
        scan.exclusive(cflag.p, cscan.p, (size_t)m + 1, s);
        HIPCHK(hipMemcpyAsync(&C, cscan.p + m, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (C > 0) {
            hipLaunchKernelGGL(k_col_map, dim3(cdiv(m, 256)), dim3(256), 0, s, m, cflag.p, cscan.p, cmap.p, clist.p);
            HIPCHK(hipGetLastError());
        }
        return C;
    }
    template <typename DT> void build(int extra_rows, DevBuf<DT> &D, DevBuf<int> &row_orig)
    {
        const i64 ldc = ((i64)C + 63) / 64 * 64;
        D.alloc(((size_t)R + (size_t)extra_rows) * (size_t)ldc);
        D.zero(s);
        row_orig.alloc((size_t)R + 1);
        if (R > 0) {
            hipLaunchKernelGGL(k_gather_int, dim3(cdiv(R, 256)), dim3(256), 0, s, R, rows.p, M.orig.p, row_orig.p);
            hipLaunchKernelGGL((k_dense_fill<DT>), dim3(cdiv((i64)R * 64, 256)), dim3(256), 0, s, R, rows.p, M.start.p, M.len.p, M.ent.p, cmap.p, D.p, (i64d)ldc);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(s));
    }
};

// the per-shard matrices built by build(k, extra_rows, D, row_orig, R) on their devices, then eliminated together
template <typename DT, class Build>
int dense_multi_run(int nshards, const std::vector<int> &dev, const ZpField &F, int C, const std::vector<const int *> &clist, Build &&build, HostU &U)
{
    const int KB = dense_kb();
    std::vector<std::unique_ptr<DenseShard<DT>>> sh((size_t)nshards);
    parallel_shards(nshards, [&](int k) {
        HIPCHK(hipSetDevice(dev[(size_t)k]));
        sh[(size_t)k].reset(new DenseShard<DT>());
        DenseShard<DT> &q = *sh[(size_t)k];
        q.me = k;
        q.C = C;
        q.ldc = ((i64)C + 63) / 64 * 64;
        q.F = F;
        q.KB = KB;
        build(k, KB, q.D, q.row_orig, q.R);
    });
    return dense_finish_multi(sh, dev, clist, U);
}

// flags[k] (m + 1 ints on device dev[k]) OR-ed over the shards, the result on every shard
inline void or_flags_across(const std::vector<int> &dev, const std::vector<int *> &flags, int m)
{
    const int nsh = (int)flags.size();
    if (nsh <= 1) return;
    HIPCHK(hipSetDevice(dev[0]));
    DevBuf<int> stage;
    stage.alloc((size_t)m + 1);
    for (int k = 1; k < nsh; k++) {
        HIPCHK(hipMemcpy(stage.p, flags[(size_t)k], ((size_t)m + 1) * sizeof(int), hipMemcpyDeviceToDevice));
        hipLaunchKernelGGL(k_or_int, dim3(cdiv((i64)m + 1, 256)), dim3(256), 0, nullptr, m + 1, flags[0], stage.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
    }
    for (int k = 1; k < nsh; k++) {
        HIPCHK(hipSetDevice(dev[(size_t)k]));
        HIPCHK(hipMemcpy(flags[(size_t)k], flags[0], ((size_t)m + 1) * sizeof(int), hipMemcpyDeviceToDevice));
    }
}
