// Tall-and-skinny dense finish -- host side (included by engine.hip inside its anonymous namespace; kernels and the algorithm:
// dense.hpp, section "Tall-and-skinny finish").  libspasm: enable_tall_and_skinny / tall_and_skinny_ratio (reference
// src/SpaSM.jl:327, :341; the strategies behind them, spasm_schur_dense_randomized and friends, prototypes :767-769).
//
// RowSource: where the dense rows come from -- fill(off, cnt, Dp) writes rows off .. off + cnt of the remainder, all C columns,
// into Dp (cnt x ldc, zero on entry): the Schur rows of a round through the dense W (SchurRowSource) or the live rows of the
// current matrix (FillRowSource).  Rows are only ever materialised a slab at a time.  For the rows beyond the first slab the source
// is asked column slab by column slab (nslabs, slab_range, prepare_slab -- the dense W of the slab, built once --, fill_slab).
#pragma once

struct TallStats {
    int R1 = 0, r1 = 0, f = 0, r2 = 0;
    double t_slab = 0, t_z = 0, t_resid = 0, t_tail = 0;
};

template <typename DT> struct TallWork {
    const ZpField &F;
    hipStream_t s;
    int ND, KB = 1024;
    DevBuf<signed char> Fd, Ut;
    DevBuf<int> live;   // seq of the GEMM: all -1
    i64 fplane = 0, uplane = 0;
    int rows_cap = 0, cols_cap = 0;
    TallWork(const ZpField &F_, hipStream_t s_) : F(F_), s(s_), ND(F_.p <= 255 ? 1 : 2) {}
    void ensure(int rows, int ncols)
    {
        const int rp = (rows + 127) / 128 * 128, cp = (ncols + 63) / 64 * 64 + 128;
        if (rp > rows_cap) {
            rows_cap = rp;
            fplane = (i64)rp * KB;
            Fd.alloc((size_t)ND * (size_t)fplane);
            live.alloc((size_t)rp);
            HIPCHK(hipMemsetAsync(live.p, 0xff, (size_t)rp * sizeof(int), s));
        }
        if (cp > cols_cap) {
            cols_cap = cp;
            uplane = (i64)cp * KB;
            Ut.alloc((size_t)ND * (size_t)uplane);
        }
    }
    // T[i][0 .. ncols) -= sum_k Src[rowidx ? rowidx[i] : i][cols[k]] * Z[k][0 .. ncols), i < nrows, k < K (K <= KB)
    void gemm_sub(DT *T, i64 ldt, int nrows, const DT *Src, i64 lds, const int *rowidx, const int *cols, int K, const DT *Z, i64 ldz, int ncols, int col_off = 0)
    {
        if (nrows <= 0 || K <= 0 || ncols <= 0) return;
        ensure(nrows, ncols);
        const int Kpad = (K + 63) / 64 * 64;
        const int ncp = (ncols + 63) / 64 * 64 + 128;
        if (ND == 1) {
            hipLaunchKernelGGL((k_tall_gather_F<1, DT>), dim3(nrows), dim3(256), 0, s, nrows, rowidx, Src, (i64d)lds, cols, K, Kpad, F, Fd.p, (i64d)fplane, KB, col_off);
            hipLaunchKernelGGL((k_tall_Ut<1, DT>), dim3(ncp), dim3(256), 0, s, Z, (i64d)ldz, K, Kpad, ncols, ncp, F, Ut.p, (i64d)uplane, KB);
            const int ntm = cdiv(nrows, 128), ntn = cdiv(ncols, 128);
            hipLaunchKernelGGL((k_gemm_i8<1, 2, 2, 2, 2, DT>), dim3((unsigned)((i64)ntm * ntn)), dim3(256), 0, s, nrows, 0, ncols, 0, Kpad, F, T, (i64d)ldt, live.p, (const int *)nullptr, 0,
                               Fd.p, (i64d)fplane, Ut.p, (i64d)uplane, KB, ntm, ntn);
        } else {
            hipLaunchKernelGGL((k_tall_gather_F<2, DT>), dim3(nrows), dim3(256), 0, s, nrows, rowidx, Src, (i64d)lds, cols, K, Kpad, F, Fd.p, (i64d)fplane, KB, col_off);
            hipLaunchKernelGGL((k_tall_Ut<2, DT>), dim3(ncp), dim3(256), 0, s, Z, (i64d)ldz, K, Kpad, ncols, ncp, F, Ut.p, (i64d)uplane, KB);
            const int ntm = cdiv(nrows, 128), ntn = cdiv(ncols, 64);
            hipLaunchKernelGGL((k_gemm_i8<2, 4, 1, 1, 2, DT>), dim3((unsigned)((i64)ntm * ntn)), dim3(256), 0, s, nrows, 0, ncols, 0, Kpad, F, T, (i64d)ldt, live.p, (const int *)nullptr, 0,
                               Fd.p, (i64d)fplane, Ut.p, (i64d)uplane, KB, ntm, ntn);
        }
        HIPCHK(hipGetLastError());
    }
};

// Z = the reduced form of r1 echelon rows on f columns without pivot: row t of the echelon form is row prow[t] of D (leading
// dimension ldc), its pivot (a 1) on column pcol[t], and it holds nothing on the pivot columns of the rows BEFORE it (pcol[s], s < t);
// Z_t = u_t[fcol] - sum_{s > t} u_t[pcol_s] Z_s.  A blocked back substitution: blocks of 1024 pivots against everything behind them on
// the int8 GEMM, 64 at a time inside a block.  (Also the dense part of spasm_kernel's basis: engine.hip, kernel_dense_rhs.)
template <typename DT>
void tall_reduced_form(TallWork<DT> &W, const DT *D, i64 ldc, int r1, const int *prow, const int *pcol, const int *fcol, int f, DT *Z, i64 ldz, const ZpField &F, hipStream_t s)
{
    if (r1 <= 0 || f <= 0) return;
    hipLaunchKernelGGL((k_tall_gather_cols<DT>), dim3(r1), dim3(256), 0, s, r1, prow, D, (i64d)ldc, fcol, f, Z, (i64d)ldz);
    HIPCHK(hipGetLastError());
    const int OB = W.KB, IB = 64;
    for (int b1 = r1; b1 > 0;) {
        const int b0 = std::max(0, (b1 - 1) / OB * OB), nb = b1 - b0;
        // rows b0 .. b1 against everything behind the block
        for (int s0 = b1; s0 < r1; s0 += OB)
            W.gemm_sub(Z + (size_t)b0 * (size_t)ldz, ldz, nb, D, ldc, prow + b0, pcol + s0, std::min(OB, r1 - s0), Z + (size_t)s0 * (size_t)ldz, ldz, f);
        // inside the block: 64 pivots at a time, the later ones of the block through the GEMM, then the back substitution
        for (int i1 = b1; i1 > b0;) {
            const int i0 = std::max(b0, (i1 - 1) / IB * IB), ni = i1 - i0;
            if (i1 < b1) W.gemm_sub(Z + (size_t)i0 * (size_t)ldz, ldz, ni, D, ldc, prow + i0, pcol + i1, b1 - i1, Z + (size_t)i1 * (size_t)ldz, ldz, f);
            hipLaunchKernelGGL((k_tall_backsub<DT>), dim3(cdiv(f, 16)), dim3(256), 0, s, i0, ni, f, F, D, (i64d)ldc, prow, pcol, Z, (i64d)ldz);
            i1 = i0;
        }
        HIPCHK(hipGetLastError());
        b1 = b0;
    }
}

// rows per slab that is eliminated whole: enough to carry every pivot a remainder of C columns can have, with some slack for
// dependent rows.  SPASM_AMD_TALL_SLAB (tests): rows of the first slab.
inline int tall_first_slab(int R, int C)
{
    i64 R1 = (i64)C + C / 8 + 1024;
    if (const char *e = getenv("SPASM_AMD_TALL_SLAB")) R1 = std::max<i64>(64, atoll(e));
    R1 = (R1 + 63) / 64 * 64;
    return (int)std::min<i64>(R1, R);
}

// does the tall-and-skinny finish apply?  libspasm: enable_tall_and_skinny and rows / columns above tall_and_skinny_ratio
inline bool tall_applies(const struct echelonize_opts *opts, const ZpField &F, i64 R, i64 C)
{
    if (!opts || !opts->enable_tall_and_skinny || !F.small || C <= 0) return false;
    const char *e = getenv("SPASM_AMD_TALL"); // 0: never, 1: whenever there are rows beyond the first slab (tests)
    if (e && atoi(e) == 0) return false;
    if (tall_first_slab((int)std::min<i64>(R, INT_MAX), (int)C) >= R) return false;
    if (e && atoi(e) == 1) return true;
    const double ratio = opts->tall_and_skinny_ratio > 0 ? opts->tall_and_skinny_ratio : 5.0;
    return (double)R > ratio * (double)C;
}

template <typename DT, class RowSource>
int dense_finish_tall(RowSource &src, int R, int C, i64 ldc, const int *clist, const int *row_orig, const ZpField &F, HostU &U, hipStream_t s)
{
    TallStats ts;
    const double t0 = spasm_wtime();
    const int R1 = tall_first_slab(R, C);
    ts.R1 = R1;
    // ---- 1. the first slab, eliminated as a whole
    DevBuf<DT> D1;
    DevBuf<int> pc1;
    D1.alloc((size_t)R1 * (size_t)ldc);
    D1.zero(s);
    pc1.alloc((size_t)C + 1);
    src.fill(0, R1, D1.p);
    HIPCHK(hipStreamSynchronize(s));
    const double tf1 = spasm_wtime() - t0;
    if ((double)R1 * (double)C > 4e9) spasm_logf("[echelonize/dense] tall and skinny: %d x %d; first slab of %d rows built [%.2fs], eliminating\n", R, C, R1, tf1);
    UStreamer us; // (the slab's rows of U leave for the host block by block while it is eliminated)
    us.init(D1.p, C, ldc, pc1.p, clist, row_orig, U);
    if (!dense_eliminate_i8(D1, R1, C, ldc, F, pc1, s, &us)) throw EngineError("dense finish: shape outside the panel kernel's range");
    HIPCHK(hipStreamSynchronize(s));
    const double te1 = spasm_wtime() - t0 - tf1;
    us.template take<DT>(C);
    const int r1 = us.found;
    ts.r1 = r1;
    ts.t_slab = spasm_wtime() - t0;
    const double tu1 = ts.t_slab - tf1 - te1;
    const int R2 = R - R1;
    const int f = C - r1;
    ts.f = f;
    if (R2 <= 0 || f <= 0) {
        src.done();
        spasm_logf("[echelonize/dense] tall and skinny: %d x %d, first slab of %d rows: %d pivots%s [rows %.2fs, elimination with the rows of U leaving block by block %.2fs, the last block's %.2fs]\n", R, C,
                   R1, r1, f <= 0 && R2 > 0 ? " = every column: the other rows cannot add any" : "", tf1, te1, tu1);
        return r1;
    }
    // ---- 2. columns with / without pivot; Z = the reduced form of the slab's pivot rows on the columns without
    const double t1 = spasm_wtime();
    if ((double)R1 * (double)C > 4e9) spasm_logf("[echelonize/dense] tall and skinny: first slab: %d pivots [%.2fs]; reduced form on the %d columns left\n", r1, te1, f);
    Scanner scan;
    DevBuf<int> pflag, pscan, pcol, prow, fcol, fclist;
    pflag.alloc((size_t)C + 1); pscan.alloc((size_t)C + 1); pcol.alloc((size_t)r1 + 1); prow.alloc((size_t)r1 + 1); fcol.alloc((size_t)f + 1); fclist.alloc((size_t)f + 1);
    hipLaunchKernelGGL(k_flag_nonneg, dim3(cdiv((i64)C + 1, 256)), dim3(256), 0, s, C, pc1.p, pflag.p);
    HIPCHK(hipGetLastError());
    scan.exclusive(pflag.p, pscan.p, (size_t)C + 1, s);
    hipLaunchKernelGGL(k_tall_split_cols, dim3(cdiv(C, 256)), dim3(256), 0, s, C, pc1.p, pscan.p, pcol.p, prow.p, fcol.p);
    hipLaunchKernelGGL(k_gather_int2, dim3(cdiv(f, 256)), dim3(256), 0, s, f, fcol.p, clist, fclist.p);
    HIPCHK(hipGetLastError());
    const i64 ldz = ((i64)f + 63) / 64 * 64;
    DevBuf<DT> Z;
    Z.alloc((size_t)std::max(r1, 1) * (size_t)ldz);
    TallWork<DT> W(F, s);
    if (r1 > 0) tall_reduced_form<DT>(W, D1.p, ldc, r1, prow.p, pcol.p, fcol.p, f, Z.p, ldz, F, s);
    HIPCHK(hipStreamSynchronize(s));
    D1.release();
    ts.t_z = spasm_wtime() - t1;
    // ---- 3. every other row in one step: t = d_N - d_P Z.  Column slab by column slab of the source (the dense W of a slab is built
    // once: config 5 at full size has six, and rebuilding them for every batch of rows took 205 of the run's 268 s), a batch of rows
    // at a time inside; Tp collects the residuals of the rows [row0, row0 + nrows) of the rows beyond the slab.
    const double t2 = spasm_wtime();
    std::vector<int> h_pcol((size_t)std::max(r1, 1)), h_fcol((size_t)std::max(f, 1));
    if (r1 > 0) HIPCHK(hipMemcpyAsync(h_pcol.data(), pcol.p, (size_t)r1 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_fcol.data(), fcol.p, (size_t)f * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    double tf2 = 0;
    const int nsl = src.nslabs();
    DevBuf<DT> Db;
    auto reduce_rows = [&](i64 row0, i64 nrows, DT *Tp) {
        for (int k = 0; k < nsl; k++) {
            i64 s0 = 0;
            int w = 0;
            src.slab_range(k, s0, w);
            const int pa = (int)(std::lower_bound(h_pcol.begin(), h_pcol.begin() + r1, (int)s0) - h_pcol.begin());
            const int pb = (int)(std::lower_bound(h_pcol.begin(), h_pcol.begin() + r1, (int)std::min<i64>(s0 + w, INT_MAX)) - h_pcol.begin());
            const int fa = (int)(std::lower_bound(h_fcol.begin(), h_fcol.begin() + f, (int)s0) - h_fcol.begin());
            const int fb = (int)(std::lower_bound(h_fcol.begin(), h_fcol.begin() + f, (int)std::min<i64>(s0 + w, INT_MAX)) - h_fcol.begin());
            if (pa == pb && fa == fb) continue;
            const double tb = spasm_wtime();
            src.prepare_slab(k);
            HIPCHK(hipStreamSynchronize(s));
            tf2 += spasm_wtime() - tb;
            size_t fr = 0, tot = 0;
            HIPCHK(hipMemGetInfo(&fr, &tot));
            // (at most 128k rows at a time: a batch buffer of tens of GB costs more to allocate and to clear than the launches of more batches)
            i64 RB = std::min<i64>(131072, std::max<i64>(4096, (i64)((fr + Db.n * sizeof(DT)) / 4) / ((i64)w * (i64)sizeof(DT)) / 128 * 128));
            if (const char *e = getenv("SPASM_AMD_TALL_BATCH")) RB = std::max<i64>(128, atoll(e) / 128 * 128); // tests: several batches
            RB = std::min<i64>(RB, (nrows + 127) / 128 * 128);
            Db.ensure((size_t)RB * (size_t)w);
            for (i64 off = 0; off < nrows; off += RB) {
                const int cnt = (int)std::min<i64>(RB, nrows - off);
                const double tc = spasm_wtime();
                HIPCHK(hipMemsetAsync(Db.p, 0, (size_t)cnt * (size_t)w * sizeof(DT), s));
                src.fill_slab(k, R1 + (int)(row0 + off), cnt, Db.p, (i64)w);
                HIPCHK(hipStreamSynchronize(s));
                tf2 += spasm_wtime() - tc;
                DT *Tb = Tp + (size_t)off * (size_t)ldz;
                if (fb > fa) {
                    hipLaunchKernelGGL((k_tall_gather_slab<DT>), dim3(cnt), dim3(256), 0, s, cnt, Db.p, (i64d)w, fcol.p + fa, fb - fa, (int)s0, F, Tb, (i64d)ldz, fa);
                    HIPCHK(hipGetLastError());
                }
                for (int c0 = pa; c0 < pb; c0 += W.KB)
                    W.gemm_sub(Tb, ldz, cnt, Db.p, (i64)w, nullptr, pcol.p + c0, std::min(W.KB, pb - c0), Z.p + (size_t)c0 * (size_t)ldz, ldz, f, (int)s0);
            }
            if ((double)R1 * (double)C > 4e9) {
                HIPCHK(hipStreamSynchronize(s));
                spasm_logf("[echelonize/dense] tall and skinny: column slab %d of %d applied to %lld other rows [%.1fs]\n", k + 1, nsl, (long long)nrows, spasm_wtime() - t2);
            }
        }
        HIPCHK(hipStreamSynchronize(s));
    };
    // Do the residuals of ALL other rows fit (R2 x f: small when the slab carried most pivots)?  A weak first slab -- many dependent
    // rows among its R1 -- leaves f large (a planted-rank matrix at config 5's size: 168 GiB); then the rows go chunk by chunk, and
    // the echelon rows E found so far ride on top of every chunk (they win their columns again: first rows, distinct leading
    // columns), so that each chunk's elimination leaves the echelon form of everything seen.  The dense W of the column slabs is then
    // rebuilt per chunk: the price of not fitting.
    size_t fr0 = 0, tot0 = 0;
    HIPCHK(hipMemGetInfo(&fr0, &tot0));
    // per row of T: its residuals, the elimination's multiplier planes (1 KB per digit), its panel entries and status word; 15 % of
    // what is free stays for the batch buffer and the rest (the dense W of a column slab is allocated already)
    const double per_row = (double)ldz * sizeof(DT) + 1024.0 * W.ND + 64.0 * sizeof(DT) + 8.0;
    i64 cap_rows = (i64)((double)fr0 * 0.85 / per_row);
    const char *chunk_env = getenv("SPASM_AMD_TALL_CHUNK"); // tests: rows of the other rows per chunk
    if (chunk_env) cap_rows = 0;
    int r2 = 0;
    int nchunks = 1;
    if ((i64)R2 <= cap_rows) {
        DevBuf<DT> T;
        T.alloc((size_t)R2 * (size_t)ldz);
        T.zero(s);
        reduce_rows(0, R2, T.p);
        Db.release();
        Z.release();
        src.done();
        ts.t_resid = spasm_wtime() - t2;
        // ---- 4. what the residuals still hold
        DevBuf<int> pc2;
        pc2.alloc((size_t)f + 1);
        if (!dense_eliminate_i8(T, R2, f, ldz, F, pc2, s)) throw EngineError("dense finish: shape outside the panel kernel's range");
        HIPCHK(hipStreamSynchronize(s));
        r2 = dense_extract_U(T.p, f, ldz, pc2.p, fclist.p, row_orig + R1, U, s);
    } else {
        // (E beside T, and its f rows ride along in T)
        i64 Rc = chunk_env ? std::max<i64>(64, atoll(chunk_env)) : (i64)(((double)fr0 * 0.85 - (double)f * (double)ldz * sizeof(DT)) / per_row) - f;
        if (Rc < 1024 && !chunk_env)
            throw EngineError("dense finish: out of device memory: the residuals on " + std::to_string(f) + " columns do not fit even in chunks of rows");
        Rc = (Rc + 63) / 64 * 64;
        nchunks = (int)(((i64)R2 + Rc - 1) / Rc);
        DevBuf<DT> T, E;
        DevBuf<int> origT, origE, pc2, pcE, p2flag, p2scan;
        T.alloc((size_t)(Rc + f) * (size_t)ldz);
        E.alloc((size_t)f * (size_t)ldz);
        origT.alloc((size_t)(Rc + f) + 1); origE.alloc((size_t)f + 1); pc2.alloc((size_t)f + 1); pcE.alloc((size_t)f + 1);
        p2flag.alloc((size_t)f + 1); p2scan.alloc((size_t)f + 1);
        HIPCHK(hipMemsetAsync(pcE.p, 0xff, ((size_t)f + 1) * sizeof(int), s));
        int nE = 0;
        double t_el = 0;
        for (i64 off0 = 0; off0 < R2 && nE < f; off0 += Rc) {
            const i64 cnt = std::min<i64>(Rc, (i64)R2 - off0);
            HIPCHK(hipMemsetAsync(T.p, 0, (size_t)(nE + cnt) * (size_t)ldz * sizeof(DT), s));
            if (nE > 0) {
                HIPCHK(hipMemcpyAsync(T.p, E.p, (size_t)nE * (size_t)ldz * sizeof(DT), hipMemcpyDeviceToDevice, s));
                HIPCHK(hipMemcpyAsync(origT.p, origE.p, (size_t)nE * sizeof(int), hipMemcpyDeviceToDevice, s));
            }
            HIPCHK(hipMemcpyAsync(origT.p + nE, row_orig + R1 + off0, (size_t)cnt * sizeof(int), hipMemcpyDeviceToDevice, s));
            reduce_rows(off0, cnt, T.p + (size_t)nE * (size_t)ldz);
            const double te = spasm_wtime();
            if (!dense_eliminate_i8(T, (int)(nE + cnt), f, ldz, F, pc2, s)) throw EngineError("dense finish: shape outside the panel kernel's range");
            hipLaunchKernelGGL(k_flag_nonneg, dim3(cdiv((i64)f + 1, 256)), dim3(256), 0, s, f, pc2.p, p2flag.p);
            HIPCHK(hipGetLastError());
            scan.exclusive(p2flag.p, p2scan.p, (size_t)f + 1, s);
            hipLaunchKernelGGL((k_tall_compact<DT>), dim3(f), dim3(256), 0, s, f, (i64d)ldz, pc2.p, p2scan.p, T.p, origT.p, E.p, origE.p, pcE.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(&nE, p2scan.p + f, sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            t_el += spasm_wtime() - te;
            spasm_logf("[echelonize/dense] tall and skinny: the residuals do not fit at once: chunk of %lld rows (%lld of %d done): %d pivots so far%s [%.1fs]\n", (long long)cnt,
                       (long long)(off0 + cnt), R2, nE, nE == f ? " = every column: the other rows cannot add any" : "", spasm_wtime() - t2);
        }
        Db.release();
        Z.release();
        T.release();
        src.done();
        ts.t_resid = spasm_wtime() - t2 - t_el;
        r2 = dense_extract_U(E.p, f, ldz, pcE.p, fclist.p, origE.p, U, s);
    }
    ts.r2 = r2;
    ts.t_tail = spasm_wtime() - t2 - ts.t_resid;
    spasm_logf("[echelonize/dense] tall and skinny: %d x %d; first slab of %d rows: %d pivots [rows %.2fs, elimination with the rows of U leaving block by block %.2fs, the last block's %.2fs]; reduced form on the "
               "%d columns left [%.2fs]; %d rows reduced in %d step%s [rows %.2fs, reduction %.2fs]; their residuals: %d pivots [%.2fs]\n", R, C, R1, r1, tf1, te1, tu1, f, ts.t_z,
               R2, nchunks, nchunks > 1 ? "s" : "", tf2, ts.t_resid - tf2, r2, ts.t_tail);
    return r1 + r2;
}
