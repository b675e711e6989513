// kernel_dense.hpp -- device side of spasm_kernel through a DENSE right-hand side (engine.hip: kernel_dense_rhs; reference call site
// src/SpaSM.jl:876-882, known answers test/runtests.jl:20-23).  Included by engine.hip inside its anonymous namespace.
//
// The kernel vector of free column j is  k = -e_j + sum_a y_a e_{pivcol(a)}  with  y_a + sum_{b != a} U[a][pivcol(b)] y_b = U[a][j]:
// a triangular solve in the topological order of the pivots (a row only refers to pivots behind it).  spasm_kernel's first
// implementation runs it with the kernels of a Schur round, one sparse "row" per free column, and needs U as one round on the
// device (32-bit offsets: 2^32 entries).  Here ALL free columns are solved at once as the columns of a dense matrix
//        Y[t][i] = y of the pivot at position t for free column i          (r x nf residues, bytes for p < 2^8)
// filled from the last position to the first:
//   * the DENSE TAIL of U (the rows of a dense finish: position t0 on) is a dense unit-triangular system -- exactly the reduced form
//     Z the tall-and-skinny finish computes for its slab (dense_tall.hpp: tall_reduced_form, on the int8 GEMM);
//   * the rows before it are sparse: level by level of their pivot graph, a workgroup per row,
//     Y[t] = U[t][free columns] - sum over the entries (c, v) of row t on other pivot columns of v * Y[pos(c)];
//   * K is the transpose of Y, compacted.
// U never exists on the device as a whole: the dense tail arrives chunk by chunk and lands in a dense block, the sparse rows are a
// fraction of it.

__global__ void k_iota_from(int n, int base, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = base + i;
}

// ---- the dense tail: entries of rows [k0, k0 + nk) of the block (row pointers relative to the chunk) into D[k][col]:
// a pivot column -> the position of its row minus t0, a selected free column -> rd + its index, anything else is dropped
template <typename DT>
__global__ __launch_bounds__(256) void k_kd_dense_rows(int nk, int k0, const i64d *__restrict__ ptr, const int *__restrict__ cj, const int *__restrict__ cx,
                                                       const int *__restrict__ qinv, const int *__restrict__ pos_of_row, const int *__restrict__ fidx, int t0, int rd,
                                                       DT *__restrict__ D, i64d ldc)
{
    const int k = blockIdx.x;
    if (k >= nk) return;
    const i64d lo = ptr[k], hi = ptr[k + 1];
    DT *row = D + (i64d)(k0 + k) * ldc;
    for (i64d e = lo + threadIdx.x; e < hi; e += 256) {
        const int c = cj[e];
        const int b = qinv[c];
        int col = -1;
        if (b >= 0) col = pos_of_row[b] - t0;
        else if (fidx[c] >= 0) col = rd + fidx[c];
        if (col >= 0) row[col] = (DT)cx[e];
    }
}

// ---- one level of the sparse rows.  rows[0 .. nrows): positions t; the entries of position t are ent[ptr[t] .. ptr[t + 1]) =
// (code, value): code >= 0 -> the position of the pivot the entry refers to, code < 0 -> free column -1 - code.
// grid (rows of the level, tiles of 1024 columns); a thread owns four consecutive columns of Y.
template <typename DT>
__global__ __launch_bounds__(256) void k_kd_level(int nrows, const int *__restrict__ rows, const i64d *__restrict__ ptr, const int2 *__restrict__ ent, ZpField F,
                                                  DT *__restrict__ Y, i64d ldz, int nf)
{
    __shared__ int2 s_e[256];
    const int t = rows[blockIdx.x];
    const int c0 = ((int)blockIdx.y * 256 + (int)threadIdx.x) * 4;
    const i64d lo = ptr[t], hi = ptr[t + 1];
    int acc[4] = {0, 0, 0, 0};
    for (i64d e0 = lo; e0 < hi; e0 += 256) {
        const int n = (int)min((i64d)256, hi - e0);
        __syncthreads();
        if ((int)threadIdx.x < n) s_e[threadIdx.x] = ent[e0 + threadIdx.x];
        __syncthreads();
        if (c0 >= nf) continue;
        for (int q = 0; q < n; q++) {
            const int2 e = s_e[q];
            if (e.x >= 0) {
                const DT *src = Y + (i64d)e.x * ldz + c0;
                const int nv = -e.y;
#pragma unroll
                for (int u = 0; u < 4; u++) acc[u] = zp_axpy_small(F, nv, (int)src[u], acc[u]); // (ldz is a multiple of 64: the four exist)
            } else {
                const int fc = -1 - e.x;
                if (fc >= c0 && fc < c0 + 4) acc[fc - c0] = zp_axpy_small(F, 1, e.y, acc[fc - c0]);
            }
        }
    }
    if (c0 < nf) {
        DT *dst = Y + (i64d)t * ldz + c0;
#pragma unroll
        for (int u = 0; u < 4; u++) dst[u] = (DT)acc[u];
    }
}

// ---- K = the transpose of Y, compacted.  Rows of Y in chunks of KD_CHUNK positions: cnt[chunk][i] = non-zeros of column i among the
// chunk's rows; after a scan in (i, chunk) order the fill pass writes every (pivot column, value) to its place -- deterministic.
#define KD_CHUNK 4096
template <typename DT>
__global__ __launch_bounds__(256) void k_kd_count(int r, int nf, const DT *__restrict__ Y, i64d ldz, i64d *__restrict__ cnt, int nchunks)
{
    const int i = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
    if (i >= nf) return;
    const int t_lo = ch * KD_CHUNK, t_hi = min(r, t_lo + KD_CHUNK);
    int n = 0;
    for (int t = t_lo; t < t_hi; t++) n += Y[(i64d)t * ldz + i] != 0;
    // (i major, chunk minor; every column starts with the entry on the free column itself: counted with chunk 0)
    cnt[(i64d)i * nchunks + ch] = (i64d)n + (ch == 0 ? 1 : 0);
}

template <typename DT>
__global__ __launch_bounds__(256) void k_kd_fill(int r, int nf, const DT *__restrict__ Y, i64d ldz, const i64d *__restrict__ off, int nchunks, const int *__restrict__ pivcol_of_pos,
                                                 const int *__restrict__ freecol, int *__restrict__ Kj, int *__restrict__ Kx)
{
    const int i = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
    if (i >= nf) return;
    const int t_lo = ch * KD_CHUNK, t_hi = min(r, t_lo + KD_CHUNK);
    i64d w = off[(i64d)i * nchunks + ch];
    if (ch == 0) { Kj[w] = freecol[i]; Kx[w] = -1; w++; }
    for (int t = t_lo; t < t_hi; t++) {
        const int v = (int)Y[(i64d)t * ldz + i];
        if (v != 0) { Kj[w] = pivcol_of_pos[t]; Kx[w] = v; w++; }
    }
}
