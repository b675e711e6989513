"""Times echelonize of a Macaulay-like matrix (BASELINE config 5 shape at 1/SCALE) on the GPU, no oracle:
   python tools/time_c5.py [scale=25]     (scale 1 = 5M x 2M)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spasm_jl_amd as S
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n, m = 5_000_000 // scale, 2_000_000 // scale
t0 = time.time(); A = S.synth_csr(2, n, m, row_nnz=40, prime=127, seed=0x5A5A0005); t1 = time.time()
print(f"generated {n} x {m}, nnz {S.nnz(A)} in {t1-t0:.2f}s", flush=True)
if "--rank-only" in sys.argv:   # spasm_amd_rank: the rows of U never leave the device (above 1/3 scale they outgrow the host)
    t0 = time.time(); r = S.rank(A, rank_only=True, verbose=("-v" in sys.argv)); t1 = time.time()
    print(f"rank only {t1-t0:.3f}s rank {r}", flush=True)
    sys.exit(0)
t0 = time.time(); fact = S.echelonize(A, verbose=("-v" in sys.argv)); t1 = time.time()
print(f"echelonize {t1-t0:.3f}s rank {fact.r} nnz(U) {S.nnz(fact.U)}", flush=True)
for r in S.last_rounds():
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ("round","rows_in","nnz_in","npiv","rows_out","nnz_out","nnz_reduced","ms_pivots","ms_solve","ms_scatter","dense")}, flush=True)
if "--verify" in sys.argv:
    t0 = time.time(); ok = S.factorization_verify(A, fact, 1); print(f"factorization_verify {ok} in {time.time()-t0:.2f}s", flush=True)
t0 = time.time(); K = S.kernel(fact, verbose=("-v" in sys.argv)); t1 = time.time()
print(f"kernel {t1-t0:.3f}s dim {K.n} nnz(K) {S.nnz(K)}", flush=True)
if "--check-kernel" in sys.argv:   # A * k^T == 0 for a sample of the basis, exact integers; rows(K) == m - rank
    import numpy as np, scipy.sparse as sp
    t0 = time.time()
    nzA = int(A.p[A.n]); As = sp.csr_matrix((np.asarray(A.x[:nzA], dtype=np.int64), np.asarray(A.j[:nzA], dtype=np.int64), np.asarray(A.p[:A.n + 1], dtype=np.int64)), shape=(A.n, A.m))
    Kp, Kj, Kx = np.asarray(K.p), np.asarray(K.j), np.asarray(K.x)
    pick = sorted(set(np.linspace(0, K.n - 1, num=min(K.n, 48), dtype=np.int64).tolist())) if K.n else []
    bad = 0
    for f in pick:
        k = np.zeros(A.m, dtype=np.int64); lo, hi = int(Kp[f]), int(Kp[f + 1]); k[Kj[lo:hi]] = Kx[lo:hi]
        bad += int(np.any((As @ k) % 127 != 0))
    print(f"kernel check: rows(K) {K.n} == m - r {A.m - fact.r}: {K.n == A.m - fact.r}; A * k^T == 0 for {len(pick) - bad} of {len(pick)} sampled vectors [{time.time()-t0:.1f}s]", flush=True)
    sys.exit(0 if (bad == 0 and K.n == A.m - fact.r) else 1)
