"""Times echelonize + kernel of BASELINE config 2 on the GPU (no oracle): python tools/time_c2.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spasm_jl_amd as S
A = S.synth_csr(0, 10000, 10000, density=1e-3, prime=42013, seed=0x5A5A0002)
for rep in range(2):
    t0 = time.time(); fact = S.echelonize(A); t1 = time.time(); K = S.kernel(fact); t2 = time.time()
    print(f"rep {rep}: echelonize {t1-t0:.3f}s rank {fact.r}  kernel {t2-t1:.3f}s dim {K.n} nnz(K) {S.nnz(K)} nnz(U) {S.nnz(fact.U)}")
for r in S.last_rounds():
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ("round","rows_in","nnz_in","npiv","rows_out","nnz_out","nnz_reduced","ms_pivots","ms_solve","ms_scatter")})
