"""Random n x n, 3 nnz/row (the fill-in phase transition of sparse elimination): python tools/time_sparse3.py [n=30000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spasm_jl_amd as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
A = S.synth_csr(1, n, n, row_nnz=3, prime=65521, seed=13)
t = time.time(); f = S.echelonize(A); dt = time.time() - t
rs = S.last_rounds()
print(f"n={n}: echelonize {dt:.2f}s rank {f.r} rounds {len(rs)} nnz(U) {S.nnz(f.U)}", flush=True)
for r in rs[:3] + rs[-3:]:
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if k in ("round","rows_in","nnz_in","npiv","rows_out","nnz_out","nnz_reduced","ms_pivots","ms_solve","ms_scatter")}, flush=True)
