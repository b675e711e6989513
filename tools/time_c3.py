"""Whole echelonize of BASELINE config 3 (1M x 1M, 20 nnz/row, p = 65521) at 1/SCALE rows and columns: python tools/time_c3.py [scale=1] [-v]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spasm_jl_amd as S
scale = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
n = 1_000_000 // scale
A = S.synth_csr(1, n, n, row_nnz=20, prime=65521, seed=0x5A5A0003)
t = time.time(); f = S.echelonize(A, verbose=("-v" in sys.argv)); dt = time.time() - t
print(f"n={n}: echelonize {dt:.2f}s rank {f.r} rounds {len(S.last_rounds())} nnz(U) {S.nnz(f.U)}", flush=True)
for r in S.last_rounds():
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if k in ("round", "rows_in", "nnz_in", "npiv", "npiv_open", "rows_out", "nnz_out", "ms_pivots", "ms_solve", "ms_scatter", "ms_uinv", "ms_w")}, flush=True)
