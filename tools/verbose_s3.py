import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spasm_jl_amd as S
A = S.synth_csr(1, 300000, 300000, row_nnz=3, prime=65521, seed=13)
t=time.time(); f = S.echelonize(A, verbose=True); print("wall", time.time()-t)
