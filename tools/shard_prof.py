import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spasm_jl_amd as S
lib = S._abi.lib()
n = 1_000_000
A = S.synth_csr(1, n, n, row_nnz=20, prime=65521, seed=0x5A5A0003)
plan = lib.spasm_amd_schur_plan_create_strided(A.data, 7, n, 8)
lib.spasm_amd_schur_plan_class_timing(plan, 0)
stream = torch.cuda.Stream(); sp = C.c_void_p(stream.cuda_stream)
for _ in range(10): lib.spasm_amd_schur_plan_run(plan, sp)
torch.cuda.synchronize()
