"""Per-row cost of a strided vs a contiguous 1/8 shard of config 3 (is the strided shard slower per row?): python tools/shard_contig.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spasm_jl_amd as S
lib = S._abi.lib()
n = 1_000_000
A = S.synth_csr(1, n, n, row_nnz=20, prime=65521, seed=0x5A5A0003)
def run(name, plan):
    assert plan, S._abi.last_error()
    lib.spasm_amd_schur_plan_class_timing(plan, 0)
    stream = torch.cuda.Stream(); sp = C.c_void_p(stream.cuda_stream)
    for _ in range(3): lib.spasm_amd_schur_plan_run(plan, sp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 20
    for _ in range(K): lib.spasm_amd_schur_plan_run(plan, sp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    st = S._abi.RoundStats(); lib.spasm_amd_schur_plan_stats(plan, C.byref(st))
    print(f"{name}: {dt*1e3:.3f} ms/step, nnz_reduced {st.nnz_reduced}, {st.nnz_reduced/dt:.3e} nnz/s, solve {st.ms_solve:.3f} scatter {st.ms_scatter:.3f}", flush=True)
    lib.spasm_amd_schur_plan_free(plan)
run("strided 7/8", lib.spasm_amd_schur_plan_create_strided(A.data, 7, n, 8))
run("contiguous [500000, 625000)", lib.spasm_amd_schur_plan_create(A.data, 500000, 625000))
run("contiguous [875000, 1000000)", lib.spasm_amd_schur_plan_create(A.data, 875000, 1000000))
