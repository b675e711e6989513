"""Times the Schur round of a 1/G row shard of config 3 on one GPU (what each rank does at N = G): python tools/shard_time.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spasm_jl_amd as S
lib = S._abi.lib()
n = 1_000_000
A = S.synth_csr(1, n, n, row_nnz=20, prime=65521, seed=0x5A5A0003)
for G in (1, 2, 4, 8):
    plan = lib.spasm_amd_schur_plan_create_strided(A.data, G - 1, n, G)  # the last strided shard
    lib.spasm_amd_schur_plan_class_timing(plan, 0)
    assert plan, S._abi.last_error()
    stream = torch.cuda.Stream(); sp = C.c_void_p(stream.cuda_stream)
    for _ in range(3): lib.spasm_amd_schur_plan_run(plan, sp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 20
    for _ in range(K): lib.spasm_amd_schur_plan_run(plan, sp)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    st = S._abi.RoundStats(); lib.spasm_amd_schur_plan_stats(plan, C.byref(st))
    print(f"G={G}: strided shard {G-1}/{G}: {dt*1e3:.3f} ms/step, nnz_reduced {st.nnz_reduced}, implied aggregate {st.nnz_reduced*G/dt:.3e} nnz/s, solve {st.ms_solve:.3f} scatter {st.ms_scatter:.3f}")
    if "-v" in sys.argv:  # one more step with an event pair around every class launch (sequential: no lanes)
        lib.spasm_amd_schur_plan_class_timing(plan, 1)
        lib.spasm_amd_schur_plan_run(plan, sp); torch.cuda.synchronize()
        lib.spasm_amd_schur_plan_stats(plan, C.byref(st))
        d = st.as_dict()
        print("   stream classes ms", [round(x, 3) for x in d["ms_class"][8:15]], "rows", d["rows_class"][8:15], "fix", round(d["ms_class"][15], 3),
              "hash ms", [round(x, 3) for x in d["ms_class"][:8]], "solve", round(d["ms_solve"], 3), "scatter", round(d["ms_scatter"], 3))
    lib.spasm_amd_schur_plan_free(plan)
