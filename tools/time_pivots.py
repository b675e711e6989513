"""What the three structural pivot searches find in round 0, what they cost, and what the Schur complement of the round weighs
(echelonize with max_round = 1, dense off, rounds read back), for several settings of the greedy search's two limits:
   python tools/time_pivots.py        (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spasm_jl_amd as S
cases = [("100k x 100k, 6 per row", 1, 100000, 100000, dict(row_nnz=6), 65521),
         ("300k x 300k, 3 per row", 1, 300000, 300000, dict(row_nnz=3), 65521),
         ("config 3 at 1/4 (250k x 250k, 20 per row)", 1, 250000, 250000, dict(row_nnz=20), 65521),
         ("config 5 at 1/25 (200k x 80k Macaulay-like)", 2, 200000, 80000, dict(row_nnz=40), 127)]
settings = [("leftmost", False, {}), ("+ on columns", True, {"SPASM_AMD_NO_CYCLE_FREE_SEARCH": "1"})]
for reach, occ in ((1024, 0), (64, 0), (8, 0), (1024, 2), (64, 2), (8, 2), (64, 1), (8, 1), (2, 1)):
    env = {"SPASM_AMD_GREEDY_REACH_MAX": str(reach)}
    if occ:
        env["SPASM_AMD_GREEDY_OCC_MAX"] = str(occ)
    settings.append((f"+ cycle-free reach<={reach} occ<={occ or 'any'}", True, env))
for name, kind, n, m, kw, p in cases:
    A = S.synth_csr(kind, n, m, prime=p, seed=0x5A5A0003, **kw)
    print(name, flush=True)
    for label, greedy, env in settings:
        for k, v in env.items():
            os.environ[k] = v
        t0 = time.time()
        S.echelonize(A, enable_greedy_pivot_search=greedy, enable_dense=False, max_round=1, enable_GPLU=False)
        dt = time.time() - t0
        for k in env:
            del os.environ[k]
        r0 = S.last_rounds()[0]
        print(f"    {label:44s} {r0['npiv']:7d} pivots (open {r0['npiv_open']}, greedy {r0['npiv_greedy']}), pivots+U+W {r0['ms_pivots']:8.1f} ms, Schur {r0['ms_solve'] + r0['ms_scatter']:8.1f} ms, nnz_out {r0['nnz_out']}, call {dt:.2f}s", flush=True)
