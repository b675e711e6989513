"""What the three structural pivot searches find in round 0 and what they cost (echelonize with max_round = 1, dense off, rounds read
back): python tools/time_pivots.py        (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spasm_jl_amd as S
cases = [("100k x 100k, 6 per row", 1, 100000, 100000, dict(row_nnz=6), 65521),
         ("300k x 300k, 3 per row", 1, 300000, 300000, dict(row_nnz=3), 65521),
         ("config 3 at 1/4 (250k x 250k, 20 per row)", 1, 250000, 250000, dict(row_nnz=20), 65521),
         ("config 5 at 1/25 (200k x 80k Macaulay-like)", 2, 200000, 80000, dict(row_nnz=40), 127)]
for name, kind, n, m, kw, p in cases:
    A = S.synth_csr(kind, n, m, prime=p, seed=0x5A5A0003, **kw)
    out = []
    for label, greedy, env in (("leftmost", False, {}), ("+ on columns", True, {"SPASM_AMD_NO_CYCLE_FREE_SEARCH": "1"}), ("+ cycle-free", True, {})):
        for k, v in env.items():
            os.environ[k] = v
        t0 = time.time()
        S.echelonize(A, enable_greedy_pivot_search=greedy, enable_dense=False, max_round=1, enable_GPLU=False)
        dt = time.time() - t0
        for k in env:
            del os.environ[k]
        r0 = S.last_rounds()[0]
        out.append(f"{label}: {r0['npiv']} pivots (open {r0['npiv_open']}, greedy {r0['npiv_greedy']}), round-0 pivots {r0['ms_pivots']:.1f} ms, nnz_out {r0['nnz_out']}, call {dt:.2f}s")
    print(name)
    for o in out:
        print("   ", o, flush=True)
