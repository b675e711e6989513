"""Times spasm_amd_echelonize_multi (row shards of one process; on a one-GPU box they share the device) against spasm_echelonize:
   python tools/time_multi.py CONFIG SCALE NSHARDS [-v] [--verify]     CONFIG 5 = Macaulay-like 5M x 2M / SCALE, p = 127;
                                                                        CONFIG 3 = 1M x 1M / SCALE, 20 per row, p = 65521"""
import sys, time, os
os.environ.setdefault("SPASM_AMD_MULTI_DENSE_MIN_BYTES", "0")  # (time the dense finish over the shards whatever the size)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spasm_jl_amd as S
cfg, scale, nsh = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if cfg == 5:
    n, m = 5_000_000 // scale, 2_000_000 // scale
    A = S.synth_csr(2, n, m, row_nnz=40, prime=127, seed=0x5A5A0005)
else:
    n = m = 1_000_000 // scale
    A = S.synth_csr(1, n, m, row_nnz=20, prime=65521, seed=0x5A5A0003)
print(f"generated {n} x {m}, nnz {S.nnz(A)}", flush=True)
v = "-v" in sys.argv
if nsh > 0:
    t0 = time.time(); fact = S.echelonize_multi(A, nsh, verbose=v); t1 = time.time()
    print(f"echelonize_multi({nsh}) {t1-t0:.3f}s rank {fact.r} nnz(U) {S.nnz(fact.U)} finish {S._abi.lib().spasm_amd_multi_last_finish()}", flush=True)
else:
    t0 = time.time(); fact = S.echelonize(A, verbose=v, enable_greedy_pivot_search=False); t1 = time.time()
    print(f"echelonize {t1-t0:.3f}s rank {fact.r} nnz(U) {S.nnz(fact.U)}", flush=True)
if "--verify" in sys.argv:
    t0 = time.time(); ok = S.factorization_verify(A, fact, 1); print(f"factorization_verify {ok} in {time.time()-t0:.2f}s", flush=True)
