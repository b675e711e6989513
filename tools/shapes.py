"""Extreme shapes through the C ABI (no oracle: the expected ranks are known by construction): python tools/shapes.py [wide|tall]
Each case prints as it goes (run it with its own timeout; write to a file under gpurun_out/ on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spasm_jl_amd as S

def from_coo(n, m, rows, cols, vals, prime):
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    p = np.zeros(n + 1, dtype=np.int64)
    np.add.at(p, rows + 1, 1)
    p = np.cumsum(p)
    return S.CSR.from_arrays(n, m, p, cols.astype(np.int32), vals.astype(np.int32), prime) if hasattr(S.CSR, "from_arrays") else None

def run(name, A, want_rank):
    t = time.time(); f = S.echelonize(A); dt = time.time() - t
    print(f"{name}: echelonize {dt:.2f}s rank {f.r}", flush=True)
    t = time.time(); K = S.kernel(f); print(f"{name}: kernel {time.time()-t:.2f}s rows {K.n}", flush=True)
    ok = f.r == want_rank and K.n == A.m - want_rank if False else f.r == want_rank
    print(f"{name}: {A.n} x {A.m}, nnz {S.nnz(A)}: rank {f.r} (want {want_rank}) kernel rows {K.n} in {dt:.2f}s {'OK' if ok else 'MISMATCH'}", flush=True)
    assert ok

rng = np.random.default_rng(1)
which = sys.argv[1] if len(sys.argv) > 1 else "wide"
if which == "wide":
    # 2000 rows over 40M columns: full row rank with overwhelming probability; the kernel has 39 998 000 vectors
    n, m = 2000, 40_000_000
    A = S.synth_csr(1, n, m, row_nnz=8, prime=65521, seed=11)
    print("generated", flush=True)
    run("wide", A, n)
if which == "tall":
    # 3M rows over 300 columns: rank 300 with overwhelming probability
    A = S.synth_csr(1, 3_000_000, 300, row_nnz=3, prime=65521, seed=12)
    print("generated", flush=True)
    run("tall", A, 300)
