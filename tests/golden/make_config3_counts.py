"""Generates tests/golden/config3_oracle_counts.json: the counters of the CPU oracle (oracle/, the libspasm-algorithm
restatement) for Schur round 0 of BASELINE config 3 at FULL size (1M x 1M, 20 nnz/row, p = 65521, seed 0x5A5A0003).
Run from the repo root:  python tests/golden/make_config3_counts.py      (about 15 s on 8 cores, 10 GB of RAM)
The matrix comes from the engine's host-side generator (spasm_amd_synth_csr, no GPU involved)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ffi as O
import spasm_jl_amd as S

A = S.synth_csr(1, 1_000_000, 1_000_000, row_nnz=20, prime=65521, seed=0x5A5A0003)
_, info = O.schur_round(A)
out = {k: int(info[k]) for k in ("npiv", "applications", "nnz_reduced", "nnz_out", "rows_out", "nnz_U")}
out["_workload"] = "synth kind 1, 1000000 x 1000000, row_nnz 20, prime 65521, seed 0x5A5A0003, Schur round 0"
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "config3_oracle_counts.json"), "w"), indent=1)
print(out)
