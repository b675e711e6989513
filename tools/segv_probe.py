"""Reproduces the SIGSEGV at exit() of a rocprofv3-profiled process that used the cooperative panel kernel, and leaves what is
needed to name the frames: /proc/self/maps as the interpreter shuts down (gpurun_out/segv/maps.txt) next to rocprofv3's stderr
with the raw backtrace (gpurun_out/segv/trace.err).  tools/segv_symbolise.py turns the two into library + nearest symbol.
SPASM_PROBE_COOP=0: the same run without the dense finish (no cooperative launch) as the control."""
import atexit
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out", "segv"), exist_ok=True)


def dump():
    tag = os.environ.get("SPASM_PROBE_TAG", "coop")
    with open(os.path.join(ROOT, "gpurun_out", "segv", f"maps_{tag}.txt"), "w") as f:
        f.write(open("/proc/self/maps").read())


atexit.register(dump)
import numpy as np  # noqa: E402

import spasm_jl_amd as S  # noqa: E402

rng = np.random.default_rng(1)
n, m, p = 400, 300, 127
D = (rng.random((n, m)) < 0.3) * rng.integers(1, p, size=(n, m))
A = S.CSR(D.T.copy(), prime=p)
coop = os.environ.get("SPASM_PROBE_COOP", "1") != "0"
fact = S.echelonize(A, enable_dense=coop)
print("rank", fact.r, "dense finish" if coop else "sparse rounds only", flush=True)
