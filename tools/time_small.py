import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import spasm_jl_amd as S
rng = np.random.default_rng(1)
for (n, m, p, d) in [(50, 60, 7, 0.2), (50, 60, 127, 0.08), (60, 45, 251, 0.5), (200, 150, 65521, 0.05)]:
    D = (rng.random((n, m)) < d) * rng.integers(1, p, size=(n, m))
    A = S.CSR(D.T.copy(), prime=p)
    for name, kw in [("default", {}), ("leftmost", dict(enable_greedy_pivot_search=False)), ("nodense", dict(enable_dense=False))]:
        S.echelonize(A, **kw)
        t = time.time()
        for _ in range(5):
            f = S.echelonize(A, **kw)
        dt = (time.time() - t) / 5
        t = time.time(); K = S.kernel(f); dk = time.time() - t
        print(n, m, p, name, "echelonize %.1f ms" % (1e3 * dt), "kernel %.1f ms" % (1e3 * dk), "rank", f.r, flush=True)
