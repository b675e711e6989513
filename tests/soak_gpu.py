"""Randomised soak of the round-3 and round-4 paths against the single-device leftmost-pivot run and the oracle (small matrices, many shapes):
the dense finish over row shards (random shard counts and block sizes), the tall-and-skinny finish (random slabs and batches), the tall finish with column slabs and row chunks, the kernel paths of round 4, the
greedy search (random limits; engine vs oracle pair for pair through max_round = 1 runs).   python tests/soak_gpu.py [seconds=120] [seed=1] [scale=1]   (test infrastructure: it uses the oracle; not collected by pytest)"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [root, os.path.join(root, "tests")]
import numpy as np
import spasm_jl_amd as S
import oracle_ffi as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
SC = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # matrix sizes times SC
LM = dict(enable_greedy_pivot_search=False)
t_end = time.time() + budget
done = {"multi": 0, "tall": 0, "greedy": 0, "kernel": 0}
KEYS = ["SPASM_AMD_TALL_CHUNK", "SPASM_AMD_MEM_BUDGET_MB", "SPASM_AMD_KERNEL_DENSE_RHS", "SPASM_AMD_KERNEL_DENSE_TAIL", "SPASM_AMD_KERNEL_REDUCE_NNZ", "SPASM_AMD_MULTI_DENSE_MIN_BYTES", "SPASM_AMD_MULTI_FINISH_NNZ", "SPASM_AMD_DENSE_KB", "SPASM_AMD_TALL", "SPASM_AMD_TALL_SLAB", "SPASM_AMD_TALL_BATCH", "SPASM_AMD_GREEDY_REACH_MAX",
        "SPASM_AMD_GREEDY_OCC_MAX", "SPASM_AMD_PANEL_GLOBAL"]


def setenv(**kw):
    for k in KEYS:
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)


def pattern(f):
    return np.asarray(f.qinv >= 0).tolist()


def random_matrix():
    p = int(rng.choice([3, 7, 127, 251, 257, 42013, 65521]))
    kind = int(rng.integers(0, 4))
    if kind == 0:      # sparse, fixed entries per row
        n, m = int(rng.integers(50, 3000 * SC)), int(rng.integers(50, 3000 * SC))
        return S.synth_csr(1, n, m, row_nnz=int(rng.integers(2, min(m, 12))), prime=p, seed=int(rng.integers(1 << 30))), p
    if kind == 1:      # Macaulay-like
        m = int(rng.integers(200, 1500 * SC)); n = int(m * rng.uniform(1.5, 4))
        return S.synth_csr(2, n, m, row_nnz=int(rng.integers(10, 40)), prime=p, seed=int(rng.integers(1 << 30))), p
    if kind == 2:      # dense, low rank pieces
        m = int(rng.integers(20, 400 * SC)); n = int(rng.integers(20, 1500 * SC)); r = int(rng.integers(1, min(n, m) + 1))
        M = (rng.integers(0, p, size=(n, r)).astype(np.int64).dot(rng.integers(0, p, size=(r, m)).astype(np.int64))) % p
        if rng.random() < 0.5:
            M[:, rng.integers(0, m, size=max(1, m // 10))] = 0
        return S.CSR(M.T.copy(), prime=p), p
    n, m = int(rng.integers(30, 1200 * SC)), int(rng.integers(30, 1200 * SC))   # Bernoulli
    return S.synth_csr(0, n, m, density=float(rng.uniform(0.01, 0.4)), prime=p, seed=int(rng.integers(1 << 30))), p


case = 0
t_say = time.time()
while time.time() < t_end:
    case += 1
    if time.time() - t_say > 60:   # (a run that says nothing for minutes is taken to be hung)
        print("soak:", case, "cases so far", done, flush=True)
        t_say = time.time()
    A, p = random_matrix()
    setenv()
    ref = S.echelonize(A, **LM)
    olu = O.echelonize(A, **LM)
    assert ref.r == olu.r and pattern(ref) == pattern(olu), ("single device vs oracle", case)
    K = S.kernel(ref).rows()
    which = case % 4
    if which == 0:
        nsh = int(rng.integers(1, 9))
        env = dict(SPASM_AMD_MULTI_DENSE_MIN_BYTES=0, SPASM_AMD_MULTI_FINISH_NNZ=int(rng.choice([1, 1000, 1 << 22])), SPASM_AMD_DENSE_KB=int(rng.choice([64, 128, 256, 1024])))
        if rng.random() < 0.2:
            env["SPASM_AMD_PANEL_GLOBAL"] = 1
        setenv(**env)
        if rng.random() < 0.4:
            # r04: the reference's default options over the shards ("FL on columns" over the shards): other pivots than `ref`, the same
            # rank, a U that verifies, a kernel of the same dimension with A k^T = 0; round 0's pivots are the single-device round's
            # without the third search
            got = S.echelonize_multi(A, nsh)
            tag = ("multi default options", nsh, env, S._abi.lib().spasm_amd_multi_last_finish())
            assert got.r == ref.r, ("rank", case, A.n, A.m, p, tag, got.r, ref.r)
            assert S.factorization_verify(A, got, 3), ("verify", case, A.n, A.m, p, tag)
            Kg = S.kernel(got)
            assert Kg.n == A.m - ref.r, ("kernel dimension", case, tag)
            Ar = A.rows()
            for kv in Kg.rows()[:: max(1, Kg.n // 5)]:
                kd = dict(kv)
                assert all(sum(v * kd.get(c, 0) for c, v in row) % p == 0 for row in Ar[:: max(1, len(Ar) // 100)]), ("A k^T", case, tag)
            done["multi"] += 1
            continue
        got = S.echelonize_multi(A, nsh, **LM)
        tag = ("multi", nsh, env, S._abi.lib().spasm_amd_multi_last_finish())
        done["multi"] += 1
    elif which == 1:
        env = dict(SPASM_AMD_TALL=1, SPASM_AMD_TALL_SLAB=int(rng.choice([64, 128, 320, 1024])), SPASM_AMD_TALL_BATCH=int(rng.choice([128, 512, 100000])),
                   SPASM_AMD_DENSE_KB=int(rng.choice([64, 256, 1024])))
        if rng.random() < 0.5:   # r04: the residuals in chunks of rows
            env["SPASM_AMD_TALL_CHUNK"] = int(rng.choice([64, 256, 1024]))
        if rng.random() < 0.5:   # r04: the dense W in column slabs (the free columns of a slab are ADDED to the residuals)
            env["SPASM_AMD_MEM_BUDGET_MB"] = int(rng.choice([1, 2, 8]))
        setenv(**env)
        got = S.echelonize(A, sparsity_threshold=float(rng.choice([0.001, 0.05, 0.3])), **LM)
        tag = ("tall", env)
        done["tall"] += 1
    elif which == 3:
        # r04: the kernel through a dense right-hand side (any split between dense tail and sparse rows) and through the closure
        # of the free columns, from the leftmost-pivot factorization and from the default one
        f2 = S.echelonize(A) if rng.random() < 0.5 else ref
        want = K if f2 is ref else None
        setenv()
        if want is None:
            want = S.kernel(f2).rows()
        if p < 65536 and rng.random() < 0.7:
            env = dict(SPASM_AMD_KERNEL_DENSE_RHS=1)
            if rng.random() < 0.8:
                env["SPASM_AMD_KERNEL_DENSE_TAIL"] = int(rng.choice([0, 1, 63, 64, 65, 300, 1 << 30]))
        else:
            env = dict(SPASM_AMD_KERNEL_DENSE_RHS=0, SPASM_AMD_KERNEL_REDUCE_NNZ=0)
        setenv(**env)
        assert S.kernel(f2).rows() == want, ("kernel paths", case, A.n, A.m, p, env)
        done["kernel"] += 1
        continue
    else:
        env = dict(SPASM_AMD_GREEDY_REACH_MAX=int(rng.choice([0, 1, 2, 8, 64, 1024])), SPASM_AMD_GREEDY_OCC_MAX=int(rng.choice([1, 2, 5, 1 << 30])))
        setenv(**env)
        # (one sparse round: what follows it differs in the ORDER of the rows of U -- the engine finishes in rounds, the oracle row by row)
        got = S.echelonize(A, enable_greedy_pivot_search=True, enable_dense=False, max_round=1)
        og = O.echelonize(A, enable_greedy_pivot_search=True, max_round=1)
        assert got.r == og.r == ref.r and pattern(got) == pattern(og), ("greedy engine vs oracle", case, A.n, A.m, p, env)
        k = S.last_rounds()[0]["npiv"]
        assert got.U.rows()[:k] == og.U.rows()[:k], ("greedy rows of U", case, A.n, A.m, p, env)
        assert S.factorization_verify(A, got, 3), ("greedy verify", case, env)
        assert S.kernel(got).rows() == O.kernel(og).rows(), ("greedy kernel", case, env)
        done["greedy"] += 1
        continue
    assert got.r == ref.r, ("rank", case, A.n, A.m, p, tag, got.r, ref.r)
    assert pattern(got) == pattern(ref), ("pivot columns", case, A.n, A.m, p, tag)
    assert S.factorization_verify(A, got, 3), ("verify", case, A.n, A.m, p, tag)
    assert S.kernel(got).rows() == K, ("kernel", case, A.n, A.m, p, tag)
setenv()
print("soak ok:", case, "cases", done)
