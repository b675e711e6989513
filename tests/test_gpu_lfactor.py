"""The L factor (echelonize_opts.L, reference src/SpaSM.jl:331, struct field :266) and what is built on it: the two-sided
factorization_verify (:934), gesv and solve (:889-923).  A[i] == sum_k L[i][k] U[k] is checked with exact integers."""
import numpy as np
import pytest
from conftest import LM  # leftmost-entry pivots only: what these tests compare does not depend on how the rounds went then

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(S):
    if S._abi.lib().spasm_amd_device_count() <= 0:
        pytest.fail("no HIP device: the -m gpu tests need the MI355X")


def lu_product_rows(L, U, p):
    Urows = U.rows()
    out = []
    for lrow in L.rows():
        acc = {}
        for k, v in lrow:
            for c, x in Urows[k]:
                acc[c] = (acc.get(c, 0) + v * x) % p
        out.append({c: v for c, v in acc.items() if v})
    return out


CASES = [
    ("fixed_nnz", 1, 1500, 1500, dict(row_nnz=5), 65521, {}),
    ("three_rounds_then_finish", 1, 1200, 1200, dict(row_nnz=6), 65521, dict(enable_greedy_pivot_search=True)),
    ("macaulay_like", 2, 2000, 800, dict(row_nnz=30), 127, {}),
    ("big_prime_wide", 1, 600, 900, dict(row_nnz=6), 0xFFFFFFFB, dict(enable_greedy_pivot_search=True)),
    ("bernoulli_tall", 0, 900, 400, dict(density=0.02), 2147483647, {}),
]


@pytest.mark.parametrize("name,kind,n,m,kw,prime,opts", CASES, ids=[c[0] for c in CASES])
def test_L_times_U_is_A(S, O, name, kind, n, m, kw, prime, opts):
    A = S.synth_csr(kind, n, m, prime=prime, seed=0x1FAC, **kw)
    fact = S.echelonize(A, L=True, **{**LM, **opts})
    olu = O.echelonize(A, **LM)
    assert fact.r == olu.r
    L, U = fact.L, fact.U
    assert L.shape == (n, fact.r)
    want = [{c: v % prime for c, v in row} for row in A.rows()]
    assert lu_product_rows(L, U, prime) == want                    # A == L * U, row for row
    # the pivotal rows of L: lower triangular with the pivots on the diagonal
    p = np.asarray(fact.p)
    Lrows = L.rows()
    for k in range(fact.r):
        row = dict(Lrows[int(p[k])])
        assert max(row) == k and row[k] % prime != 0
    for sd in (0, 1, 2):
        assert S.factorization_verify(A, fact, sd)
    # the kernel does not depend on whether L was kept
    plain = S.echelonize(A, enable_dense=False, **{**LM, **opts})
    assert np.asarray(fact.qinv >= 0).tolist() == np.asarray(plain.qinv >= 0).tolist()
    assert S.kernel(fact).rows() == S.kernel(plain).rows()


def test_L_survives_rounds_in_row_batches(S, O, monkeypatch):
    A = S.synth_csr(2, 4000, 1600, row_nnz=40, prime=127, seed=0x5A5A0005)
    ref = S.echelonize(A, L=True, **LM)
    monkeypatch.setenv("SPASM_AMD_MEM_BUDGET_MB", "8")
    got = S.echelonize(A, L=True, **LM)
    monkeypatch.delenv("SPASM_AMD_MEM_BUDGET_MB")
    assert got.r == ref.r and got.L.rows() == ref.L.rows() and got.U.rows() == ref.U.rows()
    assert S.factorization_verify(A, got, 5)


def test_two_sided_verify_catches_what_one_sided_misses(S, O):
    """A U with a junk row (a row outside the row space of A, on a free column) spans the rows of A just as well: the one-sided
    check accepts it and reports a rank that is one too high.  With L the check is two-sided and must refuse it."""
    n, m, prime = 400, 380, 65521
    A = S.synth_csr(1, n, m, row_nnz=3, prime=prime, seed=77)
    fact = S.echelonize(A, L=True, **LM)
    r = fact.r
    q = np.asarray(fact.qinv).copy()
    free = [j for j in range(m) if q[j] < 0]
    assert free
    junk_col = free[-1]                                             # (the last free column: no row of U reaches beyond it)
    Urows = [list(row) for row in fact.U.rows()] + [[(junk_col, 1)]]
    q2 = q.copy(); q2[junk_col] = r
    p = np.asarray(fact.p)
    spare = next(i for i in range(n) if i not in set(int(v) for v in p[:r]))
    p2 = np.full(max(n, m), -1, dtype=np.int32); p2[:r] = p[:r]; p2[r] = spare
    one_sided = S.LU.from_parts(S.CSR.from_rows(Urows, m, prime), q2.astype(np.int32), p2)
    assert S.factorization_verify(A, one_sided, 3)                  # the documented blind spot without L
    two_sided = S.LU.from_parts(S.CSR.from_rows(Urows, m, prime), q2.astype(np.int32), p2,
                                L=S.CSR.from_rows([list(row) for row in fact.L.rows()], r + 1, prime))
    assert not S.factorization_verify(A, two_sided, 3)
    # a changed multiplier: A != L * U
    Lrows = [list(row) for row in fact.L.rows()]
    i = next(i for i, row in enumerate(Lrows) if len(row) > 1)
    c, v = Lrows[i][0]
    Lrows[i][0] = (c, v + 1 if v + 1 <= prime // 2 else v - 1)
    if Lrows[i][0][1] != 0:
        bad = S.LU.from_parts(S.CSR.from_rows([list(row) for row in fact.U.rows()], m, prime), q.astype(np.int32), np.asarray(fact.p).astype(np.int32),
                              L=S.CSR.from_rows(Lrows, r, prime))
        assert not S.factorization_verify(A, bad, 3)


@pytest.mark.parametrize("n,m,k,prime,seed", [(400, 500, 5, 65521, 1), (500, 300, 4, 127, 2), (300, 420, 5, 2147483647, 3)])
def test_gesv_and_solve(S, O, n, m, k, prime, seed):
    A = S.synth_csr(1, n, m, row_nnz=k, prime=prime, seed=seed)
    fact = S.echelonize(A, L=True, **LM)
    Arows = A.rows()
    rng = np.random.default_rng(seed)

    def combo(cf):
        acc = {}
        for i, v in cf.items():
            for c, x in Arows[i]:
                acc[c] = (acc.get(c, 0) + v * x) % prime
        return sorted((c, v) for c, v in acc.items() if v)

    coeff = [{int(i): int(v) for i, v in zip(rng.choice(n, size=7, replace=False), rng.integers(1, min(prime, 1 << 31), size=7))} for _ in range(30)]
    good = [combo(cf) for cf in coeff] + Arows[:20]
    q = np.asarray(fact.qinv)
    free = [j for j in range(m) if q[j] < 0]
    bad = []
    for t in range(4 if free else 0):
        row = dict(good[t]); row[free[t % len(free)]] = (row.get(free[t % len(free)], 0) + 1) % prime
        bad.append(sorted((c, v) for c, v in row.items() if v))
    B = S.CSR.from_rows(good + bad, m, prime)
    X, ok = S.gesv(fact, B)
    assert X.shape == (B.n, n)
    assert ok[: len(good)].all() and not ok[len(good):].any()
    for b, xrow in enumerate(X.rows()[: len(good)]):
        assert combo({i: v % prime for i, v in xrow}) == [(c, v % prime) for c, v in good[b]]   # X[b] * A == B[b]
    # one dense vector
    bvec = np.zeros(m, dtype=np.int32)
    for c, v in good[0]:
        bvec[c] = v if v <= prime // 2 else v - prime
    x = S.solve(fact, bvec)
    assert x is not None and combo({i: int(v) % prime for i, v in enumerate(x) if v}) == [(c, v % prime) for c, v in good[0]]
    if bad:
        bvec2 = np.zeros(m, dtype=np.int32)
        for c, v in bad[0]:
            bvec2[c] = v if v <= prime // 2 else v - prime
        assert S.solve(fact, bvec2) is None
    with pytest.raises(S.SpasmError):
        S.gesv(S.echelonize(A, **LM), B)                                  # no L: the reference errors on fact.L too (:896, :916)
