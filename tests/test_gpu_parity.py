"""GPU parity tests: the HIP path, called through the C ABI (spasm_echelonize / spasm_kernel /
spasm_transpose / the Schur plan), against the CPU oracle, the committed golden vectors and an
independent dense elimination.  Bit-exact: integer / index work."""
import ctypes as C
import json
import os

import numpy as np
import pytest
from conftest import LM  # leftmost-entry pivots only: what these tests compare does not depend on how the rounds went then

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
P0 = GOLD["prime"]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(S):
    assert S._abi.lib().spasm_amd_device_count() > 0, "no HIP device visible: GPU tests need the MI355X"


def julia_sparse_of_kernel(K, p):
    D = np.zeros((K.m, K.n), dtype=np.int64)
    for f, row in enumerate(K.rows()):
        for c, v in row:
            D[c, f] = v % p
    return D


# ---- the reference's own tests, through the C ABI -------------------------------------------------

def test_construction_roundtrip(S):
    m = np.array(GOLD["roundtrip"]["m"])
    sm = S.CSR(m)
    assert (S.sparse(sm).toarray() % P0 == m % P0).all()  # reference test/runtests.jl:7-10


def test_transpose_involution(S):
    sm = S.CSR(np.array(GOLD["roundtrip"]["m"]))
    tt = S.transpose(S.transpose(sm))
    assert (S.sparse(tt) != S.sparse(sm)).nnz == 0  # reference test/runtests.jl:12-15
    t = S.transpose(sm)
    assert t.shape == (sm.m, sm.n)
    assert (t.todense() == sm.todense().T).all()


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: c["name"])
def test_reference_known_answer_kernels(S, case):
    if "m" in case:
        sm = S.CSR(np.array(case["m"]))
    else:
        base = next(c for c in GOLD["cases"] if c["name"] == case["m_transposed_of"])
        sm = S.transpose(S.CSR(np.array(base["m"])))
    fact = S.echelonize(sm, **LM)
    assert S.rank(fact) == case["rank"]
    k = S.kernel(fact)
    assert (julia_sparse_of_kernel(k, P0) == np.array(case["kernel_sparse"])).all()  # test/runtests.jl:20-23, README.md:44-47
    # kernel rows come out in ascending free-column order with K[j] = -1 first
    free = [j for j in range(sm.m) if fact.qinv[j] < 0]
    assert [int(k.j[k.p[f]]) for f in range(k.n)] == free
    assert all(int(k.x[k.p[f]]) == -1 for f in range(k.n))


def test_one_stop_kernel_and_rank(S):
    sm = S.CSR(np.array([[1, 2], [3, 6]]))
    assert S.rank(sm) == 1  # reference src/SpaSM.jl:1149
    assert (julia_sparse_of_kernel(S.kernel(sm), P0) == np.array([[3], [42012]])).all()  # :1147


# ---- random matrices vs the oracle and vs dense elimination -------------------------------------

def check_lu(S, A, fact, D, p):
    """Invariants of the returned factorization (layout reference src/SpaSM.jl:262-270, :705-712)."""
    U, qinv = fact.U, fact.qinv
    r = fact.r
    assert U.shape == (r, A.m) and len(qinv) == A.m
    pc = {int(qinv[j]): j for j in range(A.m) if qinv[j] >= 0}
    assert sorted(pc) == list(range(r))
    Ud = U.todense()
    for a in range(r):
        assert Ud[a, pc[a]] == 1  # unit pivots
    # U spans the row space of A: rank([A;U]) == rank(A) == r  (checked by the dense eliminator)
    return pc


@pytest.mark.parametrize("n,m,p,density,seed", [
    (12, 9, 7, 0.4, 1), (30, 40, 127, 0.15, 2), (60, 45, 42013, 0.08, 3), (80, 80, 65521, 0.05, 4),
    (50, 70, 0xFFFFFFFB, 0.1, 5), (40, 40, 3, 0.3, 6), (1, 17, 42013, 0.5, 7), (25, 1, 42013, 0.5, 8),
    (200, 150, 65537, 0.03, 9), (150, 220, 2147483647, 0.04, 10),
])
@pytest.mark.parametrize("enable_dense", [True, False], ids=["dense_tail", "sparse_rounds_only"])
def test_echelonize_kernel_vs_oracle_and_dense(S, O, n, m, p, density, seed, enable_dense):
    from test_oracle_golden import random_rows

    rng = np.random.default_rng(seed)
    D = random_rows(rng, n, m, p, density, rank_deficient=True)
    A = S.CSR(D.T.copy(), prime=p)
    fact = S.echelonize(A, enable_dense=enable_dense, **LM)  # reference option, src/SpaSM.jl:329
    K = S.kernel(fact)
    olu = O.echelonize(A, **LM)
    oK = O.kernel(olu)
    assert fact.r == olu.r
    assert (np.asarray(fact.qinv) >= 0).tolist() == (olu.qinv >= 0).tolist()  # identical pivot columns
    assert K.rows() == oK.rows()  # identical kernel basis, entry for entry
    Kd, piv = O.dense_kernel_normal_form(D, p)
    assert fact.r == len(piv) and sorted(np.nonzero(np.asarray(fact.qinv) >= 0)[0].tolist()) == piv
    assert (K.todense() == Kd).all()
    pc = check_lu(S, A, fact, D, p)
    # every row of U lies in the row space of A: rank of the stacked matrix does not grow
    stacked = np.vstack([D % p, fact.U.todense() % p])
    R2, piv2 = O.dense_rref(stacked, p)
    assert len(piv2) == fact.r


@pytest.mark.parametrize("n,m,p,density,seed", [
    (200, 300, 127, 0.3, 1), (260, 190, 65521, 0.25, 2), (150, 400, 16777213, 0.2, 3), (180, 180, 2147483647, 0.3, 4), (70, 500, 42013, 0.5, 5),
])
def test_dense_tail_multi_panel(S, O, n, m, p, density, seed):
    """Dense inputs go straight to the dense tail: several 64-column panels, f64-MFMA trailing updates for p <= 2^24,
    the rank-1 path above; rank, pivot columns and kernel against the oracle and the numpy eliminator."""
    from test_oracle_golden import random_rows

    rng = np.random.default_rng(seed)
    D = random_rows(rng, n, m, p, density, rank_deficient=True)
    D[:, 5] = 0          # a zero column and a repeated column: free columns inside the panels
    D[:, 70] = D[:, 3]
    A = S.CSR(D.T.copy(), prime=p)
    fact = S.echelonize(A, **LM)
    olu = O.echelonize(A, **LM)
    assert fact.r == olu.r
    assert (np.asarray(fact.qinv) >= 0).tolist() == (olu.qinv >= 0).tolist()
    Kd, piv = O.dense_kernel_normal_form(D, p)
    assert sorted(np.nonzero(np.asarray(fact.qinv) >= 0)[0].tolist()) == piv
    K = S.kernel(fact)
    assert (K.todense() == Kd).all()
    check_lu(S, A, fact, D, p)


def test_kernel_accepts_foreign_factorizations(S, O):
    """spasm_kernel on a factorization this engine did not produce: U rows in arbitrary order, pivot not the
    first entry of its row (reference src/SpaSM.jl:711 allows both); here the oracle's LU, rows shuffled."""
    import ctypes as C

    rng = np.random.default_rng(77)
    from test_oracle_golden import random_rows

    D = random_rows(rng, 60, 80, 65521, 0.06, rank_deficient=True)
    A = S.CSR(D.T.copy(), prime=65521)
    olu = O.echelonize(A, **LM)
    want = O.kernel(olu).rows()
    r = olu.r
    perm = rng.permutation(r)
    Urows = olu.U.rows()
    shuffled = [list(reversed(Urows[int(a)])) for a in perm]  # row order and entry order both scrambled
    Us = S.CSR.from_rows(shuffled, A.m, prime=65521)
    inv = np.empty(r, dtype=np.int64)
    inv[perm] = np.arange(r)
    qinv = np.array([inv[q] if q >= 0 else -1 for q in olu.qinv], dtype=np.int32)
    lu = S._abi.LuStruct()
    lu.r = r
    lu.complete = False
    lu.L = None
    lu.U = Us.data
    lu.qinv = qinv.ctypes.data_as(C.POINTER(C.c_int32))
    lu.p = None
    ptr = S._abi.lib().spasm_kernel(C.byref(lu))
    assert ptr, S._abi.last_error()
    assert S.CSR(ptr).rows() == want


def test_edge_cases(S):
    # empty matrix, zero rows, zero columns, all-zero rows, single entry
    for n, m in [(0, 5), (5, 0), (3, 4)]:
        A = S.CSR.from_rows([[] for _ in range(n)], m)
        fact = S.echelonize(A, **LM)
        assert fact.r == 0
        K = S.kernel(fact)
        assert K.shape == (m, m)
        assert K.rows() == [[(j, -1)] for j in range(m)]
    A = S.CSR.from_rows([[(3, 5)]], 6)
    fact = S.echelonize(A, **LM)
    assert fact.r == 1 and fact.qinv.tolist() == [-1, -1, -1, 0, -1, -1]
    assert fact.U.rows() == [[(3, 1)]]
    assert S.kernel(fact).rows() == [[(j, -1)] for j in (0, 1, 2, 4, 5)]
    # duplicate rows and a dense block
    rows = [[(0, 1), (1, 2), (2, 3)]] * 4 + [[(c, c + 1) for c in range(8)]]
    A = S.CSR.from_rows(rows, 8)
    assert S.rank(A) == 2


def test_config2_random_10k(S, O):
    """BASELINE config 2: random 10k x 10k, density 1e-3, p = 42013: echelonize + kernel, rank bit-exact vs CPU."""
    A = S.synth_csr(0, 10000, 10000, density=1e-3, prime=42013, seed=0x5A5A0002)
    fact = S.echelonize(A, **LM)
    olu = O.echelonize(A, **LM)
    assert fact.r == olu.r
    assert (np.asarray(fact.qinv) >= 0).tolist() == (olu.qinv >= 0).tolist()
    assert S.factorization_verify(A, fact, 2)  # the reference's self-check (src/SpaSM.jl:934) accepts the engine's LU
    K = S.kernel(fact)
    oK = O.kernel(olu)
    assert K.shape == (oK.n, oK.m)
    assert (K.p == oK.p).all()
    # same rows up to the order of entries inside a row
    kp, kj, kx = K.p, K.j, K.x
    for f in range(K.n):
        lo, hi = int(kp[f]), int(kp[f + 1])
        got = sorted(zip(kj[lo:hi].tolist(), kx[lo:hi].tolist()))
        lo2, hi2 = int(oK.p[f]), int(oK.p[f + 1])
        want = sorted(zip(oK.j[lo2:hi2].tolist(), oK.x[lo2:hi2].tolist()))
        assert got == want
    # A * k^T == 0 for every kernel vector (checked on the sparse structure, exact integers)
    import scipy.sparse as sp

    nz = S.nnz(A)
    As = sp.csr_matrix((A.x[:nz].astype(np.int64), A.j[:nz].astype(np.int64), A.p.astype(np.int64)), shape=A.shape)
    nk = S.nnz(K)
    Ks = sp.csr_matrix((K.x[:nk].astype(np.int64), K.j[:nk].astype(np.int64), K.p.astype(np.int64)), shape=K.shape)
    prod = (As @ Ks.T).tocoo()
    assert (prod.data % 42013 == 0).all()


def test_config5_macaulay_style_scaled_down(S, O):
    """BASELINE config 5 at 1/250 scale: Macaulay-like 20000 x 8000, p = 127: many FL pivots, small dense tail."""
    A = S.synth_csr(2, 20000, 8000, row_nnz=40, prime=127, seed=0x5A5A0005)
    fact = S.echelonize(A, **LM)
    rounds = S.last_rounds()
    assert rounds and rounds[0]["npiv"] > 0.5 * min(A.n, A.m) * 0.5  # the first round elects most pivots
    olu = O.echelonize(A, **LM)
    assert fact.r == olu.r
    assert (np.asarray(fact.qinv) >= 0).tolist() == (olu.qinv >= 0).tolist()
    assert S.factorization_verify(A, fact, 5)
    K = S.kernel(fact)
    oK = O.kernel(olu)
    assert K.shape == (oK.n, oK.m) and (K.p == oK.p).all()
    for f in range(0, K.n, max(1, K.n // 200)):  # spot-check kernel vectors entry for entry
        lo, hi = int(K.p[f]), int(K.p[f + 1])
        lo2, hi2 = int(oK.p[f]), int(oK.p[f + 1])
        assert sorted(zip(K.j[lo:hi].tolist(), K.x[lo:hi].tolist())) == sorted(zip(oK.j[lo2:hi2].tolist(), oK.x[lo2:hi2].tolist()))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,m,kw,prime", [(1, 2500, 2500, dict(row_nnz=10), 65521), (2, 4000, 1600, dict(row_nnz=40), 127),
                                              (0, 2500, 2500, dict(density=2e-3), 0xfffffffb)])
def test_rounds_in_row_batches_when_memory_is_short(S, O, monkeypatch, kind, n, m, kw, prime):
    """A round whose multiplier records / Schur slots exceed the device memory is reduced in batches of rows that are
    appended to the next round's matrix. Forced here by a 16 MB budget; U, rank and kernel must not change."""
    monkeypatch.setenv("SPASM_AMD_ROUND_STATS", "1")  # exact trip counters: the rounds keep to the multiplier lists
    A = S.synth_csr(kind, n, m, prime=prime, seed=0xB47C4, **kw)
    ref = S.echelonize(A, enable_dense=False, **LM)
    ref_rounds = S.last_rounds()
    monkeypatch.setenv("SPASM_AMD_MEM_BUDGET_MB", "16")
    try:
        got = S.echelonize(A, enable_dense=False, **LM)
        got_rounds = S.last_rounds()
    finally:
        monkeypatch.delenv("SPASM_AMD_MEM_BUDGET_MB")
    assert got.r == ref.r == O.echelonize(A, **LM).r
    assert np.asarray(got.qinv).tolist() == np.asarray(ref.qinv).tolist()
    assert len(got_rounds) == len(ref_rounds)
    for a, b in zip(got_rounds, ref_rounds):  # same pivots, same eliminations, same Schur complements round by round
        for key in ("npiv", "nnz_reduced", "applications", "nnz_out", "rows_out"):
            assert a[key] == b[key], (key, a, b)
    assert got.U.rows() == ref.U.rows()
    assert S.kernel(got).rows() == S.kernel(ref).rows()


@pytest.mark.parametrize("n,m,p,density,seed", [(60, 80, 65521, 0.08, 1), (120, 90, 127, 0.05, 2), (90, 140, 2147483647, 0.06, 3),
                                                (200, 260, 42013, 0.02, 4), (40, 70, 0xfffffffb, 0.1, 5)])
def test_rref_is_the_unique_reduced_echelon_form(S, O, n, m, p, density, seed):
    """rref(fact) (reference src/SpaSM.jl:871) against an independent dense Gauss-Jordan elimination: the reduced row echelon
    form of a matrix is unique, so the rows must agree entry for entry (row k of R is the row whose pivot is Rqinv^-1(k))."""
    rng = np.random.default_rng(seed)
    D = ((rng.random((n, m)) < density) * rng.integers(1, min(p, 1 << 31), size=(n, m))).astype(np.int64)
    D[n - 1] = (3 * D[0] + 5 * D[1]) % p  # rank deficiency
    A = S.CSR(D.T.copy(), prime=p)         # CSR(dense) stores the transpose, like the reference: rows of A = rows of D
    fact = S.echelonize(A, **LM)
    R, rq = S.rref(fact)
    want, piv = O.dense_rref(D, p)
    assert R.n == fact.r == len(piv)
    assert sorted(np.flatnonzero(rq >= 0).tolist()) == piv
    got = np.zeros((R.n, m), dtype=object)
    for k, row in enumerate(R.rows()):
        assert row[0][1] == 1                                  # the pivot comes first and is 1
        for c, v in row:
            got[k, c] = v % p
    for a, c in enumerate(piv):                                # dense_rref orders its rows by pivot column
        assert (got[int(rq[c])] == np.asarray(want[a], dtype=object)).all()
    assert S.factorization_verify(A, S.LU.from_parts(S.CSR.from_rows(R.rows(), m, p), rq, np.full(max(n, m), -1, dtype=np.int32)), 1)


@pytest.mark.parametrize("n,m,p,seed", [(400, 500, 65521, 1), (300, 360, 127, 2), (350, 420, 2147483647, 3), (200, 260, 0xfffffffb, 4)])
def test_sparse_triangular_solve_all_rows_at_once(S, O, n, m, p, seed):
    """sparse_triangular_solve(LU, B) (reference src/SpaSM.jl:725-755, semantics :694-713): X * U == B checked exactly with
    integers; rows outside the row space are reported unsolvable; the oracle's row-by-row solve gives the same x_b."""
    A = S.synth_csr(1, n, m, row_nnz=5, prime=p, seed=seed)
    fact = S.echelonize(A, **LM)
    U, qinv = fact.U, np.asarray(fact.qinv)
    r = fact.r
    rng = np.random.default_rng(seed)
    Urows = U.rows()
    # B: 40 random combinations of rows of U (solvable) + the rows of A themselves (solvable) + 5 rows pushed outside
    coeff = [{int(k): int(v) for k, v in zip(rng.choice(r, size=6, replace=False), rng.integers(1, min(p, 1 << 31), size=6))} for _ in range(40)]
    def combo(cf):
        acc = {}
        for k, v in cf.items():
            for c, x in Urows[k]:
                acc[c] = (acc.get(c, 0) + v * x) % p
        return sorted((c, v) for c, v in acc.items() if v)
    free = [j for j in range(m) if qinv[j] < 0]
    rowsB = [combo(cf) for cf in coeff] + A.rows()
    bad = []
    for t in range(5):
        row = dict(rowsB[t])
        row[free[t]] = (row.get(free[t], 0) + 1) % p   # + e_f, f a free column: every vector of the row space leads on a pivot column
        bad.append(sorted((c, v) for c, v in row.items() if v))
    B = S.CSR.from_rows(rowsB + bad, m, p)
    X, ok = S.api._triangular_solve(U, B, qinv)
    assert X.shape == (B.n, r)
    Xrows = X.rows()
    for kk in range(B.n):
        acc = {}
        for k, v in Xrows[kk]:
            for c, x in Urows[k]:
                acc[c] = (acc.get(c, 0) + v * x) % p
        got = {c: v for c, v in acc.items() if v}
        target = {c: v % p for c, v in (rowsB + bad)[kk]}
        if ok[kk]:
            assert got == target                              # X[k] * U == B[k] exactly
        else:
            assert got != target
    assert ok[: len(rowsB)].all() and not ok[len(rowsB):].any()
    for t, cf in enumerate(coeff):                             # the combination is recovered (U's rows are independent)
        assert {k: v % p for k, v in Xrows[t]} == {k: v % p for k, v in cf.items()}
    assert S.sparse_triangular_solve(fact, S.CSR.from_rows(rowsB, m, p)) is not None
    assert S.sparse_triangular_solve(fact, B) is None


def test_kernel_of_a_strided_subset_of_the_free_columns(S):
    """spasm_amd_kernel_strided (the multi-GPU kernel step): vectors first, first + step, ... of the whole basis."""
    A = S.synth_csr(0, 900, 1100, density=4e-3, prime=42013, seed=0xFEED)
    fact = S.echelonize(A, **LM)
    K = S.kernel(fact).rows()
    assert len(K) == A.m - fact.r > 50
    lib = S._abi.lib()
    for first, step in ((0, 1), (0, 3), (2, 3), (5, 7), (len(K) - 1, 2), (len(K) + 3, 2)):
        ptr = lib.spasm_amd_kernel_strided(fact.data, first, step)
        assert ptr, S._abi.last_error()
        assert S.CSR(ptr).rows() == K[first::step]


@pytest.mark.parametrize("kind,n,m,kw,prime", [
    (0, 900, 1100, dict(density=4e-3), 42013),          # a random matrix: the closure is most of U
    (2, 3000, 1200, dict(row_nnz=30), 127),             # Macaulay-like: dense tail, sparse kernel
    (1, 2000, 2600, dict(row_nnz=3), 65521),            # very sparse: many rows outside the closure
    (0, 300, 500, dict(density=2e-3), 42013),           # empty columns: kernel vectors of one entry, closure nearly empty
])
def test_kernel_through_the_closure_of_the_free_columns(S, O, monkeypatch, kind, n, m, kw, prime):
    """A U too large for the device as one round (32-bit offsets: 2^32 entries) is reduced on the host to the rows a kernel vector can
    be non-zero on (kernel_closure, engine.hip) and the same solve runs on what is left.  SPASM_AMD_KERNEL_REDUCE_NNZ=0 sends
    every U through that path: the basis must be the one the whole U gives, vector for vector, and the oracle's."""
    A = S.synth_csr(kind, n, m, prime=prime, seed=0xC105, **kw)
    fact = S.echelonize(A, **LM)
    whole = S.kernel(fact).rows()
    monkeypatch.setenv("SPASM_AMD_KERNEL_REDUCE_NNZ", "0")
    try:
        K = S.kernel(fact)
        lib = S._abi.lib()
        parts = []
        for first in range(3):
            ptr = lib.spasm_amd_kernel_strided(fact.data, first, 3)
            assert ptr, S._abi.last_error()
            parts.append(S.CSR(ptr).rows())
    finally:
        monkeypatch.delenv("SPASM_AMD_KERNEL_REDUCE_NNZ")
    rows = K.rows()
    assert len(rows) == A.m - fact.r
    assert rows == whole
    assert rows == O.kernel(O.echelonize(A, **LM)).rows()
    for first in range(3):
        assert parts[first] == whole[first::3]             # the closure of a SUBSET of the free columns
    # A * k^T == 0, exact integers
    Arows = A.rows()
    for k in rows[:: max(1, len(rows) // 40)]:
        kd = dict(k)
        for row in Arows[:: max(1, len(Arows) // 200)]:
            assert sum(v * kd.get(c, 0) for c, v in row) % prime == 0


@pytest.mark.parametrize("tail", ["0", "100", "1000000000", None], ids=["no_dense_tail", "tail_of_100_rows", "everything_dense", "tail_by_density"])
@pytest.mark.parametrize("kind,n,m,kw,prime,opts", [
    (0, 900, 1100, dict(density=4e-3), 42013, LM),           # several sparse rounds + a dense finish
    (2, 3000, 1200, dict(row_nnz=30), 127, LM),              # Macaulay-like, bytes: dense tail from the tall finish
    (1, 2000, 2600, dict(row_nnz=3), 65521, LM),             # very sparse, deep pivot graph, many free columns (shorts)
    (0, 300, 500, dict(density=2e-3), 42013, LM),            # empty columns
    (2, 3000, 1200, dict(row_nnz=30), 127, {}),              # default options: pivots that are not leftmost entries
    (0, 700, 900, dict(density=6e-3), 251, {}),              # p = 251: residues fill the byte
])
def test_kernel_through_a_dense_right_hand_side(S, O, monkeypatch, kind, n, m, kw, prime, opts, tail):
    """spasm_kernel with all free columns at once (csrc/kernel_dense.hpp: the dense tail of U through the reduced form on the int8
    GEMM, the sparse rows level by level, K = the transpose) -- the path of a U too large for the device as one round, forced here
    at small sizes with every split between dense tail and sparse rows.  Must give the basis of the sparse solves, vector for vector."""
    A = S.synth_csr(kind, n, m, prime=prime, seed=0xD3A5E, **kw)
    fact = S.echelonize(A, **opts)
    whole = S.kernel(fact).rows()
    monkeypatch.setenv("SPASM_AMD_KERNEL_DENSE_RHS", "1")
    if tail is not None:
        monkeypatch.setenv("SPASM_AMD_KERNEL_DENSE_TAIL", tail)
    try:
        rows = S.kernel(fact).rows()
        lib = S._abi.lib()
        parts = []
        for first in range(2):
            ptr = lib.spasm_amd_kernel_strided(fact.data, first, 2)
            assert ptr, S._abi.last_error()
            parts.append(S.CSR(ptr).rows())
    finally:
        monkeypatch.delenv("SPASM_AMD_KERNEL_DENSE_RHS")
        if tail is not None:
            monkeypatch.delenv("SPASM_AMD_KERNEL_DENSE_TAIL")
    assert len(rows) == A.m - fact.r
    assert rows == whole
    assert parts[0] == whole[0::2] and parts[1] == whole[1::2]
    if not opts:
        return
    assert rows == O.kernel(O.echelonize(A, **LM)).rows()


def test_rref_of_a_multi_round_factorization(S, O):
    """U of several sparse rounds plus a dense tail (config-2 style, scaled down): R must have no entry on a foreign pivot
    column, span the same space (verify), and reproduce the kernel through the textbook formula k[piv(a)] = R[a][j]."""
    A = S.synth_csr(0, 1500, 1600, density=3e-3, prime=42013, seed=0xABCD)
    fact = S.echelonize(A, **LM)
    assert len(S.last_rounds()) >= 2
    R, rq = S.rref(fact)
    assert R.n == fact.r
    pivcols = set(np.flatnonzero(rq >= 0).tolist())
    for k, row in enumerate(R.rows()):
        assert row[0][1] == 1 and int(rq[row[0][0]]) == k
        assert not any(c in pivcols for c, _ in row[1:])
    assert S.factorization_verify(A, S.LU.from_parts(S.CSR.from_rows(R.rows(), A.m, 42013), rq, np.full(max(A.n, A.m), -1, dtype=np.int32)), 3)
    K = S.kernel(fact)
    p = 42013
    Rrows = R.rows()
    kr = K.rows()
    free = [j for j in range(A.m) if rq[j] < 0]
    assert len(kr) == len(free)
    for f in range(0, len(free), max(1, len(free) // 25)):
        j = free[f]
        want = {j: -1}
        for k, row in enumerate(Rrows):
            for c, v in row[1:]:
                if c == j:
                    want[row[0][0]] = v
        assert dict(kr[f]) == want


# ---- one Schur round (the benchmark's unit of work) vs the oracle -------------------------------

def run_plan(S, A, lo=0, hi=None, stride=1):
    lib = S._abi.lib()
    hi = A.n if hi is None else hi
    plan = lib.spasm_amd_schur_plan_create_strided(A.data, lo, hi, stride)
    assert plan, S._abi.last_error()
    try:
        assert lib.spasm_amd_schur_plan_run(plan, None) == 0, S._abi.last_error()
        st = S._abi.RoundStats()
        assert lib.spasm_amd_schur_plan_stats(plan, C.byref(st)) == 0, S._abi.last_error()
        p_out = np.empty(max(A.n, 1), dtype=np.int32)
        ptr = lib.spasm_amd_schur_plan_fetch(plan, p_out.ctypes.data_as(C.POINTER(C.c_int32)))
        assert ptr, S._abi.last_error()
        Sc = S.CSR(ptr)
        return Sc, st.as_dict(), p_out[: Sc.n]
    finally:
        lib.spasm_amd_schur_plan_free(plan)


@pytest.mark.parametrize("n,k,p,seed", [(2000, 6, 65521, 21), (20000, 20, 65521, 0x5A5A0003), (5000, 12, 127, 23), (4000, 10, 2147483647, 24)])
def test_schur_round_vs_oracle(S, O, n, k, p, seed):
    A = S.synth_csr(1, n, n, row_nnz=k, prime=p, seed=seed)
    Sc, st, p_out = run_plan(S, A)
    So, info = O.schur_round(A)
    assert st["npiv"] == info["npiv"]
    assert st["applications"] == info["applications"]
    assert st["nnz_reduced"] == info["nnz_reduced"]  # the unit of work, counted on both sides
    assert st["nnz_out"] == info["nnz_out"] and st["rows_out"] == info["rows_out"]
    assert Sc.n == So.n
    assert np.all(np.diff(p_out) > 0)  # rows keep the input order
    assert Sc.rows() == So.rows()


@pytest.mark.parametrize("n,k,p,seed", [(20000, 20, 65521, 0x5A5A0003), (6000, 40, 127, 29), (4000, 10, 2147483647, 24)])
def test_schur_round_with_chunks_of_128_entries(S, O, monkeypatch, n, k, p, seed):
    """SPASM_AMD_CHUNK=128: the plan cuts the runs of W into chunks of 128 entries and a lane of the streaming kernel takes two entries
    per chunk (one 16-byte load; VERDICT r3 #2b).  Slower on config 3, kept for matrices with long rows of W:
    the Schur complement must be the oracle's whichever chunk size runs."""
    A = S.synth_csr(1, n, n, row_nnz=k, prime=p, seed=seed)
    monkeypatch.setenv("SPASM_AMD_CHUNK", "128")
    try:
        Sc, st, p_out = run_plan(S, A)
    finally:
        monkeypatch.delenv("SPASM_AMD_CHUNK")
    So, info = O.schur_round(A)
    assert st["npiv"] == info["npiv"] and st["nnz_out"] == info["nnz_out"] and st["rows_out"] == info["rows_out"]
    assert Sc.rows() == So.rows()


@pytest.mark.parametrize("n,k,p,seed", [(20000, 20, 65521, 0x5A5A0003), (6000, 40, 127, 29), (3000, 60, 7, 31)])
def test_schur_round_with_the_duplicate_check_as_a_filter(S, O, monkeypatch, n, k, p, seed):
    """SPASM_AMD_BLOOM=1: the streaming kernels check for duplicate columns with a bitmap filter (three bits of one word per entry, one
    returning ds_or) instead of the three exact tag tables, and resolve the suspects against the row's input at the end of the row.
    Exact -- the Schur complement must be the oracle's, with many duplicates (p = 7, 60 per row) and few -- and slower on config 3
    (DESIGN.md section 8): kept as a measured alternative."""
    A = S.synth_csr(1, n, n, row_nnz=k, prime=p, seed=seed)
    monkeypatch.setenv("SPASM_AMD_BLOOM", "1")
    try:
        Sc, st, p_out = run_plan(S, A)
    finally:
        monkeypatch.delenv("SPASM_AMD_BLOOM")
    So, info = O.schur_round(A)
    assert st["npiv"] == info["npiv"] and st["nnz_out"] == info["nnz_out"] and st["rows_out"] == info["rows_out"]
    assert Sc.rows() == So.rows()


def test_config3_full_size_counters_and_sampled_rows(S, O):
    """BASELINE config 3 at FULL size (1M x 1M, 20 nnz/row, p = 65521), the workload bench.py times.
    (a) The counters of the whole round equal the oracle's, committed as tests/golden/config3_oracle_counts.json by
        tests/golden/make_config3_counts.py (the oracle needs ~15 s and 10 GB for the full round, too much for this suite).
    (b) Three row ranges spread over the matrix are reduced as shards on the GPU and by the oracle here, with the pivots
        of the whole matrix: their Schur rows must agree entry for entry.
    (c) Size-independent properties of the full result: rows keep their order, no entry sits on a pivot column, the
        leftmost column recorded for a row is its smallest column."""
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config3_oracle_counts.json")))
    n = 1_000_000
    A = S.synth_csr(1, n, n, row_nnz=20, prime=65521, seed=0x5A5A0003)
    # (b) sampled shards first (small results)
    got_rows = 0
    for lo, hi in ((0, 20000), (500000, 501500), (998500, 1000000)):
        Sc, st, p_out = run_plan(S, A, lo, hi)
        So, info = O.schur_round(A, row_lo=lo, row_hi=hi)
        assert st["npiv"] == info["npiv"] == gold["npiv"]
        assert st["nnz_reduced"] == info["nnz_reduced"] and st["applications"] == info["applications"]
        assert st["nnz_out"] == info["nnz_out"]
        assert Sc.n == So.n and Sc.rows() == So.rows()
        assert np.all((p_out >= lo) & (p_out < hi)) and np.all(np.diff(p_out) > 0)
        got_rows += Sc.n
    assert got_rows > 3000, got_rows
    # (a) + (c) the whole round
    lib = S._abi.lib()
    plan = lib.spasm_amd_schur_plan_create(A.data, 0, n)
    assert plan, S._abi.last_error()
    try:
        assert lib.spasm_amd_schur_plan_run(plan, None) == 0, S._abi.last_error()
        st = S._abi.RoundStats()
        assert lib.spasm_amd_schur_plan_stats(plan, C.byref(st)) == 0, S._abi.last_error()
        d = st.as_dict()
        for key in ("npiv", "applications", "nnz_reduced", "nnz_out", "rows_out"):
            assert d[key] == gold[key], (key, d[key], gold[key])
        p_out = np.empty(n, dtype=np.int32)
        ptr = lib.spasm_amd_schur_plan_fetch(plan, p_out.ctypes.data_as(C.POINTER(C.c_int32)))
        assert ptr, S._abi.last_error()
        Sc = S.CSR(ptr)
    finally:
        lib.spasm_amd_schur_plan_free(plan)
    assert Sc.n == n - gold["npiv"] and S.nnz(Sc) == gold["nnz_out"]
    assert np.all(np.diff(p_out[: Sc.n]) > 0)
    # pivot columns of the round = leftmost columns elected: recompute them on the host from A (min (len, row) per leftmost column)
    Ap, Aj = A.p, A.j[: S.nnz(A)]
    assert np.all(np.diff(Ap) > 0)                     # no empty row in this generator
    lead = np.minimum.reduceat(Aj, Ap[:-1])
    is_pivot_col = np.zeros(n, dtype=bool)
    is_pivot_col[lead] = True
    assert int(is_pivot_col.sum()) == gold["npiv"]
    assert not is_pivot_col[Sc.j[: S.nnz(Sc)]].any()    # no Schur entry on a pivot column
    sp = Sc.p
    nonempty = np.flatnonzero(np.diff(sp) > 0)
    assert len(nonempty) == gold["rows_out"]


@pytest.mark.parametrize("N,nprobe", [(6000, 40), (40000, 6)], ids=["lds_dense_class", "global_memory_class"])
def test_schur_round_deep_chains(S, O, N, nprobe):
    """Pivot rows that chain through thousands of other pivots: the reach of a row exceeds the sorted-list class of
    the solve kernel (and Uinv is too dense to build), so the dense-vector classes run: in LDS up to ~30000 pivots
    per round, in global memory beyond."""
    p = 65521
    rng = np.random.default_rng(5)
    rows = []
    for i in range(N):  # bidiagonal pivot rows: pivot i chains into i+1, i+2, ...
        rows.append([(i, int(rng.integers(1, p))), (i + 1, int(rng.integers(1, p))), (N + 1 + int(rng.integers(0, 500)), int(rng.integers(1, p)))])
    for k in range(nprobe):  # probe rows entering the chain at various depths, heavier than the pivot rows
        c = int(rng.integers(0, N - 1)) if k else 0
        cols = sorted(set([c] + [int(x) for x in rng.integers(c, N + 501, size=6)]))
        rows.append([(cc, int(rng.integers(1, p))) for cc in cols])
    A = S.CSR.from_rows(rows, N + 501, prime=p)
    Sc, st, p_out = run_plan(S, A)
    So, info = O.schur_round(A)
    assert st["npiv"] == info["npiv"] == N
    assert st["applications"] == info["applications"] and st["nnz_reduced"] == info["nnz_reduced"]
    assert Sc.rows() == So.rows()


def test_schur_round_rows_beyond_every_lds_table(S, O):
    """Schur rows with more than 10240 distinct columns: the global-memory scatter class."""
    p = 65521
    m = 40000
    rng = np.random.default_rng(8)
    rows = []
    npv = 60
    for i in range(npv):  # heavy pivot rows: leading entry i, then 500 entries far to the right
        cols = sorted(set(int(x) for x in rng.integers(npv, m, size=500)))
        rows.append([(i, int(rng.integers(1, p)))] + [(c, int(rng.integers(1, p))) for c in cols])
    for k in range(12):  # probe rows hitting most pivots: heavier than any pivot row so they stay non-pivotal
        cols = sorted(set(list(range(0, npv, 1 + k % 2)) + [int(x) for x in rng.integers(npv, m, size=700)]))
        rows.append([(c, int(rng.integers(1, p))) for c in cols])
    A = S.CSR.from_rows(rows, m, prime=p)
    Sc, st, p_out = run_plan(S, A)
    So, info = O.schur_round(A)
    assert max(len(r) for r in So.rows()) > 10240
    assert st["npiv"] == info["npiv"] and st["nnz_reduced"] == info["nnz_reduced"] and st["nnz_out"] == info["nnz_out"]
    assert Sc.rows() == So.rows()


def test_schur_round_sharded_rows(S, O):
    """Row shards reduce independently against the same U (multi-GPU partitioning)."""
    A = S.synth_csr(1, 6000, 6000, row_nnz=10, prime=65521, seed=31)
    So, info = O.schur_round(A)
    full, st, p_full = run_plan(S, A)
    parts, origs, red = [], [], 0
    for lo, hi in [(0, 1500), (1500, 1501), (1501, 6000)]:
        Sc, st_s, p_out = run_plan(S, A, lo, hi)
        parts += Sc.rows()
        origs += p_out.tolist()
        assert st_s["npiv"] == info["npiv"]
        red += st_s["nnz_reduced"]
    assert origs == p_full.tolist()
    assert parts == full.rows() == So.rows()
    assert red == info["nnz_reduced"]
    # strided shards (rank r of 3: rows r, r+3, ...): balanced, and together the same rows again
    by_orig = {}
    red = 0
    for r in range(3):
        Sc, st_s, p_out = run_plan(S, A, r, A.n, 3)
        assert all(g % 3 == r for g in p_out.tolist())
        by_orig.update(zip(p_out.tolist(), Sc.rows()))
        red += st_s["nnz_reduced"]
    assert [by_orig[g] for g in p_full.tolist()] == So.rows() and red == info["nnz_reduced"]


def test_schur_round_is_idempotent_on_rerun(S):
    A = S.synth_csr(1, 3000, 3000, row_nnz=8, prime=65521, seed=41)
    lib = S._abi.lib()
    plan = lib.spasm_amd_schur_plan_create(A.data, 0, A.n)
    assert plan
    try:
        outs = []
        for _ in range(3):
            assert lib.spasm_amd_schur_plan_run(plan, None) == 0
            st = S._abi.RoundStats()
            assert lib.spasm_amd_schur_plan_stats(plan, C.byref(st)) == 0
            Sc = S.CSR(lib.spasm_amd_schur_plan_fetch(plan, None))
            outs.append((st.nnz_reduced, st.nnz_out, Sc.rows()))
        assert outs[0] == outs[1] == outs[2]
    finally:
        lib.spasm_amd_schur_plan_free(plan)


# ---- randomized differential campaign: tiny primes make cancellations, zero multipliers and empty Schur rows common ----

@pytest.mark.parametrize("enable_dense", [False, True], ids=["sparse_rounds_only", "dense_tail"])
def test_fuzz_small_primes_vs_dense_elimination(S, O, enable_dense):
    rng = np.random.default_rng(20260930 + int(enable_dense))
    cases = 0
    for trial in range(90):
        p = int(rng.choice([3, 5, 7, 11, 127, 251]))
        n, m = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        density = float(rng.choice([0.03, 0.08, 0.2, 0.5]))
        D = (rng.random((n, m)) < density) * rng.integers(1, p, size=(n, m))
        if n > 3 and rng.random() < 0.5:  # planted dependencies and duplicate rows
            D[n - 1] = (D[0] + 2 * D[1]) % p
            D[n - 2] = D[2]
        if m > 4 and rng.random() < 0.3:
            D[:, m - 1] = 0
        A = S.CSR(D.T.copy(), prime=p)
        fact = S.echelonize(A, enable_dense=enable_dense, **LM)
        Kd, piv = O.dense_kernel_normal_form(D, p)
        assert fact.r == len(piv), (trial, p, n, m)
        assert sorted(np.nonzero(np.asarray(fact.qinv) >= 0)[0].tolist()) == piv, (trial, p, n, m)
        K = S.kernel(fact)
        assert (K.todense() == Kd).all(), (trial, p, n, m)
        # U: unit pivots and rows inside the row space of A
        Ud = fact.U.todense() % p
        for a in range(fact.r):
            assert Ud[a, [j for j in range(m) if fact.qinv[j] == a][0]] == 1
        assert len(O.dense_rref(np.vstack([D % p, Ud]), p)[1]) == fact.r
        cases += 1
    assert cases == 90


def test_fuzz_schur_round_small_primes(S, O):
    """One Schur round against the oracle on matrices where multipliers and accumulators cancel to zero often."""
    rng = np.random.default_rng(77)
    for trial in range(25):
        p = int(rng.choice([3, 5, 7]))
        n, m, k = int(rng.integers(50, 400)), int(rng.integers(50, 400)), int(rng.integers(2, 9))
        A = S.synth_csr(1, n, m, row_nnz=min(k, m), prime=p, seed=1000 + trial)
        Sc, st, p_out = run_plan(S, A)
        So, info = O.schur_round(A)
        assert (st["npiv"], st["applications"], st["nnz_reduced"], st["nnz_out"]) == \
               (info["npiv"], info["applications"], info["nnz_reduced"], info["nnz_out"]), (trial, p, n, m, k)
        assert Sc.rows() == So.rows(), (trial, p, n, m, k)


@pytest.mark.gpu
def test_round_loop_options_change_the_rounds_not_the_result(S, O):
    """max_round and min_pivot_proportion (reference src/SpaSM.jl:333-335; the stop at README.md:32) decide how many sparse
    rounds run before the finish takes over; every pivot stays a leftmost entry, so rank, pivot columns and kernel do not
    move.  The round records (spasm_amd_last_rounds) must show the difference."""
    A = S.synth_csr(1, 3000, 3000, row_nnz=6, prime=65521, seed=0x0917)
    ref = S.echelonize(A, max_round=50, min_pivot_proportion=0.0, enable_dense=False, **LM)  # sparse rounds to the end
    n_ref = len(S.last_rounds())
    one = S.echelonize(A, max_round=1, min_pivot_proportion=0.0, **LM)
    n_one = len(S.last_rounds())
    none = S.echelonize(A, max_round=50, min_pivot_proportion=0.9, **LM)
    n_none = len(S.last_rounds())
    dflt = S.echelonize(A, **LM)
    assert n_ref > 2 and n_one == 1 and n_none == 0, (n_ref, n_one, n_none)
    olu = O.echelonize(A, **LM)
    for f in (ref, one, none, dflt):
        assert f.r == olu.r
        assert np.asarray(f.qinv >= 0).tolist() == np.asarray(olu.qinv >= 0).tolist()
    Kref = S.kernel(ref).rows()
    assert S.kernel(one).rows() == Kref and S.kernel(none).rows() == Kref and S.kernel(dflt).rows() == Kref == O.kernel(olu).rows()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,m,kw,prime", [(2, 4000, 1600, dict(row_nnz=40), 127), (1, 1500, 1200, dict(row_nnz=30), 65521),
                                              (0, 900, 700, dict(density=0.05), 2147483647)])
def test_schur_complement_straight_to_dense(S, O, kind, n, m, kw, prime):
    """When the density estimate (spasm_schur_estimate_density, prototype src/SpaSM.jl:763-764) says the Schur complement of a
    round is dense, it is reduced straight into the dense matrix of the finish (spasm_schur_dense, :765-766) and never built
    sparse.  The threshold sits between the density of the input (0.025 .. 0.05) and that of its first Schur complement; rank,
    pivot columns and kernel must equal the oracle's, and the record of the last round must show that no sparse Schur
    complement was counted."""
    A = S.synth_csr(kind, n, m, prime=prime, seed=0xD35E, **kw)
    fact = S.echelonize(A, sparsity_threshold=0.1, **LM)
    rounds = S.last_rounds()
    assert len(rounds) >= 1 and rounds[-1]["nnz_out"] == -1, rounds
    olu = O.echelonize(A, **LM)
    assert fact.r == olu.r
    assert np.asarray(fact.qinv >= 0).tolist() == np.asarray(olu.qinv >= 0).tolist()
    assert S.kernel(fact).rows() == O.kernel(olu).rows()
    assert S.factorization_verify(A, fact, 7)


# ---- "Faugère-Lachartre on columns" (enable_greedy_pivot_search, reference src/SpaSM.jl:326; log line README.md:22) -------------

GREEDY_CASES = [
    ("fixed_nnz", 1, 3000, 3000, dict(row_nnz=6), 65521),
    ("three_per_row", 1, 8000, 8000, dict(row_nnz=3), 65521),
    ("macaulay_like", 2, 4000, 1600, dict(row_nnz=40), 127),
    ("bernoulli", 0, 900, 1300, dict(density=0.01), 2147483647),
    ("wide_big_prime", 1, 2500, 4000, dict(row_nnz=8), 0xFFFFFFFB),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,n,m,kw,prime", GREEDY_CASES, ids=[c[0] for c in GREEDY_CASES])
def test_fl_on_columns_takes_the_oracles_pivots(S, O, name, kind, n, m, kw, prime):
    """With enable_greedy_pivot_search the sparse rounds also take pivots that are not leftmost entries: on columns no pivot row
    touches, chosen by column occupancy (kernels.hpp k_close_cols ..; oracle fl_pivots_ex).  The rule is order-free, so engine
    and oracle must elect the same pivots round by round: with the dense finish off both go through the same max_round rounds
    and finish on leftmost entries, which makes rank, pivot columns, the rows of U from the sparse rounds and the kernel basis
    comparable entry for entry."""
    A = S.synth_csr(kind, n, m, prime=prime, seed=0xF1C0, **kw)
    fact = S.echelonize(A, enable_greedy_pivot_search=True, enable_dense=False)
    rounds = S.last_rounds()
    olu = O.echelonize(A, enable_greedy_pivot_search=True)
    plain = S.echelonize(A, enable_greedy_pivot_search=False, enable_dense=False)
    assert fact.r == olu.r == plain.r
    assert np.asarray(fact.qinv >= 0).tolist() == np.asarray(olu.qinv >= 0).tolist()
    sparse_rounds = [r for r in rounds if r["round"] < 3]
    assert sum(r["npiv_open"] + r["npiv_greedy"] for r in sparse_rounds) > 0, rounds               # the searches found something
    assert all(r["npiv_open"] + r["npiv_greedy"] == 0 for r in rounds if r["round"] >= 3)          # the finish keeps to leftmost entries
    k = sum(r["npiv"] for r in sparse_rounds)
    assert fact.U.rows()[:k] == olu.U.rows()[:k]                             # same pivot rows, same order, same values
    assert S.kernel(fact).rows() == O.kernel(olu).rows()
    assert S.factorization_verify(A, fact, 11)
    # some pivot is NOT the leftmost entry of its row
    q = np.asarray(fact.qinv)
    pc = {int(q[j]): j for j in range(m) if q[j] >= 0}
    assert any(min(c for c, _ in row) != pc[a] for a, row in enumerate(fact.U.rows()[:k]))


@pytest.mark.gpu
def test_fl_on_columns_finds_more_pivots_per_round(S, O):
    """What the search is for: more structural pivots per round, fewer rounds.  On the 3-per-row matrix
    the first round must elect visibly more pivots than the leftmost-entry election alone, and the whole run must not take
    more rounds; the rank is that of the oracle."""
    A = S.synth_csr(1, 100000, 100000, row_nnz=3, prime=65521, seed=0xF1C1)
    g = S.echelonize(A, enable_greedy_pivot_search=True)
    rg = S.last_rounds()
    l = S.echelonize(A, enable_greedy_pivot_search=False)
    rl = S.last_rounds()
    assert g.r == l.r
    assert rg[0]["npiv"] - rg[0]["npiv_open"] - rg[0]["npiv_greedy"] == rl[0]["npiv"]   # the leftmost election is the same
    assert rg[0]["npiv_open"] > 0.05 * rl[0]["npiv"], (rg[0], rl[0])
    assert len(rg) <= len(rl)
    assert S.factorization_verify(A, g, 3) and S.factorization_verify(A, l, 3)
    Kg = S.kernel(g)
    assert Kg.n == A.m - g.r


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,p,density,seed", [(60, 80, 65521, 0.08, 1), (120, 90, 127, 0.05, 2), (90, 140, 2147483647, 0.06, 3)])
def test_consumers_accept_pivots_that_are_not_leftmost(S, O, n, m, p, density, seed):
    """rref, the triangular solve and the kernel number the pivots in a topological order of U (engine.hip
    pivot_topological_order), so a factorization from the "FL on columns" search -- or the oracle's -- works like any other:
    R has the unit vectors on the pivot columns and the same row space; X * U == B; K spans the right kernel."""
    rng = np.random.default_rng(seed)
    D = ((rng.random((n, m)) < density) * rng.integers(1, min(p, 1 << 31), size=(n, m))).astype(np.int64)
    D[n - 1] = (3 * D[0] + 5 * D[1]) % p
    A = S.CSR(D.T.copy(), prime=p)
    fact = S.echelonize(A, enable_greedy_pivot_search=True, enable_dense=False)
    olu = O.echelonize(A, enable_greedy_pivot_search=True)
    assert fact.r == olu.r and np.asarray(fact.qinv >= 0).tolist() == np.asarray(olu.qinv >= 0).tolist()
    q = np.asarray(fact.qinv)
    piv = [j for j in range(m) if q[j] >= 0]
    R, rq = S.rref(fact)
    Rd = np.zeros((R.n, m), dtype=object)
    for k, row in enumerate(R.rows()):
        for c, v in row:
            Rd[k, c] = v % p
    for j in piv:                                                            # identity on the pivot columns
        col = Rd[:, j]
        assert col[int(rq[j])] == 1 and sum(1 for v in col if v) == 1
    _, pr = O.dense_rref(np.vstack([D % p, np.array(Rd.tolist(), dtype=np.int64)]), p)
    assert len(pr) == fact.r == R.n                                           # same row space
    # the same through the oracle's factorization (a foreign LU, rows in the oracle's order)
    foreign = S.LU.from_parts(S.CSR.from_rows(olu.U.rows(), m, p), np.asarray(olu.qinv, dtype=np.int32), np.full(max(n, m), -1, dtype=np.int32))
    R2, rq2 = S.rref(foreign)
    assert sorted(map(tuple, (tuple(r) for r in R2.rows()))) == sorted(map(tuple, (tuple(r) for r in R.rows())))
    assert S.kernel(foreign).rows() == S.kernel(fact).rows() == O.kernel(olu).rows()
    # X * U == B for the rows of A
    X = S.sparse_triangular_solve(fact, A)
    assert X is not None
    Ud = np.array(fact.U.todense().tolist(), dtype=object) % p
    Xd = np.array(X.todense().tolist(), dtype=object) % p
    assert ((Xd.dot(Ud) - (D % p)) % p == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [dict(SPASM_AMD_DENSE_KB="128"), dict(SPASM_AMD_PANEL_GLOBAL="1", SPASM_AMD_DENSE_KB="192"),
                                 dict(SPASM_AMD_MEM_BUDGET_MB="4"), dict(SPASM_AMD_DENSE_F64="1"),
                                 dict(SPASM_AMD_PANEL_RES_WGS="2", SPASM_AMD_PANEL_RES_ROWS="128", SPASM_AMD_DENSE_KB="128"),
                                 dict(SPASM_AMD_PANEL_RES_WGS="5", SPASM_AMD_PANEL_RES_ROWS="256")],
                         ids=["several_blocks", "panel_rows_in_global_memory", "dense_W_in_column_slabs", "f64_panels",
                              "tall_panels_few_resident_rows_many_redone", "tall_panels_followers"])
@pytest.mark.parametrize("kind,n,m,kw,prime", [(2, 4000, 1600, dict(row_nnz=40), 127), (1, 1500, 1200, dict(row_nnz=30), 65521)],
                         ids=["p127_one_digit", "p65521_two_digits"])
def test_dense_finish_variants(S, O, monkeypatch, env, kind, n, m, kw, prime):
    """The code paths of the dense finish that default sizes only reach on large inputs, forced on small ones: several
    1024-column blocks (two-level updates), the panel kernel working in global memory instead of LDS, the dense W built for a
    few columns at a time, the f64 panels primes above 2^16 use, and the tall-matrix panels (a few LDS-resident rows elect the
    pivots, the others follow; panels in which a follower should have been a pivot are redone in place).  Same rank, pivot columns and kernel as the oracle; the
    factorization verifies."""
    A = S.synth_csr(kind, n, m, prime=prime, seed=0xD35E, **kw)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    fact = S.echelonize(A, sparsity_threshold=0.1, **LM)
    rounds = S.last_rounds()
    for k in env:
        monkeypatch.delenv(k)
    if "SPASM_AMD_MEM_BUDGET_MB" not in env or kind == 2:
        assert rounds[-1]["nnz_out"] == -1, rounds   # the round went straight to the dense finish
    # (with a 4 MB budget the row sample of the p = 65521 case does not fit, so no estimate is made and the round is built sparse
    # before the finish takes it; the Macaulay-like case estimates from columns and builds its dense W in slabs)
    olu = O.echelonize(A, **LM)
    assert fact.r == olu.r
    assert np.asarray(fact.qinv >= 0).tolist() == np.asarray(olu.qinv >= 0).tolist()
    assert S.kernel(fact).rows() == O.kernel(olu).rows()
    assert S.factorization_verify(A, fact, 9)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,p,seed", [(60, 80, 65521, 1), (90, 70, 127, 2), (50, 120, 0xFFFFFFFB, 3)])
def test_sparse_triangular_solve_one_row_as_the_reference_binds_it(S, O, n, m, p, seed):
    """spasm_sparse_triangular_solve(U, B, k, xj, x, qinv) (reference src/SpaSM.jl:721, semantics :694-713) and spasm_scatter (:620):
    with x_b on the pivot columns and x_a on the others, x_b * U + x_a == B[k], checked with spasm_scatter itself and exact integers;
    the pattern is xj[top:m]; against the oracle's orc_sparse_triangular_solve value for value."""
    rng = np.random.default_rng(seed)
    D = ((rng.random((n, m)) < 0.08) * rng.integers(1, min(p, 1 << 31), size=(n, m))).astype(np.int64)
    A = S.CSR(D.T.copy(), prime=p)
    fact = S.echelonize(A, **LM)
    U, q = fact.U, np.asarray(fact.qinv, dtype=np.int32)
    B = S.CSR(((rng.random((7, m)) < 0.15) * rng.integers(1, min(p, 1 << 31), size=(7, m))).astype(np.int64).T.copy(), prime=p)
    Brows = B.rows()
    Urows = U.rows()
    for k in range(B.n):
        xj = np.zeros(3 * m, dtype=np.int32)
        x = np.full(m, 12345, dtype=np.int32)  # "does not need to be initialized"
        top = S.sparse_triangular_solve_row(U, B, k, xj, x, q)
        pat = xj[top:m].tolist()
        assert len(set(pat)) == len(pat) and not xj[m:].any()
        # x_b * U + x_a == B[k]
        acc = {}
        for j in pat:
            v = int(x[j])
            if q[j] >= 0:
                for c, u in Urows[int(q[j])]:
                    acc[c] = (acc.get(c, 0) + v * int(u)) % p
            else:
                acc[j] = (acc.get(j, 0) + v) % p
        want = {c: int(v) % p for c, v in Brows[k]}
        assert {c: v for c, v in acc.items() if v} == {c: v for c, v in want.items() if v}
        # the oracle's solve of the same row: same values on its pattern
        oxj = np.zeros(3 * m, dtype=np.int32)
        ox = np.zeros(m, dtype=np.int32)
        otop = O.sparse_triangular_solve(U, B, k, oxj, ox, q)
        for j in oxj[otop:m].tolist():
            if ox[j] != 0:
                assert j in pat and int(x[j]) == int(ox[j])
        assert {j for j in pat if x[j] != 0} == {j for j in oxj[otop:m].tolist() if ox[j] != 0}
    # spasm_scatter: x += beta * U[0]
    y = np.zeros(m, dtype=np.int32)
    S.scatter(U, 0, 3, y)
    assert {c: int(y[c]) % p for c in np.nonzero(y)[0]} == {c: (3 * int(v)) % p for c, v in Urows[0] if (3 * int(v)) % p}


@pytest.mark.gpu
def test_explicit_zero_entries_are_dropped_at_ingest(S, O):
    """A C caller may hand over entries whose value is 0 mod p (SpaSM.jl's constructors drop them, reference src/SpaSM.jl:959, :979).
    Elected as a pivot such an entry would give a silently wrong U; the engine drops them when the matrix is uploaded: the result is
    that of the matrix without them, also when the zero is the leftmost entry of its row or sits on a column no pivot row touches."""
    p = 65521
    rng = np.random.default_rng(5)
    D = ((rng.random((80, 90)) < 0.08) * rng.integers(1, p, size=(80, 90))).astype(np.int64)
    A = S.CSR(D.T.copy(), prime=p)
    clean = S.echelonize(A)
    Kc = S.kernel(clean).rows()
    # the same matrix with planted zeros: every row gets a zero-valued entry on column 0 (leftmost!) or on a fresh column
    rows = [list(r) for r in A.rows()]
    planted = []
    for i, r in enumerate(rows):
        have = {c for c, _ in r}
        c = 0 if 0 not in have else next(j for j in range(89, -1, -1) if j not in have)
        planted.append([(c, p)] + r if r else [(c, p)])  # p = 0 mod p
    B = S.CSR.from_rows(planted, 90, p)
    nz = S.nnz(B)
    if nz == S.nnz(A):  # the host mirror dropped them: plant through the arrays instead
        pytest.skip("from_rows drops zeros")
    got = S.echelonize(B)
    assert got.r == clean.r and np.asarray(got.qinv >= 0).tolist() == np.asarray(clean.qinv >= 0).tolist()
    assert S.kernel(got).rows() == Kc
    assert S.factorization_verify(A, got, 3)


@pytest.mark.gpu
def test_enable_gplu_off_without_a_dense_finish_stops_short(S, O):
    """echelonize_opts.enable_GPLU = 0 (reference src/SpaSM.jl:330) with the dense finish off leaves no method to finish with: like
    libspasm the engine returns what the sparse rounds found -- r is then a lower bound of the rank and U a partial echelon form
    whose rows lie in the row space of A; with GPLU on (the default) the same options finish."""
    A = S.synth_csr(1, 3000, 3000, row_nnz=6, prime=65521, seed=0x6F1)
    full = S.echelonize(A, enable_dense=False, **LM)
    short = S.echelonize(A, enable_dense=False, enable_GPLU=False, max_round=2, **LM)
    assert 0 < short.r < full.r == O.echelonize(A, **LM).r
    assert not S.factorization_verify(A, short, 1) and S.factorization_verify(A, full, 1)
    # the rows it has are rows of a correct echelon form: the same pivot columns as the full run's first rows
    k = short.r
    assert short.U.rows() == full.U.rows()[:k]


@pytest.mark.parametrize("p", [65521, 0xFFFFFFFB], ids=["small_prime", "large_prime"])
def test_schur_round_w_build_long_dependency_lists(S, O, p):
    """The corners of the level-wise W build (csrc/wlevel.hpp) that random sparse matrices never reach: pivot rows with more than 64
    entries on other pivot columns (left by the wave kernel to the workgroup kernel with the largest table), with more than 1024 of
    them (that kernel takes its dependencies in batches), and one whose row of W exceeds the largest table -- published as not
    available, so the rows that need it go through the multiplier lists.  Two levels, the second large enough to go through the
    wave kernel; the Schur complement must still be the oracle's, entry for entry."""
    rng = np.random.default_rng(11)
    n0, n1, m = 1500, 2100, 9000        # level-0 pivots on columns n1 .. n1 + n0 - 1, level-1 pivots on columns 0 .. n1 - 1
    val = lambda: int(rng.integers(1, min(p, 1 << 31)))  # noqa: E731
    free = lambda k: sorted(set(int(x) for x in rng.integers(n1 + n0, m, size=k)))  # noqa: E731
    rows = []
    for i in range(n0):                 # level 0: the pivot, then 10 entries on columns without a pivot
        rows.append([(n1 + i, val())] + [(c, val()) for c in free(10)])
    for k in range(n1):                 # level 1: the pivot, level-0 pivot columns (a few, ~90, ~300, or all 1500), a few free columns
        if k == 0:
            deps = list(range(n1, n1 + n0))
        else:
            many = 1100 if p < 65536 else 350   # (12-byte slots for the large primes: the largest table holds 4096 entries, not 8192)
            ndep = (3, 90, 300, many)[k % 4] if k % 50 else many
            deps = sorted(set(int(x) for x in rng.integers(n1, n1 + n0, size=ndep)))
        rows.append([(k, val())] + [(c, val()) for c in deps] + [(c, val()) for c in free(5)])
    npiv = n0 + n1
    for t in range(2 * npiv + 50):      # rows to reduce: longer than the pivot row of their leftmost column
        if t % 7 == 0:                  # enters at a level-1 pivot: needs ITS row of W (pivot 0's is not available: 1500 x 10 entries)
            c0 = (t // 7) % n1
            cols = sorted(set([c0] + [int(x) for x in rng.integers(n1, n1 + n0, size=1200)] + free(400)))
        else:
            c0 = n1 + int(rng.integers(0, n0))
            cols = sorted(set([c0] + [int(x) for x in rng.integers(c0, n1 + n0, size=3)] + free(12)))
        rows.append([(c, val()) for c in cols])
    A = S.CSR.from_rows(rows, m, prime=p)
    Sc, st, p_out = run_plan(S, A)
    So, info = O.schur_round(A)
    assert st["npiv"] == info["npiv"] == npiv
    assert st["w_levels"] == 2 and st["w_long_rows"] > n1 // 3, (st["w_levels"], st["w_long_rows"])
    assert st["applications"] == info["applications"] and st["nnz_reduced"] == info["nnz_reduced"] and st["nnz_out"] == info["nnz_out"]
    assert Sc.rows() == So.rows()


# ---- tall-and-skinny finish (enable_tall_and_skinny / tall_and_skinny_ratio, reference src/SpaSM.jl:327, :341; csrc/dense_tall.hpp) ------

def _low_rank_rows(rng, rows, rank, m, p):
    return (rng.integers(0, p, size=(rows, rank)).astype(np.int64).dot(rng.integers(0, p, size=(rank, m)).astype(np.int64))) % p


@pytest.mark.gpu
@pytest.mark.parametrize("p", [127, 65521], ids=["one_digit", "two_digits"])
@pytest.mark.parametrize("shape", ["slab_short_of_the_rank", "rank_below_the_columns", "slab_finds_every_column"])
def test_tall_and_skinny_finish_dense_input(S, O, monkeypatch, p, shape):
    """A dense remainder with many more rows than columns: a first slab of rows is eliminated, the others are reduced against its
    reduced form in one step and only their residuals eliminated (dense_tall.hpp).  The three shapes: the slab's rows span only part
    of the row space (the residuals carry pivots, also LEFT of pivots the slab found -- the pivot columns must still be the leading
    columns of the whole row space); the whole matrix is rank deficient (columns without pivot at the end); the slab has full column
    rank (the other rows are never looked at).  Rank, pivot columns and kernel of the oracle; several row batches; verified."""
    rng = np.random.default_rng(11)
    m = 400
    if shape == "slab_short_of_the_rank":
        M = np.vstack([_low_rank_rows(rng, 320, 100, m, p), rng.integers(0, p, size=(900, m))])
        M[:, 37] = 0
        M[400:, 37] = rng.integers(1, p, size=M.shape[0] - 400)          # a column the slab cannot see
    elif shape == "rank_below_the_columns":
        M = np.vstack([_low_rank_rows(rng, 320, 90, m, p), _low_rank_rows(rng, 900, 150, m, p)])
    else:
        M = rng.integers(0, p, size=(1500, m))
    A = S.CSR(M.astype(np.int64).T.copy(), prime=p)
    olu = O.echelonize(A, **LM)
    monkeypatch.setenv("SPASM_AMD_TALL_SLAB", "320" if shape != "slab_finds_every_column" else "448")
    monkeypatch.setenv("SPASM_AMD_TALL_BATCH", "256")
    fact = S.echelonize(A, verbose=False, **LM)
    monkeypatch.setenv("SPASM_AMD_TALL", "0")
    plain = S.echelonize(A, **LM)                                       # the same without the strategy
    for k in ("SPASM_AMD_TALL_SLAB", "SPASM_AMD_TALL_BATCH", "SPASM_AMD_TALL"):
        monkeypatch.delenv(k)
    assert fact.r == olu.r == plain.r
    assert np.asarray(fact.qinv >= 0).tolist() == np.asarray(olu.qinv >= 0).tolist() == np.asarray(plain.qinv >= 0).tolist()
    assert S.kernel(fact).rows() == O.kernel(olu).rows()
    assert S.factorization_verify(A, fact, 9)
    pr = np.asarray(fact.p)[: fact.r]
    assert len(set(pr.tolist())) == fact.r
    # enable_tall_and_skinny = 0 switches the strategy off like the reference's option
    off = S.echelonize(A, enable_tall_and_skinny=False, **LM)
    assert off.U.rows() == plain.U.rows()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [dict(), dict(SPASM_AMD_MEM_BUDGET_MB="4"), dict(SPASM_AMD_TALL_CHUNK="640"),
                                 dict(SPASM_AMD_MEM_BUDGET_MB="1", SPASM_AMD_TALL_CHUNK="1024")],
                         ids=["W_whole", "W_in_column_slabs", "residuals_in_row_chunks", "column_slabs_and_row_chunks"])
def test_tall_and_skinny_finish_of_a_schur_complement(S, O, monkeypatch, env):
    """The same strategy fed by the Schur rows of a round through the dense W (Macaulay-like: config 5's shape), rows materialised a
    slab / a batch at a time."""
    A = S.synth_csr(2, 6000, 1500, row_nnz=40, prime=127, seed=0x7A11)
    olu = O.echelonize(A, **LM)
    monkeypatch.setenv("SPASM_AMD_TALL", "1")
    monkeypatch.setenv("SPASM_AMD_TALL_SLAB", "256")
    monkeypatch.setenv("SPASM_AMD_TALL_BATCH", "512")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    fact = S.echelonize(A, sparsity_threshold=0.1, **LM)
    rounds = S.last_rounds()
    for k in ["SPASM_AMD_TALL", "SPASM_AMD_TALL_SLAB", "SPASM_AMD_TALL_BATCH"] + list(env):
        monkeypatch.delenv(k)
    assert rounds[-1]["nnz_out"] == -1, rounds                          # the round went straight to the dense finish
    assert fact.r == olu.r
    assert np.asarray(fact.qinv >= 0).tolist() == np.asarray(olu.qinv >= 0).tolist()
    assert S.kernel(fact).rows() == O.kernel(olu).rows()
    assert S.factorization_verify(A, fact, 9)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,m,kw,prime,env", [
    (1, 3000, 3000, dict(row_nnz=6), 65521, {}),
    (2, 6000, 1500, dict(row_nnz=40), 127, dict(SPASM_AMD_TALL="1", SPASM_AMD_TALL_SLAB="256")),
    (2, 4000, 1600, dict(row_nnz=40), 127, dict(SPASM_AMD_TALL="0")),
    (0, 900, 1300, dict(density=0.01), 2147483647, {}),
    (0, 500, 300, dict(density=0.4), 0xFFFFFFFB, {}),
], ids=["sparse_rounds_then_dense", "tall_finish", "dense_finish", "f64_panels", "rank1_updates"])
def test_rank_only_counts_what_echelonize_finds(S, O, monkeypatch, kind, n, m, kw, prime, env):
    """spasm_amd_rank (rank(A), reference src/SpaSM.jl:1149, without the rows of U on the host): the same rank as the factorization
    and as the oracle, through every finish."""
    A = S.synth_csr(kind, n, m, prime=prime, seed=0x4A4B, **kw)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    full = S.echelonize(A)
    r = S.rank(A, rank_only=True)
    r_lm = S.rank(A, rank_only=True, enable_greedy_pivot_search=False)
    for k in env:
        monkeypatch.delenv(k)
    assert r == r_lm == full.r == O.echelonize(A, **LM).r


@pytest.mark.gpu
def test_dense_finish_with_the_lds_dma_gemm(S, O, monkeypatch):
    """SPASM_AMD_GEMM_GLDS=1: the large one-digit updates on the 256 x 256 LDS-DMA kernel (dense.hpp k_gemm_i8_glds; off by default:
    it ties the 128 x 128 kernel since that one writes its tiles back row-wise).  A remainder large enough to reach it (more than 4096
    rows, more than 1024 columns right of the first block); rank, pivot columns and kernel of the default path; verified."""
    A = S.synth_csr(2, 14000, 4600, row_nnz=40, prime=127, seed=0x61D5)
    monkeypatch.setenv("SPASM_AMD_TALL", "0")
    ref = S.echelonize(A, **LM)
    monkeypatch.setenv("SPASM_AMD_GEMM_GLDS", "1")
    got = S.echelonize(A, **LM)
    monkeypatch.delenv("SPASM_AMD_GEMM_GLDS")
    monkeypatch.delenv("SPASM_AMD_TALL")
    assert S.last_rounds()[-1]["nnz_out"] == -1
    assert got.r == ref.r and np.asarray(got.qinv >= 0).tolist() == np.asarray(ref.qinv >= 0).tolist()
    assert got.U.rows() == ref.U.rows()          # the same elimination, other kernels for its large updates
    assert S.factorization_verify(A, got, 5)
