"""The options users actually get: SpaSM.jl's echelonize(A) leaves echelonize_opts at its defaults (reference src/SpaSM.jl:817,
:860-866), i.e. enable_greedy_pivot_search = 1 AND enable_dense = 1 -- "FL on columns" in the sparse rounds, then the dense finish
or the Schur complement straight to dense, with pivots that are not leftmost entries.  Engine and oracle then elect different
pivot columns in general (they finish differently), so what is compared is what SURVEY 8(c)(2) prescribes when pivot sets differ:
  * the rank, exactly, against the oracle AND an independent dense elimination;
  * the kernel as a SUBSPACE: the reduced row echelon form of K (unique for the subspace) equals that of the oracle's / the dense
    elimination's kernel basis; plus libspasm's normal form of K itself (K[f] = -1 on its free column, 0 on the other free columns,
    ascending free columns: test/runtests.jl:20-23) and A * k^T == 0 with exact integers;
  * U: unit pivots on the pivot columns, rows inside the row space of A and spanning it (spasm_factorization_verify + rank).
The same groups as tests/test_gpu_parity.py (echelonize/kernel vs oracle and dense, fuzz, config 2, config 5 scaled down), plus the
round-0 pivots of the engine against the INDEPENDENT restatement of the search (tests/fl_columns_ref.py)."""
import numpy as np
import pytest

import fl_columns_ref
from test_fl_columns_rule import pivots_of_first_round

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(S):
    assert S._abi.lib().spasm_amd_device_count() > 0, "no HIP device visible: GPU tests need the MI355X"


def rows_to_dense(rows, m, p):
    D = np.zeros((len(rows), m), dtype=np.int64 if p < (1 << 31) else object)
    for i, r in enumerate(rows):
        for c, v in r:
            D[i, c] = v % p
    return D


def kernel_normal_form_ok(K, qinv, m):
    free = [j for j in range(m) if qinv[j] < 0]
    rows = K.rows()
    assert len(rows) == len(free)
    fs = set(free)
    for f, row in zip(free, rows):
        on_free = [(c, v) for c, v in row if c in fs]
        assert on_free == [(f, -1)], (f, on_free)


def sparse_times_kernel_is_zero(S, A, K, p):
    """A * k^T == 0 for every row k of K, exact integers (python ints for the large primes)."""
    Krows = [dict(r) for r in K.rows()]
    if not Krows:
        return True
    for row in A.rows():
        for k in Krows:
            acc = 0
            for c, v in row:
                w = k.get(c)
                if w is not None:
                    acc += int(v) * int(w)
            if acc % p:
                return False
    return True


def check_default_run(S, O, A, D, p, oracle=True):
    """D: the dense matrix (rows of A), or None when it is too large for the dense checks.  oracle=False: the oracle's sequential
    finish takes minutes on the matrix; the caller compares the rank with the engine's leftmost-pivot run instead, which
    tests/test_gpu_parity.py holds against the oracle."""
    m = A.m
    fact = S.echelonize(A)  # the reference's defaults
    olu = O.echelonize(A) if oracle else None
    assert olu is None or fact.r == olu.r
    K = S.kernel(fact)
    q = np.asarray(fact.qinv)
    assert K.n == m - fact.r
    kernel_normal_form_ok(K, q, m)
    assert S.factorization_verify(A, fact, 17)
    if D is not None:
        R, piv = O.dense_rref(D, p)
        assert fact.r == len(piv)
        # the kernel as a subspace: rref(K) is unique
        Kd_want, _ = O.dense_kernel_normal_form(D, p)
        got = O.dense_rref(rows_to_dense(K.rows(), m, p), p)[0]
        want = O.dense_rref(np.mod(np.asarray(Kd_want, dtype=object if p >= (1 << 31) else np.int64), p), p)[0]
        assert got.shape == want.shape and (np.asarray(got) == np.asarray(want)).all()
        okd = O.dense_rref(rows_to_dense(O.kernel(olu).rows(), m, p), p)[0]
        assert okd.shape == want.shape and (np.asarray(okd) == np.asarray(want)).all()
        # U: unit pivots, rows inside the row space of A, and as many as its rank
        Ud = rows_to_dense(fact.U.rows(), m, p)
        for a in range(fact.r):
            assert Ud[a, int(np.nonzero(q == a)[0][0])] == 1
        assert len(O.dense_rref(np.vstack([np.mod(np.asarray(D, dtype=Ud.dtype), p), Ud]), p)[1]) == fact.r
    return fact, K, olu


@pytest.mark.parametrize("n,m,p,density,seed", [
    (12, 9, 7, 0.4, 1), (30, 40, 127, 0.15, 2), (60, 45, 42013, 0.08, 3), (80, 80, 65521, 0.05, 4),
    (50, 70, 0xFFFFFFFB, 0.1, 5), (40, 40, 3, 0.3, 6), (1, 17, 42013, 0.5, 7), (25, 1, 42013, 0.5, 8),
    (200, 150, 65537, 0.03, 9), (150, 220, 2147483647, 0.04, 10),
    (200, 300, 127, 0.3, 11), (260, 190, 65521, 0.25, 12),  # reach the dense finish for real (several panels)
])
def test_defaults_echelonize_kernel_vs_oracle_and_dense(S, O, n, m, p, density, seed):
    from test_oracle_golden import random_rows

    rng = np.random.default_rng(seed)
    D = random_rows(rng, n, m, p, density, rank_deficient=True)
    A = S.CSR(D.T.copy(), prime=p)
    fact, K, olu = check_default_run(S, O, A, D, p)
    assert sparse_times_kernel_is_zero(S, A, K, p)


def test_defaults_fuzz_small_primes(S, O):
    rng = np.random.default_rng(20261004)
    for trial in range(30):
        p = int(rng.choice([3, 5, 7, 11, 127, 251]))
        n, m = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        density = float(rng.choice([0.03, 0.08, 0.2, 0.5]))
        D = (rng.random((n, m)) < density) * rng.integers(1, p, size=(n, m))
        if n > 3 and rng.random() < 0.5:
            D[n - 1] = (D[0] + 2 * D[1]) % p
            D[n - 2] = D[2]
        if m > 4 and rng.random() < 0.3:
            D[:, m - 1] = 0
        A = S.CSR(D.T.copy(), prime=p)
        check_default_run(S, O, A, D, p)


def test_defaults_config2_random_10k(S, O):
    """BASELINE config 2 under the default options: rank against the oracle's (leftmost AND default), kernel by its invariants."""
    from conftest import LM

    A = S.synth_csr(0, 10000, 10000, density=1e-3, prime=42013, seed=0x5A5A0002)
    fact, K, olu = check_default_run(S, O, A, None, 42013, oracle=False)
    assert fact.r == S.echelonize(A, **LM).r  # (== the oracle's: test_config2_random_10k)
    import scipy.sparse as sp

    nz = S.nnz(A)
    As = sp.csr_matrix((A.x[:nz].astype(np.int64), A.j[:nz].astype(np.int64), A.p.astype(np.int64)), shape=A.shape)
    nk = S.nnz(K)
    Ks = sp.csr_matrix((K.x[:nk].astype(np.int64), K.j[:nk].astype(np.int64), K.p.astype(np.int64)), shape=K.shape)
    assert ((As @ Ks.T).tocoo().data % 42013 == 0).all()
    # the same subspace as the leftmost-pivot kernel: stacking the two bases does not raise the rank (both have m - r rows in
    # normal form, so each is a basis of its span; equal spans <=> the stacked rank is m - r).  Checked by reducing the rows of
    # one modulo the other on the device: K_left * (the rows of K) == 0 is implied by A * k^T == 0 and dim = m - rank(A).
    assert K.n == A.m - fact.r


def test_defaults_config5_macaulay_scaled_down(S, O):
    from conftest import LM

    A = S.synth_csr(2, 20000, 8000, row_nnz=40, prime=127, seed=0x5A5A0005)
    fact, K, olu = check_default_run(S, O, A, None, 127)
    assert fact.r == S.echelonize(A, **LM).r
    # A * k^T == 0 on a sample of kernel vectors (exact integers)
    rows = K.rows()
    step = max(1, len(rows) // 10)
    Arows = A.rows()
    for k in rows[::step]:
        kd = dict(k)
        for r in Arows:
            assert sum(int(v) * int(kd.get(c, 0)) for c, v in r) % 127 == 0


GREEDY_CASES = [
    ("fixed_nnz", 1, 3000, 3000, dict(row_nnz=6), 65521),
    ("three_per_row", 1, 8000, 8000, dict(row_nnz=3), 65521),
    ("macaulay_like", 2, 4000, 1600, dict(row_nnz=40), 127),
    ("bernoulli", 0, 900, 1300, dict(density=0.01), 2147483647),
    ("wide_big_prime", 1, 2500, 4000, dict(row_nnz=8), 0xFFFFFFFB),
]


GREEDY_LIMITS = [dict(), dict(SPASM_AMD_GREEDY_REACH_MAX="1024", SPASM_AMD_GREEDY_OCC_MAX="2147483647"),
                 dict(SPASM_AMD_GREEDY_REACH_MAX="16", SPASM_AMD_GREEDY_OCC_MAX="3")]


@pytest.mark.parametrize("limits", GREEDY_LIMITS, ids=["default_limits", "no_limits", "reach16_occ3"])
@pytest.mark.parametrize("name,kind,n,m,kw,prime", GREEDY_CASES, ids=[c[0] for c in GREEDY_CASES])
def test_round0_pivots_match_the_independent_restatement(S, O, monkeypatch, name, kind, n, m, kw, prime, limits):
    """Un-circles the parity of the "FL on columns" search: the engine's round 0 and the oracle's are each compared with the
    plain-Python rule written from DESIGN.md section 2 -- same (column, row) pairs in the same numbering."""
    for k, v in limits.items():                                # (the engine, the oracle and the Python rule read the same variables)
        monkeypatch.setenv(k, v)
    A = S.synth_csr(kind, n, m, prime=prime, seed=0xF1C0, **kw)
    want, nopen, ngreedy = fl_columns_ref.structural_pivots3(A.rows(), m)
    if "SPASM_AMD_GREEDY_OCC_MAX" in limits:
        assert ngreedy > 0 or name == "macaulay_like"          # the case must exercise the search
    fact = S.echelonize(A, enable_greedy_pivot_search=True, enable_dense=False)
    r0 = S.last_rounds()[0]
    assert (r0["npiv"], r0["npiv_open"], r0["npiv_greedy"]) == (len(want), nopen, ngreedy)
    assert pivots_of_first_round(fact, len(want)) == want
    olu = O.echelonize(A, enable_greedy_pivot_search=True)
    assert pivots_of_first_round(olu, len(want)) == want
    assert S.factorization_verify(A, fact, 5)


@pytest.mark.parametrize("name,kind,n,m,kw,prime", GREEDY_CASES, ids=[c[0] for c in GREEDY_CASES])
def test_round0_pivots_without_the_cycle_free_search(S, O, monkeypatch, name, kind, n, m, kw, prime):
    """The first two searches alone (SPASM_AMD_NO_CYCLE_FREE_SEARCH=1, read by engine and oracle): the numbering of round 2's tests --
    open-column pivots first, then the leftmost ones by column -- is what a round keeps when the third search adds nothing."""
    A = S.synth_csr(kind, n, m, prime=prime, seed=0xF1C0, **kw)
    want, nopen = fl_columns_ref.structural_pivots(A.rows(), m, on_columns=True)
    monkeypatch.setenv("SPASM_AMD_NO_CYCLE_FREE_SEARCH", "1")
    fact = S.echelonize(A, enable_greedy_pivot_search=True, enable_dense=False)
    r0 = S.last_rounds()[0]
    olu = O.echelonize(A, enable_greedy_pivot_search=True)
    monkeypatch.delenv("SPASM_AMD_NO_CYCLE_FREE_SEARCH")
    assert (r0["npiv"], r0["npiv_open"], r0["npiv_greedy"]) == (len(want), nopen, 0)
    assert pivots_of_first_round(fact, len(want)) == want
    assert pivots_of_first_round(olu, len(want)) == want


def test_option_fields_without_an_equivalent_say_so(S, O):
    """dense_block_size, low_rank_ratio, low_rank_start_weight and complete (reference src/SpaSM.jl:332, :339, :340, :342) tune
    libspasm's dense strategies; the engine has no equivalent, so it says "ignored" once per call when they leave their defaults,
    and the result is the one the defaults give."""
    import ctypes as C

    A = S.synth_csr(1, 400, 300, row_nnz=5, prime=42013, seed=0x09)
    lines = []
    cb = S.api._LOGFUNC(lambda s: lines.append(s.decode()) or 0)
    slot = C.c_void_p.in_dll(S._abi.lib(), "logcallback")
    prev = slot.value
    slot.value = C.cast(cb, C.c_void_p).value
    try:
        base = S.echelonize(A, verbose=True)
        assert not any("ignored" in x for x in lines), lines
        del lines[:]
        fact = S.echelonize(A, verbose=True, dense_block_size=64, low_rank_ratio=0.9, complete=True)
    finally:
        slot.value = prev
    said = [x for x in lines if "ignored" in x]
    assert len(said) == 3 and "dense_block_size = 64" in said[0] and "low_rank_ratio = 0.9" in said[1] and "complete = 1" in said[2], lines
    assert fact.r == base.r == O.echelonize(A).r
    assert np.asarray(fact.qinv).tolist() == np.asarray(base.qinv).tolist()
