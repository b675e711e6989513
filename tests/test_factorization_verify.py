"""spasm_factorization_verify (reference src/SpaSM.jl:934), the host-side probabilistic self-check: accepts the oracle's
factorizations, rejects factorizations that are wrong in each of the ways it claims to detect.  CPU only."""
import ctypes as C

import numpy as np
import pytest
from conftest import LM  # leftmost-entry pivots only: what these tests compare does not depend on how the rounds went then


def _copy_lu(S, olu, n, drop_row=None, poke=None):
    """A product-side LU with the oracle's U / qinv (optionally with one row dropped or one entry changed)."""
    U = olu.U
    rows = [list(r) for r in U.rows()]
    qinv = np.array(olu.qinv, dtype=np.int32).copy()
    if poke is not None:
        k, t, v = poke
        c, _ = rows[k][t]
        rows[k][t] = (c, v)
    if drop_row is not None:
        col = int(np.flatnonzero(qinv == drop_row)[0])
        qinv[col] = -1
        qinv[qinv > drop_row] -= 1
        rows.pop(drop_row)
    Uc = S.CSR.from_rows(rows, U.m, U.prime)
    perm = np.full(max(n, U.m, 1), -1, dtype=np.int32)
    return S.LU.from_parts(Uc, qinv, perm)


@pytest.mark.parametrize("n,m,k,p,seed", [(300, 340, 5, 65521, 1), (200, 150, 4, 127, 2), (250, 250, 6, 0xfffffffb, 3), (120, 400, 3, 7, 4)])
def test_verify_accepts_correct_and_rejects_wrong(S, O, n, m, k, p, seed):
    A = S.synth_csr(1, n, m, row_nnz=k, prime=p, seed=seed)
    olu = O.echelonize(A, **LM)
    good = _copy_lu(S, olu, n)
    assert good.r == olu.r
    for sd in (0, 1, 0xDEADBEEF):
        assert S.factorization_verify(A, good, sd)
    # a U that misses one of its rows cannot span the rows of A any more
    assert not S.factorization_verify(A, _copy_lu(S, olu, n, drop_row=olu.r // 2), 5)
    # a changed non-pivot entry: some row of A no longer reduces to zero (unless U spans everything: r == m, where any
    # echelon-shaped U is a correct basis)
    rows = olu.U.rows()
    k0 = next(i for i, r in enumerate(rows) if len(r) > 1)
    pc = int(np.flatnonzero(np.asarray(olu.qinv) == k0)[0])
    t = next(i for i, (c, _) in enumerate(rows[k0]) if c != pc)
    bad_val = rows[k0][t][1] + 1 if rows[k0][t][1] + 1 <= p // 2 else rows[k0][t][1] - 1
    if bad_val != 0 and olu.r < m:  # (bad_val == 0: from_rows would store an explicit zero; skip that corner)
        assert not S.factorization_verify(A, _copy_lu(S, olu, n, poke=(k0, t, bad_val)), 6)
    # a pivot that is not 1: the echelon shape check
    tp = next(i for i, (c, _) in enumerate(rows[k0]) if c == pc)
    assert not S.factorization_verify(A, _copy_lu(S, olu, n, poke=(k0, tp, 2)), 7)


def test_verify_rejects_mismatched_matrix(S, O):
    A = S.synth_csr(1, 200, 220, row_nnz=5, prime=65521, seed=11)
    B = S.synth_csr(1, 200, 220, row_nnz=5, prime=65521, seed=12)
    lu = _copy_lu(S, O.echelonize(A, **LM), 200)
    assert S.factorization_verify(A, lu, 3)
    assert not S.factorization_verify(B, lu, 3)


@pytest.mark.parametrize("kind,n,m,kw,p,seed", [(1, 600, 640, dict(row_nnz=3), 65521, 1), (1, 500, 400, dict(row_nnz=5), 127, 2),
                                                (1, 700, 900, dict(row_nnz=8), 0xFFFFFFFB, 3), (0, 300, 500, dict(density=0.01), 2147483647, 4)])
def test_oracle_fl_on_columns_pivots(S, O, kind, n, m, kw, p, seed):
    """The oracle's restatement of the "Faugere-Lachartre on columns" search (enable_greedy_pivot_search, reference
    src/SpaSM.jl:326; oracle fl_pivots_ex): pivots that are not leftmost entries, yet U stays (permuted) triangular in the order
    it is stored, spans the rows of A (factorization_verify accepts it for several seeds), has the rank of the leftmost-only
    run, and yields a kernel basis that annihilates A.  A cycle among the pivots must be rejected by the verifier."""
    A = S.synth_csr(kind, n, m, prime=p, seed=seed, **kw)
    g = O.echelonize(A, enable_greedy_pivot_search=True)
    l = O.echelonize(A, enable_greedy_pivot_search=False)
    assert g.r == l.r
    rows = g.U.rows()
    q = np.asarray(g.qinv)
    pc = {int(q[j]): j for j in range(m) if q[j] >= 0}
    assert sorted(pc) == list(range(g.r))
    assert any(min(c for c, _ in row) != pc[a] for a, row in enumerate(rows)), "no pivot off the leftmost entry: the search did nothing"
    for a, row in enumerate(rows):  # stored in topological order: a row only touches pivot columns of later rows
        assert all(q[c] < 0 or q[c] >= a for c, _ in row)
    lu = _copy_lu(S, g, n)
    for sd in (0, 1, 2):
        assert S.factorization_verify(A, lu, sd)
    # K * A^T == 0 with exact integers
    K = O.kernel(g)
    assert K.n == m - g.r
    Arows = A.rows()
    for krow in K.rows():
        kv = dict(krow)
        for arow in Arows:
            assert sum(v * kv.get(c, 0) for c, v in arow) % p == 0
    # two rows that hold each other's pivot column: no order eliminates with them, the verifier must say no
    if g.r >= 2:
        cyc = [list(r) for r in rows]
        a, b = 0, 1
        if not any(c == pc[b] for c, _ in cyc[a]):
            cyc[a].append((pc[b], 1))
        if not any(c == pc[a] for c, _ in cyc[b]):
            cyc[b].append((pc[a], 1))
        Uc = S.CSR.from_rows(cyc, m, p)
        bad = S.LU.from_parts(Uc, np.array(q, dtype=np.int32), np.full(max(n, m, 1), -1, dtype=np.int32))
        assert not S.factorization_verify(A, bad, 1)
