"""spasm_factorization_verify (reference src/SpaSM.jl:934), the host-side probabilistic self-check: accepts the oracle's
factorizations, rejects factorizations that are wrong in each of the ways it claims to detect.  CPU only."""
import ctypes as C

import numpy as np
import pytest


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
    olu = O.echelonize(A)
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
    lu = _copy_lu(S, O.echelonize(A), 200)
    assert S.factorization_verify(A, lu, 3)
    assert not S.factorization_verify(B, lu, 3)
