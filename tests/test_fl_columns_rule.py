"""The "FL on columns" pivot search against an INDEPENDENT restatement (tests/fl_columns_ref.py, written from DESIGN.md section 2):
here the CPU oracle (no GPU needed); tests/test_gpu_default_options.py does the same for the engine.  Before this the oracle's
search was only ever compared with the engine's, and the engine's with the oracle's."""
import numpy as np
import pytest

import fl_columns_ref

CASES = [
    ("fixed_nnz", 1, 600, 600, dict(row_nnz=6), 65521),
    ("three_per_row", 1, 1500, 1500, dict(row_nnz=3), 65521),
    ("macaulay_like", 2, 800, 320, dict(row_nnz=40), 127),
    ("bernoulli", 0, 300, 450, dict(density=0.02), 2147483647),
    ("wide", 1, 500, 800, dict(row_nnz=8), 0xFFFFFFFB),
]


def pivots_of_first_round(lu, npiv0):
    """(column, row) of the first npiv0 rows of U: qinv gives the columns, p (pivotal rows first) the rows of the input."""
    q = np.asarray(lu.qinv)
    col_of = {int(q[j]): j for j in range(len(q)) if q[j] >= 0}
    p = np.asarray(lu.p)
    return [(col_of[k], int(p[k])) for k in range(npiv0)]


@pytest.mark.parametrize("name,kind,n,m,kw,prime", CASES, ids=[c[0] for c in CASES])
def test_oracle_takes_the_pivots_of_the_written_rule(S, O, monkeypatch, name, kind, n, m, kw, prime):
    A = S.synth_csr(kind, n, m, prime=prime, seed=0xF1C2, **kw)
    want, nopen = fl_columns_ref.structural_pivots(A.rows(), m, on_columns=True)
    left, _ = fl_columns_ref.structural_pivots(A.rows(), m, on_columns=False)
    assert nopen > 0 or name == "macaulay_like", "the case must exercise the search"  # (nearly all columns of that one are closed)
    monkeypatch.setenv("SPASM_AMD_NO_CYCLE_FREE_SEARCH", "1")                         # the first two searches alone
    olu = O.echelonize(A, enable_greedy_pivot_search=True, max_round=1)
    monkeypatch.delenv("SPASM_AMD_NO_CYCLE_FREE_SEARCH")
    assert pivots_of_first_round(olu, len(want)) == want
    olu2 = O.echelonize(A, enable_greedy_pivot_search=False, max_round=1)
    assert pivots_of_first_round(olu2, len(left)) == left
    # every open-column pivot sits on a column that no leftmost pivot row holds, and is not the leftmost entry of its row somewhere
    rows = A.rows()
    closed = {c for _, i in left for c, _ in rows[i]}
    assert all(c not in closed for c, _ in want[:nopen])


# the two limits of the third search: the defaults (free pivots only), and wide open (libspasm's search has none)
GREEDY_LIMITS = [dict(), dict(SPASM_AMD_GREEDY_REACH_MAX="1024", SPASM_AMD_GREEDY_OCC_MAX="2147483647"),
                 dict(SPASM_AMD_GREEDY_REACH_MAX="16", SPASM_AMD_GREEDY_OCC_MAX="3")]
GREEDY_LIMIT_IDS = ["default_limits", "no_limits", "reach16_occ3"]


@pytest.mark.parametrize("limits", GREEDY_LIMITS, ids=GREEDY_LIMIT_IDS)
@pytest.mark.parametrize("name,kind,n,m,kw,prime", CASES, ids=[c[0] for c in CASES])
def test_oracle_takes_the_pivots_of_the_written_rule_with_the_cycle_free_search(S, O, monkeypatch, name, kind, n, m, kw, prime, limits):
    """All three searches (leftmost, on columns, greedy cycle-free: tests/fl_columns_ref.py structural_pivots3) against the oracle's
    round 0, pair for pair in U's numbering; and what the third search promises: the pivots are permutable to a triangle in the
    order given (a pivot row only holds pivot columns of LATER pivots), every new pivot sits on a non-zero of its row."""
    for k, v in limits.items():
        monkeypatch.setenv(k, v)
    A = S.synth_csr(kind, n, m, prime=prime, seed=0xF1C2, **kw)
    rows = A.rows()
    want, nopen, ngreedy = fl_columns_ref.structural_pivots3(rows, m)
    if "SPASM_AMD_GREEDY_OCC_MAX" in limits:
        assert ngreedy > 0 or name == "macaulay_like", "the case must exercise the search"
    olu = O.echelonize(A, enable_greedy_pivot_search=True, max_round=1)
    assert pivots_of_first_round(olu, len(want)) == want
    idx = {c: k for k, (c, _) in enumerate(want)}
    assert len({i for _, i in want}) == len(want) == len(idx)
    for k, (c, i) in enumerate(want):
        cs = [c2 for c2, _ in rows[i]]
        assert c in cs
        assert all(idx[c2] > k for c2 in cs if c2 != c and c2 in idx)
