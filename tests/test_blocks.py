"""Block-diagonal driver (reference src/blocks.jl)."""
import numpy as np
import pytest
from conftest import LM  # leftmost-entry pivots only: what these tests compare does not depend on how the rounds went then


def make_block_matrix(S, seed=3, p=42013):
    """Three independent blocks + an isolated empty row and an isolated empty column, rows and columns interleaved."""
    rng = np.random.default_rng(seed)
    shapes = [(6, 5), (4, 7), (9, 3)]
    n = sum(a for a, _ in shapes) + 1
    m = sum(b for _, b in shapes) + 1
    rperm, cperm = rng.permutation(n), rng.permutation(m)
    rows = [[] for _ in range(n)]
    r0 = c0 = 0
    for (a, b) in shapes:
        D = (rng.random((a, b)) < 0.5) * rng.integers(1, p, size=(a, b))
        D[0, 0] = 1
        for i in range(a):
            D[i, i % b] = D[i, i % b] or 7  # keep the block connected
            if i + 1 < a:
                D[i + 1, i % b] = D[i + 1, i % b] or 5
        D[a - 1] = (D[0] * 2) % p  # rank deficiency
        for i in range(a):
            rows[rperm[r0 + i]] = [(int(cperm[c0 + c]), int(D[i, c])) for c in range(b) if D[i, c]]
        r0 += a
        c0 += b
    return S.CSR.from_rows(rows, m, prime=p), len(shapes)


def test_block_split_and_reassembly(S):
    A, nblocks = make_block_matrix(S)
    B = S.Block.from_csr(A)
    assert len(B) == nblocks + 2  # + the empty row and the empty column, each alone in its component
    assert B.shape == A.shape
    assert sorted(i for rows in B.block2row for i in rows) == list(range(A.n))
    assert sorted(c for cols in B.block2col for c in cols) == list(range(A.m))
    for b, blk in enumerate(B.blocks):
        assert blk.shape == (len(B.block2row[b]), len(B.block2col[b]))
    assert B.to_csr().rows() == A.rows()  # CSR(Block(A)) == A, src/blocks.jl:142-170


@pytest.mark.gpu
def test_block_rank_and_kernel_match_the_unsplit_matrix(S, O):
    A, _ = make_block_matrix(S, seed=11)
    B = S.Block.from_csr(A)
    E = S.blocks.echelonize(B, **LM)
    olu = O.echelonize(A, **LM)
    assert S.blocks.rank(E) == olu.r == S.rank(A)
    K = S.blocks.kernel(E).to_csr()
    want = O.kernel(olu).rows()
    assert K.shape == (len(want), A.m)
    assert sorted(K.rows()) == sorted(want)  # same kernel vectors, ordered block after block (src/blocks.jl:119-137)
    # owner=(rank, world): every block is echelonized by exactly one of two processes
    parts = [S.blocks.echelonize(B, owner=(r, 2), **LM) for r in range(2)]
    assert all((parts[0].blocks[b] is None) != (parts[1].blocks[b] is None) for b in range(len(B)))
    assert sum(S.blocks.rank(P) for P in parts) == olu.r
