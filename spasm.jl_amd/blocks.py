"""Block-diagonal driver: mirror of SpaSM.jl's src/blocks.jl over the C ABI.

`Block(A)` splits a CSR into the connected components of its row/column graph (reference
src/blocks.jl:35-105); `echelonize`, `rank`, `kernel` then run block by block (:107-139) and
`to_csr` stitches a block matrix back together (:142-170).  Blocks are independent units: with one
process per GPU, `owner=(rank, world)` makes a process work only on its share (zero communication).
"""
import numpy as np

from . import api


class Block:
    """blocks[b] with the maps row2block / col2block = (block, position, 0-based) and their inverses (src/blocks.jl:1-7)."""

    def __init__(self, blocks, row2block, col2block, block2row, block2col):
        self.blocks = blocks
        self.row2block = row2block
        self.col2block = col2block
        self.block2row = block2row
        self.block2col = block2col

    def __len__(self):
        return len(self.blocks)

    @property
    def shape(self):  # Base.size(block), src/blocks.jl:11
        return (len(self.row2block), len(self.col2block))

    @classmethod
    def from_csr(cls, A):
        """Block(A::CSR): union-find over rows and columns joined by the non-zeros (src/blocks.jl:35-105).
        Blocks are numbered by their smallest member in (rows, then columns) order."""
        n, m = A.shape
        nz = api.nnz(A)
        p, j, x = A.p, A.j[:nz], A.x[:nz]
        import scipy.sparse as sp
        from scipy.sparse.csgraph import connected_components

        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(p))
        G = sp.coo_matrix((np.ones(nz, dtype=np.int8), (rows, j.astype(np.int64) + n)), shape=(n + m, n + m))
        ncomp, lab = connected_components(G, directed=False)
        first = np.full(ncomp, n + m, dtype=np.int64)
        np.minimum.at(first, lab, np.arange(n + m))
        renum = np.empty(ncomp, dtype=np.int64)
        renum[np.argsort(first, kind="stable")] = np.arange(ncomp)
        lab = renum[lab]
        block2row = [[] for _ in range(ncomp)]
        block2col = [[] for _ in range(ncomp)]
        row2block, col2block = [], []
        for i in range(n):
            b = int(lab[i])
            row2block.append((b, len(block2row[b])))
            block2row[b].append(i)
        for c in range(m):
            b = int(lab[n + c])
            col2block.append((b, len(block2col[b])))
            block2col[b].append(c)
        colpos = np.array([q for _, q in col2block], dtype=np.int32)
        blocks = []
        for b in range(ncomp):
            rws = block2row[b]
            lens = np.array([int(p[i + 1] - p[i]) for i in rws], dtype=np.int64)
            sp_ = np.concatenate([[0], np.cumsum(lens)]) if rws else np.zeros(1, dtype=np.int64)
            idx = np.concatenate([np.arange(p[i], p[i + 1]) for i in rws]) if rws and sp_[-1] else np.zeros(0, dtype=np.int64)
            blocks.append(api.CSR.from_arrays(len(rws), len(block2col[b]), sp_, colpos[j[idx]] if len(idx) else [], x[idx] if len(idx) else [], prime=A.prime))
        return cls(blocks, row2block, col2block, block2row, block2col)

    def to_csr(self):
        """CSR(block::Block{CSR}) (src/blocks.jl:142-170)."""
        n, m = self.shape
        prime = self.blocks[0].prime if self.blocks else api.prime0
        rows = []
        cache = [b.rows() for b in self.blocks]
        for i in range(n):
            b, sub = self.row2block[i]
            rows.append([(self.block2col[b][c], v) for c, v in cache[b][sub]])
        return api.CSR.from_rows(rows, m, prime=prime)


def echelonize(block, owner=None, **kwargs):
    """echelonize(block::Block{CSR}) (src/blocks.jl:107-115).  owner=(rank, world): only blocks b % world == rank."""
    lus = []
    for b, A in enumerate(block.blocks):
        mine = owner is None or b % owner[1] == owner[0]
        lus.append(api.echelonize(A, **kwargs) if mine else None)
    return Block(lus, block.row2block, block.col2block, block.block2row, block.block2col)


def rank(block, **kwargs):
    """rank(block) = sum of the ranks (src/blocks.jl:117)."""
    return sum(api.rank(X, **kwargs) for X in block.blocks if X is not None)


def kernel(block, **kwargs):
    """kernel(block::Block{LU}) (src/blocks.jl:119-137): per-block kernels, rows numbered block after block."""
    if block.blocks and isinstance(block.blocks[0], api.CSR):
        block = echelonize(block, **kwargs)
    ks = [api.kernel(X) for X in block.blocks]
    block2row, row2block, r = [], [], 0
    for b, k in enumerate(ks):
        block2row.append(list(range(r, r + k.n)))
        row2block += [(b, i) for i in range(k.n)]
        r += k.n
    return Block(ks, row2block, block.col2block, block2row, block.block2col)
