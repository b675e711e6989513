"""Macaulay-shaped matrices whose RANK IS KNOWN BY CONSTRUCTION (config 5's rank at full size was a number
nobody could check -- the oracle is out of reach there and spasm_amd_rank drops U).

    E   n0 x m   every row has a 1 on a column of its own (its "leading" column) and 10 .. 30 more entries to the right of it, within
                 a band: a permuted echelon form, rank n0 whatever the other entries are;
    A = P * [ L * E ; D * E ]   L = I + N, N strictly lower triangular with a few entries per row (invertible: rank(L E) = n0);
                                D any sparse matrix (rows that are combinations of rows of E: they add nothing to the rank);
                                P a row permutation.

rank(A) = n0, exactly, over any field -- no probability in it.  `keep` = the share of the rows of L E that only mix in rows whose
leading column lies to the RIGHT of their own (their leftmost entry stays a column of their own: the pivot search of round 0 finds
them, as it finds the shifts of a Macaulay matrix); the other rows mix in anything, so their pivots are only found by elimination --
the dense tail of BASELINE config 5.

    python tools/planted_rank.py [scale=25] [--rank-only] [--keep=0.85] [--n=rows --m=columns] [--opt=field=value]
                                                  # 5M x 2M over scale (or n x m), rank 0.99 * columns, through the C ABI
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def planted(n, m, n0, prime=127, seed=0x5A5A0005, keep=0.85, band=4096):
    """-> (scipy CSR n x m with entries in [0, prime), rank n0 by construction)."""
    assert n0 <= m and n0 <= n
    rng = np.random.default_rng(seed)
    # E: leading columns = n0 distinct columns in ascending order; 10 .. 30 entries to the right of each, inside a band
    lead = np.sort(rng.choice(m, size=n0, replace=False)).astype(np.int64)
    k = rng.integers(10, 31, size=n0)
    rows = np.repeat(np.arange(n0, dtype=np.int64), k)
    cols = lead[rows] + 1 + rng.integers(0, band, size=rows.size)
    ok = cols < m
    rows, cols = rows[ok], cols[ok]
    vals = rng.integers(1, prime, size=rows.size)
    E = sp.csr_matrix((np.concatenate([np.ones(n0, dtype=np.int64), vals]), (np.concatenate([np.arange(n0), rows]), np.concatenate([lead, cols]))), shape=(n0, m))
    E.sum_duplicates()
    E.data %= prime
    # L = I + N: row i mixes in 2 rows j < i ... in the order of DESCENDING leading column, so "j < i" = "leading column to the right"
    # for the rows that keep their own leading entry; the others mix in rows from anywhere before them in a random order
    order = np.arange(n0)[::-1].copy()  # position -> row of E; descending leading column
    pos_of = np.empty(n0, dtype=np.int64)
    pos_of[order] = np.arange(n0)
    nmix = 2
    i = np.repeat(np.arange(1, n0, dtype=np.int64), nmix)  # positions 1 .. n0-1 mix in earlier positions
    j = (rng.random(i.size) * i).astype(np.int64)          # uniform in [0, i)
    free = rng.random(n0) >= keep                          # rows that mix in ANY other row (their leading entry is not their own any more)
    jf = rng.integers(0, n0, size=i.size)
    jj = np.where(free[order[i]], jf, j)
    # (L stays invertible: for the free rows take a second triangular order -- ascending row number -- so that N is nilpotent in
    # neither order alone but L = (I + N1)(I + N2) is a product of invertible matrices; simpler: apply the two mixings one after the other)
    tri = ~free[order[i]]
    N1 = sp.csr_matrix((rng.integers(1, prime, size=int(tri.sum())), (order[i[tri]], order[j[tri]])), shape=(n0, n0))
    A1 = E + N1 @ E
    A1.data %= prime
    fi = np.flatnonzero(free)
    fi = fi[fi > 0]
    fj = (rng.random(fi.size) * fi).astype(np.int64)       # a row with a smaller row NUMBER: strictly lower triangular again
    N2 = sp.csr_matrix((rng.integers(1, prime, size=fi.size), (fi, fj)), shape=(n0, n0))
    A1 = A1 + N2 @ A1
    A1.data %= prime
    # D: every further row a combination of 2 rows of E
    nd = n - n0
    di = np.repeat(np.arange(nd, dtype=np.int64), 2)
    D = sp.csr_matrix((rng.integers(1, prime, size=di.size), (di, rng.integers(0, n0, size=di.size))), shape=(nd, n0))
    A2 = D @ E
    A2.data %= prime
    A = sp.vstack([A1, A2], format="csr")
    A.eliminate_zeros()
    perm = rng.permutation(n)
    A = A[perm]
    A.sort_indices()
    return A


def to_engine(S, A, prime):
    vals = A.data.astype(np.int64)
    vals = np.where(vals > prime // 2, vals - prime, vals)
    keep = vals != 0
    if not keep.all():
        A = A.copy()
        A.data = vals
        A.eliminate_zeros()
        vals = A.data
    return S.CSR.from_arrays(A.shape[0], A.shape[1], A.indptr.astype(np.int64), A.indices.astype(np.int32), vals.astype(np.int32), prime)


if __name__ == "__main__":
    import spasm_jl_amd as S

    scale = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 25
    n, m = 5_000_000 // scale, 2_000_000 // scale
    n0 = int(0.99 * m)
    keep = 0.85
    for a in sys.argv:
        if a.startswith("--keep="):
            keep = float(a[7:])
        if a.startswith("--n="):
            n = int(a[4:])
        if a.startswith("--m="):
            m, n0 = int(a[4:]), int(0.99 * int(a[4:]))
    opts = {}
    for a in sys.argv:
        if a.startswith("--opt="):  # --opt=sparsity_threshold=0.0 : a field of echelonize_opts
            k, v = a[6:].split("=")
            opts[k] = float(v) if "." in v else int(v)
    t0 = time.time()
    A = planted(n, m, n0, keep=keep)
    print(f"planted {n} x {m}, rank {n0} by construction, nnz {A.nnz} in {time.time() - t0:.1f}s", flush=True)
    M = to_engine(S, A, 127)
    del A
    t0 = time.time()
    if "--rank-only" in sys.argv:
        r = S.rank(M, rank_only=True, verbose=("-v" in sys.argv), **opts)
        print(f"spasm_amd_rank: {r} in {time.time() - t0:.2f}s -> {'OK' if r == n0 else 'MISMATCH'} (want {n0})", flush=True)
    else:
        f = S.echelonize(M, verbose=("-v" in sys.argv), **opts)
        r = f.r
        print(f"spasm_echelonize: rank {r} in {time.time() - t0:.2f}s -> {'OK' if r == n0 else 'MISMATCH'} (want {n0})", flush=True)
    sys.exit(0 if r == n0 else 1)
