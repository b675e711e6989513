"""An INDEPENDENT restatement, in plain Python, of the structural pivot search of one sparse round -- the leftmost-entry election
(Faugere-Lachartre) followed by "FL on columns" -- written from the prose of DESIGN.md section 2, not from the engine's kernels or
the C oracle.  Test infrastructure only: tests compare the pivots of the engine's round 0 AND of the oracle's with this.

The rule:
  leftmost election   every non-empty row bids for its leftmost column with the key (row length, row number); per column the
                      smallest key wins.  Winners are the leftmost pivots.
  open columns        a column is closed when some pivot row holds it, open otherwise.
  a pass              among the rows that are not pivot rows: occ[c] = number of those rows holding column c; every such row with
                      an open column proposes ONE: its open column of smallest occ (ties: smallest column); per proposed column
                      the smallest (row length, row number) among its proposers wins; a winner is accepted when no OTHER column of
                      its row received a proposal from anybody.  Accepted rows become pivot rows on the column they proposed,
                      their columns are closed.  Up to four passes, or until a pass accepts nothing.
  numbering           the pivots of the last pass first, then the passes before it, each by ascending column; then the leftmost
                      pivots by ascending column.
Returns [(column, row)] in that numbering, and the number of open-column pivots."""

OPEN_PASSES = 4


def structural_pivots(rows, m, on_columns=True):
    """rows: list of lists of (column, value); m: number of columns."""
    cols = [[c for c, _ in r] for r in rows]
    n = len(rows)
    # ---- leftmost election
    best = {}
    for i, cs in enumerate(cols):
        if not cs:
            continue
        j = min(cs)
        key = (len(cs), i)
        if j not in best or key < best[j]:
            best[j] = key
    leftmost = sorted((j, k[1]) for j, k in best.items())
    if not on_columns or not leftmost:
        return leftmost, 0
    is_piv = [False] * n
    closed = [False] * m
    for j, i in leftmost:
        is_piv[i] = True
    for j, i in leftmost:
        for c in cols[i]:
            closed[c] = True
    passes = []
    for _ in range(OPEN_PASSES):
        occ = {}
        for i, cs in enumerate(cols):
            if not is_piv[i]:
                for c in cs:
                    occ[c] = occ.get(c, 0) + 1
        proposal = {}
        winner = {}
        for i, cs in enumerate(cols):
            if is_piv[i] or not cs:
                continue
            open_cols = [c for c in cs if not closed[c]]
            if not open_cols:
                continue
            c = min(open_cols, key=lambda x: (occ[x], x))
            proposal[i] = c
            key = (len(cs), i)
            if c not in winner or key < winner[c]:
                winner[c] = key
        accepted = []
        for i, c in proposal.items():
            if winner[c][1] != i:
                continue
            if any(c2 != c and c2 in winner for c2 in cols[i]):
                continue
            accepted.append((c, i))
        if not accepted:
            break
        accepted.sort()
        passes.append(accepted)
        for c, i in accepted:
            is_piv[i] = True
        for c, i in accepted:
            for c2 in cols[i]:
                closed[c2] = True
    out = []
    for acc in reversed(passes):
        out.extend(acc)
    nopen = len(out)
    out.extend(leftmost)
    return out, nopen
