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


# ---------------------------------------------------------------------------------------------------------------------------
# The third search -- "greedy alternating cycle-free search" (reference README.md:23) -- as DESIGN.md section 2 words it:
#   a pass (up to three, until one accepts nothing), for every non-empty row that is no pivot row and has at most 256 entries:
#     reach     the pivots reachable from the row: those on the pivot columns it holds, then those on the pivot columns THEIR rows
#               hold, and so on; a row that reaches more than GREEDY_REACH_MAX pivots sits the pass out;
#     touched   the columns without pivot that the rows of the reach hold;
#     occ[c]    the number of rows that are no pivot rows (at the start of the pass) and hold column c;
#     the row proposes, among its columns that carry no pivot, are not touched and have occ <= GREEDY_OCC_MAX, the one of smallest
#     occ (ties: leftmost);
#     per proposed column the smallest (row length, row number) wins;
#     a winner is accepted unless some OTHER column of (its own pivot-free columns + touched) has a winner with a smaller key.
#   Accepted rows are pivot rows from the next pass on.
#   When the search found anything, all pivots of the round are numbered by descending level (0: the row holds no other pivot column;
#   else 1 + the deepest level among the pivot columns it holds), ascending column inside a level.
# ---------------------------------------------------------------------------------------------------------------------------
GREEDY_PASSES = 3
GR_MAXLEN = 256
GR_BUDGET = 1024
GREEDY_REACH_MAX_DEFAULT = 2
GREEDY_OCC_MAX_DEFAULT = 1


def _limits():
    """The two limits of the search, with the environment variables the engine and the oracle read."""
    import os

    reach = min(GR_BUDGET, max(0, int(os.environ.get("SPASM_AMD_GREEDY_REACH_MAX", GREEDY_REACH_MAX_DEFAULT))))
    occ = max(1, int(os.environ.get("SPASM_AMD_GREEDY_OCC_MAX", GREEDY_OCC_MAX_DEFAULT)))
    return reach, occ


def greedy_extend(rows, m, pivots):
    """pivots: [(column, row)] found so far (any numbering).  Returns (all pivots renumbered -- or `pivots` unchanged when the search
    finds nothing --, number of pivots the search added)."""
    cols = [[c for c, _ in r] for r in rows]
    n = len(rows)
    prow_of_col = {c: i for c, i in pivots}
    is_piv = [False] * n
    for _, i in pivots:
        is_piv[i] = True
    added = 0
    reach_max, occ_max = _limits()
    for _ in range(GREEDY_PASSES):
        proposal, full, winner = {}, {}, {}
        occ = {}
        for i, cs in enumerate(cols):
            if not is_piv[i]:
                for c in cs:
                    occ[c] = occ.get(c, 0) + 1
        for i, cs in enumerate(cols):
            if is_piv[i] or not cs or len(cs) > GR_MAXLEN:
                continue
            reach, todo = set(), [prow_of_col[c] for c in cs if c in prow_of_col]
            reach.update(todo)
            while todo and len(reach) <= reach_max:
                r = todo.pop()
                for c in cols[r]:
                    r2 = prow_of_col.get(c)
                    if r2 is not None and r2 not in reach:
                        reach.add(r2)
                        todo.append(r2)
            if len(reach) > reach_max:
                continue
            touched = {c for r in reach for c in cols[r] if c not in prow_of_col}
            cand = [c for c in cs if c not in prow_of_col and c not in touched and occ[c] <= occ_max]
            if not cand:
                continue
            j = min(cand, key=lambda c: (occ[c], c))
            proposal[i] = j
            full[i] = touched | {c for c in cs if c not in prow_of_col}
            key = (len(cs), i)
            if j not in winner or key < winner[j]:
                winner[j] = key
        accepted = []
        for i, j in proposal.items():
            key = (len(cols[i]), i)
            if winner[j] != key:
                continue
            if any(c != j and c in winner and winner[c] < key for c in full[i]):
                continue
            accepted.append((j, i))
        if not accepted:
            break
        for j, i in accepted:
            prow_of_col[j] = i
            is_piv[i] = True
        added += len(accepted)
    if added == 0:
        return list(pivots), 0
    # levels (the pivot graph is acyclic: that is what the search guarantees)
    level = {}

    def lev(c):
        stack = [c]
        while stack:
            x = stack[-1]
            if x in level:
                stack.pop()
                continue
            deps = [c2 for c2 in cols[prow_of_col[x]] if c2 != x and c2 in prow_of_col]
            missing = [d for d in deps if d not in level]
            if missing:
                assert len(stack) <= len(prow_of_col), "cycle among the pivots"
                stack.extend(missing)
                continue
            level[x] = 1 + max((level[d] for d in deps), default=-1)
            stack.pop()
        return level[c]

    for c in prow_of_col:
        lev(c)
    order = sorted(prow_of_col, key=lambda c: (-level[c], c))
    return [(c, prow_of_col[c]) for c in order], added


def structural_pivots3(rows, m):
    """All three searches of a sparse round: [(column, row)] in U's numbering, open-column pivots, greedy pivots."""
    piv, nopen = structural_pivots(rows, m, on_columns=True)
    out, ngreedy = greedy_extend(rows, m, piv)
    return out, nopen, ngreedy
