"""The multi-GPU exchange protocol (spasm.jl_amd/sharded.py) on CPU: two gloo ranks, a numpy engine standing
in for the device, result compared with the oracle's unsharded Schur round."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from conftest import LM  # leftmost-entry pivots only: what these tests compare does not depend on how the rounds went then

INT64_MAX = np.iinfo(np.int64).max


class NumpyShardEngine:
    """Same contract as GpuShardEngine, plain numpy/python (test double; the Schur rows come from a python elimination)."""

    def __init__(self, rows, m, lo, p):
        self.rows, self.m, self.lo, self.p = rows, m, lo, p  # rows: this shard's rows, lists of (col, val)

    def elect(self):
        keys = np.full(self.m, INT64_MAX, dtype=np.int64)
        for i, r in enumerate(self.rows):
            if r:
                lead = min(c for c, _ in r)
                keys[lead] = min(keys[lead], (len(r) << 32) | (self.lo + i))
        return torch.from_numpy(keys)

    def set_keys(self, keys):
        keys = keys.numpy()
        self.pivcols = [j for j in range(self.m) if keys[j] != INT64_MAX]
        self.pivrow = [int(keys[j] & 0xFFFFFFFF) for j in self.pivcols]
        self.owned = [(idx, g - self.lo) for idx, g in enumerate(self.pivrow) if self.lo <= g < self.lo + len(self.rows)]
        return len(self.pivcols), len(self.owned), sum(len(self.rows[i]) for _, i in self.owned)

    def export(self):
        hdr = np.array([[idx, len(self.rows[i])] for idx, i in self.owned], dtype=np.int32).reshape(-1, 2)
        ent = np.array([e for _, i in self.owned for e in self.rows[i]], dtype=np.int32).reshape(-1, 2)
        return torch.from_numpy(hdr), torch.from_numpy(ent)

    def import_(self, hdr_all, ent_all):
        hdr_all, ent_all = hdr_all.numpy(), ent_all.numpy()
        self.U = {}
        k = 0
        for idx, ln in hdr_all:
            row = [(int(c), int(v)) for c, v in ent_all[k:k + ln]]
            k += ln
            pc = self.pivcols[idx]
            inv = pow(dict(row)[pc] % self.p, -1, self.p)
            self.U[pc] = [(c, v * inv % self.p) for c, v in row]

    def schur_rows(self):
        p = self.p
        mine = {i for _, i in self.owned}
        out = []
        for i, r in enumerate(self.rows):
            if i in mine or not r:
                continue
            x = {c: v % p for c, v in r}
            while True:
                cand = [c for c in x if c in self.U and x[c]]
                if not cand:
                    break
                c = min(cand)
                f = x[c]
                for cc, vv in self.U[c]:
                    x[cc] = (x.get(cc, 0) - f * vv) % p
            out.append((self.lo + i, sorted((c, v - p if 2 * v > p else v) for c, v in x.items() if v and c not in self.U)))
        return out


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, m, k, p, seed, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import spasm_jl_amd as S
    from spasm_jl_amd import sharded

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        A = S.synth_csr(1, n, m, row_nnz=k, prime=p, seed=seed)
        lo, hi = rank * n // world, (rank + 1) * n // world
        eng = NumpyShardEngine(A.rows()[lo:hi], m, lo, p)
        npiv, info = sharded.exchange_pivot_rows(eng)
        q.put((rank, npiv, info, eng.schur_rows()))
    except Exception as exc:  # surface the failure to the parent instead of letting it time out
        q.put((rank, -1, {"error": repr(exc)}, []))
        raise
    finally:
        dist.destroy_process_group()


def test_exchange_protocol_two_ranks_gloo(S, O):
    n, m, k, p, seed = 400, 450, 6, 65521, 99
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, m, k, p, seed, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=120) for _ in range(world)]
    assert all(r[1] >= 0 for r in results), results
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    A = S.synth_csr(1, n, m, row_nnz=k, prime=p, seed=seed)
    So, info = O.schur_round(A)
    results.sort()
    assert all(r[1] == info["npiv"] for r in results)
    assert sum(r[2]["owned_rows"] for r in results) == info["npiv"]
    assert all(r[2]["gathered_rows"] == info["npiv"] for r in results)
    got = [row for r in results for row in r[3]]
    got.sort()
    assert [row for _, row in got] == So.rows()


class NumpyRoundEngine(NumpyShardEngine):
    """The engine contract of sharded.echelonize_sharded on the CPU: rows lo, lo + stride, ... of the matrix stay with this
    rank over all rounds (python elimination for the Schur rows)."""

    def __init__(self, Av, lo, hi, stride=1):
        rows = Av.rows()
        self.ids = list(range(lo, hi, stride))
        NumpyShardEngine.__init__(self, [rows[g] for g in self.ids], Av.m, lo, int(Av.prime))
        self.stride = stride

    def elect(self):
        keys = np.full(self.m, INT64_MAX, dtype=np.int64)
        for i, r in enumerate(self.rows):
            if r:
                lead = min(c for c, _ in r)
                keys[lead] = min(keys[lead], (len(r) << 32) | self.ids[i])
        return torch.from_numpy(keys)

    def set_keys(self, keys):
        keys = keys.numpy()
        self.pivcols = [j for j in range(self.m) if keys[j] != INT64_MAX]
        self.pivrow = [int(keys[j] & 0xFFFFFFFF) for j in self.pivcols]
        local = {g: i for i, g in enumerate(self.ids)}
        self.owned = [(idx, local[g]) for idx, g in enumerate(self.pivrow) if g in local]
        return len(self.pivcols), len(self.owned), sum(len(self.rows[i]) for _, i in self.owned)

    def counts(self):
        return sum(1 for r in self.rows if r), sum(len(r) for r in self.rows)

    def round_U(self):
        """the imported round's rows of U in pivot order: (lengths, columns, values (balanced), pivot columns, original rows)"""
        p = self.p
        lens, cols, vals = [], [], []
        for pc in self.pivcols:
            row = self.U[pc]
            lens.append(len(row))
            cols += [c for c, _ in row]
            vals += [v - p if v > p // 2 else v for _, v in row]
        return (np.array(lens, dtype=np.int64), np.array(cols, dtype=np.int64), np.array(vals, dtype=np.int64),
                np.array(self.pivcols, dtype=np.int64), np.array(self.pivrow, dtype=np.int64))

    def advance(self):
        """the round's Schur rows replace the shard's rows (pivot rows and eliminated rows become empty)"""
        self.lo = 0
        out = dict(self.schur_rows())  # local index -> Schur row
        self.rows = [out.get(i, []) for i in range(len(self.rows))]
        return self.counts()

    def fetch_rows(self):
        keep = [i for i, r in enumerate(self.rows) if r]
        gid = np.array([self.ids[i] for i in keep], dtype=np.int64)
        lens = np.array([len(self.rows[i]) for i in keep], dtype=np.int64)
        ent = np.array([e for i in keep for e in self.rows[i]], dtype=np.int64).reshape(-1, 2)
        return gid, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), ent[:, 0].astype(np.int32), ent[:, 1].astype(np.int32)

    def close(self):
        pass


def _echelonize_worker(rank, world, port, n, m, k, p, seed, finish_nnz, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import oracle_ffi as O
    import spasm_jl_amd as S
    from spasm_jl_amd import sharded

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        A = S.synth_csr(1, n, m, row_nnz=k, prime=p, seed=seed)
        fact, info = sharded.echelonize_sharded(A, engine_cls=NumpyRoundEngine, finish=lambda M: O.echelonize(M, enable_greedy_pivot_search=False), finish_nnz=finish_nnz)
        K = O.kernel(fact)
        q.put((rank, fact.r, np.asarray(fact.qinv).tolist(), K.rows(), [(r["finish"], r["npiv"]) for r in info["rounds"]]))
    except Exception as exc:
        import traceback

        q.put((rank, -1, repr(exc) + traceback.format_exc(), [], []))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,finish_nnz", [(2, 600), (3, 0)])
def test_echelonize_sharded_rounds_gloo(S, O, world, finish_nnz):
    """The whole row-sharded echelonization (election all-reduce, pivot-row all-gather, local Schur rows, round after round,
    then the replicated finish) on CPU ranks over gloo: same rank, pivot columns and kernel as the unsharded oracle."""
    n, m, k, p, seed = 260, 300, 4, 65521, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_echelonize_worker, args=(r, world, port, n, m, k, p, seed, finish_nnz, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=300) for _ in range(world)]
    assert all(r[1] >= 0 for r in results), results
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    A = S.synth_csr(1, n, m, row_nnz=k, prime=p, seed=seed)
    olu = O.echelonize(A, **LM)
    oK = O.kernel(olu)
    results.sort()
    for rank, r, qinv, Krows, rounds in results:
        assert r == olu.r
        assert [c >= 0 for c in qinv] == (olu.qinv >= 0).tolist()
        assert Krows == oK.rows()
        assert rounds == results[0][4]                       # every rank went through the same rounds
        assert sum(1 for fin, _ in rounds if not fin) >= 1   # at least one sharded round before the finish
    assert sum(np_ for _, np_ in results[0][4]) == olu.r


def test_collectives_degenerate_to_identity_without_a_group():
    from spasm_jl_amd import sharded

    t = torch.arange(6, dtype=torch.int64)
    assert sharded.all_reduce_min(t) is t
    assert sharded.all_gather_var(t.reshape(3, 2), [3]).shape == (3, 2)
    assert sharded.all_gather_counts([4, 5]).tolist() == [[4, 5]]
