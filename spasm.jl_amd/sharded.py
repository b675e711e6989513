"""Row-sharded Schur round over a torch.distributed group (one process per GPU).

Non-pivot rows are the independent units: every rank keeps a block of rows resident, the
round's pivot rows are exchanged once, U is then identical on every rank and Schur rows never move.
torch.distributed is plumbing only (RCCL over xGMI for backend "nccl", gloo in tests); the compute is
behind the `engine` object:

    keys            = engine.elect()                     # i64[m]: (len << 32 | global row), INT64_MAX = none
    npiv, n, nnz    = engine.set_keys(keys)              # after all-reduce(MIN)
    hdr, ent        = engine.export()                    # i32[n,2] (pivot index, length), i32[nnz,2] (col, val)
    engine.import_(hdr_all, ent_all)                     # concatenation over ranks

`GpuShardEngine` implements it over the C ABI (spasm_amd_shard_*); tests also drive the protocol with a
numpy engine on CPU-only machines.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _abi


class GpuShardEngine:
    def __init__(self, A, row_lo, row_hi, device=None, stride=1):
        self.lib = _abi.lib()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.m = A.m
        self.A = A  # keeps the host matrix alive
        # rows row_lo, row_lo + stride, ... < row_hi (rank r of G: row_lo = r, stride = G balances the shards)
        self.shard = self.lib.spasm_amd_shard_create_strided(A.data, int(row_lo), int(row_hi), int(stride))
        if not self.shard:
            raise RuntimeError("spasm_amd_shard_create failed: " + _abi.last_error())
        self.plan = None
        self.n_own = self.nnz_own = 0
        self.npiv = 0

    def elect(self):
        keys = torch.empty(self.m, dtype=torch.int64, device=self.device)
        if self.lib.spasm_amd_shard_elect(self.shard, C.c_void_p(keys.data_ptr())) != 0:
            raise RuntimeError("spasm_amd_shard_elect failed: " + _abi.last_error())
        return keys

    def set_keys(self, keys):
        keys = keys.to(self.device).contiguous()
        n = C.c_int32(0)
        nnz = C.c_int64(0)
        npiv = self.lib.spasm_amd_shard_set_keys(self.shard, C.c_void_p(keys.data_ptr()), C.byref(n), C.byref(nnz))
        if npiv < 0:
            raise RuntimeError("spasm_amd_shard_set_keys failed: " + _abi.last_error())
        self.n_own, self.nnz_own = int(n.value), int(nnz.value)
        self.npiv = int(npiv)
        return int(npiv), self.n_own, self.nnz_own

    def set_keys_open(self, keys, group=None):
        """set_keys with "Faugere-Lachartre on columns" between its halves (the single-device round's second pivot search,
        enable_greedy_pivot_search; reference src/SpaSM.jl:326, README.md:23) over the row shards: the steps of
        spasm_amd_shard_open_step, each followed by the reduction of the array it leaves (closed: MAX, colcnt: SUM, best2: MIN,
        newflag: MAX -- four m-word all-reduces per pass, at most four passes).  Returns (npiv, owned rows, owned entries, pivots the
        search added)."""
        lib = self.lib
        keys = keys.to(self.device).contiguous()
        npiv0 = lib.spasm_amd_shard_assign(self.shard, C.c_void_p(keys.data_ptr()))
        if npiv0 < 0:
            raise RuntimeError("spasm_amd_shard_assign failed: " + _abi.last_error())
        added = 0
        if npiv0 > 0:
            def step(k, pass_, tin, tout):
                rc = lib.spasm_amd_shard_open_step(self.shard, k, pass_, C.c_void_p(tin.data_ptr()) if tin is not None else None,
                                                   C.c_void_p(tout.data_ptr()) if tout is not None else None)
                if rc < 0:
                    raise RuntimeError("spasm_amd_shard_open_step failed: " + _abi.last_error())
                return rc

            def reduce(t, op):
                if not dist.is_initialized() or dist.get_world_size(group) == 1:
                    return t
                x = _collective_device(t, group).contiguous()
                dist.all_reduce(x, op=op, group=group)
                return x.to(self.device)

            i32 = lambda: torch.empty(self.m, dtype=torch.int32, device=self.device)  # noqa: E731
            closed, colcnt, newflag = i32(), i32(), i32()
            best2 = torch.empty(self.m, dtype=torch.int64, device=self.device)
            step(0, 0, None, closed)
            closed = reduce(closed, dist.ReduceOp.MAX)
            for pass_ in range(1, 5):
                step(1, pass_, closed, colcnt)
                colcnt = reduce(colcnt, dist.ReduceOp.SUM)
                step(2, pass_, colcnt, best2)
                best2 = reduce(best2, dist.ReduceOp.MIN)
                step(3, pass_, best2, newflag)
                newflag = reduce(newflag, dist.ReduceOp.MAX)
                closed = i32()
                if step(4, pass_, newflag, closed) == 0:
                    break
                closed = reduce(closed, dist.ReduceOp.MAX)
            added = step(5, 0, None, None)
        n = C.c_int32(0)
        nnz = C.c_int64(0)
        npiv = lib.spasm_amd_shard_finish_keys(self.shard, C.byref(n), C.byref(nnz))
        if npiv < 0:
            raise RuntimeError("spasm_amd_shard_finish_keys failed: " + _abi.last_error())
        self.n_own, self.nnz_own = int(n.value), int(nnz.value)
        self.npiv = int(npiv)
        return int(npiv), self.n_own, self.nnz_own, int(added)

    def export(self):
        hdr = torch.empty((max(self.n_own, 1), 2), dtype=torch.int32, device=self.device)
        ent = torch.empty((max(self.nnz_own, 1), 2), dtype=torch.int32, device=self.device)
        if self.lib.spasm_amd_shard_export(self.shard, C.c_void_p(hdr.data_ptr()), C.c_void_p(ent.data_ptr())) != 0:
            raise RuntimeError("spasm_amd_shard_export failed: " + _abi.last_error())
        return hdr[: self.n_own], ent[: self.nnz_own]

    def import_(self, hdr_all, ent_all, prepare=True):
        """prepare=False: stop after U (spasm_amd_shard_import_U) -- the caller decides first whether the round's Schur complement
        goes dense, and calls prepare() when it does not."""
        hdr_all = hdr_all.to(self.device).contiguous()
        ent_all = ent_all.to(self.device).contiguous()
        fn = self.lib.spasm_amd_shard_import if prepare else self.lib.spasm_amd_shard_import_U
        plan = fn(self.shard, int(hdr_all.shape[0]), int(ent_all.shape[0]), C.c_void_p(hdr_all.data_ptr()), C.c_void_p(ent_all.data_ptr()))
        if not plan:
            raise RuntimeError("spasm_amd_shard_import failed: " + _abi.last_error())
        self.plan = plan
        return plan

    def prepare(self):
        if self.lib.spasm_amd_schur_plan_prepare(self.plan) != 0:
            raise RuntimeError("spasm_amd_schur_plan_prepare failed: " + _abi.last_error())

    def close(self):
        if self.plan:
            self.lib.spasm_amd_schur_plan_free(self.plan)
            self.plan = None
        if self.shard:
            self.lib.spasm_amd_shard_free(self.shard)
            self.shard = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _collective_device(t, group):
    """gloo moves CPU tensors; nccl (= RCCL) moves device tensors."""
    backend = dist.get_backend(group) if dist.is_initialized() else "none"
    return t.cpu() if backend == "gloo" else t


def all_reduce_min(t, group=None):
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    x = _collective_device(t, group).contiguous()
    dist.all_reduce(x, op=dist.ReduceOp.MIN, group=group)
    return x.to(t.device)


def all_gather_counts(values, group=None):
    """values: list of python ints of this rank -> i64 tensor [world, len(values)]"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = torch.tensor(values, dtype=torch.int64)
    if world == 1:
        return mine.reshape(1, -1)
    backend = dist.get_backend(group)
    if backend != "gloo":
        mine = mine.cuda()
    outs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(outs, mine, group=group)
    return torch.stack(outs).cpu()


def all_gather_var(t, counts, group=None):
    """Variable-length all-gather of the leading dimension: every rank contributes t[:counts[rank]], all receive
    the concatenation in rank order.  Padded to the largest part (one all-gather, one link-time per peer on xGMI)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return t
    mx = max(int(max(counts)), 1)
    x = _collective_device(t, group)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=x.device)
    pad[: t.shape[0]] = x
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    parts = [outs[r][: int(counts[r])] for r in range(world)]
    return torch.cat(parts, dim=0).to(t.device)


def exchange_pivot_rows(engine, group=None):
    """The round's exchange step.  Returns (npiv, info) after engine.import_ has built U on this rank."""
    keys = engine.elect()
    keys = all_reduce_min(keys, group)                       # election: sparsest row per leftmost column, ties to the lowest row
    npiv, n_own, nnz_own = engine.set_keys(keys)
    counts = all_gather_counts([n_own, nnz_own], group)      # [world, 2]
    hdr, ent = engine.export()
    hdr_all = all_gather_var(hdr, counts[:, 0].tolist(), group)
    ent_all = all_gather_var(ent, counts[:, 1].tolist(), group)
    if int(hdr_all.shape[0]) != npiv:
        raise RuntimeError(f"exchange: {hdr_all.shape[0]} pivot rows gathered, {npiv} elected")
    engine.import_(hdr_all, ent_all)
    return npiv, {"owned_rows": n_own, "owned_nnz": nnz_own, "gathered_rows": int(hdr_all.shape[0]),
                  "gathered_bytes": int(hdr_all.numel() * 4 + ent_all.numel() * 4)}


# ------------------------------------------------------------------------------------------------
# The whole echelonization, row-sharded: rounds of (election, pivot-row exchange, local Schur complement) until what is
# left is small, then every rank finishes the same remainder.
# ------------------------------------------------------------------------------------------------
class GpuRoundEngine(GpuShardEngine):
    """The shard of one rank over all rounds: the exchange protocol of GpuShardEngine plus
        counts()      (non-empty rows, entries) of the shard's current matrix,
        advance()     run the round; its Schur rows become the shard's matrix ON THE DEVICE (spasm_amd_schur_plan_advance),
        fetch_rows()  the current rows as host arrays (ids = rows of the ORIGINAL matrix, ascending) for the final gather."""

    def __init__(self, A, row_lo, row_hi, device=None, stride=1):
        import numpy as np

        GpuShardEngine.__init__(self, A, row_lo, row_hi, device=device, stride=stride)
        self.lo, self.stride = int(row_lo), int(stride)
        lens = np.diff(np.asarray(A.p))[self.lo:int(row_hi):self.stride]
        self._counts = (int((lens > 0).sum()), int(lens.sum()))

    def counts(self):
        return self._counts

    def advance(self):
        rows, nnz = C.c_int32(0), C.c_int64(0)
        plan, self.plan = self.plan, None  # consumed by the call, whatever happens
        sh = self.lib.spasm_amd_schur_plan_advance(plan, C.byref(rows), C.byref(nnz))
        if not sh:
            raise RuntimeError("spasm_amd_schur_plan_advance failed: " + _abi.last_error())
        if self.shard:
            self.lib.spasm_amd_shard_free(self.shard)
        self.shard = sh
        self._counts = (int(rows.value), int(nnz.value))
        return self._counts

    def round_U(self):
        """The pivot rows of the imported round as rows of U, from the device (spasm_amd_schur_plan_fetch_U):
        (row lengths, columns, values, pivot columns, original rows), in pivot order."""
        import numpy as np

        from .api import CSR

        npiv = int(self.npiv)
        pc = np.empty(max(npiv, 1), dtype=np.int32)
        ro = np.empty(max(npiv, 1), dtype=np.int32)
        P = C.POINTER(C.c_int32)
        ptr = self.lib.spasm_amd_schur_plan_fetch_U(self.plan, pc.ctypes.data_as(P), ro.ctypes.data_as(P))
        if not ptr:
            raise RuntimeError("spasm_amd_schur_plan_fetch_U failed: " + _abi.last_error())
        Uc = CSR(ptr)
        up = np.asarray(Uc.p)
        nz = int(up[Uc.n])
        return (np.diff(up).astype(np.int64), np.array(Uc.j[:nz], dtype=np.int64), np.array(Uc.x[:nz], dtype=np.int64),
                pc[:npiv].astype(np.int64), ro[:npiv].astype(np.int64))

    def fetch_rows(self):
        import numpy as np

        from .api import CSR

        ptr = self.lib.spasm_amd_shard_fetch(self.shard)
        if not ptr:
            raise RuntimeError("spasm_amd_shard_fetch failed: " + _abi.last_error())
        M = CSR(ptr)
        mp = np.asarray(M.p)
        lens = np.diff(mp)
        keep = np.flatnonzero(lens > 0)
        nnz = int(mp[M.n])
        ids = self.lo + keep.astype(np.int64) * self.stride
        # empty rows contribute nothing, so the entries of the kept rows are all entries, in order
        return ids, np.concatenate([[0], np.cumsum(lens[keep])]).astype(np.int64), np.array(M.j[:nnz]), np.array(M.x[:nnz])


def _virtual_csr(n, m, prime, ids, p, j, x):
    """An n-row matrix whose rows `ids` (ascending) hold the given rows and whose other rows are empty: a shard's view of
    the current matrix under the ORIGINAL row numbering, which is what keeps election keys comparable across ranks."""
    import numpy as np

    from .api import CSR

    lens = np.zeros(n + 1, dtype=np.int64)
    lens[np.asarray(ids, dtype=np.int64) + 1] = np.diff(p)
    return CSR.from_arrays(n, m, np.cumsum(lens), j, x, prime)


def _ranges(starts, lens):
    """concatenation of arange(starts[k], starts[k] + lens[k]) without a python loop"""
    import numpy as np

    starts = np.asarray(starts, dtype=np.int64)
    lens = np.asarray(lens, dtype=np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)])[:-1]
    return np.repeat(starts - offs, lens) + np.arange(total, dtype=np.int64)


def _balanced(v, prime):
    """the canonical representative in [p/2 - p + 1, p/2] (reference src/SpaSM.jl:83-88)"""
    v = v % prime
    return v - prime if v > prime // 2 else v


def _chk(rc, what):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise RuntimeError(what + " failed: " + _abi.last_error())
    return rc


def dense_round_sharded(eng, free_cols, sparsity_threshold, group=None, force=False, rows=False):
    """The round whose pivots have just been exchanged (eng.plan holds U, unprepared), when its Schur complement is dense: every
    rank's Schur rows go straight into its dense matrix and the ranks eliminate them TOGETHER (csrc/dense_multi.hpp: rows stay where
    they are; per panel of 64 columns an all-gather of the candidates' panel entries, the election on every rank, one broadcast of
    the winners' rows per owner).  Returns None when the round stays sparse (estimated density at or below the threshold), else this
    rank's share of the rows of U as (row lengths, columns, values, pivot columns, original rows) and the number of pivots found by
    all ranks.
    rows=True: no round at all -- the ranks' CURRENT rows (a remainder that is dense already) are eliminated together where they
    are (spasm_amd_dshard_open_rows; the one-process path's dense_now)."""
    import numpy as np

    from .api import CSR

    lib = eng.lib
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    ds = lib.spasm_amd_dshard_open_rows(eng.shard, rank, world) if rows else lib.spasm_amd_dshard_open(eng.plan, rank, world)
    if not ds:
        raise RuntimeError("spasm_amd_dshard_open failed: " + _abi.last_error())
    try:
        m = eng.m
        flags = torch.empty(m + 1, dtype=torch.int32, device=eng.device)
        _chk(lib.spasm_amd_dshard_flags(ds, C.c_void_p(flags.data_ptr())), "spasm_amd_dshard_flags")
        if world > 1:
            f = _collective_device(flags, group)
            dist.all_reduce(f, op=dist.ReduceOp.MAX, group=group)
            flags = f.to(eng.device)
        Cc = C.c_int32(0)
        est = lib.spasm_amd_dshard_density(ds, C.c_void_p(flags.data_ptr()), int(free_cols), C.byref(Cc))
        if est < 0:
            raise RuntimeError("spasm_amd_dshard_density failed: " + _abi.last_error())
        # rank 0's estimate decides for all (its rows are a strided sample of all rows)
        dec = torch.tensor([float(est), float(Cc.value)], dtype=torch.float64)
        if world > 1:
            d = dec if dist.get_backend(group) == "gloo" else dec.to(eng.device)
            dist.broadcast(d, src=0, group=group)
            dec = d.cpu()
        if int(dec[1]) <= 0 or not (force or float(dec[0]) > sparsity_threshold):
            return None
        _chk(lib.spasm_amd_dshard_build(ds), "spasm_amd_dshard_build")
        Cn, KB, ldc, elem, nd, cb = C.c_int32(0), C.c_int32(0), C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _chk(lib.spasm_amd_dshard_info(ds, C.byref(Cn), C.byref(KB), C.byref(ldc), C.byref(elem), C.byref(nd), C.byref(cb)), "spasm_amd_dshard_info")
        Cn, KB, ldc, elem, nd, cb = Cn.value, KB.value, ldc.value, elem.value, nd.value, cb.value
        cand = torch.empty(cb, dtype=torch.uint8, device=eng.device)
        npp, cnt, first = C.c_int32(0), (C.c_int32 * world)(), (C.c_int32 * world)()
        # Ranks that SHARE a device (rehearsals on a one-GPU box) take turns in the one step that launches the panel kernel over the
        # shard's rows: it is a persistent grid with its own barrier, and two of them from different processes cannot both be resident
        # on one device.  (The election among the stacked candidates is one workgroup: no turn needed.)  One device per rank: no barrier.
        shared = world > 1 and torch.cuda.is_available() and torch.cuda.device_count() < world
        for b0 in range(0, Cn, KB):
            b1 = min(b0 + KB, Cn)
            _chk(lib.spasm_amd_dshard_block_begin(ds), "spasm_amd_dshard_block_begin")
            q = 0
            for c0 in range(b0, b1, 64):
                w = min(c0 + 64, b1) - c0
                if shared:
                    rc = 0
                    for turn in range(world):
                        if turn == rank:
                            rc = lib.spasm_amd_dshard_candidates(ds, c0, w, C.c_void_p(cand.data_ptr()))
                        dist.barrier(group=group)
                    _chk(rc, "spasm_amd_dshard_candidates")
                else:
                    _chk(lib.spasm_amd_dshard_candidates(ds, c0, w, C.c_void_p(cand.data_ptr())), "spasm_amd_dshard_candidates")
                if world > 1:
                    cdev = _collective_device(cand, group)
                    outs = [torch.empty_like(cdev) for _ in range(world)]
                    dist.all_gather(outs, cdev, group=group)
                    stack = torch.cat(outs).to(eng.device)
                else:
                    stack = cand
                _chk(lib.spasm_amd_dshard_elect(ds, C.c_void_p(stack.data_ptr()), w, C.byref(npp), cnt, first), "spasm_amd_dshard_elect")
                for k in range(world):
                    if cnt[k] <= 0:
                        continue
                    nbytes = cnt[k] * (ldc - c0) * elem + nd * cnt[k] * KB
                    buf = torch.empty(nbytes, dtype=torch.uint8, device=eng.device)
                    if k == rank:
                        got = lib.spasm_amd_dshard_pack(ds, c0, C.c_void_p(buf.data_ptr()))
                        if got != nbytes:
                            raise RuntimeError(f"spasm_amd_dshard_pack wrote {got} bytes, {nbytes} expected: " + _abi.last_error())
                    if world > 1:
                        bdev = _collective_device(buf, group)
                        dist.broadcast(bdev, src=k, group=group)
                        buf = bdev.to(eng.device)
                    _chk(lib.spasm_amd_dshard_unpack(ds, q, c0, k, C.c_void_p(buf.data_ptr())), "spasm_amd_dshard_unpack")
                _chk(lib.spasm_amd_dshard_apply(ds, q, c0, w, b1), "spasm_amd_dshard_apply")
                q += 1
            _chk(lib.spasm_amd_dshard_block_end(ds, b0, b1, q), "spasm_amd_dshard_block_end")
        npiv = lib.spasm_amd_dshard_finish(ds)
        if npiv < 0:
            raise RuntimeError("spasm_amd_dshard_finish failed: " + _abi.last_error())
        pc = np.empty(max(Cn, 1), dtype=np.int32)
        ro = np.empty(max(Cn, 1), dtype=np.int32)
        nown = C.c_int32(0)
        P = C.POINTER(C.c_int32)
        ptr = lib.spasm_amd_dshard_fetch_U(ds, pc.ctypes.data_as(P), ro.ctypes.data_as(P), C.byref(nown))
        if not ptr:
            raise RuntimeError("spasm_amd_dshard_fetch_U failed: " + _abi.last_error())
        Uc = CSR(ptr)
        up = np.asarray(Uc.p)
        nz = int(up[Uc.n])
        k = int(nown.value)
        return ((np.diff(up).astype(np.int64), np.array(Uc.j[:nz], dtype=np.int64), np.array(Uc.x[:nz], dtype=np.int64),
                 pc[:k].astype(np.int64), ro[:k].astype(np.int64)), int(npiv))
    finally:
        lib.spasm_amd_dshard_close(ds)


def _gather_U_blocks(blk, group):
    """every rank's share of the rows of U (dense_round_sharded) on every rank, in rank order"""
    import numpy as np

    lens, cols, vals, pcs, origs = blk
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return blk
    counts = all_gather_counts([len(lens), len(cols)], group)
    dev = "cuda" if dist.get_backend(group) != "gloo" else "cpu"
    rows = torch.as_tensor(np.stack([lens, pcs, origs], axis=1).reshape(-1, 3), dtype=torch.int64, device=dev)
    ents = torch.as_tensor(np.stack([cols, vals], axis=1).reshape(-1, 2), dtype=torch.int64, device=dev)
    g_rows = all_gather_var(rows, counts[:, 0].tolist(), group).cpu().numpy()
    g_ents = all_gather_var(ents, counts[:, 1].tolist(), group).cpu().numpy()
    return (g_rows[:, 0].astype(np.int64), g_ents[:, 0].astype(np.int64), g_ents[:, 1].astype(np.int64), g_rows[:, 1].astype(np.int64),
            g_rows[:, 2].astype(np.int64))


def echelonize_sharded(A, group=None, finish_nnz=1 << 22, max_rounds=1 << 30, engine_cls=None, finish=None, dense_over_shards=None,
                       open_columns=False):
    """Row-sharded echelonize of A (every rank passes the same matrix; rank r keeps rows r, r + G, ...).
    Per round: all-reduce(MIN) of the election keys, all-gather of the elected pivot rows, local Schur complement of the
    rank's rows.  When at most `finish_nnz` entries are left in total, or the remainder is dense enough for the dense tail
    (the single-device rule), or after `max_rounds` rounds, the remaining rows are all-gathered and every rank finishes
    them with the single-device engine, so all ranks return the same LU.
    open_columns=True: the rounds also run "Faugere-Lachartre on columns" over the shards (GpuShardEngine.set_keys_open: what the
    single-device round does under enable_greedy_pivot_search, without its third, cycle-free search) -- the pivots of the sharded
    rounds are then those of the single-device rounds with SPASM_AMD_NO_CYCLE_FREE_SEARCH=1, whatever the number of ranks.
    Returns (LU, info)."""
    import numpy as np

    from . import api

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    engine_cls = engine_cls or GpuRoundEngine
    # the sharded rounds elect leftmost-entry pivots only (one all-reduce per round); the finish keeps to them as well, so that the
    # pivot columns and the kernel basis do not depend on the number of ranks
    finish = finish or (lambda M: api.echelonize(M, enable_greedy_pivot_search=False))
    sparsity_threshold = float(api.EchelonizeOpts().struct.sparsity_threshold)
    n, m, prime = A.n, A.m, int(A.prime)
    eng = engine_cls(A, rank, n, stride=world)  # the rank's rows stay on its device from here on
    # the dense finish over the ranks: primes below 2^16 (the int8 path), engines that have the steps (the numpy engine of the
    # protocol tests has not).  With it the "one round ahead" rule below is not needed: the estimate after the exchange decides.
    if dense_over_shards is None:
        dense_over_shards = True
    dense_ok = bool(dense_over_shards) and prime < 65536 and isinstance(eng, GpuRoundEngine) and world <= 16
    # .. and remainders worth it: below 16 GiB of dense matrix one device finishes faster than the ranks exchange panels (a 44k x 44k
    # remainder over two gloo ranks: 25 s of collectives against 1 s).  SPASM_AMD_MULTI_DENSE_MIN_BYTES (tests): 0.
    import os as _os

    dense_min_bytes = float(_os.environ.get("SPASM_AMD_MULTI_DENSE_MIN_BYTES", 1 << 34))
    dense_allowed, elem = dense_ok, (1 if prime <= 255 else 2)
    blocks = []  # per round: (row lengths, columns, values, pivot columns, original rows) of its rows of U
    n_u = 0
    rounds = []
    import time as _time

    try:
        while True:
            t_round = _time.time()
            tot = all_gather_counts(list(eng.counts()), group).sum(dim=0)
            rows_left, nnz_left = int(tot[0]), int(tot[1])
            if nnz_left == 0:
                break
            free_cols = m - n_u
            dense_ok = dense_allowed and rows_left * max(free_cols, 1) * elem >= dense_min_bytes
            dense_enough = nnz_left > sparsity_threshold * rows_left * max(free_cols, 1)  # the single-device rule for its dense tail
            # ... applied one round ahead as well, like the single-device density estimate (spasm_schur_estimate_density): when the
            # fill keeps growing at the rate of the last round, would the NEXT Schur complement be dense?  Then it is never built
            # sparse here (100k x 100k, 6 per row: the round that was skipped wrote 3.4e8 entries that the hand-off then had to
            # gather: 9 of 15 s).
            if not dense_ok and rounds and not rounds[-1]["finish"] and rounds[-1]["nnz"] > 0 and nnz_left > rounds[-1]["nnz"]:
                predicted = nnz_left * (nnz_left / rounds[-1]["nnz"])
                dense_enough = dense_enough or predicted > sparsity_threshold * rows_left * max(free_cols, 1)
            if dense_ok and dense_enough and nnz_left > finish_nnz and len(rounds) < max_rounds:
                # the remainder is dense ALREADY and worth sharing (these are the remainders that do not fit one device):
                # no hand-off, no further round -- all ranks eliminate their rows together where they are
                t_d = _time.time()
                got = dense_round_sharded(eng, free_cols, sparsity_threshold, group, force=True, rows=True)
                if got is not None:
                    dblk, dpiv = got
                    dblk = _gather_U_blocks(dblk, group)
                    if len(dblk[0]):
                        blocks.append(dblk)
                        n_u += len(dblk[0])
                    rounds.append({"round": len(rounds), "finish": True, "dense_over_shards": True, "dense_rows_as_they_are": True, "rows": rows_left,
                                   "nnz": nnz_left, "npiv": int(dpiv), "seconds": {"dense_finish": _time.time() - t_d}})
                    break
            if nnz_left <= finish_nnz or len(rounds) >= max_rounds or dense_enough:
                # hand-off: every rank gets all remaining rows and finishes them (deterministic, so the results agree)
                ids, p, j, x = eng.fetch_rows()
                t_fetch = _time.time()
                counts = all_gather_counts([len(ids), int(p[-1])], group)
                dev = "cuda" if dist.is_initialized() and dist.get_backend(group) != "gloo" else "cpu"
                g_ids = all_gather_var(torch.as_tensor(ids, dtype=torch.int64, device=dev), counts[:, 0].tolist(), group).cpu().numpy()
                g_len = all_gather_var(torch.as_tensor(np.diff(p), dtype=torch.int64, device=dev), counts[:, 0].tolist(), group).cpu().numpy()
                g_ent = all_gather_var(torch.as_tensor(np.stack([j, x], axis=1).reshape(-1, 2), dtype=torch.int32, device=dev),
                                       counts[:, 1].tolist(), group).cpu().numpy()
                order = np.argsort(g_ids, kind="stable")
                starts = np.concatenate([[0], np.cumsum(g_len)])
                sel = _ranges(starts[:-1][order], g_len[order])
                rest = _virtual_csr(n, m, prime, g_ids[order], np.concatenate([[0], np.cumsum(g_len[order])]), g_ent[sel, 0], g_ent[sel, 1])
                t_gather = _time.time()
                # Ranks that SHARE a device (rehearsals on a one-GPU box) take turns: the dense finish's panel kernel is a persistent
                # grid with its own barrier, and two of them from different processes cannot both be resident on one device (the
                # kernel gives up with "a grid barrier ... timed out").  One device per rank: all at once.
                shared = world > 1 and torch.cuda.is_available() and torch.cuda.device_count() < world
                if shared:
                    fact = None
                    for turn in range(world):
                        if turn == rank:
                            fact = finish(rest)
                        dist.barrier(group=group)
                else:
                    fact = finish(rest)
                t_finish = _time.time()
                Uc, fp, fq = fact.U, np.asarray(fact.p), np.asarray(fact.qinv)
                up, uj, ux = np.asarray(Uc.p), np.asarray(Uc.j), np.asarray(Uc.x)
                col_of_row = np.full(fact.r, -1, dtype=np.int64)
                col_of_row[fq[fq >= 0]] = np.flatnonzero(fq >= 0)
                nzu = int(up[fact.r])
                blocks.append((np.diff(up).astype(np.int64), np.array(uj[:nzu]), np.array(ux[:nzu]), col_of_row, fp[: fact.r].astype(np.int64)))
                n_u += fact.r
                rounds.append({"round": len(rounds), "finish": True, "rows": rows_left, "nnz": nnz_left, "npiv": int(fact.r),
                               "seconds": {"fetch_rows": t_fetch - t_round, "gather": t_gather - t_fetch, "finish": t_finish - t_gather,
                                           "collect_U": _time.time() - t_finish}})
                break
            keys = all_reduce_min(eng.elect(), group)
            n_open = 0
            if open_columns and hasattr(eng, "set_keys_open"):
                npiv, n_own, nnz_own, n_open = eng.set_keys_open(keys, group)
            else:
                npiv, n_own, nnz_own = eng.set_keys(keys)
            t_elect = _time.time()
            if npiv == 0:
                break
            counts = all_gather_counts([n_own, nnz_own], group)
            hdr, ent = eng.export()
            hdr_all = all_gather_var(hdr, counts[:, 0].tolist(), group)
            ent_all = all_gather_var(ent, counts[:, 1].tolist(), group)
            if dense_ok:
                eng.import_(hdr_all, ent_all, prepare=False)
            else:
                eng.import_(hdr_all, ent_all)
            t_exchange = _time.time()
            # the round's rows of U come from the engine as it built them on the device (scaled to a unit pivot, in pivot order):
            # nothing is recomputed on the host
            blk = eng.round_U()
            if len(blk[0]):
                blocks.append(blk)
                n_u += len(blk[0])
            t_u = _time.time()
            if dense_ok:
                # is the Schur complement of this round dense?  Then it is never built sparse: all ranks eliminate it together
                got = dense_round_sharded(eng, m - n_u, sparsity_threshold, group)
                if got is not None:
                    dblk, dpiv = got
                    dblk = _gather_U_blocks(dblk, group)
                    if len(dblk[0]):
                        blocks.append(dblk)
                        n_u += len(dblk[0])
                    rounds.append({"round": len(rounds), "finish": False, "rows": rows_left, "nnz": nnz_left, "npiv": int(npiv), "npiv_open": int(n_open), "seconds": {}})
                    rounds.append({"round": len(rounds), "finish": True, "dense_over_shards": True, "rows": rows_left, "nnz": -1, "npiv": int(dpiv),
                                   "seconds": {"dense_finish": _time.time() - t_u}})
                    break
                eng.prepare()
            eng.advance()  # the round runs; its Schur rows are the shard's matrix of the next round, still on the device
            rounds.append({"round": len(rounds), "finish": False, "rows": rows_left, "nnz": nnz_left, "npiv": int(npiv), "npiv_open": int(n_open),
                           "gathered_bytes": int(hdr_all.numel() * 4 + ent_all.numel() * 4),
                           "seconds": {"elect": t_elect - t_round, "exchange_import": t_exchange - t_elect, "fetch_U": t_u - t_exchange,
                                       "schur_advance": _time.time() - t_u}})
    finally:
        eng.close()

    r = n_u

    def cat(i, dt):
        return np.concatenate([b[i] for b in blocks]).astype(dt) if blocks else np.zeros(0, dtype=dt)

    up = np.concatenate([[0], np.cumsum(cat(0, np.int64))]).astype(np.int64)
    uj, ux = cat(1, np.int32), cat(2, np.int32)
    u_pivcol, u_orig = cat(3, np.int64), cat(4, np.int64)
    U = api.CSR.from_arrays(r, m, up, uj, ux, prime)
    qinv = np.full(max(m, 1), -1, dtype=np.int32)
    if r:
        qinv[np.asarray(u_pivcol, dtype=np.int64)] = np.arange(r, dtype=np.int32)
    plen = max(n, m, 1)
    perm = np.full(plen, -1, dtype=np.int32)
    used = np.zeros(max(n, 1), dtype=bool)
    if r:
        perm[:r] = np.asarray(u_orig, dtype=np.int32)
        used[np.asarray(u_orig, dtype=np.int64)] = True
    others = np.flatnonzero(~used[:n])
    perm[r:r + len(others)] = others
    return api.LU.from_parts(U, qinv[:m] if m else qinv, perm), {"rounds": rounds, "world": world}


def kernel_sharded(fact, group=None):
    """Right kernel of an echelonized matrix with the free columns sharded over the ranks: rank r computes the
    vectors of the free columns number r, r + G, ... (spasm_amd_kernel_strided), one variable-length all-gather puts the
    basis together in the order of spasm_kernel(fact).  Every rank returns the whole basis."""
    import numpy as np

    from . import api

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    ptr = _abi.lib().spasm_amd_kernel_strided(fact.data, rank, world)
    if not ptr:
        raise RuntimeError("spasm_amd_kernel_strided failed: " + _abi.last_error())
    Kl = api.CSR(ptr)
    if world == 1:
        return Kl
    kp = np.asarray(Kl.p)
    nnz = int(kp[Kl.n])
    counts = all_gather_counts([Kl.n, nnz], group)
    dev = "cuda" if dist.get_backend(group) != "gloo" else "cpu"
    g_len = all_gather_var(torch.as_tensor(np.diff(kp), dtype=torch.int64, device=dev), counts[:, 0].tolist(), group).cpu().numpy()
    ent = np.stack([np.asarray(Kl.j[:nnz]), np.asarray(Kl.x[:nnz])], axis=1).reshape(-1, 2)
    g_ent = all_gather_var(torch.as_tensor(ent, dtype=torch.int32, device=dev), counts[:, 1].tolist(), group).cpu().numpy()
    # vector f of the whole basis is vector f // G of rank f % G
    per = counts[:, 0].numpy().astype(np.int64)
    first_row = np.concatenate([[0], np.cumsum(per)])[:-1]            # where rank r's rows start in the gathered order
    total = int(per.sum())
    f = np.arange(total, dtype=np.int64)
    src = first_row[f % world] + f // world                           # gathered index of vector f
    starts = np.concatenate([[0], np.cumsum(g_len)])
    sel = _ranges(starts[:-1][src], g_len[src])
    p_out = np.concatenate([[0], np.cumsum(g_len[src])]).astype(np.int64)
    return api.CSR.from_arrays(total, Kl.m, p_out, g_ent[sel, 0], g_ent[sel, 1], int(Kl.prime))
