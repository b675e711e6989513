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
        return int(npiv), self.n_own, self.nnz_own

    def export(self):
        hdr = torch.empty((max(self.n_own, 1), 2), dtype=torch.int32, device=self.device)
        ent = torch.empty((max(self.nnz_own, 1), 2), dtype=torch.int32, device=self.device)
        if self.lib.spasm_amd_shard_export(self.shard, C.c_void_p(hdr.data_ptr()), C.c_void_p(ent.data_ptr())) != 0:
            raise RuntimeError("spasm_amd_shard_export failed: " + _abi.last_error())
        return hdr[: self.n_own], ent[: self.nnz_own]

    def import_(self, hdr_all, ent_all):
        hdr_all = hdr_all.to(self.device).contiguous()
        ent_all = ent_all.to(self.device).contiguous()
        plan = self.lib.spasm_amd_shard_import(self.shard, int(hdr_all.shape[0]), int(ent_all.shape[0]),
                                               C.c_void_p(hdr_all.data_ptr()), C.c_void_p(ent_all.data_ptr()))
        if not plan:
            raise RuntimeError("spasm_amd_shard_import failed: " + _abi.last_error())
        self.plan = plan
        return plan

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
