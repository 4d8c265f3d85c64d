"""Multi-GPU row-band sharding of the mapper: one process per GPU, the collectives
of the one real exchange step run through torch.distributed (backend "nccl" = RCCL over xGMI) on the
device buffers the C-ABI exposes (esvo_shard_buffers).  PyTorch is plumbing here: device-pointer
views + collectives; every kernel is in libesvo_hip.so.

Per tick (see esvo_shard_tick_phase in include/esvo_hip.h):
    phase 0 (BM of every world-th event)   -> all-reduce(SUM) match flags
    phase 1 (order, LM + cull of own ones) -> all-reduce(SUM) point flags + point slots
    phase 2 (frame, window, fuse, clean)   -> all-gather of the regulariser view's row bands
    phase 3 (regularise the band)
Foreign entries of the summed buffers are zero, so SUM on integer views is an exact union.
"""
import numpy as np

from . import lib
from .abi import DEPTH_POINT_DTYPE

POINT_WORDS = DEPTH_POINT_DTYPE.itemsize // 8  # 13 int64 words per esvo_depth_point_t


class _DevView:
    """Zero-copy view of library-owned device memory through the CUDA array interface."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, n, typestr):
    import torch
    return torch.as_tensor(_DevView(ptr, n, typestr), device="cuda")


def band_of(rank, world, height):
    rows = -(-height // world)
    return min(rank * rows, height), min((rank + 1) * rows, height)


# ---- the exchange primitives (backend agnostic: exercised with gloo on CPU in tests/test_dist.py) ----
def merge_disjoint_(t, group=None):
    """all-reduce(SUM) of an integer tensor whose non-owned entries are zero == union of the owners' entries"""
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def gather_row_bands_(full, width, height, rank, world, group=None):
    """every rank owns rows band_of(rank) of `full` (row-major, `width*k` entries per row); afterwards every rank
    holds all rows.  Equal bands use one all-gather; ragged bands fall back to one broadcast per rank."""
    import torch
    import torch.distributed as dist
    per_row = full.numel() // height
    y0, y1 = band_of(rank, world, height)
    if height % world == 0 and dist.get_backend(group) == "nccl":
        mine = full[y0 * per_row:y1 * per_row].clone()
        dist.all_gather_into_tensor(full[: height * per_row], mine, group=group)
    else:
        for r in range(world):
            a, b = band_of(r, world, height)
            if b > a:
                dist.broadcast(full[a * per_row:b * per_row], src=r, group=group)
    return full


class ShardedEsvo:
    """Same driving interface as lib.Esvo for bench.py / tests, one instance per rank."""

    counts_are_local = False

    def __init__(self, params, rig, rank, world, local_rank, group=None):
        import torch
        self.rank, self.world, self.group = rank, world, group
        self.rig, self.params = rig, params
        self.W, self.H = rig.width, rig.height
        self.dev = lib.Esvo(params, rig, device=local_rank)
        self.dev.set_stream(torch.cuda.current_stream().cuda_stream)
        self.y0, self.y1 = band_of(rank, world, self.H)
        if self.y1 <= self.y0:
            raise lib.EsvoError(f"rank {rank} of {world} would own no image rows (H={self.H})")
        self.dev.set_band(self.y0, self.y1, rank, world)
        b = self.dev.shard_buffers()
        self.t_mflags = device_tensor(b.d_match_flags, b.max_events, "<i4")
        self.t_pflags = device_tensor(b.d_point_flags, b.max_events, "<i4")
        self.t_pslots = device_tensor(b.d_point_slots, b.max_events * POINT_WORDS, "<i8")
        self.t_valid = device_tensor(b.d_reg_valid, b.n_cells, "|u1")
        self.t_ab = device_tensor(b.d_reg_ab, b.n_cells * 2, "<f8")
        self.t_cd = device_tensor(b.d_reg_cd, b.n_cells * 2, "<f8")

    # replicated stages: every rank ingests all events and renders the full Time Surfaces
    def ts_push_events(self, cam, ev):
        self.dev.ts_push_events(cam, ev)

    def ts_render(self, cam, t_ns, download=True):
        return self.dev.ts_render(cam, t_ns, download)

    def set_observation(self, *a):
        self.dev.set_observation(*a)

    def synchronize(self):
        self.dev.synchronize()

    def stats(self):
        return self.dev.stats()

    def tick(self, t_ns, stamps, poses):
        d = self.dev
        d.shard_phase(0, t_ns, stamps, poses)
        n = d.stats().last_events_in
        if n:
            merge_disjoint_(self.t_mflags[:n], self.group)
        d.shard_phase(1)
        m = d.stats().last_matches
        if m:
            merge_disjoint_(self.t_pflags[:m], self.group)
            merge_disjoint_(self.t_pslots[: m * POINT_WORDS], self.group)
        d.shard_phase(2)
        if self.params.regularization:
            gather_row_bands_(self.t_valid, self.W, self.H, self.rank, self.world, self.group)
            gather_row_bands_(self.t_ab, self.W, self.H, self.rank, self.world, self.group)
            gather_row_bands_(self.t_cd, self.W, self.H, self.rank, self.world, self.group)
        d.shard_phase(3)

    def get_band_map(self):
        """this rank's DepthMap elements; `seq` is the global creation id"""
        return self.dev.get_map()

    def get_map(self):
        """the full DepthMap on every rank, in the reference's list order"""
        import torch.distributed as dist
        parts = [None] * self.world
        dist.all_gather_object(parts, self.get_band_map(), group=self.group)
        return merge_band_maps(parts)


def merge_band_maps(parts):
    allp = np.concatenate([p for p in parts if len(p)]) if any(len(p) for p in parts) else np.zeros(0, DEPTH_POINT_DTYPE)
    order = np.argsort(allp["seq"], kind="stable")
    out = allp[order].copy()
    out["seq"] = np.arange(len(out), dtype=np.uint32)
    return out
