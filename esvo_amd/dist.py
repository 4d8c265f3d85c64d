"""Multi-GPU sharding of the mapper: one process per GPU, the two exchange steps of a
tick run through torch.distributed (backend "nccl" = RCCL over xGMI) on device buffers the C-ABI hands
out (esvo_shard_exchange).  PyTorch is plumbing here: device-pointer views + collectives; every
kernel is in libesvo_hip.so.

Per tick (see esvo_shard_tick_phase in include/esvo_hip.h):
    phase 0  BM + LM + culling of every world-th slot      -> sum of one byte per slot (matched, kept)
    phase 1  frame order from the bytes, own points placed  -> sum of the frame (104 B per kept point)
    phase 2  window policy, fusion + clean + regularisation of the row band (halo rows recomputed locally)
Entries of other ranks are zero in both buffers, so SUM over 64-bit integer words is an exact union.
"""
import numpy as np

from . import lib
from .abi import DEPTH_POINT_DTYPE


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

    def _exchange(self):
        ptr, nbytes = self.dev.shard_exchange()
        if nbytes:
            merge_disjoint_(device_tensor(ptr, nbytes // 8, "<i8"), self.group)

    def tick(self, t_ns, stamps, poses):
        d = self.dev
        d.shard_phase(0, t_ns, stamps, poses)
        self._exchange()
        d.shard_phase(1)
        self._exchange()
        d.shard_phase(2)

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
