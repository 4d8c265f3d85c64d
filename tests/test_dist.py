"""world_size-2 tests of the multi-GPU exchange steps on CPU (gloo): the two sums ShardedEsvo runs between
the phases of a sharded tick (on host tensors here), and the band bookkeeping."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from esvo_amd import dist as edist
from esvo_amd.abi import DEPTH_POINT_DTYPE


def _worker(rank, world, port, height, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        n = 1000 + height  # not a multiple of 8: the byte buffer is padded to whole 64-bit words
        words = DEPTH_POINT_DTYPE.itemsize // 8
        # exchange 1: one byte per slot (bit 0 matched, bit 1 kept); slot w belongs to rank w % world
        codes_ref = rng.choice(np.array([0, 1, 3], np.uint8), n)
        mine = np.zeros((n + 7) // 8 * 8, np.uint8)
        own = np.arange(n) % world == rank
        mine[:n][own] = codes_ref[own]
        t = torch.from_numpy(mine.view(np.int64))
        edist.merge_disjoint_(t)
        ok1 = np.array_equal(mine[:n], codes_ref) and not mine[n:].any()
        # exchange 2: the frame, 13 words per kept point (any bit pattern, negative words included), zero elsewhere
        kept = np.flatnonzero(codes_ref == 3)
        frame_ref = rng.integers(-2**63, 2**63 - 1, (len(kept), words), dtype=np.int64)
        frame = np.where((kept % world == rank)[:, None], frame_ref, 0).astype(np.int64).reshape(-1)
        t = torch.from_numpy(frame)
        edist.merge_disjoint_(t)
        ok2 = np.array_equal(frame.reshape(-1, words), frame_ref)
        q.put((rank, ok1, ok2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height", [8, 9])
def test_exchange_primitives_gloo_world2(height):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + height
    procs = [ctx.Process(target=_worker, args=(r, 2, port, height, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok1 and ok2 for _, ok1, ok2 in res), res


def test_band_bookkeeping_and_merge():
    assert [edist.band_of(r, 8, 480) for r in range(8)] == [(60 * r, 60 * r + 60) for r in range(8)]
    assert [edist.band_of(r, 4, 10) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    a = np.zeros(3, DEPTH_POINT_DTYPE); a["seq"] = [5, 1, 9]; a["row"] = [0, 0, 1]
    b = np.zeros(2, DEPTH_POINT_DTYPE); b["seq"] = [3, 7]; b["row"] = [5, 6]
    m = edist.merge_band_maps([a, b, np.zeros(0, DEPTH_POINT_DTYPE)])
    assert list(m["row"]) == [0, 5, 0, 6, 1] and list(m["seq"]) == [0, 1, 2, 3, 4]
