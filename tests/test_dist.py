"""world_size-2 tests of the multi-GPU exchange step on CPU (gloo): the collectives ShardedEsvo runs
between the phases of a sharded tick, and the band bookkeeping."""
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
        n = 1000
        # every "match" is owned by exactly one rank (the band of its row); foreign slots are zero
        owner = rng.integers(0, world, n)
        flags_ref = rng.integers(0, 2, n).astype(np.int32)
        slots_ref = rng.integers(1, 2**62, (n, edist.POINT_WORDS)).astype(np.int64) * flags_ref[:, None]
        flags = torch.from_numpy(np.where(owner == rank, flags_ref, 0).astype(np.int32))
        slots = torch.from_numpy(np.where((owner == rank)[:, None], slots_ref, 0)).reshape(-1)
        edist.merge_disjoint_(flags)
        edist.merge_disjoint_(slots)
        ok1 = np.array_equal(flags.numpy(), flags_ref) and np.array_equal(slots.numpy().reshape(n, -1), slots_ref)
        # row bands (equal and ragged heights), 2 values per cell like the (inv_depth, 2 sigma) view
        width = 7
        full_ref = rng.random((height, width, 2))
        full = torch.zeros(height * width * 2, dtype=torch.float64)
        y0, y1 = edist.band_of(rank, world, height)
        full.view(height, width, 2)[y0:y1] = torch.from_numpy(full_ref[y0:y1])
        edist.gather_row_bands_(full, width, height, rank, world)
        ok2 = np.array_equal(full.numpy().reshape(height, width, 2), full_ref)
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
