"""world_size-2 tests of the multi-GPU exchange steps on CPU (gloo): the two all-gathers ShardedEsvo runs between
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
        n = 1000 + height  # not a multiple of 8 * world: blocks are padded to whole 64-bit words
        words = DEPTH_POINT_DTYPE.itemsize // 8
        # exchange 1: one byte per OWN slot (bit 0 matched, bit 1 kept), slots rank, rank + world ... back to back
        codes_ref = rng.choice(np.array([0, 1, 3], np.uint8), n)
        blk = (-(-n // world) + 7) // 8 * 8
        mine = np.zeros(blk, np.uint8)
        own = codes_ref[rank::world]
        mine[: len(own)] = own
        recv = np.zeros(world * blk, np.uint8)
        edist.gather_blocks_(torch.from_numpy(recv.view(np.int64)), torch.from_numpy(mine.view(np.int64)), world)
        back = np.zeros(n, np.uint8)
        for r in range(world):
            back[r::world] = recv[r * blk: r * blk + len(range(r, n, world))]
        ok1 = np.array_equal(back, codes_ref)
        # exchange 2: [count | kept points] per rank, block length = the largest kept count (13 words per point, any bit
        # pattern, negative words included); the receiver places every point by the index it carries
        kept = np.flatnonzero(codes_ref == 3)
        frame_ref = rng.integers(-2**63, 2**63 - 1, (len(kept), words), dtype=np.int64)
        counts = [int((kept % world == r).sum()) for r in range(world)]
        bw = 1 + words * max(counts)
        send = np.zeros(bw, np.int64)
        idx_mine = np.flatnonzero(kept % world == rank)
        send[0] = len(idx_mine)
        send[1: 1 + words * len(idx_mine)] = frame_ref[idx_mine].reshape(-1)
        recv2 = np.zeros(world * bw, np.int64)
        edist.gather_blocks_(torch.from_numpy(recv2), torch.from_numpy(send), world)
        frame = np.zeros_like(frame_ref)
        for r in range(world):
            c = int(recv2[r * bw])
            frame[np.flatnonzero(kept % world == r)] = recv2[r * bw + 1: r * bw + 1 + words * c].reshape(c, words)
        ok2 = [int(recv2[r * bw]) for r in range(world)] == counts and np.array_equal(frame, frame_ref)
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


def _routed_worker(rank, world, port, q):
    """exchange 1 of the ROUTED band mode (kernels_shard.hip, shard_codes_routed / shard_unpack_routed, restated in numpy):
    a rank's slots are wherever its rows' events fell, so its block spans the whole tick -- two bits per slot, own slots set,
    whole 64-bit words -- and the receiver takes the one non-zero field per slot and every rank's kept count"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(11)           # same stream on every rank
        n = 1003
        owner = rng.integers(0, world, n)         # which rank's band the slot's event fell into
        codes_ref = rng.choice(np.array([0, 1, 3], np.uint8), n)
        blk = (-(-n // 16) * 4 + 7) // 8 * 8
        words = np.zeros(blk // 4, np.uint32)
        for w in np.flatnonzero(owner == rank):
            words[w >> 4] |= np.uint32(int(codes_ref[w]) << (2 * (w & 15)))
        recv = np.zeros(world * blk, np.uint8)
        edist.gather_blocks_(torch.from_numpy(recv.view(np.int64)), torch.from_numpy(words.view(np.int64)), world)
        back = np.zeros(n, np.uint8)
        kept = []
        for r in range(world):
            wr = recv[r * blk: (r + 1) * blk].view(np.uint32)
            f = np.stack([(wr >> np.uint32(2 * qq)) & np.uint32(3) for qq in range(16)], 1).reshape(-1)[:n].astype(np.uint8)
            assert not np.any((back != 0) & (f != 0))          # one owner per slot
            back |= f
            kept.append(int(((f >> 1) & 1).sum()))
        ok = np.array_equal(back, codes_ref) and kept == [int(((codes_ref[owner == r] >> 1) & 1).sum()) for r in range(world)]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_routed_exchange_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_routed_worker, args=(r, world, 29660 + world, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(res) == world and all(ok for _, ok in res), res


def test_band_bookkeeping_and_merge():
    assert [edist.band_of(r, 8, 480) for r in range(8)] == [(60 * r, 60 * r + 60) for r in range(8)]
    assert [edist.band_of(r, 4, 10) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    a = np.zeros(3, DEPTH_POINT_DTYPE); a["seq"] = [5, 1, 9]; a["row"] = [0, 0, 1]
    b = np.zeros(2, DEPTH_POINT_DTYPE); b["seq"] = [3, 7]; b["row"] = [5, 6]
    m = edist.merge_band_maps([a, b, np.zeros(0, DEPTH_POINT_DTYPE)])
    assert list(m["row"]) == [0, 5, 0, 6, 1] and list(m["seq"]) == [0, 1, 2, 3, 4]


# ---- tick-interleaved driver (dist.TickShardedEsvo) on CPU: two gloo ranks, a host-memory stand-in for lib.Esvo ------
class _FakeEsvo:
    """Records what TickShardedEsvo asks of the handle; a tick's frame is n(k) points whose 13 words encode (k, i, w)."""

    def __init__(self):
        self.log, self.k_seen, self.keep = [], [], []

    def ts_render(self, cam, t_ns, download=True):
        self.log.append(("render", cam, t_ns))

    def set_observation(self, *a):
        self.log.append(("obs", a[0]))

    def front(self, t_ns, stamps, poses):
        k = int(t_ns)                      # the test passes the tick index as time stamp
        n = 3 + (k * 7) % 5
        frame = np.empty((n, 13), np.int64)
        for i in range(n):
            frame[i] = [k * 1000 + i * 13 + w for w in range(13)]
        self.keep.append(frame)            # keeps the buffer alive, like d_pts_tmp
        self.log.append(("front", k, n))
        return frame.ctypes.data, n

    def push_frame_device(self, ptr, n, poses):
        import ctypes
        words = np.frombuffer((ctypes.c_char * (n * 104)).from_address(int(ptr)), np.int64, n * 13).reshape(n, 13).copy() if n else np.zeros((0, 13), np.int64)
        self.log.append(("push", words, np.asarray(poses).copy()))

    def fuse_async(self):
        self.log.append(("fuse",))

    def synchronize(self):
        pass


def _tick_worker(rank, world, port, n_ticks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fake = _FakeEsvo()
        drv = edist.TickShardedEsvo(None, None, rank, world, 0, dev=fake, device="cpu")
        for k in range(n_ticks):
            drv.ts_render(0, k, download=False)
            drv.set_observation(k, None, None, None)
            drv.tick(k, np.arange(2, dtype=np.uint64), np.full((2, 16), float(k)))
            if k == 2:
                drv.synchronize()            # a flush in the middle of a round (as bench.py does after its warm-up)
        drv.synchronize()
        fronts = [e[1] for e in fake.log if e[0] == "front"]
        renders = [e[2] for e in fake.log if e[0] == "render"]
        pushes = [e for e in fake.log if e[0] == "push"]
        ok = fronts == [k for k in range(n_ticks) if k % world == rank] == renders
        ok &= len(pushes) == n_ticks
        for k, (_, words, poses) in enumerate(pushes):       # every tick's frame, in tick order, content intact
            n = 3 + (k * 7) % 5
            ok &= words.shape == (n, 13) and bool((words[:, 0] == k * 1000 + 13 * np.arange(n)).all()) and bool((poses == float(k)).all())
        # the fusion of an own tick follows the push of that tick and precedes the push of the next one
        order = [("p", i) for i in range(n_ticks)]
        seq, pi = [], 0
        for e in fake.log:
            if e[0] == "push":
                seq.append(("p", pi)); pi += 1
            elif e[0] == "fuse":
                seq.append(("f", pi - 1))
        fused = [i for tag, i in seq if tag == "f"]
        ok &= fused == [k for k in range(n_ticks) if k % world == rank] and drv.last_mine == (fused[-1] if fused else -1)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_ticks", [4, 7])
def test_tick_interleaved_driver_gloo_world2(n_ticks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tick_worker, args=(r, 2, 29620 + n_ticks, n_ticks, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


# ---- band driver (dist.ShardedEsvo) on CPU: the phase / exchange protocol with a host-memory stand-in ---------------
class _FakeBand:
    """host-memory stand-in for lib.Esvo's three-phase tick: hands out (send, receive, block bytes) as esvo_shard_exchange does"""

    def __init__(self, rank, world):
        self.rank, self.world, self.phase_log = rank, world, []
        self.send = self.recv = None

    def set_band(self, y0, y1, shard, n_shards, routing=None, ts_halo_rows=-1):
        self.band = (y0, y1, shard, n_shards)
        self.routing = routing

    def shard_phase(self, phase, t_ns=0, stamps=None, poses=None):
        self.phase_log.append(phase)
        if phase == 0:      # one byte per own slot (w = rank + k world), padded to whole words
            n = 21
            own = np.arange(n)[self.rank::self.world]
            b = np.zeros((-(-n // self.world) + 7) // 8 * 8, np.uint8)
            b[: len(own)] = 1 + 2 * (own % 2)
            self.send = b.view(np.int64)
        elif phase == 1:    # [count | own points], block length = the largest count among the ranks
            mine = np.arange(5)[self.rank::self.world]
            most = len(np.arange(5)[0::self.world])
            f = np.zeros(1 + 13 * most, np.int64)
            f[0] = len(mine)
            f[1: 1 + 13 * len(mine)] = ((mine[:, None] + 1) * 100 + np.arange(13)).reshape(-1)
            self.send = f
        else:
            self.send = None
        self.recv = None if self.send is None else np.zeros(self.world * len(self.send), np.int64)

    def shard_exchange(self):
        return (self.send.ctypes.data, self.recv.ctypes.data, self.send.nbytes) if self.send is not None else (0, 0, 0)


class _Rig:
    width, height = 8, 10


def _band_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fake = _FakeBand(rank, world)
        drv = edist.ShardedEsvo(None, _Rig(), rank, world, 0, dev=fake, device="cpu")
        ok = fake.band == (*edist.band_of(rank, world, 10), rank, world) and fake.routing == "y_rect"   # (rows unless the preset forbids it)
        snap = {}
        orig = fake.shard_phase

        def spy(phase, *a, **k):
            if fake.recv is not None:
                snap[phase - 1] = fake.recv.copy()     # what the previous phase's receive buffer held after the exchange
            orig(phase, *a, **k)
        fake.shard_phase = spy
        drv.tick(1, np.zeros(1, np.uint64), np.zeros((1, 16)))
        ok &= fake.phase_log == [0, 1, 2]
        blk = len(snap[0]) // world
        codes = np.zeros(21, np.uint8)
        for r in range(world):
            codes[r::world] = snap[0][r * blk: (r + 1) * blk].view(np.uint8)[: len(range(r, 21, world))]
        ok &= bool((codes == 1 + 2 * (np.arange(21) % 2)).all())                       # every rank's bytes, in place
        bw = len(snap[1]) // world
        frame = np.zeros((5, 13), np.int64)
        for r in range(world):
            c = int(snap[1][r * bw])
            frame[r::world] = snap[1][r * bw + 1: r * bw + 1 + 13 * c].reshape(c, 13)
        ok &= bool((frame == (np.arange(5)[:, None] + 1) * 100 + np.arange(13)).all())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_band_driver_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_band_worker, args=(r, 2, 29641, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_pick_routing_follows_what_the_library_supports():
    """dist.pick_routing("auto"): rows by floor(y_rect) unless the parameters are ones esvo_shard_set_routing refuses
    (per-pixel event queues, up-down stereo) -- then the broadcast switch; an explicit choice is passed through.  Denoising
    is routed by rows since round 6."""
    from esvo_amd import calib, dist as edist, params
    rig = calib.dataset_rig("dsec")
    p, _ = params.make_params(params.PRESETS["mapping_dsec"], rig)
    assert edist.pick_routing(p, "auto") == "y_rect" and edist.pick_routing(None, "auto") == "y_rect"
    assert edist.pick_routing(p, "broadcast") == "broadcast" and edist.pick_routing(p, "y_rect") == "y_rect"
    rig_r = calib.dataset_rig("rpg")
    pr, _ = params.make_params(params.PRESETS["mvstereo_rpg"], rig_r)       # the small DAVIS configs switch Denoising on
    assert bool(pr.denoising) and edist.pick_routing(pr, "auto") == "y_rect"
    pq, _ = params.make_params(params.PRESETS["mapping_dsec"], rig)
    pq.max_event_queue_len = 20
    assert edist.pick_routing(pq, "auto") == "broadcast"
