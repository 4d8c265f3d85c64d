"""Multi-GPU operation of the mapper, one process per GPU, collectives through torch.distributed
(backend "nccl" = RCCL over xGMI) on device memory the C-ABI hands out.  Two ways to split the work:

  TickShardedEsvo  ticks dealt round-robin to the ranks, one all-gather of the round's frames (throughput scaling)
  ShardedEsvo      ONE tick split over the ranks -- per-event work by slot, per-cell work by image row band -- with
                   the two all-gathers of esvo_shard_exchange (latency scaling of a single tick; see below)

ShardedEsvo:  PyTorch is plumbing here: device-pointer views + collectives; every
kernel is in libesvo_hip.so.

Per tick (see esvo_shard_tick_phase in include/esvo_hip.h):
    phase 0  BM + LM + culling of the rank's events              -> all-gather: the (matched, kept) bits of the tick's slots
    phase 1  frame order from all bits, own kept points packed   -> all-gather: [count | points], block = largest kept count
    phase 2  points to their frame positions, window policy, fusion + clean + regularisation of the row band (halo rows
             recomputed locally)
No zero padding travels except the imbalance between the ranks' kept counts.

Which events are a rank's is the ROUTING (esvo_shard_set_routing):
    "y_rect"     SURVEY 8(e): a rank keeps the events of its image rows at ingest (every rank is handed the whole packets, as
                 eight subscribers of a topic would be), renders the Time Surfaces of its band + halo only and matches the
                 events whose floor(y_rect) falls into the band;
    "broadcast"  the A/B switch: every rank stages everything and renders full Time Surfaces, per-event work dealt by slot.
"""
import numpy as np

from . import lib
from .abi import DEPTH_POINT_DTYPE


class _DevView:
    """Zero-copy view of library-owned device memory through the CUDA array interface."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, n, typestr, device="cuda"):
    """zero-copy int/float view of `n` elements at device address `ptr` (host address when device == "cpu": tests)"""
    import torch
    if device == "cpu":
        import ctypes
        dt = np.dtype(typestr)
        buf = (ctypes.c_char * (int(n) * dt.itemsize)).from_address(int(ptr))
        return torch.from_numpy(np.frombuffer(buf, dtype=dt, count=int(n)))
    return torch.as_tensor(_DevView(ptr, n, typestr), device="cuda")


def band_of(rank, world, height):
    rows = -(-height // world)
    return min(rank * rows, height), min((rank + 1) * rows, height)


class LocalTransport:
    """The all-gather between `world` ranks that are handles of ONE process on ONE GPU, driven from `world` threads (tests, the
    closed loop on logical bands): a rendezvous + device copies.  Blocking (the device is drained on both sides), so nothing of
    the library's overlap is exercised -- only that the calls it makes, in the order it makes them, give the right bits."""

    def __init__(self, world, timeout=120):
        import threading
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()   # in the constructing thread: torch's lazy initialisation from a callback thread finds no device
        self.world = world
        self.bar = threading.Barrier(world, timeout=timeout)
        self.slots = [None] * world

    def all_gather(self, rank, d_send, d_recv, nbytes):
        import torch
        torch.cuda.synchronize()
        self.slots[rank] = d_send
        self.bar.wait()
        for r in range(self.world):
            src = device_tensor(self.slots[r], nbytes // 8, "<i8")
            device_tensor(d_recv + r * nbytes, nbytes // 8, "<i8").copy_(src)
        torch.cuda.synchronize()
        self.bar.wait()
        return 0

    def abort(self):
        self.bar.abort()


class HaloViolation(lib.EsvoError):
    """Routed band mode: a refinement of an earlier tick read outside the Time-Surface rows some rank renders; every rank
    refuses its ticks from here on (ESVO_ERR_HALO, the same tick on all of them: the count travels with exchange 2).  The map of
    the tick BEFORE this call is not to be trusted.  Recover with runner.restart(routing="broadcast") (exact whatever the
    motion) or runner.restart(ts_halo_rows=more) and re-stage the events the next ticks need."""


def _band_tick_guard(fn):
    def wrapped(self, *a, **kw):
        try:
            return fn(self, *a, **kw)
        except lib.EsvoError as e:
            if getattr(e, "code", None) == lib.ERR_HALO:
                raise HaloViolation(str(e), code=lib.ERR_HALO) from e
            raise
    return wrapped


def pick_routing(params, routing="auto"):
    """"auto": rows where the library supports it (esvo_shard_set_routing refuses per-pixel event queues and up-down stereo
    with ESVO_ERR_UNSUPPORTED), the broadcast switch otherwise"""
    if routing != "auto":
        return routing
    if params is None:
        return "y_rect"
    unsupported = int(getattr(params, "max_event_queue_len", 0)) > 0 or bool(getattr(params, "bm_updown", 0))
    return "broadcast" if unsupported else "y_rect"


# ---- the exchange primitive (backend agnostic: exercised with gloo on CPU in tests/test_dist.py) ----
def gather_blocks_(recv, send, world, group=None):
    """all-gather of one fixed-size block per rank into `recv` (rank-major): ncclAllGather under the nccl backend, a list
    gather under gloo (which has no all_gather_into_tensor on CPU tensors)"""
    import torch.distributed as dist
    if recv.device.type == "cpu":
        dist.all_gather(list(recv.view(world, -1).unbind(0)), send, group=group)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    return recv


class ShardedEsvo:
    """Same driving interface as lib.Esvo for bench.py / tests, one instance per rank."""

    counts_are_local = False

    def __init__(self, params, rig, rank, world, local_rank, group=None, dev=None, device="cuda", routing="auto", ts_halo_rows=-1):
        """dev / device: a stand-in for lib.Esvo on host memory (CPU tests of the phase / exchange logic under gloo)"""
        import torch
        self.rank, self.world, self.group = rank, world, group
        self.rig, self.params = rig, params
        self.W, self.H = rig.width, rig.height
        self.device = device
        self.routing = pick_routing(params, routing)
        if dev is None:
            self.dev = lib.Esvo(params, rig, device=local_rank)
            self.dev.set_stream(torch.cuda.current_stream().cuda_stream)
        else:
            self.dev = dev
        self.y0, self.y1 = band_of(rank, world, self.H)
        if self.y1 <= self.y0:
            raise lib.EsvoError(f"rank {rank} of {world} would own no image rows (H={self.H})")
        self.dev.set_band(self.y0, self.y1, rank, world, routing=self.routing, ts_halo_rows=ts_halo_rows)
        gather_blocks_(torch.zeros(1024 * world, dtype=torch.int64, device=device), torch.zeros(1024, dtype=torch.int64, device=device),
                       world, group)  # communicator set-up, untimed
        if device == "cuda":
            torch.cuda.synchronize()
    # every rank is handed every packet; what it stages and renders is the routing's business (inside the library)
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
        snd, rcv, nbytes = self.dev.shard_exchange()
        if nbytes and self.world > 1:
            gather_blocks_(device_tensor(rcv, self.world * nbytes // 8, "<i8", self.device),
                           device_tensor(snd, nbytes // 8, "<i8", self.device), self.world, self.group)

    @_band_tick_guard
    def tick(self, t_ns, stamps, poses):
        d = self.dev
        if d.shard_phase(0, t_ns, stamps, poses):   # Denoising on a routed handle: the mask's bits first, then phase 0 proper
            self._exchange()
            d.shard_phase(0)
        self._exchange()
        d.shard_phase(1)
        self._exchange()
        d.shard_phase(2)

    def restart(self, routing=None, ts_halo_rows=-1):
        """after a HaloViolation (collective: every rank got it at the same tick): esvo_reset, the band again with another
        routing or a wider halo; the caller re-stages events and warms the window up again"""
        self.routing = pick_routing(self.params, routing or self.routing)
        self.dev.reset()
        self.dev.set_band(self.y0, self.y1, self.rank, self.world, routing=self.routing, ts_halo_rows=ts_halo_rows)

    def get_band_map(self):
        """this rank's DepthMap elements; `seq` is the global creation id"""
        return self.dev.get_map()

    def get_map(self):
        """the full DepthMap on every rank, in the reference's list order"""
        import torch.distributed as dist
        parts = [None] * self.world
        dist.all_gather_object(parts, self.get_band_map(), group=self.group)
        return merge_band_maps(parts)


class TickShardedEsvo:
    """Tick-interleaved multi-GPU operation: rank r maps the ticks k with k % world == r completely (Time Surfaces at its
    tick, block matching, LM, fusion, clean, regularisation).  A tick depends on earlier ticks only through the frames in
    its fusion window -- the reference builds a new DepthFrame at every tick (esvo_Mapping.cpp:266-272) -- so the one
    exchange is an all-gather of the round's frames (104 B per kept point, padded to the round's largest frame), after
    which every rank pushes the `world` frames into its window in tick order and fuses when it reaches its own.  No
    kernel is split, so per-GPU efficiency is that of the single-GPU tick and throughput grows with the number of
    GPUs; the latency of one tick does not change.  Same driving interface as lib.Esvo: every rank sees every call.

    The DepthMap of tick k lives on rank k % world (get_map returns the newest one on every rank)."""

    counts_are_local = True   # stats() counts this rank's own ticks

    def __init__(self, params, rig, rank, world, local_rank, group=None, dev=None, device="cuda"):
        """dev / device: a stand-in for lib.Esvo on host memory (CPU tests of the round logic under gloo)"""
        import torch
        self.rank, self.world, self.group = rank, world, group
        self.rig, self.params = rig, params
        self.device = device
        if dev is None:
            self.dev = lib.Esvo(params, rig, device=local_rank)
            self.dev.set_stream(torch.cuda.current_stream().cuda_stream)
        else:
            self.dev = dev
        self.k = 0                 # index of the next tick
        self.round = []            # ticks of the current round: (t_ns, stamps, poses)
        self.mine = None           # (device pointer, points) of this rank's frame in the current round
        self.last_mine = -1        # index of the last tick this rank fused
        self.words = DEPTH_POINT_DTYPE.itemsize // 8
        self._cnt = torch.zeros(world, dtype=torch.int64, device=device)
        # Four alternating gather buffers, sized for 64 k points per tick up front and grown on demand.  The pushes of
        # round R copy out of the buffer on the library's BACK stream; they are enqueued there before this rank's fusion
        # of round R + 1, whose completion the front stage of round R + 3 waits for (two ticks in flight per handle) --
        # so a buffer may be overwritten by the gather of round R + 3 at the earliest, and a reuse distance of 4 is safe.
        # One untimed round trip of both collectives sets up the communicator's channels.
        prime = 65536 * self.words
        self._gather = [torch.empty(world * prime, dtype=torch.int64, device=device) for _ in range(4)]
        self._rounds = 0
        self._retired = []
        import torch.distributed as dist
        dist.all_reduce(self._cnt, op=dist.ReduceOp.SUM, group=group)
        self._all_gather(self._gather[0][: world * 1024], torch.zeros(1024, dtype=torch.int64, device=device))
        if device == "cuda":
            torch.cuda.synchronize()

    def _all_gather(self, recv, send):
        import torch.distributed as dist
        if dist.get_backend(self.group) == "gloo" and recv.device.type == "cpu":  # gloo has no all_gather_into_tensor on CPU
            dist.all_gather(list(recv.view(self.world, -1).unbind(0)), send, group=self.group)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)

    def _is_mine(self):
        return self.k % self.world == self.rank

    # every rank stages all events (the mapper walks back over them, the Time Surfaces need them all)
    def ts_push_events(self, cam, ev):
        self.dev.ts_push_events(cam, ev)

    def ts_render(self, cam, t_ns, download=True):
        if self._is_mine():  # the SAE only has to be current at this rank's own ticks
            return self.dev.ts_render(cam, t_ns, download)
        return None

    def set_observation(self, *a):
        if self._is_mine():
            self.dev.set_observation(*a)

    def tick(self, t_ns, stamps, poses):
        if self._is_mine():
            self.mine = self.dev.front(t_ns, stamps, poses)
        self.round.append((t_ns, np.ascontiguousarray(stamps), np.ascontiguousarray(poses, np.float64).reshape(-1, 16)))
        self.k += 1
        if len(self.round) == self.world:
            self._finish_round()

    def _finish_round(self):
        """all-gather the frames of the round, push them in tick order, fuse at the own tick"""
        import torch
        import torch.distributed as dist
        if not self.round:
            return
        n_mine = self.mine[1] if self.mine else 0
        self._cnt.zero_()
        self._cnt[self.rank] = n_mine
        dist.all_reduce(self._cnt, op=dist.ReduceOp.SUM, group=self.group)  # world counts (the host sizes the gather)
        counts = [int(c) for c in self._cnt.tolist()]
        stride = max(max(counts), 1) * self.words
        buf = self._gather[self._rounds % 4]
        if buf is None or buf.numel() < self.world * stride:
            self._retired.append(buf)  # copies on the back stream may still read it: never handed back to the allocator
            buf = torch.empty(self.world * stride, dtype=torch.int64, device=self.device)
            self._gather[self._rounds % 4] = buf
        send = buf.new_zeros(stride) if not n_mine else torch.empty(stride, dtype=torch.int64, device=self.device)
        if n_mine:
            send[: n_mine * self.words] = device_tensor(self.mine[0], n_mine * self.words, "<i8", self.device)
        recv = buf[: self.world * stride]
        self._all_gather(recv, send)
        base = recv.data_ptr()
        k0 = self.k - len(self.round)
        for j, (t_ns, stamps, poses) in enumerate(self.round):
            owner = (k0 + j) % self.world            # a round may start anywhere (partial rounds are flushed at syncs)
            self.dev.push_frame_device(base + owner * stride * 8, counts[owner], poses)
            if owner == self.rank:
                self.dev.fuse_async()
                self.last_mine = k0 + j
        self.round, self.mine = [], None
        self._rounds += 1

    def synchronize(self):
        self._finish_round()       # a partial last round (collective: every rank calls synchronize at the same points)
        self.dev.synchronize()

    def stats(self):
        return self.dev.stats()

    def get_map(self):
        """the newest DepthMap (of the last tick), on every rank"""
        import torch.distributed as dist
        self.synchronize()
        parts = [None] * self.world
        dist.all_gather_object(parts, (self.last_mine, self.dev.get_map() if self.last_mine >= 0 else None), group=self.group)
        return max(parts, key=lambda p: p[0])[1]


def merge_band_maps(parts):
    allp = np.concatenate([p for p in parts if len(p)]) if any(len(p) for p in parts) else np.zeros(0, DEPTH_POINT_DTYPE)
    order = np.argsort(allp["seq"], kind="stable")
    out = allp[order].copy()
    out["seq"] = np.arange(len(out), dtype=np.uint32)
    return out


# ---- the same two modes with the exchange INSIDE libesvo_hip.so (api_comm.hip: RCCL called from C) ------------------
# What a C++ ROS node would use: esvo_comm_init(ncclUniqueId ...) + esvo_comm_tick / esvo_comm_shard_tick.  These classes
# only hand the unique id around (torch.distributed is used for that one broadcast) and forward the driving calls.
def _native_comm_init(dev, rank, world, group=None):
    import torch.distributed as dist
    box = [lib.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    dev.comm_init(box[0], rank, world)


class NativeTickSharded:
    """TickShardedEsvo with the round logic and the ncclAllGather of frames in C (esvo_comm_tick): two rounds in flight, the
    exchange on a stream of its own beside the next round's front stage (api_comm.hip)."""

    counts_are_local = True

    def __init__(self, params, rig, rank, world, local_rank, group=None):
        self.rank, self.world = rank, world
        self.rig, self.params = rig, params
        self.dev = lib.Esvo(params, rig, device=local_rank)
        self._init_comm(group)
        self._T = None

    def _init_comm(self, group):
        _native_comm_init(self.dev, self.rank, self.world, group)

    def ts_push_events(self, cam, ev):
        self.dev.ts_push_events(cam, ev)

    def ts_render(self, cam, t_ns, download=True):
        if self.dev.comm_owns_next_tick():  # the SAE only has to be current at this rank's own ticks
            return self.dev.ts_render(cam, t_ns, download)
        return None

    def set_observation(self, t_ns, ts_left, ts_right, T_world_cam):
        self._T = T_world_cam  # esvo_comm_tick sets the observation on the owner (device-resident Time Surfaces)

    def tick(self, t_ns, stamps, poses):
        self.dev.comm_tick(t_ns, self._T, stamps, poses)

    def tick_resident(self, t_ns, T_world_cam, stamps, poses):
        """the four calls above in one: the owner renders both Time Surfaces inside esvo_comm_tick_resident"""
        self.dev.comm_tick_resident(t_ns, T_world_cam, stamps, poses)

    def synchronize(self):
        self.dev.comm_flush()
        self.dev.synchronize()

    def stats(self):
        return self.dev.stats()

    def comm_stats(self):
        return self.dev.comm_stats()

    def get_map(self):
        return self.dev.comm_newest_map()[0]


class CallbackTickSharded(NativeTickSharded):
    """The same C round logic (esvo_comm_tick: two rounds in flight, count-sized blocks) with the ONE collective it issues
    supplied by torch.distributed instead of the library's own RCCL binding (esvo_comm_init_callbacks): what bench.py falls
    back to when esvo_comm_init fails on a node.  The all-gather is enqueued on the stream the library names -- its exchange
    stream -- so it overlaps the next round's front stage exactly as the native one does."""

    def _init_comm(self, group):
        import torch
        import torch.distributed as dist
        self._group = group
        self._ext = {}

        def all_gather(d_send, d_recv, nbytes, stream):
            ext = self._ext.get(stream)
            if ext is None:
                ext = self._ext[stream] = torch.cuda.ExternalStream(int(stream))
            with torch.cuda.stream(ext):
                send = device_tensor(d_send, nbytes // 8, "<i8")
                recv = device_tensor(d_recv, self.world * nbytes // 8, "<i8")
                dist.all_gather_into_tensor(recv, send, group=self._group)
            return 0

        self.dev.comm_init_callbacks(self.rank, self.world, all_gather)


class NativeBandSharded:
    """ShardedEsvo with the two all-gathers of a tick and the all-gather of the DepthMap bands in C (ncclAllGather)."""

    counts_are_local = False

    def __init__(self, params, rig, rank, world, local_rank, group=None, routing="auto", ts_halo_rows=-1):
        self.rank, self.world = rank, world
        self.rig, self.params = rig, params
        self.routing = pick_routing(params, routing)
        self.dev = lib.Esvo(params, rig, device=local_rank)
        y0, y1 = band_of(rank, world, rig.height)
        if y1 <= y0:
            raise lib.EsvoError(f"rank {rank} of {world} would own no image rows (H={rig.height})")
        self.y0, self.y1 = y0, y1
        self.dev.set_band(y0, y1, rank, world, routing=self.routing, ts_halo_rows=ts_halo_rows)
        _native_comm_init(self.dev, rank, world, group)

    def ts_push_events(self, cam, ev):
        self.dev.ts_push_events(cam, ev)

    def ts_render(self, cam, t_ns, download=True):
        return self.dev.ts_render(cam, t_ns, download)

    def set_observation(self, *a):
        self.dev.set_observation(*a)

    @_band_tick_guard
    def tick(self, t_ns, stamps, poses):
        self.dev.comm_shard_tick(t_ns, stamps, poses)

    def restart(self, routing=None, ts_halo_rows=-1):
        """see ShardedEsvo.restart (the communicator stays)"""
        self.routing = pick_routing(self.params, routing or self.routing)
        self.dev.reset()
        self.dev.set_band(self.y0, self.y1, self.rank, self.world, routing=self.routing, ts_halo_rows=ts_halo_rows)

    def synchronize(self):
        self.dev.synchronize()

    def stats(self):
        return self.dev.stats()

    def get_map(self):
        return self.dev.comm_gather_map()
