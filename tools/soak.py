"""Long run against the oracle: every tick's DepthMap of the device (lazy two-stream pipeline, small event ring that wraps)
is compared with the canonical CPU oracle's.  usage: python tools/soak.py [workload] [events per tick] [ticks] [ring capacity] [resident: esvo_map_tick_resident instead of the four calls | sync: resident, and every tick's map read right away -- each tick runs alone, the latency path]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from esvo_amd import lib
from oracle import oracle

name = sys.argv[1] if len(sys.argv) > 1 else "upenn346x260"
n_ev = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ring = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 17
sync = len(sys.argv) > 5 and sys.argv[5] == "sync"
resident = sync or (len(sys.argv) > 5 and sys.argv[5] == "resident")
rig, stream, p, ticks = bench.make_workload(name, n, events_cap=n_ev)
p.event_ring_capacity = ring
dev = lib.Esvo(p, rig)
m = oracle.OracleMapper(p, rig)
m.set_mode(True, True)
m.set_threads(os.cpu_count() or 1)
ts = [oracle.OracleTS(rig.width, rig.height), oracle.OracleTS(rig.width, rig.height)]
done = [0, 0]
bad = 0
prev = None   # the device map is fetched one tick late (esvo_map_get_committed) so that two ticks stay in flight
for k, (t, stamps, poses, T) in enumerate(ticks):
    for cam, (ev, ns) in enumerate(((stream.ev_left, stream.ns_left), (stream.ev_right, stream.ns_right))):
        hi = int(np.searchsorted(ns, t, side="left"))
        dev.ts_push_events(cam, ev[done[cam]:hi])      # staged tick by tick: the ring wraps
        ts[cam].push(ev[done[cam]:hi])
        done[cam] = hi
    if resident:   # both cameras per launch; the scatter ranges wrap the ring
        dev.tick_resident(t, T, stamps, poses)
    else:
        dev.ts_render(0, t, download=False); dev.ts_render(1, t, download=False)
        dev.set_observation(t, None, None, T)
        dev.tick(t, stamps, poses)
    l = ts[0].render(t, map_x=rig.left.map_x, map_y=rig.left.map_y)
    r = ts[1].render(t, map_x=rig.right.map_x, map_y=rig.right.map_y)
    m.set_observation(t, l, r, T); m.set_poses(stamps, poses)
    staged = stream.ev_left[:done[0]]                 # the device sees what has been staged: lower_bound(t) is end() (Appendix A-3)
    idx = oracle.select_events(staged, t, p.bm_half_slice_thickness, p.process_event_num)
    m.tick(staged[idx])
    om = bench.map_sha1(m.get_map())
    if sync:   # the ROS node's pattern: tick, read the map, next tick
        gm = dev.get_map()
        ok = bench.map_sha1(gm) == om
        bad += not ok
        if not ok or k % 20 == 0:
            print(f"tick {k}: {'equal' if ok else 'DIFFERENT'} (map {len(gm)})", flush=True)
        prev = (t, om, len(m.get_map()))
        continue
    if prev is not None:
        gm, gt = dev.get_committed_map()
        ok = gt == prev[0] and bench.map_sha1(gm) == prev[1]
        if not ok and bad < 3:
            print('   stamp', gt, prev[0], 'sizes', len(gm), prev[2])
        bad += not ok
        if not ok or k % 20 == 0:
            print(f"tick {k - 1}: {'equal' if ok else 'DIFFERENT'} (map {len(gm)})", flush=True)
    prev = (t, om, len(m.get_map()))
gm = dev.get_map()
ok = bench.map_sha1(gm) == prev[1]
bad += not ok
print(f"{name}: {n} ticks of {n_ev} events, ring {ring}: {bad} ticks differ; events staged {done}, final map {len(gm)}")
sys.exit(1 if bad else 0)
