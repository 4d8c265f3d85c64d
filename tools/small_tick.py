"""One reference-faithful tick at a time (the mode the ROS node runs): ms per tick with a synchronisation after every tick.
usage: python tools/small_tick.py [upenn346x260|dsec640x480] [events] [ticks] [r01: round 1's scene | pipelined: no synchronisation between ticks]   (under rocprofv3 --kernel-trace for a timeline)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from esvo_amd import lib
name = sys.argv[1] if len(sys.argv) > 1 else "upenn346x260"
n_ev = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
r01 = len(sys.argv) > 4 and sys.argv[4] == "r01"
sync = not (len(sys.argv) > 4 and sys.argv[4] == "pipelined")
rig, stream, p, ticks = bench.make_workload(name, n + 6, events_cap=n_ev, r01_scene=r01)
dev = lib.Esvo(p, rig)
dev.ts_push_events(0, stream.ev_left); dev.ts_push_events(1, stream.ev_right)
bench.run_single(dev, stream, ticks, 0, 6, sync_each=True)
t0 = time.perf_counter()
bench.run_single(dev, stream, ticks, 6, n + 6, sync_each=sync)
dev.synchronize()
dt = time.perf_counter() - t0
s = dev.stats()
print(f"{name}{' (round-1 scene)' if r01 else ''} {n_ev} events/tick: {dt / n * 1e3:.3f} ms per tick ({'synchronised' if sync else 'pipelined'}), {s.last_matches} matches, {s.last_points} points, map {s.last_map_size}")
