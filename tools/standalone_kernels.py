"""Stand-alone kernel durations at a bench workload: ticks with a synchronisation after each (nothing overlaps), to be run
under `rocprofv3 --kernel-trace --stats`.  usage: python tools/standalone_kernels.py [workload] [ticks]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from esvo_amd import lib
name = sys.argv[1] if len(sys.argv) > 1 else "dsec640x480"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rig, stream, p, ticks = bench.make_workload(name, n)
dev = lib.Esvo(p, rig)
dev.ts_push_events(0, stream.ev_left); dev.ts_push_events(1, stream.ev_right)
bench.run_single(dev, stream, ticks, 0, n, sync_each=True)
s = dev.stats()
print(name, "events", s.last_events_in, "matches", s.last_matches, "points", s.last_points, "window points", s.last_window_points)
