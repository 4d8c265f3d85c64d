"""How much issue slack is left beside one handle's tick pipeline?  Two independent handles on one GPU, ticks interleaved
from one host thread, against one handle alone.  usage: python tools/two_handles.py [workload] [ticks]"""
import os, sys, time
os.environ.setdefault("ESVO_DEV_SWITCHES", "1")   # the library reads its A/B switches only with this set
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from esvo_amd import lib
name = sys.argv[1] if len(sys.argv) > 1 else "dsec640x480"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W = 4
rig, stream, p, ticks = bench.make_workload(name, n + W)
def mk():
    d = lib.Esvo(p, rig)
    d.ts_push_events(0, stream.ev_left); d.ts_push_events(1, stream.ev_right)
    return d
def one(d, k):
    t, stamps, poses, T = ticks[k]
    d.ts_render(0, t, download=False); d.ts_render(1, t, download=False)
    d.set_observation(t, None, None, T); d.tick(t, stamps, poses)
for rep in range(2):
    a = mk()
    for k in range(W): one(a, k)
    a.synchronize(); t0 = time.perf_counter()
    for k in range(W, W + n): one(a, k)
    a.synchronize(); t1 = time.perf_counter() - t0
    a.close()
    a, b = mk(), mk()
    for k in range(W): one(a, k); one(b, k)
    a.synchronize(); b.synchronize(); t0 = time.perf_counter()
    for k in range(W, W + n): one(a, k); one(b, k)
    a.synchronize(); b.synchronize(); t2 = time.perf_counter() - t0
    a.close(); b.close()
    print(f"{name}: one handle {t1 / n * 1e3:.3f} ms/tick; two handles {t2 / n * 1e3:.3f} ms per tick PAIR = {t2 / (2 * n) * 1e3:.3f} ms/tick ({2 * t1 / t2:.2f}x)")
