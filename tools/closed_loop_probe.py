"""Print the closed-loop run tick by tick: python tools/closed_loop_probe.py [speed] [re-reference period]."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from esvo_amd import closed_loop

speed = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
reref = int(sys.argv[2]) if len(sys.argv) > 2 else 10**9
r = closed_loop.run(speed=speed, reref=reref, verbose=True)
print({k: v for k, v in r.items() if not isinstance(v, list)})
