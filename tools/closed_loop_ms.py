"""usage (GPU box): python tools/closed_loop_ms.py   -- the closed-loop point of bench.py --extras alone: ms per cycle / tracking / mapping"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esvo_amd import closed_loop as cl  # noqa: E402

r = cl.run(n_ticks=15)
med = lambda v: float(np.median(np.asarray(v[3:])))  # noqa: E731
print("cycle %.3f ms  tracking %.3f ms  mapping %.3f ms  end error %.2f mm of %.1f mm  max rot err %.3f deg  points/cycle %d" % (
    med(r["cycle_ms"]), med(r["track_ms"]), med(r["map_ms"]), r["pos_err"][-1] * 1e3, r["gt_len"][-1] * 1e3, max(r["rot_err_deg"]),
    int(np.median(r["points"]))))
print("tracking ms per cycle:", " ".join("%.2f" % v for v in r["track_ms"]))
