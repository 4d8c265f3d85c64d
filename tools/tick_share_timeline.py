import ctypes, os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ESVO_DEV_SWITCHES"] = "1"
os.environ["ESVO_TIMELINE"] = "1"
import bench
from esvo_amd import lib
orig_close = lib.Esvo.close
def close(self):
    if getattr(self, "_cb", None) is not None:   # the comm handle
        rows = np.zeros((2000, 12), np.float32)
        nr = ctypes.c_int()
        self.lib.esvo_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        rc = self.lib.esvo_debug_timeline(self.h, rows.ctypes.data, 2000, ctypes.byref(nr))
        rows = rows[: nr.value]
        names = ["T0", "BM0", "BM1", "S1", "LM0", "LM1", "S2", "CNT", "FU0", "FU1", "CL1", "RG1"]
        i0 = 10
        t0 = rows[i0][0]
        for r in rows[i0: i0 + 6]:
            print("   " + "  ".join(f"{nm} {x - t0:7.3f}" for nm, x in zip(names, r)))
    orig_close(self)
lib.Esvo.close = close
r = bench.tick_share("dsec640x480", 0, world=8, rounds=20, one_gpu_ms_per_tick=1.258)
print(r["round_ms_pipelined"])
