#!/usr/bin/env python
"""The tolerance operating point (esvo_map_set_refine_mode(ESVO_REFINE_FAST)) against the exact refinement on the sustained
headline workload, interleaved in one process: ms per tick, LM launch time.  usage: python tools/fast_mode_probe.py [ticks] [rounds]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from esvo_amd import lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 800
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
orig = lib.Esvo.__init__
for r in range(rounds):
    for fast in (False, True):
        def init(self, *a, _f=fast, **k):
            orig(self, *a, **k)
            self.set_refine_mode(_f)
        lib.Esvo.__init__ = init
        res = bench.sustained_point("dsec640x480", n, 0, None)
        print(f"{'FAST ' if fast else 'exact'} r{r}  {res['ms_per_tick']:.4f} ms/tick  {res['events_per_s'] / 1e6:.1f} M ev/s  lm {res['kernel_ms']['lm_refine']:.3f} "
              f"fuse {res['kernel_ms']['fuse']:.3f} reg {res['kernel_ms']['regularize']:.3f} bm {res['kernel_ms']['bm_match']:.3f}  resyncs {res['pipeline_resyncs']}")
lib.Esvo.__init__ = orig
