#!/usr/bin/env python
"""bench.py's tick_share block alone (benchlib/multigpu.py): rank 0 of a world-N tick-interleaved run on ONE GPU, the other
ranks' frames pre-recorded, the all-gather emulated by device copies on the exchange stream.
    python tools/tick_share_probe.py [workload] [world] [rounds] [one_gpu_ms_per_tick]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dsec640x480"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 20
one = float(sys.argv[4]) if len(sys.argv) > 4 else None
print(json.dumps(bench.tick_share(name, 0, world=world, rounds=rounds, one_gpu_ms_per_tick=one), indent=1))
