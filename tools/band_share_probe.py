#!/usr/bin/env python
"""bench.py's band_share block alone (benchlib/multigpu.py): one rank's compute per band-mode tick with G logical shards on ONE GPU.
    python tools/band_share_probe.py [workload] [G ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dsec640x480"
shards = tuple(int(a) for a in sys.argv[2:]) or (8, 16)
print(json.dumps(bench.band_share(name, 0, shards=shards), indent=1))
