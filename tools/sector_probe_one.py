#!/usr/bin/env python
"""One mode of the sector probe, for PMC calibration on a KNOWN byte count (tools/profile_round.sh):
   tools/sector_probe_one.py <mode> [nchains] [rounds]   prints the bytes the probe kernel reads / writes per launch"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
mode = int(sys.argv[1])
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
ms = pkg._lib.sector_probe(nch, 16384, rounds, mode, iters=3)
per_round = {0: (32, 0), 1: (32, 32), 2: (64, 0), 3: (64, 64), 7: (128 / 4, 0), 8: (128 / 4, 128 / 4)}[mode]  # bytes per lane-load
n = nch * 64 * 4 * rounds
print(json.dumps({"mode": mode, "ms": ms, "read_bytes_per_launch": n * per_round[0], "written_bytes_per_launch": n * per_round[1],
                  "launches": 4}))
