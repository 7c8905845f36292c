#!/usr/bin/env python
"""Does the memory system reward WIDE requests?  Random 32-byte sectors per lane against aligned 64/128-byte units and 96-byte runs read
by lane groups (one request per unit), 4096 wavefronts over 4 GiB.  usage: tools/sector_wide.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
nch, rounds, d = 4096, 1000, 16384
modes = [(0, "32 B per lane, read", 64 * 4, 32), (1, "32 B per lane, read+write", 64 * 4, 32),
         (9, "64 B per 2 lanes, read", 32 * 4, 64), (10, "64 B per 2 lanes, read+write", 32 * 4, 64),
         (7, "128 B per 4 lanes, read", 16 * 4, 128), (8, "128 B per 4 lanes, read+write", 16 * 4, 128),
         (11, "96 B run per 8 lanes, read", 8 * 4, 96), (12, "96 B run per 8 lanes, read+write", 8 * 4, 96)]
for mode, name, units_per_wave_round, nbytes in modes:
    ms = pkg._lib.sector_probe(nch, d, rounds, mode)
    units = nch * rounds * units_per_wave_round
    print(json.dumps({"mode": name, "ms": round(ms, 3), "Gunits_per_s": round(units / (ms * 1e-3) / 1e9, 2),
                      "GB_per_s": round(units * nbytes / (ms * 1e-3) / 1e9, 1)}), flush=True)
