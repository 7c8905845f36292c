#!/usr/bin/env python
"""What one rejected proposal of the tracked kernels costs the memory system in two layouts (sector probe modes 13-16, pdmp_bps.hip):
record line + key-block line, both dirtied (as built) against record line read only + a 256-byte pair of (key, proposal time) lines with
one dirty line.  4096 wavefronts over 4 GiB.  usage: tools/sector_proposal.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
nch, rounds, d = 4096, 500, 16384
for mode, name in [(15, "record line + key line, reads only"), (13, "record line + key line, both written (as built)"),
                   (17, "record line + key line, both written with non-temporal stores"),
                   (18, "record line read only + ONE (key, time) line of a block of 8, 16 B of it written"),
                   (16, "record line + 256 B pair, reads only"), (14, "record line + 256 B pair, 16 B of the pair written")]:
    ms = pkg._lib.sector_probe(nch, d, rounds, mode)
    units = nch * rounds * 64
    print(json.dumps({"mode": name, "ms": round(ms, 3), "Gproposals_per_s": round(units / (ms * 1e-3) / 1e9, 2)}), flush=True)
