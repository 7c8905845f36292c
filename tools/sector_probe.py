#!/usr/bin/env python
"""Practical ceiling of the event loop's scattered memory traffic: random 32-byte sectors over nchains x 1 MiB of records
(one wavefront per chain, 4 independent loads per lane in flight).  usage: tools/sector_probe.py [nchains ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
d, rounds = 16384, 1000
for nch in [int(a) for a in sys.argv[1:]] or [4096]:
    out = {"nchains": nch}
    for write in (0, 1, 2, 3, 5, 6):
        ms = pkg._lib.sector_probe(nch, d, rounds, write)
        sectors = nch * 64 * 4 * rounds * {0: 1, 1: 2, 2: 2, 3: 4, 5: 3, 6: 3}[write]  # 32-byte sector operations
        out[{0: "read", 1: "read+write", 2: "read 64 B records", 3: "read+write 64 B records", 5: "read 3 hot halves at 64 B pitch", 6: "read 3 hot halves packed"}[write]] = {"ms": round(ms, 3), "sectors_per_s": sectors / (ms * 1e-3),
                                                   "GB_per_s": sectors * 32 / (ms * 1e-3) / 1e9}
    print(json.dumps(out))
