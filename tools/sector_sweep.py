#!/usr/bin/env python
"""Random-sector rate against footprint: 4096 wavefronts, records per chain d in a sweep, so the probe's working set sits in
L2 (<= 32 MiB), in the Infinity Cache (<= 256 MiB) or in HBM.  usage: tools/sector_sweep.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
nch, rounds = 4096, 1000
for d in (64, 256, 512, 1024, 2048, 4096, 16384):
    out = {"nchains": nch, "records_per_chain": d, "footprint_MiB": nch * d * 64 / 2**20}
    for write, name, mult in ((0, "read", 1), (1, "read+write", 2), (2, "read64", 2), (3, "read+write64", 4)):
        ms = pkg._lib.sector_probe(nch, d, rounds, write)
        sectors = nch * 64 * 4 * rounds * mult
        out[name] = {"ms": round(ms, 3), "Gsectors_per_s": round(sectors / (ms * 1e-3) / 1e9, 2), "GB_per_s": round(sectors * 32 / (ms * 1e-3) / 1e9, 1)}
    print(json.dumps(out), flush=True)
