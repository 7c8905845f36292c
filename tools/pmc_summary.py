#!/usr/bin/env python
"""Summarise the counter CSVs written by tools/pmc_passes.sh into the text kept under profiles/.

    python tools/pmc_summary.py gpurun_out/pmc_k "header line" > profiles/<name>_pmc.txt
"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    print("# " + (sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 --pmc passes (tools/pmc_passes.sh)"))
    print("# per-dispatch means per kernel; SQ_* in quad-cycles summed over waves; FETCH/WRITE_SIZE in KiB")
    for f in sorted(glob.glob(os.path.join(d, "pmc*_counter_collection.csv"))):
        tag = os.path.basename(f).split("_")[0]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:36], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print(f"{tag:6s} {k:36s} {c:24s} n={len(v)} mean={sum(v) / len(v):.5g}")


if __name__ == "__main__":
    main()
