#!/bin/bash
# Secondary configurations of BASELINE.json (C2, C4, C5) on one GPU; one JSON line each.  The headline (C3) is bench.py.
# usage: tools/bench_configs.sh > profiles/<round>_configs.jsonl
set -u
cd "$(dirname "$0")/.."
python tools/bench_c2.py 256 100 | tail -1
python tools/bench_c4.py 8192 10 | tail -1
python tools/bench_c5.py 4096 2 | tail -1
