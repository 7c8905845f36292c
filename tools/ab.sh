#!/bin/bash
# A/B timing of engine builds inside ONE GPU session (the same binary runs 6-12 % apart between processes, so versions are
# compared by interleaved runs):  tools/ab.sh N variantA variantB ...   ("base" = the default library)
# BENCH_ARGS="--config C5 --steps 4 --warmup 1" selects another workload
# prints ms per step of bench.py's headline workload, N rounds, variants interleaved
N=${1:-3}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for k in $(seq 1 $N); do
  for v in "$@"; do
    if [ "$v" = base ]; then lib=""; else lib="$ROOT/zigzagboomerang.jl_amd/lib/libpdmp_mi355.$v.so"; fi
    ms=$(PDMP_MI355_LIB=$lib python "$ROOT/bench.py" ${BENCH_ARGS:---steps 8 --warmup 2} --no-cpu-baseline --ess-batches 0 2>/dev/null | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo -n "$v=$ms  "
  done; echo
done
