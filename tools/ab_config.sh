#!/bin/bash
# Interleaved A/B of library variants on one bench configuration inside ONE GPU session:  tools/ab_config.sh N "ref base" "<bench.py arguments>"
N=${1:-3}; VARS=${2:-"ref base"}; ARGS=${3:-"--config C4 --steps 6 --warmup 2 --late-T 0"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for k in $(seq 1 $N); do
  for v in $VARS; do
    if [ "$v" = base ]; then lib=""; else lib="$ROOT/zigzagboomerang.jl_amd/lib/libpdmp_mi355.$v.so"; fi
    PDMP_MI355_LIB=$lib python "$ROOT/bench.py" $ARGS --no-cpu-baseline --ess-batches 0 --exact-steps 0 --no-strong-proxy --no-pipeline 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$v', '%.2f ms' % j['ms_per_step'], 'frac %.3f' % j['roofline']['frac'], j['roofline']['kernel'], 'bad', j.get('unhealthy_chains'))"
  done
done
