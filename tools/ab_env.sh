#!/bin/bash
# interleaved A/B of environment settings on the headline bench: tools/ab_env.sh N "ENV1=.." "ENV2=.." ...   ("-" = no setting)
N=${1:-3}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for k in $(seq 1 $N); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    ms=$(env $e python "$ROOT/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --ess-batches 0 2>/dev/null | grep '^{' | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo -n "[$v]=$ms  "
  done; echo
done
