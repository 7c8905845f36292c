#!/bin/bash
# ms per step of bench.py (headline workload), N repetitions: tools/ms.sh [N]
for k in $(seq 1 ${1:-3}); do python bench.py --steps 8 --warmup 2 --no-cpu-baseline --ess-batches 0 2>/dev/null | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'], end=' ')"; done; echo
