#!/bin/bash
for k in $(seq 1 ${1:-3}); do PDMP_DEBUG_PTRS=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/tmp/err.txt | tail -1 | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'], end=' ')"; grep PTRS /tmp/err.txt | tail -1; done
