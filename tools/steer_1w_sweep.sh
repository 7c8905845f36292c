#!/bin/bash
# target steering on the one-wave forms (PDMP_HELPER_STEER=gain,target,ahead reaches them too): d = 65536 at 1024 chains, d = 16384 at 2048 / 4096
for st in "" "0.3,36,1" "0.3,40,1" "0.3,44,1" "0.3,48,1" "0.3,52,1"; do
  echo -n "steer [$st]: "
  PDMP_HELPER_STEER=$st timeout 300 python tools/strong_proxy.py --evals tracked --grid 256 --widths 1024 --steps 4 2>/dev/null | python -c "
import sys,json
print(' '.join('g256/%d:%.2f' % (json.loads(l)['chains'], json.loads(l)['ms_per_step']) for l in sys.stdin), end='  ')"
  PDMP_HELPER_STEER=$st timeout 300 python tools/strong_proxy.py --evals tracked --widths 4096,2048 --steps 6 2>/dev/null | python -c "
import sys,json
print(' '.join('%d:%.2f' % (json.loads(l)['chains'], json.loads(l)['ms_per_step']) for l in sys.stdin))"
done
