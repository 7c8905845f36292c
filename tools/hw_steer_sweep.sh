#!/bin/bash
# steering of the two-wave form's selection threshold ("gain,target,prefetch distance"): one line per setting
for st in "0.3,48,2" "0.3,40,2" "0.3,52,2" "0.5,48,2" "0.15,48,2" "0.3,48,1" "0.3,48,4"; do
  echo -n "steer $st: "
  PDMP_HELPER_STEER=$st timeout 200 python tools/strong_proxy.py --evals tracked --widths ${1:-1024,512} --steps 6 2>/dev/null | python -c "
import sys,json
print(' '.join('%d:%.2f' % (json.loads(l)['chains'], json.loads(l)['ms_per_step']) for l in sys.stdin))"
done
