#!/bin/bash
# Does the full-width slice's timing mode depend on the shader clock?  Measures the slice under the default governor, then with the maximum sclk capped
# (performance-determinism mode), then restores the defaults.    tools/mode_sclk.sh "2300 2100 1900 1700" [steps]
CAPS=${1:-"2200 2000 1800"}; STEPS=${2:-40}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
one() { python "$ROOT/tools/strong_proxy.py" --evals tracked --widths ${W:-4096} --steps "$STEPS" 2>/dev/null | python -c "
import sys, json
print('$1', ' '.join('%d:%.2f ms' % (j['chains'], j['ms_per_step']) for j in map(json.loads, sys.stdin)), flush=True)"; }
one default
for c in $CAPS; do
  timeout 20 rocm-smi --setperfdeterminism $c 2>&1 | grep -i "success\|error\|fail\|not" | head -2
  one "cap$c"
done
timeout 20 rocm-smi --resetperfdeterminism 2>&1 | grep -i "success\|error\|fail\|not" | head -2
one default_again
