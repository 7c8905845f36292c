#!/bin/bash
# Interleaved A/B of library variants at the strong-scaling widths inside ONE GPU session (boxes and processes differ by 6-14 %):
#   tools/ab_widths.sh N "varA varB .." [widths] [evals]      ("base" = the default library; a variant is lib/libpdmp_mi355.<name>.so)
N=${1:-2}; VARS=${2:-"ref base"}; W=${3:-4096,1024,512}; EV=${4:-tracked}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for k in $(seq 1 $N); do
  for v in $VARS; do
    if [ "$v" = base ]; then lib=""; else lib="$ROOT/zigzagboomerang.jl_amd/lib/libpdmp_mi355.$v.so"; fi
    PDMP_MI355_LIB=$lib python "$ROOT/tools/strong_proxy.py" --evals $EV --widths $W --steps 8 2>/dev/null | python -c "
import sys, json
print('$v', ' '.join('%d:%.2f' % (j['chains'], j['ms_per_step']) + ('!' if j['bad'] else '') for j in map(json.loads, sys.stdin)))"
  done
done
