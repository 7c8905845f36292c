run() { echo "== rec=$1 kp=$2 ev=$3"; PDMP_PLACE_rec=$1 PDMP_PLACE_kp=$2 PDMP_PLACE_ev=$3 timeout 300 python tools/mode_alloc.py --rounds 3 --steps 3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['phase'][:5], j['ms'], j['placement'][:200])"; }
run 0 0 0
run 0 1 2
run 0 1 0
run 0 1 1
run 01 2 2
run 012 0 012
run 001122 2 0
