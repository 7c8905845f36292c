#!/bin/bash
# The full-width C3 slice under explicit class patterns of its arrays (PDMP_PLACE_rec / _kp / _ev: pdmp_place.hip), several ensembles per pattern.
#   tools/probes/place_matrix.sh "rec kp ev" ["rec kp ev" ...]
run() { echo "== rec=$1 kp=$2 ev=$3"; PDMP_PLACE_rec=$1 PDMP_PLACE_kp=$2 PDMP_PLACE_ev=$3 timeout 300 python tools/mode_alloc.py --rounds ${ROUNDS:-4} --steps 3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['phase'][:5], j['ms'], j['slices'], j['Mevents'], j['placement'][:100])
    else: print(l.rstrip()[:160])"; }
for c in "$@"; do run $c; done
