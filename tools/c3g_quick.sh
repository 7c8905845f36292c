#!/bin/bash
# quick look at config C3G (bench lines only): tools/c3g_quick.sh <tag>
TAG=${1:-c3g}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
A="--steps 6 --warmup 2 --no-cpu-baseline --ess-batches 0 --exact-steps 0"
python bench.py $A 2>/dev/null | grep '^{' > "$OUT/C3_tracked.json"
for g in lattice3d random6 random8; do
  python bench.py --config C3G --graph $g $A 2>"$OUT/C3G_${g}_tracked.err" | grep '^{' > "$OUT/C3G_${g}_tracked.json"
  python bench.py --config C3G --graph $g --exact --steps 2 --warmup 1 --no-cpu-baseline 2>"$OUT/C3G_${g}_exact.err" | grep '^{' > "$OUT/C3G_${g}_exact.json"
done
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        j=json.load(open(f))
        print(os.path.basename(f), j["roofline"]["kernel"], "ms/step %.2f"%j["ms_per_step"], "value %.3e"%j["value"], "frac %.3f"%j["roofline"]["frac"], "acc %.3f"%j["acceptance"], "bad", j["unhealthy_chains"])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
