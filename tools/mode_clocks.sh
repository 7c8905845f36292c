#!/bin/bash
# What the GPU's clocks, power and temperatures are WHILE the full-width C3 slice runs (DESIGN.md 5 "The two timing modes"): samples rocm-smi
# every ~0.25 s beside a bench run and prints the samples taken under load next to the run's kernel time.    tools/mode_clocks.sh [steps] [out-dir]
STEPS=${1:-120}; OUT=${2:-gpurun_out/mode_clocks}
ROOT=$(cd "$(dirname "$0")/.." && pwd); mkdir -p "$OUT"
rocm-smi --showmaxpower --showperflevel --showclkfrq 2>/dev/null | grep -v "^=\|^$" > "$OUT/static.txt"
( while true; do date +%s.%N; rocm-smi --showclocks --showpower --showtemp --showuse --showmemuse --json 2>/dev/null; echo; sleep 0.25; done ) > "$OUT/smi.log" &
SAMPLER=$!
sleep 2
python "$ROOT/tools/strong_proxy.py" --evals tracked --widths 4096 --steps "$STEPS" > "$OUT/bench.json" 2> "$OUT/bench.err"
sleep 1
kill $SAMPLER 2>/dev/null; wait $SAMPLER 2>/dev/null
python - "$OUT" <<'PY'
import json, sys, collections
out = sys.argv[1]
samples = []
for ln in open(out + "/smi.log"):
    ln = ln.strip()
    if ln.startswith("{"):
        try:
            samples.append(json.loads(ln))
        except Exception:
            pass
b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("full-width slice: %.2f ms (fast < 41, slow > 43.5)" % b["ms_per_step"])
busy = [s for s in samples if str(s.get("card0", {}).get("GPU use (%)", "0")).strip() not in ("0", "")]
print(len(samples), "samples,", len(busy), "with the GPU in use")
samples = busy or samples
cols = collections.defaultdict(list)
for s in samples:
    c = s.get("card0", {})
    for k, v in c.items():
        cols[k].append(v)
for k, v in cols.items():
    cnt = collections.Counter(v)
    print("%-60s %s" % (k, ", ".join("%s x%d" % kv for kv in cnt.most_common(6))))
PY
