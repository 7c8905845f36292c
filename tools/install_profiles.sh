#!/bin/bash
# Copies a profile_round session's summaries into profiles/ (tracked) and makes it the round's reference set:  tools/install_profiles.sh <new-tag> [<old-tag-to-drop>]
# (traffic.json is replaced; the docs still have to be told the new tag)
set -eu
NEW=$1; OLD=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
test -d "$ROOT/gpurun_out/$NEW/summary"
if [ -n "$OLD" ]; then for f in "$ROOT"/profiles/${OLD}_*; do case $f in *bench_default*) ;; *) git -C "$ROOT" rm -q --ignore-unmatch "$f" || rm -f "$f";; esac; done; fi
cp "$ROOT"/gpurun_out/$NEW/summary/${NEW}_* "$ROOT/profiles/"
cp "$ROOT/gpurun_out/$NEW/summary/traffic.json" "$ROOT/profiles/traffic.json"
[ -f "$ROOT/gpurun_out/${NEW}_bench.json" ] && cp "$ROOT/gpurun_out/${NEW}_bench.json" "$ROOT/profiles/${NEW}_bench_default.json"
python - "$ROOT" <<'PY'
import json, sys
sys.path.insert(0, sys.argv[1])
import bench
t = json.load(open(sys.argv[1] + "/profiles/traffic.json"))
print("traffic.json hash", t["source_hash"], "sources", bench.source_hash(), "OK" if t["source_hash"] == bench.source_hash() else "STALE")
PY
