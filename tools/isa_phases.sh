#!/bin/bash
# The assembly of one source file with the phase boundaries of its event loops as comments (-DPDMP_PHASE_MARKS: "; WPHASE k" / "; LPHASE k"), one
# .s file per kernel under <outdir>, and per kernel the loads, stores, vmcnt waits and phase marks in program order (what an iteration waits for):
#   tools/isa_phases.sh pdmp_trackp.hip [outdir]        (hipcc only: runs without a GPU)
SRC=$1; OUT=${2:-/tmp/isa_$(basename $SRC .hip)}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $OUT && cd $OUT || exit 1
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -DPDMP_PHASE_MARKS -I$ROOT/include --cuda-device-only -S $ROOT/zigzagboomerang.jl_amd/csrc/$SRC -o all.s || exit 1
for K in $(grep -o "^_Z[A-Za-z0-9_]*:" all.s | tr -d ':' | sort -u); do
  L=$(grep -n "^$K:" all.s | head -1 | cut -d: -f1)
  awk -v a=$L 'NR>=a{print} /s_endpgm/&&NR>a{exit}' all.s > $K.s
  echo "== $K ($(wc -l < $K.s) lines; $(grep -A14 "\.name: *$K" all.s | grep "\.vgpr_count\|sgpr_spill_count" | tr -s ' ' | tr '\n' ' '))"
  grep -n "vmcnt\|global_load\|global_store\|global_atomic\|Loop Header: Depth=1\|PHASE" $K.s | awk -F: '{printf "%s:%s; ", $1, $2}' | sed 's/\s\+/ /g; s/global_load_dword/GL/g; s/global_store_dword/GS/g; s/s_waitcnt //g'
  echo
done
