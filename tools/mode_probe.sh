#!/bin/bash
# What differs between the boxes / processes whose timings of the SAME binary are 6-13 % apart (profiles/HISTORY.md "the two timing modes")?
# One record per call: partition modes, clocks, power cap, firmware, NUMA placement, and the two headline timings.  tools/mode_probe.sh <tag>
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/mode_probe
mkdir -p $OUT
F=$OUT/${1:-probe}.txt
{
  echo "== $(date -u +%FT%TZ) host $(hostname) kernel $(uname -r)"
  rocm-smi --showcomputepartition --showmemorypartition --showperflevel --showclocks --showpower --showmaxpower --showvbios --showserial --showuniqueid --showbus --showtopo 2>&1 | grep -v "^=\|^$" | head -60
  echo "-- rocminfo (gfx950 agent)"
  rocminfo 2>/dev/null | awk '/Name:.*gfx950/{f=1} f&&/Compute Unit|Max Clock|Wavefront|Pool|Size:|Cacheline|L2|L3|Memory Properties/{print} /Agent [0-9]+/{if(f&&n++>0)exit}' | head -40
  echo "-- lscpu / numactl"
  lscpu | grep -i "model name\|socket\|numa\|^cpu(s)" | head -8
  cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo
  cat /sys/class/drm/card*/device/current_link_speed /sys/class/drm/card*/device/current_link_width 2>/dev/null | tr '\n' ' '; echo
  echo "-- timings"
  cd $ROOT
  for k in 1 2; do
    python bench.py --steps 6 --warmup 2 --no-cpu-baseline --ess-batches 0 --exact-steps 4 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('tracked ms %.2f  exact ms %.2f' % (j['ms_per_step'], j['exact']['ms_per_step']))"
  done
  rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|mclk\|fclk\|socclk\|power\|temp" | head -12
} > $F 2>&1
tail -4 $F
