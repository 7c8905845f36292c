#!/bin/bash
# Is the per-process spread of the headline kernel's duration a clock effect?  N profiled runs: kernel duration next to GRBM_GUI_ACTIVE
# (cycles, summed over the 8 XCDs; /8 / duration = effective clock), or the counters of PMC="...".   tools/clock_probe.sh <out> [N]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(realpath -m "$1"); N=${2:-6}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for k in $(seq 1 $N); do
  timeout 300 rocprofv3 --kernel-trace --pmc ${PMC:-GRBM_GUI_ACTIVE} -d $OUT -o run$k --output-format csv -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ess-batches 0 > $OUT/run$k.log 2>&1
done
python - <<PY
import csv,glob
for k in range(1,$N+1):
    dur={}
    for f in glob.glob("$OUT/**/run%d_kernel_trace.csv"%k,recursive=True):
        for r in csv.DictReader(open(f)):
            if "zz_local_track" in r["Kernel_Name"]: dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-6
    cyc={}
    for f in glob.glob("$OUT/**/run%d_counter_collection.csv"%k,recursive=True):
        for r in csv.DictReader(open(f)):
            if "zz_local_track" in r["Kernel_Name"]: cyc.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
    ids=sorted(set(dur)&set(cyc),key=int)[-2:]
    print("run",k," | ".join("%.2fms "%dur[i]+" ".join("%s=%.4g"%(n,v) for n,v in sorted(cyc[i].items())) for i in ids))
PY
