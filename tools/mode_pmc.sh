#!/bin/bash
# Hardware counters of fast and slow full-width slices side by side: tools/mode_alloc.py keeps several ensembles alive in one process (both timing modes
# occur among them); each pass collects one counter group per dispatch with the kernel trace beside it.    tools/mode_pmc.sh [out-dir]
OUT=${1:-gpurun_out/mode_pmc}; ROOT=$(cd "$(dirname "$0")/.." && pwd); mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift
  PDMP_VMM_CHUNK_MB=${CHUNK:-0} timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$n" -o p --output-format csv -- python $ROOT/tools/mode_alloc.py --rounds ${ROUNDS:-6} --steps 2 > "$OUT/$n.log" 2>&1
  python - "$OUT/$n" "$@" <<'PY'
import csv, glob, sys, collections
d, names = sys.argv[1], sys.argv[2:]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True); cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not kt or not cc:
    print("no output in", d); sys.exit(0)
dur = {}
for r in csv.DictReader(open(kt[0])):
    if "trackp" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
val = collections.defaultdict(dict)
for r in csv.DictReader(open(cc[0])):
    if r["Dispatch_Id"] in dur:
        val[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
print("%8s " % "ms" + " ".join("%28s" % n[-28:] for n in names))
for k in sorted(dur, key=int):
    print("%8.2f " % dur[k] + " ".join("%28.4g" % val[k].get(n, float("nan")) for n in names))
PY
}
pass utcl1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
pass level TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum
pass stall TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
pass tag TCC_TAG_STALL_sum TCC_IB_STALL_sum TCP_PENDING_STALL_CYCLES_sum GRBM_UTCL2_BUSY
pass dest TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_IO_32B_sum TCC_EA0_WRREQ_DRAM_sum
