#!/bin/bash
# Which TCC_EA0 request-size counter counts what on gfx950: the sector probe (known request sizes) under the read / write request counters.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for M in 0 2 7 1 3 8; do
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $OUT -o rd$M --output-format csv -- python $ROOT/tools/sector_probe_one.py $M 4096 200 > $OUT/rd$M.json 2>/dev/null
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum -d $OUT -o wr$M --output-format csv -- python $ROOT/tools/sector_probe_one.py $M 4096 200 > /dev/null 2>&1
done
python - <<PY
import csv,glob,collections,json,os
out="$OUT"
for M in (0,2,7,1,3,8):
    k=json.loads([l for l in open(f"{out}/rd{M}.json") if l.startswith("{")][-1])
    row={"mode":M,"known_read":k["read_bytes_per_launch"],"known_written":k["written_bytes_per_launch"]}
    for tag in ("rd","wr"):
        g=glob.glob(f"{out}/**/{tag}{M}_counter_collection.csv",recursive=True)
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(g[0])):
            if "sector_probe" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c,v in acc.items(): row[c]=sum(v)/len(v)
    print(json.dumps(row))
PY
