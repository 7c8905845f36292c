#!/bin/bash
# Everything the round's profiles/ files are made from, in one GPU session:  tools/profile_round.sh <tag> [configs...]
#   per configuration: bench.py's JSON line, rocprofv3 --kernel-trace --stats of the same command, and the L2<->fabric request counters / WRITE_SIZE in
#   SEPARATE --pmc passes (never combined with other trace domains); for C3 also the SQ passes; once: the PMC calibration on the
#   sector probe (a kernel whose HBM bytes are known exactly).  Raw output under gpurun_out/<tag>/, summaries are made by
#   tools/profile_collect.py (run here, copied into profiles/ by the caller).
set -u
TAG=$1; shift
CONFIGS=${@:-C3 C3X C3G C3GX C3G_random6 C3GX_random6 C3G_random8 C3GX_random8 C2 C4 C4T C5}   # C3X = C3 on the bit-identical moving kernel (bench.py --exact); C4T = C4 with tracked bounds (--tracked); C3G* = graphs off the 2-d lattice
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for C in $CONFIGS; do
  ENVV="PDMP_NOENV=1"
  case $C in C3|C3X|C3L|C3R|C3_w2048|C3_w1024|C3_w512|C3X_w1024|C3X_w512) ST=8;; C3_g256) ST=4;; C3G|C3G_random6|C3G_random8) ST=6;; C3GX|C3GX_random6|C3GX_random8) ST=3;; C2) ST=8;; C4|C4T) ST=6;; C5) ST=4;; esac
  ARGS="--config $C --steps $ST --warmup 2 --no-cpu-baseline --ess-batches 0 --exact-steps 0 --no-strong-proxy --no-pipeline --late-T 0"
  if [ "$C" = C3X ]; then ARGS="--config C3 --exact --steps $ST --warmup 2 --no-cpu-baseline --ess-batches 0 --exact-steps 0 --no-strong-proxy --no-pipeline"; fi
  if [ "$C" = C4T ]; then ARGS="--config C4 --tracked --steps $ST --warmup 2 --no-cpu-baseline --ess-batches 0 --exact-steps 0 --late-T 0"; fi
  case $C in
    C3_w2048|C3_w1024|C3_w512)   # a rank's share of the 4096-chain ensemble on 2 / 4 / 8 GPUs (the strong-scaling proxy), tracked
      ARGS="--config C3 --chains ${C#C3_w} --steps $ST --warmup 2 --no-cpu-baseline --ess-batches 0 --exact-steps 0 --no-strong-proxy --no-pipeline";;
    C3X_w1024|C3X_w512)          # ... on the bit-identical moving kernel
      ARGS="--config C3 --exact --chains ${C#C3X_w} --steps $ST --warmup 2 --no-cpu-baseline --ess-batches 0 --exact-steps 0 --no-strong-proxy --no-pipeline";;
    C3L)                         # round 6: the line layout of pdmp_trackl.hip (opt-in form)
      ENVV="PDMP_TRACK_LINES=1"; ARGS="--config C3 --steps $ST --warmup 2 --no-cpu-baseline --ess-batches 0 --exact-steps 0 --no-strong-proxy --no-pipeline";;
    C3R)                         # round 6: the flow's refresh clock on (the reference's arithmetic, 8-event kernel)
      ARGS="--config C3 --lambda-ref 1.0 --steps $ST --warmup 2 --no-cpu-baseline --ess-batches 0 --exact-steps 0 --no-strong-proxy --no-pipeline";;
    C3_g256)                     # d = 65536: the 256 x 256 lattice, 1024 chains (the same 8.6 GB of chain state)
      ARGS="--config C3 --grid 256 --chains 1024 --steps $ST --warmup 2 --no-cpu-baseline --ess-batches 0 --exact-steps 0 --no-strong-proxy --no-pipeline";;
    C3G) ARGS="--config C3G --steps $ST --warmup 2 --no-cpu-baseline";;
    C3GX) ARGS="--config C3G --exact --steps $ST --warmup 1 --no-cpu-baseline";;
    C3G_random6) ARGS="--config C3G --graph random6 --steps $ST --warmup 2 --no-cpu-baseline";;
    C3GX_random6) ARGS="--config C3G --graph random6 --exact --steps $ST --warmup 1 --no-cpu-baseline";;
    C3G_random8) ARGS="--config C3G --graph random8 --steps $ST --warmup 2 --no-cpu-baseline";;
    C3GX_random8) ARGS="--config C3G --graph random8 --exact --steps $ST --warmup 1 --no-cpu-baseline";;
  esac
  env $ENVV python $ROOT/bench.py $ARGS 2>/dev/null | grep '^{' > "$OUT/${C}_bench.json"
  timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/${C}_stats" -o st --output-format csv -- env $ENVV python $ROOT/bench.py $ARGS > "$OUT/${C}_stats.log" 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d "$OUT/${C}_pmc" -o fetch --output-format csv -- env $ENVV python $ROOT/bench.py $ARGS > "$OUT/${C}_fetch.log" 2>&1
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/${C}_pmc" -o write --output-format csv -- env $ENVV python $ROOT/bench.py $ARGS > "$OUT/${C}_write.log" 2>&1
  if true; then   # the SQ passes for every configuration (round 4: the `issue` object of bench.py comes from them)
    timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT/${C}_pmc" -o sq1 --output-format csv -- env $ENVV python $ROOT/bench.py $ARGS > "$OUT/${C}_sq1.log" 2>&1
    timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT -d "$OUT/${C}_pmc" -o sq2 --output-format csv -- env $ENVV python $ROOT/bench.py $ARGS > "$OUT/${C}_sq2.log" 2>&1
  fi
done
# calibration: the request-size counters against exactly known bytes in the event loop's own access patterns
$ROOT/tools/pmc_reqsize_cal.sh "$OUT/cal" > "$OUT/cal.jsonl" 2>"$OUT/cal.log"
python $ROOT/tools/profile_collect.py "$OUT" "$TAG"
ls "$OUT"
