X="--no-cpu-baseline --ess-batches 0 --exact-steps 0 --no-strong-proxy --no-pipeline"
t() { local s=$(date +%s.%N); "$@" 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['config']['workload'][:40], j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel'], (j.get('late') or {}).get('roofline_frac'))"; echo "  wall $(echo "$(date +%s.%N) - $s" | bc)"; }
t python bench.py --config C2 --steps 4 --warmup 1 $X
t python bench.py --config C4 --steps 3 --warmup 1 --late-T 0 $X
t python bench.py --config C4 --steps 3 --warmup 1 $X
t python bench.py --config C5 --steps 3 --warmup 1 --c5-rows 10000 $X
t python bench.py --config C3G --graph random6 --steps 3 --warmup 1 $X
t python bench.py --config C3G --graph random6 --exact --steps 2 --warmup 1 $X
t python bench.py --config C3 --grid 256 --chains 1024 --steps 2 --warmup 1 $X
