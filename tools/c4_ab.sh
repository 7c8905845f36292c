cd /root/repo
for I in 0 1; do
PDMP_BENCH_C4_INTEGRALS=$I python bench.py --config C4 --steps 6 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('integrals $I', j['roofline']['kernel'], j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['value'], j['roofline']['frac'], j['acceptance'])"
done
