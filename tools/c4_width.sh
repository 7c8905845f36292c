# C4 at several ensemble widths: what a width that is not a multiple of the resident capacity (13 chains per CU = 3328) costs, and the
# throughput of one round at partial occupancy.  Output: gpurun_out/c4_width.txt
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/c4_width.txt
for N in 1024 2048 3328 4096 6656 8192 9984; do
python bench.py --config C4 --chains $N --steps 6 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('chains $N', j['ms_per_step'], j['roofline']['kernel_ms_avg'], 'per-chain us', 1e3*j['roofline']['kernel_ms_avg']/$N, j['value'], j['roofline']['frac'])" >> gpurun_out/c4_width.txt
done
cat gpurun_out/c4_width.txt
