#!/bin/bash
# One GPU session of the inner loop of a kernel change: the tracked kernels' parity tests, then the strong-scaling widths with chain 0's phase profile.
#   tools/gpu_quick.sh <tag> [widths]     (run through gpurun; output under gpurun_out/<tag>/)
TAG=$1; W=${2:-4096,2048,1024,512,1}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_track_parity.py tests/test_gpu_trackp_two_waves.py tests/test_gpu_track_generic.py tests/test_gpu_big_lattice.py -x -q -m gpu > gpurun_out/$TAG/tests.log 2>&1
tail -2 gpurun_out/$TAG/tests.log
timeout 900 python tools/strong_proxy.py --phase --evals tracked --widths $W > gpurun_out/$TAG/strong_phase.jsonl 2> gpurun_out/$TAG/strong_phase.err
python - <<P
import json
for l in open("gpurun_out/$TAG/strong_phase.jsonl"):
    j = json.loads(l); p = j.get("phase", {})
    print(j["chains"], j["kernel"], "ms %.2f" % j["ms_per_step"], "ev/s %.4g" % j["events_per_s"], "bad", j["bad"], p.get("cycles_per_iter"), "prop/it %.1f" % p.get("proposals_per_iter", 0))
P
