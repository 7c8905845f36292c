#!/bin/bash
# two-wave (PDMP_HELPER_WAVE=1) against one-wave (=0) form of the tracked kernel across ensemble widths: where the launcher's limit belongs
for hw in 1 0; do for w in 2048 1536 1024 768 512 256; do
  echo -n "helper=$hw $w: "; PDMP_HELPER_WAVE=$hw timeout 200 python tools/strong_proxy.py --evals tracked --widths $w --steps 6 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print(j['kernel'], '%.2f ms' % j['ms_per_step'], '%.4g ev/s' % j['events_per_s'])"
done; done
