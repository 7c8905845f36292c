for st in "0.3,48,1" "0.3,56,1" "0.3,60,1" "0.3,44,1" "0.5,52,1" "0.3,52,1" "0.3,48,0.5" "0.3,48,1"; do
  echo -n "steer $st: "
  PDMP_HELPER_STEER=$st timeout 200 python tools/strong_proxy.py --evals tracked --widths 1024,512 --steps 6 2>/dev/null | python -c "
import sys,json
print(' '.join('%d:%.2f' % (json.loads(l)['chains'], json.loads(l)['ms_per_step']) for l in sys.stdin))"
done
