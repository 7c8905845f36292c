mkdir -p gpurun_out/r05c
for st in "1.05,0.97,12,2" "1.1,0.95,20,2" "1.1,0.95,30,2" "1.2,0.9,40,2" "1.05,0.97,12,1" "1.05,0.97,12,4" "1.02,0.98,5,2" "1.1,0.9,16,3"; do
  echo "steer $st"
  PDMP_HELPER_STEER=$st timeout 200 python tools/strong_proxy.py --evals tracked --phase --widths 512 --steps 6 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); p=r['phase']; print(r['chains'], round(r['ms_per_step'],2), 'prop/it %.1f sel %.1f zone %.1f'%(p['proposals_per_iter'],p['cand_selected'],p['cand_after_zone']), p['cycles_per_iter'])
"
done
