for q in 16 22 30 36 44; do for r in 56 62; do
echo "== targetq=$q raw=$r"; PDMP_HELPER_STEER=$q,$r,0 python tools/strong_proxy.py --widths 4096 --evals tracked --phase --steps 4 2>&1 | grep -o '"ms_per_step": [0-9.]*\|proposals_per_iter": [0-9.]*\|alias [0-9]* near [0-9]*\|iters [0-9]* raw [0-9]* nev [0-9]*' | tr '\n' ' '; echo
done; done
