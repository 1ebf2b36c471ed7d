# same box: configs[3] (64 inputs x 12 sub-circuits, one gkr_prove_many call per step) by the cap on a lockstep group's proofs
for cap in 0 64 128 192 320; do
  echo "GKR_LOCKSTEP_MAX_PROOFS=$cap"
  GKR_LOCKSTEP_MAX_PROOFS=$cap PROBE_REPS=40 PROBE_THREADS=14,14 python tools/proof_many_probe.py 64 2>/dev/null | grep threads | cut -c1-120
done
