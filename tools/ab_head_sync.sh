for rep in 1 2; do for lib in after before; do
  if [ $lib = before ]; then export GKR_AMD_LIB=$PWD/gkr_amd/lib/libgkr_amd_before.so; else unset GKR_AMD_LIB; fi
  echo "== $lib"; timeout 300 python tools/bench_large_r1cs.py 30 14 2>&1 | tail -1 | cut -c1-70
  PROBE_THREADS=14 PROBE_REPS=40 timeout 300 python tools/proof_many_probe.py 3 2>&1 | tail -1 | cut -c1-45
  PROBE_THREADS=14 PROBE_REPS=40 timeout 300 python tools/proof_many_probe.py 64 2>&1 | tail -1 | cut -c1-45
  timeout 300 python tools/large_r1cs_chain_probe.py 10 | tail -1 | cut -c50-140
done; done
