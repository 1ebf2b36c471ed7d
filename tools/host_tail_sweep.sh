# same box: the (20,20) lone wide layer and the k = [18,20,20] proof by host_tail_log2 (where a phase's product passes move to the host)
for t in 0 7 8 9 10 11 12; do
  echo "GKR_HOST_TAIL_LOG2=$t"
  GKR_HOST_TAIL_LOG2=$t python tools/bench_wide.py 20,20 2>&1 | tail -1
  GKR_HOST_TAIL_LOG2=$t python tools/bench_wide_prove.py 2>&1 | tail -1
done
