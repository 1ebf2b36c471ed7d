set -x
mkdir -p gpurun_out/r1d
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r1d/gpu_tests.txt
python bench.py > gpurun_out/r1d/bench_default.json 2> gpurun_out/r1d/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --proofs 0 > /tmp/stats_bench.json 2>/dev/null
cp $(ls /tmp/prof_stats/*/*kernel_stats.csv | head -1) $GRAFT_REPO_ROOT/gpurun_out/r1d/kernel_stats.csv
cp /tmp/stats_bench.json $GRAFT_REPO_ROOT/gpurun_out/r1d/stats_bench.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --proofs 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --proofs 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write 1024 20 > gpurun_out/r1d/pmc_traffic.json
cat gpurun_out/r1d/gpu_tests.txt
head -c 1500 gpurun_out/r1d/bench_default.json
head -12 gpurun_out/r1d/kernel_stats.csv
python -c "
import json; d=json.load(open('gpurun_out/r1d/pmc_traffic.json')); print({k:v for k,v in d.items() if k.startswith('k_')})"
