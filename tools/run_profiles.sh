# Round artefacts on the GPU box:  bash tools/run_profiles.sh <out-dir-under-gpurun_out>
# GPU tests, the default bench line, rocprofv3 kernel stats of the bench command, and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, as MI355X_MICROARCH.md's HBM section prescribes) -> pmc_traffic.json
set -x
OUT=gpurun_out/${1:-r2x}
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $OUT/gpu_tests.txt
GKR_BENCH_DETAIL=$OUT/bench_default_detail.json python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err   # (the driver's exact command)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export GKR_BENCH_DETAIL=/tmp/bench_detail_scratch.json   # (only the runs that name their own keep a detail file)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-verify --proofs 0 > /tmp/stats_bench.json 2>/dev/null
cp $(ls /tmp/prof_stats/*/*kernel_stats.csv | head -1) $R/$OUT/kernel_stats.csv
cp /tmp/stats_bench.json $R/$OUT/stats_bench.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify --no-profile --proofs 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify --no-profile --proofs 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_proofs -- python3 $R/bench.py --mode proofs --steps 3 --warmup 1 > /tmp/proofs_bench.json 2>/dev/null
cp $(ls /tmp/prof_proofs/*/*kernel_stats.csv | head -1) $R/$OUT/kernel_stats_mode_proofs.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_layer -- python3 $R/bench.py --mode layer-split --steps 3 --warmup 1 > /tmp/layer_bench.json 2>/dev/null
cp $(ls /tmp/prof_layer/*/*kernel_stats.csv | head -1) $R/$OUT/kernel_stats_mode_layer_split.csv
cd $R
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write 1024 20 > $OUT/pmc_traffic.json
PROBE_REPS=60 PROBE_THREADS=14,12 python tools/proof_many_probe.py 64 > $OUT/proof_many_probe_64.txt 2>/dev/null
PROBE_REPS=60 PROBE_THREADS=14 python tools/proof_many_probe.py 3 > $OUT/proof_many_probe_3.txt 2>/dev/null
GKR_BENCH_DETAIL=$OUT/bench_mode_layer_split_detail.json python bench.py --mode layer-split --steps 10 --warmup 3 > $OUT/bench_mode_layer_split.json 2>/dev/null
GKR_BENCH_FORCE_GROUP=1 python bench.py --mode layer-split --steps 10 --warmup 3 > $OUT/bench_mode_layer_split_rccl_one_rank.json 2>/dev/null
GKR_BENCH_DETAIL=$OUT/bench_mode_proofs_detail.json python bench.py --mode proofs --steps 10 --warmup 3 > $OUT/bench_mode_proofs.json 2>/dev/null
LOCAL_WORLD_SIZE=8 python bench.py --no-cpu-baseline --no-extras --no-verify --proofs 0 > $OUT/bench_emulated_8_ranks.json 2>/dev/null
python bench.py --mode mle-split --log2-points 20 > $OUT/bench_mode_mle_split_n20.json 2>/dev/null
GKR_BENCH_FORCE_GROUP=1 python bench.py --mode mle-split --log2-points 20 > $OUT/bench_mode_mle_split_n20_rccl_one_rank.json 2>/dev/null
python bench.py --mode mle-split --log2-points 30 --steps 5 > $OUT/bench_mode_mle_split_n30.json 2>/dev/null
# the default line with two ranks (two processes over gloo sharing the one GPU), started by bench.py ITSELF (--gpus 2, no launcher
# around it): every N > 1 field of the line
GKR_BENCH_DETAIL=$OUT/bench_default_gpus2_self_launched_gloo_one_gpu_detail.json GKR_BENCH_BACKEND=gloo GKR_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 4 --warmup 2 --batch 256 > $OUT/bench_default_gpus2_self_launched_gloo_one_gpu.json 2> $OUT/bench_default_gpus2_self_launched_gloo_one_gpu.err
python tools/bench_wide.py 20,15 22,16 24,18 20,20 22,22 > $OUT/bench_wide_layers.txt 2>&1
WIDE_SHAPE=circom python tools/bench_wide.py 20,20 22,22 > $OUT/bench_wide_layers_circom_shaped.txt 2>&1
bash tools/stats_large_r1cs.sh ${1:-r2x}/large_r1cs_lockstep 10 14 > $OUT/large_r1cs_lockstep_kernel_stats_summary.txt 2>&1
GKR_PROVE_MANY_LOCKSTEP=0 bash tools/stats_large_r1cs.sh ${1:-r2x}/large_r1cs_one_chain_per_item 10 14 > $OUT/large_r1cs_one_chain_per_item_kernel_stats_summary.txt 2>&1
python tools/bench_large_r1cs.py 30 14 8 4 > $OUT/large_r1cs_step_by_threads.txt 2>&1
python tools/config3_accounts.py 64 14 20 > $OUT/config3_thread_and_piece_accounts.txt 2>&1
bash tools/stats_wide.sh 20,20 24,18 > /dev/null 2>&1; cp gpurun_out/wide_stats_20_20.csv $OUT/kernel_stats_wide_layer_k_i20_k20.csv; cp gpurun_out/wide_stats_24_18.csv $OUT/kernel_stats_wide_layer_k_i24_k18.csv
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gkr_amd/csrc tools/ubench_cross.hip -o /tmp/ubench_cross > /dev/null 2>&1 && /tmp/ubench_cross > $OUT/ubench_product_pass_cross_sums.txt 2>&1
bash tools/pmc_product_passes.sh 20,20 > $OUT/product_pass_pmc_k_i20_k20.txt 2>&1
bash tools/pmc_gate_passes.sh > $OUT/seg_pass_pmc_counters_layer24.txt 2>&1
bash tools/trace_wide_timeline.sh > $OUT/wide_layer_k_i20_k20_kernel_timeline.txt 2>&1
bash tools/variants_ab.sh > $OUT/layer24_gate_pass_kernel_ms.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
bash tools/trace_layer_timeline.sh > $OUT/layer24_kernel_timeline.txt 2>&1
bash tools/trace_mle_latency_timeline.sh > $OUT/mle_batch1_kernel_timeline.txt 2>&1
cd $R
cat $OUT/gpu_tests.txt $OUT/smoke.txt
head -c 1500 $OUT/bench_default.json
head -12 $OUT/kernel_stats.csv
python -c "
import json; d=json.load(open('$OUT/pmc_traffic.json')); print({k:v for k,v in d.items() if k.startswith('k_')})"
