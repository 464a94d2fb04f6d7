#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): round 6's evidence set of the final tree into gpurun_out/<tag>/ (copy to profiles/<tag>_*).
#   bash tools/probes/build_ablate.sh   (in the container: the ablation variants travel in build/ablate/)
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/profile_r06_final.sh r06_z'
set -u
TAG=${1:-r06_z}
OUT=gpurun_out/$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RP="rocprofv3 --kernel-trace --output-format csv"
SUM="python tools/prof_summary.py"
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/pytest_gpu_tail.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> /dev/null
cp bench_extra.json $OUT/bench_20_5_extra.json 2> /dev/null
python bench.py > $OUT/bench_default.json 2> /dev/null
python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-extra > $OUT/bench_force_dist.json 2> /dev/null
CVGS_BENCH_WORLD_ON_ONE_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 5 --no-extra > $OUT/bench_gpus2_on_one_gpu.json 2> /dev/null
# kernel trace of the headline command: the fused K1 launch's average duration against timing.tick_launch.us_per_launch of the same run
timeout -k 5 300 $RP --stats -d $RAW/bench_trace -o t -- python bench.py --no-cpu --no-extra --no-regimes --no-sweep --headline-only > $OUT/bench_trace.json 2> /dev/null
$SUM kernels $RAW/bench_trace/t_kernel_trace.csv > $OUT/bench_trace_kernels.txt 2>&1
# the L2 -> fabric request census of the headline launch + the guide's FETCH_SIZE / WRITE_SIZE, separate --pmc passes, --kernel-trace only
CMD="python bench.py --eager --steps 128 --warmup 16 --no-cpu --no-extra --no-regimes --no-sweep --headline-only"
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_BUBBLE_sum TCC_READ_SECTORS_sum TCC_REQ_sum TCC_MISS_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_SECTORS_sum TCC_HIT_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 300 $RP --pmc $SET -d $RAW/req_$i -o p -- $CMD > /dev/null 2>&1
  $SUM pmc $RAW/req_$i/p_counter_collection.csv k1_resize > $OUT/pmc_requests_${i}_ticks16.txt 2>&1
done
python tools/probes/census_summary.py $OUT > $OUT/request_census.txt 2>&1
# the kernels against their own skeletons (build/ablate/: tools/probes/build_ablate.sh)
if [ -f build/ablate/libcvgs_ldst.so ]; then
  python tools/probes/tick_ablation.py --m 16 --rounds 4 --variants full,ldst,ld,st,desc --out $OUT/tick_ablation_m16.txt > /dev/null 2>&1
  python tools/probes/tick_ablation.py --m 1 --rounds 3 --variants full,ldst,ld,st,desc --out $OUT/tick_ablation_m1.txt > /dev/null 2>&1
  python tools/probes/tick_ablation.py --m 16 --rounds 3 --fixed --variants full,ldst,ld,st,desc --out $OUT/tick_ablation_cfg2a.txt > /dev/null 2>&1
  python tools/probes/tick_ablation.py --workload cfg3 --m 1 --rounds 3 --variants k4_full,k4_ldst,k4_ld,k4_st,k4_math,k4_empty --out $OUT/cfg3_ablation_m1.txt > /dev/null 2>&1
  python tools/probes/tick_ablation.py --workload resize_write --m 3 --rounds 3 --variants x4_full,x4_ldst,x4_ld,x4_st,x4_math --out $OUT/resize_write_c3_ablation.txt > /dev/null 2>&1
fi
python tools/probes/two_stream_ticks.py > $OUT/two_stream_ticks.txt 2> /dev/null
python tools/bench_tick.py > $OUT/bench_tick_m16.txt 2> /dev/null
python tools/bench_more.py > $OUT/bench_more.txt 2> /dev/null
python tools/bench_reference_tests.py > $OUT/reference_test_chains.txt 2> /dev/null
python tools/bench_upscale.py --cn 1 3 4 > $OUT/bench_upscale.txt 2> /dev/null
python tools/perf_gate.py > $OUT/perf_gate.json 2> /dev/null
./examples/bin/serving_ticks > $OUT/serving_ticks_cpp.txt 2>&1
./examples/bin/sharded_crops --iters 50 2>&1 | grep -v "version\|Hostname\|Librccl" > $OUT/sharded_crops_cpp.txt
./examples/bin/benchmark_batchresize > $OUT/benchmark_batchresize_x_split3D.csv 2> /dev/null
CVGS_FUZZ_N=60000 CVGS_FUZZ_BIG_N=600 CVGS_FUZZ_CIRCULAR_N=1500 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -2 > $OUT/fuzz.txt
CVGS_FUZZ_SUBMIT_N=600 python -m pytest tests/test_gpu_submission_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -2 >> $OUT/fuzz.txt
ls -la $OUT
