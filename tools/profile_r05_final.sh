#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): round 5's evidence set of the final tree into gpurun_out/<tag>/ (copy to profiles/<tag>_*).
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/profile_r05_final.sh r05_z'
set -u
TAG=${1:-r05_z}
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
python bench.py --submission queue --no-extra --no-regimes > $OUT/bench_submission_queue.json 2> /dev/null
# kernel trace of the headline command: the fused K1 launch's average duration against timing.tick_launch.us_per_launch of the same run
timeout -k 5 300 $RP --stats -d $RAW/bench_trace -o t -- python bench.py --no-cpu --no-extra --no-regimes --no-sweep --headline-only > $OUT/bench_trace.json 2> /dev/null
$SUM kernels $RAW/bench_trace/t_kernel_trace.csv > $OUT/bench_trace_kernels.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 $RP --pmc $C -d $RAW/pmc_$C -o p -- python bench.py --eager --steps 128 --warmup 16 --no-cpu --no-extra --no-regimes --no-sweep --headline-only > /dev/null 2>&1
  $SUM pmc $RAW/pmc_$C/p_counter_collection.csv k1_resize > $OUT/pmc_${C}_ticks16.txt 2>&1
  for W in A C; do
    timeout -k 5 300 $RP --pmc $C -d $RAW/cal_${C}_$W -o p -- python tools/calibrate_pmc.py $W > /dev/null 2>&1
    $SUM pmc $RAW/cal_${C}_$W/p_counter_collection.csv cvgs:: > $OUT/calibrate_${C}_$W.txt 2>&1
  done
done
python tools/bench_tick.py > $OUT/bench_tick_m16.txt 2> /dev/null
python tools/bench_more.py > $OUT/bench_more.txt 2> /dev/null
python tools/bench_reference_tests.py > $OUT/reference_test_chains.txt 2> /dev/null
python tools/bench_upscale.py --cn 1 3 4 > $OUT/bench_upscale.txt 2> /dev/null
python tools/perf_gate.py > $OUT/perf_gate.json 2> /dev/null
./examples/bin/serving_ticks > $OUT/serving_ticks_cpp.txt 2>&1
CVGS_FUZZ_N=60000 CVGS_FUZZ_BIG_N=600 CVGS_FUZZ_CIRCULAR_N=1500 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -2 > $OUT/fuzz.txt
CVGS_FUZZ_SUBMIT_N=600 python -m pytest tests/test_gpu_submission_fuzz.py -q -x 2>&1 | grep -E "passed|failed" | tail -2 >> $OUT/fuzz.txt
ls -la $OUT
