#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): collects the rocprofv3 evidence of one round into gpurun_out/<tag>/ as SMALL text
# summaries (the raw traces stay in /tmp on the box: gpurun merges at most 64 MiB back).
#   tools/profile_round.sh r02
# Every rocprofv3 call is wrapped in `timeout` (rocprofv3 can hang at process exit after HIP-graph replays on this
# pool; the trace is complete before that) and uses csv output.  PMC passes are separate runs with
# --kernel-trace only (gpurun refuses --pmc combined with other trace domains).
set -u
TAG=${1:-round}
OUT=gpurun_out/$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RP="rocprofv3 --kernel-trace --output-format csv"
SUM="python tools/prof_summary.py"
# 0. the un-profiled headline lines (driver's command and the default)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_unprofiled_20_5.json 2>/dev/null
python bench.py --no-cpu --no-extra > $OUT/bench_unprofiled.json 2>/dev/null
python bench.py --frames-per-launch 16 --no-cpu --no-extra > $OUT/bench_unprofiled_m16.json 2>/dev/null
# 1. the headline bench command, kernel trace + stats; the same with 16 frames fused per launch
timeout -k 5 240 $RP --stats -d $RAW/bench_trace -o t -- python bench.py --no-cpu --no-extra > $OUT/bench_trace.json 2> $OUT/bench_trace.err
$SUM kernels $RAW/bench_trace/t_kernel_trace.csv > $OUT/bench_trace_kernels.txt 2>&1
head -6 $RAW/bench_trace/t_kernel_stats.csv > $OUT/bench_trace_stats_head.txt 2>/dev/null
timeout -k 5 240 $RP --stats -d $RAW/bench_trace_m16 -o t -- python bench.py --frames-per-launch 16 --no-cpu --no-extra > $OUT/bench_trace_m16.json 2>/dev/null
$SUM kernels $RAW/bench_trace_m16/t_kernel_trace.csv > $OUT/bench_trace_m16_kernels.txt 2>&1
# 2. HBM counters (eager, fewer steps: counters serialise the kernels)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 $RP --pmc $C -d $RAW/pmc_${C}_50 -o p -- python bench.py --steps 128 --warmup 8 --eager --no-cpu --no-extra > /dev/null 2>&1
  $SUM pmc $RAW/pmc_${C}_50/p_counter_collection.csv k1_resize > $OUT/pmc_${C}_50.txt 2>&1
  timeout -k 5 200 $RP --pmc $C -d $RAW/pmc_${C}_m16 -o p -- python bench.py --frames-per-launch 16 --steps 32 --warmup 4 --eager --no-cpu --no-extra > /dev/null 2>&1
  $SUM pmc $RAW/pmc_${C}_m16/p_counter_collection.csv k1_resize > $OUT/pmc_${C}_m16.txt 2>&1
  timeout -k 5 200 $RP --pmc $C -d $RAW/pmc_${C}_3200 -o p -- python bench.py --crops 3200 --table --steps 24 --warmup 4 --eager --no-cpu --no-extra > /dev/null 2>&1
  $SUM pmc $RAW/pmc_${C}_3200/p_counter_collection.csv k1_resize > $OUT/pmc_${C}_3200.txt 2>&1
  for K in A B C; do
    timeout -k 5 200 $RP --pmc $C -d $RAW/calib_${C}_$K -o p -- python tools/calibrate_pmc.py $K > $OUT/calib_${C}_$K.txt 2>/dev/null
    $SUM pmc $RAW/calib_${C}_$K/p_counter_collection.csv cvgs:: >> $OUT/calib_${C}_$K.txt 2>&1
  done
  timeout -k 5 200 $RP --pmc $C -d $RAW/more_${C} -o p -- python tools/bench_more.py --iters 20 > /dev/null 2>&1
  $SUM pmc $RAW/more_${C}/p_counter_collection.csv > $OUT/more_${C}_pmc.txt 2>&1
done
# 3. kernel trace of the secondary configs and of the u8 colour conversions
timeout -k 5 200 $RP --stats -d $RAW/more_trace -o t -- python tools/bench_more.py --iters 50 > $OUT/more_trace.json 2>/dev/null
$SUM kernels $RAW/more_trace/t_kernel_trace.csv > $OUT/more_trace_kernels.txt 2>&1
timeout -k 5 200 $RP --stats -d $RAW/cc_trace -o t -- python tools/bench_cvtcolor.py > $OUT/cvtcolor_trace.json 2>/dev/null
$SUM kernels $RAW/cc_trace/t_kernel_trace.csv > $OUT/cvtcolor_trace_kernels.txt 2>&1
ls -la $OUT
