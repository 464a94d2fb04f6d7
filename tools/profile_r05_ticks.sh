#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the evidence set of round 5's headline regime (ticks of 16 frames as ONE cvgs_execute_many launch).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/profile_r05_ticks.sh r05_d'
set -u
TAG=${1:-r05_d}
OUT=gpurun_out/$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RP="rocprofv3 --kernel-trace --output-format csv"
SUM="python tools/prof_summary.py"
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err
cp bench_extra.json $OUT/bench_20_5_extra.json 2> /dev/null
tail -c 3000 $OUT/bench_20_5.err > $OUT/bench_20_5.err.tail; rm -f $OUT/bench_20_5.err
python bench.py --no-extra > $OUT/bench_default_no_extra.json 2> /dev/null
# kernel trace of the headline command (no side legs): the fused K1 launch's average duration must agree with timing.tick_launch.us_per_launch
timeout -k 5 300 $RP --stats -d $RAW/bench_trace -o t -- python bench.py --no-cpu --no-extra --no-regimes --no-sweep --headline-only > $OUT/bench_trace.json 2> /dev/null
$SUM kernels $RAW/bench_trace/t_kernel_trace.csv > $OUT/bench_trace_kernels.txt 2>&1
# HBM counters of the same kernel, eager (counters serialise the kernels), separate passes
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 $RP --pmc $C -d $RAW/pmc_$C -o p -- python bench.py --eager --steps 128 --warmup 16 --no-cpu --no-extra --no-regimes --no-sweep --no-queue-leg > /dev/null 2>&1
  $SUM pmc $RAW/pmc_$C/p_counter_collection.csv k1_resize > $OUT/pmc_${C}_ticks16.txt 2>&1
  for W in A C B; do
    timeout -k 5 300 $RP --pmc $C -d $RAW/cal_${C}_$W -o p -- python tools/calibrate_pmc.py $W > /dev/null 2>&1
    $SUM pmc $RAW/cal_${C}_$W/p_counter_collection.csv cvgs:: > $OUT/calibrate_${C}_$W.txt 2>&1
  done
done
python tools/bench_tick.py > $OUT/bench_tick_m16.txt 2> /dev/null
python tools/bench_tick.py --m 64 --frames 128 > $OUT/bench_tick_m64.txt 2> /dev/null
ls -la $OUT
