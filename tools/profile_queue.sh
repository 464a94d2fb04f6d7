#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the rocprofv3 evidence of the headline on the device-side descriptor queue.
#   tools/profile_queue.sh r03_c
# The dominant kernel is the server grid k1q_server: ONE call serves the whole timed region of bench.py, which prints how many
# batches that call served -- its duration / batches is the per-batch kernel time of roofline.achieved.  PMC passes are separate
# runs with --kernel-trace only.
set -u
TAG=${1:-queue}
OUT=gpurun_out/$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RP="rocprofv3 --kernel-trace --output-format csv"
SUM="python tools/prof_summary.py"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra > $OUT/bench_unprofiled_20_5.json 2>/dev/null
cp bench_extra.json $OUT/bench_unprofiled_20_5_extra.json 2>/dev/null
timeout -k 5 300 $RP --stats -d $RAW/trace -o t -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extra --no-regimes > $OUT/bench_trace.json 2> $OUT/bench_trace.err
cp bench_extra.json $OUT/bench_trace_extra.json 2>/dev/null
$SUM kernels $RAW/trace/t_kernel_trace.csv > $OUT/bench_trace_kernels.txt 2>&1
$SUM calls $RAW/trace/t_kernel_trace.csv k1q_server > $OUT/bench_trace_server_calls.txt 2>&1
# HBM counters: rocprofv3 --pmc segfaults inside bench.py on this image (with and without the stream events, staged or direct
# slots) and works on tools/queue_ab.py, which drives the same server with the same batches: one batch per server call
# (--retire-between), so every k1q_server row of the counter file is ONE 50-crop batch.
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 $RP --pmc $C -d $RAW/pmc_$C -o p -- python tools/queue_ab.py --batches 1 --replays 40 --retire-between > $OUT/queue_ab_pmc_$C.txt 2>/dev/null
  $SUM pmccalls $RAW/pmc_$C/p_counter_collection.csv k1q_server > $OUT/pmc_${C}_server_calls.txt 2>&1
done
ls -la $OUT
