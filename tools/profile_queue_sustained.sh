#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): HBM traffic of the queue's server in the regime bench.py TIMES (round 3's counters
# came from single-batch server calls -- 4-row tasks, 584 calls -- a different regime from the sustained 128-deep stream of the headline).
#   tools/profile_queue_sustained.sh r04_g
# Per counter (FETCH_SIZE, WRITE_SIZE: separate --pmc passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes):
#   1. tools/queue_ab.py --sustained: ONE server call serves 13 x 1024 batches with the ring full (128-row tasks); its counters / batches;
#   2. the same tool in round 3's regime (--batches 1 --retire-between): one batch per server call -- for the side-by-side;
#   3. tools/calibrate_pmc.py A / C / B: K1 launches with KNOWN byte counts (dense and 4:1 sparse taps) and the streaming copy, in the
#      same collection on the same box -- the correction factors for exactly these counters.
set -u
TAG=${1:-r04_pmc}
OUT=gpurun_out/$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RP="rocprofv3 --kernel-trace --output-format csv"
SUM="python tools/prof_summary.py"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 $RP --pmc $C -d $RAW/sus_$C -o p -- python tools/queue_ab.py --sustained --batches 1024 --replays 12 --depth 128 --variants "2,1,0" > $OUT/sustained_${C}_run.txt 2>/dev/null
  $SUM pmccalls $RAW/sus_$C/p_counter_collection.csv k1q_server > $OUT/sustained_${C}_server_calls.txt 2>&1
  timeout -k 5 300 $RP --pmc $C -d $RAW/one_$C -o p -- python tools/queue_ab.py --batches 1 --replays 40 --retire-between --variants "2,1,0" > $OUT/single_${C}_run.txt 2>/dev/null
  $SUM pmccalls $RAW/one_$C/p_counter_collection.csv k1q_server > $OUT/single_${C}_server_calls.txt 2>&1
  for W in A C B; do
    timeout -k 5 300 $RP --pmc $C -d $RAW/cal_${C}_$W -o p -- python tools/calibrate_pmc.py $W > /dev/null 2>&1
    $SUM pmc $RAW/cal_${C}_$W/p_counter_collection.csv > $OUT/calibrate_${C}_$W.txt 2>&1
  done
done
ls -la $OUT
