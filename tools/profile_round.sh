#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): collects the rocprofv3 evidence of one round into gpurun_out/<tag>/.
#   tools/profile_round.sh r01b
# Every rocprofv3 call is wrapped in `timeout` (rocprofv3 can hang at process exit after HIP-graph replays on this
# pool; the trace is complete before that) and uses csv output.  PMC passes are separate runs with
# --kernel-trace only (gpurun refuses --pmc combined with other trace domains).
set -u
TAG=${1:-round}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RP="rocprofv3 --kernel-trace --output-format csv"
# 1. the headline bench command, kernel trace + stats
timeout -k 5 240 $RP --stats -d $OUT/bench_trace -o t -- python bench.py --no-cpu --no-extra > $OUT/bench_trace.json 2> $OUT/bench_trace.err
# 2. HBM counters on the headline workload (eager, fewer steps: counters serialise the kernels)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 $RP --pmc $C -d $OUT/pmc_${C}_50 -o p -- python bench.py --steps 128 --warmup 8 --eager --no-cpu --no-extra > $OUT/pmc_${C}_50.json 2>/dev/null
  timeout -k 5 200 $RP --pmc $C -d $OUT/pmc_${C}_3200 -o p -- python bench.py --crops 3200 --table --steps 24 --warmup 4 --eager --no-cpu --no-extra > $OUT/pmc_${C}_3200.json 2>/dev/null
  timeout -k 5 200 $RP --pmc $C -d $OUT/calib_${C} -o p -- python tools/calibrate_pmc.py > $OUT/calib_${C}.txt 2>/dev/null
  timeout -k 5 200 $RP --pmc $C -d $OUT/more_${C} -o p -- python tools/bench_more.py --iters 20 > $OUT/more_${C}.json 2>/dev/null
done
# 3. kernel trace of the secondary configs
timeout -k 5 200 $RP --stats -d $OUT/more_trace -o t -- python tools/bench_more.py --iters 50 > $OUT/more_trace.json 2>/dev/null
ls -R $OUT | head -60
