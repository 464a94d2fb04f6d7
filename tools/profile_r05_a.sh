#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): round 5's first evidence set -- the residency question and cfg #3 / #4.
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/profile_r05_a.sh r05_a'
set -u
TAG=${1:-r05_a}
OUT=gpurun_out/$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
RP="rocprofv3 --kernel-trace --output-format csv"
SUM="python tools/prof_summary.py"
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $OUT/pytest_gpu_tail.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err
cp bench_extra.json $OUT/bench_20_5_extra.json 2> /dev/null
tail -c 2000 $OUT/bench_20_5.err > $OUT/bench_20_5.err.tail; rm -f $OUT/bench_20_5.err
# HBM counters of the server in the sustained regime, on rotations of 96 and 20 frames (touched set 4.6 x / 0.96 x the Infinity Cache)
for F in 96 20; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 300 $RP --pmc $C -d $RAW/sus_${C}_$F -o p -- python tools/queue_ab.py --frames $F --sustained --batches 1024 --replays 12 --depth 128 --variants "2,1,0" > $OUT/sustained_f${F}_${C}_run.txt 2>/dev/null
    $SUM pmccalls $RAW/sus_${C}_$F/p_counter_collection.csv k1q_server > $OUT/sustained_f${F}_${C}_server_calls.txt 2>&1
  done
done
for C in FETCH_SIZE WRITE_SIZE; do
  for W in A C B; do
    timeout -k 5 300 $RP --pmc $C -d $RAW/cal_${C}_$W -o p -- python tools/calibrate_pmc.py $W > /dev/null 2>&1
    $SUM pmc $RAW/cal_${C}_$W/p_counter_collection.csv > $OUT/calibrate_${C}_$W.txt 2>&1
  done
done
# cfg #3 / cfg #4: kernel trace + HBM counters + (cfg #3) the SQ counters that say what bounds K4
for CFG in cfg3 cfg4; do
  timeout -k 5 300 $RP --stats -d $RAW/${CFG}_trace -o t -- python tools/bench_more.py --only $CFG --iters 60 > $OUT/${CFG}_lines.txt 2>/dev/null
  $SUM kernels $RAW/${CFG}_trace/t_kernel_trace.csv > $OUT/${CFG}_trace_kernels.txt 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 300 $RP --pmc $C -d $RAW/${CFG}_$C -o p -- python tools/bench_more.py --only $CFG --iters 20 > /dev/null 2>&1
    $SUM pmc $RAW/${CFG}_$C/p_counter_collection.csv > $OUT/${CFG}_pmc_$C.txt 2>&1
  done
done
timeout -k 5 300 $RP --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $RAW/cfg3_sq1 -o p -- python tools/bench_more.py --only cfg3 --iters 20 > /dev/null 2>&1
$SUM pmc $RAW/cfg3_sq1/p_counter_collection.csv > $OUT/cfg3_pmc_sq1.txt 2>&1
timeout -k 5 300 $RP --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY -d $RAW/cfg3_sq2 -o p -- python tools/bench_more.py --only cfg3 --iters 20 > /dev/null 2>&1
$SUM pmc $RAW/cfg3_sq2/p_counter_collection.csv > $OUT/cfg3_pmc_sq2.txt 2>&1
ls -la $OUT
