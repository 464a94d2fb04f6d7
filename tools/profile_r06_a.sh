#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): round 6's first evidence set -- the headline tick against its own memory skeletons (tools/probes/tick_ablation.py,
# libraries from tools/probes/build_ablate.sh) and the L2 -> fabric request census of the same launch (raw TCC counters, separate --pmc passes).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/profile_r06_a.sh r06_a'
set -u
TAG=${1:-r06_a}
OUT=gpurun_out/$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/probes/tick_ablation.py --m 16 --rounds 4 --out $OUT/tick_ablation_m16.txt > /dev/null 2> $OUT/tick_ablation_m16.err
python tools/probes/tick_ablation.py --m 1 --rounds 3 --variants full,ldst,ld,st,desc,plain,sc1 --out $OUT/tick_ablation_m1.txt > /dev/null 2> $OUT/tick_ablation_m1.err
python tools/probes/tick_ablation.py --m 64 --frames 128 --rounds 3 --variants full,ldst,ld,st,zfast --out $OUT/tick_ablation_m64.txt > /dev/null 2> $OUT/tick_ablation_m64.err
RP="rocprofv3 --kernel-trace --output-format csv"
SUM="python tools/prof_summary.py"
CMD="python bench.py --eager --steps 128 --warmup 16 --no-cpu --no-extra --no-regimes --no-sweep --no-queue-leg"
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_BUBBLE_sum TCC_READ_SECTORS_sum TCC_REQ_sum TCC_MISS_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_SECTORS_sum TCC_HIT_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"; do
  i=$((i+1))
  timeout -k 5 300 $RP --pmc $SET -d $RAW/req_$i -o p -- $CMD > /dev/null 2>&1
  $SUM pmc $RAW/req_$i/p_counter_collection.csv k1_resize > $OUT/pmc_requests_${i}_ticks16.txt 2>&1
  # the same counters on launches of KNOWN byte counts (tools/calibrate_pmc.py: A = K1 at identity scale, dense taps; C = K1 4:1 x 1:1, sparse
  # taps that touch every 64-byte sector; B = the 16-byte streaming copy)
  for W in A C B; do
    timeout -k 5 300 $RP --pmc $SET -d $RAW/cal_${i}_$W -o p -- python tools/calibrate_pmc.py $W > /dev/null 2>&1
    $SUM pmc $RAW/cal_${i}_$W/p_counter_collection.csv cvgs:: > $OUT/pmc_requests_${i}_calibrate_$W.txt 2>&1
  done
done
ls -la $OUT
cat $OUT/tick_ablation_m16.txt | head -20
