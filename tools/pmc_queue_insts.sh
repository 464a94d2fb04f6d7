#!/bin/bash
# Runs ON THE GPU BOX: instructions the queue server executes per 50-crop batch -- SQ counters of k1q_server calls that serve
# N1 and N2 batches each (tools/queue_ab.py --retire-between: one server call per replay); (N2 - N1) batches account for the difference.
#   bash tools/pmc_queue_insts.sh <tag> [N1 N2]
TAG=${1:-qinsts}; N1=${2:-8}; N2=${3:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for N in $N1 $N2; do
  rm -rf /tmp/pq_$N
  timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES -d /tmp/pq_$N -o p -- \
    python tools/queue_ab.py --batches $N --replays 24 --retire-between --variants "2,1,0" > /dev/null 2>&1
  python tools/prof_summary.py pmccalls /tmp/pq_$N/p_counter_collection.csv k1q_server > gpurun_out/${TAG}_calls_$N.txt 2>&1
done
tail -n 12 gpurun_out/${TAG}_calls_$N1.txt gpurun_out/${TAG}_calls_$N2.txt | cut -c1-220
