#!/bin/bash
# Runs ON THE GPU BOX: cfg #3's kernel against its own skeletons (one surface per launch, and a tick of 4), and the headline's tap loads on frames
# where the crops do not overlap (is the read side bound by what the L2s fetch or by what HBM delivers?).
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/profile_r06_b.sh r06_b'
set -u
TAG=${1:-r06_b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/probes/tick_ablation.py --workload cfg3 --m 1 --rounds 4 --out $OUT/cfg3_ablation_m1.txt > /dev/null 2> $OUT/cfg3_ablation_m1.err
python tools/probes/tick_ablation.py --workload cfg3 --m 4 --rounds 3 --out $OUT/cfg3_ablation_m4.txt > /dev/null 2> $OUT/cfg3_ablation_m4.err
python tools/probes/tick_ablation.py --m 16 --rounds 3 --frame 4k --variants full,ldst,ld,st --out $OUT/tick_ablation_4k.txt > /dev/null 2> $OUT/tick_ablation_4k.err
python tools/probes/tick_ablation.py --m 16 --rounds 3 --frame 8k --frames 48 --variants full,ldst,ld,st --out $OUT/tick_ablation_8k.txt > /dev/null 2> $OUT/tick_ablation_8k.err
head -12 $OUT/cfg3_ablation_m1.txt $OUT/cfg3_ablation_m4.txt $OUT/tick_ablation_4k.txt $OUT/tick_ablation_8k.txt | cut -c1-150
tail -3 $OUT/*.err
