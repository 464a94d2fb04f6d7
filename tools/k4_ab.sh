#!/bin/bash
# A/B of the K4 (NV12 resize) kernel build variants under build/ab/ (see DESIGN.md, K4): swaps the product library in place
# on the GPU box, runs cfg #3 and the 50-crop NV12 batch with and without the FMA-corrected division, restores the library.
cp cvgpuspeedup_amd/lib/libcvgs_hip.so /tmp/orig.so
for V in 0 1 2 3; do
  if [ $V = 0 ]; then cp /tmp/orig.so cvgpuspeedup_amd/lib/libcvgs_hip.so; else cp build/ab/libcvgs_hip_$V.so cvgpuspeedup_amd/lib/libcvgs_hip.so; fi
  for FD in 1 0; do
    echo "variant $V fastdiv=$FD"
    export CVGS_K1_FASTDIV=$FD
    for W in cfg3 nv12crops; do
      python tools/bench_more.py --iters 300 --only $W 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('   ', j['config'][:40], j['us_per_launch'])"
    done
  done
done
cp /tmp/orig.so cvgpuspeedup_amd/lib/libcvgs_hip.so
