#!/bin/bash
# ABAB comparison of two builds of libcvgs_hip.so on ONE box (run under gpurun).  A-then-B orderings are not measurements on this pool:
# the boxes drift by ~3 % over a minute of benchmarks (round 5: a "no change" control read 0.35 us slower in the second half of a run).
#   bash tools/abab.sh build/ab/A_new.so build/ab/B_old.so 3 'bash tools/probes/preload_ab.sh'
# Build the two libraries in the container first (make; cp cvgpuspeedup_amd/lib/libcvgs_hip.so build/ab/A_x.so; git stash; make; ...): build/
# travels with the snapshot.  The installed library is restored at the end.
set -u
A=$1; B=$2; ROUNDS=${3:-3}; CMD=${4:-bash tools/probes/preload_ab.sh}
L=cvgpuspeedup_amd/lib/libcvgs_hip.so
cp $L /tmp/abab_keep.so
for R in $(seq 1 $ROUNDS); do
  for V in $A $B; do
    cp $V $L
    echo "== round $R $(basename $V)"
    eval "$CMD"
  done
done
cp /tmp/abab_keep.so $L
