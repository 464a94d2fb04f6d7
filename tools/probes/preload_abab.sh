# ABAB of two builds of libcvgs_hip.so (build/ab/A_*.so, build/ab/B_*.so) on the launch-bound regimes, same box, alternating (order effects:
# the box's clocks drift by ~3 % over a minute of benchmarks, so "A then B" is not a measurement)
A=$(ls build/ab/A_*.so); B=$(ls build/ab/B_*.so); L=cvgpuspeedup_amd/lib/libcvgs_hip.so
cp $L /tmp/keep.so
for R in 1 2 3; do
  for V in $A $B; do
    cp $V $L; echo "== round $R $(basename $V)"
    bash tools/probes/preload_ab.sh
    bash tools/probes/preload_ab_cfg3.sh | head -3 | cut -c1-60,200-330
  done
done
cp /tmp/keep.so $L
