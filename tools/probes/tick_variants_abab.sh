L=cvgpuspeedup_amd/lib/libcvgs_hip.so
cp $L /tmp/keep.so
for R in 1 2; do
  for V in build/ab/A_wpb2.so build/ab/B_wpb4.so build/ab/B_wpb1.so; do
    cp $V $L; echo "== round $R $(basename $V)"
    bash tools/probes/tick_variants.sh
  done
done
cp /tmp/keep.so $L
