# decode-side tick (16 NV12 surfaces x 50 crops, host descriptors, eager): descriptors in the kernel arguments against the pinned table ring
for R in 1 2 3; do for V in 1 0; do echo -n "inline=$V  "; CVGS_MANY_INLINE=$V bash tools/probes/k4_crops_line.sh | tail -1; done; done
