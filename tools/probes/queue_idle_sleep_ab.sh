#!/bin/bash
# A/B of an idle queue worker's pause between two polls of the tail (CVGS_QUEUE_IDLE_SLEEP, units of 64 clocks; product: 32):
# warm single-batch latency (server alive) by batch size, and the sustained rate, per variant.  Runs ON THE GPU BOX; build the
# variants first (build/ travels with the snapshot):
#   cd cvgpuspeedup_amd/csrc; mkdir -p ../../build/ab; OBJS=$(ls ../../build/csrc/*.o | grep -v "k_queue.hip.o\|exp")
#   for N in 32 8 2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DCVGS_QUEUE_IDLE_SLEEP=$N -c k_queue.hip -o ../../build/ab/k_queue_s$N.o
#     hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ab/libcvgs_hip_s$N.so $OBJS ../../build/ab/k_queue_s$N.o -ldl; done
cp cvgpuspeedup_amd/lib/libcvgs_hip.so /tmp/orig.so
for rep in 1 2; do
  for N in 32 8 2; do
    cp build/ab/libcvgs_hip_s$N.so cvgpuspeedup_amd/lib/libcvgs_hip.so
    for c in 1 8 50; do
      python tools/queue_ab.py --crops $c --depth 128 --batches 200 --replays 5 --variants "2,1,0" 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('idle_sleep $N crops $c: warm latency', j['warm_latency_us_median'], 'us; sustained', j['us_per_batch_median'], 'us per batch')"
    done
  done
done
cp /tmp/orig.so cvgpuspeedup_amd/lib/libcvgs_hip.so
