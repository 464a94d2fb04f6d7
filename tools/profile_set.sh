#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the whole evidence set of a round into gpurun_out/<tag>/ -- the headline's rocprofv3 trace and HBM
# counters (tools/profile_queue.sh), the default bench line, the other benches, the perf gate and the GPU test suite's tail.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/profile_set.sh r03_j'
# then copy gpurun_out/<tag>/* to profiles/<tag>_* (tools/README.md).
set -u
TAG=${1:-set}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/pytest_gpu_tail.txt
bash tools/profile_queue.sh $TAG > /dev/null 2>&1
python bench.py > $OUT/bench_default.json 2> /dev/null
cp bench_extra.json $OUT/bench_default_extra.json 2> /dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> /dev/null
cp bench_extra.json $OUT/bench_20_5_extra.json 2> /dev/null
python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-extra > $OUT/bench_force_dist.json 2> /dev/null
bash tools/profile_queue_sustained.sh $TAG > /dev/null 2>&1
(echo "== GPU_MAX_HW_QUEUES=3 (what the tools set for themselves: no caller queue shares the server's pipe), producer kernel on the stream =="; GPU_MAX_HW_QUEUES=3 python tools/probes/stream_ordered_rate.py --producer 2>&1 | grep -v amdgpu.ids
 echo; echo "== the runtime's default (GPU_MAX_HW_QUEUES=4), producer kernel on the stream =="; GPU_MAX_HW_QUEUES=4 python tools/probes/stream_ordered_rate.py --producer 2>&1 | grep -v amdgpu.ids
 echo; echo "== GPU_MAX_HW_QUEUES=3, no producer kernel =="; GPU_MAX_HW_QUEUES=3 python tools/probes/stream_ordered_rate.py 2>&1 | grep -v amdgpu.ids) > $OUT/stream_ordered_rate.txt
(echo "== GPU_MAX_HW_QUEUES=3 =="; GPU_MAX_HW_QUEUES=3 python tools/probes/gate_trace.py 2>&1 | grep -v amdgpu.ids; echo "== the runtime's default (4) =="; GPU_MAX_HW_QUEUES=4 python tools/probes/gate_trace.py 2>&1 | grep -v amdgpu.ids) > $OUT/gate_trace.txt
(for Q in 4 8 3 2; do echo "== GPU_MAX_HW_QUEUES=$Q: 24 fresh streams, us per empty one-wave kernel =="; GPU_MAX_HW_QUEUES=$Q python tools/probes/server_vs_streams.py --many 2>&1 | grep -v amdgpu.ids; done) > $OUT/server_vs_streams.txt
python tools/bench_queue_regimes.py --soak 60 2>&1 | grep -v amdgpu.ids > $OUT/queue_regimes_soak60.txt
python tools/bench_queue_regimes.py --g-sweep 2>&1 | grep -v amdgpu.ids > $OUT/coexistence_g_sweep.txt
python tools/bench_reference_tests.py > $OUT/reference_test_chains.txt 2> /dev/null
python tools/bench_more.py > $OUT/bench_more.txt 2> /dev/null
python tools/bench_upscale.py > $OUT/bench_upscale.txt 2> /dev/null
python tools/bench_upscale.py --cn 4 >> $OUT/bench_upscale.txt 2> /dev/null
python tools/bench_upscale.py --cn 1 >> $OUT/bench_upscale.txt 2> /dev/null
python tools/bench_nv12_letterbox.py > $OUT/bench_nv12_letterbox.txt 2> /dev/null
python tools/perf_gate.py > $OUT/perf_gate.json 2> /dev/null
./examples/bin/benchmark_batchresize > $OUT/benchmark_batchresize_x_split3D.csv 2> /dev/null
ls -la $OUT
