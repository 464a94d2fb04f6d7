set -x
mkdir -p gpurun_out/r02p
timeout 900 python -m pytest tests/test_gpu_k4_planar.py tests/test_yuv_layouts.py tests/test_gpu_circular_nv12.py tests/test_p010.py -m gpu -x -q -n 4 > gpurun_out/r02p/k4_u8_tests.txt 2>&1; tail -15 gpurun_out/r02p/k4_u8_tests.txt
timeout 600 python tools/bench_nv12_letterbox.py > gpurun_out/r02p/k4_u8_bench.txt 2>&1; cat gpurun_out/r02p/k4_u8_bench.txt
CVGS_FUZZ_N=20000 timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r02p/fuzz2.txt 2>&1; tail -3 gpurun_out/r02p/fuzz2.txt
