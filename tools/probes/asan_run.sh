# runs the facade programs' ASan + UBSan builds against the ASan build of the library's host code (make -C cvgpuspeedup_amd/csrc asan; make -C tests/cpp asan; make -C examples asan)
export LD_LIBRARY_PATH=$PWD/build/asan:$LD_LIBRARY_PATH
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0
mkdir -p gpurun_out/asan
for t in ${@:-test_batchresize test_circulartensor test_divergent test_pointwise test_resize test_warping serving_ticks readme_example}; do
  exe=tests/cpp/bin/${t}_asan; [ -x $exe ] || exe=examples/bin/${t}_asan
  timeout 600 $exe > gpurun_out/asan/$t.txt 2>&1; echo "== $t rc=$?"; grep -n "ERROR\|SUMMARY\|passed\|runtime error\|bit for bit" gpurun_out/asan/$t.txt | head -8
done
