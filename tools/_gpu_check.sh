set -x
mkdir -p gpurun_out/r02q
timeout 900 python -m pytest tests -m gpu -x -q -n 4 > gpurun_out/r02q/gputests.txt 2>&1; tail -2 gpurun_out/r02q/gputests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02q/smoke.txt 2>&1; tail -1 gpurun_out/r02q/smoke.txt
timeout 1500 bash tools/profile_round.sh r02q > gpurun_out/r02q/profile_round.log 2>&1
cat gpurun_out/r02q/bench_unprofiled_20_5.json | cut -c1-300
