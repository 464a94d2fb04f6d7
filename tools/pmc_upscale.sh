# SQ / HBM counters of the 1080p -> 4K packed u8 up-scaling (tools/bench_upscale.py --only 0); run on the GPU box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
RP="rocprofv3 --kernel-trace --output-format csv"
CMD="python tools/bench_upscale.py --only 0 --iters 50"
timeout -k 5 200 $RP --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/p1 -o p -- $CMD > /dev/null 2>&1
python tools/prof_summary.py pmc /tmp/p1/p_counter_collection.csv cvgs:: > gpurun_out/upscale_pmc_sq1.txt 2>&1
timeout -k 5 200 $RP --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY -d /tmp/p2 -o p -- $CMD > /dev/null 2>&1
python tools/prof_summary.py pmc /tmp/p2/p_counter_collection.csv cvgs:: > gpurun_out/upscale_pmc_sq2.txt 2>&1
timeout -k 5 200 $RP --pmc FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE -d /tmp/p3 -o p -- $CMD > /dev/null 2>&1
python tools/prof_summary.py pmc /tmp/p3/p_counter_collection.csv cvgs:: > gpurun_out/upscale_pmc_hbm.txt 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o p --output-format csv -- $CMD > /dev/null 2>&1
python tools/prof_summary.py calls /tmp/p4/p_kernel_trace.csv cvgs:: > gpurun_out/upscale_trace.txt 2>&1
cat gpurun_out/upscale_pmc_sq1.txt gpurun_out/upscale_pmc_sq2.txt gpurun_out/upscale_pmc_hbm.txt gpurun_out/upscale_trace.txt | cut -c1-220
