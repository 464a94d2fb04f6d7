# SQ counters + kernel trace of any command's cvgs kernels; run on the GPU box:  bash tools/pmc_cmd.sh <tag> <command...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
RP="rocprofv3 --kernel-trace --output-format csv"
rm -rf /tmp/pc1 /tmp/pc2 /tmp/pc4
timeout -k 5 200 $RP --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/pc1 -o p -- "$@" > /dev/null 2>&1
python tools/prof_summary.py pmc /tmp/pc1/p_counter_collection.csv cvgs:: > gpurun_out/${TAG}_pmc_sq1.txt 2>&1
timeout -k 5 200 $RP --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY -d /tmp/pc2 -o p -- "$@" > /dev/null 2>&1
python tools/prof_summary.py pmc /tmp/pc2/p_counter_collection.csv cvgs:: > gpurun_out/${TAG}_pmc_sq2.txt 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/pc4 -o p --output-format csv -- "$@" > /dev/null 2>&1
python tools/prof_summary.py kernels /tmp/pc4/p_kernel_trace.csv > gpurun_out/${TAG}_trace_kernels.txt 2>&1
cat gpurun_out/${TAG}_pmc_sq1.txt gpurun_out/${TAG}_pmc_sq2.txt gpurun_out/${TAG}_trace_kernels.txt | cut -c1-200
