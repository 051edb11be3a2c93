# Round-3 recipe: GPU head tests, then tools/headbench.py under rocprofv3 --kernel-trace --stats; per-kernel rows of the head kernels.
R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 600 python -m pytest tests/test_gpu_head.py -x -q 2>&1 | tail -2 || exit 1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_head && mkdir -p /tmp/prof_head
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_head -- python3 $R/tools/headbench.py > $R/gpurun_out/r3_headbench.json 2> $R/gpurun_out/r3_headbench.err || { tail -5 $R/gpurun_out/r3_headbench.err; exit 1; }
cp /tmp/prof_head/*/*_kernel_stats.csv $R/gpurun_out/r3_headbench_kernel_stats.csv
python3 $R/tools/trace_rows.py $(ls /tmp/prof_head/*/*_kernel_trace.csv | head -1) ksh_ 5 > $R/gpurun_out/r3_headbench_rows.txt
grep -E "gemm|splitk|ksh_ppo_loss" $R/gpurun_out/r3_headbench_rows.txt | cut -c1-150
python3 -c "
import json; d=json.load(open('$R/gpurun_out/r3_headbench.json')); print({k:v for k,v in d.items() if 'dense' in k})"
