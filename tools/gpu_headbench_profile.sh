# Recipe: tools/headbench.py under rocprofv3 --kernel-trace --stats; per-kernel rows of the head kernels (-> profiles/r02_headbench*).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_head && mkdir -p /tmp/prof_head
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_head -- python3 $R/tools/headbench.py > $R/gpurun_out/r2_headbench_prof7.json 2> $R/gpurun_out/r2_headbench_prof7.err || { tail -5 $R/gpurun_out/r2_headbench_prof7.err; exit 1; }
cp /tmp/prof_head/*/*_kernel_stats.csv $R/gpurun_out/r2_headbench_prof7_kernel_stats.csv
python3 $R/tools/trace_rows.py $(ls /tmp/prof_head/*/*_kernel_trace.csv | head -1) ksh_ 24 > $R/gpurun_out/r2_headbench_prof7_rows.txt
grep -E "head_|stats" $R/gpurun_out/r2_headbench_prof7_rows.txt | cut -c1-130
