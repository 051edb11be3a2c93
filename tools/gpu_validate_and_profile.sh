# Recipe run on the GPU box through `gpurun -- bash tools/gpu_validate_and_profile.sh` (writes under gpurun_out/; summaries are then copied into profiles/).
R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gputest_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_gputest_final.log; tail -3 gpurun_out/r2_gputest_final.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1; tail -1 gpurun_out/r2_smoke.log
timeout -k 10 400 python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; tail -1 gpurun_out/r2_bench_final.err; head -c 400 gpurun_out/r2_bench_final.json; echo
cd /tmp && export TMPDIR=/tmp && mkdir -p /tmp/prof && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r2_bench_under_rocprof.json 2> $R/gpurun_out/r2_bench_under_rocprof.err
cp /tmp/prof/*/*_kernel_stats.csv $R/gpurun_out/r2_bench_kernel_stats.csv
python3 $R/tools/trace_rows.py $(ls /tmp/prof/*/*_kernel_trace.csv | head -1) ksh_ 96 > $R/gpurun_out/r2_bench_ksh_rows.txt; grep -E "env_step|head_|gae|stats" $R/gpurun_out/r2_bench_ksh_rows.txt | cut -c1-130
