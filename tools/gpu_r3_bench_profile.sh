# Round-3 recipe: the default bench (supervisor + worker + side process), then the worker alone under rocprofv3
# (`--worker`: the program itself behind `--`, no launcher hop), kernel stats + the ksh_* rows of the trace.
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err || { tail -12 gpurun_out/r3_bench.err; exit 1; }
grep "bench" gpurun_out/r3_bench.err | tail -12; head -c 300 gpurun_out/r3_bench.json; echo
cd /tmp && export TMPDIR=/tmp && mkdir -p /tmp/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $R/bench.py --worker --no-cpu-baseline > $R/gpurun_out/r3_bench_under_rocprof.json 2> $R/gpurun_out/r3_bench_under_rocprof.err
cp /tmp/prof/*/*_kernel_stats.csv $R/gpurun_out/r3_bench_kernel_stats.csv
python3 $R/tools/trace_rows.py $(ls /tmp/prof/*/*_kernel_trace.csv | head -1) ksh_ 96 > $R/gpurun_out/r3_bench_ksh_rows.txt
grep -E "env_step" $R/gpurun_out/r3_bench_ksh_rows.txt | cut -c1-150
