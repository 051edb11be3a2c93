# Recipe run on the GPU box through `gpurun -- bash tools/gpu_validate_and_profile.sh` (writes under gpurun_out/; summaries are then copied into profiles/).
# Every GPU step only starts when the one before it succeeded (set -e): nothing else is launched after a fault or a timeout.
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2_gputest_final.log 2>&1 || { tail -30 gpurun_out/r2_gputest_final.log; exit 1; }
tail -3 gpurun_out/r2_gputest_final.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1 || { tail -20 gpurun_out/r2_smoke.log; exit 1; }
tail -1 gpurun_out/r2_smoke.log
# (run from a scratch directory: should the runtime ever report a memory fault, its GPU core dump lands there and rocgdb names the kernel)
mkdir -p /tmp/diag && cd /tmp/diag
timeout -k 10 400 python $R/bench.py > $R/gpurun_out/r2_bench_final.json 2> $R/gpurun_out/r2_bench_final.err || {
    tail -8 $R/gpurun_out/r2_bench_final.err
    if ls /tmp/diag/gpucore.* > /dev/null 2>&1; then
        timeout -k 10 300 /opt/rocm/bin/rocgdb -batch -ex "info agents" -ex "bt" -ex "x/6i \$pc" -ex "info threads" -c $(ls /tmp/diag/gpucore.* | head -1) > $R/gpurun_out/r2_bench_fault_rocgdb.txt 2>&1
        head -c 2500 $R/gpurun_out/r2_bench_fault_rocgdb.txt
    fi
    exit 1
}
cd $R
tail -1 gpurun_out/r2_bench_final.err; head -c 400 gpurun_out/r2_bench_final.json; echo
cd /tmp && export TMPDIR=/tmp && mkdir -p /tmp/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r2_bench_under_rocprof.json 2> $R/gpurun_out/r2_bench_under_rocprof.err
cp /tmp/prof/*/*_kernel_stats.csv $R/gpurun_out/r2_bench_kernel_stats.csv
python3 $R/tools/trace_rows.py $(ls /tmp/prof/*/*_kernel_trace.csv | head -1) ksh_ 96 > $R/gpurun_out/r2_bench_ksh_rows.txt
grep -E "env_step|head_|gae|stats" $R/gpurun_out/r2_bench_ksh_rows.txt | cut -c1-130
