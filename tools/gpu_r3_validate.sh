# Round-3 recipe (gpurun -- bash tools/gpu_r3_validate.sh): GPU test suite, smoke, default bench.  Each step starts only when
# the one before it succeeded; outputs under gpurun_out/.
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest.log 2>&1 || { tail -40 gpurun_out/r3_gputest.log; exit 1; }
tail -3 gpurun_out/r3_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1 || { tail -20 gpurun_out/r3_smoke.log; exit 1; }
tail -1 gpurun_out/r3_smoke.log
timeout -k 10 500 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err || { tail -12 gpurun_out/r3_bench.err; exit 1; }
grep "bench" gpurun_out/r3_bench.err | tail -25; head -c 600 gpurun_out/r3_bench.json; echo
