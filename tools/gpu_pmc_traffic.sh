# Recipe run on the GPU box through `gpurun -- bash tools/gpu_pmc_traffic.sh` (writes under gpurun_out/; summaries are then copied into profiles/).
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch2 -- python3 $R/bench.py --roofline-only > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write2 -- python3 $R/bench.py --roofline-only > /dev/null 2>&1 &&
python3 $R/tools/env_traffic.py $R/gpurun_out/pmc_fetch2 $R/gpurun_out/pmc_write2 4096 > $R/gpurun_out/env_kernel_hbm_traffic.json && head -12 $R/gpurun_out/env_kernel_hbm_traffic.json
cd $R && timeout -k 10 300 python -m pytest tests/test_gpu_env.py -x -q -k "table_modes" 2>&1 | tail -2
