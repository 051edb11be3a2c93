# Recipe run on the GPU box through `gpurun -- bash tools/gpu_pmc_env_kernel.sh` (writes under gpurun_out/; summaries are then copied into profiles/).
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --roofline-only > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --roofline-only > /dev/null 2>&1 &&
python3 $R/tools/env_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write 4096 > $R/gpurun_out/env_kernel_hbm_traffic.json && cat $R/gpurun_out/env_kernel_hbm_traffic.json | head -8 &&
for m in lds global; do export KSH_ENV_TABLES=$m
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_${m}_a -- python3 $R/bench.py --roofline-only > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_${m}_b -- python3 $R/bench.py --roofline-only > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${m}_c -- python3 $R/bench.py --roofline-only > /dev/null 2>&1 &&
  for p in a b c; do python3 $R/tools/pmc_rows.py $(find $R/gpurun_out/pmc_${m}_$p -name "*counter_collection.csv" | head -1) ksh_env_step 96; done > $R/gpurun_out/r2_env_pmc_$m.csv; done
unset KSH_ENV_TABLES
cd $R && for n in 16384 32768; do for m in lds global; do echo "games $n tables $m"; KSH_ENV_TABLES=$m python bench.py --roofline-only --games $n 2>/dev/null | cut -c1-200; done; done > gpurun_out/r2_env_tables2.log; cat gpurun_out/r2_env_tables2.log; cat gpurun_out/r2_env_pmc_lds.csv; echo ---; cat gpurun_out/r2_env_pmc_global.csv
