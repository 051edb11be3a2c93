# Recipe run on the GPU box through `gpurun -- bash tools/gpu_pmc_head_kernels.sh` (writes under gpurun_out/; summaries are then copied into profiles/).
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_head_$i -- python3 $R/tools/headdiag.py > /dev/null 2>&1
  for k in ksh_head_loss_kernel ksh_head_dw_work_kernel; do echo "== $k"; python3 $R/tools/pmc_rows.py $(find $R/gpurun_out/pmc_head_$i -name "*counter_collection.csv" | head -1) $k 50; done
done > $R/gpurun_out/r2_head_pmc.txt 2>&1
cat $R/gpurun_out/r2_head_pmc.txt
