# Diagnostic: MFMA-pipe counters of the dense-path GEMM kernels (tools/headbench.py under rocprofv3 --pmc)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/pmc_g
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_g -- python3 $R/tools/headbench.py > /dev/null 2>&1 || { echo "pmc pass failed: $set"; continue; }
  f=$(find /tmp/pmc_g -name "*counter_collection.csv" | head -1)
  for k in gemm_logits gemm_dfeat gemm_dw; do echo "== $k"; python3 $R/tools/pmc_rows.py $f $k 5; done
done
