# Diagnostic recipe: dense-path GEMM A/B builds (tile shapes / split factors overridden with -D through tools/build_variant.sh) timed
# by tools/headbench.py under rocprofv3.  usage: bash tools/gpu_gemm_variants.sh "<names>"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in $1; do
  rm -rf /tmp/prof_g
  SHOGIDRL_AMD_LIB=$R/build/variants/libksh_$n.so timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -- python3 $R/tools/headbench.py > /tmp/hb_$n.json 2>/dev/null || { echo "$n failed"; continue; }
  echo "== $n"
  python3 $R/tools/trace_rows.py $(ls /tmp/prof_g/*/*_kernel_trace.csv | head -1) ksh_head 5 | grep "last" | grep -E "gemm|splitk" | awk '$NF=="us" && $(NF-1) > 6 {print "   ", $1, $(NF-1)}'
done
