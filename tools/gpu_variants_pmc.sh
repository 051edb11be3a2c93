# Diagnostic recipe: instruction counts (SQ_INSTS_*) and time of env-kernel cut builds.  usage: bash tools/gpu_variants_pmc.sh "<names>"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in $1; do
  export SHOGIDRL_AMD_LIB=$R/build/variants/libksh_$n.so
  rm -rf /tmp/pmc_$n
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/pmc_$n -- python3 $R/bench.py --roofline-only > /tmp/pmc_$n.json 2>/dev/null || { echo "$n failed"; exit 1; }
  echo "== $n"; python3 $R/tools/pmc_rows.py $(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1) ksh_env_step 96
  python3 $R/bench.py --roofline-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('time_us,%.2f' % d['avg_launch_us'])"
done
