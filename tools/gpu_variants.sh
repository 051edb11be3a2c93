# Diagnostic recipe: env-kernel A/B builds (tools/build_variant.sh) timed with bench.py's roofline replay.
# usage (through gpurun): bash tools/gpu_variants.sh "<names>" "<sizes>"
R=$GRAFT_REPO_ROOT
cd $R
for n in $1; do
  for g in $2; do
    SHOGIDRL_AMD_LIB=$R/build/variants/libksh_$n.so KSH_ENV_TABLES=lds timeout -k 10 120 python bench.py --roofline-only --games $g 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-14s games %6d  %7.2f us  frac %.4f  deterministic %s' % ('$n', d['n_games'], d['avg_launch_us'], d['frac'], d['replay_deterministic']))" || exit 1
  done
done
