# Round-3 recipe: the long differential run of the rewritten env kernel against the oracle, in all three forms
# (LDS tables, global tables, compact formats).  Each run starts only when the one before it passed.
cd $GRAFT_REPO_ROOT
L=gpurun_out/r3_stress_vs_oracle.log
timeout -k 10 900 python tools/stress_vs_oracle.py 2>&1 | grep -v "^  " > $L && tail -1 $L &&
echo "--- KSH_ENV_TABLES=global" >> $L && KSH_ENV_TABLES=global timeout -k 10 900 python tools/stress_vs_oracle.py 2>&1 | grep -v "^  " >> $L && tail -1 $L &&
echo "--- compact formats (packed masks + bf16 channels-last observations)" >> $L && timeout -k 10 900 python tools/stress_vs_oracle.py compact 2>&1 | grep -v "^  " >> $L && tail -1 $L
