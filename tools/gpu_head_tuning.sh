# Recipe: A-B runs of the fused loss forward + backward (tools/headdiag.py) over alternative builds of the head kernels in
# build/diag/ (made in the container by substituting one constant each; see profiles/r02_head_kernels_pmc.txt).
R=$GRAFT_REPO_ROOT
cd $R
if [ -n "$WITH_TESTS" ]; then timeout -k 10 300 python -m pytest tests/test_gpu_head.py -x -q 2>&1 | tail -3 || exit 1; fi
python tools/headdiag.py 2>/dev/null | tail -1
for v in $VARIANTS; do
    SHOGIDRL_AMD_LIB=$R/build/diag/lib_$v.so python tools/headdiag.py 2>/dev/null | tail -1 || exit 1
done
python tools/headdiag.py 2>/dev/null | tail -1
