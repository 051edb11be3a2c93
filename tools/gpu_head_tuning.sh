# Recipe: A-B runs of the fused loss forward + backward (tools/headdiag.py) over alternative builds of the head kernels in
# build/diag/ (made in the container by substituting one constant each; see profiles/r02_head_kernels_pmc.txt).
R=$GRAFT_REPO_ROOT
cd $R
true
python tools/headdiag.py 2>/dev/null | tail -1
for v in ${VARIANTS:-accna3 accna2}; do
    SHOGIDRL_AMD_LIB=$R/build/diag/lib_$v.so python tools/headdiag.py 2>/dev/null | tail -1 || exit 1
done
python tools/headdiag.py 2>/dev/null | tail -1
