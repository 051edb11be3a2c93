# Round-3 recipe: the learning sanity experiments behind tests/test_gpu_learning.py and DESIGN section 6.1
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 400 python tools/learn_sanity.py "$@" > gpurun_out/r3_learn_$name.log 2>&1 && tail -1 gpurun_out/r3_learn_$name.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); u,t=d['untrained'],d['trained']
print('$name', d['seconds'],'s', 'argmax win', u['argmax']['win_rate'],'->',t['argmax']['win_rate'],'sampled win',u['sampled']['win_rate'],'->',t['sampled']['win_rate'],'draws',u['argmax']['draws'],'->',t['argmax']['draws'],'eps/iter',d['episodes_per_iter_first_fifth'],'->',d['episodes_per_iter_last_fifth'],'entropy',d['entropy_first'],'->',d['entropy_last'])"; }
run D --iters 60 --games 1024 &&
run E --iters 60 --games 1024 --lr 1e-3 &&
run F --iters 60 --games 1024 --lr 1e-3 --recompute --seed 7 &&
run G --iters 120 --games 1024 --recompute
