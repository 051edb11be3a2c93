"""TEST INFRASTRUCTURE: a stand-in for bench.py's rank body that needs no GPU -- every rank joins the process group
(gloo), the ranks all-reduce their rank numbers, rank 0 prints ONE JSON line.  `fail` makes rank 1 exit with code 3
while rank 0 would wait in a barrier forever (the launcher has to stop it)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch
import torch.distributed as dist

from shogidrl_amd import parallel

mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
rank, local_rank, world = parallel.init_from_env(backend="gloo")
if mode == "fail" and rank == 1:
    sys.exit(3)
if mode == "fail":
    time.sleep(600)          # the launcher must not wait for this
t = torch.tensor([float(rank)])
dist.all_reduce(t)
print("rank %d chatter on stdout" % rank if rank else json.dumps(
    {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "sum_of_ranks": float(t),
     "launched": os.environ.get("KSH_LAUNCHED")}), flush=True)
dist.barrier()
dist.destroy_process_group()
