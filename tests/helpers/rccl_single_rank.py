"""TEST INFRASTRUCTURE: a ONE-rank process group over the "nccl" backend (RCCL on ROCm) on cuda:0 -- the only way to run
RCCL on a one-GPU box.  Issues every collective / dtype / reduce-op combination shogidrl_amd.parallel uses, at the sizes
the bench uses, and checks the results (a one-rank reduction is the identity).  Prints ONE JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = "cuda:0"
out = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
g = torch.Generator(device=dev).manual_seed(1)
flat = torch.randn(13_240_000, device=dev, generator=g)                 # the flat gradient bucket of the 9x256 tower (53 MB)
ref = flat.clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
dist.all_reduce(flat, op=dist.ReduceOp.SUM)                                # FlatGradBucket.all_reduce_mean
torch.cuda.synchronize()
out["allreduce_53MB_ms"] = (time.perf_counter() - t0) * 1e3
out["f32_sum_ok"] = bool(torch.equal(flat, ref))
m = torch.tensor([4096.0, 1.25, 7.5], dtype=torch.float64, device=dev)    # global_advantage_moments
dist.all_reduce(m, op=dist.ReduceOp.SUM)
out["f64_sum_ok"] = m.tolist() == [4096.0, 1.25, 7.5]
t = torch.tensor([131072, -131072], dtype=torch.int64, device=dev)        # check_equal_batches
dist.all_reduce(t, op=dist.ReduceOp.MAX)
out["i64_max_ok"] = t.tolist() == [131072, -131072]
e = torch.tensor([1.5, 2.5, 3.5], dtype=torch.float64, device=dev)        # bench.py: elapsed times, MAX over ranks
dist.all_reduce(e, op=dist.ReduceOp.MAX)
out["f64_max_ok"] = e.tolist() == [1.5, 2.5, 3.5]
w = torch.randn(13527, 162, device=dev, generator=g)
w0 = w.clone()
dist.broadcast(w, src=0)                                                   # broadcast_parameters (float32)
nb = torch.tensor(7, dtype=torch.int64, device=dev)
dist.broadcast(nb, src=0)                                                  # ... and BatchNorm's num_batches_tracked (int64)
out["broadcast_ok"] = bool(torch.equal(w, w0)) and int(nb) == 7
dist.barrier()
dist.destroy_process_group()
print(json.dumps(out), flush=True)
