"""Tower forward+backward time with the fused glue kernels vs ATen's (batch 4096, ResNet 9x256 + SE, bf16 autocast,
channels_last) -- the model configuration of bench.py's PPO update."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import shogidrl_amd.models as models

torch.backends.cudnn.benchmark = os.environ.get("KSH_CUDNN_BENCHMARK", "0") == "1"     # MIOpen exhaustive find

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
only = sys.argv[3] if len(sys.argv) > 3 else ""        # "fused" / "aten": run one variant (profiling)
for fused in ((True,) if only == "fused" else (False,) if only == "aten" else (False, True)):
    models.FUSED_TOWER_GLUE = fused
    torch.manual_seed(0)
    m = models.ResTowerActorCritic().to("cuda").to(memory_format=torch.channels_last)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
    x = (torch.rand(B, 46, 9, 9, device="cuda") < 0.1).float().contiguous(memory_format=torch.channels_last)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits, value = m(x)
        loss = logits.float().pow(2).mean() + value.float().pow(2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("fused_glue=%s  batch %d  fwd+bwd+adam %.2f ms  (%.0f samples/s)" % (fused, B, ms, B / ms * 1e3), flush=True)
