"""Scratch: throughput of the PyTorch tower (fwd+bwd) under a few layouts. Not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, time, os
import torch
from shogidrl_amd.models import ResTowerActorCritic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["nchw_bf16", "nhwc_bf16", "nchw_fp32"]
dev = "cuda:0"
import torch.nn.functional as F
class NativeBN(torch.nn.BatchNorm2d):
    def forward(self, x):
        with torch.backends.cudnn.flags(enabled=False):
            return super().forward(x)
def swap_bn(mod):
    for name, ch in mod.named_children():
        if isinstance(ch, torch.nn.BatchNorm2d):
            nb = NativeBN(ch.num_features).to(ch.weight.device)
            nb.load_state_dict(ch.state_dict())
            setattr(mod, name, nb)
        else:
            swap_bn(ch)
for v in variants:
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = "bench" in v
    m = ResTowerActorCritic().to(dev)
    cl = v.startswith("nhwc")
    if cl:
        m = m.to(memory_format=torch.channels_last)
    amp = "bf16" in v
    if "nativebn" in v:
        swap_bn(m)
    x = torch.randn(B, 46, 9, 9, device=dev)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    opt = torch.optim.SGD(m.parameters(), lr=1e-5)
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            lg, val = m(x)
        loss = lg.float().square().mean() + val.float().square().mean()
        opt.zero_grad(set_to_none=False)
        loss.backward()
    t0 = time.time()
    step(); torch.cuda.synchronize()
    print(v, "first step %.2f s" % (time.time() - t0), flush=True)
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.time(); n = 5
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print("%s B=%d fwd+bwd %.1f ms -> %.0f samples/s, %.1f TFLOP/s (5.2 GF/sample)" % (v, B, dt * 1e3, B / dt, B / dt * 5.2e-3), flush=True)
    with torch.no_grad():
        t0 = time.time()
        for _ in range(n):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                m(x)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / n
    print("%s B=%d fwd only %.1f ms -> %.0f samples/s" % (v, B, dt * 1e3, B / dt), flush=True)
