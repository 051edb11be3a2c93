"""Scratch micro-benchmark of the env kernel (random legal self-play). Superseded by bench.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
import time

import torch

from shogidrl_amd import ops
from shogidrl_amd.env import DeviceTimer, VecShogiEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
variant = sys.argv[3] if len(sys.argv) > 3 else ""        # "", "packed", "obs16", "compact" (= packed + obs16)
packed = variant in ("packed", "compact")
obs16 = variant in ("obs16", "compact")
env = VecShogiEnv(N, 500, packed_masks=packed, obs_format="bf16_nhwc" if obs16 else "f32")
env.reset()
logits = torch.zeros(N, 13527, device="cuda:0")
acts = []
# pre-generate nothing: actions depend on state; time env kernel alone with events around each launch
tm = DeviceTimer()
tot_env = 0.0
for t in range(64):
    a, _ = ops.masked_sample(logits, (env.legal_bits if packed else env.legal_mask), 42, t)
    env.step(a)
torch.cuda.synchronize()
t0 = time.time()
for t in range(steps):
    a, _ = ops.masked_sample(logits, (env.legal_bits if packed else env.legal_mask), 42, 64 + t)
    tm.start()
    env.step(a, validate=False)
    tm.stop()
    tot_env += tm.elapsed_ms()
torch.cuda.synchronize()
wall = time.time() - t0
ms = tot_env / steps
bytes_per = (7452 if obs16 else 14904) + (1696 if packed else 13527) + 279
print((variant + " " if variant else "") + "N=%d env kernel %.3f ms/step -> %.1f M env-steps/s ; %.1f GB/s of the written bytes/step; loop wall %.3f ms/step" %
      (N, ms, N / ms / 1e3, N * bytes_per / ms / 1e6, wall / steps * 1e3))
