"""Scratch micro-benchmark of the learner-side kernels on realistic inputs (masks from real self-play)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
import torch
from shogidrl_amd import ops
from shogidrl_amd.env import DeviceTimer, VecShogiEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda:0"
env = VecShogiEnv(N, 500)
env.reset()
z = torch.zeros(N, 13527, device=dev)
for t in range(80):
    a, _ = ops.masked_sample(z, env.legal_mask, 1, t)
    env.step(a, validate=False)
mask = env.legal_mask.clone()
nleg = float(env.n_legal.float().mean())
tm = DeviceTimer()

def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    tm.start()
    for _ in range(reps): fn()
    tm.stop()
    return tm.elapsed_ms() / reps * 1e3

for dt, name, sz in ((torch.float32, "f32", 4), (torch.bfloat16, "bf16", 2)):
    logits = torch.randn(N, 13527, device=dev).to(dt)
    us = timeit(lambda: ops.masked_sample(logits, mask, 7, 3))
    alg = N * (13527 + nleg * 64 + 16)
    print("masked_sample %s: %.1f us  (%.0f rows/us) algorithmic %.1f MB -> %.0f GB/s" % (name, us, N / us, alg / 1e6, alg / us / 1e3))
    lg = logits.clone().requires_grad_(True)
    v = torch.randn(N, device=dev, requires_grad=True)
    a, lp = ops.masked_sample(logits, mask, 7, 3)
    adv = torch.randn(N, device=dev); ret = torch.randn(N, device=dev)
    def loss_fwd():
        ops.ppo_loss(lg, v, mask, a, lp, adv, ret)
    us = timeit(loss_fwd)
    alg = N * (13527 + 13527 * sz + nleg * 64 + 40)
    print("ppo_loss fwd+bwd %s: %.1f us  algorithmic %.1f MB -> %.0f GB/s" % (name, us, alg / 1e6, alg / us / 1e3))
T = 32
r = torch.randn(T, N, device=dev); vv = torch.randn(T, N, device=dev); d = torch.rand(T, N, device=dev) < 0.01; lv = torch.randn(N, device=dev)
us = timeit(lambda: ops.gae(r, vv, d, lv, 0.99, 0.95))
print("gae T=%d N=%d: %.1f us  (%.1f MB -> %.0f GB/s)" % (T, N, us, T * N * 17 / 1e6, T * N * 17 / us / 1e3))
print("mean n_legal %.1f" % nleg)
