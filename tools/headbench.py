#!/usr/bin/env python3
"""Micro-benchmark of the fused policy head against the path it replaces, on masks from real self-play.
usage: python tools/headbench.py [rows=4096] [K=162]
Prints hipEvent averages (back-to-back launches on one stream): fused sampler vs (hipBLASLt logits + masked_sample),
fused loss fwd+bwd vs (logits GEMM + ppo_loss kernels + the two backward GEMMs), and the all-dense MFMA path."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from shogidrl_amd import ops
from shogidrl_amd.env import DeviceTimer, VecShogiEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 162
A, dev = 13527, "cuda:0"
env = VecShogiEnv(N, 500, device=dev)
env.reset()
zeros = torch.zeros(N, A, device=dev)
for t in range(48):
    a, _ = ops.masked_sample(zeros, env.legal_bits, 1, t)
    env.step(a)
del zeros
bits, mask = env.legal_bits.clone(), env.legal_mask.clone()
torch.manual_seed(0)
feat = torch.randn(N, K, device=dev)
W = torch.randn(A, K, device=dev) * 0.1
b = torch.randn(A, device=dev) * 0.1
values, adv, ret = torch.randn(N, device=dev), torch.randn(N, device=dev), torch.randn(N, device=dev)
act, lp = ops.policy_head_sample(feat, W, b, bits, 7, 0)
tm = DeviceTimer()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps * 1e3


def fused_loss(dense=False, m=bits):
    f, w, bb, v = (t.detach().requires_grad_(True) for t in (feat, W, b, values))
    loss, _ = ops.policy_head_ppo_loss(f, w, bb, v, m, act, lp, adv, ret, None, 0.2, 0.5, 0.01, False, dense_rows_possible=dense)
    loss.backward()


def logits_loss(m=bits):
    f, w, bb, v = (t.detach().requires_grad_(True) for t in (feat, W, b, values))
    loss, _ = ops.ppo_loss(torch.addmm(bb, f, w.t()), v, m, act, lp, adv, ret, None, 0.2, 0.5, 0.01, False)
    loss.backward()


cnt = mask.sum(0)                                                   # samples per action
seg = mask.view(-1, 256, A).any(1).sum(0) if N % 256 == 0 else None  # non-empty 256-sample segments per action
out = {
    "pairs": int(cnt.sum()), "active_actions": int((cnt > 0).sum()), "max_samples_per_action": int(cnt.max()),
    "work_items_256": None if seg is None else int(seg.sum()),
    "rows": N, "K": K, "mean_legal": float(env.n_legal.float().mean()), "max_legal": int(env.n_legal.max()),
    "fused_sample_us": timeit(lambda: ops.policy_head_sample(feat, W, b, bits, 7, 1)),
    "logits_gemm_plus_sample_bits_us": timeit(lambda: ops.masked_sample(torch.addmm(b, feat, W.t()), bits, 7, 1)),
    "logits_gemm_plus_sample_bytes_us": timeit(lambda: ops.masked_sample(torch.addmm(b, feat, W.t()), mask, 7, 1)),
    "sample_kernel_bits_us": None, "sample_kernel_bytes_us": None,
    "fused_loss_fwd_bwd_us": timeit(fused_loss),
    "fused_loss_fwd_bwd_with_dense_launches_us": timeit(lambda: fused_loss(True)),
    "logits_loss_fwd_bwd_us": timeit(logits_loss),
    "fused_evaluate_us": timeit(lambda: ops.policy_head_evaluate(feat, W, b, bits, act, False)),
}
lg = torch.addmm(b, feat, W.t())
out["sample_kernel_bits_us"] = timeit(lambda: ops.masked_sample(lg, bits, 7, 1))
out["sample_kernel_bytes_us"] = timeit(lambda: ops.masked_sample(lg, mask, 7, 1))
del lg
# the dense (MFMA) path with every row dense, at a size whose scratch stays small
Bd = min(N, 1024)
dm = torch.ones(Bd, A, dtype=torch.bool, device=dev)
dbits = ops.pack_mask_bits(dm)
fd, vd = feat[:Bd].contiguous(), values[:Bd].contiguous()
ad = torch.randint(0, A, (Bd,), device=dev)


def dense_loss():
    f, w, bb, v = (t.detach().requires_grad_(True) for t in (fd, W, b, vd))
    loss, _ = ops.policy_head_ppo_loss(f, w, bb, v, dbits, ad, lp[:Bd], adv[:Bd], ret[:Bd], None, 0.2, 0.5, 0.01, False)
    loss.backward()


us = timeit(dense_loss, reps=5)
out["all_dense_rows"] = Bd
out["all_dense_loss_fwd_bwd_us"] = us
out["all_dense_gemm_flop"] = 3 * 2.0 * Bd * A * K
print(json.dumps(out))
