"""Times the fused policy head's loss forward + backward (all of its kernels, no autograd) on 4096 rows of self-play masks
with hipEvents; SHOGIDRL_AMD_LIB / KSH_HEAD_LIGHT_MAX select an alternative build / the light-heavy threshold (A-B runs)."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from shogidrl_amd import ops
from shogidrl_amd.env import DeviceTimer, VecShogiEnv
N, K, A, dev = 4096, 162, 13527, "cuda:0"
env = VecShogiEnv(N, 500, device=dev); env.reset()
zeros = torch.zeros(N, A, device=dev)
for t in range(48):
    a, _ = ops.masked_sample(zeros, env.legal_bits, 1, t); env.step(a)
bits = env.legal_bits.clone()
torch.manual_seed(0)
feat = torch.randn(N, K, device=dev); W = torch.randn(A, K, device=dev) * 0.1; b = torch.randn(A, device=dev) * 0.1
values, adv, ret = torch.randn(N, device=dev), torch.randn(N, device=dev), torch.randn(N, device=dev)
act, lp = ops.policy_head_sample(feat, W, b, bits, 7, 0)
tm = DeviceTimer()
def run():
    ops.policy_head_loss_grads(feat, W, b, values, bits, act, lp, adv, ret, None, 0.2, 0.5, 0.01, False, dense_rows_possible=False)
for _ in range(5): run()
torch.cuda.synchronize(); tm.start()
for _ in range(50): run()
tm.stop()
print(os.environ.get("SHOGIDRL_AMD_LIB", "product"), "fused loss fwd+bwd (all kernels, no autograd): %.1f us" % (tm.elapsed_ms() / 50 * 1e3))
