"""Times the forward-only calls of the fused policy head (evaluate, sampler) on 4096 rows of self-play masks with hipEvents;
SHOGIDRL_AMD_LIB selects an alternative build (A-B runs)."""
import os, sys
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
act, lp = ops.policy_head_sample(feat, W, b, bits, 7, 0)
tm = DeviceTimer()
for name, fn in (("evaluate", lambda: ops.policy_head_evaluate(feat, W, b, bits, act, False)),
                 ("sample", lambda: ops.policy_head_sample(feat, W, b, bits, 7, 0))):
    for _ in range(5): fn()
    torch.cuda.synchronize(); tm.start()
    for _ in range(50): fn()
    tm.stop()
    print(os.environ.get("SHOGIDRL_AMD_LIB", "product"), "%s: %.1f us" % (name, tm.elapsed_ms() / 50 * 1e3))
