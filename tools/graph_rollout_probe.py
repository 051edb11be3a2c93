"""Diagnostic: is a replayed ROLLOUT graph (SelfPlayRunner.collect_graphed) still bit-identical to eager rollouts after tens
of thousands of eager kernel launches on the same stream?  (The captured PPO-update graph is not: see DESIGN.md.)
usage: graph_rollout_probe.py depth width games [launches]
Run with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (what the package requires since round 3); KSH_PROBE_DEFAULT_PACKET_CAPTURE=1 (this
TOOL only) patches the guard to look at the runtime's default instead."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from shogidrl_amd import graphs
from shogidrl_amd.agent import PPOConfig, VecPPOAgent
from shogidrl_amd.buffer import VecExperienceBuffer
from shogidrl_amd.env import VecShogiEnv
from shogidrl_amd.models import ResTowerActorCritic
from shogidrl_amd.rollout import SelfPlayRunner

if os.environ.get("KSH_PROBE_DEFAULT_PACKET_CAPTURE") == "1":
    graphs.packet_capture_off = lambda: True
else:
    graphs.enable_safe_replay()
depth, width, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 30000
DEV, T = "cuda:0", 6
torch.backends.cudnn.deterministic = True


def make():
    torch.manual_seed(0)
    cfg = PPOConfig(seed=11, mixed_precision=True, channels_last=True)
    agent = VecPPOAgent(ResTowerActorCritic(46, 13527, depth, width, 0.25), cfg, DEV)
    return SelfPlayRunner(VecShogiEnv(N, 60, device=DEV), agent, VecExperienceBuffer(T, N, cfg.gamma, cfg.lambda_gae, DEV))


eager, graphed = make(), make()
bad = 0
for it in range(6):
    if it >= 2:
        x = torch.zeros(256, device=DEV)
        for k in range(launches):
            x.add_(1.0)
        del x
    eager.collect()
    graphed.collect_graphed()
    a, b = eager.buffer, graphed.buffer
    diff = [name for name in ("obs", "legal_masks", "actions", "rewards", "dones", "log_probs", "values", "advantages", "returns")
            if not torch.equal(getattr(a, name), getattr(b, name))]
    print("iteration %d (%s): %s" % (it, "after %d eager launches" % launches if it >= 2 else "no extra launches",
                                     "identical" if not diff else "DIFFERENT: " + ", ".join(diff)), flush=True)
    bad += bool(diff)
    eager.buffer.clear()
    graphed.buffer.clear()
print("rollout graph depth %d width %d games %d: %s" % (depth, width, N, "OK" if not bad else "CORRUPTED in %d iterations" % bad))
