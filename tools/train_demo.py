"""End-to-end sanity run of the whole path at the bench configuration: 4096 self-play games, ResNet 9x256+SE (bf16
autocast, fused tower glue), T = 32, 10 PPO epochs x minibatch 4096 -- a dozen iterations with the metrics the
reference logs (ppo/* keys of PPOAgent.learn, episode statistics of StepManager).  Random init, synthetic self-play."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from shogidrl_amd import graphs

graphs.enable_safe_replay()                 # the rollout is replayed from a hipGraph (before the first HIP call)
from shogidrl_amd.agent import PPOConfig, VecPPOAgent
from shogidrl_amd.buffer import VecExperienceBuffer
from shogidrl_amd.env import VecShogiEnv
from shogidrl_amd.models import ResTowerActorCritic
from shogidrl_amd.rollout import SelfPlayRunner

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
# "reference": old log-probs as stored while acting (the reference's semantics); "recompute" (PPOConfig.recompute_old_log_probs=True, this tool's default):
# re-computed under learn-time statistics at the start of learn() (deviation 6)
mode = sys.argv[2] if len(sys.argv) > 2 else "recompute"
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 10
lr = float(sys.argv[4]) if len(sys.argv) > 4 else 3e-4     # 0: the weights never move, so kl / clip_frac ARE the old-vs-new mismatch
torch.backends.cudnn.benchmark = True
torch.manual_seed(42)
cfg = PPOConfig(ppo_epochs=epochs, minibatch_size=4096, mixed_precision=True, channels_last=True, seed=42,
                recompute_old_log_probs=(mode != "reference"), learning_rate=lr)
print("old log-probs: %s; ppo_epochs %d" % (mode, epochs), flush=True)
agent = VecPPOAgent(ResTowerActorCritic(46, 13527, 9, 256, 0.25), cfg, "cuda:0")
env = VecShogiEnv(4096, 500, device="cuda:0")
buf = VecExperienceBuffer(32, 4096, cfg.gamma, cfg.lambda_gae, "cuda:0")
run = SelfPlayRunner(env, agent, buf)
t0 = time.time()
for it in range(iters):
    run.collect_graphed()
    m = run.update()
    st = run.episode_stats()
    print("iter %2d  t=%6.1fs  steps %8d  policy_loss %+.4f  value_loss %.4f  entropy %+.3f  kl %.4f  clip_frac %.3f  "
          "grad_norm %.3f | episodes %d (black %d / white %d / draws %d)"
          % (it, time.time() - t0, run.global_timestep, m["ppo/policy_loss"], m["ppo/value_loss"], m["ppo/entropy"],
             m["ppo/kl_divergence_approx"], m["ppo/clip_fraction"], agent.last_gradient_norm, st["total_episodes_completed"],
             st["black_wins"], st["white_wins"], st["draws"]), flush=True)
    assert all(v == v and abs(v) < 1e6 for v in m.values()), "non-finite metric"
print("ok: %d iterations, %.1f k env-steps/s overall" % (iters, run.global_timestep / (time.time() - t0) / 1e3))
