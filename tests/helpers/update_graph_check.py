"""TEST INFRASTRUCTURE: PPOConfig.graph_minibatch_step in a process of its own -- the option needs
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the first HIP call, which a running pytest process cannot arrange.
Prints ONE JSON line.

  equal     small tower: three eager updates, one capture, replays from then on, across two learn() calls with fresh
            advantages, against the eager loop (parameters, buffers, metrics)
  lr        the captured fused-Adam node bakes the learning rate in: after `param_groups[0]["lr"]` changes, learn() must capture a
            new graph (and drop the old one) instead of replaying the old rate -- compared with the eager loop
  survive   bench-sized tower, minibatch 64: two learn() calls over the same rollout data, graph-replayed and eager.  The
            second learn() starts with 64 eager forward passes (the re-computed old log-probs: ~2 x 10^4 launches) -- with
            the runtime's default packet capture the replays after that returned garbage (|grad| = inf)."""
import copy
import dataclasses
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from shogidrl_amd.agent import PPOConfig, VecPPOAgent
from shogidrl_amd.buffer import VecExperienceBuffer
from shogidrl_amd.env import VecShogiEnv
from shogidrl_amd.models import ResTowerActorCritic
from shogidrl_amd.rollout import SelfPlayRunner

DEV, A = "cuda:0", 13527
KEYS = ("ppo/policy_loss", "ppo/value_loss", "ppo/entropy", "ppo/kl_divergence_approx", "ppo/clip_fraction")
mode = sys.argv[1]
torch.backends.cudnn.deterministic = mode == "equal"      # (bit-level comparison there; "survive" compares regimes)
out = {"mode": mode, "env": os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")}
if mode in ("equal", "lr"):
    N, T = 64, 4
    res = []
    for graphed in (False, True):
        torch.manual_seed(0)
        cfg = PPOConfig(ppo_epochs=2, minibatch_size=32, seed=4, graph_minibatch_step=graphed)
        agent = VecPPOAgent(ResTowerActorCritic(46, A, 2, 32, 0.25), cfg, DEV)
        run = SelfPlayRunner(VecShogiEnv(N, 40, device=DEV), agent, VecExperienceBuffer(T, N, cfg.gamma, cfg.lambda_gae, DEV))
        ms = []
        for it in range(2 if mode == "equal" else 3):
            if mode == "lr" and it == 1:
                for pg in agent.optimizer.param_groups:          # what an LR scheduler does between two learn() calls
                    pg["lr"] = 1e-5
            run.collect()
            ms.append(run.update())
        res.append((ms, torch.cat([p.detach().reshape(-1) for p in agent.model.parameters()]).clone(),
                    torch.cat([b.detach().reshape(-1).float() for b in agent.model.buffers()]).clone(), agent))
    (m0, p0, b0, _), (m1, p1, b1, ag) = res
    out["graphs"] = len(ag._step_graphs)
    out["captured"] = next(iter(ag._step_graphs.values()))["graph"] is not None
    out["param_diff"] = float((p0 - p1).abs().max())
    out["buffer_diff"] = float((b0 - b1).abs().max())
    out["metric_diff"] = max(abs(x[k] - y[k]) for x, y in zip(m0, m1) for k in KEYS)
else:
    torch.manual_seed(42)
    games, T, n = 1024, 4, 4096
    cfg = PPOConfig(ppo_epochs=1, minibatch_size=1024, mixed_precision=True, channels_last=True, seed=42)
    actor = VecPPOAgent(ResTowerActorCritic(46, A, 9, 256, 0.25), cfg, DEV)
    buf = VecExperienceBuffer(T, games, cfg.gamma, cfg.lambda_gae, DEV)
    SelfPlayRunner(VecShogiEnv(games, 500, device=DEV), actor, buf).collect()
    small = VecExperienceBuffer(1, n, buf.gamma, buf.lambda_gae, DEV)
    flat = lambda x: x.reshape(-1, *x.shape[2:])[:n]
    small.obs[0].copy_(flat(buf.obs[:buf.T]))
    small.legal_masks[0].copy_(flat(buf.legal_masks[:buf.T]))
    small.legal_bits[0].copy_(flat(buf.legal_bits[:buf.T]))
    for name in ("actions", "rewards", "log_probs", "values", "dones", "advantages", "returns"):
        getattr(small, name)[0].copy_(flat(getattr(buf, name)))
    small._advantages_computed = True
    for graphed in (False, True):
        gcfg = dataclasses.replace(cfg, minibatch_size=64, graph_minibatch_step=graphed, recompute_old_log_probs=True)
        ag = VecPPOAgent(copy.deepcopy(actor.model), gcfg, DEV)
        ms = [ag.learn(small) for _ in range(2)]
        torch.cuda.synchronize()
        out["graph" if graphed else "eager"] = {"metrics": [[m[k] for k in KEYS] for m in ms], "grad_norm": ag.last_gradient_norm,
                                                "params_finite": bool(all(torch.isfinite(p).all() for p in ag.model.parameters()))}
print(json.dumps(out), flush=True)
