"""End-to-end learning sanity check through SURVEY 8 row f2: does the ASSEMBLED path -- env kernel ->
fused sampler -> rollout buffer -> GAE kernel -> fused PPO loss -> Adam, driven by SelfPlayRunner -- change a policy in a
way an evaluation match can see?  Trains K self-play iterations from random init, then plays the trained and the untrained
network against UniformRandomPolicy with `evaluate_match` (the batched counterpart of
keisei/evaluation/strategies/single_opponent.py:170-220), colours balanced.

What the reference's reward structure can and cannot teach: +1 goes to the mover who mates, nobody
is ever punished, and both colours' transitions share one GAE stream without a sign flip -- so V(s) estimates "a mate is
near", not "I am winning", and the side that walks INTO the mate is reinforced too.  Self-play under these rules learns to
END games; against a random opponent that shows up as more decided games (fewer max-move draws) and more wins.

usage: learn_sanity.py [--iters K] [--games N] [--rollout T] [--depth D --width W | --tiny] [--epochs E] [--minibatch M]
                       [--lr LR] [--recompute] [--eval-games G] [--max-moves MM] [--seed S]
Prints one line per iteration and ONE JSON line at the end."""
import argparse
import copy
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shogidrl_amd import graphs  # noqa: E402

graphs.enable_safe_replay()                 # the rollout is replayed from a hipGraph (before the first HIP call)
import torch  # noqa: E402

from shogidrl_amd.agent import PPOConfig, VecPPOAgent  # noqa: E402
from shogidrl_amd.buffer import VecExperienceBuffer  # noqa: E402
from shogidrl_amd.env import VecShogiEnv  # noqa: E402
from shogidrl_amd.evaluation import UniformRandomPolicy, evaluate_match  # noqa: E402
from shogidrl_amd.models import ResTowerActorCritic, TinyActorCritic  # noqa: E402
from shogidrl_amd.rollout import SelfPlayRunner  # noqa: E402


def run(a, log=print):
    dev = "cuda:0"
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(a.seed)
    model = TinyActorCritic() if a.tiny else ResTowerActorCritic(46, 13527, a.depth, a.width, 0.25)
    cfg = PPOConfig(ppo_epochs=a.epochs, minibatch_size=a.minibatch, mixed_precision=not a.tiny, channels_last=not a.tiny,
                    seed=a.seed, learning_rate=a.lr, recompute_old_log_probs=a.recompute, entropy_coef=a.entropy)
    agent = VecPPOAgent(model, cfg, dev)
    untrained = VecPPOAgent(copy.deepcopy(agent.model), cfg, dev)

    def match(who, seed):
        r = {}
        for name, det in (("sampled", False), ("argmax", True)):
            m = evaluate_match(who, UniformRandomPolicy(seed), a.eval_games, a.max_moves, dev, deterministic=det)
            r[name] = {"win_rate": m.win_rate, "wins": m.agent_wins, "losses": m.opponent_wins, "draws": m.draws,
                       "avg_game_length": round(m.avg_game_length, 1), "reasons": m.reasons}
        return r

    t0 = time.time()
    before = match(untrained, 777)
    log("untrained vs uniform-random: %s" % json.dumps(before))
    env = VecShogiEnv(a.games, a.max_moves, device=dev)
    buf = VecExperienceBuffer(a.rollout, a.games, cfg.gamma, cfg.lambda_gae, dev)
    runner = SelfPlayRunner(env, agent, buf)
    prev_eps, hist = 0, []
    for it in range(a.iters):
        runner.collect_graphed()
        m = runner.update()
        st = runner.episode_stats()
        eps = st["total_episodes_completed"] - prev_eps
        prev_eps = st["total_episodes_completed"]
        decided = st["black_wins"] + st["white_wins"]
        hist.append({"iter": it, "episodes": eps, "kl": m["ppo/kl_divergence_approx"], "clip": m["ppo/clip_fraction"],
                     "entropy": -m["ppo/entropy"], "value_loss": m["ppo/value_loss"]})
        if it % a.log_every == 0 or it == a.iters - 1:
            log("iter %3d t=%6.1fs steps %9d policy %+.4f value %.4f entropy %.3f kl %.4f clip %.3f | episodes this iter %d "
                "(total %d, decided %d)" % (it, time.time() - t0, runner.global_timestep, m["ppo/policy_loss"], m["ppo/value_loss"],
                                            -m["ppo/entropy"], m["ppo/kl_divergence_approx"], m["ppo/clip_fraction"], eps,
                                            st["total_episodes_completed"], decided))
        assert all(v == v for v in m.values()), "non-finite metric"
    after = match(agent, 777)
    log("trained   vs uniform-random: %s" % json.dumps(after))
    k = max(1, a.iters // 5)
    out = {"config": vars(a), "seconds": round(time.time() - t0, 1), "env_steps": runner.global_timestep,
           "untrained": before, "trained": after,
           "episodes_per_iter_first_fifth": sum(h["episodes"] for h in hist[:k]) / k,
           "episodes_per_iter_last_fifth": sum(h["episodes"] for h in hist[-k:]) / k,
           "entropy_first": hist[0]["entropy"], "entropy_last": hist[-1]["entropy"]}
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--games", type=int, default=256)
    ap.add_argument("--rollout", type=int, default=64)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--minibatch", type=int, default=2048)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--entropy", type=float, default=0.01)
    ap.add_argument("--recompute", action="store_true")
    ap.add_argument("--eval-games", type=int, default=512)
    ap.add_argument("--max-moves", type=int, default=500)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--log-every", type=int, default=5)
    return ap.parse_args(argv)


if __name__ == "__main__":
    res = run(parse())
    print(json.dumps(res), flush=True)
