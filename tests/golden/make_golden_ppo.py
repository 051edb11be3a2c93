#!/usr/bin/env python3
"""Generate the GAE / masked-categorical / PPO-loss golden fixtures by running the *reference*
(Keisei) in THIS container.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_ppo.py

Outputs (data only):
    tests/golden/gae.npz        ExperienceBuffer.compute_advantages_and_returns on random streams
    tests/golden/ppo_loss.npz   PPOAgent.learn run for ONE minibatch on a stub model whose parameters
                                are the logits / values themselves: metrics + d(loss)/d(logits), d(loss)/d(values)
    tests/golden/sample.npz     BaseActorCriticModel.get_action_and_value log-probs for given actions
                                (+ deterministic arg-max actions)

Reference objects exercised: keisei.core.experience_buffer.ExperienceBuffer,
keisei.core.ppo_agent.PPOAgent.learn, keisei.core.base_actor_critic.BaseActorCriticModel.
"""
import os
import sys

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
A = 13527

import logging  # noqa: E402

logging.disable(logging.CRITICAL)
import torch  # noqa: E402

torch.set_num_threads(4)
from keisei.core.base_actor_critic import BaseActorCriticModel  # noqa: E402
from keisei.core.experience_buffer import ExperienceBuffer  # noqa: E402
from keisei.core.ppo_agent import PPOAgent  # noqa: E402
from keisei.utils import load_config  # noqa: E402


# ------------------------------------------------------------------------------------------- GAE
def make_gae():
    rng = np.random.default_rng(1234)
    out = {}
    cases = [(1, 0.99, 0.95, 0.0), (2, 0.99, 0.95, 0.5), (5, 0.99, 0.95, 0.3), (64, 0.9, 0.8, 0.1),
             (257, 0.997, 0.9, 0.02), (2048, 0.99, 0.95, 0.01)]
    for ci, (T, gamma, lam, pdone) in enumerate(cases):
        r = (rng.random(T) < 0.1).astype(np.float32) * rng.choice([1.0, -1.0, 0.5], T).astype(np.float32)
        v = rng.normal(0, 1, T).astype(np.float32)
        d = rng.random(T) < pdone
        if T >= 5:
            d[-1] = ci % 2 == 0
        last_v = float(np.float32(rng.normal()))
        buf = ExperienceBuffer(T, gamma, lam, "cpu")
        obs = torch.zeros(46, 9, 9)
        m = torch.zeros(A, dtype=torch.bool)
        for t in range(T):
            buf.add(obs, 0, float(r[t]), 0.0, float(v[t]), bool(d[t]), m)
        buf.compute_advantages_and_returns(last_v)
        out[f"c{ci}_r"] = r
        out[f"c{ci}_v"] = v
        out[f"c{ci}_d"] = d.astype(np.uint8)
        out[f"c{ci}_hp"] = np.asarray([gamma, lam, last_v], np.float64)
        out[f"c{ci}_adv"] = buf.advantages.numpy().copy()
        out[f"c{ci}_ret"] = buf.returns.numpy().copy()
    # the reference's own known answer: tests/core/test_experience_buffer.py:98-105
    out["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(HERE, "gae.npz"), **out)
    print("gae cases:", len(cases))


# ------------------------------------------------------------------------------------ PPO loss
class StubModel(BaseActorCriticModel):
    """forward() returns stored logits / values for the sample index carried in obs[:,0,0,0]."""

    def __init__(self, logits, values):
        super().__init__()
        self.logits = torch.nn.Parameter(torch.tensor(logits))
        self.values = torch.nn.Parameter(torch.tensor(values))

    def forward(self, x):
        idx = x[:, 0, 0, 0].round().long()
        return self.logits[idx], self.values[idx]


def random_case(rng, B, kind):
    logits = rng.normal(0, 1.5, (B, A)).astype(np.float32)
    mask = np.zeros((B, A), bool)
    actions = np.zeros(B, np.int64)
    for b in range(B):
        if kind == "dense":
            n = A
        elif kind == "one":
            n = 1
        else:
            n = int(rng.choice([1, 2, 7, 30, 53, 89, 206, 593]))
        ids = np.sort(rng.choice(A, n, replace=False))
        mask[b, ids] = True
        actions[b] = ids[rng.integers(n)]
        if kind == "peaked" and n > 1:
            logits[b, ids[0]] += 40.0          # p ~ 1 for one action, < eps for the others
            actions[b] = ids[rng.integers(min(2, n))]
    old_logp = np.zeros(B, np.float32)
    adv = rng.normal(0, 1, B).astype(np.float32)
    ret = rng.normal(0, 1, B).astype(np.float32)
    values = rng.normal(0, 1, B).astype(np.float32)
    old_values = (values + rng.normal(0, 0.3, B)).astype(np.float32)
    # old log-probs near the new ones so that both clipped and unclipped branches occur
    ml = np.where(mask, logits, -np.inf)
    lse = np.log(np.exp(ml - ml.max(1, keepdims=True)).sum(1)) + ml.max(1)
    newlp = ml[np.arange(B), actions] - lse
    old_logp = (newlp + rng.normal(0, 0.25, B)).astype(np.float32)
    return logits, mask, actions, old_logp, adv, ret, values, old_values


def run_learn(case, value_clip, clip_eps=0.2, vf=0.5, ent=0.01):
    logits, mask, actions, old_logp, adv, ret, values, old_values = case
    B = logits.shape[0]
    cfg = load_config()
    cfg.training.ppo_epochs = 1
    cfg.training.minibatch_size = B
    cfg.training.normalize_advantages = False
    cfg.training.enable_value_clipping = bool(value_clip)
    cfg.training.gradient_clip_max_norm = 1e30
    cfg.training.clip_epsilon = clip_eps
    cfg.training.value_loss_coeff = vf
    cfg.training.entropy_coef = ent
    cfg.training.learning_rate = 1e-12
    cfg.training.lr_schedule_type = None
    cfg.env.device = "cpu"
    model = StubModel(logits, values)
    agent = PPOAgent(model, cfg, torch.device("cpu"))
    buf = ExperienceBuffer(B, 0.99, 0.95, "cpu")
    for b in range(B):
        obs = torch.zeros(46, 9, 9)
        obs[0, 0, 0] = float(b)
        buf.add(obs, int(actions[b]), 0.0, float(old_logp[b]), float(old_values[b]), False, torch.tensor(mask[b]))
    buf.compute_advantages_and_returns(0.0)
    buf.advantages[:] = torch.tensor(adv)
    buf.returns[:] = torch.tensor(ret)
    metrics = agent.learn(buf)
    return {
        "dlogits": model.logits.grad.numpy().copy(),
        "dvalues": model.values.grad.numpy().copy(),
        "metrics": np.asarray([metrics["ppo/policy_loss"], metrics["ppo/value_loss"], metrics["ppo/entropy"],
                               metrics["ppo/kl_divergence_approx"], metrics["ppo/clip_fraction"]], np.float64),
    }


def make_ppo_loss():
    rng = np.random.default_rng(4321)
    specs = [(2, "mixed", 0), (3, "one", 0), (64, "mixed", 0), (8, "peaked", 0), (4, "dense", 0),
             (16, "mixed", 1), (5, "peaked", 1)]
    out = {"n_cases": np.asarray(len(specs))}
    for ci, (B, kind, vclip) in enumerate(specs):
        case = random_case(rng, B, kind)
        res = run_learn(case, vclip)
        logits, mask, actions, old_logp, adv, ret, values, old_values = case
        # sparse storage: only legal logits / gradients matter (illegal gradients are exactly 0, asserted here)
        assert np.all(res["dlogits"][~mask] == 0.0)
        out[f"c{ci}_mask_packed"] = np.packbits(mask, axis=1, bitorder="little")
        nz = np.nonzero(mask)
        out[f"c{ci}_legal_logits"] = logits[nz]
        out[f"c{ci}_legal_dlogits"] = res["dlogits"][nz]
        out[f"c{ci}_illegal_seed"] = np.asarray([1000 + ci])   # illegal logits are regenerated from this seed
        for k, v in (("actions", actions), ("old_logp", old_logp), ("adv", adv), ("ret", ret), ("values", values),
                     ("old_values", old_values), ("dvalues", res["dvalues"]), ("metrics", res["metrics"]),
                     ("cfg", np.asarray([0.2, 0.5, 0.01, vclip], np.float64))):
            out[f"c{ci}_{k}"] = v
        print("ppo case", ci, B, kind, vclip, res["metrics"])
    np.savez_compressed(os.path.join(HERE, "ppo_loss.npz"), **out)


# ------------------------------------------------------------------------- masked categorical
def make_sample():
    rng = np.random.default_rng(999)
    B = 24
    logits, mask, actions, *_ = random_case(rng, B, "mixed")
    mask[-1] = False          # all-illegal row -> NaN -> uniform fallback (base_actor_critic.py:93-101)
    mask[-2] = True           # dense row

    probs = torch.softmax(torch.where(torch.tensor(mask), torch.tensor(logits), torch.tensor(float("-inf"))), dim=-1)
    probs[torch.isnan(probs).any(dim=1)] = 1.0 / A
    dist = torch.distributions.Categorical(probs=probs)
    acts = torch.tensor(actions)
    acts[-1] = 77
    logp = dist.log_prob(acts).numpy()
    # the reference calls get_action_and_value with batch size 1 (ppo_agent.py:152-186); its NaN fallback
    # replaces the WHOLE batch (base_actor_critic.py:93-101), so evaluate row by row as the reference does
    det_a, det_lp = [], []
    for b in range(B):
        class M1(BaseActorCriticModel):
            def forward(self, x, b=b):
                return torch.tensor(logits[b:b + 1]), torch.zeros(1)
        a1, lp1, _ = M1().get_action_and_value(torch.zeros(1, 46, 9, 9), legal_mask=torch.tensor(mask[b]),
                                               deterministic=True)
        det_a.append(a1[0])
        det_lp.append(lp1[0])
    det_a, det_lp = torch.stack(det_a), torch.stack(det_lp)
    nz = np.nonzero(mask)
    np.savez_compressed(os.path.join(HERE, "sample.npz"),
                        mask_packed=np.packbits(mask, axis=1, bitorder="little"),
                        legal_logits=logits[nz], actions=acts.numpy(), logp=logp.astype(np.float32),
                        det_actions=det_a.numpy(), det_logp=det_lp.detach().numpy().astype(np.float32),
                        probs_legal=dist.probs.numpy()[nz].astype(np.float32))
    print("sample rows:", B)


if __name__ == "__main__":
    make_gae()
    make_ppo_loss()
    make_sample()


# ------------------------------------------------------------------------------- learn() end to end
def make_learn_e2e():
    """PPOAgent.learn on a small real network: metrics + (clipped) gradients of the single update.
    Both sides build the network from torch.manual_seed(7) with identical module order, so the initial
    weights agree without shipping them; a few checksums of the initial weights are stored to prove it."""
    from keisei.training.models.resnet_tower import ActorCriticResTower
    rng = np.random.default_rng(77)
    B = 8
    obs = (rng.random((B, 46, 9, 9)) < 0.05).astype(np.float32)
    mask = np.zeros((B, A), bool)
    actions = np.zeros(B, np.int64)
    for b in range(B):
        ids = np.sort(rng.choice(A, int(rng.choice([3, 30, 80])), replace=False))
        mask[b, ids] = True
        actions[b] = ids[rng.integers(len(ids))]
    old_logp = rng.normal(-3.0, 0.3, B).astype(np.float32)
    old_values = rng.normal(0, 0.5, B).astype(np.float32)
    adv = rng.normal(0, 1, B).astype(np.float32)
    ret = rng.normal(0, 1, B).astype(np.float32)
    out = {}
    for name, epochs, mb, normalize in (("one_update", 1, 8, False), ("two_epochs_mb4_norm", 2, 4, True)):
        torch.manual_seed(7)
        model = ActorCriticResTower(46, A, tower_depth=2, tower_width=16, se_ratio=0.25)
        init_sums = np.asarray([float(p.detach().double().sum()) for p in model.parameters()])
        cfg = load_config()
        cfg.training.ppo_epochs = epochs
        cfg.training.minibatch_size = mb
        cfg.training.normalize_advantages = normalize
        cfg.training.enable_value_clipping = False
        cfg.training.lr_schedule_type = None
        cfg.env.device = "cpu"
        cfg.env.seed = 42
        agent = PPOAgent(model, cfg, torch.device("cpu"))
        buf = ExperienceBuffer(B, 0.99, 0.95, "cpu")
        for b in range(B):
            buf.add(torch.tensor(obs[b]), int(actions[b]), 0.0, float(old_logp[b]), float(old_values[b]), False,
                    torch.tensor(mask[b]))
        buf.compute_advantages_and_returns(0.0)
        buf.advantages[:] = torch.tensor(adv)
        buf.returns[:] = torch.tensor(ret)
        metrics = agent.learn(buf)
        out[f"{name}_metrics"] = np.asarray([metrics["ppo/policy_loss"], metrics["ppo/value_loss"], metrics["ppo/entropy"],
                                             metrics["ppo/kl_divergence_approx"], metrics["ppo/clip_fraction"]], np.float64)
        out[f"{name}_init_sums"] = init_sums
        # gradients of the LAST update after clip_grad_norm_ (in place): first 32 entries of every tensor + norms
        out[f"{name}_grad_head"] = np.concatenate([p.grad.detach().reshape(-1)[:32].numpy() for p in model.parameters()])
        out[f"{name}_grad_norms"] = np.asarray([float(p.grad.detach().norm()) for p in model.parameters()])
        out[f"{name}_last_gradient_norm"] = np.asarray(agent.last_gradient_norm)
        print("learn_e2e", name, out[f"{name}_metrics"], agent.last_gradient_norm)
    np.savez_compressed(os.path.join(HERE, "learn_e2e.npz"), obs_packed=np.packbits(obs.astype(bool).reshape(B, -1), axis=1),
                        mask_packed=np.packbits(mask, axis=1, bitorder="little"), actions=actions, old_logp=old_logp,
                        old_values=old_values, adv=adv, ret=ret, **out)


# ----------------------------------------------------------------------------- checkpoint layout
def make_checkpoint_layout():
    """State-dict keys / shapes of the reference's networks and the key set of PPOAgent.save_model's dict
    (ppo_agent.py:462-487), so that reference checkpoints load into shogidrl_amd.models and vice versa."""
    import io
    import json
    from keisei.core.neural_network import ActorCritic
    from keisei.training.models.resnet_tower import ActorCriticResTower
    out = {}
    m = ActorCriticResTower(46, A, tower_depth=9, tower_width=256, se_ratio=0.25)
    out["resnet_9x256_se025"] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    m2 = ActorCriticResTower(46, A, tower_depth=2, tower_width=16, se_ratio=None)
    out["resnet_2x16_nose"] = [[k, list(v.shape)] for k, v in m2.state_dict().items()]
    t = ActorCritic(46, A)
    out["tiny_cnn"] = [[k, list(v.shape)] for k, v in t.state_dict().items()]
    cfg = load_config()
    cfg.env.device = "cpu"
    agent = PPOAgent(m2, cfg, torch.device("cpu"))
    buf = io.BytesIO()
    agent.save_model(buf, global_timestep=123, total_episodes_completed=4, stats_to_save={"black_wins": 1, "white_wins": 2, "draws": 1})
    buf.seek(0)
    ck = torch.load(buf, weights_only=True)
    out["save_model_keys"] = sorted(ck.keys())
    out["optimizer_state_dict_keys"] = sorted(ck["optimizer_state_dict"].keys())
    out["optimizer_param_group_keys"] = sorted(ck["optimizer_state_dict"]["param_groups"][0].keys())
    with open(os.path.join(HERE, "checkpoint_layout.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("checkpoint layout:", len(out["resnet_9x256_se025"]), "entries;", out["save_model_keys"])
