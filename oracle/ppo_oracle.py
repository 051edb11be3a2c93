"""TEST INFRASTRUCTURE: numpy restatement of the reference's masked-categorical and PPO-loss
arithmetic (core/base_actor_critic.py:118-184, core/ppo_agent.py:340-383) including the gradients
torch.autograd produces for it.  Pinned against tests/golden/ppo_loss.npz and sample.npz, which come
from the reference itself (tests/golden/make_golden_ppo.py).  Never imported by the product."""
import numpy as np

A = 13527
EPS = float(np.finfo(np.float32).eps)  # torch.distributions.utils.clamp_probs


def masked_probs(logits, mask):
    """softmax(where(mask, logits, -inf)) in float64; all-illegal rows -> uniform (base_actor_critic.py:166-174)."""
    z = np.where(mask, logits.astype(np.float64), -np.inf)
    m = z.max(axis=1, keepdims=True)
    none = ~np.isfinite(m[:, 0])
    m = np.where(np.isfinite(m), m, 0.0)
    e = np.exp(z - m)
    s = e.sum(axis=1, keepdims=True)
    p = np.where(s > 0, e / np.where(s > 0, s, 1.0), 1.0 / A)
    return p, none


def log_prob_entropy(logits, mask, actions):
    p, none = masked_probs(logits, mask)
    y = p / p.sum(axis=1, keepdims=True)                       # Categorical.__init__ renormalises
    l = np.log(np.clip(y, EPS, 1.0 - EPS))                     # probs_to_logits
    rows = np.arange(len(actions))
    return l[rows, actions], -(y * l).sum(axis=1), y, l, none


def ppo_loss(logits, mask, actions, old_logp, adv, ret, values, old_values, clip_eps, vf, ent, value_clip):
    """Returns stats dict, dlogits [B,A], dvalues [B] (float64)."""
    B = logits.shape[0]
    new_lp, H, y, l, none = log_prob_entropy(logits, mask, actions)
    ratio = np.exp(new_lp - old_logp)
    surr1 = ratio * adv
    surr2 = np.clip(ratio, 1 - clip_eps, 1 + clip_eps) * adv
    policy_loss = -np.minimum(surr1, surr2).mean()
    vu = ((values - ret) ** 2).mean()
    wu, wc = 1.0, 0.0
    value_loss = vu
    if value_clip:
        vcl = old_values + np.clip(values - old_values, -clip_eps, clip_eps)
        vc = ((vcl - ret) ** 2).mean()
        value_loss = max(vu, vc)
        wu, wc = (1.0, 0.0) if vu > vc else ((0.0, 1.0) if vc > vu else (0.5, 0.5))
    entropy_loss = -H.mean()
    loss = policy_loss + vf * value_loss + ent * entropy_loss
    stats = dict(loss=loss, policy_loss=policy_loss, value_loss=value_loss, entropy_loss=entropy_loss,
                 kl=(old_logp - new_lp).mean(), clip_frac=(np.abs(ratio - 1.0) > clip_eps).mean())
    # gradients: G = dL/dy, then through y = p/sum(p) and the soft-max: dz = y * (G - sum(y G))
    u = (y >= EPS) & (y <= 1 - EPS)
    G = (ent / B) * (l + u)
    rows = np.arange(B)
    gr = np.where(surr1 <= surr2, adv, 0.0)
    ya = y[rows, actions]
    G[rows, actions] += np.where(u[rows, actions], -gr * ratio / (B * ya), 0.0)
    c = (y * G).sum(axis=1, keepdims=True)
    dz = np.where(mask, y * (G - c), 0.0)
    dz[none] = 0.0
    dv = wu * vf * 2.0 * (values - ret) / B
    if value_clip and wc:
        dvv = values - old_values
        vcl = old_values + np.clip(dvv, -clip_eps, clip_eps)
        dv = dv + wc * vf * 2.0 * (vcl - ret) / B * ((dvv >= -clip_eps) & (dvv <= clip_eps))
    return stats, dz, dv
