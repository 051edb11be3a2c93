"""Shared parity checks for the learner-side kernels (GAE, masked sampling, fused PPO loss): a backend
(`ops`) is either the host-emulated kernels (tests/emu) or the HIP library through the C-ABI."""
import os

import numpy as np

from oracle import ko, ppo_oracle

G = os.path.join(os.path.dirname(__file__), "golden")
A = 13527
# stated floating-point tolerances (north_star: "PPO losses/returns within a stated fp tolerance")
LOSS_RTOL, LOSS_ATOL = 1e-5, 1e-6      # metrics and gradients, float32 logits
LOGP_ATOL = 2e-6                        # log-probs, float32 logits
BF16_LOGP_ATOL = 2e-2                   # log-probs when the logits are rounded to bf16 first


def load_loss_case(z, ci):
    mask = np.unpackbits(z[f"c{ci}_mask_packed"], axis=1, bitorder="little")[:, :A].astype(bool)
    B = mask.shape[0]
    rng = np.random.default_rng(int(z[f"c{ci}_illegal_seed"][0]))
    logits = rng.normal(0, 1.5, (B, A)).astype(np.float32)   # illegal logits are irrelevant by construction
    nz = np.nonzero(mask)
    logits[nz] = z[f"c{ci}_legal_logits"]
    dl = np.zeros((B, A), np.float32)
    dl[nz] = z[f"c{ci}_legal_dlogits"]
    g = lambda k: z[f"c{ci}_{k}"]
    return dict(logits=logits, mask=mask, actions=g("actions"), old_logp=g("old_logp"), adv=g("adv"), ret=g("ret"),
                values=g("values"), old_values=g("old_values"), dlogits=dl, dvalues=g("dvalues"), metrics=g("metrics"),
                cfg=g("cfg"))


def check_gae_golden(ops):
    z = np.load(os.path.join(G, "gae.npz"))
    for ci in range(int(z["n_cases"])):
        r, v, d = z[f"c{ci}_r"], z[f"c{ci}_v"], z[f"c{ci}_d"]
        gamma, lam, last_v = (float(x) for x in z[f"c{ci}_hp"])
        adv, ret = ops.gae(r[:, None], v[:, None], d[:, None], np.asarray([last_v], np.float32), gamma, lam)
        assert np.array_equal(adv[:, 0], z[f"c{ci}_adv"]), ci     # bit-exact
        assert np.array_equal(ret[:, 0], z[f"c{ci}_ret"]), ci
    # the reference's own known answer (tests/core/test_experience_buffer.py:98-105): terminal step
    adv, ret = ops.gae(np.asarray([[2.0]]), np.asarray([[1.5]]), np.asarray([[1]]), np.asarray([0.7]), 0.99, 0.95)
    assert adv[0, 0] == 0.5 and ret[0, 0] == 2.0


def check_gae_vs_oracle(ops, T, N, seed):
    rng = np.random.default_rng(seed)
    r = rng.normal(0, 1, (T, N)).astype(np.float32)
    v = rng.normal(0, 1, (T, N)).astype(np.float32)
    d = (rng.random((T, N)) < 0.03).astype(np.uint8)
    lv = rng.normal(0, 1, N).astype(np.float32)
    adv, ret = ops.gae(r, v, d, lv, 0.99, 0.95)
    oadv, oret = ko.gae(r, v, d, lv, 0.99, 0.95)
    assert np.array_equal(adv, oadv) and np.array_equal(ret, oret)


def check_loss_golden(ops):
    z = np.load(os.path.join(G, "ppo_loss.npz"))
    for ci in range(int(z["n_cases"])):
        c = load_loss_case(z, ci)
        clip, vf, ent, vclip = (float(x) for x in c["cfg"])
        dl, dv, st = ops.ppo_loss(c["logits"], c["mask"], c["actions"], c["old_logp"], c["adv"], c["ret"], c["values"],
                                  c["old_values"], clip, vf, ent, int(vclip))
        pl, vl, el, kl, cf = c["metrics"]
        np.testing.assert_allclose(st[1:6], [pl, vl, el, kl, cf], rtol=LOSS_RTOL, atol=LOSS_ATOL, err_msg=str(ci))
        np.testing.assert_allclose(st[0], pl + vf * vl + ent * el, rtol=LOSS_RTOL, atol=LOSS_ATOL)
        assert np.all(dl[~c["mask"]] == 0.0), ci                      # illegal actions get exactly zero gradient
        np.testing.assert_allclose(dl, c["dlogits"], rtol=LOSS_RTOL, atol=LOSS_ATOL, err_msg=str(ci))
        np.testing.assert_allclose(dv, c["dvalues"], rtol=LOSS_RTOL, atol=LOSS_ATOL, err_msg=str(ci))


def check_oracle_loss_golden():
    """pins oracle/ppo_oracle.py itself against the reference's autograd outputs"""
    z = np.load(os.path.join(G, "ppo_loss.npz"))
    for ci in range(int(z["n_cases"])):
        c = load_loss_case(z, ci)
        clip, vf, ent, vclip = (float(x) for x in c["cfg"])
        st, dz, dv = ppo_oracle.ppo_loss(c["logits"], c["mask"], c["actions"], c["old_logp"].astype(np.float64),
                                         c["adv"].astype(np.float64), c["ret"].astype(np.float64),
                                         c["values"].astype(np.float64), c["old_values"].astype(np.float64),
                                         clip, vf, ent, int(vclip))
        pl, vl, el, kl, cf = c["metrics"]
        got = [st["policy_loss"], st["value_loss"], st["entropy_loss"], st["kl"], st["clip_frac"]]
        np.testing.assert_allclose(got, [pl, vl, el, kl, cf], rtol=LOSS_RTOL, atol=LOSS_ATOL, err_msg=str(ci))
        np.testing.assert_allclose(dz, c["dlogits"], rtol=2e-5, atol=LOSS_ATOL, err_msg=str(ci))
        np.testing.assert_allclose(dv, c["dvalues"], rtol=LOSS_RTOL, atol=LOSS_ATOL, err_msg=str(ci))


def random_loss_inputs(B, seed, dense_rows=0, empty_rows=0):
    rng = np.random.default_rng(seed)
    logits = rng.normal(0, 2.0, (B, A)).astype(np.float32)
    mask = np.zeros((B, A), bool)
    actions = np.zeros(B, np.int64)
    for b in range(B):
        n = A if b < dense_rows else int(rng.choice([1, 2, 5, 31, 48, 90, 206, 593]))
        ids = np.sort(rng.choice(A, n, replace=False))
        mask[b, ids] = True
        actions[b] = ids[rng.integers(n)]
    for b in range(B - empty_rows, B):
        mask[b] = False
    f = lambda s: rng.normal(0, s, B).astype(np.float32)
    return logits, mask, actions, f(1.0) - 3.0, f(1.0), f(1.0), f(1.0), f(1.0)


def check_loss_vs_oracle(ops, B, seed, value_clip=0, bf16=False, dense_rows=0, empty_rows=0, logit_scale=1.0):
    """logit_scale >> 1: peaked soft-max -- probabilities outside the Categorical eps clamp on both sides, and taken actions
    whose float32 probability underflows to exactly 0 (gradient 0, not 0 / 0)"""
    logits, mask, actions, old_logp, adv, ret, values, old_values = random_loss_inputs(B, seed, dense_rows, empty_rows)
    logits = (logits * np.float32(logit_scale)).astype(np.float32)
    if bf16:
        from tests.emu.emu import from_bf16_bits, to_bf16_bits
        logits = from_bf16_bits(to_bf16_bits(logits))      # the oracle sees the same rounded logits
    dl, dv, st = ops.ppo_loss(logits, mask, actions, old_logp, adv, ret, values, old_values, 0.2, 0.5, 0.01, value_clip,
                              **({"bf16": True} if bf16 else {}))
    ost, odz, odv = ppo_oracle.ppo_loss(logits, mask, actions, *(x.astype(np.float64) for x in
                                                                  (old_logp, adv, ret, values, old_values)),
                                        0.2, 0.5, 0.01, value_clip)
    want = [ost[k] for k in ("loss", "policy_loss", "value_loss", "entropy_loss", "kl", "clip_frac")]
    np.testing.assert_allclose(st[:6], want, rtol=2e-5, atol=2e-6)
    assert np.all(dl[~mask] == 0.0)
    if bf16:  # gradients are rounded to bf16 on store: 2^-8 relative
        np.testing.assert_allclose(dl, odz, rtol=1e-2, atol=1e-6)
    else:
        np.testing.assert_allclose(dl, odz, rtol=2e-5, atol=LOSS_ATOL)
    np.testing.assert_allclose(dv, odv, rtol=LOSS_RTOL, atol=LOSS_ATOL)


def load_sample_case():
    z = np.load(os.path.join(G, "sample.npz"))
    mask = np.unpackbits(z["mask_packed"], axis=1, bitorder="little")[:, :A].astype(bool)
    B = mask.shape[0]
    logits = np.random.default_rng(5).normal(0, 1.5, (B, A)).astype(np.float32)
    logits[np.nonzero(mask)] = z["legal_logits"]
    return z, logits, mask


def check_sample_golden(ops):
    z, logits, mask = load_sample_case()
    B = mask.shape[0]
    # deterministic: arg-max action and its log-prob (rows with legal actions; the NaN row picks index 0)
    for off in (0, 1, 3, 7):
        act, lp = ops.masked_sample(logits, mask, 42, 0, True, offset=off)
        assert np.array_equal(act, z["det_actions"]), off
        np.testing.assert_allclose(lp, z["det_logp"], atol=LOGP_ATOL, rtol=0)
    # stochastic: sampled actions are legal, reproducible, and their log-probs match Categorical.log_prob
    act, lp = ops.masked_sample(logits, mask, 42, 3, False)
    act2, lp2 = ops.masked_sample(logits, mask, 42, 3, False)
    assert np.array_equal(act, act2) and np.array_equal(lp, lp2)
    act3, _ = ops.masked_sample(logits, mask, 42, 4, False)
    assert not np.array_equal(act, act3)
    rows = np.arange(B - 1)
    assert mask[rows, act[:-1]].all()
    want_lp, _, _, _, _ = ppo_oracle.log_prob_entropy(logits, mask, act)
    np.testing.assert_allclose(lp, want_lp, atol=LOGP_ATOL, rtol=0)
    assert 0 <= act[-1] < A and abs(lp[-1] - np.log(1.0 / A)) < 1e-5     # uniform fallback row
    # the golden log-probs of the fixture's own actions, through the oracle (pins the oracle)
    olp, _, _, _, _ = ppo_oracle.log_prob_entropy(logits, mask, z["actions"])
    np.testing.assert_allclose(olp, z["logp"], atol=LOGP_ATOL, rtol=0)


def check_sample_distribution(ops, n_draws=4000, seed=11):
    """chi-square goodness of fit of the Gumbel-max sampler against the masked soft-max."""
    rng = np.random.default_rng(seed)
    n = 12
    ids = np.sort(rng.choice(A, n, replace=False))
    row = rng.normal(0, 1.0, A).astype(np.float32)
    m = np.zeros(A, bool)
    m[ids] = True
    p, _ = ppo_oracle.masked_probs(row[None], m[None])
    B = 200
    counts = np.zeros(A)
    for s in range(n_draws // B):
        act, _ = ops.masked_sample(np.repeat(row[None], B, 0), np.repeat(m[None], B, 0), 7, s, False)
        np.add.at(counts, act, 1)
    assert counts[~m].sum() == 0
    e = p[0, ids] * n_draws
    chi2 = ((counts[ids] - e) ** 2 / e).sum()
    assert chi2 < 40.0, chi2   # dof = 11: P(chi2 > 40) ~ 4e-5


def check_sample_bf16(ops):
    z, logits, mask = load_sample_case()
    act, lp = ops.masked_sample(logits, mask, 1, 1, False, bf16=True)
    want_lp, _, _, _, _ = ppo_oracle.log_prob_entropy(logits, mask, act)
    np.testing.assert_allclose(lp[:-1], want_lp[:-1], atol=BF16_LOGP_ATOL, rtol=0)


def check_evaluate_actions(ops, B=9, seed=4):
    """evaluate_actions' tail (log-prob of the given action + entropy per row, base_actor_critic.py:118-184) against the
    oracle, incl. a dense row (block kernel), a 1-legal row (log-prob = log(1 - eps), entropy ~ 0) and an all-illegal row."""
    logits, mask, actions, *_ = random_loss_inputs(B, seed, dense_rows=1, empty_rows=1)
    one = B - 2
    mask[one] = False
    mask[one, 777] = True
    actions[one] = 777
    lp, en = ops.evaluate_actions(logits, mask, actions)
    wlp, wen, _, _, none = ppo_oracle.log_prob_entropy(logits, mask, actions)
    np.testing.assert_allclose(lp[~none], wlp[~none], atol=LOGP_ATOL, rtol=0)
    np.testing.assert_allclose(en[~none], wen[~none], atol=2e-5, rtol=1e-5)
    assert abs(lp[one] - np.log(1.0 - ppo_oracle.EPS)) < 1e-9
    # all-illegal row: uniform over all actions (base_actor_critic.py:166-174)
    assert abs(lp[-1] - np.log(1.0 / A)) < 1e-5 and abs(en[-1] - np.log(A)) < 1e-4
    # and the golden log-probs the reference itself produced (sample.npz)
    z, lg, mk = load_sample_case()
    lp2, _ = ops.evaluate_actions(lg, mk, z["actions"])
    np.testing.assert_allclose(lp2, z["logp"], atol=LOGP_ATOL, rtol=0)
