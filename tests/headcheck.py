"""Shared parity checks for the FUSED policy head (shogidrl_amd/csrc/ksh_head_kernels.h): Linear(K, 13527) + masked
soft-max tail + PPO loss and its backward without logits in memory.  A backend (`ops`) is the host-emulated kernels
(tests/emu) or the HIP library through the C-ABI (tests/gpu_backends.py).

The reference's own fixtures give LOGITS, not features: they are fed through the head with one-hot features and
W[:, b] = logits[b, :] (K = B), which reproduces every logit exactly (1.0 * w plus exact zeros), so that
  dW[a][b] == dlogits[b][a]   and   dfeat == dlogits @ W
pin the fused path to the very same autograd outputs of the reference (tests/golden/ppo_loss.npz, sample.npz)."""
import numpy as np

from oracle import ppo_oracle
from tests import ppocheck
from tests.ppocheck import A, LOGP_ATOL, LOSS_ATOL, LOSS_RTOL

HEAD_LOGP_ATOL = 5e-6      # log-probs when the logits themselves are float32 dot products of K terms (fp32 GEMM noise)


def onehot_head(logits):
    """(feat, W) with feat @ W.T == logits exactly"""
    B = logits.shape[0]
    K = max(4, B + (B & 1))
    feat = np.zeros((B, K), np.float32)
    feat[np.arange(B), np.arange(B)] = 1.0
    W = np.zeros((A, K), np.float32)
    W[:, :B] = logits.T
    return feat, W


def check_head_loss_golden(ops, cases=None):
    z = np.load(ppocheck.os.path.join(ppocheck.G, "ppo_loss.npz"))
    for ci in (range(int(z["n_cases"])) if cases is None else cases):
        c = ppocheck.load_loss_case(z, ci)
        clip, vf, ent, vclip = (float(x) for x in c["cfg"])
        feat, W = onehot_head(c["logits"])
        B = feat.shape[0]
        zero_bias = np.zeros(A, np.float32)                           # exact (z + 0.0) and makes dbias observable
        dfeat, dW, dbias, dv, st = ops.head_loss(feat, W, zero_bias, c["mask"], c["actions"], c["old_logp"], c["adv"], c["ret"],
                                                 c["values"], c["old_values"], clip, vf, ent, int(vclip))
        pl, vl, el, kl, cf = c["metrics"]
        np.testing.assert_allclose(st[1:6], [pl, vl, el, kl, cf], rtol=LOSS_RTOL, atol=LOSS_ATOL, err_msg=str(ci))
        np.testing.assert_allclose(st[0], pl + vf * vl + ent * el, rtol=LOSS_RTOL, atol=LOSS_ATOL)
        np.testing.assert_allclose(dv, c["dvalues"], rtol=LOSS_RTOL, atol=LOSS_ATOL, err_msg=str(ci))
        got_dl = dW[:, :B].T                                          # == d loss / d logits
        assert np.all(got_dl[~c["mask"]] == 0.0), ci                  # illegal actions: exactly zero gradient
        np.testing.assert_allclose(got_dl, c["dlogits"], rtol=LOSS_RTOL, atol=LOSS_ATOL, err_msg=str(ci))
        if feat.shape[1] > B:
            assert np.all(dW[:, B:] == 0.0)
        np.testing.assert_allclose(dbias, c["dlogits"].astype(np.float64).sum(0), rtol=2e-5, atol=LOSS_ATOL, err_msg=str(ci))
        want_dfeat = c["dlogits"].astype(np.float64) @ W.astype(np.float64)
        np.testing.assert_allclose(dfeat, want_dfeat, rtol=2e-5, atol=2e-6, err_msg=str(ci))


def random_head_inputs(B, K, seed, dense_rows=0, empty_rows=0, w_scale=1.0):
    rng = np.random.default_rng(seed)
    feat = rng.normal(0, 1.0, (B, K)).astype(np.float32)
    W = rng.normal(0, w_scale * 2.0 / np.sqrt(K), (A, K)).astype(np.float32)
    bias = rng.normal(0, 0.5, A).astype(np.float32)
    _, mask, actions, old_logp, adv, ret, values, old_values = ppocheck.random_loss_inputs(B, seed, dense_rows, empty_rows)
    return feat, W, bias, mask, actions, old_logp, adv, ret, values, old_values


def head_reference(feat, W, bias, mask, actions, old_logp, adv, ret, values, old_values, clip, vf, ent, value_clip):
    """float64 oracle of the whole fused op"""
    f64 = lambda x: np.asarray(x, np.float64)
    logits = f64(feat) @ f64(W).T + (0.0 if bias is None else f64(bias))
    st, dz, dv = ppo_oracle.ppo_loss(logits, mask, actions, f64(old_logp), f64(adv), f64(ret), f64(values), f64(old_values),
                                     clip, vf, ent, value_clip)
    return st, dz @ f64(W), dz.T @ f64(feat), dz.sum(0), dv, logits


def check_head_loss_vs_oracle(ops, B, K, seed, value_clip=0, dense_rows=0, empty_rows=0, dense=True, with_bias=True, w_scale=1.0,
                              greedy_actions=False):
    """w_scale > 1 makes the soft-max peaked: probabilities beyond the Categorical eps clamp on both sides (the entries the
    one-pass input gradient corrects separately); greedy_actions: every other row takes its arg-max action (probability
    above 1 - eps, no policy gradient)"""
    feat, W, bias, mask, actions, old_logp, adv, ret, values, old_values = random_head_inputs(B, K, seed, dense_rows, empty_rows,
                                                                                              w_scale)
    if not with_bias:
        bias = None
    if greedy_actions:
        lg = np.where(mask.astype(bool), feat.astype(np.float64) @ W.astype(np.float64).T + (0.0 if bias is None else bias), -np.inf)
        actions = actions.copy()
        actions[::2] = lg.argmax(1)[::2]
    dfeat, dW, dbias, dv, st = ops.head_loss(feat, W, bias, mask, actions, old_logp, adv, ret, values, old_values, 0.2, 0.5, 0.01,
                                             value_clip, dense=dense)
    ost, odfeat, odW, odb, odv, _ = head_reference(feat, W, bias, mask, actions, old_logp, adv, ret, values, old_values, 0.2, 0.5,
                                                   0.01, value_clip)
    want = [ost[k] for k in ("loss", "policy_loss", "value_loss", "entropy_loss", "kl", "clip_frac")]
    np.testing.assert_allclose(st[:6], want, rtol=5e-5, atol=5e-6)
    never = ~mask.any(axis=0)
    assert np.all(dW[never] == 0.0) and (not with_bias or np.all(dbias[never] == 0.0))   # legal nowhere: exactly zero rows
    # Elements of a gradient row are sums that cancel: their error scales with the ROW's magnitude, not with the element.
    # float32 logits of magnitude ~13 w_scale carry a rounding error of ~1e-6 w_scale, i.e. that relative error in every
    # probability, which the float64 oracle (exact logits from the same float32 inputs) does not see -- a row with a
    # near-tie and a large PPO ratio showed 7.6e-6 of its largest element at w_scale = 30 (tools/stress_head_vs_oracle.py,
    # reproduced bit for bit by the emulator).
    def close(got, want, what):
        scale = np.abs(want).max(axis=-1, keepdims=True) if want.ndim > 1 else np.abs(want).max()
        bound = 5e-5 * np.abs(want) + 5e-6 + 2e-6 * max(1.0, w_scale) * scale
        bad = np.abs(got - want) > bound
        assert not bad.any(), "%s: %d elements off, worst %.3g (bound %.3g)" % (
            what, int(bad.sum()), float(np.abs(got - want)[bad].max()), float(np.broadcast_to(bound, got.shape)[bad].min()))
    close(dfeat, odfeat, "dfeat")
    close(dW, odW, "dW")
    if with_bias:                                                      # (no bias tensor, no bias gradient)
        close(dbias, odb, "dbias")
    np.testing.assert_allclose(dv, odv, rtol=LOSS_RTOL, atol=LOSS_ATOL)


def check_head_dense_row_without_dense_path_is_loud(ops):
    """dense_rows_possible = 0 is a promise; a dense row showing up anyway must poison the results, not be skipped"""
    feat, W, bias, mask, actions, old_logp, adv, ret, values, old_values = random_head_inputs(3, 162, 8, dense_rows=1)
    _, _, _, dv, st = ops.head_loss(feat, W, bias, mask, actions, old_logp, adv, ret, values, old_values, 0.2, 0.5, 0.01, 0,
                                    dense=False)
    assert np.isnan(st[:6]).all() and np.isnan(dv).all()
    lp, en = ops.head_evaluate(feat, W, bias, mask, actions, dense=False)      # evaluate-only: the dense row's outputs are NaN
    assert np.isnan(lp[0]) and np.isnan(en[0]) and np.isfinite(lp[1:]).all() and np.isfinite(en[1:]).all()


def check_head_evaluate(ops, B=7, K=162, seed=12):
    feat, W, bias, mask, actions, *_ = random_head_inputs(B, K, seed, dense_rows=1, empty_rows=1)
    lp, en = ops.head_evaluate(feat, W, bias, mask, actions)
    logits = feat.astype(np.float64) @ W.astype(np.float64).T + bias
    wlp, wen, _, _, none = ppo_oracle.log_prob_entropy(logits, mask, actions)
    np.testing.assert_allclose(lp[~none], wlp[~none], atol=HEAD_LOGP_ATOL, rtol=0)
    np.testing.assert_allclose(en[~none], wen[~none], atol=5e-5, rtol=2e-5)
    assert abs(lp[-1] - np.log(1.0 / A)) < 1e-5 and abs(en[-1] - np.log(A)) < 1e-4


def check_head_sample_golden(ops):
    z, logits, mask = ppocheck.load_sample_case()
    B = mask.shape[0]
    feat, W = onehot_head(logits)
    act, lp = ops.head_sample(feat, W, None, mask, 42, 0, True)
    assert np.array_equal(act, z["det_actions"])
    np.testing.assert_allclose(lp, z["det_logp"], atol=LOGP_ATOL, rtol=0)
    act, lp = ops.head_sample(feat, W, None, mask, 42, 3, False)
    act2, lp2 = ops.head_sample(feat, W, None, mask, 42, 3, False)
    assert np.array_equal(act, act2) and np.array_equal(lp, lp2)
    act3, _ = ops.head_sample(feat, W, None, mask, 42, 4, False)
    assert not np.array_equal(act, act3)
    assert mask[np.arange(B - 1), act[:-1]].all()
    want_lp, _, _, _, _ = ppo_oracle.log_prob_entropy(logits, mask, act)
    np.testing.assert_allclose(lp, want_lp, atol=LOGP_ATOL, rtol=0)
    assert 0 <= act[-1] < A and abs(lp[-1] - np.log(1.0 / A)) < 1e-5     # uniform fallback row
    # the logits-based sampler draws the very same actions from the same (seed, step): one RNG stream, two kernels
    if hasattr(ops, "masked_sample"):
        a_ref, lp_ref = ops.masked_sample(logits, mask, 42, 3, False)
        assert np.array_equal(a_ref, act)
        np.testing.assert_allclose(lp_ref, lp, atol=LOGP_ATOL, rtol=0)


def check_head_sample_vs_oracle(ops, B=9, K=162, seed=3):
    """random features: actions legal, log-probs those of the float64 head; dense row handled; arg-max = oracle's"""
    feat, W, bias, mask, *_ = random_head_inputs(B, K, seed, dense_rows=1, empty_rows=1)
    logits = feat.astype(np.float64) @ W.astype(np.float64).T + bias
    act, lp = ops.head_sample(feat, W, bias, mask, 9, 1, False)
    rows = np.arange(B - 1)
    assert mask[rows, act[:-1]].all()
    wlp, _, _, _, _ = ppo_oracle.log_prob_entropy(logits, mask, act)
    np.testing.assert_allclose(lp[:-1], wlp[:-1], atol=HEAD_LOGP_ATOL, rtol=0)
    det, dlp = ops.head_sample(feat, W, bias, mask, 9, 1, True)
    masked = np.where(mask, logits, -np.inf)
    best = masked[rows, det[:-1]]
    assert np.all(masked[:-1].max(axis=1) - best < 1e-5)                  # the arg-max up to float32 logit noise


def check_head_sample_distribution(ops, n_draws=4000, seed=11):
    rng = np.random.default_rng(seed)
    n = 12
    ids = np.sort(rng.choice(A, n, replace=False))
    row = rng.normal(0, 1.0, A).astype(np.float32)
    m = np.zeros(A, bool)
    m[ids] = True
    p, _ = ppo_oracle.masked_probs(row[None], m[None])
    Bn = 200
    feat = np.zeros((Bn, 4), np.float32)
    feat[:, 0] = 1.0
    W = np.zeros((A, 4), np.float32)
    W[:, 0] = row
    counts = np.zeros(A)
    for s in range(n_draws // Bn):
        act, _ = ops.head_sample(feat, W, None, np.repeat(m[None], Bn, 0), 7, s, False)
        np.add.at(counts, act, 1)
    assert counts[~m].sum() == 0
    e = p[0, ids] * n_draws
    chi2 = ((counts[ids] - e) ** 2 / e).sum()
    assert chi2 < 40.0, chi2


def check_pack_mask(ops, seed=2):
    rng = np.random.default_rng(seed)
    mask = rng.random((5, A)) < 0.01
    mask[1] = True
    mask[2] = False
    mask[3, -1] = True
    mask[3, 0] = True
    mask = mask.astype(np.uint8)
    mask[4, 100] = 7                                  # any non-zero byte counts as legal
    bits = ops.pack_mask_kernel(mask)
    want = np.packbits(np.pad(mask.astype(bool), ((0, 0), (0, 424 * 32 - A))), axis=1, bitorder="little").view(np.uint32)
    assert np.array_equal(bits, want)
