"""The fused policy head on the MI355X through the C-ABI (ksh_policy_head_*): the reference's golden vectors, the
float64 oracle, the MFMA dense path, and agreement with the logits-based kernels it replaces."""
import numpy as np
import pytest
import torch

from tests import headcheck

pytestmark = pytest.mark.gpu
A = 13527


@pytest.fixture(scope="module")
def ops():
    from tests.gpu_backends import GpuOps
    return GpuOps()


def test_gpu_pack_mask_kernel(ops):
    headcheck.check_pack_mask(ops)


def test_gpu_head_loss_golden_reference_autograd(ops):
    headcheck.check_head_loss_golden(ops)


@pytest.mark.parametrize("B,K,vc,dense,empty", [(5, 162, 0, 0, 0), (6, 162, 1, 1, 1), (3, 1296, 0, 0, 1), (130, 162, 0, 3, 2),
                                                (64, 1296, 1, 2, 0)])
def test_gpu_head_loss_vs_oracle(ops, B, K, vc, dense, empty):
    headcheck.check_head_loss_vs_oracle(ops, B, K, seed=B + K, value_clip=vc, dense_rows=dense, empty_rows=empty)


def test_gpu_head_loss_sparse_only_and_no_bias(ops):
    headcheck.check_head_loss_vs_oracle(ops, 300, 162, seed=5, dense=False, with_bias=False)


@pytest.mark.parametrize("K,scale,greedy", [(162, 6.0, False), (162, 25.0, True), (1296, 25.0, True), (10, 100.0, False)])
def test_gpu_head_loss_peaked_softmax_clamped_entries(ops, K, scale, greedy):
    """probabilities below eps / above 1 - eps (the entries the one-pass input gradient of K <= 192 fetches a second time),
    down to float32 probabilities of exactly 0 for the taken action (gradient 0, not 0 / 0)"""
    headcheck.check_head_loss_vs_oracle(ops, 70, K, seed=21, dense=False, w_scale=scale, greedy_actions=greedy)


def test_gpu_head_dense_row_without_dense_path_is_loud(ops):
    headcheck.check_head_dense_row_without_dense_path_is_loud(ops)


def test_gpu_head_evaluate(ops):
    headcheck.check_head_evaluate(ops)


def test_gpu_head_sample_golden(ops):
    headcheck.check_head_sample_golden(ops)


def test_gpu_head_sample_vs_oracle(ops):
    headcheck.check_head_sample_vs_oracle(ops)


def test_gpu_head_sample_distribution(ops):
    headcheck.check_head_sample_distribution(ops)


def test_gpu_mfma_dense_path_all_rows_dense_equals_torch():
    """every row dense (all-ones masks, as in the reference's unit tests): the whole op runs on the float32 MFMA GEMMs;
    compared with plain PyTorch float32 autograd of the same arithmetic"""
    from shogidrl_amd import ops as O
    torch.manual_seed(0)
    B, K = 96, 162
    dev = "cuda:0"
    feat = torch.randn(B, K, device=dev, requires_grad=True)
    W = (torch.randn(A, K, device=dev) * 0.15).requires_grad_(True)
    b = (torch.randn(A, device=dev) * 0.3).requires_grad_(True)
    values = torch.randn(B, device=dev, requires_grad=True)
    mask = torch.ones(B, A, dtype=torch.bool, device=dev)
    actions = torch.randint(0, A, (B,), device=dev)
    old_lp = torch.full((B,), -9.5, device=dev) + 0.1 * torch.randn(B, device=dev)
    adv, ret = torch.randn(B, device=dev), torch.randn(B, device=dev)
    loss, stats = O.policy_head_ppo_loss(feat, W, b, values, mask, actions, old_lp, adv, ret, None, 0.2, 0.5, 0.01, False)
    loss.backward()
    got = [feat.grad.clone(), W.grad.clone(), b.grad.clone(), values.grad.clone()]
    for t in (feat, W, b, values):
        t.grad = None
    logits = feat @ W.t() + b
    p = torch.softmax(logits, dim=1)
    dist = torch.distributions.Categorical(probs=p)
    new_lp, H = dist.log_prob(actions), dist.entropy()
    ratio = torch.exp(new_lp - old_lp)
    pl = -torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).mean()
    vl = torch.nn.functional.mse_loss(values, ret)
    ref = pl + 0.5 * vl + 0.01 * (-H.mean())
    ref.backward()
    assert abs(float(loss) - float(ref)) < 2e-5 * max(1.0, abs(float(ref)))
    for g, t, name in zip(got, (feat, W, b, values), ("dfeat", "dW", "dbias", "dvalues")):
        np.testing.assert_allclose(g.cpu().numpy(), t.grad.cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=name)


def test_gpu_mfma_dense_rows_far_apart_in_a_batch_longer_than_one_step_window():
    """16 500 rows, three of them dense (all-ones masks) at indices 5, 16 450 and 16 499: the weight-gradient product reduces
    over the SAMPLES in two slices of 8 256 / 8 244 rows, walks each slice's K-steps in windows of 512 (= 8 192 samples) and
    takes only the steps that hold a flagged row -- here one step in the first window of slice 0 and two in the SECOND window
    of slice 1, the last of them the range's tail step (16 500 = 1031 * 16 + 4).  Compared with plain PyTorch float32 autograd
    of the same arithmetic (masked soft-max for the sparse rows)."""
    from shogidrl_amd import ops as O
    torch.manual_seed(3)
    B, K, dev = 16500, 16, "cuda:0"
    feat = torch.randn(B, K, device=dev, requires_grad=True)
    W = (torch.randn(A, K, device=dev) * 0.3).requires_grad_(True)
    b = (torch.randn(A, device=dev) * 0.3).requires_grad_(True)
    values = torch.randn(B, device=dev, requires_grad=True)
    mask = torch.zeros(B, A, dtype=torch.bool, device=dev)
    idx = torch.randint(0, A, (B, 40), device=dev)
    mask.scatter_(1, idx, True)                                         # ~40 legal actions per sparse row
    for r in (5, 16450, 16499):
        mask[r] = True
    actions = idx[:, 0].clone()
    old_lp = torch.full((B,), -3.5, device=dev) + 0.1 * torch.randn(B, device=dev)
    adv, ret = torch.randn(B, device=dev), torch.randn(B, device=dev)
    loss, stats = O.policy_head_ppo_loss(feat, W, b, values, mask, actions, old_lp, adv, ret, None, 0.2, 0.5, 0.01, False)
    loss.backward()
    got = [feat.grad.clone(), W.grad.clone(), b.grad.clone(), values.grad.clone()]
    for t in (feat, W, b, values):
        t.grad = None
    logits = torch.where(mask, feat @ W.t() + b, torch.full((), float("-inf"), device=dev))
    p = torch.softmax(logits, dim=1)
    dist = torch.distributions.Categorical(probs=p)
    new_lp, H = dist.log_prob(actions), dist.entropy()
    ratio = torch.exp(new_lp - old_lp)
    pl = -torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).mean()
    vl = torch.nn.functional.mse_loss(values, ret)
    ref = pl + 0.5 * vl + 0.01 * (-H.mean())
    ref.backward()
    assert abs(float(loss) - float(ref)) < 5e-5 * max(1.0, abs(float(ref)))
    for g, t, name in zip(got, (feat, W, b, values), ("dfeat", "dW", "dbias", "dvalues")):
        np.testing.assert_allclose(g.cpu().numpy(), t.grad.cpu().numpy(), rtol=5e-4, atol=5e-7, err_msg=name)


@pytest.mark.parametrize("N", [4096, 8192])
def test_gpu_fused_head_agrees_with_logits_kernels_on_self_play_masks(N):
    """production shape: 4096 (and 8192: two 256-sample windows per segment) rows of real self-play masks, K = 162.  The fused sampler draws the same actions as the
    logits-based sampler fed hipBLASLt logits (same Philox stream; float32 logit noise can only flip near-ties), and the
    fused loss returns the gradients PyTorch's autograd derives from the logits-based loss kernel."""
    from shogidrl_amd import ops as O
    from shogidrl_amd.env import VecShogiEnv
    torch.manual_seed(1)
    K, dev = 162, "cuda:0"
    env = VecShogiEnv(N, 500, device=dev)
    env.reset()
    zeros = torch.zeros(N, A, device=dev)
    for t in range(30):                                        # desynchronise the games
        a, _ = O.masked_sample(zeros, env.legal_bits, 3, t)
        env.step(a)
    bits, mask = env.legal_bits, env.legal_mask
    assert int(env.n_legal.max()) <= 593
    if N > 4096:
        # one action legal in EVERY sample: a heavy action whose segments hold 512 samples each
        mask = mask.clone()
        mask[:, 77] = True
        bits = O.pack_mask_bits(mask)
    feat = torch.randn(N, K, device=dev)
    W = torch.randn(A, K, device=dev) * 0.1
    b = torch.randn(A, device=dev) * 0.1
    logits = feat @ W.t() + b
    a1, lp1 = O.policy_head_sample(feat, W, b, bits, 7, 5)
    a2, lp2 = O.masked_sample(logits, bits, 7, 5)
    same = a1 == a2
    assert float(same.float().mean()) > 0.999
    assert torch.allclose(lp1[same], lp2[same], atol=1e-5)
    assert bool(mask.gather(1, a1[:, None]).all())
    # loss + gradients
    values = torch.randn(N, device=dev)
    adv, ret = torch.randn(N, device=dev), torch.randn(N, device=dev)
    f1, W1, b1, v1 = (t.clone().requires_grad_(True) for t in (feat, W, b, values))
    loss1, st1 = O.policy_head_ppo_loss(f1, W1, b1, v1, bits, a1, lp1 - 0.1, adv, ret, None, 0.2, 0.5, 0.01, False,
                                        dense_rows_possible=False)
    loss1.backward()
    f2, W2, b2, v2 = (t.clone().requires_grad_(True) for t in (feat, W, b, values))
    loss2, st2 = O.ppo_loss(f2 @ W2.t() + b2, v2, bits, a1, lp1 - 0.1, adv, ret, None, 0.2, 0.5, 0.01, False)
    loss2.backward()
    assert torch.allclose(st1[:6], st2[:6], rtol=1e-4, atol=1e-6)
    for x, y, name in ((f1, f2, "dfeat"), (W1, W2, "dW"), (b1, b2, "dbias"), (v1, v2, "dvalues")):
        np.testing.assert_allclose(x.grad.cpu().numpy(), y.grad.cpu().numpy(), rtol=2e-4, atol=1e-7, err_msg=name)
    # bit-reproducible: no float atomics anywhere in the backward
    f3, W3, b3, v3 = (t.clone().requires_grad_(True) for t in (feat, W, b, values))
    loss3, _ = O.policy_head_ppo_loss(f3, W3, b3, v3, bits, a1, lp1 - 0.1, adv, ret, None, 0.2, 0.5, 0.01, False,
                                      dense_rows_possible=False)
    loss3.backward()
    assert torch.equal(W1.grad, W3.grad) and torch.equal(f1.grad, f3.grad) and torch.equal(b1.grad, b3.grad)


@pytest.mark.parametrize("B,K", [(1, 4), (3, 6), (33, 162), (64, 164), (65, 160), (257, 162), (300, 1312), (1025, 30), (5000, 162)])
def test_gpu_fused_head_shape_sweep_against_the_logits_kernels(B, K):
    """edge shapes: K with and without the trailing pair (K % 4 == 2), K = 2, the largest K, batches that are not a
    multiple of 32 / 64 / 256 (ragged last dword, last segment, last window of the transposed bit matrix), B = 1 and a
    batch larger than 4096 (more than 8 dwords per segment).  Reference = the logits-based HIP kernels on feat @ W.T + b
    computed in float64 and rounded once."""
    from shogidrl_amd import ops as O
    g = torch.Generator(device="cuda:0").manual_seed(B * 7919 + K)
    dev = "cuda:0"
    feat = torch.randn(B, K, device=dev, generator=g)
    W = torch.randn(A, K, device=dev, generator=g) * (1.5 / K ** 0.5)
    b = torch.randn(A, device=dev, generator=g) * 0.3
    # masks: 1 .. 120 legal actions per row, clustered like real move lists (runs of consecutive ids), a few shared hot actions
    mask = torch.zeros(B, A, dtype=torch.bool, device=dev)
    starts = torch.randint(0, A - 40, (B, 6), device=dev, generator=g)
    lens = torch.randint(1, 21, (B, 6), device=dev, generator=g)
    ar = torch.arange(A, device=dev)
    for c in range(6):
        mask |= (ar[None, :] >= starts[:, c:c + 1]) & (ar[None, :] < (starts[:, c:c + 1] + lens[:, c:c + 1]))
    mask[:, 4242] = True                                   # legal everywhere: the heavy-action path whenever B > 64
    mask[::3, 13526] = True                                # the last action id
    logits = (feat.double() @ W.double().t() + b.double()).float()
    values = torch.randn(B, device=dev, generator=g)
    adv, ret = torch.randn(B, device=dev, generator=g), torch.randn(B, device=dev, generator=g)
    a1, lp1 = O.policy_head_sample(feat, W, b, mask, 5, 2)
    a2, lp2 = O.masked_sample(logits, mask, 5, 2)
    same = a1 == a2
    assert float(same.float().mean()) >= 0.99 and bool(mask.gather(1, a1[:, None]).all())
    assert torch.allclose(lp1[same], lp2[same], atol=2e-5)
    e1, h1 = O.policy_head_evaluate(feat, W, b, mask, a2)
    e2, h2 = O.evaluate_actions(logits, mask, a2)
    assert torch.allclose(e1, e2, atol=2e-5) and torch.allclose(h1, h2, atol=1e-4, rtol=1e-4)
    stats, dfeat, dW, dbias, dval = O.policy_head_loss_grads(feat, W, b, values, mask, a2, lp2 - 0.05, adv, ret, None, 0.2, 0.5,
                                                             0.01, False, dense_rows_possible=False)
    lg = logits.clone().requires_grad_(True)
    v2 = values.clone().requires_grad_(True)
    loss2, st2 = O.ppo_loss(lg, v2, mask, a2, lp2 - 0.05, adv, ret, None, 0.2, 0.5, 0.01, False)
    loss2.backward()
    dl = lg.grad.double()
    assert torch.allclose(stats[:6], st2[:6], rtol=2e-4, atol=2e-6)
    scale = lambda t: float(t.abs().max()) + 1e-12
    for got, want, name in ((dfeat, dl @ W.double(), "dfeat"), (dW, dl.t() @ feat.double(), "dW"), (dbias, dl.sum(0), "dbias"),
                            (dval, v2.grad.double(), "dvalues")):
        err = float((got.double() - want).abs().max())
        assert err <= 3e-4 * scale(want) + 1e-7, (name, err, scale(want))
    assert bool((dW[~mask.any(0)] == 0).all())


def test_fused_head_rejects_unsupported_shapes():
    from shogidrl_amd import ops as O
    from shogidrl_amd._lib import KeiseiHipError
    dev = "cuda:0"
    mask = torch.zeros(2, A, dtype=torch.bool, device=dev)
    mask[:, 5] = True
    with pytest.raises(ValueError):
        O.policy_head_sample(torch.zeros(2, 163, device=dev), torch.zeros(A, 163, device=dev), None, mask, 1, 0)      # odd K
    with pytest.raises(ValueError):
        O.policy_head_sample(torch.zeros(2, 1314, device=dev), torch.zeros(A, 1314, device=dev), None, mask, 1, 0)    # K too large
    with pytest.raises(TypeError):
        O.policy_head_sample(torch.zeros(2, 162, device=dev).half(), torch.zeros(A, 162, device=dev), None, mask, 1, 0)
    with pytest.raises(KeiseiHipError):
        O.policy_head_sample(torch.zeros(2, 162), torch.zeros(A, 162), None, mask.cpu(), 1, 0)                         # CPU tensors: no CPU path


def test_gpu_fused_head_at_production_size_against_plain_pytorch_autograd():
    """4096 rows of self-play masks, K = 162: the fused op against the REFERENCE'S OWN arithmetic written in plain PyTorch
    float32 on the GPU (evaluate_actions: where(mask, logits, -inf) -> softmax -> Categorical(probs) log_prob / entropy,
    base_actor_critic.py:139-184; PPOAgent.learn's loss lines, ppo_agent.py:340-383) and torch.autograd's gradients."""
    from shogidrl_amd import ops as O
    from shogidrl_amd.env import VecShogiEnv
    torch.manual_seed(2)
    N, K, dev = 4096, 162, "cuda:0"
    env = VecShogiEnv(N, 500, device=dev)
    env.reset()
    zeros = torch.zeros(N, A, device=dev)
    for t in range(40):
        a, _ = O.masked_sample(zeros, env.legal_bits, 11, t)
        env.step(a)
    del zeros
    bits, mask = env.legal_bits, env.legal_mask
    feat = torch.randn(N, K, device=dev)
    W = torch.randn(A, K, device=dev) * 0.12
    b = torch.randn(A, device=dev) * 0.2
    values, adv, ret = torch.randn(N, device=dev), torch.randn(N, device=dev), torch.randn(N, device=dev)
    old_values = values + 0.3 * torch.randn(N, device=dev)
    actions, lp = O.policy_head_sample(feat, W, b, bits, 3, 1)
    old_lp = lp + 0.15 * torch.randn(N, device=dev)
    for value_clip in (False, True):
        stats, dfeat, dW, dbias, dval = O.policy_head_loss_grads(feat, W, b, values, bits, actions, old_lp, adv, ret, old_values,
                                                                 0.2, 0.5, 0.01, value_clip, dense_rows_possible=False)
        f, w, bb, v = (t.clone().requires_grad_(True) for t in (feat, W, b, values))
        logits = f @ w.t() + bb
        probs = torch.softmax(torch.where(mask, logits, torch.full_like(logits, float("-inf"))), dim=-1)
        dist = torch.distributions.Categorical(probs=probs)
        new_lp, ent = dist.log_prob(actions), dist.entropy()
        ratio = torch.exp(new_lp - old_lp)
        policy_loss = -torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).mean()
        if value_clip:
            vc = old_values + torch.clamp(v - old_values, -0.2, 0.2)
            value_loss = torch.max(torch.nn.functional.mse_loss(v, ret), torch.nn.functional.mse_loss(vc, ret))
        else:
            value_loss = torch.nn.functional.mse_loss(v, ret)
        entropy_loss = -ent.mean()
        loss = policy_loss + 0.5 * value_loss + 0.01 * entropy_loss
        loss.backward()
        want = [float(x.detach()) for x in (loss, policy_loss, value_loss, entropy_loss, (old_lp - new_lp).mean(),
                                   ((ratio - 1.0).abs() > 0.2).float().mean())]
        np.testing.assert_allclose(stats[:6].cpu().numpy(), want, rtol=1e-4, atol=2e-6)
        for got, ref, name in ((dfeat, f.grad, "dfeat"), (dW, w.grad, "dW"), (dbias, bb.grad, "dbias"), (dval, v.grad, "dvalues")):
            err = float((got - ref).abs().max())
            assert err <= 3e-4 * float(ref.abs().max()) + 1e-8, (name, value_clip, err, float(ref.abs().max()))
