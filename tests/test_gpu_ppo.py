"""Parity tests proper for the learner-side HIP kernels (GAE, masked sampling, fused PPO loss) on a real
MI355X through the C-ABI: golden vectors from the reference + the oracle at larger sizes."""
import pytest

from tests import ppocheck

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from tests.gpu_backends import GpuOps
    return GpuOps()


def test_gpu_gae_golden_bit_exact(ops):
    ppocheck.check_gae_golden(ops)


def test_gpu_gae_vs_oracle_full_size(ops):
    ppocheck.check_gae_vs_oracle(ops, T=64, N=4096, seed=3)


def test_gpu_ppo_loss_golden(ops):
    ppocheck.check_loss_golden(ops)


@pytest.mark.parametrize("B,vc,dense,empty", [(64, 0, 0, 0), (33, 1, 2, 1), (3, 0, 0, 3), (512, 0, 1, 0)])
def test_gpu_ppo_loss_vs_oracle(ops, B, vc, dense, empty):
    ppocheck.check_loss_vs_oracle(ops, B, seed=B, value_clip=vc, dense_rows=dense, empty_rows=empty)


def test_gpu_ppo_loss_peaked_softmax_underflowing_probabilities(ops):
    """logits spread over +-150: taken actions with float32 probability 0 (the gradient used to be y * (pol / y) = 0 / 0),
    one probability above 1 - eps per row"""
    ppocheck.check_loss_vs_oracle(ops, 24, seed=5, logit_scale=40.0, dense_rows=1)


def test_gpu_ppo_loss_bf16(ops):
    ppocheck.check_loss_vs_oracle(ops, 16, seed=21, bf16=True)


def test_gpu_sample_golden(ops):
    ppocheck.check_sample_golden(ops)


def test_gpu_sample_bf16(ops):
    ppocheck.check_sample_bf16(ops)


def test_gpu_sample_distribution(ops):
    ppocheck.check_sample_distribution(ops, n_draws=20000)


def test_gpu_packed_masks_match_byte_masks():
    """env kernel packed output == packbits(byte output); sampler / loss give the same results from either form"""
    import numpy as np
    import torch
    from shogidrl_amd import ops as O
    from shogidrl_amd.env import VecShogiEnv
    N = 130
    e_b, e_p = VecShogiEnv(N, 60), VecShogiEnv(N, 60, packed_masks=True)
    e_b.reset()
    _, bits = e_p.reset()
    assert torch.equal(O.unpack_mask(bits), e_b.legal_mask) and torch.equal(O.pack_mask(e_b.legal_mask), bits)
    logits = torch.randn(N, 13527, device="cuda:0")
    for t in range(90):                                       # max_moves 60: resets happen
        a1, lp1 = O.masked_sample(logits, e_b.legal_mask, 3, t)
        a2, lp2 = O.masked_sample(logits, e_p.legal_bits, 3, t)
        assert torch.equal(a1, a2) and torch.allclose(lp1, lp2, atol=1e-6)
        o1, r1, d1, i1 = e_b.step(a1)
        o2, r2, d2, i2 = e_p.step(a2)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(i1["reason"], i2["reason"])
        assert torch.equal(O.unpack_mask(e_p.legal_bits), e_b.legal_mask)
    lg = torch.randn(N, 13527, device="cuda:0")
    outs = []
    a, lp = O.masked_sample(lg, e_b.legal_mask, 9, 0)        # same inputs for both forms
    for mask in (e_b.legal_mask, e_p.legal_bits):
        x = lg.clone().requires_grad_(True)
        v = torch.zeros(N, device="cuda:0", requires_grad=True)
        loss, stats = O.ppo_loss(x, v, mask, a, lp - 0.1, torch.ones(N, device="cuda:0"), torch.ones(N, device="cuda:0"))
        loss.backward()
        outs.append((stats.clone(), x.grad.clone(), v.grad.clone()))
    for x, y in zip(*outs):                                   # same legal set, accumulated in a different order
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-7)
    # an illegal action is rejected through the packed previous mask too
    bad = torch.zeros(N, dtype=torch.int64, device="cuda:0")          # id 0 is never legal at these positions' from-square
    legal0 = O.unpack_mask(e_p.legal_bits)[:, 0]
    e_p.step(bad)
    assert torch.equal(e_p.reason == 5, ~legal0)


def test_gpu_rollout_with_packed_masks_learns():
    import torch
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import VecExperienceBuffer
    from shogidrl_amd.env import VecShogiEnv
    from shogidrl_amd.models import TinyActorCritic
    from shogidrl_amd.rollout import SelfPlayRunner
    torch.manual_seed(0)
    cfg = PPOConfig(ppo_epochs=1, minibatch_size=128, seed=3)
    agent = VecPPOAgent(TinyActorCritic(), cfg, "cuda:0")
    env = VecShogiEnv(32, 50, packed_masks=True)
    buf = VecExperienceBuffer(8, 32, cfg.gamma, cfg.lambda_gae, "cuda:0", packed_masks=True)
    run = SelfPlayRunner(env, agent, buf)
    m = run.iteration()
    assert all(torch.isfinite(torch.tensor(v)) for v in m.values()) and buf.legal_masks.dtype == torch.int32
