"""Learner-side kernel sources (shogidrl_amd/csrc/ksh_ppo_kernels.h) executed on the HOST through the
fiber emulator, checked against golden vectors from the reference and against the oracle."""
import pytest

from tests import ppocheck
from tests.emu.emu import EmuOps

ops = EmuOps()


def test_oracle_ppo_loss_pinned_to_reference_autograd():
    ppocheck.check_oracle_loss_golden()


def test_emu_gae_golden_bit_exact():
    ppocheck.check_gae_golden(ops)


def test_emu_gae_vs_oracle_batched():
    ppocheck.check_gae_vs_oracle(ops, T=37, N=300, seed=3)


def test_emu_ppo_loss_golden():
    ppocheck.check_loss_golden(ops)


@pytest.mark.parametrize("B,vc,dense,empty", [(5, 0, 0, 0), (7, 1, 1, 1), (3, 0, 0, 3)])
def test_emu_ppo_loss_vs_oracle(B, vc, dense, empty):
    ppocheck.check_loss_vs_oracle(ops, B, seed=B, value_clip=vc, dense_rows=dense, empty_rows=empty)


def test_emu_ppo_loss_peaked_softmax_underflowing_probabilities():
    """logits spread over +-150: taken actions with float32 probability 0 (the gradient used to be y * (pol / y) = 0 / 0),
    one probability above 1 - eps per row"""
    ppocheck.check_loss_vs_oracle(ops, 6, seed=5, logit_scale=40.0, dense_rows=1)


def test_emu_ppo_loss_bf16():
    ppocheck.check_loss_vs_oracle(ops, 4, seed=21, bf16=True)


def test_emu_sample_golden():
    ppocheck.check_sample_golden(ops)


def test_emu_sample_bf16():
    ppocheck.check_sample_bf16(ops)


def test_emu_sample_distribution():
    ppocheck.check_sample_distribution(ops, n_draws=2000)


def test_emu_packed_masks_give_identical_results():
    """the kernels fed bit-packed masks produce exactly what they produce from byte masks"""
    import numpy as np
    p = EmuOps()
    p.packed = True
    ppocheck.check_loss_golden(p)
    ppocheck.check_sample_golden(p)
    logits, mask, actions, old_logp, adv, ret, values, old_values = ppocheck.random_loss_inputs(6, 99, dense_rows=1, empty_rows=1)
    a = ops.ppo_loss(logits, mask, actions, old_logp, adv, ret, values, old_values, 0.2, 0.5, 0.01, 0)
    b = p.ppo_loss(logits, mask, actions, old_logp, adv, ret, values, old_values, 0.2, 0.5, 0.01, 0)
    for x, y in zip(a, b):                              # same set of legal ids, accumulated in a different order
        np.testing.assert_allclose(x, y, rtol=1e-5, atol=1e-7)
    (a1, lp1), (a2, lp2) = ops.masked_sample(logits, mask, 5, 6, False), p.masked_sample(logits, mask, 5, 6, False)
    assert np.array_equal(a1, a2)                       # Gumbel-max is order independent: identical actions
    np.testing.assert_allclose(lp1, lp2, atol=1e-6)     # the soft-max sum is accumulated in a different order


def test_sampler_uniform_is_strictly_inside_the_unit_interval():
    """with 24 random bits + 0.5 the largest draw rounded to u == 1.0, i.e. a Gumbel key of +inf that forced
    the action whatever its probability.  The mapping now uses 23 bits + 0.5 (exact in float32): 0 < u < 1 always."""
    import ctypes as C
    import numpy as np
    from tests.emu.emu import _plib
    L = _plib()
    L.emu_uniform_open.restype = C.c_float
    L.emu_uniform_open.argtypes = [C.c_uint]
    L.emu_gumbel_key.restype = C.c_float
    L.emu_gumbel_key.argtypes = [C.c_float, C.c_uint]
    for x in (0xFFFFFFFF, 0xFFFFFE00, 0xFFFFFF00, 0x0, 0x1FF, 0x200, 0x80000000):
        u = L.emu_uniform_open(x)
        assert 0.0 < u < 1.0, (hex(x), u)
        assert np.isfinite(L.emu_gumbel_key(0.0, x)), hex(x)
    assert L.emu_uniform_open(0xFFFFFFFF) == np.float32(1.0) - np.float32(2.0 ** -24)
    assert L.emu_uniform_open(0) == np.float32(2.0 ** -24)


def test_emu_evaluate_actions_vs_oracle_and_golden():
    ppocheck.check_evaluate_actions(ops)
    p = EmuOps()
    p.packed = True
    ppocheck.check_evaluate_actions(p)
