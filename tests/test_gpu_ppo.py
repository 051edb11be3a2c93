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


def test_gpu_ppo_loss_bf16(ops):
    ppocheck.check_loss_vs_oracle(ops, 16, seed=21, bf16=True)


def test_gpu_sample_golden(ops):
    ppocheck.check_sample_golden(ops)


def test_gpu_sample_bf16(ops):
    ppocheck.check_sample_bf16(ops)


def test_gpu_sample_distribution(ops):
    ppocheck.check_sample_distribution(ops, n_draws=20000)
