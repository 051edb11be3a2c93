"""The tower-glue kernel sources (shogidrl_amd/csrc/ksh_tower_kernels.h) executed on the HOST through the fiber
emulator, against plain PyTorch float32 references (tests/towercheck.py).  The same checks run against the HIP
library in tests/test_gpu_tower.py."""
import pytest

from tests import towercheck
from tests.emu.emu import EmuTower


@pytest.fixture(scope="module")
def be():
    return EmuTower()


@pytest.mark.parametrize("shape,bf16,relu,residual", towercheck.CASES_BN)
def test_emu_bn_train_fwd_bwd(be, shape, bf16, relu, residual):
    towercheck.check_bn(be, shape, bf16, relu, residual, seed=sum(shape) + bf16 + 2 * relu)


def test_emu_bn_statistics_with_large_mean(be):
    # |mean| >> std: the shifted sums (K = running mean) keep the variance accurate
    towercheck.check_bn(be, (4, 32, 9, 9), False, True, False, seed=3, mean_offset=300.0)


def test_emu_bn_folded_conv_bias(be):
    towercheck.check_bn(be, (4, 64, 9, 9), False, True, False, seed=5, pre_bias=True)
    towercheck.check_bn(be, (3, 32, 9, 9), True, False, True, seed=6, pre_bias=True)


def test_emu_bn_eval(be):
    towercheck.check_bn_eval(be, (3, 64, 9, 9), False, True, True, seed=1)
    towercheck.check_bn_eval(be, (2, 256, 9, 9), True, False, False, seed=2)


@pytest.mark.parametrize("shape,bf16", [((3, 32, 9, 9), False), ((2, 256, 9, 9), True), ((2, 8, 2, 3), False)])
def test_emu_se_tail(be, shape, bf16):
    towercheck.check_se(be, shape, bf16, seed=shape[1])


@pytest.mark.parametrize("shape,bf16", [((5, 32, 9, 9), False), ((3, 256, 9, 9), True), ((2, 8, 2, 3), False)])
def test_emu_merged_bn_se_pair(be, shape, bf16):
    towercheck.check_bn_se(be, shape, bf16, seed=shape[1] + 1)
