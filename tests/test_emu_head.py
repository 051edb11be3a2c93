"""Fused policy-head kernel sources (shogidrl_amd/csrc/ksh_head_kernels.h) executed on the HOST through the fiber
emulator (MFMA included), checked against the reference's golden vectors and the float64 oracle."""
import pytest

from tests import headcheck
from tests.emu.emu import EmuOps

ops = EmuOps()


def test_emu_pack_mask_kernel():
    headcheck.check_pack_mask(ops)


def test_emu_head_loss_golden_reference_autograd():
    headcheck.check_head_loss_golden(ops)


@pytest.mark.parametrize("B,K,vc,dense,empty", [(5, 162, 0, 0, 0), (6, 162, 1, 1, 1), (3, 1296, 0, 0, 1)])
def test_emu_head_loss_vs_oracle(B, K, vc, dense, empty):
    headcheck.check_head_loss_vs_oracle(ops, B, K, seed=B + K, value_clip=vc, dense_rows=dense, empty_rows=empty)


def test_emu_head_loss_heavy_actions_split_into_segments_and_no_bias():
    """the weight-gradient pipeline with every action treated as heavy: its bit row (B = 70: three dwords, ragged tail)
    is split into three one-dword segments whose partial rows the combine kernel adds in order"""
    o = EmuOps()
    o.wpa, o.light_max = 3, 0
    headcheck.check_head_loss_vs_oracle(o, 70, 162, seed=5, dense=False, with_bias=False)


def test_emu_head_loss_mixed_light_and_heavy_actions():
    o = EmuOps()
    o.wpa, o.light_max = 2, 1           # actions legal in exactly one sample are light, the others heavy
    headcheck.check_head_loss_vs_oracle(o, 40, 162, seed=6, dense=False)


def test_emu_head_dense_row_without_dense_path_is_loud():
    headcheck.check_head_dense_row_without_dense_path_is_loud(ops)


def test_emu_head_evaluate():
    headcheck.check_head_evaluate(ops)


def test_emu_head_sample_golden():
    headcheck.check_head_sample_golden(ops)


def test_emu_head_sample_vs_oracle():
    headcheck.check_head_sample_vs_oracle(ops)


def test_emu_head_sample_distribution():
    headcheck.check_head_sample_distribution(ops, n_draws=2000)
