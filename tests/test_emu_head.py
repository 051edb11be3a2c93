"""Fused policy-head kernel sources (shogidrl_amd/csrc/ksh_head_kernels.h) executed on the HOST through the fiber
emulator (MFMA included), checked against the reference's golden vectors and the float64 oracle."""
import numpy as np
import pytest

from tests import headcheck
from tests.emu.emu import EmuOps

ops = EmuOps()


def test_emu_pack_mask_kernel():
    headcheck.check_pack_mask(ops)


def test_emu_head_loss_golden_reference_autograd():
    # two of the seven golden cases on the (slow) emulator: tests/test_gpu_head.py runs all of them through the C-ABI
    import numpy as np
    from tests import ppocheck
    z = np.load(ppocheck.os.path.join(ppocheck.G, "ppo_loss.npz"))
    n = int(z["n_cases"])
    small = sorted(range(n), key=lambda ci: z[f"c{ci}_mask_packed"].shape[0])[:1]     # the smallest batch ...
    dense = [ci for ci in range(n) if np.unpackbits(z[f"c{ci}_mask_packed"], axis=1).sum(1).max() > 1024][:1]   # ... and a dense one
    headcheck.check_head_loss_golden(ops, cases=sorted(set(small + dense)))


# (K = 1296, the tiny CNN's head, and larger batches run on the GPU only: every emulated call walks all 13 527 actions)
# (the K = 10 and K = 190 cases put dense rows through the tile kernel's edge logic: a reduction shorter than one K-step, a k tail
# that is not a multiple of four, quads that would leave the last row of `feat` / `W` / the scratch)
@pytest.mark.parametrize("B,K,vc,dense,empty", [(5, 162, 0, 0, 0), (6, 162, 1, 1, 1), (3, 10, 0, 2, 0), (4, 190, 1, 3, 1)])
def test_emu_head_loss_vs_oracle(B, K, vc, dense, empty):
    headcheck.check_head_loss_vs_oracle(ops, B, K, seed=B + K, value_clip=vc, dense_rows=dense, empty_rows=empty)


def test_emu_head_loss_peaked_softmax_clamped_entries():
    """probabilities below eps and above 1 - eps, down to float32 probabilities of exactly 0 for the taken action (gradient
    0, not 0 / 0); the GPU suite runs more sizes"""
    headcheck.check_head_loss_vs_oracle(ops, 9, 162, seed=21, dense=False, w_scale=25.0, greedy_actions=True)
    headcheck.check_head_loss_vs_oracle(ops, 5, 10, seed=22, dense=False, w_scale=24.0, greedy_actions=False)


def test_emu_head_loss_heavy_actions_split_into_segments_and_no_bias():
    """the weight-gradient pipeline with every action that is legal in more than one sample treated as heavy: its bit row
    (B = 70: three dwords, ragged tail) is split into three one-dword segments whose partial rows the combine kernel adds
    in order; actions legal in exactly one sample take the light path"""
    o = EmuOps()
    o.wpa, o.light_max = 3, 1          # actions legal in one sample are light (written directly), the others heavy
    headcheck.check_head_loss_vs_oracle(o, 70, 162, seed=5, dense=False, with_bias=False)


@pytest.mark.parametrize("mode", [1, 2])
def test_emu_head_corrupt_work_list_is_contained_and_poisons_the_gradient(mode):
    """an impossible work list (counter far beyond the list: what an un-cleared scratch looks like; an item naming an action
    beyond the table) must not take the work kernel outside its buffers (run this file under the AddressSanitizer build to
    see that) and must not pass silently: row 0 of the weight gradient and dbias[0] come back NaN"""
    from tests.emu import emu
    rng = np.random.default_rng(3)
    B, K, A = 6, 162, 13527
    feat = rng.standard_normal((B, K)).astype(np.float32)
    W = (rng.standard_normal((A, K)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(A).astype(np.float32)
    mask = np.zeros((B, A), np.uint8)
    for b in range(B):
        mask[b, rng.choice(A, 30, replace=False)] = 1
    actions = np.array([np.flatnonzero(mask[b])[0] for b in range(B)], np.int64)
    z = np.zeros(B, np.float32)
    emu._plib().emu_head_corrupt_work_list(mode)
    try:
        dfeat, dW, dbias, dv, st = ops.head_loss(feat, W, bias, mask, actions, z - 1.0, z + 1.0, z, z, None, 0.2, 0.5, 0.01, False,
                                                 dense=False)
    finally:
        emu._plib().emu_head_corrupt_work_list(0)
    assert np.isnan(dW[0]).all() and np.isnan(dbias[0])


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
