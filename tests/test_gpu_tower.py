"""Tower-glue kernels on a real MI355X through the C-ABI: (1) the raw entry points against plain PyTorch float32
references (tests/towercheck.py), (2) the model with the fused glue against the same model through ATen's kernels."""
import numpy as np
import pytest
import torch

from tests import towercheck

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def be():
    from tests.gpu_backends import GpuTower
    return GpuTower()


@pytest.mark.parametrize("shape,bf16,relu,residual", towercheck.CASES_BN + [((512, 256, 9, 9), True, True, False),
                                                                           ((300, 64, 9, 9), False, True, True)])
def test_gpu_bn_train_fwd_bwd(be, shape, bf16, relu, residual):
    towercheck.check_bn(be, shape, bf16, relu, residual, seed=sum(shape) + bf16 + 2 * relu)


def test_gpu_bn_statistics_with_large_mean(be):
    towercheck.check_bn(be, (4, 32, 9, 9), False, True, False, seed=3, mean_offset=300.0)


def test_gpu_bn_folded_conv_bias(be):
    towercheck.check_bn(be, (4, 64, 9, 9), False, True, False, seed=5, pre_bias=True)
    towercheck.check_bn(be, (130, 256, 9, 9), True, False, True, seed=6, pre_bias=True)


def test_gpu_bn_eval(be):
    towercheck.check_bn_eval(be, (3, 64, 9, 9), False, True, True, seed=1)
    towercheck.check_bn_eval(be, (200, 256, 9, 9), True, False, False, seed=2)


@pytest.mark.parametrize("shape,bf16", [((3, 32, 9, 9), False), ((130, 256, 9, 9), True), ((2, 8, 2, 3), False)])
def test_gpu_se_tail(be, shape, bf16):
    towercheck.check_se(be, shape, bf16, seed=shape[1])


@pytest.mark.parametrize("shape,bf16", [((5, 32, 9, 9), False), ((1500, 256, 9, 9), True), ((2, 8, 2, 3), False)])
def test_gpu_merged_bn_se_pair(be, shape, bf16):
    towercheck.check_bn_se(be, shape, bf16, seed=shape[1] + 1)      # 1500 samples: two samples per block (spb = 2)


def test_gpu_bad_arguments_are_rejected(be):
    from shogidrl_amd._lib import KeiseiHipError
    x = np.zeros((81, 12), np.float32)                       # C = 12: not a multiple of 8
    with pytest.raises(KeiseiHipError):
        be.se_pool(x, False, 1, 81, 12)
    with pytest.raises(KeiseiHipError):
        be.bn_fwd_eval(np.zeros((81, 24), np.float32), None, False, np.zeros((3, 24), np.float32), True)   # 256 % 3 != 0


def _run_model(fused, autocast, se_ratio, seed=0, steps=2):
    import shogidrl_amd.models as models
    models.FUSED_TOWER_GLUE = fused
    try:
        torch.manual_seed(seed)
        m = models.ResTowerActorCritic(46, 13527, 2, 64, se_ratio).to(DEV).to(memory_format=torch.channels_last)
        g = torch.Generator().manual_seed(seed + 1)
        outs = []
        for _ in range(steps):                       # two steps: the second sees the updated running statistics
            x = (torch.rand(48, 46, 9, 9, generator=g) < 0.1).float().to(DEV).contiguous(memory_format=torch.channels_last)
            m.train()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                logits, value = m(x)
            loss = (logits.float() ** 2).mean() + value.float().pow(2).mean() + logits.float()[:, :7].sum() * 1e-3
            m.zero_grad()
            loss.backward()
            outs.append((logits.detach().float().cpu(), value.detach().float().cpu()))
        grads = {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}
        bufs = {k: b.detach().float().cpu() for k, b in m.named_buffers()}
        m.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            ev = m(x)[0].float().cpu()
        return outs, grads, bufs, ev
    finally:
        models.FUSED_TOWER_GLUE = True


@pytest.mark.parametrize("se_ratio", [0.25, None])
def test_model_with_fused_glue_matches_aten_fp32(se_ratio):
    a = _run_model(True, False, se_ratio)
    b = _run_model(False, False, se_ratio)
    for (la, va), (lb, vb) in zip(a[0], b[0]):
        assert torch.allclose(la, lb, rtol=1e-4, atol=1e-4) and torch.allclose(va, vb, rtol=1e-4, atol=1e-4)
    for k in b[1]:
        scale = b[1][k].abs().max().item() + 1e-6
        # (+1e-6: biases of convolutions that feed a BatchNorm have a true gradient of zero -- both sides are noise)
        assert (a[1][k] - b[1][k]).abs().max().item() <= 2e-3 * scale + 1e-6, k
    for k in b[2]:
        assert torch.allclose(a[2][k], b[2][k], rtol=1e-4, atol=1e-5), k           # running stats, num_batches_tracked
    assert torch.allclose(a[3], b[3], rtol=1e-4, atol=1e-4)                        # eval-mode forward


def test_model_with_fused_glue_bf16_autocast_is_as_accurate_as_aten():
    """Two bf16 pipelines differ from each other by rounding noise, so both are measured against the float32 run: the
    fused glue must not be less accurate than the framework kernels it replaces."""
    ref = _run_model(False, False, 0.25)
    fused = _run_model(True, True, 0.25)
    aten = _run_model(False, True, 0.25)
    worst = 0.0
    for i in range(len(ref[0])):
        e_f = (fused[0][i][0] - ref[0][i][0]).abs().max().item()
        e_a = (aten[0][i][0] - ref[0][i][0]).abs().max().item()
        assert e_f <= 3.0 * e_a + 1e-3, (i, e_f, e_a)
    for k in ref[1]:
        e_f = (fused[1][k] - ref[1][k]).abs().max().item()
        e_a = (aten[1][k] - ref[1][k]).abs().max().item()
        scale = ref[1][k].abs().max().item()
        assert e_f <= 3.0 * e_a + 0.02 * scale + 1e-4, (k, e_f, e_a, scale)       # per-tensor maxima of rounding noise
        worst = max(worst, e_f / (scale + 1e-6))
    for k in ref[2]:
        e_f = (fused[2][k] - ref[2][k]).abs().max().item()
        e_a = (aten[2][k] - ref[2][k]).abs().max().item()
        assert e_f <= 3.0 * e_a + 1e-3, (k, e_f, e_a)
