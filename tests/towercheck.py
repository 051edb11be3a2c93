"""Shared checks of the tower-glue kernels (csrc/ksh_tower_kernels.h) against plain PyTorch float32 references of the
same ops (nn.BatchNorm2d / relu / residual add; squeeze-excitation tail of keisei/training/models/resnet_tower.py:15-45).
A backend provides the raw kernel entry points on numpy arrays; tests/test_emu_tower.py runs them on the host-emulated
kernels, tests/test_gpu_tower.py on the HIP library."""
import numpy as np
import torch
import torch.nn.functional as F


def _bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).float().numpy()


def _nhwc(a):     # [N,C,H,W] -> dense [N*H*W, C]
    return np.ascontiguousarray(a.transpose(0, 2, 3, 1)).reshape(-1, a.shape[1])


def _nchw(m, shape):
    N, C, H, W = shape
    return np.ascontiguousarray(m.reshape(N, H, W, C).transpose(0, 3, 1, 2))


def check_bn(be, shape, bf16, relu, residual, seed, mean_offset=0.0, pre_bias=False):
    """training-mode forward (outputs, saved statistics, running-stat update) and backward (dx, dresidual, dw, db)"""
    N, C, H, W = shape
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) * rng.uniform(0.5, 2.0, (1, C, 1, 1)) + rng.uniform(-1, 1, (1, C, 1, 1)) + mean_offset).astype(np.float32)
    res = rng.standard_normal(shape).astype(np.float32) if residual else None
    dy = rng.standard_normal(shape).astype(np.float32)
    w = rng.uniform(0.5, 1.5, C).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, C).astype(np.float32)
    rm0 = rng.uniform(-0.2, 0.2, C).astype(np.float32) + np.float32(mean_offset)
    rv0 = rng.uniform(0.5, 1.5, C).astype(np.float32)
    pb = rng.uniform(-1.0, 1.0, C).astype(np.float32) if pre_bias else None
    if bf16:
        x, dy = _bf16_round(x), _bf16_round(dy)
        res = _bf16_round(res) if residual else None
    # ---- float32 reference on the same (already rounded) inputs
    tx = torch.tensor(x, requires_grad=True)
    tr = torch.tensor(res, requires_grad=True) if residual else None
    tw, tb = torch.tensor(w, requires_grad=True), torch.tensor(b, requires_grad=True)
    trm, trv = torch.tensor(rm0.copy()), torch.tensor(rv0.copy())
    # a folded convolution bias: the reference adds it to x before the normalisation
    ty = F.batch_norm(tx if pb is None else tx + torch.tensor(pb)[None, :, None, None], trm, trv, tw, tb, True, 0.1, 1e-5)
    if residual:
        ty = ty + tr
    if relu:
        ty = F.relu(ty)
    y_ref = ty.detach().numpy()
    ty.backward(torch.tensor(dy))
    # ---- kernels
    out = be.bn_fwd_train(_nhwc(x), None if res is None else _nhwc(res), bf16, w, b, rm0.copy(), rv0.copy(), 0.1, 1e-5, relu,
                          **({"pre_bias": pb} if pre_bias else {}))
    y, rm, rv, mean, invstd = out
    tol = dict(rtol=1e-2, atol=1e-2) if bf16 else dict(rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(_nchw(y, shape), y_ref, **tol)
    xm = x.transpose(1, 0, 2, 3).reshape(C, -1).astype(np.float64)
    np.testing.assert_allclose(mean, xm.mean(1), rtol=1e-5, atol=1e-5 * (1 + abs(mean_offset)))
    np.testing.assert_allclose(invstd, 1.0 / np.sqrt(xm.var(1) + 1e-5), rtol=2e-5)
    # save_mean is the batch mean of x itself; the running mean includes the folded bias
    np.testing.assert_allclose(rm, trm.numpy(), rtol=1e-5, atol=1e-6 * (1 + abs(mean_offset)))
    np.testing.assert_allclose(rv, trv.numpy(), rtol=2e-5, atol=1e-6)
    dx, dres, dw, db = be.bn_bwd(_nhwc(dy), _nhwc(x), y if relu else None, bf16, w, mean, invstd, relu, residual)
    gtol = dict(rtol=2e-2, atol=2e-2) if bf16 else dict(rtol=1e-4, atol=1e-4)
    # the kernel's ReLU mask comes from its own stored output: elements whose sign differs from the reference's by
    # rounding (|y| ~ 1e-6) are excluded from the element-wise comparison
    ok = np.ones(shape, bool) if not relu else ((y_ref > 0) == (_nchw(y, shape) > 0))
    assert ok.mean() > 0.995
    np.testing.assert_allclose(np.where(ok, _nchw(dx, shape), 0), np.where(ok, tx.grad.numpy(), 0), **gtol)
    if residual:
        np.testing.assert_allclose(np.where(ok, _nchw(dres, shape), 0), np.where(ok, tr.grad.numpy(), 0), **gtol)
    n = N * H * W
    stol = dict(rtol=2e-2, atol=2e-2 * np.sqrt(n)) if bf16 else dict(rtol=1e-4, atol=1e-4 * np.sqrt(n))
    np.testing.assert_allclose(dw, tw.grad.numpy(), **stol)
    np.testing.assert_allclose(db, tb.grad.numpy(), **stol)


def check_bn_eval(be, shape, bf16, relu, residual, seed):
    N, C, H, W = shape
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(shape).astype(np.float32)
    res = rng.standard_normal(shape).astype(np.float32) if residual else None
    if bf16:
        x = _bf16_round(x)
        res = _bf16_round(res) if residual else None
    w, b = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.uniform(-0.5, 0.5, C).astype(np.float32)
    rm, rv = rng.uniform(-0.5, 0.5, C).astype(np.float32), rng.uniform(0.5, 1.5, C).astype(np.float32)
    ty = F.batch_norm(torch.tensor(x), torch.tensor(rm), torch.tensor(rv), torch.tensor(w), torch.tensor(b), False, 0.1, 1e-5)
    if residual:
        ty = ty + torch.tensor(res)
    if relu:
        ty = F.relu(ty)
    scale = w / np.sqrt(rv + 1e-5)
    coef = np.stack([scale, rm, b]).astype(np.float32)
    y = be.bn_fwd_eval(_nhwc(x), None if res is None else _nhwc(res), bf16, coef, relu)
    tol = dict(rtol=1e-2, atol=1e-2) if bf16 else dict(rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(_nchw(y, shape), ty.numpy(), **tol)


def check_se(be, shape, bf16, seed):
    """pool, z = relu(x*s + r), and the backward pair, against autograd on the same float32 expression"""
    N, C, H, W = shape
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(shape).astype(np.float32)
    r = rng.standard_normal(shape).astype(np.float32)
    dz = rng.standard_normal(shape).astype(np.float32)
    s = rng.uniform(0.05, 0.95, (N, C)).astype(np.float32)
    dp = rng.standard_normal((N, C)).astype(np.float32)
    if bf16:
        x, r, dz = _bf16_round(x), _bf16_round(r), _bf16_round(dz)
    p = be.se_pool(_nhwc(x), bf16, N, H * W, C)
    np.testing.assert_allclose(p, x.mean(axis=(2, 3)), rtol=1e-5, atol=1e-6)
    tx, tr, ts = torch.tensor(x, requires_grad=True), torch.tensor(r, requires_grad=True), torch.tensor(s, requires_grad=True)
    tz = F.relu(tx * ts[:, :, None, None] + tr)
    z = be.se_apply(_nhwc(x), s, _nhwc(r), bf16, N, H * W, C)
    tol = dict(rtol=1e-2, atol=1e-2) if bf16 else dict(rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(_nchw(z, shape), tz.detach().numpy(), **tol)
    # the pooled-mean branch: total loss = <dz, z> + <dp, mean_hw(x)>
    (tz * torch.tensor(dz)).sum().add((tx.mean(dim=(2, 3)) * torch.tensor(dp)).sum()).backward()
    ds = be.se_bwd_reduce(_nhwc(dz), z, _nhwc(x), bf16, N, H * W, C)
    dx, dr = be.se_bwd_apply(_nhwc(dz), z, s, dp, bf16, N, H * W, C)
    z_ref = tz.detach().numpy()
    ok = (z_ref > 0) == (_nchw(z, shape) > 0)
    assert ok.mean() > 0.995
    gtol = dict(rtol=2e-2, atol=2e-2) if bf16 else dict(rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(np.where(ok, _nchw(dx, shape), 0), np.where(ok, tx.grad.numpy(), 0), **gtol)
    np.testing.assert_allclose(np.where(ok, _nchw(dr, shape), 0), np.where(ok, tr.grad.numpy(), 0), **gtol)
    np.testing.assert_allclose(ds, ts.grad.numpy(), rtol=2e-2 if bf16 else 1e-4, atol=(0.3 if bf16 else 1e-4))


def check_bn_se(be, shape, bf16, seed):
    """The merged pair: y = bn(x + pb) with pooled = mean_hw(y) from one pass, and the backward of
    z = relu(y * s + r) (+ a gradient dp reaching the pooled mean) down to dx, dr, dw, db.  The reference is autograd on
    the same float32 expressions, staged the way the data is staged: the SE tail is differentiated at the y the kernel
    stored and its gradient enters the batch-norm backward rounded to the activation dtype (in bf16 a reference that
    keeps y in float32 flips ReLU decisions of elements within rounding distance of zero, an O(1) change per element
    that says nothing about the kernels).  s is held fixed; its own gradient path is checked by check_se."""
    N, C, H, W = shape
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(shape) * rng.uniform(0.5, 2.0, (1, C, 1, 1)) + rng.uniform(-1, 1, (1, C, 1, 1))).astype(np.float32)
    r = rng.standard_normal(shape).astype(np.float32)
    dz = rng.standard_normal(shape).astype(np.float32)
    s = rng.uniform(0.05, 0.95, (N, C)).astype(np.float32)
    dp = rng.standard_normal((N, C)).astype(np.float32)
    w, b = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.uniform(-0.5, 0.5, C).astype(np.float32)
    pb = rng.uniform(-1.0, 1.0, C).astype(np.float32)
    rm0, rv0 = rng.uniform(-0.2, 0.2, C).astype(np.float32), rng.uniform(0.5, 1.5, C).astype(np.float32)
    if bf16:
        x, r, dz = _bf16_round(x), _bf16_round(r), _bf16_round(dz)
    # ---- forward
    tx = torch.tensor(x, requires_grad=True)
    tw, tb = torch.tensor(w, requires_grad=True), torch.tensor(b, requires_grad=True)
    trm, trv = torch.tensor(rm0.copy()), torch.tensor(rv0.copy())
    ty = F.batch_norm(tx + torch.tensor(pb)[None, :, None, None], trm, trv, tw, tb, True, 0.1, 1e-5)
    y, pooled, rm, rv, mean, invstd = be.bn_pool_fwd_train(_nhwc(x), bf16, N, H * W, w, b, rm0.copy(), rv0.copy(), 0.1, 1e-5, pre_bias=pb)
    tol = dict(rtol=1e-2, atol=1e-2) if bf16 else dict(rtol=2e-5, atol=2e-5)
    y_k = _nchw(y, shape)
    np.testing.assert_allclose(y_k, ty.detach().numpy(), **tol)
    np.testing.assert_allclose(pooled, y_k.mean(axis=(2, 3)), rtol=1e-5, atol=1e-6)                 # mean of the stored values
    np.testing.assert_allclose(rm, trm.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv, trv.numpy(), rtol=2e-5, atol=1e-6)
    # ---- SE tail at the stored y
    ty2, tr = torch.tensor(y_k, requires_grad=True), torch.tensor(r, requires_grad=True)
    tz = F.relu(ty2 * torch.tensor(s)[:, :, None, None] + tr)
    ((tz * torch.tensor(dz)).sum() + (ty2.mean(dim=(2, 3)) * torch.tensor(dp)).sum()).backward()
    z = be.se_apply(y, s, _nhwc(r), bf16, N, H * W, C)
    np.testing.assert_allclose(_nchw(z, shape), tz.detach().numpy(), **(dict(rtol=1e-2, atol=1e-2) if bf16 else dict(rtol=1e-6, atol=1e-6)))
    dy2_ref = _bf16_round(ty2.grad.numpy()) if bf16 else ty2.grad.numpy()
    ty.backward(torch.tensor(dy2_ref))
    # ---- backward of the pair
    dx, dr, dw, db = be.se_bn_bwd(_nhwc(dz), z, s, dp, _nhwc(x), bf16, N, H * W, w, mean, invstd)
    ok = (tz.detach().numpy() > 0) == (_nchw(z, shape) > 0)              # ties at exactly zero only
    assert ok.mean() > 0.9999
    gtol = dict(rtol=2e-2, atol=2e-2) if bf16 else dict(rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(np.where(ok, _nchw(dr, shape), 0), np.where(ok, tr.grad.numpy(), 0), **gtol)
    np.testing.assert_allclose(np.where(ok, _nchw(dx, shape), 0), np.where(ok, tx.grad.numpy(), 0), **gtol)
    n = N * H * W
    stol = dict(rtol=2e-2, atol=2e-2 * np.sqrt(n)) if bf16 else dict(rtol=1e-4, atol=1e-4 * np.sqrt(n))
    np.testing.assert_allclose(dw, tw.grad.numpy(), **stol)
    np.testing.assert_allclose(db, tb.grad.numpy(), **stol)


CASES_BN = [  # shape, bf16, relu, residual
    ((5, 32, 9, 9), False, True, False), ((3, 256, 9, 9), False, False, False), ((4, 64, 9, 9), False, True, True),
    ((2, 16, 9, 9), False, False, True), ((6, 256, 9, 9), True, True, False), ((3, 128, 9, 9), True, False, False),
    ((4, 32, 9, 9), True, True, True), ((1, 8, 3, 5), False, True, False),
]
