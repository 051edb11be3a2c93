"""Fused tower glue over the HIP kernels of csrc/ksh_tower_kernels.h: what sits between the convolutions of the
reference's ActorCriticResTower (keisei/training/models/resnet_tower.py):

  bn_act(x, bn, residual=None, relu=False)   <- `F.relu(self.bn1(self.conv1(x)))`, `self.bn2(...)`, stem (:39-40, 82)
  se_tail(x, se, residual)                   <- `out = self.se(out); out += x; return F.relu(out)` (:41-45, 21-26)

Activations must be CUDA tensors [N,C,H,W] that are dense in channels_last memory format (what MIOpen's NHWC
convolutions produce), float32 or bfloat16, with C % 8 == 0 and 256 % (C/8) == 0; `supported()` tells.  The module
parameters / buffers are the ordinary nn.BatchNorm2d / Conv2d ones, so state dicts are unchanged.  Semantics are
those of the PyTorch modules (training-mode batch statistics with biased variance, running statistics blended with
momentum using the unbiased variance, eval mode from the running statistics); sums are accumulated in float32 /
double, so results agree with the framework kernels to rounding (tests/test_gpu_tower.py)."""
import torch

from ._lib import check, lib

_workspaces = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _workspace(device, C):
    """Scratch for the batch-norm partial sums (statistics kernel -> finalize kernel).  It is reused by consecutive calls
    on ONE stream; two streams (the asynchronous learner's rollout stream next to learn() on the main stream) must not
    share it, so it is cached per stream.  While a hipGraph is being captured nothing is cached: the scratch then comes
    from the graph's private pool and must not outlive the capture in a global."""
    n = lib().ksh_tower_workspace_bytes(C) // 4
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(n, dtype=torch.float32, device=device)
    key = (device, C, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None:
        ws = torch.empty(n, dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws


def _dense_nhwc(t):
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def supported(x):
    """True when `x` can go through the fused kernels as it is."""
    if not x.is_cuda or x.dim() != 4 or x.dtype not in (torch.float32, torch.bfloat16):
        return False
    C = x.shape[1]
    return C % 8 == 0 and 256 % (C // 8) == 0 and _dense_nhwc(x) and x.data_ptr() % 16 == 0 and x.numel() > 0


def _code(t):
    return 0 if t.dtype == torch.float32 else 1


def _like(x):
    return torch.empty_like(x, memory_format=torch.channels_last)


def _as(t, ref):
    """gradient tensors arrive in whatever layout autograd produced; bring them to ref's dtype / channels_last"""
    if t.dtype != ref.dtype:
        t = t.to(ref.dtype)
    if not _dense_nhwc(t) or t.data_ptr() % 16:
        t = t.contiguous(memory_format=torch.channels_last)
        if t.data_ptr() % 16:
            t = t.clone(memory_format=torch.channels_last)
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


class _BNActTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, residual, pre_bias, momentum, eps, relu):
        N, C, H, W = x.shape
        M = N * H * W
        y = _like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        w = None if weight is None else weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        pb = None if pre_bias is None else pre_bias.detach().float().contiguous()
        check(lib().ksh_bn_fwd_train(x.data_ptr(), _ptr(residual), y.data_ptr(), _code(x), M, C, _ptr(pb), _ptr(w), _ptr(b),
                                     _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), int(relu),
                                     mean.data_ptr(), invstd.data_ptr(), _workspace(x.device, C).data_ptr(), _stream()),
              "ksh_bn_fwd_train")
        ctx.save_for_backward(x, y if relu else None, w, mean, invstd)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        ctx.has_w, ctx.has_b = weight is not None, bias is not None
        ctx.pre_bias = None if pre_bias is None else (pre_bias.shape, pre_bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, w, mean, invstd = ctx.saved_tensors
        N, C, H, W = x.shape
        dy = _as(dy, x)
        dx = _like(x)
        dres = _like(x) if (ctx.has_res and ctx.relu) else None
        dw = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        check(lib().ksh_bn_bwd(dy.data_ptr(), x.data_ptr(), _ptr(y), dx.data_ptr(), _ptr(dres), _code(x), N * H * W, C,
                               _ptr(w), mean.data_ptr(), invstd.data_ptr(), int(ctx.relu), dw.data_ptr(), db.data_ptr(),
                               _workspace(x.device, C).data_ptr(), _stream()), "ksh_bn_bwd")
        if ctx.has_res and not ctx.relu:
            dres = dy                                        # no ReLU in between: the residual's gradient is dy itself
        # a bias added before the normalisation cannot change the output: its gradient is exactly zero
        dpb = None if ctx.pre_bias is None else torch.zeros(ctx.pre_bias[0], dtype=ctx.pre_bias[1], device=x.device)
        return (dx, dw if ctx.has_w else None, db if ctx.has_b else None, None, None, dres if ctx.has_res else None, dpb,
                None, None, None)


def bn_act(x, bn, residual=None, relu=False, pre_bias=None):
    """[relu](bn(x [+ pre_bias]) [+ residual]) for an nn.BatchNorm2d `bn` through the fused kernels; `pre_bias` [C] is
    the bias of the convolution that produced `x`, folded into the statistics instead of being added (and reduced in
    the backward) by separate kernels.  Returns None when this call
    is not covered (CPU tensors, channel counts such as the 2-plane heads, eval mode under autograd, cumulative-average
    momentum) -- the caller then runs the same computation through ATen (models._BN)."""
    if not supported(x) or bn.momentum is None:
        return None
    if residual is not None and not (supported(residual) and residual.dtype == x.dtype and residual.shape == x.shape):
        return None
    training = bn.training or not bn.track_running_stats
    if training:
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        rm = bn.running_mean if bn.track_running_stats else None
        rv = bn.running_var if bn.track_running_stats else None
        return _BNActTrain.apply(x, bn.weight, bn.bias, rm, rv, residual, pre_bias, bn.momentum, bn.eps, relu)
    if torch.is_grad_enabled() and (x.requires_grad or (bn.weight is not None and bn.weight.requires_grad)
                                    or (pre_bias is not None and pre_bias.requires_grad)):
        return None
    with torch.no_grad():
        scale = torch.rsqrt(bn.running_var.float() + bn.eps)
        if bn.weight is not None:
            scale = scale * bn.weight.float()
        bias = bn.bias.float() if bn.bias is not None else torch.zeros_like(scale)
        mean = bn.running_mean.float() if pre_bias is None else bn.running_mean.float() - pre_bias.float()
        coef = torch.stack([scale, mean, bias]).contiguous()
        N, C, H, W = x.shape
        y = _like(x)
        check(lib().ksh_bn_fwd_eval(x.data_ptr(), _ptr(residual), y.data_ptr(), _code(x), N * H * W, C, coef.data_ptr(),
                                    int(relu), _stream()), "ksh_bn_fwd_eval")
    return y


def _tn_matmul(a, b, chunks=32):
    """a[N,P], b[N,Q] -> a.T @ b [P,Q].  The reduction runs over the batch (N = 4096) while P x Q is tiny, a shape the
    GEMM library serves with one or two workgroups (58 us); splitting N into `chunks` batched products and summing
    them keeps the whole chip busy."""
    N = a.shape[0]
    if N % chunks or N // chunks < 16:
        return a.t() @ b
    return torch.bmm(a.view(chunks, N // chunks, -1).transpose(1, 2), b.view(chunks, N // chunks, -1)).sum(0)


class _SETail(torch.autograd.Function):
    """z = relu(x * sigmoid(fc2(relu(fc1(mean_hw(x))))) + r); the two 1x1 convolutions act on [N,C] vectors and are
    evaluated (and differentiated) here as small float32 matrix products."""

    @staticmethod
    def forward(ctx, x, r, w1, b1, w2, b2):
        N, C, H, W = x.shape
        L = lib()
        p = torch.empty((N, C), dtype=torch.float32, device=x.device)
        check(L.ksh_se_pool(x.data_ptr(), _code(x), N, H * W, C, p.data_ptr(), _stream()), "ksh_se_pool")
        with torch.autocast(device_type="cuda", enabled=False):       # the [N,C] excitation vectors stay in float32
            W1 = w1.detach().float().reshape(w1.shape[0], C)
            W2 = w2.detach().float().reshape(C, w2.shape[1])
            h = torch.relu(torch.addmm(b1.detach().float(), p, W1.t()))
            s = torch.sigmoid(torch.addmm(b2.detach().float(), h, W2.t())).contiguous()
        z = _like(x)
        check(L.ksh_se_apply(x.data_ptr(), s.data_ptr(), r.data_ptr(), z.data_ptr(), _code(x), N, H * W, C, _stream()),
              "ksh_se_apply")
        ctx.save_for_backward(x, z, p, h, s, W1, W2)
        ctx.wshapes = (w1.shape, w2.shape, w1.dtype, w2.dtype, b1.dtype, b2.dtype)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, z, p, h, s, W1, W2 = ctx.saved_tensors
        N, C, H, W = x.shape
        L = lib()
        dz = _as(dz, x)
        ds = torch.empty((N, C), dtype=torch.float32, device=x.device)
        check(L.ksh_se_bwd_reduce(dz.data_ptr(), z.data_ptr(), x.data_ptr(), _code(x), N, H * W, C, ds.data_ptr(), _stream()),
              "ksh_se_bwd_reduce")
        with torch.autocast(device_type="cuda", enabled=False):
            du = ds * s * (1.0 - s)                      # through the sigmoid
            dW2 = _tn_matmul(du, h)
            db2 = du.sum(0)
            dh = (du @ W2) * (h > 0)                     # through the ReLU
            dW1 = _tn_matmul(dh, p)
            db1 = dh.sum(0)
            dp = (dh @ W1).contiguous()
        dx = _like(x)
        dr = _like(x)
        check(L.ksh_se_bwd_apply(dz.data_ptr(), z.data_ptr(), s.data_ptr(), dp.data_ptr(), dx.data_ptr(), dr.data_ptr(),
                                 _code(x), N, H * W, C, _stream()), "ksh_se_bwd_apply")
        s1, s2, t1, t2, tb1, tb2 = ctx.wshapes
        return dx, dr, dW1.reshape(s1).to(t1), db1.to(tb1), dW2.reshape(s2).to(t2), db2.to(tb2)


class _BNSETail(torch.autograd.Function):
    """z = relu(se(bn(x [+ pre_bias])) + r) in training mode, with the SE pooling riding on the batch-norm apply pass and
    the batch-norm backward reduction riding on the SE backward apply pass (two HBM passes fewer than bn_act followed
    by se_tail)."""

    @staticmethod
    def forward(ctx, x, r, weight, bias, running_mean, running_var, pre_bias, momentum, eps, w1, b1, w2, b2):
        N, C, H, W = x.shape
        L = lib()
        y = _like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        p = torch.empty((N, C), dtype=torch.float32, device=x.device)
        w = None if weight is None else weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        pb = None if pre_bias is None else pre_bias.detach().float().contiguous()
        check(L.ksh_bn_pool_fwd_train(x.data_ptr(), y.data_ptr(), _code(x), N, H * W, C, _ptr(pb), _ptr(w), _ptr(b),
                                      _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), mean.data_ptr(),
                                      invstd.data_ptr(), p.data_ptr(), _workspace(x.device, C).data_ptr(), _stream()),
              "ksh_bn_pool_fwd_train")
        with torch.autocast(device_type="cuda", enabled=False):
            W1 = w1.detach().float().reshape(w1.shape[0], C)
            W2 = w2.detach().float().reshape(C, w2.shape[1])
            h = torch.relu(torch.addmm(b1.detach().float(), p, W1.t()))
            s = torch.sigmoid(torch.addmm(b2.detach().float(), h, W2.t())).contiguous()
        z = _like(x)
        check(L.ksh_se_apply(y.data_ptr(), s.data_ptr(), r.data_ptr(), z.data_ptr(), _code(x), N, H * W, C, _stream()),
              "ksh_se_apply")
        ctx.save_for_backward(x, y, z, p, h, s, W1, W2, w, mean, invstd)
        ctx.meta = (weight is not None, bias is not None,
                    None if pre_bias is None else (pre_bias.shape, pre_bias.dtype),
                    (w1.shape, w2.shape, w1.dtype, w2.dtype, b1.dtype, b2.dtype))
        return z

    @staticmethod
    def backward(ctx, dz):
        x, y, z, p, h, s, W1, W2, w, mean, invstd = ctx.saved_tensors
        has_w, has_b, pbm, (s1, s2, t1, t2, tb1, tb2) = ctx.meta
        N, C, H, W = x.shape
        L = lib()
        dz = _as(dz, x)
        ds = torch.empty((N, C), dtype=torch.float32, device=x.device)
        check(L.ksh_se_bwd_reduce(dz.data_ptr(), z.data_ptr(), y.data_ptr(), _code(x), N, H * W, C, ds.data_ptr(), _stream()),
              "ksh_se_bwd_reduce")
        with torch.autocast(device_type="cuda", enabled=False):
            du = ds * s * (1.0 - s)
            dW2 = _tn_matmul(du, h)
            db2 = du.sum(0)
            dh = (du @ W2) * (h > 0)
            dW1 = _tn_matmul(dh, p)
            db1 = dh.sum(0)
            dp = (dh @ W1).contiguous()
        dy2, dx, dr = _like(x), _like(x), _like(x)
        dw = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        check(L.ksh_se_bn_bwd(dz.data_ptr(), z.data_ptr(), s.data_ptr(), dp.data_ptr(), x.data_ptr(), dy2.data_ptr(),
                              dx.data_ptr(), dr.data_ptr(), _code(x), N, H * W, C, _ptr(w), mean.data_ptr(), invstd.data_ptr(),
                              dw.data_ptr(), db.data_ptr(), _workspace(x.device, C).data_ptr(), _stream()), "ksh_se_bn_bwd")
        dpb = None if pbm is None else torch.zeros(pbm[0], dtype=pbm[1], device=x.device)
        return (dx, dr, dw if has_w else None, db if has_b else None, None, None, dpb, None, None,
                dW1.reshape(s1).to(t1), db1.to(tb1), dW2.reshape(s2).to(t2), db2.to(tb2))


def bn_se_tail(x, bn, se, residual, pre_bias=None):
    """relu(se(bn(x [+ pre_bias])) + residual) with the merged passes; None when not covered (then: bn_act + se_tail)."""
    if not (supported(x) and supported(residual) and residual.dtype == x.dtype and residual.shape == x.shape):
        return None
    if bn.momentum is None or not (bn.training or not bn.track_running_stats):
        return None
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BNSETail.apply(x, residual, bn.weight, bn.bias, rm, rv, pre_bias, bn.momentum, bn.eps, se.fc1.weight,
                           se.fc1.bias, se.fc2.weight, se.fc2.bias)


def se_tail(x, se, residual):
    """relu(se(x) + residual) for the reference's SqueezeExcitation module layout (`se.fc1`, `se.fc2` 1x1 convs);
    None when the tensors are not covered by the fused kernels."""
    if supported(x) and supported(residual) and residual.dtype == x.dtype and residual.shape == x.shape:
        return _SETail.apply(x, residual, se.fc1.weight, se.fc1.bias, se.fc2.weight, se.fc2.bias)
    return None
