"""Adapters that expose the PRODUCT (shogidrl_amd over the C-ABI of libkeisei_hip.so, on cuda:0) through
the numpy backend protocols of tests/envcheck.py and tests/ppocheck.py."""
import numpy as np
import torch

from shogidrl_amd import ops
from shogidrl_amd.env import VecShogiEnv

DEV = "cuda:0"


class GpuEnv:
    def __init__(self, n, max_moves, obs_format="f32"):
        self.n = n
        self.env = VecShogiEnv(n, max_moves, device=DEV, obs_format=obs_format)
        if obs_format != "f32":
            # compact observations are compared as their exact float32 upcast against the oracle's planes rounded by
            # torch's own float32 -> bfloat16 conversion (tests/envcheck.py check_random_vs_oracle)
            self.obs_round = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).float().numpy()

    def _out(self):
        e = self.env
        torch.cuda.synchronize()
        if e.emit_bits:      # contract path: the bit-set the sampler / loss kernels read is the byte mask, packed
            assert torch.equal(ops.unpack_mask(e.legal_bits), e.legal_mask), "packed legal bits != byte mask"
        return np.ascontiguousarray(e.obs.float().cpu().numpy()), e.legal_mask.cpu().numpy().astype(np.uint8), e.n_legal.cpu().numpy().copy()

    def reset(self, reset_mask=None):
        rm = None if reset_mask is None else torch.as_tensor(np.asarray(reset_mask, np.uint8), device=DEV)
        self.env.obs.fill_(-7.0)
        self.env.legal_mask.view(torch.uint8).fill_(0xCC)
        self.env.reset(rm)
        return self._out()

    def observe(self):
        self.env.obs.fill_(-7.0)
        self.env.legal_mask.view(torch.uint8).fill_(0xCC)
        self.env.observe()
        return self._out()

    def evaluate_termination(self):
        e = self.env
        e.evaluate_termination()
        obs, mask, nl = self._out()
        return (obs, mask, nl, e.done.cpu().numpy().astype(np.uint8), e.reason.cpu().numpy().copy(),
                e.winner.cpu().numpy().copy())

    def step(self, actions, auto_reset=True, check_prev=True):
        e = self.env
        e.auto_reset = bool(auto_reset)
        a = torch.as_tensor(np.asarray(actions, np.int64), device=DEV)
        if check_prev:
            # validate against the mask the previous call produced, writing the new one elsewhere
            new_mask = torch.full_like(e.legal_mask.view(torch.uint8), 0xCC).view(torch.bool)
            new_obs = torch.full_like(e.obs, -7.0)
            e.step(a, validate=True, out_obs=new_obs, out_mask=new_mask)
        else:
            e.step(a, validate=False)
        torch.cuda.synchronize()
        obs, mask, nl = self._out()
        return (obs, mask, nl, e.reward.cpu().numpy().copy(), e.done.cpu().numpy().astype(np.uint8),
                e.reason.cpu().numpy().copy(), e.winner.cpu().numpy().copy())

    def set_state(self, g, rec128):
        self.env.set_state(g, rec128)


def make_env(n, max_moves):
    return GpuEnv(n, max_moves)


class GpuOps:
    def gae(self, r, v, d, last_v, gamma, lam):
        t = lambda x, dt: torch.as_tensor(np.ascontiguousarray(x, dt), device=DEV)
        adv, ret = ops.gae(t(r, np.float32), t(v, np.float32), t(d, np.uint8), t(last_v, np.float32), gamma, lam)
        return adv.cpu().numpy(), ret.cpu().numpy()

    def masked_sample(self, logits, mask, seed, step, deterministic, bf16=False, offset=0):
        B, A = logits.shape
        dt = torch.bfloat16 if bf16 else torch.float32
        flat = torch.zeros(B * A + 16, dtype=dt, device=DEV)
        flat[offset:offset + B * A] = torch.as_tensor(logits, device=DEV).to(dt).reshape(-1)
        mflat = torch.zeros(B * A + 16, dtype=torch.uint8, device=DEV)
        mflat[offset:offset + B * A] = torch.as_tensor(np.asarray(mask, np.uint8), device=DEV).reshape(-1)
        lg = flat[offset:offset + B * A].view(B, A)       # views keep the (mis)alignment of the slice start
        mk = mflat[offset:offset + B * A].view(B, A)
        act, lp = ops.masked_sample(lg, mk, seed, step, deterministic)
        return act.cpu().numpy(), lp.cpu().numpy()

    def ppo_loss(self, logits, mask, actions, old_logp, adv, ret, values, old_values, clip_eps, vf, ent, value_clip,
                 bf16=False):
        dt = torch.bfloat16 if bf16 else torch.float32
        lg = torch.as_tensor(logits, device=DEV).to(dt).requires_grad_(True)
        vl = torch.as_tensor(np.asarray(values, np.float32), device=DEV).requires_grad_(True)
        t = lambda x: torch.as_tensor(np.asarray(x, np.float32), device=DEV)
        loss, stats = ops.ppo_loss(lg, vl, torch.as_tensor(np.asarray(mask, bool), device=DEV),
                                   torch.as_tensor(np.asarray(actions, np.int64), device=DEV), t(old_logp), t(adv), t(ret),
                                   t(old_values), clip_eps, vf, ent, bool(value_clip))
        loss.backward()
        return lg.grad.float().cpu().numpy(), vl.grad.cpu().numpy(), stats.cpu().numpy()


    def evaluate_actions(self, logits, mask, actions):
        lp, en = ops.evaluate_actions(torch.as_tensor(np.asarray(logits, np.float32), device=DEV),
                                      torch.as_tensor(np.asarray(mask, bool), device=DEV),
                                      torch.as_tensor(np.asarray(actions, np.int64), device=DEV))
        return lp.cpu().numpy(), en.cpu().numpy()


    # ---- fused policy head (masks go in as bool rows; the op packs them with the HIP kernel)
    def pack_mask_kernel(self, mask):
        return ops.pack_mask_bits(torch.as_tensor(np.asarray(mask, np.uint8), device=DEV)).cpu().numpy().view(np.uint32)

    @staticmethod
    def _hf(x):
        return None if x is None else torch.as_tensor(np.ascontiguousarray(x, np.float32), device=DEV)

    def head_sample(self, feat, W, bias, mask, seed, step, deterministic):
        act, lp = ops.policy_head_sample(self._hf(feat), self._hf(W), self._hf(bias),
                                         torch.as_tensor(np.asarray(mask, bool), device=DEV), seed, step, deterministic)
        return act.cpu().numpy(), lp.cpu().numpy()

    def head_evaluate(self, feat, W, bias, mask, actions, dense=True):
        lp, en = ops.policy_head_evaluate(self._hf(feat), self._hf(W), self._hf(bias),
                                          torch.as_tensor(np.asarray(mask, bool), device=DEV),
                                          torch.as_tensor(np.asarray(actions, np.int64), device=DEV), dense)
        return lp.cpu().numpy(), en.cpu().numpy()

    def head_loss(self, feat, W, bias, mask, actions, old_logp, adv, ret, values, old_values, clip_eps, vf, ent, value_clip,
                  dense=True):
        ft = self._hf(feat).requires_grad_(True)
        Wt = self._hf(W).requires_grad_(True)
        bt = None if bias is None else self._hf(bias).requires_grad_(True)
        vl = self._hf(values).requires_grad_(True)
        loss, stats = ops.policy_head_ppo_loss(ft, Wt, bt, vl, torch.as_tensor(np.asarray(mask, bool), device=DEV),
                                               torch.as_tensor(np.asarray(actions, np.int64), device=DEV), self._hf(old_logp),
                                               self._hf(adv), self._hf(ret), self._hf(old_values), clip_eps, vf, ent,
                                               bool(value_clip), dense_rows_possible=dense)
        loss.backward()
        db = np.zeros(W.shape[0], np.float32) if bt is None else bt.grad.cpu().numpy()
        return ft.grad.cpu().numpy(), Wt.grad.cpu().numpy(), db, vl.grad.cpu().numpy(), stats.cpu().numpy()


class GpuTower:
    """numpy backend of tests/towercheck.py over the C-ABI tower entry points (dense [M, C] activations)."""

    def __init__(self):
        from shogidrl_amd._lib import check, lib
        self.L, self.check = lib(), check
        self.st = lambda: torch.cuda.current_stream().cuda_stream

    @staticmethod
    def _t(a, bf16=False):
        if a is None:
            return None
        t = torch.as_tensor(np.ascontiguousarray(a, np.float32), device=DEV)
        return t.to(torch.bfloat16) if bf16 else t

    @staticmethod
    def _np(t):
        return t.float().cpu().numpy()

    def _ws(self, C):
        return torch.empty(self.L.ksh_tower_workspace_bytes(C) // 4, dtype=torch.float32, device=DEV)

    def bn_fwd_train(self, x, res, bf16, w, b, rm, rv, momentum, eps, relu, pre_bias=None):
        M, C = x.shape
        p = lambda t: None if t is None else t.data_ptr()
        xt, rt = self._t(x, bf16), self._t(res, bf16)
        y = torch.full_like(xt, float("nan"))
        wt, bt, rmt, rvt, pbt = self._t(w), self._t(b), self._t(rm), self._t(rv), self._t(pre_bias)
        mean, invstd, ws = torch.empty(C, device=DEV), torch.empty(C, device=DEV), self._ws(C)
        self.check(self.L.ksh_bn_fwd_train(p(xt), p(rt), p(y), int(bf16), M, C, p(pbt), p(wt), p(bt), p(rmt), p(rvt), momentum, eps,
                                           int(relu), p(mean), p(invstd), p(ws), self.st()), "ksh_bn_fwd_train")
        return self._np(y), self._np(rmt), self._np(rvt), self._np(mean), self._np(invstd)

    def bn_fwd_eval(self, x, res, bf16, coef, relu):
        M, C = x.shape
        p = lambda t: None if t is None else t.data_ptr()
        xt, rt, ct = self._t(x, bf16), self._t(res, bf16), self._t(coef)
        y = torch.full_like(xt, float("nan"))
        self.check(self.L.ksh_bn_fwd_eval(p(xt), p(rt), p(y), int(bf16), M, C, p(ct), int(relu), self.st()), "ksh_bn_fwd_eval")
        return self._np(y)

    def bn_bwd(self, dy, x, y, bf16, w, mean, invstd, relu, residual):
        M, C = x.shape
        p = lambda t: None if t is None else t.data_ptr()
        gt, xt, yt = self._t(dy, bf16), self._t(x, bf16), self._t(y, bf16)
        dx = torch.full_like(xt, float("nan"))
        dres = torch.full_like(xt, float("nan")) if (residual and relu) else None
        dw, db, ws = torch.empty(C, device=DEV), torch.empty(C, device=DEV), self._ws(C)
        wt, mt, it = self._t(w), self._t(mean), self._t(invstd)      # keep the tensors alive across the launch
        self.check(self.L.ksh_bn_bwd(p(gt), p(xt), p(yt), p(dx), p(dres), int(bf16), M, C, p(wt), p(mt), p(it), int(relu),
                                     p(dw), p(db), p(ws), self.st()), "ksh_bn_bwd")
        dres_f = None if not residual else (self._np(dres) if relu else np.asarray(dy, np.float32))
        return self._np(dx), dres_f, self._np(dw), self._np(db)

    def se_pool(self, x, bf16, N, HW, C):
        xt, out = self._t(x, bf16), torch.empty((N, C), device=DEV)
        self.check(self.L.ksh_se_pool(xt.data_ptr(), int(bf16), N, HW, C, out.data_ptr(), self.st()), "ksh_se_pool")
        return self._np(out)

    def se_apply(self, x, s, r, bf16, N, HW, C):
        xt, rt, st = self._t(x, bf16), self._t(r, bf16), self._t(s)
        z = torch.full_like(xt, float("nan"))
        self.check(self.L.ksh_se_apply(xt.data_ptr(), st.data_ptr(), rt.data_ptr(), z.data_ptr(), int(bf16), N, HW, C, self.st()),
                   "ksh_se_apply")
        return self._np(z)

    def se_bwd_reduce(self, dz, z, x, bf16, N, HW, C):
        gt, zt, xt, ds = self._t(dz, bf16), self._t(z, bf16), self._t(x, bf16), torch.empty((N, C), device=DEV)
        self.check(self.L.ksh_se_bwd_reduce(gt.data_ptr(), zt.data_ptr(), xt.data_ptr(), int(bf16), N, HW, C, ds.data_ptr(),
                                            self.st()), "ksh_se_bwd_reduce")
        return self._np(ds)

    def se_bwd_apply(self, dz, z, s, dp, bf16, N, HW, C):
        gt, zt, st, dpt = self._t(dz, bf16), self._t(z, bf16), self._t(s), self._t(dp)
        dx, dr = torch.full_like(gt, float("nan")), torch.full_like(gt, float("nan"))
        self.check(self.L.ksh_se_bwd_apply(gt.data_ptr(), zt.data_ptr(), st.data_ptr(), dpt.data_ptr(), dx.data_ptr(),
                                           dr.data_ptr(), int(bf16), N, HW, C, self.st()), "ksh_se_bwd_apply")
        return self._np(dx), self._np(dr)

    def bn_pool_fwd_train(self, x, bf16, N, HW, w, b, rm, rv, momentum, eps, pre_bias=None):
        M, C = x.shape
        p = lambda t: None if t is None else t.data_ptr()
        xt = self._t(x, bf16)
        y = torch.full_like(xt, float("nan"))
        wt, bt, rmt, rvt, pbt = self._t(w), self._t(b), self._t(rm), self._t(rv), self._t(pre_bias)
        mean, invstd, pooled, ws = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty((N, C), device=DEV), self._ws(C)
        self.check(self.L.ksh_bn_pool_fwd_train(p(xt), p(y), int(bf16), N, HW, C, p(pbt), p(wt), p(bt), p(rmt), p(rvt), momentum, eps,
                                                p(mean), p(invstd), p(pooled), p(ws), self.st()), "ksh_bn_pool_fwd_train")
        return self._np(y), self._np(pooled), self._np(rmt), self._np(rvt), self._np(mean), self._np(invstd)

    def se_bn_bwd(self, dz, z, s, dp, x, bf16, N, HW, w, mean, invstd):
        M, C = x.shape
        p = lambda t: None if t is None else t.data_ptr()
        gt, zt, xt, st, dpt = self._t(dz, bf16), self._t(z, bf16), self._t(x, bf16), self._t(s), self._t(dp)
        wt, mt, it = self._t(w), self._t(mean), self._t(invstd)
        dy2, dx, dr = (torch.full_like(xt, float("nan")) for _ in range(3))
        dw, db, ws = torch.empty(C, device=DEV), torch.empty(C, device=DEV), self._ws(C)
        self.check(self.L.ksh_se_bn_bwd(p(gt), p(zt), p(st), p(dpt), p(xt), p(dy2), p(dx), p(dr), int(bf16), N, HW, C, p(wt), p(mt),
                                        p(it), p(dw), p(db), p(ws), self.st()), "ksh_se_bn_bwd")
        return self._np(dx), self._np(dr), self._np(dw), self._np(db)
