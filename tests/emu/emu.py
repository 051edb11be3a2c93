"""TEST INFRASTRUCTURE: Python driver of the host-emulated env kernel (tests/emu/libksh_emu.so)."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
A = 13527


def _lib():
    # KSH_EMU_ASAN=1 (with libasan preloaded into python, see the Makefile): the AddressSanitizer build of the same sources
    name = "libksh_emu_asan.so" if os.environ.get("KSH_EMU_ASAN") == "1" else "libksh_emu.so"
    subprocess.check_call(["make", "-C", _DIR, "-s", name])
    L = C.CDLL(os.path.join(_DIR, name))
    L.emu_env_launch.argtypes = [C.c_void_p] * 15 + [C.c_int] * 6
    L.emu_build_tables.argtypes = [C.c_void_p, C.c_int]
    L.emu_finish_record.argtypes = [C.c_void_p, C.c_int]
    L.emu_full_hash.argtypes = [C.c_void_p]
    L.emu_full_hash.restype = C.c_uint64
    return L


class EmuEnv:
    def __init__(self, n, max_moves=500, obs_offset=0):
        self.L = _lib()
        self.n, self.max_moves = n, max_moves
        self.cap = max_moves + 1
        self.tables = np.zeros(self.L.emu_tables_size(), np.uint8)
        self.L.emu_build_tables(self.tables.ctypes.data, max_moves)
        self.state = np.zeros((n, 128), np.uint8)
        self.hist = np.zeros((n, self.cap), np.uint64)
        # obs_offset (in floats) shifts the output buffer to exercise every 16-byte alignment phase of the rows
        self._obs_store = np.zeros(n * 46 * 81 + 8, np.float32)
        self.obs = self._obs_store[obs_offset:obs_offset + n * 46 * 81].reshape(n, 46, 9, 9)
        # compact format (bf16, channels last): raw bf16 bit patterns [n, 81, 46]; obs_offset then counts dwords
        self.obs16 = False
        self._obs16_store = np.zeros(n * 81 * 46 + 16, np.uint16)
        self.obs16_bits = self._obs16_store[2 * obs_offset:2 * obs_offset + n * 81 * 46].reshape(n, 81, 46)
        self.mask = np.zeros((n, A), np.uint8)
        self.bits = np.zeros((n, 424), np.uint32)
        self.packed_only = False        # True: the kernel writes ONLY the packed mask (byte mask pointer NULL)
        self.n_legal = np.zeros(n, np.int32)
        self.reward = np.zeros(n, np.float32)
        self.done = np.zeros(n, np.uint8)
        self.reason = np.zeros(n, np.uint8)
        self.winner = np.zeros(n, np.int8)

    def _launch(self, mode, actions=None, prev_mask=None, reset_mask=None, auto_reset=1):
        p = lambda x: None if x is None else x.ctypes.data
        self.obs[:] = -7.0          # poison: every byte must be rewritten
        self._obs16_store[:] = 0x7FC1   # (a NaN pattern the kernel never produces)
        self.mask[:] = 0xCC
        self.bits[:] = 0xDEADBEEF
        prev_bits = None
        if self.packed_only and prev_mask is not None:
            prev_bits = np.ascontiguousarray(np.packbits(np.pad(prev_mask.astype(bool), ((0, 0), (0, 424 * 32 - A))), axis=1,
                                                         bitorder="little").view(np.uint32))
            prev_mask = None
        self.L.emu_env_launch(p(self.state), p(self.hist), p(self.tables), p(actions), p(prev_mask), p(prev_bits), p(reset_mask),
                              p(self.obs16_bits) if self.obs16 else p(self.obs), None if self.packed_only else p(self.mask), p(self.bits), p(self.n_legal), p(self.reward), p(self.done),
                              p(self.reason), p(self.winner), self.n, self.max_moves, self.cap, mode, auto_reset,
                              1 if self.obs16 else 0)
        if self.obs16:
            # nothing outside the rows may be touched, and the float view used by the shared checks is the exact upcast
            lo = self.obs16_bits.__array_interface__["data"][0] - self._obs16_store.__array_interface__["data"][0]
            outside = np.ones(self._obs16_store.size, bool)
            outside[lo // 2:lo // 2 + self.obs16_bits.size] = False
            assert (self._obs16_store[outside] == 0x7FC1).all(), "compact obs writer stored outside its rows"
            self.obs[:] = from_bf16_bits(self.obs16_bits).reshape(self.n, 9, 9, 46).transpose(0, 3, 1, 2)

    def _sync_masks(self):
        """packed and byte forms must describe the same set; in packed-only mode rebuild the byte view from the bits"""
        unpacked = np.unpackbits(self.bits.view(np.uint8), axis=1, bitorder="little")
        assert not unpacked[:, A:].any(), "bits beyond action 13526 must be zero"
        if self.packed_only:
            self.mask[:] = unpacked[:, :A]
        else:
            assert np.array_equal(unpacked[:, :A], self.mask), "packed mask != byte mask"

    def reset(self, reset_mask=None):
        rm = None if reset_mask is None else np.ascontiguousarray(reset_mask, np.uint8)
        self._launch(1, reset_mask=rm)
        self._sync_masks()
        return self.obs.copy(), self.mask.copy(), self.n_legal.copy()

    def observe(self):
        self._launch(0)
        self._sync_masks()
        return self.obs.copy(), self.mask.copy(), self.n_legal.copy()

    def evaluate_termination(self):
        self._launch(3)
        self._sync_masks()
        return (self.obs.copy(), self.mask.copy(), self.n_legal.copy(), self.done.copy(), self.reason.copy(),
                self.winner.copy())

    def step(self, actions, auto_reset=True, check_prev=True):
        a = np.ascontiguousarray(actions, np.int64)
        pm = self.mask.copy() if check_prev else None
        self._launch(2, actions=a, prev_mask=pm, auto_reset=int(auto_reset))
        self._sync_masks()
        return (self.obs.copy(), self.mask.copy(), self.n_legal.copy(), self.reward.copy(), self.done.copy(),
                self.reason.copy(), self.winner.copy())

    def set_state(self, g, rec128):
        rec = np.array(rec128, np.uint8).copy()
        mc = int(rec[96]) | (int(rec[97]) << 8)
        rec[102], rec[103] = mc & 0xFF, mc >> 8      # history starts empty at this ply
        rec[104:] = 0
        self.L.emu_finish_record(rec.ctypes.data, self.max_moves)
        self.state[g] = rec

    def check_hashes(self):
        """the incrementally maintained position key of every game equals the key computed from scratch"""
        for g in range(self.n):
            rec = np.ascontiguousarray(self.state[g])
            stored = int(rec[104:112].view(np.uint64)[0])
            assert stored == self.L.emu_full_hash(rec.ctypes.data), g


# ------------------------------------------------------------------ learner-side kernels
def _plib():
    L = _lib()
    L.emu_gae.argtypes = [C.c_void_p] * 4 + [C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.emu_masked_sample.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int,
                                    C.c_void_p, C.c_void_p]
    L.emu_ppo_loss.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_float, C.c_float, C.c_float,
                                                                          C.c_int] + [C.c_void_p] * 4
    vp, i32, f32, u64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64
    L.emu_pack_mask.argtypes = [vp, i32, vp]
    L.emu_head_sample.argtypes = [vp, vp, vp, i32, vp, i32, u64, u64, i32, vp, vp]
    L.emu_head_loss.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, i32, i32, i32,
                                vp, vp, vp, vp, vp, vp, vp, i32, i32]
    L.emu_evaluate_actions.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return L


def to_bf16_bits(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def from_bf16_bits(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def pack_mask(mask):
    """bool/uint8 [B,13527] -> uint32 [B,424], bit i of a row <-> action i"""
    m = np.pad(np.asarray(mask).astype(bool), ((0, 0), (0, 424 * 32 - A)))
    return np.ascontiguousarray(np.packbits(m, axis=1, bitorder="little").view(np.uint32))


class EmuOps:
    """numpy front-end with the same call shapes as tests use for the HIP library."""
    packed = False      # True: hand the kernels bit-packed masks instead of byte masks

    def gae(self, r, v, d, last_v, gamma, lam):
        r = np.ascontiguousarray(r, np.float32); v = np.ascontiguousarray(v, np.float32)
        d = np.ascontiguousarray(d, np.uint8); lv = np.ascontiguousarray(last_v, np.float32)
        T, N = r.shape
        adv = np.zeros((T, N), np.float32); ret = np.zeros((T, N), np.float32)
        _plib().emu_gae(r.ctypes.data, v.ctypes.data, d.ctypes.data, lv.ctypes.data, gamma, lam, T, N,
                        adv.ctypes.data, ret.ctypes.data)
        return adv, ret

    def masked_sample(self, logits, mask, seed, step, deterministic, bf16=False, offset=0):
        B = logits.shape[0]
        lg = to_bf16_bits(logits) if bf16 else np.ascontiguousarray(logits, np.float32)
        # `offset` shifts the buffers by that many elements to exercise every alignment phase
        buf = np.zeros(lg.size + 16, lg.dtype); buf[offset:offset + lg.size] = lg.ravel()
        mb = np.zeros(mask.size + 16, np.uint8); mb[offset:offset + mask.size] = mask.astype(np.uint8).ravel()
        act = np.zeros(B, np.int64); lp = np.zeros(B, np.float32)
        bits = pack_mask(mask) if self.packed else None
        _plib().emu_masked_sample(buf.ctypes.data + offset * buf.itemsize, int(bf16),
                                  None if self.packed else mb.ctypes.data + offset, None if bits is None else bits.ctypes.data, B,
                                  seed, step, int(deterministic), act.ctypes.data, lp.ctypes.data)
        return act, lp

    def ppo_loss(self, logits, mask, actions, old_logp, adv, ret, values, old_values, clip_eps, vf, ent, value_clip,
                 bf16=False):
        B = logits.shape[0]
        lg = to_bf16_bits(logits) if bf16 else np.ascontiguousarray(logits, np.float32)
        m8 = np.ascontiguousarray(mask, np.uint8)
        dl = np.full(lg.shape, 0x7FC1 if bf16 else np.nan, lg.dtype)
        f = lambda x: np.ascontiguousarray(x, np.float32)
        a64 = np.ascontiguousarray(actions, np.int64)
        ol, ad, rt, vl, ov = f(old_logp), f(adv), f(ret), f(values), f(old_values)
        dv = np.zeros(B, np.float32); st = np.zeros(8, np.float32); ws = np.zeros(B * 8 + 16, np.float32)
        bits = pack_mask(mask) if self.packed else None
        _plib().emu_ppo_loss(lg.ctypes.data, int(bf16), None if self.packed else m8.ctypes.data,
                             None if bits is None else bits.ctypes.data, a64.ctypes.data, ol.ctypes.data, ad.ctypes.data,
                             rt.ctypes.data, vl.ctypes.data, ov.ctypes.data, B, clip_eps, vf, ent, int(value_clip),
                             dl.ctypes.data, dv.ctypes.data, st.ctypes.data, ws.ctypes.data)
        return (from_bf16_bits(dl) if bf16 else dl), dv, st


    def evaluate_actions(self, logits, mask, actions):
        B = logits.shape[0]
        lg = np.ascontiguousarray(logits, np.float32)
        m8 = np.ascontiguousarray(mask, np.uint8)
        a64 = np.ascontiguousarray(actions, np.int64)
        bits = pack_mask(mask) if self.packed else None
        lp = np.full(B, np.nan, np.float32); en = np.full(B, np.nan, np.float32)
        _plib().emu_evaluate_actions(lg.ctypes.data, 0, None if self.packed else m8.ctypes.data,
                                     None if bits is None else bits.ctypes.data, a64.ctypes.data, B, lp.ctypes.data, en.ctypes.data)
        return lp, en


    # ---- fused policy head (ksh_head_kernels.h); masks are always handed over bit-packed
    wpa = 0          # max segments per bit row in the weight-gradient pipeline (0 = the library's choice)
    light_max = -1   # actions with at most this many samples are one work item (-1 = the library's 96)

    def pack_mask_kernel(self, mask):
        m8 = np.ascontiguousarray(mask, np.uint8)
        bits = np.full((m8.shape[0], 424), 0xDEADBEEF, np.uint32)
        _plib().emu_pack_mask(m8.ctypes.data, m8.shape[0], bits.ctypes.data)
        return bits

    def head_sample(self, feat, W, bias, mask, seed, step, deterministic):
        f = lambda x: None if x is None else np.ascontiguousarray(x, np.float32)
        feat, W, bias = f(feat), f(W), f(bias)
        B, K = feat.shape
        bits = pack_mask(mask)
        act = np.zeros(B, np.int64); lp = np.zeros(B, np.float32)
        _plib().emu_head_sample(feat.ctypes.data, W.ctypes.data, None if bias is None else bias.ctypes.data, K, bits.ctypes.data,
                                B, seed, step, int(deterministic), act.ctypes.data, lp.ctypes.data)
        return act, lp

    def _head(self, feat, W, bias, mask, actions, old_logp, adv, ret, values, old_values, clip_eps, vf, ent, value_clip,
              dense, want_grad):
        f = lambda x: None if x is None else np.ascontiguousarray(x, np.float32)
        feat, W, bias = f(feat), f(W), f(bias)
        B, K = feat.shape
        bits = pack_mask(mask)
        a64 = np.ascontiguousarray(actions, np.int64)
        ol, ad, rt, vl, ov = f(old_logp), f(adv), f(ret), f(values), f(old_values)
        dfeat = np.full((B, K), np.nan, np.float32); dW = np.full(W.shape, np.nan, np.float32)
        dbias = np.full(W.shape[0], np.nan, np.float32)
        dv = np.zeros(B, np.float32); st = np.zeros(8, np.float32)
        lp = np.full(B, np.nan, np.float32); en = np.full(B, np.nan, np.float32)
        p = lambda x: None if x is None else x.ctypes.data
        _plib().emu_head_loss(p(feat), p(W), p(bias), K, p(bits), p(a64), p(ol), p(ad), p(rt), p(vl), p(ov), B, clip_eps, vf, ent,
                              int(value_clip), int(dense), int(want_grad), p(dfeat), p(dW), p(dbias), p(dv), p(st), p(lp), p(en),
                              int(self.wpa), int(self.light_max))
        return dfeat, dW, dbias, dv, st, lp, en

    def head_loss(self, feat, W, bias, mask, actions, old_logp, adv, ret, values, old_values, clip_eps, vf, ent, value_clip,
                  dense=True):
        return self._head(feat, W, bias, mask, actions, old_logp, adv, ret, values, old_values, clip_eps, vf, ent, value_clip,
                          dense, 1)[:5]

    def head_evaluate(self, feat, W, bias, mask, actions, dense=True):
        z = np.zeros(feat.shape[0], np.float32)
        return self._head(feat, W, bias, mask, actions, z, z, z, z, None, 0.2, 0.0, 0.0, 0, dense, 0)[5:]


# ------------------------------------------------------------------ tower-glue kernels
class EmuTower:
    """numpy backend of tests/towercheck.py over the host-emulated kernels; activations are dense [M, C] float32
    arrays that are converted to bf16 bit patterns for the bf16 instantiations."""

    def __init__(self):
        self.L = _lib()
        vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
        self.L.emu_tower_workspace_floats.restype = i64
        self.L.emu_tower_workspace_floats.argtypes = [i32]
        self.L.emu_bn_fwd_train.argtypes = [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp]
        self.L.emu_bn_fwd_eval.argtypes = [vp, vp, vp, i32, i64, i32, vp, i32]
        self.L.emu_bn_bwd.argtypes = [vp, vp, vp, vp, vp, i32, i64, i32, vp, vp, vp, i32, vp, vp, vp]
        self.L.emu_se_pool.argtypes = [vp, i32, i32, i32, i32, vp]
        self.L.emu_se_apply.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32]
        self.L.emu_se_bwd_reduce.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
        self.L.emu_se_bwd_apply.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32]
        self.L.emu_bn_pool_fwd_train.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp]
        self.L.emu_se_bn_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]

    @staticmethod
    def _in(a, bf16):
        if a is None:
            return None
        a = np.ascontiguousarray(a, np.float32)
        return to_bf16_bits(a).reshape(a.shape) if bf16 else a

    @staticmethod
    def _out(shape, bf16):
        return np.full(shape, 0x7FC1, np.uint16) if bf16 else np.full(shape, np.nan, np.float32)

    @staticmethod
    def _f(a, bf16):
        return from_bf16_bits(a).reshape(a.shape) if bf16 else a

    def _ws(self, Cn):
        return np.zeros(self.L.emu_tower_workspace_floats(Cn), np.float32)

    def bn_fwd_train(self, x, res, bf16, w, b, rm, rv, momentum, eps, relu, pre_bias=None):
        M, Cn = x.shape
        p = lambda t: None if t is None else t.ctypes.data
        xi, ri, y = self._in(x, bf16), self._in(res, bf16), self._out(x.shape, bf16)
        mean, invstd = np.zeros(Cn, np.float32), np.zeros(Cn, np.float32)
        ws = self._ws(Cn)
        self.L.emu_bn_fwd_train(p(xi), p(ri), p(y), int(bf16), M, Cn, p(pre_bias), p(w), p(b), p(rm), p(rv), momentum, eps, int(relu),
                                p(mean), p(invstd), p(ws))
        return self._f(y, bf16), rm, rv, mean, invstd

    def bn_fwd_eval(self, x, res, bf16, coef, relu):
        M, Cn = x.shape
        p = lambda t: None if t is None else t.ctypes.data
        xi, ri, y = self._in(x, bf16), self._in(res, bf16), self._out(x.shape, bf16)
        coef = np.ascontiguousarray(coef, np.float32)
        self.L.emu_bn_fwd_eval(p(xi), p(ri), p(y), int(bf16), M, Cn, p(coef), int(relu))
        return self._f(y, bf16)

    def bn_bwd(self, dy, x, y, bf16, w, mean, invstd, relu, residual):
        M, Cn = x.shape
        p = lambda t: None if t is None else t.ctypes.data
        gi, xi, yi = self._in(dy, bf16), self._in(x, bf16), self._in(y, bf16)
        dx = self._out(x.shape, bf16)
        dres = self._out(x.shape, bf16) if (residual and relu) else None
        dw, db = np.zeros(Cn, np.float32), np.zeros(Cn, np.float32)
        ws = self._ws(Cn)
        self.L.emu_bn_bwd(p(gi), p(xi), p(yi), p(dx), p(dres), int(bf16), M, Cn, p(w), p(mean), p(invstd), int(relu), p(dw),
                          p(db), p(ws))
        dres_f = None if not residual else (self._f(dres, bf16) if relu else np.asarray(dy, np.float32))
        return self._f(dx, bf16), dres_f, dw, db

    def se_pool(self, x, bf16, N, HW, Cn):
        xi, out = self._in(x, bf16), np.zeros((N, Cn), np.float32)
        self.L.emu_se_pool(xi.ctypes.data, int(bf16), N, HW, Cn, out.ctypes.data)
        return out

    def se_apply(self, x, s, r, bf16, N, HW, Cn):
        xi, ri, z = self._in(x, bf16), self._in(r, bf16), self._out(x.shape, bf16)
        s = np.ascontiguousarray(s, np.float32)
        self.L.emu_se_apply(xi.ctypes.data, s.ctypes.data, ri.ctypes.data, z.ctypes.data, int(bf16), N, HW, Cn)
        return self._f(z, bf16)

    def se_bwd_reduce(self, dz, z, x, bf16, N, HW, Cn):
        gi, zi, xi, ds = self._in(dz, bf16), self._in(z, bf16), self._in(x, bf16), np.zeros((N, Cn), np.float32)
        self.L.emu_se_bwd_reduce(gi.ctypes.data, zi.ctypes.data, xi.ctypes.data, int(bf16), N, HW, Cn, ds.ctypes.data)
        return ds

    def se_bwd_apply(self, dz, z, s, dp, bf16, N, HW, Cn):
        gi, zi = self._in(dz, bf16), self._in(z, bf16)
        dx, dr = self._out(dz.shape, bf16), self._out(dz.shape, bf16)
        s, dp = np.ascontiguousarray(s, np.float32), np.ascontiguousarray(dp, np.float32)
        self.L.emu_se_bwd_apply(gi.ctypes.data, zi.ctypes.data, s.ctypes.data, dp.ctypes.data, dx.ctypes.data, dr.ctypes.data,
                                int(bf16), N, HW, Cn)
        return self._f(dx, bf16), self._f(dr, bf16)

    def bn_pool_fwd_train(self, x, bf16, N, HW, w, b, rm, rv, momentum, eps, pre_bias=None):
        M, Cn = x.shape
        p = lambda t: None if t is None else t.ctypes.data
        xi, y = self._in(x, bf16), self._out(x.shape, bf16)
        mean, invstd, pooled = np.zeros(Cn, np.float32), np.zeros(Cn, np.float32), np.zeros((N, Cn), np.float32)
        ws = self._ws(Cn)
        self.L.emu_bn_pool_fwd_train(p(xi), p(y), int(bf16), N, HW, Cn, p(pre_bias), p(w), p(b), p(rm), p(rv), momentum, eps,
                                     p(mean), p(invstd), p(pooled), p(ws))
        return self._f(y, bf16), pooled, rm, rv, mean, invstd

    def se_bn_bwd(self, dz, z, s, dp, x, bf16, N, HW, w, mean, invstd):
        M, Cn = x.shape
        p = lambda t: None if t is None else t.ctypes.data
        gi, zi, xi = self._in(dz, bf16), self._in(z, bf16), self._in(x, bf16)
        dy2, dx, dr = self._out(x.shape, bf16), self._out(x.shape, bf16), self._out(x.shape, bf16)
        s, dp = np.ascontiguousarray(s, np.float32), np.ascontiguousarray(dp, np.float32)
        dw, db, ws = np.zeros(Cn, np.float32), np.zeros(Cn, np.float32), self._ws(Cn)
        self.L.emu_se_bn_bwd(p(gi), p(zi), p(s), p(dp), p(xi), p(dy2), p(dx), p(dr), int(bf16), N, HW, Cn, p(w), p(mean), p(invstd),
                             p(dw), p(db), p(ws))
        return self._f(dx, bf16), self._f(dr, bf16), dw, db
