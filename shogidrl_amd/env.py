"""VecShogiEnv -- N Shogi games stepped at once on one MI355X by the hand-written HIP kernel.

Host-side mirror of the reference interface for this path:
  ShogiGame(max_moves_per_game) / reset / get_legal_moves / make_move / get_observation
      (keisei/shogi/shogi_game.py:39-54, 113-130, 216, 574-660, 183)
  PolicyOutputMapper.get_legal_mask (keisei/utils/utils.py:310-336)
  the Gymnasium-style vector env the reference sketches: reset() -> obs[N];
      step(actions[N]) -> (obs, rewards, done, infos) (+ terminated / truncated properties; docs/1. DESIGN.md:590-602)
All tensors live on the GPU; nothing here synchronises with the host.
"""
import numpy as np
import torch

from . import _lib
from ._lib import MASK_WORDS, NUM_ACTIONS, check, lib

REASON_NAMES = ("Game ongoing", "Tsumi", "stalemate", "Max moves reached", "Sennichite", "illegal_move")


def _stream():
    return torch.cuda.current_stream().cuda_stream


OBS_FORMATS = {"f32": 0, "bf16_nhwc": 1}     # KSH_OBS_F32_NCHW / KSH_OBS_BF16_NHWC (include/keisei_hip.h)


def empty_obs(lead, obs_format, device):
    """Observation storage of shape [*lead, 46, 9, 9] in the memory layout the env kernel writes for `obs_format`:
    float32 NCHW, or bfloat16 with the channel dimension innermost (each [46,9,9] block is 7 452 contiguous bytes;
    a [N,46,9,9] slice is a torch channels_last tensor)."""
    lead = tuple(lead)
    if obs_format == "f32":
        return torch.empty(lead + (46, 9, 9), dtype=torch.float32, device=device)
    k = len(lead)
    return torch.empty(lead + (9, 9, 46), dtype=torch.bfloat16, device=device).permute(*range(k), k + 2, k, k + 1)


def _obs_dense(obs, obs_format):
    """True when `obs` ([N,46,9,9]) is one dense block in the layout of `obs_format`."""
    if obs_format == "f32":
        return obs.dtype == torch.float32 and obs.is_contiguous()
    return obs.dtype == torch.bfloat16 and obs.dim() == 4 and obs.permute(0, 2, 3, 1).is_contiguous()


class VecShogiEnv:
    def __init__(self, num_games, max_moves_per_game=500, device="cuda:0", auto_reset=True, packed_masks=False,
                 obs_format="f32", emit_bits=True, table_mode="auto"):
        """obs_format="bf16_nhwc": observations are written as bfloat16 in channels_last memory format (7 452 B per
        game instead of 14 904) -- tensors still have shape [N,46,9,9] and equal
        `obs_f32.to(torch.bfloat16)` bit for bit; see `empty_obs`.
        Legal masks: by default the kernel writes BOTH the reference's `legal_mask` (bool [N,13527], the drop-in
        contract, utils/utils.py:310-336) and the same set as `legal_bits` (int32 [N,424], 1 696 B per game, +6 % bytes):
        the sampler / loss kernels read the bit-set instead of re-reading the 13.5 KB byte row it was expanded from.
        emit_bits=False: byte mask only (the round-1 contract path, kept for A/B measurements).
        packed_masks=True: ONLY the bit-packed form; `legal_mask` is then unavailable.
        table_mode: "auto" | "lds" | "global" -- where the kernel reads its attack tables from (ksh_env_set_table_mode;
        a tuning switch, identical results)."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.KeiseiHipError("VecShogiEnv runs on an MI355X only (device=%r); there is no CPU path" % (device,))
        self.num_games = int(num_games)
        self.max_moves_per_game = int(max_moves_per_game)
        self.auto_reset = bool(auto_reset)
        self._L = lib()
        h = _lib.C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(idx):
            torch.zeros(1, device=self.device)  # make sure the HIP context exists before the library allocates
            check(self._L.ksh_env_create(idx, self.num_games, self.max_moves_per_game, _lib.C.byref(h)), "ksh_env_create")
        self._h = h
        n, dev = self.num_games, self.device
        if obs_format not in OBS_FORMATS:
            raise ValueError("obs_format must be one of %r" % (tuple(OBS_FORMATS),))
        self.obs_format = obs_format
        check(self._L.ksh_env_set_obs_format(self._h, OBS_FORMATS[obs_format]), "ksh_env_set_obs_format")
        if table_mode != "auto":
            check(self._L.ksh_env_set_table_mode(self._h, {"lds": 1, "global": 2}[table_mode]), "ksh_env_set_table_mode")
        self.obs = empty_obs((n,), obs_format, dev)
        self.packed_masks = bool(packed_masks)
        self.emit_bits = bool(emit_bits) or self.packed_masks
        self.legal_mask = None if self.packed_masks else torch.empty((n, NUM_ACTIONS), dtype=torch.bool, device=dev)
        self.legal_bits = torch.empty((n, MASK_WORDS), dtype=torch.int32, device=dev) if self.emit_bits else None
        self.n_legal = torch.empty(n, dtype=torch.int32, device=dev)
        self.reward = torch.zeros(n, dtype=torch.float32, device=dev)
        self.done = torch.zeros(n, dtype=torch.bool, device=dev)
        self.reason = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.winner = torch.zeros(n, dtype=torch.int8, device=dev)

    def close(self):
        if getattr(self, "_h", None):
            self._L.ksh_env_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def seed(self, seed_value=None):
        """No-op like ShogiGame.seed (shogi_game.py:56-73): the env is deterministic given the actions."""
        self._seed_value = seed_value
        return self

    # ------------------------------------------------------------------ API
    @property
    def kernel_mask(self):
        """the mask form the sampler / loss kernels should read: the bit-set when the env emits it, else the byte mask"""
        return self.legal_bits if self.emit_bits else self.legal_mask

    def _bits_arg(self, t):
        if t.dtype != torch.int32 or not t.is_contiguous() or t.shape != (self.num_games, MASK_WORDS):
            raise ValueError("packed masks must be contiguous int32 [N,%d]" % MASK_WORDS)
        return t

    def reset(self, reset_mask=None, out_obs=None, out_mask=None, out_bits=None):
        """Reset all games (or those where reset_mask is True). Returns (obs, legal mask [bool, or packed bits])."""
        rm = None
        if reset_mask is not None:
            rm = reset_mask.to(device=self.device, dtype=torch.uint8).contiguous()
        if out_obs is not None:
            if not _obs_dense(out_obs, self.obs_format):
                raise ValueError("out_obs must be dense in the env's obs format")
            self.obs = out_obs
        if self.packed_masks:
            if out_mask is not None:
                self.legal_bits = out_mask
            check(self._L.ksh_env_reset_packed(self._h, rm.data_ptr() if rm is not None else None, self.obs.data_ptr(), None,
                                               self.legal_bits.data_ptr(), self.n_legal.data_ptr(), _stream()),
                  "ksh_env_reset_packed")
            return self.obs, self.legal_bits
        if out_mask is not None:
            self.legal_mask = out_mask
        if self.emit_bits:
            if out_bits is not None:
                self.legal_bits = self._bits_arg(out_bits)
            check(self._L.ksh_env_reset_packed(self._h, rm.data_ptr() if rm is not None else None, self.obs.data_ptr(),
                                               self.legal_mask.data_ptr(), self.legal_bits.data_ptr(), self.n_legal.data_ptr(),
                                               _stream()), "ksh_env_reset_packed")
            return self.obs, self.legal_mask
        check(self._L.ksh_env_reset(self._h, rm.data_ptr() if rm is not None else None, self.obs.data_ptr(),
                                    self.legal_mask.data_ptr(), self.n_legal.data_ptr(), _stream()), "ksh_env_reset")
        return self.obs, self.legal_mask

    def observe(self):
        if self.packed_masks:
            raise NotImplementedError("observe() re-encodes byte masks; use a byte-mask env for state inspection")
        check(self._L.ksh_env_observe(self._h, self.obs.data_ptr(), self.legal_mask.data_ptr(), self.n_legal.data_ptr(),
                                      _stream()), "ksh_env_observe")
        if self.emit_bits:
            from . import ops
            self.legal_bits.copy_(ops.pack_mask(self.legal_mask))   # state inspection path (tests / fixtures), not the hot loop
        return self.obs, self.legal_mask

    def evaluate_termination(self):
        """observe() + mark games whose CURRENT position is already terminal, as ShogiGame.from_sfen does
        (shogi_game.py:331-343). Returns (obs, legal_mask); done / reason / winner hold the verdicts."""
        check(self._L.ksh_env_evaluate_termination(self._h, self.obs.data_ptr(), self.legal_mask.data_ptr(),
                                                   self.n_legal.data_ptr(), self.done.data_ptr(), self.reason.data_ptr(),
                                                   self.winner.data_ptr(), _stream()), "ksh_env_evaluate_termination")
        if self.emit_bits:
            from . import ops
            self.legal_bits.copy_(ops.pack_mask(self.legal_mask))
        return self.obs, self.legal_mask

    def step(self, actions, validate=True, out_obs=None, out_mask=None, prev_mask=None, out_bits=None, prev_bits=None):
        """actions: int64 [N] on the device.  Returns (obs, reward, done, info) like ShogiGame.make_move
        (shogi_game.py:553-572), batched; `terminated` / `truncated` (the Gymnasium split the reference sketches in
        docs/1. DESIGN.md:590-602) are available as properties and computed only when asked for.

        out_obs / out_mask: where the kernel writes the next observation / mask (default: the env's own
        buffers); prev_mask: the mask the actions were drawn from (default: the mask of the previous call),
        used to reject illegal actions when validate is True."""
        if actions.dtype != torch.int64 or actions.device != self.device or actions.numel() != self.num_games:
            raise ValueError("actions must be an int64 tensor of shape [%d] on %s" % (self.num_games, self.device))
        actions = actions.contiguous()
        obs = self.obs if out_obs is None else out_obs
        if self.packed_masks:
            bits = self.legal_bits if out_mask is None else out_mask
            if not _obs_dense(obs, self.obs_format) or not bits.is_contiguous() or bits.dtype != torch.int32:
                raise ValueError("out_obs / out_mask must be dense in the env's formats (packed masks: int32 [N,424])")
            prev = (self.legal_bits if prev_mask is None else prev_mask).data_ptr() if validate else None
            check(self._L.ksh_env_step_packed(self._h, actions.data_ptr(), prev, obs.data_ptr(), None, bits.data_ptr(),
                                              self.n_legal.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(),
                                              self.reason.data_ptr(), self.winner.data_ptr(), int(self.auto_reset), _stream()),
                  "ksh_env_step_packed")
            self.obs, self.legal_bits = obs, bits
            return obs, self.reward, self.done, {"reason": self.reason, "winner": self.winner, "n_legal": self.n_legal}
        mask = self.legal_mask if out_mask is None else out_mask
        if not _obs_dense(obs, self.obs_format) or not mask.is_contiguous():
            raise ValueError("out_obs / out_mask must be dense in the env's obs format")
        if self.emit_bits:
            # contract path: byte mask AND bit-set from one launch; the action is validated against the previous bit-set
            # (one dword per game instead of a byte gather -- same verdict, tests/test_gpu_env.py)
            bits = self.legal_bits if out_bits is None else self._bits_arg(out_bits)
            prev = None
            if validate:
                if prev_bits is None and prev_mask is not None:
                    raise ValueError("this env validates against packed masks: pass prev_bits with prev_mask")
                prev = (self.legal_bits if prev_bits is None else self._bits_arg(prev_bits)).data_ptr()
            check(self._L.ksh_env_step_packed(self._h, actions.data_ptr(), prev, obs.data_ptr(), mask.data_ptr(), bits.data_ptr(),
                                              self.n_legal.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(),
                                              self.reason.data_ptr(), self.winner.data_ptr(), int(self.auto_reset), _stream()),
                  "ksh_env_step_packed")
            self.obs, self.legal_mask, self.legal_bits = obs, mask, bits
            return obs, self.reward, self.done, {"reason": self.reason, "winner": self.winner, "n_legal": self.n_legal}
        prev = None
        if validate:
            prev = (self.legal_mask if prev_mask is None else prev_mask).data_ptr()
        check(self._L.ksh_env_step(self._h, actions.data_ptr(), prev, obs.data_ptr(), mask.data_ptr(),
                                   self.n_legal.data_ptr(), self.reward.data_ptr(), self.done.data_ptr(),
                                   self.reason.data_ptr(), self.winner.data_ptr(), int(self.auto_reset), _stream()),
              "ksh_env_step")
        self.obs, self.legal_mask = obs, mask
        return obs, self.reward, self.done, {"reason": self.reason, "winner": self.winner, "n_legal": self.n_legal}

    @property
    def truncated(self):
        """"Max moves reached" (the reference's buffer treats it as done as well, step_manager.py:273-281)."""
        return self.reason == 3

    @property
    def terminated(self):
        return self.done & (self.reason != 3)

    # ------------------------------------------------------------------ host-side state exchange (tests / fixtures)
    def set_state(self, game, rec128):
        rec = np.ascontiguousarray(rec128, np.uint8)
        assert rec.size >= 104
        buf = np.zeros(128, np.uint8)
        buf[:rec.size] = rec[:128]
        check(self._L.ksh_env_set_state(self._h, int(game), buf.ctypes.data, _stream()), "ksh_env_set_state")

    def get_state(self, game):
        buf = np.zeros(128, np.uint8)
        check(self._L.ksh_env_get_state(self._h, int(game), buf.ctypes.data, _stream()), "ksh_env_get_state")
        return buf


class DeviceTimer:
    """hipEvent pair on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self._L = lib()
        self.a, self.b = _lib.C.c_void_p(), _lib.C.c_void_p()
        check(self._L.ksh_event_create(_lib.C.byref(self.a)), "ksh_event_create")
        check(self._L.ksh_event_create(_lib.C.byref(self.b)), "ksh_event_create")

    def start(self):
        check(self._L.ksh_event_record(self.a, _stream()), "ksh_event_record")

    def stop(self):
        check(self._L.ksh_event_record(self.b, _stream()), "ksh_event_record")

    def elapsed_ms(self):
        ms = _lib.C.c_float()
        check(self._L.ksh_event_elapsed_ms(self.a, self.b, _lib.C.byref(ms)), "ksh_event_elapsed_ms")
        return float(ms.value)
