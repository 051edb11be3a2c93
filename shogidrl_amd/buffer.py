"""Experience storage for the PPO path.

ExperienceBuffer      -- the reference's single-stream buffer, same constructor / methods / error behaviour
                         (keisei/core/experience_buffer.py:26-209), GAE by the HIP scan kernel (N = 1).
VecExperienceBuffer   -- the batched [T, N] counterpart the GPU rollout fills in place: the env kernel writes
                         the next observation / mask straight into slot t+1, nothing is copied per step.
"""
import warnings

import torch

from . import ops
from ._lib import MASK_WORDS, NUM_ACTIONS
from .env import empty_obs


class ExperienceBuffer:
    def __init__(self, buffer_size, gamma, lambda_gae, device="cuda:0"):
        self.buffer_size, self.gamma, self.lambda_gae = buffer_size, gamma, lambda_gae
        self.device = torch.device(device)
        d = self.device
        self.obs = torch.zeros((buffer_size, 46, 9, 9), dtype=torch.float32, device=d)
        self.actions = torch.zeros(buffer_size, dtype=torch.int64, device=d)
        self.rewards = torch.zeros(buffer_size, dtype=torch.float32, device=d)
        self.log_probs = torch.zeros(buffer_size, dtype=torch.float32, device=d)
        self.values = torch.zeros(buffer_size, dtype=torch.float32, device=d)
        self.dones = torch.zeros(buffer_size, dtype=torch.bool, device=d)
        self.legal_masks = torch.zeros((buffer_size, NUM_ACTIONS), dtype=torch.bool, device=d)
        self.advantages = torch.zeros(buffer_size, dtype=torch.float32, device=d)
        self.returns = torch.zeros(buffer_size, dtype=torch.float32, device=d)
        self.ptr = 0
        self._advantages_computed = False

    def add(self, obs, action, reward, log_prob, value, done, legal_mask):
        if self.ptr >= self.buffer_size:
            warnings.warn("ExperienceBuffer: Buffer is full. Cannot add new experience.")
            return
        i = self.ptr
        self.obs[i] = torch.as_tensor(obs).to(self.device)
        self.actions[i] = action
        self.rewards[i] = reward
        self.log_probs[i] = log_prob
        self.values[i] = value
        self.dones[i] = done
        self.legal_masks[i] = torch.as_tensor(legal_mask).to(self.device)
        self.ptr += 1

    def compute_advantages_and_returns(self, last_value):
        if self.ptr == 0:
            warnings.warn("ExperienceBuffer: compute_advantages_and_returns called on an empty buffer.")
            return
        n = self.ptr
        lv = torch.tensor([float(last_value)], dtype=torch.float32, device=self.device)
        adv, ret = ops.gae(self.rewards[:n, None], self.values[:n, None], self.dones[:n, None], lv, self.gamma,
                           self.lambda_gae)
        self.advantages[:n] = adv[:, 0]
        self.returns[:n] = ret[:, 0]
        self._advantages_computed = True

    def get_batch(self):
        if self.ptr == 0:
            warnings.warn("ExperienceBuffer: get_batch called on an empty or not-yet-computed buffer.")
            return {}
        if not self._advantages_computed:
            raise RuntimeError("Cannot get batch: compute_advantages_and_returns() must be called first")
        n = self.ptr
        return {"obs": self.obs[:n], "actions": self.actions[:n], "log_probs": self.log_probs[:n],
                "values": self.values[:n], "rewards": self.rewards[:n], "advantages": self.advantages[:n],
                "returns": self.returns[:n], "dones": self.dones[:n], "legal_masks": self.legal_masks[:n]}

    def clear(self):
        self.ptr = 0
        self._advantages_computed = False

    def __len__(self):
        return self.ptr

    def size(self):
        return self.ptr

    def capacity(self):
        return self.buffer_size


class VecExperienceBuffer:
    """[T, N] rollout storage; obs / legal_masks have T+1 slots (slot T holds the bootstrap observation)."""

    def __init__(self, horizon, num_games, gamma, lambda_gae, device="cuda:0", packed_masks=False, obs_format="f32",
                 emit_bits=True):
        """Default: `legal_masks` bool [T+1,N,13527] (the reference's buffer contract, experience_buffer.py:29-97) plus the
        same sets as `legal_bits` int32 [T+1,N,424] -- the form the sampler / loss kernels read (1.7 KB instead of
        13.5 KB per row).  emit_bits=False: byte masks only.  packed_masks=True stores ONLY the bit-packed form (then
        `legal_masks` IS the int32 tensor: 8x less memory and traffic); obs_format="bf16_nhwc" stores observations the
        way an env of that format writes them (half the bytes)."""
        self.T, self.N, self.gamma, self.lambda_gae = horizon, num_games, gamma, lambda_gae
        self.packed_masks = bool(packed_masks)
        self.emit_bits = bool(emit_bits) or self.packed_masks
        self.device = torch.device(device)
        d, T, N = self.device, horizon, num_games
        self.obs_format = obs_format
        self.obs = empty_obs((T + 1, N), obs_format, d)
        if self.packed_masks:
            self.legal_masks = torch.empty((T + 1, N, MASK_WORDS), dtype=torch.int32, device=d)
            self.legal_bits = self.legal_masks
        else:
            self.legal_masks = torch.empty((T + 1, N, NUM_ACTIONS), dtype=torch.bool, device=d)
            self.legal_bits = torch.empty((T + 1, N, MASK_WORDS), dtype=torch.int32, device=d) if self.emit_bits else None
        self.actions = torch.zeros((T, N), dtype=torch.int64, device=d)
        self.rewards = torch.zeros((T, N), dtype=torch.float32, device=d)
        self.log_probs = torch.zeros((T, N), dtype=torch.float32, device=d)
        self.values = torch.zeros((T, N), dtype=torch.float32, device=d)
        self.dones = torch.zeros((T, N), dtype=torch.bool, device=d)
        self.advantages = torch.zeros((T, N), dtype=torch.float32, device=d)
        self.returns = torch.zeros((T, N), dtype=torch.float32, device=d)
        self._advantages_computed = False
        # True only while the mask rows were written by the env kernel (SelfPlayRunner sets it): such rows never hold more
        # than 593 legal actions, which lets learn() skip the fused head's dense-row launches.  Masks from anywhere else
        # (hand-made, e.g. the all-ones masks of the reference's unit tests) keep the dense path available.
        self.masks_from_env = False

    def compute_advantages_and_returns(self, last_values):
        """last_values: float32 [N] bootstrap values of obs[T] (experience_buffer.py:99-145 per game)."""
        self.advantages, self.returns = ops.gae(self.rewards, self.values, self.dones, last_values, self.gamma,
                                                self.lambda_gae)
        self._advantages_computed = True

    def get_batch(self):
        if not self._advantages_computed:
            raise RuntimeError("Cannot get batch: compute_advantages_and_returns() must be called first")
        T, N = self.T, self.N
        f = lambda x: x.reshape(T * N, *x.shape[2:])
        out = {"obs": f(self.obs[:T]), "actions": f(self.actions), "log_probs": f(self.log_probs),
               "values": f(self.values), "rewards": f(self.rewards), "advantages": f(self.advantages),
               "returns": f(self.returns), "dones": f(self.dones), "legal_masks": f(self.legal_masks[:T])}
        if self.legal_bits is not None:
            out["legal_bits"] = f(self.legal_bits[:T])     # what the loss kernel reads; "legal_masks" stays the contract
        out["masks_from_env"] = bool(self.masks_from_env)
        return out

    def clear(self):
        # carry the bootstrap observation over as the first observation of the next rollout
        self.obs[0].copy_(self.obs[self.T])
        self.legal_masks[0].copy_(self.legal_masks[self.T])
        if self.legal_bits is not None and self.legal_bits is not self.legal_masks:
            self.legal_bits[0].copy_(self.legal_bits[self.T])
        self._advantages_computed = False
