"""VecPPOAgent -- batched counterpart of keisei/core/ppo_agent.py:PPOAgent for the GPU path.

  select_actions / select_action  <- PPOAgent.select_action (ppo_agent.py:134-223): model.train(is_training),
                                     no_grad forward, masked categorical sample / arg-max, log-prob, value
  get_values / get_value          <- PPOAgent.get_value (ppo_agent.py:225-241): eval-mode bootstrap value
  learn                           <- PPOAgent.learn (ppo_agent.py:243-460): whole-batch advantage normalisation,
                                     ppo_epochs x shuffled minibatches, clipped surrogate + value + entropy loss,
                                     grad-norm clipping, Adam, same metric keys

The network forward/backward stays in PyTorch-ROCm; everything after the logits (masked soft-max, sampling,
log-prob, entropy, the PPO loss and its gradient) runs in the HIP kernels of shogidrl_amd.ops.  Unlike the
reference there are no per-step `.item()` host syncs: metrics accumulate on the device and are read once
per learn() call."""
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import ops
from .mapper import PolicyOutputMapper
from .parallel import (FlatGradBucket, all_reduce_mean_, check_equal_batches, collectives_active, global_advantage_moments,
                       sync_buffers)


UPDATE_GRAPH_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"     # see PPOConfig.graph_minibatch_step


@dataclass
class PPOConfig:
    # defaults = default_config.yaml of the reference
    gamma: float = 0.99
    lambda_gae: float = 0.95
    clip_epsilon: float = 0.2
    value_loss_coeff: float = 0.5
    entropy_coef: float = 0.01
    ppo_epochs: int = 10
    minibatch_size: int = 64
    learning_rate: float = 3e-4
    weight_decay: float = 0.0
    gradient_clip_max_norm: float = 0.5
    normalize_advantages: bool = True
    enable_value_clipping: bool = False
    # Which log-probs are "old" in the PPO ratio.
    #   False = the reference: the ones select_action produced while acting (ppo_agent.py:150 acts in BatchNorm TRAIN
    #           mode, so they are taken under the batch statistics of ONE rollout step), although learn() re-evaluates
    #           the same samples under the statistics of shuffled cross-timestep minibatches.  Measured on the MI355X with
    #           the learning rate at zero: KL 0.043 and 49 % of the ratios outside the clip range BEFORE any update at
    #           the bench configuration (4096 games, ResNet 9x256; profiles/r02_old_logp_mismatch_lr0_*.log).
    #   True  = one extra no-grad pass over shuffled minibatches at the start of learn() re-computes them under
    #           learn-time statistics (running statistics untouched): KL 0.0000, clip fraction 0.000 in the same run.
    #   None  = automatic: True when the model has BatchNorm layers (a DELIBERATE deviation from the reference,
    #           DESIGN.md section 2), False otherwise (nothing to reconcile).
    recompute_old_log_probs: Optional[bool] = None
    # True: models that expose `policy_features(x)` / `policy_linear` (ResTowerActorCritic, TinyActorCritic) act and learn
    # through the FUSED policy head (ops.policy_head_*): the last nn.Linear of the policy head runs inside the sampler /
    # loss kernels and the [B,13527] logits and their gradient are never materialised.  False: logits through the
    # framework's GEMM, then ops.masked_sample / ops.ppo_loss.
    fused_policy_head: bool = True
    # True: the optimizer update of one minibatch (gather -> tower forward -> fused head -> tower backward -> clip -> Adam)
    # is captured ONCE per minibatch shape in a hipGraph and replayed for every later minibatch of that shape.  It pays
    # where the loop is launch-bound -- the reference's default minibatch_size of 64 (default_config.yaml:88): some 600
    # launches of a few microseconds each per update -- and changes nothing at GPU-sized minibatches.  Single rank, fused
    # policy head and a minibatch that divides the batch only; anything else takes the eager loop.
    # REQUIRES the process to run with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (set before the first HIP call): with the HIP
    # runtime's default AQL-packet capture, replays of this graph came back corrupted (garbage gradients, once a memory
    # fault) as soon as some ten thousand eager launches had gone through the same stream in between -- the re-computation
    # of the old log-probs at the start of the next learn() is enough.  Measured in profiles/r02_update_graph_corruption.txt;
    # the much smaller rollout graph (SelfPlayRunner.collect_graphed) is not affected.  VecPPOAgent refuses the option without
    # that setting.
    graph_minibatch_step: bool = False
    mixed_precision: bool = False          # bf16 autocast of the tower (the reference uses fp16 + GradScaler)
    channels_last: bool = False
    seed: int = 42


def _gather_obs(obs, idx):
    """Minibatch rows of the flattened observation store.  Compact observations (bf16, channel dimension innermost) are
    gathered through their dense [M,9,9,46] view so that the result is again a channels_last tensor, not an NCHW copy."""
    if obs.dim() == 4 and obs.stride(1) == 1 and obs.shape[1] > 1:
        return obs.permute(0, 2, 3, 1).index_select(0, idx).permute(0, 3, 1, 2)
    return obs.index_select(0, idx)


class _Acting:
    """The acting half of the agent (select_actions / get_values over self.model); shared by VecPPOAgent and by the
    weight-lagged ActingPolicy of the asynchronous learner."""

    def _init_acting(self):
        self.fused_head = bool(self.config.fused_policy_head and hasattr(self.model, "policy_features")
                               and hasattr(self.model, "policy_linear"))
        self._sample_step = 0
        self._step_tensor = None        # device-side twin of _sample_step while a captured rollout graph is in use

    # ------------------------------------------------------------------ acting
    def _forward(self, obs):
        if obs.dtype != torch.float32 and not self.config.mixed_precision:
            obs = obs.float()                                      # compact bf16 observations into an fp32 model
        if self.config.channels_last:
            obs = obs.contiguous(memory_format=torch.channels_last)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=self.config.mixed_precision):
            logits, value = self.model(obs)
        return logits, value.float().reshape(-1)

    def _features(self, obs):
        """(policy features [B,K] float32, value [B]) for the fused policy head"""
        if obs.dtype != torch.float32 and not self.config.mixed_precision:
            obs = obs.float()
        if self.config.channels_last:
            obs = obs.contiguous(memory_format=torch.channels_last)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=self.config.mixed_precision):
            feat, value = self.model.policy_features(obs)
        return feat.float(), value.float().reshape(-1)

    @torch.no_grad()
    def select_actions(self, obs, legal_mask, *, is_training=True):
        """obs [N,46,9,9], legal_mask bool [N,13527] (or the same sets bit-packed, int32 [N,424]) ->
        (action int64 [N], log_prob [N], value [N])."""
        self.model.train(is_training)                              # BatchNorm uses batch statistics while acting
        # graph-captured rollout: the Philox step counter lives on the device and the graph increments it
        step = self._step_tensor if self._step_tensor is not None else self._sample_step
        if self.fused_head:
            feat, value = self._features(obs)
            lin = self.model.policy_linear
            action, logp = ops.policy_head_sample(feat, lin.weight, lin.bias, legal_mask, self.config.seed, step,
                                                  deterministic=not is_training)
        else:
            logits, value = self._forward(obs)
            action, logp = ops.masked_sample(logits, legal_mask, self.config.seed, step, deterministic=not is_training)
        if self._step_tensor is not None:
            self._step_tensor.add_(1)
        else:
            self._sample_step += 1
        return action, logp, value

    @torch.no_grad()
    def get_values(self, obs):
        self.model.eval()
        _, value = self._features(obs) if self.fused_head else self._forward(obs)   # no logits GEMM for a value
        return value


class ActingPolicy(_Acting):
    """A copy of the learner's network that only acts: the behaviour policy of the asynchronous learner's rollout
    stream / actor ranks (the reference's SelfPlayWorker holds such a copy and refreshes it from the weights the trainer
    pushes, keisei/training/parallel/self_play_worker.py:356-435).  `load_from` is that refresh, device to device."""

    def __init__(self, model, config, device="cuda:0"):
        import copy
        self.config = config
        self.device = torch.device(device)
        self.model = copy.deepcopy(model).to(self.device)
        for p in self.model.parameters():
            p.requires_grad_(False)
        self._init_acting()
        self.weights_version = 0

    @torch.no_grad()
    def load_from(self, module, version=None):
        """parameters AND buffers (BatchNorm running statistics) of `module`, one multi-tensor copy"""
        src = [t.detach() for t in list(module.parameters()) + list(module.buffers())]
        dst = [t.detach() for t in list(self.model.parameters()) + list(self.model.buffers())]
        assert len(src) == len(dst)
        torch._foreach_copy_(dst, src)
        if version is not None:
            self.weights_version = version


class VecPPOAgent(_Acting):
    def __init__(self, model, config=None, device="cuda:0", group=None, name="PPOAgent"):
        self.config = config or PPOConfig()
        if self.config.graph_minibatch_step and os.environ.get(UPDATE_GRAPH_ENV) != "0" \
                and os.environ.get("KSH_UNSAFE_UPDATE_GRAPH") != "1":
            raise RuntimeError(
                "PPOConfig.graph_minibatch_step needs %s=0 in the environment before the process makes its first HIP call: with "
                "the runtime's default packet capture, replays of the captured update graph are corrupted once ~10^4 eager "
                "launches have run on the same stream (profiles/r02_update_graph_corruption.txt)" % UPDATE_GRAPH_ENV)
        self.device = torch.device(device)
        self.name = name
        self.model = model.to(self.device)
        if self.config.channels_last:
            self.model = self.model.to(memory_format=torch.channels_last)
        self.group = group
        self.policy_output_mapper = PolicyOutputMapper()
        self.num_actions_total = self.policy_output_mapper.get_total_actions()
        params = [p for p in self.model.parameters() if p.requires_grad]
        self.bucket = FlatGradBucket(params, group)
        kw = dict(lr=self.config.learning_rate, weight_decay=self.config.weight_decay)
        if self.config.graph_minibatch_step:
            kw["capturable"] = True                            # step counters live on the device (hipGraph replay)
        try:
            self.optimizer = torch.optim.Adam(params, fused=True, **kw)
        except (RuntimeError, TypeError, ValueError):
            self.optimizer = torch.optim.Adam(params, **kw)
        self._step_graphs = {}                                 # (minibatch rows, batch signature) -> captured update
        self._learn_id = 0
        self._init_acting()
        self._rng = np.random.default_rng(self.config.seed)       # minibatch shuffling (ppo_agent.py:100, 298)
        self._recompute_rng = np.random.default_rng(self.config.seed + 0x5EED)
        self.last_kl_div = 0.0
        self.last_gradient_norm = 0.0

    # ------------------------------------------------------------------ single-game signatures of the reference
    def select_action(self, obs, legal_mask, *, is_training=True):
        """Single-game signature of the reference: (move tuple | None, index, log_prob, value)."""
        o = torch.as_tensor(obs, dtype=torch.float32, device=self.device).reshape(1, 46, 9, 9)
        m = torch.as_tensor(legal_mask, device=self.device).reshape(1, -1).to(torch.bool)
        a, lp, v = self.select_actions(o, m, is_training=is_training)
        idx = int(a.item())
        return self.policy_output_mapper.policy_index_to_shogi_move(idx), idx, float(lp.item()), float(v.item())

    def get_value(self, obs):
        o = torch.as_tensor(obs, dtype=torch.float32, device=self.device).reshape(1, 46, 9, 9)
        return float(self.get_values(o).item())

    # ------------------------------------------------------------------ learning
    @torch.no_grad()
    def _recompute_log_probs(self, obs, masks, actions, dense=True):
        """log-probs of the stored actions under the CURRENT weights and learn-time (train-mode, minibatch) BatchNorm
        statistics; running statistics and num_batches_tracked are left exactly as they were."""
        bns = [m for m in self.model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        saved = [(m.momentum, m.num_batches_tracked.clone() if m.num_batches_tracked is not None else None) for m in bns]
        for m in bns:
            m.momentum = 0.0                                       # running = 1 * running + 0 * batch
        out = torch.empty(obs.shape[0], dtype=torch.float32, device=self.device)
        # SHUFFLED minibatches, like the ones learn() will use: a chunk of consecutive rows of a [T,N] rollout is one
        # timestep's N boards, i.e. exactly the batch the action was taken under -- re-evaluating it reproduces the stored
        # log-prob bit for bit and reconciles nothing.  Own generator: the learn-time shuffling stream is not disturbed.
        order = self._recompute_rng.permutation(obs.shape[0])
        perm = torch.from_numpy(order).to(self.device)
        try:
            for start in range(0, obs.shape[0], self.config.minibatch_size):
                idx = perm[start:start + self.config.minibatch_size]
                m_, a_ = masks.index_select(0, idx), actions.index_select(0, idx)
                if self.fused_head:
                    feat, _ = self._features(_gather_obs(obs, idx))
                    lin = self.model.policy_linear
                    lp, _ = ops.policy_head_evaluate(feat, lin.weight, lin.bias, m_, a_, dense)
                else:
                    logits, _ = self._forward(_gather_obs(obs, idx))
                    lp, _ = ops.evaluate_actions(logits, m_, a_)
                out.index_copy_(0, idx, lp)
        finally:
            for m, (mom, nbt) in zip(bns, saved):
                m.momentum = mom
                if nbt is not None:
                    m.num_batches_tracked.copy_(nbt)
        return out

    def learn(self, buffer):
        cfg = self.config
        self.model.train()
        batch = buffer.get_batch()
        lr = self.optimizer.param_groups[0]["lr"]
        if not batch or batch["obs"].shape[0] == 0:
            return {"ppo/policy_loss": 0.0, "ppo/value_loss": 0.0, "ppo/entropy": 0.0,
                    "ppo/kl_divergence_approx": self.last_kl_div, "ppo/learning_rate": lr}
        obs, actions, old_logp = batch["obs"], batch["actions"], batch["log_probs"]
        old_values, adv, ret = batch["values"], batch["advantages"], batch["returns"]
        masks = batch.get("legal_bits", batch["legal_masks"])     # the kernels read the bit-set when the buffer has it
        if cfg.normalize_advantages:
            n, mean, std = global_advantage_moments(adv, self.group)
            ok = (std > 1e-8) & (n > 1)                            # ppo_agent.py:277-282, decided on the device
            adv = torch.where(ok, ((adv.double() - mean) / torch.where(ok, std, torch.ones_like(std))).float(), adv)
        recompute = cfg.recompute_old_log_probs
        if recompute is None:
            recompute = any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in self.model.modules())
        # masks written by the env kernel never have more than 593 legal actions: the fused head can skip its dense path
        dense = not bool(batch.get("masks_from_env", False))
        if self.fused_head and not ops.is_packed(masks):
            masks = ops.pack_mask_bits(masks)                      # user-provided bool masks: packed once per learn()
        if recompute:
            old_logp = self._recompute_log_probs(obs, masks, actions, dense)
        num = obs.shape[0]
        check_equal_batches(num, self.device, self.group)          # same minibatch count (= collectives) on every rank
        indices = np.arange(num)
        self._learn_id += 1
        graphed = bool(cfg.graph_minibatch_step and self.fused_head and num % cfg.minibatch_size == 0
                       and not collectives_active(self.group))
        acc = torch.zeros(8, dtype=torch.float32, device=self.device)
        gnorm = torch.zeros((), dtype=torch.float32, device=self.device)
        updates = 0
        for _ in range(cfg.ppo_epochs):
            self._rng.shuffle(indices)
            perm = torch.from_numpy(indices.copy()).to(self.device)   # a copy: `indices` is reshuffled in place next epoch
            for start in range(0, num, cfg.minibatch_size):
                idx = perm[start:start + cfg.minibatch_size]
                if graphed:
                    gnorm = self._graphed_update(idx, obs, masks, actions, old_logp, adv, ret, old_values, dense, acc)
                    updates += 1
                    continue
                rest = (masks.index_select(0, idx), actions.index_select(0, idx), old_logp.index_select(0, idx),
                        adv.index_select(0, idx), ret.index_select(0, idx), old_values.index_select(0, idx),
                        cfg.clip_epsilon, cfg.value_loss_coeff, cfg.entropy_coef, cfg.enable_value_clipping)
                self.bucket.begin()
                if self.fused_head:
                    # fused policy head: the kernels return d loss / d (features, values) and write the Linear layer's
                    # weight / bias gradients straight into the flat all-reduce buffer; the tower's backward starts
                    # from the two small tensors (no autograd node, no scaling or copy pass over the 8.8 MB dW)
                    feat, values = self._features(_gather_obs(obs, idx))
                    lin = self.model.policy_linear
                    gw = self.bucket.view_of(lin.weight)
                    gb = self.bucket.view_of(lin.bias) if lin.bias is not None else None
                    direct = gw.is_contiguous()
                    stats, dfeat, dW, dbias, dvalues = ops.policy_head_loss_grads(
                        feat.detach(), lin.weight.detach(), None if lin.bias is None else lin.bias.detach(), values.detach(),
                        *rest, dense_rows_possible=dense, out_dweight=gw if direct else None, out_dbias=gb if direct else None)
                    torch.autograd.backward([feat, values], [dfeat, dvalues.to(values.dtype)])
                    lin.weight.grad = dW
                    if lin.bias is not None:
                        lin.bias.grad = dbias
                else:
                    logits, values = self._forward(_gather_obs(obs, idx))
                    loss, stats = ops.ppo_loss(logits, values, *rest)
                    loss.backward()
                self.bucket.gather()
                self.bucket.all_reduce_mean()                      # RCCL over xGMI when world_size > 1
                flat = self.bucket.flat
                gnorm = torch.linalg.vector_norm(flat)
                flat.mul_(torch.clamp(cfg.gradient_clip_max_norm / (gnorm + 1e-6), max=1.0))  # clip_grad_norm_
                self.optimizer.step()
                acc += stats
                updates += 1
        sync_buffers(self.model, self.group)                       # BatchNorm running statistics: mean over ranks
        acc = all_reduce_mean_(acc / max(updates, 1), self.group).cpu()   # the single host sync of learn()
        self.last_gradient_norm = float(gnorm)
        self.last_kl_div = float(acc[4])
        return {"ppo/policy_loss": float(acc[1]), "ppo/value_loss": float(acc[2]), "ppo/entropy": float(acc[3]),
                "ppo/kl_divergence_approx": float(acc[4]), "ppo/clip_fraction": float(acc[5]),
                "ppo/learning_rate": self.optimizer.param_groups[0]["lr"]}

    # ------------------------------------------------------------------ hipGraph-captured minibatch update
    def _update_from(self, idx, obs, masks, actions, old_logp, adv, ret, old_values, dense, acc):
        """one optimizer update on the rows `idx` (fused policy head, single rank): the body the graph captures"""
        cfg = self.config
        rest = (masks.index_select(0, idx), actions.index_select(0, idx), old_logp.index_select(0, idx),
                adv.index_select(0, idx), ret.index_select(0, idx), old_values.index_select(0, idx),
                cfg.clip_epsilon, cfg.value_loss_coeff, cfg.entropy_coef, cfg.enable_value_clipping)
        self.bucket.begin()
        feat, values = self._features(_gather_obs(obs, idx))
        lin = self.model.policy_linear
        gw = self.bucket.view_of(lin.weight)
        gb = self.bucket.view_of(lin.bias) if lin.bias is not None else None
        stats, dfeat, dW, dbias, dvalues = ops.policy_head_loss_grads(
            feat.detach(), lin.weight.detach(), None if lin.bias is None else lin.bias.detach(), values.detach(), *rest,
            dense_rows_possible=dense, out_dweight=gw, out_dbias=gb)
        torch.autograd.backward([feat, values], [dfeat, dvalues.to(values.dtype)])
        lin.weight.grad = dW
        if lin.bias is not None:
            lin.bias.grad = dbias
        self.bucket.gather()
        flat = self.bucket.flat
        gnorm = torch.linalg.vector_norm(flat)
        flat.mul_(torch.clamp(cfg.gradient_clip_max_norm / (gnorm + 1e-6), max=1.0))
        self.optimizer.step()
        acc += stats
        return gnorm

    def _graphed_update(self, idx, obs, masks, actions, old_logp, adv, ret, old_values, dense, acc):
        """Replay (capturing it on first use) the hipGraph of one minibatch update.  The graph reads the minibatch's row
        indices, the per-sample vectors that change between learn() calls (old log-probs, advantages) and the metric
        accumulator from STATIC tensors that are refreshed by copies (returns too: the GAE op hands out new tensors); the
        big buffers (observations, masks, actions, values) are captured by address, which is valid as long as learn() is handed the same buffer."""
        key = (int(idx.numel()), obs.data_ptr(), masks.data_ptr(), actions.data_ptr(), old_values.data_ptr(), tuple(obs.shape),
               bool(dense))
        g = self._step_graphs.get(key)
        if g is None:
            st = {"idx": idx.clone(), "old_logp": old_logp.clone(), "adv": adv.clone(), "ret": ret.clone(),
                  "acc": torch.zeros_like(acc), "warm": 0, "graph": None, "gnorm": None}
            self._step_graphs[key] = g = st
        g["idx"].copy_(idx)
        if g.get("learn_id") != self._learn_id:             # once per learn(): this call's old log-probs / advantages
            g["learn_id"] = self._learn_id
            g["old_logp"].copy_(old_logp)
            g["adv"].copy_(adv)
            g["ret"].copy_(ret)                               # (the GAE op returns fresh tensors every rollout)
        if g["graph"] is None and g["warm"] < 3:
            # eager passes first: lazy initialisations (Adam state, MIOpen solver choice, workspaces) must not be captured
            g["warm"] += 1
            gn = self._update_from(g["idx"], obs, masks, actions, g["old_logp"], g["adv"], g["ret"], old_values, dense, acc)
            return gn
        if g["graph"] is None:
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    g["gnorm"] = self._update_from(g["idx"], obs, masks, actions, g["old_logp"], g["adv"], g["ret"], old_values,
                                                   dense, g["acc"])
            torch.cuda.current_stream(self.device).wait_stream(side)
            g["graph"] = graph
        g["acc"].zero_()
        g["graph"].replay()
        acc += g["acc"]
        return g["gnorm"]

    # ------------------------------------------------------------------ checkpoints (ppo_agent.py:462-537 dict layout)
    def save_model(self, file_path, global_timestep=0, total_episodes_completed=0, stats_to_save=None):
        """Same dict as the reference: model_state_dict, optimizer_state_dict, global_timestep,
        total_episodes_completed (+ black_wins / white_wins / draws from stats_to_save)."""
        d = {"model_state_dict": self.model.state_dict(), "optimizer_state_dict": self.optimizer.state_dict(),
             "global_timestep": global_timestep, "total_episodes_completed": total_episodes_completed}
        if stats_to_save:
            d.update(stats_to_save)
        torch.save(d, file_path)

    def load_model(self, file_path):
        """Returns the reference's result dict; a missing / broken file yields zeros and an "error" entry
        (ppo_agent.py:489-537) instead of raising.  Files are read with weights_only=True."""
        blank = {"global_timestep": 0, "total_episodes_completed": 0, "black_wins": 0, "white_wins": 0, "draws": 0}
        if not (hasattr(file_path, "read") or os.path.exists(file_path)):
            return dict(blank, error="File not found")
        try:
            ck = torch.load(file_path, map_location=self.device, weights_only=True)
            self.model.load_state_dict(ck["model_state_dict"])
            self.optimizer.load_state_dict(ck["optimizer_state_dict"])
            return {"global_timestep": ck.get("global_timestep", 0),
                    "total_episodes_completed": ck.get("total_episodes_completed", 0),
                    "black_wins": ck.get("black_wins", 0), "white_wins": ck.get("white_wins", 0),
                    "draws": ck.get("draws", 0), "lr_schedule_type": ck.get("lr_schedule_type", None),
                    "lr_schedule_step_on": ck.get("lr_schedule_step_on", "epoch")}
        except (KeyError, RuntimeError, EOFError) as e:
            return dict(blank, error=str(e))
