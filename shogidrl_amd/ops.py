"""Learner-side operators over the HIP kernels (no torch fallbacks).

  masked_sample   <- BaseActorCriticModel.get_action_and_value tail (keisei/core/base_actor_critic.py:64-116)
  gae             <- ExperienceBuffer.compute_advantages_and_returns (keisei/core/experience_buffer.py:99-145)
  ppo_loss        <- evaluate_actions tail + PPOAgent.learn loss lines (base_actor_critic.py:147-184,
                     keisei/core/ppo_agent.py:340-383), forward and backward fused in one kernel
"""
import torch

from ._lib import MASK_WORDS, NUM_ACTIONS, KeiseiHipError, check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and t.device.type != "cuda":
            raise KeiseiHipError("shogidrl_amd ops run on an MI355X only; got a %s tensor (no CPU path)" % t.device)


def _dtype_code(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise TypeError("logits must be float32 or bfloat16, got %s" % t.dtype)


def _mask_u8(mask):
    if mask.dtype not in (torch.bool, torch.uint8):
        raise TypeError("legal mask must be bool / uint8 [B,13527] or bit-packed int32 [B,424]")
    return mask.contiguous()


def is_packed(mask):
    """bit-packed legal mask: int32 [B, 424], bit i of a row <-> action id i"""
    return mask.dtype == torch.int32 and mask.shape[-1] == MASK_WORDS


def pack_mask(mask):
    """bool [B,13527] -> int32 [B,424] (host-side helper for tests / interop; the env kernel writes this form itself)"""
    B = mask.shape[0]
    m = torch.zeros((B, MASK_WORDS * 32), dtype=torch.bool, device=mask.device)
    m[:, :NUM_ACTIONS] = mask.to(torch.bool)
    w = (m.view(B, MASK_WORDS, 32).to(torch.int64) << torch.arange(32, device=mask.device)).sum(-1)
    return w.to(torch.int32)     # two's-complement wrap of the 32-bit pattern


def unpack_mask(bits):
    """int32 [B,424] -> bool [B,13527]"""
    B = bits.shape[0]
    w = bits.to(torch.int64) & 0xFFFFFFFF
    m = ((w.unsqueeze(-1) >> torch.arange(32, device=bits.device)) & 1).to(torch.bool)
    return m.view(B, MASK_WORDS * 32)[:, :NUM_ACTIONS].contiguous()


def masked_sample(logits, legal_mask, seed, step, deterministic=False):
    """logits [B,13527] f32|bf16, legal_mask [B,13527] bool (or packed int32 [B,424]) -> (action int64 [B],
    log_prob f32 [B]).  `step`: the Philox step counter -- a Python int, or a one-element int64 CUDA tensor read by the
    kernel at run time (hipGraph-captured rollouts: increment the tensor inside the graph)."""
    _need_cuda(logits, legal_mask)
    B = logits.shape[0]
    logits = logits.contiguous()
    action = torch.empty(B, dtype=torch.int64, device=logits.device)
    logp = torch.empty(B, dtype=torch.float32, device=logits.device)
    if isinstance(step, torch.Tensor):
        if step.dtype != torch.int64 or step.numel() != 1 or step.device != logits.device:
            raise TypeError("a device step counter must be a one-element int64 tensor on the logits' device")
        packed = is_packed(legal_mask)
        assert logits.shape == (B, NUM_ACTIONS) and legal_mask.shape == (B, MASK_WORDS if packed else NUM_ACTIONS)
        legal_mask = legal_mask.contiguous() if packed else _mask_u8(legal_mask)
        check(lib().ksh_masked_sample_graph(logits.data_ptr(), _dtype_code(logits), None if packed else legal_mask.data_ptr(),
                                            legal_mask.data_ptr() if packed else None, B, int(seed), step.data_ptr(),
                                            int(bool(deterministic)), action.data_ptr(), logp.data_ptr(), _stream()),
              "ksh_masked_sample_graph")
        return action, logp
    if is_packed(legal_mask):
        assert logits.shape == (B, NUM_ACTIONS) and legal_mask.shape == (B, MASK_WORDS)
        legal_mask = legal_mask.contiguous()
        check(lib().ksh_masked_sample_packed(logits.data_ptr(), _dtype_code(logits), legal_mask.data_ptr(), B, int(seed),
                                             int(step), int(bool(deterministic)), action.data_ptr(), logp.data_ptr(),
                                             _stream()), "ksh_masked_sample_packed")
        return action, logp
    assert logits.shape == (B, NUM_ACTIONS) and legal_mask.shape == (B, NUM_ACTIONS)
    legal_mask = _mask_u8(legal_mask)
    check(lib().ksh_masked_sample(logits.data_ptr(), _dtype_code(logits), legal_mask.data_ptr(), B, int(seed), int(step),
                                  int(bool(deterministic)), action.data_ptr(), logp.data_ptr(), _stream()),
          "ksh_masked_sample")
    return action, logp


def gae(rewards, values, dones, last_values, gamma, lam):
    """[T,N] float32 rewards/values, [T,N] bool dones, [N] last_values -> (advantages, returns) [T,N]."""
    _need_cuda(rewards, values, dones, last_values)
    T, N = rewards.shape
    rewards = rewards.contiguous().float()
    values = values.contiguous().float()
    dones = dones.contiguous()
    if dones.dtype not in (torch.bool, torch.uint8):
        raise TypeError("dones must be bool or uint8")
    last_values = last_values.contiguous().float()
    adv = torch.empty_like(rewards)
    ret = torch.empty_like(rewards)
    check(lib().ksh_gae_scan(rewards.data_ptr(), values.data_ptr(), dones.data_ptr(), last_values.data_ptr(),
                             float(gamma), float(lam), T, N, adv.data_ptr(), ret.data_ptr(), _stream()), "ksh_gae_scan")
    return adv, ret


def evaluate_actions(logits, legal_mask, actions):
    """(log_probs [B], entropy [B]) of `actions` under the masked soft-max of `logits` -- the tail of
    BaseActorCriticModel.evaluate_actions (base_actor_critic.py:118-184), no gradient."""
    _need_cuda(logits, legal_mask, actions)
    B = logits.shape[0]
    logits = logits.contiguous()
    packed = is_packed(legal_mask)
    assert logits.shape == (B, NUM_ACTIONS) and legal_mask.shape == (B, MASK_WORDS if packed else NUM_ACTIONS)
    mask = legal_mask.contiguous() if packed else _mask_u8(legal_mask)
    if actions.dtype != torch.int64:
        raise TypeError("actions must be int64")
    logp = torch.empty(B, dtype=torch.float32, device=logits.device)
    ent = torch.empty(B, dtype=torch.float32, device=logits.device)
    L = lib()
    ws = torch.empty(int(L.ksh_evaluate_actions_workspace_bytes(B)), dtype=torch.uint8, device=logits.device)
    check(L.ksh_evaluate_actions(logits.data_ptr(), _dtype_code(logits), None if packed else mask.data_ptr(),
                                 mask.data_ptr() if packed else None, actions.contiguous().data_ptr(), B, logp.data_ptr(),
                                 ent.data_ptr(), ws.data_ptr(), _stream()), "ksh_evaluate_actions")
    return logp, ent


class _PPOLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, values, legal_mask, actions, old_log_probs, advantages, returns, old_values, clip_eps,
                vf_coef, ent_coef, value_clip):
        _need_cuda(logits, values, legal_mask)
        B = logits.shape[0]
        logits_c = logits.contiguous()
        values_c = values.contiguous().float()
        packed = is_packed(legal_mask)
        mask = legal_mask.contiguous() if packed else _mask_u8(legal_mask)
        f = lambda t: t.contiguous().float()
        dlogits = torch.empty_like(logits_c)
        dvalues = torch.empty(B, dtype=torch.float32, device=logits.device)
        stats = torch.empty(8, dtype=torch.float32, device=logits.device)
        L = lib()
        ws = torch.empty(int(L.ksh_ppo_loss_workspace_bytes(B)), dtype=torch.uint8, device=logits.device)
        ov = f(old_values) if old_values is not None else None
        fn = L.ksh_ppo_loss_fwd_bwd_packed if packed else L.ksh_ppo_loss_fwd_bwd
        check(fn(logits_c.data_ptr(), _dtype_code(logits_c), mask.data_ptr(),
                                     actions.contiguous().data_ptr(), f(old_log_probs).data_ptr(), f(advantages).data_ptr(),
                                     f(returns).data_ptr(), values_c.data_ptr(), ov.data_ptr() if ov is not None else None,
                                     B, float(clip_eps), float(vf_coef), float(ent_coef), int(bool(value_clip)),
                                     dlogits.data_ptr(), dvalues.data_ptr(), stats.data_ptr(), ws.data_ptr(), _stream()),
              "ksh_ppo_loss_fwd_bwd_packed" if packed else "ksh_ppo_loss_fwd_bwd")
        ctx.save_for_backward(dlogits, dvalues)
        ctx.values_dtype = values.dtype
        ctx.mark_non_differentiable(stats)
        return stats[0].clone(), stats

    @staticmethod
    def backward(ctx, grad_loss, _grad_stats):
        dlogits, dvalues = ctx.saved_tensors
        return (dlogits * grad_loss.to(dlogits.dtype), (dvalues * grad_loss).to(ctx.values_dtype),
                None, None, None, None, None, None, None, None, None, None)


def ppo_loss(logits, values, legal_mask, actions, old_log_probs, advantages, returns, old_values=None,
             clip_eps=0.2, vf_coef=0.5, ent_coef=0.01, value_clip=False):
    """Returns (loss, stats) with stats = [loss, policy_loss, value_loss, entropy_loss(-mean H), kl, clip_frac, 0, 0].
    `loss` is differentiable w.r.t. logits and values (gradients come from the fused kernel)."""
    if value_clip and old_values is None:
        raise ValueError("value_clip=True needs old_values")
    if actions.dtype != torch.int64:
        raise TypeError("actions must be int64")
    return _PPOLossFn.apply(logits, values, legal_mask, actions, old_log_probs, advantages, returns, old_values,
                            clip_eps, vf_coef, ent_coef, value_clip)
