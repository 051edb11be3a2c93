"""Learner-side operators over the HIP kernels (no torch fallbacks).

  masked_sample   <- BaseActorCriticModel.get_action_and_value tail (keisei/core/base_actor_critic.py:64-116)
  gae             <- ExperienceBuffer.compute_advantages_and_returns (keisei/core/experience_buffer.py:99-145)
  ppo_loss        <- evaluate_actions tail + PPOAgent.learn loss lines (base_actor_critic.py:147-184,
                     keisei/core/ppo_agent.py:340-383), forward and backward fused in one kernel
  policy_head_*   <- the same three tails FUSED with the policy head's last nn.Linear (training/models/resnet_tower.py:63-69):
                     they take the Linear's input features + weight + bias instead of logits; logits [B,13527] and their
                     gradient never exist in memory (csrc/ksh_head_kernels.h)
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


# ---------------------------------------------------------------------------------------------- fused policy head
def pack_mask_bits(mask):
    """bool / uint8 [B,13527] -> int32 [B,424] by the HIP kernel (pack_mask above is the torch helper for tests)"""
    _need_cuda(mask)
    mask = _mask_u8(mask)
    B = mask.shape[0]
    assert mask.shape == (B, NUM_ACTIONS)
    bits = torch.empty((B, MASK_WORDS), dtype=torch.int32, device=mask.device)
    check(lib().ksh_pack_mask(mask.data_ptr(), B, bits.data_ptr(), _stream()), "ksh_pack_mask")
    return bits


def _head_args(feat, weight, bias, legal_mask):
    _need_cuda(feat, weight, bias, legal_mask)
    if feat.dtype != torch.float32 or weight.dtype != torch.float32 or (bias is not None and bias.dtype != torch.float32):
        raise TypeError("the fused policy head runs in float32 (features, weight and bias)")
    B, K = feat.shape
    if weight.shape != (NUM_ACTIONS, K) or (bias is not None and bias.shape != (NUM_ACTIONS,)):
        raise ValueError("weight must be [13527, K] and bias [13527] for features [B, K]")
    if K % 2 or K < 4 or K > 1312:
        raise ValueError("the fused policy head needs an even K with 4 <= K <= 1312, got %d" % K)
    bits = legal_mask.contiguous() if is_packed(legal_mask) else pack_mask_bits(legal_mask)
    if bits.shape != (B, MASK_WORDS):
        raise ValueError("legal mask must have one row per feature row")
    return feat.contiguous(), weight.contiguous(), None if bias is None else bias.contiguous(), bits, B, K


def policy_head_sample(feat, weight, bias, legal_mask, seed, step, deterministic=False):
    """get_action_and_value's tail fused with the policy head's Linear layer: feat [B,K] float32, weight [13527,K], bias
    [13527] or None, legal_mask bool [B,13527] or packed int32 [B,424] -> (action int64 [B], log_prob [B]).  Same RNG
    stream and results as masked_sample(feat @ weight.T + bias, ...) up to float32 summation order of the logits.
    `step`: Python int or a one-element int64 CUDA tensor (hipGraph-captured rollouts)."""
    feat, weight, bias, bits, B, K = _head_args(feat, weight, bias, legal_mask)
    action = torch.empty(B, dtype=torch.int64, device=feat.device)
    logp = torch.empty(B, dtype=torch.float32, device=feat.device)
    step_dev = None
    if isinstance(step, torch.Tensor):
        if step.dtype != torch.int64 or step.numel() != 1 or step.device != feat.device:
            raise TypeError("a device step counter must be a one-element int64 tensor on the features' device")
        step_dev, step = step.data_ptr(), 0
    check(lib().ksh_policy_head_sample(feat.data_ptr(), weight.data_ptr(), None if bias is None else bias.data_ptr(), K,
                                       bits.data_ptr(), B, int(seed), int(step), step_dev, int(bool(deterministic)),
                                       action.data_ptr(), logp.data_ptr(), _stream()), "ksh_policy_head_sample")
    return action, logp


def policy_head_evaluate(feat, weight, bias, legal_mask, actions, dense_rows_possible=True):
    """evaluate_actions' tail fused with the Linear layer, no gradient: (log_probs [B], entropy [B])."""
    feat, weight, bias, bits, B, K = _head_args(feat, weight, bias, legal_mask)
    if actions.dtype != torch.int64:
        raise TypeError("actions must be int64")
    logp = torch.empty(B, dtype=torch.float32, device=feat.device)
    ent = torch.empty(B, dtype=torch.float32, device=feat.device)
    L = lib()
    d = int(bool(dense_rows_possible))
    ws = torch.empty(int(L.ksh_policy_head_evaluate_workspace_bytes(B, K, d)), dtype=torch.uint8, device=feat.device)
    check(L.ksh_policy_head_evaluate(feat.data_ptr(), weight.data_ptr(), None if bias is None else bias.data_ptr(), K,
                                     bits.data_ptr(), actions.contiguous().data_ptr(), B, d, logp.data_ptr(), ent.data_ptr(),
                                     ws.data_ptr(), _stream()), "ksh_policy_head_evaluate")
    return logp, ent


@torch.no_grad()
def policy_head_loss_grads(feat, weight, bias, values, legal_mask, actions, old_log_probs, advantages, returns, old_values=None,
                           clip_eps=0.2, vf_coef=0.5, ent_coef=0.01, value_clip=False, dense_rows_possible=True,
                           out_dweight=None, out_dbias=None):
    """The fused head's forward + backward as a plain function (no autograd node): returns
    (stats [8], dfeat [B,K], dweight [13527,K], dbias [13527] | None, dvalues [B]) = the loss terms and d loss / d input.
    out_dweight / out_dbias: dense float32 tensors the kernels write the parameter gradients into (e.g. views of a flat
    all-reduce bucket), so that no copy or scaling pass touches the 8.8 MB weight gradient afterwards."""
    if value_clip and old_values is None:
        raise ValueError("value_clip=True needs old_values")
    if actions.dtype != torch.int64:
        raise TypeError("actions must be int64")
    feat_c, weight_c, bias_c, bits, B, K = _head_args(feat, weight, bias, legal_mask)
    values_c = values.contiguous().float()
    f = lambda t: t.contiguous().float()
    dev = feat.device
    dfeat = torch.empty_like(feat_c)
    dW = out_dweight if out_dweight is not None else torch.empty_like(weight_c)
    if dW.shape != weight_c.shape or dW.dtype != torch.float32 or not dW.is_contiguous():
        raise ValueError("out_dweight must be a contiguous float32 tensor shaped like weight")
    dbias = None
    if bias is not None:
        dbias = out_dbias if out_dbias is not None else torch.empty(NUM_ACTIONS, dtype=torch.float32, device=dev)
        if dbias.shape != (NUM_ACTIONS,) or dbias.dtype != torch.float32 or not dbias.is_contiguous():
            raise ValueError("out_dbias must be a contiguous float32 tensor of 13527 elements")
    dvalues = torch.empty(B, dtype=torch.float32, device=dev)
    stats = torch.empty(8, dtype=torch.float32, device=dev)
    L = lib()
    d = int(bool(dense_rows_possible))
    ws = torch.empty(int(L.ksh_policy_head_workspace_bytes(B, K, d)), dtype=torch.uint8, device=dev)
    ov = f(old_values) if old_values is not None else None
    check(L.ksh_policy_head_loss_fwd_bwd(
        feat_c.data_ptr(), weight_c.data_ptr(), None if bias_c is None else bias_c.data_ptr(), K, bits.data_ptr(),
        actions.contiguous().data_ptr(), f(old_log_probs).data_ptr(), f(advantages).data_ptr(), f(returns).data_ptr(),
        values_c.data_ptr(), ov.data_ptr() if ov is not None else None, B, float(clip_eps), float(vf_coef), float(ent_coef),
        int(bool(value_clip)), d, dfeat.data_ptr(), dW.data_ptr(), None if dbias is None else dbias.data_ptr(),
        dvalues.data_ptr(), stats.data_ptr(), ws.data_ptr(), _stream()), "ksh_policy_head_loss_fwd_bwd")
    return stats, dfeat, dW, dbias, dvalues


class _HeadLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, weight, bias, values, legal_mask, actions, old_log_probs, advantages, returns, old_values, clip_eps,
                vf_coef, ent_coef, value_clip, dense):
        stats, dfeat, dW, dbias, dvalues = policy_head_loss_grads(feat, weight, bias, values, legal_mask, actions, old_log_probs,
                                                                  advantages, returns, old_values, clip_eps, vf_coef, ent_coef,
                                                                  value_clip, dense)
        ctx.save_for_backward(dfeat, dW, dvalues, *([dbias] if dbias is not None else []))
        ctx.has_bias = dbias is not None
        ctx.values_dtype = values.dtype
        ctx.mark_non_differentiable(stats)
        return stats[0].clone(), stats

    @staticmethod
    def backward(ctx, grad_loss, _grad_stats):
        saved = ctx.saved_tensors
        dfeat, dW, dvalues = saved[0], saved[1], saved[2]
        dbias = saved[3] if ctx.has_bias else None
        g = grad_loss
        # the kernels produced d loss / d inputs; autograd's incoming gradient is a scalar (1.0 when loss.backward())
        return (dfeat * g, dW * g, None if dbias is None else dbias * g, (dvalues * g).to(ctx.values_dtype)) + (None,) * 11


def policy_head_ppo_loss(feat, weight, bias, values, legal_mask, actions, old_log_probs, advantages, returns, old_values=None,
                         clip_eps=0.2, vf_coef=0.5, ent_coef=0.01, value_clip=False, dense_rows_possible=True):
    """ppo_loss with logits = feat @ weight.T + bias never materialised.  Returns (loss, stats); `loss` is differentiable
    w.r.t. feat, weight, bias and values.  dense_rows_possible=False skips the dense (MFMA) path and its scratch -- only
    for masks produced by the env kernel (<= 593 legal actions per row); a dense row then makes the loss NaN."""
    if value_clip and old_values is None:
        raise ValueError("value_clip=True needs old_values")
    if actions.dtype != torch.int64:
        raise TypeError("actions must be int64")
    return _HeadLossFn.apply(feat, weight, bias, values, legal_mask, actions, old_log_probs, advantages, returns, old_values,
                             clip_eps, vf_coef, ent_coef, value_clip, dense_rows_possible)
