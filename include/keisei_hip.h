/*
 * keisei_hip.h -- C-ABI of libkeisei_hip.so, the MI355X (gfx950) implementation of the Keisei
 * (tachyon-beep/shogidrl) self-play + PPO hot path.
 *
 * The reference has no FFI: the path is ordinary Python method calls.  Each entry
 * point below therefore cites the reference *Python* interface it replaces (paths relative to
 * /root/reference/keisei/), and INTEGRATION.md shows the ctypes binding a Keisei maintainer would add.
 *
 * Conventions
 *   - every pointer marked "dev" is caller-owned DEVICE memory (e.g. torch.Tensor.data_ptr()) on the
 *     device the handle was created on; "host" pointers are ordinary host memory;
 *   - every launch goes to the caller-supplied stream (hipStream_t passed as void*; NULL = default
 *     stream); no entry point synchronises unless it says so, none allocates after *_create, so
 *     all *_step / *_reset / kernels are hipGraph-capturable;
 *   - return value: 0 = OK, negative = KSH_E_* ; never throws; handles are not thread-safe;
 *   - tensors are dense row-major with exactly the reference's shapes and dtypes:
 *       obs   float32 [N,46,9,9]   (shogi/shogi_game_io.py:434-539)
 *       mask  uint8   [N,13527]    (utils/utils.py:310-336; torch.bool has the same 1-byte layout)
 *       action int64  [N]          (utils/utils.py:183-266 index space)
 */
#ifndef KEISEI_HIP_H
#define KEISEI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSH_NUM_ACTIONS 13527 /* constants.py:15 */
#define KSH_OBS_PLANES 46
#define KSH_OBS_FLOATS (46 * 81)
#define KSH_STATE_BYTES 128

/* error codes */
#define KSH_OK 0
#define KSH_E_ARG (-1)     /* bad argument (null pointer, size, misaligned buffer) */
#define KSH_E_HIP (-2)     /* a HIP runtime call failed; see ksh_last_error() */
#define KSH_E_STATE (-3)   /* invalid game state handed to ksh_env_set_state */

/* termination reason codes written to reason[] (shogi/shogi_core_definitions.py:135-141) */
#define KSH_REASON_ONGOING 0    /* "Game ongoing" (shogi_game.py:567-569) */
#define KSH_REASON_TSUMI 1      /* "Tsumi" */
#define KSH_REASON_STALEMATE 2  /* "stalemate" */
#define KSH_REASON_MAX_MOVES 3  /* "Max moves reached" */
#define KSH_REASON_SENNICHITE 4 /* "Sennichite" */
#define KSH_REASON_ILLEGAL 5    /* action rejected (the reference's ValueError -> episode reset,
                                   training/step_manager.py:303-348) */

const char *ksh_last_error(void);
int ksh_version(void);

/* ------------------------------------------------------------------------------------------------
 * Batched environment.  Replaces, for N games at once:
 *   ShogiGame(max_moves_per_game)                     shogi/shogi_game.py:39-54
 *   ShogiGame.reset()                                  shogi/shogi_game.py:113-130
 *   ShogiGame.get_legal_moves()                        shogi/shogi_game.py:216 -> shogi_rules_logic.py:486-635
 *   PolicyOutputMapper.get_legal_mask(moves, device)   utils/utils.py:310-336
 *   ShogiGame.make_move(move)                          shogi/shogi_game.py:574-660 (+ termination :408-450,
 *                                                      sennichite shogi_rules_logic.py:638-695)
 *   ShogiGame.get_observation()                        shogi/shogi_game_io.py:434-539
 *   StepManager episode reset on done                  training/step_manager.py:437-468
 * ---------------------------------------------------------------------------------------------- */
int ksh_env_create(int device, int n_games, int max_moves, void **handle);
int ksh_env_destroy(void *handle);
int ksh_env_num_games(void *handle);
int ksh_env_max_moves(void *handle);

/* Observation format of every later call on this handle (SURVEY 8 f4 -- a bandwidth reduction beyond the reference
 * contract, off by default):
 *   KSH_OBS_F32_NCHW   float32 [N][46][9][9], ShogiGame.get_observation's layout (shogi_game_io.py:434-539)
 *   KSH_OBS_BF16_NHWC  bfloat16 [N][9][9][46] (channels last), 7 452 B per game: bit for bit
 *                      obs.to(bfloat16) in channels_last memory format, i.e. the tensor a bf16-autocast stem
 *                      convolution (PPOAgent.learn's autocast, core/ppo_agent.py:321-330) reads.  The `obs`
 *                      arguments below then point at that buffer (still 4-byte aligned). */
#define KSH_OBS_F32_NCHW 0
#define KSH_OBS_BF16_NHWC 1
int ksh_env_set_obs_format(void *handle, int obs_format);
int ksh_env_obs_format(void *handle);

/* Where the env kernel reads its 24.9 KB of attack tables from (a tuning switch, results are identical):
 *   KSH_TABLES_LDS     every block stages a copy into LDS (33.6 KB of LDS per block, 4 waves per SIMD) -- faster up to
 *                      ~32 768 games per GPU (4096 games: 24.6 vs 26.6 us per step)
 *   KSH_TABLES_GLOBAL  lookups go to global memory (L1 / L2 resident; 9.4 KB of LDS, 7 waves per SIMD) -- faster at
 *                      65 536 games (369 vs 387 us); float32 observation format only
 *   KSH_TABLES_AUTO    the library's choice by n_games (the default at ksh_env_create; the environment variable
 *                      KSH_ENV_TABLES=lds|global overrides it for A-B runs) */
#define KSH_TABLES_AUTO 0
#define KSH_TABLES_LDS 1
#define KSH_TABLES_GLOBAL 2
int ksh_env_set_table_mode(void *handle, int table_mode);
int ksh_env_table_mode(void *handle);

/* Reset the games whose reset_mask[g] != 0 (all games when reset_mask == NULL) to the initial position
 * and write obs / legal mask / legal-move count of EVERY game (non-reset games are re-observed). */
int ksh_env_reset(void *handle, const uint8_t *reset_mask /*dev [N] or NULL*/, float *obs /*dev*/,
                  uint8_t *legal_mask /*dev*/, int32_t *n_legal /*dev [N]*/, void *stream);

/* Re-encode the current state of every game without changing it (observation + legal mask). */
int ksh_env_observe(void *handle, float *obs /*dev*/, uint8_t *legal_mask /*dev*/, int32_t *n_legal /*dev*/, void *stream);

/* Like ksh_env_observe, and additionally evaluates the termination of every (not yet finished) game's current
 * position the way ShogiGame.from_sfen does after loading one (shogi/shogi_game.py:331-343): no legal move ->
 * Tsumi (winner = the side NOT to move) or stalemate; otherwise move_count >= max_moves -> "Max moves reached".
 * Finished games are marked in their state record; done / reason / winner receive the verdicts. */
int ksh_env_evaluate_termination(void *handle, float *obs /*dev*/, uint8_t *legal_mask /*dev*/, int32_t *n_legal /*dev*/,
                                 uint8_t *done /*dev*/, uint8_t *reason /*dev*/, int8_t *winner /*dev*/, void *stream);

/* One env step for every game: apply actions[g], detect termination, optionally auto-reset finished
 * games, then write the NEXT observation / legal mask (of the reset position for finished games when
 * auto_reset != 0, exactly as StepManager discards the terminal observation).
 *   reward[g]  +1.0 iff the mover delivered Tsumi, else 0.0      (shogi_game.py:553-566)
 *   done[g]    1 iff the game ended with this move
 *   reason[g]  KSH_REASON_*, winner[g]  -1 none / 0 black / 1 white
 * prev_mask (nullable): the legal mask of the position the action is taken from (normally the buffer
 * the previous call wrote; it may alias next_mask).  When given, an action whose mask byte is 0 is
 * rejected with KSH_REASON_ILLEGAL instead of being applied. */
int ksh_env_step(void *handle, const int64_t *actions /*dev [N]*/, const uint8_t *prev_mask /*dev or NULL*/,
                 float *next_obs /*dev*/, uint8_t *next_mask /*dev*/, int32_t *n_legal /*dev*/,
                 float *reward /*dev*/, uint8_t *done /*dev*/, uint8_t *reason /*dev*/, int8_t *winner /*dev*/,
                 int auto_reset, void *stream);

/* Bit-packed legal masks (beyond the drop-in contract): the same information as the bool[13527]
 * mask in 1 696 bytes per game -- uint32 [N][424], bit i of a row <-> action id i (little-endian bit order, i.e.
 * numpy.packbits(mask, bitorder="little")), rows 16-byte aligned, bits >= 13527 zero.  The *_packed entry points
 * write / read this form; the byte mask output becomes optional (pass NULL to skip its 13.5 KB per game-step). */
#define KSH_MASK_WORDS 424
int ksh_env_reset_packed(void *handle, const uint8_t *reset_mask /*dev or NULL*/, float *obs /*dev*/,
                         uint8_t *legal_mask /*dev or NULL*/, uint32_t *legal_bits /*dev [N,424]*/, int32_t *n_legal /*dev*/,
                         void *stream);
int ksh_env_step_packed(void *handle, const int64_t *actions /*dev*/, const uint32_t *prev_bits /*dev or NULL*/,
                        float *next_obs /*dev*/, uint8_t *next_mask /*dev or NULL*/, uint32_t *next_bits /*dev [N,424]*/,
                        int32_t *n_legal /*dev*/, float *reward /*dev*/, uint8_t *done /*dev*/, uint8_t *reason /*dev*/,
                        int8_t *winner /*dev*/, int auto_reset, void *stream);
int ksh_masked_sample_packed(const void *logits /*dev [B,A]*/, int logits_dtype, const uint32_t *mask_bits /*dev [B,424]*/,
                             int B, uint64_t seed, uint64_t step, int deterministic, int64_t *action /*dev*/,
                             float *log_prob /*dev*/, void *stream);
int ksh_ppo_loss_fwd_bwd_packed(const void *logits /*dev [B,A]*/, int logits_dtype, const uint32_t *mask_bits /*dev [B,424]*/,
                                const int64_t *actions, const float *old_log_probs, const float *advantages,
                                const float *returns, const float *values, const float *old_values, int B, float clip_eps,
                                float vf_coef, float ent_coef, int value_clip, void *dlogits, float *dvalues, float *stats,
                                void *workspace, void *stream);

/* Host-side state exchange for tests / fixtures (ShogiGame.from_sfen, shogi_game.py:283-345, is host
 * code: parse the SFEN on the host, then hand the packed 128-byte record over).  Both calls
 * synchronise the given stream.  Record layout: bytes 0-80 board (0 empty, else 1+type+14*color,
 * sq=row*9+col), 81-87 black hand P..R, 88-94 white hand, 95 side to move, 96-97 move_count (LE),
 * 98 game_over, 99 reason, 100 winner+1, 102-103 history base (LE), rest zero. */
int ksh_env_set_state(void *handle, int game, const uint8_t *state128 /*host*/, void *stream);
int ksh_env_get_state(void *handle, int game, uint8_t *state128 /*host*/, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Masked categorical action selection.  Replaces BaseActorCriticModel.get_action_and_value's tail
 * (core/base_actor_critic.py:64-116): masked softmax -> Categorical(probs) sample (or argmax when
 * deterministic) -> log_prob, for B rows at once.  logits_dtype: 0 = float32, 1 = bfloat16.
 * RNG: counter-based Philox4x32-10 keyed by (seed, step), one stream per row.
 * ---------------------------------------------------------------------------------------------- */
int ksh_masked_sample(const void *logits /*dev [B,A]*/, int logits_dtype, const uint8_t *mask /*dev [B,A]*/,
                      int B, uint64_t seed, uint64_t step, int deterministic,
                      int64_t *action /*dev [B]*/, float *log_prob /*dev [B]*/, void *stream);
/* The same sampler for hipGraph-captured rollouts: the Philox step counter is read from device memory at run time
 * (`step_counter`, one uint64 the caller increments inside the graph), so a replayed graph draws fresh numbers.  Exactly
 * one of mask (bytes) / mask_bits (packed) is non-NULL.  ksh_masked_sample(..., step = s) == this with *step_counter = s. */
int ksh_masked_sample_graph(const void *logits /*dev [B,A]*/, int logits_dtype, const uint8_t *mask, const uint32_t *mask_bits,
                            int B, uint64_t seed, const uint64_t *step_counter /*dev*/, int deterministic,
                            int64_t *action /*dev [B]*/, float *log_prob /*dev [B]*/, void *stream);

/* ------------------------------------------------------------------------------------------------
 * GAE.  Replaces ExperienceBuffer.compute_advantages_and_returns (core/experience_buffer.py:99-145)
 * for N independent streams stored [T,N] row-major (N = 1 is the reference's single stream).
 * Bit-exact float32 op order; gamma/lambda are doubles because the reference folds them in Python
 * doubles before the float32 multiply.
 * ---------------------------------------------------------------------------------------------- */
int ksh_gae_scan(const float *rewards /*dev [T,N]*/, const float *values /*dev [T,N]*/,
                 const uint8_t *dones /*dev [T,N]*/, const float *last_values /*dev [N]*/,
                 double gamma, double lambda, int T, int N,
                 float *advantages /*dev [T,N]*/, float *returns /*dev [T,N]*/, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused PPO clipped-surrogate loss, forward + backward.  Replaces evaluate_actions' tail
 * (core/base_actor_critic.py:147-184) and the loss lines of PPOAgent.learn (core/ppo_agent.py:340-383)
 * for one minibatch of B samples:
 *   loss = policy_loss + vf_coef * value_loss + ent_coef * (-mean entropy)
 * Outputs d(loss)/d(logits) [B,A] (same dtype as logits), d(loss)/d(values) [B] and
 * stats[8] = {loss, policy_loss, value_loss, entropy_loss(-mean H), kl_approx, clip_fraction, 0, 0}.
 * value_clip != 0 selects the clipped value loss (ppo_agent.py:357-369).
 * workspace: dev scratch of at least ksh_ppo_loss_workspace_bytes(B) bytes.
 * ---------------------------------------------------------------------------------------------- */
uint64_t ksh_ppo_loss_workspace_bytes(int B);
int ksh_ppo_loss_fwd_bwd(const void *logits /*dev [B,A]*/, int logits_dtype, const uint8_t *mask /*dev [B,A]*/,
                         const int64_t *actions /*dev [B]*/, const float *old_log_probs /*dev [B]*/,
                         const float *advantages /*dev [B]*/, const float *returns /*dev [B]*/,
                         const float *values /*dev [B]*/, const float *old_values /*dev [B]*/, int B,
                         float clip_eps, float vf_coef, float ent_coef, int value_clip,
                         void *dlogits /*dev [B,A]*/, float *dvalues /*dev [B]*/, float *stats /*dev [8]*/,
                         void *workspace /*dev*/, void *stream);

/* evaluate_actions' tail alone (core/base_actor_critic.py:118-184): log-prob of actions[b] under the masked soft-max
 * and the entropy of the row's distribution, with Categorical(probs)'s eps-clamp semantics -- the loss kernels with the
 * gradient switched off.  Exactly one of mask (bytes) / mask_bits (packed) is non-NULL.
 * workspace: at least ksh_evaluate_actions_workspace_bytes(B) bytes of device scratch. */
uint64_t ksh_evaluate_actions_workspace_bytes(int B);
int ksh_evaluate_actions(const void *logits /*dev [B,A]*/, int logits_dtype, const uint8_t *mask, const uint32_t *mask_bits,
                         const int64_t *actions /*dev [B]*/, int B, float *log_probs /*dev [B]*/, float *entropy /*dev [B]*/,
                         void *workspace /*dev*/, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused policy head.  Replaces the LAST LAYER of the actor-critic's policy head together with the tail that consumes
 * its output, so that logits [B,13527] (and, in the backward, dlogits) never exist in memory:
 *   nn.Linear(2*9*9, 13527)                               training/models/resnet_tower.py:63-69
 *   (nn.Linear(16*9*9, 13527) of the tiny CNN             core/neural_network.py:17)
 *   + get_action_and_value's tail                         core/base_actor_critic.py:64-116   -> ksh_policy_head_sample
 *   + evaluate_actions' tail                              core/base_actor_critic.py:139-184  -> ksh_policy_head_evaluate
 *   + PPOAgent.learn's loss lines and their autograd      core/ppo_agent.py:340-383          -> ksh_policy_head_loss_fwd_bwd
 * feat  float32 [B,K]  the Linear layer's input (K even, 4 <= K <= 1312; 162 for the ResNet tower), 8-byte aligned
 * W     float32 [13527,K] = nn.Linear.weight,  bias float32 [13527] (or NULL) = nn.Linear.bias
 * mask_bits uint32 [B,424]  packed legal masks (KSH_MASK_WORDS; ksh_pack_mask converts bool[13527] rows)
 * Sparse rows (<= 1024 legal actions; a Shogi position has at most 593) gather only the legal rows of W; rows with
 * more legal actions (hand-made masks) go through float32 MFMA GEMMs -- set dense_rows_possible = 0 only when the
 * masks come from the env kernel: the dense path's launches and scratch are then skipped, and a dense row, should one
 * appear anyway, turns stats / dvalues into NaN instead of being silently mis-evaluated.
 * Semantics, RNG and stats[8] are those of ksh_masked_sample / ksh_ppo_loss_fwd_bwd with logits = feat W^T + bias;
 * outputs dfeat [B,K], dW [13527,K], dbias [13527] (or NULL), dvalues [B] are the gradients of `loss`.
 * workspace: 256-byte aligned device scratch of ksh_policy_head_workspace_bytes (loss) /
 * ksh_policy_head_evaluate_workspace_bytes (evaluate) bytes (229 MB / 0.2 MB at B = 4096 without the dense path; the
 * transposed dz matrix [13527][B] is most of it).  Diagnostic switches of the weight-gradient launch, read per call from
 * the environment: KSH_HEAD_LIGHT_MAX (samples up to which one wave takes an action, default 96) and
 * KSH_HEAD_WORK_BLOCKS (grid of the work kernel, default 768).
 * ---------------------------------------------------------------------------------------------- */
int ksh_pack_mask(const uint8_t *mask /*dev [B,13527]*/, int B, uint32_t *mask_bits /*dev [B,424]*/, void *stream);
int ksh_policy_head_sample(const float *feat, const float *W, const float *bias, int K, const uint32_t *mask_bits, int B,
                           uint64_t seed, uint64_t step, const uint64_t *step_counter /*dev or NULL: added to step*/,
                           int deterministic, int64_t *action /*dev [B]*/, float *log_prob /*dev [B]*/, void *stream);
uint64_t ksh_policy_head_evaluate_workspace_bytes(int B, int K, int dense_rows_possible);
int ksh_policy_head_evaluate(const float *feat, const float *W, const float *bias, int K, const uint32_t *mask_bits,
                             const int64_t *actions, int B, int dense_rows_possible, float *log_probs /*dev [B]*/,
                             float *entropy /*dev [B]*/, void *workspace, void *stream);
uint64_t ksh_policy_head_workspace_bytes(int B, int K, int dense_rows_possible);
int ksh_policy_head_loss_fwd_bwd(const float *feat, const float *W, const float *bias, int K, const uint32_t *mask_bits,
                                 const int64_t *actions, const float *old_log_probs, const float *advantages,
                                 const float *returns, const float *values, const float *old_values, int B, float clip_eps,
                                 float vf_coef, float ent_coef, int value_clip, int dense_rows_possible, float *dfeat, float *dW,
                                 float *dbias, float *dvalues, float *stats, void *workspace, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Tower glue: the normalisation / activation / residual work between the convolutions of the actor-critic
 * (training/models/resnet_tower.py:28-46 ResidualBlock.forward, :15-26 SqueezeExcitation.forward, :82-83 stem),
 * fused; the convolutions and linear layers themselves stay in PyTorch-ROCm.  Activations are channels-last:
 * dense [M][C] matrices, M = N*H*W, dtype 0 = float32 / 1 = bfloat16, 16-byte aligned, C % 8 == 0 and
 * 256 % (C/8) == 0.  Per-channel parameters and statistics are float32.  `workspace`: device scratch of
 * ksh_tower_workspace_bytes(C) bytes, reusable by consecutive calls on one stream.
 *
 * ksh_bn_fwd_train  nn.BatchNorm2d in training mode (+ optional residual add, + optional ReLU):
 *                   y = [relu](w * (x - mean) / sqrt(var + eps) + b [+ residual]); batch statistics with the
 *                   biased variance; running_mean / running_var (may be NULL) blended with `momentum`, the
 *                   running variance from the unbiased estimate; save_mean / save_invstd [C] for the backward.
 *                   pre_bias [C] (optional): bias of the convolution that produced x, folded in instead of being
 *                   added to x first -- BN(x + pb) == BN(x) with the batch mean moved by pb, so it only enters
 *                   the running mean (its gradient is exactly zero).
 * ksh_bn_fwd_eval   y = [relu]((x - coef[1][c]) * coef[0][c] + coef[2][c] [+ residual]) with caller-computed
 *                   coefficients (eval mode: coef = {w / sqrt(running_var + eps), running_mean, b}).
 * ksh_bn_bwd        gradients of ksh_bn_fwd_train: dx, optional dresidual (= dy masked by the ReLU; pass NULL
 *                   when there was no residual), dweight / dbias [C]; `y` is only read when relu != 0.
 * ksh_se_pool       pooled[n][c] = mean over the HW positions of x[n] (adaptive_avg_pool2d(x, 1)).
 * ksh_se_apply      z = relu(x * scale[n][c] + residual)   (x * s; out += x; relu)
 * ksh_se_bwd_reduce dscale[n][c] = sum over HW of dz * x, dz = dy * (z > 0)
 * ksh_se_bwd_apply  dx = dz * scale + dpooled / HW, dresidual = dz
 * ---------------------------------------------------------------------------------------------- */
uint64_t ksh_tower_workspace_bytes(int C);
int ksh_bn_fwd_train(const void *x, const void *residual /*or NULL*/, void *y, int dtype, int64_t M, int C,
                     const float *pre_bias /*or NULL*/, const float *weight, const float *bias, float *running_mean, float *running_var, float momentum,
                     float eps, int relu, float *save_mean, float *save_invstd, void *workspace, void *stream);
int ksh_bn_fwd_eval(const void *x, const void *residual /*or NULL*/, void *y, int dtype, int64_t M, int C,
                    const float *coef /*dev [3][C]*/, int relu, void *stream);
int ksh_bn_bwd(const void *dy, const void *x, const void *y, void *dx, void *dresidual /*or NULL*/, int dtype, int64_t M,
               int C, const float *weight, const float *save_mean, const float *save_invstd, int relu, float *dweight,
               float *dbias, void *workspace, void *stream);
int ksh_se_pool(const void *x, int dtype, int N, int HW, int C, float *pooled /*dev [N][C]*/, void *stream);
int ksh_se_apply(const void *x, const float *scale /*dev [N][C]*/, const void *residual, void *z, int dtype, int N, int HW,
                 int C, void *stream);
int ksh_se_bwd_reduce(const void *dy, const void *z, const void *x, int dtype, int N, int HW, int C,
                      float *dscale /*dev [N][C]*/, void *stream);
int ksh_se_bwd_apply(const void *dy, const void *z, const float *scale, const float *dpooled /*dev [N][C]*/, void *dx,
                     void *dresidual, int dtype, int N, int HW, int C, void *stream);

/* bn2 + SE tail of a residual block with two passes merged into their neighbours (training mode):
 * ksh_bn_pool_fwd_train = ksh_bn_fwd_train without ReLU / residual that also returns pooled[n][c] = mean over HW of y
 *                         (ksh_se_pool's result) from the same apply pass;
 * ksh_se_bn_bwd         = ksh_se_bwd_apply whose pass also accumulates the batch-norm backward sums, followed by the
 *                         batch-norm finalize and apply: dz -> (dresidual, dx of the batch norm's input x, dweight, dbias).
 *                         dy2: device scratch of the activation's size.  N * HW rows; one block per ceil(N / 1024) samples. */
int ksh_bn_pool_fwd_train(const void *x, void *y, int dtype, int N, int HW, int C, const float *pre_bias /*or NULL*/,
                          const float *weight, const float *bias, float *running_mean, float *running_var, float momentum,
                          float eps, float *save_mean, float *save_invstd, float *pooled /*dev [N][C]*/, void *workspace,
                          void *stream);
int ksh_se_bn_bwd(const void *dz, const void *z, const float *scale /*dev [N][C]*/, const float *dpooled /*dev [N][C]*/,
                  const void *x, void *dy2 /*scratch*/, void *dx, void *dresidual, int dtype, int N, int HW, int C,
                  const float *weight, const float *save_mean, const float *save_invstd, float *dweight, float *dbias,
                  void *workspace, void *stream);

/* hipEvent helpers so that host code without a HIP binding (bench.py) can time launches on the very
 * stream they were issued on.  ksh_event_elapsed_ms synchronises on `stop`. */
int ksh_event_create(void **event);
int ksh_event_record(void *event, void *stream);
int ksh_event_elapsed_ms(void *start, void *stop, float *ms /*host*/);
int ksh_event_destroy(void *event);

#ifdef __cplusplus
}
#endif
#endif
