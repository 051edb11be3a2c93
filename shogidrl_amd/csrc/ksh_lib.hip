// ksh_lib.hip -- libkeisei_hip.so: the C-ABI of include/keisei_hip.h over the gfx950 kernels.
// Built by __graft_entry__.build() with: hipcc -O3 --offload-arch=gfx950 -shared -fPIC
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/keisei_hip.h"
#include "ksh_env_kernel.h"
#include "ksh_ppo_kernels.h"
#include "ksh_head_kernels.h"
#include "ksh_tower_kernels.h"

static thread_local std::string g_err;

static int fail_hip(hipError_t e, const char *what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return KSH_E_HIP;
}
static int fail_arg(const char *what) {
    g_err = what;
    return KSH_E_ARG;
}
#define HIP_TRY(x)                                   \
    do {                                             \
        hipError_t _e = (x);                         \
        if (_e != hipSuccess) return fail_hip(_e, #x); \
    } while (0)

struct KshEnv {
    int device;
    int n;
    int max_moves;
    int hist_cap;
    int obs_format;  // KSH_OBS_F32_NCHW | KSH_OBS_BF16_NHWC
    int table_mode;  // 0: attack tables staged into LDS by every block; 1: read from global memory (float32 obs format only)
    uint8_t *state;
    unsigned long long *hist;
    KshTables *tables;
};

extern "C" {

const char *ksh_last_error(void) { return g_err.c_str(); }
int ksh_version(void) { return 200; }

int ksh_env_create(int device, int n_games, int max_moves, void **handle) {
    if (!handle || n_games <= 0 || max_moves <= 0 || max_moves > 65534) return fail_arg("ksh_env_create: bad argument");
    HIP_TRY(hipSetDevice(device));
    KshEnv *e = new KshEnv();
    e->device = device;
    e->n = n_games;
    e->max_moves = max_moves;
    e->hist_cap = max_moves + 1;
    e->obs_format = KSH_OBS_F32_NCHW;
    {   // diagnostic / A-B switch; the default was chosen by measurement
        // measured (profiles/r02_env_tables_lds_vs_global.log): LDS copy wins at 4096 games (24.6 vs 26.6 us), ties at
        // 8192 .. 32 768 (within 2-4 % either way), loses at 65 536 (387 vs 369 us) where 7 instead of 4 waves per SIMD hide more latency
        const char *tm = getenv("KSH_ENV_TABLES");
        e->table_mode = n_games >= 65536 ? 1 : 0;
        if (tm != nullptr && strcmp(tm, "global") == 0) e->table_mode = 1;
        if (tm != nullptr && strcmp(tm, "lds") == 0) e->table_mode = 0;
    }
    e->state = nullptr;
    e->hist = nullptr;
    e->tables = nullptr;
    KshTables *t = new KshTables();
    ksh_build_tables(t);
    ksh_finish_record(t->init_state, max_moves);   // position key + next move-count plane value
    hipError_t err = hipMalloc(&e->state, (size_t)n_games * 128);
    if (err == hipSuccess) err = hipMalloc(&e->hist, (size_t)n_games * e->hist_cap * sizeof(unsigned long long));
    if (err == hipSuccess) err = hipMalloc(&e->tables, sizeof(KshTables));
    if (err == hipSuccess) err = hipMemcpy(e->tables, t, sizeof(KshTables), hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        std::vector<uint8_t> init((size_t)n_games * 128);
        for (int g = 0; g < n_games; g++) memcpy(&init[(size_t)g * 128], t->init_state, 128);
        err = hipMemcpy(e->state, init.data(), init.size(), hipMemcpyHostToDevice);
    }
    if (err == hipSuccess) err = hipMemset(e->hist, 0, (size_t)n_games * e->hist_cap * sizeof(unsigned long long));
    delete t;
    if (err != hipSuccess) {
        if (e->state) (void)hipFree(e->state);
        if (e->hist) (void)hipFree(e->hist);
        if (e->tables) (void)hipFree(e->tables);
        delete e;
        return fail_hip(err, "ksh_env_create");
    }
    *handle = e;
    return KSH_OK;
}

int ksh_env_destroy(void *handle) {
    if (!handle) return fail_arg("ksh_env_destroy: null handle");
    KshEnv *e = (KshEnv *)handle;
    (void)hipFree(e->state);
    (void)hipFree(e->hist);
    (void)hipFree(e->tables);
    delete e;
    return KSH_OK;
}

int ksh_env_num_games(void *handle) { return handle ? ((KshEnv *)handle)->n : KSH_E_ARG; }
int ksh_env_max_moves(void *handle) { return handle ? ((KshEnv *)handle)->max_moves : KSH_E_ARG; }

int ksh_env_set_obs_format(void *handle, int obs_format) {
    if (!handle) return fail_arg("ksh_env_set_obs_format: null handle");
    if (obs_format != KSH_OBS_F32_NCHW && obs_format != KSH_OBS_BF16_NHWC) return fail_arg("ksh_env_set_obs_format: unknown format");
    ((KshEnv *)handle)->obs_format = obs_format;
    return KSH_OK;
}
int ksh_env_obs_format(void *handle) { return handle ? ((KshEnv *)handle)->obs_format : KSH_E_ARG; }

int ksh_env_set_table_mode(void *handle, int table_mode) {
    if (!handle) return fail_arg("ksh_env_set_table_mode: null handle");
    KshEnv *e = (KshEnv *)handle;
    if (table_mode == KSH_TABLES_AUTO) e->table_mode = e->n >= 65536 ? 1 : 0;
    else if (table_mode == KSH_TABLES_LDS || table_mode == KSH_TABLES_GLOBAL) e->table_mode = table_mode == KSH_TABLES_GLOBAL ? 1 : 0;
    else return fail_arg("ksh_env_set_table_mode: unknown mode");
    return KSH_OK;
}
int ksh_env_table_mode(void *handle) {
    return handle ? (((KshEnv *)handle)->table_mode ? KSH_TABLES_GLOBAL : KSH_TABLES_LDS) : KSH_E_ARG;
}

static unsigned long long *g_stamps = nullptr;  // diagnostic builds only (ksh_debug_set_stamps)

static int launch_env(KshEnv *e, int mode, const int64_t *actions, const uint8_t *prev_mask, const uint8_t *reset_mask,
                      float *obs, uint8_t *mask, int32_t *n_legal, float *reward, uint8_t *done, uint8_t *reason,
                      int8_t *winner, int auto_reset, void *stream, const uint32_t *prev_bits = nullptr,
                      uint32_t *mask_bits = nullptr) {
    if (!obs || !n_legal) return fail_arg("env: obs / n_legal must not be null");
    if (!mask && !mask_bits) return fail_arg("env: give a byte mask, a packed mask, or both");
    if ((reinterpret_cast<size_t>(mask_bits) & 15) != 0) return fail_arg("env: packed mask must be 16-byte aligned");
    if ((reinterpret_cast<size_t>(obs) & 3) != 0) return fail_arg("env: obs must be 4-byte aligned");
    if (mode == 2 && (!actions || !reward || !done || !reason || !winner)) return fail_arg("env step: null output");
    KshStepArgs a;
    a.state = e->state;
    a.hist = e->hist;
    a.tables = e->tables;
    a.actions = reinterpret_cast<const long long *>(actions);
    a.prev_mask = prev_mask;
    a.prev_bits = prev_bits;
    a.mask_bits = mask_bits;
    a.reset_mask = reset_mask;
    a.obs = obs;
    a.mask = mask;
    a.n_legal = n_legal;
    a.reward = reward;
    a.done = done;
    a.reason = reason;
    a.winner = reinterpret_cast<signed char *>(winner);
    a.n_games = e->n;
    a.max_moves = e->max_moves;
    a.hist_cap = e->hist_cap;
    a.mode = mode;
    a.auto_reset = auto_reset;
    a.stamps = g_stamps;
    const int blocks = (e->n + KSH_WPB - 1) / KSH_WPB;
    if (e->obs_format == KSH_OBS_BF16_NHWC)
        hipLaunchKernelGGL(ksh_env_step_c16_kernel, dim3(blocks), dim3(64 * KSH_WPB), 0, (hipStream_t)stream, a);
    else if (e->table_mode == 1)
        hipLaunchKernelGGL(ksh_env_step_gt_kernel, dim3(blocks), dim3(64 * KSH_WPB), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(ksh_env_step_kernel, dim3(blocks), dim3(64 * KSH_WPB), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

int ksh_env_reset(void *handle, const uint8_t *reset_mask, float *obs, uint8_t *legal_mask, int32_t *n_legal,
                  void *stream) {
    if (!handle) return fail_arg("ksh_env_reset: null handle");
    return launch_env((KshEnv *)handle, 1, nullptr, nullptr, reset_mask, obs, legal_mask, n_legal, nullptr, nullptr,
                      nullptr, nullptr, 0, stream);
}

int ksh_env_observe(void *handle, float *obs, uint8_t *legal_mask, int32_t *n_legal, void *stream) {
    if (!handle) return fail_arg("ksh_env_observe: null handle");
    return launch_env((KshEnv *)handle, 0, nullptr, nullptr, nullptr, obs, legal_mask, n_legal, nullptr, nullptr,
                      nullptr, nullptr, 0, stream);
}

int ksh_env_evaluate_termination(void *handle, float *obs, uint8_t *legal_mask, int32_t *n_legal, uint8_t *done,
                                 uint8_t *reason, int8_t *winner, void *stream) {
    if (!handle) return fail_arg("ksh_env_evaluate_termination: null handle");
    if (!done || !reason || !winner) return fail_arg("ksh_env_evaluate_termination: null output");
    return launch_env((KshEnv *)handle, 3, nullptr, nullptr, nullptr, obs, legal_mask, n_legal, nullptr, done, reason, winner,
                      0, stream);
}

int ksh_env_step(void *handle, const int64_t *actions, const uint8_t *prev_mask, float *next_obs, uint8_t *next_mask,
                 int32_t *n_legal, float *reward, uint8_t *done, uint8_t *reason, int8_t *winner, int auto_reset,
                 void *stream) {
    if (!handle) return fail_arg("ksh_env_step: null handle");
    return launch_env((KshEnv *)handle, 2, actions, prev_mask, nullptr, next_obs, next_mask, n_legal, reward, done,
                      reason, winner, auto_reset, stream);
}

int ksh_env_reset_packed(void *handle, const uint8_t *reset_mask, float *obs, uint8_t *legal_mask, uint32_t *legal_bits,
                         int32_t *n_legal, void *stream) {
    if (!handle) return fail_arg("ksh_env_reset_packed: null handle");
    return launch_env((KshEnv *)handle, 1, nullptr, nullptr, reset_mask, obs, legal_mask, n_legal, nullptr, nullptr, nullptr,
                      nullptr, 0, stream, nullptr, legal_bits);
}

int ksh_env_step_packed(void *handle, const int64_t *actions, const uint32_t *prev_bits, float *next_obs, uint8_t *next_mask,
                        uint32_t *next_bits, int32_t *n_legal, float *reward, uint8_t *done, uint8_t *reason, int8_t *winner,
                        int auto_reset, void *stream) {
    if (!handle) return fail_arg("ksh_env_step_packed: null handle");
    return launch_env((KshEnv *)handle, 2, actions, nullptr, nullptr, next_obs, next_mask, n_legal, reward, done, reason, winner,
                      auto_reset, stream, prev_bits, next_bits);
}

int ksh_env_set_state(void *handle, int game, const uint8_t *state128, void *stream) {
    if (!handle || !state128) return fail_arg("ksh_env_set_state: null argument");
    KshEnv *e = (KshEnv *)handle;
    if (game < 0 || game >= e->n) return fail_arg("ksh_env_set_state: game index out of range");
    uint8_t rec[128];
    memset(rec, 0, sizeof rec);
    memcpy(rec, state128, 104);
    int pieces = 0;
    for (int s = 0; s < 81; s++) {
        if (rec[s] > 28) { g_err = "ksh_env_set_state: invalid piece code"; return KSH_E_STATE; }
        pieces += rec[s] != 0;
    }
    if (pieces > 64) { g_err = "ksh_env_set_state: more than 64 pieces on the board"; return KSH_E_STATE; }
    if (rec[S_SIDE] > 1) { g_err = "ksh_env_set_state: side to move must be 0 or 1"; return KSH_E_STATE; }
    rec[S_HBASE] = rec[S_MC];  // the position history starts empty at this ply (from_sfen: shogi_game.py:320-321)
    rec[S_HBASE + 1] = rec[S_MC + 1];
    ksh_finish_record(rec, e->max_moves);
    HIP_TRY(hipMemcpyAsync(e->state + (size_t)game * 128, rec, 128, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return KSH_OK;
}

int ksh_env_get_state(void *handle, int game, uint8_t *state128, void *stream) {
    if (!handle || !state128) return fail_arg("ksh_env_get_state: null argument");
    KshEnv *e = (KshEnv *)handle;
    if (game < 0 || game >= e->n) return fail_arg("ksh_env_get_state: game index out of range");
    HIP_TRY(hipMemcpyAsync(state128, e->state + (size_t)game * 128, 128, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return KSH_OK;
}

// ---------------------------------------------------------------------------------- PPO-side kernels
static int sample_impl(const void *logits, int logits_dtype, const uint8_t *mask, const uint32_t *mask_bits, int B,
                       uint64_t seed, uint64_t step, int deterministic, int64_t *action, float *log_prob, void *stream,
                       const uint64_t *step_dev = nullptr);

int ksh_masked_sample_graph(const void *logits, int logits_dtype, const uint8_t *mask, const uint32_t *mask_bits, int B,
                            uint64_t seed, const uint64_t *step_counter, int deterministic, int64_t *action, float *log_prob,
                            void *stream) {
    if ((!mask && !mask_bits) || !step_counter) return fail_arg("ksh_masked_sample_graph: bad argument");
    return sample_impl(logits, logits_dtype, mask_bits ? nullptr : mask, mask_bits, B, seed, 0, deterministic, action,
                       log_prob, stream, step_counter);
}

int ksh_masked_sample(const void *logits, int logits_dtype, const uint8_t *mask, int B, uint64_t seed, uint64_t step,
                      int deterministic, int64_t *action, float *log_prob, void *stream) {
    if (!mask) return fail_arg("ksh_masked_sample: bad argument");
    return sample_impl(logits, logits_dtype, mask, nullptr, B, seed, step, deterministic, action, log_prob, stream);
}

int ksh_masked_sample_packed(const void *logits, int logits_dtype, const uint32_t *mask_bits, int B, uint64_t seed,
                             uint64_t step, int deterministic, int64_t *action, float *log_prob, void *stream) {
    if (!mask_bits) return fail_arg("ksh_masked_sample_packed: bad argument");
    return sample_impl(logits, logits_dtype, nullptr, mask_bits, B, seed, step, deterministic, action, log_prob, stream);
}

static int sample_impl(const void *logits, int logits_dtype, const uint8_t *mask, const uint32_t *mask_bits, int B,
                       uint64_t seed, uint64_t step, int deterministic, int64_t *action, float *log_prob, void *stream,
                       const uint64_t *step_dev) {
    if (!logits || !action || !log_prob || B <= 0) return fail_arg("ksh_masked_sample: bad argument");
    if (logits_dtype != 0 && logits_dtype != 1) return fail_arg("ksh_masked_sample: logits_dtype must be 0 (f32) or 1 (bf16)");
    const dim3 grid((B + KSH_SAMPLE_WPB - 1) / KSH_SAMPLE_WPB), block(64 * KSH_SAMPLE_WPB);
    if (logits_dtype == 0)
        hipLaunchKernelGGL(ksh_masked_sample_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float *)logits, mask,
                           mask_bits, B, seed, step, reinterpret_cast<const unsigned long long *>(step_dev), deterministic,
                           reinterpret_cast<long long *>(action), log_prob);
    else
        hipLaunchKernelGGL(ksh_masked_sample_kernel<ksh_bf16>, grid, block, 0, (hipStream_t)stream,
                           (const ksh_bf16 *)logits, mask, mask_bits, B, seed, step,
                           reinterpret_cast<const unsigned long long *>(step_dev), deterministic,
                           reinterpret_cast<long long *>(action), log_prob);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

int ksh_gae_scan(const float *rewards, const float *values, const uint8_t *dones, const float *last_values, double gamma,
                 double lambda, int T, int N, float *advantages, float *returns, void *stream) {
    if (!rewards || !values || !dones || !last_values || !advantages || !returns || T <= 0 || N <= 0)
        return fail_arg("ksh_gae_scan: bad argument");
    const float g = (float)gamma;             // python float -> float32 scalar (experience_buffer.py:135)
    const float gl = (float)(gamma * lambda); // folded in double first (experience_buffer.py:138)
    const int threads = 256;
    hipLaunchKernelGGL(ksh_gae_kernel, dim3((N + threads - 1) / threads), dim3(threads), 0, (hipStream_t)stream, rewards,
                       values, dones, last_values, g, gl, T, N, advantages, returns);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

uint64_t ksh_ppo_loss_workspace_bytes(int B) { return (uint64_t)(B > 0 ? B : 0) * 8 * sizeof(float) + 64; }

static int loss_impl(const void *logits, int logits_dtype, const uint8_t *mask, const uint32_t *mask_bits,
                     const int64_t *actions, const float *old_log_probs, const float *advantages, const float *returns,
                     const float *values, const float *old_values, int B, float clip_eps, float vf_coef, float ent_coef,
                     int value_clip, void *dlogits, float *dvalues, float *stats, void *workspace, void *stream,
                     float *row_logp = nullptr, float *row_entropy = nullptr);

uint64_t ksh_evaluate_actions_workspace_bytes(int B) {
    // partial [B][8] + 5 dummy per-row input vectors (zeros are fine: only log-prob / entropy are read back) + dvalues + stats
    return ksh_ppo_loss_workspace_bytes(B) + (uint64_t)(B > 0 ? B : 0) * 2 * sizeof(float) + 64;
}

int ksh_evaluate_actions(const void *logits, int logits_dtype, const uint8_t *mask, const uint32_t *mask_bits,
                         const int64_t *actions, int B, float *log_probs, float *entropy, void *workspace, void *stream) {
    if ((!mask && !mask_bits) || !log_probs || !entropy || !workspace) return fail_arg("ksh_evaluate_actions: bad argument");
    // the loss kernels with the gradient switched off: per-row scalars that only feed the (discarded) loss terms read
    // a zeroed vector
    float *ws = reinterpret_cast<float *>(workspace);
    float *zeros = ws + (size_t)B * 8 + 16, *dval = zeros + B;
    HIP_TRY(hipMemsetAsync(zeros, 0, (size_t)B * sizeof(float), (hipStream_t)stream));
    return loss_impl(logits, logits_dtype, mask_bits ? nullptr : mask, mask_bits, actions, zeros, zeros, zeros, zeros, nullptr,
                     B, 0.2f, 0.0f, 0.0f, 0, nullptr, dval, ws + (size_t)B * 8, workspace, stream, log_probs, entropy);
}

int ksh_ppo_loss_fwd_bwd(const void *logits, int logits_dtype, const uint8_t *mask, const int64_t *actions,
                         const float *old_log_probs, const float *advantages, const float *returns, const float *values,
                         const float *old_values, int B, float clip_eps, float vf_coef, float ent_coef, int value_clip,
                         void *dlogits, float *dvalues, float *stats, void *workspace, void *stream) {
    if (!mask) return fail_arg("ksh_ppo_loss_fwd_bwd: bad argument");
    return loss_impl(logits, logits_dtype, mask, nullptr, actions, old_log_probs, advantages, returns, values, old_values, B,
                     clip_eps, vf_coef, ent_coef, value_clip, dlogits, dvalues, stats, workspace, stream);
}

int ksh_ppo_loss_fwd_bwd_packed(const void *logits, int logits_dtype, const uint32_t *mask_bits, const int64_t *actions,
                                const float *old_log_probs, const float *advantages, const float *returns,
                                const float *values, const float *old_values, int B, float clip_eps, float vf_coef,
                                float ent_coef, int value_clip, void *dlogits, float *dvalues, float *stats, void *workspace,
                                void *stream) {
    if (!mask_bits) return fail_arg("ksh_ppo_loss_fwd_bwd_packed: bad argument");
    return loss_impl(logits, logits_dtype, nullptr, mask_bits, actions, old_log_probs, advantages, returns, values, old_values,
                     B, clip_eps, vf_coef, ent_coef, value_clip, dlogits, dvalues, stats, workspace, stream);
}

static int loss_impl(const void *logits, int logits_dtype, const uint8_t *mask, const uint32_t *mask_bits,
                     const int64_t *actions, const float *old_log_probs, const float *advantages, const float *returns,
                     const float *values, const float *old_values, int B, float clip_eps, float vf_coef, float ent_coef,
                     int value_clip, void *dlogits, float *dvalues, float *stats, void *workspace, void *stream,
                     float *row_logp, float *row_entropy) {
    if (!logits || !actions || !old_log_probs || !advantages || !returns || !values || (!dlogits && !row_logp) || !dvalues ||
        !stats || !workspace || B <= 0)
        return fail_arg("ksh_ppo_loss_fwd_bwd: bad argument");
    if (value_clip && !old_values) return fail_arg("ksh_ppo_loss_fwd_bwd: value_clip needs old_values");
    if (logits_dtype != 0 && logits_dtype != 1) return fail_arg("ksh_ppo_loss_fwd_bwd: logits_dtype must be 0 or 1");
    KshLossArgs a;
    a.mask = mask;
    a.mask_bits = mask_bits;
    a.actions = reinterpret_cast<const long long *>(actions);
    a.old_logp = old_log_probs;
    a.adv = advantages;
    a.ret = returns;
    a.values = values;
    a.old_values = old_values;
    a.B = B;
    a.clip_eps = clip_eps;
    a.vf_coef = vf_coef;
    a.ent_coef = ent_coef;
    a.value_clip = value_clip;
    a.dvalues = dvalues;
    a.stats = stats;
    a.partial = reinterpret_cast<float *>(workspace);
    a.row_logp = row_logp;
    a.row_entropy = row_entropy;
    a.overflow_is_error = 0;
    // sparse wave-per-row kernel first; rows it flags (more legal actions than its list holds) go to the dense kernel
    a.only_overflow = 1;
    const dim3 sgrid((B + KSH_LOSS_WPB - 1) / KSH_LOSS_WPB), sblock(64 * KSH_LOSS_WPB);
    if (logits_dtype == 0) {
        hipLaunchKernelGGL(ksh_ppo_loss_sparse_kernel<float>, sgrid, sblock, 0, (hipStream_t)stream, a, (const float *)logits,
                           (float *)dlogits);
        hipLaunchKernelGGL(ksh_ppo_loss_kernel<float>, dim3(B), dim3(KSH_ROW_THREADS), 0, (hipStream_t)stream, a,
                           (const float *)logits, (float *)dlogits);
    } else {
        hipLaunchKernelGGL(ksh_ppo_loss_sparse_kernel<ksh_bf16>, sgrid, sblock, 0, (hipStream_t)stream, a,
                           (const ksh_bf16 *)logits, (ksh_bf16 *)dlogits);
        hipLaunchKernelGGL(ksh_ppo_loss_kernel<ksh_bf16>, dim3(B), dim3(KSH_ROW_THREADS), 0, (hipStream_t)stream, a,
                           (const ksh_bf16 *)logits, (ksh_bf16 *)dlogits);
    }
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(ksh_ppo_stats_kernel, dim3(1), dim3(KSH_STATS_THREADS), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

// ---------------------------------------------------------------------------------- fused policy head
// workspace layout: partial [B][8] floats | 256 B (word 0: dense-row counter; words 1-3: light-list length, "lists corrupt"
//                   flag, heavy-list length) | bitsT [A][BW] | dzT [A][B] | light list [A] | heavy list [A]
//                   | (dense) logits scratch [B][A] | (dense) dlogits scratch [B][A] | (dense) split-K tiles
struct HeadWs {
    float *partial;
    unsigned *n_over, *n_items;
    uint32_t *bitsT;
    float *dzT;
    uint32_t *items, *heavy;
    float *zbuf, *dlbuf, *cpart;
    int BW, SEGW, NSEG;
    size_t bytes;
};
#ifndef KSH_HEAD_KSPLIT
#define KSH_HEAD_KSPLIT 32
#endif
#ifndef KSH_HEAD_DW_KSPLIT
#define KSH_HEAD_DW_KSPLIT 2       // dW = dl^T feat: 212 row tiles for 256 CUs, one wave per SIMD -> two slices of the sample range
#endif
static size_t head_align(size_t x) { return (x + 255) & ~(size_t)255; }
static HeadWs head_ws(void *workspace, int B, int K, int dense, int grad) {      // grad = 0: the evaluate-only layout (no backward scratch)
    HeadWs w;
    char *p = (char *)workspace;
    w.partial = (float *)p;
    p += head_align((size_t)B * 32 + 64);
    w.n_over = (unsigned *)p;                 // 256 bytes of counters / flags, zeroed by one small memset per call
    w.n_items = w.n_over + 1;
    p += 256;
    w.BW = (B + 31) / 32;
    const int maxseg = K <= 192 ? 16 : 4;                 // segments of a heavy action = waves of a work-kernel block that share it
    w.SEGW = (w.BW + maxseg - 1) / maxseg;
    w.NSEG = (w.BW + w.SEGW - 1) / w.SEGW;
    w.bitsT = nullptr; w.dzT = nullptr; w.items = w.heavy = nullptr;
    if (grad) {
        w.bitsT = (uint32_t *)p;
        p += head_align((size_t)KSH_PA * w.BW * 4);
        w.dzT = (float *)p;
        p += head_align((size_t)KSH_PA * B * 4);
        w.items = (uint32_t *)p;
        p += head_align((size_t)KSH_PA * 4);
        w.heavy = (uint32_t *)p;
        p += head_align((size_t)KSH_PA * 4);
    }
    w.zbuf = w.dlbuf = w.cpart = nullptr;
    if (dense) {
        w.zbuf = (float *)p;
        p += head_align((size_t)KSH_PA * B * 4);
        if (grad) {
            w.dlbuf = (float *)p;
            p += head_align((size_t)KSH_PA * B * 4);
            w.cpart = (float *)p;                          // split-K partial tiles of the dense dfeat / dW products (used in turn)
            const size_t c1 = (size_t)KSH_HEAD_KSPLIT * B * K * 4, c2 = (size_t)KSH_HEAD_DW_KSPLIT * KSH_PA * (K + 1) * 4;
            p += head_align(c1 > c2 ? c1 : c2);
        }
    }
    w.bytes = (size_t)(p - (char *)workspace);
    return w;
}

uint64_t ksh_policy_head_workspace_bytes(int B, int K, int dense_rows_possible) {
    if (B <= 0 || K <= 0) return 0;
    return head_ws(nullptr, B, K, dense_rows_possible, 1).bytes;
}

int ksh_pack_mask(const uint8_t *mask, int B, uint32_t *mask_bits, void *stream) {
    if (!mask || !mask_bits || B <= 0) return fail_arg("ksh_pack_mask: bad argument");
    if ((reinterpret_cast<size_t>(mask_bits) & 3) != 0) return fail_arg("ksh_pack_mask: mask_bits must be 4-byte aligned");
    hipLaunchKernelGGL(ksh_pack_mask_kernel, dim3(B), dim3(KSH_ROW_THREADS), 0, (hipStream_t)stream, mask, B, mask_bits);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

static int head_common_ok(const char *who, const void *feat, const void *W, int K, const void *bits, int B) {
    if (!feat || !W || !bits || B <= 0) { g_err = std::string(who) + ": null pointer or B <= 0"; return 0; }
    if (K < 4 || (K & 1) != 0 || K > 1312) { g_err = std::string(who) + ": K must be even, >= 4 and <= 1312"; return 0; }
    if ((reinterpret_cast<size_t>(feat) & 7) != 0 || (reinterpret_cast<size_t>(W) & 7) != 0) {
        g_err = std::string(who) + ": feat / W must be 8-byte aligned";
        return 0;
    }
    return 1;
}

int ksh_policy_head_sample(const float *feat, const float *W, const float *bias, int K, const uint32_t *mask_bits, int B,
                           uint64_t seed, uint64_t step, const uint64_t *step_counter, int deterministic, int64_t *action,
                           float *log_prob, void *stream) {
    if (!head_common_ok("ksh_policy_head_sample", feat, W, K, mask_bits, B)) return KSH_E_ARG;
    if (!action || !log_prob) return fail_arg("ksh_policy_head_sample: null output");
    KshHeadArgs a;
    memset(&a, 0, sizeof a);
    a.feat = feat; a.W = W; a.bias = bias; a.K = K; a.mask_bits = mask_bits; a.B = B;
    a.seed = seed; a.step = step; a.step_dev = reinterpret_cast<const unsigned long long *>(step_counter);
    a.deterministic = deterministic;
    a.action = reinterpret_cast<long long *>(action);
    a.log_prob = log_prob;
    const dim3 grid((B + KSH_HEAD_WPB - 1) / KSH_HEAD_WPB), block(64 * KSH_HEAD_WPB);
    if (K <= 192) hipLaunchKernelGGL(ksh_head_sample_kernel<192>, grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ksh_head_sample_kernel<1312>, grid, block, 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

// shared by ksh_policy_head_evaluate (want_grad = 0) and ksh_policy_head_loss_fwd_bwd
static int head_loss_impl(const float *feat, const float *W, const float *bias, int K, const uint32_t *mask_bits,
                          const int64_t *actions, const float *old_log_probs, const float *advantages, const float *returns,
                          const float *values, const float *old_values, int B, float clip_eps, float vf_coef, float ent_coef,
                          int value_clip, int dense, int want_grad, float *dfeat, float *dW, float *dbias, float *dvalues,
                          float *stats, float *row_logp, float *row_entropy, void *workspace, hipStream_t st) {
    const HeadWs w = head_ws(workspace, B, K, dense, want_grad);
    KshHeadArgs a;
    memset(&a, 0, sizeof a);
    a.feat = feat; a.W = W; a.bias = bias; a.K = K; a.mask_bits = mask_bits; a.B = B;
    KshLossArgs &L = a.L;
    L.mask = nullptr;
    L.mask_bits = mask_bits;
    L.actions = reinterpret_cast<const long long *>(actions);
    L.old_logp = old_log_probs; L.adv = advantages; L.ret = returns; L.values = values; L.old_values = old_values;
    L.B = B; L.clip_eps = clip_eps; L.vf_coef = vf_coef; L.ent_coef = ent_coef; L.value_clip = value_clip;
    L.dvalues = dvalues; L.stats = stats; L.partial = w.partial; L.only_overflow = 1;
    L.row_logp = row_logp; L.row_entropy = row_entropy;
    L.overflow_is_error = dense ? 0 : 1;
    a.want_grad = want_grad;
    a.dfeat = dfeat; a.dzT = w.dzT; a.bitsT = w.bitsT; a.BW = w.BW; a.n_over = w.n_over;
    // the dense-row counter only matters to the dense path's launches (the work-list counters are zeroed by the transpose
    // launch): without a dense path there is nothing to clear
    if (dense) HIP_TRY(hipMemsetAsync(w.n_over, 0, 256, st));
    const dim3 grid((B + KSH_HEAD_WPB - 1) / KSH_HEAD_WPB), block(64 * KSH_HEAD_WPB);
    if (K <= 192) hipLaunchKernelGGL(ksh_head_loss_kernel<192>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(ksh_head_loss_kernel<1312>, grid, block, 0, st, a);
    KshGemmArgs g;
    memset(&g, 0, sizeof g);
    g.flags = w.partial + 6;
    g.n_over = w.n_over;
    const dim3 gb(256);
    if (dense) {
        // dense rows, forward: logits[r][a] = <feat_r, W_a> + bias_a on the matrix cores, then the block-per-row kernel
        g.mode = 0;
        g.A = feat; g.sAm = K; g.sAk = 1;
        g.Bm = W; g.sBk = 1; g.sBn = K;
        g.C = w.zbuf; g.sCm = KSH_PA;
        g.M = B; g.N = KSH_PA; g.Kd = K; g.bias = bias;
        hipLaunchKernelGGL(ksh_head_gemm_logits_kernel, dim3((B + KSH_GEMM_M0 - 1) / KSH_GEMM_M0, (KSH_PA + KSH_GEMM_N0 - 1) / KSH_GEMM_N0), gb, 0, st, g);
        hipLaunchKernelGGL(ksh_ppo_loss_kernel<float>, dim3(B), dim3(KSH_ROW_THREADS), 0, st, L, (const float *)w.zbuf,
                           want_grad ? w.dlbuf : (float *)nullptr);
    }
    if (want_grad) {
        // where each action was legal: the packed masks transposed (rows the loss kernel flagged as dense left out)
        // (+ one row of blocks: work-list counters zeroed, batch statistics -- see the kernel)
        hipLaunchKernelGGL(ksh_head_bits_transpose_kernel,
                           dim3((w.BW + KSH_BT_TS - 1) / KSH_BT_TS, (KSH_PBITROW + KSH_BT_TA - 1) / KSH_BT_TA + 1),
                           dim3(KSH_BT_THREADS), 0, st, mask_bits, (const float *)(w.partial + 6), B, w.BW, w.bitsT, L, w.n_items);
        KshHeadDwArgs d;
        d.feat = feat; d.dzT = w.dzT; d.bitsT = w.bitsT; d.dW = dW; d.dbias = dbias;
        d.B = B; d.BW = w.BW; d.K = K; d.A = KSH_PA;
        d.SEGW = w.SEGW; d.NSEG = w.NSEG; d.n_items = w.n_items; d.items = w.items; d.heavy = w.heavy;
        d.light_max = 96;                                                             // (sweep: profiles/r02_head_kernels_pmc.txt)
        if (const char *lm = getenv("KSH_HEAD_LIGHT_MAX")) d.light_max = atoi(lm);     // tuning switch (tools/headdiag.py)
        hipLaunchKernelGGL(ksh_head_dw_scan_kernel,
                           dim3((KSH_PA + KSH_DW_SCAN_WPB * KSH_DW_SCAN_APW - 1) / (KSH_DW_SCAN_WPB * KSH_DW_SCAN_APW)),
                           dim3(64 * KSH_DW_SCAN_WPB), 0, st, d);
        // a fixed grid walks the two lists (their lengths are only known on the device)
        int wblocks = 768;
        if (const char *wb = getenv("KSH_HEAD_WORK_BLOCKS")) wblocks = atoi(wb) > 0 ? atoi(wb) : wblocks;      // tuning switch
        if (K <= 192) hipLaunchKernelGGL(ksh_head_dw_work_kernel<192>, dim3(wblocks), dim3(64 * KSH_DW_WPB), 0, st, d);
        else hipLaunchKernelGGL(ksh_head_dw_work_kernel<1312>, dim3(wblocks), dim3(64 * KSH_DW_WPB), 0, st, d);
        if (dense) {
            // dense rows, backward: dfeat_r = dl_r W ; dW += dl^T feat and dbias += column sums over the dense rows
            g.mode = 1;
            g.A = w.dlbuf; g.sAm = KSH_PA; g.sAk = 1;
            g.Bm = W; g.sBk = K; g.sBn = 1;
            g.C = dfeat; g.sCm = K;
            g.M = B; g.N = K; g.Kd = KSH_PA; g.bias = nullptr;
            // 13 527 reduction steps for a handful of output tiles: split the reduction 32 ways, add the slices in order
            g.ksplit = KSH_HEAD_KSPLIT;
            g.kchunk = ((KSH_PA + KSH_HEAD_KSPLIT - 1) / KSH_HEAD_KSPLIT + KSH_GEMM_BK - 1) / KSH_GEMM_BK * KSH_GEMM_BK;
            g.Cpart = w.cpart;
            hipLaunchKernelGGL(ksh_head_gemm_dfeat_kernel, dim3((B + KSH_GEMM_M1 - 1) / KSH_GEMM_M1, (K + KSH_GEMM_N1 - 1) / KSH_GEMM_N1, KSH_HEAD_KSPLIT), gb, 0, st, g);
            hipLaunchKernelGGL(ksh_head_splitk_reduce_kernel, dim3(B), gb, 0, st, g);
            g.mode = 2;
            g.A = w.dlbuf; g.sAm = 1; g.sAk = KSH_PA;
            g.Bm = feat; g.sBk = K; g.sBn = 1;
            g.C = dW; g.sCm = K; g.C2 = dbias;
            g.M = KSH_PA; g.N = K + 1; g.Kd = B;
            // 212 row tiles would leave one wave per SIMD and 44 CUs idle: two slices of the sample range, added in order
            g.ksplit = KSH_HEAD_DW_KSPLIT;
            g.kchunk = ((B + KSH_HEAD_DW_KSPLIT - 1) / KSH_HEAD_DW_KSPLIT + KSH_GEMM_BK - 1) / KSH_GEMM_BK * KSH_GEMM_BK;
            hipLaunchKernelGGL(ksh_head_gemm_dw_kernel, dim3((KSH_PA + KSH_GEMM_M2 - 1) / KSH_GEMM_M2, (K + 1 + KSH_GEMM_N2 - 1) / KSH_GEMM_N2, KSH_HEAD_DW_KSPLIT), gb, 0, st, g);
            hipLaunchKernelGGL(ksh_head_splitk_reduce_dw_kernel, dim3((KSH_PA + 3) / 4), gb, 0, st, g);
        }
    }
    HIP_TRY(hipGetLastError());
    // (with gradients the batch statistics ran inside the transpose launch; the evaluate-only call has per-row outputs
    // only -- a dense row met without a dense path gets NaN there from the row kernel -- and needs none)
    return KSH_OK;
}

int ksh_policy_head_loss_fwd_bwd(const float *feat, const float *W, const float *bias, int K, const uint32_t *mask_bits,
                                 const int64_t *actions, const float *old_log_probs, const float *advantages,
                                 const float *returns, const float *values, const float *old_values, int B, float clip_eps,
                                 float vf_coef, float ent_coef, int value_clip, int dense_rows_possible, float *dfeat, float *dW,
                                 float *dbias, float *dvalues, float *stats, void *workspace, void *stream) {
    if (!head_common_ok("ksh_policy_head_loss_fwd_bwd", feat, W, K, mask_bits, B)) return KSH_E_ARG;
    if (!actions || !old_log_probs || !advantages || !returns || !values || !dfeat || !dW || !dvalues || !stats || !workspace)
        return fail_arg("ksh_policy_head_loss_fwd_bwd: null pointer");
    if (value_clip && !old_values) return fail_arg("ksh_policy_head_loss_fwd_bwd: value_clip needs old_values");
    if ((reinterpret_cast<size_t>(dfeat) & 7) != 0 || (reinterpret_cast<size_t>(workspace) & 255) != 0)
        return fail_arg("ksh_policy_head_loss_fwd_bwd: dfeat must be 8-byte, workspace 256-byte aligned");
    return head_loss_impl(feat, W, bias, K, mask_bits, actions, old_log_probs, advantages, returns, values, old_values, B, clip_eps,
                          vf_coef, ent_coef, value_clip, dense_rows_possible, 1, dfeat, dW, dbias, dvalues, stats, nullptr, nullptr,
                          workspace, (hipStream_t)stream);
}

uint64_t ksh_policy_head_evaluate_workspace_bytes(int B, int K, int dense_rows_possible) {
    if (B <= 0 || K <= 0) return 0;
    // the forward part of the loss layout, plus per-row dummies for the loss-only inputs / outputs
    return head_ws(nullptr, B, K, dense_rows_possible, 0).bytes + head_align((size_t)B * 8 + 64);
}

int ksh_policy_head_evaluate(const float *feat, const float *W, const float *bias, int K, const uint32_t *mask_bits,
                             const int64_t *actions, int B, int dense_rows_possible, float *log_probs, float *entropy,
                             void *workspace, void *stream) {
    if (!head_common_ok("ksh_policy_head_evaluate", feat, W, K, mask_bits, B)) return KSH_E_ARG;
    if (!actions || !log_probs || !entropy || !workspace) return fail_arg("ksh_policy_head_evaluate: null pointer");
    if ((reinterpret_cast<size_t>(workspace) & 255) != 0) return fail_arg("ksh_policy_head_evaluate: workspace must be 256-byte aligned");
    float *extra = (float *)((char *)workspace + head_ws(nullptr, B, K, dense_rows_possible, 0).bytes);
    float *zeros = extra, *dval = extra + B;        // loss-only inputs read zeros; dvalues / stats are discarded
    float *stats = dval + B;
    if (!dense_rows_possible)       // the row kernel reads null loss inputs as zeros: one launch, nothing to clear
        return head_loss_impl(feat, W, bias, K, mask_bits, actions, nullptr, nullptr, nullptr, nullptr, nullptr, B, 0.2f, 0.0f,
                              0.0f, 0, 0, 0, nullptr, nullptr, nullptr, dval, stats, log_probs, entropy, workspace,
                              (hipStream_t)stream);
    HIP_TRY(hipMemsetAsync(zeros, 0, (size_t)B * sizeof(float), (hipStream_t)stream));      // (the dense row kernel wants arrays)
    return head_loss_impl(feat, W, bias, K, mask_bits, actions, zeros, zeros, zeros, zeros, nullptr, B, 0.2f, 0.0f, 0.0f, 0,
                          dense_rows_possible, 0, nullptr, nullptr, nullptr, dval, stats, log_probs, entropy, workspace,
                          (hipStream_t)stream);
}

#ifdef KSH_STAMPS
int ksh_debug_set_stamps(void *dev_buffer) { g_stamps = (unsigned long long *)dev_buffer; return KSH_OK; }
#endif

// ---------------------------------------------------------------------------------- tower glue
// workspace layout (floats): partial [KSH_TW_MAXG][2][C], coef [3][C]
uint64_t ksh_tower_workspace_bytes(int C) { return C > 0 ? ((uint64_t)KSH_TW_MAXG * 2 + 3) * (uint64_t)C * 4 : 0; }

static int tower_shape_ok(const char *who, int dtype, int64_t M, int C) {
    if (dtype != 0 && dtype != 1) { g_err = std::string(who) + ": dtype must be 0 (f32) or 1 (bf16)"; return 0; }
    if (M <= 0 || C <= 0 || (C & 7) != 0 || (KSH_TW_THREADS % (C >> 3)) != 0 || (C >> 3) > KSH_TW_THREADS) {
        g_err = std::string(who) + ": need M > 0, C % 8 == 0 and 256 % (C/8) == 0";
        return 0;
    }
    return 1;
}
static int aligned16(const void *p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

extern "C++" {
template <typename T>
static void launch_bn_apply(const void *x, const void *res, void *y, const KshRowSplit &sp, const float *coef, int relu,
                            hipStream_t st) {
    const T *xp = (const T *)x, *rp = (const T *)res;
    T *yp = (T *)y;
    const dim3 g(sp.G), b(KSH_TW_THREADS);
    if (relu && res) hipLaunchKernelGGL((ksh_bn_apply_kernel<T, 1, 1>), g, b, 0, st, xp, rp, yp, sp, coef);
    else if (relu) hipLaunchKernelGGL((ksh_bn_apply_kernel<T, 1, 0>), g, b, 0, st, xp, rp, yp, sp, coef);
    else if (res) hipLaunchKernelGGL((ksh_bn_apply_kernel<T, 0, 1>), g, b, 0, st, xp, rp, yp, sp, coef);
    else hipLaunchKernelGGL((ksh_bn_apply_kernel<T, 0, 0>), g, b, 0, st, xp, rp, yp, sp, coef);
}
}  // extern "C++"

int ksh_bn_fwd_train(const void *x, const void *residual, void *y, int dtype, int64_t M, int C, const float *pre_bias,
                     const float *weight, const float *bias, float *running_mean, float *running_var, float momentum, float eps, int relu,
                     float *save_mean, float *save_invstd, void *workspace, void *stream) {
    if (!tower_shape_ok("ksh_bn_fwd_train", dtype, M, C)) return KSH_E_ARG;
    if (!x || !y || !save_mean || !save_invstd || !workspace) return fail_arg("ksh_bn_fwd_train: null pointer");
    if (!aligned16(x) || !aligned16(y) || !aligned16(residual)) return fail_arg("ksh_bn_fwd_train: activations must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    float *partial = (float *)workspace, *coef = partial + (size_t)KSH_TW_MAXG * 2 * C;
    const KshRowSplit sp = ksh_row_split(M, C, KSH_TW_MAXG);
    if (dtype == 0) hipLaunchKernelGGL((ksh_bn_stats_kernel<float>), dim3(sp.G), dim3(KSH_TW_THREADS), 0, st, (const float *)x, sp, (const float *)running_mean, pre_bias, partial);
    else hipLaunchKernelGGL((ksh_bn_stats_kernel<ksh_bf16>), dim3(sp.G), dim3(KSH_TW_THREADS), 0, st, (const ksh_bf16 *)x, sp, (const float *)running_mean, pre_bias, partial);
    hipLaunchKernelGGL(ksh_bn_finalize_kernel, dim3((C + KSH_TW_FIN_CH - 1) / KSH_TW_FIN_CH), dim3(KSH_TW_THREADS), 0, st,
                       (const float *)partial, sp, (const float *)running_mean, pre_bias, weight, bias, running_mean, running_var, momentum, eps, save_mean,
                       save_invstd, coef);
    const KshRowSplit sa = ksh_row_split(M, C, 4 * KSH_TW_MAXG);
    if (dtype == 0) launch_bn_apply<float>(x, residual, y, sa, coef, relu, st);
    else launch_bn_apply<ksh_bf16>(x, residual, y, sa, coef, relu, st);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

int ksh_bn_fwd_eval(const void *x, const void *residual, void *y, int dtype, int64_t M, int C, const float *coef, int relu,
                    void *stream) {
    if (!tower_shape_ok("ksh_bn_fwd_eval", dtype, M, C)) return KSH_E_ARG;
    if (!x || !y || !coef) return fail_arg("ksh_bn_fwd_eval: null pointer");
    if (!aligned16(x) || !aligned16(y) || !aligned16(residual)) return fail_arg("ksh_bn_fwd_eval: activations must be 16-byte aligned");
    const KshRowSplit sa = ksh_row_split(M, C, 4 * KSH_TW_MAXG);
    if (dtype == 0) launch_bn_apply<float>(x, residual, y, sa, coef, relu, (hipStream_t)stream);
    else launch_bn_apply<ksh_bf16>(x, residual, y, sa, coef, relu, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

extern "C++" {
template <typename T>
static void launch_bn_bwd(const void *dy, const void *x, const void *y, void *dx, void *dres, const KshRowSplit &sp,
                          const KshRowSplit &sa, const float *weight, const float *mean, const float *invstd, int relu,
                          float *dw, float *db, float *partial, float *coef, hipStream_t st) {
    const T *gp = (const T *)dy, *xp = (const T *)x, *yp = (const T *)y;
    T *dxp = (T *)dx, *drp = (T *)dres;
    const dim3 b(KSH_TW_THREADS);
    if (relu) hipLaunchKernelGGL((ksh_bn_bwd_reduce_kernel<T, 1>), dim3(sp.G), b, 0, st, gp, xp, yp, sp, mean, partial);
    else hipLaunchKernelGGL((ksh_bn_bwd_reduce_kernel<T, 0>), dim3(sp.G), b, 0, st, gp, xp, yp, sp, mean, partial);
    hipLaunchKernelGGL(ksh_bn_bwd_finalize_kernel, dim3((sp.C + KSH_TW_FIN_CH - 1) / KSH_TW_FIN_CH), b, 0, st, (const float *)partial, sp, weight, invstd,
                       dw, db, coef);
    const dim3 g(sa.G);
    if (relu && dres) hipLaunchKernelGGL((ksh_bn_bwd_apply_kernel<T, 1, 1>), g, b, 0, st, gp, xp, yp, dxp, drp, sa, mean, (const float *)coef);
    else if (relu) hipLaunchKernelGGL((ksh_bn_bwd_apply_kernel<T, 1, 0>), g, b, 0, st, gp, xp, yp, dxp, drp, sa, mean, (const float *)coef);
    else if (dres) hipLaunchKernelGGL((ksh_bn_bwd_apply_kernel<T, 0, 1>), g, b, 0, st, gp, xp, yp, dxp, drp, sa, mean, (const float *)coef);
    else hipLaunchKernelGGL((ksh_bn_bwd_apply_kernel<T, 0, 0>), g, b, 0, st, gp, xp, yp, dxp, drp, sa, mean, (const float *)coef);
}
}  // extern "C++"

int ksh_bn_bwd(const void *dy, const void *x, const void *y, void *dx, void *dres, int dtype, int64_t M, int C,
               const float *weight, const float *save_mean, const float *save_invstd, int relu, float *dweight,
               float *dbias, void *workspace, void *stream) {
    if (!tower_shape_ok("ksh_bn_bwd", dtype, M, C)) return KSH_E_ARG;
    if (!dy || !x || !dx || !save_mean || !save_invstd || !workspace || (relu && !y)) return fail_arg("ksh_bn_bwd: null pointer");
    if (!aligned16(dy) || !aligned16(x) || !aligned16(y) || !aligned16(dx) || !aligned16(dres))
        return fail_arg("ksh_bn_bwd: activations must be 16-byte aligned");
    float *partial = (float *)workspace, *coef = partial + (size_t)KSH_TW_MAXG * 2 * C;
    const KshRowSplit sp = ksh_row_split(M, C, KSH_TW_MAXG), sa = ksh_row_split(M, C, 4 * KSH_TW_MAXG);
    if (dtype == 0) launch_bn_bwd<float>(dy, x, y, dx, dres, sp, sa, weight, save_mean, save_invstd, relu, dweight, dbias, partial, coef, (hipStream_t)stream);
    else launch_bn_bwd<ksh_bf16>(dy, x, y, dx, dres, sp, sa, weight, save_mean, save_invstd, relu, dweight, dbias, partial, coef, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

static int se_args_ok(const char *who, int dtype, int N, int HW, int C) {
    if (N <= 0 || HW <= 0) { g_err = std::string(who) + ": need N > 0 and HW > 0"; return 0; }
    return tower_shape_ok(who, dtype, (int64_t)N * HW, C);
}

int ksh_se_pool(const void *x, int dtype, int N, int HW, int C, float *pooled, void *stream) {
    if (!se_args_ok("ksh_se_pool", dtype, N, HW, C)) return KSH_E_ARG;
    if (!x || !pooled || !aligned16(x)) return fail_arg("ksh_se_pool: null or misaligned pointer");
    if (dtype == 0) hipLaunchKernelGGL((ksh_se_pool_kernel<float>), dim3(N), dim3(KSH_TW_THREADS), 0, (hipStream_t)stream, (const float *)x, HW, C, pooled);
    else hipLaunchKernelGGL((ksh_se_pool_kernel<ksh_bf16>), dim3(N), dim3(KSH_TW_THREADS), 0, (hipStream_t)stream, (const ksh_bf16 *)x, HW, C, pooled);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

int ksh_se_apply(const void *x, const float *scale, const void *residual, void *z, int dtype, int N, int HW, int C,
                 void *stream) {
    if (!se_args_ok("ksh_se_apply", dtype, N, HW, C)) return KSH_E_ARG;
    if (!x || !scale || !residual || !z || !aligned16(x) || !aligned16(residual) || !aligned16(z))
        return fail_arg("ksh_se_apply: null or misaligned pointer");
    if (dtype == 0) hipLaunchKernelGGL((ksh_se_apply_kernel<float>), dim3(N), dim3(KSH_TW_THREADS), 0, (hipStream_t)stream, (const float *)x, scale, (const float *)residual, (float *)z, HW, C);
    else hipLaunchKernelGGL((ksh_se_apply_kernel<ksh_bf16>), dim3(N), dim3(KSH_TW_THREADS), 0, (hipStream_t)stream, (const ksh_bf16 *)x, scale, (const ksh_bf16 *)residual, (ksh_bf16 *)z, HW, C);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

int ksh_se_bwd_reduce(const void *dy, const void *z, const void *x, int dtype, int N, int HW, int C, float *dscale,
                      void *stream) {
    if (!se_args_ok("ksh_se_bwd_reduce", dtype, N, HW, C)) return KSH_E_ARG;
    if (!dy || !z || !x || !dscale || !aligned16(dy) || !aligned16(z) || !aligned16(x))
        return fail_arg("ksh_se_bwd_reduce: null or misaligned pointer");
    if (dtype == 0) hipLaunchKernelGGL((ksh_se_bwd_reduce_kernel<float>), dim3(N), dim3(KSH_TW_THREADS), 0, (hipStream_t)stream, (const float *)dy, (const float *)z, (const float *)x, HW, C, dscale);
    else hipLaunchKernelGGL((ksh_se_bwd_reduce_kernel<ksh_bf16>), dim3(N), dim3(KSH_TW_THREADS), 0, (hipStream_t)stream, (const ksh_bf16 *)dy, (const ksh_bf16 *)z, (const ksh_bf16 *)x, HW, C, dscale);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

int ksh_se_bwd_apply(const void *dy, const void *z, const float *scale, const float *dpooled, void *dx, void *dresidual,
                     int dtype, int N, int HW, int C, void *stream) {
    if (!se_args_ok("ksh_se_bwd_apply", dtype, N, HW, C)) return KSH_E_ARG;
    if (!dy || !z || !scale || !dpooled || !dx || !dresidual || !aligned16(dy) || !aligned16(z) || !aligned16(dx) ||
        !aligned16(dresidual))
        return fail_arg("ksh_se_bwd_apply: null or misaligned pointer");
    if (dtype == 0) hipLaunchKernelGGL((ksh_se_bwd_apply_kernel<float>), dim3(N), dim3(KSH_TW_THREADS), 0, (hipStream_t)stream, (const float *)dy, (const float *)z, scale, dpooled, (float *)dx, (float *)dresidual, HW, C);
    else hipLaunchKernelGGL((ksh_se_bwd_apply_kernel<ksh_bf16>), dim3(N), dim3(KSH_TW_THREADS), 0, (hipStream_t)stream, (const ksh_bf16 *)dy, (const ksh_bf16 *)z, scale, dpooled, (ksh_bf16 *)dx, (ksh_bf16 *)dresidual, HW, C);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

// bn2 + SE tail with merged passes (training mode).  Forward part 1: statistics -> finalize -> apply + pool.
int ksh_bn_pool_fwd_train(const void *x, void *y, int dtype, int N, int HW, int C, const float *pre_bias, const float *weight,
                          const float *bias, float *running_mean, float *running_var, float momentum, float eps,
                          float *save_mean, float *save_invstd, float *pooled, void *workspace, void *stream) {
    if (!se_args_ok("ksh_bn_pool_fwd_train", dtype, N, HW, C)) return KSH_E_ARG;
    if (!x || !y || !save_mean || !save_invstd || !pooled || !workspace) return fail_arg("ksh_bn_pool_fwd_train: null pointer");
    if (!aligned16(x) || !aligned16(y)) return fail_arg("ksh_bn_pool_fwd_train: activations must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = (int64_t)N * HW;
    float *partial = (float *)workspace, *coef = partial + (size_t)KSH_TW_MAXG * 2 * C;
    const KshRowSplit sp = ksh_row_split(M, C, KSH_TW_MAXG);
    if (dtype == 0) hipLaunchKernelGGL((ksh_bn_stats_kernel<float>), dim3(sp.G), dim3(KSH_TW_THREADS), 0, st, (const float *)x, sp, (const float *)running_mean, pre_bias, partial);
    else hipLaunchKernelGGL((ksh_bn_stats_kernel<ksh_bf16>), dim3(sp.G), dim3(KSH_TW_THREADS), 0, st, (const ksh_bf16 *)x, sp, (const float *)running_mean, pre_bias, partial);
    hipLaunchKernelGGL(ksh_bn_finalize_kernel, dim3((C + KSH_TW_FIN_CH - 1) / KSH_TW_FIN_CH), dim3(KSH_TW_THREADS), 0, st,
                       (const float *)partial, sp, (const float *)running_mean, pre_bias, weight, bias, running_mean, running_var,
                       momentum, eps, save_mean, save_invstd, coef);
    if (dtype == 0) hipLaunchKernelGGL((ksh_bn_apply_pool_kernel<float>), dim3(N), dim3(KSH_TW_THREADS), 0, st, (const float *)x, (float *)y, HW, C, (const float *)coef, pooled);
    else hipLaunchKernelGGL((ksh_bn_apply_pool_kernel<ksh_bf16>), dim3(N), dim3(KSH_TW_THREADS), 0, st, (const ksh_bf16 *)x, (ksh_bf16 *)y, HW, C, (const float *)coef, pooled);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

// Backward of the pair: SE backward apply (+ batch-norm partial sums) -> batch-norm finalize -> batch-norm apply.
// dy2: scratch of the activation's size for the gradient between the two (written, then read).
int ksh_se_bn_bwd(const void *dz, const void *z, const float *scale, const float *dpooled, const void *x, void *dy2, void *dx,
                  void *dresidual, int dtype, int N, int HW, int C, const float *weight, const float *save_mean,
                  const float *save_invstd, float *dweight, float *dbias, void *workspace, void *stream) {
    if (!se_args_ok("ksh_se_bn_bwd", dtype, N, HW, C)) return KSH_E_ARG;
    if (!dz || !z || !scale || !dpooled || !x || !dy2 || !dx || !dresidual || !save_mean || !save_invstd || !workspace)
        return fail_arg("ksh_se_bn_bwd: null pointer");
    if (!aligned16(dz) || !aligned16(z) || !aligned16(x) || !aligned16(dy2) || !aligned16(dx) || !aligned16(dresidual))
        return fail_arg("ksh_se_bn_bwd: activations must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = (int64_t)N * HW;
    float *partial = (float *)workspace, *coef = partial + (size_t)KSH_TW_MAXG * 2 * C;
    const int spb = (N + KSH_TW_MAXG - 1) / KSH_TW_MAXG;          // samples per block: at most KSH_TW_MAXG partial rows
    KshRowSplit sp;
    sp.M = M;
    sp.R = (long long)spb * HW;
    sp.C = C;
    sp.G = (N + spb - 1) / spb;
    const dim3 b(KSH_TW_THREADS);
    if (dtype == 0) hipLaunchKernelGGL((ksh_se_bwd_apply_bnred_kernel<float>), dim3(sp.G), b, 0, st, (const float *)dz, (const float *)z, scale, dpooled, (const float *)x, save_mean, (float *)dy2, (float *)dresidual, N, HW, C, spb, partial);
    else hipLaunchKernelGGL((ksh_se_bwd_apply_bnred_kernel<ksh_bf16>), dim3(sp.G), b, 0, st, (const ksh_bf16 *)dz, (const ksh_bf16 *)z, scale, dpooled, (const ksh_bf16 *)x, save_mean, (ksh_bf16 *)dy2, (ksh_bf16 *)dresidual, N, HW, C, spb, partial);
    hipLaunchKernelGGL(ksh_bn_bwd_finalize_kernel, dim3((C + KSH_TW_FIN_CH - 1) / KSH_TW_FIN_CH), b, 0, st, (const float *)partial, sp,
                       weight, save_invstd, dweight, dbias, coef);
    const KshRowSplit sa = ksh_row_split(M, C, 4 * KSH_TW_MAXG);
    if (dtype == 0) hipLaunchKernelGGL((ksh_bn_bwd_apply_kernel<float, 0, 0>), dim3(sa.G), b, 0, st, (const float *)dy2, (const float *)x, (const float *)nullptr, (float *)dx, (float *)nullptr, sa, save_mean, (const float *)coef);
    else hipLaunchKernelGGL((ksh_bn_bwd_apply_kernel<ksh_bf16, 0, 0>), dim3(sa.G), b, 0, st, (const ksh_bf16 *)dy2, (const ksh_bf16 *)x, (const ksh_bf16 *)nullptr, (ksh_bf16 *)dx, (ksh_bf16 *)nullptr, sa, save_mean, (const float *)coef);
    HIP_TRY(hipGetLastError());
    return KSH_OK;
}

// ---------------------------------------------------------------------------------- timing helpers
int ksh_event_create(void **ev) {
    if (!ev) return fail_arg("ksh_event_create: null");
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *ev = (void *)e;
    return KSH_OK;
}
int ksh_event_record(void *ev, void *stream) {
    HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return KSH_OK;
}
int ksh_event_elapsed_ms(void *start, void *stop, float *ms) {
    if (!ms) return fail_arg("ksh_event_elapsed_ms: null");
    HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return KSH_OK;
}
int ksh_event_destroy(void *ev) {
    HIP_TRY(hipEventDestroy((hipEvent_t)ev));
    return KSH_OK;
}

}  // extern "C"
