// TEST INFRASTRUCTURE: runs the learner-side kernels of shogidrl_amd/csrc/ksh_ppo_kernels.h on the host.
#include <hip/hip_runtime.h>

#include <vector>

#include "../../shogidrl_amd/csrc/ksh_ppo_kernels.h"

namespace {
struct GaeA { const float *r, *v; const uint8_t *d; const float *lv; float g, gl; int T, N; float *adv, *ret; } ga;
void gae_entry(void *) { ksh_gae_kernel(ga.r, ga.v, ga.d, ga.lv, ga.g, ga.gl, ga.T, ga.N, ga.adv, ga.ret); }

struct SampA { const void *logits; int dtype; const uint8_t *mask; const uint32_t *bits; int B; unsigned long long seed, step; int det; long long *act; float *lp; } sa;
void samp_entry(void *) {
    if (sa.dtype == 0) ksh_masked_sample_kernel<float>((const float *)sa.logits, sa.mask, sa.bits, sa.B, sa.seed, sa.step, nullptr, sa.det, sa.act, sa.lp);
    else ksh_masked_sample_kernel<ksh_bf16>((const ksh_bf16 *)sa.logits, sa.mask, sa.bits, sa.B, sa.seed, sa.step, nullptr, sa.det, sa.act, sa.lp);
}
KshLossArgs la;
float *g_row_logp = nullptr, *g_row_entropy = nullptr;   // set by emu_evaluate_actions around one emu_ppo_loss call
const void *l_logits; void *l_dlogits; int l_dtype;
void loss_entry(void *) {
    if (l_dtype == 0) ksh_ppo_loss_kernel<float>(la, (const float *)l_logits, (float *)l_dlogits);
    else ksh_ppo_loss_kernel<ksh_bf16>(la, (const ksh_bf16 *)l_logits, (ksh_bf16 *)l_dlogits);
}
void sparse_entry(void *) {
    if (l_dtype == 0) ksh_ppo_loss_sparse_kernel<float>(la, (const float *)l_logits, (float *)l_dlogits);
    else ksh_ppo_loss_sparse_kernel<ksh_bf16>(la, (const ksh_bf16 *)l_logits, (ksh_bf16 *)l_dlogits);
}
void stats_entry(void *) { ksh_ppo_stats_kernel(la); }
}  // namespace

extern "C" {
// the sampler's uniform mapping and the Gumbel key it feeds, for a given 32-bit Philox output
float emu_uniform_open(unsigned int x) { return ksh_uniform_open(x); }
float emu_gumbel_key(float z, unsigned int x) { return z - logf(-logf(ksh_uniform_open(x))); }
void emu_gae(const float *r, const float *v, const uint8_t *d, const float *lv, double gamma, double lam, int T, int N,
             float *adv, float *ret) {
    ga = GaeA{r, v, d, lv, (float)gamma, (float)(gamma * lam), T, N, adv, ret};
    for (int b = 0; b < (N + 255) / 256; b++) emu::run_block(gae_entry, nullptr, b, 256);
}
void emu_masked_sample(const void *logits, int dtype, const uint8_t *mask, const uint32_t *bits, int B, unsigned long long seed,
                       unsigned long long step, int det, long long *action, float *logp) {
    sa = SampA{logits, dtype, mask, bits, B, seed, step, det, action, logp};
    for (int b = 0; b < (B + KSH_SAMPLE_WPB - 1) / KSH_SAMPLE_WPB; b++) emu::run_block(samp_entry, nullptr, b, 64 * KSH_SAMPLE_WPB);
}
void emu_ppo_loss(const void *logits, int dtype, const uint8_t *mask, const uint32_t *bits, const long long *actions, const float *old_logp,
                  const float *adv, const float *ret, const float *values, const float *old_values, int B, float clip_eps,
                  float vf, float ent, int value_clip, void *dlogits, float *dvalues, float *stats, float *workspace) {
    la.mask = mask; la.mask_bits = bits; la.actions = actions; la.old_logp = old_logp; la.adv = adv; la.ret = ret; la.values = values;
    la.old_values = old_values; la.B = B; la.clip_eps = clip_eps; la.vf_coef = vf; la.ent_coef = ent;
    la.value_clip = value_clip; la.dvalues = dvalues; la.stats = stats; la.partial = workspace;
    l_logits = logits; l_dlogits = dlogits; l_dtype = dtype;
    la.only_overflow = 1;
    la.row_logp = g_row_logp; la.row_entropy = g_row_entropy;
    for (int b = 0; b < (B + KSH_LOSS_WPB - 1) / KSH_LOSS_WPB; b++) emu::run_block(sparse_entry, nullptr, b, 64 * KSH_LOSS_WPB);
    for (int b = 0; b < B; b++) emu::run_block(loss_entry, nullptr, b, KSH_ROW_THREADS);
    emu::run_block(stats_entry, nullptr, 0, 256);
}
// ksh_evaluate_actions: the loss kernels with the gradient switched off (dlogits == nullptr), exactly as ksh_lib.hip calls them
void emu_evaluate_actions(const void *logits, int dtype, const uint8_t *mask, const uint32_t *bits, const long long *actions, int B,
                          float *logp, float *entropy) {
    std::vector<float> zeros(B, 0.0f), dv(B), ws((size_t)B * 8 + 16);
    float st[8];
    g_row_logp = logp; g_row_entropy = entropy;
    emu_ppo_loss(logits, dtype, mask, bits, actions, zeros.data(), zeros.data(), zeros.data(), zeros.data(), nullptr, B, 0.2f, 0.0f,
                 0.0f, 0, nullptr, dv.data(), st, ws.data());
    g_row_logp = nullptr; g_row_entropy = nullptr;
}
}

#include "emu_head.inc"
