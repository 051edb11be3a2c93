// TEST INFRASTRUCTURE: runs the env kernel of shogidrl_amd/csrc/ksh_env_kernel.h on the host through
// the fiber emulator, one block after another, on ordinary host buffers.
#include <hip/hip_runtime.h>

#include "../../shogidrl_amd/csrc/ksh_env_kernel.h"

static KshStepArgs g_args;
static void entry(void *) { ksh_env_step_kernel(g_args); }
static void entry_c16(void *) { ksh_env_step_c16_kernel(g_args); }

extern "C" {
int emu_tables_size() { return (int)sizeof(KshTables); }
void emu_build_tables(void *t, int max_moves) {
    ksh_build_tables((KshTables *)t);
    ksh_finish_record(((KshTables *)t)->init_state, max_moves);
}
void emu_finish_record(uint8_t *rec, int max_moves) { ksh_finish_record(rec, max_moves); }
unsigned long long emu_full_hash(const uint8_t *rec) { return ksh_full_hash(rec); }
void emu_env_launch(uint8_t *state, unsigned long long *hist, const void *tables, const long long *actions,
                    const uint8_t *prev_mask, const uint32_t *prev_bits, const uint8_t *reset_mask, float *obs, uint8_t *mask,
                    uint32_t *mask_bits, int *n_legal,
                    float *reward, uint8_t *done, uint8_t *reason, signed char *winner, int n_games, int max_moves,
                    int hist_cap, int mode, int auto_reset, int obs_format) {
    KshStepArgs a;
    a.state = state; a.hist = hist; a.tables = (const KshTables *)tables; a.actions = actions;
    a.prev_mask = prev_mask; a.prev_bits = prev_bits; a.reset_mask = reset_mask; a.obs = obs; a.mask = mask;
    a.mask_bits = mask_bits; a.n_legal = n_legal;
    a.reward = reward; a.done = done; a.reason = reason; a.winner = winner; a.n_games = n_games;
    a.max_moves = max_moves; a.hist_cap = hist_cap; a.mode = mode; a.auto_reset = auto_reset; a.stamps = nullptr;
    g_args = a;
    const int nblocks = (n_games + KSH_WPB - 1) / KSH_WPB;
    for (int b = 0; b < nblocks; b++) emu::run_block(obs_format ? entry_c16 : entry, nullptr, b, 64 * KSH_WPB);
}
}
