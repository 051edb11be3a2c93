/*
 * keisei_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded, mailbox-board restatement of the reference's (tachyon-beep/shogidrl,
 * "Keisei") Shogi rules, observation/mask encoders, GAE scan and masked-categorical / PPO-loss
 * arithmetic.  It follows the reference algorithm literally (make-move / attack-test / undo per
 * candidate), so it is slow and obviously faithful; every function cites the reference file:line it
 * restates.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (shogidrl_amd/, libkeisei_hip.so) never links, imports or calls anything in oracle/.
 *
 * Parity pin: checked against golden vectors produced by importing the reference in the build
 * container (tests/golden/make_golden_*.py -> tests/golden/*.npz|json), see tests/test_oracle_*.py.
 */
#ifndef KEISEI_ORACLE_H
#define KEISEI_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KO_NUM_ACTIONS 13527      /* keisei/constants.py:15 FULL_ACTION_SPACE */
#define KO_OBS_PLANES 46          /* shogi_core_definitions.py:272-283 */
#define KO_OBS_SIZE (46 * 81)
#define KO_MAX_LEGAL 1024
#define KO_STATE_BYTES 98         /* 81 board + 16 hands + 1 side: exact sennichite key */

/* piece types: shogi_core_definitions.py:64-84 */
enum { KO_PAWN = 0, KO_LANCE, KO_KNIGHT, KO_SILVER, KO_GOLD, KO_BISHOP, KO_ROOK, KO_KING,
       KO_PRO_PAWN, KO_PRO_LANCE, KO_PRO_KNIGHT, KO_PRO_SILVER, KO_PRO_BISHOP, KO_PRO_ROOK };
enum { KO_BLACK = 0, KO_WHITE = 1 };
/* termination reasons: shogi_core_definitions.py:135-141 (only the four the engine can produce) */
enum { KO_ONGOING = 0, KO_TSUMI = 1, KO_STALEMATE = 2, KO_MAX_MOVES = 3, KO_SENNICHITE = 4 };

typedef struct KoGame {
    int8_t  board[81];      /* 0 empty, else 1 + type + 14*color; sq = row*9 + col               */
    uint8_t hands[2][8];    /* [color][P,L,N,S,G,B,R,(K: king-capture quirk, never droppable)]  */
    int32_t side;           /* current_player */
    int32_t move_count;
    int32_t max_moves;
    int32_t game_over;
    int32_t winner;         /* -1 none, 0 black, 1 white */
    int32_t reason;
    int32_t hist_len;       /* number of post-move states in move_history */
    int32_t hist_cap;
    uint8_t *hist;          /* hist_len * KO_STATE_BYTES, exact states (shogi_game.py:347-372) */
} KoGame;

KoGame *ko_new(int max_moves);
void    ko_free(KoGame *g);
void    ko_reset(KoGame *g);                                   /* shogi_game.py:113-130 */
int     ko_from_sfen(KoGame *g, const char *sfen);             /* shogi_game.py:283-345; 0 ok, <0 rejected */
int     ko_legal_ids(KoGame *g, uint16_t *ids);                /* shogi_rules_logic.py:486-635 + utils.py:310-336; sorted */
int     ko_is_in_check(const KoGame *g, int color);            /* shogi_rules_logic.py:36-67 */
/* shogi_game.py:574-660.  Returns 0 ok, -1 malformed/illegal pattern (the reference's ValueError). */
int     ko_make_move(KoGame *g, int action, float *reward, int *done);
void    ko_observation(const KoGame *g, float *obs);           /* shogi_game_io.py:434-539 */
int     ko_legal_mask(KoGame *g, uint8_t *mask);               /* utils.py:310-336; returns n_legal */
void    ko_export_state(const KoGame *g, uint8_t *out128);     /* packed like the HIP engine's state record */

/* action codec, utils.py:183-266 in closed form.  from = -1 for drops (aux = piece type), else aux = promote */
int     ko_encode_action(int from_sq, int to_sq, int aux);
int     ko_decode_action(int action, int *from_sq, int *to_sq, int *aux);

/* Batched helper used as the CPU baseline: n independent games, one env step each with auto-reset
 * (step_manager.py:98-348, 437-468).  Outputs have the same layout as ksh_env_step. */
long long ko_selfplay_random(KoGame **games, int n, int steps, uint64_t seed, float *obs, uint8_t *mask);
void    ko_vec_step(KoGame **games, int n, const int64_t *actions, float *obs, uint8_t *mask,
                    int32_t *n_legal, float *reward, uint8_t *done, uint8_t *reason, int8_t *winner,
                    int auto_reset);

/* experience_buffer.py:99-145 on a [T,N] row-major layout (N independent streams). */
void    ko_gae(const float *rewards, const float *values, const uint8_t *dones, const float *last_values,
               double gamma, double lam, int T, int N, float *adv, float *ret);

#ifdef __cplusplus
}
#endif
#endif
