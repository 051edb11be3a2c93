/*
 * keisei_oracle.c -- TEST INFRASTRUCTURE ONLY (see keisei_oracle.h).
 *
 * Literal CPU restatement of the reference's Shogi path.  Deliberately written the way the
 * reference computes it (simulate every candidate on a copy, regenerate every enemy piece's
 * pseudo-moves to test for check) so that agreement with the HIP engine -- which uses a different
 * algorithm (pins / checkers / attack maps) -- is meaningful.  All "ref:" citations are relative to
 * /root/reference/keisei/.
 */
#include "keisei_oracle.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PC(type, color) ((int8_t)(1 + (type) + 14 * (color)))
#define PTYPE(pc) (((pc) - 1) % 14)
#define PCOLOR(pc) (((pc) - 1) / 14)

static int on_board(int r, int c) { return r >= 0 && r < 9 && c >= 0 && c < 9; }
static int is_promoted_type(int t) { return t >= KO_PRO_PAWN; }
/* ref: shogi/shogi_core_definitions.py:211-224 */
static int promote_type(int t) {
    switch (t) {
    case KO_PAWN: return KO_PRO_PAWN;
    case KO_LANCE: return KO_PRO_LANCE;
    case KO_KNIGHT: return KO_PRO_KNIGHT;
    case KO_SILVER: return KO_PRO_SILVER;
    case KO_BISHOP: return KO_PRO_BISHOP;
    case KO_ROOK: return KO_PRO_ROOK;
    default: return -1;
    }
}
static int base_type(int t) {
    switch (t) {
    case KO_PRO_PAWN: return KO_PAWN;
    case KO_PRO_LANCE: return KO_LANCE;
    case KO_PRO_KNIGHT: return KO_KNIGHT;
    case KO_PRO_SILVER: return KO_SILVER;
    case KO_PRO_BISHOP: return KO_BISHOP;
    case KO_PRO_ROOK: return KO_ROOK;
    default: return t;
    }
}

/* ------------------------------------------------------------------ action codec */
/* ref: utils/utils.py:208-266 -- enumeration order (r_from,c_from,r_to,c_to skipping from==to)x{F,T}
 * then (r_to,c_to)x{P..R}; closed form verified against the dumped table (tests/golden/codec.npz). */
int ko_encode_action(int from_sq, int to_sq, int aux) {
    if (from_sq < 0) return 12960 + to_sq * 7 + aux;
    return ((from_sq * 80 + to_sq - (to_sq > from_sq ? 1 : 0)) * 2) + (aux ? 1 : 0);
}
int ko_decode_action(int action, int *from_sq, int *to_sq, int *aux) {
    if (action < 0 || action >= KO_NUM_ACTIONS) return -1; /* ref: utils.py:300-308 IndexError */
    if (action >= 12960) {
        int k = action - 12960;
        *from_sq = -1; *to_sq = k / 7; *aux = k % 7;
        return 0;
    }
    int pair = action >> 1;
    int f = pair / 80, j = pair % 80;
    *from_sq = f; *to_sq = j + (j >= f ? 1 : 0); *aux = action & 1;
    return 0;
}

/* ------------------------------------------------------------------ lifetime */
KoGame *ko_new(int max_moves) {
    KoGame *g = (KoGame *)calloc(1, sizeof(KoGame));
    g->max_moves = max_moves;
    g->hist_cap = 64;
    g->hist = (uint8_t *)malloc((size_t)g->hist_cap * KO_STATE_BYTES);
    ko_reset(g);
    return g;
}
void ko_free(KoGame *g) {
    if (!g) return;
    free(g->hist);
    free(g);
}

/* ref: shogi/shogi_game.py:79-111 */
static void setup_initial_board(KoGame *g) {
    static const int back[9] = {KO_LANCE, KO_KNIGHT, KO_SILVER, KO_GOLD, KO_KING,
                                KO_GOLD, KO_SILVER, KO_KNIGHT, KO_LANCE};
    memset(g->board, 0, sizeof g->board);
    for (int i = 0; i < 9; i++) {
        g->board[2 * 9 + i] = PC(KO_PAWN, KO_WHITE);
        g->board[6 * 9 + i] = PC(KO_PAWN, KO_BLACK);
        g->board[0 * 9 + i] = PC(back[i], KO_WHITE);
        g->board[8 * 9 + i] = PC(back[i], KO_BLACK);
    }
    g->board[1 * 9 + 1] = PC(KO_ROOK, KO_WHITE);
    g->board[1 * 9 + 7] = PC(KO_BISHOP, KO_WHITE);
    g->board[7 * 9 + 1] = PC(KO_BISHOP, KO_BLACK);
    g->board[7 * 9 + 7] = PC(KO_ROOK, KO_BLACK);
}

/* ref: shogi/shogi_game.py:113-130 */
void ko_reset(KoGame *g) {
    setup_initial_board(g);
    memset(g->hands, 0, sizeof g->hands);
    g->side = KO_BLACK;
    g->move_count = 0;
    g->game_over = 0;
    g->winner = -1;
    g->reason = KO_ONGOING;
    g->hist_len = 0; /* move_history = [] ; the initial position lives only in board_history */
}

/* ------------------------------------------------------------------ pseudo-legal moves */
/* ref: shogi/shogi_rules_logic.py:82-208.  Returns destination squares (no duplicates possible:
 * the step sets and slide sets of +B / +R are disjoint, so the reference's set() is a no-op). */
static int piece_potential_moves(const KoGame *g, int pc, int r, int c, int *out) {
    int n = 0;
    int color = PCOLOR(pc), t = PTYPE(pc);
    int fw = (color == KO_BLACK) ? -1 : 1; /* :96-98 */
    int offs[8][2];
    int no = 0;
#define OFF(a, b) do { offs[no][0] = (a); offs[no][1] = (b); no++; } while (0)
    if (t == KO_PAWN) {
        OFF(fw, 0);
    } else if (t == KO_KNIGHT) {
        OFF(fw * 2, -1); OFF(fw * 2, 1);
    } else if (t == KO_SILVER) {
        OFF(fw, 0); OFF(fw, -1); OFF(fw, 1); OFF(-fw, -1); OFF(-fw, 1);
    } else if (t == KO_GOLD || t == KO_PRO_PAWN || t == KO_PRO_LANCE || t == KO_PRO_KNIGHT ||
               t == KO_PRO_SILVER) {
        OFF(fw, 0); OFF(fw, -1); OFF(fw, 1); OFF(0, -1); OFF(0, 1); OFF(-fw, 0);
    } else if (t == KO_KING) {
        OFF(-1, -1); OFF(-1, 0); OFF(-1, 1); OFF(0, -1); OFF(0, 1); OFF(1, -1); OFF(1, 0); OFF(1, 1);
    } else if (t == KO_PRO_BISHOP) { /* :176-182 */
        OFF(-1, 0); OFF(1, 0); OFF(0, -1); OFF(0, 1);
    } else if (t == KO_PRO_ROOK) { /* :186-192 */
        OFF(-1, -1); OFF(-1, 1); OFF(1, -1); OFF(1, 1);
    }
#undef OFF
    for (int i = 0; i < no; i++) {
        int nr = r + offs[i][0], nc = c + offs[i][1];
        if (!on_board(nr, nc)) continue;
        int tp = g->board[nr * 9 + nc];
        if (tp == 0 || PCOLOR(tp) != color) out[n++] = nr * 9 + nc;
    }
    int dirs[4][2];
    int nd = 0;
    if (t == KO_LANCE) {
        dirs[0][0] = fw; dirs[0][1] = 0; nd = 1;
    } else if (t == KO_BISHOP || t == KO_PRO_BISHOP) {
        static const int d[4][2] = {{-1, -1}, {-1, 1}, {1, -1}, {1, 1}};
        memcpy(dirs, d, sizeof d); nd = 4;
    } else if (t == KO_ROOK || t == KO_PRO_ROOK) {
        static const int d[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}};
        memcpy(dirs, d, sizeof d); nd = 4;
    }
    for (int k = 0; k < nd; k++) { /* :194-206 */
        for (int i = 1; i < 9; i++) {
            int nr = r + dirs[k][0] * i, nc = c + dirs[k][1] * i;
            if (!on_board(nr, nc)) break;
            int tp = g->board[nr * 9 + nc];
            if (tp == 0) {
                out[n++] = nr * 9 + nc;
            } else {
                if (PCOLOR(tp) != color) out[n++] = nr * 9 + nc;
                break;
            }
        }
    }
    return n;
}

/* ref: shogi/shogi_rules_logic.py:25-32 (first king in row-major order) */
static int find_king(const KoGame *g, int color) {
    for (int s = 0; s < 81; s++)
        if (g->board[s] == PC(KO_KING, color)) return s;
    return -1;
}

/* ref: shogi/shogi_rules_logic.py:234-272 */
static int square_attacked(const KoGame *g, int sq, int attacker_color) {
    int tmp[32];
    for (int s = 0; s < 81; s++) {
        int pc = g->board[s];
        if (pc == 0 || PCOLOR(pc) != attacker_color) continue;
        int n = piece_potential_moves(g, pc, s / 9, s % 9, tmp);
        for (int i = 0; i < n; i++)
            if (tmp[i] == sq) return 1;
    }
    return 0;
}

/* ref: shogi/shogi_rules_logic.py:36-67 and :362-379 (king missing => "in check") */
int ko_is_in_check(const KoGame *g, int color) {
    int k = find_king(g, color);
    if (k < 0) return 1;
    return square_attacked(g, k, 1 - color);
}

/* ref: shogi/shogi_game.py:821-825 */
static int in_promo_zone(int row, int color) { return color == KO_BLACK ? (row <= 2) : (row >= 6); }
/* ref: shogi/shogi_rules_logic.py:382-401 */
static int can_promote(int pc, int r_from, int r_to) {
    int t = PTYPE(pc);
    if (t == KO_GOLD || t == KO_KING || is_promoted_type(t)) return 0;
    if (promote_type(t) < 0) return 0;
    return in_promo_zone(r_from, PCOLOR(pc)) || in_promo_zone(r_to, PCOLOR(pc));
}
/* ref: shogi/shogi_rules_logic.py:404-421 */
static int must_promote(int pc, int r_to) {
    int t = PTYPE(pc), col = PCOLOR(pc);
    if (t == KO_PAWN || t == KO_LANCE) return (col == KO_BLACK && r_to == 0) || (col == KO_WHITE && r_to == 8);
    if (t == KO_KNIGHT) return (col == KO_BLACK && r_to <= 1) || (col == KO_WHITE && r_to >= 7);
    return 0;
}

/* ref: shogi/shogi_move_execution.py:24-138 (board / hands only) */
static void apply_to_board(KoGame *g, int from_sq, int to_sq, int aux, int mover) {
    if (from_sq < 0) { /* drop :55-71 */
        g->board[to_sq] = PC(aux, mover);
        g->hands[mover][aux] -= 1;
        return;
    }
    int pc = g->board[from_sq];
    int tp = g->board[to_sq];
    if (tp) { /* capture :102-114 : base type goes to the mover's hand (king included, quirk) */
        int bt = base_type(PTYPE(tp));
        g->hands[mover][bt] += 1;
    }
    g->board[to_sq] = (int8_t)pc;
    g->board[from_sq] = 0;
    if (aux) g->board[to_sq] = PC(promote_type(PTYPE(pc)), PCOLOR(pc)); /* :122-132 */
}

static int legal_moves_impl(KoGame *g, int is_ufz_check, uint16_t *ids);

/* ref: shogi/shogi_rules_logic.py:211-231 */
static int nifu(const KoGame *g, int color, int col) {
    for (int r = 0; r < 9; r++)
        if (g->board[r * 9 + col] == PC(KO_PAWN, color)) return 1;
    return 0;
}

/* ref: shogi/shogi_rules_logic.py:275-359 */
static int uchi_fu_zume(KoGame *g, int r, int c, int color) {
    int opp = 1 - color;
    int sq = r * 9 + c;
    int original_player = g->side;
    if (g->board[sq] != 0) return 0;
    if (g->hands[color][KO_PAWN] == 0) return 0;
    g->board[sq] = PC(KO_PAWN, color);
    g->hands[color][KO_PAWN] -= 1;
    int result = 0;
    int ok = find_king(g, opp);
    if (ok >= 0 && square_attacked(g, ok, color)) {
        uint16_t tmp[KO_MAX_LEGAL];
        g->side = opp;
        int n = legal_moves_impl(g, 1, tmp);
        result = (n == 0);
    }
    g->board[sq] = 0;
    g->hands[color][KO_PAWN] += 1;
    g->side = original_player;
    return result;
}

/* ref: shogi/shogi_rules_logic.py:424-483 */
static int can_drop(KoGame *g, int t, int r, int c, int color, int is_escape_check) {
    if (g->board[r * 9 + c] != 0) return 0;
    int last_rank = color == KO_BLACK ? 0 : 8;
    int second_last = color == KO_BLACK ? 1 : 7;
    if (t == KO_PAWN) {
        if (nifu(g, color, c)) return 0;
        if (r == last_rank) return 0;
        if (!is_escape_check && uchi_fu_zume(g, r, c, color)) return 0;
    } else if (t == KO_LANCE) {
        if (r == last_rank) return 0;
    } else if (t == KO_KNIGHT) {
        if (r == last_rank || r == second_last) return 0;
    }
    return 1;
}

/* ref: shogi/shogi_rules_logic.py:486-635.  The simulation (shogi_game.py:634-648 + undo :728-763)
 * is a copy / apply / test / restore of board+hands, exactly as the reference deep-copies them. */
static int legal_moves_impl(KoGame *g, int is_ufz_check, uint16_t *ids) {
    int n = 0;
    int me = g->side;
    int dst[32];
    int8_t board_save[81];
    uint8_t hands_save[2][8];
    for (int s = 0; s < 81; s++) {
        int pc = g->board[s];
        if (pc == 0 || PCOLOR(pc) != me) continue;
        int r_from = s / 9;
        int nd = piece_potential_moves(g, pc, r_from, s % 9, dst);
        for (int i = 0; i < nd; i++) {
            int to = dst[i];
            int cp = can_promote(pc, r_from, to / 9);
            int mp = must_promote(pc, to / 9);
            for (int promo = 0; promo <= 1; promo++) {
                if (promo == 0 && mp) continue;       /* :514-519 */
                if (promo == 1 && !(cp || mp)) continue;
                memcpy(board_save, g->board, sizeof board_save);
                memcpy(hands_save, g->hands, sizeof hands_save);
                apply_to_board(g, s, to, promo, me);
                int safe = !ko_is_in_check(g, me);    /* :550-554 */
                memcpy(g->board, board_save, sizeof board_save);
                memcpy(g->hands, hands_save, sizeof hands_save);
                if (safe) ids[n++] = (uint16_t)ko_encode_action(s, to, promo);
            }
        }
    }
    for (int t = 0; t < 7; t++) { /* :562-583 (hands hold P..R; a captured king is never droppable/mappable) */
        if (g->hands[me][t] == 0) continue;
        for (int sq = 0; sq < 81; sq++) {
            if (!can_drop(g, t, sq / 9, sq % 9, me, is_ufz_check)) continue;
            memcpy(board_save, g->board, sizeof board_save);
            memcpy(hands_save, g->hands, sizeof hands_save);
            apply_to_board(g, -1, sq, t, me);
            int safe = !ko_is_in_check(g, me);        /* :616-622 */
            memcpy(g->board, board_save, sizeof board_save);
            memcpy(g->hands, hands_save, sizeof hands_save);
            if (safe) ids[n++] = (uint16_t)ko_encode_action(-1, sq, t);
        }
    }
    return n;
}

static int cmp_u16(const void *a, const void *b) {
    return (int)*(const uint16_t *)a - (int)*(const uint16_t *)b;
}

int ko_legal_ids(KoGame *g, uint16_t *ids) {
    int n = legal_moves_impl(g, 0, ids);
    qsort(ids, (size_t)n, sizeof(uint16_t), cmp_u16);
    return n;
}

int ko_legal_mask(KoGame *g, uint8_t *mask) {
    uint16_t ids[KO_MAX_LEGAL];
    int n = legal_moves_impl(g, 0, ids);
    memset(mask, 0, KO_NUM_ACTIONS);
    for (int i = 0; i < n; i++) mask[ids[i]] = 1; /* ref: utils/utils.py:323-327 */
    return n;
}

/* ------------------------------------------------------------------ state key / termination */
/* ref: shogi/shogi_game.py:347-372 -- (board grid, both hands' counts, side to move); exact */
static void state_key(const KoGame *g, uint8_t *out) {
    memcpy(out, g->board, 81);
    memcpy(out + 81, g->hands, 16);
    out[97] = (uint8_t)g->side;
}
void ko_export_state(const KoGame *g, uint8_t *out128) {
    memset(out128, 0, 128);
    memcpy(out128, g->board, 81);
    for (int c = 0; c < 2; c++)
        for (int t = 0; t < 7; t++) out128[81 + c * 7 + t] = g->hands[c][t];
    out128[95] = (uint8_t)g->side;
    out128[96] = (uint8_t)(g->move_count & 0xff);
    out128[97] = (uint8_t)((g->move_count >> 8) & 0xff);
    out128[98] = (uint8_t)g->game_over;
    out128[99] = (uint8_t)g->reason;
    out128[100] = (uint8_t)(g->winner + 1);
}

/* ref: shogi/shogi_rules_logic.py:638-695 */
static int sennichite(const KoGame *g) {
    if (g->hist_len == 0) return 0;
    const uint8_t *last = g->hist + (size_t)(g->hist_len - 1) * KO_STATE_BYTES;
    int count = 0;
    for (int i = 0; i < g->hist_len; i++)
        if (memcmp(g->hist + (size_t)i * KO_STATE_BYTES, last, KO_STATE_BYTES) == 0) count++;
    return count >= 4;
}

/* ref: shogi/shogi_game.py:408-450 */
static void check_termination(KoGame *g, int player_who_just_moved) {
    if (g->game_over) return;
    uint16_t ids[KO_MAX_LEGAL];
    int n = legal_moves_impl(g, 0, ids);
    if (n == 0) {
        g->game_over = 1;
        if (ko_is_in_check(g, g->side)) {
            g->winner = player_who_just_moved;
            g->reason = KO_TSUMI;
        } else {
            g->winner = -1;
            g->reason = KO_STALEMATE;
        }
        return;
    }
    if (g->move_count >= g->max_moves) {
        g->game_over = 1; g->winner = -1; g->reason = KO_MAX_MOVES;
        return;
    }
    if (sennichite(g)) {
        g->game_over = 1; g->winner = -1; g->reason = KO_SENNICHITE;
    }
}

/* ref: shogi/shogi_game.py:574-660 (real move) + :518-546 validation + :553-572 return */
int ko_make_move(KoGame *g, int action, float *reward, int *done) {
    if (g->game_over) { /* :589-593: nominal mover = opponent of current player */
        int nominal = 1 - g->side;
        *reward = (g->winner == nominal) ? 1.0f : (g->winner < 0 ? 0.0f : -1.0f);
        *done = 1;
        return 0;
    }
    int from, to, aux;
    if (ko_decode_action(action, &from, &to, &aux) != 0) return -1;
    int mover = g->side;
    if (from >= 0) {
        int pc = g->board[from];
        if (pc == 0) return -1;                 /* :529-530 */
        if (PCOLOR(pc) != mover) return -1;     /* :531-534 */
        int dst[32], ok = 0;
        int nd = piece_potential_moves(g, pc, from / 9, from % 9, dst);
        for (int i = 0; i < nd; i++) ok |= (dst[i] == to);
        if (!ok) return -1;                     /* :539-544 */
        if (aux && promote_type(PTYPE(pc)) < 0) return -1; /* shogi_move_execution.py:122-127 */
    } else {
        /* drops are not validated by make_move (shogi_move_execution.py:62-71); a drop the mover
         * cannot make would corrupt the reference's state, so reject it instead. */
        if (g->hands[mover][aux] == 0 || g->board[to] != 0) return -1;
    }
    apply_to_board(g, from, to, aux, mover);
    g->move_count += 1;                          /* shogi_move_execution.py:153-156 */
    g->side = 1 - g->side;
    if (g->hist_len == g->hist_cap) {
        g->hist_cap *= 2;
        g->hist = (uint8_t *)realloc(g->hist, (size_t)g->hist_cap * KO_STATE_BYTES);
    }
    state_key(g, g->hist + (size_t)g->hist_len * KO_STATE_BYTES); /* :651-654 (after the flip) */
    g->hist_len += 1;
    check_termination(g, mover);
    *reward = 0.0f;
    if (g->game_over) *reward = (g->winner == mover) ? 1.0f : (g->winner < 0 ? 0.0f : -1.0f);
    *done = g->game_over;
    return 0;
}

/* ------------------------------------------------------------------ observation */
/* ref: shogi/shogi_game_io.py:434-539 */
void ko_observation(const KoGame *g, float *obs) {
    memset(obs, 0, sizeof(float) * KO_OBS_SIZE);
    int black_pov = (g->side == KO_BLACK);
    for (int r = 0; r < 9; r++)
        for (int c = 0; c < 9; c++) {
            int pc = g->board[r * 9 + c];
            if (!pc) continue;
            int fr = black_pov ? r : 8 - r, fc = black_pov ? c : 8 - c; /* :468-469 */
            int t = PTYPE(pc), mine = (PCOLOR(pc) == g->side);
            int ch;
            if (is_promoted_type(t)) {
                /* OBS_PROMOTED_ORDER = +P,+L,+N,+S,+B,+R -> index t-8 */
                ch = (mine ? 8 : 22) + (t - KO_PRO_PAWN);
            } else {
                ch = (mine ? 0 : 14) + t; /* OBS_UNPROMOTED_ORDER = P,L,N,S,G,B,R,K */
            }
            obs[ch * 81 + fr * 9 + fc] = 1.0f;
        }
    int opp = 1 - g->side;
    for (int i = 0; i < 7; i++) {
        int n = g->hands[g->side][i];
        if (n > 0) { /* :511-515, Python double division then stored to float32 */
            float v = (float)((double)n / 18.0);
            for (int k = 0; k < 81; k++) obs[(28 + i) * 81 + k] = v;
        }
        n = g->hands[opp][i];
        if (n > 0) {
            float v = (float)((double)n / 18.0);
            for (int k = 0; k < 81; k++) obs[(35 + i) * 81 + k] = v;
        }
    }
    float ind = black_pov ? 1.0f : 0.0f; /* :530-532 */
    double mm = (double)g->max_moves;
    float mc = (float)(mm > 0 ? (double)g->move_count / mm : 0.0); /* :535-536 */
    for (int k = 0; k < 81; k++) {
        obs[42 * 81 + k] = ind;
        obs[43 * 81 + k] = mc;
    }
}

/* ------------------------------------------------------------------ SFEN (test fixtures) */
/* ref: shogi/shogi_game_io.py:169-306 and shogi_game.py:283-345 */
static int sym_type(char up) {
    switch (up) {
    case 'P': return KO_PAWN; case 'L': return KO_LANCE; case 'N': return KO_KNIGHT;
    case 'S': return KO_SILVER; case 'G': return KO_GOLD; case 'B': return KO_BISHOP;
    case 'R': return KO_ROOK; case 'K': return KO_KING; default: return -1;
    }
}
int ko_from_sfen(KoGame *g, const char *sfen) {
    char buf[256];
    char *tok[4];
    size_t len = strlen(sfen);
    if (len >= sizeof buf) return -1;
    strcpy(buf, sfen);
    int nt = 0;
    for (char *p = strtok(buf, " \t"); p && nt < 4; p = strtok(NULL, " \t")) tok[nt++] = p;
    if (nt != 4) return -1;
    if (strtok(NULL, " \t")) return -1;
    if (!(tok[1][0] == 'b' || tok[1][0] == 'w') || tok[1][1]) return -1;
    for (const char *p = tok[3]; *p; p++) if (!isdigit((unsigned char)*p)) return -1;
    long mv = strtol(tok[3], NULL, 10);
    if (mv < 1) return -1;
    int8_t board[81];
    uint8_t hands[2][8];
    memset(board, 0, sizeof board);
    memset(hands, 0, sizeof hands);
    int r = 0, c = 0, promo = 0;
    for (const char *p = tok[0];; p++) {
        char ch = *p;
        if (ch == '/' || ch == 0) {
            if (c != 9 || promo) return -1;
            r++; c = 0;
            if (ch == 0) break;
            if (r >= 9) return -1;
            continue;
        }
        if (r >= 9) return -1;
        if (ch == '+') { if (promo) return -1; promo = 1; continue; }
        if (isdigit((unsigned char)ch)) {
            if (promo || ch == '0') return -1;
            int e = ch - '0';
            if (e > 9 - c) return -1;
            c += e;
            continue;
        }
        int t = sym_type((char)toupper((unsigned char)ch));
        if (t < 0 || c >= 9) return -1;
        int color = isupper((unsigned char)ch) ? KO_BLACK : KO_WHITE;
        if (promo) { t = promote_type(t); if (t < 0) return -1; }
        board[r * 9 + c] = PC(t, color);
        c++; promo = 0;
    }
    if (r != 9) return -1;
    if (strcmp(tok[2], "-") != 0) {
        int white_seen = 0;
        const char *p = tok[2];
        while (*p) {
            int cnt = 0, have = 0;
            while (isdigit((unsigned char)*p)) { cnt = cnt * 10 + (*p - '0'); have = 1; p++; }
            if (!*p) return -1;
            int t = sym_type((char)toupper((unsigned char)*p));
            if (t < 0 || t == KO_KING) return -1;
            int white = islower((unsigned char)*p) != 0;
            if (white) white_seen = 1; else if (white_seen) return -1;
            if (!have) cnt = 1;
            if (hands[white][t] + cnt > 255) return -1;
            hands[white][t] = (uint8_t)(hands[white][t] + cnt);
            p++;
        }
    }
    memcpy(g->board, board, sizeof board);
    memcpy(g->hands, hands, sizeof hands);
    g->side = tok[1][0] == 'b' ? KO_BLACK : KO_WHITE;
    g->move_count = (int32_t)(mv - 1);
    g->game_over = 0; g->winner = -1; g->reason = KO_ONGOING;
    g->hist_len = 0;
    check_termination(g, 1 - g->side); /* shogi_game.py:343 */
    return 0;
}

/* ------------------------------------------------------------------ batched step (CPU baseline) */
/* ref: training/step_manager.py:98-348 (one env step) and :437-468 (reset on done, the terminal
 * next_obs is discarded and replaced by the reset observation). */
void ko_vec_step(KoGame **games, int n, const int64_t *actions, float *obs, uint8_t *mask,
                 int32_t *n_legal, float *reward, uint8_t *done, uint8_t *reason, int8_t *winner,
                 int auto_reset) {
    for (int i = 0; i < n; i++) {
        KoGame *g = games[i];
        float rw = 0.0f;
        int dn = 0;
        int rc = ko_make_move(g, (int)actions[i], &rw, &dn);
        if (rc != 0) { rw = 0.0f; dn = 0; }
        reward[i] = rw;
        done[i] = (uint8_t)dn;
        reason[i] = (uint8_t)g->reason;
        winner[i] = (int8_t)g->winner;
        if (dn && auto_reset) ko_reset(g);
        ko_observation(g, obs + (size_t)i * KO_OBS_SIZE);
        n_legal[i] = ko_legal_mask(g, mask + (size_t)i * KO_NUM_ACTIONS);
    }
}

/* Uniform-random legal self-play of n games for `steps` batched steps, entirely in C (no interpreter between steps):
 * the CPU-baseline loop of bench.py when several host threads run at once.  Each step is ko_vec_step's work (move,
 * termination, auto-reset, 46-plane observation, 13 527-byte mask) with the action drawn from the previous mask by a
 * xorshift64* stream.  obs / mask: caller-provided scratch of n rows.  Returns the number of env steps made. */
long long ko_selfplay_random(KoGame **games, int n, int steps, uint64_t seed, float *obs, uint8_t *mask) {
    uint64_t x = seed ? seed : 0x9E3779B97F4A7C15ull;
    long long made = 0;
    int32_t *nl = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    if (!nl) return 0;
    for (int i = 0; i < n; i++) nl[i] = ko_legal_mask(games[i], mask + (size_t)i * KO_NUM_ACTIONS);
    for (int s = 0; s < steps; s++) {
        for (int i = 0; i < n; i++) {
            KoGame *g = games[i];
            uint8_t *m = mask + (size_t)i * KO_NUM_ACTIONS;
            if (nl[i] <= 0) { ko_reset(g); nl[i] = ko_legal_mask(g, m); }
            x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
            int k = (int)(((x * 0x2545F4914F6CDD1Dull) >> 33) % (uint64_t)nl[i]);
            int a = 0;
            for (; a < KO_NUM_ACTIONS; a++)
                if (m[a] && k-- == 0) break;
            float rw = 0.0f;
            int dn = 0;
            (void)ko_make_move(g, a, &rw, &dn);
            if (dn) ko_reset(g);
            ko_observation(g, obs + (size_t)i * KO_OBS_SIZE);
            nl[i] = ko_legal_mask(g, m);
            made++;
        }
    }
    free(nl);
    return made;
}

/* ------------------------------------------------------------------ GAE */
/* ref: core/experience_buffer.py:99-145.  torch evaluates, on 0-d float32 tensors:
 *   delta = r[t] + gamma * next_v * mask[t] - v[t]          (left to right, gamma is a Python float
 *                                                            -> scalar multiply in float32)
 *   gae   = delta + gamma * lambda * mask[t] * gae           (gamma*lambda folded in Python double,
 *                                                            then a float32 scalar)
 * Each binary op rounds to float32; no FMA contraction (compile with -ffp-contract=off). */
void ko_gae(const float *rewards, const float *values, const uint8_t *dones, const float *last_values,
            double gamma, double lam, int T, int N, float *adv, float *ret) {
    const float g = (float)gamma;              /* python float -> float32 scalar */
    const float gl = (float)(gamma * lam);     /* folded in Python double first */
    for (int n = 0; n < N; n++) {
        volatile float gae = 0.0f;
        float next_v = last_values[n];
        for (int t = T - 1; t >= 0; t--) {
            size_t i = (size_t)t * N + n;
            float m = 1.0f - (dones[i] ? 1.0f : 0.0f);
            volatile float a = g * next_v;
            volatile float b = a * m;
            volatile float c = rewards[i] + b;
            volatile float delta = c - values[i];
            volatile float d = gl * m;
            volatile float e = d * gae;
            gae = delta + e;
            adv[i] = gae;
            volatile float rr = gae + values[i];
            ret[i] = rr;
            next_v = values[i];
        }
    }
}
