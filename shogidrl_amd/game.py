"""ShogiGame -- single-game facade with the reference's method surface for this path, running on the HIP
env kernel (a VecShogiEnv of one game).  It exists so that Keisei's EnvManager / StepManager style callers
(and evaluation loops) can drive the GPU engine through the familiar calls:

    ShogiGame(max_moves_per_game) / seed / reset / get_observation / get_legal_moves / make_move /
    current_player / move_count / game_over / winner / termination_reason / from_sfen / to_sfen_string /
    to_string / move_history, and game_to_kif(game, filename, sente_player_name, gote_player_name)

(keisei/shogi/shogi_game.py:39-54, 56-73, 113-130, 183-218, 283-345, 574-660; SFEN format
keisei/shogi/shogi_game_io.py:169-379, text board :540-583, KIF export :586-733).  Colours / piece types are plain ints (0 = Black, 1 = White;
0..13 = P,L,N,S,G,B,R,K,+P,+L,+N,+S,+B,+R); moves are the reference's 5-tuples.  Every call synchronises with
the device -- this is the plumbing path (BASELINE configs[0]), the batched path is VecShogiEnv."""
import datetime
import re

import numpy as np
import torch

from .env import REASON_NAMES, VecShogiEnv
from .mapper import PolicyOutputMapper

_SYM = "PLNSGBRK"
_HAND_ORDER = (6, 5, 4, 3, 2, 1, 0)        # R,B,G,S,N,L,P: SFEN canonical hand order (shogi_game_io.py SFEN_HAND_PIECE_CANONICAL_ORDER)
_PROMOTE = {0: 8, 1: 9, 2: 10, 3: 11, 5: 12, 6: 13}
_SFEN_RE = re.compile(r"^\s*([^ ]+)\s+([bw])\s+([^ ]+)\s+(\d+)\s*$")
_HAND_RE = re.compile(r"(\d*)([PLNSGBRplnsgbr])")


def parse_sfen(sfen):
    """SFEN string -> packed 104-byte state record (include/keisei_hip.h). Raises ValueError like the reference."""
    m = _SFEN_RE.match(sfen.strip())
    if not m:
        raise ValueError("Invalid SFEN string structure: %r" % sfen)
    board_s, turn, hands_s, num = m.groups()
    move_number = int(num)
    if move_number < 1:
        raise ValueError("SFEN move number must be positive")
    rec = np.zeros(128, np.uint8)
    rows = board_s.split("/")
    if len(rows) != 9:
        raise ValueError("Expected 9 ranks")
    for r, row in enumerate(rows):
        c, promo = 0, False
        for ch in row:
            if ch == "+":
                if promo:
                    raise ValueError("Invalid piece character sequence starting with '+'")
                promo = True
            elif ch.isdigit():
                if promo or ch == "0" or int(ch) > 9 - c:
                    raise ValueError("Invalid SFEN board row %r" % row)
                c += int(ch)
            else:
                t = _SYM.find(ch.upper())
                if t < 0 or c >= 9:
                    raise ValueError("Invalid SFEN piece character for board: %s" % ch)
                if promo:
                    if t not in _PROMOTE:
                        raise ValueError("Invalid promotion: '+' applied to non-promotable piece")
                    t = _PROMOTE[t]
                rec[r * 9 + c] = 1 + t + (0 if ch.isupper() else 14)
                c += 1
                promo = False
        if c != 9 or promo:
            raise ValueError("Row %d (%r) does not describe 9 columns" % (r + 1, row))
    if hands_s != "-":
        pos, white_seen = 0, False
        while pos < len(hands_s):
            hm = _HAND_RE.match(hands_s, pos)
            if not hm:
                raise ValueError("Invalid character sequence in SFEN hands")
            cnt = int(hm.group(1)) if hm.group(1) else 1
            ch = hm.group(2)
            white = ch.islower()
            if white:
                white_seen = True
            elif white_seen:
                raise ValueError("Invalid SFEN hands: Black's pieces must precede White's pieces.")
            slot = 81 + (7 if white else 0) + _SYM.find(ch.upper())
            if rec[slot] + cnt > 255:
                raise ValueError("hand count too large")
            rec[slot] += cnt
            pos = hm.end()
    rec[95] = 0 if turn == "b" else 1
    mc = move_number - 1
    rec[96], rec[97] = mc & 0xFF, (mc >> 8) & 0xFF
    return rec


def state_to_sfen(rec):
    out = []
    for r in range(9):
        row, empty = "", 0
        for c in range(9):
            pc = int(rec[r * 9 + c])
            if pc == 0:
                empty += 1
                continue
            if empty:
                row += str(empty)
                empty = 0
            color, t = (1, pc - 15) if pc >= 15 else (0, pc - 1)
            base = t if t < 8 else {8: 0, 9: 1, 10: 2, 11: 3, 12: 5, 13: 6}[t]
            ch = _SYM[base] if color == 0 else _SYM[base].lower()
            row += ("+" if t >= 8 else "") + ch
        out.append(row + (str(empty) if empty else ""))
    hands = ""
    for color in (0, 1):
        for t in _HAND_ORDER:
            n = int(rec[81 + 7 * color + t])
            if n:
                hands += (str(n) if n > 1 else "") + (_SYM[t] if color == 0 else _SYM[t].lower())
    mc = int(rec[96]) | (int(rec[97]) << 8)
    return "%s %s %s %d" % ("/".join(out), "b" if rec[95] == 0 else "w", hands or "-", mc + 1)


_TYPE_NAMES = ("PAWN", "LANCE", "KNIGHT", "SILVER", "GOLD", "BISHOP", "ROOK")
_KIF_HAND = ((6, "HI"), (5, "KA"), (4, "KI"), (3, "GI"), (2, "KE"), (1, "KY"), (0, "FU"))   # shogi_game_io.py:633-641
_KIF_TERMINATION = {"Tsumi": "詰み", "Toryo": "投了", "Sennichite": "千日手", "Stalemate": "持将棋",
                    "Max moves reached": "持将棋"}                                              # shogi_game_io.py:690-697
_KIF_DRAW_REASONS = ("Sennichite", "impasse", "Max moves reached")                              # shogi_game_io.py:722-726


def _symbol(pc):
    t, color = (pc - 1) % 14, (pc - 1) // 14
    base = t if t < 8 else {8: 0, 9: 1, 10: 2, 11: 3, 12: 5, 13: 6}[t]
    ch = _SYM[base] if color == 0 else _SYM[base].lower()
    return ("+" if t >= 8 else "") + ch


def state_to_text(rec):
    """Text board of a packed state record, character for character the reference's
    convert_game_to_text_representation (shogi_game_io.py:540-583)."""
    lines = []
    for r in range(9):
        cells = []
        for c in range(9):
            pc = int(rec[r * 9 + c])
            if pc == 0:
                cells.append(" . ")
            else:
                sym = _symbol(pc)
                cells.append(" %s " % sym if len(sym) == 1 else "%s " % sym)
        lines.append("%d " % (9 - r) + "".join(cells))
    lines.append("a b c d e f g h i")
    mc = int(rec[96]) | (int(rec[97]) << 8)
    lines.append("Turn: %s, Move: %d" % ("BLACK" if rec[95] == 0 else "WHITE", mc + 1))
    for color, who in ((0, "Black"), (1, "White")):
        hand = {_TYPE_NAMES[t]: int(rec[81 + 7 * color + t]) for t in range(7) if rec[81 + 7 * color + t] > 0}
        lines.append("%s's hand: %s" % (who, hand))
    return "\n".join(lines)


def record_to_kif(rec, moves, sente_player_name="Sente", gote_player_name="Gote", date=None):
    """KIF text of a game: `rec` = its current packed state record, `moves` = the 5-tuples played so far.
    Reproduces the reference's game_to_kif (shogi_game_io.py:586-733) including its quirks: the board block is
    always the HIRATE start, the hand lines and the side-to-move line describe the CURRENT position, drop moves
    are skipped (their tuple starts with None, :664-671) while the move numbers keep counting, and a stalemate
    prints its reason but no RESULT line."""
    date = date or datetime.date.today().strftime("%Y/%m/%d")
    lines = ["#KIF version=2.0 encoding=UTF-8", "*Event: Casual Game", "*Site: Local Machine", "*Date: %s" % date,
             "*Player Sente: %s" % sente_player_name, "*Player Gote: %s" % gote_player_name, "*Handicap: HIRATE",
             "P1-KY-KE-GI-KI-OU-KI-GI-KE-KY", "P2 * -HI * * * * * -KA * ", "P3-FU-FU-FU-FU-FU-FU-FU-FU-FU",
             "P4 * * * * * * * * * ", "P5 * * * * * * * * * ", "P6 * * * * * * * * * ",
             "P7+FU+FU+FU+FU+FU+FU+FU+FU+FU", "P8 * +KA * * * * * +HI * ", "P9+KY+KE+GI+KI+OU+KI+GI+KE+KY"]
    lines.append("P+" + "".join("%02d%s" % (int(rec[81 + t]), sym) for t, sym in _KIF_HAND))
    lines.append("P-" + "".join("%02d%s" % (int(rec[88 + t]), sym) for t, sym in _KIF_HAND))
    lines.append("+" if rec[95] == 0 else "-")
    lines.append("moves")
    for i, mv in enumerate(moves):
        if not mv or mv[0] is None or mv[1] is None or mv[2] is None or mv[3] is None:
            continue
        lines.append("%d %d%s%d%s%s" % (i + 1, mv[0] + 1, chr(mv[1] + ord("a")), mv[2] + 1, chr(mv[3] + ord("a")),
                                        "+" if mv[4] else ""))
    if rec[98]:
        reason = REASON_NAMES[int(rec[99])]
        shown = _KIF_TERMINATION.get(reason, reason)
        if shown:
            lines.append(shown)
        w = int(rec[100]) - 1
        if w == 0:
            lines.append("RESULT:SENTE_WIN")
        elif w == 1:
            lines.append("RESULT:GOTE_WIN")
        elif reason in _KIF_DRAW_REASONS:
            lines.append("RESULT:DRAW")
    lines.append("*EOF")
    return "\n".join(lines)


def game_to_kif(game, filename=None, sente_player_name="Sente", gote_player_name="Gote"):
    """Reference signature (shogi_game_io.py:586-591): returns the KIF string, or writes `filename` and returns None."""
    text = record_to_kif(game._rec(), [h["move"] for h in game.move_history], sente_player_name, gote_player_name)
    if filename:
        with open(filename, "w", encoding="utf-8") as f:
            f.write(text)
        return None
    return text


class ShogiGame:
    def __init__(self, max_moves_per_game=500, device="cuda:0"):
        self._env = VecShogiEnv(1, max_moves_per_game, device=device, auto_reset=False)
        self._mapper = PolicyOutputMapper()
        self._seed_value = None
        self.reset()

    max_moves_per_game = property(lambda self: self._env.max_moves_per_game)

    def seed(self, seed_value=None):
        self._seed_value = seed_value          # no-op like the reference (shogi_game.py:56-73)
        return self

    def _rec(self):
        return self._env.get_state(0)

    def reset(self):
        self._env.reset()
        self.move_history = []             # [{"move": 5-tuple}], the part of the reference's entries KIF export reads
        return self.get_observation()

    @classmethod
    def from_sfen(cls, sfen_str, max_moves_for_game_instance=500, device="cuda:0"):
        g = cls(max_moves_for_game_instance, device)
        g._env.set_state(0, parse_sfen(sfen_str))
        g.move_history = []
        g._env.evaluate_termination()      # the reference evaluates termination for the loaded position (shogi_game.py:343)
        return g

    # ------------------------------------------------------------------ reference-style accessors
    @property
    def current_player(self):
        return int(self._rec()[95])

    @property
    def move_count(self):
        r = self._rec()
        return int(r[96]) | (int(r[97]) << 8)

    @property
    def game_over(self):
        return bool(self._rec()[98])

    @property
    def winner(self):
        w = int(self._rec()[100]) - 1
        return None if w < 0 else w

    @property
    def termination_reason(self):
        r = self._rec()
        return REASON_NAMES[int(r[99])] if r[98] else None

    def get_piece(self, row, col):
        if not (0 <= row < 9 and 0 <= col < 9):
            return None
        pc = int(self._rec()[row * 9 + col])
        return None if pc == 0 else ((pc - 1) % 14, (pc - 1) // 14)   # (type, color)

    def get_observation(self):
        return self._env.obs[0].cpu().numpy().copy()

    get_state = get_observation

    def get_legal_mask(self):
        return self._env.legal_mask[0]

    def get_legal_moves(self):
        ids = torch.nonzero(self._env.legal_mask[0]).flatten().tolist()
        return [self._mapper.policy_index_to_shogi_move(i) for i in ids]

    def make_move(self, move_tuple):
        """(obs, reward, done, info) like shogi_game.py:574-660; ValueError on a move the engine rejects."""
        idx = self._mapper.shogi_move_to_policy_index(tuple(move_tuple))
        a = torch.tensor([idx], dtype=torch.int64, device=self._env.device)
        was_over = self.game_over
        self._env.step(a, validate=not was_over)
        reason = int(self._env.reason.item())
        if reason == 5:
            raise ValueError("Illegal move %r for the current position" % (move_tuple,))
        if not was_over:
            self.move_history.append({"move": tuple(move_tuple)})
        done = bool(self._env.done.item())
        info = {"reason": REASON_NAMES[reason] if done else "Game ongoing"}
        w = int(self._env.winner.item())
        if done and w >= 0:
            info["winner"] = "BLACK" if w == 0 else "WHITE"
        return self.get_observation(), float(self._env.reward.item()), done, info

    def to_sfen_string(self):
        return state_to_sfen(self._rec())

    to_sfen = to_sfen_string

    def to_string(self):
        return state_to_text(self._rec())
