"""PolicyOutputMapper -- action id <-> Shogi move tuple, host side.

Mirror of keisei/utils/utils.py:180-336 (same method names, argument meaning and error behaviour) in
closed form instead of a 13 527-entry dict: board move id = ((from_sq*80 + to_sq - (to_sq > from_sq))*2
+ promote, drop id = 12960 + to_sq*7 + piece_type, sq = row*9 + col.  Move tuples are
(from_r, from_c, to_r, to_c, promote: bool) or (None, None, to_r, to_c, piece_type) where piece_type is
either an int 0..6 (P,L,N,S,G,B,R) or any enum whose .value is that int (the reference's PieceType)."""
import torch

from ._lib import NUM_ACTIONS

N_BOARD_ACTIONS = 12960


def _ptype(x):
    v = getattr(x, "value", x)
    if not isinstance(v, int) or isinstance(v, bool) or not 0 <= v <= 6:
        raise ValueError("drop piece type must be 0..6 (P,L,N,S,G,B,R), got %r" % (x,))
    return v


class PolicyOutputMapper:
    def __init__(self, piece_type_enum=None):
        # optional enum class (e.g. keisei's PieceType) used to build drop tuples
        self._enum = piece_type_enum

    def get_total_actions(self):
        return NUM_ACTIONS

    def shogi_move_to_policy_index(self, move):
        try:
            fr, fc, tr, tc, x = move
        except (TypeError, ValueError):
            raise ValueError("Move %r not found in PolicyOutputMapper's known moves." % (move,))
        if not all(isinstance(v, int) and 0 <= v <= 8 for v in (tr, tc)):
            raise ValueError("Move %r not found in PolicyOutputMapper's known moves." % (move,))
        to = tr * 9 + tc
        if fr is None and fc is None:
            return N_BOARD_ACTIONS + to * 7 + _ptype(x)
        if not all(isinstance(v, int) and 0 <= v <= 8 for v in (fr, fc)) or not isinstance(x, bool):
            raise ValueError("Move %r not found in PolicyOutputMapper's known moves." % (move,))
        frm = fr * 9 + fc
        if frm == to:
            raise ValueError("Move %r not found in PolicyOutputMapper's known moves." % (move,))
        return (frm * 80 + to - (1 if to > frm else 0)) * 2 + (1 if x else 0)

    def policy_index_to_shogi_move(self, idx):
        if not 0 <= idx < NUM_ACTIONS:
            raise IndexError("Policy index %d is out of bounds (0-%d)." % (idx, NUM_ACTIONS - 1))
        if idx >= N_BOARD_ACTIONS:
            k = idx - N_BOARD_ACTIONS
            to, pt = divmod(k, 7)
            return (None, None, to // 9, to % 9, self._enum(pt) if self._enum is not None else pt)
        pair, promo = divmod(idx, 2)
        frm, j = divmod(pair, 80)
        to = j + (1 if j >= frm else 0)
        return (frm // 9, frm % 9, to // 9, to % 9, bool(promo))

    def get_legal_mask(self, legal_shogi_moves, device):
        """bool[13527] with True at the ids of the given moves (utils.py:310-336). Host-side helper for
        callers that still hold move lists; the batched env writes masks on the GPU directly."""
        idx = []
        for m in legal_shogi_moves:
            try:
                idx.append(self.shogi_move_to_policy_index(m))
            except ValueError as e:
                raise ValueError("CRITICAL: Legal move %r could not be mapped to policy index. Original error: %s" % (m, e)) from e
        mask = torch.zeros(NUM_ACTIONS, dtype=torch.bool)
        if idx:
            mask[torch.tensor(idx, dtype=torch.long)] = True
        return mask.to(device)
