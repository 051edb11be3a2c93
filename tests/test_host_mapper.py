"""Host-side PolicyOutputMapper mirror vs the reference's dumped table (tests/golden/codec.npz) and the
known answers of the reference's tests/shogi/test_shogi_utils.py."""
import os

import numpy as np
import pytest
import torch

from shogidrl_amd.mapper import PolicyOutputMapper

G = os.path.join(os.path.dirname(__file__), "golden")


def test_total_actions_and_known_ids():
    m = PolicyOutputMapper()
    assert m.get_total_actions() == 13527                                   # test_shogi_utils.py:27
    a = m.shogi_move_to_policy_index((0, 0, 0, 1, False))
    assert m.shogi_move_to_policy_index((0, 0, 0, 1, True)) == a + 1         # :43 promo id = non-promo id + 1
    assert m.shogi_move_to_policy_index((None, None, 4, 4, 0)) == 13240      # :50-54 pawn drop at (4,4)
    assert m.shogi_move_to_policy_index((None, None, 0, 0, 1)) == 12961      # :60-66 lance drop at (0,0)
    with pytest.raises(IndexError):
        m.policy_index_to_shogi_move(13527)                                  # :75-80
    with pytest.raises(IndexError):
        m.policy_index_to_shogi_move(-1)
    with pytest.raises(ValueError):
        m.shogi_move_to_policy_index((0, 0, 0, 0, False))                    # null move is not in the table


def test_round_trip_matches_reference_table():
    tab = np.load(os.path.join(G, "codec.npz"))["table"]
    m = PolicyOutputMapper()
    for idx in range(13527):
        f, t, x = (int(v) for v in tab[idx])
        mv = m.policy_index_to_shogi_move(idx)
        if f < 0:
            assert mv == (None, None, t // 9, t % 9, x)
        else:
            assert mv == (f // 9, f % 9, t // 9, t % 9, bool(x))
        assert m.shogi_move_to_policy_index(mv) == idx


def test_legal_mask_and_enum_piece_types():
    import enum

    class PT(enum.Enum):
        PAWN = 0
        ROOK = 6

    m = PolicyOutputMapper(piece_type_enum=PT)
    moves = [(6, 4, 5, 4, False), (None, None, 4, 4, PT.PAWN), (None, None, 0, 8, PT.ROOK)]
    mask = m.get_legal_mask(moves, torch.device("cpu"))
    assert mask.dtype == torch.bool and mask.shape == (13527,) and int(mask.sum()) == 3
    assert m.policy_index_to_shogi_move(13240) == (None, None, 4, 4, PT.PAWN)
    with pytest.raises(ValueError):
        m.get_legal_mask([(9, 0, 0, 0, False)], torch.device("cpu"))
