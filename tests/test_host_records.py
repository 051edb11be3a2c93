"""Game-record formats: the text board and the KIF export of shogidrl_amd.game against the text the
reference produced for the same games (tests/golden/kif.json, made by tests/golden/make_golden_kif.py).  The host
formatting code works on packed state records; here the records come from the oracle replaying the fixture's action
stream, in tests/test_gpu_agent.py from the HIP engine."""
import json
import os

import pytest

from oracle import ko
from shogidrl_amd.game import parse_sfen, record_to_kif, state_to_sfen, state_to_text
from shogidrl_amd.mapper import PolicyOutputMapper

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "kif.json"), encoding="utf-8"))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_text_kif_sfen_match_reference(case):
    mapper = PolicyOutputMapper()
    g = ko.Game(case["max_moves"], sfen=case["sfen"])
    moves = []
    n_checked = 0
    for ply in range(len(case["actions"]) + 1):
        snap = case["snaps"].get(str(ply))
        if snap is not None:
            rec = g.export_state()
            assert state_to_text(rec) == snap["text"]
            assert state_to_sfen(rec) == snap["sfen"]
            assert record_to_kif(rec, moves, date="{DATE}", **snap["kwargs"]) == snap["kif"]
            n_checked += 1
        if ply < len(case["actions"]):
            a = case["actions"][ply]
            g.make_move(a)
            moves.append(mapper.policy_index_to_shogi_move(a))
    assert n_checked == len(case["snaps"])
    assert g.game_over == case["game_over"]


def test_sfen_round_trip_of_fixture_positions():
    for case in CASES:
        for snap in case["snaps"].values():
            assert state_to_sfen(parse_sfen(snap["sfen"])) == snap["sfen"]


def test_kif_stalemate_prints_reason_without_result_line():
    # "stalemate" is not in the reference's termination map nor in its draw list (shogi_game_io.py:690-726)
    rec = parse_sfen("4k4/9/9/9/9/9/9/9/4K4 b - 1")
    rec[98], rec[99], rec[100] = 1, 2, 0
    lines = record_to_kif(rec, [], date="x").split("\n")
    assert lines[-2] == "stalemate" and lines[-1] == "*EOF"
    rec[99] = 4
    assert record_to_kif(rec, [], date="x").split("\n")[-3:] == ["千日手", "RESULT:DRAW", "*EOF"]
