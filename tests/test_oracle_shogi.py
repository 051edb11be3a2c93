"""CPU oracle (oracle/keisei_oracle.c) pinned against golden vectors generated from the reference
(tests/golden/make_golden_shogi.py).  Mirrors the reference's tests/shogi/test_legal_mask_generation.py,
test_shogi_utils.py, test_shogi_game_core_logic.py (sennichite, max-moves) known answers."""
import json
import os
import zlib

import numpy as np
import pytest

from oracle import ko

G = os.path.join(os.path.dirname(__file__), "golden")


def ids_crc(ids):
    return zlib.crc32(np.asarray(ids, dtype="<u2").tobytes())


def obs_crc(obs):
    return zlib.crc32(np.ascontiguousarray(obs, dtype="<f4").tobytes())


def test_codec_closed_form_matches_reference_table():
    tab = np.load(os.path.join(G, "codec.npz"))["table"]
    assert tab.shape == (13527, 3)
    for a in range(13527):
        f, t, x = ko.decode_action(a)
        assert (f, t, x) == tuple(int(v) for v in tab[a]), a
        assert ko.encode_action(f, t, x) == a
    # reference known answers: tests/shogi/test_shogi_utils.py:43-66
    assert ko.encode_action(-1, 4 * 9 + 4, 0) == 13240
    assert ko.encode_action(-1, 0, 1) == 12961
    with pytest.raises(IndexError):
        ko.decode_action(13527)


def test_initial_position_known_answers():
    g = ko.Game()
    assert len(g.legal_ids()) == 30          # test_legal_mask_generation.py:20-37
    assert g.legal_mask().sum() == 30
    assert not g.in_check(0) and not g.in_check(1)
    obs = g.observation()
    assert obs.shape == (46, 9, 9) and obs[42].min() == 1.0 and obs[43].max() == 0.0


@pytest.mark.parametrize("sfen,n", [
    ("4k4/4r4/9/9/9/9/9/9/4K4 b - 1", 4),          # test_legal_mask_generation.py:82-103
    ("9/9/9/9/9/4G4/4r4/4g4/4K4 b - 1", 0),        # :123-139 checkmate
    ("9/9/9/9/4K4/9/9/9/4k4 b P 1", 78),           # :154-184
    ("4k4/9/9/9/9/9/9/9/4K4 b - 1", 5),            # :421-440
])
def test_reference_test_known_counts(sfen, n):
    g = ko.Game(sfen=sfen)
    assert len(g.legal_ids()) == n


def test_positions_fixture():
    recs = json.load(open(os.path.join(G, "positions.json")))
    nvalid = 0
    for rec in recs:
        if not rec["valid"]:
            if rec.get("error", "").startswith("unmappable"):
                continue
            with pytest.raises(ValueError):
                ko.Game(sfen=rec["sfen"])
            continue
        nvalid += 1
        g = ko.Game(max_moves=rec["max_moves"], sfen=rec["sfen"])
        assert g.legal_ids().tolist() == rec["legal"], rec["sfen"]
        assert [g.in_check(0), g.in_check(1)] == rec["in_check"], rec["sfen"]
        assert g.game_over == rec["game_over"], rec["sfen"]
        assert g.reason == rec["reason"] and g.winner == rec["winner"], rec["sfen"]
        assert obs_crc(g.observation()) == rec["obs_crc"], rec["sfen"]
        assert g.side == rec["side"] and g.move_count == rec["move_count"]
        for ply in rec.get("playout", []):
            r, d = g.make_move(ply["a"])
            assert (r, d) == (ply["r"], ply["done"]), rec["sfen"]
            assert g.reason == ply["reason"] and g.winner == ply["winner"]
            assert obs_crc(g.observation()) == ply["obs_crc"]
            ids = [] if d else g.legal_ids().tolist()
            assert len(ids) == ply["n_legal"] and ids_crc(ids) == ply["legal_crc"], rec["sfen"]
    assert nvalid >= 80


def test_sequences_fixture_sennichite_and_max_moves():
    seqs = json.load(open(os.path.join(G, "sequences.json")))
    names = set()
    for s in seqs:
        names.add(s["name"])
        g = ko.Game(max_moves=s["max_moves"], sfen=s["sfen"]) if s["sfen"] else ko.Game(max_moves=s["max_moves"])
        for i, ply in enumerate(s["plies"]):
            r, d = g.make_move(ply["a"])
            assert (r, d, g.reason, g.winner) == (ply["r"], ply["done"], ply["reason"], ply["winner"]), (s["name"], i)
            assert obs_crc(g.observation()) == ply["obs_crc"]
            ids = [] if d else g.legal_ids().tolist()
            assert len(ids) == ply["n_legal"] and ids_crc(ids) == ply["legal_crc"]
    assert {"start_shuffle_16", "king_shuffle", "shuffle_maxmoves_13"} <= names
    # the untouched initial position is not in move_history -> Sennichite only at ply 16
    s16 = [s for s in seqs if s["name"] == "start_shuffle_16"][0]
    assert len(s16["plies"]) == 16 and s16["plies"][-1]["reason"] == 4 and not s16["plies"][11]["done"]


def test_deep_sequences_fixture_sennichite_far_back_and_inert_finished_games():
    """sequences_deep.json (reference output, tests/golden/make_golden_deep.py): repetitions whose earlier occurrences lie
    70+ / 520+ plies back in move_history, and make_move called again on a finished game (shogi_game.py:589-593)."""
    seqs = json.load(open(os.path.join(G, "sequences_deep.json")))
    seen = set()
    for s in seqs:
        g = ko.Game(max_moves=s["max_moves"])
        if "actions" in s:                                  # action stream whose per-ply outputs traces.npz already pins
            for a in s["actions"][:-1]:
                g.make_move(a)
        for i, ply in enumerate(s["plies"]):
            r, d = g.make_move(ply["a"])
            assert (r, d, g.reason, g.winner) == (ply["r"], ply["done"], ply["reason"], ply["winner"]), (s["name"], i)
            assert obs_crc(g.observation()) == ply["obs_crc"], (s["name"], i)
            ids = [] if d else g.legal_ids().tolist()
            assert len(ids) == ply["n_legal"] and ids_crc(ids) == ply["legal_crc"], (s["name"], i)
        assert g.game_over
        for i, ply in enumerate(s["after"]):                # the finished game is inert
            r, d = g.make_move(ply["a"])
            assert (r, d, g.reason, g.winner) == (ply["r"], ply["done"], ply["reason"], ply["winner"]), (s["name"], "after", i)
            assert obs_crc(g.observation()) == ply["obs_crc"]
        seen.add((s["plies"][-1]["reason"], s["plies"][-1]["winner"]))
    deep = {s["name"]: s for s in seqs}
    assert len(deep["deep_sennichite_70"]["plies"]) == 87 and len(deep["deep_sennichite_520"]["plies"]) == 536
    assert all(p["reason"] == 0 for p in deep["deep_sennichite_520"]["plies"][:-1])     # nothing ends before the 4th occurrence
    assert {(4, -1), (1, 0), (1, 1), (3, -1)} <= seen
    assert [p["r"] for p in deep["inert_after_reason1_winner0"]["after"]] == [1.0, 1.0, 1.0]   # nominal mover = the winner


def test_traces_fixture_full_games():
    z = np.load(os.path.join(G, "traces.npz"))
    off = z["game_off"]
    meta = z["meta"]
    full_lookup = {(int(g), int(p)): i for i, (g, p) in enumerate(z["full_game_ply"])}
    full_obs_lookup = {(int(g), int(p)): i for i, (g, p) in enumerate(z["full_obs_idx"])}
    n_games = len(off) - 1
    assert n_games >= 32
    seen_reasons = set()
    for gi in range(n_games):
        seed, max_moves, policy, nl0, lc0, oc0 = (int(x) for x in meta[gi])
        g = ko.Game(max_moves=max_moves)
        ids = g.legal_ids().tolist()
        assert len(ids) == nl0 and ids_crc(ids) == lc0 and obs_crc(g.observation()) == oc0
        for k, i in enumerate(range(off[gi], off[gi + 1])):
            r, d = g.make_move(int(z["action"][i]))
            assert r == float(z["reward"][i]) and d == bool(z["done"][i]), (gi, k)
            assert g.reason == int(z["reason"][i]) and g.winner == int(z["winner"][i]), (gi, k)
            obs = g.observation()
            assert obs_crc(obs) == int(z["obs_crc"][i]), (gi, k)
            ids = [] if d else g.legal_ids().tolist()
            assert len(ids) == int(z["n_legal"][i]) and ids_crc(ids) == int(z["legal_crc"][i]), (gi, k)
            key = (gi, k + 1)
            if key in full_lookup:
                j = full_lookup[key]
                assert ids == z["full_ids"][z["full_off"][j]:z["full_off"][j + 1]].tolist()
            if key in full_obs_lookup:
                assert np.array_equal(obs, z["full_obs"][full_obs_lookup[key]])
        assert g.game_over
        seen_reasons.add(g.reason)
    assert {1, 3} <= seen_reasons


def test_c_selfplay_loop_is_the_checked_step_path():
    """bench.py's CPU-baseline loop (ko_selfplay_random) makes the same moves as the Python-driven loop over the pinned
    entry points when the same xorshift64* stream picks the actions."""
    import numpy as np
    from oracle import ko
    vec = ko.VecGames(3, 60)
    steps, seed = 90, 77
    assert ko.selfplay_random(vec, steps, seed) == 3 * steps
    ref = [ko.Game(60) for _ in range(3)]
    x = seed
    M = (1 << 64) - 1
    for _ in range(steps):
        for g in ref:
            ids = g.legal_ids()
            if len(ids) == 0:
                g.reset()
                ids = g.legal_ids()
            x ^= x >> 12
            x = (x ^ (x << 25)) & M
            x ^= x >> 27
            k = ((x * 0x2545F4914F6CDD1D) & M) >> 33
            _, done = g.make_move(int(np.sort(ids)[k % len(ids)]))
            if done:
                g.reset()
    for a, b in zip(vec.games, ref):
        assert np.array_equal(a.export_state(), b.export_state())
