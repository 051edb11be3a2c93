#!/usr/bin/env python3
"""Generate the Shogi golden fixtures by running the *reference* (Keisei) in THIS container.

Run (build container only -- /root/reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_shogi.py [--games 48] [--procs 8]

Outputs (committed, data only -- inputs and the reference's outputs):

    tests/golden/codec.npz       every action id -> (from_sq, to_sq, promo) | (-1, to_sq, piece)
    tests/golden/positions.json  SFEN -> legal action ids, in-check, termination, obs/hash facts
    tests/golden/traces.npz      seeded full games: action stream + per-ply outputs
    tests/golden/sequences.json  scripted move sequences (sennichite cases) + per-ply outputs

The reference objects exercised: keisei.shogi.ShogiGame (get_legal_moves / make_move /
get_observation / from_sfen / is_in_check) and keisei.utils.PolicyOutputMapper.
"""
import argparse
import json
import multiprocessing as mp
import os
import re
import sys
import zlib

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

REASONS = {None: 0, "Game ongoing": 0, "Tsumi": 1, "stalemate": 2,
           "Max moves reached": 3, "Sennichite": 4}


def _imports():
    import logging
    logging.disable(logging.CRITICAL)
    from keisei.shogi.shogi_core_definitions import Color, PieceType
    from keisei.shogi.shogi_game import ShogiGame
    from keisei.utils import PolicyOutputMapper
    return Color, PieceType, ShogiGame, PolicyOutputMapper


def legal_ids(game, mapper):
    moves = game.get_legal_moves()
    ids = sorted(mapper.shogi_move_to_policy_index(m) for m in moves)
    assert len(set(ids)) == len(ids)
    return ids


def ids_crc(ids):
    return zlib.crc32(np.asarray(ids, dtype="<u2").tobytes())


def obs_crc(obs):
    obs = np.ascontiguousarray(obs, dtype="<f4")
    assert obs.shape == (46, 9, 9)
    return zlib.crc32(obs.tobytes())


def winner_code(game):
    return -1 if game.winner is None else int(game.winner.value)


# --------------------------------------------------------------------------- codec
def make_codec():
    _, PieceType, _, PolicyOutputMapper = _imports()
    mapper = PolicyOutputMapper()
    n = mapper.get_total_actions()
    tab = np.zeros((n, 3), dtype=np.int16)
    for i in range(n):
        m = mapper.policy_index_to_shogi_move(i)
        if m[0] is None:
            tab[i] = (-1, m[2] * 9 + m[3], m[4].value)
        else:
            tab[i] = (m[0] * 9 + m[1], m[2] * 9 + m[3], int(m[4]))
    np.savez_compressed(os.path.join(HERE, "codec.npz"), table=tab)
    print("codec:", n)


# ----------------------------------------------------------------------- positions
# Hand-written probe positions (inputs only; expected outputs come from the reference).
EXTRA_SFENS = [
    # uchi-fu-zume family (Black drops P at 5b / (1,4) against king on 5a)
    "3GkG3/3G1G3/9/9/9/9/9/9/8K b P 1",          # true uchi-fu-zume
    "4k4/9/9/9/9/9/9/9/8K b P 1",                # king escapes / captures pawn
    "4k4/r8/9/9/9/9/9/9/8K b P 1",               # rook captures the pawn
    "3GkG3/3G1G3/4g4/9/4R4/9/9/9/8K b P 1",      # capturer un-pinned by the dropped pawn
    "3GkG3/r2G1G3/9/9/9/9/9/9/8K b P 1",         # rook on rank b blocked by gold -> still mate
    "3GkG3/3G1G1b1/9/9/9/9/9/9/8K b P 1",        # bishop far away, cannot capture
    "3GkG3/3G1Gb2/9/9/9/9/9/9/8K b P 1",         # (1,6) bishop cannot reach (1,4)
    "2bGkG3/3G1G3/9/9/9/9/9/9/8K b P 1",
    "3GkG3/3G1G3/5n3/9/9/9/9/9/8K b P 1",        # knight (white) attacks backwards? no
    "3+RkG3/5G3/4L4/9/9/9/9/9/8K b P 1",
    "4k4/9/4G4/9/9/9/9/9/8K b P 1",              # pawn protected by gold, king escapes sideways
    "3sks3/9/4G4/9/9/9/9/9/8K b P 1",            # silvers capture
    "3gkg3/9/4G4/9/9/9/9/9/8K b P 1",            # golds next to king capture sideways? (no: golds move)
    "3lkl3/3p1p3/4G4/9/9/9/9/9/8K b P 1",        # boxed-in king, protected pawn -> uchi-fu-zume
    "3lkl3/3p1p3/9/9/9/9/9/9/8K b P 1",          # boxed-in king, unprotected pawn -> king takes
    # white drops pawn against black king (mirror)
    "8k/9/9/9/9/9/9/3g1g3/3gKg3 w p 1",
    "8k/9/9/9/9/9/4g4/3P1P3/3LKL3 w p 1",
    "8k/9/9/9/9/9/9/3P1P3/3LKL3 w p 1",
    # nifu / last-rank drop restrictions / many hands
    "4k4/9/9/9/9/9/P1P1P1P1P/9/4K4 b 2P2L2N 1",
    "4k4/9/9/9/9/9/+P1P1P1P1P/9/4K4 b PLNSGBR 1",
    "4k4/p1p1p1p1p/9/9/9/9/9/9/4K4 w 2p2l2nsgbr 2",
    "4k4/9/9/9/9/9/9/9/4K4 b 18P4L4N4S4G2B2R 1",
    "4k4/9/9/9/9/9/9/9/4K4 w 18p4l4n4s4g2b2r 2",
    # checks: single slider (interpositions by drop), contact, double check, knight check
    "4k4/9/9/9/4r4/9/9/9/4K4 b GS 1",
    "4k4/9/9/9/9/9/9/4g4/4K4 b GS 1",
    "4k4/9/9/9/4r4/9/9/3n5/4K4 b GS 1",
    "4k4/9/9/9/9/9/2b6/9/4K3r b GSP 1",
    "4k4/9/9/9/9/9/3n5/9/4K4 b G 1",
    "4k4/9/9/9/8b/9/9/9/4K3l w - 1",
    # pins: rook/bishop/lance pins, pinned piece moving along the ray
    "4k4/4r4/9/9/9/9/4S4/9/4K4 b - 1",
    "4k4/4r4/9/9/9/9/4R4/9/4K4 b - 1",
    "4k4/4l4/9/9/9/9/4L4/9/4K4 b - 1",
    "k8/1b7/9/9/9/9/6B2/9/8K b - 1",
    "k8/1b7/9/9/9/9/6G2/9/8K b - 1",
    "4k4/9/9/9/r3N3K/9/9/9/9 b - 1",
    "4k4/4R4/9/9/9/9/4s4/9/4K4 w - 2",
    # promotion zone: optional / forced promotions for every type, both colours
    "4k4/9/PLNSGBR2/9/9/9/9/9/4K4 b - 1",
    "4k4/PLNSGBR2/9/9/9/9/9/9/4K4 b - 1",
    "4k4/9/9/PLNSGBR2/9/9/9/9/4K4 b - 1",
    "1PLNSGBR1/4k4/9/9/9/9/9/9/4K4 b - 1",
    "4k4/9/9/9/9/9/plnsgbr2/9/4K4 w - 2",
    "4k4/9/9/9/9/9/9/plnsgbr2/4K4 w - 2",
    "4k4/9/9/9/9/plnsgbr2/9/9/4K4 w - 2",
    "4k4/9/+P+L+N+S+B+R3/9/9/9/+p+l+n+s+b+r3/9/4K4 b - 1",
    "4k4/9/+P+L+N+S+B+R3/9/9/9/+p+l+n+s+b+r3/9/4K4 w - 2",
    # kings adjacent / missing kings / stalemate-like
    "9/9/9/4k4/4K4/9/9/9/9 b - 1",
    "9/9/9/9/4K4/9/9/9/9 b P 1",
    "9/9/9/9/4k4/9/9/9/9 b P 1",
    "9/9/9/9/4K4/9/9/9/4k4 w p 2",
    "8k/9/9/9/9/9/9/9/K8 b - 1",
    "7+Rk/9/9/9/9/9/9/9/K7+r b - 1",
    "k8/9/9/9/9/9/9/9/+r+r6K b - 1",
    "+B7k/9/9/9/9/9/9/1+r7/K8 b - 1",
    # middle-game like positions
    "ln1g1g1nl/1ks2r3/1pppp1bpp/p3spp2/9/P1P1P4/1P1PSPPPP/1BK1GS1R1/LN3G1NL b - 17",
    "l6nl/5+P1gk/2np1S3/p1p4Pp/3P2Sp1/1PPb2P1P/P5GS1/R8/LN4bKL w RGgsn5p 80",
    "lnsgkgsnl/1r5b1/ppppppppp/9/9/9/PPPPPPPPP/1B5R1/LNSGKGSNL w - 2",
    "lnsgk1snl/1r4gb1/p1pppp1pp/6p2/1p7/2P6/PP1PPPPPP/1B5R1/LNSGKGSNL b - 7",
    "ln1gk2nl/1r1s2g2/p1pp1p1pp/4p1p2/1p7/2PP5/PPB1PPPPP/2GS3R1/LN2KGSNL b Bb 15",
]


def harvest_test_sfens():
    pat = re.compile(r'"([^"\n]{5,120} [bw] \S+ \d+)"')
    out = []
    tdir = os.path.join(REF, "tests", "shogi")
    for fn in sorted(os.listdir(tdir)):
        if fn.endswith(".py"):
            with open(os.path.join(tdir, fn), "r", encoding="utf-8") as f:
                out += pat.findall(f.read())
    for fn in ["tests/evaluation/test_evaluate_evaluator.py"]:
        p = os.path.join(REF, fn)
        if os.path.exists(p):
            with open(p, "r", encoding="utf-8") as f:
                out += pat.findall(f.read())
    seen, uniq = set(), []
    for s in out:
        if s not in seen and s.count("/") == 8:
            seen.add(s)
            uniq.append(s)
    return uniq


def position_record(sfen, max_moves=500, playout=0, seed=0):
    Color, PieceType, ShogiGame, PolicyOutputMapper = _imports()
    mapper = position_record.mapper
    try:
        g = ShogiGame.from_sfen(sfen, max_moves_for_game_instance=max_moves)
    except Exception as e:  # malformed test strings: recorded as rejected
        return {"sfen": sfen, "valid": False, "error": type(e).__name__}
    rec = {"sfen": sfen, "valid": True, "max_moves": max_moves}
    # read the termination fields BEFORE get_legal_moves(): its simulate/undo path clears
    # game_over/winner/reason as a side effect (shogi_move_execution.py:218-221)
    rec["game_over"] = bool(g.game_over)
    rec["reason"] = REASONS[g.termination_reason]
    rec["winner"] = winner_code(g)
    try:
        ids = legal_ids(g, mapper)
    except ValueError as e:
        # e.g. a king in hand after a scripted king capture: not mappable
        return {"sfen": sfen, "valid": False, "error": "unmappable:" + str(e)[:40]}
    rec["legal"] = ids
    rec["in_check"] = [bool(g.is_in_check(Color.BLACK)), bool(g.is_in_check(Color.WHITE))]
    rec["obs_crc"] = obs_crc(g.get_observation())
    rec["move_count"] = int(g.move_count)
    rec["side"] = int(g.current_player.value)
    # short random playout from the position: action stream + outputs per ply
    if playout and not rec["game_over"] and ids:
        rng = np.random.default_rng(seed)
        plies = []
        for _ in range(playout):
            ids = legal_ids(g, mapper)
            if not ids or g.game_over:
                break
            a = int(ids[rng.integers(len(ids))])
            mv = mapper.policy_index_to_shogi_move(a)
            obs, rew, done, info = g.make_move(mv)
            try:
                nxt = [] if done else legal_ids(g, mapper)
            except ValueError:
                break
            plies.append({"a": a, "r": float(rew), "done": bool(done),
                          "reason": REASONS[info["reason"]], "winner": winner_code(g),
                          "n_legal": len(nxt), "legal_crc": ids_crc(nxt),
                          "obs_crc": obs_crc(obs)})
            if done:
                break
        rec["playout"] = plies
    return rec


def make_positions():
    _, _, _, PolicyOutputMapper = _imports()
    position_record.mapper = PolicyOutputMapper()
    sfens = harvest_test_sfens()
    print("harvested", len(sfens), "SFEN strings from the reference's tests")
    recs = []
    for i, s in enumerate(sfens + EXTRA_SFENS):
        recs.append(position_record(s, playout=24, seed=1000 + i))
    nvalid = sum(r["valid"] for r in recs)
    with open(os.path.join(HERE, "positions.json"), "w") as f:
        json.dump(recs, f, separators=(",", ":"))
    print("positions:", len(recs), "valid:", nvalid)


# -------------------------------------------------------------------------- traces
def play_game(args):
    gi, seed, max_moves, policy, keep_full = args
    Color, PieceType, ShogiGame, PolicyOutputMapper = _imports()
    mapper = PolicyOutputMapper()
    rng = np.random.default_rng(seed)
    g = ShogiGame(max_moves_per_game=max_moves)
    obs0 = g.reset()
    ids = legal_ids(g, mapper)
    rows = []          # per ply: a, r, done, reason, winner, n_legal(next), legal_crc(next), obs_crc(next)
    full = [ids] if keep_full else None
    full_obs = []
    first = (len(ids), ids_crc(ids), obs_crc(obs0))
    while True:
        if policy == "uniform":
            a = int(ids[rng.integers(len(ids))])
        else:  # "drops": prefer drops half of the time, promotions otherwise uniform
            drops = [i for i in ids if i >= 12960]
            if drops and rng.random() < 0.5:
                a = int(drops[rng.integers(len(drops))])
            else:
                a = int(ids[rng.integers(len(ids))])
        mv = mapper.policy_index_to_shogi_move(a)
        obs, rew, done, info = g.make_move(mv)
        ids = [] if done else legal_ids(g, mapper)
        rows.append((a, rew, int(done), REASONS[info["reason"]], winner_code(g),
                     len(ids), ids_crc(ids), obs_crc(obs)))
        if keep_full:
            full.append(ids)
            if len(rows) % 16 == 0 or done:
                full_obs.append((len(rows), obs.copy()))
        if done:
            break
    return gi, seed, max_moves, policy, first, rows, full, full_obs


def make_traces(n_games, procs):
    jobs = []
    for gi in range(n_games):
        seed = 100 + gi
        if gi % 6 == 5:
            max_moves = 60 + 7 * (gi % 5)     # short truncation games ("Max moves reached")
        else:
            max_moves = 500
        policy = "drops" if gi % 3 == 1 else "uniform"
        keep_full = gi < 12
        jobs.append((gi, seed, max_moves, policy, keep_full))
    with mp.Pool(procs) as pool:
        res = pool.map(play_game, jobs, chunksize=1)
    res.sort(key=lambda r: r[0])
    game_off = [0]
    A, R, D, RS, W, NL, LC, OC = [], [], [], [], [], [], [], []
    meta = []
    full_ids, full_off, full_game_ply = [], [0], []
    fobs, fobs_idx = [], []
    for gi, seed, max_moves, policy, first, rows, full, full_obs in res:
        meta.append((seed, max_moves, 0 if policy == "uniform" else 1, first[0], first[1], first[2]))
        for (a, r, d, rs, w, nl, lc, oc) in rows:
            A.append(a); R.append(r); D.append(d); RS.append(rs); W.append(w)
            NL.append(nl); LC.append(lc); OC.append(oc)
        game_off.append(len(A))
        if full is not None:
            for ply, ids in enumerate(full):
                full_ids += ids
                full_off.append(len(full_ids))
                full_game_ply.append((gi, ply))
            for ply, o in full_obs:
                fobs.append(o)
                fobs_idx.append((gi, ply))
    np.savez_compressed(
        os.path.join(HERE, "traces.npz"),
        game_off=np.asarray(game_off, np.int64),
        meta=np.asarray(meta, np.int64),       # seed, max_moves, policy, n_legal0, legal_crc0, obs_crc0
        action=np.asarray(A, np.int16),
        reward=np.asarray(R, np.float32),
        done=np.asarray(D, np.uint8),
        reason=np.asarray(RS, np.uint8),
        winner=np.asarray(W, np.int8),
        n_legal=np.asarray(NL, np.int16),
        legal_crc=np.asarray(LC, np.uint32),
        obs_crc=np.asarray(OC, np.uint32),
        full_ids=np.asarray(full_ids, np.int16),
        full_off=np.asarray(full_off, np.int64),
        full_game_ply=np.asarray(full_game_ply, np.int32),
        full_obs=np.asarray(fobs, np.float32),
        full_obs_idx=np.asarray(fobs_idx, np.int32),
    )
    rs = np.asarray(RS)
    print("traces: games", n_games, "plies", len(A),
          "tsumi", int((rs == 1).sum()), "stalemate", int((rs == 2).sum()),
          "maxmoves", int((rs == 3).sum()), "sennichite", int((rs == 4).sum()),
          "max n_legal", int(max(NL)))


# ----------------------------------------------------------------------- sequences
def run_sequence(name, moves, max_moves=500, sfen=None):
    Color, PieceType, ShogiGame, PolicyOutputMapper = _imports()
    mapper = PolicyOutputMapper()
    g = ShogiGame.from_sfen(sfen, max_moves) if sfen else ShogiGame(max_moves_per_game=max_moves)
    out = []
    for mv in moves:
        mv = tuple(mv)
        if mv[0] is None:
            mv = (None, None, mv[2], mv[3], PieceType(mv[4]))
        else:
            mv = (mv[0], mv[1], mv[2], mv[3], bool(mv[4]))
        a = mapper.shogi_move_to_policy_index(mv)
        obs, rew, done, info = g.make_move(mv)
        ids = [] if done else legal_ids(g, mapper)
        out.append({"a": a, "r": float(rew), "done": bool(done), "reason": REASONS[info["reason"]],
                    "winner": winner_code(g), "n_legal": len(ids), "legal_crc": ids_crc(ids),
                    "obs_crc": obs_crc(obs)})
        if done:
            break
    return {"name": name, "sfen": sfen, "max_moves": max_moves, "plies": out}


def make_sequences():
    seqs = []
    # (1) start-position shuffles with different pieces: sennichite only at ply 16
    sh = []
    for (b_from, b_to, w_from, w_to) in [((7, 7), (7, 6), (1, 1), (1, 2)),
                                         ((8, 3), (7, 3), (0, 3), (1, 3)),
                                         ((8, 5), (7, 5), (0, 5), (1, 5)),
                                         ((8, 4), (7, 4), (0, 4), (1, 4))]:
        sh += [(*b_from, *b_to, 0), (*w_from, *w_to, 0), (*b_to, *b_from, 0), (*w_to, *w_from, 0)]
    seqs.append(run_sequence("start_shuffle_16", sh))
    # (2) pure king shuffle from the start position (same piece each time)
    ks = [(8, 4, 7, 4, 0), (0, 4, 1, 4, 0), (7, 4, 8, 4, 0), (1, 4, 0, 4, 0)] * 5
    seqs.append(run_sequence("king_shuffle", ks))
    # (3) rook shuffle
    rs_ = [(7, 7, 7, 6, 0), (1, 1, 1, 2, 0), (7, 6, 7, 7, 0), (1, 2, 1, 1, 0)] * 5
    seqs.append(run_sequence("rook_shuffle", rs_))
    # (4) bare kings + rooks from a SFEN, repetition with a position that is NOT the initial one
    seqs.append(run_sequence("sfen_shuffle", [(8, 4, 8, 3, 0), (0, 4, 0, 5, 0), (8, 3, 8, 4, 0), (0, 5, 0, 4, 0)] * 5,
                             sfen="4k4/9/9/9/9/9/9/9/4K4 b - 1"))
    # (5) repetition cut short by max_moves (max-moves wins over sennichite in check order)
    seqs.append(run_sequence("shuffle_maxmoves_12", ks, max_moves=12))
    seqs.append(run_sequence("shuffle_maxmoves_13", ks, max_moves=13))
    # (6) capture + drop cycle so hands matter for the hash
    seqs.append(run_sequence("hand_cycle", [(4, 4, 3, 4, 0), (0, 0, 1, 0, 0), (None, None, 5, 5, 0), (1, 0, 0, 0, 0)]
                             + [(8, 8, 7, 8, 0), (0, 0, 1, 0, 0), (7, 8, 8, 8, 0), (1, 0, 0, 0, 0)] * 4,
                             sfen="k8/9/9/4p4/4R4/9/9/9/8K b - 1"))
    # (7) fool's-mate style quick Tsumi: mating move at ply == max_moves is still a win
    mate = [(6, 2, 5, 2, 0), (2, 3, 3, 3, 0), (7, 1, 2, 6, 1)]  # bishop takes 3c+ : not mate, just a trace
    seqs.append(run_sequence("bishop_promote_capture", mate))
    with open(os.path.join(HERE, "sequences.json"), "w") as f:
        json.dump(seqs, f, separators=(",", ":"))
    for s in seqs:
        last = s["plies"][-1]
        print("sequence", s["name"], "plies", len(s["plies"]), "end reason", last["reason"], "done", last["done"])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=48)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    todo = a.only.split(",") if a.only else ["codec", "positions", "sequences", "traces"]
    if "codec" in todo:
        make_codec()
    if "positions" in todo:
        make_positions()
    if "sequences" in todo:
        make_sequences()
    if "traces" in todo:
        make_traces(a.games, a.procs)
