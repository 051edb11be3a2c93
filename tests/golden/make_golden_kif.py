#!/usr/bin/env python3
"""Generate the game-record golden fixture (KIF export + text board) by running the *reference* here.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_kif.py

Output: tests/golden/kif.json -- per case: action-id stream (uniform-random legal play from a seed, or from an SFEN),
and the reference's game_to_kif() / to_string() / to_sfen_string() text at a few plies and at the end.  The KIF date
header is today's date in the reference (shogi_game_io.py:613); it is replaced by the token {DATE} in the fixture.
Reference objects exercised: keisei.shogi.ShogiGame, keisei.shogi.shogi_game_io.game_to_kif,
convert_game_to_text_representation."""
import datetime
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    import logging
    logging.disable(logging.CRITICAL)
    from keisei.shogi.shogi_game import ShogiGame
    from keisei.shogi.shogi_game_io import game_to_kif
    from keisei.utils import PolicyOutputMapper
    mapper = PolicyOutputMapper()
    today = datetime.date.today().strftime("%Y/%m/%d")

    def snap(game, **kw):
        kif = game_to_kif(game, **kw).replace("*Date: " + today, "*Date: {DATE}")
        return {"kif": kif, "text": game.to_string(), "sfen": game.to_sfen_string(), "kwargs": kw}

    cases = []
    specs = [("random_seed3_full", None, 500, 3, None), ("random_seed11_80plies", None, 500, 11, 80),
             ("random_seed5_max60", None, 60, 5, None),
             ("midgame_sfen", "ln1g1g1nl/1ks2r3/1pppp1bpp/p3spp2/9/P1P1P4/1P1PSPPPP/1BK1GR3/LNSG3NL b - 17", 500, 7, 40)]
    for name, sfen, max_moves, seed, stop in specs:
        game = ShogiGame.from_sfen(sfen, max_moves_for_game_instance=max_moves) if sfen else ShogiGame(max_moves_per_game=max_moves)
        rng = np.random.default_rng(seed)
        actions, snaps = [], {}
        snaps["0"] = snap(game)
        while not game.game_over and (stop is None or len(actions) < stop):
            ids = sorted(mapper.shogi_move_to_policy_index(m) for m in game.get_legal_moves())
            a = int(ids[int(rng.integers(len(ids)))])
            game.make_move(mapper.policy_index_to_shogi_move(a))
            actions.append(a)
            if len(actions) in (1, 7, 33):
                snaps[str(len(actions))] = snap(game)
        snaps[str(len(actions))] = snap(game, sente_player_name="Agent A", gote_player_name="Agent B")
        cases.append({"name": name, "sfen": sfen, "max_moves": max_moves, "actions": actions, "snaps": snaps,
                      "game_over": bool(game.game_over), "reason": game.termination_reason})
        print(name, len(actions), game.termination_reason)
    with open(os.path.join(HERE, "kif.json"), "w", encoding="utf-8") as f:
        json.dump(cases, f, ensure_ascii=False, indent=0)


if __name__ == "__main__":
    main()
