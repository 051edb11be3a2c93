"""Batched evaluation games on the HIP engine -- the GPU counterpart of the game loops inside Keisei's
evaluation strategies (keisei/evaluation/strategies/single_opponent.py:170-220: `while not game.game_over:
legal = game.get_legal_moves(); mask = mapper.get_legal_mask(...); move = player.select_action(...);
game.make_move(move)`), which are a second consumer of the same env / select_action API.  N games are
played to the end at once; finished games stop consuming moves (auto-reset is off)."""
from dataclasses import dataclass

import torch

from . import ops
from .env import VecShogiEnv


@dataclass
class EvaluationResult:
    games: int
    black_wins: int
    white_wins: int
    draws: int
    reasons: dict          # termination reason name -> count
    avg_game_length: float

    @property
    def win_rate_black(self):
        return self.black_wins / max(self.games, 1)


class UniformRandomPolicy:
    """Stand-in for keisei/utils/opponents.py SimpleRandomOpponent: uniform over the legal actions."""

    def __init__(self, seed=0):
        self.seed, self._step = seed, 0

    def select_actions(self, obs, legal_mask, *, is_training=False):
        logits = torch.zeros(legal_mask.shape, dtype=torch.float32, device=legal_mask.device)
        a, lp = ops.masked_sample(logits, legal_mask, self.seed, self._step, deterministic=False)
        self._step += 1
        return a, lp, torch.zeros(legal_mask.shape[0], device=legal_mask.device)


@torch.no_grad()
def play_games(black, white, num_games, max_moves_per_game=500, device="cuda:0", deterministic=True):
    """`black` / `white`: objects with select_actions(obs, legal_mask, is_training=...) (VecPPOAgent or a policy
    above).  Every game is played from the start position until it ends; returns an EvaluationResult."""
    from .env import REASON_NAMES
    env = VecShogiEnv(num_games, max_moves_per_game, device=device, auto_reset=False)
    obs, mask = env.reset()
    over = torch.zeros(num_games, dtype=torch.bool, device=env.device)
    reason = torch.zeros(num_games, dtype=torch.uint8, device=env.device)
    winner = torch.full((num_games,), -1, dtype=torch.int8, device=env.device)
    length = torch.zeros(num_games, dtype=torch.int32, device=env.device)
    for ply in range(max_moves_per_game):
        player = black if ply % 2 == 0 else white              # all games are in lock-step: same side to move
        a, _, _ = player.select_actions(obs, mask, is_training=not deterministic)
        obs, _, done, info = env.step(a, validate=False)       # finished games ignore further actions (shogi_game.py:589-593)
        newly = done & ~over
        reason = torch.where(newly, info["reason"], reason)
        winner = torch.where(newly, info["winner"], winner)
        length = torch.where(newly, torch.full_like(length, ply + 1), length)
        over |= done
        mask = env.legal_mask
        if ply % 16 == 15 and bool(over.all()):               # one host sync every 16 plies
            break
    r = reason.cpu().tolist()
    w = winner.cpu().tolist()
    return EvaluationResult(
        games=num_games, black_wins=sum(1 for x in w if x == 0), white_wins=sum(1 for x in w if x == 1),
        draws=sum(1 for x in w if x < 0),
        reasons={REASON_NAMES[k]: r.count(k) for k in sorted(set(r))},
        avg_game_length=float(length.float().mean()))
