"""Batched evaluation games on the HIP engine -- the GPU counterpart of the game loops inside Keisei's
evaluation strategies (keisei/evaluation/strategies/single_opponent.py:170-220: `while not game.game_over:
legal = game.get_legal_moves(); mask = mapper.get_legal_mask(...); move = player.select_action(...);
game.make_move(move)`), which are a second consumer of the same env / select_action API.  N games are
played to the end at once; finished games stop consuming moves (auto-reset is off)."""
from dataclasses import dataclass

import torch

from . import ops
from ._lib import NUM_ACTIONS
from .env import VecShogiEnv


@dataclass
class EvaluationResult:
    games: int
    black_wins: int
    white_wins: int
    draws: int
    reasons: dict          # termination reason name -> count
    avg_game_length: float
    winners: object = None       # per game: -1 draw / 0 black / 1 white      (numpy int8 [games])
    reason_codes: object = None  # per game: KSH_REASON_* code                (numpy uint8 [games])
    lengths: object = None       # per game: number of plies played           (numpy int32 [games])
    actions: object = None       # with record_actions: int64 [plies, games], the action handed to every game at every ply

    @property
    def win_rate_black(self):
        return self.black_wins / max(self.games, 1)


class UniformRandomPolicy:
    """Stand-in for keisei/utils/opponents.py SimpleRandomOpponent: uniform over the legal actions."""

    def __init__(self, seed=0):
        self.seed, self._step, self._zeros = seed, 0, None

    def select_actions(self, obs, legal_mask, *, is_training=False):
        if self._zeros is None or self._zeros.shape[0] != legal_mask.shape[0] or self._zeros.device != legal_mask.device:
            self._zeros = torch.zeros((legal_mask.shape[0], NUM_ACTIONS), dtype=torch.float32, device=legal_mask.device)
        logits = self._zeros                              # legal_mask may be the byte mask or the packed bit-set
        a, lp = ops.masked_sample(logits, legal_mask, self.seed, self._step, deterministic=False)
        self._step += 1
        return a, lp, torch.zeros(legal_mask.shape[0], device=legal_mask.device)


@torch.no_grad()
def play_games(black, white, num_games, max_moves_per_game=500, device="cuda:0", deterministic=True, record_actions=False):
    """`black` / `white`: objects with select_actions(obs, legal_mask, is_training=...) (VecPPOAgent or a policy
    above).  Every game is played from the start position until it ends; returns an EvaluationResult.
    record_actions=True keeps the whole action stream (tests replay it on the oracle)."""
    from .env import REASON_NAMES
    env = VecShogiEnv(num_games, max_moves_per_game, device=device, auto_reset=False)
    obs, _ = env.reset()
    mask = env.kernel_mask
    over = torch.zeros(num_games, dtype=torch.bool, device=env.device)
    reason = torch.zeros(num_games, dtype=torch.uint8, device=env.device)
    winner = torch.full((num_games,), -1, dtype=torch.int8, device=env.device)
    length = torch.zeros(num_games, dtype=torch.int32, device=env.device)
    stream = []
    for ply in range(max_moves_per_game):
        player = black if ply % 2 == 0 else white              # all games are in lock-step: same side to move
        a, _, _ = player.select_actions(obs, mask, is_training=not deterministic)
        if record_actions:
            stream.append(a.clone())
        obs, _, done, info = env.step(a, validate=False)       # finished games ignore further actions (shogi_game.py:589-593)
        newly = done & ~over
        reason = torch.where(newly, info["reason"], reason)
        winner = torch.where(newly, info["winner"], winner)
        length = torch.where(newly, torch.full_like(length, ply + 1), length)
        over |= done
        mask = env.kernel_mask
        if ply % 16 == 15 and bool(over.all()):               # one host sync every 16 plies
            break
    r = reason.cpu().tolist()
    w = winner.cpu().tolist()
    return EvaluationResult(
        games=num_games, black_wins=sum(1 for x in w if x == 0), white_wins=sum(1 for x in w if x == 1),
        draws=sum(1 for x in w if x < 0),
        reasons={REASON_NAMES[k]: r.count(k) for k in sorted(set(r))},
        avg_game_length=float(length.float().mean()),
        winners=winner.cpu().numpy(), reason_codes=reason.cpu().numpy(), lengths=length.cpu().numpy(),
        actions=torch.stack(stream).cpu().numpy() if record_actions else None)


@dataclass
class MatchResult:
    """Agent-centric result of a match (keisei/evaluation/core/evaluation_result.py SummaryStats fields for this
    path: total_games / agent_wins / opponent_wins / draws / win_rate / avg_game_length)."""
    games: int
    agent_wins: int
    opponent_wins: int
    draws: int
    agent_sente_games: int
    avg_game_length: float
    reasons: dict

    @property
    def win_rate(self):
        return self.agent_wins / max(self.games, 1)


def evaluate_match(agent, opponent, num_games, max_moves_per_game=500, device="cuda:0", deterministic=True,
                   balance_colors=True):
    """Single-opponent evaluation (keisei/evaluation/strategies/single_opponent.py:170-330): the agent plays Sente in
    half of the games and Gote in the other half when `balance_colors` (the reference's `play_as_both_colors`), each
    half as one batched play_games call; winners are mapped back to agent / opponent (tournament.py:181-190)."""
    n_sente = (num_games + 1) // 2 if balance_colors else num_games
    halves = [(n_sente, True)] + ([(num_games - n_sente, False)] if num_games - n_sente > 0 else [])
    wins = losses = draws = 0
    reasons, length_sum = {}, 0.0
    for n, agent_is_sente in halves:
        r = play_games(agent if agent_is_sente else opponent, opponent if agent_is_sente else agent, n,
                       max_moves_per_game, device, deterministic)
        wins += r.black_wins if agent_is_sente else r.white_wins
        losses += r.white_wins if agent_is_sente else r.black_wins
        draws += r.draws
        length_sum += r.avg_game_length * n
        for k, v in r.reasons.items():
            reasons[k] = reasons.get(k, 0) + v
    return MatchResult(num_games, wins, losses, draws, n_sente, length_sum / max(num_games, 1), reasons)


def tournament(agent, opponents, num_games_per_opponent, **kw):
    """Round of matches against several opponents (keisei/evaluation/strategies/tournament.py:161-260):
    `opponents` maps a name to a policy; returns name -> MatchResult plus the aggregate under "__total__"."""
    out = {name: evaluate_match(agent, opp, num_games_per_opponent, **kw) for name, opp in opponents.items()}
    tot = MatchResult(0, 0, 0, 0, 0, 0.0, {})
    for r in out.values():
        tot.avg_game_length = (tot.avg_game_length * tot.games + r.avg_game_length * r.games) / max(tot.games + r.games, 1)
        tot.games += r.games
        tot.agent_wins += r.agent_wins
        tot.opponent_wins += r.opponent_wins
        tot.draws += r.draws
        tot.agent_sente_games += r.agent_sente_games
        for k, v in r.reasons.items():
            tot.reasons[k] = tot.reasons.get(k, 0) + v
    out["__total__"] = tot
    return out
