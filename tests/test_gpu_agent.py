"""Host side of the path on the MI355X: reference-API mirrors (ExperienceBuffer), VecPPOAgent.learn against the
reference's PPOAgent.learn (tests/golden/learn_e2e.npz), and the batched rollout driver."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
A = 13527
DEV = "cuda:0"


def test_experience_buffer_mirrors_reference_contract():
    # mirrors /root/reference/tests/core/test_experience_buffer.py (structure + the terminal-step known answer)
    from shogidrl_amd.buffer import ExperienceBuffer
    buf = ExperienceBuffer(buffer_size=4, gamma=0.99, lambda_gae=0.95, device=DEV)
    assert len(buf) == 0 and buf.capacity() == 4
    with pytest.warns(UserWarning):
        assert buf.get_batch() == {}
    obs = torch.randn(46, 9, 9)
    mask = torch.zeros(A, dtype=torch.bool)
    mask[:5] = True
    buf.add(obs, 1, 2.0, -0.5, 1.5, True, mask)
    with pytest.raises(RuntimeError):
        buf.get_batch()                                     # GAE not computed yet (experience_buffer.py:161-164)
    buf.compute_advantages_and_returns(0.7)
    b = buf.get_batch()
    assert float(b["advantages"][0]) == 0.5 and float(b["returns"][0]) == 2.0   # test_experience_buffer.py:98-105
    assert b["obs"].shape == (1, 46, 9, 9) and b["legal_masks"].dtype == torch.bool and b["actions"].dtype == torch.int64
    for _ in range(3):
        buf.add(obs, 1, 0.0, -0.5, 0.1, False, mask)
    with pytest.warns(UserWarning):
        buf.add(obs, 1, 0.0, -0.5, 0.1, False, mask)        # full: warns, does not grow (:92-97)
    assert buf.size() == 4
    buf.clear()
    assert len(buf) == 0


def _load_e2e():
    z = np.load(os.path.join(G, "learn_e2e.npz"))
    B = len(z["actions"])
    obs = np.unpackbits(z["obs_packed"], axis=1)[:, :46 * 81].reshape(B, 46, 9, 9).astype(np.float32)
    mask = np.unpackbits(z["mask_packed"], axis=1, bitorder="little")[:, :A].astype(bool)
    return z, obs, mask


@pytest.mark.parametrize("name,epochs,mb,normalize", [("one_update", 1, 8, False), ("two_epochs_mb4_norm", 2, 4, True)])
def test_learn_matches_reference_ppo_agent(name, epochs, mb, normalize):
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import ExperienceBuffer
    from shogidrl_amd.models import ResTowerActorCritic
    z, obs, mask = _load_e2e()
    B = obs.shape[0]
    torch.manual_seed(7)
    model = ResTowerActorCritic(46, A, tower_depth=2, tower_width=16, se_ratio=0.25)
    sums = np.asarray([float(p.detach().double().sum()) for p in model.parameters()])
    np.testing.assert_allclose(sums, z[f"{name}_init_sums"], rtol=1e-9, atol=1e-9)    # same initial weights as the reference
    cfg = PPOConfig(ppo_epochs=epochs, minibatch_size=mb, normalize_advantages=normalize, seed=42,
                    recompute_old_log_probs=False)      # the reference's semantics: old log-probs as stored
    agent = VecPPOAgent(model, cfg, DEV)
    buf = ExperienceBuffer(B, 0.99, 0.95, DEV)
    for b in range(B):
        buf.add(torch.tensor(obs[b]), int(z["actions"][b]), 0.0, float(z["old_logp"][b]), float(z["old_values"][b]), False,
                torch.tensor(mask[b]))
    buf.compute_advantages_and_returns(0.0)
    buf.advantages[:] = torch.tensor(z["adv"], device=DEV)
    buf.returns[:] = torch.tensor(z["ret"], device=DEV)
    metrics = agent.learn(buf)
    got = [metrics[k] for k in ("ppo/policy_loss", "ppo/value_loss", "ppo/entropy", "ppo/kl_divergence_approx",
                                "ppo/clip_fraction")]
    # one update: fp32 conv/GEMM on the GPU vs the reference on the CPU -> 1e-4; after several Adam steps the two
    # trajectories drift slightly (Adam's sign-like first steps amplify rounding) -> 2e-2
    tol = 1e-4 if epochs == 1 else 2e-2
    np.testing.assert_allclose(got, z[f"{name}_metrics"], rtol=tol, atol=tol)
    if epochs == 1:
        heads = torch.cat([p.grad.detach().reshape(-1)[:32] for p in agent.model.parameters()]).cpu().numpy()
        np.testing.assert_allclose(heads, z[f"{name}_grad_head"], rtol=2e-3, atol=2e-6)
        norms = np.asarray([float(p.grad.detach().norm()) for p in agent.model.parameters()])
        np.testing.assert_allclose(norms, z[f"{name}_grad_norms"], rtol=2e-3, atol=1e-6)
        assert abs(sum(n * n for n in norms) ** 0.5 - float(z[f"{name}_last_gradient_norm"])) < 1e-4   # clipped to 0.5


def test_rollout_runner_matches_oracle_and_learns():
    from oracle import ko
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import VecExperienceBuffer
    from shogidrl_amd.env import VecShogiEnv
    from shogidrl_amd.models import TinyActorCritic
    from shogidrl_amd.rollout import SelfPlayRunner
    torch.manual_seed(0)
    N, T = 48, 12
    cfg = PPOConfig(ppo_epochs=2, minibatch_size=96, seed=3)
    agent = VecPPOAgent(TinyActorCritic(), cfg, DEV)
    env = VecShogiEnv(N, 60, device=DEV)
    buf = VecExperienceBuffer(T, N, cfg.gamma, cfg.lambda_gae, DEV)
    run = SelfPlayRunner(env, agent, buf)
    ora = ko.VecGames(N, 60)
    for it in range(6):
        run.collect()
        # the stored trajectory replays bit-exactly on the oracle: obs_t, mask_t -> action -> reward, done, obs_{t+1}
        for t in range(T):
            o_obs, o_mask, _ = ora.observe() if (it == 0 and t == 0) else (nobs, nmask, None)
            assert np.array_equal(buf.obs[t].cpu().numpy(), o_obs)
            assert np.array_equal(buf.legal_masks[t].cpu().numpy().astype(np.uint8), o_mask)
            a = buf.actions[t].cpu().numpy()
            assert buf.legal_masks[t].gather(1, buf.actions[t][:, None]).all()
            nobs, nmask, _, rew, done, _, _ = ora.step(a, auto_reset=True)
            assert np.array_equal(buf.rewards[t].cpu().numpy(), rew)
            assert np.array_equal(buf.dones[t].cpu().numpy().astype(np.uint8), done)
        oadv, oret = ko.gae(buf.rewards.cpu().numpy(), buf.values.cpu().numpy(), buf.dones.cpu().numpy().astype(np.uint8),
                            agent.get_values(buf.obs[T]).cpu().numpy(), cfg.gamma, cfg.lambda_gae)
        assert np.array_equal(buf.advantages.cpu().numpy(), oadv) and np.array_equal(buf.returns.cpu().numpy(), oret)
        before = [p.detach().clone() for p in agent.model.parameters()]
        m = run.update()
        assert set(m) == {"ppo/policy_loss", "ppo/value_loss", "ppo/entropy", "ppo/kl_divergence_approx",
                          "ppo/clip_fraction", "ppo/learning_rate"}
        assert all(np.isfinite(v) for v in m.values()) and m["ppo/entropy"] < 0
        assert any(not torch.equal(a_, b_.detach()) for a_, b_ in zip(before, agent.model.parameters()))
    st = run.episode_stats()
    assert st["total_episodes_completed"] >= N and run.global_timestep == 6 * T * N   # max_moves 60 -> every game ended at least once
    assert st["black_wins"] + st["white_wins"] + st["draws"] == st["total_episodes_completed"]


def test_compact_formats_run_the_same_rollout_and_update():
    """with packed masks + bf16 channels-last observations the whole loop (rollout, GAE, PPO update) makes
    the same decisions as with the reference formats: under bf16 autocast the stem convolution reads identical bits
    either way, so actions, rewards and dones are identical and the learning metrics agree to fp tolerance."""
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import VecExperienceBuffer
    from shogidrl_amd.env import VecShogiEnv
    from shogidrl_amd.models import TinyActorCritic
    from shogidrl_amd.rollout import SelfPlayRunner
    N, T = 64, 8
    runs = []
    for fmt in ({}, dict(packed_masks=True, obs_format="bf16_nhwc")):
        torch.manual_seed(0)
        cfg = PPOConfig(ppo_epochs=1, minibatch_size=256, seed=3, mixed_precision=True, channels_last=True)
        agent = VecPPOAgent(TinyActorCritic(), cfg, DEV)
        run = SelfPlayRunner(VecShogiEnv(N, 80, device=DEV, **fmt), agent,
                             VecExperienceBuffer(T, N, cfg.gamma, cfg.lambda_gae, DEV, **fmt))
        run.collect()
        buf = run.buffer
        snap = dict(actions=buf.actions.clone(), rewards=buf.rewards.clone(), dones=buf.dones.clone(),
                    values=buf.values.clone(), logp=buf.log_probs.clone(), adv=buf.advantages.clone(),
                    obs=buf.obs[:T].float().contiguous().clone())
        snap["metrics"] = run.update()
        runs.append(snap)
    a, b = runs
    assert b["obs"].dtype == torch.float32 and torch.equal(b["obs"], a["obs"].to(torch.bfloat16).float())
    assert torch.equal(a["actions"], b["actions"]) and torch.equal(a["rewards"], b["rewards"]) and torch.equal(a["dones"], b["dones"])
    assert torch.allclose(a["values"], b["values"], atol=1e-6) and torch.allclose(a["logp"], b["logp"], atol=1e-5)
    assert torch.allclose(a["adv"], b["adv"], atol=1e-5)
    for k in ("ppo/policy_loss", "ppo/value_loss", "ppo/entropy", "ppo/kl_divergence_approx", "ppo/clip_fraction"):
        assert abs(a["metrics"][k] - b["metrics"][k]) < 1e-4, k


def test_graph_captured_rollout_replays_the_eager_rollout():
    """BASELINE configs[4]: the whole T-step self-play rollout (tower forward in BatchNorm train mode, sampler with a
    device-side Philox counter, env kernel, bookkeeping, GAE) captured in one hipGraph produces, replay after replay,
    exactly the trajectories of the eager loop."""
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import VecExperienceBuffer
    from shogidrl_amd.env import VecShogiEnv
    from shogidrl_amd.models import ResTowerActorCritic
    from shogidrl_amd.rollout import SelfPlayRunner
    N, T = 96, 6

    def make():
        torch.manual_seed(0)
        cfg = PPOConfig(seed=11, mixed_precision=True, channels_last=True)
        agent = VecPPOAgent(ResTowerActorCritic(46, 13527, 2, 32, 0.25), cfg, DEV)
        return SelfPlayRunner(VecShogiEnv(N, 16, device=DEV), agent, VecExperienceBuffer(T, N, cfg.gamma, cfg.lambda_gae, DEV))
    eager, graphed = make(), make()
    for it in range(4):                       # graphed: eager pass, capture + replay, replay, replay
        eager.collect()
        graphed.collect_graphed()
        a, b = eager.buffer, graphed.buffer
        for name in ("obs", "legal_masks", "actions", "rewards", "dones", "log_probs", "values", "advantages", "returns"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (it, name)
        assert torch.equal(eager.episodes, graphed.episodes) and eager.global_timestep == graphed.global_timestep
        for (k, x), (_, y) in zip(eager.agent.model.state_dict().items(), graphed.agent.model.state_dict().items()):
            assert torch.equal(x, y), (it, k)                         # BatchNorm running statistics advance identically
        assert graphed.buffer.get_batch()["obs"].shape[0] == T * N      # usable by learn() after a replay
        eager.buffer.clear()
        graphed.buffer.clear()
    assert graphed._graph is not None and int(graphed.episodes[0]) > 0
    with pytest.raises(RuntimeError):
        graphed.collect()


def test_reference_style_single_game_calls():
    from shogidrl_amd.agent import VecPPOAgent
    from shogidrl_amd.env import VecShogiEnv
    from shogidrl_amd.models import TinyActorCritic
    agent = VecPPOAgent(TinyActorCritic(), device=DEV)
    env = VecShogiEnv(1, 500, device=DEV)
    obs, mask = env.reset()
    move, idx, lp, v = agent.select_action(obs[0].cpu().numpy(), mask[0], is_training=True)
    assert bool(mask[0, idx]) and move == agent.policy_output_mapper.policy_index_to_shogi_move(idx)
    assert lp <= 0 and isinstance(v, float) and isinstance(agent.get_value(obs[0].cpu().numpy()), float)
    move2, idx2, _, _ = agent.select_action(obs[0].cpu().numpy(), mask[0], is_training=False)
    assert bool(mask[0, idx2])


def test_single_game_facade_matches_positions_fixture():
    """ShogiGame facade (from_sfen / get_legal_moves / make_move / to_sfen_string) against the golden positions."""
    import json
    from shogidrl_amd.game import ShogiGame, parse_sfen, state_to_sfen
    from shogidrl_amd.mapper import PolicyOutputMapper
    from tests.envcheck import obs_crc
    mp = PolicyOutputMapper()
    recs = json.load(open(os.path.join(G, "positions.json")))
    n = 0
    for rec in recs[:70]:
        if not rec["valid"]:
            if not rec.get("error", "").startswith("unmappable"):
                with pytest.raises(ValueError):
                    parse_sfen(rec["sfen"])
            continue
        g = ShogiGame.from_sfen(rec["sfen"])
        assert g.game_over == rec["game_over"], rec["sfen"]
        if rec["game_over"]:
            assert REASONS[g.termination_reason] == rec["reason"] and (g.winner if g.winner is not None else -1) == rec["winner"]
        assert sorted(mp.shogi_move_to_policy_index(m) for m in g.get_legal_moves()) == rec["legal"], rec["sfen"]
        assert obs_crc(g.get_observation()) == rec["obs_crc"]
        assert g.current_player == rec["side"] and g.move_count == rec["move_count"]
        assert state_to_sfen(parse_sfen(rec["sfen"])).split()[0] == rec["sfen"].split()[0]
        for ply in rec.get("playout", [])[:6]:
            obs, r, d, info = g.make_move(mp.policy_index_to_shogi_move(ply["a"]))
            assert (r, d, REASONS[info["reason"]]) == (ply["r"], ply["done"], ply["reason"])
            assert obs_crc(obs) == ply["obs_crc"]
            n += 1
    g = ShogiGame()
    assert len(g.get_legal_moves()) == 30 and g.to_sfen_string() == "lnsgkgsnl/1r5b1/ppppppppp/9/9/9/PPPPPPPPP/1B5R1/LNSGKGSNL b - 1"
    with pytest.raises(ValueError):
        g.make_move((0, 0, 5, 5, False))            # not a legal move: the reference raises ValueError too
    assert n > 100


def test_single_game_facade_kif_and_text_export(tmp_path):
    """game_to_kif / to_string on the HIP engine against the reference's text for the same games (kif.json)."""
    import datetime
    import json
    from shogidrl_amd.game import ShogiGame, game_to_kif
    from shogidrl_amd.mapper import PolicyOutputMapper
    mp = PolicyOutputMapper()
    today = datetime.date.today().strftime("%Y/%m/%d")
    cases = {c["name"]: c for c in json.load(open(os.path.join(G, "kif.json"), encoding="utf-8"))}
    for name in ("random_seed5_max60", "midgame_sfen"):
        c = cases[name]
        g = ShogiGame.from_sfen(c["sfen"], c["max_moves"]) if c["sfen"] else ShogiGame(c["max_moves"])
        for ply in range(len(c["actions"]) + 1):
            snap = c["snaps"].get(str(ply))
            if snap is not None:
                assert g.to_string() == snap["text"] and g.to_sfen_string() == snap["sfen"]
                assert game_to_kif(g, **snap["kwargs"]) == snap["kif"].replace("{DATE}", today)
            if ply < len(c["actions"]):
                g.make_move(mp.policy_index_to_shogi_move(c["actions"][ply]))
        assert g.game_over == c["game_over"] and len(g.move_history) == len(c["actions"])
        out = tmp_path / (name + ".kif")
        assert game_to_kif(g, filename=str(out)) is None
        assert out.read_text(encoding="utf-8").endswith("*EOF")


REASONS = {None: 0, "Game ongoing": 0, "Tsumi": 1, "stalemate": 2, "Max moves reached": 3, "Sennichite": 4}


def test_batched_evaluation_games_finish_with_consistent_counts():
    from shogidrl_amd.evaluation import UniformRandomPolicy, play_games
    res = play_games(UniformRandomPolicy(1), UniformRandomPolicy(2), num_games=96, max_moves_per_game=120,
                     deterministic=False)
    assert res.games == 96 and res.black_wins + res.white_wins + res.draws == 96
    assert sum(res.reasons.values()) == 96 and set(res.reasons) <= {"Tsumi", "stalemate", "Max moves reached", "Sennichite"}
    assert 20 <= res.avg_game_length <= 120
    assert res.reasons.get("Tsumi", 0) == res.black_wins + res.white_wins     # only Tsumi produces a winner


def test_evaluation_loop_replays_on_the_oracle_without_auto_reset():
    """f2 parity (keisei/evaluation/strategies/single_opponent.py:170-220 game loop): the action stream play_games fed the
    env -- including the moves it kept handing to games that were already over -- is replayed on oracle games stepped
    WITHOUT auto-reset.  Winners, termination reasons and game lengths must agree game by game, and from its last ply on
    every oracle game must report exactly what the evaluation loop latched (finished games are inert)."""
    from oracle import ko
    from shogidrl_amd.evaluation import UniformRandomPolicy, play_games
    n, mm = 64, 90
    res = play_games(UniformRandomPolicy(11), UniformRandomPolicy(12), num_games=n, max_moves_per_game=mm,
                     deterministic=False, record_actions=True)
    acts = res.actions
    assert acts.shape[1] == n and acts.shape[0] >= int(res.lengths.max())
    ora = ko.VecGames(n, mm)
    first_done = np.zeros(n, np.int32)
    o_reason = np.zeros(n, np.uint8)
    o_winner = np.full(n, -1, np.int8)
    for ply in range(acts.shape[0]):
        _, _, _, rew, done, reason, winner = ora.step(acts[ply], auto_reset=False)
        newly = (done != 0) & (first_done == 0)
        first_done[newly] = ply + 1
        o_reason[newly] = reason[newly]
        o_winner[newly] = winner[newly]
        over = first_done > 0
        # inert afterwards: same verdict on every later ply, reward +1 only for a decided game (shogi_game.py:589-593)
        assert np.array_equal(reason[over], o_reason[over]) and np.array_equal(winner[over], o_winner[over])
        assert np.array_equal(rew[over & ~newly], (o_winner[over & ~newly] >= 0).astype(np.float32))
    assert (first_done > 0).all()
    assert np.array_equal(res.lengths, first_done)
    assert np.array_equal(res.reason_codes, o_reason) and np.array_equal(res.winners, o_winner)
    assert res.black_wins == int((o_winner == 0).sum()) and res.white_wins == int((o_winner == 1).sum())
    assert res.draws == int((o_winner < 0).sum())
    assert abs(res.avg_game_length - float(first_done.mean())) < 1e-4


def test_match_and_tournament_account_seats_like_the_reference():
    """evaluate_match = the two seatings of play_games mapped back to agent / opponent; tournament aggregates."""
    from shogidrl_amd.evaluation import UniformRandomPolicy, evaluate_match, play_games, tournament
    kw = dict(max_moves_per_game=100, deterministic=False)
    m = evaluate_match(UniformRandomPolicy(3), UniformRandomPolicy(4), 65, **kw)
    a, b = UniformRandomPolicy(3), UniformRandomPolicy(4)          # same seeds and call order -> same games
    first = play_games(a, b, 33, **kw)
    second = play_games(b, a, 32, **kw)
    assert (m.games, m.agent_sente_games) == (65, 33)
    assert m.agent_wins == first.black_wins + second.white_wins
    assert m.opponent_wins == first.white_wins + second.black_wins
    assert m.draws == first.draws + second.draws and m.agent_wins + m.opponent_wins + m.draws == 65
    assert abs(m.avg_game_length - (first.avg_game_length * 33 + second.avg_game_length * 32) / 65) < 1e-4
    t = tournament(UniformRandomPolicy(5), {"r6": UniformRandomPolicy(6), "r7": UniformRandomPolicy(7)}, 16, **kw)
    assert set(t) == {"r6", "r7", "__total__"} and t["__total__"].games == 32
    assert t["__total__"].agent_wins == t["r6"].agent_wins + t["r7"].agent_wins
    assert sum(t["__total__"].reasons.values()) == 32 and 0.0 <= t["__total__"].win_rate <= 1.0


def test_config0_single_game_episode_plumbing():
    """BASELINE configs[0]: one ShogiGame self-play episode with the tiny CNN through the reference-style single-game
    calls (StepManager.execute_step order, step_manager.py:98-348): legal moves -> mask -> select_action -> make_move
    -> buffer.add; then compute_advantages_and_returns + learn."""
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import ExperienceBuffer
    from shogidrl_amd.game import ShogiGame
    from shogidrl_amd.mapper import PolicyOutputMapper
    from shogidrl_amd.models import TinyActorCritic
    torch.manual_seed(1)
    mapper = PolicyOutputMapper()
    agent = VecPPOAgent(TinyActorCritic(), PPOConfig(ppo_epochs=2, minibatch_size=16, seed=42), DEV)
    game = ShogiGame(max_moves_per_game=40)
    buf = ExperienceBuffer(64, 0.99, 0.95, DEV)
    obs = game.reset()
    steps, episodes = 0, 0
    while steps < 64:
        legal = game.get_legal_moves()
        assert legal, "a live game always has legal moves"
        mask = mapper.get_legal_mask(legal, DEV)
        assert torch.equal(mask, game.get_legal_mask())                 # host-built mask == kernel-built mask
        move, idx, logp, value = agent.select_action(obs, mask, is_training=True)
        assert move in legal
        next_obs, reward, done, info = game.make_move(move)
        buf.add(torch.as_tensor(obs), idx, reward, logp, value, done, mask)
        obs = next_obs
        steps += 1
        if done:
            assert info["reason"] in ("Tsumi", "stalemate", "Max moves reached", "Sennichite")
            episodes += 1
            obs = game.reset()
    assert episodes >= 1                                               # max_moves 40 guarantees an episode end
    buf.compute_advantages_and_returns(agent.get_value(obs))
    m = agent.learn(buf)
    assert np.isfinite(list(m.values())).all() and len(buf) == 64


def test_first_minibatch_ratio_before_any_update():
    """the stored log-probs come from acting under ONE rollout step's BatchNorm batch statistics, learn()
    re-evaluates under shuffled-minibatch statistics.  With the learning rate at zero the weights never move, so the
    KL / clip fraction learn() reports IS the mismatch.  Reference semantics (default) are measured and only required to
    be finite; with recompute_old_log_probs=True the ratio must be 1 up to minibatch-composition noise."""
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import VecExperienceBuffer
    from shogidrl_amd.env import VecShogiEnv
    from shogidrl_amd.models import ResTowerActorCritic
    from shogidrl_amd.rollout import SelfPlayRunner
    N, T = 256, 8
    out = {}
    for recompute in (False, True):
        torch.manual_seed(0)
        cfg = PPOConfig(ppo_epochs=1, minibatch_size=512, learning_rate=0.0, seed=5, mixed_precision=True, channels_last=True,
                        recompute_old_log_probs=recompute)
        agent = VecPPOAgent(ResTowerActorCritic(46, A, 2, 32, 0.25), cfg, DEV)
        run = SelfPlayRunner(VecShogiEnv(N, 120, device=DEV), agent, VecExperienceBuffer(T, N, cfg.gamma, cfg.lambda_gae, DEV))
        run.collect()
        stats_before = {k: v.clone() for k, v in agent.model.state_dict().items() if "running" in k or "num_batches" in k}
        if recompute:
            b = run.buffer.get_batch()
            lp = agent._recompute_log_probs(b["obs"], b["legal_bits"], b["actions"])
            for k, v in agent.model.state_dict().items():               # the extra pass leaves BatchNorm's buffers alone
                if k in stats_before:
                    assert torch.equal(v, stats_before[k]), k
            assert torch.isfinite(lp).all()
            out["max_abs_logp_shift"] = float((lp - b["log_probs"]).abs().max())
        m = run.update()
        out[recompute] = (m["ppo/kl_divergence_approx"], m["ppo/clip_fraction"])
    print("first-minibatch mismatch (kl, clip_frac): reference semantics %r, recomputed %r; largest per-sample log-prob "
          "shift between acting and learn-time statistics %.3f" % (out[False], out[True], out["max_abs_logp_shift"]))
    assert all(np.isfinite(v) for v in out[False])
    assert abs(out[True][0]) < 2e-3 and out[True][1] < 0.01, out
    assert abs(out[True][0]) <= abs(out[False][0]) + 1e-6


def test_fused_policy_head_and_logits_path_learn_the_same_update():
    """VecPPOAgent with the fused policy head (features -> HIP kernels, no logits) against the same agent taking the logits
    path (framework GEMM + ops.masked_sample / ops.ppo_loss): same rollout decisions, same update."""
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import VecExperienceBuffer
    from shogidrl_amd.env import VecShogiEnv
    from shogidrl_amd.models import ResTowerActorCritic
    from shogidrl_amd.rollout import SelfPlayRunner
    N, T = 128, 6
    res = []
    for fused in (True, False):
        torch.manual_seed(0)
        cfg = PPOConfig(ppo_epochs=2, minibatch_size=256, seed=9, fused_policy_head=fused, recompute_old_log_probs=False)
        agent = VecPPOAgent(ResTowerActorCritic(46, A, 2, 32, 0.25), cfg, DEV)
        assert agent.fused_head == fused
        run = SelfPlayRunner(VecShogiEnv(N, 60, device=DEV), agent, VecExperienceBuffer(T, N, cfg.gamma, cfg.lambda_gae, DEV))
        run.collect()
        snap = dict(actions=run.buffer.actions.clone(), logp=run.buffer.log_probs.clone(), values=run.buffer.values.clone())
        snap["metrics"] = run.update()
        snap["params"] = torch.cat([p.detach().reshape(-1) for p in agent.model.parameters()]).clone()
        res.append(snap)
    a, b = res
    same = a["actions"] == b["actions"]                       # float32 logit noise may flip a near-tie of the Gumbel keys
    assert float(same.float().mean()) > 0.995
    if bool(same.all()):
        assert torch.allclose(a["logp"], b["logp"], atol=2e-5) and torch.allclose(a["values"], b["values"], atol=1e-6)
        for k in ("ppo/policy_loss", "ppo/value_loss", "ppo/entropy", "ppo/kl_divergence_approx", "ppo/clip_fraction"):
            assert abs(a["metrics"][k] - b["metrics"][k]) < 2e-4, k
        # Adam moves every weight by about lr per update whatever the gradient's size: compare at that scale
        assert float((a["params"] - b["params"]).abs().max()) < 4 * 3e-4 * 2
        assert float((a["params"] - b["params"]).abs().mean()) < 1e-5


def _update_graph_check(mode, packet_capture_off=True):
    """tests/helpers/update_graph_check.py in a process of its own (the runtime flag has to be set before the first HIP call)"""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None)
    if packet_capture_off:
        env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "helpers", "update_graph_check.py"), mode], env=env, capture_output=True,
                       text=True, timeout=600)
    return r, ([json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")] or [None])[-1]


def test_graph_captured_minibatch_update_equals_the_eager_loop():
    """PPOConfig.graph_minibatch_step: the per-minibatch optimizer update replayed from a hipGraph (three eager updates,
    one capture, replays from then on, across two learn() calls with fresh advantages) against the eager loop."""
    r, o = _update_graph_check("equal")
    assert r.returncode == 0 and o is not None, r.stderr[-2000:]
    assert o["graphs"] == 1 and o["captured"]
    assert o["param_diff"] < 2e-5 and o["buffer_diff"] < 1e-4 and o["metric_diff"] < 1e-4, o


def test_graph_captured_update_follows_a_changed_learning_rate():
    """the learning rate is baked into the captured fused-Adam node.  After a scheduler-style change of
    `param_groups[...]["lr"]` between two learn() calls the graphed learner must track the eager one (a stale graph would keep
    stepping with 3e-4 instead of 1e-5: 30x larger steps), and hold ONE graph, not one per rate."""
    r, o = _update_graph_check("lr")
    assert r.returncode == 0 and o is not None, r.stderr[-2000:]
    assert o["graphs"] == 1 and o["captured"]
    assert o["param_diff"] < 2e-5 and o["metric_diff"] < 1e-4, o


def test_graph_captured_update_survives_the_eager_launches_between_learn_calls():
    """bench-sized tower, minibatch 64, two learn() calls: the 64 eager forward passes that re-compute the old log-probs at
    the start of the second call are what corrupted the replays under the runtime's default packet capture (gradient norm
    inf, profiles/r02_update_graph_corruption.txt).  With the required setting the replayed learner tracks the eager one."""
    r, o = _update_graph_check("survive")
    assert r.returncode == 0 and o is not None, r.stderr[-2000:]
    g, e = o["graph"], o["eager"]
    assert g["params_finite"] and e["params_finite"]
    assert np.isfinite(g["grad_norm"]) and g["grad_norm"] < 20.0, g
    for mg, me in zip(g["metrics"], e["metrics"]):
        # two independent learners on the same data (MIOpen's atomics, Adam): same regime, not the same bits
        assert abs(mg[0] - me[0]) < 0.02 and abs(mg[1] - me[1]) < 0.01 and abs(mg[3] - me[3]) < 0.02, (mg, me)


def test_graph_captured_update_is_refused_without_the_runtime_setting():
    r, o = _update_graph_check("equal", packet_capture_off=False)
    assert r.returncode != 0 and "DEBUG_CLR_GRAPH_PACKET_CAPTURE" in r.stderr


def test_hand_made_dense_masks_in_a_vec_buffer_take_the_dense_path():
    """only buffers filled by the env kernel (SelfPlayRunner sets `masks_from_env`) may promise "no dense
    rows".  A VecExperienceBuffer filled by hand with the reference-style all-ones masks
    (tests/core/test_ppo_agent_learning.py:53-129 use torch.ones(13527, dtype=bool)) must learn through the dense path:
    finite loss, finite update -- and agree with the single-stream ExperienceBuffer holding the same rows."""
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import ExperienceBuffer, VecExperienceBuffer
    from shogidrl_amd.models import TinyActorCritic
    N = 8
    g = torch.Generator().manual_seed(5)
    obs = torch.rand(N, 46, 9, 9, generator=g)
    act = torch.randint(0, A, (N,), generator=g)
    res = []
    for vec in (True, False):
        torch.manual_seed(1)
        agent = VecPPOAgent(TinyActorCritic(), PPOConfig(ppo_epochs=2, minibatch_size=4, seed=3), DEV)
        if vec:
            buf = VecExperienceBuffer(1, N, 0.99, 0.95, DEV)
            assert buf.masks_from_env is False
            buf.obs[0].copy_(obs)
            buf.legal_masks[0].fill_(True)
            buf.legal_bits[0].copy_(__import__("shogidrl_amd").ops.pack_mask_bits(buf.legal_masks[0]))
            buf.actions[0].copy_(act)
            buf.log_probs[0].fill_(-9.5)
            buf.values[0].fill_(0.1)
            buf.rewards[0].copy_(torch.arange(N) % 2)
            buf.compute_advantages_and_returns(torch.zeros(N, device=DEV))
            assert buf.get_batch()["masks_from_env"] is False
        else:
            buf = ExperienceBuffer(N, 0.99, 0.95, DEV)
            for b in range(N):
                buf.add(obs[b], int(act[b]), float(b % 2), -9.5, 0.1, True, torch.ones(A, dtype=torch.bool))
            buf.compute_advantages_and_returns(0.0)
        m = agent.learn(buf)
        assert all(np.isfinite(v) for v in m.values()), m
        assert all(bool(torch.isfinite(p).all()) for p in agent.model.parameters())
        res.append(m)
    for k in res[0]:
        assert abs(res[0][k] - res[1][k]) < 1e-5, (k, res)
