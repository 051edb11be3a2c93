"""Shared parity checks: drive an env backend (host-emulated kernel or the HIP library through the
C-ABI) with the golden action streams and compare every output with the golden vectors / the oracle.

Backend protocol (numpy in/out): n, reset(reset_mask=None) -> (obs, mask, n_legal); observe();
step(actions, auto_reset, check_prev) -> (obs, mask, n_legal, reward, done, reason, winner);
set_state(g, rec128)."""
import json
import os
import zlib

import numpy as np

from oracle import ko

G = os.path.join(os.path.dirname(__file__), "golden")


def ids_crc(ids):
    return zlib.crc32(np.asarray(ids, dtype="<u2").tobytes())


def obs_crc(obs):
    return zlib.crc32(np.ascontiguousarray(obs, dtype="<f4").tobytes())


def mask_ids(mask_row):
    assert set(np.unique(mask_row).tolist()) <= {0, 1}, "mask bytes must be exactly 0/1"
    return np.nonzero(mask_row)[0].tolist()


def check_traces(make_env, max_games=None, auto_reset=True):
    """Replay the golden full games as ONE batch (one env slot per golden game)."""
    z = np.load(os.path.join(G, "traces.npz"))
    off, meta = z["game_off"], z["meta"]
    n_games = len(off) - 1 if max_games is None else min(max_games, len(off) - 1)
    # group by max_moves (an env instance has a single max_moves)
    groups = {}
    for gi in range(n_games):
        groups.setdefault(int(meta[gi][1]), []).append(gi)
    init = ko.Game()
    init_obs, init_ids = init.observation(), init.legal_ids().tolist()
    total_plies = 0
    for max_moves, gis in sorted(groups.items()):
        env = make_env(len(gis), max_moves)
        obs, mask, nl = env.reset()
        for s, gi in enumerate(gis):
            assert nl[s] == int(meta[gi][3]) and ids_crc(mask_ids(mask[s])) == int(meta[gi][4])
            assert obs_crc(obs[s]) == int(meta[gi][5])
        lens = [int(off[gi + 1] - off[gi]) for gi in gis]
        rng = np.random.default_rng(7)
        for k in range(max(lens)):
            acts = np.zeros(len(gis), np.int64)
            live = []
            for s, gi in enumerate(gis):
                if k < lens[s]:
                    acts[s] = int(z["action"][off[gi] + k])
                    live.append(s)
                else:  # finished slots keep playing random legal moves of whatever game they are in now
                    ids = np.nonzero(mask[s])[0]
                    acts[s] = int(ids[rng.integers(len(ids))]) if len(ids) else 0
            obs, mask, nl, rew, done, reason, winner = env.step(acts, auto_reset=auto_reset, check_prev=True)
            if not auto_reset:
                # finished games stay put (shogi_game.py:589-593; pinned on the reference by sequences_deep.json "after"):
                # same verdict and observation as on their last ply, reward +1 exactly when somebody won
                for s, gi in enumerate(gis):
                    if k >= lens[s]:
                        i = off[gi + 1] - 1
                        ctx = (gi, k, "finished")
                        assert bool(done[s]) and int(reason[s]) == int(z["reason"][i]) and int(winner[s]) == int(z["winner"][i]), ctx
                        assert rew[s] == (1.0 if int(z["winner"][i]) >= 0 else 0.0), ctx
                        assert obs_crc(obs[s]) == int(z["obs_crc"][i]), ctx
                        total_plies += 1
            for s in live:
                i = off[gis[s]] + k
                ctx = (gis[s], k)
                assert rew[s] == float(z["reward"][i]) and bool(done[s]) == bool(z["done"][i]), ctx
                assert int(reason[s]) == int(z["reason"][i]) and int(winner[s]) == int(z["winner"][i]), ctx
                if done[s] and auto_reset:
                    assert np.array_equal(obs[s], init_obs) and mask_ids(mask[s]) == init_ids and nl[s] == 30, ctx
                else:
                    assert obs_crc(obs[s]) == int(z["obs_crc"][i]), ctx
                    if not done[s]:
                        ids = mask_ids(mask[s])
                        assert nl[s] == len(ids) == int(z["n_legal"][i]), ctx
                        assert ids_crc(ids) == int(z["legal_crc"][i]), ctx
                total_plies += 1
    return total_plies


def check_positions(make_env):
    recs = [r for r in json.load(open(os.path.join(G, "positions.json"))) if r["valid"]]
    env = make_env(len(recs), 500)
    env.reset()
    games = []
    for s, rec in enumerate(recs):
        g = ko.Game(max_moves=rec["max_moves"], sfen=rec["sfen"])
        games.append(g)
        st = g.export_state()
        st[98:101] = 0          # drop the oracle's verdict: the device must evaluate the loaded position itself
        env.set_state(s, st)
    # from_sfen semantics: the loaded position's termination is evaluated on the device (shogi_game.py:331-343)
    obs, mask, nl, done, reason, winner = env.evaluate_termination()
    for s, rec in enumerate(recs):
        assert (bool(done[s]), int(reason[s]), int(winner[s])) == (rec["game_over"], rec["reason"], rec["winner"]), rec["sfen"]
    obs, mask, nl = env.observe()
    for s, rec in enumerate(recs):
        assert mask_ids(mask[s]) == rec["legal"], rec["sfen"]
        assert nl[s] == len(rec["legal"]), rec["sfen"]
        assert obs_crc(obs[s]) == rec["obs_crc"], rec["sfen"]
    # playouts (only for positions that were not already over)
    maxlen = max(len(r.get("playout", [])) for r in recs)
    live = {s for s, r in enumerate(recs) if r.get("playout")}
    init_ids = ko.Game().legal_ids().tolist()
    checked = 0
    for k in range(maxlen):
        acts = np.zeros(len(recs), np.int64)
        cur = [s for s in live if k < len(recs[s]["playout"])]
        for s in range(len(recs)):
            if s in cur:
                acts[s] = recs[s]["playout"][k]["a"]
            else:
                ids = np.nonzero(mask[s])[0]
                acts[s] = int(ids[0]) if len(ids) else 0
        obs, mask, nl, rew, done, reason, winner = env.step(acts, auto_reset=True, check_prev=False)
        for s in cur:
            ply = recs[s]["playout"][k]
            ctx = (recs[s]["sfen"], k)
            assert (float(rew[s]), bool(done[s]), int(reason[s]), int(winner[s])) == \
                   (ply["r"], ply["done"], ply["reason"], ply["winner"]), ctx
            if not done[s]:
                ids = mask_ids(mask[s])
                assert obs_crc(obs[s]) == ply["obs_crc"], ctx
                assert len(ids) == ply["n_legal"] and ids_crc(ids) == ply["legal_crc"], ctx
            else:
                assert mask_ids(mask[s]) == init_ids, ctx
            checked += 1
    return checked


def check_sequences(make_env, deep=False):
    """scripted sequences from the reference.  deep=True: sequences_deep.json -- repetitions 70+ / 520+ plies back in the
    history (max_moves 700: the kernel's hv[1..7] registers and its `base = 512` loop) and steps on finished games."""
    seqs = json.load(open(os.path.join(G, "sequences_deep.json" if deep else "sequences.json")))
    n = 0
    for s in seqs:
        env = make_env(1, s["max_moves"])
        env.reset()
        if s["sfen"]:
            env.set_state(0, ko.Game(max_moves=s["max_moves"], sfen=s["sfen"]).export_state())
            env.observe()
        for a in s.get("actions", [])[:-1]:          # lead-in whose outputs traces.npz pins
            env.step(np.array([a]), auto_reset=False, check_prev=True)
        for k, ply in enumerate(s["plies"]):
            obs, mask, nl, rew, done, reason, winner = env.step(np.array([ply["a"]]), auto_reset=False, check_prev=True)
            ctx = (s["name"], k)
            assert (float(rew[0]), bool(done[0]), int(reason[0]), int(winner[0])) == \
                   (ply["r"], ply["done"], ply["reason"], ply["winner"]), ctx
            assert obs_crc(obs[0]) == ply["obs_crc"], ctx
            if not done[0]:
                ids = mask_ids(mask[0])
                assert len(ids) == ply["n_legal"] and ids_crc(ids) == ply["legal_crc"], ctx
            n += 1
        for k, ply in enumerate(s.get("after", [])):
            # the game is over and auto-reset is off: further steps change nothing and report the reference's early
            # return (shogi_game.py:589-593): same verdict, reward from the nominal mover's point of view
            obs, mask, nl, rew, done, reason, winner = env.step(np.array([ply["a"]]), auto_reset=False, check_prev=False)
            ctx = (s["name"], "after", k)
            assert (float(rew[0]), bool(done[0]), int(reason[0]), int(winner[0])) == \
                   (ply["r"], ply["done"], ply["reason"], ply["winner"]), ctx
            assert obs_crc(obs[0]) == ply["obs_crc"], ctx
            n += 1
    return n


def check_random_vs_oracle(make_env, n_games, n_steps, seed, max_moves=500, drop_bias=0.0):
    """Seeded random self-play: env backend vs the oracle stepped side by side (auto-reset on)."""
    env = make_env(n_games, max_moves)
    ora = ko.VecGames(n_games, max_moves)
    obs, mask, nl = env.reset()
    o_obs, o_mask, o_nl = ora.observe()
    # a backend in the compact observation format reports the exact upcast of its bf16 planes: the oracle's float32
    # planes are rounded the same way (round to nearest even) before the bit-exact comparison
    obs_round = getattr(env, "obs_round", None) or (lambda x: x)
    assert np.array_equal(obs, obs_round(o_obs)) and np.array_equal(mask, o_mask) and np.array_equal(nl, o_nl)
    rng = np.random.default_rng(seed)
    ndone = 0
    for k in range(n_steps):
        acts = np.zeros(n_games, np.int64)
        for s in range(n_games):
            ids = np.nonzero(mask[s])[0]
            if drop_bias > 0:
                d = ids[ids >= 12960]
                if len(d) and rng.random() < drop_bias:
                    ids = d
            acts[s] = int(ids[rng.integers(len(ids))])
        obs, mask, nl, rew, done, reason, winner = env.step(acts, auto_reset=True, check_prev=True)
        o = ora.step(acts, auto_reset=True)
        o = (obs_round(o[0]),) + tuple(o[1:])
        names = ["obs", "mask", "n_legal", "reward", "done", "reason", "winner"]
        for name, x, y in zip(names, (obs, mask, nl, rew, done, reason, winner), o):
            if not np.array_equal(x, y):
                bad = np.nonzero((x != y).reshape(n_games, -1).any(axis=1))[0]
                raise AssertionError("step %d: %s differs for games %s" % (k, name, bad[:8].tolist()))
        ndone += int(done.sum())
        if hasattr(env, "check_hashes"):
            env.check_hashes()
    return ndone


def check_illegal_actions(make_env):
    """Actions outside the legal set are rejected the way StepManager treats the reference's ValueError
    (step_manager.py:303-348): done with reason 5, the episode restarts (auto-reset) or the state stays untouched."""
    init = ko.Game()
    init_obs, init_ids = init.observation(), init.legal_ids().tolist()
    bad = [0, 1, 13526, -1, 13527, 10 ** 12, 12960]       # illegal ids in the start position + out-of-range ids
    env = make_env(len(bad), 500)
    env.reset()
    # first make one legal move everywhere so that a reset is observable
    obs, mask, nl, *_ = env.step(np.full(len(bad), init_ids[0], np.int64), auto_reset=True, check_prev=True)
    before_obs, before_mask = obs.copy(), mask.copy()
    obs, mask, nl, rew, done, reason, winner = env.step(np.asarray(bad, np.int64), auto_reset=True, check_prev=True)
    assert done.tolist() == [1] * len(bad) and reason.tolist() == [5] * len(bad) and rew.tolist() == [0.0] * len(bad)
    for s in range(len(bad)):
        assert np.array_equal(obs[s], init_obs) and mask_ids(mask[s]) == init_ids
    # without auto-reset the position is left as it was
    env2 = make_env(len(bad), 500)
    env2.reset()
    env2.step(np.full(len(bad), init_ids[0], np.int64), auto_reset=False, check_prev=True)
    obs, mask, nl, rew, done, reason, winner = env2.step(np.asarray(bad, np.int64), auto_reset=False, check_prev=True)
    assert reason.tolist() == [5] * len(bad)
    assert np.array_equal(obs, before_obs) and np.array_equal(mask, before_mask)
    # and the game can continue with a legal move afterwards
    ids = mask_ids(mask[0])
    obs, mask, nl, rew, done, reason, winner = env2.step(np.full(len(bad), ids[0], np.int64), auto_reset=False, check_prev=True)
    assert reason.tolist() == [0] * len(bad) and done.tolist() == [0] * len(bad)
