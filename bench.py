#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the Keisei self-play + PPO hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
        N > 1: either started by torch.distributed.run with --nproc-per-node N (the driver), or directly -- then this
        process only launches N fresh ranks, one per GPU (shogidrl_amd/launch.py), and exits with their status.

One "step" = one full pass of the hot path over one batch: a rollout of T plies for every game of the
batch (policy forward -> masked sample kernel -> env step kernel, next obs / mask written in place), the
GAE scan, and the PPO update (ppo_epochs x minibatches: forward, fused PPO-loss kernel fwd+bwd, backward,
gradient all-reduce, clip, Adam).  Workload = BASELINE.json configs[2]: 4096 batched self-play games per
GPU, ResNet 9x256 + SE 0.25 actor-critic (13.24 M parameters, random init), bf16 autocast tower, synthetic
self-play data (games start from the standard position; there is no dataset on this path).

`value` = env-steps/s of the WHOLE job (all ranks, rollout + update), i.e. the "steps per second" the
reference displays (keisei/training/training_loop_manager.py:609-616).  The same JSON line carries the
phase rates (self-play env-steps/s, PPO samples/s), the roofline of the dominant hand-written kernel
(ksh_env_step_kernel, HBM-bound) measured live with hipEvents on the launch stream, and the CPU baseline
(the oracle = port of the reference's algorithm, timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if "--graph" not in sys.argv and "--async-learner" not in sys.argv:
    # The default run contains ONE hipGraph: the minibatch-64 side figure's captured PPO update, which needs the runtime's
    # AQL-packet capture off (PPOConfig.graph_minibatch_step says why).  Nothing in the timed region uses a graph, so the
    # headline number is not touched by this; runs that replay rollout graphs (--graph, --async-learner) keep the default.
    os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ENV_BYTES_PER_STEP = 28710       # obs 14 904 + mask 13 527 + state/scalars 279
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


_REAL_STDOUT = None


def quiet_stdout():
    """From here on file descriptor 1 is stderr: whatever a library prints to stdout (RCCL writes a five-line version banner
    there when its first communicator comes up) cannot end up next to the ONE JSON line, which emit() writes to the real
    stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096, help="games per GPU")
    ap.add_argument("--rollout", type=int, default=32, help="plies per game per step (T)")
    ap.add_argument("--epochs", type=int, default=10, help="ppo_epochs (reference default 10)")
    ap.add_argument("--minibatch", type=int, default=4096, help="GPU-sized minibatch (reference default 64)")
    ap.add_argument("--depth", type=int, default=9)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--replay", type=int, default=96, help="env launches in the roofline replay")
    ap.add_argument("--compact", action="store_true",
                    help="SURVEY 8f4 formats for the whole-job loop: bit-packed masks + bf16 channels-last observations "
                         "(NOT the reference contract; the default run and the roofline replay keep float32 / byte masks)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the T-step self-play rollout from one captured hipGraph (BASELINE configs[4])")
    ap.add_argument("--async-learner", action="store_true",
                    help="BASELINE configs[4]: rollout i+1 on a second HIP stream (actor copy of the network, refreshed every "
                         "--sync-interval iterations) while learn() consumes rollout i; changes the on-policy semantics, "
                         "so it is not the default")
    ap.add_argument("--sync-interval", type=int, default=1)
    ap.add_argument("--roofline-only", action="store_true", help="only time the env kernel (quick check)")
    return ap.parse_args()


def _cpu_selfplay(threads, seconds):
    """Uniform-random legal self-play on the oracle (plain-C port of the reference's algorithm), `threads` host threads
    with 16 games each, the whole loop inside the C library (ctypes releases the GIL)."""
    import threading
    from oracle import ko
    vecs = [ko.VecGames(16, 500) for _ in range(threads)]
    ko.selfplay_random(vecs[0], 20, 1)                                   # warm the caches, measure the rate
    t0 = time.perf_counter()
    probe = ko.selfplay_random(vecs[0], 100, 2)
    rate = probe / (time.perf_counter() - t0)
    steps = max(100, int(rate * seconds / 16))                           # batched steps per thread for ~`seconds`
    done = [0] * threads

    def work(i):
        done[i] = ko.selfplay_random(vecs[i], steps, 1000 + i)
    th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return sum(done), dt


def cpu_baseline(seconds):
    """The oracle on ONE host core and on the box's CPU share (bounded samples, about `seconds` each)."""
    n1, d1 = _cpu_selfplay(1, seconds)
    share = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    nk, dk = _cpu_selfplay(share, seconds * 0.5)
    what = ("random-legal self-play env steps (16 games per thread from the start position, max_moves 500, auto-reset; "
            "step + 46-plane obs + 13527-byte mask per step, loop inside the C port)")
    return {"value": n1 / d1, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d %s in %.1f s on one host core of %d; the Python reference itself measured 18-34 "
                      "env-steps/s/core (BASELINE.md)" % (n1, what, d1, os.cpu_count() or 0),
            "all_cores": {"value": nk / dk, "unit": "env-steps/s", "cores": share, "kind": "port",
                          "sample": "%d env steps in %.1f s on %d threads" % (nk, dk, share)}}


def minibatch64_rate(agent, buf, dev, updates=64, note=lambda msg: None):
    """PPO update rate at the REFERENCE's minibatch size (default_config.yaml:88 `minibatch_size: 64`), on a bounded
    sample: `updates` optimizer updates of 64 samples each drawn from the rollout buffer (one epoch over 64 x updates
    samples), timed with hipEvents after the headline measurement.  Reported next to the GPU-sized minibatch of the
    headline number ("default-64 run also reported for parity")."""
    from shogidrl_amd.buffer import VecExperienceBuffer
    from shogidrl_amd.env import DeviceTimer
    n = 64 * updates
    small = VecExperienceBuffer(1, n, buf.gamma, buf.lambda_gae, dev)
    flat = lambda x: x.reshape(-1, *x.shape[2:])[:n]
    small.obs[0].copy_(flat(buf.obs[:buf.T]))
    small.legal_masks[0].copy_(flat(buf.legal_masks[:buf.T]))
    small.legal_bits[0].copy_(flat(buf.legal_bits[:buf.T]))
    for name in ("actions", "rewards", "log_probs", "values", "dones", "advantages", "returns"):
        getattr(small, name)[0].copy_(flat(getattr(buf, name)))
    small._advantages_computed = True
    saved = (agent.config.minibatch_size, agent.config.ppo_epochs)
    agent.config.minibatch_size, agent.config.ppo_epochs = 64, 1
    try:
        agent.learn(small)                       # warm-up (MIOpen picks its kernels for batch 64)
        tm = DeviceTimer()
        torch.cuda.synchronize()
        tm.start()
        agent.learn(small)
        tm.stop()
        ms = tm.elapsed_ms()
        note("minibatch-64: %d eager updates in %.1f ms" % (updates, ms))
    finally:
        agent.config.minibatch_size, agent.config.ppo_epochs = saved
    out = {"value": n / (ms * 1e-3), "unit": "PPO samples/s", "minibatch_size": 64, "updates": updates,
           "ms_per_update": ms / updates,
           "sample": "%d optimizer updates of 64 samples (one epoch over %d transitions of the rollout buffer), after the "
                     "timed region" % (updates, n)}
    # the same updates replayed from one captured hipGraph (PPOConfig.graph_minibatch_step): at this size the loop is
    # launch-bound, which is what graphs are for
    try:
        import copy
        import dataclasses
        from shogidrl_amd.agent import UPDATE_GRAPH_ENV, VecPPOAgent
        if os.environ.get(UPDATE_GRAPH_ENV) != "0":
            raise RuntimeError("not measured: %s=0 is not set in this run" % UPDATE_GRAPH_ENV)
        gcfg = dataclasses.replace(agent.config, minibatch_size=64, ppo_epochs=1, graph_minibatch_step=True)
        ga = VecPPOAgent(copy.deepcopy(agent.model), gcfg, dev)
        ga.learn(small)                          # eager warm-up updates + capture
        torch.cuda.synchronize()
        note("minibatch-64: update graph captured")
        tm.start()
        ga.learn(small)
        tm.stop()
        gms = tm.elapsed_ms()
        out["hipgraph_replay"] = {"value": n / (gms * 1e-3), "ms_per_update": gms / updates}
        del ga
    except Exception as e:                       # a side figure: never let it take the headline measurement down
        out["hipgraph_replay"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def fused_head_times(agent, buf, dev, reps=20):
    """Device time of the fused policy head on one real minibatch of the rollout buffer: sampler, and loss
    forward + backward (all of its kernels: row kernel, bit-matrix transpose + statistics, scan / work / combine), between
    hipEvents on the launch stream, after the timed region."""
    from shogidrl_amd import ops
    from shogidrl_amd.env import DeviceTimer
    if not getattr(agent, "fused_head", False):
        return None
    n = buf.N
    with torch.no_grad():
        agent.model.train()
        feat, values = agent._features(buf.obs[0])
    lin = agent.model.policy_linear
    W, b = lin.weight.detach(), (None if lin.bias is None else lin.bias.detach())
    bits, act = buf.legal_bits[0], buf.actions[0]
    lp, adv, ret = buf.log_probs[0], buf.advantages[0], buf.returns[0]
    tm = DeviceTimer()
    out = {"rows": int(n), "K": int(feat.shape[1]),
           "mean_legal_actions": float(ops.unpack_mask(bits).float().sum(1).mean())}
    for name, fn in (("sample_us", lambda: ops.policy_head_sample(feat, W, b, bits, 7, 0)),
                     ("loss_fwd_bwd_us", lambda: ops.policy_head_loss_grads(feat, W, b, values, bits, act, lp, adv, ret, None, 0.2,
                                                                            0.5, 0.01, False, dense_rows_possible=False))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        tm.start()
        for _ in range(reps):
            fn()
        tm.stop()
        out[name] = round(tm.elapsed_ms() / reps * 1e3, 1)
    return out


def env_roofline(dev, n_games, replay_steps, seed):
    """Average device time of ksh_env_step_kernel from `replay_steps` back-to-back launches between two
    hipEvents on the launch stream: a recorded self-play action stream is replayed on a second env."""
    from shogidrl_amd import ops
    from shogidrl_amd.env import DeviceTimer, VecShogiEnv
    live = VecShogiEnv(n_games, 500, device=dev)
    live.reset()
    logits = torch.zeros(n_games, 13527, device=dev)
    warm = 48                                   # desynchronise the games first (opening plies are the cheap ones)
    acts = []
    for t in range(warm + replay_steps):
        a, _ = ops.masked_sample(logits, live.kernel_mask, seed, t)
        live.step(a, validate=False)
        acts.append(a)
    rep = VecShogiEnv(n_games, 500, device=dev)
    rep.reset()
    for t in range(warm):
        rep.step(acts[t], validate=False)
    tm = DeviceTimer()
    torch.cuda.synchronize()
    tm.start()
    for t in range(warm, warm + replay_steps):
        rep.step(acts[t], validate=False)
    tm.stop()
    ms = tm.elapsed_ms() / replay_steps
    same = bool(torch.equal(rep.obs, live.obs) and torch.equal(rep.legal_mask, live.legal_mask))
    del live, rep
    # context for the fraction below: what a plain write-only fill of the same number of bytes takes on this GPU
    # (the kernel's traffic is 96 % stores; torch's vectorised fill kernel, same event timing, back-to-back launches)
    scratch = torch.empty(n_games * ENV_BYTES_PER_STEP // 4, dtype=torch.float32, device=dev)
    for _ in range(3):
        scratch.fill_(1.0)
    torch.cuda.synchronize()
    tm.start()
    for _ in range(replay_steps):
        scratch.fill_(1.0)
    tm.stop()
    fill_us = tm.elapsed_ms() / replay_steps * 1e3
    del scratch
    achieved = n_games * ENV_BYTES_PER_STEP / (ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes (tools/env_traffic.py): only when they were taken on THIS kernel source
    traffic, traffic_note = None, "no PMC measurement for this size"
    tfile = os.path.join(ROOT, "profiles", "env_kernel_hbm_traffic.json")
    if os.path.exists(tfile):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from env_traffic import env_source_hash
            t = json.load(open(tfile))
            if int(t.get("n_games", -1)) != n_games:
                traffic_note = "PMC file is for %s games" % t.get("n_games")
            elif t.get("env_source_hash") != env_source_hash():
                traffic_note = "stale: PMC passes were taken on another version of the env kernel (tools/env_traffic.py regenerates)"
            else:
                traffic, traffic_note = t.get("bytes_per_launch"), "rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes (profiles/env_kernel_hbm_traffic.json)"
        except (ValueError, OSError, ImportError):
            traffic = None
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
            "kernel": "ksh_env_step_kernel",
            "avg_launch_us": round(ms * 1e3, 2), "launches": replay_steps,
            "algorithmic_bytes_per_launch": n_games * ENV_BYTES_PER_STEP, "replay_deterministic": same,
            "fill_same_bytes_us": round(fill_us, 2)}


def async_main(args, rank, world, dev, agent, env, buf, cfg, fmt, barrier, note):
    """--async-learner: every timed step is one learner iteration with the NEXT rollout in flight on the side stream.
    Reports the same whole-job metric; no roofline / cpu_baseline objects (the default run carries them)."""
    from shogidrl_amd.async_learner import OverlappedSelfPlayLearner
    from shogidrl_amd.buffer import VecExperienceBuffer
    buf2 = VecExperienceBuffer(args.rollout, args.games, cfg.gamma, cfg.lambda_gae, dev, **fmt)
    ov = OverlappedSelfPlayLearner(env, agent, [buf, buf2], sync_interval=args.sync_interval, graphed=args.graph)
    for i in range(max(args.warmup, 3 if args.graph else 1)):     # (graphs: eager pass + capture for both buffers)
        ov.step()
        ov.synchronize()
        note("warmup %d done" % i)
    barrier()
    t0 = time.perf_counter()
    metrics = {}
    for _ in range(args.steps):
        metrics = ov.step()
    ov.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():       # (more than one rank, or a one-rank rehearsal group: KSH_DIST_REHEARSAL=1)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.cpu()[0])
    transitions = world * args.steps * args.rollout * args.games
    if rank == 0:
        emit({
            "metric": "env-steps/sec self-play + PPO samples/sec, 4096 games, 1/2/4/8 MI355X",
            "value": transitions / elapsed, "unit": "env-steps/s", "n_gpus": world,
            "world_size": dist.get_world_size() if dist.is_initialized() else 1,
            "backend": dist.get_backend() if dist.is_initialized() else "none (single rank)",
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4] schedule on configs[2] sizes: %d games per GPU, ResNet %dx%d+SE0.25, "
                                   "rollout T=%d on a side stream (%s) overlapping PPO %d epochs x minibatch %d; actor weights "
                                   "refreshed every %d iteration(s)" % (args.games, args.depth, args.width, args.rollout,
                                                                        "hipGraph replay" if args.graph else "eager launches",
                                                                        args.epochs, args.minibatch, args.sync_interval),
                       "games_per_gpu": args.games, "rollout_len": args.rollout, "async_learner": True,
                       "sync_interval": args.sync_interval, "parallelism": "dp%d" % world},
            "rollout_weight_versions": ov.rollout_versions[-4:], "learner_version": ov.learner_version,
            "ppo_metrics": metrics, "episodes": ov.episode_stats()})
    if dist.is_initialized():       # (more than one rank, or a one-rank rehearsal group: KSH_DIST_REHEARSAL=1)
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not started by torch.distributed.run: be the launcher.  This parent makes no GPU call; it starts --gpus fresh
        # ranks (one per GPU), rank 0 prints the JSON line, and any failing rank fails the whole run.
        from shogidrl_amd import launch
        sys.exit(launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    from shogidrl_amd import parallel
    quiet_stdout()
    rank, local_rank, world = parallel.init_from_env()
    if world != args.gpus:
        # the driver passes the same N to the launcher and to --gpus; anything else is a mis-launch, not a note
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node == --gpus)" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if args.roofline_only:
        emit(env_roofline(dev, args.games, args.replay, 42))
        return
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import VecExperienceBuffer
    from shogidrl_amd.env import DeviceTimer, VecShogiEnv
    from shogidrl_amd.models import ResTowerActorCritic
    from shogidrl_amd.rollout import SelfPlayRunner

    torch.backends.cudnn.benchmark = True        # MIOpen exhaustive find during the warm-up: tower step 47.6 -> 42.1 ms
    torch.manual_seed(42)                        # same init on every rank (then kept in sync by the all-reduce)
    model = ResTowerActorCritic(46, 13527, args.depth, args.width, 0.25)
    cfg = PPOConfig(ppo_epochs=args.epochs, minibatch_size=args.minibatch, mixed_precision=True, channels_last=True,
                    seed=42 + rank)
    agent = VecPPOAgent(model, cfg, dev)
    if dist.is_initialized():       # (more than one rank, or a one-rank rehearsal group: KSH_DIST_REHEARSAL=1)
        parallel.broadcast_parameters(agent.model, src=0)     # identical seeds already give identical weights; make it certain
    fmt = dict(packed_masks=True, obs_format="bf16_nhwc") if args.compact else {}
    env = VecShogiEnv(args.games, 500, device=dev, **fmt)
    buf = VecExperienceBuffer(args.rollout, args.games, cfg.gamma, cfg.lambda_gae, dev, **fmt)
    runner = None if args.async_learner else SelfPlayRunner(env, agent, buf)

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():       # (more than one rank, or a one-rank rehearsal group: KSH_DIST_REHEARSAL=1)
            dist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print("[bench %7.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    if args.async_learner:
        return async_main(args, rank, world, dev, agent, env, buf, cfg, fmt, barrier, note)
    collect = runner.collect
    if args.graph:
        runner.collect_graphed()                 # eager (allocations, MIOpen solver choice) ...
        runner.buffer.clear()
        runner.collect_graphed()                 # ... capture + first replay; from here on replays only
        runner.buffer.clear()
        collect = runner.collect_graphed
    for i in range(args.warmup):
        collect()
        torch.cuda.synchronize()
        note("warmup %d: rollout done" % i)
        runner.update()
        note("warmup %d: update done" % i)
    t_roll = DeviceTimer()
    t_upd = DeviceTimer()
    roll_ms = upd_ms = 0.0
    metrics = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t_roll.start()
        collect()
        t_roll.stop()
        t_upd.start()
        metrics = runner.update()
        t_upd.stop()
        roll_ms += t_roll.elapsed_ms()
        upd_ms += t_upd.elapsed_ms()
        note("timed step done (rollout %.2f s, update %.2f s so far)" % (roll_ms * 1e-3, upd_ms * 1e-3))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():       # (more than one rank, or a one-rank rehearsal group: KSH_DIST_REHEARSAL=1)
        t = torch.tensor([elapsed, roll_ms, upd_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, roll_ms, upd_ms = (float(x) for x in t.cpu())
    transitions = world * args.steps * args.rollout * args.games
    out = None
    if rank == 0:
        roof = env_roofline(dev, args.games, args.replay, 42)
        torch.cuda.synchronize()
        note("env kernel replay done (%.2f us per launch)" % roof["avg_launch_us"])
        out = {
            "metric": "env-steps/sec self-play + PPO samples/sec, 4096 games, 1/2/4/8 MI355X",
            "value": transitions / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "world_size": dist.get_world_size() if dist.is_initialized() else 1,      # as torch.distributed sees it
            "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if dist.is_initialized() else "none (single rank)",
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: %d batched self-play games per GPU, ResNet %dx%d+SE0.25 actor-critic "
                            "(random init), bf16 autocast tower / fp32 heads, rollout T=%d, GAE, PPO %d epochs x "
                            "minibatch %d, Adam" % (args.games, args.depth, args.width, args.rollout, args.epochs, args.minibatch),
                "games_per_gpu": args.games, "rollout_len": args.rollout, "ppo_epochs": args.epochs,
                "minibatch_size": args.minibatch, "rollout": "hipGraph replay" if args.graph else "eager launches",
                "parallelism": "dp%d (games sharded, flat-gradient all-reduce)" % world,
                "env_dtype": ("u32 bitboards -> bf16 channels-last obs / bit-packed mask (compact, SURVEY 8f4)" if args.compact
                              else "u32 bitboards -> f32 obs / u8 mask"),
            },
            "selfplay_env_steps_per_s": transitions / (roll_ms * 1e-3),
            "ppo_samples_per_s": transitions * args.epochs / (upd_ms * 1e-3),
            "env_kernel_env_steps_per_s": args.games / (roof["avg_launch_us"] * 1e-6),
            "roofline": roof,
            "ppo_metrics": metrics,
            "episodes": runner.episode_stats(),
        }
        try:
            out["fused_policy_head"] = fused_head_times(agent, buf, dev)
        except Exception as e:                   # a side figure: never let it take the headline measurement down
            out["fused_policy_head"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["ppo_reference_minibatch64"] = minibatch64_rate(agent, buf, dev, note=note)
            except Exception as e:               # a side figure (needs 4096 transitions in the buffer): never the headline's problem
                out["ppo_reference_minibatch64"] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.synchronize()
            note("minibatch-64 sample done")
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
            note("cpu baseline done")
    if dist.is_initialized():       # (more than one rank, or a one-rank rehearsal group: KSH_DIST_REHEARSAL=1)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(out)


if __name__ == "__main__":
    main()
