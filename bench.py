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

Process structure (round 3: the line must survive anything that happens after the timed region).
    rank 0 = a SUPERVISOR that never touches the GPU (stdlib only).  It starts the WORKER (`--worker`, the same command in a
    fresh process: process group, warm-up, the timed steps, then the figures that ride along), which appends every finished
    part to a stage file the moment it exists -- "headline" (value, ms_per_step, ...) right after the timed region and BEFORE
    any barrier / process-group teardown, then "roofline", "cpu_baseline", the side figures, "complete".  The supervisor
    prints the ONE JSON line from whatever stages arrived: a GPU fault, an abort or a hung RCCL teardown after the headline
    cannot remove value / roofline / cpu_baseline from stdout (what did not arrive is named under "worker_exit").
    The one figure that replays a hipGraph of the PPO update (`ppo_reference_minibatch64.hipgraph_replay`) runs in a THIRD
    process (`--side update-graph`), started by the supervisor after the worker has exited, with
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 set for that process only: the timed region keeps the runtime's defaults, and no
    process that holds the headline ever captures a graph.  ranks > 0 run the worker body directly.
    `python bench.py --worker ...` runs the worker alone and prints the line itself (what to put behind `rocprofv3 --`).
"""
import argparse
import json
import os
import signal
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

METRIC = "env-steps/sec self-play + PPO samples/sec, 4096 games, 1/2/4/8 MI355X"
ENV_BYTES_PER_STEP = 28710       # obs 14 904 + mask 13 527 + state/scalars 279
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CACHE_GATHER_PEAK_GBS = 8600.0   # MI355X_MICROARCH.md "38 MB table, uniformly random rows (Infinity Cache)": 8.6 TB/s
STAGES_ENV = "KSH_BENCH_STAGES"
ABORT_ENV = "KSH_BENCH_TEST_ABORT_AT"   # TEST HOOK: the worker calls os.abort() when it reaches the named stage (tests only)

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


def parse(argv=None):
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
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="quick mode: no cpu_baseline object and no minibatch-64 side figures")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="bounded CPU sample: one core for this long, all cores for 0.6x")
    ap.add_argument("--replay", type=int, default=96, help="env launches in the roofline replay")
    ap.add_argument("--recompute-old-log-probs", action="store_true",
                    help="PPOConfig.recompute_old_log_probs=True (opt-in deviation 6 of DESIGN.md; default = the reference's stored log-probs)")
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
    ap.add_argument("--worker", action="store_true", help="(internal) run the rank body in this process")
    ap.add_argument("--side", default=None, choices=["update-graph"], help="(internal) a side figure in a process of its own")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ stage protocol
class Stages:
    """Finished parts of the result, appended to the supervisor's stage file one JSON line at a time (flushed and fsync'ed:
    what was written survives the death of this process)."""

    def __init__(self, path):
        self.path = path
        self.parts = {}

    def put(self, name, obj):
        self.parts[name] = obj
        if self.path:
            with open(self.path, "a") as f:
                f.write(json.dumps({"stage": name, "data": obj}) + "\n")
                f.flush()
                os.fsync(f.fileno())


def _test_abort(name):
    """TEST HOOK (tests/test_bench_process.py): die the way a GPU fault kills a process -- abruptly, no exception, no
    cleanup -- just before the work of stage `name` starts."""
    if os.environ.get(ABORT_ENV) == name:
        sys.stderr.write("[bench] test hook: aborting before stage %r\n" % name)
        sys.stderr.flush()
        os.abort()


def read_stages(path):
    parts, order = {}, []
    try:
        with open(path) as f:
            for line in f:
                try:
                    rec = json.loads(line)
                except ValueError:
                    continue                      # a line cut short by the writer's death
                parts[rec["stage"]] = rec["data"]
                order.append(rec["stage"])
    except OSError:
        pass
    return parts, order


def assemble(parts):
    """headline keys first, then every other stage under its own name"""
    out = dict(parts.get("headline", {}))
    for k, v in parts.items():
        if k not in ("headline", "complete"):
            out[k] = v
    return out


# ------------------------------------------------------------------------------------------------ CPU baseline (oracle)
def _cpu_selfplay(threads, seconds):
    """Uniform-random legal self-play on the oracle (plain-C port of the reference's algorithm), `threads` host threads
    with 16 games each, the whole loop inside the C library (ctypes releases the GIL).  Bounded by WALL time: every thread
    plays chunks of about 50 ms until the deadline (a fixed step count per thread took 190 s on a box whose affinity mask
    shows 256 cores but whose cgroup grants 16)."""
    import threading
    from oracle import ko
    vecs = [ko.VecGames(16, 500) for _ in range(threads)]
    ko.selfplay_random(vecs[0], 20, 1)                                   # warm the caches, measure the rate
    t0 = time.perf_counter()
    probe = ko.selfplay_random(vecs[0], 100, 2)
    rate = probe / (time.perf_counter() - t0)
    chunk = max(10, int(rate * 0.05 / 16))                               # batched steps per call
    done = [0] * threads
    deadline = [0.0]

    def work(i):
        n = k = 0
        while time.perf_counter() < deadline[0]:
            n += ko.selfplay_random(vecs[i], chunk, 1000 + i + 7919 * k)
            k += 1
        done[i] = n
    th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    deadline[0] = t0 + seconds
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return sum(done), dt


def _cgroup_cpu_limit():
    """CPUs the cgroup of this process may use at once (cpu.max quota / period), None when unlimited or unreadable"""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                return None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            q = float(txt[0])
            return None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError, IndexError):
            continue
    return None


def cpu_baseline(seconds):
    """The oracle on ONE host core and on ALL host cores this process may run on ("single-thread and
    all-host-cores, nproc stated"); bounded samples of about `seconds` and 0.6 x `seconds` of wall time."""
    n1, d1 = _cpu_selfplay(1, seconds)
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nk, dk = _cpu_selfplay(usable, seconds * 0.6)
    limit = _cgroup_cpu_limit()
    what = ("random-legal self-play env steps (16 games per thread from the start position, max_moves 500, auto-reset; "
            "step + 46-plane obs + 13527-byte mask per step, loop inside the C port)")
    like = ("like-for-like with `env_kernel_env_steps_per_s` (the env kernel alone), NOT with `value` "
            "(= env + network forward + 10 PPO epochs)")
    return {"value": n1 / d1, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d %s in %.1f s on one host core; %s; the Python reference itself measured 18-34 "
                      "env-steps/s/core (BASELINE.md)" % (n1, what, d1, like),
            "all_cores": {"value": nk / dk, "unit": "env-steps/s", "cores": usable, "kind": "port",
                          "host_cpu_count": os.cpu_count() or 0, "usable_cores": usable, "cgroup_cpu_limit": limit,
                          "sample": "%d env steps in %.1f s on %d threads = every core in this process's affinity mask "
                                    "(os.cpu_count() = %d; the cgroup grants %s CPUs at once); %s"
                                    % (nk, dk, usable, os.cpu_count() or 0, "%.1f" % limit if limit else "all of those", like)}}


# ------------------------------------------------------------------------------------------------ GPU side figures
def _small_buffer(buf, n, dev):
    """the first `n` transitions of a rollout buffer as a one-step buffer of n 'games' (minibatch-64 figures)"""
    from shogidrl_amd.buffer import VecExperienceBuffer
    small = VecExperienceBuffer(1, n, buf.gamma, buf.lambda_gae, dev)
    flat = lambda x: x.reshape(-1, *x.shape[2:])[:n]
    small.obs[0].copy_(flat(buf.obs[:buf.T]))
    small.legal_masks[0].copy_(flat(buf.legal_masks[:buf.T]))
    small.legal_bits[0].copy_(flat(buf.legal_bits[:buf.T]))
    for name in ("actions", "rewards", "log_probs", "values", "dones", "advantages", "returns"):
        getattr(small, name)[0].copy_(flat(getattr(buf, name)))
    small._advantages_computed = True
    small.masks_from_env = True
    return small


def minibatch64_rate(agent, buf, dev, updates=64, note=lambda msg: None):
    """PPO update rate at the REFERENCE's minibatch size (default_config.yaml:88 `minibatch_size: 64`), on a bounded
    sample: `updates` optimizer updates of 64 samples each drawn from the rollout buffer (one epoch over 64 x updates
    samples), eager launches, timed with hipEvents after the headline measurement.  Reported next to the GPU-sized
    minibatch of the headline number ("default-64 run also reported for parity")."""
    import torch
    from shogidrl_amd.env import DeviceTimer
    n = 64 * updates
    small = _small_buffer(buf, n, dev)
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
    return {"value": n / (ms * 1e-3), "unit": "PPO samples/s", "minibatch_size": 64, "updates": updates,
            "ms_per_update": ms / updates,
            "sample": "%d optimizer updates of 64 samples (one epoch over %d transitions of the rollout buffer), eager launches, "
                      "after the timed region; like-for-like with the Python reference's 118 PPO samples/s on 8 CPU cores "
                      "(BASELINE.md), not with `value`" % (updates, n)}


def fused_head_times(agent, buf, dev, reps=20):
    """Device time of the fused policy head on one real minibatch of the rollout buffer: sampler, and loss
    forward + backward (all of its kernels: row kernel, bit-matrix transpose + statistics, scan, work), between hipEvents
    on the launch stream, after the timed region -- with the gathered bytes as the denominator: every legal (sample, action)
    pair costs one 648-byte row of W for the logit, the same row again for dfeat, and one 648-byte feature row for dW."""
    import torch
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
    pairs = int(ops.unpack_mask(bits).sum())
    row_bytes = int(feat.shape[1]) * 4
    out = {"rows": int(n), "K": int(feat.shape[1]), "legal_pairs": pairs, "mean_legal_actions": pairs / float(n)}
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
    gathered = 3 * pairs * row_bytes
    rate = gathered / (out["loss_fwd_bwd_us"] * 1e-6) / 1e9
    out["loss_gathered_bytes"] = gathered
    out["loss_gather"] = {"achieved": round(rate, 1), "peak": CACHE_GATHER_PEAK_GBS, "unit": "GB/s",
                          "frac": round(rate / CACHE_GATHER_PEAK_GBS, 4),
                          "note": "3 x legal_pairs x %d B (W row for the logit, W row for dfeat, feature row for dW) / "
                                  "loss_fwd_bwd_us; the traffic is cache traffic (W = 8.8 MB), so the denominator is the guide's "
                                  "random-row Infinity-Cache gather rate, not HBM; four dependent launches" % row_bytes}
    srate = pairs * row_bytes / (out["sample_us"] * 1e-6) / 1e9
    out["sample_gather"] = {"achieved": round(srate, 1), "peak": CACHE_GATHER_PEAK_GBS, "unit": "GB/s",
                            "frac": round(srate / CACHE_GATHER_PEAK_GBS, 4)}
    return out


def env_roofline(dev, n_games, replay_steps, seed):
    """Average device time of ksh_env_step_kernel from `replay_steps` back-to-back launches between two
    hipEvents on the launch stream: a recorded self-play action stream is replayed on a second env."""
    import torch
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
    del logits
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
            "kernel": "ksh_env_step_kernel", "n_games": n_games,
            "avg_launch_us": round(ms * 1e3, 2), "launches": replay_steps,
            "algorithmic_bytes_per_launch": n_games * ENV_BYTES_PER_STEP, "replay_deterministic": same,
            "fill_same_bytes_us": round(fill_us, 2)}


# ------------------------------------------------------------------------------------------------ the job of one rank
class SelfPlayJob:
    """What one rank runs: env + network + buffer + runner on its GPU.  (tests/helpers/bench_rehearsal.py swaps in a CPU
    stand-in through JOB_FACTORY to rehearse everything AROUND this class -- launcher, supervisor, stage file, process
    group, collectives, teardown -- with 8 gloo ranks on a box without GPUs.)"""

    def __init__(self, args, rank, local_rank, world):
        import torch
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
        from shogidrl_amd import parallel
        from shogidrl_amd.agent import PPOConfig, VecPPOAgent
        from shogidrl_amd.buffer import VecExperienceBuffer
        from shogidrl_amd.env import VecShogiEnv
        from shogidrl_amd.models import ResTowerActorCritic
        from shogidrl_amd.rollout import SelfPlayRunner
        import torch.distributed as dist
        self.args, self.rank, self.world = args, rank, world
        torch.cuda.set_device(local_rank)
        self.device = dev = "cuda:%d" % local_rank
        torch.backends.cudnn.benchmark = True        # MIOpen exhaustive find during the warm-up: tower step 47.6 -> 42.1 ms
        torch.manual_seed(42)                        # same init on every rank (then kept in sync by the all-reduce)
        model = ResTowerActorCritic(46, 13527, args.depth, args.width, 0.25)
        self.cfg = cfg = PPOConfig(ppo_epochs=args.epochs, minibatch_size=args.minibatch, mixed_precision=True, channels_last=True,
                                   seed=42 + rank, recompute_old_log_probs=bool(args.recompute_old_log_probs))
        self.agent = VecPPOAgent(model, cfg, dev)
        if dist.is_initialized():       # (more than one rank, or a one-rank rehearsal group: KSH_DIST_REHEARSAL=1)
            parallel.broadcast_parameters(self.agent.model, src=0)   # identical seeds already give identical weights; make it certain
        self.fmt = dict(packed_masks=True, obs_format="bf16_nhwc") if args.compact else {}
        self.env = VecShogiEnv(args.games, 500, device=dev, **self.fmt)
        self.buf = VecExperienceBuffer(args.rollout, args.games, cfg.gamma, cfg.lambda_gae, dev, **self.fmt)
        self.runner = None if args.async_learner else SelfPlayRunner(self.env, self.agent, self.buf)
        self.ov = None

    def synchronize(self):
        import torch
        torch.cuda.synchronize()

    def timer(self):
        from shogidrl_amd.env import DeviceTimer
        return DeviceTimer()

    def prepare(self, note):
        """everything that must not be inside the warm-up accounting: graph capture, the asynchronous schedule's priming"""
        args = self.args
        if args.async_learner:
            from shogidrl_amd.async_learner import OverlappedSelfPlayLearner
            from shogidrl_amd.buffer import VecExperienceBuffer
            buf2 = VecExperienceBuffer(args.rollout, args.games, self.cfg.gamma, self.cfg.lambda_gae, self.device, **self.fmt)
            self.ov = OverlappedSelfPlayLearner(self.env, self.agent, [self.buf, buf2], sync_interval=args.sync_interval,
                                                graphed=args.graph)
            for i in range(3 if args.graph else 0):       # (graphs: eager pass + capture for both buffers)
                self.ov.step()
                self.ov.synchronize()
                note("graph priming step %d done" % i)
        elif args.graph:
            self.runner.collect_graphed()                 # eager (allocations, MIOpen solver choice) ...
            self.runner.buffer.clear()
            self.runner.collect_graphed()                 # ... capture + first replay; from here on replays only
            self.runner.buffer.clear()

    def collect(self):
        if self.ov is not None:
            return                                        # the overlapped learner's step() does both halves
        self.runner.collect_graphed() if self.args.graph else self.runner.collect()

    def update(self):
        if self.ov is not None:
            return self.ov.step()
        return self.runner.update()

    def finish(self):
        if self.ov is not None:
            self.ov.synchronize()

    def episode_stats(self):
        return self.ov.episode_stats() if self.ov is not None else self.runner.episode_stats()

    def workload(self):
        a = self.args
        if a.async_learner:
            return ("BASELINE configs[4] schedule on configs[2] sizes: %d games per GPU, ResNet %dx%d+SE0.25, rollout T=%d on a "
                    "side stream (%s) overlapping PPO %d epochs x minibatch %d; actor weights refreshed every %d iteration(s)"
                    % (a.games, a.depth, a.width, a.rollout, "hipGraph replay" if a.graph else "eager launches", a.epochs,
                       a.minibatch, a.sync_interval))
        return ("BASELINE configs[2]: %d batched self-play games per GPU, ResNet %dx%d+SE0.25 actor-critic (random init), bf16 "
                "autocast tower / fp32 heads, rollout T=%d, GAE, PPO %d epochs x minibatch %d, Adam"
                % (a.games, a.depth, a.width, a.rollout, a.epochs, a.minibatch))

    def extra_headline(self):
        if self.ov is not None:
            return {"rollout_weight_versions": self.ov.rollout_versions[-4:], "learner_version": self.ov.learner_version}
        return {}

    def side_figures(self):
        """[(stage name, callable)] for rank 0 after the headline stage, in the order they run: the two objects the contract
        requires first (the CPU baseline cannot fault a GPU), then the figures that ride along"""
        args, dev = self.args, self.device
        figs = [("roofline", lambda: env_roofline(dev, args.games, args.replay, 42))]
        if self.world == 1 and not args.no_cpu_baseline:
            figs.append(("cpu_baseline", lambda: cpu_baseline(args.cpu_seconds)))
        if args.async_learner:
            return figs
        if args.games != 8192:
            # the per-GPU size of BASELINE configs[4] with the same method, so that the 0.65 of DESIGN 4.1 is timed by whoever
            # runs this file, not quoted
            figs.append(("roofline_8192", lambda: env_roofline(dev, 8192, args.replay, 42)))
        figs.append(("fused_policy_head", lambda: fused_head_times(self.agent, self.buf, dev)))
        if self.world == 1 and not args.no_cpu_baseline:
            figs.append(("ppo_reference_minibatch64", lambda: minibatch64_rate(self.agent, self.buf, dev)))
        return figs


JOB_FACTORY = SelfPlayJob


def _barrier_fn(job, local_rank):
    import torch.distributed as dist

    def barrier():
        job.synchronize()
        if dist.is_initialized():       # (more than one rank, or a one-rank rehearsal group: KSH_DIST_REHEARSAL=1)
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local_rank])     # RCCL: name the device, do not let the barrier guess it
            else:
                dist.barrier()
        job.synchronize()
    return barrier


def worker_main(args):
    """The rank body.  With KSH_BENCH_STAGES set (started by the supervisor) every finished part goes to the stage file
    and nothing is printed; without it (ranks > 0, or `--worker` by hand) rank 0 prints the line itself -- before the
    process-group teardown."""
    stages = Stages(os.environ.get(STAGES_ENV))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    from shogidrl_amd import graphs, launch, parallel
    if world_env > 1:
        # N ranks, N exhaustive MIOpen searches: each rank gets its own user find-db / kernel-cache directory instead of
        # queueing on the file locks of one shared sqlite file (set before MIOpen initialises, i.e. before the first conv)
        for k, v in launch.per_rank_miopen_env(int(os.environ.get("LOCAL_RANK", "0"))).items():
            os.environ.setdefault(k, v)
    if args.graph or args.async_learner:
        graphs.enable_safe_replay()                 # before the first HIP call: every hipGraph replay path requires it
    import torch
    import torch.distributed as dist
    quiet_stdout()
    rank, local_rank, world = parallel.init_from_env()
    if world != args.gpus:
        # the driver passes the same N to the launcher and to --gpus; anything else is a mis-launch, not a note
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node == --gpus)" % (args.gpus, world))
    t_start = time.perf_counter()

    def note(msg):
        if rank == 0:
            print("[bench %7.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    if args.roofline_only:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
        torch.cuda.set_device(local_rank)
        emit(env_roofline("cuda:%d" % local_rank, args.games, args.replay, 42))
        return 0
    stand_in = JOB_FACTORY is not SelfPlayJob
    if stand_in and os.environ.get("KSH_BENCH_REHEARSAL_STAND_IN") != "1":
        # there is no CPU path: anything but the GPU job is the test suite's dress rehearsal, and has to say so
        raise SystemExit("bench.py: JOB_FACTORY was replaced without KSH_BENCH_REHEARSAL_STAND_IN=1 (tests/helpers/bench_rehearsal.py)")
    job = JOB_FACTORY(args, rank, local_rank, world)
    barrier = _barrier_fn(job, local_rank)
    job.prepare(note)
    t_w0 = time.perf_counter()
    first_warm = None
    for i in range(args.warmup):
        job.collect()
        job.synchronize()
        note("warmup %d: rollout done" % i)
        job.update()
        job.synchronize()
        if first_warm is None:
            first_warm = time.perf_counter() - t_w0
        note("warmup %d: update done" % i)
    warm_s = time.perf_counter() - t_w0
    t_roll, t_upd = job.timer(), job.timer()
    roll_ms = upd_ms = 0.0
    metrics = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t_roll.start()
        job.collect()
        t_roll.stop()
        t_upd.start()
        metrics = job.update()
        t_upd.stop()
        if not args.async_learner:                # (event reads synchronise: the overlapped schedule is timed as a whole)
            roll_ms += t_roll.elapsed_ms()
            upd_ms += t_upd.elapsed_ms()
        note("timed step done (rollout %.2f s, update %.2f s so far)" % (roll_ms * 1e-3, upd_ms * 1e-3))
    job.finish()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():       # (more than one rank, or a one-rank rehearsal group: KSH_DIST_REHEARSAL=1)
        t = torch.tensor([elapsed, roll_ms, upd_ms, warm_s, first_warm or 0.0], dtype=torch.float64, device=job.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, roll_ms, upd_ms, warm_s, first_warm = (float(x) for x in t.cpu())
    transitions = world * args.steps * args.rollout * args.games
    if rank == 0:
        backend = "none (single rank)"
        if dist.is_initialized():
            backend = dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")
        head = {
            "metric": METRIC,
            "value": transitions / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "world_size": dist.get_world_size() if dist.is_initialized() else 1,      # as torch.distributed sees it
            "backend": backend,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": job.workload(),
                "games_per_gpu": args.games, "rollout_len": args.rollout, "ppo_epochs": args.epochs,
                "minibatch_size": args.minibatch, "rollout": "hipGraph replay" if args.graph else "eager launches",
                "async_learner": bool(args.async_learner), "sync_interval": args.sync_interval,
                "old_log_probs": ("re-computed at learn time (opt-in deviation 6)" if args.recompute_old_log_probs
                                  else "as stored while acting (the reference's semantics)"),
                "parallelism": "dp%d (games sharded, flat-gradient all-reduce)" % world,
                "env_dtype": ("u32 bitboards -> bf16 channels-last obs / bit-packed mask (compact, SURVEY 8f4)" if args.compact
                              else "u32 bitboards -> f32 obs / u8 mask"),
            },
            "warmup_s": round(warm_s, 2), "first_warmup_step_s": round(first_warm or 0.0, 2),   # (MIOpen's exhaustive find lives here)
            "ppo_metrics": metrics,
            "episodes": job.episode_stats(),
        }
        if not args.async_learner and roll_ms > 0:
            head["selfplay_env_steps_per_s"] = transitions / (roll_ms * 1e-3)
            head["ppo_samples_per_s"] = transitions * args.epochs / (upd_ms * 1e-3)
        head.update(job.extra_headline())
        if stand_in:
            head["rehearsal_stand_in"] = True     # NOT a measurement: the rank's GPU job was replaced by the tests' CPU stand-in
        stages.put("headline", head)              # from here on the measurement cannot be lost (supervisor) ...
        note("headline: %.1f env-steps/s (%.1f ms per step)" % (head["value"], head["ms_per_step"]))
        for name, fn in job.side_figures():
            _test_abort(name)
            try:
                stages.put(name, fn())
            except Exception as e:               # recorded, never fatal (a GPU fault or abort is the supervisor's case)
                stages.put(name, {"error": "%s: %s" % (type(e).__name__, e)})
            job.synchronize()
            note("%s done" % name)
            if name == "roofline" and "avg_launch_us" in stages.parts[name]:
                stages.put("env_kernel_env_steps_per_s", args.games / (stages.parts[name]["avg_launch_us"] * 1e-6))
        _test_abort("complete")
        stages.put("complete", {})
        if not stages.path:
            emit(assemble(stages.parts))          # ... and a worker on its own prints BEFORE the process-group teardown
    if dist.is_initialized():       # (more than one rank, or a one-rank rehearsal group: KSH_DIST_REHEARSAL=1)
        barrier()
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------------ side process: update graph
def side_update_graph_main(args):
    """`ppo_reference_minibatch64.hipgraph_replay`: the minibatch-64 PPO update (default_config.yaml:88) replayed from ONE
    captured hipGraph (PPOConfig.graph_minibatch_step) next to the same updates launched eagerly -- in a process of its own
    with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, on its own data: 64 games x 64 plies of self-play by a random-init network of the
    bench's architecture (4096 transitions; the rollout's forward has the minibatch's batch size)."""
    from shogidrl_amd import graphs
    graphs.enable_safe_replay()
    import copy
    import dataclasses
    import torch
    quiet_stdout()
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import VecExperienceBuffer
    from shogidrl_amd.env import DeviceTimer, VecShogiEnv
    from shogidrl_amd.models import ResTowerActorCritic
    from shogidrl_amd.rollout import SelfPlayRunner
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev = "cuda:0"
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(42)
    updates, games = 64, 64
    n = 64 * updates
    T = n // games
    cfg = PPOConfig(ppo_epochs=1, minibatch_size=64, mixed_precision=True, channels_last=True, seed=42,
                    recompute_old_log_probs=bool(args.recompute_old_log_probs))
    actor = VecPPOAgent(ResTowerActorCritic(46, 13527, args.depth, args.width, 0.25), cfg, dev)
    buf = VecExperienceBuffer(T, games, cfg.gamma, cfg.lambda_gae, dev)
    SelfPlayRunner(VecShogiEnv(games, 500, device=dev), actor, buf).collect()
    small = _small_buffer(buf, n, dev)
    tm = DeviceTimer()
    out = {"updates": updates, "minibatch_size": 64,
           "sample": "%d updates of 64 samples on %d games x %d plies of self-play generated in this process (random-init ResNet "
                     "%dx%d), DEBUG_CLR_GRAPH_PACKET_CAPTURE=%s" % (updates, games, T, args.depth, args.width,
                                                                     os.environ.get(graphs.PACKET_CAPTURE_ENV))}
    for name, graphed in (("eager", False), ("graph", True)):
        ag = VecPPOAgent(copy.deepcopy(actor.model), dataclasses.replace(cfg, graph_minibatch_step=graphed), dev)
        ag.learn(small)                          # warm-up: MIOpen's choice; graph: three eager updates, capture, replays
        ag.learn(small)                          # (a second learn() between capture and measurement: the case that used to corrupt)
        torch.cuda.synchronize()
        tm.start()
        m = ag.learn(small)
        tm.stop()
        ms = tm.elapsed_ms()
        out[name] = {"value": n / (ms * 1e-3), "unit": "PPO samples/s", "ms_per_update": ms / updates,
                     "grad_norm": ag.last_gradient_norm, "finite": bool(all(v == v for v in m.values()))}
        del ag
    emit(out)
    return 0


# ------------------------------------------------------------------------------------------------ supervisor (rank 0)
def _run_side(argv_tail, env_extra, timeout):
    cmd = [sys.executable, os.path.abspath(sys.argv[0])] + argv_tail
    env = dict(os.environ)
    env.pop(STAGES_ENV, None)
    env.pop(ABORT_ENV, None)
    env.update(env_extra)
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=None, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "side process exceeded %d s and was stopped" % timeout}
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "side process exited with code %d%s" % (r.returncode, "" if lines else " and printed no result")}
    try:
        return json.loads(lines[-1])
    except ValueError as e:
        return {"error": "side process printed an unreadable result: %s" % e}


def _die_with_parent():
    """preexec_fn: PR_SET_PDEATHSIG = SIGTERM (a supervisor that is killed outright must not leave its worker on the GPU)"""
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(1, signal.SIGTERM)
    except (OSError, AttributeError):
        pass


def supervisor_main(args):
    """Rank 0's outer process: no torch, no HIP.  Starts the worker, keeps what it reports, prints the ONE line."""
    fd, path = tempfile.mkstemp(prefix="ksh_bench_stages_", suffix=".jsonl")
    os.close(fd)
    env = dict(os.environ)
    env[STAGES_ENV] = path
    cmd = [sys.executable, os.path.abspath(sys.argv[0])] + sys.argv[1:] + ["--worker"]
    child = subprocess.Popen(cmd, env=env, stdout=sys.stderr, preexec_fn=_die_with_parent)   # (the worker prints nothing on stdout; make that certain)

    def forward(signum, frame):                                     # torch.distributed.run stops its ranks with SIGTERM
        if child.poll() is None:
            child.terminate()
    for s in (signal.SIGTERM, signal.SIGINT):
        signal.signal(s, forward)
    complete_at, stopped = None, False
    grace = float(os.environ.get("KSH_BENCH_TEARDOWN_GRACE_S", "120"))
    while child.poll() is None:
        time.sleep(0.2)
        if complete_at is None:
            try:
                if os.path.getsize(path) and "complete" in read_stages(path)[0]:
                    complete_at = time.time()
            except OSError:
                pass
        elif time.time() - complete_at > grace and not stopped:
            # every figure is in; what is left is the process-group teardown -- a hang there must not eat the line
            sys.stderr.write("[bench] worker still alive %.0f s after its last stage (teardown hang?): stopping it\n" % grace)
            child.terminate()
            stopped = True
            try:
                child.wait(timeout=15)
            except subprocess.TimeoutExpired:
                child.kill()
    rc = child.wait()
    parts, order = read_stages(path)
    try:
        os.unlink(path)
    except OSError:
        pass
    if "headline" not in parts:
        sys.stderr.write("[bench] worker exited with code %d before the timed region finished: nothing to report\n" % rc)
        return rc if rc > 0 else 1
    out = assemble(parts)
    if "complete" not in parts or (rc != 0 and not stopped):
        out["worker_exit"] = {"code": rc, "last_stage": order[-1] if order else None,
                              "note": "the worker died after the timed region; stages it had finished are reported, the rest is missing"}
        sys.stderr.write("[bench] worker exited with code %d after stage %r; reporting what it had finished\n" % (rc, order[-1]))
    elif stopped:
        out["worker_exit"] = {"code": rc, "last_stage": "complete", "note": "all figures reported; the worker was stopped in its teardown"}
    wants_side = args.gpus == 1 and not args.no_cpu_baseline and not args.async_learner and not args.graph
    if wants_side and isinstance(out.get("ppo_reference_minibatch64"), dict):
        tail = ["--side", "update-graph", "--depth", str(args.depth), "--width", str(args.width)]
        if args.recompute_old_log_probs:
            tail.append("--recompute-old-log-probs")
        sys.stderr.write("[bench] side process: minibatch-64 update replayed from a hipGraph\n")
        out["ppo_reference_minibatch64"]["hipgraph_replay"] = _run_side(tail, {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}, 300)
    emit(out)
    need = ["value", "roofline"] + (["cpu_baseline"] if args.gpus == 1 and not args.no_cpu_baseline else [])
    good = lambda k: k in out and not (isinstance(out[k], dict) and "error" in out[k])   # (a figure that raised is not a figure)
    return 0 if all(good(k) for k in need) else (rc if rc > 0 else 1)


def main():
    args = parse()
    if args.side == "update-graph":
        return side_update_graph_main(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.worker:
        # not started by torch.distributed.run: be the launcher.  This parent makes no GPU call; it starts --gpus fresh
        # ranks (one per GPU), rank 0 prints the JSON line, and any failing rank fails the whole run.
        from shogidrl_amd import launch
        return launch.spawn_ranks([sys.executable, os.path.abspath(sys.argv[0])] + sys.argv[1:], args.gpus)
    if args.worker or args.roofline_only or int(os.environ.get("RANK", "0")) != 0:
        return worker_main(args)
    return supervisor_main(args)


if __name__ == "__main__":
    sys.exit(main())
