"""Asynchronous learner (BASELINE configs[4]: "hipGraph-captured rollout step + async learner").

Reference semantics being mirrored (keisei/training/parallel/): self-play workers act with their OWN copy of the network
while the trainer keeps learning; the trainer pushes fresh weights to them only every `sync_interval` steps
(model_sync.py:50 `should_sync`, parallel_manager.py:175-221, default_config.yaml:458 `sync_interval: 100`), so the
behaviour policy lags the learner.  The reference moves gzip'd state dicts and pickled experience batches through
multiprocessing queues (communication.py:63-89); none of that is reproduced.  MI355X-first design instead:

  OverlappedSelfPlayLearner   every GPU is actor AND learner.  With a 13 s update and a 0.4 s rollout (4096 games, ResNet
      9x256) a dedicated actor GPU would idle 97 % of the time, so the rollout runs on a second HIP stream of the SAME
      GPU, concurrently with learn() on the main stream: rollout i+1 is played by an ActingPolicy (a device-side copy of
      the network, refreshed by one multi-tensor copy every `sync_interval` iterations) into the second of two rollout
      buffers while learn() consumes rollout i.  The policy that generates the data is therefore (at least) one
      iteration stale -- exactly the reference's worker lag.  Across GPUs nothing changes: games stay sharded, gradients
      are all-reduced over RCCL by learn(), every rank refreshes its own actor copy locally (no weight broadcast at all).
      Optionally each buffer's rollout is replayed from its own captured hipGraph.
  ActorLearnerLoop            the reference's role split for process groups that want it (dedicated actor ranks and
      learner ranks): experience goes actor -> learner with point-to-point sends, weights go learner -> actors with
      parallel.broadcast_parameters every `sync_interval` rounds.  Host logic only (any backend; the CPU tests run it
      over gloo).

Both are OFF by default: an asynchronous learner changes the on-policy semantics of PPO (bench.py --async-learner)."""
import torch
import torch.distributed as dist

from . import graphs, parallel
from .agent import ActingPolicy
from .rollout import SelfPlayRunner


class PolicySync:
    """When to refresh the actors' weights: the reference's ModelSynchronizer.should_sync (model_sync.py:50-61)."""

    def __init__(self, sync_interval=1):
        if sync_interval < 1:
            raise ValueError("sync_interval must be >= 1")
        self.sync_interval = int(sync_interval)
        self.last_sync_step = 0
        self.sync_count = 0

    def should_sync(self, current_step):
        return current_step - self.last_sync_step >= self.sync_interval

    def mark(self, current_step):
        self.last_sync_step = current_step
        self.sync_count += 1


class OverlappedSelfPlayLearner:
    """Rollout i+1 on a side stream while learn() works on rollout i (one GPU, two HIP streams, two rollout buffers).

    step() = one learner iteration.  Data flow and the events that order it:
        side stream : [wait learn(i-1) done] refresh actor weights if due -> carry games over -> rollout i+1 -> E_roll
        main stream : [wait E_roll of rollout i] learn(rollout i) -> E_learn
    The buffer rollout i+1 overwrites was last read by learn(i-1), which precedes learn(i) on the main stream; the
    actor copy is only written on the side stream, between two rollouts, from weights a finished learn() produced.
    overlap=False runs the very same schedule serially on the main stream (tests compare the two bit for bit)."""

    def __init__(self, env, agent, buffers, sync_interval=1, graphed=False, overlap=True):
        assert len(buffers) == 2
        if graphed:
            graphs.require_safe_replay("OverlappedSelfPlayLearner(graphed=True)")
        self.env, self.agent, self.buffers = env, agent, buffers
        self.actor = ActingPolicy(agent.model, agent.config, agent.device)
        self.runners = [SelfPlayRunner(env, self.actor, buffers[0]), SelfPlayRunner(env, self.actor, buffers[1], reset=False)]
        self.sync = PolicySync(sync_interval)
        self.graphed = bool(graphed)
        self.overlap = bool(overlap)
        dev = agent.device
        self.side = torch.cuda.Stream(device=dev) if self.overlap else None
        self.iteration = 0                      # learner iterations completed
        self.learner_version = 0                # == iteration: how many learn() calls produced the learner's weights
        self._e_roll = [None, None]             # rollout into buffer k finished (recorded on the rollout stream)
        self._e_learn = None                    # last learn() finished (recorded on the main stream)
        self._filled = None                     # index of the buffer holding the rollout the next learn() consumes
        self.rollout_versions = []              # weights version that played each rollout (diagnostics / tests)
        self._prime()

    # ---- rollout side -------------------------------------------------------------------------------------------
    def _rollout_stream(self):
        return self.side if self.overlap else torch.cuda.current_stream(self.agent.device)

    def _collect_into(self, k, carry_from=None):
        """enqueue: (weight refresh if due) + rollout into buffer k, on the rollout stream"""
        main = torch.cuda.current_stream(self.agent.device)
        st = self._rollout_stream()
        if self.overlap:
            st.wait_stream(main) if self._e_learn is None else st.wait_event(self._e_learn)
        with torch.cuda.stream(st):
            if self.sync.should_sync(self.learner_version) and self.actor.weights_version != self.learner_version:
                self.actor.load_from(self.agent.model, version=self.learner_version)
                self.sync.mark(self.learner_version)
                if self.overlap:
                    # the next learn() must not start changing the learner's tensors while this copy still reads them
                    loaded = torch.cuda.Event()
                    loaded.record(st)
                    main.wait_event(loaded)
            if carry_from is not None:
                self.runners[k].carry_from(self.buffers[carry_from])
            r = self.runners[k]
            r.collect_graphed() if self.graphed else r.collect()
            ev = torch.cuda.Event()
            ev.record(st)
        self._e_roll[k] = ev
        self.rollout_versions.append(self.actor.weights_version)

    def _prime(self):
        self.actor.load_from(self.agent.model, version=0)
        self._collect_into(0)
        self._filled = 0

    # ---- one learner iteration ------------------------------------------------------------------------------------
    def step(self):
        cur, nxt = self._filled, 1 - self._filled
        main = torch.cuda.current_stream(self.agent.device)
        # next rollout first: its launches go to the side stream and overlap the learn() enqueued right after
        self._collect_into(nxt, carry_from=cur)
        main.wait_event(self._e_roll[cur])
        # the rollout was played by the lagging actor copy: its stored log-probs ARE the behaviour policy; they are never
        # re-computed from the learner's newer weights (that would start the PPO ratio at 1 against the wrong policy)
        metrics = self.agent.learn(self.buffers[cur], behaviour_is_current=False)
        self.buffers[cur]._advantages_computed = False
        self._e_learn = torch.cuda.Event()
        self._e_learn.record(main)
        self.iteration += 1
        self.learner_version = self.iteration
        self._filled = nxt
        return metrics

    def synchronize(self):
        if self.side is not None:
            self.side.synchronize()
        torch.cuda.current_stream(self.agent.device).synchronize()

    @property
    def global_timestep(self):
        return sum(r.global_timestep for r in self.runners)

    def episode_stats(self):
        e = (self.runners[0].episodes + self.runners[1].episodes).cpu().tolist()
        return {"total_episodes_completed": e[0], "black_wins": e[1], "white_wins": e[2], "draws": e[3]}


class ActorLearnerLoop:
    """Role split over a process group: ranks < n_learners learn, the others act.

    One `round` = every actor collects one batch (a dict of tensors, same keys / shapes every round) with ITS copy of
    the model and sends it to learner `actor_index % n_learners`; every learner receives its actors' batches, runs
    `learn_fn(list_of_batches)`; after the round, when `PolicySync.should_sync(round)` holds on all ranks (a pure
    function of the round number, so every rank takes the same branch and no rank waits for a collective the others
    skip), the learners' model is broadcast from rank 0 to everybody.  Actors do not wait for the learner between two
    sync points: they run up to `sync_interval` rounds ahead with stale weights -- the reference's worker lag.

    `collect_fn(model, round) -> dict[str, Tensor]`, `learn_fn(batches, round) -> Any` are supplied by the caller (on
    the MI355X: SelfPlayRunner.collect + buffer.get_batch, and VecPPOAgent.learn on the concatenated batch)."""

    def __init__(self, model, n_learners, sync_interval=1, group=None, template=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("ActorLearnerLoop needs an initialised process group")
        self.model, self.group = model, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if not 0 < n_learners < self.world:
            raise ValueError("need at least one learner and one actor rank")
        self.n_learners = n_learners
        self.is_learner = self.rank < n_learners
        self.sync = PolicySync(sync_interval)
        self.template = template                # learner side: {key: (shape, dtype)} of an actor batch
        self.round = 0
        self.weights_round = 0                  # round after which this rank's weights were last refreshed
        parallel.broadcast_parameters(self.model, src=0, group=group)    # everybody starts from the learner's weights

    def my_actors(self):
        return [r for r in range(self.n_learners, self.world) if (r - self.n_learners) % self.n_learners == self.rank]

    def my_learner(self):
        return (self.rank - self.n_learners) % self.n_learners

    def run_round(self, collect_fn=None, learn_fn=None):
        """Returns learn_fn's result on learners, the batch the actor sent on actors."""
        out = None
        if self.is_learner:
            batches = []
            for src in self.my_actors():
                b = {}
                for key in sorted(self.template):
                    shape, dtype = self.template[key]
                    t = torch.empty(shape, dtype=dtype, device=next(self.model.parameters()).device)
                    dist.recv(t, src=src, group=self.group)
                    b[key] = t
                batches.append(b)
            out = learn_fn(batches, self.round)
        else:
            batch = collect_fn(self.model, self.round)
            for key in sorted(batch):
                dist.send(batch[key].contiguous(), dst=self.my_learner(), group=self.group)
            out = batch
        self.round += 1
        if self.sync.should_sync(self.round):
            parallel.broadcast_parameters(self.model, src=0, group=self.group)
            self.sync.mark(self.round)
            self.weights_round = self.round
        return out
