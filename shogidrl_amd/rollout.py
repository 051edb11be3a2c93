"""Batched rollout driver -- the GPU counterpart of StepManager.execute_step +
TrainingLoopManager._run_epoch_sequential + Trainer.perform_ppo_update
(keisei/training/step_manager.py:98-348, training_loop_manager.py:440-454, trainer.py:214-269).

Per step, for all N games at once: (obs_t, mask_t) -> policy forward -> masked sample -> env step, storing
(obs_t, a, r, logp, v, done, mask_t); finished games are reset inside the env kernel and continue from the
reset observation (the terminal next_obs is discarded, step_manager.py:437-468).  After T steps the value of
obs_T bootstraps GAE and the agent learns.  Nothing in the loop synchronises with the host."""
import torch

from . import graphs


class SelfPlayRunner:
    def __init__(self, env, agent, buffer, reset=True):
        """reset=False: attach to an env that is already running (a second runner over another buffer of the same env --
        the asynchronous learner's double buffering); call carry_from(other_buffer) before collecting."""
        assert env.num_games == buffer.N and env.packed_masks == buffer.packed_masks and env.emit_bits == buffer.emit_bits
        assert env.obs_format == getattr(buffer, "obs_format", "f32")
        self.env, self.agent, self.buffer = env, agent, buffer
        self.global_timestep = 0
        self.episodes = torch.zeros(4, dtype=torch.int64, device=env.device)   # episodes, black wins, white wins, draws
        env.auto_reset = True
        buffer.masks_from_env = True             # from here on every mask row of this buffer is written by the env kernel
        # both mask forms per step (contract bytes + the bit-set the kernels read) unless one of them is switched off
        self._both = env.emit_bits and not env.packed_masks
        if not reset:
            return
        if self._both:
            env.reset(out_obs=buffer.obs[0], out_mask=buffer.legal_masks[0], out_bits=buffer.legal_bits[0])
        else:
            env.reset(out_obs=buffer.obs[0], out_mask=buffer.legal_masks[0])

    def carry_from(self, other):
        """continue the games where the rollout into `other` (a VecExperienceBuffer of the same shape) left them: its
        bootstrap observation / masks become slot 0 of this runner's buffer"""
        b, T = self.buffer, other.T
        b.obs[0].copy_(other.obs[T])
        b.legal_masks[0].copy_(other.legal_masks[T])
        if b.legal_bits is not None and b.legal_bits is not b.legal_masks:
            b.legal_bits[0].copy_(other.legal_bits[T])

    def collect_graphed(self):
        """collect() replayed from one hipGraph (BASELINE configs[4]: "hipGraph-captured rollout step").  Everything in
        the T-step loop is device work on fixed buffers -- the policy forward, the sampler (its Philox counter is a
        device tensor the graph increments), the env kernel writing slot t+1, the bookkeeping -- so the whole rollout
        is captured once and replayed with a single launch; trajectories are identical to collect()'s
        (tests/test_gpu_agent.py).  The first call runs one eager rollout (allocations, MIOpen solver choice), the
        second captures, later calls replay.  Needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 from before the first HIP call
        (shogidrl_amd/graphs.py), like every graph replay of this package."""
        graphs.require_safe_replay("SelfPlayRunner.collect_graphed")
        agent = self.agent
        if getattr(self, "_graph", None) is not None:
            self._graph.replay()
            self.buffer._advantages_computed = True      # host-side flag the captured GAE call set only once
            agent._sample_step += self.buffer.T
            self.global_timestep += self.buffer.T * self.buffer.N
            return
        if not getattr(self, "_graph_warm", False):
            self._graph_warm = True
            return self.collect()
        if agent._step_tensor is None:     # shared by every graph captured over this policy (two runners, two buffers)
            agent._step_tensor = torch.tensor([agent._sample_step], dtype=torch.int64, device=self.env.device)
        side = torch.cuda.Stream(device=self.env.device)
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                self._collect_body()
        torch.cuda.current_stream().wait_stream(side)
        self._graph = graph
        graph.replay()                     # capture records, it does not execute
        self.buffer._advantages_computed = True
        agent._sample_step += self.buffer.T
        self.global_timestep += self.buffer.T * self.buffer.N

    def collect(self):
        if getattr(self, "_graph", None) is not None:
            raise RuntimeError("this runner replays a captured rollout graph; call collect_graphed()")
        self._collect_body()
        self.global_timestep += self.buffer.T * self.buffer.N

    def _collect_body(self):
        env, agent, buf = self.env, self.agent, self.buffer
        for t in range(buf.T):
            if self._both:
                a, logp, v = agent.select_actions(buf.obs[t], buf.legal_bits[t])
                env.step(a, prev_bits=buf.legal_bits[t], out_obs=buf.obs[t + 1], out_mask=buf.legal_masks[t + 1],
                         out_bits=buf.legal_bits[t + 1])
            else:
                a, logp, v = agent.select_actions(buf.obs[t], buf.legal_masks[t])
                env.step(a, prev_mask=buf.legal_masks[t], out_obs=buf.obs[t + 1], out_mask=buf.legal_masks[t + 1])
            buf.actions[t] = a
            buf.log_probs[t] = logp
            buf.values[t] = v
            buf.rewards[t].copy_(env.reward)
            buf.dones[t].copy_(env.done)
            w = env.winner
            d = env.done
            self.episodes += torch.stack([d.sum(), (d & (w == 0)).sum(), (d & (w == 1)).sum(), (d & (w < 0)).sum()])
        last_values = agent.get_values(buf.obs[buf.T])
        buf.compute_advantages_and_returns(last_values)

    def update(self):
        metrics = self.agent.learn(self.buffer)
        self.buffer.clear()
        return metrics

    def iteration(self, graphed=False):
        self.collect_graphed() if graphed else self.collect()
        return self.update()

    def episode_stats(self):
        e = self.episodes.cpu().tolist()
        return {"total_episodes_completed": e[0], "black_wins": e[1], "white_wins": e[2], "draws": e[3]}
