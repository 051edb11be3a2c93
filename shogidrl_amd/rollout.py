"""Batched rollout driver -- the GPU counterpart of StepManager.execute_step +
TrainingLoopManager._run_epoch_sequential + Trainer.perform_ppo_update
(keisei/training/step_manager.py:98-348, training_loop_manager.py:440-454, trainer.py:214-269).

Per step, for all N games at once: (obs_t, mask_t) -> policy forward -> masked sample -> env step, storing
(obs_t, a, r, logp, v, done, mask_t); finished games are reset inside the env kernel and continue from the
reset observation (the terminal next_obs is discarded, step_manager.py:437-468).  After T steps the value of
obs_T bootstraps GAE and the agent learns.  Nothing in the loop synchronises with the host."""
import torch


class SelfPlayRunner:
    def __init__(self, env, agent, buffer):
        assert env.num_games == buffer.N and env.packed_masks == buffer.packed_masks
        assert env.obs_format == getattr(buffer, "obs_format", "f32")
        self.env, self.agent, self.buffer = env, agent, buffer
        self.global_timestep = 0
        self.episodes = torch.zeros(4, dtype=torch.int64, device=env.device)   # episodes, black wins, white wins, draws
        env.auto_reset = True
        env.reset(out_obs=buffer.obs[0], out_mask=buffer.legal_masks[0])

    def collect(self):
        env, agent, buf = self.env, self.agent, self.buffer
        for t in range(buf.T):
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
        self.global_timestep += buf.T * buf.N

    def update(self):
        metrics = self.agent.learn(self.buffer)
        self.buffer.clear()
        return metrics

    def iteration(self):
        self.collect()
        return self.update()

    def episode_stats(self):
        e = self.episodes.cpu().tolist()
        return {"total_episodes_completed": e[0], "black_wins": e[1], "white_wins": e[2], "draws": e[3]}
