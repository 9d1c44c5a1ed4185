"""TEST INFRASTRUCTURE ONLY (see oracle/oracle.py): numpy restatement of the episode bookkeeping of
`PufferGPUDrive.step()` (reference gpudrive/env/env_puffer.py:250-403, rewards
gpudrive/env/env_torch.py:469-505, Info columns gpudrive/datatypes/info.py:11-15), statement by
statement, over any simulator with the SimManager tensor surface (the CPU oracle in the tests).
Parity unpinned: env_puffer needs pufferlib / wandb / gymnasium and cannot be imported here."""
import numpy as np

f32 = np.float32
STAT_NAMES = ("episodes", "finished_agents", "return_sum", "off_road_agents", "collided_agents", "goal_achieved",
              "truncated_agents", "length_sum", "total_collisions", "total_off_road")


class OracleEpisodeTracker:
    def __init__(self, sim, collision_weight=-0.5, goal_achieved_weight=1.0, off_road_weight=-0.5,
                 reward_type="weighted_combination"):
        self.sim = sim
        self.cw, self.gw, self.ow = f32(collision_weight), f32(goal_achieved_weight), f32(off_road_weight)
        self.reward_type = reward_type
        self.controlled_agent_mask = np.array(sim.controlled_state_tensor())[..., 0] == 1   # cont_agent_mask
        W, A = self.controlled_agent_mask.shape
        # reset(), env_puffer.py:200-236
        self.agent_episode_returns = np.zeros((W, A), f32)
        self.episode_lengths = np.zeros((W, A), f32)
        self.live_agent_mask = np.ones((W, A), bool)
        self.collided_in_episode = np.zeros((W, A), f32)
        self.offroad_in_episode = np.zeros((W, A), f32)
        self.world_stats = np.zeros((W, 12), f32)

    def get_rewards(self):  # env_torch.py:469-505
        info = np.array(self.sim.info_tensor())
        off_road = info[:, :, 0].astype(f32)
        collided = info[:, :, 1:3].astype(f32).sum(axis=2)
        goal_achieved = info[:, :, 3].astype(f32)
        if self.reward_type == "sparse_on_goal_achieved":
            return np.array(self.sim.reward_tensor())[..., 0].copy()
        return self.cw * collided + self.gw * goal_achieved + self.ow * off_road

    def step(self):
        """Everything PufferGPUDrive.step does after `self.env.step_dynamics(self.actions)`."""
        self.sim.step()
        reward = self.get_rewards()
        terminal = np.array(self.sim.done_tensor())[..., 0].astype(f32).astype(bool)
        controlled_per_world = self.controlled_agent_mask.sum(axis=1)
        done_worlds = np.where((terminal * self.controlled_agent_mask).sum(axis=1) == controlled_per_world)[0]
        self.agent_episode_returns[self.live_agent_mask] += reward[self.live_agent_mask]
        self.episode_lengths += 1
        info = np.array(self.sim.info_tensor())
        self.offroad_in_episode += info[:, :, 0]
        self.collided_in_episode += info[:, :, 1:3].sum(axis=2)
        masks = self.live_agent_mask.copy()
        self.live_agent_mask[terminal] = 0
        goal_achieved = info[:, :, 3]
        truncated = np.logical_and(~self.offroad_in_episode.astype(bool),
                                   np.logical_and(~self.collided_in_episode.astype(bool), ~goal_achieved.astype(bool)))
        done_flags = np.zeros(len(controlled_per_world), np.int32)
        if len(done_worlds) > 0:
            done_flags[done_worlds] = 1
            for w in done_worlds:  # the reference aggregates over all finished worlds; kept per world here
                cm = self.controlled_agent_mask[w]
                self.world_stats[w, :10] = [
                    1, cm.sum(), _tree_sum(np.where(cm, self.agent_episode_returns[w], f32(0))),
                    (self.offroad_in_episode[w][cm] > 0).sum(), (self.collided_in_episode[w][cm] > 0).sum(),
                    goal_achieved[w][cm].sum(), truncated[w][cm].sum(), _tree_sum(self.episode_lengths[w]),
                    _tree_sum(self.collided_in_episode[w]), _tree_sum(self.offroad_in_episode[w])]
            self.sim.reset([int(w) for w in done_worlds])
            self.agent_episode_returns[done_worlds, :] = 0
            self.episode_lengths[done_worlds, :] = 0
            self.live_agent_mask[done_worlds] = self.controlled_agent_mask[done_worlds]
            self.offroad_in_episode[done_worlds, :] = 0
            self.collided_in_episode[done_worlds, :] = 0
        return reward, terminal, truncated, masks, done_flags


def _tree_sum(x):
    """float32 pairwise tree (what a power-of-two block reduction computes)."""
    x = np.asarray(x, f32).copy()
    n = len(x)
    while n > 1:
        n //= 2
        x[:n] = x[:n] + x[n:2 * n]
    return x[0]
