"""Episode bookkeeping on the device.

`EpisodeTracker` keeps what `PufferGPUDrive.step()` keeps (reference gpudrive/env/env_puffer.py:250-403):
live-agent mask, per-agent episode returns and lengths, collision / off-road counts, detection of
finished worlds, their statistics and their asynchronous reset -- in ONE kernel per step plus the
device-side reset pass, with no `.item()` / `.cpu()` synchronisation.  Field names follow the reference."""
import ctypes as C

import torch

from . import _capi

REWARD_TYPES = {"weighted_combination": 0, "sparse_on_goal_achieved": 1}


class EpisodeTracker:
    def __init__(self, sim, collision_weight=-0.5, goal_achieved_weight=1.0, off_road_weight=-0.5,
                 reward_type="weighted_combination", auto_reset=True):
        self.sim = sim
        self._L = _capi.lib()
        W, A = sim._W, sim._A
        dev = sim.controlled_state_tensor().to_torch().device
        # cont_agent_mask, captured once like gpudrive/env/env_torch.py:61-63 does at construction
        self.controlled_agent_mask = sim.controlled_state_tensor().to_torch().clone().squeeze(-1) == 1
        self.num_agents = None  # filled lazily: a device->host sync the step path never needs
        self.cfg = _capi.GdEpisodeConfig(float(collision_weight), float(goal_achieved_weight), float(off_road_weight),
                                         REWARD_TYPES[reward_type], 1 if auto_reset else 0)
        z = lambda dt: torch.zeros((W, A), dtype=dt, device=dev)
        self.agent_episode_returns, self.episode_lengths = z(torch.float32), z(torch.float32)
        self.collided_in_episode, self.offroad_in_episode = z(torch.float32), z(torch.float32)
        self.live_agent_mask = torch.ones((W, A), dtype=torch.bool, device=dev)  # env_puffer.py:221-223
        self.rewards, self.terminals = z(torch.float32), z(torch.bool)
        self.truncations, self.masks = z(torch.bool), z(torch.bool)
        self.done_worlds = torch.zeros((W,), dtype=torch.int32, device=dev)
        self.stats = torch.zeros((_capi.EPISODE_STATS,), dtype=torch.float32, device=dev)
        self.world_stats = torch.zeros((W, _capi.EPISODE_STATS), dtype=torch.float32, device=dev)
        self._cmask_u8 = self.controlled_agent_mask.to(torch.uint8).contiguous()
        p = lambda t: C.c_void_p(t.data_ptr())
        self._bufs = _capi.GdEpisodeBuffers(
            p(self._cmask_u8), p(self.agent_episode_returns), p(self.episode_lengths), p(self.collided_in_episode),
            p(self.offroad_in_episode), p(self.live_agent_mask), p(self.rewards), p(self.terminals), p(self.truncations),
            p(self.masks), p(self.done_worlds), p(self.stats), p(self.world_stats))

    def step(self, step_sim=True):
        """sim.step() (actions are already in the action tensor), then the bookkeeping kernel and the
        device-side reset of the worlds that just finished.  Returns full [W, A] tensors
        (rewards, terminals, truncations, masks); index them with `controlled_agent_mask` to get the
        flat per-agent views PufferGPUDrive returns.  Nothing here waits for the device."""
        if step_sim:
            self.sim.step()
        self.sim._bind_stream()
        _capi.check(self._L.gd_episode_step(self.sim._h, C.byref(self.cfg), C.byref(self._bufs)), "gd_episode_step")
        return self.rewards, self.terminals, self.truncations, self.masks

    def pop_stats(self):
        """Running sums over the episodes finished since the last call, as the dictionary PufferGPUDrive
        logs (env_puffer.py:352-371); the one place that synchronises with the device."""
        s = self.stats.cpu().tolist()
        self.stats.zero_()
        if self.num_agents is None:
            self.num_agents = int(self.controlled_agent_mask.sum().item())
        ep, fin = s[0], s[1]
        if ep == 0 or fin == 0:
            return {}
        A = self.controlled_agent_mask.shape[1]
        return {
            "mean_episode_reward_per_agent": s[2] / fin,
            "perc_goal_achieved": s[5] / fin,
            "perc_off_road": s[3] / fin,
            "perc_veh_collisions": s[4] / fin,
            "total_controlled_agents": self.num_agents,
            "control_density": self.num_agents / self.controlled_agent_mask.numel(),
            "episode_length": s[7] / (ep * A),
            "perc_truncated": s[6] / fin,
            "num_completed_episodes": int(ep),
            "total_collisions": s[8],
            "total_off_road": s[9],
        }
