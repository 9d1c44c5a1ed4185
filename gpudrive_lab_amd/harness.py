"""Torch-only mirror of the call sequence the reference's gym wrapper drives the simulator with
(last row): `GPUDriveTorchEnv.step_dynamics -> get_rewards -> get_dones -> get_obs`
(reference gpudrive/env/env_torch.py:606-613, 453-505, 897-945), starting from discrete action INDICES that go
through the action table (`_set_discrete_action_space`, :666-724) and the in-place `[:, :, :3].copy_()` write
into the exported action tensor (`_copy_actions_to_simulator`, :645-664); and the calls around an episode:
`reset` (:403-451), `get_infos` (:462-467), `get_controlled_agents_mask` (:1218-1222), `get_expert_actions`
(:1445-1509), `advance_sim_with_log_playback` (:1274-1293), `remove_agents_by_id` (:1295-1349), `swap_data_batch`
(:1351-1384), `get_env_filenames` / `get_scenario_ids` (:1511-1537), the partner / road masks (:1224-1272), the optional
LiDAR and BEV observations (:898-945) and frame stacking (:1209-1216).

The reference wrapper itself keeps working on top of the drop-in `madrona_gpudrive` module; this class exists
because it cannot be imported where `gymnasium` / `pufferlib` are missing (the GPU box), and the boundary still
has to be exercised exactly the way the wrapper exercises it.  Same method names and argument meaning as the
reference's class; nothing here computes simulator results.
"""
from itertools import product

import torch

# gpudrive/env/config.py:62-75 (classic / bicycle), :72-77 (delta_local)
_ROUND = lambda t: torch.round(t, decimals=3)


def default_action_values(dynamics_model):
    if dynamics_model in ("classic", "bicycle"):
        return (_ROUND(torch.linspace(-4.0, 4.0, 7)), _ROUND(torch.linspace(-torch.pi, torch.pi, 13)), torch.Tensor([0]))
    if dynamics_model == "delta_local":
        return (_ROUND(torch.linspace(-6.0, 6.0, 20)), _ROUND(torch.linspace(-6.0, 6.0, 20)),
                _ROUND(torch.linspace(-torch.pi, torch.pi, 20)))
    raise ValueError("Invalid dynamics model: %s" % dynamics_model)


# gpudrive/env/constants.py:6-21
MAX_SPEED, MAX_VEH_LEN, MAX_VEH_WIDTH = 100, 30, 15
MIN_REL_GOAL_COORD, MAX_REL_GOAL_COORD = -1000, 1000
MIN_REL_AGENT_POS, MAX_REL_AGENT_POS = -1000, 1000
MAX_ORIENTATION_RAD = 2 * torch.pi
MIN_RG_COORD, MAX_RG_COORD = -1000, 1000
MAX_ROAD_LINE_SEGMENT_LEN = 100
MAX_ROAD_SCALE = 100
ROAD_TYPES = 7
NUM_MADRONA_ENTITY_TYPES = 11  # gpudrive/env/constants.py:32


def _normalize_min_max(x, lo, hi):  # gpudrive/utils/geometry.py normalize_min_max
    return 2 * ((x - lo) / (hi - lo)) - 1


class TorchCallSequence:
    """`sim` is a `madrona_gpudrive.SimManager`; `dynamics_model` one of "classic", "bicycle", "delta_local", "state"."""

    def __init__(self, sim, dynamics_model="classic", reward_type="sparse_on_goal_achieved", norm_obs=True,
                 action_values=None, vehicle_scale=0.7, init_steps=0, episode_len=91, num_stack=1):
        self.sim = sim
        self.dynamics_model = dynamics_model
        self.reward_type = reward_type
        self.norm_obs = norm_obs
        self.vehicle_scale = vehicle_scale
        self.device = sim.action_tensor().to_torch().device
        self._action_values = action_values
        self.action_keys_tensor = None
        if dynamics_model != "state":
            self._set_discrete_action_space()
        done = sim.done_tensor().to_torch()
        self.num_worlds, self.max_agent_count = done.shape[0], done.shape[1]
        self.world_time_steps = torch.zeros(self.num_worlds, dtype=torch.short, device=self.device)
        self.init_steps, self.episode_len, self.num_stack = init_steps, episode_len, num_stack
        self.stacked_obs = None
        self.data_batch = None
        if hasattr(sim, "controlled_state_tensor"):
            self._refresh_masks()

    # ---- actions: env_torch.py:666-724, 615-664 ----
    def _set_discrete_action_space(self):
        a1, a2, a3 = self._action_values or default_action_values(self.dynamics_model)
        self.action_key_to_values = {}
        self.values_to_action_key = {}
        for action_idx, (v1, v2, v3) in enumerate(product(a1, a2, a3)):
            self.action_key_to_values[action_idx] = [v1.item(), v2.item(), v3.item()]
            self.values_to_action_key[round(v1.item(), 5), round(v2.item(), 5), round(v3.item(), 5)] = action_idx
        self.action_keys_tensor = torch.tensor(
            [self.action_key_to_values[key] for key in sorted(self.action_key_to_values.keys())]).to(self.device)
        return len(self.action_key_to_values)

    def _apply_actions(self, actions):
        if self.dynamics_model in ("classic", "bicycle", "delta_local"):
            if actions.dim() == 2:  # (num_worlds, max_agent_count): indices
                actions = torch.nan_to_num(actions, nan=0).long().to(self.device)
                action_value_tensor = self.action_keys_tensor[actions]
            elif actions.dim() == 3:
                if actions.shape[2] == 1:
                    actions = actions.squeeze(dim=2).to(self.device)
                    action_value_tensor = self.action_keys_tensor[actions]
                else:  # the actual action values
                    action_value_tensor = actions.to(self.device)
            else:
                raise ValueError(f"Invalid action shape: {actions.shape}")
        else:
            action_value_tensor = actions.to(self.device)
        self._copy_actions_to_simulator(action_value_tensor)

    def _copy_actions_to_simulator(self, actions):
        if self.dynamics_model in ("classic", "bicycle", "delta_local"):
            self.sim.action_tensor().to_torch()[:, :, :3].copy_(actions)
        elif self.dynamics_model == "state":
            self.sim.action_tensor().to_torch()[:, :, :10].copy_(actions)
        else:
            raise ValueError(f"Invalid dynamics model: {self.dynamics_model}")

    # ---- step: env_torch.py:606-613 ----
    def step_dynamics(self, actions):
        if actions is not None:
            self._apply_actions(actions)
        self.sim.step()
        not_done_worlds = ~self.get_dones().any(dim=1)
        self.world_time_steps[not_done_worlds] += 1

    # ---- results: env_torch.py:453-505 ----
    def get_dones(self):
        return self.sim.done_tensor().to_torch().clone().squeeze(dim=2).to(torch.float)

    def get_rewards(self, collision_weight=-0.5, goal_achieved_weight=1.0, off_road_weight=-0.5):
        info_tensor = self.sim.info_tensor().to_torch().clone()
        off_road = info_tensor[:, :, 0].to(torch.float)
        collided = info_tensor[:, :, 1:3].to(torch.float).sum(axis=2)
        goal_achieved = info_tensor[:, :, 3].to(torch.float)
        if self.reward_type == "sparse_on_goal_achieved":
            return self.sim.reward_tensor().to_torch().clone().squeeze(dim=2)
        if self.reward_type == "weighted_combination":
            return collision_weight * collided + goal_achieved_weight * goal_achieved + off_road_weight * off_road
        raise ValueError("reward_type %r is outside the harness" % self.reward_type)

    # ---- observation: env_torch.py:756-896, 897-945; gpudrive/datatypes/observation.py, roadgraph.py ----
    def _get_ego_state(self):
        so = self.sim.self_observation_tensor().to_torch().clone()
        speed, length, width = so[:, :, 0], so[:, :, 1] * self.vehicle_scale, so[:, :, 2] * self.vehicle_scale
        gx, gy, collided = so[:, :, 4], so[:, :, 5], so[:, :, 6]
        if self.norm_obs:
            speed = speed / MAX_SPEED
            length = length / MAX_VEH_LEN
            width = width / MAX_VEH_WIDTH
            gx = _normalize_min_max(gx, MIN_REL_GOAL_COORD, MAX_REL_GOAL_COORD)
            gy = _normalize_min_max(gy, MIN_REL_GOAL_COORD, MAX_REL_GOAL_COORD)
        return torch.stack([speed, length, width, gx, gy, collided], dim=-1)

    def _get_partner_obs(self):
        po = self.sim.partner_observations_tensor().to_torch().clone()
        speed, x, y, yaw = po[..., 0:1], po[..., 1:2], po[..., 2:3], po[..., 3:4]
        length, width = po[..., 4:5] * self.vehicle_scale, po[..., 5:6] * self.vehicle_scale
        self.partner_ids = po[..., 8]
        if self.norm_obs:
            speed = speed / MAX_SPEED
            x = _normalize_min_max(x, MIN_REL_AGENT_POS, MAX_REL_AGENT_POS)
            y = _normalize_min_max(y, MIN_REL_AGENT_POS, MAX_REL_AGENT_POS)
            yaw = yaw / MAX_ORIENTATION_RAD
            length = length / MAX_VEH_LEN
            width = width / MAX_VEH_WIDTH
        return torch.concat([speed, x, y, yaw, length, width], dim=-1).flatten(start_dim=2)

    def _get_road_map_obs(self):
        rm = self.sim.agent_roadmap_tensor().to_torch().clone()
        x, y, seg_len, seg_w, seg_h, yaw = (rm[..., k] for k in range(6))
        types = torch.nn.functional.one_hot(rm[..., 6].long(), num_classes=ROAD_TYPES)
        if self.norm_obs:
            x = _normalize_min_max(x, MIN_RG_COORD, MAX_RG_COORD)
            y = _normalize_min_max(y, MIN_RG_COORD, MAX_RG_COORD)
            seg_len = seg_len / MAX_ROAD_LINE_SEGMENT_LEN
            seg_w = seg_w / MAX_ROAD_SCALE
            seg_h = seg_h / MAX_ROAD_SCALE
            yaw = yaw / MAX_ORIENTATION_RAD
        return torch.cat([x.unsqueeze(-1), y.unsqueeze(-1), seg_len.unsqueeze(-1), seg_w.unsqueeze(-1),
                          seg_h.unsqueeze(-1), yaw.unsqueeze(-1), types], dim=-1).flatten(start_dim=2)

    def get_obs(self, mask=None, reset=False):
        """env_torch.py:1172-1216: ego | partners | roads per agent slot, the last `num_stack` frames side by side (zeros for the
        frames before a reset), and the partner mask of the frame kept for `get_partner_mask`."""
        partner = self._get_partner_obs()
        obs = torch.cat((self._get_ego_state(), partner, self._get_road_map_obs()), dim=-1)
        if hasattr(self.sim, "response_type_tensor"):
            self.partner_mask = self.make_partner_mask(partner)
        if self.num_stack > 1:
            if reset or self.stacked_obs is None:
                prev = torch.zeros_like(obs).repeat(1, 1, self.num_stack - 1)
            else:
                prev = self.stacked_obs[..., obs.shape[-1]:]
            self.stacked_obs = torch.cat([prev, obs], dim=-1)
            obs = self.stacked_obs.clone()
        return obs if mask is None else obs[mask]

    # ---- masks and the optional sensors: env_torch.py:1224-1272, 898-945 ----
    def make_partner_mask(self, partner_observations):
        """Per ego and partner slot: 0 a partner that acts, 1 a `Static` one with a non-zero row, 2 nobody (id <= -1).  (The
        reference writes this fork's 127 partner slots as a literal; here it is max_agent_count - 1.)"""
        B, A, _ = partner_observations.shape
        partner_sum = partner_observations.reshape(B, A, A - 1, 6).sum(-1)
        static_mask = (self.sim.response_type_tensor().to_torch().clone().to(self.device) == 2).squeeze(-1)
        eye_mask = ~torch.eye(A, dtype=torch.bool)
        relative_static_mask = static_mask.unsqueeze(1).expand(-1, A, -1)[:, eye_mask].reshape(B, A, -1)
        filtered_static_mask = relative_static_mask & (partner_sum != 0)
        partner_ids = self.sim.partner_observations_tensor().to_torch().clone().to(self.device)[..., 8]
        return torch.where(filtered_static_mask, 1, torch.where(partner_ids <= -1, 2, 0))

    def get_partner_mask(self):
        return self.partner_mask.clone()

    def get_road_mask(self):
        """True where a road row is padding (id -1)."""
        return self.sim.agent_roadmap_tensor().to_torch().clone().to(self.device)[..., 7] == -1

    def _get_lidar_obs(self, mask=None):
        """[W, A, 3 planes, rays, 4] -> agent | road-edge | road-line samples side by side per agent."""
        lidar = self.sim.lidar_tensor().to_torch().clone().to(self.device)
        planes = [lidar[:, :, 0, :, :], lidar[:, :, 1, :, :], lidar[:, :, 2, :, :]]
        if mask is not None:
            return [p[mask] for p in planes]
        return torch.cat(planes, dim=-1).flatten(start_dim=2)

    def _get_bev_obs(self, mask=None):
        """The 200 x 200 entity-type raster, one-hot over Madrona's 11 entity types."""
        bev = torch.nn.functional.one_hot(self.sim.bev_observation_tensor().to_torch().clone().to(self.device).long(),
                                          num_classes=NUM_MADRONA_ENTITY_TYPES)
        return bev[mask].flatten(start_dim=1) if mask is not None else bev.flatten(start_dim=2)

    # ---- around an episode: env_torch.py:403-451, 462-467, 1218-1222 ----
    def _refresh_masks(self):
        self.cont_agent_mask = self.get_controlled_agents_mask()
        self.max_agent_count = self.cont_agent_mask.shape[1]
        self.num_valid_controlled_agents_across_worlds = self.cont_agent_mask.sum().item()

    def get_controlled_agents_mask(self):
        return (self.sim.controlled_state_tensor().to_torch().clone() == 1).squeeze(axis=2)

    def reset(self, mask=None, env_idx_list=None):
        """All worlds (or `env_idx_list`) back to their first step, the clock of every world to zero, the logged warm-up if
        `init_steps` was given; returns the observations (of the agents in `mask`, if given)."""
        if env_idx_list is None:
            env_idx_list = list(range(self.num_worlds))
        self.sim.reset(env_idx_list)
        self.world_time_steps.zero_()
        if self.init_steps > 0:
            self.advance_sim_with_log_playback(init_steps=self.init_steps)
        return self.get_obs(mask, reset=True)

    class Info:
        """gpudrive/datatypes/info.py:11-15: off_road = column 0, collided = columns 1 + 2, goal_achieved = column 3."""

        def __init__(self, info):
            self.off_road = info[:, :, 0]
            self.collided = info[:, :, 1:3].sum(axis=2)
            self.goal_achieved = info[:, :, 3]

        @property
        def shape(self):
            return self.off_road.shape

    def get_infos(self):
        return TorchCallSequence.Info(self.sim.info_tensor().to_torch().clone().to(self.device))

    # ---- the logged trajectories: env_torch.py:1445-1509, gpudrive/datatypes/trajectory.py:21-40 ----
    def get_expert_actions(self):
        """(inferred_actions, pos_xy, vel_xy, yaw, valids) over the 91 logged steps: the expert rows are 2 x 91 positions,
        2 x 91 velocities, 91 headings, 91 valid flags and 91 x 10 inferred action columns per agent slot; the inferred actions
        are clamped per dynamics model (classic / bicycle: acceleration +-6, steering +-0.3; delta_local: +-6, +-6, +-pi),
        `state` takes (x, y, 1, yaw, vx, vy, 0, 0, 0, 0) instead."""
        T = 91
        raw = self.sim.expert_trajectory_tensor().to_torch().clone()
        W, A = self.num_worlds, raw.shape[1]
        pos_xy = raw[:, :, :2 * T].view(W, A, T, -1)
        vel_xy = raw[:, :, 2 * T:4 * T].view(W, A, T, -1)
        yaw = raw[:, :, 4 * T:5 * T].view(W, A, T, -1)
        valids = raw[:, :, 5 * T:6 * T].view(W, A, T, -1).to(torch.int32)
        inferred = raw[:, :, 6 * T:16 * T].view(W, A, T, -1)
        if self.dynamics_model == "delta_local":
            act = inferred[..., :3]
            act[..., 0] = torch.clamp(act[..., 0], -6, 6)
            act[..., 1] = torch.clamp(act[..., 1], -6, 6)
            act[..., 2] = torch.clamp(act[..., 2], -torch.pi, torch.pi)
        elif self.dynamics_model == "state":
            ones = torch.ones((*pos_xy.shape[:-1], 1), device=raw.device)
            zeros = torch.zeros((*pos_xy.shape[:-1], 4), device=raw.device)
            act = torch.cat((pos_xy, ones, yaw, vel_xy, zeros), dim=-1)
        else:
            act = inferred[..., :3]
            act[..., 0] = torch.clamp(act[..., 0], -6, 6)
            act[..., 1] = torch.clamp(act[..., 1], -0.3, 0.3)
        return act, pos_xy, vel_xy, yaw, valids

    def advance_sim_with_log_playback(self, init_steps=0):
        if init_steps >= self.episode_len:
            raise ValueError("The length of the expert trajectory is 91,"
                             f"so init_steps = {init_steps} should be < than 91.")
        self.log_playback_traj, _, _, _, _ = self.get_expert_actions()
        for time_step in range(init_steps):
            self.step_dynamics(actions=self.log_playback_traj[:, :, time_step, :])

    # ---- the scenes under the worlds: env_torch.py:1295-1384, 1511-1537 ----
    def remove_agents_by_id(self, perc_to_rmv_per_scene, remove_controlled_agents=True, generator=None):
        """Deletes a share of every world's (controlled, or uncontrolled) agents by their ids through `sim.deleteAgents`, at
        least one per world that has any, then re-reads the controlled mask."""
        if perc_to_rmv_per_scene <= 0.0:
            return
        agent_ids = self.sim.self_observation_tensor().to_torch().clone()[:, :, 7]
        agent_mask = self.cont_agent_mask if remove_controlled_agents else (~self.cont_agent_mask) & (agent_ids != -1)
        for env_idx in range(self.num_worlds):
            scene_agent_ids = agent_ids[env_idx, :][agent_mask[env_idx]].long()
            if scene_agent_ids.numel() > 0:
                num_to_sample = max(1, int(perc_to_rmv_per_scene * scene_agent_ids.size(0)))
                sampled = scene_agent_ids[torch.randperm(scene_agent_ids.size(0), generator=generator)[:num_to_sample]]
                self.sim.deleteAgents({env_idx: sampled.tolist()})
        self._refresh_masks()

    def swap_data_batch(self, data_batch):
        """New scenes under every world (`sim.set_maps`), the controlled mask re-read."""
        if len(data_batch) != self.num_worlds:
            raise ValueError(f"Data batch size ({len(data_batch)}) does not match "
                             f"the expected number of worlds ({self.num_worlds}).")
        self.data_batch = data_batch
        self.sim.set_maps(self.data_batch)
        self._refresh_masks()

    def _names(self, tensor):
        ints = tensor.to_torch()
        return {i: "".join(chr(c) for c in ints[i].tolist() if c != 0) for i in range(self.num_worlds)}

    def get_env_filenames(self):
        return self._names(self.sim.map_name_tensor())

    def get_scenario_ids(self):
        return self._names(self.sim.scenario_id_tensor())
