"""CPU suite: sanity of the episode-bookkeeping restatement (oracle/episode.py, parity unpinned) on
properties PufferGPUDrive.step guarantees by construction (reference gpudrive/env/env_puffer.py:250-403)."""
import numpy as np

from tests.conftest import SCENE_4, TEST_JSON


def test_timeouts_finish_every_world_at_step_91(oracle_mod):
    from oracle.episode import OracleEpisodeTracker
    O = oracle_mod
    p = O.default_params(polylineReductionThreshold=0.5, observationRadius=10.0, collisionBehaviour=2, rewardType=1,
                         distanceToGoalThreshold=0.0, maxNumControlledAgents=2)
    sim = O.OracleSim([TEST_JSON, SCENE_4], p, max_agents=64)
    tr = OracleEpisodeTracker(sim)
    cm = tr.controlled_agent_mask
    assert cm.sum(axis=1).tolist() == [2, 2]
    for k in range(1, 95):
        reward, terminal, truncated, masks, done = tr.step()
        if k < 91:
            assert done.sum() == 0 and (tr.episode_lengths == k).all()
            assert np.isin(reward[cm], [0.0, -0.5, -1.0]).all()  # goal unreachable (thr 0): only the two penalties
        elif k == 91:
            assert done.tolist() == [1, 1] and terminal[cm].all()
            assert (tr.episode_lengths == 0).all() and (tr.agent_episode_returns == 0).all()
            assert np.array_equal(tr.live_agent_mask, cm)  # env_puffer.py:386-388
            assert (tr.world_stats[:, 0] == 1).all() and (tr.world_stats[:, 1] == 2).all()
            assert (tr.world_stats[:, 7] == 91 * 64).all()  # episode_lengths summed over all slots
            assert (np.array(sim.steps_remaining_tensor())[cm] == 91).all()  # the worlds were reset
        else:
            assert done.sum() == 0 and (tr.episode_lengths == k - 91).all()
