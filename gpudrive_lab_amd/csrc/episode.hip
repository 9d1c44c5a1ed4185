// Episode bookkeeping on the device: what PufferGPUDrive.step() does
// with ~25 torch ops, two host synchronisations (`.item()`, `.cpu().numpy()`) and a host-driven reset
// per step (reference gpudrive/env/env_puffer.py:250-403; rewards gpudrive/env/env_torch.py:469-505;
// Info columns gpudrive/datatypes/info.py:11-15), as one kernel per step:
//
//   reward   = collision_w * (info[1] + info[2]) + goal_w * info[3] + off_road_w * info[0]
//              (weighted_combination) or the simulator's reward (sparse_on_goal_achieved)
//   terminal = done != 0
//   returns[live] += reward;  lengths += 1;  offroad += info[0];  collided += info[1] + info[2]
//   mask     = live (before the update);  live[terminal] = 0
//   truncated = !offroad && !collided && !goal_achieved
//   world done <=> every controlled agent is terminal: its episode sums go to `stats`, its trackers are
//   zeroed, live <- controlled, and its reset flag is raised ON THE DEVICE; the reset pass that follows
//   (gd_sim::reset_flagged) is launched unconditionally and returns at once when nothing was flagged.
// One workgroup per world, one thread per agent slot.
#include <hip/hip_runtime.h>

#include "engine.hpp"

namespace gd {

namespace {

template <int A_T>
__device__ __forceinline__ float block_sum(float v, float *scratch, int a) {
    // deterministic tree reduction (the statistics are compared bit for bit against the oracle per world)
    scratch[a] = v;
    __syncthreads();
#pragma unroll
    for (int s = A_T / 2; s > 0; s >>= 1) {
        if (a < s) scratch[a] += scratch[a + s];
        __syncthreads();
    }
    const float r = scratch[0];
    __syncthreads();
    return r;
}

template <int A_T>
__global__ __launch_bounds__(A_T) void k_episode_step(DevSim d, gd_episode_config c, gd_episode_buffers b) {
    const int w = blockIdx.x, a = threadIdx.x;
    const size_t i = (size_t)w * A_T + a;
    __shared__ float scratch[A_T];

    const int32_t *info = d.info + i * 5;
    const float off_road = (float)info[0];
    const float collided = (float)(info[1] + info[2]);
    const float goal = (float)info[3];
    const float reward = c.reward_type == GD_EPISODE_REWARD_SPARSE ? d.reward[i]
                                                                 : (c.collision_weight * collided + c.goal_achieved_weight * goal) +
                                                                       c.off_road_weight * off_road;
    const bool terminal = d.done[i] != 0;
    const bool controlled = b.controlled_mask[i] != 0;
    const bool live = b.live_agent_mask[i] != 0;

    float ret = b.agent_episode_returns[i];
    if (live) ret += reward;
    const float len = b.episode_lengths[i] + 1.f;
    const float off_ep = b.offroad_in_episode[i] + off_road;
    const float col_ep = b.collided_in_episode[i] + collided;
    const bool truncated = !(off_ep != 0.f) && !(col_ep != 0.f) && !(goal != 0.f);

    b.reward_out[i] = reward;
    b.terminal_out[i] = terminal ? 1 : 0;
    b.truncated_out[i] = truncated ? 1 : 0;
    b.mask_out[i] = live ? 1 : 0;

    const int n_controlled = __syncthreads_count(controlled);
    const int n_terminal = __syncthreads_count(controlled && terminal);
    const bool world_done = n_terminal == n_controlled;  // also true for a world without controlled agents
    if (world_done) {
        const float fc = controlled ? 1.f : 0.f;
        float sums[8];
        sums[0] = block_sum<A_T>(fc * ret, scratch, a);
        sums[1] = block_sum<A_T>(controlled && off_ep > 0.f ? 1.f : 0.f, scratch, a);
        sums[2] = block_sum<A_T>(controlled && col_ep > 0.f ? 1.f : 0.f, scratch, a);
        sums[3] = block_sum<A_T>(fc * goal, scratch, a);
        sums[4] = block_sum<A_T>(controlled && truncated ? 1.f : 0.f, scratch, a);
        sums[5] = block_sum<A_T>(len, scratch, a);
        sums[6] = block_sum<A_T>(col_ep, scratch, a);
        sums[7] = block_sum<A_T>(off_ep, scratch, a);
        if (a == 0) {
            atomicAdd(&b.stats[GD_EPISODE_STAT_EPISODES], 1.f);
            atomicAdd(&b.stats[GD_EPISODE_STAT_FINISHED_AGENTS], (float)n_controlled);
            atomicAdd(&b.stats[GD_EPISODE_STAT_RETURN_SUM], sums[0]);
            atomicAdd(&b.stats[GD_EPISODE_STAT_OFF_ROAD_AGENTS], sums[1]);
            atomicAdd(&b.stats[GD_EPISODE_STAT_COLLIDED_AGENTS], sums[2]);
            atomicAdd(&b.stats[GD_EPISODE_STAT_GOAL_ACHIEVED], sums[3]);
            atomicAdd(&b.stats[GD_EPISODE_STAT_TRUNCATED_AGENTS], sums[4]);
            atomicAdd(&b.stats[GD_EPISODE_STAT_LENGTH_SUM], sums[5]);
            atomicAdd(&b.stats[GD_EPISODE_STAT_TOTAL_COLLISIONS], sums[6]);
            atomicAdd(&b.stats[GD_EPISODE_STAT_TOTAL_OFF_ROAD], sums[7]);
            // per-world record of the episode that just ended (deterministic, unlike the running sums)
            float *ws = b.world_stats + (size_t)w * GD_EPISODE_STATS;
            ws[GD_EPISODE_STAT_EPISODES] = 1.f;
            ws[GD_EPISODE_STAT_FINISHED_AGENTS] = (float)n_controlled;
            ws[GD_EPISODE_STAT_RETURN_SUM] = sums[0];
            ws[GD_EPISODE_STAT_OFF_ROAD_AGENTS] = sums[1];
            ws[GD_EPISODE_STAT_COLLIDED_AGENTS] = sums[2];
            ws[GD_EPISODE_STAT_GOAL_ACHIEVED] = sums[3];
            ws[GD_EPISODE_STAT_TRUNCATED_AGENTS] = sums[4];
            ws[GD_EPISODE_STAT_LENGTH_SUM] = sums[5];
            ws[GD_EPISODE_STAT_TOTAL_COLLISIONS] = sums[6];
            ws[GD_EPISODE_STAT_TOTAL_OFF_ROAD] = sums[7];
            b.done_worlds[w] = 1;
            if (c.auto_reset) { d.reset_flags[w] = 1; *d.any_reset = 1; }
        }
        // env_puffer.py:381-391: empty the storage of the finished worlds
        b.agent_episode_returns[i] = 0.f;
        b.episode_lengths[i] = 0.f;
        b.offroad_in_episode[i] = 0.f;
        b.collided_in_episode[i] = 0.f;
        b.live_agent_mask[i] = controlled ? 1 : 0;
    } else {
        if (a == 0) b.done_worlds[w] = 0;
        b.agent_episode_returns[i] = ret;
        b.episode_lengths[i] = len;
        b.offroad_in_episode[i] = off_ep;
        b.collided_in_episode[i] = col_ep;
        b.live_agent_mask[i] = live && !terminal ? 1 : 0;
    }
}

}  // namespace

void launch_episode_step(const DevSim &d, hipStream_t st, const gd_episode_config &c, const gd_episode_buffers &b) {
    if (d.A == 64) hipLaunchKernelGGL(k_episode_step<64>, dim3(d.W), dim3(64), 0, st, d, c, b);
    else hipLaunchKernelGGL(k_episode_step<128>, dim3(d.W), dim3(128), 0, st, d, c, b);
}

}  // namespace gd
