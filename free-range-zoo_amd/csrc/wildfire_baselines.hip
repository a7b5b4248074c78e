// wildfire_baselines.hip — the scripted wildfire baselines as device-side policies.
//
// Reference: free_range_zoo/envs/wildfire/baselines/strongest.py:32-62 and weakest.py (per-env Python loops over the
// padded task observation).  One thread per env here, on the jagged observation buffers themselves.  Reproduced as written:
//   * an agent whose action mapping is empty in EVERY env answers [-1, -1] everywhere (strongest.py:41-43);
//   * an env with an empty mapping answers [-1, -1] (:50-52);
//   * the candidates are the intensities of the env's FIRST n task rows, n = length of the agent's action mapping in that env
//     (:47-48 index the padded task tensor by position, not through the mapping);
//   * ties are broken uniformly (:54-57); the reference draws from torch's global generator, here the draw is
//     word 0 of Philox4x32-10(counter (env, 0, step, step >> 32), key (seed, seed >> 32)): member floor(u32 * ties / 2^32);
//   * agents without suppressant (observation self[:, 3] == 0) turn their answer into a noop, keeping the index (:62).
#include "frz_device.h"

#include "../../include/frz.h"

namespace {

__global__ void __launch_bounds__(frz::kBlock) wf_extreme_policy_kernel(const int64_t* __restrict__ task_values, const int64_t* __restrict__ task_offsets,
                                                                         const int64_t* __restrict__ map_offsets, const int64_t* __restrict__ map_lengths,
                                                                         const float* __restrict__ obs_self, int64_t B, int weakest, uint32_t seed_lo,
                                                                         uint32_t seed_hi, uint32_t step_lo, uint32_t step_hi, int64_t first_env,
                                                                         int32_t* __restrict__ actions) {
    const int64_t b = (int64_t)blockIdx.x * frz::kBlock + threadIdx.x;
    if (b >= B) return;
    int32_t idx = -1, act = -1;
    const bool any_mapping = map_offsets[B] - map_offsets[0] > 0;
    const int64_t n = map_lengths[b];
    if (any_mapping && n > 0) {
        const int64_t* rows = task_values + task_offsets[b] * 4;
        int64_t best = rows[3];
        int ties = 1;
        for (int64_t k = 1; k < n; ++k) {
            const int64_t v = rows[k * 4 + 3];
            const bool better = weakest ? v < best : v > best;
            ties = better ? 1 : (v == best ? ties + 1 : ties);
            best = better ? v : best;
        }
        const frz::Philox4 w = frz::philox4x32_10((uint32_t)(b + first_env), 0u, step_lo, step_hi, seed_lo, seed_hi);
        int pick = (int)(((uint64_t)w.w[0] * (uint64_t)ties) >> 32);
        for (int64_t k = 0; k < n; ++k) {
            if (rows[k * 4 + 3] == best) {
                if (pick == 0) {
                    idx = (int32_t)k;
                    break;
                }
                --pick;
            }
        }
        act = 0;
    }
    if (any_mapping && obs_self[b * 4 + 3] == 0.0f) act = -1;
    reinterpret_cast<int2*>(actions)[b] = make_int2(idx, act);
}

}  // namespace

extern "C" int frz_wildfire_extreme_fire_policy(const int64_t* task_values, const int64_t* task_offsets, const int64_t* map_offsets,
                                                const int64_t* map_lengths, const float* obs_self, int64_t parallel_envs, int weakest,
                                                uint64_t seed, uint64_t step, int64_t first_env_index, int32_t* actions_out, void* stream) {
    if (!task_values || !task_offsets || !map_offsets || !map_lengths || !obs_self || !actions_out || parallel_envs <= 0) return FRZ_E_INVALID;
    const int blocks = (int)((parallel_envs + frz::kBlock - 1) / frz::kBlock);
    hipLaunchKernelGGL(wf_extreme_policy_kernel, dim3(blocks), dim3(frz::kBlock), 0, static_cast<hipStream_t>(stream), task_values, task_offsets,
                       map_offsets, map_lengths, obs_self, parallel_envs, weakest, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step,
                       (uint32_t)(step >> 32), first_env_index, actions_out);
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}
