// cybersecurity_baselines.hip — the stateful scripted cybersecurity baselines as one device-side policy.
//
// Reference: free_range_zoo/envs/cybersecurity/baselines/patched.py:33-71 (kind 0 PatchedAttackerBaseline), exploited.py:39-88
// (1 ExploitedAttackerBaseline), patched.py:96-152 (2 PatchedDefenderBaseline), exploited.py:120-167 (3 ExploitedDefenderBaseline),
// camp.py:32-60 (4 CampDefenderBaseline): torch ops plus a per-env Python loop for the tie-break.  One thread per env here; the
// agent's state (target_node, time_focused, actions) lives in caller-owned device arrays that the kernel updates in place.
// Reproduced as written:
//   * an action mapping without any element answers [-100, -1] everywhere and leaves the state alone;
//   * the candidates are observation['tasks'][:, 0]: row_len values tasks[b * env_stride + k * elem_stride] — on the env's
//     [B, N, F] observation the F features of node 0 (what the reference computes); the strides also express "feature 0 of
//     every node";
//   * key: 0 value, min; 1 value with (subnetwork_states - 1) -> -100, max; 2 value with -100 and 0 -> 1000, min; 3 value, max;
//     one uniform draw per env among the positions holding the extreme: tie_draws[b] when given (replayed torch.randint
//     draws), else floor(u32 * ties / 2^32) of word 0 of Philox4x32-10(counter (first_env + b, 0, step lo, step hi), key (seed));
//   * attackers adopt the new target when they have none, defenders whenever the row holds no -100;
//   * answers and the three-step focus counter as in the oracle restatement (oracle/frz_oracle_rng.c), fills applied in the
//     reference's order (camp: an absent agent standing on its node still answers -2).
#include "frz_device.h"

#include "../../include/frz.h"

namespace {

template <int KIND>
__device__ __forceinline__ int64_t focus_key(int64_t x, int32_t states) {
    if constexpr (KIND == 1) return x == states - 1 ? -100 : x;
    if constexpr (KIND == 2) return (x == -100 || x == 0) ? 1000 : x;
    return x;
}

template <int KIND>
__global__ void __launch_bounds__(frz::kBlock) cy_focus_policy_kernel(const int64_t* __restrict__ tasks, int64_t env_stride, int64_t elem_stride,
                                                                       int32_t row_len, const float* __restrict__ obs_self, int32_t self_width,
                                                                       int64_t B, int32_t states, int32_t camp_target, int mapping_empty,
                                                                       uint32_t seed_lo, uint32_t seed_hi, uint32_t step_lo, uint32_t step_hi,
                                                                       int64_t first_env, const int64_t* __restrict__ tie_draws,
                                                                       int32_t* __restrict__ target_node, int32_t* __restrict__ time_focused,
                                                                       int32_t* __restrict__ actions) {
    const int64_t b = (int64_t)blockIdx.x * frz::kBlock + threadIdx.x;
    if (b >= B) return;
    int2* answer = reinterpret_cast<int2*>(actions) + b;
    if (mapping_empty) {
        *answer = make_int2(-100, -1);
        return;
    }
    const float* me = obs_self + b * self_width;
    const bool absent = me[1] == 0.0f;
    int32_t a1 = answer->y;
    if constexpr (KIND == 4) {
        const bool at = me[2] == (float)camp_target;
        a1 = at ? -2 : (absent ? -1 : 0);
        *answer = make_int2(camp_target, a1);
        return;
    } else {
        constexpr bool want_max = KIND == 1 || KIND == 3, defender = KIND >= 2;
        const int64_t* row = tasks + b * env_stride;
        int64_t best = 0;
        int ties = 0;
        bool monitored = true;
        for (int32_t k = 0; k < row_len; ++k) {
            const int64_t x = row[k * elem_stride];
            monitored &= x != -100;
            const int64_t key = focus_key<KIND>(x, states);
            const bool better = k == 0 || (want_max ? key > best : key < best);
            ties = better ? 1 : (key == best ? ties + 1 : ties);
            best = better ? key : best;
        }
        int64_t pick;
        if (tie_draws) {
            pick = tie_draws[b];
        } else {
            const frz::Philox4 w = frz::philox4x32_10((uint32_t)(b + first_env), 0u, step_lo, step_hi, seed_lo, seed_hi);
            pick = (int64_t)(((uint64_t)w.w[0] * (uint64_t)ties) >> 32);
        }
        int32_t fresh = 0;
        for (int32_t k = 0; k < row_len; ++k) {
            if (focus_key<KIND>(row[k * elem_stride], states) == best) {
                if (pick == 0) {
                    fresh = k;
                    break;
                }
                --pick;
            }
        }
        int32_t target = target_node[b], focused = time_focused[b];
        if (defender ? monitored : target == -1) target = fresh;
        const bool targeted = target != -1 && !absent, targetless = target == -1 && !absent;
        if constexpr (!defender) {
            if (targeted) a1 = 0;
            if (absent) a1 = -1;
            if (targeted) ++focused;
        } else {
            const bool at = me[2] == (float)target;
            if (targeted && !at) a1 = 0;
            if (absent) a1 = -1;
            if (targeted && at) a1 = -2;
            if (targetless && !monitored) a1 = -3;
            if (targeted && at) ++focused;
        }
        *answer = make_int2(target, a1);
        if (focused >= 3) target = -1, focused = 0;
        target_node[b] = target;
        time_focused[b] = focused;
    }
}

}  // namespace

extern "C" int frz_cybersecurity_focus_policy(const int64_t* tasks, int64_t env_stride, int64_t elem_stride, int32_t row_len,
                                              const float* obs_self, int32_t self_width, int64_t parallel_envs, int kind,
                                              int32_t subnetwork_states, int32_t camp_target, int64_t mapping_numel, uint64_t seed, uint64_t step,
                                              int64_t first_env_index, const int64_t* tie_draws, int32_t* target_node, int32_t* time_focused,
                                              int32_t* actions, void* stream) {
    if (!tasks || !obs_self || !target_node || !time_focused || !actions || parallel_envs <= 0 || kind < 0 || kind > 4) return FRZ_E_INVALID;
    if (row_len <= 0 || env_stride < 0 || elem_stride < 0 || self_width < (kind >= 2 ? 3 : 2)) return FRZ_E_INVALID;
    if (reinterpret_cast<uintptr_t>(actions) & 7) return FRZ_E_INVALID;
    const int blocks = (int)((parallel_envs + frz::kBlock - 1) / frz::kBlock);
    const hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t seed_lo = (uint32_t)seed, seed_hi = (uint32_t)(seed >> 32), step_lo = (uint32_t)step, step_hi = (uint32_t)(step >> 32);
    const int mapping_empty = mapping_numel == 0;
#define FRZ_CY_POLICY(K)                                                                                                                    \
    hipLaunchKernelGGL((cy_focus_policy_kernel<K>), dim3(blocks), dim3(frz::kBlock), 0, s, tasks, env_stride, elem_stride, row_len, obs_self, \
                       self_width, parallel_envs, subnetwork_states, camp_target, mapping_empty, seed_lo, seed_hi, step_lo, step_hi,          \
                       first_env_index, tie_draws, target_node, time_focused, actions)
    switch (kind) {
        case 0: FRZ_CY_POLICY(0); break;
        case 1: FRZ_CY_POLICY(1); break;
        case 2: FRZ_CY_POLICY(2); break;
        case 3: FRZ_CY_POLICY(3); break;
        default: FRZ_CY_POLICY(4); break;
    }
#undef FRZ_CY_POLICY
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}
