// rideshare_baselines.hip — the scripted rideshare baselines as one device-side policy.
//
// Reference: free_range_zoo/envs/rideshare/baselines/greedy_Tfocus.py:48-115, greedy_Tglobal.py:48-99, fifo_Tfocus.py:36-80,
// fifo_Tglobal.py:36-74 (per-env Python loops over the padded task observation).  One thread per env here, on the jagged
// observation rows (y, x, y_dest, x_dest, accepted_by, riding_by, fare, entered_step) themselves.  Reproduced as written:
//   * no mapped task in the env -> [-1, -1] (an observation without any task answers [-1, -1] everywhere);
//   * the candidates are the env's FIRST n task rows, n = length of the agent's action mapping in that env;
//   * greedy key = trip length + my distance to the passenger in float32, |dy| + |dx| on the 4-connected grid and
//     sqrt(dy^2 + dx^2) with diagonal travel (the agents build MovementTransition with fast_travel=True,
//     transitions/movement.py:56-86); greedy_Tglobal keeps the key in an int64 buffer: truncated toward zero;
//   * fifo key = entered_step;
//   * Tfocus: with an accepted row in the env every row that is not accepted gets the largest key, then with a riding row every
//     row that is not riding does (FLT_MAX greedy, +inf fifo) — an env showing both ends with all keys equal (greedy_Tfocus
//     asserts on those observations; the arithmetic carries on here);
//   * uniform draw among the rows holding the minimum: tie_draws[b] when the caller supplies the member (replaying
//     torch.randint draws), else floor(u32 * ties / 2^32) of word 0 of Philox4x32-10(counter (first_env + b, 0, step lo,
//     step hi), key (seed lo, seed hi));
//   * second component from the chosen row: riding 2 (drop), accepted 1 (pick), otherwise 0 (accept).
#include "frz_device.h"

#include "../../include/frz.h"

#include <cfloat>

namespace {

struct Row {
    int4 lo, hi;  // (y, x, y_dest, x_dest), (accepted_by, riding_by, fare, entered_step)
};

__device__ __forceinline__ Row load_row(const int32_t* __restrict__ rows, int64_t k) {
    const int4* p = reinterpret_cast<const int4*>(rows + k * 8);
    return Row{p[0], p[1]};
}

template <int KIND, bool DIAGONAL>
__device__ __forceinline__ float row_key(const Row& r, int my_y, int my_x, bool any_accepted, bool any_riding) {
    constexpr bool greedy = KIND < 2, focus = (KIND & 1) == 0;
    float key;
    if constexpr (greedy) {
        const float ty = (float)(r.lo.z - r.lo.x), tx = (float)(r.lo.w - r.lo.y);
        const float my = (float)(r.lo.x - my_y), mx = (float)(r.lo.y - my_x);
        const float trip = DIAGONAL ? sqrtf(ty * ty + tx * tx) : fabsf(ty) + fabsf(tx);
        const float mine = DIAGONAL ? sqrtf(my * my + mx * mx) : fabsf(my) + fabsf(mx);
        key = trip + mine;
    } else {
        key = (float)r.hi.w;
    }
    if constexpr (focus) {
        const float masked = greedy ? FLT_MAX : INFINITY;
        if (any_accepted && r.hi.x < 0) key = masked;
        if (any_riding && r.hi.y < 0) key = masked;
    }
    if constexpr (KIND == 1) key = (float)(int64_t)key;
    return key;
}

template <int KIND, bool DIAGONAL>
__global__ void __launch_bounds__(frz::kBlock) rs_task_policy_kernel(const int32_t* __restrict__ task_values, const int64_t* __restrict__ task_offsets,
                                                                      const int64_t* __restrict__ task_lengths,
                                                                      const int64_t* __restrict__ map_lengths, const int32_t* __restrict__ obs_self,
                                                                      int64_t B, uint32_t seed_lo, uint32_t seed_hi, uint32_t step_lo,
                                                                      uint32_t step_hi, int64_t first_env, const int64_t* __restrict__ tie_draws,
                                                                      int32_t* __restrict__ actions) {
    const int64_t b = (int64_t)blockIdx.x * frz::kBlock + threadIdx.x;
    if (b >= B) return;
    int32_t idx = -1, act = -1;
    const int64_t count = task_lengths[b];
    const int64_t n = min(map_lengths[b], count);
    if (n > 0) {
        const int32_t* rows = task_values + task_offsets[b] * 8;
        const int2 me = reinterpret_cast<const int2*>(obs_self)[b * 2];
        bool any_accepted = false, any_riding = false;
        if constexpr ((KIND & 1) == 0) {
            for (int64_t k = 0; k < count; ++k) {
                const int4 hi = reinterpret_cast<const int4*>(rows + k * 8)[1];
                any_accepted |= hi.x >= 0;
                any_riding |= hi.y >= 0;
            }
        }
        float best = row_key<KIND, DIAGONAL>(load_row(rows, 0), me.x, me.y, any_accepted, any_riding);
        int ties = 1;
        for (int64_t k = 1; k < n; ++k) {
            const float v = row_key<KIND, DIAGONAL>(load_row(rows, k), me.x, me.y, any_accepted, any_riding);
            ties = v < best ? 1 : (v == best ? ties + 1 : ties);
            best = v < best ? v : best;
        }
        int64_t pick;
        if (tie_draws) {
            pick = tie_draws[b];
        } else {
            const frz::Philox4 w = frz::philox4x32_10((uint32_t)(b + first_env), 0u, step_lo, step_hi, seed_lo, seed_hi);
            pick = (int64_t)(((uint64_t)w.w[0] * (uint64_t)ties) >> 32);
        }
        for (int64_t k = 0; k < n; ++k) {
            const Row r = load_row(rows, k);
            if (row_key<KIND, DIAGONAL>(r, me.x, me.y, any_accepted, any_riding) == best) {
                if (pick == 0) {
                    idx = (int32_t)k;
                    act = r.hi.y >= 0 ? 2 : (r.hi.x >= 0 ? 1 : 0);
                    break;
                }
                --pick;
            }
        }
    }
    reinterpret_cast<int2*>(actions)[b] = make_int2(idx, act);
}

}  // namespace

extern "C" int frz_rideshare_task_policy(const int32_t* task_values, const int64_t* task_offsets, const int64_t* task_lengths,
                                         const int64_t* map_lengths, const int32_t* obs_self, int64_t parallel_envs, int kind, int diagonal,
                                         uint64_t seed, uint64_t step, int64_t first_env_index, const int64_t* tie_draws, int32_t* actions_out,
                                         void* stream) {
    if (!task_values || !task_offsets || !task_lengths || !map_lengths || !obs_self || !actions_out || parallel_envs <= 0) return FRZ_E_INVALID;
    if (kind < 0 || kind > 3) return FRZ_E_INVALID;
    if ((reinterpret_cast<uintptr_t>(task_values) & 15) || (reinterpret_cast<uintptr_t>(obs_self) & 7) || (reinterpret_cast<uintptr_t>(actions_out) & 7))
        return FRZ_E_INVALID;  // rows are read as two 16-byte words
    const int blocks = (int)((parallel_envs + frz::kBlock - 1) / frz::kBlock);
    const hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t seed_lo = (uint32_t)seed, seed_hi = (uint32_t)(seed >> 32), step_lo = (uint32_t)step, step_hi = (uint32_t)(step >> 32);
#define FRZ_RS_POLICY(K, D)                                                                                                               \
    hipLaunchKernelGGL((rs_task_policy_kernel<K, D>), dim3(blocks), dim3(frz::kBlock), 0, s, task_values, task_offsets, task_lengths,      \
                       map_lengths, obs_self, parallel_envs, seed_lo, seed_hi, step_lo, step_hi, first_env_index, tie_draws, actions_out)
    switch (kind * 2 + (diagonal ? 1 : 0)) {
        case 0: FRZ_RS_POLICY(0, false); break;
        case 1: FRZ_RS_POLICY(0, true); break;
        case 2: FRZ_RS_POLICY(1, false); break;
        case 3: FRZ_RS_POLICY(1, true); break;
        case 4: FRZ_RS_POLICY(2, false); break;
        case 5: FRZ_RS_POLICY(2, true); break;
        case 6: FRZ_RS_POLICY(3, false); break;
        default: FRZ_RS_POLICY(3, true); break;
    }
#undef FRZ_RS_POLICY
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}
