// wildfire_common.h — declarations shared by the wildfire kernels (lane-per-env and field/crew pairs) and the host C-ABI.
#pragma once

#include "frz_device.h"

#include <hip/hip_ext.h>

#include "../../include/frz.h"

#include <cstdlib>
#include <type_traits>

namespace frz_wf {

using frz::kBlock;

constexpr int kTotalsStride = 32;  // uint32 words per totals slot (128 B)
constexpr int kSmallCells = 64;    // cells of the env-per-lane kernels' per-cell tables; larger grids run wildfire_grid.hip (cells across lanes)

enum Mode { kStep = 0, kRebuild = 1, kReset = 2 };  // kReset (field/crew kernel): the configured initial state instead of loads, then rebuild

enum Flag : uint32_t {
    kStochIncrease = 1u << 0, kStochBurnouts = 1u << 1, kStochDecrease = 1u << 2, kUseFuel = 1u << 3, kStochSuppDecrease = 1u << 4,
    kStochRefill = 1u << 5, kStochSwitch = 1u << 6, kStochRepair = 1u << 7, kStochDegrade = 1u << 8, kCritical = 1u << 9,
    kShowBad = 1u << 10, kObsPower = 1u << 11, kObsSupp = 1u << 12, kPenaltyScaled = 1u << 13, kLocalize = 1u << 14,
    kTrackCumulative = 1u << 15, kTruncate = 1u << 16,
};

// Device-resident configuration block at arena offset 0.
// WfHot = the scalars / small tables the step kernels read: every workgroup stages it through LDS once and keeps what
// it uses in registers for the whole launch (in-kernel stamps showed scalar-cache loads at the point of use, each behind
// its own wait, costing more than the arithmetic: a 2x3 env step spent ~50 % of its cycles waiting on them).
struct WfHot {
    int32_t B, H, W, HW, A, S, K, nchunks, nch, others_k, max_steps, num_fire_states;
    uint32_t flags;
    float p_increase, p_burnout, p_decrease, decrease_bonus, p_supp_decrease, p_refill, p_switch, p_repair, p_degrade, p_critical;
    float spread_n, spread_w, spread_e, spread_s, random_ignition;
    float bad_attack_penalty, burnout_penalty, termination_reward, termination_kappa;
    float cum[FRZ_MAX_CAPACITIES];
    int32_t ay[FRZ_MAX_AGENTS], ax[FRZ_MAX_AGENTS];
    float power[FRZ_MAX_AGENTS];
    uint64_t has_n, has_w, has_e, has_s;  // cells that have a north/west/east/south neighbour
    // row indices of the [rows][B] blocks
    int32_t r_fires, r_intensity, r_fuel, r_supp, r_cap, r_equip, r_moves, r_burnouts, r_rewards, r_cum, r_atc, r_seeds, r_mti, n_rows4;
    int32_t q_burnouts, q_putouts, q_etc, n_rows8;
    int32_t u_term, u_trunc, u_frozen, n_rows1;
    int32_t roles;  // 1: field/crew wavefront-pair kernel (wildfire_roles.hip), 0: lane-per-env kernel
    int32_t pad_[3];
    // byte offsets from the arena base
    int64_t off_rows4, off_rows8, off_rows1, off_obs_self, off_obs_others, off_task_values, off_task_offsets, off_obs_map,
        off_act_values, off_act_offsets, off_bad_values, off_bad_offsets, off_mt_state, off_actions, off_error, off_epoch, off_totals,
        off_agg, off_prefix, off_rand_field, off_rand_agent, total_bytes, off_metrics;
    float fire_rewards[kSmallCells];
    int32_t ignition[kSmallCells];
    int32_t cell_yx[kSmallCells];  // (y << 16) | x
};
static_assert(sizeof(WfHot) % 16 == 0, "WfHot is staged with 16-byte copies");

// WfStaged = WfHot + the tables a lane indexes by its own state (they stay in LDS): one 16-byte load per thread stages it.
struct WfStaged : WfHot {
    float caps[FRZ_MAX_CAPACITIES];
    float eq[FRZ_MAX_EQUIPMENT_STATES][4];  // (capacity, power, range, -)
    uint64_t range_mask[FRZ_MAX_AGENTS][FRZ_MAX_EQUIPMENT_STATES];  // cells agent a reaches at equipment state s
    // the configured initial state (wildfire.py:347-354) of the field/crew kernels' grids (<= 24 cells): what a multi-step launch that
    // starts with a reset, or resets finished envs between its steps, puts in its registers
    int32_t init_fires[24], init_intensity[24], init_fuel[24];
    int32_t init_equipment;
    float init_suppressant, init_capacity;
    int32_t pad2_;
};
static_assert(sizeof(WfStaged) % 16 == 0 && sizeof(WfStaged) / 16 <= kBlock, "WfStaged is staged with one 16-byte copy per thread");

struct WfDev : WfStaged {
    int32_t initial_fuel, initial_equipment;
    float initial_suppressant, initial_capacity;
    int32_t fire_types[kSmallCells], lit[kSmallCells];
    int32_t grid;  // 1: one env per wavefront, cells across lanes (wildfire_grid.hip): the cell arrays are env-major [B][H*W]
};
static_assert(sizeof(WfDev) <= 8192, "configuration block too large");
constexpr int64_t kDevBlockBytes = 8192;

// What a step kernel needs before the configuration block is staged (passed by value: kernel arguments are there at
// wave start, so the state loads, the byte rows and the epoch/totals words are all in flight from the first instruction)
struct WfLaunch {
    int32_t batch;
    uint32_t policy;  // 1: sample the actions in-kernel (uniform random policy, frz_wildfire_step_random_policy)
    int64_t off_rows1, off_epoch, off_totals, off_mt_state;
    uint32_t policy_seed_lo, policy_seed_hi, policy_step_lo, policy_step_hi;
    int32_t* actions_out;  // where the sampled actions are left (policy == 1), int32 [A][B][2]
    uint32_t ticketed;     // 1: chunks are handed out in arrival order (more chunks than resident workgroups)
    uint32_t skip;         // diagnostic builds only (-DFRZ_WF_EXPERIMENT): store groups to leave out when timing; 0 in the product
    int32_t seed_increment;  // reset only (frz_wildfire_reset_reseed): added to every env seed
    int32_t n_steps;         // multi-step launches (wf_roles_kernel<..., PERSIST>): steps of the rollout this launch performs
    int64_t scratch_delta;   // multi-step launches: byte distance from the packed list buffers to their second copy
    double* metrics_out;     // multi-step launches: frz_wildfire_episode_metrics folded into the launch's tail (nullptr: not asked for)
    // frz_wildfire_rollout (multi-step launches): what drives the steps and what they leave behind besides the last step's outputs
    uint32_t rollout_flags;      // FRZ_ROLLOUT_* (include/frz.h)
    uint32_t seed_stride;        // FRZ_ROLLOUT_AUTO_RESET: added (mod 2^32) to the seed of an env each time the launch resets it
    int64_t tape_actions_step;   // elements between two steps of the action tape (`actions` = its step 0); 0: no tape
    int64_t list_record_delta;   // bytes from the arena's list block (off_task_offsets) to step 0's copy in the list record; 0: no record
    int64_t list_record_step;    // bytes between two steps' copies
    float* reward_tape;          // optional float32 [n_steps][A][B]: every step's rewards
    uint8_t* done_tape;          // optional uint8 [n_steps][2][B]: every step's (terminated, truncated)
    int64_t actions_out_step;    // elements between two steps of actions_out (0: every step overwrites the one buffer)
    float* supp_tape;            // optional float32 [n_steps][A][B]: every step's suppressants (frz_rollout_spec.obs_tape with FRZ_ROLLOUT_OBS_COMPACT)
    int32_t* state_tape;         // optional: n_steps copies of the state rows [3 HW + 3 A][B] (frz_rollout_spec.state_tape)
};

struct WfArgs {
    char* arena;
    const int32_t* actions;
    const float* field_rand;
    const float* agent_rand;
    const WfDev* host_dev;  // host copy of the configuration block (launch-side decisions)
    bool policy = false;    // fused uniform random policy (field/crew kernels only)
    uint64_t policy_seed = 0, policy_step = 0;
    int32_t* actions_out = nullptr;
    // optional: events that receive the step dispatch's own begin / end timestamps (frz_wildfire_step_random_policy_timed)
    hipEvent_t start_event = nullptr, stop_event = nullptr;
    bool ticketed = false;  // field/crew kernels: one workgroup per chunk, chunks handed out in arrival order
    int32_t seed_increment = 0;  // reset launches only
    int32_t n_steps = 1;         // > 1: one multi-step launch (frz_wildfire_rollout_random_policy, field/crew exact Philox kernels)
    int64_t scratch_delta = 0;
    double* metrics_out = nullptr;
    uint32_t rollout_flags = 0, seed_stride = 0;
    int64_t tape_actions_step = 0, list_record_delta = 0, list_record_step = 0, actions_out_step = 0;
    float* reward_tape = nullptr;
    uint8_t* done_tape = nullptr;
    float* supp_tape = nullptr;
    int32_t* state_tape = nullptr;
};

// The (CMAX, AMAX) instantiations of the step kernels: X(index, CMAX, AMAX, exact).  An env runs the first entry that holds its shape;
// exact entries (listed first) match only their own (cells, agents) pair: loops of exactly the shape's size and the random streams
// generated inside the step launch.  Both kernels (wildfire.hip, wildfire_roles.hip) are instantiated from this one list.
#define FRZ_WF_VARIANT_LIST(X)                                                                                                      \
    X(0, 6, 3, true) X(1, 6, 2, true) X(2, 9, 3, true) X(3, 9, 4, true) X(4, 8, 3, true) X(5, 8, 4, true) X(6, 12, 3, true)           \
    X(7, 12, 4, true) X(8, 16, 3, true) X(9, 16, 4, true) X(10, 4, 2, true) X(11, 4, 3, true) X(12, 8, 2, true) X(13, 9, 2, true)     \
    X(14, 12, 2, true) X(15, 16, 2, true) X(16, 6, 4, true)                                                                         \
    X(17, 8, 4, false) X(18, 16, 4, false) X(19, 8, 8, false) X(20, 16, 8, false) X(21, 24, 8, false)

#define FRZ_WF_ROLES_GROUPS 6  // translation units the field/crew instantiations are dealt to (wildfire_roles_g<k>.hip)

// launch a step kernel; with timing events the dispatch itself is bracketed (what a profiler's kernel trace reports)
template <typename K, typename... Args>
inline void launch_step_kernel(const WfArgs& a, K kernel, int grid, int block, hipStream_t stream, Args... args) {
    if (a.start_event && a.stop_event)
        hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, a.start_event, a.stop_event, 0, args...);
    else
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, args...);
}

inline uint32_t experiment_skip() {
#ifdef FRZ_WF_EXPERIMENT
    const char* v = std::getenv("FRZ_WF_SKIP");
    return v ? (uint32_t)std::atoi(v) : 0u;
#else
    return 0u;
#endif
}
#ifdef FRZ_WF_EXPERIMENT
#define FRZ_SKIP(bit) (EXACT && ((launch.skip >> (bit)) & 1u))  // (the exact-shape kernels only: the bench shape is what the experiments are about)
#define FRZ_WF_LOCAL_PROOF (!FRZ_SKIP(5))  // (A/B: every workgroup waits for the batch totals between the steps of a launch, as rounds 2-3 did)
#else
#define FRZ_SKIP(bit) false
#define FRZ_WF_LOCAL_PROOF true
#endif

inline WfLaunch make_launch(const WfArgs& a) {
    const WfDev* host = a.host_dev;
    return WfLaunch{host->B, a.policy ? 1u : 0u, host->off_rows1, host->off_epoch, host->off_totals, host->off_mt_state, (uint32_t)a.policy_seed,
                    (uint32_t)(a.policy_seed >> 32), (uint32_t)a.policy_step, (uint32_t)(a.policy_step >> 32), a.actions_out, a.ticketed ? 1u : 0u, experiment_skip(), a.seed_increment, a.n_steps, a.scratch_delta, a.metrics_out,
                    a.rollout_flags, a.seed_stride, a.tape_actions_step, a.list_record_delta, a.list_record_step, a.reward_tape, a.done_tape, a.actions_out_step, a.supp_tape, a.state_tape};
}

// Staging the configuration: the 16-byte piece is requested by the kernel's FIRST vector-memory instruction (before the
// state loads, so that waiting for it does not wait for them: the counter retires in issue order), parked in LDS, and the
// hot scalars come back by value: the fields a kernel uses end up in registers.
__device__ __forceinline__ uint4 stage_request(const WfDev* dev) {
    const uint4* src = reinterpret_cast<const uint4*>(static_cast<const WfStaged*>(dev));
    return threadIdx.x < sizeof(WfStaged) / 16 ? src[threadIdx.x] : make_uint4(0, 0, 0, 0);
}
__device__ __forceinline__ WfHot stage_commit(WfStaged& lds, const uint4& piece) {
    if (threadIdx.x < sizeof(WfStaged) / 16) reinterpret_cast<uint4*>(&lds)[threadIdx.x] = piece;
    __syncthreads();
    return lds;
}

// row access with a 32-bit element index: lets the compiler address as (uniform base) + (32-bit lane offset); assignment through the
// result is a write-through store (frz_device.h: a wavefront writes whole lines of a [rows][B] row)
template <typename T>
__device__ __forceinline__ frz::RowRef<T> at32(T* base, uint32_t index) {
    using Byte = std::conditional_t<std::is_const_v<T>, const char, char>;
    return frz::RowRef<T>{reinterpret_cast<T*>(reinterpret_cast<Byte*>(base) + (uint64_t)(index * (uint32_t)sizeof(T)))};
}

__device__ __forceinline__ float clamp01(float p) {
    p = p < 0.0f ? 0.0f : p;
    return p > 1.0f ? 1.0f : p;
}

template <typename M>
__device__ __forceinline__ int popc(M m) {
    if constexpr (sizeof(M) == 8)
        return __popcll(m);
    else
        return __popc(m);
}



// field/crew wavefront-pair kernels for grids of <= 8 cells (wildfire_roles.hip); variant as in wildfire.hip's table
int launch_roles(const WfArgs& args, int variant, int grid, int rng, int mode, hipStream_t stream);
// workgroups of the variant's multi-step kernel one CU holds at once (hipOccupancyMaxActiveBlocksPerMultiprocessor); 0: none / no device
int roles_persist_occupancy(int variant);

// ---- grids above 16 cells: one env per wavefront, cells across its lanes (wildfire_grid.hip) -------------------------------------
// Configuration of the grid kernels, passed BY VALUE as a kernel argument.  Tables indexed by a lane (per agent, per cell, range sets)
// live in the arena behind off_agent_table / off_cell_tables / off_range.
struct WgAgentTable {  // arena block read by the agent lanes
    float power[FRZ_MAX_AGENTS];
    int32_t ay[FRZ_MAX_AGENTS], ax[FRZ_MAX_AGENTS];
    float eq[FRZ_MAX_EQUIPMENT_STATES][4];  // (capacity, power, range, -) bonus per equipment state
    float caps[FRZ_MAX_CAPACITIES];
};
struct WgDev {
    int32_t B, H, W, HW, A, S, K, nchunks, others_k, max_steps, num_fire_states;
    uint32_t flags, inv_w;  // inv_w: c / W = (c * inv_w) >> 16 for c < H * W (checked at create)
    uint32_t inv_others;    // q / (A - 1) = (q * inv_others) >> 16 for q < A * (A - 1)
    float p_increase, p_burnout, p_decrease, decrease_bonus, p_supp_decrease, p_refill, p_switch, p_repair, p_degrade, p_critical;
    float spread_n, spread_w, spread_e, spread_s, random_ignition;
    float bad_attack_penalty, burnout_penalty, termination_reward, termination_kappa;
    float cum[FRZ_MAX_CAPACITIES];
    int32_t initial_fuel, initial_equipment;
    float initial_suppressant, initial_capacity;
    // rows of the [rows][B] blocks (shared with the helper kernels of wildfire.hip through the WfDev block at arena offset 0)
    int32_t r_supp, r_cap, r_equip, r_moves, r_burnouts, r_rewards, r_cum, r_atc, r_seeds, r_mti;
    int32_t q_burnouts, q_putouts, q_etc;
    int32_t u_term, u_trunc, u_frozen;
    int64_t off_rows4, off_rows8, off_rows1, off_cells /* fires, intensity, fuel: each int32 [B][HW] */, off_agent_table,
        off_cell_tables /* float fire_rewards[HW]; int32 ignition[HW], initial fires[HW], initial intensity[HW], initial fuel[HW] */,
        off_range /* uint64 [A][S][chunks]: cells agent a reaches at equipment state s, bit i of chunk k = cell 64 k + i */,
        off_obs_self, off_obs_others, off_task_values, off_task_offsets, off_obs_map, off_act_values, off_act_offsets, off_bad_values,
        off_bad_offsets, off_error, off_epoch, off_totals, off_agg, off_prefix,
        off_litmap /* uint64 [chunks][B]: the lit cells (fires > 0) of every env as mask words, left by wg_env_kernel for wg_lists_kernel */,
        off_okmap /* uint64 [A][chunks][B]: agent a's attackable cells (lit, in range at its equipment state, suppressant left) */,
        off_lit_cells /* int2 [B][HW rounded up to even]: (fires, intensity) of env b's lit cells in row-major (= task) order */;
    int32_t parity;  // overlapped rollouts (launch_cpl): which copy of the three arrays above this step's launches use (0 otherwise)
};
struct WgPolicy {  // the uniform random policy sampled inside the step launch
    uint32_t on, seed_lo, seed_hi, step_lo, step_hi;
    int32_t* actions_out;
};
// overlapped steps of a grid-family rollout (wildfire_grid.hip launch_cpl): the lists of step t on a second stream beside the env launch of step t + 1
struct WgOverlap {
    hipStream_t side = nullptr;                            // the second stream (owned by the handle)
    hipEvent_t scan_done = nullptr, lists_done = nullptr;  // scan of this step enqueued on the step's stream / lists of this step enqueued on `side`
    int64_t copy_delta = 0;                                // bytes from the mask words / lit cells to their second copies (steps of odd parity)
    int parity = 0;
    bool wait_previous_lists = false;                      // the previous step's lists launch is (possibly) still reading the offsets this step's scan rewrites
};
int launch_grid(const WgDev& dev, char* arena, const WfArgs& args, int rng, int mode, bool ticketed, hipStream_t stream, const WgOverlap* overlap = nullptr);

}  // namespace frz_wf
