// rideshare.hip — rideshare environment step for gfx950 (MI355X): one environment per WAVEFRONT, passenger slots across its lanes.
//
// One ParallelEnv.step() of the reference (rideshare.py:248-467) = three stream-ordered launches:
//   rs_env_kernel      (four envs per workgroup: a wavefront per env for the passenger table, one crew wavefront for the agents of all
//                                      four) action decode through the open action mapping (optionally the uniform random policy, sampled in
//                                      the launch) -> movement -> passenger state (accept-conflict resolution, picks) -> passenger exit
//                                      (drops, fares, ordered in-place compaction by ballot + lane rank) -> passenger entry (schedule)
//                                      -> rewards -> truncation -> agent observations, per-env task counts
//   rs_offsets_kernel  (lane per env)  launch-wide exclusive prefix sums of the counts (frz_scan.h) -> the jagged offsets + the
//                                      batch totals the next step's freeze test reads
//   rs_emit_kernel     (wave per env)  update_actions / update_observations: the env's task rows staged once in LDS, every list
//                                      (all passengers, each agent's visible ones) compacted by ballot + lane rank and written as
//                                      CONTIGUOUS 16-byte pieces (one wave store = up to 1 KiB of consecutive bytes)
//
// The reference keeps one global passenger table sorted by env and re-sorts / boolean-compacts it every step; here each env owns
// max_passengers slots in table order, env-major records [B][slot][10]: lane s of the env's wavefront holds slot s (and s + 64 when an
// env has more than 64 slots) and moves its 40-byte record with three wide accesses (the columns a step can change come first, so a
// changed slot is two stores).  In the env launch the per-agent quantities of a workgroup's four envs live in the lanes of ONE of its
// wavefronts (the crew: lane 16 e + a = agent a of env e) and what crosses between slot lanes and agent lanes is an LDS word — masks,
// per-slot claim / effect words, the agents' moves — across three workgroup barriers; in the emit launch a wavefront is one env and the
// crossings are ballots, v_writelane and ds_bpermute.  Deterministic integer/byte work, HBM-bound: no MFMA.
#include "frz_scan.h"
#include "frz_wave.h"

#include "../../include/frz.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <type_traits>
#include <utility>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

using frz::kBlock;
using frz::at32;
using frz::for_each_index;
using frz::from_lane;
using frz::lane_rank;
using frz::last_bit;
using frz::popc_words;
using frz::read_lane;
using frz::select_nth;
using frz::wave_lds_sync;
using frz::write_lane_c;

enum Mode { kStep = 0, kRebuild = 1 };
enum Flag : uint32_t { kFast = 1u << 0, kDiagonal = 1u << 1, kVariableMove = 1u << 2, kWaiting = 1u << 3, kTrackCumulative = 1u << 4, kTruncate = 1u << 5 };
// a passenger record: bytes [0, 16) (y, x, state, driver) and [16, 24) (accepted, picked) are what a step can change, [24, 40) never changes
enum Col { PY = 0, PX, PSTATE, PDRIVER, PACCEPTED, PPICKED, PYD, PXD, PFARE, PENTERED, PCOLS };
constexpr int kNone = -100;
constexpr int kEnvsPerBlock = kBlock / 64;  // one env per wavefront

struct RsDev {
    int32_t B, A, P, nchunks, max_steps, pool_limit, long_wait_time, schedule_rows, max_time, first_env_index;
    int32_t wait_limit[3];
    uint32_t flags;
    float move_cost, drop_cost, noop_cost, accept_cost, pool_limit_cost, general_wait_cost, long_wait_cost;
    int32_t start_y[FRZ_MAX_AGENTS], start_x[FRZ_MAX_AGENTS];
    uint32_t inv_others;  // ceil(2^16 / (A - 1)): q / (A - 1) = (q * inv_others) >> 16 for the q < A * (A - 1) the kernels divide
    int32_t r_count, r_moves, r_rewards, r_cum, r_atc, n_rows4;
    int32_t u_term, u_trunc, u_frozen, n_rows1;
    int64_t off_rows4, off_rows1, off_agents, off_passengers, off_etc, off_obs_self, off_obs_others, off_task_values, off_task_offsets,
        off_agent_task_values, off_agent_map_values, off_agent_offsets, off_agent_task_states, off_schedule, off_schedule_index,
        off_actions, off_error, off_epoch, off_totals, off_agg, off_prefix, total_bytes;
};
constexpr int64_t kDevBlockBytes = 4096;
static_assert(sizeof(RsDev) <= kDevBlockBytes, "configuration block too large");

// the uniform random policy sampled inside rs_env_kernel (frz_rideshare_step_random_policy)
struct RsPolicy {
    uint32_t on, seed_lo, seed_hi, step_lo, step_hi;
    int32_t* actions_out;
};

// one passenger record (40 bytes, 8-byte aligned) of an env's table, as three accesses with immediate offsets
typedef int int4u __attribute__((ext_vector_type(4), aligned(8)));  // records are 8-byte aligned
typedef int int2u __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void load_record(const int32_t* table, uint32_t slot, int (&v)[10]) {
    const char* r = reinterpret_cast<const char*>(table) + (uint64_t)(slot * 40u);
    const int4u a = *reinterpret_cast<const int4u*>(r), c = *reinterpret_cast<const int4u*>(r + 24);
    const int2u m = *reinterpret_cast<const int2u*>(r + 16);
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = m.x, v[5] = m.y, v[6] = c.x, v[7] = c.y, v[8] = c.z, v[9] = c.w;
}
__device__ __forceinline__ void store_hot(int32_t* table, uint32_t slot, const int (&v)[10]) {  // what a step can change
    char* r = reinterpret_cast<char*>(table) + (uint64_t)(slot * 40u);
    *reinterpret_cast<int4u*>(r) = int4u{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<int2u*>(r + 16) = int2u{v[4], v[5]};
}
__device__ __forceinline__ void store_cold(int32_t* table, uint32_t slot, const int (&v)[10]) {
    *reinterpret_cast<int4u*>(reinterpret_cast<char*>(table) + (uint64_t)(slot * 40u) + 24) = int4u{v[6], v[7], v[8], v[9]};
}

// schedule rows of one timestep are contiguous in the time-sorted device copy: [index[t], index[t + 1])
struct Schedule {
    const int32_t* rows;   // [S][7] stable-sorted by timestep
    const int32_t* index;  // [max_time + 2]
    int max_time;
};

// transitions/passenger_entry.py:24-72 for one env: the rows of timestep t (this env or wildcard) are appended in schedule order behind
// the `count` passengers of the table; 64 schedule rows per pass, their places by ballot + lane rank.  The loads of the first pass are
// issued by entry_prefetch (long before the table is ready for them), the records are written by entry_commit.
struct EntryPlan {
    int first, last;  // the timestep's rows [first, last) of the time-sorted schedule
    int row[7];       // lane's row of the first pass (first + lane)
};
__device__ __forceinline__ EntryPlan entry_prefetch(const Schedule& sch, int t) {
    EntryPlan plan;
    plan.first = plan.last = 0;
    if (t >= 0 && t <= sch.max_time) {
        plan.first = sch.index[t];
        plan.last = sch.index[t + 1];
    }
    const int lane = threadIdx.x & 63;
    const int r = plan.first + lane < plan.last ? plan.first + lane : plan.first;
    const bool any = plan.first < plan.last;
#pragma unroll
    for (int c = 0; c < 7; ++c) plan.row[c] = any ? at32(sch.rows, (uint32_t)r * 7u + (uint32_t)c) : 0;
    return plan;
}
// returns the number of passengers appended
__device__ __forceinline__ int entry_commit(const Schedule& sch, const EntryPlan& plan, int32_t* table, int P, int b, int t, int count, uint32_t& err) {
    const int lane = threadIdx.x & 63;
    int appended = 0;
    for (int r0 = plan.first; r0 < plan.last; r0 += 64) {
        int row[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) row[c] = plan.row[c];
        const bool in = r0 + lane < plan.last;
        if (r0 != plan.first) {
            const int r = in ? r0 + lane : plan.first;
#pragma unroll
            for (int c = 0; c < 7; ++c) row[c] = at32(sch.rows, (uint32_t)r * 7u + (uint32_t)c);
        }
        const bool match = in && (row[1] == -1 || row[1] == b);
        const uint64_t m = __ballot(match);
        if (m == 0) continue;
        const int pos = count + appended + lane_rank(m);
        if (match) {
            if (pos < P) {
                const int v[10] = {row[2], row[3], 0, -1, -1, -1, row[4], row[5], row[6], t};  // state 0, no driver, never accepted / picked
                store_hot(table, (uint32_t)pos, v);
                store_cold(table, (uint32_t)pos, v);
            } else {
                err |= FRZ_ERR_OVERFLOW;
            }
        }
        appended = min(appended + (int)__popcll(m), P - count);
    }
    return appended;
}

// rideshare.py:185-222 + utils/env.py:137-160: agents at their start positions, bookkeeping zeroed, step-0 passengers enter
__global__ void __launch_bounds__(kBlock) rs_fill_kernel(char* __restrict__ arena, const RsDev d) {
    const int b = frz::env_of_wave<kEnvsPerBlock>(d.B);
    if (b < 0) return;
    const int lane = threadIdx.x & 63;
    const int64_t B = d.B;
    int32_t* const rows = reinterpret_cast<int32_t*>(arena + d.off_rows4);
    float* const rowsf = reinterpret_cast<float*>(arena + d.off_rows4);
    uint8_t* const rows1 = reinterpret_cast<uint8_t*>(arena + d.off_rows1);
    int32_t* const pas = reinterpret_cast<int32_t*>(arena + d.off_passengers) + (int64_t)b * PCOLS * d.P;
    if (lane < d.A) {
        reinterpret_cast<int2*>(arena + d.off_agents)[(int64_t)b * d.A + lane] = make_int2(d.start_y[lane], d.start_x[lane]);
        rowsf[(d.r_rewards + lane) * B + b] = 0.0f;
        rowsf[(d.r_cum + lane) * B + b] = 0.0f;
        rows1[(d.u_term + lane) * B + b] = 0;
        rows1[(d.u_trunc + lane) * B + b] = 0;
    }
    const Schedule sch{reinterpret_cast<const int32_t*>(arena + d.off_schedule), reinterpret_cast<const int32_t*>(arena + d.off_schedule_index),
                       d.max_time};
    uint32_t err = 0;
    const int count = entry_commit(sch, entry_prefetch(sch, 0), pas, d.P, b, 0, 0, err);
    if (lane == 0) {
        rows[d.r_moves * B + b] = 0;
        rows1[d.u_frozen * B + b] = 0;
        rows[d.r_count * B + b] = count;
    }
    if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// rs_env_kernel: everything of a step that concerns the FOUR envs of a workgroup.  SPL = slots per lane (1: up to 64 passenger slots per
// env, 2: up to 128).
//   * every wavefront is the FIELD of one env: "slot-lane" values, lane s holds slot s + 64 * k in element k.  It loads the table, turns
//     it into masks (who drives what, the three passenger states) and parks what an agent may ask about a passenger in LDS; after the crew
//     has decided it applies the effects (accepted / picked / dropped, riding passengers follow their driver), compacts the table in
//     place, lets the next timestep's passengers enter and evaluates the waiting costs;
//   * ONE wavefront of the workgroup (blockIdx % envs: the four SIMDs share that work) is also the CREW of all four envs: lane 16 e + a
//     holds agent a of env e.  It decodes the actions through the open action mapping (or samples the uniform policy), moves the agents,
//     settles the accept conflicts, and after the fields are done computes rewards, truncation, the counts of the rebuilt spaces and the
//     agent observations.  (Round 2's kernel had every wavefront do both for its own env: the agent phases ran at 8 of 64 lanes, four
//     times per workgroup, and a launch of these kernels is bound by the instructions its wavefronts issue.)
// The fields and the crew meet at three workgroup barriers and exchange through LDS: masks, the per-slot claim / effect words, the agents'
// moves, the fields' counts.
// ------------------------------------------------------------------------------------------------------------------------------------
struct RsFieldOut {  // what an env's field leaves for the crew once the table is updated
    int count, entered;
    float global;  // the waiting costs every agent of the env pays
    int pad_;
};
enum TargetCol { TSTATE = 0, TY, TX, TYD, TXD, TFARE, TCOLS };

template <int AMAX, int SPL, int MODE>
__global__ void __launch_bounds__(kBlock) rs_env_kernel(char* __restrict__ arena, const RsDev d, const int32_t* __restrict__ actions,
                                                         const RsPolicy pol) {
    constexpr int W = 2 * SPL;  // 32-bit words of a slot mask
    constexpr int E = kEnvsPerBlock, AP = 64 / E, S = 64 * SPL;
    static_assert(AMAX <= AP, "a crew lane per agent");
    __shared__ uint32_t s_claim[E][S];              // accepting agents per slot
    __shared__ uint32_t s_effect[E][S];             // (winning agent + 1) << 8 | picked << 1 | dropped << 2
    __shared__ int s_target[E][S][TCOLS];           // what an agent reads of the passenger it chose (state before the step)
    __shared__ uint32_t s_driven[E][AMAX][W];       // the passengers each agent drives, as mask words
    __shared__ uint64_t s_mask[E][4][SPL];          // unaccepted / accepted / riding / kept slots; after the fields' second phase: of the new table
    __shared__ int s_move[E][AP];                   // the agents' moves (y | x << 16)
    __shared__ RsFieldOut s_out[E];
    __shared__ int4 s_self[E][AMAX];                // the agents' self observation rows
    const int b = frz::env_of_wave<E>(d.B);
    if (b < 0) return;  // before any barrier: a barrier does not wait for wavefronts that have ended
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t B = d.B;
    const int A = d.A, P = d.P;
    const uint32_t flags = d.flags;
    const int b0 = b - wave;                                         // the workgroup's first env
    const int n_envs = (int)(B - b0 < E ? B - b0 : E);
    const bool crew = wave == (int)(blockIdx.x % (uint32_t)n_envs);  // wave-uniform
    const int ce = lane / AP, ca = lane % AP;                        // crew lane -> (env of the workgroup, agent)
    const bool is_agent = ca < A && ce < n_envs;
    const int agent = ca < A ? ca : A - 1;                           // clamped: unconditional loads
    const int cb = b0 + (ce < n_envs ? ce : n_envs - 1);             // the crew lane's env
    int32_t* const rows = reinterpret_cast<int32_t*>(arena + d.off_rows4);
    float* const rowsf = reinterpret_cast<float*>(arena + d.off_rows4);
    uint8_t* const rows1 = reinterpret_cast<uint8_t*>(arena + d.off_rows1);
    int32_t* const pas = reinterpret_cast<int32_t*>(arena + d.off_passengers) + (int64_t)b * PCOLS * P;  // this env's table [slot][column]
    int2* const agents = reinterpret_cast<int2*>(arena + d.off_agents);
    const uint32_t Bu = (uint32_t)B, bu = (uint32_t)b, cbu = (uint32_t)cb;

    // ---------------------------------------------------------------- loads, field: everything the step reads of the env's table,
    // requested before anything is waited for except the passenger count (lanes past it re-read the last live record: the same cache
    // lines, no extra traffic)
    const uint32_t epoch = *reinterpret_cast<const uint32_t*>(arena + d.off_epoch);
    const uint32_t* const totals = reinterpret_cast<const uint32_t*>(arena + d.off_totals);
    const uint32_t left0[2] = {totals[A + 1], totals[A + 2]}, left1[2] = {totals[frz::kTotalsStride + A + 1], totals[frz::kTotalsStride + A + 2]};
    const int count0 = at32(rows, (uint32_t)d.r_count * Bu + bu);
    const int nm = MODE == kStep ? at32(rows, (uint32_t)d.r_moves * Bu + bu) : 0;
    int v[SPL][PCOLS];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int slot = lane + 64 * k;
        load_record(pas, (uint32_t)(slot < count0 ? slot : (count0 > 0 ? count0 - 1 : 0)), v[k]);
    }
    const Schedule sch{reinterpret_cast<const int32_t*>(arena + d.off_schedule), reinterpret_cast<const int32_t*>(arena + d.off_schedule_index),
                       d.max_time};
    EntryPlan plan;
    if (MODE == kStep) plan = entry_prefetch(sch, nm + 1);  // the next timestep's schedule rows (rideshare.py:308): in flight early
    // ---------------------------------------------------------------- loads, crew (lane 16 e + a: agent a of the workgroup's env e)
    int2 pos0 = make_int2(0, 0);
    int act_idx = 0, act_id = -1, c_nm = 0;
    float cum0 = 0.0f;
    bool trunc = false;
    if (crew) {
        pos0 = at32(agents, cbu * (uint32_t)A + (uint32_t)agent);
        trunc = at32(rows1, (uint32_t)d.u_trunc * Bu + cbu) != 0;
        if (MODE == kStep) {
            c_nm = at32(rows, (uint32_t)d.r_moves * Bu + cbu);
            if (!pol.on) {
                const int2 a2 = at32(reinterpret_cast<const int2*>(actions), (uint32_t)agent * Bu + cbu);
                act_idx = a2.x;
                act_id = is_agent ? a2.y : -1;
            }
            if (flags & kTrackCumulative) cum0 = at32(rowsf, (uint32_t)(d.r_cum + agent) * Bu + cbu);
        }
    }

    if (MODE == kStep) {
        // utils/env.py:211-213 (terminations never set, rideshare.py:252): frozen once every env is truncated.  Channels A + 1 / A + 2 of
        // the batch totals rs_offsets_kernel left after the previous step = number of envs not terminated / not truncated.  (Batch
        // totals: every wavefront takes this branch or none.)
        const bool odd = ((epoch + 1u) & 1u) != 0;
        const uint32_t left_alive = odd ? left1[0] : left0[0], left_running = odd ? left1[1] : left0[1];
        if (left_alive == 0u || left_running == 0u) {
            // the parallel adapter sums the stale rewards once per agent call (utils/conversions.py:87-90)
            if (crew && !at32(rows1, (uint32_t)d.u_frozen * Bu + cbu)) {
                if (is_agent) {
                    const float r = at32(rowsf, (uint32_t)(d.r_rewards + agent) * Bu + cbu);
                    float acc = 0.0f;
                    for (int j = 0; j < A; ++j) acc = acc + r;
                    at32(rowsf, (uint32_t)(d.r_rewards + agent) * Bu + cbu) = acc;
                }
                wave_lds_sync();
                if (ca == 0 && ce < n_envs) at32(rows1, (uint32_t)d.u_frozen * Bu + cbu) = 1;
            }
            return;
        }
    }

    // ---------------------------------------------------------------- field: the table as masks
    bool live[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        live[k] = lane + 64 * k < count0;
        v[k][PSTATE] = live[k] ? v[k][PSTATE] : -1;  // lanes past the count hold no passenger
        v[k][PDRIVER] = live[k] ? v[k][PDRIVER] : -2;
    }
    uint64_t st0[SPL], st1[SPL], st2[SPL];  // unaccepted / accepted / riding passengers (wave-uniform masks)
    uint64_t kept_mask[SPL];                // the slots that stay in the table
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        st0[k] = __ballot(v[k][PSTATE] == 0);
        st1[k] = __ballot(v[k][PSTATE] == 1);
        st2[k] = __ballot(v[k][PSTATE] == 2);
        kept_mask[k] = __ballot(live[k]);
    }
    // the passengers each agent drives (slot 64 k + 32 h + i = bit i of word 2 k + h), for the crew: every driven slot sets its bit in
    // its driver's words (one LDS atomic per slot lane; a ballot per agent was eight times the instructions)
    if (lane < AMAX * W) (&s_driven[wave][0][0])[lane] = 0u;
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int drv = v[k][PDRIVER];
        if (drv >= 0 && drv < AMAX) atomicOr(&s_driven[wave][drv][2 * k + (lane >> 5)], 1u << (lane & 31));
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < SPL; ++k) s_mask[wave][0][k] = st0[k], s_mask[wave][1][k] = st1[k], s_mask[wave][2][k] = st2[k], s_mask[wave][3][k] = kept_mask[k];
    }
    uint32_t err = 0;
    int count = count0, entered = 0;
    float global = 0.0f;

    // crew state that lives across the barriers
    uint32_t driven[W];
#pragma unroll
    for (int i = 0; i < W; ++i) driven[i] = 0u;
    int ay = pos0.x, ax = pos0.y;
    float cost = 0.0f;
    int fare = 0, tslot = 0;
    bool wins = false;

    if (MODE == kStep) {
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            const int slot = lane + 64 * k;
            s_claim[wave][slot] = 0u, s_effect[wave][slot] = 0u;  // the LDS words of this env's slots start clear
            int* const t = s_target[wave][slot];
            t[TSTATE] = v[k][PSTATE], t[TY] = v[k][PY], t[TX] = v[k][PX], t[TYD] = v[k][PYD], t[TXD] = v[k][PXD], t[TFARE] = v[k][PFARE];
        }
        __syncthreads();

        if (crew) {
            // -------------------------------------------------------- (1) crew: action decode (rideshare.py:256-300)
            // the agent's action mapping lists, in table order, the passengers that are unaccepted or its own (rideshare.py:378-392): entry i
            // of the mapping = the i-th set bit of (unaccepted | driven)
#pragma unroll
            for (int i = 0; i < W; ++i) driven[i] = is_agent ? s_driven[ce][agent][i] : 0u;
            uint32_t seen[W];
#pragma unroll
            for (int k = 0; k < SPL; ++k) {
                const uint64_t u = s_mask[ce][0][k];
                seen[2 * k] = driven[2 * k] | (uint32_t)u;
                seen[2 * k + 1] = driven[2 * k + 1] | (uint32_t)(u >> 32);
            }
            const int n_visible = popc_words(seen);
            if (pol.on) {
                // uniform member of OneOf([Discrete(1, start=state_t) for visible task t] + [noop]) (spaces/actions.py:10-50), the stream
                // of frz_rideshare_random_policy: word 0 of Philox(counter (agent, global env, step), key seed), all agents at once
                const frz::Philox4 w = frz::philox4x32_10((uint32_t)agent, (uint32_t)(cb + d.first_env_index), pol.step_lo, pol.step_hi, pol.seed_lo, pol.seed_hi);
                act_idx = (int)(((uint64_t)w.w[0] * (uint64_t)(n_visible + 1)) >> 32);
            }
            const int target = is_agent ? select_nth(seen, act_idx) : -1;  // slot of the chosen passenger, -1 = none
            const int* const chosen = s_target[ce][target < 0 ? 0 : target];  // its record as the field parked it
            const int t_state = chosen[TSTATE];
            if (pol.on) {
                act_id = (is_agent && target >= 0) ? t_state : -1;
                if (is_agent) at32(reinterpret_cast<int2*>(pol.actions_out), (uint32_t)agent * Bu + cbu) = make_int2(act_idx, act_id);
            }
            // act_idx inside the mapping <=> target >= 0 (lanes that are no agent carry act_id -1 and target -1)
            if (__ballot(act_id != -1 && target < 0)) err |= FRZ_ERR_BAD_ACTION_INDEX;  // the reference reads a garbage row
            const int kind = target >= 0 ? act_id : -1;
            const bool accept = kind == 0, pick = kind == 1, drop = kind == 2;
            const bool has_vec = (uint32_t)kind <= 2u;
            // goal of the task vector: the passenger's position, or its destination for a drop — from the state BEFORE movement
            const int t_y = chosen[TY], t_x = chosen[TX], t_yd = chosen[TYD], t_xd = chosen[TXD], t_fare = chosen[TFARE];
            const int gy = drop ? t_yd : t_y, gx = drop ? t_xd : t_x;
            // -------------------------------------------------------- (2) crew: movement (transitions/movement.py:56-116)
            int my = 0, mx = 0;
            uint32_t dist2 = 0;  // squared pre-move distance to the goal (coordinates within +-16383: fits): sqrt is monotonic, zero iff zero
            {
                const int dy = ay - gy, dx = ax - gx;
                uint32_t best = (uint32_t)(dy * dy) + (uint32_t)(dx * dx);
                dist2 = has_vec ? best : 0u;
                int by = 0, bx = 0;
                if (flags & kFast) {
                    by = -dy;
                    bx = -dx;
                } else {  // first minimum over {stay, N, E, S, W(, NW, NE, SE, SW)}
                    const int cy[9] = {0, -1, 0, 1, 0, -1, -1, 1, 1}, cx[9] = {0, 0, 1, 0, -1, -1, 1, 1, -1};
                    const int ndirs = (flags & kDiagonal) ? 9 : 5;
#pragma unroll
                    for (int k = 1; k < 9; ++k) {
                        if (k < ndirs) {
                            const int ey = dy + cy[k], ex = dx + cx[k];
                            const uint32_t e = (uint32_t)(ey * ey) + (uint32_t)(ex * ex);
                            const bool better = e < best;
                            best = better ? e : best;
                            by = better ? cy[k] : by;
                            bx = better ? cx[k] : bx;
                        }
                    }
                }
                my = has_vec ? by : 0;
                mx = has_vec ? bx : 0;
                const float fy = (float)my, fx = (float)mx;
                // sqrtf, not __fsqrt_rn: the intrinsic is the bare v_sqrt_f32 (1 ulp) on this target, sqrtf the correctly rounded sequence the
                // reference's CPU norm gives (a fast-travel diagonal move such as (5, 9) differed in the last bit: found by tests/test_hip_fuzz.py)
                cost = (flags & kDiagonal) ? sqrtf(__fadd_rn(__fmul_rn(fy, fy), __fmul_rn(fx, fx))) : __fadd_rn(fabsf(fy), fabsf(fx));
                ay += my;
                ax += mx;
            }
            // -------------------------------------------------------- (3) crew: accept conflicts (passenger_state.py:54-74)
            // while a passenger is claimed by several accepting agents, per env only the closest of ALL contested agents keeps its claim
            // (lowest index on ties); uncontested accepts survive.  One pass settles an env.  Claims are counted per slot in LDS.
            tslot = target < 0 ? 0 : target;
            if (accept) atomicAdd(&s_claim[ce][tslot], 1u);
            wave_lds_sync();
            const bool contested = accept && s_claim[ce][tslot] > 1u;
            int winner = -1;
            if (__ballot(contested)) {  // rare; every lane looks at the agents of its own env, in agent order
                uint32_t best = 0;
#pragma unroll
                for (int o = 0; o < AMAX; ++o) {
                    const bool c_o = from_lane(ce * AP + o, contested ? 1 : 0) != 0;
                    const uint32_t d_o = (uint32_t)from_lane(ce * AP + o, (int)dist2);
                    const bool better = o < A && c_o && (winner < 0 || d_o < best);
                    best = better ? d_o : best;
                    winner = better ? o : winner;
                }
            }
            wins = accept && (!contested || ca == winner);
            const bool picked = pick && dist2 == 0;   // distance < 1e-6: the agent already stood on the passenger (:88-90)
            const bool dropped = drop && dist2 == 0;  // transitions/passenger_exit.py:43-46
            fare = dropped ? t_fare : 0;
            // what this step does to each slot, and where the agents go (riding passengers follow their driver)
            const uint32_t effect = (wins ? (uint32_t)(ca + 1) << 8 : 0u) | (picked ? 2u : 0u) | (dropped ? 4u : 0u);
            if (effect) atomicOr(&s_effect[ce][tslot], effect);
            s_move[ce][ca] = is_agent ? ((my & 0xFFFF) | (mx << 16)) : 0;
        }
        __syncthreads();

        // ------------------------------------------------------------ (2b/3/4) field: riding passengers follow their driver, winners
        // accept, picks ride, drops leave (order-preserving compaction: a kept slot's new place = its rank among the kept ones)
        uint64_t taken_mask[SPL], boarded_mask[SPL];
        int place[SPL];
        bool keep[SPL], moved[SPL], taken[SPL], boarded[SPL];
        int kept = 0;
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            const uint32_t e = s_effect[wave][lane + 64 * k];
            // best_moves[env, driver]; driver -1 wraps to the last agent like Python's index (movement.py:107-108)
            const int drv0 = v[k][PDRIVER] < 0 ? A + v[k][PDRIVER] : v[k][PDRIVER];
            const bool rides = v[k][PSTATE] == 2 && drv0 >= 0 && drv0 < A;
            const int word = s_move[wave][rides ? drv0 : 0];
            const int sy = (int)(short)(word & 0xFFFF), sx = word >> 16;
            moved[k] = rides && word != 0;
            v[k][PY] += rides ? sy : 0;
            v[k][PX] += rides ? sx : 0;
            taken[k] = live[k] && (e >> 8) != 0u;
            if (taken[k]) {
                v[k][PSTATE] = 1;
                v[k][PACCEPTED] = nm;
                v[k][PDRIVER] = (int)(e >> 8) - 1;
            }
            boarded[k] = live[k] && (e & 2u) != 0u;
            if (boarded[k]) {
                v[k][PSTATE] = 2;
                v[k][PPICKED] = nm;
            }
            keep[k] = live[k] && !(e & 4u);
            kept_mask[k] = __ballot(keep[k]);
            taken_mask[k] = __ballot(taken[k]);
            boarded_mask[k] = __ballot(boarded[k]);
            place[k] = kept + lane_rank(kept_mask[k]);
            kept += (int)__popcll(kept_mask[k]);
        }
        // every load of the table has landed before the first store into it (a slot is only ever written at or below its own index,
        // i.e. where ANOTHER lane's value was read from)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            // a slot is rewritten only where it changes: the whole record once a removal has shifted it, otherwise its changeable part
            const bool shifted = keep[k] && place[k] != lane + 64 * k;
            if (shifted) store_cold(pas, (uint32_t)place[k], v[k]);
            if (shifted || (keep[k] && (moved[k] || taken[k] || boarded[k]))) store_hot(pas, (uint32_t)place[k], v[k]);
        }
        // ------------------------------------------------------------ (5) field: entry of the next timestep (rideshare.py:308)
        entered = entry_commit(sch, plan, pas, P, b, nm + 1, kept, err);
        count = kept + entered;
        // the new table as masks over the OLD slot numbers (counts do not care)
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            st0[k] = st0[k] & ~taken_mask[k] & ~boarded_mask[k] & kept_mask[k];
            st1[k] = ((st1[k] | taken_mask[k]) & ~boarded_mask[k]) & kept_mask[k];
            st2[k] = (st2[k] | boarded_mask[k]) & kept_mask[k];
        }
        // ------------------------------------------------------------ (6) field: the waiting costs of the env (rideshare.py:310-339)
        int unaccepted = entered;
#pragma unroll
        for (int k = 0; k < SPL; ++k) unaccepted += (int)__popcll(st0[k]);
        if (flags & kWaiting) {
            // `global_rewards[envs] += cost` is an index_put without accumulation: per statement only the LAST passenger (table order) of
            // the env in that state takes effect (:323-333); the passengers that just entered are the last unaccepted ones
            auto last_wait = [&](const uint64_t (&mask)[SPL], int since_col, bool& any) {
                int wait = 0;
                any = false;
#pragma unroll
                for (int k = 0; k < SPL; ++k)
                    if (mask[k]) {
                        wait = nm - read_lane(v[k][since_col], last_bit(mask[k]));
                        any = true;
                    }
                return wait;
            };
            bool any0, any1, any2;
            int w0 = last_wait(st0, PENTERED, any0);
            const int w1 = last_wait(st1, PACCEPTED, any1), w2 = last_wait(st2, PPICKED, any2);
            if (entered > 0) w0 = nm - (nm + 1), any0 = true;
            if (any0) global = __fadd_rn(global, __fmul_rn(w0 >= d.wait_limit[0] ? 1.0f : 0.0f, d.general_wait_cost));
            if (any1) global = __fadd_rn(global, __fmul_rn(w1 >= d.wait_limit[1] ? 1.0f : 0.0f, d.general_wait_cost));
            if (any2) global = __fadd_rn(global, __fmul_rn(w2 >= d.wait_limit[2] ? 1.0f : 0.0f, d.general_wait_cost));
            if (any0) global = __fadd_rn(global, __fmul_rn(w0 >= d.long_wait_time ? 1.0f : 0.0f, d.long_wait_cost));
            const int slots = A * d.pool_limit;
            global = __fadd_rn(global, __fmul_rn(__fmul_rn(unaccepted >= slots - count ? 1.0f : 0.0f, -0.5f), (float)(slots - count)));
        }
        if (lane == 0) {
            at32(rows, (uint32_t)d.r_moves * Bu + bu) = nm + 1;
            at32(rows, (uint32_t)d.r_count * Bu + bu) = count;
#pragma unroll
            for (int k = 0; k < SPL; ++k) s_mask[wave][0][k] = st0[k], s_mask[wave][1][k] = st1[k], s_mask[wave][2][k] = st2[k], s_mask[wave][3][k] = kept_mask[k];
        }
    }
    if (lane == 0) {
        s_out[wave] = RsFieldOut{count, entered, global, 0};
        reinterpret_cast<int64_t*>(arena + d.off_etc)[b] = count;
    }
    __syncthreads();
    if (!crew) {
        if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);
        return;
    }

    // ---------------------------------------------------------------- crew: rewards (rideshare.py:340-363), truncation, the counts of
    // update_actions / update_observations that are not lists (rideshare.py:427-463), self = (y, x, #accepted, #riding), others = the
    // other agents' self rows
    const RsFieldOut out = s_out[ce < n_envs ? ce : 0];
    uint64_t n0[SPL], n1[SPL], n2[SPL], nk[SPL];  // the new table's masks
#pragma unroll
    for (int k = 0; k < SPL; ++k) n0[k] = s_mask[ce][0][k], n1[k] = s_mask[ce][1][k], n2[k] = s_mask[ce][2][k], nk[k] = s_mask[ce][3][k];
    if (MODE == kStep) {
        // an accepted passenger had no driver or this one (it was visible to the agent): the winner's bit joins its driven set
        {
            const uint32_t bit = wins ? 1u << (tslot & 31) : 0u;
            const int word = tslot >> 5;
#pragma unroll
            for (int i = 0; i < W; ++i) driven[i] |= word == i ? bit : 0u;
        }
        int owned = 0;  // passengers whose driver is this agent, any state (rideshare.py:343-344)
#pragma unroll
        for (int k = 0; k < SPL; ++k) owned += __popc(driven[2 * k] & (uint32_t)nk[k]) + __popc(driven[2 * k + 1] & (uint32_t)(nk[k] >> 32));
        float r = 0.0f;
        r = __fadd_rn(r, owned > d.pool_limit ? d.pool_limit_cost : 0.0f);
        r = __fadd_rn(r, __fmul_rn(act_id == -1 ? 1.0f : 0.0f, d.noop_cost));
        r = __fadd_rn(r, __fmul_rn(act_id == 0 ? 1.0f : 0.0f, d.accept_cost));  // the accept ACTION, won or not
        r = __fadd_rn(r, fare > 0 ? __fsub_rn((float)fare, d.drop_cost) : 0.0f);
        float dr = __fmul_rn(cost, d.move_cost);
        if (flags & kVariableMove) dr = __fdiv_rn(dr, (float)(owned + 1));
        r = __fadd_rn(r, dr);
        const float reward = __fadd_rn(r, out.global);
        trunc = (flags & kTruncate) ? c_nm + 1 >= d.max_steps : trunc;
        if (is_agent) {
            at32(rowsf, (uint32_t)(d.r_rewards + agent) * Bu + cbu) = reward;
            if (flags & kTrackCumulative) at32(rowsf, (uint32_t)(d.r_cum + agent) * Bu + cbu) = __fadd_rn(cum0, reward);
            if (flags & kTruncate) at32(rows1, (uint32_t)(d.u_trunc + agent) * Bu + cbu) = (uint8_t)trunc;
            at32(agents, cbu * (uint32_t)A + (uint32_t)agent) = make_int2(ay, ax);
        }
    } else {
#pragma unroll
        for (int i = 0; i < W; ++i) driven[i] = is_agent ? s_driven[ce][agent][i] : 0u;
    }
    int n_accepted = 0, n_riding = 0, visible = out.entered;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const uint32_t lo = driven[2 * k] & (uint32_t)nk[k], hi = driven[2 * k + 1] & (uint32_t)(nk[k] >> 32);
        n_accepted += __popc(lo & (uint32_t)n1[k]) + __popc(hi & (uint32_t)(n1[k] >> 32));
        n_riding += __popc(lo & (uint32_t)n2[k]) + __popc(hi & (uint32_t)(n2[k] >> 32));
        // visible = unaccepted or driven by the agent
        visible += __popc(lo | (uint32_t)n0[k]) + __popc(hi | (uint32_t)(n0[k] >> 32));
    }
    const int4 self = make_int4(ay, ax, n_accepted, n_riding);
    if (is_agent) {
        at32(reinterpret_cast<int4*>(arena + d.off_obs_self), (uint32_t)agent * Bu + cbu) = self;
        at32(rows, (uint32_t)(d.r_atc + agent) * Bu + cbu) = visible;
        s_self[ce][agent] = self;
    }
    wave_lds_sync();
    {   // others[a][j] = self of the j-th other agent: one lane per (agent, other) pair
        int4* const obs_others = reinterpret_cast<int4*>(arena + d.off_obs_others);
        const int others = A - 1, pairs = A * others;
        for (int e = 0; e < n_envs; ++e)
            for (int q = lane; q < pairs; q += 64) {
                const int a = (int)(((uint32_t)q * d.inv_others) >> 16), j = q - a * others;
                at32(obs_others, ((uint32_t)a * Bu + (uint32_t)(b0 + e)) * (uint32_t)others + (uint32_t)j) = s_self[e][j < a ? j : j + 1];
            }
    }
    if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// rs_offsets_kernel: one env per lane.  Exclusive prefix sums over the batch of (passengers, visible tasks of agent 0, 1, ...) = where
// each env's segment of each jagged list starts; the batch totals (and the number of envs not yet truncated) stay for the next step.
// ------------------------------------------------------------------------------------------------------------------------------------
template <int AMAX>
__global__ void __launch_bounds__(kBlock) rs_offsets_kernel(char* __restrict__ arena, const RsDev d, uint32_t ticketed) {
    __shared__ frz::ScanShared<AMAX + 1> s_scan;
    __shared__ int s_ticket;
    const int tid = threadIdx.x;
    const int64_t B = d.B;
    const int A = d.A;
    frz::ScanWorkspace ws{reinterpret_cast<uint32_t*>(arena + d.off_epoch), reinterpret_cast<uint32_t*>(arena + d.off_totals),
                          reinterpret_cast<uint64_t*>(arena + d.off_agg), reinterpret_cast<uint64_t*>(arena + d.off_prefix)};
    const frz::ScanLaunch launch = frz::scan_begin(ws);
    const int chunk = frz::scan_take_chunk(ws, d.nchunks, ticketed != 0, &s_ticket);
    const int64_t b = (int64_t)chunk * kBlock + tid;
    const bool active = b < B;
    const int64_t bl = active ? b : B - 1;
    const int32_t* rows = reinterpret_cast<const int32_t*>(arena + d.off_rows4);
    const uint8_t* rows1 = reinterpret_cast<const uint8_t*>(arena + d.off_rows1);
    uint32_t cnt[AMAX + 1], excl[AMAX + 1];
    cnt[0] = active ? (uint32_t)rows[(int64_t)d.r_count * B + bl] : 0u;
#pragma unroll
    for (int a = 0; a < AMAX; ++a) cnt[a + 1] = (active && a < A) ? (uint32_t)rows[(int64_t)(d.r_atc + (a < A ? a : 0)) * B + bl] : 0u;
    const bool trunc = rows1[(int64_t)d.u_trunc * B + bl] != 0;
    uint32_t err = 0;
    frz::scan_chunk<AMAX + 1>(s_scan, ws, launch, cnt, active, active && !trunc, A + 1, chunk, d.nchunks, excl, &err);
    if (active) {
        int64_t* const task_offsets = reinterpret_cast<int64_t*>(arena + d.off_task_offsets);
        int64_t* const agent_offsets = reinterpret_cast<int64_t*>(arena + d.off_agent_offsets);
        task_offsets[b] = excl[0];
        if (b == B - 1) task_offsets[B] = (int64_t)excl[0] + cnt[0];
#pragma unroll
        for (int a = 0; a < AMAX; ++a)
            if (a < A) {
                agent_offsets[a * (B + 1) + b] = excl[a + 1];
                if (b == B - 1) agent_offsets[a * (B + 1) + B] = (int64_t)excl[a + 1] + cnt[a + 1];
            }
    }
    if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);
    frz::scan_end(ws, launch, chunk, d.nchunks);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// rs_emit_kernel: the lists of update_actions / update_observations (rideshare.py:367-467) for ONE env per wavefront.  The env's task
// rows (y, x, y_dest, x_dest, accepted_by | -100, riding_by | -100, fare, entered) are built once and parked in LDS.  The list of all
// passengers is those rows in order.  The agents' lists (the tasks an agent sees: unaccepted or its own, rideshare.py:378-392, 446-456)
// are short and many, so they are written as ONE flat sequence: every (agent, visible slot) pair gets a place by ballot + lane rank,
// the places of all agents are laid end to end in LDS, and the wavefront then streams rows to their destinations, 16 bytes per lane, a
// whole wave store at a time whatever agent a piece belongs to (its agent's destination comes from a small LDS table).
// ------------------------------------------------------------------------------------------------------------------------------------
// list stores: FRZ_RS_STORE = 0 plain, 1 write-through (sc1), 2 non-temporal (nt), 3 sc0 sc1 — experiment switch, see DESIGN.md
#ifndef FRZ_RS_STORE
#define FRZ_RS_STORE 0
#endif
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void list_store(int4* at, const int4& value) {
    const v4i v = {value.x, value.y, value.z, value.w};
#if FRZ_RS_STORE == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(at), "v"(v) : "memory");
#elif FRZ_RS_STORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(at), "v"(v) : "memory");
#elif FRZ_RS_STORE == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(at), "v"(v) : "memory");
#else
    *at = value;
#endif
}
__device__ __forceinline__ void list_store(int64_t* at, int64_t value) {
    const v2i v = {(int)(uint32_t)value, (int)(value >> 32)};
#if FRZ_RS_STORE == 1
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(at), "v"(v) : "memory");
#elif FRZ_RS_STORE == 2
    asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(at), "v"(v) : "memory");
#elif FRZ_RS_STORE == 3
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(at), "v"(v) : "memory");
#else
    *at = value;
#endif
}
__device__ __forceinline__ void list_store(int32_t* at, int32_t value) {
#if FRZ_RS_STORE == 1
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(at), "v"(value) : "memory");
#elif FRZ_RS_STORE == 2
    asm volatile("global_store_dword %0, %1, off nt" ::"v"(at), "v"(value) : "memory");
#elif FRZ_RS_STORE == 3
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(at), "v"(value) : "memory");
#else
    *at = value;
#endif
}

// diagnostic builds only (timing experiments: -DFRZ_RS_SKIP=<bits> leaves store groups out): 1 task rows, 2 agent rows, 4 maps / states
#ifdef FRZ_RS_SKIP
#define FRZ_RS_SKIPPED(bit) ((FRZ_RS_SKIP & (bit)) != 0)
#else
#define FRZ_RS_SKIPPED(bit) false
#endif

template <int AMAX, int SPL>
__global__ void __launch_bounds__(kBlock) rs_emit_kernel(char* __restrict__ arena, const RsDev d) {
    __shared__ int4 s_rows[kEnvsPerBlock][SPL * 64][2];
    __shared__ uint16_t s_pick[kEnvsPerBlock][AMAX * SPL * 64];  // flat place -> slot | state << 8 | agent << 12
    __shared__ int64_t s_dest[kEnvsPerBlock][AMAX];             // agent -> (first row of its segment in the [A][cap] outputs) - (its first flat place)
    const int b = frz::env_of_wave<kEnvsPerBlock>(d.B);
    if (b < 0) return;  // no workgroup barrier in this kernel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t B = d.B;
    const int A = d.A, P = d.P;
    const int32_t* rows = reinterpret_cast<const int32_t*>(arena + d.off_rows4);
    const int32_t* pas = reinterpret_cast<const int32_t*>(arena + d.off_passengers) + (int64_t)b * PCOLS * P;
    const int count = rows[(int64_t)d.r_count * B + b];
    const int64_t cap = B * (int64_t)P;
    const int64_t task_base = reinterpret_cast<const int64_t*>(arena + d.off_task_offsets)[b];
    const int agent = lane < A ? lane : A - 1;
    const int64_t agent_first = reinterpret_cast<const int64_t*>(arena + d.off_agent_offsets)[(int64_t)agent * (B + 1) + b];  // agent-lane
    int st[SPL], drv[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
        const int slot = lane + 64 * k;
        const char* r = reinterpret_cast<const char*>(pas) + (uint64_t)((uint32_t)(slot < count ? slot : (count > 0 ? count - 1 : 0)) * 40u);
        const int4u hot = *reinterpret_cast<const int4u*>(r), cold = *reinterpret_cast<const int4u*>(r + 24);
        const bool live = slot < count;
        const int st_s = hot.z, drv_s = hot.w;
        s_rows[wave][slot][0] = make_int4(hot.x, hot.y, cold.x, cold.y);
        s_rows[wave][slot][1] = make_int4(st_s == 1 ? drv_s : kNone, st_s == 2 ? drv_s : kNone, cold.z, cold.w);
        st[k] = live ? st_s : -1;  // lanes past the count hold no passenger
        drv[k] = live ? drv_s : -2;
    }
    // flat places: agent 0's visible slots in table order, then agent 1's, ...
    uint32_t first_place = 0;  // agent-lane: where the agent's places start
    int total = 0;
    for_each_index(std::make_integer_sequence<int, AMAX>{}, [&](auto ic) {
        constexpr int a = decltype(ic)::value;
        first_place = write_lane_c<a>(first_place, (uint32_t)total);
#pragma unroll
        for (int k = 0; k < SPL; ++k) {
            const bool mine = st[k] == 0 || drv[k] == a;
            const uint64_t m = __ballot(mine);
            if (mine) s_pick[wave][total + lane_rank(m)] = (uint16_t)((lane + 64 * k) | (st[k] << 8) | (a << 12));
            total += (int)__popcll(m);
        }
    });
    if (lane < A) s_dest[wave][lane] = (int64_t)lane * cap + agent_first - (int64_t)first_place;
    wave_lds_sync();
    // all passengers, in table order (task_store, rideshare.py:415-425)
    {
        int4* const out = reinterpret_cast<int4*>(arena + d.off_task_values) + task_base * 2;
        const int4* const src = &s_rows[wave][0][0];
        for (int p = lane; p < (FRZ_RS_SKIPPED(1) ? 0 : 2 * count); p += 64) list_store(out + p, src[p]);
    }
    // agents >= A of a wider instantiation saw the unaccepted passengers too: their places lie behind those of the real agents and are
    // not written (the flat sequence is cut at the last real agent's end)
    const int real_total = A < AMAX ? (int)read_lane((int)first_place, A) : total;
    // the agents' task rows
    {
        int4* const out = reinterpret_cast<int4*>(arena + d.off_agent_task_values);
        for (int p = lane; p < (FRZ_RS_SKIPPED(2) ? 0 : 2 * real_total); p += 64) {
            const int r = p >> 1, e = s_pick[wave][r];
            list_store(out + (s_dest[wave][e >> 12] + r) * 2 + (p & 1), s_rows[wave][e & 0xFF][p & 1]);
        }
    }
    // their positions in the env's passenger list (action = observation mapping) and their states (the action id each OneOf member carries)
    {
        int64_t* const map_out = reinterpret_cast<int64_t*>(arena + d.off_agent_map_values);
        int32_t* const state_out = reinterpret_cast<int32_t*>(arena + d.off_agent_task_states);
        for (int r = lane; r < (FRZ_RS_SKIPPED(4) ? 0 : real_total); r += 64) {
            const int e = s_pick[wave][r];
            const int64_t at = s_dest[wave][e >> 12] + r;
            list_store(map_out + at, (int64_t)(e & 0xFF));
            list_store(state_out + at, (int32_t)((e >> 8) & 0xF));
        }
    }
}

// uniform member of OneOf([Discrete(1, start=state_t) for visible task t] + [noop]) (spaces/actions.py:10-50)
__global__ void __launch_bounds__(kBlock) rs_policy_kernel(const char* arena, uint32_t seed_lo, uint32_t seed_hi, uint32_t step_lo,
                                                             uint32_t step_hi, int32_t* actions) {
    const RsDev& d = *reinterpret_cast<const RsDev*>(arena);
    const int64_t B = d.B;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)d.A * B) return;
    const int a = (int)(i / B);
    const int64_t b = i % B;
    const int32_t* rows = reinterpret_cast<const int32_t*>(arena + d.off_rows4);
    const int64_t* agent_offsets = reinterpret_cast<const int64_t*>(arena + d.off_agent_offsets);
    const int32_t* agent_states = reinterpret_cast<const int32_t*>(arena + d.off_agent_task_states);
    const int n = rows[d.r_atc * B + i];
    const int64_t env_global = b + d.first_env_index;  // sharding-invariant stream: (agent, global env index, step)
    const frz::Philox4 w = frz::philox4x32_10((uint32_t)a, (uint32_t)env_global, step_lo, step_hi, seed_lo, seed_hi);
    const int j = (int)(((uint64_t)w.w[0] * (uint64_t)(n + 1)) >> 32);
    const int64_t cap = B * (int64_t)d.P;
    const int value = j < n ? agent_states[a * cap + agent_offsets[a * (B + 1) + b] + j] : -1;
    reinterpret_cast<int2*>(actions)[i] = make_int2(j, value);
}

}  // namespace

// ================================================================================================================
// host side of the C-ABI
// ================================================================================================================
struct frz_rideshare_env {
    frz_rideshare_cfg cfg;
    RsDev dev;
    std::vector<int32_t> schedule;        // time-sorted copy
    std::vector<int32_t> schedule_index;  // [max_time + 2]
    char* arena = nullptr;
    bool was_reset = false;
    bool ticketed = false;  // rs_offsets_kernel: more chunks than CUs, chunks are handed out in arrival order (frz_scan.h)
    int variant = 0;
    std::vector<hipEvent_t> timing_events;  // pool of frz_rideshare_timed_rollout
};

namespace {

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct Timing {
    hipEvent_t start = nullptr, stop = nullptr;  // begin of the step's first dispatch / end of its last one
};

template <typename K, typename... Args>
void launch_kernel(K kernel, dim3 grid, hipStream_t stream, hipEvent_t start, hipEvent_t stop, Args... args) {
    if (start || stop)
        hipExtLaunchKernelGGL(kernel, grid, dim3(kBlock), 0, stream, start, stop, 0, args...);
    else
        hipLaunchKernelGGL(kernel, grid, dim3(kBlock), 0, stream, args...);
}

template <int AMAX, int SPL>
void launch_variant(frz_rideshare_env* env, const int32_t* actions, int mode, const RsPolicy& policy, hipStream_t stream, const Timing& timing) {
    const RsDev& d = env->dev;
    const dim3 waves((unsigned)((d.B + kEnvsPerBlock - 1) / kEnvsPerBlock)), lanes((unsigned)d.nchunks);
    if (mode == kRebuild)
        launch_kernel(rs_env_kernel<AMAX, SPL, kRebuild>, waves, stream, timing.start, nullptr, env->arena, d, actions, policy);
    else
        launch_kernel(rs_env_kernel<AMAX, SPL, kStep>, waves, stream, timing.start, nullptr, env->arena, d, actions, policy);
    launch_kernel(rs_offsets_kernel<AMAX>, lanes, stream, nullptr, nullptr, env->arena, d, env->ticketed ? 1u : 0u);
    launch_kernel(rs_emit_kernel<AMAX, SPL>, waves, stream, nullptr, timing.stop, env->arena, d);
}

int launch(frz_rideshare_env* env, const int32_t* actions, int mode, const RsPolicy& policy, hipStream_t stream, const Timing& timing = Timing()) {
    const bool wide = env->dev.P > 64;
    switch (env->variant * 2 + (wide ? 1 : 0)) {
        case 0: launch_variant<4, 1>(env, actions, mode, policy, stream, timing); break;
        case 1: launch_variant<4, 2>(env, actions, mode, policy, stream, timing); break;
        case 2: launch_variant<8, 1>(env, actions, mode, policy, stream, timing); break;
        case 3: launch_variant<8, 2>(env, actions, mode, policy, stream, timing); break;
        case 4: launch_variant<16, 1>(env, actions, mode, policy, stream, timing); break;
        default: launch_variant<16, 2>(env, actions, mode, policy, stream, timing); break;
    }
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

template <typename T>
T* at(char* arena, int64_t off) {
    return reinterpret_cast<T*>(arena + off);
}

}  // namespace

extern "C" {

int frz_rideshare_create(const frz_rideshare_cfg* cfg, const int32_t* schedule, frz_rideshare_env** out) {
    if (!cfg || !out || (cfg->schedule_rows > 0 && !schedule)) return FRZ_E_INVALID;
    const int A = cfg->num_agents, P = cfg->max_passengers;
    if (cfg->parallel_envs <= 0 || A <= 0 || A > FRZ_MAX_AGENTS || P <= 0 || P > FRZ_MAX_PASSENGERS || cfg->schedule_rows < 0)
        return FRZ_E_INVALID;
    // arrays the kernels address with 32-bit byte offsets from a wave-uniform base stay below 4 GiB
    if ((int64_t)(4 * A + 8) * cfg->parallel_envs >= (int64_t)1 << 28 || (int64_t)A * (A - 1) * cfg->parallel_envs * 16 >= (int64_t)1 << 32 ||
        (int64_t)cfg->schedule_rows * 28 >= (int64_t)1 << 32)
        return FRZ_E_INVALID;
    // coordinates stay inside +-16383: squared distances fit 32 bits, an agent's move (a difference of two positions with fast
    // travel) travels between lanes as two 16-bit halves of one word
    auto small = [](int32_t v) { return v > -16384 && v < 16384; };
    for (int a = 0; a < A; ++a)
        if (!small(cfg->start_y[a]) || !small(cfg->start_x[a])) return FRZ_E_INVALID;
    for (int64_t r = 0; r < cfg->schedule_rows; ++r)
        for (int c = 2; c < 6; ++c)
            if (!small(schedule[r * 7 + c])) return FRZ_E_INVALID;
    frz_rideshare_env* env = new (std::nothrow) frz_rideshare_env();
    if (!env) return FRZ_E_INVALID;
    env->cfg = *cfg;
    env->variant = A <= 4 ? 0 : (A <= 8 ? 1 : 2);
    RsDev& p = env->dev;
    std::memset(&p, 0, sizeof(p));
    const int64_t B = cfg->parallel_envs;
    p.B = cfg->parallel_envs, p.A = A, p.P = P;
    p.nchunks = (cfg->parallel_envs + kBlock - 1) / kBlock;
    p.max_steps = cfg->max_steps;
    p.pool_limit = cfg->pool_limit;
    p.long_wait_time = cfg->long_wait_time;
    p.first_env_index = cfg->first_env_index;
    p.schedule_rows = cfg->schedule_rows;
    std::memcpy(p.wait_limit, cfg->wait_limit, sizeof(p.wait_limit));
    auto flag = [&](int on, uint32_t bit) { p.flags |= on ? bit : 0u; };
    flag(cfg->use_fast_travel, kFast);
    flag(cfg->use_diagonal_travel, kDiagonal);
    flag(cfg->use_variable_move_cost, kVariableMove);
    flag(cfg->use_waiting_costs, kWaiting);
    flag(cfg->track_cumulative_rewards, kTrackCumulative);
    flag(cfg->max_steps >= 0, kTruncate);
    p.move_cost = cfg->move_cost, p.drop_cost = cfg->drop_cost, p.noop_cost = cfg->noop_cost, p.accept_cost = cfg->accept_cost;
    p.pool_limit_cost = cfg->pool_limit_cost, p.general_wait_cost = cfg->general_wait_cost, p.long_wait_cost = cfg->long_wait_cost;
    p.inv_others = A > 1 ? (uint32_t)((65536 + A - 2) / (A - 1)) : 0u;
    for (int q = 0; q < A * (A - 1); ++q)
        if ((int)(((uint32_t)q * p.inv_others) >> 16) != q / (A - 1)) {  // cannot happen for A <= FRZ_MAX_AGENTS; checked, not assumed
            delete env;
            return FRZ_E_INVALID;
        }
    std::memcpy(p.start_y, cfg->start_y, sizeof(p.start_y));
    std::memcpy(p.start_x, cfg->start_x, sizeof(p.start_x));

    // schedule, stable-sorted by timestep (rows of one timestep keep their schedule order, transitions/passenger_entry.py:57);
    // rows with a negative timestep can never match a step and are dropped
    std::vector<int> order;
    for (int r = 0; r < cfg->schedule_rows; ++r)
        if (schedule[r * 7] >= 0) order.push_back(r);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return schedule[x * 7] < schedule[y * 7]; });
    const int S = (int)order.size();
    const int max_time = S > 0 ? schedule[order[S - 1] * 7] : -1;
    p.max_time = max_time;
    p.schedule_rows = S;
    env->schedule.assign((size_t)(S > 0 ? S : 1) * 7, 0);
    for (int r = 0; r < S; ++r) std::memcpy(&env->schedule[(size_t)r * 7], &schedule[(size_t)order[r] * 7], 7 * sizeof(int32_t));
    env->schedule_index.assign((size_t)max_time + 3, S);  // index[t] = first sorted row with timestep >= t
    for (int t = max_time + 1, row = S; t >= 0; --t) {
        while (row > 0 && env->schedule[(size_t)(row - 1) * 7] >= t) --row;
        env->schedule_index[t] = row;
    }

    int r = 0;
    p.r_count = r++;
    p.r_moves = r++;
    p.r_rewards = r, r += A;
    p.r_cum = r, r += A;
    p.r_atc = r, r += A;
    p.n_rows4 = r;
    p.u_term = 0, p.u_trunc = A, p.u_frozen = 2 * A, p.n_rows1 = 2 * A + 1;
    const int64_t cap = B * P;
    const int nch_total = A + 3;
    int64_t off = kDevBlockBytes;
    auto take = [&](int64_t bytes) {
        const int64_t here = off;
        off = align_up(off + (bytes > 0 ? bytes : 1), 256);
        return here;
    };
    p.off_rows4 = take((int64_t)p.n_rows4 * B * 4);
    p.off_rows1 = take((int64_t)p.n_rows1 * B);
    p.off_agents = take((int64_t)A * B * 8);
    p.off_passengers = take((int64_t)PCOLS * P * B * 4);
    p.off_etc = take(B * 8);
    p.off_obs_self = take((int64_t)A * B * 16);
    p.off_obs_others = take((int64_t)A * B * (A - 1) * 16);
    p.off_task_offsets = take((B + 1) * 8);
    p.off_agent_offsets = take((int64_t)A * (B + 1) * 8);
    p.off_task_values = take(cap * 32);
    p.off_agent_task_values = take((int64_t)A * cap * 32);
    p.off_agent_map_values = take((int64_t)A * cap * 8);
    p.off_agent_task_states = take((int64_t)A * cap * 4);
    p.off_schedule = take((int64_t)env->schedule.size() * 4);
    p.off_schedule_index = take((int64_t)env->schedule_index.size() * 4);
    p.off_actions = take((int64_t)A * B * 8);
    p.off_error = take(256);
    p.off_epoch = take(256);
    p.off_totals = take(2 * frz::kTotalsStride * 4);
    p.off_agg = take((int64_t)p.nchunks * nch_total * 8);
    p.off_prefix = take((int64_t)p.nchunks * nch_total * 8);
    p.total_bytes = off;

    int device = 0, cus = 256;
    if (hipGetDevice(&device) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    env->ticketed = p.nchunks > cus;  // one workgroup per chunk; one 256-thread workgroup per CU is always resident
    frz::handle_register(env, 3, A, cfg->parallel_envs, P);
    *out = env;
    return FRZ_OK;
}

void frz_rideshare_destroy(frz_rideshare_env* env) {
    if (!env) return;
    frz::handle_unregister(env);
    for (hipEvent_t e : env->timing_events) (void)hipEventDestroy(e);
    delete env;
}

int64_t frz_rideshare_arena_bytes(const frz_rideshare_env* env) { return env ? env->dev.total_bytes : FRZ_E_INVALID; }

int frz_rideshare_bind(frz_rideshare_env* env, void* arena, void* stream) {
    if (!env || !arena || reinterpret_cast<uintptr_t>(arena) % 256 != 0) return FRZ_E_INVALID;
    env->arena = static_cast<char*>(arena);
    env->was_reset = false;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(arena, &env->dev, sizeof(RsDev), hipMemcpyHostToDevice, s) != hipSuccess) return FRZ_E_LAUNCH;
    if (hipMemcpyAsync(env->arena + env->dev.off_schedule, env->schedule.data(), env->schedule.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess)
        return FRZ_E_LAUNCH;
    if (hipMemcpyAsync(env->arena + env->dev.off_schedule_index, env->schedule_index.data(), env->schedule_index.size() * 4,
                       hipMemcpyHostToDevice, s) != hipSuccess)
        return FRZ_E_LAUNCH;
    return hipStreamSynchronize(s) == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

int frz_rideshare_get_bufs(const frz_rideshare_env* env, frz_rideshare_bufs* out) {
    if (!env || !out) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    const RsDev& p = env->dev;
    char* a = env->arena;
    const int64_t B = p.B;
    auto row4 = [&](int r) { return a + p.off_rows4 + (int64_t)r * B * 4; };
    auto row1 = [&](int r) { return a + p.off_rows1 + (int64_t)r * B; };
    out->agents = at<int32_t>(a, p.off_agents);
    out->passenger_count = reinterpret_cast<int32_t*>(row4(p.r_count));
    out->num_moves = reinterpret_cast<int32_t*>(row4(p.r_moves));
    out->rewards = reinterpret_cast<float*>(row4(p.r_rewards));
    out->cumulative_rewards = reinterpret_cast<float*>(row4(p.r_cum));
    out->agent_task_count = reinterpret_cast<int32_t*>(row4(p.r_atc));
    out->terminations = reinterpret_cast<uint8_t*>(row1(p.u_term));
    out->truncations = reinterpret_cast<uint8_t*>(row1(p.u_trunc));
    out->frozen_scaled = reinterpret_cast<uint8_t*>(row1(p.u_frozen));
    out->passengers = at<int32_t>(a, p.off_passengers);
    out->env_task_count = at<int64_t>(a, p.off_etc);
    out->obs_self = at<int32_t>(a, p.off_obs_self);
    out->obs_others = at<int32_t>(a, p.off_obs_others);
    out->task_values = at<int32_t>(a, p.off_task_values);
    out->task_offsets = at<int64_t>(a, p.off_task_offsets);
    out->agent_task_values = at<int32_t>(a, p.off_agent_task_values);
    out->agent_map_values = at<int64_t>(a, p.off_agent_map_values);
    out->agent_offsets = at<int64_t>(a, p.off_agent_offsets);
    out->agent_task_states = at<int32_t>(a, p.off_agent_task_states);
    out->schedule = at<int32_t>(a, p.off_schedule);
    out->actions = at<int32_t>(a, p.off_actions);
    out->error_flags = at<uint32_t>(a, p.off_error);
    return FRZ_OK;
}

int frz_rideshare_rebuild(frz_rideshare_env* env, void* stream) {
    if (!env) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    env->was_reset = true;
    return launch(env, nullptr, kRebuild, RsPolicy{}, static_cast<hipStream_t>(stream));
}

int frz_rideshare_reset(frz_rideshare_env* env, void* stream) {
    if (!env) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    const unsigned blocks = (unsigned)((env->cfg.parallel_envs + kEnvsPerBlock - 1) / kEnvsPerBlock);
    hipLaunchKernelGGL(rs_fill_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), env->arena, env->dev);
    if (hipGetLastError() != hipSuccess) return FRZ_E_LAUNCH;
    return frz_rideshare_rebuild(env, stream);
}

int frz_rideshare_step(frz_rideshare_env* env, const int32_t* actions, void* stream) {
    if (!env || !actions) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;
    return launch(env, actions, kStep, RsPolicy{}, static_cast<hipStream_t>(stream));
}

static RsPolicy make_policy(uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out) {
    return RsPolicy{1u, (uint32_t)policy_seed, (uint32_t)(policy_seed >> 32), (uint32_t)policy_step, (uint32_t)(policy_step >> 32), actions_out};
}

int frz_rideshare_step_random_policy(frz_rideshare_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out, void* stream) {
    if (!env || !actions_out) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;
    return launch(env, actions_out, kStep, make_policy(policy_seed, policy_step, actions_out), static_cast<hipStream_t>(stream));
}

int frz_rideshare_list_block(const frz_rideshare_env* env, void** block, int64_t* bytes) {
    if (!env || !block || !bytes) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    *block = env->arena + env->dev.off_task_offsets;  // task / agent offsets, task rows, per-agent task rows, index maps, task states
    *bytes = env->dev.off_schedule - env->dev.off_task_offsets;
    return FRZ_OK;
}

// utils/conversions.py:59-99 over n steps for this domain: one launch sequence per step (the env launch's fields and crew, the offsets,
// the lists), the records copied out between the steps.  The domain draws nothing (rideshare.py:248-365): rng_mode and the randomness
// tapes are not looked at; it has no partial reset (rideshare.py:246 raises) and no metrics entry: those options are refused.
int frz_rideshare_rollout(frz_rideshare_env* env, const frz_rollout_spec* spec, void* stream) {
    if (!env || !spec || spec->n_steps < 0) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;
    const RsDev& p = env->dev;
    const bool policy = spec->action_tape == nullptr, reset_first = (spec->flags & FRZ_ROLLOUT_RESET_FIRST) != 0;
    if (policy && !spec->actions_out) return FRZ_E_INVALID;
    if ((spec->flags & FRZ_ROLLOUT_AUTO_RESET) || spec->metrics || spec->seed_increment != 0) return FRZ_E_INVALID;
    if (spec->obs_tape || spec->state_tape || (spec->flags & FRZ_ROLLOUT_OBS_COMPACT)) return FRZ_E_INVALID;  // (observation / state tapes: wildfire, cybersecurity)
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (reset_first) {
        const int rc = frz_rideshare_reset(env, stream);
        if (rc != FRZ_OK) return rc;
    }
    const int64_t B = p.B, A = p.A, AB2 = A * B * 2;
    const int64_t block_bytes = p.off_schedule - p.off_task_offsets;
    for (int32_t t = 0; t < spec->n_steps; ++t) {
        int rc;
        if (policy)
            rc = frz_rideshare_step_random_policy(env, spec->policy_seed, spec->first_step + (uint64_t)t,
                                                  spec->actions_out + (spec->record_actions ? (int64_t)t * AB2 : 0), stream);
        else
            rc = frz_rideshare_step(env, spec->action_tape + (int64_t)t * AB2, stream);
        if (rc != FRZ_OK) return rc;
        bool ok = true;
        if (spec->reward_tape)
            ok = ok && hipMemcpyAsync(spec->reward_tape + (int64_t)t * A * B, env->arena + p.off_rows4 + (int64_t)p.r_rewards * B * 4, (size_t)(A * B * 4),
                                      hipMemcpyDeviceToDevice, s) == hipSuccess;
        if (spec->done_tape) {
            ok = ok && hipMemcpyAsync(spec->done_tape + ((int64_t)t * 2 + 0) * B, env->arena + p.off_rows1 + (int64_t)p.u_term * B, (size_t)B,
                                      hipMemcpyDeviceToDevice, s) == hipSuccess;
            ok = ok && hipMemcpyAsync(spec->done_tape + ((int64_t)t * 2 + 1) * B, env->arena + p.off_rows1 + (int64_t)p.u_trunc * B, (size_t)B,
                                      hipMemcpyDeviceToDevice, s) == hipSuccess;
        }
        if (spec->list_record && t < spec->n_steps - 1)
            ok = ok && hipMemcpyAsync(static_cast<char*>(spec->list_record) + (int64_t)t * block_bytes, env->arena + p.off_task_offsets, (size_t)block_bytes,
                                      hipMemcpyDeviceToDevice, s) == hipSuccess;
        if (!ok) return FRZ_E_LAUNCH;
    }
    return FRZ_OK;
}

int frz_rideshare_timed_rollout(frz_rideshare_env* env, uint64_t policy_seed, uint64_t first_step, int32_t n_steps, int32_t* actions_out,
                                void* stream, float* step_ms) {
    if (!env || !actions_out || !step_ms || n_steps <= 0) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;
    while ((int)env->timing_events.size() < 2 * n_steps) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return FRZ_E_LAUNCH;
        env->timing_events.push_back(e);
    }
    for (int i = 0; i < n_steps; ++i) {  // back to back: no host synchronisation between the steps
        const Timing timing{env->timing_events[2 * i], env->timing_events[2 * i + 1]};
        const int rc = launch(env, actions_out, kStep, make_policy(policy_seed, first_step + (uint64_t)i, actions_out),
                              static_cast<hipStream_t>(stream), timing);
        if (rc != FRZ_OK) return rc;
    }
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return FRZ_E_LAUNCH;
    for (int i = 0; i < n_steps; ++i)
        if (hipEventElapsedTime(&step_ms[i], env->timing_events[2 * i], env->timing_events[2 * i + 1]) != hipSuccess) return FRZ_E_LAUNCH;
    return FRZ_OK;
}

int frz_rideshare_random_policy(frz_rideshare_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out, void* stream) {
    if (!env || !actions_out) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    const int64_t n = (int64_t)env->dev.A * env->dev.B;
    hipLaunchKernelGGL(rs_policy_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       env->arena, (uint32_t)policy_seed, (uint32_t)(policy_seed >> 32), (uint32_t)policy_step,
                       (uint32_t)(policy_step >> 32), actions_out);
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

}  // extern "C"
