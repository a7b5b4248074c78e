// rideshare.hip — fused rideshare environment step for gfx950 (MI355X), one environment per lane.
//
// One launch = one ParallelEnv.step() of the reference (rideshare.py:248-467) for the whole batch:
//   action decode through the open action mapping -> movement -> passenger state (accept-conflict resolution, picks)
//   -> passenger exit (drops, fares, ordered compaction) -> passenger entry (schedule) -> rewards -> truncation
//   -> update_actions / update_observations: per-agent visible-task lists compacted with the launch-wide single-pass
//      prefix scan of frz_scan.h.
//
// The reference keeps one global passenger table sorted by env and re-sorts / boolean-compacts it every step; here each
// env owns max_passengers slots in table order, struct-of-arrays [column][slot][B] in the device arena, so that the 64
// lanes of a wavefront stream slot s of 64 consecutive envs as one 256-byte segment per column.  A lane walks its env's
// slots three times per step: decode (2 columns), transform + ordered in-place compaction (10 columns), emission of the
// observation rows (8 columns, re-read from L2).  Deterministic integer/byte work, HBM-bound: no MFMA, no randomness.
#include "frz_scan.h"

#include "../../include/frz.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

using frz::kBlock;

enum Mode { kStep = 0, kRebuild = 1 };
enum Flag : uint32_t { kFast = 1u << 0, kDiagonal = 1u << 1, kVariableMove = 1u << 2, kWaiting = 1u << 3, kTrackCumulative = 1u << 4, kTruncate = 1u << 5 };
constexpr int kBatch = 4;  // slots whose loads are in flight together in the three walks over an env's passenger slots
enum Col { PY = 0, PX, PYD, PXD, PFARE, PSTATE, PDRIVER, PENTERED, PACCEPTED, PPICKED, PCOLS };
constexpr int kNone = -100;

struct RsDev {
    int32_t B, A, P, nchunks, max_steps, pool_limit, long_wait_time, schedule_rows, max_time, first_env_index;
    int32_t wait_limit[3];
    uint32_t flags;
    float move_cost, drop_cost, noop_cost, accept_cost, pool_limit_cost, general_wait_cost, long_wait_cost;
    int32_t start_y[FRZ_MAX_AGENTS], start_x[FRZ_MAX_AGENTS];
    int32_t r_agents, r_count, r_moves, r_rewards, r_cum, r_atc, n_rows4;
    int32_t u_term, u_trunc, u_frozen, n_rows1;
    int64_t off_rows4, off_rows1, off_passengers, off_etc, off_obs_self, off_obs_others, off_task_values, off_task_offsets,
        off_agent_task_values, off_agent_map_values, off_agent_offsets, off_agent_task_states, off_schedule, off_schedule_index,
        off_actions, off_error, off_epoch, off_totals, off_agg, off_prefix, total_bytes;
};
constexpr int64_t kDevBlockBytes = 4096;
static_assert(sizeof(RsDev) <= kDevBlockBytes, "configuration block too large");

template <typename T>
__device__ __forceinline__ T& at32(T* base, uint32_t index) {
    return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (uint64_t)(index * (uint32_t)sizeof(T)));
}

// schedule rows of one timestep are contiguous in the time-sorted device copy: [index[t], index[t + 1])
struct Schedule {
    const int32_t* rows;   // [S][7] stable-sorted by timestep
    const int32_t* index;  // [max_time + 2]
    int max_time;
};

// transitions/passenger_entry.py:24-72 for one env: append the rows of timestep t (this env or wildcard) in schedule order
template <typename OnEntry>
__device__ __forceinline__ void passenger_entry(const RsDev& d, const Schedule& sch, int32_t* pas, uint32_t Bu, uint32_t bl, int64_t b,
                                                int t, int& count, uint32_t& err, bool active, OnEntry on_entry) {
    if (t < 0 || t > sch.max_time) return;
    const int first = sch.index[t], last = sch.index[t + 1];
    for (int r = first; r < last; ++r) {
        const int32_t* row = sch.rows + r * 7;
        const int env = row[1];
        if (!(env == -1 || env == (int)b)) continue;
        if (count >= d.P) {
            if (active) err |= FRZ_ERR_OVERFLOW;
            continue;
        }
        const uint32_t s = (uint32_t)count;
        if (active) {
            at32(pas, ((uint32_t)PY * d.P + s) * Bu + bl) = row[2];
            at32(pas, ((uint32_t)PX * d.P + s) * Bu + bl) = row[3];
            at32(pas, ((uint32_t)PYD * d.P + s) * Bu + bl) = row[4];
            at32(pas, ((uint32_t)PXD * d.P + s) * Bu + bl) = row[5];
            at32(pas, ((uint32_t)PFARE * d.P + s) * Bu + bl) = row[6];
            at32(pas, ((uint32_t)PSTATE * d.P + s) * Bu + bl) = 0;
            at32(pas, ((uint32_t)PDRIVER * d.P + s) * Bu + bl) = -1;
            at32(pas, ((uint32_t)PENTERED * d.P + s) * Bu + bl) = t;
            at32(pas, ((uint32_t)PACCEPTED * d.P + s) * Bu + bl) = -1;
            at32(pas, ((uint32_t)PPICKED * d.P + s) * Bu + bl) = -1;
        }
        on_entry(t);
        ++count;
    }
}

// rideshare.py:185-222 + utils/env.py:137-160: agents at their start positions, bookkeeping zeroed, step-0 passengers enter
__global__ void __launch_bounds__(kBlock) rs_fill_kernel(char* arena) {
    const RsDev& d = *reinterpret_cast<const RsDev*>(arena);
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x, B = d.B;
    if (b >= B) return;
    int32_t* rows = reinterpret_cast<int32_t*>(arena + d.off_rows4);
    float* rowsf = reinterpret_cast<float*>(arena + d.off_rows4);
    uint8_t* rows1 = reinterpret_cast<uint8_t*>(arena + d.off_rows1);
    int32_t* pas = reinterpret_cast<int32_t*>(arena + d.off_passengers);
    for (int a = 0; a < d.A; ++a) {
        rows[(d.r_agents + 2 * a) * B + b] = d.start_y[a];
        rows[(d.r_agents + 2 * a + 1) * B + b] = d.start_x[a];
        rowsf[(d.r_rewards + a) * B + b] = 0.0f;
        rowsf[(d.r_cum + a) * B + b] = 0.0f;
        rows1[(d.u_term + a) * B + b] = 0;
        rows1[(d.u_trunc + a) * B + b] = 0;
    }
    rows[d.r_moves * B + b] = 0;
    rows1[d.u_frozen * B + b] = 0;
    const Schedule sch{reinterpret_cast<const int32_t*>(arena + d.off_schedule), reinterpret_cast<const int32_t*>(arena + d.off_schedule_index),
                       d.max_time};
    int count = 0;
    uint32_t err = 0;
    passenger_entry(d, sch, pas, (uint32_t)B, (uint32_t)b, b, 0, count, err, true, [](int) {});
    rows[d.r_count * B + b] = count;
    if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);
}

// diagnostic builds only (tools/exp_skip.sh): -DFRZ_RS_SKIP_MASK=<bits> leaves store groups of the emission pass out
#ifdef FRZ_RS_SKIP_MASK
#define FRZ_RS_SKIP(bit) ((FRZ_RS_SKIP_MASK & (bit)) != 0)
#else
#define FRZ_RS_SKIP(bit) false
#endif

template <int AMAX, int MODE>
__global__ void __launch_bounds__(kBlock) rs_step_kernel(char* __restrict__ arena, const RsDev d, const int32_t* __restrict__ actions,
                                                          uint32_t ticketed) {
    // the configuration block arrives BY VALUE (424 bytes of kernel arguments): read through a pointer into the arena, every field
    // had to be re-read after each store that might alias it — a scalar-memory round trip per use, hundreds per env slot walked
    __shared__ frz::ScanShared<AMAX + 1> s_scan;
    __shared__ int s_ticket;
    // What this step does to each slot of each env, looked up by slot in the ordered pass instead of comparing every agent's target with
    // every slot: bits 0-4 the (last) agent whose accept won the slot, plus one; bit 5 picked up; bit 6 dropped off.  And the agents' moves,
    // looked up by a riding passenger's driver.  [slot][lane] / [agent][lane]: a lane only ever reads what it wrote itself.
    __shared__ uint8_t s_effect[MODE == kStep ? FRZ_MAX_PASSENGERS : 1][kBlock];
    __shared__ short2 s_move[MODE == kStep ? AMAX : 1][kBlock];  // (dy, dx): |move| < grid size < 2^15 (checked by frz_rideshare_create)

    const int tid = threadIdx.x;
    const int64_t B = d.B;
    const uint32_t Bu = (uint32_t)d.B, P = (uint32_t)d.P;
    const int A = d.A;
    const uint32_t flags = d.flags;
    frz::ScanWorkspace ws{reinterpret_cast<uint32_t*>(arena + d.off_epoch), reinterpret_cast<uint32_t*>(arena + d.off_totals),
                          reinterpret_cast<uint64_t*>(arena + d.off_agg), reinterpret_cast<uint64_t*>(arena + d.off_prefix)};
    const frz::ScanLaunch launch = frz::scan_begin(ws);
    int32_t* const rows = reinterpret_cast<int32_t*>(arena + d.off_rows4);
    float* const rowsf = reinterpret_cast<float*>(arena + d.off_rows4);
    uint8_t* const rows1 = reinterpret_cast<uint8_t*>(arena + d.off_rows1);
    int32_t* const pas = reinterpret_cast<int32_t*>(arena + d.off_passengers);
    const Schedule sch{reinterpret_cast<const int32_t*>(arena + d.off_schedule), reinterpret_cast<const int32_t*>(arena + d.off_schedule_index),
                       d.max_time};
    auto pcol = [&](int col, int slot, uint32_t env) -> int32_t& { return at32(pas, ((uint32_t)col * P + (uint32_t)slot) * Bu + env); };

    // utils/env.py:211-213 (terminations never set, rideshare.py:252): frozen once every env is truncated.
    // totals channels A + 1 / A + 2 = number of envs not terminated / not truncated after the previous launch
    bool frozen = false;
    if (MODE == kStep) frozen = launch.prev[A + 1] == 0u || launch.prev[A + 2] == 0u;

    {  // one chunk per workgroup (no chunk loop: see wildfire_roles.hip)
        const int chunk = frz::scan_take_chunk(ws, d.nchunks, ticketed != 0, &s_ticket);
        const int64_t b = (int64_t)chunk * kBlock + tid;
        const bool active = b < B;
        const uint32_t bl = (uint32_t)(active ? b : B - 1);

        if (frozen) {  // the parallel adapter sums the stale rewards once per agent call (utils/conversions.py:87-90)
            if (active && !at32(rows1, (uint32_t)d.u_frozen * Bu + bl)) {
                for (int a = 0; a < A; ++a) {
                    const float r = at32(rowsf, (uint32_t)(d.r_rewards + a) * Bu + bl);
                    float acc = 0.0f;
                    for (int j = 0; j < A; ++j) acc = acc + r;
                    at32(rowsf, (uint32_t)(d.r_rewards + a) * Bu + bl) = acc;
                }
                at32(rows1, (uint32_t)d.u_frozen * Bu + bl) = 1;
            }
            return;
        }

        int count = at32(rows, (uint32_t)d.r_count * Bu + bl);
        bool trunc = at32(rows1, (uint32_t)d.u_trunc * Bu + bl) != 0;
        int ay[AMAX], ax[AMAX];
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            ay[a] = a < A ? at32(rows, (uint32_t)(d.r_agents + 2 * a) * Bu + bl) : 0;
            ax[a] = a < A ? at32(rows, (uint32_t)(d.r_agents + 2 * a + 1) * Bu + bl) : 0;
        }
        uint32_t err = 0;
        int visible[AMAX], n_accepted[AMAX], n_riding[AMAX];  // per agent: visible tasks, own accepted / riding passengers
#pragma unroll
        for (int a = 0; a < AMAX; ++a) visible[a] = n_accepted[a] = n_riding[a] = 0;

        if (MODE == kStep) {
            const int nm = at32(rows, (uint32_t)d.r_moves * Bu + bl);
            // ---------------------------------------------------------------- (1) action decode (rideshare.py:256-300)
            int act_idx[AMAX], act_id[AMAX], target[AMAX], seen[AMAX];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                const int2 v = a < A ? reinterpret_cast<const int2*>(actions)[a * B + bl] : make_int2(0, -1);
                act_idx[a] = v.x;
                act_id[a] = v.y;
                target[a] = kNone;
                seen[a] = 0;
            }
            // the agent's action mapping lists, in table order, the passengers that are unaccepted or its own
            // (rideshare.py:378-392): walk the slots and pick the act_idx-th visible one
            // (slots are walked kBatch at a time with every load of a batch issued before its first use: a lane-per-env walk is a
            // chain of dependent memory round trips otherwise; slots past `count` are read from slot P - 1 and ignored)
            for (int s0 = 0; s0 < count; s0 += kBatch) {
                int st[kBatch], drv[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int s = min(s0 + u, (int)P - 1);
                    st[u] = pcol(PSTATE, s, bl);
                    drv[u] = pcol(PDRIVER, s, bl);
                }
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int s = s0 + u;
                    const bool live = s < count;
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) {
                        const bool vis = live && (st[u] == 0 || drv[u] == a);
                        target[a] = (vis && seen[a] == act_idx[a]) ? s : target[a];
                        seen[a] += vis ? 1 : 0;
                    }
                }
            }
            bool accept[AMAX], pick[AMAX], drop[AMAX], has_vec[AMAX];
            int gy[AMAX], gx[AMAX];  // goal of the task vector (passenger position, or destination for a drop)
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                const bool noop = act_id[a] == -1;
                const bool valid = a < A && !noop && act_idx[a] >= 0 && act_idx[a] < seen[a];
                if (a < A && !noop && !valid && active) err |= FRZ_ERR_BAD_ACTION_INDEX;  // the reference reads a garbage row
                accept[a] = valid && act_id[a] == 0;
                pick[a] = valid && act_id[a] == 1;
                drop[a] = valid && act_id[a] == 2;
                has_vec[a] = accept[a] || pick[a] || drop[a];
                gy[a] = gx[a] = kNone;
                if (has_vec[a]) {
                    gy[a] = pcol(drop[a] ? PYD : PY, target[a], bl);
                    gx[a] = pcol(drop[a] ? PXD : PX, target[a], bl);
                }
            }
            // ---------------------------------------------------------------- (2) movement (transitions/movement.py:56-116)
            int my[AMAX], mx[AMAX];
            float cost[AMAX];
            int64_t dist2[AMAX];  // squared pre-move distance to the goal: sqrt is monotonic, zero iff zero
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                my[a] = mx[a] = 0;
                cost[a] = 0.0f;
                dist2[a] = 0;
                if (has_vec[a]) {
                    const int dy = ay[a] - gy[a], dx = ax[a] - gx[a];
                    dist2[a] = (int64_t)dy * dy + (int64_t)dx * dx;
                    if (flags & kFast) {
                        my[a] = -dy;
                        mx[a] = -dx;
                    } else {  // first minimum over {stay, N, E, S, W(, NW, NE, SE, SW)}
                        int64_t best = dist2[a];
                        const int cy[9] = {0, -1, 0, 1, 0, -1, -1, 1, 1}, cx[9] = {0, 0, 1, 0, -1, -1, 1, 1, -1};
                        const int ndirs = (flags & kDiagonal) ? 9 : 5;
#pragma unroll
                        for (int k = 1; k < 9; ++k) {
                            if (k < ndirs) {
                                const int64_t ey = dy + cy[k], ex = dx + cx[k];
                                const int64_t e = ey * ey + ex * ex;
                                const bool better = e < best;
                                best = better ? e : best;
                                my[a] = better ? cy[k] : my[a];
                                mx[a] = better ? cx[k] : mx[a];
                            }
                        }
                    }
                    const float fy = (float)my[a], fx = (float)mx[a];
                    cost[a] = (flags & kDiagonal) ? __fsqrt_rn(__fadd_rn(__fmul_rn(fy, fy), __fmul_rn(fx, fx))) : __fadd_rn(fabsf(fy), fabsf(fx));
                    ay[a] += my[a];
                    ax[a] += mx[a];
                }
            }
            // ---------------------------------------------------------------- (3) accept conflicts (passenger_state.py:54-74)
            // while a passenger is claimed by several accepting agents, per env only the closest of ALL contested agents keeps
            // its claim (lowest index on ties); uncontested accepts survive.  One pass settles an env.
            bool wins[AMAX];
            {
                bool contested[AMAX];
                bool any = false;
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    contested[a] = false;
#pragma unroll
                    for (int o = 0; o < AMAX; ++o) contested[a] = contested[a] || (o != a && accept[a] && accept[o] && target[o] == target[a]);
                    any = any || contested[a];
                }
                int winner = -1;
                int64_t best = 0;
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    const bool better = contested[a] && (winner < 0 || dist2[a] < best);
                    best = better ? dist2[a] : best;
                    winner = better ? a : winner;
                }
#pragma unroll
                for (int a = 0; a < AMAX; ++a) wins[a] = accept[a] && (!contested[a] || a == winner);
                (void)any;
            }
            bool picked[AMAX], dropped[AMAX];
            int fares[AMAX];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                picked[a] = pick[a] && dist2[a] == 0;    // distance < 1e-6: the agent already stood on the passenger (:88-90)
                dropped[a] = drop[a] && dist2[a] == 0;   // transitions/passenger_exit.py:43-46
                fares[a] = dropped[a] ? pcol(PFARE, target[a], bl) : 0;
            }
            // the slot effects of this step (cleared cooperatively: the table is per launch)
            {
                uint4* const table = reinterpret_cast<uint4*>(&s_effect[0][0]);
                for (uint32_t i = (uint32_t)tid; i < P * (kBlock / 16); i += kBlock) table[i] = make_uint4(0, 0, 0, 0);
                __syncthreads();
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    const bool any = wins[a] || picked[a] || dropped[a];
                    if (any) {
                        uint32_t e = s_effect[target[a]][tid];
                        e = wins[a] ? ((e & ~31u) | (uint32_t)(a + 1)) : e;
                        e |= (picked[a] ? 32u : 0u) | (dropped[a] ? 64u : 0u);
                        s_effect[target[a]][tid] = (uint8_t)e;
                    }
                    s_move[a][tid] = make_short2((short)my[a], (short)mx[a]);
                }
            }
            // ---------------------------------------------------------------- (2b/3/4) one ordered pass over the env's slots:
            // riding passengers follow their driver, winners accept, picks ride, drops leave (order-preserving compaction)
            int kept = 0, unaccepted = 0;
            int wait_last[3] = {0, 0, 0};
            bool has_state[3] = {false, false, false};
            int owned[AMAX];  // passengers whose driver is agent a, any state (rideshare.py:343-344)
            // per-agent counters packed one byte per agent (a count is at most max_passengers <= 128): [0] driver == a, [1] accepted by a,
            // [2] riding with a, [3] driver == a and not unaccepted (the visible-but-not-general ones)
            constexpr int KW = (AMAX + 7) / 8;
            uint64_t tally[4][KW];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int w = 0; w < KW; ++w) tally[k][w] = 0;
            auto settle_slot = [&](const int s, const int (&was)[PCOLS], const uint32_t effect) {
                int v[PCOLS];
#pragma unroll
                for (int c = 0; c < PCOLS; ++c) v[c] = was[c];
                if (v[PSTATE] == 2) {  // best_moves[env, driver]; driver -1 wraps to the last agent like Python's index
                    const int drv = v[PDRIVER] < 0 ? A + v[PDRIVER] : v[PDRIVER];
                    const short2 move = s_move[min(max(drv, 0), AMAX - 1)][tid];
                    const bool real = drv >= 0 && drv < AMAX;
                    v[PY] += real ? move.x : 0;
                    v[PX] += real ? move.y : 0;
                }
                const int winner = (int)(effect & 31u) - 1;
                if (winner >= 0) {
                    v[PSTATE] = 1;
                    v[PACCEPTED] = nm;
                    v[PDRIVER] = winner;
                }
                if (effect & 32u) {
                    v[PSTATE] = 2;
                    v[PPICKED] = nm;
                }
                const bool removed = (effect & 64u) != 0;
                if (!removed) {
                    if (active) {
                        // a slot is rewritten only where it changes: every column once a removal has shifted the table
                        // (kept < s), otherwise just the columns this step touched (most slots: none)
                        const bool shifted = kept != s;
#pragma unroll
                        for (int c = 0; c < PCOLS; ++c)
                            if (shifted || v[c] != was[c]) pcol(c, kept, bl) = v[c];
                    }
                    const int st = v[PSTATE];
                    const int since = st == 0 ? v[PENTERED] : (st == 1 ? v[PACCEPTED] : v[PPICKED]);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        wait_last[k] = st == k ? nm - since : wait_last[k];
                        has_state[k] = has_state[k] || st == k;
                    }
                    unaccepted += st == 0 ? 1 : 0;
                    const int drv = v[PDRIVER];
#pragma unroll
                    for (int w = 0; w < KW; ++w) {
                        const int local = drv - 8 * w;
                        const uint64_t one = (local >= 0 && local < 8) ? (uint64_t)1 << (8 * local) : (uint64_t)0;
                        tally[0][w] += one;
                        tally[1][w] += st == 1 ? one : (uint64_t)0;
                        tally[2][w] += st == 2 ? one : (uint64_t)0;
                        tally[3][w] += st != 0 ? one : (uint64_t)0;
                    }
                    ++kept;
                }
            };
            for (int s0 = 0; s0 < count; s0 += kBatch) {
                int was[kBatch][PCOLS];
                uint32_t effect[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int s = min(s0 + u, (int)P - 1);
#pragma unroll
                    for (int c = 0; c < PCOLS; ++c) was[u][c] = pcol(c, s, bl);
                    effect[u] = s_effect[s][tid];
                }
                // a slot is only ever written at or below its own index (kept <= s), so the batch's loads see the old table
#pragma unroll
                for (int u = 0; u < kBatch; ++u)
                    if (s0 + u < count) settle_slot(s0 + u, was[u], effect[u]);
            }
            count = kept;
            // unpack the tallies (visible = the unaccepted ones, seen by every agent, plus the agent's own accepted / riding ones)
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                const int w = a >> 3, sh = 8 * (a & 7);
                owned[a] = (int)((tally[0][w] >> sh) & 0xFFu);
                n_accepted[a] = (int)((tally[1][w] >> sh) & 0xFFu);
                n_riding[a] = (int)((tally[2][w] >> sh) & 0xFFu);
                visible[a] = unaccepted + (int)((tally[3][w] >> sh) & 0xFFu);
            }
            // ---------------------------------------------------------------- (5) entry of the next timestep (rideshare.py:308)
            passenger_entry(d, sch, pas, Bu, bl, b, nm + 1, count, err, active, [&](int t) {
                wait_last[0] = nm - t;
                has_state[0] = true;
                ++unaccepted;
#pragma unroll
                for (int a = 0; a < AMAX; ++a) visible[a] += 1;
            });
            // ---------------------------------------------------------------- (6) rewards (rideshare.py:310-363)
            float global = 0.0f;
            if (flags & kWaiting) {
                // `global_rewards[envs] += cost` is an index_put without accumulation: per statement only the LAST passenger
                // (table order) of the env in that state takes effect (:323-333)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (has_state[k]) global = __fadd_rn(global, __fmul_rn(wait_last[k] >= d.wait_limit[k] ? 1.0f : 0.0f, d.general_wait_cost));
                if (has_state[0]) global = __fadd_rn(global, __fmul_rn(wait_last[0] >= d.long_wait_time ? 1.0f : 0.0f, d.long_wait_cost));
                const int slots = A * d.pool_limit;
                global = __fadd_rn(global, __fmul_rn(__fmul_rn(unaccepted >= slots - count ? 1.0f : 0.0f, -0.5f), (float)(slots - count)));
            }
            const int nm1 = nm + 1;
            trunc = (flags & kTruncate) ? nm1 >= d.max_steps : trunc;
            if (active) {
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    if (a < A) {
                        float r = 0.0f;
                        r = __fadd_rn(r, owned[a] > d.pool_limit ? d.pool_limit_cost : 0.0f);
                        r = __fadd_rn(r, __fmul_rn(act_id[a] == -1 ? 1.0f : 0.0f, d.noop_cost));
                        r = __fadd_rn(r, __fmul_rn(act_id[a] == 0 ? 1.0f : 0.0f, d.accept_cost));  // the accept ACTION, won or not
                        r = __fadd_rn(r, fares[a] > 0 ? __fsub_rn((float)fares[a], d.drop_cost) : 0.0f);
                        float dr = __fmul_rn(cost[a], d.move_cost);
                        if (flags & kVariableMove) dr = __fdiv_rn(dr, (float)(owned[a] + 1));
                        r = __fadd_rn(r, dr);
                        r = __fadd_rn(r, global);
                        at32(rowsf, (uint32_t)(d.r_rewards + a) * Bu + bl) = r;
                        if (flags & kTrackCumulative) {
                            float& cum = at32(rowsf, (uint32_t)(d.r_cum + a) * Bu + bl);
                            cum = __fadd_rn(cum, r);
                        }
                        if (flags & kTruncate) at32(rows1, (uint32_t)(d.u_trunc + a) * Bu + bl) = (uint8_t)trunc;
                        at32(rows, (uint32_t)(d.r_agents + 2 * a) * Bu + bl) = ay[a];
                        at32(rows, (uint32_t)(d.r_agents + 2 * a + 1) * Bu + bl) = ax[a];
                    }
                }
                at32(rows, (uint32_t)d.r_moves * Bu + bl) = nm1;
                at32(rows, (uint32_t)d.r_count * Bu + bl) = count;
            }
        } else {
            // rebuild only: statistics of the table as it stands
            for (int s = 0; s < count; ++s) {
                const int st = pcol(PSTATE, s, bl), drv = pcol(PDRIVER, s, bl);
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    visible[a] += (st == 0 || drv == a) ? 1 : 0;
                    n_accepted[a] += (st == 1 && drv == a) ? 1 : 0;
                    n_riding[a] += (st == 2 && drv == a) ? 1 : 0;
                }
            }
        }

        // ======================================================================================================
        // update_actions + update_observations (rideshare.py:367-467)
        // ======================================================================================================
        uint32_t cnt[AMAX + 1], excl[AMAX + 1];
        cnt[0] = active ? (uint32_t)count : 0u;
#pragma unroll
        for (int a = 0; a < AMAX; ++a) cnt[a + 1] = (active && a < A) ? (uint32_t)visible[a] : 0u;
        frz::scan_chunk<AMAX + 1>(s_scan, ws, launch, cnt, active, active && !trunc, A + 1, chunk, d.nchunks, excl, &err);

        if (active) {
            const int64_t cap = B * (int64_t)P;
            int32_t* const obs_self = reinterpret_cast<int32_t*>(arena + d.off_obs_self);
            int32_t* const obs_others = reinterpret_cast<int32_t*>(arena + d.off_obs_others);
            int64_t* const etc = reinterpret_cast<int64_t*>(arena + d.off_etc);
            int32_t* const task_values = reinterpret_cast<int32_t*>(arena + d.off_task_values);
            int64_t* const task_offsets = reinterpret_cast<int64_t*>(arena + d.off_task_offsets);
            int32_t* const agent_tasks = reinterpret_cast<int32_t*>(arena + d.off_agent_task_values);
            int64_t* const agent_maps = reinterpret_cast<int64_t*>(arena + d.off_agent_map_values);
            int64_t* const agent_offsets = reinterpret_cast<int64_t*>(arena + d.off_agent_offsets);
            int32_t* const agent_states = reinterpret_cast<int32_t*>(arena + d.off_agent_task_states);
            // agent observations: self = (y, x, #accepted, #riding); others = the other agents' self rows (:427-463)
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (a < A) {
                    reinterpret_cast<int4*>(obs_self)[a * B + b] = make_int4(ay[a], ax[a], n_accepted[a], n_riding[a]);
                    int4* others = reinterpret_cast<int4*>(obs_others) + (a * B + b) * (int64_t)(A - 1);
                    int j = 0;
#pragma unroll
                    for (int o = 0; o < AMAX; ++o)
                        if (o < A && o != a) others[j++] = make_int4(ay[o], ax[o], n_accepted[o], n_riding[o]);
                    at32(rows, (uint32_t)(d.r_atc + a) * Bu + bl) = visible[a];
                    agent_offsets[a * (B + 1) + b] = excl[a + 1];
                    if (b == B - 1) agent_offsets[a * (B + 1) + B] = (int64_t)excl[a + 1] + visible[a];
                }
            }
            etc[b] = count;
            task_offsets[b] = excl[0];
            if (b == B - 1) task_offsets[B] = (int64_t)excl[0] + count;
            // task rows (y, x, y_dest, x_dest, accepted_by | -100, riding_by | -100, fare, entered) (:405-414)
            int next[AMAX];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) next[a] = 0;
            auto emit_slot = [&](const int s, const int st, const int drv, const int4 lo, const int4 hi) {
                int4* row = reinterpret_cast<int4*>(task_values) + ((int64_t)excl[0] + s) * 2;
                if (!FRZ_RS_SKIP(2)) {
                    row[0] = lo;
                    row[1] = hi;
                }
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    if (!FRZ_RS_SKIP(1) && a < A && (st == 0 || drv == a)) {  // general or exclusive task of agent a (:378-380)
                        const int64_t at = a * cap + (int64_t)excl[a + 1] + next[a];
                        int4* arow = reinterpret_cast<int4*>(agent_tasks) + at * 2;
                        arow[0] = lo;
                        arow[1] = hi;
                        agent_maps[at] = s;
                        agent_states[at] = st;
                        ++next[a];
                    }
                }
            };
            for (int s0 = 0; s0 < (FRZ_RS_SKIP(4) ? 0 : count); s0 += kBatch) {
                int st[kBatch], drv[kBatch];
                int4 lo[kBatch], hi[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int s = min(s0 + u, (int)P - 1);
                    st[u] = pcol(PSTATE, s, bl);
                    drv[u] = pcol(PDRIVER, s, bl);
                    lo[u] = make_int4(pcol(PY, s, bl), pcol(PX, s, bl), pcol(PYD, s, bl), pcol(PXD, s, bl));
                    hi[u] = make_int4(0, 0, pcol(PFARE, s, bl), pcol(PENTERED, s, bl));
                }
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    hi[u].x = st[u] == 1 ? drv[u] : kNone;
                    hi[u].y = st[u] == 2 ? drv[u] : kNone;
                    if (s0 + u < count) emit_slot(s0 + u, st[u], drv[u], lo[u], hi[u]);
                }
            }
        }
        if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);
        frz::scan_end(ws, launch, chunk, d.nchunks);
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
    bool ticketed = false;  // more chunks than CUs: chunks are handed out in arrival order (frz_scan.h)
    int variant = 0;
};

namespace {

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

template <int AMAX>
void launch_variant(frz_rideshare_env* env, const int32_t* actions, int mode, hipStream_t stream) {
    if (mode == kRebuild)
        hipLaunchKernelGGL((rs_step_kernel<AMAX, kRebuild>), dim3(env->dev.nchunks), dim3(kBlock), 0, stream, env->arena, env->dev, actions, env->ticketed ? 1u : 0u);
    else
        hipLaunchKernelGGL((rs_step_kernel<AMAX, kStep>), dim3(env->dev.nchunks), dim3(kBlock), 0, stream, env->arena, env->dev, actions, env->ticketed ? 1u : 0u);
}

int launch(frz_rideshare_env* env, const int32_t* actions, int mode, hipStream_t stream) {
    switch (env->variant) {
        case 0: launch_variant<4>(env, actions, mode, stream); break;
        case 1: launch_variant<8>(env, actions, mode, stream); break;
        default: launch_variant<16>(env, actions, mode, stream); break;
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
    if ((int64_t)PCOLS * P * cfg->parallel_envs >= (int64_t)1 << 30 || (int64_t)(4 * A + 8) * cfg->parallel_envs >= (int64_t)1 << 30)
        return FRZ_E_INVALID;  // 32-bit element indices
    // coordinates stay inside +-16383: an agent's move (a difference of two positions with fast travel) travels through a 16-bit table
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
    p.r_agents = r, r += 2 * A;
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
    if ((int64_t)p.n_rows4 * B * 4 >= (int64_t)1 << 32) {  // the row blocks are addressed with 32-bit byte offsets
        delete env;
        return FRZ_E_INVALID;
    }
    p.off_rows1 = take((int64_t)p.n_rows1 * B);
    p.off_passengers = take((int64_t)PCOLS * P * B * 4);
    if ((int64_t)PCOLS * P * B * 4 >= (int64_t)1 << 32) {  // the passenger slots are addressed with 32-bit byte offsets
        delete env;
        return FRZ_E_INVALID;
    }
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
    *out = env;
    return FRZ_OK;
}

void frz_rideshare_destroy(frz_rideshare_env* env) { delete env; }

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
    out->agents = reinterpret_cast<int32_t*>(row4(p.r_agents));
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
    return launch(env, nullptr, kRebuild, static_cast<hipStream_t>(stream));
}

int frz_rideshare_reset(frz_rideshare_env* env, void* stream) {
    if (!env) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    const int blocks = (env->cfg.parallel_envs + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(rs_fill_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), env->arena);
    if (hipGetLastError() != hipSuccess) return FRZ_E_LAUNCH;
    return frz_rideshare_rebuild(env, stream);
}

int frz_rideshare_step(frz_rideshare_env* env, const int32_t* actions, void* stream) {
    if (!env || !actions) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;
    return launch(env, actions, kStep, static_cast<hipStream_t>(stream));
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
