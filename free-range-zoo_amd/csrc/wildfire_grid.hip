// wildfire_grid.hip — the wildfire step for grids above 16 cells: the grid's cells across the lanes of a wavefront.
//
// The reference's fire spread is a conv2d over any H x W (transitions/fire_spreads.py:32-59) and its open task / action sets are
// `nonzero()` lists over the whole grid (wildfire.py:586-717).  Here lane l of an env's FIELD wavefront holds cells l, l + 64, l + 128, ...
// (row-major cell index c = y * W + x; CPL = cells per lane, up to 16: 32 x 32 grids), so
//   * the cell arrays are env-major [B][H*W] (the reference's own layout): a row of 64 cells is one coalesced access;
//   * the lit map is CPL wave-uniform mask words: the 4-neighbour stencil is that map shifted by a row / a column (scalar shifts), the
//     shifted word being the lane predicate of the add (N, W, E, S in the reference's accumulation order);
//   * "the lit fires in row-major order" is a ballot per 64-cell chunk: a fire's task index = lit bits below it (lane rank) + the
//     chunks before; an agent's attackable set = lit mask & its precomputed range mask (per equipment state), so the index-th listed
//     task is a population-count descent and the per-agent counts are popcounts;
//   * the agents of a workgroup's four envs live in the lanes of ONE of its wavefronts (the CREW: lane 16 e + a) and hand their
//     fire-fighting power to the fields through an LDS array, added in agent order.
// One ParallelEnv.step() = two stream-ordered launches: wg_env_kernel (decode, optionally the uniform random policy, the 7 transitions,
// rewards / termination / bookkeeping, agent observations, per-env counts and the mask words the lists are made from) and wg_lists_kernel
// (launch-wide prefix sums -> jagged offsets + the batch totals the next step reads, then task rows, observation map and per-agent
// action / bad lists, one output entry per lane).  Same results as the env-per-lane kernels of wildfire.hip (the parity tests run both on
// shapes both accept).  No MFMA: integer / float32 elementwise work bound by instruction issue and HBM.
#include "frz_scan.h"
#include "frz_wave.h"

#include <cstddef>

#include "../../include/frz.h"

#include "wildfire_common.h"

namespace frz_wf {

namespace {

using frz::lane_rank;
using frz::popc_words;
using frz::read_lane;
using frz::select_nth;
using frz::wave_lds_sync;

constexpr int kEnvsPerBlock = kBlock / 64;  // one env per wavefront

struct CellTables {  // arena block behind off_cell_tables, HW entries each
    const float* fire_rewards;
    const int32_t *ignition, *fires0, *intensity0, *fuel0;
};
__device__ __forceinline__ CellTables cell_tables(const char* arena, const WgDev& d) {
    const char* base = arena + d.off_cell_tables;
    const int64_t n = (int64_t)d.HW * 4;
    return CellTables{reinterpret_cast<const float*>(base), reinterpret_cast<const int32_t*>(base + n), reinterpret_cast<const int32_t*>(base + 2 * n),
                      reinterpret_cast<const int32_t*>(base + 3 * n), reinterpret_cast<const int32_t*>(base + 4 * n)};
}

// ------------------------------------------------------------------------------------------------------------------------------------
// wg_env_kernel: everything of a step that concerns the FOUR envs of a workgroup (MODE kStep), or the counts / observations of their state
// as it stands (kRebuild), or of the configured initial state (kReset).
//   * every wavefront is the FIELD of one env: "cell-lane" values, element k of lane l belongs to cell 64 k + l.  It loads the cells,
//     makes the env's Philox draws, runs the fire transitions and the spread, writes the cells a transition touched, the lit mask words
//     and the lit cells' (fires, intensity) for the lists launch, and reduces what the rewards need;
//   * ONE wavefront of the workgroup (a different one from workgroup to workgroup, so the four SIMDs share that work) is also the CREW of
//     all four envs: lane 16 e + a holds agent a of the workgroup's env e.  It decodes the actions (or samples the uniform policy),
//     hands the fire-fighting power to the fields, runs the agent transitions, and after the fields are done computes rewards,
//     termination, the attackable sets and the observations.  An agent's rows are 4-byte pieces of [rows][B] arrays: one crew wavefront
//     touches 16 bytes of each line where four did 4 bytes each (the launch was bound by the number of such requests and by vector issue:
//     the agent phases ran at 12 / 64 lanes, four times per workgroup).
// The fields and the crew meet at three workgroup barriers and exchange through LDS.
// ------------------------------------------------------------------------------------------------------------------------------------
struct FieldSums {  // what an env's field leaves for the crew
    float fire_reward_sum, burnout_total;
    int n_put, n_burn, dead, pad_;
};

// FULL: H * W == 64 * CPL (8 x 8, 16 x 16, 32 x 32 ...): every lane's every cell exists and the `inside` tests fold away.
template <int CPL, int MODE, int RNG, bool FULL = false>
__global__ void __launch_bounds__(kBlock) wg_env_kernel(char* __restrict__ arena, const WgDev d, const int32_t* __restrict__ actions,
                                                         const float* __restrict__ field_rand, const float* __restrict__ agent_rand, const WgPolicy pol) {
    constexpr int W2 = 2 * CPL;  // 32-bit words of a cell mask
    constexpr int kCells = 64 * CPL;
    constexpr int E = kEnvsPerBlock, AP = 64 / E;  // envs per workgroup; crew lanes per env
    static_assert(AP >= FRZ_MAX_AGENTS, "a crew lane per agent");
    constexpr bool kStepping = MODE == kStep;
    constexpr bool kPhilox = kStepping && RNG == FRZ_RNG_PHILOX;
    __shared__ float s_draw[E][kPhilox ? 3 * kCells + 5 * FRZ_MAX_AGENTS + 8 : 1];  // the step's draws, by draw number (+ the last block's tail)
    __shared__ uint32_t s_policy[E][kPhilox ? 16 : 1];                          // the fused policy's Philox words: word a of env e
    __shared__ float s_power[E][kStepping ? kCells : 1];                        // fire-fighting power applied to each cell
    __shared__ uint32_t s_hits[E][kStepping ? kCells : 1];                      // agents fighting each cell
    __shared__ uint8_t s_put[E][kStepping ? kCells : 4];                        // cells put out this step (localized rewards)
    __shared__ float s_supp[E][FRZ_MAX_AGENTS];                                 // suppressants after the agent transitions
    __shared__ uint64_t s_lit[2][E][CPL];                                       // lit mask words: [0] of the loaded state, [1] after the step
    __shared__ FieldSums s_sums[E];
    // The small tables a crew lane indexes with values it has just loaded (equipment state -> bonuses, capacity index -> capacity, (agent,
    // equipment state) -> cells in range) are staged in LDS by the crew's FIRST loads: a lookup behind a loaded index is then an LDS read,
    // not one more trip to memory.
    constexpr int kTableWords = (int)(sizeof(WgAgentTable) / 4), kRangeStaged = CPL <= 4 ? 128 : 1;
    static_assert(kTableWords <= 128, "WgAgentTable is staged by two loads per lane");
    __shared__ uint32_t s_table[128];
    __shared__ uint64_t s_range[kRangeStaged];
    const int b = frz::env_of_wave<E>(d.B);
    if (b < 0) return;  // before any barrier: a barrier does not wait for wavefronts that have ended
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t B = d.B;
    const uint32_t Bu = (uint32_t)d.B, bu = (uint32_t)b;
    const int HW = d.HW, A = d.A, Wd = d.W;
    const uint32_t flags = d.flags;
    const int b0 = b - wave;                                                   // the workgroup's first env
    const int n_envs = (int)(B - b0 < E ? B - b0 : E);                         // its envs (the last workgroup of a ragged batch has fewer)
    const bool crew = wave == (int)(blockIdx.x % (uint32_t)n_envs);            // wave-uniform
    const int ce = lane / AP, ca = lane % AP;                                  // crew lane -> (env of the workgroup, agent)
    const bool is_agent = ca < A && ce < n_envs;
    const int agent = ca < A ? ca : A - 1;                                     // clamped: unconditional loads
    const uint32_t cbu = (uint32_t)(b0 + (ce < n_envs ? ce : n_envs - 1));     // the crew lane's env
    int32_t* const rows = reinterpret_cast<int32_t*>(arena + d.off_rows4);
    float* const rowsf = reinterpret_cast<float*>(arena + d.off_rows4);
    int64_t* const rows8 = reinterpret_cast<int64_t*>(arena + d.off_rows8);
    uint8_t* const rows1 = reinterpret_cast<uint8_t*>(arena + d.off_rows1);
    int32_t* const fires = reinterpret_cast<int32_t*>(arena + d.off_cells) + (int64_t)b * HW;  // this env's cells
    int32_t* const intensity = fires + B * HW;
    int32_t* const fuel = intensity + B * HW;
    const WgAgentTable* const table = reinterpret_cast<const WgAgentTable*>(arena + d.off_agent_table);
    const CellTables cells = cell_tables(arena, d);
    const uint64_t* const range = reinterpret_cast<const uint64_t*>(arena + d.off_range);
    const int range_words = A * d.S * CPL;
    const bool range_staged = CPL <= 4 && range_words <= kRangeStaged;  // uniform

    // ---------------------------------------------------------------- loads: field
    const uint32_t epoch = *reinterpret_cast<const uint32_t*>(arena + d.off_epoch);
    const uint32_t* const totals = reinterpret_cast<const uint32_t*>(arena + d.off_totals);
    int f[CPL], in[CPL], fu[CPL], ign[kStepping ? CPL : 1];
    float fire_reward[kStepping ? CPL : 1];
    bool inside[CPL];  // the lane's k-th cell exists
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c = lane + 64 * k;
        inside[k] = FULL || c < HW;
        const uint32_t cc = (uint32_t)(inside[k] ? c : HW - 1);
        if (kStepping) ign[k] = frz::at32(cells.ignition, cc), fire_reward[k] = frz::at32(cells.fire_rewards, cc);  // with the first loads, not behind the tests that need them
        if (MODE == kReset) {  // wildfire.py:347-351: +type on the configured lit cells, -type elsewhere, ...
            f[k] = frz::at32(cells.fires0, cc), in[k] = frz::at32(cells.intensity0, cc), fu[k] = frz::at32(cells.fuel0, cc);
        } else {
            f[k] = frz::at32(fires, cc), in[k] = frz::at32(intensity, cc), fu[k] = frz::at32(fuel, cc);
        }
    }
    int nm = 0;         // the field's env: moves so far (Philox counter)
    uint32_t seed = 0;  // and its seed
    uint32_t left_alive = 1, left_running = 1;
    if (kStepping) {
        nm = frz::at32(rows, (uint32_t)d.r_moves * Bu + bu);
        seed = (uint32_t)frz::at32(rows, (uint32_t)d.r_seeds * Bu + bu);
        // what the lists launch left after the previous step: slot (epoch + 1) & 1 — both slots are requested with everything else and
        // the epoch picks one (a load whose address waits for the epoch would be one more trip to memory)
        const bool odd = ((epoch + 1u) & 1u) != 0u;
        const uint32_t tl0 = totals[A + 1], tl1 = totals[kTotalsStride + A + 1], tr0 = totals[A + 2], tr1 = totals[kTotalsStride + A + 2];
        left_alive = odd ? tl1 : tl0;
        left_running = odd ? tr1 : tr0;
    }
    // ---------------------------------------------------------------- loads: crew (lane 16 e + a: agent a of the workgroup's env e)
    float supp = d.initial_suppressant, capa = d.initial_capacity, cum0 = 0.0f;
    int eqs = d.initial_equipment;
    int c_nm = 0, c_nb = 0;
    uint32_t c_seed = 0;
    bool term0 = false, trunc0 = false;
    int act_idx = 0, act_id = -1;
    uint32_t agent_tasks_anywhere = 1;  // the batch total of the agent's attackable tasks after the previous launch
    uint32_t table_w0 = 0, table_w1 = 0;
    uint64_t range_w0 = 0, range_w1 = 0;
    if (crew) {
        table_w0 = reinterpret_cast<const uint32_t*>(table)[lane < kTableWords ? lane : 0];
        table_w1 = reinterpret_cast<const uint32_t*>(table)[lane + 64 < kTableWords ? lane + 64 : 0];
        if (CPL <= 4) {
            range_w0 = range[lane < range_words ? lane : 0];
            range_w1 = range[lane + 64 < range_words ? lane + 64 : 0];
        }
        if (MODE != kReset) {
            supp = frz::at32(rowsf, (uint32_t)(d.r_supp + agent) * Bu + cbu);
            capa = frz::at32(rowsf, (uint32_t)(d.r_cap + agent) * Bu + cbu);
            eqs = frz::at32(rows, (uint32_t)(d.r_equip + agent) * Bu + cbu);
            term0 = frz::at32(rows1, (uint32_t)d.u_term * Bu + cbu) != 0;
            trunc0 = frz::at32(rows1, (uint32_t)d.u_trunc * Bu + cbu) != 0;
        }
        if (kStepping) {
            c_nm = frz::at32(rows, (uint32_t)d.r_moves * Bu + cbu);
            c_nb = frz::at32(rows, (uint32_t)d.r_burnouts * Bu + cbu);
            c_seed = (uint32_t)frz::at32(rows, (uint32_t)d.r_seeds * Bu + cbu);
            if (flags & kTrackCumulative) cum0 = frz::at32(rowsf, (uint32_t)(d.r_cum + agent) * Bu + cbu);
            if (!pol.on) {
                const int2 a2 = frz::at32(reinterpret_cast<const int2*>(actions), (uint32_t)agent * Bu + cbu);
                act_idx = a2.x;
                act_id = is_agent ? a2.y : -1;
            }
            const uint32_t ta0 = totals[1 + agent], ta1 = totals[kTotalsStride + 1 + agent];
            agent_tasks_anywhere = ((epoch + 1u) & 1u) ? ta1 : ta0;
        }
    }

    if (kStepping) {
        // utils/env.py:211-213 — every per-agent step() is a no-op once ALL envs are terminated or ALL are truncated; the parallel adapter
        // (utils/conversions.py:87-90) then adds the stale aec rewards once per agent call.  (Batch totals: every wavefront takes this branch or none.)
        // (words 40 / 41 of the epoch block, by the step's parity: "this step changed nothing" — read by the lists launch of an overlapped
        // rollout, whose mask words of this parity are then not this step's; the lists the previous step left stay as they are)
        if (blockIdx.x == 0 && threadIdx.x == 0)
            reinterpret_cast<uint32_t*>(arena + d.off_epoch)[40 + (d.parity & 1)] = (left_alive == 0u || left_running == 0u) ? 1u : 0u;
        if (left_alive == 0u || left_running == 0u) {
            if (crew && !frz::at32(rows1, (uint32_t)d.u_frozen * Bu + cbu)) {
                if (is_agent) {
                    const float r = frz::at32(rowsf, (uint32_t)(d.r_rewards + agent) * Bu + cbu);
                    float acc = 0.0f;
                    for (int j = 0; j < A; ++j) acc = acc + r;
                    frz::at32(rowsf, (uint32_t)(d.r_rewards + agent) * Bu + cbu) = acc;
                }
                wave_lds_sync();  // (nothing in LDS: keeps the flag's store behind every lane's read of it)
                if (ca == 0 && ce < n_envs) frz::at32(rows1, (uint32_t)d.u_frozen * Bu + cbu) = 1;
            }
            return;
        }
    }

    uint64_t lit[CPL];  // the field's lit fires (fires > 0) per 64-cell chunk, wave-uniform
#pragma unroll
    for (int k = 0; k < CPL; ++k) lit[k] = __ballot(inside[k] && f[k] > 0);
    uint32_t err = 0;

    // cells a crew lane's agent reaches at an equipment state (utils/in_range_check.py:5-23 precomputed per (agent, state) at create) that
    // are lit in its env, as mask words
    auto attackable_words = [&](int slot, float suppressant, int equipment, uint32_t (&out)[W2]) {
        const int at = (agent * d.S + equipment) * CPL;  // the table's stride is the kernel's chunk count
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const uint64_t mine = range_staged ? s_range[CPL <= 4 ? at + k : 0] : range[at + k];
            const uint64_t m = suppressant > 0.0f ? (s_lit[slot][ce][k] & mine) : 0ull;  // no suppressant: nothing to attack (wildfire.py:604-623)
            out[2 * k] = (uint32_t)m;
            out[2 * k + 1] = (uint32_t)(m >> 32);
        }
    };
    constexpr int kEqAt = (int)(offsetof(WgAgentTable, eq) / 4), kCapsAt = (int)(offsetof(WgAgentTable, caps) / 4);
    constexpr int kAyAt = (int)(offsetof(WgAgentTable, ay) / 4), kAxAt = (int)(offsetof(WgAgentTable, ax) / 4);
    const float* const tablef = reinterpret_cast<const float*>(s_table);

    float reward = 0.0f;
    FieldSums sums{0.0f, 0.0f, 0, 0, 0, 0};  // field: what this step leaves for the crew
    int hit = -1, tcell = 0;
    bool good = false, refill = false;  // crew: the agent fights a listed task / refills
    float r_field[kStepping ? 3 : 1][CPL];
    if (kStepping) {
        // ------------------------------------------------------------ (1) field: the env's draws, the loaded state's lit words
        // FRZ_RNG_PHILOX (include/frz.h): draw u = 24-bit field u % 5 of block (u / 5, step, 0, 0) keyed by (seed, 0x46525A00); field
        // event e of cell c is draw e * HW + c, agent event e of agent a is draw 3 * HW + e * A + a.  Lane j computes blocks j, j + 64, ...
        // and parks their draws in LDS by draw number.  The fused policy draws Philox blocks too (ceil(A / 4) of them, another key and
        // counter): when they fit into the lanes the step's own blocks leave idle, both streams are ONE Philox evaluation with per-lane
        // inputs; the policy words go to s_policy.
        const int U = 3 * HW + 5 * A, NB = (U + 4) / 5, PB = (A + 3) / 4;
        const bool merged = kPhilox && pol.on && NB + PB <= 64;
        if (kPhilox) {
            for (int j = lane; j < NB + (merged ? PB : 0); j += 64) {
                const bool mine = j < NB;  // a block of the step's draws; otherwise policy block j - NB
                const frz::Philox4 w = frz::philox4x32_10(mine ? (uint32_t)j : (uint32_t)(j - NB), mine ? (uint32_t)nm : 0u, mine ? 0u : pol.step_lo,
                                                          mine ? 0u : pol.step_hi, mine ? seed : (pol.seed_lo ^ seed), mine ? 0x46525A00u : pol.seed_hi);
                if (mine) {
                    float* const out = &s_draw[wave][5 * j];  // the tail past U stays inside the array (sized for 5 * NB)
                    out[0] = frz::philox_unit24<0>(w);
                    out[1] = frz::philox_unit24<1>(w);
                    out[2] = frz::philox_unit24<2>(w);
                    out[3] = frz::philox_unit24<3>(w);
                    out[4] = frz::philox_unit24<4>(w);
                } else {
                    uint32_t* const out = &s_policy[wave][4 * (j - NB)];
                    out[0] = w.w[0], out[1] = w.w[1], out[2] = w.w[2], out[3] = w.w[3];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            s_power[wave][lane + 64 * k] = 0.0f, s_hits[wave][lane + 64 * k] = 0u;
            if (lane == 0) s_lit[0][wave][k] = lit[k];
        }
        if (crew) {
            s_table[lane] = table_w0, s_table[lane + 64] = table_w1;
            if constexpr (CPL <= 4) {
                if (range_staged) s_range[lane] = range_w0, s_range[lane + 64] = range_w1;
            }
        }
        __syncthreads();

        // ------------------------------------------------------------ (2) crew: action decode (wildfire.py:427-483).  Only what the fields
        // wait for — who fights which cell with what power; the agent transitions follow when the fields are done.
        // The action mapping of the previous rebuild is a pure function of the state it was built from, which is the state just
        // loaded: attackable set of agent a = lit fires within its (equipment-adjusted) range, non-empty only while it has suppressant.
        if (crew) {
            const float base_power = tablef[agent];
            uint32_t ok0[W2], sel[W2];
            attackable_words(0, supp, eqs, ok0);
            const bool show_bad = (flags & kShowBad) != 0;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {  // the tasks the agent's action space lists
                const uint64_t l = s_lit[0][ce][k];
                sel[2 * k] = show_bad ? (uint32_t)l : ok0[2 * k];
                sel[2 * k + 1] = show_bad ? (uint32_t)(l >> 32) : ok0[2 * k + 1];
            }
            const int n_listed = popc_words(sel);
            if (pol.on) {
                // uniform random policy over OneOf([task] * n + [noop]) (spaces/actions.py:23-41, baselines/random.py:20), the stream of
                // frz_wildfire_random_policy: agent a draws word a % 4 of Philox(counter (a / 4, 0, step), key (policy seed ^ env seed));
                // member j ~ U{0..n}; j < n -> [j, 0] (fight task j), j == n -> [n, -1] (noop / refill)
                uint32_t word;
                if (merged) {
                    word = s_policy[ce][agent];
                } else {
                    const frz::Philox4 w = frz::philox4x32_10((uint32_t)agent >> 2, 0u, pol.step_lo, pol.step_hi, pol.seed_lo ^ c_seed, pol.seed_hi);
                    word = w.w[0];
                    word = (agent & 3) == 1 ? w.w[1] : word;
                    word = (agent & 3) == 2 ? w.w[2] : word;
                    word = (agent & 3) == 3 ? w.w[3] : word;
                }
                const int j = (int)(((uint64_t)word * (uint64_t)(n_listed + 1)) >> 32);
                act_idx = j < n_listed ? j : n_listed;
                act_id = (is_agent && j < n_listed) ? 0 : -1;
                if (is_agent) frz::at32(reinterpret_cast<int2*>(pol.actions_out), (uint32_t)agent * Bu + cbu) = make_int2(act_idx, act_id);
            }
            refill = act_id == -1;
            const bool skipped = agent_tasks_anywhere == 0u;  // quirk wildfire.py:434-435: no attackable task in ANY env of the batch
            const bool fight = is_agent && !refill && !skipped;
            const int target = select_nth(sel, act_idx);  // the cell of the index-th listed task, -1 outside the list
            const bool valid = target >= 0;
            tcell = valid ? target : 0;
            bool attackable = false;
#pragma unroll
            for (int i = 0; i < W2; ++i) attackable = attackable || ((tcell >> 5) == i && ((ok0[i] >> (tcell & 31)) & 1u));
            good = fight && valid && (!show_bad || attackable);
            if (__ballot(fight && !valid)) err |= FRZ_ERR_BAD_ACTION_INDEX;
            const float power = base_power + tablef[kEqAt + 4 * eqs + 1];
            hit = good ? tcell : -1;
            reward = (fight && !good) ? d.bad_attack_penalty : 0.0f;  // assignment, wildfire.py:477
            // applied power per cell (wildfire.py:455-470).  The reference adds the agents' powers in agent order; float addition of TWO
            // terms does not depend on the order, so the agents add theirs with one LDS atomic each unless some cell is hit by three or
            // more agents — then one agent of every env at a time, in order.
            if (good) atomicAdd(&s_hits[ce][tcell], 1u);
            wave_lds_sync();
            if (__ballot(good && s_hits[ce][tcell] > 2u)) {
                for (int a = 0; a < A; ++a) {
                    if (ca == a && good) s_power[ce][tcell] = s_power[ce][tcell] + power;
                    wave_lds_sync();
                }
            } else {
                if (good) atomicAdd(&s_power[ce][tcell], power);
            }
        }
        // ------------------------------------------------------------ field: the cells' draws (read while the crew decodes)
        if (kPhilox) {
#pragma unroll
            for (int e = 0; e < 3; ++e)
#pragma unroll
                for (int k = 0; k < CPL; ++k) r_field[e][k] = inside[k] ? s_draw[wave][e * HW + lane + 64 * k] : 1.0f;
        } else {  // the tensor generator.generate(B, 3, (H, W)) returns (wildfire.py:409-410)
#pragma unroll
            for (int e = 0; e < 3; ++e)
#pragma unroll
                for (int k = 0; k < CPL; ++k) r_field[e][k] = inside[k] ? field_rand[((int64_t)e * B + b) * HW + lane + 64 * k] : 1.0f;
        }
        __syncthreads();

        int f_in[CPL], in_in[CPL], fu_in[CPL];  // the state as loaded
#pragma unroll
        for (int k = 0; k < CPL; ++k) f_in[k] = f[k], in_in[k] = in[k], fu_in[k] = fu[k];
        // ------------------------------------------------------------ (4) field: fire increase / decrease per cell
        const int almost_state = d.num_fire_states - 2, burnout_state = d.num_fire_states - 1;
        const float p_unmet = (flags & kStochIncrease) ? d.p_increase : 1.0f;
        const float p_almost = (flags & kStochBurnouts) ? d.p_burnout : d.p_increase;  // fire_increase.py:77-80
        bool burned[CPL], put_out[CPL];
        uint64_t burning[CPL];  // wave-uniform mask words, as `lit`
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const float ap = s_power[wave][lane + 64 * k];
            {  // transitions/fire_increase.py:61-91
                const int required = f[k] >= 0 ? f[k] : 0;
                const float diff = (float)required - ap;
                const bool burns = f[k] > 0 && in[k] > 0;
                const bool unmet = diff > 0.0f && burns;
                const bool almost = unmet && in[k] == almost_state;
                float prob = unmet ? (almost ? p_almost : p_unmet) : 0.0f;
                prob = clamp01(prob);
                const bool inc = inside[k] && r_field[0][k] < prob;
                in[k] += inc ? 1 : 0;
                const bool bo = inc && in[k] >= burnout_state;
                f[k] = bo ? -f[k] : f[k];
                fu[k] = bo ? (fu[k] - 1 < 0 ? 0 : fu[k] - 1) : fu[k];
                burned[k] = bo;
            }
            {  // transitions/fire_decrease.py:56-77: p = p_dec + ((-1 * diff) * bonus), each op rounded
                const int required = f[k] >= 0 ? f[k] : 0;
                const float diff = (float)required - ap;
                const bool burns = f[k] > 0 && in[k] > 0;
                const bool met = diff <= 0.0f && burns;
                const float stoch_p = __fadd_rn(d.p_decrease, __fmul_rn(__fmul_rn(-1.0f, diff), d.decrease_bonus));
                float prob = met ? ((flags & kStochDecrease) ? stoch_p : 1.0f) : 0.0f;
                prob = clamp01(prob);
                const bool dec = inside[k] && r_field[1][k] < prob;
                in[k] -= dec ? 1 : 0;
                const bool po = dec && in[k] <= 0;
                f[k] = po ? -f[k] : f[k];
                fu[k] = po ? fu[k] - 1 : fu[k];  // unclamped, :75
                put_out[k] = po;
            }
            burning[k] = __ballot(inside[k] && f[k] > 0 && in[k] > 0);  // lit map after increase / decrease
            if (flags & kLocalize) s_put[wave][lane + 64 * k] = (uint8_t)put_out[k];
        }
        // ------------------------------------------------------------ (5) field: fire spread, 4-neighbour stencil on the lit map
        // (transitions/fire_spreads.py:44-57; conv2d accumulation order N, W, E, S).  The map is CPL wave-uniform mask words, so "my
        // northern neighbour burns" is the map shifted by a row — scalar shifts — and the shifted word is the lane predicate of the add.
        bool any_fire = false;
        int fuel_sum = 0;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c = lane + 64 * k;
            const int y = (int)(((uint32_t)c * d.inv_w) >> 16), x = c - y * Wd;
            const uint64_t first_col = __ballot(x == 0), last_col = __ballot(x == Wd - 1);
            const uint64_t below = k > 0 ? burning[k > 0 ? k - 1 : 0] : 0ull, above = k + 1 < CPL ? burning[k + 1 < CPL ? k + 1 : k] : 0ull;
            const uint64_t from_n = (burning[k] << Wd) | (below >> (64 - Wd));        // cell c - W burns (W <= 32; nothing shifts in above row 0)
            const uint64_t from_s = (burning[k] >> Wd) | (above << (64 - Wd));        // cell c + W burns (no bits past the last cell)
            const uint64_t from_w = ((burning[k] << 1) | (below >> 63)) & ~first_col;  // cell c - 1, same row
            const uint64_t from_e = ((burning[k] >> 1) | (above << 63)) & ~last_col;   // cell c + 1, same row
            float prob = 0.0f;
            prob = __fadd_rn(prob, frz::where_lane(from_n, d.spread_n));
            prob = __fadd_rn(prob, frz::where_lane(from_w, d.spread_w));
            prob = __fadd_rn(prob, frz::where_lane(from_e, d.spread_e));
            prob = __fadd_rn(prob, frz::where_lane(from_s, d.spread_s));
            bool unlit = f[k] < 0 && in[k] == 0;
            unlit = unlit && (!(flags & kUseFuel) || fu[k] > 0);
            prob = unlit ? __fadd_rn(prob, d.random_ignition) : 0.0f;
            const bool spread = inside[k] && r_field[2][k] < prob;
            f[k] = spread ? -f[k] : f[k];
            in[k] = spread ? ign[k] : in[k];
            any_fire = any_fire || (inside[k] && f[k] > 0);
            fuel_sum += inside[k] ? fu[k] : 0;
        }
        // termination test (wildfire.py:560-570): no lit fire left (and no fuel when fuel is tracked)
        bool dead = __ballot(any_fire) == 0ull;
        if (flags & kUseFuel) dead = dead && (int)frz::wave_sum((uint32_t)fuel_sum) <= 0;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            f[k] = dead ? 0 : f[k];  // :570
            // a 64-cell run that no transition touched (most of a large grid, most steps) is not written back
            const uint32_t c = (uint32_t)(lane + 64 * k);
            if (__ballot(inside[k] && f[k] != f_in[k])) {
                if (inside[k]) frz::at32(fires, c) = f[k];
            }
            if (__ballot(inside[k] && in[k] != in_in[k])) {
                if (inside[k]) frz::at32(intensity, c) = in[k];
            }
            if (__ballot(inside[k] && fu[k] != fu_in[k])) {
                if (inside[k]) frz::at32(fuel, c) = fu[k];
            }
            lit[k] = __ballot(inside[k] && f[k] > 0);
        }
        // ------------------------------------------------------------ (6) field: what the rewards need (wildfire.py:534-582)
        // put-out cells pay their reward, burnt-out cells their penalty: summed in cell order (few cells: a walk over the set bits)
        float fire_reward_sum = 0.0f, burnout_total = 0.0f;
        int n_burn = 0, n_put = 0;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const uint64_t pm = __ballot(put_out[k]), bm = __ballot(burned[k]);
            n_put += (int)__popcll(pm);
            n_burn += (int)__popcll(bm);
            if (pm | bm) {
                const float fr = fire_reward[k];
                for (uint64_t m = pm; m; m &= m - 1) fire_reward_sum = __fadd_rn(fire_reward_sum, __int_as_float(read_lane(__float_as_int(fr), frz::first_bit(m))));
                for (uint64_t m = bm; m; m &= m - 1) {
                    const float r = __int_as_float(read_lane(__float_as_int(fr), frz::first_bit(m)));
                    burnout_total = __fadd_rn(burnout_total, (flags & kPenaltyScaled) ? __fmul_rn(-1.0f, r) : d.burnout_penalty);
                }
            }
        }
        sums = FieldSums{fire_reward_sum, burnout_total, n_put, n_burn, dead ? 1 : 0, 0};
    }
    if (MODE == kReset) {
#pragma unroll
        for (int k = 0; k < CPL; ++k)
            if (inside[k]) {
                const uint32_t c = (uint32_t)(lane + 64 * k);
                frz::at32(fires, c) = f[k], frz::at32(intensity, c) = in[k], frz::at32(fuel, c) = fu[k];
            }
    }
    // ---------------------------------------------------------------- field: what the lists launch reads instead of the cells — the lit
    // mask words and (fires, intensity) of the lit cells in task order
    {
        uint64_t* const litmap = reinterpret_cast<uint64_t*>(arena + d.off_litmap);
        int2* const lit_cells = reinterpret_cast<int2*>(arena + d.off_lit_cells) + (int64_t)b * ((HW + 1) & ~1);  // 16-byte aligned rows
        int before = 0;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            if ((lit[k] >> lane) & 1ull) frz::at32(lit_cells, (uint32_t)(before + lane_rank(lit[k]))) = make_int2(f[k], in[k]);
            before += (int)__popcll(lit[k]);
        }
        if (lane == 0) {  // (one predicated region for everything lane 0 leaves: each costs three scalar instructions)
#pragma unroll
            for (int k = 0; k < CPL; ++k) litmap[(int64_t)k * B + b] = lit[k], s_lit[1][wave][k] = lit[k];
            frz::at32(rows8, (uint32_t)d.q_etc * Bu + bu) = before;  // environment_task_count
            if (kStepping) s_sums[wave] = sums;
        }
    }
    if (!kStepping && crew) {
        s_table[lane] = table_w0, s_table[lane + 64] = table_w1;
        if constexpr (CPL <= 4) {
            if (range_staged) s_range[lane] = range_w0, s_range[lane + 64] = range_w1;
        }
    }
    __syncthreads();
    if (!crew) {
        if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);
        return;
    }

    // ---------------------------------------------------------------- crew: rewards and termination (wildfire.py:534-582), bookkeeping,
    // state rows of the agents, counts of the rebuilt spaces (wildfire.py:586-666), agent observations (wildfire.py:668-717)
    const float base_power = tablef[agent];
    bool term = term0, trunc = trunc0;
    if (kStepping) {
        // the agents' draws (the fields' LDS is theirs until the workgroup ends)
        float r_agent[5];
        if (kPhilox) {
#pragma unroll
            for (int e = 0; e < 5; ++e) r_agent[e] = s_draw[ce][3 * HW + e * A + agent];
        } else {  // the tensor generator.generate(B, 5, (A,)) returns (wildfire.py:409-410)
#pragma unroll
            for (int e = 0; e < 5; ++e) r_agent[e] = agent_rand[((int64_t)e * B + cbu) * A + agent];
        }
        // -------------------------------------------------------- (3) crew: agent transitions (suppressant decrease, equipment, refill, capacity)
        {
            // transitions/suppressant_decrease.py:56-61
            const bool dec = good && (!(flags & kStochSuppDecrease) || r_agent[0] < d.p_supp_decrease);
            float s = dec ? supp - 1.0f : supp;
            s = s < 0.0f ? 0.0f : s;
            // transitions/equipment.py:51-75 (masks from the value before any write)
            const int e0 = eqs, top = d.S - 1;
            const bool pristine = e0 == top, damaged = e0 == 0, inter = !pristine && !damaged;
            const float r1 = r_agent[1];
            const bool repairs = (flags & kStochRepair) ? (damaged && r1 < d.p_repair) : damaged;
            const bool crit = (flags & kCritical) && pristine && r1 < d.p_critical;
            bool degr = (flags & kStochDegrade) ? ((pristine || inter) && r1 < d.p_degrade) : (inter || pristine);
            degr = degr && !crit;
            int e = repairs ? top : e0;
            e = crit ? 0 : e;
            e = degr ? e - 1 : e;
            // transitions/suppressant_refill.py:63-70 (bonus from the NEW equipment state)
            const bool inc = refill && (!(flags & kStochRefill) || r_agent[2] < d.p_refill);
            s = inc ? capa + tablef[kEqAt + 4 * e] : s;
            // transitions/capacity.py:52-64: bucketize(r, cumsum) = #{j : cum[j] < r} (cum padded with +inf, clamped to the last capacity)
            int ci = 0;
#pragma unroll
            for (int j = 0; j < FRZ_MAX_CAPACITIES; ++j) ci += r_agent[3] > d.cum[j] ? 1 : 0;
            ci = ci > d.K - 1 ? d.K - 1 : ci;
            const float new_max = tablef[kCapsAt + ci];
            const bool sw = inc && (!(flags & kStochSwitch) || r_agent[4] < d.p_switch);
            const float bonus = s - capa;
            capa = sw ? new_max : capa;
            s = sw ? new_max + bonus : s;
            supp = s;
            eqs = e;
        }
        const FieldSums fsum = s_sums[ce < n_envs ? ce : 0];
        const bool dead = fsum.dead != 0;
        const bool newly = !term0 && dead;
        // correctly rounded float32 log via double (matches the oracle bit for bit; the reference's torch.log is a <=1-ulp float32 log)
        float log_burnouts = 0.0f;
        if (newly && d.termination_kappa != 0.0f) log_burnouts = (float)log((double)c_nb + 1.0);
        float term_reward = __fsub_rn(d.termination_reward, __fmul_rn(d.termination_kappa, log_burnouts));
        term_reward = term_reward < 0.0f ? 0.0f : term_reward;
        float base_reward = fsum.fire_reward_sum;
        if (flags & kLocalize) {  // only the put-outs this agent last hit (:546-553)
            const bool mine = hit >= 0 && s_put[ce][tcell] != 0;
            base_reward = mine ? frz::at32(cells.fire_rewards, (uint32_t)tcell) : 0.0f;
        }
        reward = __fadd_rn(reward, __fadd_rn(base_reward, fsum.burnout_total));
        reward = newly ? __fadd_rn(reward, term_reward) : reward;
        const int nm1 = c_nm + 1;
        trunc = (flags & kTruncate) ? nm1 >= d.max_steps : trunc0;
        term = term0 || dead;
        if (ca == 0 && ce < n_envs) {
            frz::at32(rows, (uint32_t)d.r_moves * Bu + cbu) = nm1;
            frz::at32(rows, (uint32_t)d.r_burnouts * Bu + cbu) = c_nb + fsum.n_burn;
            frz::at32(rows8, (uint32_t)d.q_burnouts * Bu + cbu) = fsum.n_burn;
            frz::at32(rows8, (uint32_t)d.q_putouts * Bu + cbu) = fsum.n_put;
        }
    }
    if (MODE == kReset && ca == 0 && ce < n_envs) {
        frz::at32(rows, (uint32_t)d.r_moves * Bu + cbu) = 0;
        frz::at32(rows, (uint32_t)d.r_burnouts * Bu + cbu) = 0;
        frz::at32(rows8, (uint32_t)d.q_burnouts * Bu + cbu) = 0;
        frz::at32(rows8, (uint32_t)d.q_putouts * Bu + cbu) = 0;
        frz::at32(rows1, (uint32_t)d.u_frozen * Bu + cbu) = 0;
        if (pol.on)  // frz_wildfire_reset_reseed: seed increment, modulo 2^32
            frz::at32(rows, (uint32_t)d.r_seeds * Bu + cbu) = (int32_t)((uint32_t)frz::at32(rows, (uint32_t)d.r_seeds * Bu + cbu) + pol.seed_lo);
    }
    uint32_t ok1[W2];
    attackable_words(1, supp, eqs, ok1);
    const int n_attackable = popc_words(ok1);
    if (is_agent) {
        if (MODE != kRebuild) {
            frz::at32(rowsf, (uint32_t)(d.r_supp + agent) * Bu + cbu) = supp;
            frz::at32(rowsf, (uint32_t)(d.r_cap + agent) * Bu + cbu) = capa;
            frz::at32(rows, (uint32_t)(d.r_equip + agent) * Bu + cbu) = eqs;
            frz::at32(rowsf, (uint32_t)(d.r_rewards + agent) * Bu + cbu) = reward;
            frz::at32(rows1, (uint32_t)(d.u_term + agent) * Bu + cbu) = (uint8_t)term;
            if (MODE == kReset || (flags & kTruncate)) frz::at32(rows1, (uint32_t)(d.u_trunc + agent) * Bu + cbu) = (uint8_t)trunc;
            if (MODE == kReset || (flags & kTrackCumulative)) frz::at32(rowsf, (uint32_t)(d.r_cum + agent) * Bu + cbu) = __fadd_rn(cum0, reward);
        }
        frz::at32(rows, (uint32_t)(d.r_atc + agent) * Bu + cbu) = n_attackable;
        s_supp[ce][agent] = supp;
        uint64_t* const okmap = reinterpret_cast<uint64_t*>(arena + d.off_okmap);
#pragma unroll
        for (int k = 0; k < CPL; ++k) okmap[((int64_t)agent * CPL + k) * B + cbu] = (uint64_t)ok1[2 * k] | ((uint64_t)ok1[2 * k + 1] << 32);
    }
    // agent observations: self = (y, x, fire_reduction_power, suppressant); others = (y, x[, power][, suppressant]) of the other agents.
    // Agents do not move and their base power is configuration, so only the suppressant column changes in a step — but whole records are
    // written (a 4-byte piece of a 16-byte record leaves every sector of these arrays partly dirty).
    float* const obs_self = reinterpret_cast<float*>(arena + d.off_obs_self);
    float* const obs_others = reinterpret_cast<float*>(arena + d.off_obs_others);
    if (is_agent)
        frz::at32(reinterpret_cast<float4*>(obs_self), (uint32_t)agent * Bu + cbu) = make_float4((float)(int)s_table[kAyAt + agent], (float)(int)s_table[kAxAt + agent], base_power, supp);
    wave_lds_sync();
    {
        const int others = A - 1, pairs = A * others, width = d.others_k;
        const bool op = (flags & kObsPower) != 0, os = (flags & kObsSupp) != 0;
        if (!kStepping || os)
            for (int e = 0; e < n_envs; ++e)
                for (int q = lane; q < pairs; q += 64) {  // one lane per (agent, other agent) record
                    const int a = (int)(((uint32_t)q * d.inv_others) >> 16), j = q - a * others, o = j < a ? j : j + 1;
                    float* const rec = obs_others + ((int64_t)a * B + b0 + e) * (others * width) + j * width;
                    const float so = s_supp[e][o], y = (float)(int)s_table[kAyAt + o], x = (float)(int)s_table[kAxAt + o];
                    if (width == 4) {
                        *reinterpret_cast<float4*>(rec) = make_float4(y, x, tablef[o], so);
                    } else {
                        rec[0] = y;
                        rec[1] = x;
                        if (op) rec[2] = tablef[o];
                        if (os) rec[width - 1] = so;
                    }
                }
    }
    if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// wg_lists_kernel: offsets AND lists in one launch, one env per LANE.  A 1024-thread workgroup owns a 256-env chunk: wavefronts 0-3 run
// the launch-wide scan (one env per lane) and park every env's segment starts in LDS; then all 16 wavefronts write lists — wavefront w
// takes the 64 envs of group w & 3 and the items (0 = task rows, 1 = observation map, 2 + a = agent a's lists) congruent to w >> 2
// modulo 4.  The 64 segments of a wavefront's envs are ONE contiguous range of the output (env-major order), so the lanes walk the set
// bits of their env's mask words (left by wg_env_kernel: lit cells, attackable cells per agent, (fires, intensity) of the lit cells in
// task order), stage their entries in an LDS tile at the positions the scan gave them, and the wavefront copies the tile out with
// lane-consecutive 8-byte stores: the cells are not read again, an env costs a lane's few loop iterations instead of a wavefront, and
// the stores are whole lines however many fires burn.
// ------------------------------------------------------------------------------------------------------------------------------------
template <int CPL>
struct ListShared {  // per wavefront: its 64 envs' mask words and segment starts, addressed by env
    static constexpr int PW = CPL < 4 ? 4 : CPL;  // (four 16-bit prefixes are read as one 8-byte word)
    uint64_t lit[64][CPL], sel[64][CPL];
    alignas(8) uint16_t lit_before[64][PW], sel_before[64][PW];  // set bits of the env's words before word k
    int rel[65];
};

// position of the n-th (0-based) set bit of a 64-bit mask (n below its population count)
__device__ __forceinline__ int select_nth64(uint64_t m, int n) {
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    const int c = __popc(lo);
    const bool up0 = n >= c;
    int rem = up0 ? n - c : n, pos = up0 ? 32 : 0;
    uint32_t w = up0 ? hi : lo;
#pragma unroll
    for (int width = 16; width >= 1; width >>= 1) {
        const int cc = __popc(w & ((1u << width) - 1u));
        const bool up = rem >= cc;
        rem -= up ? cc : 0;
        w = up ? w >> width : w;
        pos += up ? width : 0;
    }
    return pos;
}

// One list of one item for the wavefront's 64 envs, one OUTPUT entry per lane: the 64 segments are one contiguous range of dst, so entry
// i of that range belongs to the env e with rel[e] <= i < rel[e + 1] (binary search in LDS) and is the (i - rel[e])-th listed cell of e
// (a population-count descent on its mask words): every lane works whatever the fires per env, and every store instruction writes 64
// consecutive entries.  sel: the cells the lane's env lists (all lanes pass theirs); first: where its segment starts in dst;
// emit(entry index in dst, env slot, position in the env's segment, cell, task index of the cell).
template <int CPL, bool RANK, typename L, typename F>
__device__ __forceinline__ void output_parallel_list(ListShared<CPL>& sh, const uint64_t (&sel)[CPL], int64_t first, int total, int lane, int tile0, int tile_stride,
                                                     L&& fetch, F&& emit) {
    // tiles of 64 entries tile0, tile0 + tile_stride, ... of the wavefront's `total` entries are this wavefront's (the others belong to
    // the wavefronts that serve the same 64 envs)
    const int64_t wave_first = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)first) | ((int64_t)__builtin_amdgcn_readfirstlane((int)(first >> 32)) << 32);
    const int rel = (int)(first - wave_first);
    wave_lds_sync();  // the previous list's readers are done
    {
        int before = 0;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            sh.sel[lane][k] = sel[k];
            sh.sel_before[lane][k] = (uint16_t)before;
            before += (int)__popcll(sel[k]);
        }
    }
    sh.rel[lane] = rel;
    if (lane == 63) sh.rel[64] = total;
    wave_lds_sync();
    const int ntiles = (total + 63) >> 6;
    auto tiles = [&](auto width) {
      constexpr int U = decltype(width)::value;
      for (int tile = tile0; tile < ntiles; tile += U * tile_stride) {
        int e[U], j[U], c[U], rank[U];
        decltype(fetch(0, 0)) got[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = 64 * (tile + u * tile_stride) + lane;
            e[u] = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) e[u] += sh.rel[e[u] + step] <= i ? step : 0;  // the last env that starts at or before i
            j[u] = i - sh.rel[e[u]];
            // the word that holds the env's j-th listed cell (the per-word prefix counts, four to an 8-byte LDS word), then the bit in it
            int kk = 0, skip = 0;
            if (CPL > 1) {
                if (CPL <= 4) {
                    const uint64_t pre = *reinterpret_cast<const uint64_t*>(sh.sel_before[e[u]]);
#pragma unroll
                    for (int k = 1; k < CPL; ++k) {
                        const int p = (int)((pre >> (16 * k)) & 0xFFFFull);
                        const bool past = j[u] >= p;
                        kk = past ? k : kk, skip = past ? p : skip;
                    }
                } else {
#pragma unroll
                    for (int k = 1; k < CPL; ++k) {
                        const int p = sh.sel_before[e[u]][k];
                        const bool past = j[u] >= p;
                        kk = past ? k : kk, skip = past ? p : skip;
                    }
                }
            }
            const bool valid = i < total;
            const int bit = select_nth64(sh.sel[e[u]][kk], j[u] - skip);
            c[u] = valid ? 64 * kk + bit : -1;  // -1 past the range
            if (valid) got[u] = fetch(e[u], j[u]);
            rank[u] = j[u];
            if (RANK) {  // the cell's rank among the env's lit cells
                const uint64_t v = sh.lit[e[u]][kk];
                rank[u] = (int)sh.lit_before[e[u]][kk] + (int)__popcll(v & ((1ull << bit) - 1ull));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c[u] >= 0) emit(wave_first + 64 * (tile + u * tile_stride) + lane, j[u], c[u], rank[u], got[u]);
      }
    };
    // four of the wavefront's tiles in flight when it has more than one (what an entry fetches from memory is requested for all four
    // before the first is used); a sparse step has one tile per list
    if (tile0 + tile_stride < ntiles)
        tiles(std::integral_constant<int, 2>{});
    else
        tiles(std::integral_constant<int, 1>{});
}

// SPLIT (round 4: the lists of step t beside the env launch of step t + 1, see launch_cpl): 0 = offsets and lists in one launch (above);
// 1 = the scan alone, on the step's stream — 256 threads per chunk: offsets, batch totals, epoch, what the NEXT env launch waits for;
// 2 = the lists alone, on a second stream: the segment starts come back from the offsets arrays the scan launch wrote.
template <int AMAX, int CPL, int BITS, int SPLIT = 0>
__global__ void __launch_bounds__(64 * (CPL >= 8 ? 4 : 16)) wg_lists_kernel(char* __restrict__ arena, const WgDev d, uint32_t ticketed) {
    constexpr int kListWaves = CPL >= 8 ? 4 : 16, kListParts = kListWaves / frz::kWaves;
    constexpr int kChannels = AMAX + 1;
    constexpr int kItems = (AMAX + 2 + kListParts - 1) / kListParts;  // items per wavefront
    __shared__ frz::ScanShared<kChannels, BITS> s_scan;
    __shared__ int s_ticket;
    __shared__ uint32_t s_first[kChannels][kBlock + 1];  // exclusive prefix of channel ch at the chunk's env e; [kBlock]: at the chunk's end
    __shared__ ListShared<CPL> s_list[kListWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, group = wave & (frz::kWaves - 1), part = wave >> 2;
    const int64_t B = d.B;
    const int A = d.A, HW = d.HW, Wd = d.W;
    frz::ScanWorkspace ws{reinterpret_cast<uint32_t*>(arena + d.off_epoch), reinterpret_cast<uint32_t*>(arena + d.off_totals),
                          reinterpret_cast<uint64_t*>(arena + d.off_agg), reinterpret_cast<uint64_t*>(arena + d.off_prefix)};
    frz::ScanLaunch launch{};
    int chunk = (int)blockIdx.x;
    if constexpr (SPLIT != 2) {
        launch = frz::scan_begin(ws);
        chunk = frz::scan_take_chunk(ws, d.nchunks, ticketed != 0, &s_ticket);
    }
    const int el = group * 64 + lane;  // the lane's env within the chunk
    const int64_t b = (int64_t)chunk * kBlock + el;
    const bool active = b < B;
    const int64_t bl = active ? b : B - 1;
    const int32_t* rows = reinterpret_cast<const int32_t*>(arena + d.off_rows4);
    const int64_t* rows8 = reinterpret_cast<const int64_t*>(arena + d.off_rows8);
    const uint8_t* rows1 = reinterpret_cast<const uint8_t*>(arena + d.off_rows1);

    // ---------------------------------------------------------------- loads of the list phase, in flight while the scan runs
    const uint64_t* const litmap = reinterpret_cast<const uint64_t*>(arena + d.off_litmap);
    const uint64_t* const okmap = reinterpret_cast<const uint64_t*>(arena + d.off_okmap);
    uint64_t lit[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) lit[k] = (active && SPLIT != 1) ? litmap[(int64_t)k * B + bl] : 0ull;
    ListShared<CPL>& sh = s_list[wave];
    {
        int before = 0;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            sh.lit[lane][k] = lit[k];
            sh.lit_before[lane][k] = (uint16_t)before;
            before += (int)__popcll(lit[k]);
        }
    }

    // ---------------------------------------------------------------- the scan (wavefronts 0-3)
    uint32_t err = 0;
    if constexpr (SPLIT == 2) {  // the scan ran as a launch of its own: every env's segment starts are in the offsets arrays
        if (part == 0) {
            const int64_t* const task_offsets = reinterpret_cast<const int64_t*>(arena + d.off_task_offsets);
            const int64_t* const act_offsets = reinterpret_cast<const int64_t*>(arena + d.off_act_offsets);
            const int64_t at = b < B ? b : B, chunk_end = (int64_t)(chunk + 1) * kBlock < B ? (int64_t)(chunk + 1) * kBlock : B;
            s_first[0][el] = (uint32_t)task_offsets[at];
            if (el == kBlock - 1) s_first[0][kBlock] = (uint32_t)task_offsets[chunk_end];
#pragma unroll
            for (int a = 0; a < AMAX; ++a)
                if (a < A) {
                    s_first[a + 1][el] = (uint32_t)act_offsets[a * (B + 1) + at];
                    if (el == kBlock - 1) s_first[a + 1][kBlock] = (uint32_t)act_offsets[a * (B + 1) + chunk_end];
                }
        }
    } else if (part == 0) {
        uint32_t cnt[kChannels], excl[kChannels];
        cnt[0] = active ? (uint32_t)rows8[(int64_t)d.q_etc * B + bl] : 0u;
#pragma unroll
        for (int a = 0; a < AMAX; ++a) cnt[a + 1] = (active && a < A) ? (uint32_t)rows[(int64_t)(d.r_atc + (a < A ? a : 0)) * B + bl] : 0u;
        const bool term = rows1[(int64_t)d.u_term * B + bl] != 0, trunc = rows1[(int64_t)d.u_trunc * B + bl] != 0;
        frz::scan_chunk<kChannels, BITS>(s_scan, ws, launch, cnt, active && !term, active && !trunc, A + 1, chunk, d.nchunks, excl, &err);
#pragma unroll
        for (int ch = 0; ch < kChannels; ++ch) {
            s_first[ch][el] = excl[ch];
            if (el == kBlock - 1) s_first[ch][kBlock] = excl[ch] + cnt[ch];
        }
        if (active) {
            int64_t* const task_offsets = reinterpret_cast<int64_t*>(arena + d.off_task_offsets);
            int64_t* const act_offsets = reinterpret_cast<int64_t*>(arena + d.off_act_offsets);
            int64_t* const bad_offsets = reinterpret_cast<int64_t*>(arena + d.off_bad_offsets);
            const bool show_bad = (d.flags & kShowBad) != 0;
            task_offsets[b] = excl[0];
            if (b == B - 1) task_offsets[B] = (int64_t)excl[0] + cnt[0];
#pragma unroll
            for (int a = 0; a < AMAX; ++a)
                if (a < A) {
                    act_offsets[a * (B + 1) + b] = excl[a + 1];
                    if (b == B - 1) act_offsets[a * (B + 1) + B] = (int64_t)excl[a + 1] + cnt[a + 1];
                    if (show_bad) {  // bad = listed but not attackable
                        bad_offsets[a * (B + 1) + b] = (int64_t)excl[0] - excl[a + 1];
                        if (b == B - 1) bad_offsets[a * (B + 1) + B] = ((int64_t)excl[0] + cnt[0]) - ((int64_t)excl[a + 1] + cnt[a + 1]);
                    }
                }
        }
    } else {
        frz::scan_chunk_passive_front();
        frz::scan_chunk_passive_back();
    }
    __syncthreads();
    if constexpr (SPLIT != 2) {
        if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);
        frz::scan_end(ws, launch, chunk, d.nchunks);
    }
    if constexpr (SPLIT == 1) return;  // (the lists: a launch of their own, on the second stream)
    if constexpr (SPLIT == 2) {  // a step that changed nothing (every env finished): its parity's mask words are two steps old, the lists stay
        if (reinterpret_cast<const uint32_t*>(arena + d.off_epoch)[40 + (d.parity & 1)] != 0u) return;
    }

    // ---------------------------------------------------------------- the lists (wildfire.py:586-717)
    const int64_t cap = B * (int64_t)HW;
    const int64_t task_first = s_first[0][el];
    const bool show_bad = (d.flags & kShowBad) != 0;
    const int64_t wave_env0 = (int64_t)chunk * kBlock + group * 64;  // env of lane 0
    // Every wavefront of a group walks ALL the items; tile t of item m is written by the wavefront with part == (t + m) mod parts: a
    // sparse step has one tile per list and the items spread over the parts as (m mod parts); in the first steps of an episode on a
    // large grid a list has tens of tiles and the parts share them (one wavefront writing all the task rows held the launch up).
    const int ends = group * 64 + 64;  // the chunk's env behind this wavefront's last one
    for (int j = 0; j < kItems; ++j) {
        for (int q = 0; q < kListParts; ++q) {
            const int item = q + kListParts * j;
            if (item >= A + 2) break;
            const int tile0 = (part - item) & (kListParts - 1);
            const int task_total = (int)(s_first[0][ends] - s_first[0][group * 64]);
            if (item <= 1) {
                if (64 * tile0 >= task_total) continue;
                if (item == 0) {
                    // task rows (y, x, fires level, intensity) of the lit fires in row-major order
                    longlong2* const task_values = reinterpret_cast<longlong2*>(arena + d.off_task_values);
                    const int2* const lit_cells = reinterpret_cast<const int2*>(arena + d.off_lit_cells);
                    const int64_t stride = (HW + 1) & ~1;
                    output_parallel_list<CPL, false>(
                        sh, lit, task_first, task_total, lane, tile0, kListParts, [&](int e, int idx) { return lit_cells[(wave_env0 + e) * stride + idx]; },
                        [&](int64_t at, int, int c, int, int2 cell) {
                            const int y = (int)(((uint32_t)c * d.inv_w) >> 16), x = c - y * Wd;
                            task_values[2 * at] = make_longlong2(y, x);
                            task_values[2 * at + 1] = make_longlong2(cell.x, cell.y);
                        });
                } else {
                    // the observation map: task j of the env observes task j
                    int64_t* const obs_map = reinterpret_cast<int64_t*>(arena + d.off_obs_map);
                    output_parallel_list<CPL, false>(sh, lit, task_first, task_total, lane, tile0, kListParts, [](int, int) { return 0; },
                                                     [&](int64_t at, int idx, int, int, int) { obs_map[at] = idx; });
                }
                continue;
            }
            // agent a: the task indices of its attackable fires (and, with show_bad_actions, of the listed-but-not-attackable ones)
            const int a = item - 2;
            const int ok_total = (int)(s_first[a + 1][ends] - s_first[a + 1][group * 64]), bad_total = show_bad ? task_total - ok_total : 0;
            if (64 * tile0 >= ok_total && 64 * tile0 >= bad_total) continue;
            // (the mask words are loaded here, per item: holding every item's words from before the scan cost registers the 1024-thread
            // workgroup does not have — it spilled — and the launch was 3 us slower for it)
            uint64_t mine[CPL];
#pragma unroll
            for (int k = 0; k < CPL; ++k) mine[k] = active ? okmap[((int64_t)a * CPL + k) * B + bl] : 0ull;
            const int64_t first = s_first[a + 1][el];
            int64_t* const act_values = reinterpret_cast<int64_t*>(arena + d.off_act_values) + (int64_t)a * cap;
            if (64 * tile0 < ok_total)
                output_parallel_list<CPL, true>(sh, mine, first, ok_total, lane, tile0, kListParts, [](int, int) { return 0; },
                                                [&](int64_t at, int, int, int rank, int) { act_values[at] = rank; });
            if (64 * tile0 < bad_total) {
                uint64_t bad[CPL];
#pragma unroll
                for (int k = 0; k < CPL; ++k) bad[k] = lit[k] & ~mine[k];
                int64_t* const bad_values = reinterpret_cast<int64_t*>(arena + d.off_bad_values) + (int64_t)a * cap;
                output_parallel_list<CPL, true>(sh, bad, task_first - first, bad_total, lane, tile0, kListParts, [](int, int) { return 0; },
                                                [&](int64_t at, int, int, int rank, int) { bad_values[at] = rank; });
            }
        }
    }
}

template <int CPL>
int launch_cpl(const WgDev& dev_in, char* arena, const WfArgs& args, int rng, int mode, bool ticketed, hipStream_t stream, const WgOverlap* overlap) {
    // Overlapped steps of a rollout (round 4): the env launch of step t + 1 needs nothing of step t's lists but the batch
    // totals and the epoch, which the SCAN produces — so the scan runs as a small launch of its own on the step's stream, and the lists
    // (13-31 us of mostly stores) go to a second stream, beside the next env launch.  What the lists launch reads of the env launch (the
    // lit / attackable mask words, the lit cells' (fires, intensity)) is double-buffered by the step's parity; the offsets arrays it also
    // reads are rewritten by the NEXT scan, which therefore waits for this step's lists first (they are long done: an env launch lies between).
    const bool split = overlap != nullptr && overlap->side != nullptr && mode == kStep;
    WgDev dev = dev_in;
    dev.parity = 0;
    if (split && overlap->parity) dev.off_litmap += overlap->copy_delta, dev.off_okmap += overlap->copy_delta, dev.off_lit_cells += overlap->copy_delta, dev.parity = 1;
    const dim3 waves((unsigned)((dev.B + kEnvsPerBlock - 1) / kEnvsPerBlock)), lanes((unsigned)dev.nchunks), block(kBlock);
    WgPolicy policy{args.policy ? 1u : 0u, (uint32_t)args.policy_seed, (uint32_t)(args.policy_seed >> 32), (uint32_t)args.policy_step,
                    (uint32_t)(args.policy_step >> 32), args.actions_out};
    auto go = [&](auto kernel) {
        if (args.start_event)
            hipExtLaunchKernelGGL(kernel, waves, block, 0, stream, args.start_event, nullptr, 0, arena, dev, args.actions, args.field_rand, args.agent_rand, policy);
        else
            hipLaunchKernelGGL(kernel, waves, block, 0, stream, arena, dev, args.actions, args.field_rand, args.agent_rand, policy);
    };
    if (mode == kReset) {
        policy = WgPolicy{args.seed_increment != 0 ? 1u : 0u, (uint32_t)args.seed_increment, 0u, 0u, 0u, nullptr};  // seed increment rides in the policy block
        go(wg_env_kernel<CPL, kReset, FRZ_RNG_INJECTED>);
    } else if (mode == kRebuild) {
        go(wg_env_kernel<CPL, kRebuild, FRZ_RNG_INJECTED>);
    } else if (rng == FRZ_RNG_PHILOX) {
        if (dev.HW == 64 * CPL)
            go(wg_env_kernel<CPL, kStep, FRZ_RNG_PHILOX, true>);
        else
            go(wg_env_kernel<CPL, kStep, FRZ_RNG_PHILOX>);
    } else {
        if (dev.HW == 64 * CPL)
            go(wg_env_kernel<CPL, kStep, FRZ_RNG_INJECTED, true>);
        else
            go(wg_env_kernel<CPL, kStep, FRZ_RNG_INJECTED>);
    }
    {
        const uint32_t tk = ticketed ? 1u : 0u;
        const dim3 list_block(64 * (CPL >= 8 ? 4 : 16));
        auto lists = [&](auto kernel) {
            if (args.stop_event)
                hipExtLaunchKernelGGL(kernel, lanes, list_block, 0, stream, nullptr, args.stop_event, 0, arena, dev, tk);
            else
                hipLaunchKernelGGL(kernel, lanes, list_block, 0, stream, arena, dev, tk);
        };
        // the split pair: the scan on the step's stream (after the previous step's lists, whose offsets it rewrites), the lists on the second one
        auto split_pair = [&](auto scan_kernel, auto emit_kernel) {
            if (overlap->wait_previous_lists) (void)hipStreamWaitEvent(stream, overlap->lists_done, 0);
            hipLaunchKernelGGL(scan_kernel, lanes, dim3(kBlock), 0, stream, arena, dev, tk);
            (void)hipEventRecord(overlap->scan_done, stream);
            (void)hipStreamWaitEvent(overlap->side, overlap->scan_done, 0);
            hipLaunchKernelGGL(emit_kernel, lanes, list_block, 0, overlap->side, arena, dev, tk);
            (void)hipEventRecord(overlap->lists_done, overlap->side);
        };
        constexpr int BITS = CPL <= 2 ? 16 : 32;  // the scan packs 16-bit counts while an env has at most 255 cells
        // (AMAX 12: with the 17 scan channels of AMAX 16 the scan's registers do not fit the 128 a 1024-thread workgroup leaves a
        // wavefront, and the kernel spills)
        auto with = [&](auto amax, auto bits) {
            constexpr int AM = decltype(amax)::value, BT = decltype(bits)::value;
            if (split) split_pair(wg_lists_kernel<AM, CPL, BT, 1>, wg_lists_kernel<AM, CPL, BT, 2>);
            else lists(wg_lists_kernel<AM, CPL, BT, 0>);
        };
        auto by_agents = [&](auto bits) {
            if (dev.A <= 4) with(std::integral_constant<int, 4>{}, bits);
            else if (dev.A <= 8) with(std::integral_constant<int, 8>{}, bits);
            else if (dev.A <= 12) with(std::integral_constant<int, 12>{}, bits);
            else with(std::integral_constant<int, 16>{}, bits);
        };
        if (CPL == 4 && dev.HW < 256) by_agents(std::integral_constant<int, 16>{});
        else by_agents(std::integral_constant<int, BITS>{});
    }
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

}  // namespace

// rng: FRZ_RNG_PHILOX (drawn in the launch) or FRZ_RNG_INJECTED (args.field_rand / agent_rand; MT19937 draws are staged by the caller)
int launch_grid(const WgDev& dev, char* arena, const WfArgs& args, int rng, int mode, bool ticketed, hipStream_t stream, const WgOverlap* overlap) {
    const int cpl = (dev.HW + 63) / 64;
    if (cpl <= 1) return launch_cpl<1>(dev, arena, args, rng, mode, ticketed, stream, overlap);
    if (cpl <= 2) return launch_cpl<2>(dev, arena, args, rng, mode, ticketed, stream, overlap);
    if (cpl <= 4) return launch_cpl<4>(dev, arena, args, rng, mode, ticketed, stream, overlap);
    if (cpl <= 8) return launch_cpl<8>(dev, arena, args, rng, mode, ticketed, stream, overlap);
    if (cpl <= 16) return launch_cpl<16>(dev, arena, args, rng, mode, ticketed, stream, overlap);
    return FRZ_E_INVALID;
}

}  // namespace frz_wf
