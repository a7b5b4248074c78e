// wildfire.hip — fused wildfire environment step for gfx950 (MI355X), one environment per lane.
//
// One launch = one ParallelEnv.step() of the reference for the whole batch:
//   action decode through the open action mapping (wildfire.py:427-483) -> 7 transitions (transitions/*.py)
//   -> rewards / termination (wildfire.py:534-582) -> AEC bookkeeping (utils/env.py:228-235)
//   -> update_observations + update_actions (wildfire.py:586-717) with the variable-length task lists compacted by
//      wavefront prefix sums and a single-pass inter-workgroup prefix hand-off.
//
// Memory: ONE device arena per env object.  Every per-env 4-byte array is a row of a [rows][B] block (env index
// innermost), so a wavefront touches one contiguous 256-byte segment per field and the kernel needs a single base
// pointer; the configuration lives in a device block read through the scalar cache; per-lane-indexed tables
// (in-range cell sets, equipment bonuses) sit in LDS.  A lane keeps its env (grid cells, agents) in registers; the lit
// set and the per-agent attackable sets are cell bitmasks.  Nothing is re-read: the algorithmic bytes per env-step
//  are each touched once.
//
// HBM-bound integer/byte work: no MFMA.  Built with -ffp-contract=off (float32 ops round once, like eager torch).
#include "frz_device.h"

#include "../../include/frz.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "frz_scan.h"
#include "wildfire_common.h"

namespace {

using namespace frz_wf;

// wildfire.py:347-354 + utils/env.py:137-160: state from the configuration, bookkeeping zeroed.
__global__ void __launch_bounds__(kBlock) wf_fill_kernel(char* arena, int32_t seed_increment) {
    const WfDev& d = *reinterpret_cast<const WfDev*>(arena);
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t B = d.B;
    if (b >= B) return;
    int32_t* rows = reinterpret_cast<int32_t*>(arena + d.off_rows4);
    // frz_wildfire_reset_reseed: fresh env seeds per episode — modulo 2^32 (a graph replayed for days wraps around; signed overflow would be UB)
    if (seed_increment != 0) reinterpret_cast<uint32_t*>(rows)[d.r_seeds * B + b] += (uint32_t)seed_increment;
    float* rowsf = reinterpret_cast<float*>(arena + d.off_rows4);
    int64_t* rows8 = reinterpret_cast<int64_t*>(arena + d.off_rows8);
    uint8_t* rows1 = reinterpret_cast<uint8_t*>(arena + d.off_rows1);
    for (int c = 0; c < d.HW; ++c) {
        const int type = d.fire_types[c];
        const int f = d.lit[c] ? type : -type;
        rows[(d.r_fires + c) * B + b] = f;
        rows[(d.r_intensity + c) * B + b] = d.lit[c] ? d.ignition[c] : 0;
        rows[(d.r_fuel + c) * B + b] = f != 0 ? d.initial_fuel : 0;
    }
    for (int a = 0; a < d.A; ++a) {
        rowsf[(d.r_supp + a) * B + b] = d.initial_suppressant;
        rowsf[(d.r_cap + a) * B + b] = d.initial_capacity;
        rows[(d.r_equip + a) * B + b] = d.initial_equipment;
        rowsf[(d.r_rewards + a) * B + b] = 0.0f;
        rowsf[(d.r_cum + a) * B + b] = 0.0f;
        rows1[(d.u_term + a) * B + b] = 0;
        rows1[(d.u_trunc + a) * B + b] = 0;
    }
    rows[d.r_moves * B + b] = 0;
    rows[d.r_burnouts * B + b] = 0;
    rows8[d.q_burnouts * B + b] = 0;
    rows8[d.q_putouts * B + b] = 0;
    rows1[d.u_frozen * B + b] = 0;
}

// frz_wildfire_reset_masked: wf_fill_kernel on the envs a device-side mask selects (mask == nullptr: the finished ones — all agents
// terminated or all truncated; agents share one value per env, row 0 is read) — the state part of reset_batches (utils/env.py:162-189,
// wildfire.py:376-397); the rebuild launch that follows refreshes observations and lists.  cells_env_major: the grid family's layout.
__global__ void __launch_bounds__(kBlock) wf_masked_fill_kernel(char* arena, const uint8_t* mask, uint32_t seed_increment, int64_t off_cells,
                                                                int64_t off_cell_tables, frz_wildfire_saved_state saved) {
    const WfDev& d = *reinterpret_cast<const WfDev*>(arena);
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t B = d.B;
    if (b >= B) return;
    int32_t* rows = reinterpret_cast<int32_t*>(arena + d.off_rows4);
    float* rowsf = reinterpret_cast<float*>(arena + d.off_rows4);
    uint8_t* rows1 = reinterpret_cast<uint8_t*>(arena + d.off_rows1);
    const bool selected = mask ? mask[b] != 0 : (rows1[d.u_term * B + b] != 0 || rows1[d.u_trunc * B + b] != 0);
    if (!selected) return;
    reinterpret_cast<uint32_t*>(rows)[d.r_seeds * B + b] += seed_increment;  // modulo 2^32
    if (saved.fires != nullptr) {  // the state saved at reset (frz_wildfire_set_saved_initial: a caller-given initial state)
        const int64_t HW = d.HW;
        int32_t* cells = reinterpret_cast<int32_t*>(arena + off_cells);
        for (int64_t c = 0; c < HW; ++c) {
            const int32_t f = saved.fires[b * saved.fires_stride_env + c * saved.fires_stride_item];
            const int32_t i = saved.intensity[b * saved.intensity_stride_env + c * saved.intensity_stride_item];
            const int32_t u = saved.fuel[b * saved.fuel_stride_env + c * saved.fuel_stride_item];
            if (off_cells != 0) {
                cells[(0 * B + b) * HW + c] = f, cells[(1 * B + b) * HW + c] = i, cells[(2 * B + b) * HW + c] = u;
            } else {
                rows[(d.r_fires + c) * B + b] = f, rows[(d.r_intensity + c) * B + b] = i, rows[(d.r_fuel + c) * B + b] = u;
            }
        }
    } else if (off_cells != 0) {  // grid family: cells env-major [3][B][HW]; initial fires / intensity / fuel are tables 2..4 of the cell-table block
        const int32_t* tables = reinterpret_cast<const int32_t*>(arena + off_cell_tables);
        int32_t* cells = reinterpret_cast<int32_t*>(arena + off_cells);
        const int64_t HW = d.HW;
        for (int k = 0; k < 3; ++k)
            for (int64_t c = 0; c < HW; ++c) cells[((int64_t)k * B + b) * HW + c] = tables[(2 + k) * HW + c];
    } else {
        for (int c = 0; c < d.HW; ++c) {
            const int type = d.fire_types[c];
            const int f = d.lit[c] ? type : -type;
            rows[(d.r_fires + c) * B + b] = f;
            rows[(d.r_intensity + c) * B + b] = d.lit[c] ? d.ignition[c] : 0;
            rows[(d.r_fuel + c) * B + b] = f != 0 ? d.initial_fuel : 0;
        }
    }
    for (int a = 0; a < d.A; ++a) {
        const bool s = saved.fires != nullptr;
        rowsf[(d.r_supp + a) * B + b] = s ? saved.suppressants[b * saved.suppressants_stride_env + a * saved.suppressants_stride_item] : d.initial_suppressant;
        rowsf[(d.r_cap + a) * B + b] = s ? saved.capacity[b * saved.capacity_stride_env + a * saved.capacity_stride_item] : d.initial_capacity;
        rows[(d.r_equip + a) * B + b] = s ? saved.equipment[b * saved.equipment_stride_env + a * saved.equipment_stride_item] : d.initial_equipment;
        rowsf[(d.r_rewards + a) * B + b] = 0.0f;
        rowsf[(d.r_cum + a) * B + b] = 0.0f;
        rows1[(d.u_term + a) * B + b] = 0;
        rows1[(d.u_trunc + a) * B + b] = 0;
    }
    rows[d.r_moves * B + b] = 0;
    rows[d.r_burnouts * B + b] = 0;
    rows1[d.u_frozen * B + b] = 0;
}

// EXACT: the grid has exactly CMAX cells and AMAX agents (every loop bound is a compile-time constant).
// Diagnostic build only (-DFRZ_WF_STAMPS, tools/stamps.py): thread 0 of workgroup 0 records the shader clock at phase
// boundaries into a buffer nothing else reads.  No stamp executes in the production library.
#ifdef FRZ_WF_STAMPS
#define FRZ_STAMP(i)                                                                                          \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (blockIdx.x == 0 && threadIdx.x == 0 && MODE == kStep)                                             \
            reinterpret_cast<unsigned long long*>(arena + d.off_rand_agent)[i] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
#else
#define FRZ_STAMP(i) \
    do {             \
    } while (0)
#endif

// One 256-env chunk per loop iteration.  Order of one iteration (what is stored where matters as much as the math: a
// CU drains ~20-35 B/clk of stores, ~half that for record-strided ones (tools/ubench/store_rate.hip), so a 2x3 env-step
// has ~8 K cycles of store traffic per workgroup; every store group is issued as soon as its values are final so that
// the write path drains behind the arithmetic that follows instead of after it):
//   loads (issued before the configuration is staged; the next chunk's are issued before the jagged tail)
//   -> randomness -> action decode -> agent transitions            => agent rows + agent observations stored
//   -> fire increase/decrease -> spread (per cell)                  => intensity / fuel rows stored cell by cell
//   -> dead-env test                                                => fires rows stored
//   -> open-task masks, per-env counts, wavefront + workgroup scan, chunk sums published
//   -> rewards / termination / bookkeeping (hides the hand-off)     => reward / counter rows stored
//   -> look-back over the preceding chunks -> jagged task / action lists.
// Lanes past the batch end (last chunk only) shadow env B-1: they compute and store the same values as its owner (all
// dense stores are pure functions of the loaded state), contribute nothing to the scan and write no list entries.
template <int CMAX, int AMAX, bool EXACT, int RNG, int MODE>
__global__ void __launch_bounds__(kBlock) wf_step_kernel(char* __restrict__ arena, const WfDev* __restrict__ dev,
                                                          const int32_t* __restrict__ actions, const float* __restrict__ field_rand,
                                                          const float* __restrict__ agent_rand, const WfLaunch launch) {
    const int32_t batch = launch.batch;
    using mask_t = std::conditional_t<(CMAX <= 32), uint32_t, uint64_t>;
    constexpr int PW = (AMAX + 1 + 3) / 4;                                // packed scan words (four 16-bit channels each)
    constexpr int NCHP = AMAX + 3 <= 8 ? 8 : (AMAX + 3 <= 16 ? 16 : 32);  // scan channels padded to a power of two

    __shared__ uint64_t s_wave_scan[frz::kWaves][PW];
    __shared__ uint32_t s_wave_live[frz::kWaves][2];
    __shared__ uint32_t s_reduce[frz::kWaves][32];
    __shared__ uint32_t s_prefix[32];
    __shared__ WfStaged s_cfg;  // hot scalars (copied to registers below) + the per-lane lookup tables (range sets, equipment, capacities)

    const int tid = threadIdx.x, lane = frz::lane_id(), wave = frz::wave_id();
    const uint4 cfg_piece = stage_request(dev);  // first vector-memory instruction of the kernel
    const int64_t B = batch;
    const uint32_t Bu = (uint32_t)batch;
    const int nchunks = (int)((B + kBlock - 1) / kBlock);
    const int HW = EXACT ? CMAX : dev->HW, A = EXACT ? AMAX : dev->A;
    // rows of the [rows][B] block: a fixed function of (HW, A) (create() lays them out in this order), and the block
    // starts right after the configuration block: the state loads need nothing but the kernel arguments
    const int r_fires = 0, r_intensity = HW, r_fuel = 2 * HW, r_supp = 3 * HW, r_cap = r_supp + A, r_equip = r_cap + A;
    const int r_moves = r_equip + A, r_burnouts = r_moves + 1, r_rewards = r_moves + 2, r_cum = r_rewards + A, r_atc = r_cum + A;
    const int r_seeds = r_atc + A;
    int32_t* const rows = reinterpret_cast<int32_t*>(arena + kDevBlockBytes);
    float* const rowsf = reinterpret_cast<float*>(arena + kDevBlockBytes);
    uint8_t* const rows1 = reinterpret_cast<uint8_t*>(arena + launch.off_rows1);
    const uint32_t u_term = 0, u_trunc = (uint32_t)A, u_frozen = 2u * (uint32_t)A;

    // plain (cacheable, wave-uniform) loads: the words were last written by the previous launch, and this launch only
    // rewrites the epoch after every workgroup has read it.  Both totals slots are fetched beside the epoch (no dependent
    // load) and the previous launch's slot is selected afterwards.
    uint32_t* const epoch_ptr = reinterpret_cast<uint32_t*>(arena + launch.off_epoch);
    uint32_t* const totals = reinterpret_cast<uint32_t*>(arena + launch.off_totals);
    const uint32_t epoch = *epoch_ptr;
    uint32_t totals0[AMAX + 3], totals1[AMAX + 3];
#pragma unroll
    for (int i = 0; i < AMAX + 3; ++i) totals0[i] = totals[i], totals1[i] = totals[kTotalsStride + i];

    struct Env {
        int f[CMAX], in[CMAX], fu[CMAX], eqs[AMAX], act_idx[AMAX], act_id[AMAX], nm, nb;
        float supp[AMAX], capa[AMAX], cum[AMAX];
        uint32_t seed, term, trunc;
    };
    struct Injected {
        float r_field[3][CMAX], r_agent[5][AMAX];
    };
    struct Nothing {};
    using Draws = std::conditional_t<(RNG == FRZ_RNG_INJECTED && MODE == kStep), Injected, Nothing>;
    auto load_env = [&](int chunk, Draws& draws) {
        Env e;
        const int64_t b = (int64_t)chunk * kBlock + tid;
        const uint32_t bl = (uint32_t)(b < B ? b : B - 1);
        // Unconditional loads (rows past the env's cell / agent count are read from the last valid row and replaced below): a load
        // under `if (a < A)` with a runtime A is a basic block of its own that ends in a wait for every load issued so far.
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            const int cc = EXACT ? c : min(c, HW - 1);
            e.f[c] = at32(rows, (uint32_t)(r_fires + cc) * Bu + bl);
            e.in[c] = at32(rows, (uint32_t)(r_intensity + cc) * Bu + bl);
            e.fu[c] = at32(rows, (uint32_t)(r_fuel + cc) * Bu + bl);
        }
        int2 pair[AMAX];
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            const int aa = EXACT ? a : min(a, A - 1);
            e.supp[a] = at32(rowsf, (uint32_t)(r_supp + aa) * Bu + bl);
            e.capa[a] = at32(rowsf, (uint32_t)(r_cap + aa) * Bu + bl);
            e.eqs[a] = at32(rows, (uint32_t)(r_equip + aa) * Bu + bl);
            e.cum[a] = 0.0f;
            pair[a] = make_int2(0, -1);
            if (MODE == kStep) {
                pair[a] = reinterpret_cast<const int2*>(actions)[(int64_t)aa * B + bl];
                e.cum[a] = at32(rowsf, (uint32_t)(r_cum + aa) * Bu + bl);
            }
        }
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c >= HW) e.f[c] = e.in[c] = e.fu[c] = 0;
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            const bool real = a < A;
            e.supp[a] = real ? e.supp[a] : 0.0f;
            e.capa[a] = real ? e.capa[a] : 0.0f;
            e.cum[a] = real ? e.cum[a] : 0.0f;
            e.eqs[a] = real ? e.eqs[a] : 0;
            e.act_idx[a] = real ? pair[a].x : 0;
            e.act_id[a] = real ? pair[a].y : -1;
        }
        // agents share one termination / truncation value (wildfire.py:579, utils/env.py:231-233): row 0 is read,
        // all A rows are written
        e.term = at32(rows1, u_term * Bu + bl);
        e.trunc = at32(rows1, u_trunc * Bu + bl);
        e.nm = e.nb = 0;
        e.seed = 0;
        if (MODE == kStep) {
            e.nm = at32(rows, (uint32_t)r_moves * Bu + bl);
            e.nb = at32(rows, (uint32_t)r_burnouts * Bu + bl);
            if (RNG == FRZ_RNG_PHILOX) e.seed = (uint32_t)at32(rows, (uint32_t)r_seeds * Bu + bl);
        }
        if constexpr (RNG == FRZ_RNG_INJECTED && MODE == kStep) {
#pragma unroll
            for (int ev = 0; ev < 3; ++ev)
#pragma unroll
                for (int c = 0; c < CMAX; ++c) draws.r_field[ev][c] = c < HW ? field_rand[((int64_t)ev * B + bl) * HW + c] : 1.0f;
#pragma unroll
            for (int ev = 0; ev < 5; ++ev)
#pragma unroll
                for (int a = 0; a < AMAX; ++a) draws.r_agent[ev][a] = a < A ? agent_rand[((int64_t)ev * B + bl) * A + a] : 1.0f;
        }
        // A shadow lane reads rows its env's owner (another lane of this workgroup) stores later in the iteration: in
        // the one chunk that has shadow lanes the loads complete before the workgroup barriers that precede those stores.
        if (chunk == nchunks - 1 && (B % kBlock) != 0) __builtin_amdgcn_s_waitcnt(0);
        return e;
    };

    // One chunk per workgroup (no chunk loop: wildfire_roles.hip explains what the loop cost); chunk = blockIdx.x while
    // the grid is co-resident, otherwise chunks are handed out in arrival order so that the hand-off never waits on a
    // workgroup that has not started.
    __shared__ int s_ticket;
    int chunk = blockIdx.x;
    if (launch.ticketed) {
        uint32_t* const counter = reinterpret_cast<uint32_t*>(arena + launch.off_epoch) + 32;
        if (tid == 0) {
            const uint32_t t = atomicAdd(counter, 1u);
            if (t == (uint32_t)nchunks - 1u) atomicExch(counter, 0u);  // every ticket of this launch is out
            s_ticket = (int)t;
        }
        __syncthreads();
        chunk = s_ticket;
    }
    Draws cur_draws;
    Env cur = load_env(chunk, cur_draws);  // in flight while the configuration is staged

    const WfHot d = stage_commit(s_cfg, cfg_piece);  // configuration block at arena offset 0; never written by a kernel
    FRZ_STAMP(0);
    const int W = d.W;
    const int nch = d.nch;  // A + 3
    const int ch_nt = A + 1, ch_ntr = A + 2;
    const uint32_t flags = d.flags;

    const uint32_t tag = epoch + 1u;  // never 0 on a zero-filled arena
    uint32_t* cur_totals = totals + (epoch & 1u) * kTotalsStride;
    uint32_t prev[AMAX + 3];
#pragma unroll
    for (int i = 0; i < AMAX + 3; ++i) prev[i] = (epoch & 1u) ? totals0[i] : totals1[i];

    int64_t* const rows8 = reinterpret_cast<int64_t*>(arena + d.off_rows8);
    const uint32_t q_burnouts = 0, q_putouts = 1, q_etc = 2;

    // utils/env.py:211-213 — every per-agent step() is a no-op once ALL envs are terminated or ALL are truncated.
    if (MODE == kStep) {
        uint32_t nt = prev[0], ntr = prev[0];
#pragma unroll
        for (int i = 0; i < AMAX + 3; ++i) {
            nt = i == ch_nt ? prev[i] : nt;
            ntr = i == ch_ntr ? prev[i] : ntr;
        }
        if (nt == 0u || ntr == 0u) {
            // The parallel adapter (utils/conversions.py:87-90) then adds the stale aec rewards once per agent call.
            {
                const int64_t b = (int64_t)chunk * kBlock + tid;
                if (b < B && !at32(rows1, u_frozen * Bu + (uint32_t)b)) {
                    for (int a = 0; a < A; ++a) {
                        const float r = at32(rowsf, (uint32_t)(r_rewards + a) * Bu + (uint32_t)b);
                        float acc = 0.0f;
                        for (int j = 0; j < A; ++j) acc = acc + r;
                        at32(rowsf, (uint32_t)(r_rewards + a) * Bu + (uint32_t)b) = acc;
                    }
                    at32(rows1, u_frozen * Bu + (uint32_t)b) = 1;
                }
            }
            return;
        }
    }

    float* const obs_self = reinterpret_cast<float*>(arena + d.off_obs_self);
    float* const obs_others = reinterpret_cast<float*>(arena + d.off_obs_others);
    uint64_t* const agg = reinterpret_cast<uint64_t*>(arena + d.off_agg);
    uint64_t* const prefix = reinterpret_cast<uint64_t*>(arena + d.off_prefix);

    FRZ_STAMP(1);
    {
        const int64_t b = (int64_t)chunk * kBlock + tid;
        const bool active = b < B;
        const uint32_t bl = (uint32_t)(active ? b : B - 1);  // lanes past the end shadow the last env

        int f[CMAX], in[CMAX], fu[CMAX], eqs[AMAX];
        float supp[AMAX], capa[AMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) f[c] = cur.f[c], in[c] = cur.in[c], fu[c] = cur.fu[c];
#pragma unroll
        for (int a = 0; a < AMAX; ++a) supp[a] = cur.supp[a], capa[a] = cur.capa[a], eqs[a] = cur.eqs[a];
        const bool term0 = cur.term != 0, trunc0 = cur.trunc != 0;
        bool term = term0, trunc = trunc0;

        float rew[AMAX];
        int hit[AMAX];
        uint32_t err = 0;
        mask_t burned = 0, put_out = 0;
        int nm = cur.nm, nb = cur.nb;
        bool dead = false;

        float r_field[3][CMAX], r_agent[5][AMAX];  // uniform draws of this step (kStep only)
        float ap[CMAX];                            // fire-fighting power applied to each cell
        FRZ_STAMP(2);
        if (MODE == kStep) {
            // ---------------------------------------------------------------------------------- randomness
            if constexpr (RNG == FRZ_RNG_INJECTED && MODE == kStep) {
#pragma unroll
                for (int e = 0; e < 3; ++e)
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) r_field[e][c] = cur_draws.r_field[e][c];
#pragma unroll
                for (int e = 0; e < 5; ++e)
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) r_agent[e][a] = cur_draws.r_agent[e][a];
            } else if constexpr (RNG == FRZ_RNG_PHILOX) {
                // FRZ_RNG_PHILOX stream (include/frz.h): draw u of the step = 24-bit field u % 5 of block (u / 5, step, 0, 0);
                // field event e of cell c is draw e * HW + c, agent event e of agent a is draw 3 * HW + e * A + a.
                // In-kernel only where HW and A are compile-time (every draw index is then a constant).
                static_assert(EXACT || RNG != FRZ_RNG_PHILOX || MODE != kStep, "runtime shapes stage their draws (wf_philox_fill_kernel)");
                constexpr int U = 3 * CMAX + 5 * AMAX, NB = (U + 4) / 5, NB_EVENT0 = (3 * CMAX + AMAX + 4) / 5;
                const bool need_late = (flags & (kStochRepair | kStochDegrade | kCritical | kStochRefill | kStochSwitch)) != 0 || d.K > 1;
                const int nb_needed = need_late ? NB : NB_EVENT0;  // agent events 1..4 are only drawn when something reads them
                float uni[NB * 5];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    uni[5 * j] = uni[5 * j + 1] = uni[5 * j + 2] = uni[5 * j + 3] = uni[5 * j + 4] = 0.0f;
                    if (j < nb_needed) {
                        const frz::Philox4 w = frz::philox4x32_10((uint32_t)j, (uint32_t)nm, 0u, 0u, cur.seed, 0x46525A00u);
                        uni[5 * j] = frz::philox_unit24<0>(w);
                        uni[5 * j + 1] = frz::philox_unit24<1>(w);
                        uni[5 * j + 2] = frz::philox_unit24<2>(w);
                        uni[5 * j + 3] = frz::philox_unit24<3>(w);
                        uni[5 * j + 4] = frz::philox_unit24<4>(w);
                    }
                }
#pragma unroll
                for (int e = 0; e < 3; ++e)
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) r_field[e][c] = uni[e * CMAX + c];
#pragma unroll
                for (int e = 0; e < 5; ++e)
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) r_agent[e][a] = uni[3 * CMAX + e * AMAX + a];
            }

            FRZ_STAMP(3);
            // ------------------------------------------------------------- action decode (wildfire.py:427-483)
            // The action mapping of the previous rebuild is a pure function of the state it was built from, which is
            // the state just loaded: attackable set of agent a = lit fires within its (equipment-adjusted) range,
            // non-empty only while it has suppressant (wildfire.py:604-623).
            mask_t lit0 = 0;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) lit0 |= (mask_t)(f[c] > 0) << c;

#pragma unroll
            for (int c = 0; c < CMAX; ++c) ap[c] = 0.0f;
            bool users[AMAX], refill[AMAX];
            const bool show_bad = (flags & kShowBad) != 0;
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                users[a] = false;
                refill[a] = false;
                hit[a] = -1;
                rew[a] = 0.0f;
                if (a < A) {
                    const mask_t ok = supp[a] > 0.0f ? (lit0 & (mask_t)s_cfg.range_mask[a][eqs[a]]) : (mask_t)0;
                    refill[a] = cur.act_id[a] == -1;
                    // quirk wildfire.py:434-435: an agent with no attackable task in ANY env of the batch is skipped
                    const bool skipped = prev[1 + a] == 0u;
                    const bool fight = !refill[a] && !skipped;
                    const mask_t sel = show_bad ? lit0 : ok;
                    const bool valid = cur.act_idx[a] >= 0 && cur.act_idx[a] < popc(sel);
                    int target = 0, seen = 0;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        const int bit = (int)((sel >> c) & 1);
                        target = (bit && seen == cur.act_idx[a]) ? c : target;
                        seen += bit;
                    }
                    const bool attackable = ((ok >> target) & 1) != 0;
                    const bool good = fight && valid && (!show_bad || attackable);
                    if (fight && !valid && active) err |= FRZ_ERR_BAD_ACTION_INDEX;
                    const float power = d.power[a] + s_cfg.eq[eqs[a]][1];
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) ap[c] = ap[c] + ((good && target == c) ? power : 0.0f);  // agent order
                    users[a] = good;
                    hit[a] = good ? target : -1;
                    rew[a] = (fight && !good) ? d.bad_attack_penalty : 0.0f;  // assignment, :477
                }
            }

            FRZ_STAMP(4);
            // ---------------------------------------------- agent transitions (suppressant/equipment/capacity)
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (a < A) {
                    // transitions/suppressant_decrease.py:56-61
                    const bool dec = users[a] && (!(flags & kStochSuppDecrease) || r_agent[0][a] < d.p_supp_decrease);
                    float s = dec ? supp[a] - 1.0f : supp[a];
                    s = s < 0.0f ? 0.0f : s;
                    // transitions/equipment.py:51-75 (masks from the value before any write)
                    const int e0 = eqs[a], top = d.S - 1;
                    const bool pristine = e0 == top, damaged = e0 == 0, inter = !pristine && !damaged;
                    const float r1 = r_agent[1][a];
                    const bool repairs = (flags & kStochRepair) ? (damaged && r1 < d.p_repair) : damaged;
                    const bool crit = (flags & kCritical) && pristine && r1 < d.p_critical;
                    bool degr = (flags & kStochDegrade) ? ((pristine || inter) && r1 < d.p_degrade) : (inter || pristine);
                    degr = degr && !crit;
                    int e = repairs ? top : e0;
                    e = crit ? 0 : e;
                    e = degr ? e - 1 : e;
                    // transitions/suppressant_refill.py:63-70 (bonus from the NEW equipment state)
                    const bool inc = refill[a] && (!(flags & kStochRefill) || r_agent[2][a] < d.p_refill);
                    s = inc ? capa[a] + s_cfg.eq[e][0] : s;
                    // transitions/capacity.py:52-64: bucketize(r, cumsum) = #{j : cum[j] < r} (cum padded with +inf,
                    // clamped to the last capacity where the reference would raise IndexError)
                    int ci = 0;
#pragma unroll
                    for (int j = 0; j < FRZ_MAX_CAPACITIES; ++j) ci += r_agent[3][a] > d.cum[j] ? 1 : 0;
                    ci = ci > d.K - 1 ? d.K - 1 : ci;
                    const float new_max = s_cfg.caps[ci];
                    const bool sw = inc && (!(flags & kStochSwitch) || r_agent[4][a] < d.p_switch);
                    const float bonus = s - capa[a];
                    capa[a] = sw ? new_max : capa[a];
                    s = sw ? new_max + bonus : s;
                    supp[a] = s;
                    eqs[a] = e;
                    at32(rowsf, (uint32_t)(r_supp + a) * Bu + bl) = supp[a];
                    at32(rowsf, (uint32_t)(r_cap + a) * Bu + bl) = capa[a];
                    at32(rows, (uint32_t)(r_equip + a) * Bu + bl) = eqs[a];
                }
            }
        }

        // ------------------------------------------ agent observations (wildfire.py:677-681, 704-716)
        // final as soon as the suppressants are: stored here so that they drain behind the fire transitions
        {
            const int k = d.others_k, width = (A - 1) * k;  // k = 2 + power column + suppressant column
            const bool op = (flags & kObsPower) != 0;
#pragma unroll
            for (int a = 0; a < AMAX; ++a)
                if (a < A) {
                    reinterpret_cast<float4*>(obs_self)[a * B + bl] = make_float4((float)d.ay[a], (float)d.ax[a], d.power[a], supp[a]);
                    float* const others = obs_others + (a * B + bl) * (int64_t)width;
                    int j = 0;  // record index: the other agents in agent order
#pragma unroll
                    for (int o = 0; o < AMAX; ++o)
                        if (o < A && o != a) {
                            float* const rec = others + j * k;
                            const float y = (float)d.ay[o], x = (float)d.ax[o];
                            if (k == 4) {
                                *reinterpret_cast<float4*>(rec) = make_float4(y, x, d.power[o], supp[o]);
                            } else if (k == 3) {
                                rec[0] = y;
                                rec[1] = x;
                                rec[2] = op ? d.power[o] : supp[o];
                            } else {
                                *reinterpret_cast<float2*>(rec) = make_float2(y, x);
                            }
                            ++j;
                        }
                }
        }

        FRZ_STAMP(5);
        if (MODE == kStep) {
            // ------------------------------------------------------------ fire increase / decrease per cell
            mask_t lit2 = 0;
            const int almost_state = d.num_fire_states - 2, burnout_state = d.num_fire_states - 1;
            const float p_unmet = (flags & kStochIncrease) ? d.p_increase : 1.0f;
            const float p_almost = (flags & kStochBurnouts) ? d.p_burnout : d.p_increase;  // fire_increase.py:77-80
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                if (c < HW) {
                    {  // transitions/fire_increase.py:61-91
                        const int required = f[c] >= 0 ? f[c] : 0;
                        const float diff = (float)required - ap[c];
                        const bool lit = f[c] > 0 && in[c] > 0;
                        const bool unmet = diff > 0.0f && lit;
                        const bool almost = unmet && in[c] == almost_state;
                        float prob = unmet ? (almost ? p_almost : p_unmet) : 0.0f;
                        prob = clamp01(prob);
                        const bool inc = r_field[0][c] < prob;
                        in[c] += inc ? 1 : 0;
                        const bool bo = inc && in[c] >= burnout_state;
                        f[c] = bo ? -f[c] : f[c];
                        fu[c] = bo ? (fu[c] - 1 < 0 ? 0 : fu[c] - 1) : fu[c];
                        burned |= (mask_t)bo << c;
                    }
                    {  // transitions/fire_decrease.py:56-77: p = p_dec + ((-1 * diff) * bonus), each op rounded
                        const int required = f[c] >= 0 ? f[c] : 0;
                        const float diff = (float)required - ap[c];
                        const bool lit = f[c] > 0 && in[c] > 0;
                        const bool met = diff <= 0.0f && lit;
                        const float stoch_p = __fadd_rn(d.p_decrease, __fmul_rn(__fmul_rn(-1.0f, diff), d.decrease_bonus));
                        float prob = met ? ((flags & kStochDecrease) ? stoch_p : 1.0f) : 0.0f;
                        prob = clamp01(prob);
                        const bool dec = r_field[1][c] < prob;
                        in[c] -= dec ? 1 : 0;
                        const bool po = dec && in[c] <= 0;
                        f[c] = po ? -f[c] : f[c];
                        fu[c] = po ? fu[c] - 1 : fu[c];  // unclamped, :75
                        put_out |= (mask_t)po << c;
                    }
                    lit2 |= (mask_t)(f[c] > 0 && in[c] > 0) << c;
                }
            }
            // ---------------------------------------- fire spread stencil (transitions/fire_spreads.py:44-57)
            int fuel_sum = 0;
            bool any_fire = false;
            {
                const mask_t from_n = (lit2 << W) & (mask_t)d.has_n, from_s = (lit2 >> W) & (mask_t)d.has_s;
                const mask_t from_w = (lit2 << 1) & (mask_t)d.has_w, from_e = (lit2 >> 1) & (mask_t)d.has_e;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    if (c < HW) {
                        float prob = 0.0f;  // conv2d accumulation order: N, W, E, S
                        prob = __fadd_rn(prob, ((from_n >> c) & 1) ? d.spread_n : 0.0f);
                        prob = __fadd_rn(prob, ((from_w >> c) & 1) ? d.spread_w : 0.0f);
                        prob = __fadd_rn(prob, ((from_e >> c) & 1) ? d.spread_e : 0.0f);
                        prob = __fadd_rn(prob, ((from_s >> c) & 1) ? d.spread_s : 0.0f);
                        bool unlit = f[c] < 0 && in[c] == 0;
                        unlit = unlit && (!(flags & kUseFuel) || fu[c] > 0);
                        prob = unlit ? __fadd_rn(prob, d.random_ignition) : 0.0f;
                        const bool spread = r_field[2][c] < prob;
                        f[c] = spread ? -f[c] : f[c];
                        in[c] = spread ? d.ignition[c] : in[c];
                        at32(rows, (uint32_t)(r_intensity + c) * Bu + bl) = in[c];  // final: stored cell by cell
                        at32(rows, (uint32_t)(r_fuel + c) * Bu + bl) = fu[c];
                        fuel_sum += fu[c];
                        any_fire = any_fire || f[c] > 0;
                    }
                }
            }
            // termination test (wildfire.py:560-570): no lit fire left (and no fuel when fuel is tracked)
            dead = !any_fire;
            if (flags & kUseFuel) dead = dead && fuel_sum <= 0;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                f[c] = dead ? 0 : f[c];  // :570
                if (c < HW) at32(rows, (uint32_t)(r_fires + c) * Bu + bl) = f[c];
            }
            nm += 1;
            trunc = (flags & kTruncate) ? nm >= d.max_steps : trunc0;
            term = term0 || dead;
        }

        // ======================================================================================================
        // update_observations + update_actions on the new state (wildfire.py:586-717)
        // ======================================================================================================
        FRZ_STAMP(6);
        mask_t lit1 = 0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) lit1 |= (mask_t)(f[c] > 0) << c;
        lit1 = active ? lit1 : (mask_t)0;
        mask_t ok1[AMAX];
        uint64_t packed[PW];
#pragma unroll
        for (int w = 0; w < PW; ++w) packed[w] = 0;
        const int F = popc(lit1);
        packed[0] = (uint64_t)F;
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            ok1[a] = 0;
            if (a < A) {
                ok1[a] = supp[a] > 0.0f ? (lit1 & (mask_t)s_cfg.range_mask[a][eqs[a]]) : (mask_t)0;
                packed[(a + 1) >> 2] |= (uint64_t)popc(ok1[a]) << (16 * ((a + 1) & 3));
            }
        }

        // workgroup-exclusive prefix of the per-env counts (four 16-bit channels per word)
        uint64_t incl[PW];
#pragma unroll
        for (int w = 0; w < PW; ++w) incl[w] = frz::wave_inclusive_scan(packed[w]);
        const uint32_t live_nt = (uint32_t)__popcll(__ballot(active && !term));
        const uint32_t live_ntr = (uint32_t)__popcll(__ballot(active && !trunc));
        __syncthreads();  // LDS reuse across chunks of a persistent workgroup
        if (lane == 63) {
#pragma unroll
            for (int w = 0; w < PW; ++w) s_wave_scan[wave][w] = incl[w];
            s_wave_live[wave][0] = live_nt;
            s_wave_live[wave][1] = live_ntr;
        }
        __syncthreads();
        uint64_t base[PW], block_total[PW];
#pragma unroll
        for (int w = 0; w < PW; ++w) {
            base[w] = 0;
            block_total[w] = 0;
#pragma unroll
            for (int j = 0; j < frz::kWaves; ++j) {
                const uint64_t t = s_wave_scan[j][w];
                base[w] += j < wave ? t : 0ull;
                block_total[w] += t;
            }
        }

        FRZ_STAMP(7);
        // publish this chunk's channel sums; the bookkeeping below runs while the hand-off is in flight
        const int round_first = chunk & ~(frz::kRound - 1);  // chunks are handed off in windows of kRound
        uint32_t my_total = 0;                         // this chunk's sum of channel `tid` (tid < nch)
        if (tid < nch) {
            if (tid <= A) {
                uint64_t word = block_total[0];
#pragma unroll
                for (int w = 1; w < PW; ++w) word = (tid >> 2) == w ? block_total[w] : word;
                my_total = (uint32_t)((word >> (16 * (tid & 3))) & 0xFFFFull);
            } else {
                const int which = tid - ch_nt;
#pragma unroll
                for (int j = 0; j < frz::kWaves; ++j) my_total += s_wave_live[j][which];
            }
            frz::granule_store(agg + (int64_t)chunk * nch + tid, tag, my_total);
        }

        if (MODE == kStep) {
            // -------------------------------------------------- rewards and termination (wildfire.py:534-582)
            float fire_reward_sum = 0.0f, burnout_total = 0.0f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                if (c < HW) {
                    const float fr = d.fire_rewards[c];
                    fire_reward_sum = __fadd_rn(fire_reward_sum, ((put_out >> c) & 1) ? fr : 0.0f);
                    const float pen = (flags & kPenaltyScaled) ? __fmul_rn(-1.0f, fr) : d.burnout_penalty;
                    burnout_total = __fadd_rn(burnout_total, ((burned >> c) & 1) ? pen : 0.0f);
                }
            }
            const bool newly = !term0 && dead;
            // correctly rounded float32 log via double (matches the oracle bit for bit; the reference's torch.log is
            // a <=1-ulp float32 log).  Only evaluated by wavefronts that hold a newly terminated env.
            float log_burnouts = 0.0f;
            if (newly && d.termination_kappa != 0.0f) log_burnouts = (float)log((double)nb + 1.0);
            const float penalty = __fmul_rn(d.termination_kappa, log_burnouts);
            float term_reward = __fsub_rn(d.termination_reward, penalty);
            term_reward = term_reward < 0.0f ? 0.0f : term_reward;
            const int n_burn = popc(burned), n_put = popc(put_out);
            nb += n_burn;

            const bool localize = (flags & kLocalize) != 0;
            const bool track = (flags & kTrackCumulative) != 0, write_trunc = (flags & kTruncate) != 0;
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (a < A) {
                    float base_reward = fire_reward_sum;
                    if (localize) {
                        base_reward = 0.0f;
#pragma unroll
                        for (int c = 0; c < CMAX; ++c)
                            if (c < HW) base_reward = (hit[a] == c && ((put_out >> c) & 1)) ? d.fire_rewards[c] : base_reward;
                    }
                    rew[a] = __fadd_rn(rew[a], __fadd_rn(base_reward, burnout_total));
                    rew[a] = newly ? __fadd_rn(rew[a], term_reward) : rew[a];
                    at32(rowsf, (uint32_t)(r_rewards + a) * Bu + bl) = rew[a];
                    at32(rows1, (u_term + (uint32_t)a) * Bu + bl) = (uint8_t)term;
                    if (write_trunc) at32(rows1, (u_trunc + (uint32_t)a) * Bu + bl) = (uint8_t)trunc;
                    if (track) at32(rowsf, (uint32_t)(r_cum + a) * Bu + bl) = __fadd_rn(cur.cum[a], rew[a]);
                }
            }
            at32(rows, (uint32_t)r_moves * Bu + bl) = nm;
            at32(rows, (uint32_t)r_burnouts * Bu + bl) = nb;
            at32(rows8, q_burnouts * Bu + bl) = n_burn;
            at32(rows8, q_putouts * Bu + bl) = n_put;
        }
        if (active) {
#pragma unroll
            for (int a = 0; a < AMAX; ++a)
                if (a < A) at32(rows, (uint32_t)(r_atc + a) * Bu + bl) = popc(ok1[a]);
            at32(rows8, q_etc * Bu + bl) = F;
        }

        FRZ_STAMP(8);
        // -------------------------------------------------- inter-workgroup exclusive prefix (single pass)
        // chunk j needs the channel sums of all chunks < j: those of this round that precede it (their workgroups
        // are co-resident and have published or are about to) + the inclusive prefix the previous round's last
        // chunk published.  Thread t sums channel (t % NCHP) over predecessors t / NCHP, t / NCHP + PP, ...; the
        // window's loads are unconditional (lanes without a predecessor read granule 0 and ignore it) so that they
        // are in flight together: one L2 round trip when the granules are already there.
        bool timed_out = false;
        uint32_t acc = 0;
        {
            const int ch = tid & (NCHP - 1), slot = tid / NCHP;
            constexpr int PP = kBlock / NCHP, UNR = 8;
            for (int first = round_first; first < chunk; first += PP * UNR) {
                uint32_t part = 0;
                for (int spin = 0;; ++spin) {  // bounded: every granule of the window must carry this launch's tag
                    bool all = true;
                    part = 0;
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int pred = first + u * PP + slot;
                        const bool valid = pred < chunk && ch < nch;
                        const uint64_t g = frz::granule_load(agg + (valid ? (int64_t)pred * nch + ch : (int64_t)0));
                        all = all && (!valid || (uint32_t)(g >> 32) == tag);
                        part += valid ? (uint32_t)g : 0u;
                    }
                    if (all) break;
                    if (spin >= (1 << 22)) {
                        timed_out = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                acc += part;
            }
            if (round_first > 0 && tid < nch) acc += frz::granule_wait(prefix + (int64_t)(round_first - 1) * nch + tid, tag, &timed_out);
#pragma unroll
            for (int dd = NCHP; dd < 64; dd <<= 1) acc += __shfl_xor(acc, dd, 64);
            if (lane < NCHP) s_reduce[wave][lane] = acc;
        }
        __syncthreads();
        if (tid < nch) {
            uint32_t s = 0;
#pragma unroll
            for (int j = 0; j < frz::kWaves; ++j) s += s_reduce[j][tid];
            s_prefix[tid] = s;
            const bool round_last = (chunk & (frz::kRound - 1)) == frz::kRound - 1 || chunk == nchunks - 1;
            if (round_last) {
                frz::granule_store(prefix + (int64_t)chunk * nch + tid, tag, s + my_total);
                if (chunk == nchunks - 1) cur_totals[tid] = s + my_total;  // batch totals, read by the next launch
            }
        }
        __syncthreads();
        if (timed_out) err |= FRZ_ERR_SCAN_TIMEOUT;

        FRZ_STAMP(9);
        // ------------------------------------------------------------------ jagged stores (values + offsets)
        if (active) {
            const int64_t cap = B * HW;
            int64_t* const task_values = reinterpret_cast<int64_t*>(arena + d.off_task_values);
            int64_t* const task_offsets = reinterpret_cast<int64_t*>(arena + d.off_task_offsets);
            int64_t* const obs_map = reinterpret_cast<int64_t*>(arena + d.off_obs_map);
            const int64_t off_f = (int64_t)s_prefix[0] + (int64_t)((base[0] + incl[0] - packed[0]) & 0xFFFFull);
            task_offsets[b] = off_f;
            if (b == B - 1) task_offsets[B] = off_f + F;
            // Row of cell c's task inside the env's segment = number of lit cells below it: every store address is a
            // closed form of the lit mask (no running counter carried through divergent control flow).
            int64_t* const trow = task_values + off_f * 4;
            int64_t* const omap = obs_map + off_f;
            int rk[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                rk[c] = popc(lit1 & (mask_t)(((mask_t)1 << c) - 1));
                if ((lit1 >> c) & 1) {
                    const int yx = d.cell_yx[c];
                    longlong2* const row = reinterpret_cast<longlong2*>(trow + rk[c] * 4);
                    row[0] = make_longlong2(yx >> 16, yx & 0xFFFF);
                    row[1] = make_longlong2(f[c], in[c]);
                    omap[rk[c]] = rk[c];
                }
            }
            int64_t* const act_values = reinterpret_cast<int64_t*>(arena + d.off_act_values);
            int64_t* const act_offsets = reinterpret_cast<int64_t*>(arena + d.off_act_offsets);
            int64_t* const bad_values = reinterpret_cast<int64_t*>(arena + d.off_bad_values);
            int64_t* const bad_offsets = reinterpret_cast<int64_t*>(arena + d.off_bad_offsets);
            const bool show_bad = (flags & kShowBad) != 0;
#pragma unroll
            for (int a = 0; a < AMAX; ++a)
                if (a < A) {
                    const int w = (a + 1) >> 2, sh = 16 * ((a + 1) & 3);
                    uint64_t excl = base[0] + incl[0] - packed[0];
#pragma unroll
                    for (int ww = 1; ww < PW; ++ww) excl = w == ww ? base[ww] + incl[ww] - packed[ww] : excl;
                    const int64_t off_a = (int64_t)s_prefix[a + 1] + (int64_t)((excl >> sh) & 0xFFFFull);
                    const int fa = popc(ok1[a]);
                    act_offsets[a * (B + 1) + b] = off_a;
                    if (b == B - 1) act_offsets[a * (B + 1) + B] = off_a + fa;
                    int64_t* av = act_values + a * cap + off_a;
                    int64_t* bv = bad_values + a * cap + (off_f - off_a);  // bad = listed but not attackable
                    if (show_bad) {
                        bad_offsets[a * (B + 1) + b] = off_f - off_a;
                        if (b == B - 1) bad_offsets[a * (B + 1) + B] = (off_f - off_a) + (F - fa);
                    }
                    // entry by entry (as in wildfire_roles.inl, round 4): the j-th store writes every env's j-th entry — the row number of the
                    // j-th member cell — and the wavefront stops at the longest list among its envs, instead of one predicated store per
                    // cell whatever the lists hold (24 cells x 8 agents here)
                    {
                        mask_t m = ok1[a];
                        for (int j = 0; j < CMAX; ++j) {
                            if (!__any(m != 0)) break;
                            if (m != 0) av[j] = popc(lit1 & (mask_t)((m & (mask_t)(0 - m)) - 1));
                            m &= (mask_t)(m - 1);
                        }
                        if (show_bad) {
                            m = lit1 & ~ok1[a];
                            for (int j = 0; j < CMAX; ++j) {
                                if (!__any(m != 0)) break;
                                if (m != 0) bv[j] = popc(lit1 & (mask_t)((m & (mask_t)(0 - m)) - 1));
                                m &= (mask_t)(m - 1);
                            }
                        }
                    }
                }
        }
        FRZ_STAMP(10);
        if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);

        // The workgroup owning the last chunk finished its look-back only after every other chunk published, i.e.
        // after every workgroup of this launch read the epoch: it can advance it for the next launch.
        if (chunk == nchunks - 1 && tid == 0) __hip_atomic_store(epoch_ptr, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// FRZ_RNG_PHILOX randomness staged in the arena for the variants with runtime grid shapes (their draw indices are not
// compile-time constants): one thread per (env, Philox block), see include/frz.h for the stream.
__global__ void __launch_bounds__(kBlock) wf_philox_fill_kernel(char* __restrict__ arena, const WfDev* __restrict__ dev) {
    const WfDev& d = *dev;
    const int64_t B = d.B;
    const int HW = d.HW, A = d.A;
    const int U = 3 * HW + 5 * A, per_env = (U + 4) / 5;
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= B * per_env) return;
    const int64_t b = t / per_env;
    const int j = (int)(t % per_env);
    const int32_t* rows = reinterpret_cast<const int32_t*>(arena + d.off_rows4);
    float* field = reinterpret_cast<float*>(arena + d.off_rand_field);
    float* agent = reinterpret_cast<float*>(arena + d.off_rand_agent);
    const uint32_t seed = (uint32_t)rows[d.r_seeds * B + b], step = (uint32_t)rows[d.r_moves * B + b];
    const frz::Philox4 w = frz::philox4x32_10((uint32_t)j, step, 0u, 0u, seed, 0x46525A00u);
    for (int k = 0; k < 5; ++k) {
        const int u = 5 * j + k;
        const float r = frz::philox_unit24(w, k);
        if (u < 3 * HW) {
            field[((int64_t)(u / HW) * B + b) * HW + u % HW] = r;
        } else if (u < U) {
            const int v = u - 3 * HW;
            agent[((int64_t)(v / A) * B + b) * A + v % A] = r;
        }
    }
}

// uniform random policy over OneOf([task] * n + [noop]) (spaces/actions.py:23-41; baselines/random.py:20):
// member index j ~ U{0..n}; j < n -> [j, 0] (fight task j of the action mapping), j == n -> [n, -1] (noop/refill)
__global__ void __launch_bounds__(kBlock) wf_policy_kernel(const char* arena, uint32_t seed_lo, uint32_t seed_hi, uint32_t step_lo,
                                                             uint32_t step_hi, int32_t* actions) {
    const WfDev& d = *reinterpret_cast<const WfDev*>(arena);
    const int64_t B = d.B;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)d.A * B) return;
    const int64_t b = i % B;
    const int32_t* rows = reinterpret_cast<const int32_t*>(arena + d.off_rows4);
    const int64_t* rows8 = reinterpret_cast<const int64_t*>(arena + d.off_rows8);
    const int n = (d.flags & kShowBad) ? (int)rows8[d.q_etc * B + b] : rows[d.r_atc * B + i];
    const uint32_t env_seed = (uint32_t)rows[d.r_seeds * B + b];
    // one block serves four agents: agent a draws word a % 4 of block (a / 4, step), keyed by the env seed
    const uint32_t agent = (uint32_t)(i / B);
    const frz::Philox4 w = frz::philox4x32_10(agent >> 2, 0u, step_lo, step_hi, seed_lo ^ env_seed, seed_hi);
    uint32_t word = w.w[0];
    word = (agent & 3u) == 1u ? w.w[1] : word;
    word = (agent & 3u) == 2u ? w.w[2] : word;
    word = (agent & 3u) == 3u ? w.w[3] : word;
    const int j = (int)(((uint64_t)word * (uint64_t)(n + 1)) >> 32);
    reinterpret_cast<int2*>(actions)[i] = j < n ? make_int2(j, 0) : make_int2(n, -1);
}

// Episode metrics in one launch (the reductions a rollout loop takes after each episode): out[a] += sum_b cumulative
// reward of agent a, out[A] += sum_b num_moves (env-steps taken), out[A + 1] += envs whose agents are all terminated or all
// truncated.  Deterministic: each of the launch's workgroups reduces a fixed slice in a fixed order into its own partial
// row; the workgroup that arrives last (atomic ticket) adds the rows up in a fixed tree.  Hand-off between workgroups: the
// partial rows are written and read with agent-scope (L1-bypassing, write-through) accesses, every storing wavefront drains its
// stores before the workgroup's barrier, one lane then takes the ticket (MI355X_MICROARCH.md, inter-workgroup visibility).
constexpr int kMetricBlocks = 256;

// frz_wildfire_export_totals / import_totals: the batch totals the last executed step left (channel 0: lit fires, 1 + a: fires agent a can
// attack, A + 1 / A + 2: envs not terminated / not truncated) — what the next step's batch-global tests read (utils/env.py:211-213,
// wildfire.py:434-435) — copied out of / into the slot the current epoch selects.  One wavefront.
__global__ void __launch_bounds__(64) wf_totals_kernel(char* arena, int32_t* staging, int import) {
    const WfDev& d = *reinterpret_cast<const WfDev*>(arena);
    uint32_t* const epoch_block = reinterpret_cast<uint32_t*>(arena + d.off_epoch);  // [0] epoch, [1] epoch of the last export, [2 ..] the shard's own totals then
    const uint32_t epoch = epoch_block[0];
    uint32_t* const slot = reinterpret_cast<uint32_t*>(arena + d.off_totals) + ((epoch + 1u) & 1u) * kTotalsStride;  // written under epoch - 1
    const int i = threadIdx.x;
    if (i >= d.nch) return;
    if (import) {
        slot[i] = (uint32_t)staging[i];
        return;
    }
    // a launch that found the batch finished left epoch and totals alone, and the slot then already holds the job's sums: the shard's own
    // totals are kept beside the epoch they belong to, so that exporting twice under one epoch exports the same values
    const bool fresh = epoch_block[1] != epoch;
    const uint32_t mine = fresh ? slot[i] : epoch_block[2 + i];
    if (fresh) epoch_block[2 + i] = mine;
    staging[i] = (int32_t)mine;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (fresh && i == 0) epoch_block[1] = epoch;
}

// ended_only (step-by-step rollouts with FRZ_ROLLOUT_AUTO_RESET, between a step and the reset of its finished envs): the returns of the
// finished envs only, one env-step per env, the number of finished envs
__global__ void __launch_bounds__(kBlock) wf_metrics_kernel(char* __restrict__ arena, double* __restrict__ out, int ended_only) {
    const WfDev& d = *reinterpret_cast<const WfDev*>(arena);
    const int64_t B = d.B;
    const int A = d.A, nrow = A + 2;
    const int nblocks = (int)gridDim.x;
    __shared__ double s_part[frz::kWaves][FRZ_MAX_AGENTS + 2];
    __shared__ int s_last;
    const float* rowsf = reinterpret_cast<const float*>(arena + d.off_rows4);
    const int32_t* rows = reinterpret_cast<const int32_t*>(arena + d.off_rows4);
    const uint8_t* rows1 = reinterpret_cast<const uint8_t*>(arena + d.off_rows1);
    double* const partial = reinterpret_cast<double*>(arena + d.off_metrics);
    uint32_t* const counter = reinterpret_cast<uint32_t*>(arena + d.off_epoch) + 48;
    double acc[FRZ_MAX_AGENTS + 2];
    for (int i = 0; i < nrow; ++i) acc[i] = 0.0;
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += (int64_t)nblocks * kBlock) {
        bool all_term = true, all_trunc = true;
        for (int a = 0; a < A; ++a) {
            all_term = all_term && rows1[(int64_t)(d.u_term + a) * B + b] != 0;
            all_trunc = all_trunc && rows1[(int64_t)(d.u_trunc + a) * B + b] != 0;
        }
        const bool finished = all_term || all_trunc;
        for (int a = 0; a < A; ++a) acc[a] += (ended_only && !finished) ? 0.0 : (double)rowsf[(int64_t)(d.r_cum + a) * B + b];
        acc[A] += ended_only ? 1.0 : (double)rows[(int64_t)d.r_moves * B + b];
        acc[A + 1] += finished ? 1.0 : 0.0;
    }
    const int lane = frz::lane_id(), wave = frz::wave_id();
    for (int i = 0; i < nrow; ++i) {
        double v = acc[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);  // fixed tree: deterministic
        if (lane == 0) s_part[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < nrow) {
        double v = 0.0;
        for (int w = 0; w < frz::kWaves; ++w) v += s_part[w][threadIdx.x];
        __hip_atomic_store(&partial[(int64_t)blockIdx.x * nrow + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wavefront drains before the barrier the ticket follows
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(counter, 1u);
        s_last = t == (uint32_t)nblocks - 1u;
        if (s_last) atomicExch(counter, 0u);
    }
    __syncthreads();
    if (s_last) {  // thread t takes partial row t (nblocks <= kBlock); rows are summed lane-tree, then wave 0..3
        for (int i = 0; i < nrow; ++i) {
            double v = (int)threadIdx.x < nblocks
                           ? __hip_atomic_load(&partial[(int64_t)threadIdx.x * nrow + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                           : 0.0;
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) s_part[wave][i] = v;
        }
        __syncthreads();
        if (threadIdx.x < nrow) {
            double v = 0.0;
            for (int w = 0; w < frz::kWaves; ++w) v += s_part[w][threadIdx.x];
            out[threadIdx.x] += v;
        }
    }
}

}  // namespace

// ================================================================================================================
// host side of the C-ABI
// ================================================================================================================
struct frz_wildfire_env {
    frz_wildfire_cfg cfg;
    WfDev dev;
    char* arena = nullptr;
    bool was_reset = false;
    int variant = 0;       // index into the (CMAX, AMAX) instantiation table
    int lane_variant = 0;  // the first runtime-shape entry that holds the shape (what the lane-per-env kernel runs)
    // set for the duration of one frz_wildfire_step_random_policy call
    bool fused_policy = false;
    uint64_t policy_seed = 0, policy_step = 0;
    int32_t* actions_out = nullptr;
    hipEvent_t start_event = nullptr, stop_event = nullptr;  // the pair the next timed launch records into
    std::vector<hipEvent_t> timing_events;                   // pool of frz_wildfire_timed_rollout
    bool timed = false;
    bool ticketed = false;  // field/crew kernels: more chunks than resident workgroups
    // multi-step launches of frz_wildfire_rollout_random_policy (wf_roles_kernel<..., PERSIST>): byte distance from the packed list
    // buffers to their second copy (0: not available for this env), and the steps the launch being enqueued performs
    int64_t list_copy_delta = 0;
    int32_t rollout_steps = 1;
    double* rollout_metrics = nullptr;  // frz_wildfire_rollout_random_policy_metrics: folded into the multi-step launch being enqueued
    const frz_rollout_spec* rollout_spec = nullptr;  // frz_wildfire_rollout: the spec of the multi-step launch being enqueued
    bool exclusive_device = false;  // frz_wildfire_set_exclusive_device: multi-step launches allowed
    // grid family, overlapped rollouts (wildfire_grid.hip launch_cpl): the second stream and the events of the hand-overs, made on first use
    hipStream_t side_stream = nullptr;
    hipEvent_t scan_done = nullptr, lists_done = nullptr;
    const frz_wf::WgOverlap* overlap = nullptr;  // set for the duration of a step launch of an overlapped rollout
    int64_t grid_copy_delta = 0;  // bytes from the mask words / lit cells to their second copies
    frz_wildfire_saved_state saved = {};  // frz_wildfire_set_saved_initial: what a partial reset restores (fires == nullptr: the configured state)
    // grids above 16 cells (wildfire_grid.hip): the kernels' configuration and the tables uploaded into the arena at bind
    WgDev gdev;
    WgAgentTable agent_table;
    std::vector<int32_t> cell_tables;    // fire_rewards (float bits), ignition, initial fires / intensity / fuel: HW entries each
    std::vector<uint64_t> range_words;   // [A][S][chunks]
};

namespace {

struct Variant {
    int cmax, amax;
    bool exact;
};
// the lane-per-env kernel unrolls its cell and agent loops to (CMAX, AMAX): a shape runs the smallest instantiation that holds it
// (3x3 / 3x4 / 4x4 grids do not pay for the 24-cell one); grids of <= 16 cells with <= 4 agents run the field/crew kernel by default
#define FRZ_X(i, c, a, e) {c, a, e},
constexpr Variant kVariants[] = {FRZ_WF_VARIANT_LIST(FRZ_X)};
#undef FRZ_X
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// FRZ_RNG_PHILOX for the variants with runtime shapes: stage the draws, then run the injected-randomness step
void stage_philox(WfArgs& a, int& rng, hipStream_t stream) {
    const WfDev* host = a.host_dev;
    const int per_env = (3 * host->HW + 5 * host->A + 4) / 5;
    const int64_t n = (int64_t)host->B * per_env;
    hipLaunchKernelGGL(wf_philox_fill_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, a.arena,
                       reinterpret_cast<const WfDev*>(a.arena));
    a.field_rand = reinterpret_cast<const float*>(a.arena + host->off_rand_field);
    a.agent_rand = reinterpret_cast<const float*>(a.arena + host->off_rand_agent);
    rng = FRZ_RNG_INJECTED;
}

template <int CMAX, int AMAX, bool EXACT>
void launch_variant(const WfArgs& args, int grid, int rng, int mode, hipStream_t stream) {
    WfArgs a = args;
    const WfDev* dev = reinterpret_cast<const WfDev*>(a.arena);
    if constexpr (!EXACT) {
        if (mode == kStep && rng == FRZ_RNG_PHILOX) stage_philox(a, rng, stream);
    }
    const WfLaunch launch = make_launch(a);
    if (mode == kRebuild) {
        launch_step_kernel(a, wf_step_kernel<CMAX, AMAX, EXACT, FRZ_RNG_INJECTED, kRebuild>, grid, kBlock, stream, a.arena, dev, a.actions,
                           a.field_rand, a.agent_rand, launch);
    } else if (rng == FRZ_RNG_PHILOX) {
        if constexpr (EXACT)
            launch_step_kernel(a, wf_step_kernel<CMAX, AMAX, EXACT, FRZ_RNG_PHILOX, kStep>, grid, kBlock, stream, a.arena, dev, a.actions,
                               a.field_rand, a.agent_rand, launch);
    } else {
        launch_step_kernel(a, wf_step_kernel<CMAX, AMAX, EXACT, FRZ_RNG_INJECTED, kStep>, grid, kBlock, stream, a.arena, dev, a.actions,
                           a.field_rand, a.agent_rand, launch);
    }
}

// The lane-per-env kernel is instantiated for the runtime-shape entries of the list only: every exact shape has a field/crew kernel
// (wildfire_roles.hip), and with FRZ_WF_KERNEL=lane such a shape runs the smallest runtime-shape instantiation that holds it
// (env->lane_variant) — the cross-check the parity tests want, at a third of this file's build time and code size.
int launch_lane(frz_wildfire_env* env, const WfArgs& args, int rng, int mode, hipStream_t stream) {
    const int grid = env->dev.nchunks;  // one workgroup per chunk
    switch (env->lane_variant) {
#define FRZ_X(i, c, a, e)                                                          \
    case i:                                                                        \
        if constexpr (!e) launch_variant<c, a, e>(args, grid, rng, mode, stream); \
        else return FRZ_E_INVALID;                                                 \
        break;
        FRZ_WF_VARIANT_LIST(FRZ_X)
#undef FRZ_X
        default: return FRZ_E_INVALID;
    }
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

int launch(frz_wildfire_env* env, const WfArgs& args, int rng, int mode, hipStream_t stream) {
    if (env->dev.grid) return launch_grid(env->gdev, env->arena, args, rng, mode, env->ticketed, stream, env->overlap);
    if (!env->dev.roles) {
        WfArgs a = args;
        a.ticketed = env->ticketed;
        return launch_lane(env, a, rng, mode, stream);
    }
    {
        WfArgs a = args;
        // (round 4: the runtime-shape field/crew variants up to 16 cells draw in the kernel, single steps included; <24, 8> and the lane-per-env
        // kernel read staged draws)
        if (!kVariants[env->variant].exact && kVariants[env->variant].cmax > 16 && mode == kStep && rng == FRZ_RNG_PHILOX) stage_philox(a, rng, stream);
        a.ticketed = env->ticketed;
        a.n_steps = env->rollout_steps;
        a.scratch_delta = env->list_copy_delta;
        a.metrics_out = env->rollout_metrics;
        if (const frz_rollout_spec* spec = env->rollout_spec) {  // one multi-step launch driven by a rollout spec
            const WfDev& p = env->dev;
            const int64_t AB2 = (int64_t)p.A * p.B * 2;
            a.rollout_flags = spec->flags;
            a.seed_increment = spec->seed_increment;
            a.seed_stride = spec->seed_stride;
            a.tape_actions_step = spec->action_tape ? AB2 : 0;
            if (spec->list_record) {
                a.list_record_delta = static_cast<char*>(spec->list_record) - (env->arena + p.off_task_offsets);
                a.list_record_step = p.off_actions - p.off_task_offsets;
            }
            a.reward_tape = spec->reward_tape;
            a.done_tape = spec->done_tape;
            a.actions_out_step = spec->record_actions ? AB2 : 0;
            a.supp_tape = (spec->flags & FRZ_ROLLOUT_OBS_COMPACT) ? static_cast<float*>(spec->obs_tape) : nullptr;
            a.state_tape = static_cast<int32_t*>(spec->state_tape);
        }
        return launch_roles(a, env->variant, env->dev.nchunks, rng, mode, stream);  // one workgroup per chunk
    }
}

template <typename T>
T* at(char* arena, int64_t off) {
    return reinterpret_cast<T*>(arena + off);
}

// frz_wildfire_create for the grid family (wildfire_grid.hip): cell arrays env-major [B][H*W]; everything per env that is not a cell
// array keeps the [rows][B] blocks of the other kernels, described by the same WfDev block at arena offset 0, so that the helper kernels
// (random policy, episode metrics) serve both families unchanged.
int create_grid(const frz_wildfire_cfg* cfg, frz_wildfire_env** out) {
    const int H = cfg->grid_height, W = cfg->grid_width, HW = H * W, A = cfg->num_agents, S = cfg->num_equipment_states;
    if (S <= 0 || S > FRZ_MAX_EQUIPMENT_STATES || cfg->num_capacities <= 0 || cfg->num_capacities > FRZ_MAX_CAPACITIES) return FRZ_E_INVALID;
    if (cfg->num_fire_states < 2 || cfg->initial_equipment_state < 0 || cfg->initial_equipment_state >= S) return FRZ_E_INVALID;
    const int64_t B = cfg->parallel_envs;
    // arrays addressed with 32-bit byte offsets from a uniform base: the [rows][B] blocks, one env's cells, the observation records
    if ((int64_t)(6 * A + 8) * B >= (int64_t)1 << 28 || (int64_t)A * (A - 1) * B * 16 >= (int64_t)1 << 32) return FRZ_E_INVALID;
    frz_wildfire_env* env = new (std::nothrow) frz_wildfire_env();
    if (!env) return FRZ_E_INVALID;
    env->cfg = *cfg;
    WfDev& p = env->dev;
    WgDev& g = env->gdev;
    std::memset(&p, 0, sizeof(p));
    std::memset(&g, 0, sizeof(g));
    int chunks = 1;
    while (chunks * 64 < HW) chunks *= 2;  // the kernels are instantiated for 1, 2, 4, 8, 16 cells per lane
    p.grid = 1;
    p.B = g.B = cfg->parallel_envs, p.H = g.H = H, p.W = g.W = W, p.HW = g.HW = HW, p.A = g.A = A, p.S = g.S = S;
    p.K = g.K = cfg->num_capacities;
    p.nchunks = g.nchunks = (int)((B + kBlock - 1) / kBlock);
    p.nch = A + 3;
    p.others_k = g.others_k = 2 + (cfg->observe_other_power ? 1 : 0) + (cfg->observe_other_suppressant ? 1 : 0);
    p.max_steps = g.max_steps = cfg->max_steps;
    p.num_fire_states = g.num_fire_states = cfg->num_fire_states;
    auto flag = [&](int on, uint32_t bit) { p.flags |= on ? bit : 0u; };
    flag(cfg->stochastic_increase, kStochIncrease);
    flag(cfg->stochastic_burnouts, kStochBurnouts);
    flag(cfg->stochastic_decrease, kStochDecrease);
    flag(cfg->use_fire_fuel, kUseFuel);
    flag(cfg->stochastic_suppressant_decrease, kStochSuppDecrease);
    flag(cfg->stochastic_refill, kStochRefill);
    flag(cfg->stochastic_switch, kStochSwitch);
    flag(cfg->stochastic_repair, kStochRepair);
    flag(cfg->stochastic_degrade, kStochDegrade);
    flag(cfg->critical_error, kCritical);
    flag(cfg->show_bad_actions, kShowBad);
    flag(cfg->observe_other_power, kObsPower);
    flag(cfg->observe_other_suppressant, kObsSupp);
    flag(cfg->burnout_penalty_scaled, kPenaltyScaled);
    flag(cfg->localize_putouts, kLocalize);
    flag(cfg->track_cumulative_rewards, kTrackCumulative);
    flag(cfg->max_steps >= 0, kTruncate);
    g.flags = p.flags;
    g.inv_w = (uint32_t)((65536 + W - 1) / W);
    for (int c = 0; c < HW; ++c)
        if ((int)(((uint32_t)c * g.inv_w) >> 16) != c / W) {  // cannot happen for HW <= FRZ_MAX_CELLS; checked, not assumed
            delete env;
            return FRZ_E_INVALID;
        }
    g.inv_others = A > 1 ? (uint32_t)((65536 + A - 2) / (A - 1)) : 0u;
    for (int q = 0; q < A * (A - 1); ++q)
        if ((int)(((uint32_t)q * g.inv_others) >> 16) != q / (A - 1)) {
            delete env;
            return FRZ_E_INVALID;
        }
    g.p_increase = cfg->intensity_increase_probability, g.p_burnout = cfg->burnout_probability, g.p_decrease = cfg->intensity_decrease_probability;
    g.decrease_bonus = cfg->extra_power_decrease_bonus, g.p_supp_decrease = cfg->suppressant_decrease_probability;
    g.p_refill = cfg->suppressant_refill_probability, g.p_switch = cfg->tank_switch_probability, g.p_repair = cfg->repair_probability;
    g.p_degrade = cfg->degrade_probability, g.p_critical = cfg->critical_error_probability;
    g.spread_n = cfg->spread_n, g.spread_w = cfg->spread_w, g.spread_e = cfg->spread_e, g.spread_s = cfg->spread_s;
    g.random_ignition = cfg->random_ignition;
    g.bad_attack_penalty = cfg->bad_attack_penalty, g.burnout_penalty = cfg->burnout_penalty;
    g.termination_reward = cfg->termination_reward, g.termination_kappa = cfg->termination_kappa;
    for (int j = 0; j < FRZ_MAX_CAPACITIES; ++j) g.cum[j] = j < cfg->num_capacities ? cfg->capacity_cumprobs[j] : __builtin_inff();
    g.initial_fuel = cfg->initial_fuel, g.initial_equipment = cfg->initial_equipment_state;
    g.initial_suppressant = cfg->initial_suppressant, g.initial_capacity = cfg->initial_capacity;
    p.initial_fuel = cfg->initial_fuel, p.initial_equipment = cfg->initial_equipment_state;  // (wf_masked_fill_kernel reads the WfDev block)
    p.initial_suppressant = cfg->initial_suppressant, p.initial_capacity = cfg->initial_capacity;

    // tables a lane indexes by agent / equipment state / cell
    WgAgentTable& t = env->agent_table;
    std::memset(&t, 0, sizeof(t));
    std::memcpy(t.power, cfg->fire_reduction_power, sizeof(t.power));
    std::memcpy(t.ay, cfg->agent_y, sizeof(t.ay));
    std::memcpy(t.ax, cfg->agent_x, sizeof(t.ax));
    std::memcpy(t.caps, cfg->possible_capacities, sizeof(t.caps));
    for (int s = 0; s < FRZ_MAX_EQUIPMENT_STATES; ++s)
        for (int j = 0; j < 3; ++j) t.eq[s][j] = cfg->equipment_states[s][j];
    env->cell_tables.assign((size_t)5 * HW, 0);
    for (int c = 0; c < HW; ++c) {
        const int type = cfg->fire_types[c], f0 = cfg->lit[c] ? type : -type;  // wildfire.py:347-351
        std::memcpy(&env->cell_tables[c], &cfg->fire_rewards[c], 4);
        env->cell_tables[(size_t)HW + c] = cfg->ignition_temp[c];
        env->cell_tables[(size_t)2 * HW + c] = f0;
        env->cell_tables[(size_t)3 * HW + c] = cfg->lit[c] ? cfg->ignition_temp[c] : 0;
        env->cell_tables[(size_t)4 * HW + c] = f0 != 0 ? cfg->initial_fuel : 0;
    }
    // in-range cell sets: chebyshev(agent, cell) <= attack_range + equipment range bonus, float32 compare
    // (utils/in_range_check.py:5-23, wildfire.py:604-616)
    env->range_words.assign((size_t)A * S * chunks, 0ull);
    for (int a = 0; a < A; ++a)
        for (int s = 0; s < S; ++s) {
            const float true_range = cfg->attack_range[a] + cfg->equipment_states[s][2];
            for (int c = 0; c < HW; ++c) {
                const int dy = std::abs(cfg->agent_y[a] - c / W), dx = std::abs(cfg->agent_x[a] - c % W);
                if ((float)(dy > dx ? dy : dx) <= true_range) env->range_words[((size_t)a * S + s) * chunks + c / 64] |= 1ull << (c % 64);
            }
        }

    // ---- arena layout
    int r = 0;
    p.r_supp = r, r += A;
    p.r_cap = r, r += A;
    p.r_equip = r, r += A;
    p.r_moves = r++;
    p.r_burnouts = r++;
    p.r_rewards = r, r += A;
    p.r_cum = r, r += A;
    p.r_atc = r, r += A;
    p.r_seeds = r++;
    p.r_mti = r++;
    p.n_rows4 = r;
    p.q_burnouts = 0, p.q_putouts = 1, p.q_etc = 2, p.n_rows8 = 3;
    p.u_term = 0, p.u_trunc = A, p.u_frozen = 2 * A, p.n_rows1 = 2 * A + 1;
    const int64_t cap = B * HW, ok = p.others_k;
    int64_t off = kDevBlockBytes;
    auto take = [&](int64_t bytes) {
        const int64_t here = off;
        off = align_up(off + (bytes > 0 ? bytes : 1), 256);
        return here;
    };
    p.off_rows4 = take((int64_t)p.n_rows4 * B * 4);
    p.off_rows8 = take((int64_t)p.n_rows8 * B * 8);
    p.off_rows1 = take((int64_t)p.n_rows1 * B);
    g.off_cells = take(3 * cap * 4);
    g.off_agent_table = take(sizeof(WgAgentTable));
    g.off_cell_tables = take((int64_t)5 * HW * 4);
    g.off_range = take((int64_t)A * S * chunks * 8);
    g.off_litmap = take((int64_t)chunks * B * 8);
    g.off_okmap = take((int64_t)A * chunks * B * 8);
    g.off_lit_cells = take(B * (int64_t)((HW + 1) & ~1) * 8);
    {   // second copies of the three arrays above, in the same order (steps of odd parity of an overlapped rollout: launch_cpl)
        const int64_t first = g.off_litmap, bytes = off - g.off_litmap;
        env->grid_copy_delta = take(bytes) - first;
    }
    p.off_obs_self = take((int64_t)A * B * 16);
    p.off_obs_others = take((int64_t)A * B * (A - 1) * ok * 4);
    p.off_task_offsets = take((B + 1) * 8);
    p.off_act_offsets = take((int64_t)A * (B + 1) * 8);
    p.off_bad_offsets = take((int64_t)A * (B + 1) * 8);
    p.off_task_values = take(cap * 32);
    p.off_obs_map = take(cap * 8);
    p.off_act_values = take((int64_t)A * cap * 8);
    p.off_bad_values = take(cfg->show_bad_actions ? (int64_t)A * cap * 8 : 256);
    p.off_actions = take((int64_t)A * B * 8);
    p.off_error = take(256);
    p.off_epoch = take(256);
    p.off_metrics = take((int64_t)kMetricBlocks * (FRZ_MAX_AGENTS + 2) * 8);
    p.off_totals = take(2 * kTotalsStride * 4);
    p.off_agg = take((int64_t)p.nchunks * p.nch * 8);
    p.off_prefix = take((int64_t)p.nchunks * p.nch * 8);
    p.off_rand_field = take(3 * cap * 4);
    p.off_rand_agent = take(5 * B * A * 4);
    p.off_mt_state = take(624 * B * 4);
    p.total_bytes = off;
    g.r_supp = p.r_supp, g.r_cap = p.r_cap, g.r_equip = p.r_equip, g.r_moves = p.r_moves, g.r_burnouts = p.r_burnouts, g.r_rewards = p.r_rewards;
    g.r_cum = p.r_cum, g.r_atc = p.r_atc, g.r_seeds = p.r_seeds, g.r_mti = p.r_mti;
    g.q_burnouts = p.q_burnouts, g.q_putouts = p.q_putouts, g.q_etc = p.q_etc;
    g.u_term = p.u_term, g.u_trunc = p.u_trunc, g.u_frozen = p.u_frozen;
    g.off_rows4 = p.off_rows4, g.off_rows8 = p.off_rows8, g.off_rows1 = p.off_rows1;
    g.off_obs_self = p.off_obs_self, g.off_obs_others = p.off_obs_others, g.off_task_values = p.off_task_values;
    g.off_task_offsets = p.off_task_offsets, g.off_obs_map = p.off_obs_map, g.off_act_values = p.off_act_values;
    g.off_act_offsets = p.off_act_offsets, g.off_bad_values = p.off_bad_values, g.off_bad_offsets = p.off_bad_offsets;
    g.off_error = p.off_error, g.off_epoch = p.off_epoch, g.off_totals = p.off_totals, g.off_agg = p.off_agg, g.off_prefix = p.off_prefix;

    int device = 0, cus = 256;
    if (hipGetDevice(&device) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    env->ticketed = p.nchunks > cus;  // wg_offsets_kernel: one workgroup per chunk, handed out in arrival order beyond one per CU
    frz::handle_register(env, 1, A, cfg->parallel_envs, HW);
    *out = env;
    return FRZ_OK;
}

}  // namespace

extern "C" {

int frz_abi_version(void) { return FRZ_ABI_VERSION; }

int frz_wildfire_create(const frz_wildfire_cfg* cfg, frz_wildfire_env** out) {
    if (!cfg || !out) return FRZ_E_INVALID;
    const int H = cfg->grid_height, W = cfg->grid_width, HW = H * W, A = cfg->num_agents;
    if (cfg->parallel_envs <= 0 || H <= 0 || W <= 0 || HW > FRZ_MAX_CELLS || A <= 0 || A > FRZ_MAX_AGENTS) return FRZ_E_INVALID;
    // Kernel family.  Grids above 24 cells, or with more than 8 agents, run one env per wavefront with the cells across its lanes
    // (wildfire_grid.hip: ~500 instructions per env whatever its size); smaller ones one env per lane (field/crew wavefront pairs, or
    // the lane-per-env kernel below, whose cost grows with its unrolled cell x agent loops: 68 us per step at 4 x 5 with 4 agents against
    // 95 us for the wavefront-per-env kernels, 440 us against 121 us at 8 x 8 with 12).  FRZ_WF_KERNEL=grid forces the former.
    const char* family = std::getenv("FRZ_WF_KERNEL");
    if ((family && std::strcmp(family, "grid") == 0) || HW > 24 || A > 8) return create_grid(cfg, out);
    if (cfg->num_equipment_states <= 0 || cfg->num_equipment_states > FRZ_MAX_EQUIPMENT_STATES) return FRZ_E_INVALID;
    if (cfg->num_capacities <= 0 || cfg->num_capacities > FRZ_MAX_CAPACITIES) return FRZ_E_INVALID;
    if (cfg->num_fire_states < 2) return FRZ_E_INVALID;
    if (cfg->initial_equipment_state < 0 || cfg->initial_equipment_state >= cfg->num_equipment_states) return FRZ_E_INVALID;
    if ((int64_t)cfg->parallel_envs * HW >= (int64_t)1 << 31) return FRZ_E_INVALID;  // 32-bit scan channels

    frz_wildfire_env* env = new (std::nothrow) frz_wildfire_env();
    if (!env) return FRZ_E_INVALID;
    env->cfg = *cfg;
    env->variant = kNumVariants - 1;
    for (int i = 0; i < kNumVariants; ++i) {
        const Variant& v = kVariants[i];
        if (v.exact ? (HW == v.cmax && A == v.amax) : (HW <= v.cmax && A <= v.amax)) {
            env->variant = i;
            break;
        }
    }
    env->lane_variant = kNumVariants - 1;
    for (int i = 0; i < kNumVariants; ++i) {
        const Variant& v = kVariants[i];
        if (!v.exact && HW <= v.cmax && A <= v.amax) {
            env->lane_variant = i;
            break;
        }
    }
    if ((int64_t)(3 * HW + 6 * A + 8) * cfg->parallel_envs >= (int64_t)1 << 30) {  // 32-bit row indices
        delete env;
        return FRZ_E_INVALID;
    }

    WfDev& p = env->dev;
    std::memset(&p, 0, sizeof(p));
    const int64_t B = cfg->parallel_envs;
    p.B = cfg->parallel_envs;
    p.H = H;
    p.W = W;
    p.HW = HW;
    p.A = A;
    p.S = cfg->num_equipment_states;
    p.K = cfg->num_capacities;
    // Kernel choice for grids of <= 8 cells: the field/crew wavefront-pair kernel (wildfire_roles.hip, two wavefronts per
    // 64 envs) or the lane-per-env kernel below; FRZ_WF_KERNEL=lane|roles overrides the default.
    const char* want = std::getenv("FRZ_WF_KERNEL");
    const bool small = HW <= 24 && A <= 8;  // shapes the field/crew kernel has an instantiation for (round 4: up to <24, 8>)
    // default: the field/crew kernel up to 16 cells; on 17-24 cells only from six agents on (measured per step at B = 65 536, lane -> field/crew:
    // 3x4 / 5 agents 52 -> 46 us, 4x4 / 6 67 -> 56, 4x6 / 8 99 -> 85, 4x5 / 6 72 -> 70, but 4x6 / 5 73 -> 77, 4x5 / 4 62 -> 68, 3x6 / 3 55 -> 62:
    // tools/dbg/lane_probe.py)
    p.roles = (small && (HW <= 16 || A >= 6)) ? 1 : 0;
    if (want && std::strcmp(want, "lane") == 0) p.roles = 0;
    if (want && std::strcmp(want, "roles") == 0 && small) p.roles = 1;
    const int envs_per_chunk = kBlock;
    p.nchunks = (cfg->parallel_envs + envs_per_chunk - 1) / envs_per_chunk;
    p.nch = A + 3;
    p.others_k = 2 + (cfg->observe_other_power ? 1 : 0) + (cfg->observe_other_suppressant ? 1 : 0);
    p.max_steps = cfg->max_steps;
    p.num_fire_states = cfg->num_fire_states;
    auto flag = [&](int on, uint32_t bit) { p.flags |= on ? bit : 0u; };
    flag(cfg->stochastic_increase, kStochIncrease);
    flag(cfg->stochastic_burnouts, kStochBurnouts);
    flag(cfg->stochastic_decrease, kStochDecrease);
    flag(cfg->use_fire_fuel, kUseFuel);
    flag(cfg->stochastic_suppressant_decrease, kStochSuppDecrease);
    flag(cfg->stochastic_refill, kStochRefill);
    flag(cfg->stochastic_switch, kStochSwitch);
    flag(cfg->stochastic_repair, kStochRepair);
    flag(cfg->stochastic_degrade, kStochDegrade);
    flag(cfg->critical_error, kCritical);
    flag(cfg->show_bad_actions, kShowBad);
    flag(cfg->observe_other_power, kObsPower);
    flag(cfg->observe_other_suppressant, kObsSupp);
    flag(cfg->burnout_penalty_scaled, kPenaltyScaled);
    flag(cfg->localize_putouts, kLocalize);
    flag(cfg->track_cumulative_rewards, kTrackCumulative);
    flag(cfg->max_steps >= 0, kTruncate);
    p.initial_fuel = cfg->initial_fuel;
    p.initial_equipment = cfg->initial_equipment_state;
    p.initial_suppressant = cfg->initial_suppressant;
    p.initial_capacity = cfg->initial_capacity;
    p.p_increase = cfg->intensity_increase_probability;
    p.p_burnout = cfg->burnout_probability;
    p.p_decrease = cfg->intensity_decrease_probability;
    p.decrease_bonus = cfg->extra_power_decrease_bonus;
    p.p_supp_decrease = cfg->suppressant_decrease_probability;
    p.p_refill = cfg->suppressant_refill_probability;
    p.p_switch = cfg->tank_switch_probability;
    p.p_repair = cfg->repair_probability;
    p.p_degrade = cfg->degrade_probability;
    p.p_critical = cfg->critical_error_probability;
    p.spread_n = cfg->spread_n;
    p.spread_w = cfg->spread_w;
    p.spread_e = cfg->spread_e;
    p.spread_s = cfg->spread_s;
    p.random_ignition = cfg->random_ignition;
    p.bad_attack_penalty = cfg->bad_attack_penalty;
    p.burnout_penalty = cfg->burnout_penalty;
    p.termination_reward = cfg->termination_reward;
    p.termination_kappa = cfg->termination_kappa;
    std::memcpy(p.caps, cfg->possible_capacities, sizeof(p.caps));
    for (int j = 0; j < FRZ_MAX_CAPACITIES; ++j) p.cum[j] = j < cfg->num_capacities ? cfg->capacity_cumprobs[j] : __builtin_inff();
    for (int s = 0; s < FRZ_MAX_EQUIPMENT_STATES; ++s)
        for (int j = 0; j < 3; ++j) p.eq[s][j] = cfg->equipment_states[s][j];
    std::memcpy(p.ay, cfg->agent_y, sizeof(p.ay));
    std::memcpy(p.ax, cfg->agent_x, sizeof(p.ax));
    std::memcpy(p.power, cfg->fire_reduction_power, sizeof(p.power));
    std::memcpy(p.fire_rewards, cfg->fire_rewards, sizeof(p.fire_rewards));
    std::memcpy(p.ignition, cfg->ignition_temp, sizeof(p.ignition));
    std::memcpy(p.fire_types, cfg->fire_types, sizeof(p.fire_types));
    std::memcpy(p.lit, cfg->lit, sizeof(p.lit));
    for (int c = 0; c < HW && c < 24; ++c) {  // the configured initial state of a cell (wildfire.py:347-351)
        const int f0 = cfg->lit[c] ? cfg->fire_types[c] : -cfg->fire_types[c];
        p.init_fires[c] = f0;
        p.init_intensity[c] = cfg->lit[c] ? cfg->ignition_temp[c] : 0;
        p.init_fuel[c] = f0 != 0 ? cfg->initial_fuel : 0;
    }
    p.init_equipment = cfg->initial_equipment_state;
    p.init_suppressant = cfg->initial_suppressant;
    p.init_capacity = cfg->initial_capacity;
    for (int c = 0; c < HW; ++c) {
        const int y = c / W, x = c % W;
        p.cell_yx[c] = (y << 16) | x;
        if (y > 0) p.has_n |= 1ull << c;
        if (y < H - 1) p.has_s |= 1ull << c;
        if (x > 0) p.has_w |= 1ull << c;
        if (x < W - 1) p.has_e |= 1ull << c;
    }
    // in-range cell sets: chebyshev(agent, cell) <= attack_range + equipment range bonus, float32 compare
    // (utils/in_range_check.py:5-23, wildfire.py:604-616)
    for (int a = 0; a < A; ++a)
        for (int s = 0; s < cfg->num_equipment_states; ++s) {
            const float true_range = cfg->attack_range[a] + cfg->equipment_states[s][2];
            uint64_t m = 0;
            for (int c = 0; c < HW; ++c) {
                const int dy = std::abs(cfg->agent_y[a] - c / W), dx = std::abs(cfg->agent_x[a] - c % W);
                const int dist = dy > dx ? dy : dx;
                if ((float)dist <= true_range) m |= 1ull << c;
            }
            p.range_mask[a][s] = m;
        }

    // ---- arena layout
    int r = 0;
    p.r_fires = r, r += HW;
    p.r_intensity = r, r += HW;
    p.r_fuel = r, r += HW;
    p.r_supp = r, r += A;
    p.r_cap = r, r += A;
    p.r_equip = r, r += A;
    p.r_moves = r++;
    p.r_burnouts = r++;
    p.r_rewards = r, r += A;
    p.r_cum = r, r += A;
    p.r_atc = r, r += A;
    p.r_seeds = r++;
    p.r_mti = r++;
    p.n_rows4 = r;
    p.q_burnouts = 0, p.q_putouts = 1, p.q_etc = 2, p.n_rows8 = 3;
    p.u_term = 0, p.u_trunc = A, p.u_frozen = 2 * A, p.n_rows1 = 2 * A + 1;
    const int64_t cap = B * HW, ok = p.others_k;
    int64_t off = kDevBlockBytes;
    auto take = [&](int64_t bytes) {
        const int64_t here = off;
        off = align_up(off + bytes, 256);
        return here;
    };
    p.off_rows4 = take((int64_t)p.n_rows4 * B * 4);
    if ((int64_t)p.n_rows4 * B * 4 >= (int64_t)1 << 32) {  // the row blocks are addressed with 32-bit byte offsets
        delete env;
        return FRZ_E_INVALID;
    }
    // the step kernels derive these from (HW, A) and the batch size alone (their loads start before the configuration
    // block is staged): keep both sides in step
    if (p.off_rows4 != kDevBlockBytes || p.r_fires != 0 || p.r_intensity != HW || p.r_fuel != 2 * HW || p.r_supp != 3 * HW ||
        p.r_cap != p.r_supp + A || p.r_equip != p.r_cap + A || p.r_moves != p.r_equip + A || p.r_burnouts != p.r_moves + 1 ||
        p.r_rewards != p.r_moves + 2 || p.r_cum != p.r_rewards + A || p.r_atc != p.r_cum + A || p.r_seeds != p.r_atc + A || p.r_mti != p.r_seeds + 1 ||
        p.u_term != 0 || p.u_trunc != A || p.u_frozen != 2 * A || p.q_burnouts != 0 || p.q_putouts != 1 || p.q_etc != 2) {
        delete env;
        return FRZ_E_INVALID;
    }
    p.off_rows8 = take((int64_t)p.n_rows8 * B * 8);
    p.off_rows1 = take((int64_t)p.n_rows1 * B);
    p.off_obs_self = take((int64_t)A * B * 16);
    p.off_obs_others = take((int64_t)A * B * (A - 1) * ok * 4);
    p.off_task_offsets = take((B + 1) * 8);
    p.off_act_offsets = take((int64_t)A * (B + 1) * 8);
    p.off_bad_offsets = take((int64_t)A * (B + 1) * 8);
    p.off_task_values = take(cap * 32);
    p.off_obs_map = take(cap * 8);
    p.off_act_values = take((int64_t)A * cap * 8);
    p.off_bad_values = take(cfg->show_bad_actions ? (int64_t)A * cap * 8 : 256);
    p.off_actions = take((int64_t)A * B * 8);
    p.off_error = take(256);
    p.off_epoch = take(256);
    p.off_metrics = take((int64_t)kMetricBlocks * (FRZ_MAX_AGENTS + 2) * 8);  // partial rows of frz_wildfire_episode_metrics
    p.off_totals = take(2 * kTotalsStride * 4);
    p.off_agg = take(2 * (int64_t)p.nchunks * p.nch * 8);  // two copies: a multi-step launch double-buffers its chunk sums by the step's parity
    p.off_prefix = take((int64_t)p.nchunks * p.nch * 8);
    p.off_rand_field = take(3 * B * HW * 4);
    p.off_rand_agent = take(5 * B * A * 4);
    p.off_mt_state = take(624 * B * 4);
    // second copy of the packed list buffers (task rows, observation map, action / bad-action maps: contiguous above) for the
    // multi-step launches of the exact field/crew kernels
    // (round 4: the runtime-shape field/crew variants have a multi-step launch too — except <24, 8>, whose scratch would not fit the LDS)
    const bool multi_step = p.roles && (kVariants[env->variant].exact || !(kVariants[env->variant].cmax > 16 && kVariants[env->variant].amax > 4));
    if (multi_step) env->list_copy_delta = take(p.off_actions - p.off_task_values) - p.off_task_values;
    p.total_bytes = off;

    // One workgroup per chunk.  With no more chunks than CUs the whole grid is resident (one 256- or 512-thread workgroup
    // per CU always fits) and chunk = blockIdx.x; otherwise chunks are handed out in arrival order (ticket), which needs no
    // residency assumption: a chunk only waits on chunks whose workgroups have started.
    int device = 0, cus = 256;
    if (hipGetDevice(&device) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    env->ticketed = p.nchunks > cus;
    frz::handle_register(env, 1, A, cfg->parallel_envs, HW);
    *out = env;
    return FRZ_OK;
}

void frz_wildfire_destroy(frz_wildfire_env* env) {
    if (!env) return;
    frz::handle_unregister(env);
    for (hipEvent_t e : env->timing_events) (void)hipEventDestroy(e);
    if (env->scan_done) (void)hipEventDestroy(env->scan_done);
    if (env->lists_done) (void)hipEventDestroy(env->lists_done);
    if (env->side_stream) (void)hipStreamDestroy(env->side_stream);
    delete env;
}

int64_t frz_wildfire_arena_bytes(const frz_wildfire_env* env) { return env ? env->dev.total_bytes : FRZ_E_INVALID; }

int frz_wildfire_bind(frz_wildfire_env* env, void* arena, void* stream) {
    if (!env || !arena) return FRZ_E_INVALID;
    if (reinterpret_cast<uintptr_t>(arena) % 256 != 0) return FRZ_E_INVALID;
    env->arena = static_cast<char*>(arena);
    env->was_reset = false;
    if (env->dev.grid && !env->side_stream) {  // the second stream of overlapped rollouts (never created inside a stream-ordered entry point)
        if (hipStreamCreateWithFlags(&env->side_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&env->scan_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&env->lists_done, hipEventDisableTiming) != hipSuccess)
            return FRZ_E_LAUNCH;
    }
    if (hipMemcpyAsync(arena, &env->dev, sizeof(WfDev), hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)) != hipSuccess)
        return FRZ_E_LAUNCH;
    if (env->dev.grid) {
        hipStream_t s = static_cast<hipStream_t>(stream);
        const WgDev& g = env->gdev;
        if (hipMemcpyAsync(env->arena + g.off_agent_table, &env->agent_table, sizeof(WgAgentTable), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(env->arena + g.off_cell_tables, env->cell_tables.data(), env->cell_tables.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(env->arena + g.off_range, env->range_words.data(), env->range_words.size() * 8, hipMemcpyHostToDevice, s) != hipSuccess)
            return FRZ_E_LAUNCH;
    }
    // the handle owns the pageable source for the life of the copy
    return hipStreamSynchronize(static_cast<hipStream_t>(stream)) == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

int frz_wildfire_get_bufs(const frz_wildfire_env* env, frz_wildfire_bufs* out) {
    if (!env || !out) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    const WfDev& p = env->dev;
    char* a = env->arena;
    const int64_t B = p.B;
    auto row4 = [&](int r) { return a + p.off_rows4 + (int64_t)r * B * 4; };
    auto row8 = [&](int r) { return a + p.off_rows8 + (int64_t)r * B * 8; };
    auto row1 = [&](int r) { return a + p.off_rows1 + (int64_t)r * B; };
    if (p.grid) {  // env-major cell arrays
        const int64_t cells = B * p.HW;
        out->fires = at<int32_t>(a, env->gdev.off_cells);
        out->intensity = out->fires + cells;
        out->fuel = out->intensity + cells;
    } else {
        out->fires = reinterpret_cast<int32_t*>(row4(p.r_fires));
        out->intensity = reinterpret_cast<int32_t*>(row4(p.r_intensity));
        out->fuel = reinterpret_cast<int32_t*>(row4(p.r_fuel));
    }
    out->cells_env_major = p.grid;
    out->suppressants = reinterpret_cast<float*>(row4(p.r_supp));
    out->capacity = reinterpret_cast<float*>(row4(p.r_cap));
    out->equipment = reinterpret_cast<int32_t*>(row4(p.r_equip));
    out->num_moves = reinterpret_cast<int32_t*>(row4(p.r_moves));
    out->num_burnouts = reinterpret_cast<int32_t*>(row4(p.r_burnouts));
    out->rewards = reinterpret_cast<float*>(row4(p.r_rewards));
    out->cumulative_rewards = reinterpret_cast<float*>(row4(p.r_cum));
    out->agent_task_count = reinterpret_cast<int32_t*>(row4(p.r_atc));
    out->seeds = reinterpret_cast<int32_t*>(row4(p.r_seeds));
    out->mt_index = reinterpret_cast<int32_t*>(row4(p.r_mti));
    out->burnouts = reinterpret_cast<int64_t*>(row8(p.q_burnouts));
    out->putouts = reinterpret_cast<int64_t*>(row8(p.q_putouts));
    out->env_task_count = reinterpret_cast<int64_t*>(row8(p.q_etc));
    out->terminations = reinterpret_cast<uint8_t*>(row1(p.u_term));
    out->truncations = reinterpret_cast<uint8_t*>(row1(p.u_trunc));
    out->frozen_scaled = reinterpret_cast<uint8_t*>(row1(p.u_frozen));
    out->obs_self = at<float>(a, p.off_obs_self);
    out->obs_others = at<float>(a, p.off_obs_others);
    out->task_values = at<int64_t>(a, p.off_task_values);
    out->task_offsets = at<int64_t>(a, p.off_task_offsets);
    out->obs_map_values = at<int64_t>(a, p.off_obs_map);
    out->act_map_values = at<int64_t>(a, p.off_act_values);
    out->act_map_offsets = at<int64_t>(a, p.off_act_offsets);
    out->bad_map_values = at<int64_t>(a, p.off_bad_values);
    out->bad_map_offsets = at<int64_t>(a, p.off_bad_offsets);
    out->mt_state = at<uint32_t>(a, p.off_mt_state);
    out->actions = at<int32_t>(a, p.off_actions);
    out->error_flags = at<uint32_t>(a, p.off_error);
    return FRZ_OK;
}

int frz_wildfire_rebuild(frz_wildfire_env* env, void* stream) {
    if (!env) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    env->was_reset = true;
    const WfArgs args{env->arena, nullptr, nullptr, nullptr, &env->dev};
    return launch(env, args, FRZ_RNG_INJECTED, kRebuild, static_cast<hipStream_t>(stream));
}

int frz_wildfire_reset(frz_wildfire_env* env, void* stream) { return frz_wildfire_reset_reseed(env, 0, stream); }

int frz_wildfire_reset_reseed(frz_wildfire_env* env, int32_t seed_increment, void* stream) {
    if (!env) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    env->saved = frz_wildfire_saved_state{};  // a full reset saves the configured initial state again (utils/env.py:140-160)
    if (env->dev.roles || env->dev.grid) {  // the configured initial state is produced inside the rebuild kernel
        WfArgs args{env->arena, nullptr, nullptr, nullptr, &env->dev};
        args.seed_increment = seed_increment;
        const int rc = launch(env, args, FRZ_RNG_INJECTED, kReset, static_cast<hipStream_t>(stream));
        if (rc == FRZ_OK) env->was_reset = true;
        return rc;
    }
    const int blocks = (env->cfg.parallel_envs + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(wf_fill_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), env->arena, seed_increment);
    if (hipGetLastError() != hipSuccess) return FRZ_E_LAUNCH;
    return frz_wildfire_rebuild(env, stream);
}

int frz_mt19937_generate_pair(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events, int64_t count, float* out2, int64_t events2,
                              int64_t count2, int64_t B, void* stream);

int frz_wildfire_random_policy(frz_wildfire_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out, void* stream);

int frz_wildfire_step(frz_wildfire_env* env, const int32_t* actions, int rng_mode, const float* field_randomness,
                      const float* agent_randomness, void* stream) {
    if (!env || !actions) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;  // reset()/rebuild() must precede the first step
    const frz_wildfire_cfg& c = env->cfg;
    const WfDev& p = env->dev;
    WfArgs args{env->arena, actions, nullptr, nullptr, &env->dev};
    if (env->fused_policy) {
        args.policy = true;
        args.policy_seed = env->policy_seed;
        args.policy_step = env->policy_step;
        args.actions_out = env->actions_out;
    }
    if (env->timed) {
        args.start_event = env->start_event;
        args.stop_event = env->stop_event;
    }
    if (rng_mode == FRZ_RNG_INJECTED) {
        if (!field_randomness || !agent_randomness) return FRZ_E_INVALID;
        args.field_rand = field_randomness;
        args.agent_rand = agent_randomness;
    } else if (rng_mode == FRZ_RNG_MT19937 && env->dev.roles && !env->dev.grid &&
               (kVariants[env->variant].exact || (env->rollout_steps > 1 && env->list_copy_delta != 0))) {
        // the field/crew kernel advances the per-env MT19937 streams itself (wildfire_roles.hip; runtime shapes: in their multi-step launch only)
    } else if (rng_mode == FRZ_RNG_MT19937) {
        // per-env MT19937 streams: field draws first, then agent draws (wildfire.py:409-410), staged in the arena
        const int64_t B = c.parallel_envs;
        uint32_t* mt_state = at<uint32_t>(env->arena, p.off_mt_state);
        int32_t* mt_index = at<int32_t>(env->arena, p.off_rows4 + (int64_t)p.r_mti * B * 4);
        float* rf = at<float>(env->arena, p.off_rand_field);
        float* ra = at<float>(env->arena, p.off_rand_agent);
        // (a frozen batch draws nothing: the step launch below will be a no-op, and so is the reference's step then)
        const int rc = frz::mt19937_generate_pair_gated(mt_state, mt_index, rf, 3, (int64_t)c.grid_height * c.grid_width, ra, 5, c.num_agents, B,
                                                        at<uint32_t>(env->arena, p.off_epoch), at<uint32_t>(env->arena, p.off_totals),
                                                        c.num_agents + 1, kTotalsStride, stream);
        if (rc != FRZ_OK) return rc;
        args.field_rand = rf;
        args.agent_rand = ra;
        rng_mode = FRZ_RNG_INJECTED;
    } else if (rng_mode != FRZ_RNG_PHILOX) {
        return FRZ_E_INVALID;
    }
    return launch(env, args, rng_mode, kStep, static_cast<hipStream_t>(stream));
}

int frz_wildfire_step_random_policy(frz_wildfire_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out, int rng_mode,
                                    const float* field_randomness, const float* agent_randomness, void* stream) {
    if (!env || !actions_out) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->dev.roles && !env->dev.grid) {  // no fused kernel for this grid shape: policy launch, then step launch
        const int rc = frz_wildfire_random_policy(env, policy_seed, policy_step, actions_out, stream);
        return rc != FRZ_OK ? rc : frz_wildfire_step(env, actions_out, rng_mode, field_randomness, agent_randomness, stream);
    }
    env->fused_policy = true;
    env->policy_seed = policy_seed;
    env->policy_step = policy_step;
    env->actions_out = actions_out;
    const int rc = frz_wildfire_step(env, actions_out, rng_mode, field_randomness, agent_randomness, stream);
    env->fused_policy = false;
    return rc;
}

int frz_wildfire_timed_rollout(frz_wildfire_env* env, uint64_t policy_seed, uint64_t first_step, int32_t n_steps, int32_t* actions_out,
                               int rng_mode, void* stream, float* kernel_ms) {
    if (!env || !kernel_ms || n_steps <= 0) return FRZ_E_INVALID;
    while ((int)env->timing_events.size() < 2 * n_steps) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return FRZ_E_LAUNCH;
        env->timing_events.push_back(e);
    }
    int rc = FRZ_OK;
    for (int i = 0; i < n_steps && rc == FRZ_OK; ++i) {  // back to back: no host synchronisation between the launches
        env->start_event = env->timing_events[2 * i];
        env->stop_event = env->timing_events[2 * i + 1];
        env->timed = true;
        rc = frz_wildfire_step_random_policy(env, policy_seed, first_step + (uint64_t)i, actions_out, rng_mode, nullptr, nullptr, stream);
        env->timed = false;
    }
    if (rc != FRZ_OK) return rc;
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return FRZ_E_LAUNCH;
    for (int i = 0; i < n_steps; ++i)
        if (hipEventElapsedTime(&kernel_ms[i], env->timing_events[2 * i], env->timing_events[2 * i + 1]) != hipSuccess) return FRZ_E_LAUNCH;
    return FRZ_OK;
}

int frz_exclusive_launch_fits(int64_t workgroups, int workgroups_per_cu, int compute_units, int cu_mask_set) {
    // every workgroup of a multi-step launch must be resident at once: they wait for each other inside the kernel
    if (cu_mask_set) return 0;  // a CU mask shrinks the device without changing the properties the runtime reports
    if (workgroups <= 0 || workgroups_per_cu <= 0 || compute_units <= 0) return 0;
    return workgroups <= (int64_t)workgroups_per_cu * compute_units ? 1 : 0;
}

int frz_wildfire_set_exclusive_device(frz_wildfire_env* env, int exclusive) {
    if (!env) return FRZ_E_INVALID;
    if (!exclusive) {
        env->exclusive_device = false;
        return FRZ_OK;
    }
    if (env->list_copy_delta == 0 || !env->dev.roles || env->dev.grid) {  // no multi-step kernel for this shape: nothing to allow, nothing to guard
        env->exclusive_device = true;
        return FRZ_OK;
    }
    // the caller's promise covers OTHER work on the device; whether this env's own grid fits is checked here: the device that owns the
    // arena (when bound), the occupancy of the multi-step instantiation, no CU mask in force
    int device = 0;
    if (env->arena) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, env->arena) == hipSuccess) device = attr.device;
        else (void)hipGetLastError();
    } else if (hipGetDevice(&device) != hipSuccess) {
        return FRZ_E_NODEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FRZ_E_NODEVICE;
    const int per_cu = roles_persist_occupancy(env->variant);
    const bool masked = std::getenv("ROC_GLOBAL_CU_MASK") != nullptr || std::getenv("HSA_CU_MASK") != nullptr;
    int cus = prop.multiProcessorCount;
    if (const char* assumed = std::getenv("FRZ_ASSUME_COMPUTE_UNITS")) cus = std::atoi(assumed);  // tests: a smaller device than the one at hand
    if (!frz_exclusive_launch_fits(env->dev.nchunks, per_cu, cus, masked ? 1 : 0)) return FRZ_E_INVALID;
    env->exclusive_device = true;
    return FRZ_OK;
}

int frz_wildfire_export_totals(frz_wildfire_env* env, int32_t* staging, void* stream) {
    if (!env || !staging) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    hipLaunchKernelGGL(wf_totals_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), env->arena, staging, 0);
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

int frz_wildfire_import_totals(frz_wildfire_env* env, const int32_t* staging, void* stream) {
    if (!env || !staging) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    hipLaunchKernelGGL(wf_totals_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), env->arena, const_cast<int32_t*>(staging), 1);
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

int frz_wildfire_rollout_launches(const frz_wildfire_env* env, int32_t n_steps, int rng_mode) {
    if (!env || n_steps < 0) return FRZ_E_INVALID;
    bool one = n_steps > 1 && env->exclusive_device && env->list_copy_delta != 0 && !env->ticketed &&
               (rng_mode == FRZ_RNG_PHILOX || rng_mode == FRZ_RNG_INJECTED || rng_mode == FRZ_RNG_MT19937) && env->dev.roles && !env->dev.grid;
    if (one && !kVariants[env->variant].exact) {
        // Runtime-shape variants: the multi-step kernel exists for all of them up to <16, 8>, but it is only CHOSEN where it is the faster way
        // (50-step rollouts at B = 65 536, one launch per step against one launch, us per step — since round 4 a single step of these shapes draws
        // in the kernel too, which halved it: profiles/r04_experiments.txt section 12): Philox <8, 4> 13.6 / 10.4, <16, 4> 20.3 / 24.7, <8, 8>
        // 23.3 / 25.9, <16, 8> 27.2 / 35.4; MT19937 (a single step reads streams staged by a generator launch) <8, 4> 35.6 / 19.0, <16, 4>
        // 65.4 / 46.0, <8, 8> 63.9 / 50.9, <16, 8> 69.1 / 76.4.  FRZ_WF_MULTI_STEP=all takes it wherever it exists (the parity tests do).
        const Variant& v = kVariants[env->variant];
        const bool big = v.cmax > 8 && v.amax > 4;
        const bool faster = rng_mode == FRZ_RNG_INJECTED || (rng_mode == FRZ_RNG_PHILOX && v.cmax <= 8 && v.amax <= 4) || (rng_mode == FRZ_RNG_MT19937 && !big);
        const char* const force = std::getenv("FRZ_WF_MULTI_STEP");
        one = faster || (force && std::strcmp(force, "all") == 0);
    }
    return one ? 1 : n_steps;
}

int frz_wildfire_timed_rollout_launch(frz_wildfire_env* env, uint64_t policy_seed, uint64_t first_step, int32_t n_steps, int32_t* actions_out,
                                      int rng_mode, void* stream, float* launch_ms) {
    if (!env || !launch_ms || frz_wildfire_rollout_launches(env, n_steps, rng_mode) != 1) return FRZ_E_INVALID;
    while ((int)env->timing_events.size() < 2) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return FRZ_E_LAUNCH;
        env->timing_events.push_back(e);
    }
    env->start_event = env->timing_events[0];
    env->stop_event = env->timing_events[1];
    env->timed = true;
    const int rc = frz_wildfire_rollout_random_policy(env, policy_seed, first_step, n_steps, actions_out, rng_mode, stream);
    env->timed = false;
    if (rc != FRZ_OK) return rc;
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return FRZ_E_LAUNCH;
    return hipEventElapsedTime(launch_ms, env->timing_events[0], env->timing_events[1]) == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

int frz_wildfire_rollout_random_policy(frz_wildfire_env* env, uint64_t policy_seed, uint64_t first_step, int32_t n_steps, int32_t* actions_out,
                                       int rng_mode, void* stream) {
    if (!env || !actions_out || n_steps < 0) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;
    if (n_steps == 0) return FRZ_OK;
    // Exact field/crew shapes, Philox draws, the whole grid resident: ONE launch whose workgroups keep their chunk's state in
    // registers from step to step (wildfire_roles.hip, PERSIST); same results as n_steps single-step launches.  Otherwise one launch
    // per step.
    if (frz_wildfire_rollout_launches(env, n_steps, rng_mode) == 1 && n_steps > 1) {
        env->rollout_steps = n_steps;
        const int rc = frz_wildfire_step_random_policy(env, policy_seed, first_step, actions_out, rng_mode, nullptr, nullptr, stream);
        env->rollout_steps = 1;
        return rc;
    }
    for (int32_t t = 0; t < n_steps; ++t) {
        const int rc = frz_wildfire_step_random_policy(env, policy_seed, first_step + (uint64_t)t, actions_out, rng_mode, nullptr, nullptr, stream);
        if (rc != FRZ_OK) return rc;
    }
    return FRZ_OK;
}

int frz_wildfire_rollout_random_policy_metrics(frz_wildfire_env* env, uint64_t policy_seed, uint64_t first_step, int32_t n_steps, int32_t* actions_out,
                                               int rng_mode, double* metrics, void* stream) {
    if (!env || !metrics) return FRZ_E_INVALID;
    if (frz_wildfire_rollout_launches(env, n_steps, rng_mode) == 1 && n_steps > 1) {  // the reductions ride in the multi-step launch's tail
        env->rollout_metrics = metrics;
        const int rc = frz_wildfire_rollout_random_policy(env, policy_seed, first_step, n_steps, actions_out, rng_mode, stream);
        env->rollout_metrics = nullptr;
        return rc;
    }
    const int rc = frz_wildfire_rollout_random_policy(env, policy_seed, first_step, n_steps, actions_out, rng_mode, stream);
    return rc != FRZ_OK ? rc : frz_wildfire_episode_metrics(env, metrics, stream);
}

int frz_wildfire_list_block(const frz_wildfire_env* env, void** block, int64_t* bytes) {
    if (!env || !block || !bytes) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    *block = env->arena + env->dev.off_task_offsets;
    *bytes = env->dev.off_actions - env->dev.off_task_offsets;
    return FRZ_OK;
}

int frz_wildfire_obs_block(const frz_wildfire_env* env, void** block, int64_t* bytes, int64_t* others_offset) {
    if (!env || !block || !bytes || !others_offset) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    *block = env->arena + env->dev.off_obs_self;
    *bytes = env->dev.off_task_offsets - env->dev.off_obs_self;
    *others_offset = env->dev.off_obs_others - env->dev.off_obs_self;
    return FRZ_OK;
}

int frz_wildfire_state_block(const frz_wildfire_env* env, void** cells, int64_t* cells_bytes, void** agents, int64_t* agents_bytes) {
    if (!env || !cells || !cells_bytes || !agents || !agents_bytes) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    const WfDev& p = env->dev;
    const int64_t B = p.B;
    *cells = env->dev.grid ? env->arena + env->gdev.off_cells : env->arena + p.off_rows4 + (int64_t)p.r_fires * B * 4;
    *cells_bytes = (int64_t)3 * p.HW * B * 4;
    *agents = env->arena + p.off_rows4 + (int64_t)p.r_supp * B * 4;
    *agents_bytes = (int64_t)3 * p.A * B * 4;
    return FRZ_OK;
}

int frz_wildfire_reset_masked(frz_wildfire_env* env, const uint8_t* mask, int32_t seed_increment, void* stream) {
    if (!env) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;
    const int blocks = (env->cfg.parallel_envs + kBlock - 1) / kBlock;
    const int64_t off_cells = env->dev.grid ? env->gdev.off_cells : 0, off_tables = env->dev.grid ? env->gdev.off_cell_tables : 0;
    hipLaunchKernelGGL(wf_masked_fill_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), env->arena, mask,
                       (uint32_t)seed_increment, off_cells, off_tables, env->saved);
    if (hipGetLastError() != hipSuccess) return FRZ_E_LAUNCH;
    return frz_wildfire_rebuild(env, stream);
}

int frz_wildfire_set_saved_initial(frz_wildfire_env* env, const frz_wildfire_saved_state* saved) {
    if (!env) return FRZ_E_INVALID;
    if (saved && (!saved->fires || !saved->intensity || !saved->fuel || !saved->suppressants || !saved->capacity || !saved->equipment)) return FRZ_E_INVALID;
    env->saved = saved ? *saved : frz_wildfire_saved_state{};
    return FRZ_OK;
}

int frz_wildfire_rollout(frz_wildfire_env* env, const frz_rollout_spec* spec, void* stream) {
    if (!env || !spec || spec->n_steps < 0) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;
    const WfDev& p = env->dev;
    const int mode = spec->rng_mode;
    const bool policy = spec->action_tape == nullptr;
    const bool auto_reset = (spec->flags & FRZ_ROLLOUT_AUTO_RESET) != 0, reset_first = (spec->flags & FRZ_ROLLOUT_RESET_FIRST) != 0;
    if (mode != FRZ_RNG_INJECTED && mode != FRZ_RNG_PHILOX && mode != FRZ_RNG_MT19937) return FRZ_E_INVALID;
    if (mode == FRZ_RNG_INJECTED && spec->n_steps > 0 && (!spec->randomness_tape_a || !spec->randomness_tape_b)) return FRZ_E_INVALID;
    if (policy && !spec->actions_out) return FRZ_E_INVALID;
    if (auto_reset && mode == FRZ_RNG_MT19937) return FRZ_E_INVALID;  // (an in-kernel re-seed of a 624-word stream; not built)
    if (spec->n_steps == 0) return reset_first ? frz_wildfire_reset_reseed(env, spec->seed_increment, stream) : FRZ_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t B = p.B, A = p.A, HW = p.HW, AB2 = A * B * 2;
    const int64_t block_bytes = p.off_actions - p.off_task_offsets;

    // ONE multi-step launch (wildfire_roles.inl, PERSIST) ...
    const bool mt_reset = reset_first && mode == FRZ_RNG_MT19937;  // the opening reset re-seeds the streams from the host side: separate launches
    if (reset_first) env->saved = frz_wildfire_saved_state{};
    const bool saved_restart = auto_reset && env->saved.fires != nullptr;  // the in-kernel restart fills from the configured tables: not for a saved state
    const bool obs_compact = (spec->flags & FRZ_ROLLOUT_OBS_COMPACT) != 0;
    const bool full_obs_tape = spec->obs_tape != nullptr && !obs_compact;  // whole observation blocks: copied out between the steps' launches
    if (spec->n_steps > 1 && !saved_restart && !full_obs_tape && frz_wildfire_rollout_launches(env, spec->n_steps, mode) == 1) {  // (a one-step rollout takes the per-step path below: it honours every option)
        frz_rollout_spec inner = *spec;
        if (mt_reset) {
            const int rc = frz_wildfire_reset_reseed(env, spec->seed_increment, stream);
            if (rc != FRZ_OK) return rc;
            const int rs = frz_mt19937_seed(at<uint32_t>(env->arena, p.off_mt_state), at<int32_t>(env->arena, p.off_rows4 + (int64_t)p.r_mti * B * 4),
                                            at<int32_t>(env->arena, p.off_rows4 + (int64_t)p.r_seeds * B * 4), nullptr, 0, B, stream);
            if (rs != FRZ_OK) return rs;
            inner.flags &= ~FRZ_ROLLOUT_RESET_FIRST;
        }
        env->rollout_spec = &inner;
        env->rollout_steps = spec->n_steps;
        env->rollout_metrics = spec->metrics;
        int rc;
        if (policy) {
            rc = frz_wildfire_step_random_policy(env, spec->policy_seed, spec->first_step, spec->actions_out, mode, spec->randomness_tape_a,
                                                 spec->randomness_tape_b, stream);
        } else {
            rc = frz_wildfire_step(env, spec->action_tape, mode, spec->randomness_tape_a, spec->randomness_tape_b, stream);
        }
        env->rollout_spec = nullptr;
        env->rollout_steps = 1;
        env->rollout_metrics = nullptr;
        return rc;
    }

    // ... or the same thing step by step (every other shape, shared devices): one launch per step, the records copied out between them
    if (reset_first) {
        const int rc = frz_wildfire_reset_reseed(env, spec->seed_increment, stream);
        if (rc != FRZ_OK) return rc;
        if (mode == FRZ_RNG_MT19937) {
            const int rs = frz_mt19937_seed(at<uint32_t>(env->arena, p.off_mt_state), at<int32_t>(env->arena, p.off_rows4 + (int64_t)p.r_mti * B * 4),
                                            at<int32_t>(env->arena, p.off_rows4 + (int64_t)p.r_seeds * B * 4), nullptr, 0, B, stream);
            if (rs != FRZ_OK) return rs;
        }
    }
    // grid family: the lists of step t beside the env launch of step t + 1 (wildfire_grid.hip launch_cpl) — when nothing between two steps
    // reads the lists (no list record, no restart of finished envs, no observation / state tapes)
    // OFF unless FRZ_WG_OVERLAP=1: measured SLOWER than the two launches back to back on one stream (8x8 / 12 agents: 80 against 70 us per
    // step, 16x16 / 6: 156 against 150) — each step pays two cross-queue hand-overs (~8 us apiece on this part) and the lists launch,
    // fighting the env launch for CUs, takes 39 instead of 20 us; profiles/r04_experiments.txt section 7.  Kept as a tested option.
    const char* const overlap_switch = std::getenv("FRZ_WG_OVERLAP");
    const bool overlap_allowed = overlap_switch && overlap_switch[0] == '1';
    WgOverlap overlap;
    const bool overlapped = overlap_allowed && env->dev.grid && env->side_stream && spec->n_steps > 1 && !auto_reset && !spec->list_record && !spec->obs_tape &&
                            !spec->state_tape;
    if (overlapped) overlap.side = env->side_stream, overlap.scan_done = env->scan_done, overlap.lists_done = env->lists_done, overlap.copy_delta = env->grid_copy_delta;
    struct OverlapGuard {  // the step's stream waits for the last lists launch before anything else is enqueued on it
        frz_wildfire_env* env;
        hipStream_t s;
        bool armed = false;
        ~OverlapGuard() {
            env->overlap = nullptr;
            if (armed) (void)hipStreamWaitEvent(s, env->lists_done, 0);
        }
    } guard{env, s};
    for (int32_t t = 0; t < spec->n_steps; ++t) {
        const float* ra = spec->randomness_tape_a ? spec->randomness_tape_a + (int64_t)t * 3 * B * HW : nullptr;
        const float* rb = spec->randomness_tape_b ? spec->randomness_tape_b + (int64_t)t * 5 * B * A : nullptr;
        if (overlapped) {
            overlap.parity = t & 1;
            overlap.wait_previous_lists = t > 0;
            env->overlap = &overlap;
            guard.armed = true;
        }
        int rc;
        if (policy)
            rc = frz_wildfire_step_random_policy(env, spec->policy_seed, spec->first_step + (uint64_t)t,
                                                 spec->actions_out + (spec->record_actions ? (int64_t)t * AB2 : 0), mode, ra, rb, stream);
        else
            rc = frz_wildfire_step(env, spec->action_tape + (int64_t)t * AB2, mode, ra, rb, stream);
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
        if (!ok) return FRZ_E_LAUNCH;
        if (auto_reset) {
            if (spec->metrics) {  // returns of the envs about to be reset
                const int blocks = p.nchunks < kMetricBlocks ? p.nchunks : kMetricBlocks;
                hipLaunchKernelGGL(wf_metrics_kernel, dim3(blocks), dim3(kBlock), 0, s, env->arena, spec->metrics, 1);
                if (hipGetLastError() != hipSuccess) return FRZ_E_LAUNCH;
            }
            rc = frz_wildfire_reset_masked(env, nullptr, (int32_t)spec->seed_stride, stream);
            if (rc != FRZ_OK) return rc;
        }
        if (spec->list_record && t < spec->n_steps - 1 &&  // the lists as the step (and the reset of its finished envs) left them
            hipMemcpyAsync(static_cast<char*>(spec->list_record) + (int64_t)t * block_bytes, env->arena + p.off_task_offsets, (size_t)block_bytes,
                           hipMemcpyDeviceToDevice, s) != hipSuccess)
            return FRZ_E_LAUNCH;
        if (spec->obs_tape) {  // ... and the observations
            const int64_t supp_bytes = A * B * 4, obs_bytes = p.off_task_offsets - p.off_obs_self;
            const hipError_t e = obs_compact ? hipMemcpyAsync(static_cast<char*>(spec->obs_tape) + (int64_t)t * supp_bytes,
                                                              env->arena + p.off_rows4 + (int64_t)p.r_supp * B * 4, (size_t)supp_bytes, hipMemcpyDeviceToDevice, s)
                                             : hipMemcpyAsync(static_cast<char*>(spec->obs_tape) + (int64_t)t * obs_bytes, env->arena + p.off_obs_self,
                                                              (size_t)obs_bytes, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return FRZ_E_LAUNCH;
        }
        if (spec->state_tape) {  // ... and the state
            void *cells = nullptr, *agents = nullptr;
            int64_t cells_bytes = 0, agents_bytes = 0;
            if (frz_wildfire_state_block(env, &cells, &cells_bytes, &agents, &agents_bytes) != FRZ_OK) return FRZ_E_INVALID;
            char* const dst = static_cast<char*>(spec->state_tape) + (int64_t)t * (cells_bytes + agents_bytes);
            if (hipMemcpyAsync(dst, cells, (size_t)cells_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess ||
                hipMemcpyAsync(dst + cells_bytes, agents, (size_t)agents_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
                return FRZ_E_LAUNCH;
        }
    }
    if (spec->metrics && !auto_reset) return frz_wildfire_episode_metrics(env, spec->metrics, stream);
    return FRZ_OK;
}

int frz_wildfire_timed_rollout_spec(frz_wildfire_env* env, const frz_rollout_spec* spec, void* stream, float* launch_ms) {
    if (!env || !spec || !launch_ms || frz_wildfire_rollout_launches(env, spec->n_steps, spec->rng_mode) != 1) return FRZ_E_INVALID;
    if ((spec->flags & FRZ_ROLLOUT_RESET_FIRST) && spec->rng_mode == FRZ_RNG_MT19937) return FRZ_E_INVALID;  // (then the reset is its own launch)
    while ((int)env->timing_events.size() < 2) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return FRZ_E_LAUNCH;
        env->timing_events.push_back(e);
    }
    env->start_event = env->timing_events[0];
    env->stop_event = env->timing_events[1];
    env->timed = true;
    const int rc = frz_wildfire_rollout(env, spec, stream);
    env->timed = false;
    if (rc != FRZ_OK) return rc;
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return FRZ_E_LAUNCH;
    return hipEventElapsedTime(launch_ms, env->timing_events[0], env->timing_events[1]) == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

int frz_wildfire_episode_metrics(frz_wildfire_env* env, double* metrics, void* stream) {
    if (!env || !metrics) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    const int blocks = env->dev.nchunks < kMetricBlocks ? env->dev.nchunks : kMetricBlocks;
    hipLaunchKernelGGL(wf_metrics_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), env->arena, metrics, 0);
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

int frz_wildfire_random_policy(frz_wildfire_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out, void* stream) {
    if (!env || !actions_out) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    const int64_t n = (int64_t)env->cfg.num_agents * env->cfg.parallel_envs;
    const int blocks = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(wf_policy_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), env->arena,
                       (uint32_t)policy_seed, (uint32_t)(policy_seed >> 32), (uint32_t)policy_step, (uint32_t)(policy_step >> 32),
                       actions_out);
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

}  // extern "C"
