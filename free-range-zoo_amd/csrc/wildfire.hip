// wildfire.hip — fused wildfire environment step for gfx950 (MI355X), one environment per lane.
//
// One launch = one ParallelEnv.step() of the reference for the whole batch:
//   action decode through the open action mapping (wildfire.py:427-483) -> 7 transitions (transitions/*.py)
//   -> rewards / termination (wildfire.py:534-582) -> AEC bookkeeping (utils/env.py:228-235)
//   -> update_observations + update_actions (wildfire.py:586-717) with the variable-length task lists compacted by
//      wavefront prefix sums and a single-pass inter-workgroup prefix hand-off.
//
// Memory: state is struct-of-arrays with the env index innermost (include/frz.h) so every per-field access of a
// wavefront is one contiguous 256-byte segment.  A lane keeps its whole env (grid cells, agents) in registers; the
// lit-fire set and the per-agent attackable sets are 64-bit cell masks.  Nothing is re-read: algorithmic bytes per
// env-step are each touched once.
//
// HBM-bound integer/byte work: no MFMA.  Built with -ffp-contract=off (float32 ops round once, like eager torch).
#include "frz_device.h"

#include "../../include/frz.h"

#include <cstdio>
#include <cstring>
#include <new>

namespace {

using frz::kBlock;

constexpr int kMaxChannels = FRZ_MAX_AGENTS + 3;  // F, F_a..., not-terminated, not-truncated
constexpr int kTotalsStride = 32;                 // uint32 words per totals slot (128 B)

enum Mode { kStep = 0, kRebuild = 1 };

struct WfParams {
    frz_wildfire_bufs buf;
    // workspace carve-up
    uint32_t* epoch;    // [1]   launch epoch, bumped by the workgroup that owns the last chunk
    uint32_t* totals;   // [2][kTotalsStride] batch totals of the scan channels, slot = epoch & 1
    uint64_t* agg;      // [nchunks][nch]  per-chunk channel sums   (granules)
    uint64_t* prefix;   // [nchunks][nch]  inclusive prefix after a round's last chunk (granules)
    const int32_t* actions;
    const float* field_rand;
    const float* agent_rand;
    int B, H, W, HW, A, S, K, nchunks, nch, others_k, max_steps, num_fire_states;
    int stochastic_increase, stochastic_burnouts, stochastic_decrease, use_fire_fuel, stochastic_supp_decrease, stochastic_refill,
        stochastic_switch, stochastic_repair, stochastic_degrade, critical_error, show_bad_actions, observe_other_power,
        observe_other_suppressant, burnout_penalty_scaled, localize_putouts, track_cumulative;
    float p_increase, p_burnout, p_decrease, decrease_bonus, p_supp_decrease, p_refill, p_switch, p_repair, p_degrade, p_critical;
    float spread_n, spread_w, spread_e, spread_s, random_ignition;
    float bad_attack_penalty, burnout_penalty, termination_reward, termination_kappa;
    float eq[FRZ_MAX_EQUIPMENT_STATES][3];
    float caps[FRZ_MAX_CAPACITIES], cum[FRZ_MAX_CAPACITIES];
    int ay[FRZ_MAX_AGENTS], ax[FRZ_MAX_AGENTS];
    float power[FRZ_MAX_AGENTS];
    uint64_t range_mask[FRZ_MAX_AGENTS][FRZ_MAX_EQUIPMENT_STATES];  // cells agent a reaches at equipment state s
    uint64_t has_n, has_w, has_e, has_s;                            // cells that have a north/west/east/south neighbour
    float fire_rewards[FRZ_MAX_CELLS];
    int ignition[FRZ_MAX_CELLS];
    int cell_yx[FRZ_MAX_CELLS];  // (y << 16) | x
};

struct FillParams {
    frz_wildfire_bufs buf;
    int B, HW, A, initial_fuel, initial_equipment;
    float initial_suppressant, initial_capacity;
    int fire_types[FRZ_MAX_CELLS], lit[FRZ_MAX_CELLS], ignition[FRZ_MAX_CELLS];
};

__device__ __forceinline__ float clamp01(float p) {
    p = p < 0.0f ? 0.0f : p;
    return p > 1.0f ? 1.0f : p;
}

template <int N>
__device__ __forceinline__ float table3(const float (&tab)[N][3], int n, int index, int col) {
    float v = tab[0][col];
#pragma unroll
    for (int s = 1; s < N; ++s)
        if (s < n) v = index == s ? tab[s][col] : v;
    return v;
}

__device__ __forceinline__ uint64_t mask_lookup(const uint64_t (&tab)[FRZ_MAX_EQUIPMENT_STATES], int n, int index) {
    uint64_t v = tab[0];
#pragma unroll
    for (int s = 1; s < FRZ_MAX_EQUIPMENT_STATES; ++s)
        if (s < n) v = index == s ? tab[s] : v;
    return v;
}

// wildfire.py:347-354 + utils/env.py:137-160: state from the configuration, bookkeeping zeroed.
__global__ void __launch_bounds__(kBlock) wf_fill_kernel(const FillParams p) {
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (b >= p.B) return;
    const int64_t B = p.B;
    for (int c = 0; c < p.HW; ++c) {
        const int type = p.fire_types[c];
        const int f = p.lit[c] ? type : -type;
        p.buf.fires[c * B + b] = f;
        p.buf.intensity[c * B + b] = p.lit[c] ? p.ignition[c] : 0;
        p.buf.fuel[c * B + b] = f != 0 ? p.initial_fuel : 0;
    }
    for (int a = 0; a < p.A; ++a) {
        p.buf.suppressants[a * B + b] = p.initial_suppressant;
        p.buf.capacity[a * B + b] = p.initial_capacity;
        p.buf.equipment[a * B + b] = p.initial_equipment;
        p.buf.rewards[a * B + b] = 0.0f;
        if (p.buf.cumulative_rewards) p.buf.cumulative_rewards[a * B + b] = 0.0f;
        p.buf.terminations[a * B + b] = 0;
        p.buf.truncations[a * B + b] = 0;
    }
    p.buf.num_moves[b] = 0;
    p.buf.num_burnouts[b] = 0;
    p.buf.burnouts[b] = 0;
    p.buf.putouts[b] = 0;
    p.buf.frozen_scaled[b] = 0;
}

template <int CMAX, int AMAX, int RNG, int MODE>
__global__ void __launch_bounds__(kBlock) wf_step_kernel(const WfParams p) {
    __shared__ uint64_t s_wave_scan[frz::kWaves][(AMAX + 1 + 3) / 4];
    __shared__ uint32_t s_wave_live[frz::kWaves][2];
    __shared__ uint32_t s_reduce[frz::kWaves][32];
    __shared__ uint32_t s_prefix[32];

    constexpr int PW = (AMAX + 1 + 3) / 4;  // packed scan words (four 16-bit channels each)
    constexpr int NCHP = AMAX + 3 <= 8 ? 8 : (AMAX + 3 <= 16 ? 16 : 32);  // channels padded to a power of two
    const int tid = threadIdx.x, lane = frz::lane_id(), wave = frz::wave_id();
    const int64_t B = p.B;
    const int HW = p.HW, A = p.A, W = p.W;
    const int nch = p.nch;  // A + 3
    const int ch_nt = A + 1, ch_ntr = A + 2;

    const uint32_t epoch = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t tag = epoch + 1u;  // never 0 on a zero-filled workspace
    const uint32_t* prev = p.totals + ((epoch + 1u) & 1u) * kTotalsStride;
    uint32_t* cur = p.totals + (epoch & 1u) * kTotalsStride;

    // utils/env.py:211-213 — every per-agent step() is a no-op once ALL envs are terminated or ALL are truncated.
    bool frozen = false;
    if (MODE == kStep) frozen = prev[ch_nt] == 0u || prev[ch_ntr] == 0u;

    for (int chunk = blockIdx.x; chunk < p.nchunks; chunk += gridDim.x) {
        const int64_t b = (int64_t)chunk * kBlock + tid;
        const bool active = b < B;

        if (frozen) {
            // The parallel adapter (utils/conversions.py:87-90) then adds the stale aec rewards once per agent call.
            if (active && !p.buf.frozen_scaled[b]) {
                for (int a = 0; a < A; ++a) {
                    const float r = p.buf.rewards[a * B + b];
                    float acc = 0.0f;
                    for (int k = 0; k < A; ++k) acc = acc + r;
                    p.buf.rewards[a * B + b] = acc;
                }
                p.buf.frozen_scaled[b] = 1;
            }
            continue;
        }

        // ------------------------------------------------------------------------------------------ load state
        int f[CMAX], in[CMAX], fu[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            const bool on = active && c < HW;
            f[c] = on ? p.buf.fires[c * B + b] : 0;
            in[c] = on ? p.buf.intensity[c * B + b] : 0;
            fu[c] = on ? p.buf.fuel[c * B + b] : 0;
        }
        float supp[AMAX], capa[AMAX];
        int eqs[AMAX];
        uint8_t term[AMAX];
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            const bool on = active && a < A;
            supp[a] = on ? p.buf.suppressants[a * B + b] : 0.0f;
            capa[a] = on ? p.buf.capacity[a * B + b] : 0.0f;
            eqs[a] = on ? p.buf.equipment[a * B + b] : 0;
            term[a] = on ? p.buf.terminations[a * B + b] : (uint8_t)1;
        }
        uint8_t trunc0 = active ? p.buf.truncations[b] : (uint8_t)1;

        float rew[AMAX];
        uint32_t err = 0;

        if (MODE == kStep) {
            int act_idx[AMAX], act_id[AMAX];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                const bool on = active && a < A;
                const int2 v = on ? reinterpret_cast<const int2*>(p.actions)[a * B + b] : make_int2(0, -1);
                act_idx[a] = v.x;
                act_id[a] = v.y;
            }
            int nm = active ? p.buf.num_moves[b] : 0;
            int nb = active ? p.buf.num_burnouts[b] : 0;

            // ---------------------------------------------------------------------------------- randomness
            float r_field[3][CMAX], r_agent[5][AMAX];
            if (RNG == FRZ_RNG_INJECTED) {
#pragma unroll
                for (int e = 0; e < 3; ++e)
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) r_field[e][c] = (active && c < HW) ? p.field_rand[((int64_t)e * B + b) * HW + c] : 1.0f;
#pragma unroll
                for (int e = 0; e < 5; ++e)
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) r_agent[e][a] = (active && a < A) ? p.agent_rand[((int64_t)e * B + b) * A + a] : 1.0f;
            } else {
                // FRZ_RNG_PHILOX: stream e = field event e (draw = cell), stream 3 + e = agent event e (draw = agent)
                const uint32_t seed = active ? (uint32_t)p.buf.seeds[b] : 0u;
#pragma unroll
                for (int e = 0; e < 3; ++e)
#pragma unroll
                    for (int q = 0; q < (CMAX + 3) / 4; ++q) {
                        if (q * 4 < HW) {
                            const frz::Philox4 w = frz::philox4x32_10((uint32_t)q, (uint32_t)nm, (uint32_t)e, 0u, seed, 0x46525A00u);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (q * 4 + j < CMAX) r_field[e][q * 4 + j] = frz::u32_to_unit_float(w.w[j]);
                        }
                    }
#pragma unroll
                for (int e = 0; e < 5; ++e)
#pragma unroll
                    for (int q = 0; q < (AMAX + 3) / 4; ++q) {
                        if (q * 4 < A) {
                            const frz::Philox4 w = frz::philox4x32_10((uint32_t)q, (uint32_t)nm, (uint32_t)(3 + e), 0u, seed, 0x46525A00u);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (q * 4 + j < AMAX) r_agent[e][q * 4 + j] = frz::u32_to_unit_float(w.w[j]);
                        }
                    }
            }

            // ------------------------------------------------------------- action decode (wildfire.py:427-483)
            // The action mapping of the previous rebuild is a pure function of the state it was built from, which is
            // the state just loaded: attackable set of agent a = lit fires within its (equipment-adjusted) range,
            // non-empty only while it has suppressant (wildfire.py:604-623).
            uint64_t lit0 = 0;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) lit0 |= (uint64_t)(f[c] > 0) << c;

            float ap[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) ap[c] = 0.0f;
            bool users[AMAX], refill[AMAX];
            int hit[AMAX];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                users[a] = false;
                refill[a] = false;
                hit[a] = -1;
                rew[a] = 0.0f;
                if (a < A) {
                    const uint64_t ok = supp[a] > 0.0f ? (lit0 & mask_lookup(p.range_mask[a], p.S, eqs[a])) : 0ull;
                    refill[a] = act_id[a] == -1;
                    // quirk wildfire.py:434-435: an agent with no attackable task in ANY env of the batch is skipped
                    const bool skipped = prev[1 + a] == 0u;
                    const bool fight = active && !refill[a] && !skipped;
                    const uint64_t sel = p.show_bad_actions ? lit0 : ok;
                    const bool valid = act_idx[a] >= 0 && act_idx[a] < __popcll(sel);
                    int target = 0, seen = 0;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        const int bit = (int)((sel >> c) & 1ull);
                        target = (bit && seen == act_idx[a]) ? c : target;
                        seen += bit;
                    }
                    const bool attackable = ((ok >> target) & 1ull) != 0ull;
                    const bool good = fight && valid && (!p.show_bad_actions || attackable);
                    if (fight && !valid) err |= FRZ_ERR_BAD_ACTION_INDEX;
                    const float power = p.power[a] + table3(p.eq, p.S, eqs[a], 1);
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) ap[c] = ap[c] + ((good && target == c) ? power : 0.0f);  // agent order
                    users[a] = good;
                    hit[a] = good ? target : -1;
                    rew[a] = (fight && !good) ? p.bad_attack_penalty : 0.0f;  // assignment, :477
                }
            }

            // ---------------------------------------------- agent transitions (suppressant/equipment/capacity)
            bool increased[AMAX];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                increased[a] = false;
                if (a < A) {
                    // transitions/suppressant_decrease.py:56-61
                    const bool dec = users[a] && (!p.stochastic_supp_decrease || r_agent[0][a] < p.p_supp_decrease);
                    float s = dec ? supp[a] - 1.0f : supp[a];
                    s = s < 0.0f ? 0.0f : s;
                    // transitions/equipment.py:51-75 (masks from the value before any write)
                    const int e0 = eqs[a], top = p.S - 1;
                    const bool pristine = e0 == top, damaged = e0 == 0, inter = !pristine && !damaged;
                    const float r1 = r_agent[1][a];
                    const bool repairs = p.stochastic_repair ? (damaged && r1 < p.p_repair) : damaged;
                    const bool crit = p.critical_error && pristine && r1 < p.p_critical;
                    bool degr = p.stochastic_degrade ? ((pristine || inter) && r1 < p.p_degrade) : (inter || pristine);
                    degr = degr && !crit;
                    int e = repairs ? top : e0;
                    e = crit ? 0 : e;
                    e = degr ? e - 1 : e;
                    // transitions/suppressant_refill.py:63-70 (bonus from the NEW equipment state)
                    const bool inc = refill[a] && (!p.stochastic_refill || r_agent[2][a] < p.p_refill);
                    s = inc ? capa[a] + table3(p.eq, p.S, e, 0) : s;
                    // transitions/capacity.py:52-64
                    int ci = 0;
#pragma unroll
                    for (int k = 0; k < FRZ_MAX_CAPACITIES; ++k)
                        if (k < p.K) ci += r_agent[3][a] > p.cum[k] ? 1 : 0;
                    ci = ci > p.K - 1 ? p.K - 1 : ci;
                    float new_max = p.caps[0];
#pragma unroll
                    for (int k = 1; k < FRZ_MAX_CAPACITIES; ++k)
                        if (k < p.K) new_max = ci == k ? p.caps[k] : new_max;
                    const bool sw = inc && (!p.stochastic_switch || r_agent[4][a] < p.p_switch);
                    const float bonus = s - capa[a];
                    capa[a] = sw ? new_max : capa[a];
                    s = sw ? new_max + bonus : s;
                    supp[a] = s;
                    eqs[a] = e;
                    increased[a] = inc;
                }
            }
            (void)increased;

            // ------------------------------------------------------------ fire increase / decrease per cell
            uint64_t burned = 0, put_out = 0, lit2 = 0;
            const int almost_state = p.num_fire_states - 2, burnout_state = p.num_fire_states - 1;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                if (c < HW) {
                    // transitions/fire_increase.py:61-91
                    {
                        const int required = f[c] >= 0 ? f[c] : 0;
                        const float diff = (float)required - ap[c];
                        const bool lit = f[c] > 0 && in[c] > 0;
                        const bool unmet = diff > 0.0f && lit;
                        const bool almost = unmet && in[c] == almost_state;
                        const bool increasing = unmet && !almost;
                        float prob = 0.0f;
                        prob = increasing ? (p.stochastic_increase ? p.p_increase : 1.0f) : prob;
                        prob = almost ? (p.stochastic_burnouts ? p.p_burnout : p.p_increase) : prob;
                        prob = clamp01(prob);
                        const bool inc = r_field[0][c] < prob;
                        in[c] += inc ? 1 : 0;
                        const bool bo = inc && in[c] >= burnout_state;
                        f[c] = bo ? -f[c] : f[c];
                        fu[c] = bo ? (fu[c] - 1 < 0 ? 0 : fu[c] - 1) : fu[c];
                        burned |= (uint64_t)bo << c;
                    }
                    // transitions/fire_decrease.py:56-77: p = p_dec + ((-1 * diff) * bonus), each op rounded
                    {
                        const int required = f[c] >= 0 ? f[c] : 0;
                        const float diff = (float)required - ap[c];
                        const bool lit = f[c] > 0 && in[c] > 0;
                        const bool met = diff <= 0.0f && lit;
                        const float stoch_p = __fadd_rn(p.p_decrease, __fmul_rn(__fmul_rn(-1.0f, diff), p.decrease_bonus));
                        float prob = met ? (p.stochastic_decrease ? stoch_p : 1.0f) : 0.0f;
                        prob = clamp01(prob);
                        const bool dec = r_field[1][c] < prob;
                        in[c] -= dec ? 1 : 0;
                        const bool po = dec && in[c] <= 0;
                        f[c] = po ? -f[c] : f[c];
                        fu[c] = po ? fu[c] - 1 : fu[c];  // unclamped, :75
                        put_out |= (uint64_t)po << c;
                    }
                    lit2 |= (uint64_t)(f[c] > 0 && in[c] > 0) << c;
                }
            }
            // ---------------------------------------- fire spread stencil (transitions/fire_spreads.py:44-57)
            {
                const uint64_t from_n = (lit2 << W) & p.has_n, from_s = (lit2 >> W) & p.has_s;
                const uint64_t from_w = (lit2 << 1) & p.has_w, from_e = (lit2 >> 1) & p.has_e;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    if (c < HW) {
                        float prob = 0.0f;  // conv2d accumulation order: N, W, E, S
                        prob = __fadd_rn(prob, ((from_n >> c) & 1ull) ? p.spread_n : 0.0f);
                        prob = __fadd_rn(prob, ((from_w >> c) & 1ull) ? p.spread_w : 0.0f);
                        prob = __fadd_rn(prob, ((from_e >> c) & 1ull) ? p.spread_e : 0.0f);
                        prob = __fadd_rn(prob, ((from_s >> c) & 1ull) ? p.spread_s : 0.0f);
                        bool unlit = f[c] < 0 && in[c] == 0;
                        unlit = unlit && (!p.use_fire_fuel || fu[c] > 0);
                        prob = unlit ? __fadd_rn(prob, p.random_ignition) : 0.0f;
                        const bool spread = r_field[2][c] < prob;
                        f[c] = spread ? -f[c] : f[c];
                        in[c] = spread ? p.ignition[c] : in[c];
                    }
                }
            }

            // -------------------------------------------------- rewards and termination (wildfire.py:534-582)
            float fire_reward_sum = 0.0f, burnout_total = 0.0f;
            int fuel_sum = 0;
            bool any_fire = false;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                if (c < HW) {
                    fire_reward_sum = __fadd_rn(fire_reward_sum, ((put_out >> c) & 1ull) ? p.fire_rewards[c] : 0.0f);
                    const float pen = p.burnout_penalty_scaled ? __fmul_rn(-1.0f, p.fire_rewards[c]) : p.burnout_penalty;
                    burnout_total = __fadd_rn(burnout_total, ((burned >> c) & 1ull) ? pen : 0.0f);
                    fuel_sum += fu[c];
                    any_fire = any_fire || f[c] > 0;
                }
            }
            bool dead = !any_fire;
            if (p.use_fire_fuel) dead = dead && fuel_sum <= 0;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) f[c] = dead ? 0 : f[c];  // :570
            bool terminated = true;
#pragma unroll
            for (int a = 0; a < AMAX; ++a)
                if (a < A) terminated = terminated && term[a] != 0;
            const bool newly = !terminated && dead;
            // correctly rounded float32 log via double (matches the oracle bit for bit; the reference's torch.log is
            // a <=1-ulp float32 log).  Only evaluated by wavefronts that hold a newly terminated env.
            float log_burnouts = 0.0f;
            if (newly && p.termination_kappa != 0.0f) log_burnouts = (float)log((double)nb + 1.0);
            const float penalty = __fmul_rn(p.termination_kappa, log_burnouts);
            float term_reward = __fsub_rn(p.termination_reward, penalty);
            term_reward = term_reward < 0.0f ? 0.0f : term_reward;
            const int n_burn = __popcll(burned), n_put = __popcll(put_out);
            nb += n_burn;
            nm += 1;
            const bool truncated = p.max_steps >= 0 ? nm >= p.max_steps : trunc0 != 0;
            trunc0 = (uint8_t)truncated;

#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (a < A) {
                    float add;
                    if (p.localize_putouts) {
                        float mine = 0.0f;
#pragma unroll
                        for (int c = 0; c < CMAX; ++c) mine = (hit[a] == c && ((put_out >> c) & 1ull)) ? p.fire_rewards[c] : mine;
                        add = __fadd_rn(mine, burnout_total);
                    } else {
                        add = __fadd_rn(fire_reward_sum, burnout_total);
                    }
                    rew[a] = __fadd_rn(rew[a], add);
                    rew[a] = newly ? __fadd_rn(rew[a], term_reward) : rew[a];
                    term[a] = (uint8_t)(term[a] | (dead ? 1 : 0));
                }
            }

            // ------------------------------------------------------------------------ dense stores (state)
            if (active) {
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < HW) {
                        p.buf.fires[c * B + b] = f[c];
                        p.buf.intensity[c * B + b] = in[c];
                        p.buf.fuel[c * B + b] = fu[c];
                    }
#pragma unroll
                for (int a = 0; a < AMAX; ++a)
                    if (a < A) {
                        p.buf.suppressants[a * B + b] = supp[a];
                        p.buf.capacity[a * B + b] = capa[a];
                        p.buf.equipment[a * B + b] = eqs[a];
                        p.buf.rewards[a * B + b] = rew[a];
                        p.buf.terminations[a * B + b] = term[a];
                        if (p.max_steps >= 0) p.buf.truncations[a * B + b] = trunc0;
                        if (p.track_cumulative) p.buf.cumulative_rewards[a * B + b] = __fadd_rn(p.buf.cumulative_rewards[a * B + b], rew[a]);
                    }
                p.buf.num_moves[b] = nm;
                p.buf.num_burnouts[b] = nb;
                p.buf.burnouts[b] = n_burn;
                p.buf.putouts[b] = n_put;
            }
        }

        // ======================================================================================================
        // update_observations + update_actions on the new state (wildfire.py:586-717)
        // ======================================================================================================
        uint64_t lit1 = 0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) lit1 |= (uint64_t)(f[c] > 0) << c;
        uint64_t ok1[AMAX];
        uint64_t packed[PW];
#pragma unroll
        for (int w = 0; w < PW; ++w) packed[w] = 0;
        const int F = __popcll(lit1);
        packed[0] = (uint64_t)F;
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            ok1[a] = 0;
            if (a < A) {
                ok1[a] = supp[a] > 0.0f ? (lit1 & mask_lookup(p.range_mask[a], p.S, eqs[a])) : 0ull;
                packed[(a + 1) >> 2] |= (uint64_t)__popcll(ok1[a]) << (16 * ((a + 1) & 3));
            }
        }

        // workgroup-exclusive prefix of the per-env counts (four 16-bit channels per word)
        uint64_t incl[PW];
#pragma unroll
        for (int w = 0; w < PW; ++w) incl[w] = frz::wave_inclusive_scan(packed[w]);
        const bool alive = active && !(term[0] != 0);  // agents share one termination value (wildfire.py:579)
        const uint32_t live_nt = (uint32_t)__popcll(__ballot(alive));
        const uint32_t live_ntr = (uint32_t)__popcll(__ballot(active && trunc0 == 0));
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
            for (int k = 0; k < frz::kWaves; ++k) {
                const uint64_t t = s_wave_scan[k][w];
                base[w] += k < wave ? t : 0ull;
                block_total[w] += t;
            }
        }

        // publish this chunk's channel sums, then (while the hand-off is in flight) do the dense observation stores
        const int round_first = chunk - blockIdx.x;  // first chunk of this round
        if (tid < nch) {
            uint32_t v;
            if (tid <= A)
                v = (uint32_t)((block_total[tid >> 2] >> (16 * (tid & 3))) & 0xFFFFull);
            else {
                v = 0;
                const int which = tid - ch_nt;
#pragma unroll
                for (int k = 0; k < frz::kWaves; ++k) v += s_wave_live[k][which];
            }
            frz::granule_store(p.agg + (int64_t)chunk * nch + tid, tag, v);
        }

        if (active) {
            // agent observations (wildfire.py:677-681, 704-716)
#pragma unroll
            for (int a = 0; a < AMAX; ++a)
                if (a < A) {
                    reinterpret_cast<float4*>(p.buf.obs_self)[a * B + b] =
                        make_float4((float)p.ay[a], (float)p.ax[a], p.power[a], supp[a]);
                    float* others = p.buf.obs_others + (a * B + b) * (int64_t)((A - 1) * p.others_k);
                    int col = 0;
#pragma unroll
                    for (int o = 0; o < AMAX; ++o)
                        if (o < A && o != a) {
                            others[col++] = (float)p.ay[o];
                            others[col++] = (float)p.ax[o];
                            if (p.observe_other_power) others[col++] = p.power[o];
                            if (p.observe_other_suppressant) others[col++] = supp[o];
                        }
                    p.buf.agent_task_count[a * B + b] = __popcll(ok1[a]);
                }
            p.buf.env_task_count[b] = F;
        }

        // -------------------------------------------------- inter-workgroup exclusive prefix (single pass)
        // chunk j needs sum of channel sums of all chunks < j: the chunks of this round that precede it (their
        // workgroups are co-resident and have published or are about to) + the inclusive prefix the previous
        // round's last chunk published.
        bool timed_out = false;
        uint32_t acc = 0;
        {
            // thread t sums channel (t % NCHP) over predecessors t / NCHP, t / NCHP + PP, ...; loads are issued in
            // batches so one L2 round trip covers the whole window when the granules are already published
            const int ch = tid & (NCHP - 1), slot = tid / NCHP;
            constexpr int PP = kBlock / NCHP, UNR = 8;
            for (int first = round_first; first < chunk; first += PP * UNR) {
                uint64_t g[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int pred = first + u * PP + slot;
                    g[u] = (pred < chunk && ch < nch) ? frz::granule_load(p.agg + (int64_t)pred * nch + ch) : ((uint64_t)tag << 32);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    uint32_t v = (uint32_t)g[u];
                    if ((uint32_t)(g[u] >> 32) != tag) v = frz::granule_wait(p.agg + (int64_t)(first + u * PP + slot) * nch + ch, tag, &timed_out);
                    acc += v;
                }
            }
            if (round_first > 0 && tid < nch) acc += frz::granule_wait(p.prefix + (int64_t)(round_first - 1) * nch + tid, tag, &timed_out);
#pragma unroll
            for (int d = NCHP; d < 64; d <<= 1) acc += __shfl_xor(acc, d, 64);
            if (lane < NCHP) s_reduce[wave][lane] = acc;
        }
        __syncthreads();
        if (tid < nch) {
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < frz::kWaves; ++k) s += s_reduce[k][tid];
            s_prefix[tid] = s;
        }
        __syncthreads();
        if (timed_out) err |= FRZ_ERR_SCAN_TIMEOUT;

        const bool round_last = blockIdx.x == gridDim.x - 1 || chunk == p.nchunks - 1;
        if (round_last && tid < nch) {
            uint32_t mine;
            if (tid <= A)
                mine = (uint32_t)((block_total[tid >> 2] >> (16 * (tid & 3))) & 0xFFFFull);
            else {
                mine = 0;
#pragma unroll
                for (int k = 0; k < frz::kWaves; ++k) mine += s_wave_live[k][tid - ch_nt];
            }
            const uint32_t inclusive = s_prefix[tid] + mine;
            frz::granule_store(p.prefix + (int64_t)chunk * nch + tid, tag, inclusive);
            if (chunk == p.nchunks - 1) cur[tid] = inclusive;  // batch totals, read by the next launch
        }

        // ------------------------------------------------------------------ jagged stores (values + offsets)
        if (active) {
            const int64_t cap = B * HW;
            const int64_t off_f = (int64_t)s_prefix[0] + (int64_t)(((base[0] + incl[0] - packed[0]) >> 0) & 0xFFFFull);
            p.buf.task_offsets[b] = off_f;
            if (b == B - 1) p.buf.task_offsets[B] = off_f + F;
            int r = 0;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                if ((lit1 >> c) & 1ull) {
                    int64_t* row = p.buf.task_values + (off_f + r) * 4;
                    const int yx = p.cell_yx[c];
                    reinterpret_cast<longlong2*>(row)[0] = make_longlong2(yx >> 16, yx & 0xFFFF);
                    reinterpret_cast<longlong2*>(row)[1] = make_longlong2(f[c], in[c]);
                    p.buf.obs_map_values[off_f + r] = r;
                    ++r;
                }
            }
#pragma unroll
            for (int a = 0; a < AMAX; ++a)
                if (a < A) {
                    const int w = (a + 1) >> 2, sh = 16 * ((a + 1) & 3);
                    const int64_t off_a = (int64_t)s_prefix[a + 1] + (int64_t)(((base[w] + incl[w] - packed[w]) >> sh) & 0xFFFFull);
                    const int fa = __popcll(ok1[a]);
                    p.buf.act_map_offsets[a * (B + 1) + b] = off_a;
                    if (b == B - 1) p.buf.act_map_offsets[a * (B + 1) + B] = off_a + fa;
                    int64_t* av = p.buf.act_map_values + a * cap + off_a;
                    int64_t* bv = nullptr;
                    if (p.show_bad_actions) {
                        const int64_t off_bad = off_f - off_a;  // bad = listed but not attackable
                        p.buf.bad_map_offsets[a * (B + 1) + b] = off_bad;
                        if (b == B - 1) p.buf.bad_map_offsets[a * (B + 1) + B] = off_bad + (F - fa);
                        bv = p.buf.bad_map_values + a * cap + off_bad;
                    }
                    int local = 0, n_ok = 0, n_bad = 0;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        if ((lit1 >> c) & 1ull) {
                            if ((ok1[a] >> c) & 1ull)
                                av[n_ok++] = local;
                            else if (bv)
                                bv[n_bad++] = local;
                            ++local;
                        }
                    }
                }
        }
        if (err) atomicOr(p.buf.error_flags, err);

        // The workgroup owning the last chunk finished its look-back only after every other chunk published, i.e.
        // after every workgroup of this launch read the epoch: it can advance it for the next launch.
        if (chunk == p.nchunks - 1 && tid == 0) __hip_atomic_store(p.epoch, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// uniform random policy over OneOf([task] * n + [noop]) (spaces/actions.py:23-41; baselines/random.py:20):
// member index j ~ U{0..n}; j < n -> [j, 0] (fight task j of the action mapping), j == n -> [n, -1] (noop/refill)
__global__ void __launch_bounds__(kBlock) wf_policy_kernel(const int32_t* agent_task_count, const int64_t* env_task_count,
                                                             int show_bad_actions, int A, int64_t B, uint32_t seed_lo, uint32_t seed_hi,
                                                             uint32_t step_lo, uint32_t step_hi, int32_t* actions) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)A * B) return;
    const int64_t b = i % B;
    const int n = show_bad_actions ? (int)env_task_count[b] : agent_task_count[i];
    const frz::Philox4 w = frz::philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), step_lo, step_hi, seed_lo, seed_hi);
    const int j = (int)(((uint64_t)w.w[0] * (uint64_t)(n + 1)) >> 32);
    reinterpret_cast<int2*>(actions)[i] = j < n ? make_int2(j, 0) : make_int2(n, -1);
}

}  // namespace

// ================================================================================================================
// host side of the C-ABI
// ================================================================================================================
struct frz_wildfire_env {
    frz_wildfire_cfg cfg;
    WfParams params;
    FillParams fill;
    bool bound = false;
    bool was_reset = false;
    int grid = 0;
    int variant = 0;  // index into the (CMAX, AMAX) instantiation table
    float* rand_field = nullptr;
    float* rand_agent = nullptr;
};

namespace {

struct Variant {
    int cmax, amax;
};
constexpr Variant kVariants[] = {{8, 4}, {24, 8}, {64, 16}};

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct WorkspaceLayout {
    int64_t epoch, totals, agg, prefix, rand_field, rand_agent, total;
};

WorkspaceLayout workspace_layout(const frz_wildfire_cfg* cfg) {
    const int64_t B = cfg->parallel_envs, HW = (int64_t)cfg->grid_height * cfg->grid_width, A = cfg->num_agents;
    const int64_t nchunks = (B + kBlock - 1) / kBlock, nch = A + 3;
    WorkspaceLayout l;
    int64_t at = 0;
    l.epoch = at;
    at += 128;
    l.totals = at;
    at += 2 * kTotalsStride * 4;
    l.agg = at;
    at = align_up(at + nchunks * nch * 8, 128);
    l.prefix = at;
    at = align_up(at + nchunks * nch * 8, 128);
    l.rand_field = at;
    at = align_up(at + 3 * B * HW * 4, 128);
    l.rand_agent = at;
    at = align_up(at + 5 * B * A * 4, 128);
    l.total = at;
    return l;
}

template <int CMAX, int AMAX>
void launch_variant(const WfParams& p, int grid, int rng, int mode, hipStream_t stream) {
    if (mode == kRebuild) {
        hipLaunchKernelGGL((wf_step_kernel<CMAX, AMAX, FRZ_RNG_INJECTED, kRebuild>), dim3(grid), dim3(kBlock), 0, stream, p);
    } else if (rng == FRZ_RNG_PHILOX) {
        hipLaunchKernelGGL((wf_step_kernel<CMAX, AMAX, FRZ_RNG_PHILOX, kStep>), dim3(grid), dim3(kBlock), 0, stream, p);
    } else {
        hipLaunchKernelGGL((wf_step_kernel<CMAX, AMAX, FRZ_RNG_INJECTED, kStep>), dim3(grid), dim3(kBlock), 0, stream, p);
    }
}

int launch(frz_wildfire_env* env, int rng, int mode, hipStream_t stream) {
    switch (env->variant) {
        case 0: launch_variant<8, 4>(env->params, env->grid, rng, mode, stream); break;
        case 1: launch_variant<24, 8>(env->params, env->grid, rng, mode, stream); break;
        default: launch_variant<64, 16>(env->params, env->grid, rng, mode, stream); break;
    }
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

}  // namespace

extern "C" {

int frz_abi_version(void) { return FRZ_ABI_VERSION; }

int64_t frz_wildfire_workspace_bytes(const frz_wildfire_cfg* cfg) {
    if (!cfg || cfg->parallel_envs <= 0) return FRZ_E_INVALID;
    return workspace_layout(cfg).total;
}

int frz_wildfire_create(const frz_wildfire_cfg* cfg, frz_wildfire_env** out) {
    if (!cfg || !out) return FRZ_E_INVALID;
    const int H = cfg->grid_height, W = cfg->grid_width, HW = H * W, A = cfg->num_agents;
    if (cfg->parallel_envs <= 0 || H <= 0 || W <= 0 || HW > FRZ_MAX_CELLS || A <= 0 || A > FRZ_MAX_AGENTS) return FRZ_E_INVALID;
    if (cfg->num_equipment_states <= 0 || cfg->num_equipment_states > FRZ_MAX_EQUIPMENT_STATES) return FRZ_E_INVALID;
    if (cfg->num_capacities <= 0 || cfg->num_capacities > FRZ_MAX_CAPACITIES) return FRZ_E_INVALID;
    if (cfg->num_fire_states < 2) return FRZ_E_INVALID;
    if ((int64_t)cfg->parallel_envs * HW >= (int64_t)1 << 31) return FRZ_E_INVALID;  // 32-bit scan channels

    frz_wildfire_env* env = new (std::nothrow) frz_wildfire_env();
    if (!env) return FRZ_E_INVALID;
    env->cfg = *cfg;
    int variant = -1;
    for (int i = 0; i < 3; ++i)
        if (HW <= kVariants[i].cmax && A <= kVariants[i].amax) {
            variant = i;
            break;
        }
    env->variant = variant;

    WfParams& p = env->params;
    std::memset(&p, 0, sizeof(p));
    p.B = cfg->parallel_envs;
    p.H = H;
    p.W = W;
    p.HW = HW;
    p.A = A;
    p.S = cfg->num_equipment_states;
    p.K = cfg->num_capacities;
    p.nchunks = (cfg->parallel_envs + kBlock - 1) / kBlock;
    p.nch = A + 3;
    p.others_k = 2 + (cfg->observe_other_power ? 1 : 0) + (cfg->observe_other_suppressant ? 1 : 0);
    p.max_steps = cfg->max_steps;
    p.num_fire_states = cfg->num_fire_states;
    p.stochastic_increase = cfg->stochastic_increase;
    p.stochastic_burnouts = cfg->stochastic_burnouts;
    p.stochastic_decrease = cfg->stochastic_decrease;
    p.use_fire_fuel = cfg->use_fire_fuel;
    p.stochastic_supp_decrease = cfg->stochastic_suppressant_decrease;
    p.stochastic_refill = cfg->stochastic_refill;
    p.stochastic_switch = cfg->stochastic_switch;
    p.stochastic_repair = cfg->stochastic_repair;
    p.stochastic_degrade = cfg->stochastic_degrade;
    p.critical_error = cfg->critical_error;
    p.show_bad_actions = cfg->show_bad_actions;
    p.observe_other_power = cfg->observe_other_power;
    p.observe_other_suppressant = cfg->observe_other_suppressant;
    p.burnout_penalty_scaled = cfg->burnout_penalty_scaled;
    p.localize_putouts = cfg->localize_putouts;
    p.track_cumulative = cfg->track_cumulative_rewards;
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
    std::memcpy(p.eq, cfg->equipment_states, sizeof(p.eq));
    std::memcpy(p.caps, cfg->possible_capacities, sizeof(p.caps));
    std::memcpy(p.cum, cfg->capacity_cumprobs, sizeof(p.cum));
    std::memcpy(p.ay, cfg->agent_y, sizeof(p.ay));
    std::memcpy(p.ax, cfg->agent_x, sizeof(p.ax));
    std::memcpy(p.power, cfg->fire_reduction_power, sizeof(p.power));
    std::memcpy(p.fire_rewards, cfg->fire_rewards, sizeof(p.fire_rewards));
    std::memcpy(p.ignition, cfg->ignition_temp, sizeof(p.ignition));
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
                const int d = dy > dx ? dy : dx;
                if ((float)d <= true_range) m |= 1ull << c;
            }
            p.range_mask[a][s] = m;
        }

    FillParams& fp = env->fill;
    std::memset(&fp, 0, sizeof(fp));
    fp.B = cfg->parallel_envs;
    fp.HW = HW;
    fp.A = A;
    fp.initial_fuel = cfg->initial_fuel;
    fp.initial_equipment = cfg->initial_equipment_state;
    fp.initial_suppressant = cfg->initial_suppressant;
    fp.initial_capacity = cfg->initial_capacity;
    std::memcpy(fp.fire_types, cfg->fire_types, sizeof(fp.fire_types));
    std::memcpy(fp.lit, cfg->lit, sizeof(fp.lit));
    std::memcpy(fp.ignition, cfg->ignition_temp, sizeof(fp.ignition));

    // Co-resident persistent grid: every workgroup of the launch must be resident for the single-pass prefix hand-off
    // (a chunk waits on chunks owned by other workgroups).  One 256-thread workgroup per CU is always resident.
    int device = 0, cus = 256;
    if (hipGetDevice(&device) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    int per_cu = p.nchunks >= 4 * cus ? 2 : 1;
    env->grid = p.nchunks < cus * per_cu ? p.nchunks : cus * per_cu;
    *out = env;
    return FRZ_OK;
}

void frz_wildfire_destroy(frz_wildfire_env* env) { delete env; }

int frz_wildfire_bind(frz_wildfire_env* env, const frz_wildfire_bufs* bufs) {
    if (!env || !bufs || !bufs->workspace || !bufs->error_flags || !bufs->fires || !bufs->frozen_scaled) return FRZ_E_INVALID;
    if (env->cfg.show_bad_actions && (!bufs->bad_map_values || !bufs->bad_map_offsets)) return FRZ_E_INVALID;
    env->params.track_cumulative = env->cfg.track_cumulative_rewards && bufs->cumulative_rewards != nullptr;
    env->params.buf = *bufs;
    env->fill.buf = *bufs;
    const WorkspaceLayout l = workspace_layout(&env->cfg);
    char* ws = static_cast<char*>(bufs->workspace);
    env->params.epoch = reinterpret_cast<uint32_t*>(ws + l.epoch);
    env->params.totals = reinterpret_cast<uint32_t*>(ws + l.totals);
    env->params.agg = reinterpret_cast<uint64_t*>(ws + l.agg);
    env->params.prefix = reinterpret_cast<uint64_t*>(ws + l.prefix);
    env->rand_field = reinterpret_cast<float*>(ws + l.rand_field);
    env->rand_agent = reinterpret_cast<float*>(ws + l.rand_agent);
    env->bound = true;
    return FRZ_OK;
}

int frz_wildfire_rebuild(frz_wildfire_env* env, void* stream) {
    if (!env) return FRZ_E_INVALID;
    if (!env->bound) return FRZ_E_UNBOUND;
    env->was_reset = true;
    return launch(env, FRZ_RNG_INJECTED, kRebuild, static_cast<hipStream_t>(stream));
}

int frz_wildfire_reset(frz_wildfire_env* env, void* stream) {
    if (!env) return FRZ_E_INVALID;
    if (!env->bound) return FRZ_E_UNBOUND;
    const int blocks = (env->cfg.parallel_envs + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(wf_fill_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), env->fill);
    if (hipGetLastError() != hipSuccess) return FRZ_E_LAUNCH;
    return frz_wildfire_rebuild(env, stream);
}

int frz_mt19937_generate(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events, int64_t count, int64_t B, void* stream);

int frz_wildfire_step(frz_wildfire_env* env, const int32_t* actions, int rng_mode, const float* field_randomness,
                      const float* agent_randomness, void* stream) {
    if (!env || !actions) return FRZ_E_INVALID;
    if (!env->bound) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;  // reset()/rebuild() must precede the first step
    const frz_wildfire_cfg& c = env->cfg;
    env->params.actions = actions;
    if (rng_mode == FRZ_RNG_INJECTED) {
        if (!field_randomness || !agent_randomness) return FRZ_E_INVALID;
        env->params.field_rand = field_randomness;
        env->params.agent_rand = agent_randomness;
    } else if (rng_mode == FRZ_RNG_MT19937) {
        // per-env MT19937 streams: field draws first, then agent draws (wildfire.py:409-410), staged in the workspace
        if (!env->params.buf.mt_state || !env->params.buf.mt_index) return FRZ_E_INVALID;
        const int64_t B = c.parallel_envs;
        int rc = frz_mt19937_generate(env->params.buf.mt_state, env->params.buf.mt_index, env->rand_field, 3,
                                      (int64_t)c.grid_height * c.grid_width, B, stream);
        if (rc != FRZ_OK) return rc;
        rc = frz_mt19937_generate(env->params.buf.mt_state, env->params.buf.mt_index, env->rand_agent, 5, c.num_agents, B, stream);
        if (rc != FRZ_OK) return rc;
        env->params.field_rand = env->rand_field;
        env->params.agent_rand = env->rand_agent;
        rng_mode = FRZ_RNG_INJECTED;
    } else if (rng_mode == FRZ_RNG_PHILOX) {
        if (!env->params.buf.seeds) return FRZ_E_INVALID;
    } else {
        return FRZ_E_INVALID;
    }
    return launch(env, rng_mode, kStep, static_cast<hipStream_t>(stream));
}

int frz_wildfire_random_policy(frz_wildfire_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out, void* stream) {
    if (!env || !actions_out) return FRZ_E_INVALID;
    if (!env->bound) return FRZ_E_UNBOUND;
    const int64_t n = (int64_t)env->cfg.num_agents * env->cfg.parallel_envs;
    const int blocks = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(wf_policy_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       env->params.buf.agent_task_count, env->params.buf.env_task_count, env->cfg.show_bad_actions, env->cfg.num_agents,
                       (int64_t)env->cfg.parallel_envs, (uint32_t)policy_seed, (uint32_t)(policy_seed >> 32), (uint32_t)policy_step,
                       (uint32_t)(policy_step >> 32), actions_out);
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

}  // extern "C"
