// wildfire_group.hip — wildfire step with G lanes per environment (small grids: H*W <= G, agents + 1 <= G).
//
// Why: at the benchmark's batch (65 536 envs) one env per lane fills only one wavefront per SIMD, and a lone wave issues
// one VALU instruction per ~4 cycles.  Here lane l of a G-lane group owns cell l (fire transitions) and, for
// 1 <= l <= A, agent l-1 (suppressant / equipment / capacity transitions, rewards, its action mapping); lane 0 also
// keeps the env-level bookkeeping.  B * G lanes give 8 waves per SIMD at G = 8, each lane runs a ~10x shorter program,
// and the exchange inside an env is wave-local: __ballot for the lit / put-out / burnt cell sets, ds_bpermute shuffles
// for attack powers and offsets.  No barrier, no LDS traffic inside an env.
//
// The variable-length task lists are compacted with a strided wavefront scan (lane l scans channel l over the envs of
// the wave), a workgroup combine through LDS and a two-level single-pass inter-workgroup prefix hand-off (granules
// tagged with the launch epoch, frz_device.h).  Same arena / same outputs / same random streams as wildfire.hip.
#include "wildfire_common.h"

namespace {

using namespace frz_wf;

template <int G>
__device__ __forceinline__ uint32_t group_bits(uint64_t ballot, int group_base) {
    return (uint32_t)(ballot >> group_base) & ((1u << G) - 1u);
}

// value held by lane `src` of this lane's group
template <int G, typename T>
__device__ __forceinline__ T group_read(T v, int src) {
    return __shfl(v, src, G);
}

template <int G, int RNG, int MODE>
__global__ void __launch_bounds__(kBlock) wf_group_kernel(char* __restrict__ arena, const WfDev* __restrict__ dev,
                                                           const int32_t* __restrict__ actions, const float* __restrict__ field_rand,
                                                           const float* __restrict__ agent_rand) {
    constexpr int EPW = 64 / G;       // envs per wavefront
    constexpr int EPB = kBlock / G;   // envs per workgroup chunk
    constexpr int NCHP = G + 2 <= 8 ? 8 : (G + 2 <= 16 ? 16 : (G + 2 <= 32 ? 32 : 64));  // scan channels (<= G + 2) padded
    static_assert(G >= 8 && G <= 32, "group width");

    __shared__ uint32_t s_wave_scan[frz::kWaves][G];
    __shared__ uint32_t s_wave_live[frz::kWaves][2];
    __shared__ uint32_t s_reduce[frz::kWaves][NCHP];
    __shared__ uint32_t s_reduce2[frz::kWaves][NCHP];
    __shared__ uint32_t s_prefix[NCHP];
    __shared__ uint32_t s_range[FRZ_MAX_AGENTS][FRZ_MAX_EQUIPMENT_STATES];
    __shared__ float s_eq[FRZ_MAX_EQUIPMENT_STATES][4];
    __shared__ float s_caps[FRZ_MAX_CAPACITIES];
    __shared__ float s_fire_rewards[G];

    const WfDev& d = *dev;
    const int tid = threadIdx.x, lane = frz::lane_id(), wave = frz::wave_id();
    const int l = lane & (G - 1);           // role inside the env: cell l, agent l - 1, leader if 0
    const int group_base = lane & ~(G - 1);  // first lane of this env's group
    const int64_t B = d.B;
    const uint32_t Bu = (uint32_t)d.B;
    const int HW = d.HW, A = d.A, W = d.W;
    const int nch = d.nch;  // A + 3: F, F_a..., not-terminated, not-truncated
    const int ch_nt = A + 1, ch_ntr = A + 2;
    const uint32_t flags = d.flags;
    const bool is_cell = l < HW, is_agent = l >= 1 && l <= A, is_leader = l == 0;
    const int a = is_agent ? l - 1 : 0;

    if (tid < FRZ_MAX_AGENTS * FRZ_MAX_EQUIPMENT_STATES)
        (&s_range[0][0])[tid] = (uint32_t)d.range_mask[tid / FRZ_MAX_EQUIPMENT_STATES][tid % FRZ_MAX_EQUIPMENT_STATES];
    if (tid < FRZ_MAX_EQUIPMENT_STATES * 4) (&s_eq[0][0])[tid] = (&d.eq[0][0])[tid];
    if (tid < FRZ_MAX_CAPACITIES) s_caps[tid] = d.caps[tid];
    if (tid < G) s_fire_rewards[tid] = tid < HW ? d.fire_rewards[tid] : 0.0f;

    // per-lane constants of this lane's roles
    const float my_power = d.power[a];
    const float my_ay = (float)d.ay[a], my_ax = (float)d.ax[a];
    const int my_ignition = d.ignition[is_cell ? l : 0];
    const int my_yx = d.cell_yx[is_cell ? l : 0];

    uint32_t* const epoch_ptr = reinterpret_cast<uint32_t*>(arena + d.off_epoch);
    uint32_t* const totals = reinterpret_cast<uint32_t*>(arena + d.off_totals);
    // plain (cacheable, wave-uniform) load: the word was last written by the previous launch, and this launch only
    // rewrites it after every workgroup has read it — an agent-scope load here would send one L2 request per wavefront
    // of the grid to a single address
    const uint32_t epoch = *epoch_ptr;
    const uint32_t tag = epoch + 1u;
    const uint32_t* prev = totals + ((epoch + 1u) & 1u) * kTotalsStride;
    uint32_t* cur = totals + (epoch & 1u) * kTotalsStride;

    int32_t* const rows = reinterpret_cast<int32_t*>(arena + d.off_rows4);
    float* const rowsf = reinterpret_cast<float*>(arena + d.off_rows4);
    int64_t* const rows8 = reinterpret_cast<int64_t*>(arena + d.off_rows8);
    uint8_t* const rows1 = reinterpret_cast<uint8_t*>(arena + d.off_rows1);
    uint64_t* const agg = reinterpret_cast<uint64_t*>(arena + d.off_agg);
    uint64_t* const gtot = reinterpret_cast<uint64_t*>(arena + d.off_gtot);
    uint64_t* const prefix = reinterpret_cast<uint64_t*>(arena + d.off_prefix);

    // utils/env.py:211-213 — every per-agent step() is a no-op once ALL envs are terminated or ALL are truncated.
    bool frozen = false;
    if (MODE == kStep) frozen = prev[ch_nt] == 0u || prev[ch_ntr] == 0u;
    // quirk wildfire.py:434-435: an agent with no attackable task in ANY env of the batch is skipped (channel l total)
    const bool skipped = MODE == kStep && is_agent && prev[l] == 0u;
    __syncthreads();

    for (int chunk = blockIdx.x; chunk < d.nchunks; chunk += gridDim.x) {
        const int64_t b = (int64_t)chunk * EPB + (tid / G);
        const bool active = b < B;
        const uint32_t bl = (uint32_t)(active ? b : B - 1);  // lanes past the batch shadow the last env; their stores are masked

        if (frozen) {
            // The parallel adapter (utils/conversions.py:87-90) then adds the stale aec rewards once per agent call.
            const bool todo = active && !at32(rows1, (uint32_t)d.u_frozen * Bu + bl);
            if (todo && is_agent) {
                const float r = at32(rowsf, (uint32_t)(d.r_rewards + a) * Bu + bl);
                float acc = 0.0f;
                for (int j = 0; j < A; ++j) acc = acc + r;
                at32(rowsf, (uint32_t)(d.r_rewards + a) * Bu + bl) = acc;
            }
            // every lane of the group has read the flag before the leader sets it (same wavefront, program order)
            if (todo && is_leader) at32(rows1, (uint32_t)d.u_frozen * Bu + bl) = 1;
            continue;
        }

        // ------------------------------------------------------------------------------------------ load state
        int f = 0, in = 0, fu = 0;
        if (is_cell) {
            f = at32(rows, (uint32_t)(d.r_fires + l) * Bu + bl);
            in = at32(rows, (uint32_t)(d.r_intensity + l) * Bu + bl);
            fu = at32(rows, (uint32_t)(d.r_fuel + l) * Bu + bl);
        }
        float supp = 0.0f, capa = 0.0f;
        int eqs = 0;
        if (is_agent) {
            supp = at32(rowsf, (uint32_t)(d.r_supp + a) * Bu + bl);
            capa = at32(rowsf, (uint32_t)(d.r_cap + a) * Bu + bl);
            eqs = at32(rows, (uint32_t)(d.r_equip + a) * Bu + bl);
        }
        // agents share one termination / truncation value (wildfire.py:579, utils/env.py:231-233)
        bool term = at32(rows1, (uint32_t)d.u_term * Bu + bl) != 0;
        bool trunc = at32(rows1, (uint32_t)d.u_trunc * Bu + bl) != 0;

        float rew = 0.0f;
        uint32_t err = 0;

        if (MODE == kStep) {
            int act_idx = 0, act_id = -1;
            if (is_agent) {
                const int2 v = reinterpret_cast<const int2*>(actions)[a * B + bl];
                act_idx = v.x;
                act_id = v.y;
            }
            int nm = at32(rows, (uint32_t)d.r_moves * Bu + bl);
            int nb = at32(rows, (uint32_t)d.r_burnouts * Bu + bl);

            // ---------------------------------------------------------------------------------- randomness
            // FRZ_RNG_PHILOX stream (include/frz.h): lane l draws Philox(counter = (l, step, block, 0), key = (seed, tag)):
            // block 0 -> field events 0..2 of cell l and agent event 0 of agent l-1; block 1 -> agent events 1..4.
            float rf0 = 1.0f, rf1 = 1.0f, rf2 = 1.0f, ra0 = 0.0f, ra1 = 0.0f, ra2 = 0.0f, ra3 = 0.0f, ra4 = 0.0f;
            if (RNG == FRZ_RNG_INJECTED) {
                if (is_cell) {
                    rf0 = field_rand[((int64_t)0 * B + bl) * HW + l];
                    rf1 = field_rand[((int64_t)1 * B + bl) * HW + l];
                    rf2 = field_rand[((int64_t)2 * B + bl) * HW + l];
                }
                if (is_agent) {
                    ra0 = agent_rand[((int64_t)0 * B + bl) * A + a];
                    ra1 = agent_rand[((int64_t)1 * B + bl) * A + a];
                    ra2 = agent_rand[((int64_t)2 * B + bl) * A + a];
                    ra3 = agent_rand[((int64_t)3 * B + bl) * A + a];
                    ra4 = agent_rand[((int64_t)4 * B + bl) * A + a];
                }
            } else {
                const uint32_t seed = (uint32_t)at32(rows, (uint32_t)d.r_seeds * Bu + bl);
                const frz::Philox4 w0 = frz::philox4x32_10((uint32_t)l, (uint32_t)nm, 0u, 0u, seed, 0x46525A00u);
                rf0 = frz::u32_to_unit_float(w0.w[0]);
                rf1 = frz::u32_to_unit_float(w0.w[1]);
                rf2 = frz::u32_to_unit_float(w0.w[2]);
                ra0 = frz::u32_to_unit_float(w0.w[3]);
                const bool need_block1 = (flags & (kStochRepair | kStochDegrade | kCritical | kStochRefill | kStochSwitch)) != 0 || d.K > 1;
                if (need_block1 && is_agent) {
                    const frz::Philox4 w1 = frz::philox4x32_10((uint32_t)l, (uint32_t)nm, 1u, 0u, seed, 0x46525A00u);
                    ra1 = frz::u32_to_unit_float(w1.w[0]);
                    ra2 = frz::u32_to_unit_float(w1.w[1]);
                    ra3 = frz::u32_to_unit_float(w1.w[2]);
                    ra4 = frz::u32_to_unit_float(w1.w[3]);
                }
            }

            // ------------------------------------------------------------- action decode (wildfire.py:427-483)
            // The action mapping of the previous rebuild is a pure function of the state it was built from = the state
            // just loaded: attackable set of an agent = lit fires within its (equipment-adjusted) range, non-empty only
            // while it has suppressant (wildfire.py:604-623).
            const uint32_t lit0 = group_bits<G>(__ballot(f > 0), group_base);
            const bool show_bad = (flags & kShowBad) != 0;
            const uint32_t ok = (is_agent && supp > 0.0f) ? (lit0 & s_range[a][eqs]) : 0u;
            const bool refill = is_agent && act_id == -1;
            const bool fight = is_agent && !refill && !skipped;
            const uint32_t sel = show_bad ? lit0 : ok;
            const bool valid = act_idx >= 0 && act_idx < __popc(sel);
            int target = 0;
            {
                int seen = 0;
#pragma unroll
                for (int c = 0; c < G; ++c) {
                    const int bit = (int)((sel >> c) & 1u);
                    target = (bit && seen == act_idx) ? c : target;
                    seen += bit;
                }
            }
            const bool attackable = ((ok >> target) & 1u) != 0u;
            const bool good = fight && valid && (!show_bad || attackable);
            if (fight && !valid && active) err |= FRZ_ERR_BAD_ACTION_INDEX;
            const float power = my_power + s_eq[eqs][1];
            const int hit = good ? target : -1;
            rew = (fight && !good) ? d.bad_attack_penalty : 0.0f;  // assignment, :477

            // attack power per cell, accumulated in agent order (wildfire.py:470)
            float ap = 0.0f;
            for (int j = 1; j <= A; ++j) {
                const int t = group_read<G>(hit, j);
                const float pw = group_read<G>(power, j);
                ap = ap + (t == l ? pw : 0.0f);
            }

            // ---------------------------------------------- agent transitions (suppressant/equipment/capacity)
            if (is_agent) {
                // transitions/suppressant_decrease.py:56-61
                const bool dec = good && (!(flags & kStochSuppDecrease) || ra0 < d.p_supp_decrease);
                float s = dec ? supp - 1.0f : supp;
                s = s < 0.0f ? 0.0f : s;
                // transitions/equipment.py:51-75 (masks from the value before any write)
                const int e0 = eqs, top = d.S - 1;
                const bool pristine = e0 == top, damaged = e0 == 0, inter = !pristine && !damaged;
                const bool repairs = (flags & kStochRepair) ? (damaged && ra1 < d.p_repair) : damaged;
                const bool crit = (flags & kCritical) && pristine && ra1 < d.p_critical;
                bool degr = (flags & kStochDegrade) ? ((pristine || inter) && ra1 < d.p_degrade) : (inter || pristine);
                degr = degr && !crit;
                int e = repairs ? top : e0;
                e = crit ? 0 : e;
                e = degr ? e - 1 : e;
                // transitions/suppressant_refill.py:63-70 (bonus from the NEW equipment state)
                const bool inc = refill && (!(flags & kStochRefill) || ra2 < d.p_refill);
                s = inc ? capa + s_eq[e][0] : s;
                // transitions/capacity.py:52-64: bucketize(r, cumsum) = #{j : cum[j] < r} (cum padded with +inf)
                int ci = 0;
#pragma unroll
                for (int j = 0; j < FRZ_MAX_CAPACITIES; ++j) ci += ra3 > d.cum[j] ? 1 : 0;
                ci = ci > d.K - 1 ? d.K - 1 : ci;
                const float new_max = s_caps[ci];
                const bool sw = inc && (!(flags & kStochSwitch) || ra4 < d.p_switch);
                const float bonus = s - capa;
                capa = sw ? new_max : capa;
                s = sw ? new_max + bonus : s;
                supp = s;
                eqs = e;
            }

            // ------------------------------------------------------------ fire increase / decrease (cell lanes)
            bool bo = false, po = false;
            if (is_cell) {
                const int almost_state = d.num_fire_states - 2, burnout_state = d.num_fire_states - 1;
                {  // transitions/fire_increase.py:61-91
                    const int required = f >= 0 ? f : 0;
                    const float diff = (float)required - ap;
                    const bool lit = f > 0 && in > 0;
                    const bool unmet = diff > 0.0f && lit;
                    const bool almost = unmet && in == almost_state;
                    const float p_unmet = (flags & kStochIncrease) ? d.p_increase : 1.0f;
                    const float p_almost = (flags & kStochBurnouts) ? d.p_burnout : d.p_increase;  // :77-80
                    float prob = unmet ? (almost ? p_almost : p_unmet) : 0.0f;
                    prob = clamp01(prob);
                    const bool inc = rf0 < prob;
                    in += inc ? 1 : 0;
                    bo = inc && in >= burnout_state;
                    f = bo ? -f : f;
                    fu = bo ? (fu - 1 < 0 ? 0 : fu - 1) : fu;
                }
                {  // transitions/fire_decrease.py:56-77: p = p_dec + ((-1 * diff) * bonus), each op rounded
                    const int required = f >= 0 ? f : 0;
                    const float diff = (float)required - ap;
                    const bool lit = f > 0 && in > 0;
                    const bool met = diff <= 0.0f && lit;
                    const float stoch_p = __fadd_rn(d.p_decrease, __fmul_rn(__fmul_rn(-1.0f, diff), d.decrease_bonus));
                    float prob = met ? ((flags & kStochDecrease) ? stoch_p : 1.0f) : 0.0f;
                    prob = clamp01(prob);
                    const bool dec = rf1 < prob;
                    in -= dec ? 1 : 0;
                    po = dec && in <= 0;
                    f = po ? -f : f;
                    fu = po ? fu - 1 : fu;  // unclamped, :75
                }
            }
            const uint32_t burned = group_bits<G>(__ballot(bo), group_base);
            const uint32_t put_out = group_bits<G>(__ballot(po), group_base);
            const uint32_t lit2 = group_bits<G>(__ballot(f > 0 && in > 0), group_base);
            // ---------------------------------------- fire spread stencil (transitions/fire_spreads.py:44-57)
            if (is_cell) {
                const uint32_t from_n = (lit2 << W) & (uint32_t)d.has_n, from_s = (lit2 >> W) & (uint32_t)d.has_s;
                const uint32_t from_w = (lit2 << 1) & (uint32_t)d.has_w, from_e = (lit2 >> 1) & (uint32_t)d.has_e;
                float prob = 0.0f;  // conv2d accumulation order: N, W, E, S
                prob = __fadd_rn(prob, ((from_n >> l) & 1u) ? d.spread_n : 0.0f);
                prob = __fadd_rn(prob, ((from_w >> l) & 1u) ? d.spread_w : 0.0f);
                prob = __fadd_rn(prob, ((from_e >> l) & 1u) ? d.spread_e : 0.0f);
                prob = __fadd_rn(prob, ((from_s >> l) & 1u) ? d.spread_s : 0.0f);
                bool unlit = f < 0 && in == 0;
                unlit = unlit && (!(flags & kUseFuel) || fu > 0);
                prob = unlit ? __fadd_rn(prob, d.random_ignition) : 0.0f;
                const bool spread = rf2 < prob;
                f = spread ? -f : f;
                in = spread ? my_ignition : in;
            }

            // -------------------------------------------------- rewards and termination (wildfire.py:534-582)
            const bool any_fire = group_bits<G>(__ballot(f > 0), group_base) != 0u;
            bool dead = !any_fire;
            if (flags & kUseFuel) {
                int fuel_sum = fu;
#pragma unroll
                for (int dd = 1; dd < G; dd <<= 1) fuel_sum += __shfl_xor(fuel_sum, dd, G);
                dead = dead && fuel_sum <= 0;
            }
            f = dead ? 0 : f;  // :570
            const bool newly = !term && dead;
            const int n_burn = __popc(burned), n_put = __popc(put_out);
            if (is_agent) {
                float fire_reward_sum = 0.0f, burnout_total = 0.0f;  // sequential over cells, as the oracle
                for (int c = 0; c < HW; ++c) {
                    const float fr = s_fire_rewards[c];
                    fire_reward_sum = __fadd_rn(fire_reward_sum, ((put_out >> c) & 1u) ? fr : 0.0f);
                    const float pen = (flags & kPenaltyScaled) ? __fmul_rn(-1.0f, fr) : d.burnout_penalty;
                    burnout_total = __fadd_rn(burnout_total, ((burned >> c) & 1u) ? pen : 0.0f);
                }
                float base_reward = fire_reward_sum;
                if (flags & kLocalize) base_reward = (hit >= 0 && ((put_out >> hit) & 1u)) ? s_fire_rewards[hit < 0 ? 0 : hit] : 0.0f;
                rew = __fadd_rn(rew, __fadd_rn(base_reward, burnout_total));
                // correctly rounded float32 log via double (bit-identical to the oracle; the reference's torch.log is a
                // <=1-ulp float32 log); only wavefronts holding a newly terminated env evaluate it
                float log_burnouts = 0.0f;
                if (newly && d.termination_kappa != 0.0f) log_burnouts = (float)log((double)nb + 1.0);
                float term_reward = __fsub_rn(d.termination_reward, __fmul_rn(d.termination_kappa, log_burnouts));
                term_reward = term_reward < 0.0f ? 0.0f : term_reward;
                rew = newly ? __fadd_rn(rew, term_reward) : rew;
            }
            nb += n_burn;
            nm += 1;
            trunc = (flags & kTruncate) ? nm >= d.max_steps : trunc;
            term = term || dead;

            // ------------------------------------------------------------------------ dense stores (state)
            if (active) {
                if (is_cell) {
                    at32(rows, (uint32_t)(d.r_fires + l) * Bu + bl) = f;
                    at32(rows, (uint32_t)(d.r_intensity + l) * Bu + bl) = in;
                    at32(rows, (uint32_t)(d.r_fuel + l) * Bu + bl) = fu;
                }
                if (is_agent) {
                    at32(rowsf, (uint32_t)(d.r_supp + a) * Bu + bl) = supp;
                    at32(rowsf, (uint32_t)(d.r_cap + a) * Bu + bl) = capa;
                    at32(rows, (uint32_t)(d.r_equip + a) * Bu + bl) = eqs;
                    at32(rowsf, (uint32_t)(d.r_rewards + a) * Bu + bl) = rew;
                    at32(rows1, (uint32_t)(d.u_term + a) * Bu + bl) = (uint8_t)term;
                    if (flags & kTruncate) at32(rows1, (uint32_t)(d.u_trunc + a) * Bu + bl) = (uint8_t)trunc;
                    if (flags & kTrackCumulative) {
                        float& cum = at32(rowsf, (uint32_t)(d.r_cum + a) * Bu + bl);
                        cum = __fadd_rn(cum, rew);
                    }
                }
                if (is_leader) {
                    at32(rows, (uint32_t)d.r_moves * Bu + bl) = nm;
                    at32(rows, (uint32_t)d.r_burnouts * Bu + bl) = nb;
                    at32(rows8, (uint32_t)d.q_burnouts * Bu + bl) = n_burn;
                    at32(rows8, (uint32_t)d.q_putouts * Bu + bl) = n_put;
                }
            }
        }

        // ======================================================================================================
        // update_observations + update_actions on the new state (wildfire.py:586-717)
        // ======================================================================================================
        const uint32_t lit1 = active ? group_bits<G>(__ballot(f > 0), group_base) : 0u;
        const int F = __popc(lit1);
        const uint32_t ok1 = (is_agent && supp > 0.0f) ? (lit1 & s_range[a][eqs]) : 0u;
        const int Fa = __popc(ok1);
        // channel l of this env: 0 -> number of tasks, a + 1 -> number of tasks agent a can attack
        const uint32_t cnt = is_leader ? (uint32_t)F : (uint32_t)Fa;

        // lane l scans channel l over the envs of the wave (stride G), then over the waves of the workgroup
        uint32_t incl = cnt;
#pragma unroll
        for (int dd = G; dd < 64; dd <<= 1) {
            const uint32_t up = __shfl_up(incl, dd, 64);
            if (lane >= dd) incl += up;
        }
        const uint32_t live_nt = (uint32_t)__popcll(__ballot(active && is_leader && !term));
        const uint32_t live_ntr = (uint32_t)__popcll(__ballot(active && is_leader && !trunc));
        __syncthreads();  // LDS reuse across chunks of a persistent workgroup
        if (lane >= 64 - G) s_wave_scan[wave][l] = incl;
        if (lane == 0) {
            s_wave_live[wave][0] = live_nt;
            s_wave_live[wave][1] = live_ntr;
        }
        __syncthreads();
        uint32_t wave_base = 0, chunk_total = 0;
#pragma unroll
        for (int j = 0; j < frz::kWaves; ++j) {
            const uint32_t t = s_wave_scan[j][l];
            wave_base += j < wave ? t : 0u;
            chunk_total += t;
        }
        const uint32_t excl = wave_base + incl - cnt;  // envs of this chunk before mine, channel l

        // publish this chunk's channel sums
        uint32_t my_total = 0;  // channel `tid` (tid < nch)
        if (tid < nch) {
            if (tid <= A) {
                my_total = chunk_total;  // tid < G: lane l == tid holds channel tid
            } else {
#pragma unroll
                for (int j = 0; j < frz::kWaves; ++j) my_total += s_wave_live[j][tid - ch_nt];
            }
            frz::granule_store(agg + (int64_t)chunk * nch + tid, tag, my_total);
        }

        if (active) {
            // agent observations (wildfire.py:677-681, 704-716)
            if (is_agent) {
                float* const obs_self = reinterpret_cast<float*>(arena + d.off_obs_self);
                reinterpret_cast<float4*>(obs_self)[a * B + b] = make_float4(my_ay, my_ax, my_power, supp);
                at32(rows, (uint32_t)(d.r_atc + a) * Bu + bl) = Fa;
            }
            if (is_leader) at32(rows8, (uint32_t)d.q_etc * Bu + bl) = F;
        }
        {
            // others: (y, x[, power][, suppressant]) of every other agent; suppressants come from the other agent lanes
            float* const obs_others = reinterpret_cast<float*>(arena + d.off_obs_others);
            const int width = (A - 1) * d.others_k;
            const bool op = (flags & kObsPower) != 0, os = (flags & kObsSupp) != 0;
            float* others = obs_others + (a * B + (int64_t)bl) * (int64_t)width;
            int col = 0;
            for (int o = 0; o < A; ++o) {
                const float other_supp = os ? group_read<G>(supp, o + 1) : 0.0f;
                if (active && is_agent && o != a) {
                    others[col++] = (float)d.ay[o];
                    others[col++] = (float)d.ax[o];
                    if (op) others[col++] = d.power[o];
                    if (os) others[col++] = other_supp;
                }
            }
        }

        // -------------------------------------------------- inter-workgroup exclusive prefix (single pass, two levels)
        // chunk j needs the channel sums of all chunks < j =
        //     level 1: the chunks of its look-back group (GRP consecutive chunks of a round) that precede it
        //   + level 2: the totals of the groups of this round that precede its group (published by each group's last chunk)
        //   + the inclusive prefix published by the previous round's last chunk.
        // All those workgroups are co-resident (persistent grid <= resident capacity), so every wait terminates.
        constexpr int GRP = kBlock / NCHP;  // chunks per look-back group = predecessors one pass of the workgroup reads
        const int round_first = chunk - blockIdx.x;
        const int gi = (chunk - round_first) / GRP;
        const int group_first = round_first + gi * GRP;
        const bool round_last = blockIdx.x == gridDim.x - 1 || chunk == d.nchunks - 1;
        const bool group_last = chunk == group_first + GRP - 1 || round_last;
        bool timed_out = false;
        const int ch = tid & (NCHP - 1), slot = tid / NCHP;  // slot in [0, GRP)
        // ---- level 1: sums of the preceding chunks of my group; the group's last chunk publishes the group total at once
        // (it must not wait for level 2 first, or the groups of a round would form a serial chain)
        {
            const int pred = group_first + slot;
            uint32_t acc1 = 0;
            if (ch < nch && pred < chunk) acc1 = frz::granule_wait(agg + (int64_t)pred * nch + ch, tag, &timed_out);
#pragma unroll
            for (int dd = NCHP; dd < 64; dd <<= 1) acc1 += __shfl_xor(acc1, dd, 64);
            if (lane < NCHP) s_reduce[wave][lane] = acc1;
        }
        __syncthreads();
        uint32_t level1 = 0;
        if (tid < nch) {
#pragma unroll
            for (int j = 0; j < frz::kWaves; ++j) level1 += s_reduce[j][tid];
            if (group_last) frz::granule_store(gtot + (int64_t)group_first * nch + tid, tag, level1 + my_total);
        }
        // ---- level 2: totals of the preceding groups of this round + the previous round's inclusive prefix
        {
            uint32_t acc2 = 0;
            if (ch < nch)
                for (int pg = slot; pg < gi; pg += GRP) acc2 += frz::granule_wait(gtot + (int64_t)(round_first + pg * GRP) * nch + ch, tag, &timed_out);
            if (round_first > 0 && tid < nch) acc2 += frz::granule_wait(prefix + (int64_t)(round_first - 1) * nch + tid, tag, &timed_out);
#pragma unroll
            for (int dd = NCHP; dd < 64; dd <<= 1) acc2 += __shfl_xor(acc2, dd, 64);
            if (lane < NCHP) s_reduce2[wave][lane] = acc2;
        }
        __syncthreads();
        if (tid < nch) {
            uint32_t level2 = 0;
#pragma unroll
            for (int j = 0; j < frz::kWaves; ++j) level2 += s_reduce2[j][tid];
            const uint32_t exclusive = level1 + level2;
            s_prefix[tid] = exclusive;
            if (round_last) {
                frz::granule_store(prefix + (int64_t)chunk * nch + tid, tag, exclusive + my_total);
                if (chunk == d.nchunks - 1) cur[tid] = exclusive + my_total;  // batch totals, read by the next launch
            }
        }
        __syncthreads();
        if (timed_out) err |= FRZ_ERR_SCAN_TIMEOUT;

        // ------------------------------------------------------------------ jagged stores (values + offsets)
        {
            const int64_t cap = B * HW;
            const uint32_t my_off = (l <= A ? s_prefix[l] : 0u) + excl;  // global offset of this env in channel l's list
            const uint32_t off_f = group_read<G>(my_off, 0);
            int64_t* const task_values = reinterpret_cast<int64_t*>(arena + d.off_task_values);
            int64_t* const task_offsets = reinterpret_cast<int64_t*>(arena + d.off_task_offsets);
            int64_t* const obs_map = reinterpret_cast<int64_t*>(arena + d.off_obs_map);
            int64_t* const act_values = reinterpret_cast<int64_t*>(arena + d.off_act_values);
            int64_t* const act_offsets = reinterpret_cast<int64_t*>(arena + d.off_act_offsets);
            int64_t* const bad_values = reinterpret_cast<int64_t*>(arena + d.off_bad_values);
            int64_t* const bad_offsets = reinterpret_cast<int64_t*>(arena + d.off_bad_offsets);
            const bool show_bad = (flags & kShowBad) != 0;
            const uint32_t below = (1u << l) - 1u;
            const bool lit_here = active && ((lit1 >> l) & 1u);
            const int rank = __popc(lit1 & below);  // local task index of this cell
            if (active && is_leader) {
                task_offsets[b] = off_f;
                if (b == B - 1) task_offsets[B] = (int64_t)off_f + F;
            }
            if (lit_here) {
                int64_t* row = task_values + ((int64_t)off_f + rank) * 4;
                reinterpret_cast<longlong2*>(row)[0] = make_longlong2(my_yx >> 16, my_yx & 0xFFFF);
                reinterpret_cast<longlong2*>(row)[1] = make_longlong2(f, in);
                obs_map[(int64_t)off_f + rank] = rank;
            }
            if (active && is_agent) {
                act_offsets[a * (B + 1) + b] = my_off;
                if (b == B - 1) act_offsets[a * (B + 1) + B] = (int64_t)my_off + Fa;
                if (show_bad) {
                    bad_offsets[a * (B + 1) + b] = (int64_t)off_f - my_off;  // bad = listed but not attackable
                    if (b == B - 1) bad_offsets[a * (B + 1) + B] = ((int64_t)off_f - my_off) + (F - Fa);
                }
            }
            // each lit cell appends its local index to the lists of the agents that can (cannot) attack it
            for (int j = 0; j < A; ++j) {
                const uint32_t okj = group_read<G>(ok1, j + 1);
                const uint32_t offj = group_read<G>(my_off, j + 1);
                if (lit_here) {
                    if ((okj >> l) & 1u)
                        act_values[j * cap + (int64_t)offj + __popc(okj & below)] = rank;
                    else if (show_bad)
                        bad_values[j * cap + ((int64_t)off_f - offj) + __popc(~okj & lit1 & below)] = rank;
                }
            }
        }
        if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);

        // The workgroup owning the last chunk finished its look-back only after every other chunk published, i.e.
        // after every workgroup of this launch read the epoch: it can advance it for the next launch.
        if (chunk == d.nchunks - 1 && tid == 0) __hip_atomic_store(epoch_ptr, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int G>
void launch_g(const WfArgs& a, int grid, int rng, int mode, hipStream_t stream) {
    const WfDev* dev = reinterpret_cast<const WfDev*>(a.arena);
    if (mode == kRebuild) {
        hipLaunchKernelGGL((wf_group_kernel<G, FRZ_RNG_INJECTED, kRebuild>), dim3(grid), dim3(kBlock), 0, stream, a.arena, dev, a.actions,
                           a.field_rand, a.agent_rand);
    } else if (rng == FRZ_RNG_PHILOX) {
        hipLaunchKernelGGL((wf_group_kernel<G, FRZ_RNG_PHILOX, kStep>), dim3(grid), dim3(kBlock), 0, stream, a.arena, dev, a.actions,
                           a.field_rand, a.agent_rand);
    } else {
        hipLaunchKernelGGL((wf_group_kernel<G, FRZ_RNG_INJECTED, kStep>), dim3(grid), dim3(kBlock), 0, stream, a.arena, dev, a.actions,
                           a.field_rand, a.agent_rand);
    }
}

}  // namespace

namespace frz_wf {

int launch_group(const WfArgs& args, int G, int grid, int rng, int mode, hipStream_t stream) {
    if (G == 8)
        launch_g<8>(args, grid, rng, mode, stream);
    else
        return FRZ_E_INVALID;
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

int group_blocks_per_cu(int G) {
    int n = 0;
    hipError_t rc = hipErrorInvalidValue;
    if (G == 8) rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wf_group_kernel<8, FRZ_RNG_PHILOX, kStep>, kBlock, 0);
    return rc == hipSuccess ? n : 1;
}

}  // namespace frz_wf
