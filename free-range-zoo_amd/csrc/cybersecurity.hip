// cybersecurity.hip — fused cybersecurity environment step for gfx950 (MI355X), one environment per lane.
//
// One launch = one ParallelEnv.step() of the reference (cybersecurity.py:295-526) for the whole batch:
//   attacker / defender action decode (attack sets, patch sets, moves) -> movement -> presence (agent openness)
//   -> subnetwork danger-score transition -> criticality-weighted rewards -> truncation bookkeeping
//   -> update_observations (partial observability) + update_actions (presence-gated action mappings, compacted with
//      the launch-wide single-pass prefix scan of frz_scan.h).
//
// The state is dense and tiny (N nodes, D defenders, A agents per env): struct-of-arrays rows [k][B] in one device
// arena, every access a contiguous 256-byte segment per wavefront.  The danger score tanh((patches - attacks) / T) only
// depends on WHICH defenders patch and WHICH attackers attack a node (sums accumulated in agent order), so it is a
// 2^A-entry table built on the host with the same libm tanhf the oracle uses; the kernel looks it up (LDS when small).
// HBM-bound integer/byte work: no MFMA.  Built with -ffp-contract=off.
#include "frz_scan.h"
#include "frz_wave.h"

#include "../../include/frz.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

using frz::kBlock;

enum Mode { kStep = 0, kRebuild = 1 };
enum Flag : uint32_t {
    kStochState = 1u << 0, kShowBad = 1u << 1, kPartial = 1u << 2, kObsPower = 1u << 3, kObsPresence = 1u << 4, kObsLocation = 1u << 5,
    kTrackCumulative = 1u << 6, kTruncate = 1u << 7,
};
constexpr int kLdsLutEntries = 1024;  // danger tables up to this size are staged in LDS

struct CyDev {
    int32_t B, N, Att, D, A, S, nchunks, max_steps, lut_entries;
    uint32_t flags;
    float patch_reward;
    float threat[FRZ_MAX_AGENTS], mitigation[FRZ_MAX_AGENTS], persist[FRZ_MAX_AGENTS], back[FRZ_MAX_AGENTS];
    int32_t initial_presence[FRZ_MAX_AGENTS], initial_location[FRZ_MAX_AGENTS], initial_state[FRZ_MAX_NODES];
    int32_t criticality[FRZ_MAX_NODES];
    float state_rewards[FRZ_MAX_NETWORK_STATES];
    int32_t r_state, r_loc, r_last, r_moves, r_rewards, r_cum, r_atc, r_etc, r_seeds, r_mti, n_rows4;
    int32_t u_presence, u_term, u_trunc, u_frozen, n_rows1;
    int64_t off_rows4, off_rows1, off_self_att, off_self_def, off_others_att, off_others_def, off_tasks, off_act_values,
        off_act_offsets, off_obs_map, off_obs_map_offsets, off_lut, off_actions, off_error, off_epoch, off_totals, off_agg, off_prefix,
        off_rand_net, off_rand_agent, off_mt_state, total_bytes;
};
constexpr int64_t kDevBlockBytes = 4096;
static_assert(sizeof(CyDev) <= kDevBlockBytes, "configuration block too large");
constexpr int kCfgPieces = (int)((sizeof(CyDev) + 15) / 16);  // 16-byte pieces staged through LDS, one per thread
static_assert(kCfgPieces <= kBlock && kCfgPieces * 16 <= kDevBlockBytes, "configuration staged with one 16-byte load per thread");

// What the step kernel needs before the configuration block is staged (by value: kernel arguments are there at wave start,
// so the state loads and the epoch / totals words are in flight from the first instruction; see wildfire_roles.hip)
struct CyLaunch {
    int32_t B, N, Att, D, A;
    uint32_t ticketed;
    int64_t off_rows1, off_epoch, off_totals;
    // fused uniform random policy (frz_cybersecurity_step_random_policy): the actions are sampled in the step launch itself
    uint32_t policy, policy_seed_lo, policy_seed_hi, policy_step_lo, policy_step_hi;
    int32_t* actions_out;
    int64_t off_mt_state;  // FRZ_RNG_MT19937 inside the step: the per-env generator states, word j of env b at [j][b]
    int64_t off_lut;       // the danger table and its size (cy_roles_kernel requests it with its first loads)
    int32_t lut_entries;
    int32_t n_steps;       // multi-step launches (cy_roles_kernel<..., PERSIST>): steps of the rollout this launch performs
    int64_t copy_delta;    // multi-step launches: byte distance from the packed action-mapping values to their second copy
    // frz_cybersecurity_rollout (cy_roles_kernel<..., PERSIST, EXTRA>): what drives the steps and what they leave besides the last step's outputs
    uint32_t rollout_flags;     // FRZ_ROLLOUT_RESET_FIRST
    uint32_t pad_;
    int64_t tape_actions_step;  // elements between two steps of the action tape (`actions` = its step 0); 0: no tape
    int64_t list_record_delta;  // bytes from the arena's list block (off_act_values) to step 0's copy in the list record; 0: none
    int64_t list_record_step;
    float* reward_tape;         // optional float32 [n_steps][A][B]
    uint8_t* done_tape;         // optional uint8 [n_steps][2][B]
    int64_t actions_out_step;   // elements between two steps of actions_out (0: one buffer)
    int64_t obs_tape_delta;     // bytes from the arena's observation block (off_self_att) to step 0's copy in the observation tape; 0: none
    int64_t obs_tape_step;      // bytes between two steps' copies
    char* state_tape;           // optional: n_steps copies of (state rows [N + 2 D][B] int32, presence [A][B] uint8), state_tape_step bytes apart
    int64_t state_tape_step;
};

// (plain stores: written through, frz_device.h, the rows of this kernel gained nothing — 11.4 vs 11.3 us — most of its bytes are
// record-strided observation rows that need the L2 to merge them)
template <typename T>
__device__ __forceinline__ T& at32(T* base, uint32_t index) {
    return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (uint64_t)(index * (uint32_t)sizeof(T)));
}

// frz_cybersecurity_reset_masked: the state part of reset_batches (utils/env.py:162-189, cybersecurity.py:268-292) on the envs a device-side
// mask selects (mask == nullptr: the finished ones — all agents share one truncation value, terminations are never set); the rebuild
// launch that follows refreshes observations and mappings
__global__ void __launch_bounds__(kBlock) cy_masked_fill_kernel(char* arena, const uint8_t* mask, uint32_t seed_increment, frz_cybersecurity_saved_state saved) {
    const CyDev& d = *reinterpret_cast<const CyDev*>(arena);
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x, B = d.B;
    if (b >= B) return;
    int32_t* rows = reinterpret_cast<int32_t*>(arena + d.off_rows4);
    float* rowsf = reinterpret_cast<float*>(arena + d.off_rows4);
    uint8_t* rows1 = reinterpret_cast<uint8_t*>(arena + d.off_rows1);
    const bool selected = mask ? mask[b] != 0 : (rows1[d.u_term * B + b] != 0 || rows1[d.u_trunc * B + b] != 0);
    if (!selected) return;
    reinterpret_cast<uint32_t*>(rows)[d.r_seeds * B + b] += seed_increment;  // modulo 2^32
    const bool s = saved.network_state != nullptr;
    for (int n = 0; n < d.N; ++n)
        rows[(d.r_state + n) * B + b] = s ? saved.network_state[b * saved.network_state_stride_env + n * saved.network_state_stride_item] : d.initial_state[n];
    for (int k = 0; k < d.D; ++k) {
        rows[(d.r_loc + k) * B + b] = s ? saved.location[b * saved.location_stride_env + k * saved.location_stride_item] : d.initial_location[k];
        rows[(d.r_last + k) * B + b] = -2;
    }
    int2* const actions = reinterpret_cast<int2*>(arena + d.off_actions);
    for (int a = 0; a < d.A; ++a) {
        rows1[(d.u_presence + a) * B + b] =
            s ? (uint8_t)(saved.presence[b * saved.presence_stride_env + a * saved.presence_stride_item] != 0) : (uint8_t)(d.initial_presence[a] != 0);
        rowsf[(d.r_rewards + a) * B + b] = 0.0f;
        rowsf[(d.r_cum + a) * B + b] = 0.0f;
        rows1[(d.u_term + a) * B + b] = 0;
        rows1[(d.u_trunc + a) * B + b] = 0;
        actions[a * B + b] = make_int2(-2, -2);  // cybersecurity.py:233-236
    }
    rows[d.r_moves * B + b] = 0;
    rows1[d.u_frozen * B + b] = 0;
}

// cybersecurity.py:218-266 + utils/env.py:137-160
__global__ void __launch_bounds__(kBlock) cy_fill_kernel(char* arena) {
    const CyDev& d = *reinterpret_cast<const CyDev*>(arena);
    const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x, B = d.B;
    if (b >= B) return;
    int32_t* rows = reinterpret_cast<int32_t*>(arena + d.off_rows4);
    float* rowsf = reinterpret_cast<float*>(arena + d.off_rows4);
    uint8_t* rows1 = reinterpret_cast<uint8_t*>(arena + d.off_rows1);
    for (int n = 0; n < d.N; ++n) rows[(d.r_state + n) * B + b] = d.initial_state[n];
    for (int k = 0; k < d.D; ++k) {
        rows[(d.r_loc + k) * B + b] = d.initial_location[k];
        rows[(d.r_last + k) * B + b] = -2;  // cybersecurity.py:233-236
    }
    for (int a = 0; a < d.A; ++a) {
        rows1[(d.u_presence + a) * B + b] = (uint8_t)(d.initial_presence[a] != 0);
        rowsf[(d.r_rewards + a) * B + b] = 0.0f;
        rowsf[(d.r_cum + a) * B + b] = 0.0f;
        rows1[(d.u_term + a) * B + b] = 0;
        rows1[(d.u_trunc + a) * B + b] = 0;
    }
    rows[d.r_moves * B + b] = 0;
    rows1[d.u_frozen * B + b] = 0;
}

// ATT >= 0: an exact instantiation — the env has exactly NMAX nodes, ATT attackers and AMAX - ATT defenders, so every shape test
// below folds at compile time (the runtime-shape instantiations, ATT = -1, pay for them with select chains: ~25 % of their
// instructions)
template <int NMAX, int AMAX, int ATT, int RNG, int MODE>
__global__ void __launch_bounds__(kBlock) cy_step_kernel(char* __restrict__ arena, const CyDev* __restrict__ dev,
                                                          const int32_t* __restrict__ actions, const float* __restrict__ net_rand,
                                                          const float* __restrict__ agent_rand, const CyLaunch L) {
    __shared__ frz::ScanShared<AMAX> s_scan;
    __shared__ int s_ticket;
    __shared__ float s_lut[kLdsLutEntries];
    __shared__ uint4 s_cfg[kCfgPieces];  // the configuration block; per-lane-indexed tables (state rewards) are read from here

    const int tid = threadIdx.x;
    // the configuration piece is the kernel's first vector-memory instruction (waiting for it must not wait for the state loads)
    const uint4 cfg_piece = tid < kCfgPieces ? reinterpret_cast<const uint4*>(dev)[tid] : make_uint4(0, 0, 0, 0);
    const int64_t B = L.B;
    const uint32_t Bu = (uint32_t)L.B;
    const int N = ATT >= 0 ? NMAX : L.N, Att = ATT >= 0 ? ATT : L.Att, D = ATT >= 0 ? AMAX - ATT : L.D, A = ATT >= 0 ? AMAX : L.A;
    const int nchunks = (int)((B + kBlock - 1) / kBlock);
    // rows of the [rows][B] block: a fixed function of (N, D, A) (frz_cybersecurity_create lays them out in this order)
    const int r_state = 0, r_loc = N, r_last = N + D, r_moves = N + 2 * D, r_seeds = r_moves + 3 * A + 2;
    const uint32_t u_presence = 0, u_trunc = 2u * (uint32_t)A;
    int32_t* const rows = reinterpret_cast<int32_t*>(arena + kDevBlockBytes);
    float* const rowsf = reinterpret_cast<float*>(arena + kDevBlockBytes);
    uint8_t* const rows1 = reinterpret_cast<uint8_t*>(arena + L.off_rows1);

    frz::ScanWorkspace ws{reinterpret_cast<uint32_t*>(arena + L.off_epoch), reinterpret_cast<uint32_t*>(arena + L.off_totals), nullptr, nullptr};
    const int chunk = frz::scan_take_chunk(ws, nchunks, L.ticketed != 0, &s_ticket);
    const frz::ScanLaunch launch = frz::scan_begin(ws);
    const int64_t b = (int64_t)chunk * kBlock + tid;
    const bool active = b < B;
    const uint32_t bl = (uint32_t)(active ? b : B - 1);

    // ---------------------------------------------------------------------------------------------- load state
    // (issued before the configuration is staged; lanes past the end shadow the last env and store nothing)
    // Every load below is UNCONDITIONAL: rows past the env's node / defender / agent count are read from the last valid row and
    // replaced afterwards.  A load under `if (a < A)` becomes its own basic block that ends in a wait for everything issued so far:
    // one memory round trip per agent instead of one for the whole prologue.
    int state[NMAX], loc[AMAX], last[AMAX];
    uint32_t pres_raw[AMAX];
    float cum_in[AMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) state[n] = at32(rows, (uint32_t)(r_state + min(n, N - 1)) * Bu + bl);
#pragma unroll
    for (int k = 0; k < AMAX; ++k) {
        loc[k] = at32(rows, (uint32_t)(r_loc + min(k, D - 1)) * Bu + bl);
        last[k] = at32(rows, (uint32_t)(r_last + min(k, D - 1)) * Bu + bl);
    }
#pragma unroll
    for (int a = 0; a < AMAX; ++a) pres_raw[a] = at32(rows1, (u_presence + (uint32_t)min(a, A - 1)) * Bu + bl);
    const uint32_t trunc_raw = at32(rows1, u_trunc * Bu + bl);
    int nm = 0, mti = 0;
    uint32_t seed = 0;
    int2 act_in[AMAX];
    float r_net_in[NMAX], r_agent_in[AMAX];
    if (MODE == kStep) {
        nm = at32(rows, (uint32_t)r_moves * Bu + bl);
        if (RNG == FRZ_RNG_PHILOX || L.policy) seed = (uint32_t)at32(rows, (uint32_t)r_seeds * Bu + bl);
        if (RNG == FRZ_RNG_MT19937) mti = at32(rows, (uint32_t)(r_seeds + 1) * Bu + bl);  // position of the env's MT19937 stream
        if (!L.policy) {
#pragma unroll
            for (int a = 0; a < AMAX; ++a) act_in[a] = reinterpret_cast<const int2*>(actions)[(int64_t)min(a, A - 1) * B + bl];
        } else {
#pragma unroll
            for (int a = 0; a < AMAX; ++a) act_in[a] = make_int2(0, -1);
        }
        // cumulative rewards (row block r_cum = r_moves + 1 + A, frz_cybersecurity_create): read here, added and stored at the end
#pragma unroll
        for (int a = 0; a < AMAX; ++a) cum_in[a] = at32(rowsf, (uint32_t)(r_moves + 1 + A + min(a, A - 1)) * Bu + bl);
        if (RNG == FRZ_RNG_INJECTED) {
#pragma unroll
            for (int n = 0; n < NMAX; ++n) r_net_in[n] = net_rand[(int64_t)bl * N + min(n, N - 1)];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) r_agent_in[a] = agent_rand[(int64_t)bl * A + min(a, A - 1)];
        }
    }

#pragma unroll
    for (int n = 0; n < NMAX; ++n) state[n] = n < N ? state[n] : 0;
#pragma unroll
    for (int k = 0; k < AMAX; ++k) {
        loc[k] = k < D ? loc[k] : -1;
        last[k] = k < D ? last[k] : -2;
    }
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
        pres_raw[a] = a < A ? pres_raw[a] : 0u;
        if (MODE == kStep) act_in[a] = a < A ? act_in[a] : make_int2(0, -1);
    }

    if (tid < kCfgPieces) s_cfg[tid] = cfg_piece;
    __syncthreads();
    const CyDev* const cfg_lds = reinterpret_cast<const CyDev*>(s_cfg);
    const CyDev d = *cfg_lds;  // hot scalars end up in registers
    const uint32_t flags = d.flags;
    const float* lut = reinterpret_cast<const float*>(arena + d.off_lut);
    const bool lut_in_lds = d.lut_entries <= kLdsLutEntries;
    if (MODE == kStep) {
        if (lut_in_lds)
            for (int i = tid; i < d.lut_entries; i += kBlock) s_lut[i] = lut[i];
    }
    ws.agg = reinterpret_cast<uint64_t*>(arena + d.off_agg);
    ws.prefix = reinterpret_cast<uint64_t*>(arena + d.off_prefix);

    // utils/env.py:211-213: no-op once ALL envs are terminated (never, cybersecurity.py:298) or ALL are truncated;
    // totals channels A / A + 1 = number of envs not terminated / not truncated after the previous launch
    bool frozen = false;
    if (MODE == kStep) frozen = launch.prev[A] == 0u || launch.prev[A + 1] == 0u;
    __syncthreads();

    {  // one chunk per workgroup (no chunk loop: see wildfire_roles.hip)

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

        bool pres[AMAX];
#pragma unroll
        for (int a = 0; a < AMAX; ++a) pres[a] = pres_raw[a] != 0;
        bool trunc = trunc_raw != 0;
        uint32_t err = 0;

        if (MODE == kStep) {
            if (L.policy) {
                // uniform member of each agent's OneOf action space, the stream of cy_policy_kernel / frz_cybersecurity_random_policy:
                // agent a draws word a % 4 of Philox(counter (a / 4, 0, step lo, step hi), key (seed lo ^ env seed, seed hi)); the agent's
                // task count is what the previous launch published (N while present, everything with show_bad_actions)
                frz::Philox4 policy_words[(AMAX + 3) / 4];
#pragma unroll
                for (int q = 0; q < (AMAX + 3) / 4; ++q)
                    if (q * 4 < A) policy_words[q] = frz::philox4x32_10((uint32_t)q, 0u, L.policy_step_lo, L.policy_step_hi, L.policy_seed_lo ^ seed,
                                                                       L.policy_seed_hi);
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    if (a < A) {
                        const int n = (flags & kShowBad) ? N : (pres[a] ? N : 0);
                        int tail1 = -3, nt = 1;
                        if (a >= Att && n > 0) {
                            int home_loc = 0;
#pragma unroll
                            for (int k = 0; k < AMAX; ++k) home_loc = (a == Att + k) ? loc[k] : home_loc;
                            const bool patchable = (flags & kShowBad) || home_loc != -1;
                            tail1 = patchable ? -2 : -3;
                            nt = patchable ? 3 : 2;
                        }
                        const int j = (int)(((uint64_t)policy_words[a >> 2].w[a & 3] * (uint64_t)(n + nt)) >> 32);
                        const int value = j < n ? 0 : (j - n == 0 ? -1 : (j - n == 1 ? tail1 : -3));
                        act_in[a] = make_int2(j, value);
                        if (active) reinterpret_cast<int2*>(L.actions_out)[(int64_t)a * B + b] = act_in[a];
                    }
                }
            }
            // ---------------------------------------------------------------------------------- randomness
            float r_net[NMAX], r_agent[AMAX];
            if (RNG == FRZ_RNG_INJECTED) {
#pragma unroll
                for (int n = 0; n < NMAX; ++n) r_net[n] = r_net_in[n];
#pragma unroll
                for (int a = 0; a < AMAX; ++a) r_agent[a] = r_agent_in[a];
            } else if constexpr (RNG == FRZ_RNG_MT19937) {
                // The env's own MT19937 stream (mt19937.hip: state word j of env b at [j][b], twisted lazily, one word per draw),
                // bit-identical to the reference's per-env torch CPU generator: generate(B, 1, (N,)) then generate(B, 1, (A,))
                // (cybersecurity.py:304-315), i.e. node n is draw n and agent a is draw N + a of the step's N + A consecutive floats.
                // N + A <= 227, so no word read here is rewritten by this batch: every load is issued before the first use
                // (wildfire_roles.hip has the same generator; a frozen step returns above and leaves the stream alone).
                constexpr int U = NMAX + AMAX, kN = 624, kM = 397;
                static_assert(U <= kN - kM, "a batch must not read a word it rewrites");
                uint32_t* const mt = reinterpret_cast<uint32_t*>(arena + L.off_mt_state);
                const int used = N + A;
                uint32_t w[U + 1], far[U];
#pragma unroll
                for (int k = 0; k <= U; ++k) {
                    int j = mti + k;
                    j -= j >= kN ? kN : 0;
                    w[k] = mt[(int64_t)j * B + bl];
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    int j = mti + k + kM;
                    j -= j >= kN ? kN : 0;
                    j -= j >= kN ? kN : 0;
                    far[k] = mt[(int64_t)j * B + bl];
                }
                float uni[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const uint32_t y = (w[k] & 0x80000000u) | (w[k + 1] & 0x7fffffffu);
                    uint32_t v = far[k] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                    int j = mti + k;
                    j -= j >= kN ? kN : 0;
                    if (active && k < used) mt[(int64_t)j * B + bl] = v;
                    v ^= v >> 11;
                    v ^= (v << 7) & 0x9d2c5680u;
                    v ^= (v << 15) & 0xefc60000u;
                    v ^= v >> 18;
                    uni[k] = (float)(v & 0xFFFFFFu) * (1.0f / 16777216.0f);
                }
                if (active) {
                    int j = mti + used;
                    j -= j >= kN ? kN : 0;
                    at32(rows, (uint32_t)(r_seeds + 1) * Bu + bl) = j;
                }
#pragma unroll
                for (int n = 0; n < NMAX; ++n) r_net[n] = uni[n];
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    float u = 0.0f;
#pragma unroll
                    for (int k = 0; k < U; ++k) u = (k == N + a) ? uni[k] : u;
                    r_agent[a] = u;
                }
            } else {  // FRZ_RNG_PHILOX stream of include/frz.h
#pragma unroll
                for (int q = 0; q < (NMAX + 3) / 4; ++q) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (q * 4 + j < NMAX) r_net[q * 4 + j] = 0.0f;
                    if (q * 4 < N && (flags & kStochState)) {
                        const frz::Philox4 w = frz::philox4x32_10((uint32_t)q, (uint32_t)nm, 0u, 0u, seed, 0x46525A01u);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (q * 4 + j < NMAX) r_net[q * 4 + j] = frz::u32_to_unit_float(w.w[j]);
                    }
                }
#pragma unroll
                for (int q = 0; q < (AMAX + 3) / 4; ++q) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (q * 4 + j < AMAX) r_agent[q * 4 + j] = 0.0f;
                    if (q * 4 < A) {
                        const frz::Philox4 w = frz::philox4x32_10((uint32_t)q, (uint32_t)nm, 1u, 0u, seed, 0x46525A01u);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (q * 4 + j < AMAX) r_agent[q * 4 + j] = frz::u32_to_unit_float(w.w[j]);
                    }
                }
            }

            // ----------------------------------------------- action decode (cybersecurity.py:326-384), agent order
            uint32_t attack_set[NMAX], patch_set[NMAX];  // bit a: attacker a attacks / defender a patches this node
#pragma unroll
            for (int n = 0; n < NMAX; ++n) attack_set[n] = patch_set[n] = 0u;
            float rew[AMAX];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                rew[a] = 0.0f;
                if (a < A) {
                    const int2 v = act_in[a];
                    const int idx = v.x, act = v.y;
                    const bool bad_target = act == 0 && (idx < 0 || idx >= N);  // reference raises ValueError (:341-346, :358-363)
                    if (bad_target && active) err |= FRZ_ERR_INVALID_TARGET;
                    if (!bad_target && !(flags & kShowBad) && !pres[a] && act != -1 && active) err |= FRZ_ERR_ABSENT_ACTION;
                    if (a < Att) {
                        const bool attack = act == 0 && !bad_target;  // no presence check (:348-350)
#pragma unroll
                        for (int n = 0; n < NMAX; ++n) attack_set[n] |= (attack && idx == n) ? (1u << a) : 0u;
                    }
                }
            }
            // defenders: lane-local index k = a - Att (separate static loop keeps every array index a compile-time constant)
#pragma unroll
            for (int k = 0; k < AMAX; ++k) {
                if (k < D) {
                    int2 v = make_int2(0, -1);
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) v = (a == Att + k) ? act_in[a] : v;
                    const int idx = v.x, act = v.y;
                    const bool bad_target = act == 0 && (idx < 0 || idx >= N);
                    const bool move = act == 0 && !bad_target;
                    const bool patch = act == -2 && loc[k] != -1 && !bad_target;  // at the CURRENT (pre-move) location (:354)
#pragma unroll
                    for (int n = 0; n < NMAX; ++n) patch_set[n] |= (patch && loc[k] == n) ? (1u << k) : 0u;
                    // patch reward lands in the defender's reward slot Att + k; bad_patch (:380-382) can never apply
                    const float pr = patch ? d.patch_reward : 0.0f;
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) rew[a] = (a == Att + k) ? rew[a] + pr : rew[a];
                    loc[k] = move ? idx : loc[k];             // transitions/movement.py:30
                    last[k] = bad_target ? last[k] : act;
                }
            }
            // ------------------------------------------------- presence (transitions/presence.py:46-58), same draw for both tests
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (a < A) {
                    const bool ret = !pres[a] && r_agent[a] < d.back[a];
                    const bool leave = pres[a] && r_agent[a] >= d.persist[a];
                    pres[a] = ret ? true : (leave ? false : pres[a]);
#pragma unroll
                    for (int k = 0; k < AMAX; ++k) loc[k] = (ret && a == Att + k) ? -1 : loc[k];  // returning defenders start at home
                }
            }
            // ------------------------------------------------- subnetwork transition (transitions/subnetwork.py:53-70)
            float net_reward = 0.0f;
#pragma unroll
            for (int n = 0; n < NMAX; ++n) {
                if (n < N) {
                    const uint32_t index = (patch_set[n] << Att) | attack_set[n];
                    const float danger = lut_in_lds ? s_lut[index] : lut[index];
                    bool better = danger > 0.0f, worse = danger < 0.0f;
                    if (flags & kStochState) {  // larger |danger| => LESS likely to move, as written (:57-59)
                        const bool gate = fabsf(danger) <= r_net[n];
                        better = better && gate;
                        worse = worse && gate;
                    }
                    int s = state[n] - (better ? 1 : 0) + (worse ? 1 : 0);
                    s = s < 0 ? 0 : (s > d.S - 1 ? d.S - 1 : s);
                    state[n] = s;
                    // :396-399 criticality-weighted state rewards, sequential float32 dot product
                    net_reward = __fadd_rn(net_reward, __fmul_rn(cfg_lds->state_rewards[s], (float)d.criticality[n]));
                }
            }
            nm += 1;
            trunc = (flags & kTruncate) ? nm >= d.max_steps : trunc;

            if (active) {
#pragma unroll
                for (int n = 0; n < NMAX; ++n)
                    if (n < N) at32(rows, (uint32_t)(d.r_state + n) * Bu + bl) = state[n];
#pragma unroll
                for (int k = 0; k < AMAX; ++k)
                    if (k < D) {
                        at32(rows, (uint32_t)(d.r_loc + k) * Bu + bl) = loc[k];
                        at32(rows, (uint32_t)(d.r_last + k) * Bu + bl) = last[k];
                    }
#pragma unroll
                for (int a = 0; a < AMAX; ++a)
                    if (a < A) {
                        const float r = a < Att ? __fadd_rn(rew[a], __fmul_rn(net_reward, -1.0f)) : __fadd_rn(rew[a], net_reward);
                        at32(rows1, (uint32_t)(d.u_presence + a) * Bu + bl) = (uint8_t)pres[a];
                        at32(rowsf, (uint32_t)(d.r_rewards + a) * Bu + bl) = r;
                        if (flags & kTruncate) at32(rows1, (uint32_t)(d.u_trunc + a) * Bu + bl) = (uint8_t)trunc;
                        if (flags & kTrackCumulative) at32(rowsf, (uint32_t)(d.r_cum + a) * Bu + bl) = __fadd_rn(cum_in[a], r);
                    }
                at32(rows, (uint32_t)d.r_moves * Bu + bl) = nm;
            }
        }

        // ======================================================================================================
        // update_observations + update_actions (cybersecurity.py:413-526)
        // ======================================================================================================
        uint32_t cnt[AMAX], excl[AMAX];
#pragma unroll
        for (int a = 0; a < AMAX; ++a) cnt[a] = (active && a < A && pres[a]) ? 1u : 0u;
        frz::scan_chunk<AMAX>(s_scan, ws, launch, cnt, active, active && !trunc, A, chunk, nchunks, excl, &err);

        if (active) {
            float* const self_att = reinterpret_cast<float*>(arena + d.off_self_att);
            float* const self_def = reinterpret_cast<float*>(arena + d.off_self_def);
            float* const others_att = reinterpret_cast<float*>(arena + d.off_others_att);
            float* const others_def = reinterpret_cast<float*>(arena + d.off_others_def);
            int64_t* const tasks = reinterpret_cast<int64_t*>(arena + d.off_tasks);
            int32_t* const act_values = reinterpret_cast<int32_t*>(arena + d.off_act_values);
            int64_t* const act_offsets = reinterpret_cast<int64_t*>(arena + d.off_act_offsets);
            const bool op = (flags & kObsPower) != 0, opr = (flags & kObsPresence) != 0, ol = (flags & kObsLocation) != 0;
            const int ka = (op ? 1 : 0) + (opr ? 1 : 0), kd = ka + (ol ? 1 : 0);
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (a < Att) {  // attackers: (threat, presence) (:481-484)
                    reinterpret_cast<float2*>(self_att)[a * B + b] = make_float2(d.threat[a], pres[a] ? 1.0f : 0.0f);
                    float* others = others_att + (a * B + b) * (int64_t)((Att - 1) * ka);
                    int col = 0;
#pragma unroll
                    for (int o = 0; o < AMAX; ++o)
                        if (o < Att && o != a) {
                            if (op) others[col++] = d.threat[o];
                            if (opr) others[col++] = pres[o] ? 1.0f : 0.0f;
                        }
                }
            }
#pragma unroll
            for (int k = 0; k < AMAX; ++k) {
                if (k < D) {  // defenders: (mitigation, presence, location) (:475-479)
                    bool present_k = false;
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) present_k = (a == Att + k) ? pres[a] : present_k;
                    float* self = self_def + (k * B + b) * 3;
                    self[0] = d.mitigation[k];
                    self[1] = present_k ? 1.0f : 0.0f;
                    self[2] = (float)loc[k];
                    float* others = others_def + (k * B + b) * (int64_t)((D - 1) * kd);
                    int col = 0;
#pragma unroll
                    for (int o = 0; o < AMAX; ++o)
                        if (o < D && o != k) {
                            bool present_o = false;
#pragma unroll
                            for (int a = 0; a < AMAX; ++a) present_o = (a == Att + o) ? pres[a] : present_o;
                            if (op) others[col++] = d.mitigation[o];
                            if (opr) others[col++] = present_o ? 1.0f : 0.0f;
                            if (ol) others[col++] = (float)loc[o];
                        }
                }
            }
            // tasks (state, criticality) per agent; a defender sees them only right after monitoring (:497, :510-511)
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (a < A) {
                    bool hidden = false;
                    if (flags & kPartial) {
#pragma unroll
                        for (int k = 0; k < AMAX; ++k) hidden = (a == Att + k && k < D) ? last[k] != -3 : hidden;
                    }
                    int64_t* t = tasks + (a * B + b) * (int64_t)(N * 2);
#pragma unroll
                    for (int n = 0; n < NMAX; ++n)
                        if (n < N)
                            reinterpret_cast<longlong2*>(t)[n] =
                                hidden ? make_longlong2(-100, -100) : make_longlong2(state[n], d.criticality[n]);
                    // action mapping: arange(N) while present, empty otherwise (:441-457)
                    const int64_t off = (int64_t)excl[a] * N;
                    act_offsets[a * (B + 1) + b] = off;
                    if (b == B - 1) act_offsets[a * (B + 1) + B] = off + (pres[a] ? N : 0);
                    // (the packed values are position mod N wherever a segment lies — every segment is arange(N) and starts at a multiple
                    // of N: cy_prefill_kernel wrote them at bind, a step only moves the offsets)
                    if (pres[a] && MODE != kStep) {
                        int32_t* v = act_values + a * B * N + off;
#pragma unroll
                        for (int n = 0; n < NMAX; ++n)
                            if (n < N) v[n] = n;
                    }
                    at32(rows, (uint32_t)(d.r_atc + a) * Bu + bl) = pres[a] ? N : 0;
                }
            }
            if (MODE == kRebuild) {  // constants of the env: written by reset()/rebuild() only
                int32_t* const obs_map = reinterpret_cast<int32_t*>(arena + d.off_obs_map);
                int64_t* const obs_map_offsets = reinterpret_cast<int64_t*>(arena + d.off_obs_map_offsets);
                for (int n = 0; n < N; ++n) obs_map[b * N + n] = n;
                obs_map_offsets[b] = b;
                if (b == B - 1) obs_map_offsets[B] = B;
                at32(rows, (uint32_t)d.r_etc * Bu + bl) = N;
            }
        }
        if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);
        frz::scan_end(ws, launch, chunk, nchunks);
    }
}

// ----------------------------------------------------------------------------------------------------------------
// The step as two roles (the shapes up to 8 nodes / 8 agents).  At 65 536 envs the lane-per-env kernel above is one
// wavefront per SIMD and what it waits for is itself: ~2 200 dependent instructions between a memory round trip at the
// start and the stores at the end.  Here a 512-thread workgroup owns the chunk; lane i of wavefront w and of wavefront
// w + 4 hold the same env:
//   * wavefronts 0-3 (**state**): loads, fused policy, action decode, presence, subnetwork transition, rewards and
//     state rows, then the scan of the presence counts and the action mappings;
//   * wavefronts 4-7 (**view**): the step's random draws (Philox blocks / the env's MT19937 stream / the injected
//     rows) while the state role decodes, then — from the post-transition state the state role leaves in LDS —
//     every observation row (self, others, tasks), i.e. two thirds of the step's stores.
// Three workgroup barriers (configuration staged, draws ready, state ready) + the four of the scan, which the view
// role passes between its store groups.
// ----------------------------------------------------------------------------------------------------------------
constexpr int kRoleBlock = 2 * kBlock;
#ifndef FRZ_CY_TASKS_LDS
#ifndef FRZ_CY_ROWS_LDS
#define FRZ_CY_ROWS_LDS 0
#endif
#define FRZ_CY_TASKS_LDS 1  // the view role's task rows staged through LDS into whole lines (0: a lane stores its own rows; profiles/r04_experiments.txt)
#endif

// PERSIST (FRZ_RNG_PHILOX, fused policy): L.n_steps steps in ONE launch, the state role keeping its envs in registers from step to step
// (frz_cybersecurity_rollout_random_policy).  Same scheme as the wildfire field/crew kernel (wildfire_roles.hip, PERSIST): env-indexed
// outputs are rewritten by the same workgroup every step; the packed action mappings go to the caller's buffers at the last step only
// and to a second copy, which nobody reads, before it; the batch totals of the step that just ended (all-truncated test) arrive as the
// last chunk's inclusive-prefix granules — requested at the top of a step, looked at before the step's first store.
// EXTRA (multi-step launches): the options of a frz_rollout_spec beyond the plain random-policy rollout — an action tape, randomness tapes
// (FRZ_RNG_INJECTED), reward / done / action records, the list record, the opening reset — as a separate instantiation (see wildfire_roles.inl).
template <int NMAX, int AMAX, int ATT, int RNG, bool PERSIST = false, bool EXTRA = false>
__global__ void __launch_bounds__(kRoleBlock, 2) cy_roles_kernel(char* __restrict__ arena, const CyDev* __restrict__ dev,
                                                              const int32_t* __restrict__ actions, const float* __restrict__ net_rand,
                                                              const float* __restrict__ agent_rand, const CyLaunch L) {
    static_assert(AMAX <= 8, "the danger table is staged with one load per state-role thread");
    static_assert(!PERSIST || RNG == FRZ_RNG_PHILOX || (RNG == FRZ_RNG_INJECTED && EXTRA), "multi-step launches draw with Philox, or from tapes");
    static_assert(!EXTRA || PERSIST, "the rollout options belong to the multi-step launch");
    const bool reset_first = EXTRA && (L.rollout_flags & FRZ_ROLLOUT_RESET_FIRST) != 0;
    __shared__ frz::ScanShared<AMAX> s_scan;
    __shared__ int s_ticket;
    __shared__ float s_lut[1 << AMAX];
    __shared__ uint4 s_cfg[kCfgPieces];
    __shared__ float s_draw[NMAX + AMAX][kBlock];     // view -> state: this step's uniforms (nodes, then agents)
    __shared__ int s_post[NMAX + 2 * AMAX + 1][kBlock];  // state -> view: state, location, last action, presence bits after the step
    __shared__ int s_stop;  // multi-step launches: the workgroup's verdict on "every env is finished" (one per workgroup: both roles leave at the same barrier)
#if FRZ_CY_TASKS_LDS
    // view role: one agent's rows (tasks: N x 16 bytes per env; self / others: up to 3 (AMAX - 1) floats) of a wavefront's 64 envs, in output order
    __shared__ longlong2 s_task_stage[4][64 * (NMAX > (3 * (AMAX > 1 ? AMAX - 1 : 1) + 3) / 4 ? NMAX : (3 * (AMAX > 1 ? AMAX - 1 : 1) + 3) / 4)];
#endif

    const int tid = threadIdx.x & (kBlock - 1);
    const bool view = threadIdx.x >= kBlock;  // wave-uniform
    const uint4 cfg_piece = threadIdx.x < kCfgPieces ? reinterpret_cast<const uint4*>(dev)[threadIdx.x] : make_uint4(0, 0, 0, 0);
    const float lut_piece = (!view && tid < L.lut_entries) ? reinterpret_cast<const float*>(arena + L.off_lut)[tid] : 0.0f;
    const int64_t B = L.B;
    const uint32_t Bu = (uint32_t)L.B;
    const int N = ATT >= 0 ? NMAX : L.N, Att = ATT >= 0 ? ATT : L.Att, D = ATT >= 0 ? AMAX - ATT : L.D, A = ATT >= 0 ? AMAX : L.A;
    const int nchunks = (int)((B + kBlock - 1) / kBlock);
    const int r_state = 0, r_loc = N, r_last = N + D, r_moves = N + 2 * D, r_seeds = r_moves + 3 * A + 2;
    const uint32_t u_presence = 0, u_trunc = 2u * (uint32_t)A;
    int32_t* const rows = reinterpret_cast<int32_t*>(arena + kDevBlockBytes);
    float* const rowsf = reinterpret_cast<float*>(arena + kDevBlockBytes);
    uint8_t* const rows1 = reinterpret_cast<uint8_t*>(arena + L.off_rows1);

    frz::ScanWorkspace ws{reinterpret_cast<uint32_t*>(arena + L.off_epoch), reinterpret_cast<uint32_t*>(arena + L.off_totals), nullptr, nullptr};
    const int chunk = frz::scan_take_chunk(ws, nchunks, L.ticketed != 0, &s_ticket);
    const frz::ScanLaunch launch = frz::scan_begin(ws);
    const int64_t b = (int64_t)chunk * kBlock + tid;
    const bool active = b < B;
    const uint32_t bl = (uint32_t)(active ? b : B - 1);

    // ------------------------------------------------------------------------------------ loads of both roles (unconditional, see above)
    int state[NMAX], loc[AMAX], last[AMAX];
    uint32_t pres_raw[AMAX];
    float cum_in[AMAX];
    uint32_t trunc_raw = 0;
    int2 act_in[AMAX];
    float r_in[NMAX + AMAX];
    const int nm_in = reset_first ? 0 : (int)at32(rows, (uint32_t)r_moves * Bu + bl);
    const uint32_t seed = (RNG == FRZ_RNG_PHILOX || L.policy) ? (uint32_t)at32(rows, (uint32_t)r_seeds * Bu + bl) : 0u;
    int mti = 0;
    if (!view) {
#pragma unroll
        for (int n = 0; n < NMAX; ++n) state[n] = at32(rows, (uint32_t)(r_state + min(n, N - 1)) * Bu + bl);
#pragma unroll
        for (int k = 0; k < AMAX; ++k) {
            loc[k] = at32(rows, (uint32_t)(r_loc + min(k, D - 1)) * Bu + bl);
            last[k] = at32(rows, (uint32_t)(r_last + min(k, D - 1)) * Bu + bl);
        }
#pragma unroll
        for (int a = 0; a < AMAX; ++a) pres_raw[a] = at32(rows1, (u_presence + (uint32_t)min(a, A - 1)) * Bu + bl);
        trunc_raw = at32(rows1, u_trunc * Bu + bl);
        if (!L.policy) {
#pragma unroll
            for (int a = 0; a < AMAX; ++a) act_in[a] = reinterpret_cast<const int2*>(actions)[(int64_t)min(a, A - 1) * B + bl];
        }
#pragma unroll
        for (int a = 0; a < AMAX; ++a) cum_in[a] = at32(rowsf, (uint32_t)(r_moves + 1 + A + min(a, A - 1)) * Bu + bl);
    } else {
        if (RNG == FRZ_RNG_MT19937) mti = at32(rows, (uint32_t)(r_seeds + 1) * Bu + bl);
        if (RNG == FRZ_RNG_INJECTED) {
#pragma unroll
            for (int n = 0; n < NMAX; ++n) r_in[n] = net_rand[(int64_t)bl * N + min(n, N - 1)];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) r_in[NMAX + a] = agent_rand[(int64_t)bl * A + min(a, A - 1)];
        }
    }

    if (threadIdx.x < kCfgPieces) s_cfg[threadIdx.x] = cfg_piece;
    if (!view && tid < (1 << AMAX)) s_lut[tid] = lut_piece;
    __syncthreads();  // (1) configuration and danger table staged
    const CyDev* const cfg_lds = reinterpret_cast<const CyDev*>(s_cfg);
    const CyDev d = *cfg_lds;
    const uint32_t flags = d.flags;
    ws.agg = reinterpret_cast<uint64_t*>(arena + d.off_agg);
    ws.prefix = reinterpret_cast<uint64_t*>(arena + d.off_prefix);
    if constexpr (EXTRA) {
        if (reset_first && !view) {  // cybersecurity.py:218-266 + utils/env.py:137-160, in registers (the rows are written by the first step)
#pragma unroll
            for (int n = 0; n < NMAX; ++n) state[n] = cfg_lds->initial_state[n < N ? n : 0];
#pragma unroll
            for (int k = 0; k < AMAX; ++k) loc[k] = cfg_lds->initial_location[k < D ? k : 0], last[k] = -2;
#pragma unroll
            for (int a = 0; a < AMAX; ++a) pres_raw[a] = cfg_lds->initial_presence[a < A ? a : 0] != 0 ? 1u : 0u, cum_in[a] = 0.0f;
            trunc_raw = 0;
            if (active) at32(rows1, (uint32_t)d.u_frozen * Bu + bl) = 0;
        }
    }

    if (!reset_first && (launch.prev[A] == 0u || launch.prev[A + 1] == 0u)) {  // frozen batch (utils/env.py:211-213): see cy_step_kernel
        if (!view && active && !at32(rows1, (uint32_t)d.u_frozen * Bu + bl)) {
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

    // ---- the steps of this launch (one, or L.n_steps of a multi-step launch)
    const int n_steps = PERSIST ? L.n_steps : 1;
    const int nch_total = A + 2;
    uint32_t epoch_now = launch.epoch;
    bool gave_up = false, finished = false;
    uint64_t requested[AMAX + 2];
    auto request_totals = [&]() {  // granules of the last chunk's inclusive prefix = the batch totals of the step that just ended
        const uint64_t* const last = ws.prefix + (int64_t)(nchunks - 1) * nch_total;
#pragma unroll
        for (int i = 0; i < AMAX + 2; ++i) requested[i] = frz::granule_load(last + (i < nch_total ? i : 0));
    };
    auto await_totals = [&]() {  // -> finished: every env truncated (or terminated) after that step: utils/env.py:211-213
        epoch_now += 1u;
        const uint32_t ended = epoch_now;  // the tag that step published under
        const uint64_t* const last = ws.prefix + (int64_t)(nchunks - 1) * nch_total;
        uint32_t not_terminated = 1, not_truncated = 1;
        for (int spin = 0;; ++spin) {  // bounded
            bool all = true;
#pragma unroll
            for (int i = 0; i < AMAX + 2; ++i) {
                const uint64_t g = spin == 0 ? requested[i] : frz::granule_load(last + (i < nch_total ? i : 0));
                all = all && (uint32_t)(g >> 32) == ended;
                not_terminated = i == A ? (uint32_t)g : not_terminated;
                not_truncated = i == A + 1 ? (uint32_t)g : not_truncated;
            }
            if (all) break;
            if (gave_up || spin >= (1 << 20)) {
                if (!gave_up && tid == 0) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), (uint32_t)FRZ_ERR_SCAN_TIMEOUT);
                gave_up = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        finished = not_terminated == 0u || not_truncated == 0u;
    };

    if (view) {
        // ================================================================================================ view role
        // The view role's stores of one step as a function of where they go: `delta` = byte distance from the arena's observation block
        // (self rows, others rows, task rows: contiguous, frz_cybersecurity_obs_block) to the copy being written — 0: the env's own
        // buffers; a step's copy of an observation tape (frz_rollout_spec.obs_tape) otherwise.  With a tape every step but the launch's
        // last writes ONLY its tape copy (the env's own rows are rewritten by the next step anyway), the last one both; a launch that
        // finds the batch finished early writes the last executed step's rows once more, into the env's own buffers (below).
        uint32_t pres_bits_v = 0;
        auto emit_obs = [&](int64_t delta) {
            float* const self_att = reinterpret_cast<float*>(arena + d.off_self_att + delta);
            float* const self_def = reinterpret_cast<float*>(arena + d.off_self_def + delta);
            float* const others_att = reinterpret_cast<float*>(arena + d.off_others_att + delta);
            float* const others_def = reinterpret_cast<float*>(arena + d.off_others_def + delta);
            int64_t* const tasks = reinterpret_cast<int64_t*>(arena + d.off_tasks + delta);
            const bool op = (flags & kObsPower) != 0, opr = (flags & kObsPresence) != 0, ol = (flags & kObsLocation) != 0;
            const int ka = (op ? 1 : 0) + (opr ? 1 : 0), kd = ka + (ol ? 1 : 0);
            bool pres[AMAX];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) pres[a] = (pres_bits_v >> a) & 1u;
            // Every row kind below is, for this wavefront's 64 envs, ONE run of consecutive bytes of its output array (row (agent, env) at
            // (agent * B + env) * K).  The task rows (192 of the step's 420 bytes per env) go through LDS into whole lines (below); doing the
            // same to the small self / others rows costs more in LDS round trips than their stores do (FRZ_CY_ROWS_LDS, off).
#if FRZ_CY_TASKS_LDS
            const int vw = (int)(threadIdx.x >> 6) & 3, ln = (int)(threadIdx.x & 63);
            const int64_t b_first = (int64_t)chunk * kBlock + vw * 64;
            const int live = (int)(B - b_first < 64 ? (B - b_first < 0 ? 0 : B - b_first) : 64);
#endif
#if FRZ_CY_TASKS_LDS && FRZ_CY_ROWS_LDS  // (the small self / others rows the same way: measured SLOWER, 6.59 against 6.11 us per step — off)
            float* const stage_f = reinterpret_cast<float*>(&s_task_stage[vw][0]);
            constexpr int KMAX = 3 * (AMAX > 1 ? AMAX - 1 : 1);
            auto staged_rows = [&](float* out, const float (&vals)[KMAX], int K) {  // out: the row of the wavefront's FIRST env
                if (K <= 0) return;
#pragma unroll
                for (int j = 0; j < KMAX; ++j)
                    if (j < K) stage_f[ln * K + j] = vals[j];
                frz::wave_lds_sync();
                if (live == 64 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
                    for (int piece = ln; piece < 16 * K; piece += 64) reinterpret_cast<float4*>(out)[piece] = reinterpret_cast<const float4*>(stage_f)[piece];
                } else {
                    for (int piece = ln; piece < live * K; piece += 64) out[piece] = stage_f[piece];
                }
                frz::wave_lds_sync();
            };
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (a < Att) {  // attackers: (threat, presence) (:481-484)
                    if (active) reinterpret_cast<float2*>(self_att)[a * B + b] = make_float2(d.threat[a], pres[a] ? 1.0f : 0.0f);
                    float vals[KMAX];
                    int col = 0;
#pragma unroll
                    for (int o = 0; o < AMAX; ++o)
                        if (o < Att && o != a) {
                            if (op) vals[col++] = d.threat[o];
                            if (opr) vals[col++] = pres[o] ? 1.0f : 0.0f;
                        }
                    staged_rows(others_att + ((int64_t)a * B + b_first) * (int64_t)((Att - 1) * ka), vals, (Att - 1) * ka);
                }
            }
#pragma unroll
            for (int k = 0; k < AMAX; ++k) {
                if (k < D) {  // defenders: (mitigation, presence, location) (:475-479)
                    bool present_k = false;
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) present_k = (a == Att + k) ? pres[a] : present_k;
                    float own[KMAX];
                    own[0] = d.mitigation[k];
                    if (KMAX > 1) own[KMAX > 1 ? 1 : 0] = present_k ? 1.0f : 0.0f;
                    if (KMAX > 2) own[KMAX > 2 ? 2 : 0] = (float)loc[k];
                    if (KMAX >= 3) {
                        staged_rows(self_def + ((int64_t)k * B + b_first) * 3, own, 3);
                    } else if (active) {
                        float* self = self_def + (k * B + b) * 3;
                        self[0] = d.mitigation[k], self[1] = present_k ? 1.0f : 0.0f, self[2] = (float)loc[k];
                    }
                    float vals[KMAX];
                    int col = 0;
#pragma unroll
                    for (int o = 0; o < AMAX; ++o)
                        if (o < D && o != k) {
                            bool present_o = false;
#pragma unroll
                            for (int a = 0; a < AMAX; ++a) present_o = (a == Att + o) ? pres[a] : present_o;
                            if (op) vals[col++] = d.mitigation[o];
                            if (opr) vals[col++] = present_o ? 1.0f : 0.0f;
                            if (ol) vals[col++] = (float)loc[o];
                        }
                    staged_rows(others_def + ((int64_t)k * B + b_first) * (int64_t)((D - 1) * kd), vals, (D - 1) * kd);
                }
            }
            if (false) {
#else
            if (active) {
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    if (a < Att) {  // attackers: (threat, presence) (:481-484)
                        reinterpret_cast<float2*>(self_att)[a * B + b] = make_float2(d.threat[a], pres[a] ? 1.0f : 0.0f);
                        float* others = others_att + (a * B + b) * (int64_t)((Att - 1) * ka);
                        int col = 0;
#pragma unroll
                        for (int o = 0; o < AMAX; ++o)
                            if (o < Att && o != a) {
                                if (op) others[col++] = d.threat[o];
                                if (opr) others[col++] = pres[o] ? 1.0f : 0.0f;
                            }
                    }
                }
#pragma unroll
                for (int k = 0; k < AMAX; ++k) {
                    if (k < D) {  // defenders: (mitigation, presence, location) (:475-479)
                        bool present_k = false;
#pragma unroll
                        for (int a = 0; a < AMAX; ++a) present_k = (a == Att + k) ? pres[a] : present_k;
                        float* self = self_def + (k * B + b) * 3;
                        self[0] = d.mitigation[k];
                        self[1] = present_k ? 1.0f : 0.0f;
                        self[2] = (float)loc[k];
                        float* others = others_def + (k * B + b) * (int64_t)((D - 1) * kd);
                        int col = 0;
#pragma unroll
                        for (int o = 0; o < AMAX; ++o)
                            if (o < D && o != k) {
                                bool present_o = false;
#pragma unroll
                                for (int a = 0; a < AMAX; ++a) present_o = (a == Att + o) ? pres[a] : present_o;
                                if (op) others[col++] = d.mitigation[o];
                                if (opr) others[col++] = present_o ? 1.0f : 0.0f;
                                if (ol) others[col++] = (float)loc[o];
                            }
                    }
                }
#endif
#if !FRZ_CY_TASKS_LDS
                // tasks (state, criticality) per agent; a defender sees them only right after monitoring (:497, :510-511)
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    if (a < A) {
                        bool hidden = false;
                        if (flags & kPartial) {
#pragma unroll
                            for (int k = 0; k < AMAX; ++k) hidden = (a == Att + k && k < D) ? last[k] != -3 : hidden;
                        }
                        int64_t* t = tasks + (a * B + b) * (int64_t)(N * 2);
#pragma unroll
                        for (int n = 0; n < NMAX; ++n)
                            if (n < N)
                                reinterpret_cast<longlong2*>(t)[n] =
                                    hidden ? make_longlong2(-100, -100) : make_longlong2(state[n], d.criticality[n]);
                    }
                }
#endif
            }
#if FRZ_CY_TASKS_LDS
            {   // tasks (state, criticality) per agent; a defender sees them only right after monitoring (:497, :510-511).  An agent's rows of
                // this wavefront's 64 envs are 64 * N * 16 CONSECUTIVE bytes of the output: staged in LDS in that order and written as
                // lane-consecutive 16-byte pieces (whole lines per store instruction) instead of N pieces per lane at a stride of N * 16
                // bytes — round 4: 7.3 -> 6.3 us per step of the episode launch (the lever was the store pattern, not the bytes)
                longlong2* const stage = &s_task_stage[vw][0];
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    if (a < A) {
                        bool hidden = false;
                        if (flags & kPartial) {
#pragma unroll
                            for (int k = 0; k < AMAX; ++k) hidden = (a == Att + k && k < D) ? last[k] != -3 : hidden;
                        }
#pragma unroll
                        for (int n = 0; n < NMAX; ++n)
                            if (n < N) stage[ln * N + n] = hidden ? make_longlong2(-100, -100) : make_longlong2(state[n], d.criticality[n]);
                        frz::wave_lds_sync();
                        longlong2* const out = reinterpret_cast<longlong2*>(tasks) + ((int64_t)a * B + b_first) * N;
                        for (int piece = ln; piece < live * N; piece += 64) out[piece] = stage[piece];
                        frz::wave_lds_sync();
                    }
                }
            }
#endif
        };
        const int64_t obs_tape_delta = EXTRA ? L.obs_tape_delta : (int64_t)0;
        for (int t = 0; t < n_steps; ++t) {
            // (no per-step opaque copies here, unlike wildfire_roles.hip: with them this kernel needs 160 instead of 219 VGPRs and is slower,
            // 7.7 against 7.4 us per step — its row addresses are better computed once, above the step loop)
            const int nm_step = nm_in + t;
            // ---------------------------------------------------------------- the step's draws (streams: cy_step_kernel above)
            if (RNG == FRZ_RNG_INJECTED) {
                if constexpr (EXTRA) {
                    if (t > 0) {  // the randomness tapes' next pair
#pragma unroll
                        for (int n = 0; n < NMAX; ++n) r_in[n] = net_rand[((int64_t)t * B + bl) * N + min(n, N - 1)];
#pragma unroll
                        for (int a = 0; a < AMAX; ++a) r_in[NMAX + a] = agent_rand[((int64_t)t * B + bl) * A + min(a, A - 1)];
                    }
                }
#pragma unroll
                for (int k = 0; k < NMAX + AMAX; ++k) s_draw[k][tid] = r_in[k];
            } else if constexpr (RNG == FRZ_RNG_MT19937) {
                constexpr int U = NMAX + AMAX, kN = 624, kM = 397;
                static_assert(U <= kN - kM, "a batch must not read a word it rewrites");
                uint32_t* const mt = reinterpret_cast<uint32_t*>(arena + L.off_mt_state);
                const int used = N + A;
                uint32_t w[U + 1], far[U];
#pragma unroll
                for (int k = 0; k <= U; ++k) {
                    int j = mti + k;
                    j -= j >= kN ? kN : 0;
                    w[k] = mt[(int64_t)j * B + bl];
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    int j = mti + k + kM;
                    j -= j >= kN ? kN : 0;
                    j -= j >= kN ? kN : 0;
                    far[k] = mt[(int64_t)j * B + bl];
                }
                float uni[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const uint32_t y = (w[k] & 0x80000000u) | (w[k + 1] & 0x7fffffffu);
                    uint32_t v = far[k] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                    int j = mti + k;
                    j -= j >= kN ? kN : 0;
                    if (active && k < used) mt[(int64_t)j * B + bl] = v;
                    v ^= v >> 11;
                    v ^= (v << 7) & 0x9d2c5680u;
                    v ^= (v << 15) & 0xefc60000u;
                    v ^= v >> 18;
                    uni[k] = (float)(v & 0xFFFFFFu) * (1.0f / 16777216.0f);
                }
                if (active) {
                    int j = mti + used;
                    j -= j >= kN ? kN : 0;
                    at32(rows, (uint32_t)(r_seeds + 1) * Bu + bl) = j;
                }
#pragma unroll
                for (int n = 0; n < NMAX; ++n) s_draw[n][tid] = uni[n];  // node n is draw n, agent a is draw N + a
#pragma unroll
                for (int a = 0; a < AMAX; ++a) {
                    float u = 0.0f;
#pragma unroll
                    for (int k = 0; k < U; ++k) u = (k == N + a) ? uni[k] : u;
                    s_draw[NMAX + a][tid] = u;
                }
            } else {
#pragma unroll
                for (int q = 0; q < (NMAX + 3) / 4; ++q) {
                    frz::Philox4 w{{0u, 0u, 0u, 0u}};
                    const bool drawn = q * 4 < N && (flags & kStochState);
                    if (drawn) w = frz::philox4x32_10((uint32_t)q, (uint32_t)nm_step, 0u, 0u, seed, 0x46525A01u);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (q * 4 + j < NMAX) s_draw[q * 4 + j][tid] = drawn ? frz::u32_to_unit_float(w.w[j]) : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < (AMAX + 3) / 4; ++q) {
                    frz::Philox4 w{{0u, 0u, 0u, 0u}};
                    const bool drawn = q * 4 < A;
                    if (drawn) w = frz::philox4x32_10((uint32_t)q, (uint32_t)nm_step, 1u, 0u, seed, 0x46525A01u);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (q * 4 + j < AMAX) s_draw[NMAX + q * 4 + j][tid] = drawn ? frz::u32_to_unit_float(w.w[j]) : 0.0f;
                }
            }
            __syncthreads();  // (2) draws ready
            if constexpr (PERSIST) {
                if (s_stop) {  // the state role's verdict (it has looked at the totals of the step that just ended): nothing more happens
                    if (obs_tape_delta != 0 && t > 0) emit_obs(0);  // (the last executed step's rows went to its tape copy only)
                    break;
                }
            }
            __syncthreads();  // (3) post-transition state ready
#pragma unroll
            for (int n = 0; n < NMAX; ++n) state[n] = s_post[n][tid];
#pragma unroll
            for (int k = 0; k < AMAX; ++k) {
                loc[k] = s_post[NMAX + k][tid];
                last[k] = s_post[NMAX + AMAX + k][tid];
            }
            pres_bits_v = (uint32_t)s_post[NMAX + 2 * AMAX][tid];
            frz::scan_chunk_passive_front();
            if (obs_tape_delta != 0) emit_obs(obs_tape_delta + (int64_t)t * L.obs_tape_step);
            if (obs_tape_delta == 0 || t == n_steps - 1) emit_obs(0);
            frz::scan_chunk_passive_back();
        }  // steps of this launch
        return;
    }

    // ==================================================================================================== state role
    bool pres[AMAX];
#pragma unroll
    for (int n = 0; n < NMAX; ++n) state[n] = n < N ? state[n] : 0;
#pragma unroll
    for (int k = 0; k < AMAX; ++k) {
        loc[k] = k < D ? loc[k] : -1;
        last[k] = k < D ? last[k] : -2;
    }
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
        pres[a] = a < A && pres_raw[a] != 0;
        if (L.policy || a >= A) act_in[a] = make_int2(0, -1);
    }
    bool trunc = trunc_raw != 0;
    uint32_t err = 0;
    int nm = nm_in, executed = 0;
    uint32_t excl[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) excl[a] = 0;
    // action mapping: arange(N) while present, empty otherwise (:441-457); `copy`: byte distance to the copy of the packed values to write
    // The packed VALUES are position mod N wherever a segment lies (every segment is arange(N) and starts at a multiple of N): the env's own
    // buffers hold that pattern since bind (cy_prefill_kernel) and a step only moves the offsets; only a list record — a caller's buffer —
    // gets the values written (`values`).
    auto emit_mappings = [&](int64_t copy, int64_t ocopy, bool values) {
        if (active) {
            int32_t* const act_values = reinterpret_cast<int32_t*>(arena + d.off_act_values + copy);
            int64_t* const act_offsets = reinterpret_cast<int64_t*>(arena + d.off_act_offsets + ocopy);
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (a < A) {
                    const int64_t off = (int64_t)excl[a] * N;
                    act_offsets[a * (B + 1) + b] = off;
                    if (b == B - 1) act_offsets[a * (B + 1) + B] = off + (pres[a] ? N : 0);
                    if (pres[a] && values) {
                        int32_t* v = act_values + a * B * N + off;
#pragma unroll
                        for (int n = 0; n < NMAX; ++n)
                            if (n < N) v[n] = n;
                    }
                }
            }
        }
    };
    for (int t = 0; t < n_steps; ++t) {
        // the caller's buffers at the last step only; before it the second copy, or — with a list record — that step's copy of the whole
        // list block, offsets included
        const bool recorded = EXTRA && L.list_record_delta != 0 && t < n_steps - 1;
        const int64_t copy = recorded ? L.list_record_delta + (int64_t)t * L.list_record_step : (int64_t)0;
        const int64_t ocopy = recorded ? copy : (int64_t)0;
        if constexpr (PERSIST) {
            if (t > 0) {
                request_totals();
#pragma unroll
                for (int a = 0; a < AMAX; ++a) act_in[a] = make_int2(0, -1);
                if constexpr (EXTRA) {
                    if (!L.policy) {  // the action tape's next step
                        const int2* const tape = reinterpret_cast<const int2*>(actions + (int64_t)t * L.tape_actions_step);
#pragma unroll
                        for (int a = 0; a < AMAX; ++a)
                            if (a < A) act_in[a] = tape[(int64_t)a * B + bl];
                    }
                }
            }
        }

        if (L.policy) {  // the stream of cy_policy_kernel (see cy_step_kernel)
            frz::Philox4 policy_words[(AMAX + 3) / 4];
#pragma unroll
            for (int q = 0; q < (AMAX + 3) / 4; ++q)
                if (q * 4 < A) {
                    const uint64_t policy_step = (((uint64_t)L.policy_step_hi << 32) | L.policy_step_lo) + (uint64_t)(PERSIST ? t : 0);
                    policy_words[q] = frz::philox4x32_10((uint32_t)q, 0u, (uint32_t)policy_step, (uint32_t)(policy_step >> 32), L.policy_seed_lo ^ seed,
                                                         L.policy_seed_hi);
                }
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                if (a < A) {
                    const int n = (flags & kShowBad) ? N : (pres[a] ? N : 0);
                    int tail1 = -3, nt = 1;
                    if (a >= Att && n > 0) {
                        int home_loc = 0;
#pragma unroll
                        for (int k = 0; k < AMAX; ++k) home_loc = (a == Att + k) ? loc[k] : home_loc;
                        const bool patchable = (flags & kShowBad) || home_loc != -1;
                        tail1 = patchable ? -2 : -3;
                        nt = patchable ? 3 : 2;
                    }
                    const int j = (int)(((uint64_t)policy_words[a >> 2].w[a & 3] * (uint64_t)(n + nt)) >> 32);
                    const int value = j < n ? 0 : (j - n == 0 ? -1 : (j - n == 1 ? tail1 : -3));
                    act_in[a] = make_int2(j, value);
                    if constexpr (!PERSIST) {
                        if (active) reinterpret_cast<int2*>(L.actions_out)[(int64_t)a * B + b] = act_in[a];
                    }
                }
            }
        }
        // --------------------------------------------------- action decode (cybersecurity.py:326-384), agent order
        uint32_t attack_set[NMAX], patch_set[NMAX];
#pragma unroll
        for (int n = 0; n < NMAX; ++n) attack_set[n] = patch_set[n] = 0u;
        float rew[AMAX];
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            rew[a] = 0.0f;
            if (a < A) {
                const int2 v = act_in[a];
                const int idx = v.x, act = v.y;
                const bool bad_target = act == 0 && (idx < 0 || idx >= N);
                if (bad_target && active) err |= FRZ_ERR_INVALID_TARGET;
                if (!bad_target && !(flags & kShowBad) && !pres[a] && act != -1 && active) err |= FRZ_ERR_ABSENT_ACTION;
                if (a < Att) {
                    const bool attack = act == 0 && !bad_target;
#pragma unroll
                    for (int n = 0; n < NMAX; ++n) attack_set[n] |= (attack && idx == n) ? (1u << a) : 0u;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < AMAX; ++k) {
            if (k < D) {
                int2 v = make_int2(0, -1);
#pragma unroll
                for (int a = 0; a < AMAX; ++a) v = (a == Att + k) ? act_in[a] : v;
                const int idx = v.x, act = v.y;
                const bool bad_target = act == 0 && (idx < 0 || idx >= N);
                const bool move = act == 0 && !bad_target;
                const bool patch = act == -2 && loc[k] != -1 && !bad_target;
#pragma unroll
                for (int n = 0; n < NMAX; ++n) patch_set[n] |= (patch && loc[k] == n) ? (1u << k) : 0u;
                const float pr = patch ? d.patch_reward : 0.0f;
#pragma unroll
                for (int a = 0; a < AMAX; ++a) rew[a] = (a == Att + k) ? rew[a] + pr : rew[a];
                loc[k] = move ? idx : loc[k];
                last[k] = bad_target ? last[k] : act;
            }
        }
        if constexpr (PERSIST) {
            // nothing of this step has left the registers yet: now the totals of the step before it (requested above) must be here.  One
            // verdict per workgroup (the first thread's), read by both roles behind the barrier
            if (t > 0) await_totals();
            if (threadIdx.x == 0) s_stop = (t > 0 && finished) ? 1 : 0;
        }
        __syncthreads();  // (2) draws ready
        if constexpr (PERSIST) {
            if (s_stop) {  // utils/env.py:211-213: nothing more happens in this launch
                if (active && !at32(rows1, (uint32_t)d.u_frozen * Bu + bl)) {  // stale rewards, once (utils/conversions.py:87-90)
                    for (int a = 0; a < A; ++a) {
                        const float r = at32(rowsf, (uint32_t)(d.r_rewards + a) * Bu + bl);
                        float acc = 0.0f;
                        for (int j = 0; j < A; ++j) acc = acc + r;
                        at32(rowsf, (uint32_t)(d.r_rewards + a) * Bu + bl) = acc;
                    }
                    at32(rows1, (uint32_t)d.u_frozen * Bu + bl) = 1;
                }
                emit_mappings(0, 0, false);  // (offsets only: the values never change, see emit_mappings)
                break;
            }
            if (L.policy && active) {
                int2* const out = reinterpret_cast<int2*>(L.actions_out + (EXTRA ? (int64_t)t * L.actions_out_step : (int64_t)0));
#pragma unroll
                for (int a = 0; a < AMAX; ++a)
                    if (a < A) out[(int64_t)a * B + b] = act_in[a];
            }
        }
        // ------------------------------------------------- presence (transitions/presence.py:46-58), same draw for both tests
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            if (a < A) {
                const float u = s_draw[NMAX + a][tid];
                const bool ret = !pres[a] && u < d.back[a];
                const bool leave = pres[a] && u >= d.persist[a];
                pres[a] = ret ? true : (leave ? false : pres[a]);
#pragma unroll
                for (int k = 0; k < AMAX; ++k) loc[k] = (ret && a == Att + k) ? -1 : loc[k];
            }
        }
        // ------------------------------------------------- subnetwork transition (transitions/subnetwork.py:53-70)
        float net_reward = 0.0f;
#pragma unroll
        for (int n = 0; n < NMAX; ++n) {
            if (n < N) {
                const uint32_t index = (patch_set[n] << Att) | attack_set[n];
                const float danger = s_lut[index];
                bool better = danger > 0.0f, worse = danger < 0.0f;
                if (flags & kStochState) {
                    const bool gate = fabsf(danger) <= s_draw[n][tid];
                    better = better && gate;
                    worse = worse && gate;
                }
                int s = state[n] - (better ? 1 : 0) + (worse ? 1 : 0);
                s = s < 0 ? 0 : (s > d.S - 1 ? d.S - 1 : s);
                state[n] = s;
                net_reward = __fadd_rn(net_reward, __fmul_rn(cfg_lds->state_rewards[s], (float)d.criticality[n]));
            }
        }
        uint32_t pres_bits = 0;
#pragma unroll
        for (int a = 0; a < AMAX; ++a) pres_bits |= pres[a] ? (1u << a) : 0u;
#pragma unroll
        for (int n = 0; n < NMAX; ++n) s_post[n][tid] = state[n];
#pragma unroll
        for (int k = 0; k < AMAX; ++k) {
            s_post[NMAX + k][tid] = loc[k];
            s_post[NMAX + AMAX + k][tid] = last[k];
        }
        s_post[NMAX + 2 * AMAX][tid] = (int)pres_bits;
        __syncthreads();  // (3) post-transition state ready
        nm += 1;
        trunc = (flags & kTruncate) ? nm >= d.max_steps : trunc;

        uint32_t cnt[AMAX];
#pragma unroll
        for (int a = 0; a < AMAX; ++a) cnt[a] = (active && a < A && pres[a]) ? 1u : 0u;
        if (active) {
#pragma unroll
            for (int n = 0; n < NMAX; ++n)
                if (n < N) at32(rows, (uint32_t)(d.r_state + n) * Bu + bl) = state[n];
#pragma unroll
            for (int k = 0; k < AMAX; ++k)
                if (k < D) {
                    at32(rows, (uint32_t)(d.r_loc + k) * Bu + bl) = loc[k];
                    at32(rows, (uint32_t)(d.r_last + k) * Bu + bl) = last[k];
                }
#pragma unroll
            for (int a = 0; a < AMAX; ++a)
                if (a < A) {
                    const float r = a < Att ? __fadd_rn(rew[a], __fmul_rn(net_reward, -1.0f)) : __fadd_rn(rew[a], net_reward);
                    at32(rows1, (uint32_t)(d.u_presence + a) * Bu + bl) = (uint8_t)pres[a];
                    at32(rowsf, (uint32_t)(d.r_rewards + a) * Bu + bl) = r;
                    if constexpr (EXTRA) {
                        if (L.reward_tape != nullptr) L.reward_tape[((int64_t)t * A + a) * B + bl] = r;
                    }
                    if (flags & kTruncate) at32(rows1, (uint32_t)(d.u_trunc + a) * Bu + bl) = (uint8_t)trunc;
                    if (flags & kTrackCumulative) {
                        const float total = __fadd_rn(cum_in[a], r);
                        at32(rowsf, (uint32_t)(d.r_cum + a) * Bu + bl) = total;
                        if constexpr (PERSIST) cum_in[a] = total;
                    }
                    at32(rows, (uint32_t)(d.r_atc + a) * Bu + bl) = pres[a] ? N : 0;
                }
            at32(rows, (uint32_t)d.r_moves * Bu + bl) = nm;
            if constexpr (EXTRA) {
                if (L.state_tape != nullptr) {  // frz_rollout_spec.state_tape: this step's state rows, then the presence bytes
                    char* const step_copy = L.state_tape + (int64_t)t * L.state_tape_step;
                    int32_t* const st = reinterpret_cast<int32_t*>(step_copy);
                    uint8_t* const pr = reinterpret_cast<uint8_t*>(step_copy + (int64_t)(N + 2 * D) * B * 4);
#pragma unroll
                    for (int n = 0; n < NMAX; ++n)
                        if (n < N) st[(int64_t)n * B + bl] = state[n];
#pragma unroll
                    for (int k = 0; k < AMAX; ++k)
                        if (k < D) st[(int64_t)(N + k) * B + bl] = loc[k], st[(int64_t)(N + D + k) * B + bl] = last[k];
#pragma unroll
                    for (int a = 0; a < AMAX; ++a)
                        if (a < A) pr[(int64_t)a * B + bl] = (uint8_t)pres[a];
                }
                if (L.done_tape != nullptr) {  // terminations never set (cybersecurity.py:298)
                    L.done_tape[((int64_t)t * 2 + 0) * B + bl] = (uint8_t)0;
                    L.done_tape[((int64_t)t * 2 + 1) * B + bl] = (uint8_t)trunc;
                }
            }
        }
        {
            const frz::ScanLaunch step{epoch_now, epoch_now + 1u, nullptr, ws.totals + (epoch_now & 1u) * frz::kTotalsStride};
            frz::scan_chunk<AMAX>(s_scan, ws, step, cnt, active, active && !trunc, A, chunk, nchunks, excl, &err);
        }
        emit_mappings(copy, ocopy, recorded);
        executed = t + 1;
    }  // steps of this launch
    if (err) atomicOr(reinterpret_cast<uint32_t*>(arena + d.off_error), err);
    // The workgroup that owns the last chunk finished its last look-back only after every other chunk published, i.e. after every
    // workgroup of the launch read the epoch: it advances it, by the steps the launch executed.
    if (executed > 0 && chunk == nchunks - 1 && threadIdx.x == 0) *ws.epoch = launch.epoch + (uint32_t)executed;
}


// the packed action-mapping values: position mod N (see emit_mappings), written once per bind over the whole capacity
__global__ void __launch_bounds__(kBlock) cy_prefill_kernel(int32_t* values, int64_t count, int32_t N) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < count) values[i] = (int32_t)(i % N);
}

// uniform member of each agent's OneOf action space (spaces/actions.py:11-99), see oracle/frz_oracle_cybersecurity.c
__global__ void __launch_bounds__(kBlock) cy_policy_kernel(const char* arena, uint32_t seed_lo, uint32_t seed_hi, uint32_t step_lo,
                                                             uint32_t step_hi, int32_t* actions) {
    const CyDev& d = *reinterpret_cast<const CyDev*>(arena);
    const int64_t B = d.B;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)d.A * B) return;
    const int a = (int)(i / B);
    const int64_t b = i % B;
    const int32_t* rows = reinterpret_cast<const int32_t*>(arena + d.off_rows4);
    const int n = (d.flags & kShowBad) ? d.N : rows[d.r_atc * B + i];
    int tail[3], nt = 0;
    tail[nt++] = -1;
    if (a >= d.Att && n > 0) {
        const bool home = rows[(d.r_loc + (a - d.Att)) * B + b] == -1;
        if ((d.flags & kShowBad) || !home) tail[nt++] = -2;
        tail[nt++] = -3;
    }
    const uint32_t env_seed = (uint32_t)rows[d.r_seeds * B + b];
    const uint32_t agent = (uint32_t)(i / B);  // agent a draws word a % 4 of block a / 4 of the (env, step) stream
    const frz::Philox4 w = frz::philox4x32_10(agent >> 2, 0u, step_lo, step_hi, seed_lo ^ env_seed, seed_hi);
    const uint32_t lane_word = agent & 3u;  // selected with compares: a runtime index into the block would put it in scratch
    const uint32_t word = lane_word == 0u ? w.w[0] : (lane_word == 1u ? w.w[1] : (lane_word == 2u ? w.w[2] : w.w[3]));
    const int j = (int)(((uint64_t)word * (uint64_t)(n + nt)) >> 32);
    const int value = j < n ? 0 : (j - n == 0 ? tail[0] : (j - n == 1 ? tail[1] : tail[2]));
    reinterpret_cast<int2*>(actions)[i] = make_int2(j, value);
}

}  // namespace

// ================================================================================================================
// host side of the C-ABI
// ================================================================================================================
struct frz_cybersecurity_env {
    frz_cybersecurity_cfg cfg;
    CyDev dev;
    std::vector<float> lut;
    char* arena = nullptr;
    bool was_reset = false;
    bool ticketed = false;  // more chunks than CUs: chunks are handed out in arrival order (frz_scan.h)
    int variant = 0;
    bool roles = false;  // steps run cy_roles_kernel (shapes up to 8 nodes / 8 agents, unless FRZ_CY_KERNEL=lane)
    // multi-step launches (frz_cybersecurity_rollout_random_policy): allowed by frz_cybersecurity_set_exclusive_device; byte distance from
    // the packed action-mapping values to their second copy; steps of the launch being enqueued
    bool exclusive_device = false;
    int64_t copy_delta = 0;
    int32_t rollout_steps = 1;
    frz_cybersecurity_saved_state saved = {};  // frz_cybersecurity_set_saved_initial: what a partial reset restores (network_state == nullptr: the configured state)
    struct RolloutOptions {  // frz_cybersecurity_rollout: the options of the multi-step launch being enqueued
        bool extra = false;
        uint32_t flags = 0;
        int64_t tape_actions_step = 0, list_record_delta = 0, list_record_step = 0, actions_out_step = 0;
        float* reward_tape = nullptr;
        uint8_t* done_tape = nullptr;
        int64_t obs_tape_delta = 0, obs_tape_step = 0, state_tape_step = 0;
        char* state_tape = nullptr;
    } rollout;
};

namespace {

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// variants whose step kernel advances the per-env MT19937 streams itself (the 16-node / 16-agent one stages its draws)
template <int NMAX, int AMAX>
constexpr bool kMtInKernel = NMAX + AMAX <= 16;

struct Policy {  // fused uniform random policy of a step launch
    bool on = false;
    uint64_t seed = 0, step = 0;
    int32_t* actions_out = nullptr;
};

template <int NMAX, int AMAX, int ATT = -1>
void launch_variant(frz_cybersecurity_env* env, const int32_t* actions, const float* nr, const float* ar, int rng, int mode,
                    hipStream_t stream, const Policy& policy) {
    const CyDev* dev = reinterpret_cast<const CyDev*>(env->arena);
    const dim3 grid(env->dev.nchunks), block(kBlock);
    const CyDev& p = env->dev;
    const CyLaunch L{p.B, p.N, p.Att, p.D, p.A, env->ticketed ? 1u : 0u, p.off_rows1, p.off_epoch, p.off_totals, policy.on ? 1u : 0u,
                     (uint32_t)policy.seed, (uint32_t)(policy.seed >> 32), (uint32_t)policy.step, (uint32_t)(policy.step >> 32), policy.actions_out,
                     p.off_mt_state, p.off_lut, p.lut_entries, env->rollout_steps, env->copy_delta,
                     env->rollout.flags, 0u, env->rollout.tape_actions_step, env->rollout.list_record_delta, env->rollout.list_record_step,
                     env->rollout.reward_tape, env->rollout.done_tape, env->rollout.actions_out_step,
                     env->rollout.obs_tape_delta, env->rollout.obs_tape_step, env->rollout.state_tape, env->rollout.state_tape_step};
    if constexpr (AMAX <= 8) {  // (the state / view kernel stages the danger table of 2^A entries in LDS: up to 8 agents; up to 16 nodes since round 4)
        if (mode == kStep && env->roles) {  // state / view roles: two wavefronts per 64 envs (cy_roles_kernel)
            const dim3 wide(kRoleBlock);
            // (the multi-step instantiations exist up to 8 nodes: at <16, 8> they need 256 VGPRs plus ~1 KB of scratch per lane and run SLOWER
            // than one launch per step — 42 against 31 us per step at 12 nodes, 89 against 47 at 16: round 4, tools/dbg/cy_nodes_probe.py)
            bool launched = false;
            if constexpr (NMAX <= 8) {
                launched = true;
                if (rng == FRZ_RNG_INJECTED && env->rollout_steps > 1)
                    hipLaunchKernelGGL((cy_roles_kernel<NMAX, AMAX, ATT, FRZ_RNG_INJECTED, true, true>), grid, wide, 0, stream, env->arena, dev, actions, nr, ar, L);
                else if (rng == FRZ_RNG_PHILOX && env->rollout_steps > 1 && env->rollout.extra)
                    hipLaunchKernelGGL((cy_roles_kernel<NMAX, AMAX, ATT, FRZ_RNG_PHILOX, true, true>), grid, wide, 0, stream, env->arena, dev, actions, nr, ar, L);
                else if (rng == FRZ_RNG_PHILOX && env->rollout_steps > 1)
                    hipLaunchKernelGGL((cy_roles_kernel<NMAX, AMAX, ATT, FRZ_RNG_PHILOX, true>), grid, wide, 0, stream, env->arena, dev, actions, nr, ar, L);
                else
                    launched = false;
            }
            if (launched) return;
            if (rng == FRZ_RNG_PHILOX)
                hipLaunchKernelGGL((cy_roles_kernel<NMAX, AMAX, ATT, FRZ_RNG_PHILOX>), grid, wide, 0, stream, env->arena, dev, actions, nr, ar, L);
            else if (rng == FRZ_RNG_MT19937) {
                if constexpr (kMtInKernel<NMAX, AMAX>)  // (otherwise the draws were staged and rng arrives as FRZ_RNG_INJECTED)
                    hipLaunchKernelGGL((cy_roles_kernel<NMAX, AMAX, ATT, FRZ_RNG_MT19937>), grid, wide, 0, stream, env->arena, dev, actions, nr, ar, L);
            } else
                hipLaunchKernelGGL((cy_roles_kernel<NMAX, AMAX, ATT, FRZ_RNG_INJECTED>), grid, wide, 0, stream, env->arena, dev, actions, nr, ar, L);
            return;
        }
    }
    if (mode == kRebuild)
        hipLaunchKernelGGL((cy_step_kernel<NMAX, AMAX, ATT, FRZ_RNG_INJECTED, kRebuild>), grid, block, 0, stream, env->arena, dev, actions, nr, ar, L);
    else if (rng == FRZ_RNG_PHILOX)
        hipLaunchKernelGGL((cy_step_kernel<NMAX, AMAX, ATT, FRZ_RNG_PHILOX, kStep>), grid, block, 0, stream, env->arena, dev, actions, nr, ar, L);
    else if (rng == FRZ_RNG_MT19937) {
        if constexpr (kMtInKernel<NMAX, AMAX>)
            hipLaunchKernelGGL((cy_step_kernel<NMAX, AMAX, ATT, FRZ_RNG_MT19937, kStep>), grid, block, 0, stream, env->arena, dev, actions, nr, ar, L);
    }
    else
        hipLaunchKernelGGL((cy_step_kernel<NMAX, AMAX, ATT, FRZ_RNG_INJECTED, kStep>), grid, block, 0, stream, env->arena, dev, actions, nr, ar, L);
}

int launch(frz_cybersecurity_env* env, const int32_t* actions, const float* nr, const float* ar, int rng, int mode, hipStream_t stream,
           const Policy& policy = Policy()) {
    switch (env->variant) {
        case 0: launch_variant<4, 4>(env, actions, nr, ar, rng, mode, stream, policy); break;
        case 1: launch_variant<8, 8>(env, actions, nr, ar, rng, mode, stream, policy); break;
        case 3: launch_variant<3, 4, 2>(env, actions, nr, ar, rng, mode, stream, policy); break;  // exact: 3 nodes, 2 attackers, 2 defenders
        case 4: launch_variant<16, 8>(env, actions, nr, ar, rng, mode, stream, policy); break;  // 9-16 nodes, up to 8 agents: state / view kernel too (round 4)
        default: launch_variant<16, 16>(env, actions, nr, ar, rng, mode, stream, policy); break;
    }
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

template <typename T>
T* at(char* arena, int64_t off) {
    return reinterpret_cast<T*>(arena + off);
}

}  // namespace

extern "C" {

int frz_cybersecurity_create(const frz_cybersecurity_cfg* cfg, frz_cybersecurity_env** out) {
    if (!cfg || !out) return FRZ_E_INVALID;
    const int N = cfg->num_nodes, Att = cfg->num_attackers, D = cfg->num_defenders, A = Att + D;
    if (cfg->parallel_envs <= 0 || N <= 0 || N > FRZ_MAX_NODES || Att < 0 || D < 0 || A <= 0 || A > FRZ_MAX_AGENTS) return FRZ_E_INVALID;
    if (cfg->num_states <= 0 || cfg->num_states > FRZ_MAX_NETWORK_STATES) return FRZ_E_INVALID;
    if ((int64_t)(N + 2 * D + 4 * A + 8) * cfg->parallel_envs >= (int64_t)1 << 30) return FRZ_E_INVALID;
    frz_cybersecurity_env* env = new (std::nothrow) frz_cybersecurity_env();
    if (!env) return FRZ_E_INVALID;
    env->cfg = *cfg;
    env->variant = (N <= 4 && A <= 4) ? 0 : ((N <= 8 && A <= 8) ? 1 : (A <= 8 ? 4 : 2));
    if (N == 3 && Att == 2 && D == 2) env->variant = 3;  // the reference's own test / competition shape gets an exact instantiation
    const char* family = std::getenv("FRZ_CY_KERNEL");
    env->roles = env->variant != 2 && !(family && std::strcmp(family, "lane") == 0);
    CyDev& p = env->dev;
    std::memset(&p, 0, sizeof(p));
    const int64_t B = cfg->parallel_envs;
    p.B = cfg->parallel_envs, p.N = N, p.Att = Att, p.D = D, p.A = A, p.S = cfg->num_states;
    p.nchunks = (cfg->parallel_envs + kBlock - 1) / kBlock;
    p.max_steps = cfg->max_steps;
    p.lut_entries = 1 << A;
    auto flag = [&](int on, uint32_t bit) { p.flags |= on ? bit : 0u; };
    flag(cfg->stochastic_state, kStochState);
    flag(cfg->show_bad_actions, kShowBad);
    flag(cfg->partially_observable, kPartial);
    flag(cfg->observe_other_power, kObsPower);
    flag(cfg->observe_other_presence, kObsPresence);
    flag(cfg->observe_other_location, kObsLocation);
    flag(cfg->track_cumulative_rewards, kTrackCumulative);
    flag(cfg->max_steps >= 0, kTruncate);
    p.patch_reward = cfg->patch_reward;
    std::memcpy(p.threat, cfg->threat, sizeof(p.threat));
    std::memcpy(p.mitigation, cfg->mitigation, sizeof(p.mitigation));
    std::memcpy(p.persist, cfg->persist_probs, sizeof(p.persist));
    std::memcpy(p.back, cfg->return_probs, sizeof(p.back));
    std::memcpy(p.initial_presence, cfg->initial_presence, sizeof(p.initial_presence));
    std::memcpy(p.initial_location, cfg->initial_location, sizeof(p.initial_location));
    std::memcpy(p.initial_state, cfg->initial_state, sizeof(p.initial_state));
    std::memcpy(p.criticality, cfg->criticality, sizeof(p.criticality));
    std::memcpy(p.state_rewards, cfg->network_state_rewards, sizeof(p.state_rewards));

    // danger table: tanh((patches - attacks) / T) for every (defender subset, attacker subset); the sums are the float32
    // accumulations the reference performs in agent order (cybersecurity.py:350, :377), tanhf is the libm the oracle uses
    env->lut.resize((size_t)1 << A);
    for (uint32_t pm = 0; pm < (1u << D); ++pm) {
        float patches = 0.0f;
        for (int k = 0; k < D; ++k)
            if (pm & (1u << k)) patches = patches + cfg->mitigation[k];
        for (uint32_t am = 0; am < (1u << Att); ++am) {
            float attacks = 0.0f;
            for (int a = 0; a < Att; ++a)
                if (am & (1u << a)) attacks = attacks + cfg->threat[a];
            const float diff = patches - attacks;
            env->lut[((size_t)pm << Att) | am] = std::tanh(diff / cfg->temperature);
        }
    }

    int r = 0;
    p.r_state = r, r += N;
    p.r_loc = r, r += D;
    p.r_last = r, r += D;
    p.r_moves = r++;
    p.r_rewards = r, r += A;
    p.r_cum = r, r += A;
    p.r_atc = r, r += A;
    p.r_etc = r++;
    p.r_seeds = r++;
    p.r_mti = r++;
    p.n_rows4 = r;
    p.u_presence = 0, p.u_term = A, p.u_trunc = 2 * A, p.u_frozen = 3 * A, p.n_rows1 = 3 * A + 1;
    const int ka = (cfg->observe_other_power ? 1 : 0) + (cfg->observe_other_presence ? 1 : 0);
    const int kd = ka + (cfg->observe_other_location ? 1 : 0);
    const int nch_total = A + 2;
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
    // the step kernel derives these from (N, D, A) alone (its loads start before the configuration block is staged)
    if (p.off_rows4 != kDevBlockBytes || p.r_state != 0 || p.r_loc != N || p.r_last != N + D || p.r_moves != N + 2 * D ||
        p.r_seeds != p.r_moves + 3 * A + 2 || p.u_presence != 0 || p.u_trunc != 2 * A) {
        delete env;
        return FRZ_E_INVALID;
    }
    p.off_self_att = take((int64_t)Att * B * 8);
    p.off_self_def = take((int64_t)D * B * 12);
    p.off_others_att = take((int64_t)Att * B * (Att > 0 ? Att - 1 : 0) * ka * 4);
    p.off_others_def = take((int64_t)D * B * (D > 0 ? D - 1 : 0) * kd * 4);
    p.off_tasks = take((int64_t)A * B * N * 16);
    p.off_act_values = take((int64_t)A * B * N * 4);
    p.off_act_offsets = take((int64_t)A * (B + 1) * 8);
    p.off_obs_map = take(B * N * 4);
    p.off_obs_map_offsets = take((B + 1) * 8);
    p.off_lut = take((int64_t)p.lut_entries * 4);
    p.off_actions = take((int64_t)A * B * 8);
    p.off_error = take(256);
    p.off_epoch = take(256);
    p.off_totals = take(2 * frz::kTotalsStride * 4);
    p.off_agg = take((int64_t)p.nchunks * nch_total * 8);
    p.off_prefix = take((int64_t)p.nchunks * nch_total * 8);
    p.off_rand_net = take(B * N * 4);
    p.off_rand_agent = take(B * A * 4);
    p.off_mt_state = take(624 * B * 4);
    // second copy of the packed action-mapping values: where the shape has a multi-step launch (state / view kernel, up to 8 nodes)
    if (env->roles && env->variant != 4) env->copy_delta = take((int64_t)A * B * N * 4) - p.off_act_values;
    p.total_bytes = off;

    int device = 0, cus = 256;
    if (hipGetDevice(&device) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    env->ticketed = p.nchunks > cus;  // one workgroup per chunk; one 256-thread workgroup per CU is always resident
    frz::handle_register(env, 2, A, cfg->parallel_envs, N);
    *out = env;
    return FRZ_OK;
}

void frz_cybersecurity_destroy(frz_cybersecurity_env* env) {
    if (env) frz::handle_unregister(env);
    delete env;
}

int64_t frz_cybersecurity_arena_bytes(const frz_cybersecurity_env* env) { return env ? env->dev.total_bytes : FRZ_E_INVALID; }

int frz_cybersecurity_bind(frz_cybersecurity_env* env, void* arena, void* stream) {
    if (!env || !arena || reinterpret_cast<uintptr_t>(arena) % 256 != 0) return FRZ_E_INVALID;
    env->arena = static_cast<char*>(arena);
    env->was_reset = false;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(arena, &env->dev, sizeof(CyDev), hipMemcpyHostToDevice, s) != hipSuccess) return FRZ_E_LAUNCH;
    if (hipMemcpyAsync(env->arena + env->dev.off_lut, env->lut.data(), env->lut.size() * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess)
        return FRZ_E_LAUNCH;
    {   // every agent's packed values are B * N ints, agent after agent: position mod N within an agent's block
        const CyDev& p = env->dev;
        const int64_t per_agent = (int64_t)p.B * p.N;
        for (int a = 0; a < p.A; ++a)
            hipLaunchKernelGGL(cy_prefill_kernel, dim3((unsigned)((per_agent + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                               reinterpret_cast<int32_t*>(env->arena + p.off_act_values) + a * per_agent, per_agent, p.N);
        if (hipGetLastError() != hipSuccess) return FRZ_E_LAUNCH;
    }
    return hipStreamSynchronize(s) == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

int frz_cybersecurity_get_bufs(const frz_cybersecurity_env* env, frz_cybersecurity_bufs* out) {
    if (!env || !out) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    const CyDev& p = env->dev;
    char* a = env->arena;
    const int64_t B = p.B;
    auto row4 = [&](int r) { return a + p.off_rows4 + (int64_t)r * B * 4; };
    auto row1 = [&](int r) { return a + p.off_rows1 + (int64_t)r * B; };
    out->network_state = reinterpret_cast<int32_t*>(row4(p.r_state));
    out->location = reinterpret_cast<int32_t*>(row4(p.r_loc));
    out->last_action = reinterpret_cast<int32_t*>(row4(p.r_last));
    out->num_moves = reinterpret_cast<int32_t*>(row4(p.r_moves));
    out->rewards = reinterpret_cast<float*>(row4(p.r_rewards));
    out->cumulative_rewards = reinterpret_cast<float*>(row4(p.r_cum));
    out->agent_task_count = reinterpret_cast<int32_t*>(row4(p.r_atc));
    out->env_task_count = reinterpret_cast<int32_t*>(row4(p.r_etc));
    out->seeds = reinterpret_cast<int32_t*>(row4(p.r_seeds));
    out->mt_index = reinterpret_cast<int32_t*>(row4(p.r_mti));
    out->presence = reinterpret_cast<uint8_t*>(row1(p.u_presence));
    out->terminations = reinterpret_cast<uint8_t*>(row1(p.u_term));
    out->truncations = reinterpret_cast<uint8_t*>(row1(p.u_trunc));
    out->frozen_scaled = reinterpret_cast<uint8_t*>(row1(p.u_frozen));
    out->obs_self_attackers = at<float>(a, p.off_self_att);
    out->obs_self_defenders = at<float>(a, p.off_self_def);
    out->obs_others_attackers = at<float>(a, p.off_others_att);
    out->obs_others_defenders = at<float>(a, p.off_others_def);
    out->obs_tasks = at<int64_t>(a, p.off_tasks);
    out->act_map_values = at<int32_t>(a, p.off_act_values);
    out->act_map_offsets = at<int64_t>(a, p.off_act_offsets);
    out->obs_map_values = at<int32_t>(a, p.off_obs_map);
    out->obs_map_offsets = at<int64_t>(a, p.off_obs_map_offsets);
    out->mt_state = at<uint32_t>(a, p.off_mt_state);
    out->actions = at<int32_t>(a, p.off_actions);
    out->error_flags = at<uint32_t>(a, p.off_error);
    return FRZ_OK;
}

int frz_cybersecurity_rebuild(frz_cybersecurity_env* env, void* stream) {
    if (!env) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    env->was_reset = true;
    return launch(env, nullptr, nullptr, nullptr, FRZ_RNG_INJECTED, kRebuild, static_cast<hipStream_t>(stream));
}

int frz_cybersecurity_reset_masked(frz_cybersecurity_env* env, const uint8_t* mask, int32_t seed_increment, void* stream) {
    if (!env) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;
    const int blocks = (env->cfg.parallel_envs + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(cy_masked_fill_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), env->arena, mask, (uint32_t)seed_increment,
                       env->saved);
    if (hipGetLastError() != hipSuccess) return FRZ_E_LAUNCH;
    return frz_cybersecurity_rebuild(env, stream);
}

int frz_cybersecurity_set_saved_initial(frz_cybersecurity_env* env, const frz_cybersecurity_saved_state* saved) {
    if (!env) return FRZ_E_INVALID;
    if (saved && (!saved->network_state || !saved->location || !saved->presence)) return FRZ_E_INVALID;
    env->saved = saved ? *saved : frz_cybersecurity_saved_state{};
    return FRZ_OK;
}

int frz_cybersecurity_reset(frz_cybersecurity_env* env, void* stream) {
    if (!env) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    env->saved = frz_cybersecurity_saved_state{};  // a full reset saves the configured initial state again
    const int blocks = (env->cfg.parallel_envs + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(cy_fill_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), env->arena);
    if (hipGetLastError() != hipSuccess) return FRZ_E_LAUNCH;
    return frz_cybersecurity_rebuild(env, stream);
}

int frz_mt19937_generate_pair(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events, int64_t count, float* out2, int64_t events2,
                              int64_t count2, int64_t B, void* stream);

static int step_impl(frz_cybersecurity_env* env, const int32_t* actions, int rng_mode, const float* network_randomness,
                     const float* agent_randomness, void* stream, const Policy& policy);

int frz_cybersecurity_step(frz_cybersecurity_env* env, const int32_t* actions, int rng_mode, const float* network_randomness,
                           const float* agent_randomness, void* stream) {
    return step_impl(env, actions, rng_mode, network_randomness, agent_randomness, stream, Policy());
}

int frz_cybersecurity_step_random_policy(frz_cybersecurity_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out,
                                         int rng_mode, const float* network_randomness, const float* agent_randomness, void* stream) {
    if (!actions_out) return FRZ_E_INVALID;
    Policy policy;
    policy.on = true, policy.seed = policy_seed, policy.step = policy_step, policy.actions_out = actions_out;
    return step_impl(env, actions_out, rng_mode, network_randomness, agent_randomness, stream, policy);
}

static int step_impl(frz_cybersecurity_env* env, const int32_t* actions, int rng_mode, const float* network_randomness,
                     const float* agent_randomness, void* stream, const Policy& policy) {
    if (!env || !actions) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;
    const CyDev& p = env->dev;
    if (rng_mode == FRZ_RNG_INJECTED) {
        if (!network_randomness || !agent_randomness) return FRZ_E_INVALID;
    } else if (rng_mode == FRZ_RNG_MT19937 && env->variant != 2 && env->variant != 4) {
        // the step kernel advances the env's own stream (network draws first, then agent draws, cybersecurity.py:304-315)
    } else if (rng_mode == FRZ_RNG_MT19937) {  // the 16-node variants: the same draws staged in the arena by a generator launch
        const int64_t B = p.B;
        uint32_t* mt_state = at<uint32_t>(env->arena, p.off_mt_state);
        int32_t* mt_index = at<int32_t>(env->arena, p.off_rows4 + (int64_t)p.r_mti * B * 4);
        float* rn = at<float>(env->arena, p.off_rand_net);
        float* ra = at<float>(env->arena, p.off_rand_agent);
        // (a frozen batch draws nothing: the step launch below will be a no-op, and so is the reference's step then)
        const int rc = frz::mt19937_generate_pair_gated(mt_state, mt_index, rn, 1, p.N, ra, 1, p.A, B, at<uint32_t>(env->arena, p.off_epoch),
                                                        at<uint32_t>(env->arena, p.off_totals), p.A, frz::kTotalsStride, stream);
        if (rc != FRZ_OK) return rc;
        network_randomness = rn;
        agent_randomness = ra;
        rng_mode = FRZ_RNG_INJECTED;
    } else if (rng_mode != FRZ_RNG_PHILOX) {
        return FRZ_E_INVALID;
    }
    return launch(env, actions, network_randomness, agent_randomness, rng_mode, kStep, static_cast<hipStream_t>(stream), policy);
}

extern "C" int frz_exclusive_launch_fits(int64_t workgroups, int workgroups_per_cu, int compute_units, int cu_mask_set);

int frz_cybersecurity_set_exclusive_device(frz_cybersecurity_env* env, int exclusive) {
    if (!env) return FRZ_E_INVALID;
    if (!exclusive || !env->roles || env->copy_delta == 0) {  // switching off / no multi-step kernel for this shape: nothing to guard
        env->exclusive_device = exclusive != 0;
        return FRZ_OK;
    }
    // the library's own part of the residency promise (see frz_wildfire_set_exclusive_device): one 512-thread workgroup per chunk, all at once,
    // on the device that owns the arena.  The multi-step instantiations need a whole CU's registers: one workgroup per CU.
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
    int cus = prop.multiProcessorCount;
    if (const char* assumed = std::getenv("FRZ_ASSUME_COMPUTE_UNITS")) cus = std::atoi(assumed);
    const bool masked = std::getenv("ROC_GLOBAL_CU_MASK") != nullptr || std::getenv("HSA_CU_MASK") != nullptr;
    if (!frz_exclusive_launch_fits(env->dev.nchunks, 1, cus, masked ? 1 : 0)) return FRZ_E_INVALID;
    env->exclusive_device = true;
    return FRZ_OK;
}

int frz_cybersecurity_rollout_launches(const frz_cybersecurity_env* env, int32_t n_steps, int rng_mode) {
    if (!env || n_steps < 0) return FRZ_E_INVALID;
    const bool one = n_steps > 1 && env->exclusive_device && env->roles && env->copy_delta != 0 && !env->ticketed &&
                     (rng_mode == FRZ_RNG_PHILOX || rng_mode == FRZ_RNG_INJECTED);
    return one ? 1 : n_steps;
}

int frz_cybersecurity_rollout_random_policy(frz_cybersecurity_env* env, uint64_t policy_seed, uint64_t first_step, int32_t n_steps,
                                            int32_t* actions_out, int rng_mode, void* stream) {
    if (!env || !actions_out || n_steps < 0) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;
    if (n_steps == 0) return FRZ_OK;
    if (n_steps > 1 && frz_cybersecurity_rollout_launches(env, n_steps, rng_mode) == 1) {  // one launch, the envs in registers across its steps
        env->rollout_steps = n_steps;
        const int rc = frz_cybersecurity_step_random_policy(env, policy_seed, first_step, actions_out, rng_mode, nullptr, nullptr, stream);
        env->rollout_steps = 1;
        return rc;
    }
    for (int32_t t = 0; t < n_steps; ++t) {
        const int rc = frz_cybersecurity_step_random_policy(env, policy_seed, first_step + (uint64_t)t, actions_out, rng_mode, nullptr, nullptr, stream);
        if (rc != FRZ_OK) return rc;
    }
    return FRZ_OK;
}

int frz_cybersecurity_list_block(const frz_cybersecurity_env* env, void** block, int64_t* bytes) {
    if (!env || !block || !bytes) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    *block = env->arena + env->dev.off_act_values;
    *bytes = env->dev.off_obs_map - env->dev.off_act_values;
    return FRZ_OK;
}

int frz_cybersecurity_obs_block(const frz_cybersecurity_env* env, void** block, int64_t* bytes) {
    if (!env || !block || !bytes) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    *block = env->arena + env->dev.off_self_att;
    *bytes = env->dev.off_act_values - env->dev.off_self_att;
    return FRZ_OK;
}

int frz_cybersecurity_state_block(const frz_cybersecurity_env* env, void** rows, int64_t* rows_bytes, void** presence, int64_t* presence_bytes) {
    if (!env || !rows || !rows_bytes || !presence || !presence_bytes) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    const CyDev& p = env->dev;
    *rows = env->arena + p.off_rows4 + (int64_t)p.r_state * p.B * 4;
    *rows_bytes = (int64_t)(p.N + 2 * p.D) * p.B * 4;
    *presence = env->arena + p.off_rows1 + (int64_t)p.u_presence * p.B;
    *presence_bytes = (int64_t)p.A * p.B;
    return FRZ_OK;
}

int frz_cybersecurity_rollout(frz_cybersecurity_env* env, const frz_rollout_spec* spec, void* stream) {
    if (!env || !spec || spec->n_steps < 0) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    if (!env->was_reset) return FRZ_E_INVALID;
    const CyDev& p = env->dev;
    const int mode = spec->rng_mode;
    const bool policy = spec->action_tape == nullptr, reset_first = (spec->flags & FRZ_ROLLOUT_RESET_FIRST) != 0;
    if (mode != FRZ_RNG_INJECTED && mode != FRZ_RNG_PHILOX && mode != FRZ_RNG_MT19937) return FRZ_E_INVALID;
    if (mode == FRZ_RNG_INJECTED && spec->n_steps > 0 && (!spec->randomness_tape_a || !spec->randomness_tape_b)) return FRZ_E_INVALID;
    if (policy && !spec->actions_out) return FRZ_E_INVALID;
    if ((spec->flags & FRZ_ROLLOUT_AUTO_RESET) || spec->metrics) return FRZ_E_INVALID;  // (episodes end together here: cybersecurity.py:298; no metrics entry)
    if (spec->n_steps == 0) return reset_first ? frz_cybersecurity_reset(env, stream) : FRZ_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t B = p.B, A = p.A, N = p.N, AB2 = A * B * 2;
    const int64_t block_bytes = p.off_obs_map - p.off_act_values;

    if (spec->flags & FRZ_ROLLOUT_OBS_COMPACT) return FRZ_E_INVALID;  // (every column of this domain's observation rows can change)
    const int64_t obs_block_bytes = p.off_act_values - p.off_self_att;
    const int64_t state_step_bytes = align_up((int64_t)(N + 2 * p.D) * B * 4 + A * B, 256);
    if (spec->n_steps > 1 && frz_cybersecurity_rollout_launches(env, spec->n_steps, mode) == 1) {  // ONE multi-step launch
        frz_cybersecurity_env::RolloutOptions& o = env->rollout;
        o.extra = !policy || spec->list_record || spec->reward_tape || spec->done_tape || spec->record_actions || reset_first || spec->obs_tape || spec->state_tape;
        o.obs_tape_delta = spec->obs_tape ? static_cast<char*>(spec->obs_tape) - (env->arena + p.off_self_att) : 0;
        o.obs_tape_step = obs_block_bytes;
        o.state_tape = static_cast<char*>(spec->state_tape);
        o.state_tape_step = state_step_bytes;
        o.flags = spec->flags;
        o.tape_actions_step = policy ? 0 : AB2;
        o.list_record_delta = spec->list_record ? static_cast<char*>(spec->list_record) - (env->arena + p.off_act_values) : 0;
        o.list_record_step = block_bytes;
        o.reward_tape = spec->reward_tape, o.done_tape = spec->done_tape;
        o.actions_out_step = spec->record_actions ? AB2 : 0;
        env->rollout_steps = spec->n_steps;
        int rc;
        if (policy)
            rc = frz_cybersecurity_step_random_policy(env, spec->policy_seed, spec->first_step, spec->actions_out, mode, spec->randomness_tape_a,
                                                      spec->randomness_tape_b, stream);
        else
            rc = frz_cybersecurity_step(env, spec->action_tape, mode, spec->randomness_tape_a, spec->randomness_tape_b, stream);
        env->rollout_steps = 1;
        o = frz_cybersecurity_env::RolloutOptions();
        return rc;
    }
    // the same step by step
    if (reset_first) {
        const int rc = frz_cybersecurity_reset(env, stream);
        if (rc != FRZ_OK) return rc;
    }
    for (int32_t t = 0; t < spec->n_steps; ++t) {
        const float* ra = spec->randomness_tape_a ? spec->randomness_tape_a + (int64_t)t * B * N : nullptr;
        const float* rb = spec->randomness_tape_b ? spec->randomness_tape_b + (int64_t)t * B * A : nullptr;
        int rc;
        if (policy)
            rc = frz_cybersecurity_step_random_policy(env, spec->policy_seed, spec->first_step + (uint64_t)t,
                                                      spec->actions_out + (spec->record_actions ? (int64_t)t * AB2 : 0), mode, ra, rb, stream);
        else
            rc = frz_cybersecurity_step(env, spec->action_tape + (int64_t)t * AB2, mode, ra, rb, stream);
        if (rc != FRZ_OK) return rc;
        bool ok = true;
        if (spec->reward_tape)
            ok = ok && hipMemcpyAsync(spec->reward_tape + (int64_t)t * A * B, env->arena + p.off_rows4 + (int64_t)p.r_rewards * B * 4, (size_t)(A * B * 4),
                                      hipMemcpyDeviceToDevice, s) == hipSuccess;
        if (spec->done_tape) {
            ok = ok && hipMemsetAsync(spec->done_tape + ((int64_t)t * 2 + 0) * B, 0, (size_t)B, s) == hipSuccess;
            ok = ok && hipMemcpyAsync(spec->done_tape + ((int64_t)t * 2 + 1) * B, env->arena + p.off_rows1 + (int64_t)p.u_trunc * B, (size_t)B,
                                      hipMemcpyDeviceToDevice, s) == hipSuccess;
        }
        if (spec->list_record && t < spec->n_steps - 1)
            ok = ok && hipMemcpyAsync(static_cast<char*>(spec->list_record) + (int64_t)t * block_bytes, env->arena + p.off_act_values, (size_t)block_bytes,
                                      hipMemcpyDeviceToDevice, s) == hipSuccess;
        if (spec->obs_tape) {
            const int64_t obs_bytes = p.off_act_values - p.off_self_att;
            ok = ok && hipMemcpyAsync(static_cast<char*>(spec->obs_tape) + (int64_t)t * obs_bytes, env->arena + p.off_self_att, (size_t)obs_bytes,
                                      hipMemcpyDeviceToDevice, s) == hipSuccess;
        }
        if (spec->state_tape) {
            const int64_t rows_bytes = (int64_t)(N + 2 * p.D) * B * 4, pres_bytes = A * B;
            char* const dst = static_cast<char*>(spec->state_tape) + (int64_t)t * align_up(rows_bytes + pres_bytes, 256);  // (a step starts aligned)
            ok = ok && hipMemcpyAsync(dst, env->arena + p.off_rows4 + (int64_t)p.r_state * B * 4, (size_t)rows_bytes, hipMemcpyDeviceToDevice, s) == hipSuccess;
            ok = ok && hipMemcpyAsync(dst + rows_bytes, env->arena + p.off_rows1 + (int64_t)p.u_presence * B, (size_t)pres_bytes, hipMemcpyDeviceToDevice, s) ==
                           hipSuccess;
        }
        if (!ok) return FRZ_E_LAUNCH;
    }
    return FRZ_OK;
}

int frz_cybersecurity_random_policy(frz_cybersecurity_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out,
                                    void* stream) {
    if (!env || !actions_out) return FRZ_E_INVALID;
    if (!env->arena) return FRZ_E_UNBOUND;
    const int64_t n = (int64_t)env->dev.A * env->dev.B;
    hipLaunchKernelGGL(cy_policy_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       env->arena, (uint32_t)policy_seed, (uint32_t)(policy_seed >> 32), (uint32_t)policy_step,
                       (uint32_t)(policy_step >> 32), actions_out);
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

}  // extern "C"
