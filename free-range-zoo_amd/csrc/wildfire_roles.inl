// wildfire_roles.inl — the fused wildfire step for small grids (<= 8 cells) with TWO wavefronts per 64 environments.
//
// Why: at batch 65 536 the lane-per-env kernel (wildfire.hip) puts ONE wavefront on each SIMD of the chip, and a lone
// wavefront issues a vector instruction every ~4 cycles where the SIMD could take one every 2 (MI355X_MICROARCH.md,
// 'vector-instruction ISSUE cost'); its stores also block its own arithmetic while they issue.  Here a 512-thread
// workgroup owns a 256-env chunk and splits the step by ROLE, not by lane: wavefronts 0-3 ("field") run the fire
// transitions, the cell rows and the task list of the chunk's envs, wavefronts 4-7 ("crew") run the action decode, the
// agent transitions, rewards/bookkeeping, the open-action scan and the action lists of the SAME envs (lane i of wave w
// and of wave w + 4 hold the same env).  Each SIMD then carries one field and one crew wavefront whose instruction
// streams interleave, and the same total work finishes in roughly half the cycles.  The roles exchange four small
// per-env words through LDS: applied power per cell (crew -> field), lit mask and burned/put-out/dead bits
// (field -> crew), the env's offset inside the chunk's task segment (crew -> field).
//
// Same arena layout, same hand-off protocol, same results as wildfire.hip (the parity tests run both).
#include "frz_device.h"

#include "../../include/frz.h"

#include <type_traits>

#include "wildfire_common.h"

namespace frz_wf {

namespace {

constexpr int kRoleBlock = 2 * kBlock;  // 4 field + 4 crew wavefronts per 256-env chunk
constexpr int kRound = 256;             // look-back window: a chunk sums at most kRound - 1 predecessors' granules + one prefix granule

// Diagnostic build only (-DFRZ_WF_STAMPS, tools/stamps.py): the first thread of each role of ONE workgroup (0, or bits 16.. of FRZ_WF_SKIP) records the
// shader clock at phase boundaries into a buffer nothing else reads.  No stamp executes in the production library.
#ifdef FRZ_WF_STAMPS
#define FRZ_RSTAMP(i)                                                                                                       \
    do {                                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        if (EXACT && blockIdx.x == ((launch.skip >> 16) & 0xFFFu) && slot == 0 && MODE == kStep && frz_stamp_step)         \
            reinterpret_cast<unsigned long long*>(arena + dev->off_rand_agent)[(crew ? 16 : 0) + (i)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
    } while (0)
// per-workgroup wall clock (100 MHz): [2 * blockIdx.x] = first instruction, [2 * blockIdx.x + 1] = last, both roles' minimum/maximum
#define FRZ_RWALL(which)                                                                                                     \
    do {                                                                                                                     \
        if (EXACT && slot == 0 && MODE == kStep)                                                                             \
            reinterpret_cast<unsigned long long*>(arena + dev->off_rand_field)[4 * blockIdx.x + 2 * (crew ? 1 : 0) + (which)] = \
                __builtin_amdgcn_s_memrealtime();                                                                            \
    } while (0)
#else
#define FRZ_RSTAMP(i) \
    do {              \
    } while (0)
#define FRZ_RWALL(which) \
    do {                 \
    } while (0)
#endif

// PERSIST (exact shapes, FRZ_RNG_PHILOX, fused policy): launch.n_steps steps in ONE launch — the workgroup keeps its chunk's state in
// registers from step to step and only the per-step outputs leave the CU.  What makes that legal on a part whose eight L2s are not
// coherent with each other: (i) every array a step writes at env-indexed addresses is rewritten by the SAME workgroup at the next
// step; (ii) the packed lists, whose addresses move between workgroups from step to step (two XCDs' L2s could end up holding dirty
// bytes of one address from different steps, written back in any order), go to the caller's buffers only at the LAST step of the
// launch — the one whose lists are visible afterwards — and to a second copy of those buffers, which nobody reads, at every step
// before it (a launch that finds the batch finished early writes its last lists once more, into the caller's buffers); (iii) what
// crosses workgroups (chunk sums, batch totals) travels in tagged granules read with agent-scope loads.  The totals of step t — which every workgroup needs before step t + 1:
// all-done test, skip-agent quirk — are the last chunk's inclusive-prefix granules: waiting for them is the only inter-step barrier.
// EXTRA (multi-step launches only): everything a frz_rollout_spec can ask for beyond "policy sampled in-kernel, opening reset, episode
// metrics" — an action tape, randomness tapes, reward / done / action records, the list record, FRZ_ROLLOUT_AUTO_RESET.  A separate
// instantiation because the plain rollout is the kernel the bench times: with these as run-time options its step was 0.45 us longer.
template <int CMAX, int AMAX, bool EXACT, int RNG, int MODE, bool PERSIST = false, bool EXTRA = false>
__global__ void __launch_bounds__(kRoleBlock, ((RNG == FRZ_RNG_MT19937 || CMAX > 8 || !EXACT || PERSIST) ? 2 : 4)) wf_roles_kernel(char* __restrict__ arena, const WfDev* __restrict__ dev,
                                                               const int32_t* __restrict__ actions, const float* __restrict__ field_rand,
                                                               const float* __restrict__ agent_rand, const WfLaunch launch) {
    const int32_t batch = launch.batch;
    using mask_t = uint32_t;
    constexpr int MB = CMAX <= 8 ? 8 : (CMAX <= 16 ? 16 : 32);             // bits of a cell mask in the exchange words
    static_assert(CMAX <= 24, "cell masks are 32-bit words");
    // the agents' attackable-cell masks travel crew -> field as 8- or 16-bit fields of ONE word where they fit (<= 64 bits: every shape of
    // rounds 1-3), as a word per agent otherwise (<16, 8>, <24, 8>: round 4)
    constexpr bool kOkPacked = MB <= 16 && AMAX * MB <= 64;
    using pack_t = std::conditional_t<(kOkPacked && AMAX * MB <= 32), uint32_t, uint64_t>;  // one mask per agent
    using fate_t = std::conditional_t<(MB == 8), uint32_t, uint64_t>;  // burned | put_out << MB | dead << 2 MB (MB == 32: dead travels in x_dead)
    constexpr int PW = (AMAX + 1 + 3) / 4;                                // packed scan words (four 16-bit channels each)
    constexpr int NCHP = AMAX + 3 <= 8 ? 8 : (AMAX + 3 <= 16 ? 16 : 32);  // scan channels padded to a power of two
    constexpr bool kPhilox = RNG == FRZ_RNG_PHILOX && MODE == kStep;
    constexpr bool kMt = RNG == FRZ_RNG_MT19937 && MODE == kStep;  // per-env MT19937 streams advanced inside the step
    constexpr bool kInjected = RNG == FRZ_RNG_INJECTED && MODE == kStep;
    // multi-step launches with in-kernel Philox draws: the field role writes EVERY list (see emit_field).  Not with the per-env MT19937 streams:
    // there the field role's own phase 1 (66 stream words in, 33 twisted and written back) is the long one, and the old split is faster
    // (round 4: 12.2 against 10.7 us per step of the default-RNG block when it wrote all the lists too)
#ifndef FRZ_WF_CREW_LISTS
#define FRZ_WF_CREW_LISTS 0  // agents whose action lists the crew role writes in such a launch (the first n; experiment builds vary it)
#endif
    constexpr bool kFieldWritesAllLists = PERSIST && RNG == FRZ_RNG_PHILOX;
    // which role writes agent a's action lists
    auto crew_writes = [](int a) constexpr { return kFieldWritesAllLists ? a < FRZ_WF_CREW_LISTS : (a & 1) == 0; };
    // runtime shapes draw in the kernel too (round 4; their MT19937 streams only in a multi-step launch: a single step reads frz_mt19937_generate's output)
    // like the exact shapes do, its draw indices (which depend on H * W and A) resolved by scattering them to their places in LDS (x_fd, x_draw)
    static_assert(EXACT || !kMt || PERSIST, "runtime shapes keep their MT19937 streams outside a single step's kernel (frz_mt19937_generate)");

    __shared__ uint64_t s_wave_scan[frz::kWaves][PW];
    __shared__ uint32_t s_wave_live[frz::kWaves][2];
    __shared__ uint32_t s_reduce[frz::kWaves][32];
    __shared__ uint32_t s_prefix[32];
    __shared__ WfStaged s_cfg;  // hot scalars (copied to registers below) + the per-lane lookup tables (range sets, equipment, capacities)
    __shared__ float x_power[CMAX][kBlock];  // crew -> field: fire-fighting power applied to each cell
    __shared__ uint32_t x_lit[kBlock];       // field -> crew: lit cells after the transitions
    __shared__ fate_t x_fate[kBlock];        // field -> crew: burned | put_out << MB | dead << 2 MB
    __shared__ uint64_t x_excl[PW][kBlock];  // crew -> both: packed per-env counts of the preceding envs of the same wavefront
    __shared__ pack_t x_ok[kOkPacked ? kBlock : 1];  // crew -> both: attackable cells of agent a in field a (MB bits each, AMAX * MB <= 64)
    __shared__ uint32_t x_okv[kOkPacked ? 1 : AMAX][kOkPacked ? 1 : kBlock];  // ... or a word per agent
    __shared__ uint8_t x_dead[MB == 32 ? kBlock : 1];
    __shared__ float x_supp[AMAX][kBlock];   // crew -> field: suppressant after the agent transitions (agent observations)
    // field -> crew: the step's agent draws (in-kernel RNG).  Two copies in a multi-step Philox launch: the field parks the NEXT step's
    // draws in copy (t + 1) & 1 as soon as it has made them, instead of holding 5 * AMAX registers across its list phase and the loop's
    // back edge (they were spilled)
    // runtime shapes, in-kernel Philox: draw u of the env (u = e * H * W + c for the field events, 3 * H * W + e * A + a for the agent events)
    // Two ways, by LDS budget: a scratch column per env that holds every draw of the step, read back at the runtime positions (x_uni: the
    // faster one — 1 x 7 / 3 agents 10.1 against 12.8 us per step — but 65 KB at <8, 8>, which would take that kernel past the 160 KB of
    // LDS), or every draw stored to ITS place (x_fd[e * CMAX + c] for the field draws, the crew's x_draw rows for the agent draws)
    constexpr bool kRuntimeDraws = !EXACT && (kPhilox || kMt);
    constexpr bool kGatherDraws = kRuntimeDraws && AMAX <= 4, kScatterDraws = kRuntimeDraws && AMAX > 4;
    __shared__ float x_uni[kGatherDraws ? 5 * ((3 * CMAX + 5 * AMAX + 4) / 5) : 1][kGatherDraws ? kBlock : 1];
    __shared__ float x_fd[kScatterDraws ? 3 * CMAX : 1][kScatterDraws ? kBlock : 1];
    // (the 8-agent runtime-shape variants keep ONE copy, 40 KB instead of 80: the next step's draws are parked behind barrier 3, by when the
    // crew has long moved this step's into registers — phase 2 — so the second copy only buys slack the exact kernels' schedule wants)
    constexpr int kDrawCopies = (RNG == FRZ_RNG_PHILOX && PERSIST && !(!EXACT && AMAX > 4)) ? 2 : 1;
    __shared__ float x_draw[kDrawCopies][(kPhilox || kMt) ? 5 * AMAX : 1][kBlock];
    // FRZ_ROLLOUT_AUTO_RESET: returns of the episodes that ended inside this launch (float64, per env slot: deterministic) and their number
    __shared__ double x_return[EXTRA ? AMAX : 1][EXTRA ? kBlock : 1];
    __shared__ uint32_t x_ended[EXTRA ? kBlock : 1];

    const int tid = threadIdx.x;
    const uint4 cfg_piece = stage_request(dev);  // first vector-memory instruction of the kernel
    // wave-uniform role.  A workgroup's wavefronts are started one after the other (the last ≈1 000 cycles after the first); the crew is
    // the role the first barrier waits for, so it gets the wavefronts that start first
    const bool crew = __builtin_amdgcn_readfirstlane(tid >> 8) == 0;
    const int slot = tid & (kBlock - 1);                                // env slot inside the chunk; the role's thread index
    const int lane = frz::lane_id(), wave = (tid >> 6) & 3;
    const int64_t B = batch;
    const uint32_t Bu = (uint32_t)batch;
    const int nchunks = (int)((B + kBlock - 1) / kBlock);
    const int HW = EXACT ? CMAX : dev->HW, A = EXACT ? AMAX : dev->A;
    // rows of the [rows][B] block: a fixed function of (HW, A) (frz_wildfire_create lays them out in this order)
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

    // ---- per-role registers of one env
    struct Cells {  // both roles read the fires (the crew derives the open task set from them)
        int f[CMAX];
    };
    struct FieldRegs {
        int in[CMAX], fu[CMAX], nm, mti;
        uint32_t seed;
    };
    struct CrewRegs {
        int eqs[AMAX], act_idx[AMAX], act_id[AMAX], nm, nb;
        float supp[AMAX], capa[AMAX], cum[AMAX];
        uint32_t seed, term, trunc;
    };
    struct FieldDraws {
        float r[3][CMAX];
    };
    struct CrewDraws {
        float r[5][AMAX];
    };
    struct Nothing {};
    using FDraws = std::conditional_t<kInjected, FieldDraws, Nothing>;
    using CDraws = std::conditional_t<kInjected, CrewDraws, Nothing>;

    auto env_index = [&](int chunk) {
        const int64_t b = (int64_t)chunk * kBlock + slot;
        return (uint32_t)(b < B ? b : B - 1);  // lanes past the end shadow the last env
    };
    auto settle = [&](int chunk) {
        // A shadow lane reads rows its env's owner stores later in the iteration: in the one chunk that has shadow
        // lanes the loads complete before the workgroup barriers that precede those stores.
        if (chunk == nchunks - 1 && (B % kBlock) != 0) __builtin_amdgcn_s_waitcnt(0);
    };
    // multi-step launches (frz_wildfire_rollout): FRZ_ROLLOUT_RESET_FIRST starts from the configured initial state (taken from the staged
    // configuration below instead of loaded here), FRZ_ROLLOUT_AUTO_RESET resets an env at the end of the step that finishes it
    const bool reset_first = PERSIST && (launch.rollout_flags & FRZ_ROLLOUT_RESET_FIRST) != 0;
    const bool auto_reset = EXTRA && (launch.rollout_flags & FRZ_ROLLOUT_AUTO_RESET) != 0;
    auto load_cells = [&](int chunk) {
        Cells e;
        const uint32_t bl = env_index(chunk);
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (MODE == kReset)  // wildfire.py:347-349: +type on the configured lit cells, -type elsewhere
                e.f[c] = c < HW ? (dev->lit[c] ? dev->fire_types[c] : -dev->fire_types[c]) : 0;
            else if (reset_first)
                e.f[c] = 0;
            else
                e.f[c] = c < HW ? at32(rows, (uint32_t)(r_fires + c) * Bu + bl) : 0;
        }
        return e;
    };
    auto load_field_draws = [&](int chunk, int t, FDraws& draws) {  // FRZ_RNG_INJECTED: the step's field tensor ([3][B][H*W], one per step on a tape)
        if constexpr (kInjected) {
            const uint32_t bl = env_index(chunk);
            const float* const src = field_rand + (int64_t)t * 3 * B * HW;
#pragma unroll
            for (int ev = 0; ev < 3; ++ev)
#pragma unroll
                for (int c = 0; c < CMAX; ++c) draws.r[ev][c] = c < HW ? src[((int64_t)ev * B + bl) * HW + c] : 1.0f;
        }
    };
    auto load_crew_draws = [&](int chunk, int t, CDraws& draws) {  // ... and its agent tensor ([5][B][A])
        if constexpr (kInjected) {
            const uint32_t bl = env_index(chunk);
            const float* const src = agent_rand + (int64_t)t * 5 * B * A;
#pragma unroll
            for (int ev = 0; ev < 5; ++ev)
#pragma unroll
                for (int a = 0; a < AMAX; ++a) draws.r[ev][a] = a < A ? src[((int64_t)ev * B + bl) * A + a] : 1.0f;
        }
    };
    auto load_field = [&](int chunk, FDraws& draws) {
        FieldRegs e;
        const uint32_t bl = env_index(chunk);
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (MODE == kReset) {  // wildfire.py:350-351
                e.in[c] = (c < HW && dev->lit[c]) ? dev->ignition[c] : 0;
                e.fu[c] = (c < HW && dev->fire_types[c] != 0) ? dev->initial_fuel : 0;
            } else if (reset_first) {
                e.in[c] = e.fu[c] = 0;
            } else {
                e.in[c] = c < HW ? at32(rows, (uint32_t)(r_intensity + c) * Bu + bl) : 0;
                e.fu[c] = c < HW ? at32(rows, (uint32_t)(r_fuel + c) * Bu + bl) : 0;
            }
        }
        e.nm = e.mti = 0;
        e.seed = 0;
        if (MODE == kStep) {
            e.nm = reset_first ? 0 : (int)at32(rows, (uint32_t)r_moves * Bu + bl);
            if (kPhilox) e.seed = (uint32_t)at32(rows, (uint32_t)r_seeds * Bu + bl) + (reset_first ? (uint32_t)launch.seed_increment : 0u);
            if (kMt) e.mti = at32(rows, (uint32_t)(r_seeds + 1) * Bu + bl);  // position of the env's MT19937 stream
        }
        load_field_draws(chunk, 0, draws);
        settle(chunk);
        return e;
    };
    auto load_crew = [&](int chunk, CDraws& draws) {
        CrewRegs e;
        const uint32_t bl = env_index(chunk);
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            e.supp[a] = e.capa[a] = e.cum[a] = 0.0f;
            e.eqs[a] = e.act_idx[a] = 0;
            e.act_id[a] = -1;
            if (a < A && MODE == kReset) {  // wildfire.py:352-354
                e.supp[a] = dev->initial_suppressant;
                e.capa[a] = dev->initial_capacity;
                e.eqs[a] = dev->initial_equipment;
            } else if (a < A && !reset_first) {
                e.supp[a] = at32(rowsf, (uint32_t)(r_supp + a) * Bu + bl);
                e.capa[a] = at32(rowsf, (uint32_t)(r_cap + a) * Bu + bl);
                e.eqs[a] = at32(rows, (uint32_t)(r_equip + a) * Bu + bl);
                if (MODE == kStep) e.cum[a] = at32(rowsf, (uint32_t)(r_cum + a) * Bu + bl);
            }
        }
        // agents share one termination / truncation value (wildfire.py:579, utils/env.py:231-233): row 0 is read,
        // all A rows are written
        e.term = (MODE == kReset || reset_first) ? 0u : at32(rows1, u_term * Bu + bl);
        e.trunc = (MODE == kReset || reset_first) ? 0u : at32(rows1, u_trunc * Bu + bl);
        e.nm = e.nb = 0;
        e.seed = 0;
        if (MODE == kStep) {
            if (!reset_first) {
                e.nm = at32(rows, (uint32_t)r_moves * Bu + bl);
                e.nb = at32(rows, (uint32_t)r_burnouts * Bu + bl);
            }
            if (kPhilox || launch.policy || PERSIST)
                e.seed = (uint32_t)at32(rows, (uint32_t)r_seeds * Bu + bl) + (reset_first ? (uint32_t)launch.seed_increment : 0u);
        }
        if (MODE == kStep && !launch.policy) {  // last, in one block: the pairs are in flight together, behind every other load of the role
            int2 v[AMAX];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) v[a] = a < A ? reinterpret_cast<const int2*>(actions)[a * B + bl] : make_int2(0, -1);
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                e.act_idx[a] = v[a].x;
                e.act_id[a] = v[a].y;
            }
        }
        load_crew_draws(chunk, 0, draws);
        settle(chunk);
        return e;
    };

    bool frz_stamp_step = true;  // (stamp builds: which step of a multi-step launch leaves its stamps — the one before the last, an ordinary step)
    FRZ_RSTAMP(0);
    FRZ_RWALL(0);
    // One chunk per workgroup.  A launch with no more chunks than resident workgroups maps chunk = blockIdx.x; a larger one
    // hands chunks out in arrival order (a ticket): every chunk a workgroup waits on in the hand-off then belongs to a
    // workgroup that has already started, so the launch cannot stall on a workgroup that is not resident yet.  No loop
    // over chunks: the kernel body is straight-line code per role (a chunk loop made the compiler hoist every
    // loop-invariant scalar in front of it: +~700 instructions and ~60 spilled scalar registers per wavefront).
    __shared__ int s_ticket;
    __shared__ int s_stop;  // multi-step launches: the workgroup's verdict on "the batch is finished", see `stop` in the crew role
    int chunk = blockIdx.x;
    if (launch.ticketed) {
        uint32_t* const counter = reinterpret_cast<uint32_t*>(arena + launch.off_epoch) + 32;
        if (tid == 0) {
            const uint32_t t = atomicAdd(counter, 1u);
            if (t == (uint32_t)nchunks - 1u) atomicExch(counter, 0u);  // every ticket of this launch is out: ready for the next launch
            s_ticket = (int)t;
        }
        __syncthreads();
        chunk = s_ticket;
    }
    // the chunk's loads: in flight while the configuration is staged
    FDraws fdraws;
    CDraws cdraws;
    Cells cells = load_cells(chunk);
    FieldRegs fld;
    CrewRegs crw;
    if (!crew) fld = load_field(chunk, fdraws);
    else crw = load_crew(chunk, cdraws);

#ifdef FRZ_WF_STAMPS  // the same staging with a stamp after each step (diagnostic build only)
    FRZ_RSTAMP(11);
    if (threadIdx.x < sizeof(WfStaged) / 16) reinterpret_cast<uint4*>(&s_cfg)[threadIdx.x] = cfg_piece;
    FRZ_RSTAMP(12);
    __syncthreads();
    FRZ_RSTAMP(13);
    const WfHot d_launch = s_cfg;
#else
    const WfHot d_launch = stage_commit(s_cfg, cfg_piece);  // configuration block at arena offset 0; never written by a kernel
#endif
    const WfHot& d = d_launch;
    FRZ_RSTAMP(1);
    const int W = d.W;
    const int nch = d.nch;  // A + 3
    const int ch_nt = A + 1, ch_ntr = A + 2;
    const uint32_t flags_word = d.flags;

    static_assert(!PERSIST || MODE == kStep, "the multi-step launch is a loop of steps");
    static_assert(!EXTRA || PERSIST, "the rollout options belong to the multi-step launch");
    static_assert(!(PERSIST && kInjected) || EXTRA, "randomness tapes are a rollout option");
    const int n_steps = PERSIST ? launch.n_steps : 1;
    if constexpr (PERSIST) {
        if (reset_first) {  // the configured initial state (wildfire.py:347-354) + zeroed bookkeeping (utils/env.py:137-160), in registers
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                cells.f[c] = s_cfg.init_fires[c];
                if (!crew) fld.in[c] = s_cfg.init_intensity[c], fld.fu[c] = s_cfg.init_fuel[c];
            }
            if (crew) {
#pragma unroll
                for (int a = 0; a < AMAX; ++a)
                    if (a < A) crw.supp[a] = s_cfg.init_suppressant, crw.capa[a] = s_cfg.init_capacity, crw.eqs[a] = s_cfg.init_equipment;
            }
        }
        if constexpr (EXTRA) {
            if (auto_reset) {
#pragma unroll
                for (int a = 0; a < AMAX; ++a) x_return[a][slot] = 0.0;
                x_ended[slot] = 0u;
            }
        }
    }
    uint32_t epoch_now = epoch;     // the epoch the current step runs under (advances with every executed step of a multi-step launch)
    uint32_t tag = epoch + 1u;      // never 0 on a zero-filled arena
    uint32_t* cur_totals = totals + (epoch & 1u) * kTotalsStride;
    uint32_t prev[AMAX + 3];
#pragma unroll
    for (int i = 0; i < AMAX + 3; ++i) prev[i] = (epoch & 1u) ? totals0[i] : totals1[i];
    if constexpr (PERSIST) {
        if (reset_first) {  // the batch totals a reset launch would have left: every env holds the initial state's lists and is live
            mask_t lit_init = 0;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) lit_init |= (mask_t)(s_cfg.init_fires[c] > 0) << c;
            prev[0] = Bu * (uint32_t)popc(lit_init);
#pragma unroll
            for (int a = 0; a < AMAX; ++a) {
                const mask_t ok = s_cfg.init_suppressant > 0.0f ? (lit_init & (mask_t)s_cfg.range_mask[a][s_cfg.init_equipment]) : (mask_t)0;
                prev[1 + a] = Bu * (uint32_t)popc(ok);
            }
#pragma unroll
            for (int i = 0; i < AMAX + 3; ++i) prev[i] = (i == ch_nt || i == ch_ntr) ? Bu : prev[i];  // nobody is terminated, nobody truncated
        }
    }

    int64_t* const rows8 = reinterpret_cast<int64_t*>(arena + d.off_rows8);
    const uint32_t q_burnouts = 0, q_putouts = 1, q_etc = 2;

    // utils/env.py:211-213 — every per-agent step() is a no-op once ALL envs are terminated or ALL are truncated.
    auto is_frozen = [&]() {
        uint32_t nt = prev[0], ntr = prev[0];
#pragma unroll
        for (int i = 0; i < AMAX + 3; ++i) {
            nt = i == ch_nt ? prev[i] : nt;
            ntr = i == ch_ntr ? prev[i] : ntr;
        }
        return !auto_reset && (nt == 0u || ntr == 0u);  // (a rollout that resets its finished envs never comes to rest)
    };
    auto frozen_step = [&]() {  // The parallel adapter (utils/conversions.py:87-90) then adds the stale aec rewards once per agent call.
        if (!crew) {
            const int64_t b = (int64_t)chunk * kBlock + slot;
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
    };
    if (MODE == kStep && !PERSIST) {
        if (is_frozen()) {
            frozen_step();
            return;
        }
    }

    FRZ_RSTAMP(2);
    float* const obs_self = reinterpret_cast<float*>(arena + d.off_obs_self);
    float* const obs_others = reinterpret_cast<float*>(arena + d.off_obs_others);
    uint64_t* const agg = reinterpret_cast<uint64_t*>(arena + d.off_agg);
    uint64_t* const prefix = reinterpret_cast<uint64_t*>(arena + d.off_prefix);

    uint32_t* const error_word = reinterpret_cast<uint32_t*>(arena + d.off_error);
    // Multi-step launch, between two steps: the totals the step that just ended left — the last chunk's inclusive-prefix granules carry
    // its tag — replace `prev`, and the epoch advances.
    // (the granules are requested at the top of the step — request_totals — and looked at where the step first needs them: the memory
    // round trip of an agent-scope load passes behind the work in between; only a wavefront that was too early polls)
    // a wait that hit its bound (workgroups of this launch not resident: the device is shared after all) flags FRZ_ERR_SCAN_TIMEOUT; every
    // later wait of the launch then gives up at once, so that a launch that cannot work ends in seconds, not minutes
    // The five barriers of a step carry LDS data from role to role.  __syncthreads() also waits for every global store the wavefront has
    // in flight (s_waitcnt vmcnt(0)) — five drains of the store queue per step that nothing in a multi-step Philox / injected-randomness
    // launch needs: its roles hand nothing to each other through global memory (the MT19937 streams are: their kernels keep the full
    // barrier, and so do the single-step kernels).
    auto role_barrier = [&]() {
        if constexpr (PERSIST && !kMt) {
            if (!FRZ_SKIP(4)) {  // (bit 4: timing experiments — the full barrier)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                return;
            }
        }
        __syncthreads();
    };
    bool gave_up = false;
    // the chunk sums of a multi-step launch are double-buffered by the parity of the step's epoch (see `owed` below)
    auto agg_now = [&]() { return agg + (PERSIST ? (int64_t)(epoch_now & 1u) * nchunks * nch : (int64_t)0); };
    uint64_t requested[AMAX + 3];
    auto request_totals = [&]() {
        const uint64_t* const last = prefix + (int64_t)(nchunks - 1) * nch;
#pragma unroll
        for (int i = 0; i < AMAX + 3; ++i) requested[i] = frz::granule_load(last + (i < nch ? i : 0));
    };
    // The chunk's OWN sums of the step that just ended (uniform over the workgroup; set where the chunk publishes them).  Everything a step
    // asks of the batch totals is "is channel i zero?" (an agent nobody can use: wildfire.py:434-435; every env terminated / truncated:
    // utils/env.py:211-213) — and a channel that is non-zero in this chunk alone is non-zero in the batch.  A workgroup whose own sums
    // answer every question does not wait for the totals at all (`own_proves`): the launch then has ONE dependent cross-chip hand-off per
    // step (the look-back) instead of two.  What the totals also were — the inter-step barrier that kept a chunk from overwriting the sums
    // its successors still read — is kept as a debt (`owed`): the chunk sums are double-buffered by the step's parity, and a chunk looks
    // at the last chunk's granule of step t - 1 (requested behind barrier 1 of step t, long there by then) before it publishes its sums of step
    // t over those of step t - 2: the last chunk's look-back of step t - 1 read every chunk's sums of step t - 1, which each chunk published
    // after its own look-back of step t - 2.
    uint32_t own[AMAX + 3];
#pragma unroll
    for (int i = 0; i < AMAX + 3; ++i) own[i] = 0;
    bool own_proves = false, owed = false;
    uint32_t owed_tag = 0;
    uint64_t owed_granule = 0;  // the last chunk's granule as requested behind barrier 1 of the step that owes the look
    auto advance_epoch = [&]() {
        epoch_now += 1u;
        owed_tag = tag;
        tag = epoch_now + 1u;
        cur_totals = totals + (epoch_now & 1u) * kTotalsStride;
    };
    auto settle_owed = [&]() {  // before this chunk's sums of the step replace those of two steps ago
        if (!owed) return;
        owed = false;
        const uint64_t* const last = prefix + (int64_t)(nchunks - 1) * nch;
        bool timed_out = false;
        uint64_t g = owed_granule;
        for (int spin = 0; (uint32_t)(g >> 32) != owed_tag; ++spin) {  // bounded
            if (gave_up || spin >= (1 << 20)) {
                timed_out = gave_up = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            g = frz::granule_load(last);
        }
        if (timed_out && slot == 0) atomicOr(error_word, (uint32_t)FRZ_ERR_SCAN_TIMEOUT);
    };
    auto await_totals = [&]() {
        advance_epoch();
        const uint32_t ended = owed_tag;
        const uint64_t* const last = prefix + (int64_t)(nchunks - 1) * nch;
        bool timed_out = false;
        for (int spin = 0;; ++spin) {  // bounded
            bool all = true;
#pragma unroll
            for (int i = 0; i < AMAX + 3; ++i) {
                const uint64_t g = spin == 0 ? requested[i] : frz::granule_load(last + (i < nch ? i : 0));
                all = all && (uint32_t)(g >> 32) == ended;
                prev[i] = (uint32_t)g;
            }
            if (all) break;
            if (gave_up || spin >= (1 << 20)) {
                timed_out = gave_up = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (timed_out && slot == 0) atomicOr(error_word, (uint32_t)FRZ_ERR_SCAN_TIMEOUT);
    };
    // which copy of the packed lists step t writes: the caller's buffers (byte offset 0) at the last planned step only
    // ... or, with a list record (frz_rollout_spec.list_record), into that step's copy of the whole list block — offsets included —
    // which stays: every step's lists can then be read back (tests: against the oracle's)
    auto list_copy = [&](int t) -> int64_t {
        if (!PERSIST || t >= n_steps - 1) return 0;
        return (EXTRA && launch.list_record_delta != 0) ? launch.list_record_delta + (int64_t)t * launch.list_record_step : launch.scratch_delta;
    };
    auto offsets_copy = [&](int t) -> int64_t { return (EXTRA && t < n_steps - 1 && launch.list_record_delta != 0) ? list_copy(t) : (int64_t)0; };
    int executed = 0;    // steps the field role performed (it advances the epoch by as many)
    int crew_steps = 0;  // ... and the crew
    // After barrier 5 either role can place any list of its env: the chunk's offsets (s_prefix), the sums of the chunk's
    // preceding wavefronts (s_wave_scan) and the env's position inside its wavefront (x_excl) are all in LDS.
    struct Placement {
        uint64_t ex[PW];  // packed counts of the chunk's envs that precede this env
    };
    auto placement = [&]() {
        Placement p;
#pragma unroll
        for (int w = 0; w < PW; ++w) {
            uint64_t before = x_excl[w][slot];
#pragma unroll
            for (int j = 0; j < frz::kWaves; ++j) before += j < wave ? s_wave_scan[j][w] : 0ull;
            p.ex[w] = before;
        }
        return p;
    };
    auto channel_offset = [&](const Placement& p, int ch) {
        uint64_t word = p.ex[0];
#pragma unroll
        for (int w = 1; w < PW; ++w) word = (ch >> 2) == w ? p.ex[w] : word;
        return (int64_t)s_prefix[ch] + (int64_t)((word >> (16 * (ch & 3))) & 0xFFFFull);
    };
    // open action list of agent a (and, with show_bad_actions, its listed-but-not-attackable list): wildfire.py:586-717
    auto emit_agent_lists = [&](int a, mask_t lit1, mask_t ok, int64_t off_f, int64_t off_a, int64_t b, int64_t copy, int64_t ocopy) {
        const int64_t cap = B * HW;
        int64_t* const act_values = reinterpret_cast<int64_t*>(arena + d.off_act_values + copy);
        int64_t* const act_offsets = reinterpret_cast<int64_t*>(arena + d.off_act_offsets + ocopy);
        int64_t* const bad_values = reinterpret_cast<int64_t*>(arena + d.off_bad_values + copy);
        int64_t* const bad_offsets = reinterpret_cast<int64_t*>(arena + d.off_bad_offsets + ocopy);
        const bool show_bad = (flags_word & kShowBad) != 0;
        const int F = popc(lit1), fa = popc(ok);
        frz::store_through(&act_offsets[a * (B + 1) + b], off_a);  // offsets rows are whole lines per wavefront
        if (b == B - 1) act_offsets[a * (B + 1) + B] = off_a + fa;
        int64_t* av = act_values + a * cap + off_a;
        int64_t* bv = bad_values + a * cap + (off_f - off_a);  // bad = listed but not attackable
        if (show_bad) {
            frz::store_through(&bad_offsets[a * (B + 1) + b], off_f - off_a);
            if (b == B - 1) bad_offsets[a * (B + 1) + B] = (off_f - off_a) + (F - fa);
        }
        if (FRZ_SKIP(3)) return;  // (timing experiments: the list values left out — profiles/r04_experiments.txt §8)
        if (!FRZ_SKIP(4)) {
            // entry by entry instead of cell by cell: the j-th store instruction writes every env's j-th entry, and the wavefront stops at
            // the longest list among its 64 envs (2-3 entries at the bench shape's list sizes, not CMAX cells)
            uint32_t m = (uint32_t)ok;
#pragma unroll
            for (int j = 0; j < CMAX; ++j) {
                if (!__any(m != 0u)) break;
                if (m != 0u) av[j] = popc(lit1 & (mask_t)((m & (0u - m)) - 1u));
                m &= m - 1u;
            }
            if (show_bad) {
                m = (uint32_t)(mask_t)(lit1 & ~ok);
#pragma unroll
                for (int j = 0; j < CMAX; ++j) {
                    if (!__any(m != 0u)) break;
                    if (m != 0u) bv[j] = popc(lit1 & (mask_t)((m & (0u - m)) - 1u));
                    m &= m - 1u;
                }
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            const mask_t below = (mask_t)(((mask_t)1 << c) - 1);
            const int rk = popc(lit1 & below);
            if ((ok >> c) & 1)
                av[popc(ok & below)] = rk;
            else if (show_bad && ((lit1 >> c) & 1))
                bv[popc(lit1 & ~ok & below)] = rk;
        }
    };

    // The two roles run the same chunk sequence and meet at five workgroup barriers per chunk; each role's loop is its own
    // region of the program so that its registers are allocated for that role alone.
    if (!crew) {
        // ============================================================================================ FIELD ROLE
        {
            const int64_t b = (int64_t)chunk * kBlock + slot;
            const bool active = b < B;
            const uint32_t bl_launch = (uint32_t)(active ? b : B - 1), Bu_launch = Bu;
            int f[CMAX], in[CMAX], fu[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) f[c] = cells.f[c], in[c] = fld.in[c], fu[c] = fld.fu[c];
            // FRZ_RNG_PHILOX (include/frz.h): draw u = 24-bit field u % 5 of block (u / 5, step, 0, 0); field event e of cell c is draw
            // e * HW + c, agent event e of agent a is draw 3 * HW + e * A + a.  The field role draws for both roles while the crew decodes
            // the actions; agent events 1..4 are only drawn when something reads them.
            // Runtime shapes: draw u (a compile-time position of `uni`) is field event u / HW of cell u % HW for u < 3 HW, agent event
            // (u - 3 HW) / A of agent (u - 3 HW) % A behind that — places known at run time only.  Each draw is stored to ITS place in LDS
            // (the field draws in x_fd, the agent draws in copy `copy` of x_draw, where the crew reads them anyway), the event / cell / agent
            // counters running along in scalar registers; the field draws are then read back at compile-time rows.  Only this thread
            // touches its column: no barrier.
            auto scatter_draws = [&](const auto& uni, int copy, float (&field_out)[3][CMAX]) {
                constexpr int N = (int)(sizeof(uni) / sizeof(float));
                if constexpr (kGatherDraws) {  // through this env's scratch column (only this thread touches it: no barrier)
#pragma unroll
                    for (int u = 0; u < N; ++u) x_uni[u][slot] = uni[u];
#pragma unroll
                    for (int ev = 0; ev < 3; ++ev)
#pragma unroll
                        for (int cc = 0; cc < CMAX; ++cc) field_out[ev][cc] = cc < HW ? x_uni[ev * HW + cc][slot] : 1.0f;
#pragma unroll
                    for (int ev = 0; ev < 5; ++ev)
#pragma unroll
                        for (int a = 0; a < AMAX; ++a) x_draw[copy][ev * AMAX + a][slot] = a < A ? x_uni[3 * HW + ev * A + a][slot] : 1.0f;
                    return;
                }
                int e = 0, c = 0, ea = 0, aa = 0;
#pragma unroll
                for (int u = 0; u < N; ++u) {
                    if (u < 3 * HW) {
                        x_fd[e * CMAX + c][slot] = uni[u];
                        if (++c == HW) c = 0, ++e;
                    } else if (u < 3 * HW + 5 * A) {
                        x_draw[copy][ea * AMAX + aa][slot] = uni[u];
                        if (++aa == A) aa = 0, ++ea;
                    }
                }
#pragma unroll
                for (int ev = 0; ev < 5; ++ev)
#pragma unroll
                    for (int a = 0; a < AMAX; ++a)
                        if (a >= A) x_draw[copy][ev * AMAX + a][slot] = 1.0f;  // (agents the env does not have: what the staged path leaves there)
#pragma unroll
                for (int ev = 0; ev < 3; ++ev)
#pragma unroll
                    for (int cc = 0; cc < CMAX; ++cc) field_out[ev][cc] = cc < HW ? x_fd[ev * CMAX + cc][slot] : 1.0f;
            };
            // (runtime shapes: the agent draws are left in copy `copy` of x_draw, `agent_out` stays untouched)
            auto philox_draws = [&](int moves, uint32_t flags, float (&field_out)[3][CMAX], float (&agent_out)[5 * AMAX], int copy) {
                constexpr int U = 3 * CMAX + 5 * AMAX, NB = (U + 4) / 5;
                const int nb_all = EXACT ? NB : (3 * HW + 5 * A + 4) / 5, nb_event0 = EXACT ? (3 * CMAX + AMAX + 4) / 5 : (3 * HW + A + 4) / 5;
                const bool need_late = (flags & (kStochRepair | kStochDegrade | kCritical | kStochRefill | kStochSwitch)) != 0 || s_cfg.K > 1;
                const int nb_needed = need_late ? nb_all : nb_event0;
                float uni[NB * 5];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    uni[5 * j] = uni[5 * j + 1] = uni[5 * j + 2] = uni[5 * j + 3] = uni[5 * j + 4] = 0.0f;
                    if (j < nb_needed) {
                        const frz::Philox4 w = frz::philox4x32_10((uint32_t)j, (uint32_t)moves, 0u, 0u, fld.seed, 0x46525A00u);
                        uni[5 * j] = frz::philox_unit24<0>(w);
                        uni[5 * j + 1] = frz::philox_unit24<1>(w);
                        uni[5 * j + 2] = frz::philox_unit24<2>(w);
                        uni[5 * j + 3] = frz::philox_unit24<3>(w);
                        uni[5 * j + 4] = frz::philox_unit24<4>(w);
                    }
                }
                if constexpr (EXACT) {
#pragma unroll
                    for (int e = 0; e < 3; ++e)
#pragma unroll
                        for (int c = 0; c < CMAX; ++c) field_out[e][c] = uni[e * CMAX + c];
#pragma unroll
                    for (int i = 0; i < 5 * AMAX; ++i) agent_out[i] = uni[3 * CMAX + i];
                } else {
                    if constexpr (kRuntimeDraws) scatter_draws(uni, copy, field_out);
                }
            };
            // multi-step launches: the NEXT step's draws (a function of the env seed and the step number only), made while this role
            // waits for the crew's hand-off — between barriers 3 and 4 it has nothing else to do for about as long as the draws take
            float next_field[kPhilox && PERSIST ? 3 : 1][kPhilox && PERSIST ? CMAX : 1];
            // phase 6 as a function of (lit cells, copy of the packed lists): a multi-step launch that ends early writes its last lists twice
            auto emit_field = [&](mask_t lit1, int64_t copy, int64_t ocopy) {
                if (active) {
                    const Placement place = placement();
                    const int64_t off_f = channel_offset(place, 0);
                    auto ok_of = [&](int a) -> mask_t {
                        if constexpr (kOkPacked) return (mask_t)((x_ok[slot] >> (MB * a)) & (pack_t)((1u << (MB & 31)) - 1u));
                        else return (mask_t)x_okv[a][slot];
                    };
                    // single-step launch: the odd agents' lists (the crew writes the even ones, both roles finish together).  Multi-step
                    // launch with Philox draws: EVERY agent's lists — the crew's decode of the next step (which includes the wait for the totals of the step
                    // that just ended) then runs beside this phase instead of behind it.  (Round 4, from the phase stamps of
                    // tools/dbg/stamps_multistep.py: the crew was busy for 14 300 of a step's 15 800 cycles, this role for 9 800.  Worth
                    // 1-2 % only: what bounds a step of the launch is the two dependent agent-scope round trips of its hand-off — the
                    // look-back behind the rewards, then the batch totals — not either role's instruction count: profiles/r04_experiments.txt.)
#pragma unroll
                    for (int a = 0; a < AMAX; ++a)
                        if (a < A && !crew_writes(a)) emit_agent_lists(a, lit1, ok_of(a), off_f, channel_offset(place, a + 1), b, copy, ocopy);
                    int64_t* const task_values = reinterpret_cast<int64_t*>(arena + d.off_task_values + copy);
                    int64_t* const task_offsets = reinterpret_cast<int64_t*>(arena + d.off_task_offsets + ocopy);
                    int64_t* const obs_map = reinterpret_cast<int64_t*>(arena + d.off_obs_map + copy);
                    frz::store_through(&task_offsets[b], off_f);
                    if (b == B - 1) task_offsets[B] = off_f + popc(lit1);
                    // row of cell c's task inside the env's segment = number of lit cells below it
                    int64_t* const trow = task_values + off_f * 4;
                    int64_t* const omap = obs_map + off_f;
                    if (!FRZ_SKIP(4) && !FRZ_SKIP(3)) {  // entry by entry, as in emit_agent_lists: row j = the env's j-th lit cell
                        uint32_t m = (uint32_t)lit1;
#pragma unroll
                        for (int j = 0; j < CMAX; ++j) {
                            if (!__any(m != 0u)) break;
                            if (m != 0u) {
                                const int c = __ffs((int)m) - 1;
                                int fc = f[0], ic = in[0];
#pragma unroll
                                for (int k = 1; k < CMAX; ++k) fc = c == k ? f[k] : fc, ic = c == k ? in[k] : ic;
                                const int yx = s_cfg.cell_yx[c];
                                longlong2* const row = reinterpret_cast<longlong2*>(trow + j * 4);
                                row[0] = make_longlong2(yx >> 16, yx & 0xFFFF);
                                row[1] = make_longlong2(fc, ic);
                                omap[j] = j;
                            }
                            m &= m - 1u;
                        }
                    } else
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        const int rk = popc(lit1 & (mask_t)(((mask_t)1 << c) - 1));
                        if (((lit1 >> c) & 1) && !FRZ_SKIP(3)) {
                            const int yx = d.cell_yx[c];
                            longlong2* const row = reinterpret_cast<longlong2*>(trow + rk * 4);
                            row[0] = make_longlong2(yx >> 16, yx & 0xFFFF);
                            row[1] = make_longlong2(f[c], in[c]);
                            omap[rk] = rk;
                        }
                    }
                }
            };
            for (int t = 0; t < n_steps; ++t) {
                frz_stamp_step = t == (n_steps > 1 ? n_steps - 2 : 0);
                const int64_t copy = list_copy(t), ocopy = offsets_copy(t);
                // per-step opaque copies: the flag tests stay next to their uses, and in a multi-step launch so do the row addresses — hoisted
                // out of the step loop, the ~100 store addresses and ~40 row bases of a step would all be live from the top of the kernel
                // (256 VGPRs, >100 spilled scalars)
                uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)flags_word);
                asm volatile("" : "+s"(flags));
                uint32_t bl = bl_launch, Bu = Bu_launch;
                if constexpr (PERSIST) {
                    asm volatile("" : "+v"(bl));
                    asm volatile("" : "+s"(Bu));
                    asm volatile("" ::: "memory");  // and the configuration is read from LDS where a step uses it, not once above the loop
                }
                const WfHot& d = PERSIST ? static_cast<const WfHot&>(s_cfg) : d_launch;
                // ---- phase 1: the step's field draws
                float r_field[3][CMAX];
                // FRZ_RNG_MT19937: the twisted words of this step and where they go — stored behind barrier (1).  The lanes that shadow the
                // last env of a ragged batch read that env's stream too and must see it as its owner does: every load of a step is then
                // ordered before every store of the step by a workgroup barrier (they used to be ordered by timing only)
                uint32_t mt_twisted[kMt ? 3 * CMAX + 5 * AMAX : 1];
                int mt_first = 0;
                if (MODE == kStep) {
                    if constexpr (kInjected) {
                        if (EXTRA && t > 0) load_field_draws(chunk, t, fdraws);  // the tape's next pair
#pragma unroll
                        for (int e = 0; e < 3; ++e)
#pragma unroll
                            for (int c = 0; c < CMAX; ++c) r_field[e][c] = fdraws.r[e][c];
                    } else if constexpr (kPhilox) {
                        if (PERSIST && t > 0) {  // drawn while this role waited for the previous step's hand-off (below)
#pragma unroll
                            for (int e = 0; e < 3; ++e)
#pragma unroll
                                for (int c = 0; c < CMAX; ++c) r_field[e][c] = next_field[e][c];
                        } else {
                            float agent_draws[5 * AMAX];
                            philox_draws(fld.nm, flags, r_field, agent_draws, 0);
                            if constexpr (EXACT) {
#pragma unroll
                                for (int i = 0; i < 5 * AMAX; ++i) x_draw[0][i][slot] = agent_draws[i];  // (t == 0: copy 0)
                            }
                        }
                    } else if constexpr (kMt) {
                        // FRZ_RNG_MT19937: the env's own MT19937 stream (mt19937.hip: state word j of env b at [j][b], twisted
                        // lazily, one word per draw), bit-identical to the reference's per-env torch CPU generator.  The step
                        // draws U consecutive floats: generate(B, 3, (H, W)) then generate(B, 5, (A,)) (wildfire.py:409-410), i.e.
                        // field event e of cell c is draw e * HW + c, agent event e of agent a is draw 3 * HW + e * A + a.
                        // U <= 227, so no word read here is rewritten by this batch: every load is issued before the first use.
                        constexpr int U = 3 * CMAX + 5 * AMAX, kN = 624, kM = 397;
                        static_assert(U <= kN - kM, "a batch must not read a word it rewrites");
                        uint32_t* const mt = reinterpret_cast<uint32_t*>(arena + launch.off_mt_state);
                        const int i0 = fld.mti;
                        uint32_t w[U + 1], far[U];
#pragma unroll
                        for (int k = 0; k <= U; ++k) {
                            int j = i0 + k;
                            j -= j >= kN ? kN : 0;
                            w[k] = mt[(int64_t)j * B + bl];
                        }
#pragma unroll
                        for (int k = 0; k < U; ++k) {
                            int j = i0 + k + kM;
                            j -= j >= kN ? kN : 0;
                            far[k] = mt[(int64_t)j * B + bl];
                        }
                        float uni[U];
#pragma unroll
                        for (int k = 0; k < U; ++k) {
                            const uint32_t y = (w[k] & 0x80000000u) | (w[k + 1] & 0x7fffffffu);
                            uint32_t v = far[k] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                            mt_twisted[k] = v;
                            v ^= v >> 11;
                            v ^= (v << 7) & 0x9d2c5680u;
                            v ^= (v << 15) & 0xefc60000u;
                            v ^= v >> 18;
                            uni[k] = (float)(v & 0xFFFFFFu) * (1.0f / 16777216.0f);
                        }
                        mt_first = i0;
                        if constexpr (EXACT) {
#pragma unroll
                            for (int e = 0; e < 3; ++e)
#pragma unroll
                                for (int c = 0; c < CMAX; ++c) r_field[e][c] = uni[e * CMAX + c];
#pragma unroll
                            for (int i = 0; i < 5 * AMAX; ++i) x_draw[0][i][slot] = uni[3 * CMAX + i];
                        } else {  // runtime H * W and A: the step uses the first 3 H W + 5 A words, each stored to its place (scatter_draws)
                            if constexpr (kRuntimeDraws) scatter_draws(uni, 0, r_field);
                        }
                    }
                }
                if (MODE == kStep) {
                    // the draws are final BEFORE the barrier: the field role reaches it early (the crew's decode is longer) and
                    // the compiler would otherwise sink the generator arithmetic behind it, into the transitions on the critical path
#pragma unroll
                    for (int e = 0; e < 3; ++e)
#pragma unroll
                        for (int c = 0; c < CMAX; ++c) asm volatile("" : "+v"(r_field[e][c]));
                }
                FRZ_RSTAMP(3);
                role_barrier();  // (1) applied power visible
                FRZ_RSTAMP(4);
                if constexpr (PERSIST) {
                    // Is the batch finished (utils/env.py:211-213: nothing more happens in this launch)?  The crew's first wavefront has
                    // looked at the totals of the step that just ended and left its verdict in LDS: one verdict per workgroup, so that both
                    // roles leave the loop at the same barrier whatever each wavefront's own polls returned.  (An MT19937 stream has not
                    // moved yet: its words are stored below.)
                    if (s_stop) {
                        frozen_step();
                        if (t > 0) {  // the last lists went to the second copy: once more, into the caller's buffers
                            mask_t lit_last = 0;
#pragma unroll
                            for (int c = 0; c < CMAX; ++c) lit_last |= (mask_t)(f[c] > 0) << c;
                            emit_field(active ? lit_last : (mask_t)0, 0, 0);
                        }
                        break;
                    }
                }
                if constexpr (kMt) {  // the stream moves on: twisted words in place, new position
                    constexpr int U = 3 * CMAX + 5 * AMAX, kN = 624;
                    const int used = EXACT ? U : 3 * HW + 5 * A;  // words the step consumed (runtime shapes: fewer than the unrolled U)
                    uint32_t* const mt = reinterpret_cast<uint32_t*>(arena + launch.off_mt_state);
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        int j = mt_first + k;
                        j -= j >= kN ? kN : 0;
                        if (EXACT || k < used) mt[(int64_t)j * B + bl] = mt_twisted[k];
                    }
                    int j = mt_first + used;
                    j -= j >= kN ? kN : 0;
                    at32(rows, (uint32_t)(r_seeds + 1) * Bu + bl) = j;
                    if constexpr (PERSIST) fld.mti = j;
                }

                // ---- phase 2: fire increase / decrease, spread, dead test
                mask_t burned = 0, put_out = 0, lit1 = 0;
                bool dead = false;
                if (MODE == kStep) {
                    float ap[CMAX];
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) ap[c] = c < HW ? x_power[c][slot] : 0.0f;
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
                    // fire spread stencil (transitions/fire_spreads.py:44-57)
                    int fuel_sum = 0;
                    bool any_fire = false;
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
                            fuel_sum += fu[c];
                            any_fire = any_fire || f[c] > 0;
                        }
                    }
                    // termination test (wildfire.py:560-570): no lit fire left (and no fuel when fuel is tracked)
                    dead = !any_fire;
                    if (flags & kUseFuel) dead = dead && fuel_sum <= 0;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) f[c] = dead ? 0 : f[c];  // :570
                }
                // FRZ_ROLLOUT_AUTO_RESET: the env this step finished (terminated: no fire left; truncated: the horizon) starts over — this
                // role's share of reset_batches (wildfire.py:376-397): the cells, the step counter behind the draws, the seed that keys them
                bool fresh = false;
                if constexpr (EXTRA) {
                    if (auto_reset) {
                        fresh = dead || ((flags & kTruncate) && fld.nm + 1 >= d.max_steps);
#pragma unroll
                        for (int c = 0; c < CMAX; ++c) {
                            f[c] = fresh ? s_cfg.init_fires[c] : f[c];
                            in[c] = fresh ? s_cfg.init_intensity[c] : in[c];
                            fu[c] = fresh ? s_cfg.init_fuel[c] : fu[c];
                        }
                        fld.seed += fresh ? launch.seed_stride : 0u;
                        fld.nm = fresh ? -1 : fld.nm;  // + 1 below
                    }
                }
                if constexpr (EXTRA) {
                    if (launch.state_tape != nullptr) {  // frz_rollout_spec.state_tape: this step's cell rows (the agent rows: crew role)
                        int32_t* const st = launch.state_tape + (int64_t)t * (int64_t)(3 * HW + 3 * A) * B;
#pragma unroll
                        for (int c = 0; c < CMAX; ++c)
                            if (c < HW) {
                                st[(int64_t)(r_fires + c) * B + bl] = f[c];
                                st[(int64_t)(r_intensity + c) * B + bl] = in[c];
                                st[(int64_t)(r_fuel + c) * B + bl] = fu[c];
                            }
                    }
                }
#pragma unroll
                for (int c = 0; c < CMAX; ++c) lit1 |= (mask_t)(f[c] > 0) << c;
                x_lit[slot] = lit1;  // as it is also for the lanes that shadow the last env: a multi-step launch steps them like their owner
                lit1 = active ? lit1 : (mask_t)0;
                if constexpr (MB == 32) {
                    x_fate[slot] = (fate_t)burned | ((fate_t)put_out << 32);
                    x_dead[slot] = (uint8_t)dead;
                } else {
                    x_fate[slot] = (fate_t)burned | ((fate_t)put_out << MB) | ((fate_t)dead << (2 * MB));
                }
                FRZ_RSTAMP(5);
                role_barrier();  // (2) lit mask and fates visible to the crew
                FRZ_RSTAMP(6);

                // ---- phase 3: cell rows (the crew scans meanwhile)
                if ((MODE == kStep || MODE == kReset) && !(FRZ_SKIP(2) && PERSIST && t < n_steps - 1)) {  // (bit 2: timing experiments — the cell rows of the launch's last step only)
#pragma unroll
                    for (int c = 0; c < CMAX; ++c)
                        if (c < HW) {
                            at32(rows, (uint32_t)(r_fires + c) * Bu + bl) = f[c];
                            at32(rows, (uint32_t)(r_intensity + c) * Bu + bl) = in[c];
                            at32(rows, (uint32_t)(r_fuel + c) * Bu + bl) = fu[c];
                        }
                }
                if (!FRZ_SKIP(1)) {  // agent observations (wildfire.py:677-681, 704-716): the suppressants arrive from the crew
                    // Agents do not move and their base power is configuration: of an observation record only the suppressant
                    // column changes from step to step.  reset / rebuild write whole records; a step rewrites only that column
                    // (the records stay what the reference would rebuild; nothing is written twice with the same bytes).
                    float supp[AMAX];
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) supp[a] = a < A ? (fresh ? s_cfg.init_suppressant : x_supp[a][slot]) : 0.0f;
                    const int k = d.others_k, width = (A - 1) * k;  // k = 2 + power column + suppressant column
                    const bool op = (flags & kObsPower) != 0, os = (flags & kObsSupp) != 0;
                    const bool whole = MODE != kStep || FRZ_SKIP(6);
#pragma unroll
                    for (int a = 0; a < AMAX; ++a)
                        if (a < A) {
                            if (whole)
                                reinterpret_cast<float4*>(obs_self)[a * B + bl] = make_float4((float)d.ay[a], (float)d.ax[a], d.power[a], supp[a]);
                            else
                                frz::store_through(&obs_self[(a * B + bl) * 4 + 3], supp[a]);
                            float* const others = obs_others + (a * B + bl) * (int64_t)width;
                            int j = 0;  // record index: the other agents in agent order
#pragma unroll
                            for (int o = 0; o < AMAX; ++o)
                                if (o < A && o != a) {
                                    float* const rec = others + j * k;
                                    if (whole) {
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
                                    } else if (os) {
                                        frz::store_through(&rec[k - 1], supp[o]);  // the suppressant column is the last one
                                    }
                                    ++j;
                                }
                        }
                }
                FRZ_RSTAMP(7);
                role_barrier();  // (3) wavefront sums visible

                // ---- phase 4 belongs to the crew (hand-off)
                if constexpr (kPhilox && PERSIST) {
                    if (t + 1 < n_steps) {
                        float next_agent[5 * AMAX];
                        philox_draws(fld.nm + 1, flags, next_field, next_agent, (t + 1) & (kDrawCopies - 1));
                        if constexpr (EXACT) {
#pragma unroll
                            for (int i = 0; i < 5 * AMAX; ++i) x_draw[(t + 1) & (kDrawCopies - 1)][i][slot] = next_agent[i];
                        }
                    }
                }
                FRZ_RSTAMP(8);
                role_barrier();  // (4)
                role_barrier();  // (5) chunk prefix visible
                FRZ_RSTAMP(9);

                // ---- phase 6: task list (wildfire.py:586-717)
                emit_field(lit1, copy, ocopy);
                FRZ_RSTAMP(10);
                FRZ_RWALL(1);
                if constexpr (PERSIST) fld.nm += 1;
                executed = t + 1;
            }  // steps of this launch
            // The workgroup owning the last chunk finished its look-back only after every other chunk published, i.e.
            // after every workgroup of this launch read the epoch: it can advance it for the next launch.
            if (executed > 0 && chunk == nchunks - 1 && slot == 0)
                __hip_atomic_store(epoch_ptr, epoch + (uint32_t)executed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        // ============================================================================================= CREW ROLE
        {
            const int64_t b = (int64_t)chunk * kBlock + slot;
            const bool active = b < B;
            const uint32_t bl_launch = (uint32_t)(active ? b : B - 1), Bu_launch = Bu;
            uint32_t err = 0;
            float supp[AMAX], capa[AMAX], rew[AMAX];
            int eqs[AMAX], hit[AMAX];
            bool users[AMAX], refill[AMAX];
#pragma unroll
            for (int a = 0; a < AMAX; ++a) supp[a] = crw.supp[a], capa[a] = crw.capa[a], eqs[a] = crw.eqs[a];
            mask_t lit_before = 0;  // lit cells of the state the step starts from (later steps of a multi-step launch: the previous step's)
#pragma unroll
            for (int c = 0; c < CMAX; ++c) lit_before |= (mask_t)(cells.f[c] > 0) << c;
            // phase 6 as a function of (lit cells, attackable cells per agent, copy of the packed lists)
            auto emit_crew = [&](mask_t lit1, const mask_t (&ok1)[AMAX], int64_t copy, int64_t ocopy) {
                if constexpr (kFieldWritesAllLists && FRZ_WF_CREW_LISTS == 0) return;  // (the field role writes every list of such a launch, see emit_field)
                if (active) {
                    const Placement place = placement();
                    const int64_t off_f = channel_offset(place, 0);
#pragma unroll
                    for (int a = 0; a < AMAX; ++a)
                        if (a < A && crew_writes(a)) emit_agent_lists(a, lit1, ok1[a], off_f, channel_offset(place, a + 1), b, copy, ocopy);
                }
            };
            for (int t = 0; t < n_steps; ++t) {
                frz_stamp_step = t == (n_steps > 1 ? n_steps - 2 : 0);
                const int64_t copy = list_copy(t), ocopy = offsets_copy(t);
                // per-step opaque copies: the flag tests stay next to their uses, and in a multi-step launch so do the row addresses — hoisted
                // out of the step loop, the ~100 store addresses and ~40 row bases of a step would all be live from the top of the kernel
                // (256 VGPRs, >100 spilled scalars)
                uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)flags_word);
                asm volatile("" : "+s"(flags));
                uint32_t bl = bl_launch, Bu = Bu_launch;
                if constexpr (PERSIST) {
                    asm volatile("" : "+v"(bl));
                    asm volatile("" : "+s"(Bu));
                    asm volatile("" ::: "memory");  // and the configuration is read from LDS where a step uses it, not once above the loop
                }
                const WfHot& d = PERSIST ? static_cast<const WfHot&>(s_cfg) : d_launch;
                if constexpr (PERSIST) {
                    // (the totals of the step that just ended are requested further down, behind the policy's Philox block: a request issued
                    // HERE reaches memory before the last chunk's prefix has landed there, finds the old tag, and the retry costs a whole
                    // round trip more — round 4, `FRZ_WF_SKIP` delay sweep in profiles/r04_experiments.txt: 6.83 -> 6.63 us per step)
                    if (EXTRA && t > 0 && !launch.policy) {  // the action tape's next step (frz_rollout_spec.action_tape)
                        const int2* const tape = reinterpret_cast<const int2*>(actions + (int64_t)t * launch.tape_actions_step);
                        int2 v[AMAX];
#pragma unroll
                        for (int a = 0; a < AMAX; ++a) v[a] = a < A ? tape[a * B + bl] : make_int2(0, -1);
#pragma unroll
                        for (int a = 0; a < AMAX; ++a) crw.act_idx[a] = v[a].x, crw.act_id[a] = v[a].y;
                    }
                    if (EXTRA && t > 0) load_crew_draws(chunk, t, cdraws);
                }
#pragma unroll
                for (int a = 0; a < AMAX; ++a) rew[a] = 0.0f, hit[a] = -1, users[a] = false, refill[a] = false;
                const bool term0 = crw.term != 0, trunc0 = crw.trunc != 0;

                // ---- phase 1: action decode (wildfire.py:427-483) -> applied power per cell
                // The action mapping of the previous rebuild is a pure function of the state it was built from, which is the
                // state just loaded: attackable set of agent a = lit fires within its (equipment-adjusted) range, non-empty
                // only while it has suppressant (wildfire.py:604-623).
                bool stop = false;  // multi-step launch: the batch turned out to be finished
                int2 sampled[AMAX];  // the policy's choices and the decode's error bits: kept until the workgroup's verdict (behind barrier 1)
                uint32_t err1 = 0;
#pragma unroll
                for (int a = 0; a < AMAX; ++a) sampled[a] = make_int2(0, -1);
                if (MODE == kStep) {
                    const mask_t lit0 = lit_before;
                    float ap[CMAX];
                    const bool show_bad = (flags & kShowBad) != 0;
                    // stream of frz_wildfire_random_policy: agent a draws word a % 4 of block (a / 4, policy step), keyed by the env seed
                    frz::Philox4 policy_block[(AMAX + 3) / 4]{};  // agent a draws word a % 4 of block a / 4
                    if (launch.policy) {
                        // (a multi-step launch samples step t of its rollout with policy step first + t)
                        const uint64_t policy_step = (((uint64_t)launch.policy_step_hi << 32) | launch.policy_step_lo) + (uint64_t)(PERSIST ? t : 0);
#pragma unroll
                        for (int q = 0; q < (AMAX + 3) / 4; ++q)
                            if (q * 4 < A) policy_block[q] = frz::philox4x32_10((uint32_t)q, 0u, (uint32_t)policy_step, (uint32_t)(policy_step >> 32),
                                                                               launch.policy_seed_lo ^ crw.seed, launch.policy_seed_hi);
                    }
                    if constexpr (PERSIST) {
                        if (t > 0) {
#ifdef FRZ_WF_EXPERIMENT  // timing experiments: FRZ_WF_SKIP bits 8..15 = quarters of a thousand cycles to wait before the totals are requested
                            for (uint32_t w = 0; w < ((launch.skip >> 8) & 0xFFu); ++w) __builtin_amdgcn_s_sleep(4);
#endif
                            if (!(FRZ_WF_LOCAL_PROOF && own_proves)) {  // (a chunk whose own sums answer the step's questions does not ask)
                                if (!launch.policy) __builtin_amdgcn_s_sleep(8);  // (no Philox block in front of the request: wait as long as one takes)
                                request_totals();
                            }
                        }
                    }
                    // A later step of a multi-step launch decodes BEFORE the totals of the step that just ended are here (they are the launch's
                    // inter-step barrier: a memory round trip after the last chunk has published them), assuming what is true of every
                    // ordinary step — no agent is skipped, the batch is not finished — and only repeats the decode when the totals say
                    // otherwise.  Nothing of this phase leaves the registers before that test.
                    for (int attempt = 0; attempt < (PERSIST ? 2 : 1); ++attempt) {
                    const bool assume_ordinary = PERSIST && t > 0 && attempt == 0;
                    err1 = 0;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) ap[c] = 0.0f;
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) {
                        sampled[a] = make_int2(0, -1);
                        if (a < A) {
                            const mask_t ok = supp[a] > 0.0f ? (lit0 & (mask_t)s_cfg.range_mask[a][eqs[a]]) : (mask_t)0;
                            const mask_t sel = show_bad ? lit0 : ok;  // the tasks the agent's action space lists
                            int act_idx = crw.act_idx[a], act_id = crw.act_id[a];
                            if (launch.policy) {
                                // uniform random policy over OneOf([task] * n + [noop]) (spaces/actions.py:23-41,
                                // baselines/random.py:20), the stream of frz_wildfire_random_policy: member j ~ U{0..n};
                                // j < n -> [j, 0] (fight task j), j == n -> [n, -1] (noop / refill)
                                const int n = popc(sel);
                                const int j = (int)(((uint64_t)policy_block[a >> 2].w[a & 3] * (uint64_t)(n + 1)) >> 32);
                                act_idx = j < n ? j : n;
                                act_id = j < n ? 0 : -1;
                                sampled[a] = make_int2(act_idx, act_id);
                                if constexpr (!PERSIST) frz::store_through(&reinterpret_cast<int2*>(launch.actions_out)[a * B + bl], sampled[a]);
                            }
                            refill[a] = act_id == -1;
                            // quirk wildfire.py:434-435: an agent with no attackable task in ANY env of the batch is skipped
                            const bool skipped = !assume_ordinary && prev[1 + a] == 0u;
                            const bool fight = !refill[a] && !skipped;
                            const bool valid = act_idx >= 0 && act_idx < popc(sel);
                            int target = 0, seen = 0;
#pragma unroll
                            for (int c = 0; c < CMAX; ++c) {
                                const int bit = (int)((sel >> c) & 1);
                                target = (bit && seen == act_idx) ? c : target;
                                seen += bit;
                            }
                            const bool attackable = ((ok >> target) & 1) != 0;
                            const bool good = fight && valid && (!show_bad || attackable);
                            if (fight && !valid && active) err1 |= FRZ_ERR_BAD_ACTION_INDEX;
                            const float power = d.power[a] + s_cfg.eq[eqs[a]][1];
#pragma unroll
                            for (int c = 0; c < CMAX; ++c) ap[c] = ap[c] + ((good && target == c) ? power : 0.0f);  // agent order
                            users[a] = good;
                            hit[a] = good ? target : -1;
                            rew[a] = (fight && !good) ? d.bad_attack_penalty : 0.0f;  // assignment, :477
                        }
                    }
                    if constexpr (PERSIST) {
                        if (!assume_ordinary) break;
                        if (FRZ_WF_LOCAL_PROOF && own_proves) {  // this chunk's own sums say "ordinary": nothing to wait for here
                            advance_epoch();
#pragma unroll
                            for (int i = 0; i < AMAX + 3; ++i) prev[i] = own[i];
                            owed = true;
                            break;
                        }
                        await_totals();
                        bool someone_skipped = false;
#pragma unroll
                        for (int a = 0; a < AMAX; ++a) someone_skipped = someone_skipped || (a < A && prev[1 + a] == 0u);
                        if (!someone_skipped) break;
                    }
                    }  // attempts
#pragma unroll
                    for (int c = 0; c < CMAX; ++c)
                        if (c < HW) x_power[c][slot] = ap[c];
                    if constexpr (PERSIST) {
                        if (threadIdx.x == 0) s_stop = is_frozen() ? 1 : 0;  // the workgroup's verdict (see the field role, behind barrier 1)
                    }
                }
                FRZ_RSTAMP(3);
                role_barrier();  // (1) applied power visible to the field role
                FRZ_RSTAMP(4);
                if constexpr (PERSIST) {
                    // (the debt of a step that did not wait for the totals: asked for HERE, not at the top of the step — a request issued
                    // there can reach memory a few hundred clocks before the last chunk's granule does, and the second look would then cost a
                    // whole round trip right where this chunk publishes its sums, which every later chunk's look-back waits for)
                    if (owed) owed_granule = frz::granule_load(prefix + (int64_t)(nchunks - 1) * nch);
                    stop = s_stop != 0;
                    if (stop) {  // utils/env.py:211-213: nothing more happens in this launch
                        if (t > 0) {  // the last lists went to the second copy: once more, into the caller's buffers
                            mask_t ok_last[AMAX];
#pragma unroll
                            for (int a = 0; a < AMAX; ++a)
                                ok_last[a] = (a < A && supp[a] > 0.0f) ? (lit_before & (mask_t)s_cfg.range_mask[a][eqs[a]]) : (mask_t)0;
                            emit_crew(lit_before, ok_last, 0, 0);
                        }
                        break;
                    }
                    err |= err1;
                    if (launch.policy) {  // nothing of the step had left the registers before the verdict
                        int2* const out = reinterpret_cast<int2*>(launch.actions_out + (EXTRA ? (int64_t)t * launch.actions_out_step : (int64_t)0));
#pragma unroll
                        for (int a = 0; a < AMAX; ++a)
                            if (a < A) frz::store_through(&out[a * B + bl], sampled[a]);
                    }
                    if (t == 0 && reset_first) {  // the opening reset's own stores: the fresh seed, the stale-reward mark
                        at32(rows, (uint32_t)r_seeds * Bu + bl) = (int32_t)crw.seed;
                        at32(rows1, u_frozen * Bu + bl) = (uint8_t)0;
                    }
                } else {
                    err |= err1;
                }

                // ---- phase 2: agent draws, agent transitions, agent rows, agent observations
                if (MODE == kStep) {
                    float r_agent[5][AMAX];
                    if constexpr (kInjected) {
#pragma unroll
                        for (int e = 0; e < 5; ++e)
#pragma unroll
                            for (int a = 0; a < AMAX; ++a) r_agent[e][a] = cdraws.r[e][a];
                    } else if constexpr (kPhilox || kMt) {
#pragma unroll
                        for (int e = 0; e < 5; ++e)
#pragma unroll
                            for (int a = 0; a < AMAX; ++a) r_agent[e][a] = x_draw[t & (kDrawCopies - 1)][e * AMAX + a][slot];  // drawn by the field role
                    }
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
                if (MODE == kReset) {  // the configured agent state + zeroed bookkeeping (utils/env.py:137-160)
#pragma unroll
                    for (int a = 0; a < AMAX; ++a)
                        if (a < A) {
                            at32(rowsf, (uint32_t)(r_supp + a) * Bu + bl) = supp[a];
                            at32(rowsf, (uint32_t)(r_cap + a) * Bu + bl) = capa[a];
                            at32(rows, (uint32_t)(r_equip + a) * Bu + bl) = eqs[a];
                            at32(rowsf, (uint32_t)(r_rewards + a) * Bu + bl) = 0.0f;
                            at32(rowsf, (uint32_t)(r_cum + a) * Bu + bl) = 0.0f;
                            at32(rows1, (u_term + (uint32_t)a) * Bu + bl) = (uint8_t)0;
                            at32(rows1, (u_trunc + (uint32_t)a) * Bu + bl) = (uint8_t)0;
                        }
                    at32(rows, (uint32_t)r_moves * Bu + bl) = 0;
                    at32(rows, (uint32_t)r_burnouts * Bu + bl) = 0;
                    if (launch.seed_increment != 0 && active)  // fresh seeds per episode, modulo 2^32
                        at32(rows, (uint32_t)r_seeds * Bu + bl) = (int32_t)((uint32_t)at32(rows, (uint32_t)r_seeds * Bu + bl) + (uint32_t)launch.seed_increment);
                    at32(rows8, q_burnouts * Bu + bl) = 0;
                    at32(rows8, q_putouts * Bu + bl) = 0;
                    at32(rows1, u_frozen * Bu + bl) = (uint8_t)0;
                }
#pragma unroll
                for (int a = 0; a < AMAX; ++a)
                    if (a < A) x_supp[a][slot] = supp[a];  // the field role stores the agent observations
                FRZ_RSTAMP(5);
                role_barrier();  // (2) lit mask and fates visible
                FRZ_RSTAMP(6);

                // ---- phase 3: open-task sets, per-env counts, wavefront scan
                const mask_t lit_all = x_lit[slot];
                const mask_t lit1 = active ? lit_all : (mask_t)0;
                const fate_t fate = x_fate[slot];
                constexpr fate_t kFateMask = MB == 32 ? (fate_t)0xFFFFFFFFull : (fate_t)((1u << (MB & 31)) - 1u);
                const mask_t burned = (mask_t)(fate & kFateMask), put_out = (mask_t)((fate >> MB) & kFateMask);
                bool dead;
                if constexpr (MB == 32) dead = x_dead[slot] != 0;
                else dead = ((fate >> ((2 * MB) & 63)) & 1u) != 0;
                bool term = term0, trunc = trunc0;
                bool fresh = false;  // FRZ_ROLLOUT_AUTO_RESET: this step finished the env and it starts over (this role's share below)
                if (MODE == kStep) {
                    const int nm = crw.nm + 1;
                    trunc = (flags & kTruncate) ? nm >= d.max_steps : trunc0;
                    term = term0 || dead;
                    if constexpr (EXTRA) {
                        fresh = auto_reset && (term || trunc);
                        if (fresh) {  // wildfire.py:352-354 on this env; the rows were written with the step's values above
#pragma unroll
                            for (int a = 0; a < AMAX; ++a)
                                if (a < A) {
                                    supp[a] = s_cfg.init_suppressant, capa[a] = s_cfg.init_capacity, eqs[a] = s_cfg.init_equipment;
                                    at32(rowsf, (uint32_t)(r_supp + a) * Bu + bl) = supp[a];
                                    at32(rowsf, (uint32_t)(r_cap + a) * Bu + bl) = capa[a];
                                    at32(rows, (uint32_t)(r_equip + a) * Bu + bl) = eqs[a];
                                }
                        }
                    }
                    at32(rows, (uint32_t)r_moves * Bu + bl) = fresh ? 0 : nm;
                    if constexpr (EXTRA) {  // what a learner keeps of the step: the agents' state as the step (and the restart) left it
                        if (launch.supp_tape != nullptr) {  // FRZ_ROLLOUT_OBS_COMPACT: the one column of the self / others records that moves
                            float* const tape = launch.supp_tape + (int64_t)t * A * B;
#pragma unroll
                            for (int a = 0; a < AMAX; ++a)
                                if (a < A) tape[(int64_t)a * B + bl] = supp[a];
                        }
                        if (launch.state_tape != nullptr) {
                            int32_t* const st = launch.state_tape + (int64_t)t * (int64_t)(3 * HW + 3 * A) * B;
#pragma unroll
                            for (int a = 0; a < AMAX; ++a)
                                if (a < A) {
                                    reinterpret_cast<float*>(st)[(int64_t)(r_supp + a) * B + bl] = supp[a];
                                    reinterpret_cast<float*>(st)[(int64_t)(r_cap + a) * B + bl] = capa[a];
                                    st[(int64_t)(r_equip + a) * B + bl] = eqs[a];
                                }
                        }
                    }
                }
                mask_t ok1[AMAX];
                uint64_t packed[PW], incl[PW], base[PW];
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
#pragma unroll
                for (int w = 0; w < PW; ++w) incl[w] = frz::wave_inclusive_scan(packed[w]);
                const uint32_t live_nt = (uint32_t)__popcll(__ballot(active && !term));
                const uint32_t live_ntr = (uint32_t)__popcll(__ballot(active && !trunc));
                if (lane == 63) {
#pragma unroll
                    for (int w = 0; w < PW; ++w) s_wave_scan[wave][w] = incl[w];
                    s_wave_live[wave][0] = live_nt;
                    s_wave_live[wave][1] = live_ntr;
                }
#pragma unroll
                for (int w = 0; w < PW; ++w) x_excl[w][slot] = incl[w] - packed[w];
                if constexpr (kOkPacked) {
                    pack_t oks = 0;
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) oks |= (pack_t)ok1[a] << ((MB * a) & 63);
                    x_ok[slot] = oks;
                } else {
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) x_okv[a][slot] = (uint32_t)ok1[a];
                }
                FRZ_RSTAMP(7);
                role_barrier();  // (3) wavefront sums visible

                // ---- phase 4: chunk sums published; rewards / bookkeeping hide the hand-off; look-back
                const int round_first = chunk & ~(kRound - 1);  // chunks are handed off in windows of kRound
                uint64_t block_total[PW];
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
                uint32_t my_total = 0;  // this chunk's sum of channel `slot` (slot < nch)
                if (slot < nch) {
                    if (slot <= A) {
                        uint64_t word = block_total[0];
#pragma unroll
                        for (int w = 1; w < PW; ++w) word = (slot >> 2) == w ? block_total[w] : word;
                        my_total = (uint32_t)((word >> (16 * (slot & 3))) & 0xFFFFull);
                    } else {
                        const int which = slot - ch_nt;
#pragma unroll
                        for (int j = 0; j < frz::kWaves; ++j) my_total += s_wave_live[j][which];
                    }
                    if constexpr (PERSIST) settle_owed();
                    frz::granule_store(agg_now() + (int64_t)chunk * nch + slot, tag, my_total);
                }
                FRZ_RSTAMP(13);
                if constexpr (PERSIST) {  // what this chunk alone can say about the batch totals the next step asks for
                    bool proves = true;
#pragma unroll
                    for (int i = 0; i <= AMAX; ++i) {
                        uint64_t word = block_total[0];
#pragma unroll
                        for (int w = 1; w < PW; ++w) word = (i >> 2) == w ? block_total[w] : word;
                        own[i] = (uint32_t)((word >> (16 * (i & 3))) & 0xFFFFull);
                        proves = proves && (i == 0 || i > A || own[i] != 0u);
                    }
                    uint32_t live_own[2] = {0u, 0u};
#pragma unroll
                    for (int j = 0; j < frz::kWaves; ++j) live_own[0] += s_wave_live[j][0], live_own[1] += s_wave_live[j][1];
#pragma unroll
                    for (int i = 0; i < AMAX + 3; ++i) own[i] = i == ch_nt ? live_own[0] : (i == ch_ntr ? live_own[1] : own[i]);
                    own_proves = proves && (auto_reset || (live_own[0] != 0u && live_own[1] != 0u));
                }
                if (MODE == kStep) {
                    // rewards and termination (wildfire.py:534-582)
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
                    if (newly && d.termination_kappa != 0.0f) log_burnouts = (float)log((double)crw.nb + 1.0);
                    const float penalty = __fmul_rn(d.termination_kappa, log_burnouts);
                    float term_reward = __fsub_rn(d.termination_reward, penalty);
                    term_reward = term_reward < 0.0f ? 0.0f : term_reward;
                    const int n_burn = popc(burned), n_put = popc(put_out);
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
                            // (an env that starts over shows what reset_batches leaves — utils/env.py:176-188: zeros; the step's own values
                            // go to the reward / done tapes)
                            at32(rowsf, (uint32_t)(r_rewards + a) * Bu + bl) = fresh ? 0.0f : rew[a];
                            at32(rows1, (u_term + (uint32_t)a) * Bu + bl) = (uint8_t)(term && !fresh);
                            if (write_trunc) at32(rows1, (u_trunc + (uint32_t)a) * Bu + bl) = (uint8_t)(trunc && !fresh);
                            if constexpr (EXTRA) {
                                if (launch.reward_tape != nullptr) launch.reward_tape[((int64_t)t * A + a) * B + bl] = rew[a];
                            }
                            if (track) {
                                const float total = __fadd_rn(crw.cum[a], rew[a]);
                                if (fresh && active) x_return[a][slot] += (double)total;  // the episode's return, harvested before reset_batches zeroes it
                                at32(rowsf, (uint32_t)(r_cum + a) * Bu + bl) = fresh ? 0.0f : total;
                                if constexpr (PERSIST) crw.cum[a] = fresh ? 0.0f : total;
                            }
                        }
                    }
                    if constexpr (EXTRA) {
                        if (launch.done_tape != nullptr) {
                            launch.done_tape[((int64_t)t * 2 + 0) * B + bl] = (uint8_t)term;
                            launch.done_tape[((int64_t)t * 2 + 1) * B + bl] = (uint8_t)trunc;
                        }
                        if (fresh) {  // utils/env.py:162-189: bookkeeping of the env zeroed; its seed moves on
                            if (active) x_ended[slot] += 1u;
                            crw.seed += launch.seed_stride;
                            at32(rows, (uint32_t)r_seeds * Bu + bl) = (int32_t)crw.seed;
                        }
                    }
                    at32(rows, (uint32_t)r_burnouts * Bu + bl) = fresh ? 0 : crw.nb + n_burn;
                    if constexpr (PERSIST) crw.nb = fresh ? 0 : crw.nb + n_burn;
                    at32(rows8, q_burnouts * Bu + bl) = n_burn;
                    at32(rows8, q_putouts * Bu + bl) = n_put;
                }
                FRZ_RSTAMP(14);
                if (active) {
#pragma unroll
                    for (int a = 0; a < AMAX; ++a)
                        if (a < A) at32(rows, (uint32_t)(r_atc + a) * Bu + bl) = popc(ok1[a]);
                    at32(rows8, q_etc * Bu + bl) = F;
                }
                FRZ_RSTAMP(15);
                // inter-workgroup exclusive prefix (single pass), as in wildfire.hip: crew thread t sums channel (t % NCHP) over
                // predecessors t / NCHP, t / NCHP + PP, ...; the window's loads are unconditional so they are in flight together
                bool timed_out = false;
                uint32_t acc = 0;
                {
                    const uint64_t* const agg_step = agg_now();
                    const int ch = slot & (NCHP - 1), pslot = slot / NCHP;
                    constexpr int PP = kBlock / NCHP, UNR = 8;
                    for (int first = round_first; first < (FRZ_SKIP(7) ? round_first : chunk); first += PP * UNR) {  // (bit 7: timing experiments)
                        uint32_t part = 0;
                        for (int spin = 0;; ++spin) {  // bounded: every granule of the window must carry this launch's tag
                            bool all = true;
                            part = 0;
#pragma unroll
                            for (int u = 0; u < UNR; ++u) {
                                const int pred = first + u * PP + pslot;
                                const bool valid = pred < chunk && ch < nch;
                                const uint64_t g = frz::granule_load(agg_step + (valid ? (int64_t)pred * nch + ch : (int64_t)0));
                                all = all && (!valid || (uint32_t)(g >> 32) == tag);
                                part += valid ? (uint32_t)g : 0u;
                            }
                            if (all) break;
                            if (gave_up || spin >= (1 << 22)) {
                                timed_out = gave_up = true;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(2);
                        }
                        acc += part;
                    }
                    if (round_first > 0 && slot < nch) acc += frz::granule_wait(prefix + (int64_t)(round_first - 1) * nch + slot, tag, &timed_out);
#pragma unroll
                    for (int dd = NCHP; dd < 64; dd <<= 1) acc += __shfl_xor(acc, dd, 64);
                    if (lane < NCHP) s_reduce[wave][lane] = acc;
                }
                FRZ_RSTAMP(8);
                role_barrier();  // (4) look-back partial sums visible

                // ---- phase 5: chunk prefix
                if (slot < nch) {
                    uint32_t s = 0;
#pragma unroll
                    for (int j = 0; j < frz::kWaves; ++j) s += s_reduce[j][slot];
                    s_prefix[slot] = s;
                    const bool round_last = (chunk & (kRound - 1)) == kRound - 1 || chunk == nchunks - 1;
                    if (round_last) {
                        frz::granule_store(prefix + (int64_t)chunk * nch + slot, tag, s + my_total);
                        if (chunk == nchunks - 1) cur_totals[slot] = s + my_total;  // batch totals, read by the next launch
                    }
                }
                role_barrier();  // (5) chunk prefix visible
                FRZ_RSTAMP(9);
                if (timed_out) err |= FRZ_ERR_SCAN_TIMEOUT;

                // ---- phase 6: the even agents' action lists (the field role writes the odd ones and the task list)
                emit_crew(lit1, ok1, copy, ocopy);
                FRZ_RSTAMP(10);
                FRZ_RWALL(1);
                if constexpr (PERSIST) {  // what the next step of this launch starts from
                    lit_before = lit_all;
                    crw.term = (term && !fresh) ? 1u : 0u, crw.trunc = (trunc && !fresh) ? 1u : 0u;
                    crw.nm = fresh ? 0 : crw.nm + 1;
                    crew_steps = t + 1;
                }
            }  // steps of this launch
            if (err) atomicOr(error_word, err);
        }
    }
    if constexpr (PERSIST) {
        // frz_wildfire_rollout_random_policy_metrics: the episode metrics (frz_wildfire_episode_metrics: sum of every agent's cumulative
        // reward, of num_moves, number of finished envs) from the values the crew role still holds, in the standalone kernel's
        // summation order (one env per thread, lane tree, wavefronts 0..3, then the chunks' partial rows by the last workgroup to arrive:
        // lane tree, wavefronts 0..3) — the same float64 results bit for bit, one launch and its gap less per episode.
        if (launch.metrics_out != nullptr) {
            __shared__ double s_metric[frz::kWaves][AMAX + 2];
            __shared__ int s_last_workgroup;
            const int nrow = A + 2;
            double* const partial = reinterpret_cast<double*>(arena + d_launch.off_metrics);
            auto reduce_rows = [&](const double (&mine)[AMAX + 2]) {
#pragma unroll
                for (int i = 0; i < AMAX + 2; ++i) {
                    double v = mine[i];
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);  // fixed tree: deterministic
                    if (lane == 0) s_metric[wave][i] = v;
                }
            };
            if (crew) {
                const bool active = (int64_t)chunk * kBlock + slot < B;
                double mine[AMAX + 2];
#pragma unroll
                for (int i = 0; i < AMAX + 2; ++i) mine[i] = 0.0;
                if (EXTRA && active && auto_reset) {  // returns of the episodes that ended in this launch, env-steps executed, episodes ended
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) mine[a] = a < A ? x_return[a][slot] : 0.0;
#pragma unroll
                    for (int i = 0; i < AMAX + 2; ++i) {
                        mine[i] = i == A ? (double)crew_steps : mine[i];
                        mine[i] = i == A + 1 ? (double)x_ended[slot] : mine[i];
                    }
                } else if (active) {
#pragma unroll
                    for (int a = 0; a < AMAX; ++a) mine[a] = a < A ? (double)crw.cum[a] : 0.0;
#pragma unroll
                    for (int i = 0; i < AMAX + 2; ++i) {
                        mine[i] = i == A ? (double)crw.nm : mine[i];
                        mine[i] = i == A + 1 ? ((crw.term != 0u || crw.trunc != 0u) ? 1.0 : 0.0) : mine[i];
                    }
                }
                reduce_rows(mine);
            }
            __syncthreads();
            if (crew && slot < nrow) {
                double v = 0.0;
#pragma unroll
                for (int w = 0; w < frz::kWaves; ++w) v += s_metric[w][slot];
                __hip_atomic_store(&partial[(int64_t)chunk * nrow + slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial row has left before the ticket is taken
            __syncthreads();
            if (crew && slot == 0) {
                uint32_t* const counter = reinterpret_cast<uint32_t*>(arena + launch.off_epoch) + 48;
                const uint32_t ticket = atomicAdd(counter, 1u);
                s_last_workgroup = ticket == (uint32_t)nchunks - 1u;
                if (s_last_workgroup) atomicExch(counter, 0u);
            }
            __syncthreads();
            if (s_last_workgroup) {
                if (crew) {
                    double mine[AMAX + 2];
#pragma unroll
                    for (int i = 0; i < AMAX + 2; ++i)
                        mine[i] = (i < nrow && slot < nchunks)
                                      ? __hip_atomic_load(&partial[(int64_t)slot * nrow + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : 0.0;
                    reduce_rows(mine);
                }
                __syncthreads();
                if (crew && slot < nrow) {
                    double v = 0.0;
#pragma unroll
                    for (int w = 0; w < frz::kWaves; ++w) v += s_metric[w][slot];
                    launch.metrics_out[slot] += v;
                }
            }
        }
    }
}

template <int CMAX, int AMAX, bool EXACT>
void launch_roles_variant(const WfArgs& a, const WfDev* dev, int grid, int rng, int mode, hipStream_t stream) {
    const WfLaunch batch = make_launch(a);
    auto go = [&](auto kernel) { launch_step_kernel(a, kernel, grid, kRoleBlock, stream, a.arena, dev, a.actions, a.field_rand, a.agent_rand, batch); };
    if (mode == kReset) return go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_INJECTED, kReset>);
    if (mode == kRebuild) return go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_INJECTED, kRebuild>);
    if constexpr (!EXACT && !(CMAX > 16 && AMAX > 4)) {  // (<24, 8>: single steps only — its multi-step instantiation would not fit the LDS)
        if (a.n_steps > 1) {  // runtime shapes: the multi-step launch draws in the kernel (Philox, or the env's MT19937 stream) or reads the tapes
            const bool extra = !a.policy || a.tape_actions_step != 0 || a.list_record_delta != 0 || a.reward_tape || a.done_tape || a.actions_out_step != 0 ||
                               (a.rollout_flags & FRZ_ROLLOUT_AUTO_RESET) != 0 || a.supp_tape || a.state_tape;
            if (rng == FRZ_RNG_PHILOX)
                return extra ? go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_PHILOX, kStep, true, true>) : go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_PHILOX, kStep, true, false>);
            if (rng == FRZ_RNG_MT19937)
                return extra ? go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_MT19937, kStep, true, true>) : go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_MT19937, kStep, true, false>);
            return go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_INJECTED, kStep, true, true>);
        }
    }
    if constexpr (EXACT) {
        if (a.n_steps > 1) {  // one launch for the whole rollout (the caller has checked residency and the second list copy)
            // the plain rollout (policy in-kernel, opening reset, metrics) or the one with every option of a frz_rollout_spec
            const bool extra = !a.policy || a.tape_actions_step != 0 || a.list_record_delta != 0 || a.reward_tape || a.done_tape || a.actions_out_step != 0 ||
                               (a.rollout_flags & FRZ_ROLLOUT_AUTO_RESET) != 0 || a.supp_tape || a.state_tape;
            if (rng == FRZ_RNG_PHILOX)
                return extra ? go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_PHILOX, kStep, true, true>) : go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_PHILOX, kStep, true, false>);
            if (rng == FRZ_RNG_MT19937)
                return extra ? go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_MT19937, kStep, true, true>) : go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_MT19937, kStep, true, false>);
            return go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_INJECTED, kStep, true, true>);  // action + randomness tapes: golden trajectories
        }
        if (rng == FRZ_RNG_PHILOX) return go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_PHILOX, kStep>);
        if (rng == FRZ_RNG_MT19937) return go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_MT19937, kStep>);
    }
    if constexpr (!EXACT && CMAX <= 16) {  // (round 4: a single step of a runtime shape draws in the kernel as well — the staging launch cost more than the step; not <24, 8>: LDS)
        if (rng == FRZ_RNG_PHILOX) return go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_PHILOX, kStep>);
    }
    if (rng == FRZ_RNG_INJECTED) go(wf_roles_kernel<CMAX, AMAX, EXACT, FRZ_RNG_INJECTED, kStep>);
}

}  // namespace

// The instantiations are dealt to FRZ_WF_ROLES_GROUPS translation units (wildfire_roles_g<k>.hip defines FRZ_WF_ROLES_GROUP = k and
// includes this file): variant i is compiled in unit i % FRZ_WF_ROLES_GROUPS, so that `make -j` builds them side by side (one unit
// with every instantiation took two minutes).  variant = index into FRZ_WF_VARIANT_LIST (wildfire_common.h); only entries of <= 16
// cells and <= 4 agents have a field/crew kernel.  The caller has already staged the Philox draws for the runtime-shape variants
// (rng arrives as FRZ_RNG_INJECTED)
#define FRZ_WF_CONCAT2(a, b) a##b
#define FRZ_WF_CONCAT(a, b) FRZ_WF_CONCAT2(a, b)
int FRZ_WF_CONCAT(launch_roles_group_, FRZ_WF_ROLES_GROUP)(const WfArgs& args, int variant, int grid, int rng, int mode, hipStream_t stream) {
    const WfDev* dev = reinterpret_cast<const WfDev*>(args.arena);
    switch (variant) {
#define FRZ_X(i, c, a, e)                                                                                                          \
    case i:                                                                                                                        \
        if constexpr ((i) % FRZ_WF_ROLES_GROUPS == FRZ_WF_ROLES_GROUP && c <= 24)                                                   \
            launch_roles_variant<c, a, e>(args, dev, grid, rng, mode, stream);                                                     \
        else return FRZ_E_INVALID;                                                                                                 \
        break;
        FRZ_WF_VARIANT_LIST(FRZ_X)
#undef FRZ_X
        default: return FRZ_E_INVALID;
    }
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}


// resident workgroups per CU: the MINIMUM over every multi-step instantiation frz_wildfire_rollout can launch for the shape (three RNG
// modes, with and without the EXTRA options — the EXTRA kernels use 7-9 KB more LDS and some spill: ADVICE r3), so that the residency
// guard of frz_wildfire_set_exclusive_device holds for whichever of them a spec picks
template <int C, int A, bool E>
int persist_occupancy_min() {
    if constexpr (!E && C > 16 && A > 4) return 0;  // (no multi-step instantiation: see launch_roles_variant)
    int least = 1 << 30;
    auto probe = [&least](auto kernel) {
        int blocks = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, kRoleBlock, 0) != hipSuccess) {
            (void)hipGetLastError();
            blocks = 0;
        }
        if (blocks < least) least = blocks;
    };
    if constexpr (E || !(C > 16 && A > 4)) {
        probe(wf_roles_kernel<C, A, E, FRZ_RNG_PHILOX, kStep, true, false>);
        probe(wf_roles_kernel<C, A, E, FRZ_RNG_PHILOX, kStep, true, true>);
        probe(wf_roles_kernel<C, A, E, FRZ_RNG_MT19937, kStep, true, false>);
        probe(wf_roles_kernel<C, A, E, FRZ_RNG_MT19937, kStep, true, true>);
        probe(wf_roles_kernel<C, A, E, FRZ_RNG_INJECTED, kStep, true, true>);
    }
    return least;
}

// resident workgroups per CU of the variant's multi-step instantiations, 0 if the variant is not in this unit
int FRZ_WF_CONCAT(roles_persist_occupancy_group_, FRZ_WF_ROLES_GROUP)(int variant) {
    int blocks = 0;
    switch (variant) {
#define FRZ_X(i, c, a, e)                                                                                                                  \
    case i:                                                                                                                                \
        if constexpr ((i) % FRZ_WF_ROLES_GROUPS == FRZ_WF_ROLES_GROUP && c <= 24) {                                                         \
            blocks = persist_occupancy_min<c, a, e>();                                                                                      \
        }                                                                                                                                  \
        break;
        FRZ_WF_VARIANT_LIST(FRZ_X)
#undef FRZ_X
        default: break;
    }
    return blocks;
}

}  // namespace frz_wf
