/*
 * frz.h — C-ABI of the MI355X-native batched environment step path.
 *
 * Drop-in boundary for free-range-zoo's batched parallel-env step loop
 * (reference: free_range_zoo/utils/conversions.py:59-99 -> utils/env.py:203-242 ->
 * envs/<domain>/env/<domain>.py step_environment/update_actions/update_observations).
 *
 * Two libraries export (a subset of) these symbols:
 *   libfrz_hip.so    — the product: hand-written HIP kernels for gfx950 (free-range-zoo_amd/csrc)
 *   libfrz_oracle.so — test infrastructure only: scalar CPU restatement (oracle/), prefix frz_oracle_
 *
 * Conventions
 *   - plain C types only; every pointer inside a *_bufs struct is a DEVICE pointer for libfrz_hip
 *     (HOST pointer for the oracle), owned by the caller, never retained after destroy().
 *   - HBM layout is struct-of-arrays with the environment index innermost ("[k][B]"), so that the 64
 *     lanes of a wavefront (one environment per lane) touch 256 contiguous bytes per field.
 *   - jagged outputs are (values, offsets) pairs in the reference's order: env-major, row-major
 *     inside an env; offsets are int64[B+1] (torch.nested jagged convention).
 *   - all entry points are stream-ordered, never synchronise, never allocate; return 0 or FRZ_E_*.
 */
#ifndef FRZ_H_
#define FRZ_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRZ_ABI_VERSION 5

#define FRZ_MAX_AGENTS 16
#define FRZ_MAX_CELLS 1024 /* 32 x 32; grids above 16 cells run one env per wavefront with the cells across its lanes */
#define FRZ_MAX_EQUIPMENT_STATES 8
#define FRZ_MAX_CAPACITIES 8
#define FRZ_MAX_NODES 16
#define FRZ_MAX_NETWORK_STATES 16

/* error codes */
#define FRZ_OK 0
#define FRZ_E_INVALID (-1)     /* bad argument / unsupported size */
#define FRZ_E_UNBOUND (-2)     /* buffers not bound */
#define FRZ_E_LAUNCH (-3)      /* HIP launch failure (hipGetLastError != success) */
#define FRZ_E_NODEVICE (-4)

/* randomness source of a step */
#define FRZ_RNG_INJECTED 0 /* caller passes the tensors RandomGenerator.generate() would return */
#define FRZ_RNG_PHILOX 1   /* counter-based Philox4x32-10 keyed by the env seed; stream definition at frz_wildfire_step */
#define FRZ_RNG_MT19937 2  /* per-env MT19937 streams, seed-identical to the reference's CPU generator */

/* bits of the device-side error word (bufs.error_flags[0]); sticky until cleared by the caller */
#define FRZ_ERR_BAD_ACTION_INDEX 1u /* task index outside the agent's action mapping */
#define FRZ_ERR_SCAN_TIMEOUT 2u     /* inter-workgroup prefix hand-off spin bound hit (never expected) */
#define FRZ_ERR_INVALID_TARGET 4u   /* cybersecurity: attack/move target outside [0, num_nodes) */
#define FRZ_ERR_ABSENT_ACTION 8u    /* cybersecurity: non-noop action by an absent agent (show_bad_actions off) */
#define FRZ_ERR_OVERFLOW 16u        /* rideshare: more live passengers than max_passengers slots */

/* ------------------------------------------------------------------------------------------------
 * Wildfire  (reference: free_range_zoo/envs/wildfire/env/wildfire.py, transitions/,
 *            structures/configuration.py)
 * ---------------------------------------------------------------------------------------------- */

typedef struct frz_wildfire_cfg {
    int32_t parallel_envs; /* B                                  utils/env.py:45 */
    int32_t grid_height;   /* H                                  configuration.py:331 */
    int32_t grid_width;    /* W */
    int32_t num_agents;    /* A                                  configuration.py:206 */
    int32_t max_steps;     /* < 0 means None (no truncation)     utils/env.py:229 */
    int32_t num_fire_states;
    int32_t num_equipment_states;
    int32_t num_capacities;

    /* StochasticConfiguration (configuration.py:276-316) + env flags (wildfire.py:174-179) */
    int32_t stochastic_increase;
    int32_t stochastic_burnouts; /* special_burnout_probability */
    int32_t stochastic_decrease;
    int32_t use_fire_fuel;
    int32_t stochastic_suppressant_decrease;
    int32_t stochastic_refill;
    int32_t stochastic_switch;
    int32_t stochastic_repair;
    int32_t stochastic_degrade;
    int32_t critical_error;
    int32_t show_bad_actions;
    int32_t observe_other_power;
    int32_t observe_other_suppressant;
    int32_t burnout_penalty_scaled;
    int32_t localize_putouts;
    int32_t track_cumulative_rewards; /* keep BatchedAECEnv._cumulative_rewards (utils/env.py:244-247) */

    /* probabilities, float32 exactly as the reference's registered buffers */
    float intensity_increase_probability;
    float burnout_probability;
    float intensity_decrease_probability;
    float extra_power_decrease_bonus;
    float suppressant_decrease_probability;
    float suppressant_refill_probability;
    float tank_switch_probability;
    float repair_probability;
    float degrade_probability;
    float critical_error_probability;
    /* fire_spread_weights 3x3 cross filter (configuration.py:346-363): p = N*lit[y-1,x] + W*lit[y,x-1]
     * + E*lit[y,x+1] + S*lit[y+1,x], accumulated in that order in float32 */
    float spread_n, spread_w, spread_e, spread_s;
    float random_ignition; /* fire_random_spread_weight (configuration.py:365-371) */

    /* RewardConfiguration (configuration.py:12-45) */
    float bad_attack_penalty;
    float burnout_penalty;
    float termination_reward;
    float termination_kappa;

    /* initial state scalars (wildfire.py:347-354) */
    int32_t initial_fuel;
    float initial_suppressant;
    float initial_capacity;
    int32_t initial_equipment_state;

    /* per agent (shared by all envs) */
    int32_t agent_y[FRZ_MAX_AGENTS];
    int32_t agent_x[FRZ_MAX_AGENTS];
    float fire_reduction_power[FRZ_MAX_AGENTS];
    float attack_range[FRZ_MAX_AGENTS];
    float equipment_states[FRZ_MAX_EQUIPMENT_STATES][3]; /* (capacity, power, range) bonus per state */
    float possible_capacities[FRZ_MAX_CAPACITIES];
    float capacity_cumprobs[FRZ_MAX_CAPACITIES]; /* float32 running sum of capacity_probabilities (capacity.py:27) */

    /* per cell, row-major c = y*W + x */
    float fire_rewards[FRZ_MAX_CELLS];
    int32_t ignition_temp[FRZ_MAX_CELLS];
    int32_t fire_types[FRZ_MAX_CELLS];
    int32_t lit[FRZ_MAX_CELLS];
} frz_wildfire_cfg;

/* Arrays of one wildfire env object.  They all live in ONE contiguous device arena whose layout the library fixes
 * (frz_wildfire_arena_bytes / frz_wildfire_bind); frz_wildfire_get_bufs() reports where each array sits so the caller
 * can wrap it as a typed view.  One base pointer keeps the kernels' scalar-register footprint small: every per-env
 * 4-byte array is a row of one [rows][B] block, addressed as base + (row * B + env) * 4.  "cap" = B*H*W rows.
 * The three cell arrays have one of two layouts (cells_env_major): grids of up to 16 cells are stepped one env per lane and keep the
 * env index innermost, [H*W][B]; larger grids are stepped one env per wavefront with the cells across its lanes and are env-major,
 * [B][H*W] — the reference's own layout. */
typedef struct frz_wildfire_bufs {
    /* state (WildfireState, structures/state.py:10-68), SoA */
    int32_t* fires;      /* [H*W][B], or [B][H*W] when cells_env_major */
    int32_t* intensity;  /* same */
    int32_t* fuel;       /* same */
    float* suppressants; /* [A][B] */
    float* capacity;     /* [A][B] */
    int32_t* equipment;  /* [A][B] */
    /* AEC bookkeeping (utils/env.py:94-160) */
    int32_t* num_moves;        /* [B] */
    int32_t* num_burnouts;     /* [B]            wildfire.py:357 */
    float* rewards;            /* [A][B]  out */
    float* cumulative_rewards; /* [A][B]  in/out BatchedAECEnv._cumulative_rewards */
    uint8_t* terminations;     /* [A][B]  in/out */
    uint8_t* truncations;      /* [A][B]  out */
    int64_t* burnouts;         /* [B] out  infos['burnouts'] (wildfire.py:581) */
    int64_t* putouts;          /* [B] out  infos['putouts'] */
    /* observations (wildfire.py:668-717) */
    float* obs_self;         /* [A][B][4] */
    float* obs_others;       /* [A][B][(A-1)*k], k = 2 + observe_other_power + observe_other_suppressant */
    int64_t* task_values;    /* [cap][4]  (y, x, fires, intensity) of lit fires */
    int64_t* task_offsets;   /* [B+1]  shared by task_values and obs_map_values */
    /* action/observation index maps + counts (wildfire.py:586-666) */
    int64_t* obs_map_values;    /* [cap]      local task indices 0..F_b-1 */
    int64_t* act_map_values;    /* [A][cap]   local indices of attackable fires */
    int64_t* act_map_offsets;   /* [A][B+1] */
    int64_t* bad_map_values;    /* [A][cap]   listed-but-not-attackable fires (filled only if show_bad_actions) */
    int64_t* bad_map_offsets;   /* [A][B+1] */
    int64_t* env_task_count;    /* [B] */
    int32_t* agent_task_count;  /* [A][B] */
    uint8_t* frozen_scaled;     /* [B] 1 once the stale-reward scaling of a frozen step (utils/env.py:211-213 +
                                   utils/conversions.py:87-90) was applied to env b; cleared by reset */
    /* per-env RNG state */
    int32_t* seeds;       /* [B]  (FRZ_RNG_PHILOX key; the seeds the MT19937 streams were started from) */
    uint32_t* mt_state;   /* [624][B] */
    int32_t* mt_index;    /* [B]       number of draws consumed modulo 624 */
    int32_t* actions;     /* [A][B][2] default action buffer (frz_wildfire_step accepts any device pointer) */
    uint32_t* error_flags; /* [1] sticky FRZ_ERR_* bits */
    int32_t cells_env_major; /* layout of fires / intensity / fuel, see above */
} frz_wildfire_bufs;

typedef struct frz_wildfire_env frz_wildfire_env; /* opaque host handle */

int frz_abi_version(void);
/* What a binding that is handed a handle as a plain integer can check before using it: 0 = not a live handle of this library, 1 wildfire,
 * 2 cybersecurity, 3 rideshare; and the sizes the handle was created with (units = cells / nodes / passenger slots per env). */
int frz_handle_kind(const void* handle);
int frz_handle_shape(const void* handle, int64_t* agents, int64_t* envs, int64_t* units);
int frz_wildfire_create(const frz_wildfire_cfg* cfg, frz_wildfire_env** out);
void frz_wildfire_destroy(frz_wildfire_env* env);
/* size of the device arena (state + outputs + RNG state + library scratch) for this configuration */
int64_t frz_wildfire_arena_bytes(const frz_wildfire_env* env);
/* attach a ZERO-FILLED device arena of frz_wildfire_arena_bytes(); uploads the configuration block (stream-ordered) */
int frz_wildfire_bind(frz_wildfire_env* env, void* arena, void* stream);
/* where each array of the bound arena lives */
int frz_wildfire_get_bufs(const frz_wildfire_env* env, frz_wildfire_bufs* out);
/* replaces raw_env.reset()'s state fill (wildfire.py:347-354) + bookkeeping zeroing (utils/env.py:137-160)
 * followed by update_observations/update_actions */
int frz_wildfire_reset(frz_wildfire_env* env, void* stream);
/* frz_wildfire_reset that first adds seed_increment (modulo 2^32: the seeds wrap around, no signed overflow) to every env seed (bufs.seeds; the FRZ_RNG_PHILOX key and the value the next
 * frz_mt19937_seed starts the env's stream from): the "fresh seeds, reset" pair at the top of every episode of a rollout loop
 * (the reference's `env.reset(seed=...)`, utils/env.py:94-160) as one launch */
int frz_wildfire_reset_reseed(frz_wildfire_env* env, int32_t seed_increment, void* stream);
/* replaces update_observations() + update_actions() (wildfire.py:586-717) on the bound state */
int frz_wildfire_rebuild(frz_wildfire_env* env, void* stream);
/* replaces one ParallelEnv.step(): BatchedAECEnv.step x A + step_environment + truncation + rebuild.
 * actions: int32 [A][B][2] (each agent's block is the reference's per-agent IntTensor[B,2]).
 * rng_mode FRZ_RNG_INJECTED: field_randomness float32 [3][B][H*W], agent_randomness float32 [5][B][A]
 * (= generator.generate(B,3,(H,W)) / generate(B,5,(A,)), wildfire.py:409-410); otherwise both NULL.
 * rng_mode FRZ_RNG_PHILOX draws the same tensors from Philox4x32-10 with key (seeds[b], 0x46525A00), step = num_moves[b]
 * before the step.  A 128-bit block (word 0 least significant) is read as five 24-bit uniforms:
 *   draw u of the step = bits [24k, 24k + 24) of block counter (u / 5, step, 0, 0), k = u % 5, float = field * 2^-24
 *   field event e of cell c = draw e * H*W + c;  agent event e of agent a = draw 3 * H*W + e * A + a */
int frz_wildfire_step(frz_wildfire_env* env, const int32_t* actions, int rng_mode, const float* field_randomness,
                      const float* agent_randomness, void* stream);
/* uniform random policy over OneOf([task]*n + [noop]) (spaces/actions.py:23-41): writes int32 [A][B][2];
 * member j = (word agent % 4 of Philox(counter (agent / 4, 0, step, step >> 32), key (seed ^ seeds[b], seed >> 32)) * (n + 1)) >> 32:
 * the stream of an env depends on its seed only, so a sharded batch draws what the unsharded one draws */
int frz_wildfire_random_policy(frz_wildfire_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out,
                               void* stream);
/* frz_wildfire_random_policy + frz_wildfire_step as ONE launch (the random-policy rollout of baselines/random.py): the
 * actions are sampled inside the step kernel from the same stream and left in actions_out (int32 [A][B][2]); results are
 * identical to the two calls in sequence (once every env is terminated or truncated the step ignores its actions and
 * actions_out is left as it is).  Grid shapes without a fused kernel run the two launches. */
int frz_wildfire_step_random_policy(frz_wildfire_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out,
                                    int rng_mode, const float* field_randomness, const float* agent_randomness, void* stream);
/* n_steps random-policy steps (policy steps first_step, first_step + 1, ...): what n_steps calls of frz_wildfire_step_random_policy
 * leave — state, rewards, the last step's observations, lists and sampled actions; intermediate steps' outputs are produced and
 * overwritten, as a loop over step() overwrites them — stream-ordered; rng_mode FRZ_RNG_PHILOX or FRZ_RNG_MT19937 (capture it into a
 * HIP graph for rollouts).  One launch per step, or — after frz_wildfire_set_exclusive_device(env, 1), for exact field/crew shapes
 * with FRZ_RNG_PHILOX or FRZ_RNG_MT19937 and parallel_envs <= 256 x CUs — ONE launch whose workgroups keep their envs in registers
 * from step to step
 * (frz_wildfire_rollout_launches tells which). */
int frz_wildfire_rollout_random_policy(frz_wildfire_env* env, uint64_t policy_seed, uint64_t first_step, int32_t n_steps,
                                       int32_t* actions_out, int rng_mode, void* stream);

/* Scripted baselines "always fight the strongest / weakest fire" (envs/wildfire/baselines/strongest.py:32-62, weakest.py) as a
 * device-side policy on the observation buffers: task_values int64 [*][4] + task_offsets int64 [B+1] (the `tasks` observation),
 * map_offsets int64 [B+1] / map_lengths int64 [B] (the agent's action mapping), obs_self float32 [B][4].  Candidates are
 * the intensities of the env's first map_lengths[b] task rows (indexed by position, as the reference does); ties are broken
 * by word 0 of Philox(counter (first_env_index + b, 0, step, step >> 32), key (seed, seed >> 32)); actions_out int32 [B][2]. */
int frz_wildfire_extreme_fire_policy(const int64_t* task_values, const int64_t* task_offsets, const int64_t* map_offsets,
                                     const int64_t* map_lengths, const float* obs_self, int64_t parallel_envs, int weakest,
                                     uint64_t seed, uint64_t step, int64_t first_env_index, int32_t* actions_out, void* stream);

/* Scripted rideshare baselines (envs/rideshare/baselines/greedy_Tfocus.py:48-115, greedy_Tglobal.py:48-99, fifo_Tfocus.py:36-80,
 * fifo_Tglobal.py:36-74) as one device-side policy on the observation buffers.  kind: 0 greedy_Tfocus, 1 greedy_Tglobal,
 * 2 fifo_Tfocus, 3 fifo_Tglobal.  task_values int32 [*][8] (16-byte aligned; rows y, x, y_dest, x_dest, accepted_by, riding_by,
 * fare, entered_step), task_offsets int64 [B] first row of each env, task_lengths int64 [B] rows per env (the `tasks`
 * observation), map_lengths int64 [B] (the agent's action mapping), obs_self int32 [B][4], diagonal = the agent configuration's
 * use_diagonal_travel.  The answer is drawn among the rows holding the minimal key: member tie_draws[b] (int64 [B]) when
 * tie_draws is given — e.g. replayed torch.randint draws — else floor(u32 * ties / 2^32) of word 0 of
 * Philox(counter (first_env_index + b, 0, step, step >> 32), key (seed, seed >> 32)).  actions_out int32 [B][2]. */
int frz_rideshare_task_policy(const int32_t* task_values, const int64_t* task_offsets, const int64_t* task_lengths,
                              const int64_t* map_lengths, const int32_t* obs_self, int64_t parallel_envs, int kind, int diagonal,
                              uint64_t seed, uint64_t step, int64_t first_env_index, const int64_t* tie_draws, int32_t* actions_out,
                              void* stream);

/* Stateful scripted cybersecurity baselines (envs/cybersecurity/baselines/patched.py:33-71,96-152, exploited.py:39-88,120-167,
 * camp.py:32-60) as one device-side policy.  kind: 0 patched attacker, 1 exploited attacker, 2 patched defender, 3 exploited
 * defender, 4 camp defender.  The candidates of env b are the row_len values tasks[b * env_stride + k * elem_stride] (int64) —
 * the reference reads observation['tasks'][:, 0], i.e. (N * F, 1, F) on the [B][N][F] task observation.  obs_self float32
 * [B][self_width] (attackers: threat, presence; defenders: mitigation, presence, location).  subnetwork_states is used by kind 1,
 * camp_target (agent index % nodes) by kind 4; mapping_numel == 0 (agent absent everywhere) answers [-100, -1] and keeps the
 * state.  One uniform draw per env among the positions holding the extreme key: member tie_draws[b] (int64 [B], nullable) or
 * floor(u32 * ties / 2^32) of word 0 of Philox(counter (first_env_index + b, 0, step, step >> 32), key (seed, seed >> 32)).
 * target_node int32 [B] (-1 = none), time_focused int32 [B] and actions int32 [B][2] are the agent's persistent state, updated
 * in place; actions is the answer. */
int frz_cybersecurity_focus_policy(const int64_t* tasks, int64_t env_stride, int64_t elem_stride, int32_t row_len, const float* obs_self,
                                   int32_t self_width, int64_t parallel_envs, int kind, int32_t subnetwork_states, int32_t camp_target,
                                   int64_t mapping_numel, uint64_t seed, uint64_t step, int64_t first_env_index, const int64_t* tie_draws,
                                   int32_t* target_node, int32_t* time_focused, int32_t* actions, void* stream);

/* Episode metrics in one launch (what a rollout loop reduces after an episode; utils/env.py:137-160 bookkeeping arrays):
 * metrics[a] += sum over envs of agent a's cumulative reward, metrics[A] += sum of num_moves, metrics[A + 1] += number of
 * envs whose agents are all terminated or all truncated.  metrics: float64 [A + 2] on the device, accumulated in place;
 * deterministic summation order. */
int frz_wildfire_episode_metrics(frz_wildfire_env* env, double* metrics, void* stream);
/* frz_wildfire_rollout_random_policy followed by frz_wildfire_episode_metrics — an episode of a rollout loop and its reductions — with
 * the same results (the float64 sums bit for bit: same summation order); where the rollout is one multi-step launch the reductions
 * are made in that launch's tail from the values its workgroups still hold, otherwise it is the two calls. */
int frz_wildfire_rollout_random_policy_metrics(frz_wildfire_env* env, uint64_t policy_seed, uint64_t first_step, int32_t n_steps,
                                               int32_t* actions_out, int rng_mode, double* metrics, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Rollouts: n_steps ParallelEnv.step() calls of a rollout loop enqueued by ONE call (reference: the loop of
 * utils/conversions.py:59-99 over given actions; docs/source/events/moasei-2026/evaluation.md `test()`; the partial resets
 * of utils/env.py:162-189 + wildfire.py:376-397 between steps).  The same spec drives every domain.
 * ---------------------------------------------------------------------------------------------- */
#define FRZ_ROLLOUT_RESET_FIRST 1u /* the rollout starts from the configured initial state: frz_<dom>_reset_reseed(seed_increment) folded
                                      into the first step (state, bookkeeping, seeds; its lists are never visible) */
#define FRZ_ROLLOUT_AUTO_RESET 2u  /* continuous rollouts at fixed B: an env that is finished (all agents terminated or truncated) at the
                                      end of step t is reset inside the same step — initial state, bookkeeping zeroed, seed += seed_stride
                                      (mod 2^32), observations / lists rebuilt from the fresh state — i.e. `env.step(...)` followed by
                                      `env.reset_batches(finished.nonzero(), seed)`: like the reference's, the reset zeroes the env's
                                      rewards / terminations / truncations, so what the step itself produced is read from the reward /
                                      done tapes (or the metrics).  The all-done early-out (utils/env.py:211-213) cannot trigger. */

#define FRZ_ROLLOUT_OBS_COMPACT 4u /* wildfire: obs_tape holds only what a step changes of the self / others records — each agent's suppressant,
                                      float32 [n_steps][A][B] (positions and base power are configuration: wildfire.py:677-716) */

typedef struct frz_rollout_spec {
    int32_t n_steps;
    int32_t rng_mode;        /* FRZ_RNG_* */
    uint32_t flags;          /* FRZ_ROLLOUT_* */
    int32_t seed_increment;  /* FRZ_ROLLOUT_RESET_FIRST: added (mod 2^32) to every env seed by the opening reset */
    uint32_t seed_stride;    /* FRZ_ROLLOUT_AUTO_RESET: added (mod 2^32) to an env's seed by each of its resets */
    uint32_t pad_;
    /* what the agents do: a tape of given actions, or (NULL) the uniform random policy of frz_<dom>_random_policy sampled in-kernel with
     * policy steps first_step, first_step + 1, ... */
    const int32_t* action_tape; /* int32 [n_steps][A][B][2] */
    uint64_t policy_seed, first_step;
    int32_t* actions_out;       /* policy only: the sampled actions, int32 [A][B][2] (the last executed step's), or every step's,
                                   int32 [n_steps][A][B][2], with record_actions != 0 */
    int32_t record_actions;
    int32_t pad2_;
    /* FRZ_RNG_INJECTED: the tensors RandomGenerator.generate() would return, one pair per step
     * (wildfire: float32 [n_steps][3][B][H*W] and [n_steps][5][B][A]; cybersecurity: [n_steps][B][N] and [n_steps][B][A]) */
    const float* randomness_tape_a;
    const float* randomness_tape_b;
    /* optional per-step records (NULL: not kept).  Without them a rollout leaves what a loop over step() leaves: the last step's outputs */
    float* reward_tape;  /* float32 [n_steps][A][B] */
    uint8_t* done_tape;  /* uint8 [n_steps][2][B]: (terminated, truncated) as the step set them */
    void* list_record;   /* n_steps - 1 copies of the env's packed-list block (frz_<dom>_list_block: offsets and values of every jagged
                            output, in arena order), copy t = the lists of step t; the last step's lists are in the env's own buffers */
    double* metrics;     /* float64 [A + 2], accumulated in place: frz_wildfire_episode_metrics at the end of the rollout; with
                            FRZ_ROLLOUT_AUTO_RESET instead: [a] += returns of the episodes that ENDED inside the rollout (needs
                            track_cumulative_rewards), [A] += env-steps executed, [A + 1] += episodes ended */
    /* what a learner keeps of a rollout besides rewards / dones / actions (reference: the `observations` the loop of
     * utils/conversions.py:92-99 hands back at every step, utils/env.py:223-225) — v5 */
    void* obs_tape;      /* n_steps copies of the dense observation block (frz_<dom>_obs_block: the self rows, then the others rows [,
                            cybersecurity: then every agent's task rows], as the step — and the reset of FRZ_ROLLOUT_AUTO_RESET — left
                            them); with FRZ_ROLLOUT_OBS_COMPACT (wildfire) float32 [n_steps][A][B] suppressants instead.  The jagged part
                            of an observation (wildfire `tasks`) is in the list record */
    void* state_tape;    /* n_steps copies of the state block (frz_<dom>_state_block: what env.state() shows after the step) */
} frz_rollout_spec;

/* n_steps steps driven by `spec`; same results as the equivalent loop over frz_wildfire_step / frz_wildfire_step_random_policy (and
 * frz_wildfire_reset_masked for FRZ_ROLLOUT_AUTO_RESET).  ONE multi-step launch where the library has one (see
 * frz_wildfire_rollout_launches: exact field/crew shapes, any rng_mode, frz_wildfire_set_exclusive_device on), otherwise one launch per step
 * (plus the reset launches).  FRZ_ROLLOUT_AUTO_RESET is not available with FRZ_RNG_MT19937 (FRZ_E_INVALID). */
int frz_wildfire_rollout(frz_wildfire_env* env, const frz_rollout_spec* spec, void* stream);
/* the packed-list block of the bound arena: device pointer and size in bytes (task_offsets, act_map_offsets, bad_map_offsets, task_values,
 * obs_map_values, act_map_values, bad_map_values, contiguous in this order, each 256-byte aligned); a list record holds copies of it */
int frz_wildfire_list_block(const frz_wildfire_env* env, void** block, int64_t* bytes);
/* the dense observation block of the bound arena (obs_self float32 [A][B][4], then — 256-byte aligned — obs_others float32
 * [A][B][A - 1][k]): device pointer, size in bytes, and the byte offset of the others rows inside it; an obs tape holds copies of it */
int frz_wildfire_obs_block(const frz_wildfire_env* env, void** block, int64_t* bytes, int64_t* others_offset);
/* the state of the bound arena as env.state() shows it, in two pieces: the cell arrays (fires, intensity, fuel: int32, `[3 H W][B]` rows —
 * or `[3][B][H W]` env-major for the grid family, frz_wildfire_bufs.cells_env_major) and the agent arrays (suppressants, capacity:
 * float32, equipment: int32; `[3 A][B]` rows).  One step of a state tape = the cell piece followed by the agent piece. */
int frz_wildfire_state_block(const frz_wildfire_env* env, void** cells, int64_t* cells_bytes, void** agents, int64_t* agents_bytes);
/* replaces reset_batches (utils/env.py:162-189 + wildfire.py:376-397) with the selection on the device: env b is reset when mask[b] != 0
 * (mask uint8 [B]), or — mask NULL — when it is finished (all agents terminated or all truncated); seeds[b] += seed_increment (mod 2^32) for
 * the reset envs (the FRZ_RNG_PHILOX key; MT19937 streams are re-seeded by the caller: frz_mt19937_seed_masked); then the observations and
 * lists of the whole batch are rebuilt.  Two launches, no host synchronisation. */
int frz_wildfire_reset_masked(frz_wildfire_env* env, const uint8_t* mask, int32_t seed_increment, void* stream);
/* The state a partial reset restores.  The reference's reset_batches puts back the state SAVED at reset (`self._state.restore_initial(
 * batch_indices)`, wildfire.py:391; utils/state.py:36-60) — which is the caller's `options['initial_state']` when reset() was given one, not
 * the configured initial state.  A binding that loads such a state tells the library where its saved copy lives: six strided device
 * arrays (element (env b, item i) at ptr[b * stride_env + i * stride_item], in elements; items = cells for fires / intensity / fuel
 * (int32), agents for suppressants / capacity (float32) / equipment (int32)).  From then on frz_wildfire_reset_masked — and
 * FRZ_ROLLOUT_AUTO_RESET, which then always takes the per-step path — restore from it.  NULL: back to the configured initial state (what
 * a plain reset() saves).  The arrays must stay alive and unchanged until the next call. */
typedef struct frz_wildfire_saved_state {
    const int32_t* fires;
    int64_t fires_stride_env, fires_stride_item;
    const int32_t* intensity;
    int64_t intensity_stride_env, intensity_stride_item;
    const int32_t* fuel;
    int64_t fuel_stride_env, fuel_stride_item;
    const float* suppressants;
    int64_t suppressants_stride_env, suppressants_stride_item;
    const float* capacity;
    int64_t capacity_stride_env, capacity_stride_item;
    const int32_t* equipment;
    int64_t equipment_stride_env, equipment_stride_item;
} frz_wildfire_saved_state;
int frz_wildfire_set_saved_initial(frz_wildfire_env* env, const frz_wildfire_saved_state* saved);

/* The caller states that nothing else runs on the device while this env's rollouts do (no other process, no concurrent stream): the
 * precondition of the multi-step launch, whose workgroups wait INSIDE the kernel for the other workgroups of their own grid (the batch
 * totals of step t gate step t + 1) and therefore must all be resident at once — sharing the CUs with another kernel that also waits
 * for its own stragglers can stall both until their bounded spins give up (FRZ_ERR_SCAN_TIMEOUT).  Off by default. */
int frz_wildfire_set_exclusive_device(frz_wildfire_env* env, int exclusive);
/* The library's own check behind frz_<dom>_set_exclusive_device(env, 1), which returns FRZ_E_INVALID when it fails: the launch's workgroups
 * (one per 256-env chunk) all fit on the device that owns the arena — occupancy of the multi-step instantiation
 * (hipOccupancyMaxActiveBlocksPerMultiprocessor) x its compute units — and no CU mask (ROC_GLOBAL_CU_MASK / HSA_CU_MASK) is in force.  The
 * decision itself, exported so that it can be unit-tested without a device: 1 = fits. */
int frz_exclusive_launch_fits(int64_t workgroups, int workgroups_per_cu, int compute_units, int cu_mask_set);
/* Globally consistent batch semantics under sharding (optional): the two batch-global tests of a step — "every env is finished"
 * (utils/env.py:211-213) and "agent a has no task in ANY env" (wildfire.py:434-435) — read the batch totals the previous step left.  Between
 * two steps a sharded job sums them over its ranks: export (int32 [A + 3], device), all_reduce(SUM), import.  Stream-ordered, no host read. */
int frz_wildfire_export_totals(frz_wildfire_env* env, int32_t* staging, void* stream);
int frz_wildfire_import_totals(frz_wildfire_env* env, const int32_t* staging, void* stream);
/* How many kernel launches frz_wildfire_rollout_random_policy(n_steps) enqueues for this env and RNG mode: 1 when the whole rollout runs
 * as one multi-step launch (exact field/crew shapes, FRZ_RNG_PHILOX or FRZ_RNG_MT19937, every chunk's workgroup resident at once,
 * frz_wildfire_set_exclusive_device on), n_steps otherwise. */
int frz_wildfire_rollout_launches(const frz_wildfire_env* env, int32_t n_steps, int rng_mode);
/* Measurement aid: frz_wildfire_rollout_random_policy(n_steps) when it is ONE launch, bracketed by a pair of HIP events that take that
 * dispatch's begin / end timestamps on `stream`; synchronises and returns the duration in milliseconds (FRZ_E_INVALID when the rollout
 * would take several launches). */
int frz_wildfire_timed_rollout_launch(frz_wildfire_env* env, uint64_t policy_seed, uint64_t first_step, int32_t n_steps, int32_t* actions_out,
                                      int rng_mode, void* stream, float* launch_ms);

/* Measurement aid: frz_wildfire_rollout(spec) when it is ONE launch, bracketed like frz_wildfire_timed_rollout_launch (FRZ_E_INVALID otherwise) */
int frz_wildfire_timed_rollout_spec(frz_wildfire_env* env, const frz_rollout_spec* spec, void* stream, float* launch_ms);

/* Measurement aid: n_steps launches of frz_wildfire_step_random_policy (policy steps first_step ...), back to back with no
 * host synchronisation in between, each bracketed by its own pair of HIP events that take the step dispatch's begin / end
 * timestamps on `stream` (what a profiler's kernel trace reports); synchronises once at the end and returns the durations
 * in milliseconds.  Randomness is drawn in-kernel or from the env's MT19937 streams (rng_mode != FRZ_RNG_INJECTED). */
int frz_wildfire_timed_rollout(frz_wildfire_env* env, uint64_t policy_seed, uint64_t first_step, int32_t n_steps, int32_t* actions_out,
                               int rng_mode, void* stream, float* kernel_ms);

/* ------------------------------------------------------------------------------------------------
 * Cybersecurity  (reference: free_range_zoo/envs/cybersecurity/env/cybersecurity.py, transitions/,
 *                 structures/configuration.py, utils/masking.py)
 * Agents are ordered attackers first, then defenders (cybersecurity.py:189-193).
 * ---------------------------------------------------------------------------------------------- */

typedef struct frz_cybersecurity_cfg {
    int32_t parallel_envs;  /* B */
    int32_t num_nodes;      /* N   configuration.py:205 */
    int32_t num_attackers;
    int32_t num_defenders;
    int32_t max_steps;      /* < 0 means None */
    int32_t num_states;     /* patched + vulnerable + exploited (configuration.py:210) */
    int32_t stochastic_state;          /* StochasticConfiguration.network_state */
    int32_t show_bad_actions;          /* cybersecurity.py:161-170 env flags */
    int32_t partially_observable;
    int32_t observe_other_power;
    int32_t observe_other_presence;
    int32_t observe_other_location;
    int32_t track_cumulative_rewards;
    float temperature;
    float bad_action_penalty;
    float patch_reward;
    float threat[FRZ_MAX_AGENTS];        /* per attacker */
    float mitigation[FRZ_MAX_AGENTS];    /* per defender */
    float persist_probs[FRZ_MAX_AGENTS]; /* all agents, attackers first (configuration.py:61-64) */
    float return_probs[FRZ_MAX_AGENTS];
    int32_t initial_presence[FRZ_MAX_AGENTS]; /* all agents */
    int32_t initial_location[FRZ_MAX_AGENTS]; /* per defender, -1 = home */
    int32_t initial_state[FRZ_MAX_NODES];
    int32_t criticality[FRZ_MAX_NODES]; /* adjacency row sums (configuration.py:199-201) */
    float network_state_rewards[FRZ_MAX_NETWORK_STATES];
} frz_cybersecurity_cfg;

/* Arrays of one cybersecurity env object (one device arena, like wildfire).  A = attackers + defenders. */
typedef struct frz_cybersecurity_bufs {
    int32_t* network_state; /* [N][B]   0 = best .. num_states-1 */
    int32_t* location;      /* [D][B]   -1 = home node */
    uint8_t* presence;      /* [A][B] */
    int32_t* last_action;   /* [D][B]   action id of the defender's last action (monitor = -3), -2 after reset */
    int32_t* num_moves;     /* [B] */
    float* rewards;            /* [A][B] */
    float* cumulative_rewards; /* [A][B] */
    uint8_t* terminations;     /* [A][B] always 0 (cybersecurity.py:298) */
    uint8_t* truncations;      /* [A][B] */
    float* obs_self_attackers;   /* [Att][B][2]  (threat, presence)               cybersecurity.py:481-484 */
    float* obs_self_defenders;   /* [D][B][3]    (mitigation, presence, location) cybersecurity.py:475-479 */
    float* obs_others_attackers; /* [Att][B][(Att-1)*ka] columns kept by utils/masking.py:6-28 */
    float* obs_others_defenders; /* [D][B][(D-1)*kd] */
    int64_t* obs_tasks;          /* [A][B][N][2] (state, criticality); -100 for a defender whose last action was not monitor */
    int32_t* act_map_values;     /* [A][B*N]  arange(N) per present env (cybersecurity.py:441-451) */
    int64_t* act_map_offsets;    /* [A][B+1] */
    int32_t* obs_map_values;     /* [B][N]    arange(N) per env; offsets are arange(B+1) (cybersecurity.py:434-439) */
    int64_t* obs_map_offsets;    /* [B+1] */
    int32_t* env_task_count;     /* [B]  = N  (int32: filled in place, cybersecurity.py:422) */
    int32_t* agent_task_count;   /* [A][B] = N * presence */
    uint8_t* frozen_scaled;      /* [B] */
    int32_t* seeds;       /* [B] */
    uint32_t* mt_state;   /* [624][B] */
    int32_t* mt_index;    /* [B] */
    int32_t* actions;     /* [A][B][2] default action buffer */
    uint32_t* error_flags;
} frz_cybersecurity_bufs;

typedef struct frz_cybersecurity_env frz_cybersecurity_env;

int frz_cybersecurity_create(const frz_cybersecurity_cfg* cfg, frz_cybersecurity_env** out);
void frz_cybersecurity_destroy(frz_cybersecurity_env* env);
int64_t frz_cybersecurity_arena_bytes(const frz_cybersecurity_env* env);
int frz_cybersecurity_bind(frz_cybersecurity_env* env, void* arena, void* stream);
int frz_cybersecurity_get_bufs(const frz_cybersecurity_env* env, frz_cybersecurity_bufs* out);
/* cybersecurity.py:218-266 (state from the configuration, last actions = -2) + update_observations/update_actions */
int frz_cybersecurity_reset(frz_cybersecurity_env* env, void* stream);
int frz_cybersecurity_rebuild(frz_cybersecurity_env* env, void* stream);
/* one ParallelEnv.step() (cybersecurity.py:295-526).  actions int32 [A][B][2]: attackers (node, 0) attack / (_, -1) noop;
 * defenders (node, 0) move / (_, -1) noop / (_, -2) patch / (_, -3) monitor.
 * FRZ_RNG_INJECTED: network_randomness float32 [1][B][N], agent_randomness float32 [1][B][A] (cybersecurity.py:304-315).
 * FRZ_RNG_PHILOX (key (seeds[b], 0x46525A01), step = num_moves[b] before the step, float = (word >> 8) * 2^-24):
 *   node n draw  = word n & 3 of counter (n >> 2, step, 0, 0);  agent a draw = word a & 3 of counter (a >> 2, step, 1, 0) */
int frz_cybersecurity_step(frz_cybersecurity_env* env, const int32_t* actions, int rng_mode, const float* network_randomness,
                           const float* agent_randomness, void* stream);
/* uniform random policy over each agent's OneOf action space (spaces/actions.py:11-99): agent a draws its member from word a % 4 of
 * Philox(counter (a / 4, 0, step, step >> 32), key (seed ^ seeds[b], seed >> 32)) — the wildfire policy's stream */
int frz_cybersecurity_random_policy(frz_cybersecurity_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out,
                                    void* stream);
/* frz_cybersecurity_random_policy + frz_cybersecurity_step as ONE launch (same results as the two calls): the actions are sampled
 * inside the step kernel from the state the launch starts with and left in actions_out (int32 [A][B][2]; not written once every env
 * is finished, when the step is a no-op).  A random rollout = the reference's `env.step({a: action_space(a).sample_nested()})` loop
 * (baselines/random.py:20) without a second launch per step. */
int frz_cybersecurity_step_random_policy(frz_cybersecurity_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out,
                                         int rng_mode, const float* network_randomness, const float* agent_randomness, void* stream);
/* n_steps random-policy steps (policy steps first_step, first_step + 1, ...): what n_steps calls of frz_cybersecurity_step_random_policy
 * leave; one launch per step, or — after frz_cybersecurity_set_exclusive_device(env, 1), shapes up to 8 nodes / 8 agents, FRZ_RNG_PHILOX,
 * parallel_envs <= 256 x CUs — ONE launch whose workgroups keep their envs in registers from step to step (see
 * frz_wildfire_rollout_random_policy / frz_wildfire_set_exclusive_device: same scheme, same precondition). */
int frz_cybersecurity_rollout_random_policy(frz_cybersecurity_env* env, uint64_t policy_seed, uint64_t first_step, int32_t n_steps,
                                            int32_t* actions_out, int rng_mode, void* stream);
/* n_steps steps driven by a frz_rollout_spec (see frz_wildfire_rollout): an action tape or the in-kernel uniform policy, FRZ_RNG_INJECTED tapes
 * (network float32 [n_steps][B][N], agent float32 [n_steps][B][A]), FRZ_ROLLOUT_RESET_FIRST, reward / done / action records and the list record
 * (frz_cybersecurity_list_block: act_map_values, act_map_offsets).  Episodes of this domain end together (truncation only,
 * cybersecurity.py:298): FRZ_ROLLOUT_AUTO_RESET and `metrics` are refused (FRZ_E_INVALID).  ONE multi-step launch for shapes up to 8 nodes / 8
 * agents with FRZ_RNG_PHILOX or FRZ_RNG_INJECTED after frz_cybersecurity_set_exclusive_device, otherwise one launch per step. */
int frz_cybersecurity_rollout(frz_cybersecurity_env* env, const frz_rollout_spec* spec, void* stream);
int frz_cybersecurity_list_block(const frz_cybersecurity_env* env, void** block, int64_t* bytes);
/* the dense observation block (attackers' self rows float32 [Att][B][2], defenders' self rows [D][B][3], the others rows of both, every
 * agent's task rows int64 [A][B][N][2]; each 256-byte aligned, in this order — frz_cybersecurity_bufs has the pointers) and the state
 * block in two pieces (network state, defender locations, last actions: int32 `[N + 2 D][B]` rows; presence: uint8 `[A][B]` rows).  One step
 * of a state tape = the rows piece followed by the presence piece, padded to a multiple of 256 bytes. */
/* reset_batches with the selection on the device (utils/env.py:162-189 + cybersecurity.py:268-292), as frz_wildfire_reset_masked: env b is
 * reset when mask[b] != 0 (uint8 [B]) or — mask NULL — when it is finished; seeds[b] += seed_increment (mod 2^32); state, last actions
 * (-2), the staged actions (-2) and the bookkeeping of the reset envs as a reset leaves them; then observations and mappings of the
 * batch are rebuilt.  The state put back is the configured initial state, or — after frz_cybersecurity_set_saved_initial — the state the
 * binding saved at reset (the caller's `options['initial_state']`): strided device arrays, element (env b, item i) at
 * ptr[b * stride_env + i * stride_item]; NULL clears it (a full reset does too). */
typedef struct frz_cybersecurity_saved_state {
    const int32_t* network_state; /* [B][N] */
    int64_t network_state_stride_env, network_state_stride_item;
    const int32_t* location; /* [B][D] */
    int64_t location_stride_env, location_stride_item;
    const uint8_t* presence; /* [B][A] (bool bytes) */
    int64_t presence_stride_env, presence_stride_item;
} frz_cybersecurity_saved_state;
int frz_cybersecurity_reset_masked(frz_cybersecurity_env* env, const uint8_t* mask, int32_t seed_increment, void* stream);
int frz_cybersecurity_set_saved_initial(frz_cybersecurity_env* env, const frz_cybersecurity_saved_state* saved);
int frz_cybersecurity_obs_block(const frz_cybersecurity_env* env, void** block, int64_t* bytes);
int frz_cybersecurity_state_block(const frz_cybersecurity_env* env, void** rows, int64_t* rows_bytes, void** presence, int64_t* presence_bytes);
int frz_cybersecurity_set_exclusive_device(frz_cybersecurity_env* env, int exclusive);
int frz_cybersecurity_rollout_launches(const frz_cybersecurity_env* env, int32_t n_steps, int rng_mode);

/* ------------------------------------------------------------------------------------------------
 * Rideshare  (reference: free_range_zoo/envs/rideshare/env/rideshare.py, transitions/,
 *             structures/configuration.py).  Deterministic: no randomness is drawn.
 * The reference keeps ONE global passenger table sorted by env (stable: entry order inside an env); here every env owns
 * max_passengers slots in that same order, env-major records [B][slot][FRZ_PASSENGER_COLUMNS]: one env is stepped by one wavefront whose
 * lanes hold its slots; a record is (y, x, state, driver, accepted, picked, y_dest, x_dest, fare, entered) — the columns a step can change
 * first, so that a changed slot is rewritten with two wide stores.
 * ---------------------------------------------------------------------------------------------- */
#define FRZ_MAX_PASSENGERS 128
#define FRZ_PASSENGER_COLUMNS 10 /* y, x, state, driver, accepted, picked, y_dest, x_dest, fare, entered */

typedef struct frz_rideshare_cfg {
    int32_t parallel_envs;
    int32_t grid_height, grid_width;
    int32_t num_agents;
    int32_t max_steps;       /* < 0 means None */
    int32_t max_passengers;  /* slots per env (<= FRZ_MAX_PASSENGERS); overflow sets FRZ_ERR_OVERFLOW */
    int32_t pool_limit;
    int32_t use_fast_travel, use_diagonal_travel; /* transitions/movement.py:15-53 */
    int32_t use_variable_move_cost, use_waiting_costs;
    int32_t track_cumulative_rewards;
    int32_t wait_limit[3]; /* per passenger state (unaccepted, accepted, riding) */
    int32_t long_wait_time;
    float move_cost, drop_cost, noop_cost, accept_cost, pool_limit_cost, general_wait_cost, long_wait_cost;
    int32_t start_y[FRZ_MAX_AGENTS], start_x[FRZ_MAX_AGENTS];
    int32_t schedule_rows;   /* rows of the schedule passed to create() */
    int32_t first_env_index; /* global index of env 0 of this shard (random-policy stream only) */
} frz_rideshare_cfg;

typedef struct frz_rideshare_bufs {
    int32_t* agents;           /* [B][A][2]  (y, x): the reference's own layout (structures/state.py:10-66) */
    int32_t* passengers;       /* [B][max_passengers][FRZ_PASSENGER_COLUMNS] */
    int32_t* passenger_count;  /* [B] */
    int32_t* num_moves;        /* [B] */
    float* rewards;            /* [A][B] */
    float* cumulative_rewards; /* [A][B] */
    uint8_t* terminations;     /* [A][B] always 0 (rideshare.py:252) */
    uint8_t* truncations;      /* [A][B] */
    int32_t* obs_self;         /* [A][B][4]        (y, x, #accepted, #riding)  rideshare.py:427-441 */
    int32_t* obs_others;       /* [A][B][A-1][4] */
    int32_t* task_values;      /* [cap][8] all passengers (y, x, y_dest, x_dest, accepted_by|-100, riding_by|-100, fare, entered) */
    int64_t* task_offsets;     /* [B+1]            task_store (rideshare.py:415-425); cap = B*max_passengers */
    int32_t* agent_task_values;  /* [A][cap][8]    the tasks agent a sees: unaccepted or its own (rideshare.py:446-456) */
    int64_t* agent_map_values;   /* [A][cap]       their positions in the env's passenger list (action = observation mapping) */
    int64_t* agent_offsets;      /* [A][B+1] */
    int32_t* agent_task_states;  /* [A][cap]       passenger state per visible task = the action id its OneOf member carries */
    int64_t* env_task_count;   /* [B] */
    int32_t* agent_task_count; /* [A][B] */
    int32_t* schedule;         /* [S][7] (timestep, env or -1, y, x, y_dest, x_dest, fare), device copy owned by the arena */
    uint8_t* frozen_scaled;    /* [B] */
    int32_t* actions;          /* [A][B][2] */
    uint32_t* error_flags;
} frz_rideshare_bufs;

typedef struct frz_rideshare_env frz_rideshare_env;

/* schedule: host pointer to int32 [schedule_rows][7] (PassengerConfiguration.schedule), copied at create.  Start positions and
 * schedule coordinates must lie within +-16383 (FRZ_E_INVALID otherwise). */
int frz_rideshare_create(const frz_rideshare_cfg* cfg, const int32_t* schedule, frz_rideshare_env** out);
void frz_rideshare_destroy(frz_rideshare_env* env);
int64_t frz_rideshare_arena_bytes(const frz_rideshare_env* env);
int frz_rideshare_bind(frz_rideshare_env* env, void* arena, void* stream);
int frz_rideshare_get_bufs(const frz_rideshare_env* env, frz_rideshare_bufs* out);
/* rideshare.py:185-222: agents at their start positions, passengers scheduled for step 0 enter, spaces rebuilt */
int frz_rideshare_reset(frz_rideshare_env* env, void* stream);
int frz_rideshare_rebuild(frz_rideshare_env* env, void* stream);
/* one ParallelEnv.step() (rideshare.py:248-467).  actions int32 [A][B][2]: (task index in the agent's mapping, action id)
 * with id -1 noop / 0 accept / 1 pick / 2 drop */
int frz_rideshare_step(frz_rideshare_env* env, const int32_t* actions, void* stream);
/* uniform member of OneOf([Discrete(1, start=state_t) for visible task t] + [noop]) (spaces/actions.py:10-50): agent a of env b draws
 * member floor(u32 * (n + 1) / 2^32) from word 0 of Philox(counter (a, first_env_index + b, step, step >> 32), key (seed, seed >> 32)) */
int frz_rideshare_random_policy(frz_rideshare_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out, void* stream);
/* frz_rideshare_random_policy + frz_rideshare_step with the policy sampled inside the step's first launch (same stream, same results as the
 * two calls; the reference's `env.step({a: action_space(a).sample_nested()})` loop, baselines/random.py:20); the sampled actions are left in
 * actions_out (int32 [A][B][2]; not written once every env is finished, when the step is a no-op) */
int frz_rideshare_step_random_policy(frz_rideshare_env* env, uint64_t policy_seed, uint64_t policy_step, int32_t* actions_out, void* stream);
/* n steps of the rollout loop (utils/conversions.py:59-99 over rideshare.py:248-467) enqueued by one call, driven by an action tape or the
 * in-launch uniform policy, with the reward / done / action tapes and the list record of frz_rollout_spec (frz_rideshare_list_block: task and
 * agent offsets, task rows, per-agent task rows, index maps, task states): one launch sequence per step, the records copied out between
 * the steps.  The domain draws nothing (rng_mode and the randomness tapes are not looked at); FRZ_ROLLOUT_RESET_FIRST is frz_rideshare_reset;
 * FRZ_ROLLOUT_AUTO_RESET, a seed increment and metrics are refused (FRZ_E_INVALID): the reference has no partial reset here
 * (rideshare.py:246) and the episodes of the batch end together */
int frz_rideshare_rollout(frz_rideshare_env* env, const frz_rollout_spec* spec, void* stream);
int frz_rideshare_list_block(const frz_rideshare_env* env, void** block, int64_t* bytes);
/* Measurement aid (bench.py): n_steps x frz_rideshare_step_random_policy back to back with no host synchronisation in between; a pair of
 * HIP events on `stream` takes the begin timestamp of each step's first dispatch and the end timestamp of its last one; synchronises once
 * at the end and returns the step durations in milliseconds */
int frz_rideshare_timed_rollout(frz_rideshare_env* env, uint64_t policy_seed, uint64_t first_step, int32_t n_steps, int32_t* actions_out,
                                void* stream, float* step_ms);

/* ------------------------------------------------------------------------------------------------
 * Per-env MT19937 streams  (reference: free_range_zoo/utils/random_generator.py:49-146; torch CPU
 * generator = MT19937 init_genrand(seed), float32 = (u32 & 0xFFFFFF) * 2^-24)
 * ---------------------------------------------------------------------------------------------- */
int frz_mt19937_seed(uint32_t* mt_state /*[624][B]*/, int32_t* mt_index /*[B]*/, const int32_t* seeds /*[B]*/,
                     const int32_t* batch_indices /* NULL = all */, int64_t n, int64_t B, void* stream);
/* out float32 [events][B][count]: env b draws events*count consecutive floats (generate(), unbuffered) */
int frz_mt19937_generate(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events, int64_t count, int64_t B,
                         void* stream);
/* two consecutive generate() calls on the same streams in one launch: `out` (events x count) first, then `out2`
 * (the randomness of one step: wildfire.py:409-410, cybersecurity.py:304-315) */
int frz_mt19937_generate_pair(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events, int64_t count, float* out2, int64_t events2,
                              int64_t count2, int64_t parallel_envs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FRZ_H_ */
