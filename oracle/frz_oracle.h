/*
 * frz_oracle.h — CPU restatement of the reference algorithm.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / reported CPU baseline — never as a product path.  The product (libfrz_hip.so) has no CPU fallback.
 *
 * Parity status: PINNED.  tests/test_oracle_*.py check every function here against (a) the reference's own
 * known-answer transition tests, recorded as data under tests/golden/ka_*.npz, and (b) trajectories of the
 * unmodified reference run in the build container (tests/golden/traj_*.npz, generator: tools/refharness/).
 *
 * Layout: the oracle keeps the REFERENCE's batch-major layout for state ([B][H*W], [B][A]) so that every loop
 * reads like the reference's tensor expression; outputs use the same layout as include/frz.h.
 */
#ifndef FRZ_ORACLE_H_
#define FRZ_ORACLE_H_

#include "../include/frz.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct frz_oracle_wildfire_bufs {
    /* state, batch-major as in WildfireState (structures/state.py:10-68) */
    int32_t* fires;      /* [B][H*W] */
    int32_t* intensity;  /* [B][H*W] */
    int32_t* fuel;       /* [B][H*W] */
    float* suppressants; /* [B][A] */
    float* capacity;     /* [B][A] */
    int32_t* equipment;  /* [B][A] */
    int32_t* num_moves;  /* [B] */
    int32_t* num_burnouts;
    float* rewards;            /* [A][B] */
    float* cumulative_rewards; /* [A][B] */
    uint8_t* terminations;     /* [A][B] */
    uint8_t* truncations;      /* [A][B] */
    int64_t* burnouts;
    int64_t* putouts;
    float* obs_self;   /* [A][B][4] */
    float* obs_others; /* [A][B][(A-1)*k] */
    int64_t* task_values;
    int64_t* task_offsets;
    int64_t* obs_map_values;
    int64_t* act_map_values;  /* [A][cap] */
    int64_t* act_map_offsets; /* [A][B+1] */
    int64_t* bad_map_values;
    int64_t* bad_map_offsets;
    int64_t* env_task_count;
    int32_t* agent_task_count; /* [A][B] */
    uint32_t* error_flags;
    /* BatchedAECEnv.step early-out bookkeeping (utils/env.py:211-213): 1 once every env is terminated or every
     * env is truncated; further steps leave everything untouched except rewards (see frz_oracle_wildfire_step) */
    int32_t* frozen; /* [2]: {frozen, stale_rewards_already_scaled} */
    /* optional (NULL: this batch is the whole batch): totals of the batch this one is a SHARD of, int64 [A + 3] = (lit fires, fires agent a can
     * attack ..., envs not terminated, envs not truncated), as they stand before the step — the two batch-global tests of a step
     * (utils/env.py:211-213, wildfire.py:434-435) are then evaluated on them: the reference's semantics for the unsharded batch */
    const int64_t* global_totals;
} frz_oracle_wildfire_bufs;

int frz_oracle_wildfire_reset(const frz_wildfire_cfg* cfg, frz_oracle_wildfire_bufs* s);
int frz_oracle_wildfire_rebuild(const frz_wildfire_cfg* cfg, frz_oracle_wildfire_bufs* s);
int frz_oracle_wildfire_step(const frz_wildfire_cfg* cfg, frz_oracle_wildfire_bufs* s, const int32_t* actions,
                             const float* field_randomness, const float* agent_randomness);

int frz_oracle_wildfire_reset_masked(const frz_wildfire_cfg* cfg, frz_oracle_wildfire_bufs* s, const uint8_t* mask, int32_t* seeds,
                                     int32_t seed_increment);

/* single transitions on batch-major arrays, for the reference's known-answer transition tests */
void frz_oracle_wf_suppressant_decrease(const frz_wildfire_cfg* cfg, float* supp, const uint8_t* users, const float* r,
                                        int64_t n);
void frz_oracle_wf_equipment(const frz_wildfire_cfg* cfg, int32_t* equipment, const float* r, int64_t n);
void frz_oracle_wf_suppressant_refill(const frz_wildfire_cfg* cfg, float* supp, const float* cap, const int32_t* equipment,
                                      const uint8_t* refills, const float* r, uint8_t* increased, int64_t n);
void frz_oracle_wf_capacity(const frz_wildfire_cfg* cfg, float* supp, float* cap, const uint8_t* targets, const float* r_size,
                            const float* r_switch, int64_t n);
void frz_oracle_wf_fire_increase(const frz_wildfire_cfg* cfg, int32_t* fires, int32_t* intensity, int32_t* fuel,
                                 const float* attack, const float* r, uint8_t* burned, int64_t n);
void frz_oracle_wf_fire_decrease(const frz_wildfire_cfg* cfg, int32_t* fires, int32_t* intensity, int32_t* fuel,
                                 const float* attack, const float* r, uint8_t* put_out, int64_t n);
void frz_oracle_wf_fire_spread(const frz_wildfire_cfg* cfg, int32_t* fires, int32_t* intensity, const int32_t* fuel,
                               const float* r, int64_t B);
int frz_oracle_in_range_chebyshev(int32_t ay, int32_t ax, int32_t ty, int32_t tx, float attack_range);

/* ---------------------------------------------------------------------------------------------- cybersecurity */
typedef struct frz_oracle_cybersecurity_bufs {
    /* state, batch-major as in CybersecurityState (structures/state.py:12-53) */
    int32_t* network_state; /* [B][N] */
    int32_t* location;      /* [B][D] */
    uint8_t* presence;      /* [B][A] */
    int32_t* last_action;   /* [B][D] */
    int32_t* num_moves;
    float* rewards;            /* [A][B] */
    float* cumulative_rewards; /* [A][B] */
    uint8_t* terminations;
    uint8_t* truncations;
    float* obs_self_attackers;
    float* obs_self_defenders;
    float* obs_others_attackers;
    float* obs_others_defenders;
    int64_t* obs_tasks;
    int32_t* act_map_values;
    int64_t* act_map_offsets;
    int32_t* obs_map_values;
    int64_t* obs_map_offsets;
    int32_t* env_task_count;
    int32_t* agent_task_count;
    uint32_t* error_flags;
    int32_t* frozen;
} frz_oracle_cybersecurity_bufs;

int frz_oracle_cybersecurity_reset(const frz_cybersecurity_cfg* cfg, frz_oracle_cybersecurity_bufs* s);
int frz_oracle_cybersecurity_rebuild(const frz_cybersecurity_cfg* cfg, frz_oracle_cybersecurity_bufs* s);
int frz_oracle_cybersecurity_step(const frz_cybersecurity_cfg* cfg, frz_oracle_cybersecurity_bufs* s, const int32_t* actions,
                                  const float* network_randomness, const float* agent_randomness);
void frz_oracle_cy_movement(int32_t* location, const int32_t* targets, const uint8_t* mask, int64_t n);
void frz_oracle_cy_presence(const frz_cybersecurity_cfg* cfg, uint8_t* presence, int32_t* location, const float* r, int64_t B);
void frz_oracle_cy_subnetwork(const frz_cybersecurity_cfg* cfg, int32_t* network_state, const float* patches, const float* attacks,
                              const float* r, int64_t n);
void frz_oracle_cybersecurity_philox_randomness(const frz_cybersecurity_cfg* cfg, const int32_t* seeds, const int32_t* num_moves,
                                                float* network, float* agent);
void frz_oracle_cybersecurity_random_policy(const frz_cybersecurity_cfg* cfg, const int32_t* agent_task_count, const int32_t* location,
                                            const int32_t* env_seeds, uint64_t seed, uint64_t step, int32_t* actions);

/* ------------------------------------------------------------------------------------------------- rideshare */
typedef struct frz_oracle_rideshare_bufs {
    int32_t* agents;           /* [B][A][2] batch-major as RideshareState.agents (structures/state.py:10-66) */
    int32_t* passengers;       /* [B][max_passengers][10] the env's rows of the reference's global table, in table order */
    int32_t* passenger_count;  /* [B] */
    int32_t* num_moves;
    float* rewards;
    float* cumulative_rewards;
    uint8_t* terminations;
    uint8_t* truncations;
    int32_t* obs_self;
    int32_t* obs_others;
    int32_t* task_values;
    int64_t* task_offsets;
    int32_t* agent_task_values;
    int64_t* agent_map_values;
    int64_t* agent_offsets;
    int32_t* agent_task_states;
    int64_t* env_task_count;
    int32_t* agent_task_count;
    uint32_t* error_flags;
    int32_t* frozen;
} frz_oracle_rideshare_bufs;

int frz_oracle_rideshare_reset(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, const int32_t* schedule);
int frz_oracle_rideshare_rebuild(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s);
int frz_oracle_rideshare_step(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, const int32_t* schedule, const int32_t* actions);
void frz_oracle_rs_move(const frz_rideshare_cfg* cfg, const int32_t vec[4], int32_t move[2], float* cost);
/* single transitions on the slot table (known-answer vectors of the reference's transition tests) */
int frz_oracle_rs_passenger_state(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, const uint8_t* accepts, const uint8_t* picks,
                                  const int32_t* targets, const int32_t* vectors, const int32_t* timesteps);
int frz_oracle_rs_passenger_exit(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, const uint8_t* drops, const int32_t* targets,
                                 const int32_t* vectors, int32_t* fares_out);
int frz_oracle_rs_passenger_entry(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, const int32_t* schedule, const int32_t* timesteps);
void frz_oracle_rideshare_random_policy(const frz_rideshare_cfg* cfg, const frz_oracle_rideshare_bufs* s, uint64_t seed, uint64_t step,
                                        int32_t* actions);

/* MT19937 per-env streams (utils/random_generator.py:76-114); state batch-major [B][624] */
void frz_oracle_mt19937_seed(uint32_t* mt_state, int32_t* mt_index, const int32_t* seeds, int64_t B);
void frz_oracle_mt19937_generate(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events, int64_t count, int64_t B);
/* Philox4x32-10 (Salmon et al., SC'11) block function */
void frz_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* the float stream FRZ_RNG_PHILOX defines: draw d of env with seed s at step t */
float frz_oracle_philox_uniform(int32_t seed, uint32_t step, uint32_t draw, uint32_t stream);
/* the scripted strongest / weakest fire baselines (envs/wildfire/baselines/strongest.py, weakest.py) on the jagged observation */
void frz_oracle_wildfire_extreme_policy(const int64_t* task_values, const int64_t* task_offsets, const int64_t* map_offsets,
                                        const int64_t* map_lengths, const float* obs_self, int64_t B, int weakest, uint64_t seed, uint64_t step,
                                        int64_t first_env, int32_t* actions);
/* the scripted greedy / FIFO rideshare baselines (envs/rideshare/baselines/{greedy,fifo}_T{focus,global}.py); kind 0 greedy_Tfocus,
 * 1 greedy_Tglobal, 2 fifo_Tfocus, 3 fifo_Tglobal; forced_pick (nullable) replays recorded tie-break draws, ties_out is nullable */
void frz_oracle_rideshare_task_policy(const int32_t* task_values, const int64_t* task_offsets, const int64_t* task_lengths,
                                      const int64_t* map_lengths, const int32_t* obs_self, int64_t B, int kind, int diagonal, uint64_t seed,
                                      uint64_t step, int64_t first_env, const int64_t* forced_pick, int64_t* ties_out, int32_t* actions);
/* the stateful patched / exploited / camp cybersecurity baselines (envs/cybersecurity/baselines/{patched,exploited,camp}.py);
 * kind 0 patched attacker, 1 exploited attacker, 2 patched defender, 3 exploited defender, 4 camp defender; target_node,
 * time_focused and actions are the agent's persistent state, updated in place */
void frz_oracle_cyber_focus_policy(const int64_t* tasks, int64_t env_stride, int64_t elem_stride, int32_t row_len, const float* obs_self,
                                   int32_t self_width, int64_t B, int kind, int32_t subnetwork_states, int32_t camp_target,
                                   int64_t mapping_numel, uint64_t seed, uint64_t step, int64_t first_env, const int64_t* forced_pick,
                                   int64_t* ties_out, int32_t* target_node, int32_t* time_focused, int32_t* actions);
void frz_oracle_wildfire_philox_randomness(const frz_wildfire_cfg* cfg, const int32_t* seeds, const int32_t* num_moves, float* field,
                                           float* agent);
void frz_oracle_wildfire_random_policy(const frz_wildfire_cfg* cfg, const int32_t* agent_task_count, const int64_t* env_task_count,
                                       const int32_t* env_seeds, uint64_t seed, uint64_t step, int32_t* actions);

int frz_oracle_cybersecurity_rollout(const frz_cybersecurity_cfg* cfg, frz_oracle_cybersecurity_bufs* s, const int32_t* env_seeds, uint64_t policy_seed,
                                     uint64_t first_step, int32_t n_steps, int32_t* actions, float* network, float* agent);
/* n_steps x (random policy, Philox randomness, step) in one call (bench.py's cpu_baseline threads) */
int frz_oracle_wildfire_rollout(const frz_wildfire_cfg* cfg, frz_oracle_wildfire_bufs* s, const int32_t* env_seeds, uint64_t policy_seed,
                                uint64_t first_step, int32_t n_steps, int32_t* actions, float* field, float* agent);

#ifdef __cplusplus
}
#endif
#endif
