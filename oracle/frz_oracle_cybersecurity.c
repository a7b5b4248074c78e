/*
 * frz_oracle_cybersecurity.c — scalar CPU restatement of the reference cybersecurity step path.
 * TEST INFRASTRUCTURE ONLY (see frz_oracle.h).  Build with -ffp-contract=off.
 * Reference lines are relative to /root/reference/free_range_zoo/envs/cybersecurity/env/.
 */
#include "frz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static int others_cols_att(const frz_cybersecurity_cfg* c) { return (c->observe_other_power ? 1 : 0) + (c->observe_other_presence ? 1 : 0); }
static int others_cols_def(const frz_cybersecurity_cfg* c) {
    return (c->observe_other_power ? 1 : 0) + (c->observe_other_presence ? 1 : 0) + (c->observe_other_location ? 1 : 0);
}

/* transitions/movement.py:16-32 */
void frz_oracle_cy_movement(int32_t* location, const int32_t* targets, const uint8_t* mask, int64_t n) {
    for (int64_t i = 0; i < n; ++i)
        if (mask[i]) location[i] = targets[i];
}

/* transitions/presence.py:34-60; arrays are [B][A] (presence, r) and [B][D] (location) */
void frz_oracle_cy_presence(const frz_cybersecurity_cfg* cfg, uint8_t* presence, int32_t* location, const float* r, int64_t B) {
    const int A = cfg->num_attackers + cfg->num_defenders, D = cfg->num_defenders, Att = cfg->num_attackers;
    for (int64_t b = 0; b < B; ++b)
        for (int a = 0; a < A; ++a) {
            const int p = presence[b * A + a];
            const int ret = !p && (r[b * A + a] < cfg->return_probs[a]);
            const int leave = p && (r[b * A + a] >= cfg->persist_probs[a]);
            if (ret) presence[b * A + a] = 1;
            if (leave) presence[b * A + a] = 0;
            if (ret && a >= Att) location[b * D + (a - Att)] = -1; /* returning defenders start at the home node */
        }
}

/* transitions/subnetwork.py:39-72; arrays [B][N] */
void frz_oracle_cy_subnetwork(const frz_cybersecurity_cfg* cfg, int32_t* network_state, const float* patches, const float* attacks,
                              const float* r, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        const float diff = patches[i] - attacks[i];
        const float danger = tanhf(diff / cfg->temperature);
        int better, worse;
        if (cfg->stochastic_state) { /* larger |danger| => LESS likely to move, as written (:57-59) */
            const float ad = fabsf(danger);
            better = danger > 0.0f && ad <= r[i];
            worse = danger < 0.0f && ad <= r[i];
        } else {
            better = danger > 0.0f;
            worse = danger < 0.0f;
        }
        int32_t s = network_state[i];
        if (better) s -= 1;
        if (worse) s += 1;
        if (s < 0) s = 0;
        if (s > cfg->num_states - 1) s = cfg->num_states - 1;
        network_state[i] = s;
    }
}

/* cybersecurity.py:459-526 (update_observations) then :413-457 (update_actions) */
int frz_oracle_cybersecurity_rebuild(const frz_cybersecurity_cfg* cfg, frz_oracle_cybersecurity_bufs* s) {
    const int64_t B = cfg->parallel_envs;
    const int N = cfg->num_nodes, Att = cfg->num_attackers, D = cfg->num_defenders, A = Att + D;
    const int ka = others_cols_att(cfg), kd = others_cols_def(cfg);
    for (int a = 0; a < A; ++a) s->act_map_offsets[(int64_t)a * (B + 1)] = 0;
    s->obs_map_offsets[0] = 0;
    for (int64_t b = 0; b < B; ++b) {
        for (int a = 0; a < Att; ++a) {
            float* self = s->obs_self_attackers + ((int64_t)a * B + b) * 2;
            self[0] = cfg->threat[a];
            self[1] = (float)s->presence[b * A + a];
            float* others = s->obs_others_attackers + ((int64_t)a * B + b) * (Att - 1) * ka;
            int col = 0;
            for (int o = 0; o < Att; ++o) {
                if (o == a) continue;
                if (cfg->observe_other_power) others[col++] = cfg->threat[o];
                if (cfg->observe_other_presence) others[col++] = (float)s->presence[b * A + o];
            }
        }
        for (int d = 0; d < D; ++d) {
            float* self = s->obs_self_defenders + ((int64_t)d * B + b) * 3;
            self[0] = cfg->mitigation[d];
            self[1] = (float)s->presence[b * A + Att + d];
            self[2] = (float)s->location[b * D + d];
            float* others = s->obs_others_defenders + ((int64_t)d * B + b) * (D - 1) * kd;
            int col = 0;
            for (int o = 0; o < D; ++o) {
                if (o == d) continue;
                if (cfg->observe_other_power) others[col++] = cfg->mitigation[o];
                if (cfg->observe_other_presence) others[col++] = (float)s->presence[b * A + Att + o];
                if (cfg->observe_other_location) others[col++] = (float)s->location[b * D + o];
            }
        }
        /* tasks: (state, criticality) int64; a defender sees them only right after a monitor action (:497, :510-511) */
        for (int a = 0; a < A; ++a) {
            int hidden = 0;
            if (a >= Att && cfg->partially_observable) hidden = s->last_action[b * D + (a - Att)] != -3;
            int64_t* t = s->obs_tasks + ((int64_t)a * B + b) * N * 2;
            for (int n = 0; n < N; ++n) {
                t[n * 2 + 0] = hidden ? -100 : s->network_state[b * N + n];
                t[n * 2 + 1] = hidden ? -100 : cfg->criticality[n];
            }
        }
        /* action / observation mappings and counts */
        s->env_task_count[b] = N;
        for (int n = 0; n < N; ++n) s->obs_map_values[b * N + n] = n;
        s->obs_map_offsets[b + 1] = b + 1; /* one row of N entries per env (:434-439) */
        for (int a = 0; a < A; ++a) {
            const int p = s->presence[b * A + a];
            s->agent_task_count[(int64_t)a * B + b] = p ? N : 0;
            int64_t* off = s->act_map_offsets + (int64_t)a * (B + 1);
            if (p)
                for (int n = 0; n < N; ++n) s->act_map_values[(int64_t)a * B * N + off[b] + n] = n;
            off[b + 1] = off[b] + (p ? N : 0);
        }
    }
    return FRZ_OK;
}

/* cybersecurity.py:218-266 + utils/env.py:94-160 */
int frz_oracle_cybersecurity_reset(const frz_cybersecurity_cfg* cfg, frz_oracle_cybersecurity_bufs* s) {
    const int64_t B = cfg->parallel_envs;
    const int N = cfg->num_nodes, Att = cfg->num_attackers, D = cfg->num_defenders, A = Att + D;
    for (int64_t b = 0; b < B; ++b) {
        for (int n = 0; n < N; ++n) s->network_state[b * N + n] = cfg->initial_state[n];
        for (int d = 0; d < D; ++d) {
            s->location[b * D + d] = cfg->initial_location[d];
            s->last_action[b * D + d] = -2; /* cybersecurity.py:233-236 */
        }
        for (int a = 0; a < A; ++a) {
            s->presence[b * A + a] = (uint8_t)(cfg->initial_presence[a] != 0);
            s->rewards[(int64_t)a * B + b] = 0.0f;
            s->cumulative_rewards[(int64_t)a * B + b] = 0.0f;
            s->terminations[(int64_t)a * B + b] = 0;
            s->truncations[(int64_t)a * B + b] = 0;
        }
        s->num_moves[b] = 0;
    }
    s->frozen[0] = s->frozen[1] = 0;
    return frz_oracle_cybersecurity_rebuild(cfg, s);
}

/* One ParallelEnv.step(): utils/conversions.py:59-99 -> utils/env.py:203-242 -> cybersecurity.py:295-411 -> rebuild */
int frz_oracle_cybersecurity_step(const frz_cybersecurity_cfg* cfg, frz_oracle_cybersecurity_bufs* s, const int32_t* actions,
                                  const float* network_randomness, const float* agent_randomness) {
    const int64_t B = cfg->parallel_envs;
    const int N = cfg->num_nodes, Att = cfg->num_attackers, D = cfg->num_defenders, A = Att + D;
    { /* utils/env.py:211-213 early-out + utils/conversions.py:87-90 stale-reward accumulation (see wildfire oracle) */
        int all_term = 1, all_trunc = 1;
        for (int64_t b = 0; b < B; ++b) {
            all_term = all_term && s->terminations[b];
            all_trunc = all_trunc && s->truncations[b];
        }
        if (all_term || all_trunc) {
            if (!s->frozen[1]) {
                for (int64_t i = 0; i < (int64_t)A * B; ++i) {
                    float acc = 0.0f;
                    for (int k = 0; k < A; ++k) acc = acc + s->rewards[i];
                    s->rewards[i] = acc;
                }
                s->frozen[1] = 1;
            }
            s->frozen[0] = 1;
            return FRZ_OK;
        }
    }
    float* patches = (float*)calloc((size_t)(B * N), sizeof(float));
    float* attacks = (float*)calloc((size_t)(B * N), sizeof(float));
    int32_t* targets = (int32_t*)calloc((size_t)(B * (D > 0 ? D : 1)), sizeof(int32_t));
    uint8_t* moving = (uint8_t*)calloc((size_t)(B * (D > 0 ? D : 1)), 1);
    for (int64_t i = 0; i < (int64_t)A * B; ++i) s->rewards[i] = 0.0f;

    for (int a = 0; a < A; ++a) { /* self.actions order = possible_agents = attackers then defenders */
        for (int64_t b = 0; b < B; ++b) {
            const int32_t idx = actions[((int64_t)a * B + b) * 2 + 0], act = actions[((int64_t)a * B + b) * 2 + 1];
            const int present = s->presence[b * A + a];
            /* :341-346 / :358-363 raise ValueError; flagged here */
            if (act == 0 && (idx < 0 || idx >= N)) {
                s->error_flags[0] |= FRZ_ERR_INVALID_TARGET;
                continue;
            }
            if (!cfg->show_bad_actions && !present && act != -1) s->error_flags[0] |= FRZ_ERR_ABSENT_ACTION;
            if (a < Att) {
                if (act == 0) attacks[b * N + idx] = attacks[b * N + idx] + cfg->threat[a]; /* no presence check (:348-350) */
            } else {
                const int d = a - Att;
                const int32_t loc = s->location[b * D + d];
                if (act == 0) {
                    moving[b * D + d] = 1;
                    targets[b * D + d] = idx;
                }
                if (act == -2 && loc != -1) { /* patch at the CURRENT (pre-move) location (:354, :372-378) */
                    patches[b * N + loc] = patches[b * N + loc] + cfg->mitigation[d];
                    s->rewards[(int64_t)a * B + b] = s->rewards[(int64_t)a * B + b] + cfg->patch_reward;
                    /* bad_patch (:380-382) needs location == -1, which `patch` excludes: never applies */
                }
                s->last_action[b * D + d] = act;
            }
        }
    }
    frz_oracle_cy_movement(s->location, targets, moving, B * D);
    frz_oracle_cy_presence(cfg, s->presence, s->location, agent_randomness, B);
    frz_oracle_cy_subnetwork(cfg, s->network_state, patches, attacks, network_randomness, B * N);
    for (int64_t b = 0; b < B; ++b) {
        /* :396-399 matmul(state_rewards[network_state], criticality.float()): sequential float32 dot product */
        float net = 0.0f;
        for (int n = 0; n < N; ++n) net = net + cfg->network_state_rewards[s->network_state[b * N + n]] * (float)cfg->criticality[n];
        for (int a = 0; a < A; ++a) {
            float* r = &s->rewards[(int64_t)a * B + b];
            *r = a < Att ? *r + net * -1.0f : *r + net;
        }
        s->num_moves[b] += 1;
        for (int a = 0; a < A; ++a) {
            if (cfg->max_steps >= 0) s->truncations[(int64_t)a * B + b] = s->num_moves[b] >= cfg->max_steps;
            if (cfg->track_cumulative_rewards)
                s->cumulative_rewards[(int64_t)a * B + b] = s->cumulative_rewards[(int64_t)a * B + b] + s->rewards[(int64_t)a * B + b];
        }
    }
    free(patches);
    free(attacks);
    free(targets);
    free(moving);
    return frz_oracle_cybersecurity_rebuild(cfg, s);
}

/* The randomness tensors a FRZ_RNG_PHILOX cybersecurity step consumes (include/frz.h) */
void frz_oracle_cybersecurity_philox_randomness(const frz_cybersecurity_cfg* cfg, const int32_t* seeds, const int32_t* num_moves,
                                                float* network, float* agent) {
    const int64_t B = cfg->parallel_envs;
    const int N = cfg->num_nodes, A = cfg->num_attackers + cfg->num_defenders;
    for (int64_t b = 0; b < B; ++b) {
        const uint32_t key[2] = {(uint32_t)seeds[b], 0x46525A01u};
        for (int i = 0; i < N; ++i) {
            const uint32_t ctr[4] = {(uint32_t)(i >> 2), (uint32_t)num_moves[b], 0u, 0u};
            uint32_t out[4];
            frz_oracle_philox4x32_10(ctr, key, out);
            network[b * N + i] = (float)(out[i & 3] >> 8) * (1.0f / 16777216.0f);
        }
        for (int i = 0; i < A; ++i) {
            const uint32_t ctr[4] = {(uint32_t)(i >> 2), (uint32_t)num_moves[b], 1u, 0u};
            uint32_t out[4];
            frz_oracle_philox4x32_10(ctr, key, out);
            agent[b * A + i] = (float)(out[i & 3] >> 8) * (1.0f / 16777216.0f);
        }
    }
}

/* Uniform member of each agent's OneOf action space (spaces/actions.py:11-99): n task members then the tail
 * attacker [noop]; defender [noop, patch, monitor] (patch dropped at the home node unless show_bad_actions; noop only
 * when the agent has no task).  Agent a draws word a % 4 of Philox(counter (a / 4, 0, step lo, step hi), key (seed lo ^ env seed, seed hi)):
 * the stream the wildfire policy defines (four agents share a block). */
void frz_oracle_cybersecurity_random_policy(const frz_cybersecurity_cfg* cfg, const int32_t* agent_task_count, const int32_t* location,
                                            const int32_t* env_seeds, uint64_t seed, uint64_t step, int32_t* actions) {
    const int64_t B = cfg->parallel_envs;
    const int N = cfg->num_nodes, Att = cfg->num_attackers, D = cfg->num_defenders, A = Att + D;
    for (int64_t i = 0; i < (int64_t)A * B; ++i) {
        const int a = (int)(i / B);
        const int64_t b = i % B;
        const int n = cfg->show_bad_actions ? N : agent_task_count[i];
        int tail[3], nt = 0;
        tail[nt++] = -1;
        if (a >= Att && n > 0) {
            const int home = location[b * D + (a - Att)] == -1;
            if (cfg->show_bad_actions || !home) tail[nt++] = -2;
            tail[nt++] = -3;
        }
        const uint32_t ctr[4] = {(uint32_t)(a >> 2), 0u, (uint32_t)step, (uint32_t)(step >> 32)};
        const uint32_t key[2] = {(uint32_t)seed ^ (uint32_t)env_seeds[b], (uint32_t)(seed >> 32)};
        uint32_t out[4];
        frz_oracle_philox4x32_10(ctr, key, out);
        const int j = (int)(((uint64_t)out[a & 3] * (uint64_t)(n + nt)) >> 32);
        actions[i * 2 + 0] = j;
        actions[i * 2 + 1] = j < n ? 0 : tail[j - n];
    }
}
