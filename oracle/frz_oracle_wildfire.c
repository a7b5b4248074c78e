/*
 * frz_oracle_wildfire.c — scalar CPU restatement of the reference wildfire step path.
 * TEST INFRASTRUCTURE ONLY (see frz_oracle.h).  Build with -ffp-contract=off: every float32 operation below
 * must round exactly once, as the reference's eager torch ops do.
 *
 * Each function cites the reference lines (relative to /root/reference/free_range_zoo/) it restates.
 */
#include "frz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float clamp01(float p) { /* torch.clamp(p, 0, 1) */
    p = p < 0.0f ? 0.0f : p;
    return p > 1.0f ? 1.0f : p;
}

/* envs/wildfire/env/utils/in_range_check.py:5-23 — int Chebyshev distance compared with a float32 range */
int frz_oracle_in_range_chebyshev(int32_t ay, int32_t ax, int32_t ty, int32_t tx, float attack_range) {
    int32_t dy = abs(ay - ty), dx = abs(ax - tx);
    int32_t d = dy > dx ? dy : dx;
    return (float)d <= attack_range;
}

/* transitions/suppressant_decrease.py:33-63 */
void frz_oracle_wf_suppressant_decrease(const frz_wildfire_cfg* cfg, float* supp, const uint8_t* users, const float* r,
                                        int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        int mask = users[i];
        if (cfg->stochastic_suppressant_decrease) mask = mask && (r[i] < cfg->suppressant_decrease_probability);
        float v = mask ? supp[i] - 1.0f : supp[i];
        supp[i] = v < 0.0f ? 0.0f : v; /* clamp(min=0) */
    }
}

/* transitions/equipment.py:41-77 — all masks come from the equipment value BEFORE any write */
void frz_oracle_wf_equipment(const frz_wildfire_cfg* cfg, int32_t* equipment, const float* r, int64_t n) {
    const int32_t top = cfg->num_equipment_states - 1;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t e0 = equipment[i];
        const int pristine = e0 == top;
        const int damaged = e0 == 0;
        const int intermediate = !pristine && !damaged;
        const int repairs = cfg->stochastic_repair ? (damaged && r[i] < cfg->repair_probability) : damaged;
        int32_t e = e0;
        if (repairs) e = top;
        int criticals = 0;
        if (cfg->critical_error) {
            criticals = pristine && r[i] < cfg->critical_error_probability;
            if (criticals) e = 0;
        }
        int degrades = cfg->stochastic_degrade ? ((pristine || intermediate) && r[i] < cfg->degrade_probability)
                                               : (intermediate || pristine);
        if (cfg->critical_error) degrades = degrades && !criticals;
        if (degrades) e -= 1;
        equipment[i] = e;
    }
}

/* transitions/suppressant_refill.py:42-74 — bonus uses the equipment state AFTER the equipment transition */
void frz_oracle_wf_suppressant_refill(const frz_wildfire_cfg* cfg, float* supp, const float* cap, const int32_t* equipment,
                                      const uint8_t* refills, const float* r, uint8_t* increased, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        int inc = refills[i];
        if (cfg->stochastic_refill) inc = inc && (r[i] < cfg->suppressant_refill_probability);
        if (inc) supp[i] = cap[i] + cfg->equipment_states[equipment[i]][0];
        increased[i] = (uint8_t)inc;
    }
}

/* transitions/capacity.py:38-66 — torch.bucketize(right=False): first index with r <= cum[index] */
void frz_oracle_wf_capacity(const frz_wildfire_cfg* cfg, float* supp, float* cap, const uint8_t* targets, const float* r_size,
                            const float* r_switch, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        int32_t idx = 0;
        while (idx < cfg->num_capacities && !(r_size[i] <= cfg->capacity_cumprobs[idx])) ++idx;
        if (idx >= cfg->num_capacities) idx = cfg->num_capacities - 1; /* reference would raise IndexError */
        const float new_max = cfg->possible_capacities[idx];
        int sw = 0;
        if (targets[i]) sw = cfg->stochastic_switch ? (r_switch[i] < cfg->tank_switch_probability) : 1;
        const float bonus = supp[i] - cap[i];
        if (sw) {
            cap[i] = new_max;
            supp[i] = new_max + bonus;
        }
    }
}

/* transitions/fire_increase.py:42-95 */
void frz_oracle_wf_fire_increase(const frz_wildfire_cfg* cfg, int32_t* fires, int32_t* intensity, int32_t* fuel,
                                 const float* attack, const float* r, uint8_t* burned, int64_t n) {
    const int32_t almost_state = cfg->num_fire_states - 2, burnout_state = cfg->num_fire_states - 1;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t required = fires[i] >= 0 ? fires[i] : 0;
        const float diff = (float)required - attack[i];
        const int lit = fires[i] > 0 && intensity[i] > 0;
        const int unmet = diff > 0.0f && lit;
        const int almost = unmet && intensity[i] == almost_state;
        const int increasing = unmet && !almost;
        float p = 0.0f;
        if (increasing) p = cfg->stochastic_increase ? cfg->intensity_increase_probability : 1.0f;
        if (almost) p = cfg->stochastic_burnouts ? cfg->burnout_probability : cfg->intensity_increase_probability; /* :77-80 quirk */
        p = clamp01(p);
        const int inc = r[i] < p;
        if (inc) intensity[i] += 1;
        const int b = inc && intensity[i] >= burnout_state;
        if (b) {
            fires[i] *= -1;
            const int32_t f = fuel[i] - 1;
            fuel[i] = f < 0 ? 0 : f;
        }
        burned[i] = (uint8_t)b;
    }
}

/* transitions/fire_decrease.py:35-80 — probability = p_dec + ((-1 * diff) * bonus), each op rounded to f32 */
void frz_oracle_wf_fire_decrease(const frz_wildfire_cfg* cfg, int32_t* fires, int32_t* intensity, int32_t* fuel,
                                 const float* attack, const float* r, uint8_t* put_out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        const int32_t required = fires[i] >= 0 ? fires[i] : 0;
        const float diff = (float)required - attack[i];
        const int lit = fires[i] > 0 && intensity[i] > 0;
        const int met = diff <= 0.0f && lit;
        float p = 0.0f;
        if (met) {
            if (cfg->stochastic_decrease) {
                const float neg = -1.0f * diff;
                const float scaled = neg * cfg->extra_power_decrease_bonus;
                p = cfg->intensity_decrease_probability + scaled;
            } else {
                p = 1.0f;
            }
        }
        p = clamp01(p);
        const int dec = r[i] < p;
        if (dec) intensity[i] -= 1;
        const int po = dec && intensity[i] <= 0;
        if (po) {
            fires[i] *= -1;
            fuel[i] -= 1; /* unclamped, :75 */
        }
        put_out[i] = (uint8_t)po;
    }
}

/* transitions/fire_spreads.py:32-59 + structures/configuration.py:346-363.  conv2d(1->1, 3x3, pad 1) of the 0/1 lit map
 * with the cross filter == N*lit[y-1,x] + W*lit[y,x-1] + E*lit[y,x+1] + S*lit[y+1,x] accumulated in filter row-major order
 * (checked against torch.nn.functional.conv2d for random weights/shapes, tests/golden/conv_order.npz). */
void frz_oracle_wf_fire_spread(const frz_wildfire_cfg* cfg, int32_t* fires, int32_t* intensity, const int32_t* fuel,
                               const float* r, int64_t B) {
    const int32_t H = cfg->grid_height, W = cfg->grid_width, HW = H * W;
    uint8_t* lit = (uint8_t*)malloc((size_t)HW);
    for (int64_t b = 0; b < B; ++b) {
        int32_t* f = fires + b * HW;
        int32_t* in = intensity + b * HW;
        const int32_t* fu = fuel + b * HW;
        for (int32_t c = 0; c < HW; ++c) lit[c] = f[c] > 0 && in[c] > 0;
        for (int32_t y = 0; y < H; ++y) {
            for (int32_t x = 0; x < W; ++x) {
                const int32_t c = y * W + x;
                float p = 0.0f;
                p = p + cfg->spread_n * (y > 0 ? (float)lit[c - W] : 0.0f);
                p = p + cfg->spread_w * (x > 0 ? (float)lit[c - 1] : 0.0f);
                p = p + cfg->spread_e * (x < W - 1 ? (float)lit[c + 1] : 0.0f);
                p = p + cfg->spread_s * (y < H - 1 ? (float)lit[c + W] : 0.0f);
                int unlit = f[c] < 0 && in[c] == 0;
                if (cfg->use_fire_fuel) unlit = unlit && fu[c] > 0;
                p = unlit ? p + cfg->random_ignition : 0.0f;
                if (r[b * HW + c] < p) {
                    f[c] *= -1;
                    in[c] = cfg->ignition_temp[c];
                }
            }
        }
    }
    free(lit);
}

static int others_width(const frz_wildfire_cfg* cfg) { return 2 + (cfg->observe_other_power ? 1 : 0) + (cfg->observe_other_suppressant ? 1 : 0); }

/* wildfire.py:668-717 (update_observations) then :586-666 (update_actions) */
int frz_oracle_wildfire_rebuild(const frz_wildfire_cfg* cfg, frz_oracle_wildfire_bufs* s) {
    const int64_t B = cfg->parallel_envs;
    const int32_t H = cfg->grid_height, W = cfg->grid_width, HW = H * W, A = cfg->num_agents;
    const int64_t cap = B * HW;
    const int k = others_width(cfg);
    s->task_offsets[0] = 0;
    for (int32_t a = 0; a < A; ++a) {
        s->act_map_offsets[(int64_t)a * (B + 1)] = 0;
        if (s->bad_map_offsets) s->bad_map_offsets[(int64_t)a * (B + 1)] = 0;
    }
    for (int64_t b = 0; b < B; ++b) {
        /* agent observations: (y, x, fire_reduction_power, suppressants), others masked by observe_other_* (:230-233) */
        for (int32_t a = 0; a < A; ++a) {
            float* self = s->obs_self + ((int64_t)a * B + b) * 4;
            self[0] = (float)cfg->agent_y[a];
            self[1] = (float)cfg->agent_x[a];
            self[2] = cfg->fire_reduction_power[a];
            self[3] = s->suppressants[b * A + a];
            float* others = s->obs_others + ((int64_t)a * B + b) * (A - 1) * k;
            int32_t j = 0;
            for (int32_t o = 0; o < A; ++o) {
                if (o == a) continue;
                float* row = others + j * k;
                int32_t col = 0;
                row[col++] = (float)cfg->agent_y[o];
                row[col++] = (float)cfg->agent_x[o];
                if (cfg->observe_other_power) row[col++] = cfg->fire_reduction_power[o];
                if (cfg->observe_other_suppressant) row[col++] = s->suppressants[b * A + o];
                ++j;
            }
        }
        /* lit fires (fires > 0), row-major: nonzero() order */
        int64_t base = s->task_offsets[b];
        int64_t F = 0;
        for (int32_t c = 0; c < HW; ++c) {
            if (s->fires[b * HW + c] > 0) {
                int64_t* row = s->task_values + (base + F) * 4;
                row[0] = c / W;
                row[1] = c % W;
                row[2] = s->fires[b * HW + c];
                row[3] = s->intensity[b * HW + c];
                s->obs_map_values[base + F] = F;
                ++F;
            }
        }
        s->task_offsets[b + 1] = base + F;
        s->env_task_count[b] = F;
        /* per agent: in range (chebyshev, range + equipment range bonus) and has suppressant (:604-623) */
        for (int32_t a = 0; a < A; ++a) {
            const float true_range = cfg->attack_range[a] + cfg->equipment_states[s->equipment[b * A + a]][2];
            const int has = s->suppressants[b * A + a] > 0.0f;
            int64_t* ao = s->act_map_offsets + (int64_t)a * (B + 1);
            int64_t* av = s->act_map_values + (int64_t)a * cap;
            int64_t* bo = s->bad_map_offsets ? s->bad_map_offsets + (int64_t)a * (B + 1) : NULL;
            int64_t* bv = s->bad_map_values ? s->bad_map_values + (int64_t)a * cap : NULL;
            int64_t n_ok = 0, n_bad = 0;
            for (int64_t t = 0; t < F; ++t) {
                const int64_t* row = s->task_values + (base + t) * 4;
                const int ok = frz_oracle_in_range_chebyshev(cfg->agent_y[a], cfg->agent_x[a], (int32_t)row[0], (int32_t)row[1],
                                                             true_range) && has;
                if (ok) {
                    av[ao[b] + n_ok++] = t;
                } else if (bv) {
                    bv[bo[b] + n_bad++] = t;
                }
            }
            ao[b + 1] = ao[b] + n_ok;
            if (bo) bo[b + 1] = bo[b] + n_bad;
            s->agent_task_count[(int64_t)a * B + b] = (int32_t)n_ok;
        }
    }
    return FRZ_OK;
}

/* wildfire.py:290-373 (state from configuration) + utils/env.py:94-160 (bookkeeping) */
int frz_oracle_wildfire_reset(const frz_wildfire_cfg* cfg, frz_oracle_wildfire_bufs* s) {
    const int64_t B = cfg->parallel_envs;
    const int32_t HW = cfg->grid_height * cfg->grid_width, A = cfg->num_agents;
    for (int64_t b = 0; b < B; ++b) {
        for (int32_t c = 0; c < HW; ++c) {
            const int32_t type = cfg->fire_types[c];
            const int32_t f = cfg->lit[c] ? type : -type;
            s->fires[b * HW + c] = f;
            s->intensity[b * HW + c] = cfg->lit[c] ? cfg->ignition_temp[c] : 0;
            s->fuel[b * HW + c] = f != 0 ? cfg->initial_fuel : 0;
        }
        for (int32_t a = 0; a < A; ++a) {
            s->suppressants[b * A + a] = cfg->initial_suppressant;
            s->capacity[b * A + a] = cfg->initial_capacity;
            s->equipment[b * A + a] = cfg->initial_equipment_state;
            s->rewards[(int64_t)a * B + b] = 0.0f;
            if (s->cumulative_rewards) s->cumulative_rewards[(int64_t)a * B + b] = 0.0f;
            s->terminations[(int64_t)a * B + b] = 0;
            s->truncations[(int64_t)a * B + b] = 0;
        }
        s->num_moves[b] = 0;
        s->num_burnouts[b] = 0;
        s->burnouts[b] = 0;
        s->putouts[b] = 0;
    }
    s->frozen[0] = s->frozen[1] = 0;
    return frz_oracle_wildfire_rebuild(cfg, s);
}

/* BatchedAECEnv.reset_batches (utils/env.py:162-189) + raw_env.reset_batches (wildfire.py:376-397) with the selection given as a mask:
 * for every selected env b — rewards, cumulative rewards, terminations, truncations zeroed, num_moves = 0 (utils/env.py:176-188);
 * state restored from the saved initial state (wildfire.py:383-384: here the configured one, what reset() saved), num_burnouts = 0
 * (wildfire.py:386); then update_observations / update_actions of the whole batch (wildfire.py:394-397).  mask NULL selects the
 * finished envs (all agents terminated or all truncated: utils/env.py:331-359).  `seeds` (nullable, in/out) += seed_increment for the
 * selected envs, modulo 2^32 (the generator.seed(..., partial_seeding) call of utils/env.py:170-174 with seed = old seed + increment). */
int frz_oracle_wildfire_reset_masked(const frz_wildfire_cfg* cfg, frz_oracle_wildfire_bufs* s, const uint8_t* mask, int32_t* seeds,
                                     int32_t seed_increment) {
    const int64_t B = cfg->parallel_envs;
    const int32_t HW = cfg->grid_height * cfg->grid_width, A = cfg->num_agents;
    for (int64_t b = 0; b < B; ++b) {
        int selected;
        if (mask) {
            selected = mask[b] != 0;
        } else {
            int all_term = 1, all_trunc = 1;
            for (int32_t a = 0; a < A; ++a) {
                all_term = all_term && s->terminations[(int64_t)a * B + b];
                all_trunc = all_trunc && s->truncations[(int64_t)a * B + b];
            }
            selected = all_term || all_trunc;
        }
        if (!selected) continue;
        if (seeds) seeds[b] = (int32_t)((uint32_t)seeds[b] + (uint32_t)seed_increment);
        for (int32_t c = 0; c < HW; ++c) {
            const int32_t type = cfg->fire_types[c];
            const int32_t f = cfg->lit[c] ? type : -type;
            s->fires[b * HW + c] = f;
            s->intensity[b * HW + c] = cfg->lit[c] ? cfg->ignition_temp[c] : 0;
            s->fuel[b * HW + c] = f != 0 ? cfg->initial_fuel : 0;
        }
        for (int32_t a = 0; a < A; ++a) {
            s->suppressants[b * A + a] = cfg->initial_suppressant;
            s->capacity[b * A + a] = cfg->initial_capacity;
            s->equipment[b * A + a] = cfg->initial_equipment_state;
            s->rewards[(int64_t)a * B + b] = 0.0f;
            if (s->cumulative_rewards) s->cumulative_rewards[(int64_t)a * B + b] = 0.0f;
            s->terminations[(int64_t)a * B + b] = 0;
            s->truncations[(int64_t)a * B + b] = 0;
        }
        s->num_moves[b] = 0;
        s->num_burnouts[b] = 0;
    }
    s->frozen[0] = s->frozen[1] = 0;
    return frz_oracle_wildfire_rebuild(cfg, s);
}

/* One ParallelEnv.step(): utils/conversions.py:59-99 -> utils/env.py:203-242 -> wildfire.py:399-584 -> rebuild */
int frz_oracle_wildfire_step(const frz_wildfire_cfg* cfg, frz_oracle_wildfire_bufs* s, const int32_t* actions,
                             const float* field_randomness, const float* agent_randomness) {
    const int64_t B = cfg->parallel_envs;
    const int32_t H = cfg->grid_height, W = cfg->grid_width, HW = H * W, A = cfg->num_agents;
    const int64_t cap = B * HW;

    /* utils/env.py:211-213 — once ALL envs are terminated or ALL are truncated (for the selected = first agent),
     * every per-agent step() returns early: nothing changes, aec rewards stay stale, and the parallel adapter
     * (utils/conversions.py:87-90) sums the stale rewards once per agent call => A x stale, accumulated by addition. */
    {
        int all_term = 1, all_trunc = 1;
        for (int64_t b = 0; b < B; ++b) {
            all_term = all_term && s->terminations[b];
            all_trunc = all_trunc && s->truncations[b];
        }
        if (s->global_totals) { /* this batch is a shard: "all" ranges over the whole batch */
            all_term = s->global_totals[A + 1] == 0;
            all_trunc = s->global_totals[A + 2] == 0;
        }
        if (all_term || all_trunc) {
            if (!s->frozen[1]) {
                for (int64_t i = 0; i < (int64_t)A * B; ++i) {
                    float acc = 0.0f;
                    for (int32_t k = 0; k < A; ++k) acc = acc + s->rewards[i];
                    s->rewards[i] = acc;
                }
                s->frozen[1] = 1;
            }
            s->frozen[0] = 1;
            return FRZ_OK;
        }
    }

    float* attack = (float*)calloc((size_t)(B * HW), sizeof(float));
    uint8_t* users = (uint8_t*)calloc((size_t)(B * A), 1);   /* [B][A] (already transposed, :484-485) */
    uint8_t* refills = (uint8_t*)calloc((size_t)(B * A), 1); /* [B][A] */
    uint8_t* increased = (uint8_t*)calloc((size_t)(B * A), 1);
    int32_t* last_hit = (int32_t*)malloc((size_t)(B * A) * sizeof(int32_t)); /* cell hit by (b, a) or -1 (:481-483) */
    uint8_t* burned = (uint8_t*)calloc((size_t)(B * HW), 1);
    uint8_t* put_out = (uint8_t*)calloc((size_t)(B * HW), 1);
    float* tmp = (float*)malloc((size_t)(B * (HW > A ? HW : A)) * sizeof(float));
    for (int64_t i = 0; i < B * A; ++i) last_hit[i] = -1;
    for (int64_t i = 0; i < (int64_t)A * B; ++i) s->rewards[i] = 0.0f;

    /* (3) wildfire.py:427-483 — decode each agent's action through its stored action mapping (from the last rebuild).
     * Quirk (:434-435): an agent whose attackable-task count is 0 in EVERY env of the batch is skipped after its
     * refills are recorded: no fight, no bad-attack penalty.  With show_bad_actions the agent can still pick a listed
     * (bad) task, so the skip is observable and is restated here; it is a batch-global condition.
     * A task index outside the agent's mapping makes the reference raise or read garbage; here it is flagged
     * (FRZ_ERR_BAD_ACTION_INDEX) and treated as a bad action. */
    for (int32_t a = 0; a < A; ++a) {
        int64_t agent_tasks_anywhere = 0;
        for (int64_t b = 0; b < B; ++b) agent_tasks_anywhere += s->agent_task_count[(int64_t)a * B + b];
        if (s->global_totals) agent_tasks_anywhere = s->global_totals[1 + a];
        for (int64_t b = 0; b < B; ++b) {
            const int32_t idx = actions[((int64_t)a * B + b) * 2 + 0];
            const int32_t act = actions[((int64_t)a * B + b) * 2 + 1];
            const int refill = act == -1;
            refills[b * A + a] = (uint8_t)refill;
            if (refill || agent_tasks_anywhere == 0) continue;
            const int64_t* moff = cfg->show_bad_actions ? s->task_offsets : s->act_map_offsets + (int64_t)a * (B + 1);
            const int64_t* mval = cfg->show_bad_actions ? s->obs_map_values : s->act_map_values + (int64_t)a * cap;
            const int64_t n = moff[b + 1] - moff[b];
            int good = 0;
            int32_t cell = -1;
            if (idx < 0 || idx >= n) {
                s->error_flags[0] |= FRZ_ERR_BAD_ACTION_INDEX;
            } else {
                const int64_t t = mval[moff[b] + idx]; /* local task index */
                const int64_t* row = s->task_values + (s->task_offsets[b] + t) * 4;
                cell = (int32_t)(row[0] * W + row[1]);
                good = 1;
                if (cfg->show_bad_actions) { /* :464-468 — attacked index listed in agent_bad_actions */
                    const int64_t* bo = s->bad_map_offsets + (int64_t)a * (B + 1);
                    const int64_t* bv = s->bad_map_values + (int64_t)a * cap;
                    for (int64_t j = bo[b]; j < bo[b + 1]; ++j)
                        if (bv[j] == idx) good = 0;
                }
            }
            if (good) {
                const float power = cfg->fire_reduction_power[a] + cfg->equipment_states[s->equipment[b * A + a]][1];
                attack[b * HW + cell] = attack[b * HW + cell] + power; /* agents accumulate in index order */
                users[b * A + a] = 1;
                last_hit[b * A + a] = cell;
            } else {
                s->rewards[(int64_t)a * B + b] = cfg->bad_attack_penalty; /* assignment, :477 */
            }
        }
    }

    /* (4) transitions in the reference's order (:489-532); randomness is [events][B][...] */
    const float* ar = agent_randomness;
    const float* fr = field_randomness;
    const int64_t BA = B * A, BHW = B * HW;
    frz_oracle_wf_suppressant_decrease(cfg, s->suppressants, users, ar + 0 * BA, BA);
    frz_oracle_wf_equipment(cfg, s->equipment, ar + 1 * BA, BA);
    frz_oracle_wf_suppressant_refill(cfg, s->suppressants, s->capacity, s->equipment, refills, ar + 2 * BA, increased, BA);
    frz_oracle_wf_capacity(cfg, s->suppressants, s->capacity, increased, ar + 3 * BA, ar + 4 * BA, BA);
    frz_oracle_wf_fire_increase(cfg, s->fires, s->intensity, s->fuel, attack, fr + 0 * BHW, burned, BHW);
    frz_oracle_wf_fire_decrease(cfg, s->fires, s->intensity, s->fuel, attack, fr + 1 * BHW, put_out, BHW);
    frz_oracle_wf_fire_spread(cfg, s->fires, s->intensity, s->fuel, fr + 2 * BHW, B);

    /* (5)(6) rewards and termination (:534-582) */
    for (int64_t b = 0; b < B; ++b) {
        float fire_reward_sum = 0.0f, burnout_total = 0.0f;
        int32_t n_burn = 0, n_put = 0;
        for (int32_t c = 0; c < HW; ++c) {
            if (put_out[b * HW + c]) {
                fire_reward_sum = fire_reward_sum + cfg->fire_rewards[c];
                ++n_put;
            }
            if (burned[b * HW + c]) {
                burnout_total = burnout_total + (cfg->burnout_penalty_scaled ? -1.0f * cfg->fire_rewards[c] : cfg->burnout_penalty);
                ++n_burn;
            }
        }
        for (int32_t a = 0; a < A; ++a) {
            float add;
            if (cfg->localize_putouts) { /* :546-553 — only put-outs this agent last hit */
                const int32_t cell = last_hit[b * A + a];
                const float mine = (cell >= 0 && put_out[b * HW + cell]) ? cfg->fire_rewards[cell] : 0.0f;
                add = mine + burnout_total;
            } else {
                add = fire_reward_sum + burnout_total;
            }
            s->rewards[(int64_t)a * B + b] = s->rewards[(int64_t)a * B + b] + add;
        }
        int32_t fmax = s->fires[b * HW];
        int64_t fuel_sum = 0;
        for (int32_t c = 0; c < HW; ++c) {
            if (s->fires[b * HW + c] > fmax) fmax = s->fires[b * HW + c];
            fuel_sum += s->fuel[b * HW + c];
        }
        int dead = fmax <= 0;
        if (cfg->use_fire_fuel) dead = dead && fuel_sum <= 0;
        if (dead)
            for (int32_t c = 0; c < HW; ++c) s->fires[b * HW + c] = 0; /* :570 */
        int terminated = 1; /* self.terminated = all agents' terminations (utils/env.py:343-350) */
        for (int32_t a = 0; a < A; ++a) terminated = terminated && s->terminations[(int64_t)a * B + b];
        const int newly = !terminated && dead;
        /* termination_reward - kappa*log(num_burnouts + 1) clamped at 0, with num_burnouts BEFORE this step (:573-580) */
        /* torch.log(float32) is a <=1-ulp vectorised logf; the restatement (and the HIP kernel) round a double log to
         * float32 so that both agree bit for bit with each other (and within 1 ulp of the reference) */
        const float penalty = cfg->termination_kappa * (float)log((double)s->num_burnouts[b] + 1.0);
        float term_reward = cfg->termination_reward - penalty;
        term_reward = term_reward < 0.0f ? 0.0f : term_reward;
        for (int32_t a = 0; a < A; ++a) {
            if (newly) s->rewards[(int64_t)a * B + b] = s->rewards[(int64_t)a * B + b] + term_reward;
            s->terminations[(int64_t)a * B + b] = (uint8_t)(s->terminations[(int64_t)a * B + b] | dead);
        }
        s->num_burnouts[b] += n_burn;
        s->burnouts[b] = n_burn;
        s->putouts[b] = n_put;
        /* utils/env.py:228-235 */
        s->num_moves[b] += 1;
        for (int32_t a = 0; a < A; ++a) {
            if (cfg->max_steps >= 0) s->truncations[(int64_t)a * B + b] = s->num_moves[b] >= cfg->max_steps;
            if (s->cumulative_rewards && cfg->track_cumulative_rewards)
                s->cumulative_rewards[(int64_t)a * B + b] = s->cumulative_rewards[(int64_t)a * B + b] + s->rewards[(int64_t)a * B + b];
        }
    }
    free(attack);
    free(users);
    free(refills);
    free(increased);
    free(last_hit);
    free(burned);
    free(put_out);
    free(tmp);
    return frz_oracle_wildfire_rebuild(cfg, s);
}
