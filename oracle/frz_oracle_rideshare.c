/*
 * frz_oracle_rideshare.c — scalar CPU restatement of the reference rideshare step path.
 * TEST INFRASTRUCTURE ONLY (see frz_oracle.h).  Build with -ffp-contract=off.
 * Reference lines are relative to /root/reference/free_range_zoo/envs/rideshare/env/.
 *
 * The reference keeps one global passenger table sorted (stably) by env; envs never interact, so each env's rows are
 * kept here as its own ordered list of max_passengers slots: columns (y, x, y_dest, x_dest, fare, state, driver,
 * entered_step, accepted_step, picked_step) = reference columns 1..10 (column 0, the env id, is implicit).
 */
#include "frz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { PY = 0, PX, PYD, PXD, PFARE, PSTATE, PDRIVER, PENTERED, PACCEPTED, PPICKED, PCOLS };
#define NONE (-100)

static int32_t* prow(frz_oracle_rideshare_bufs* s, const frz_rideshare_cfg* cfg, int64_t b, int slot) {
    return s->passengers + (b * cfg->max_passengers + slot) * PCOLS;
}

/* transitions/passenger_entry.py:24-72: schedule rows of this timestep for this env (or wildcard -1), in schedule order,
 * appended behind the env's existing passengers (stable sort by env, :69-70) */
static void passenger_entry(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, const int32_t* schedule, int64_t b, int32_t t) {
    for (int r = 0; r < cfg->schedule_rows; ++r) {
        const int32_t* row = schedule + r * 7;
        if (row[0] != t || !(row[1] == b || row[1] == -1)) continue;
        if (s->passenger_count[b] >= cfg->max_passengers) {
            s->error_flags[0] |= FRZ_ERR_OVERFLOW;
            continue;
        }
        int32_t* p = prow(s, cfg, b, s->passenger_count[b]++);
        p[PY] = row[2], p[PX] = row[3], p[PYD] = row[4], p[PXD] = row[5], p[PFARE] = row[6];
        p[PSTATE] = 0, p[PDRIVER] = -1, p[PENTERED] = t, p[PACCEPTED] = -1, p[PPICKED] = -1;
    }
}

static void task_row(const int32_t* p, int32_t* out) { /* rideshare.py:405-414 */
    out[0] = p[PY], out[1] = p[PX], out[2] = p[PYD], out[3] = p[PXD];
    out[4] = p[PSTATE] == 1 ? p[PDRIVER] : NONE;
    out[5] = p[PSTATE] == 2 ? p[PDRIVER] : NONE;
    out[6] = p[PFARE], out[7] = p[PENTERED];
}

/* rideshare.py:367-395 (update_actions) and :397-467 (update_observations) */
int frz_oracle_rideshare_rebuild(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s) {
    const int64_t B = cfg->parallel_envs, cap = B * cfg->max_passengers;
    const int A = cfg->num_agents;
    s->task_offsets[0] = 0;
    for (int a = 0; a < A; ++a) s->agent_offsets[(int64_t)a * (B + 1)] = 0;
    for (int64_t b = 0; b < B; ++b) {
        const int n = s->passenger_count[b];
        for (int k = 0; k < n; ++k) task_row(prow(s, cfg, b, k), s->task_values + (s->task_offsets[b] + k) * 8);
        s->task_offsets[b + 1] = s->task_offsets[b] + n;
        s->env_task_count[b] = n;
        for (int a = 0; a < A; ++a) {
            int64_t* off = s->agent_offsets + (int64_t)a * (B + 1);
            int visible = 0, accepted = 0, riding = 0;
            for (int k = 0; k < n; ++k) {
                const int32_t* p = prow(s, cfg, b, k);
                if (p[PSTATE] == 0 || p[PDRIVER] == a) { /* general or exclusive task (:378-380) */
                    const int64_t at = (int64_t)a * cap + off[b] + visible;
                    s->agent_map_values[at] = k;
                    s->agent_task_states[at] = p[PSTATE];
                    task_row(p, s->agent_task_values + at * 8);
                    ++visible;
                }
                accepted += p[PSTATE] == 1 && p[PDRIVER] == a;
                riding += p[PSTATE] == 2 && p[PDRIVER] == a;
            }
            off[b + 1] = off[b] + visible;
            s->agent_task_count[(int64_t)a * B + b] = visible;
            int32_t* self = s->obs_self + ((int64_t)a * B + b) * 4;
            self[0] = s->agents[(b * A + a) * 2 + 0], self[1] = s->agents[(b * A + a) * 2 + 1], self[2] = accepted, self[3] = riding;
        }
        for (int a = 0; a < A; ++a) { /* others = the other agents' self rows, in agent order (:458-463) */
            int32_t* others = s->obs_others + ((int64_t)a * B + b) * (A - 1) * 4;
            int j = 0;
            for (int o = 0; o < A; ++o) {
                if (o == a) continue;
                memcpy(others + j * 4, s->obs_self + ((int64_t)o * B + b) * 4, 4 * sizeof(int32_t));
                ++j;
            }
        }
    }
    return FRZ_OK;
}

/* rideshare.py:185-222 + utils/env.py:94-160 */
int frz_oracle_rideshare_reset(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, const int32_t* schedule) {
    const int64_t B = cfg->parallel_envs;
    const int A = cfg->num_agents;
    for (int64_t b = 0; b < B; ++b) {
        for (int a = 0; a < A; ++a) {
            s->agents[(b * A + a) * 2 + 0] = cfg->start_y[a];
            s->agents[(b * A + a) * 2 + 1] = cfg->start_x[a];
            s->rewards[(int64_t)a * B + b] = 0.0f;
            s->cumulative_rewards[(int64_t)a * B + b] = 0.0f;
            s->terminations[(int64_t)a * B + b] = 0;
            s->truncations[(int64_t)a * B + b] = 0;
        }
        s->num_moves[b] = 0;
        s->passenger_count[b] = 0;
        passenger_entry(cfg, s, schedule, b, 0);
    }
    s->frozen[0] = s->frozen[1] = 0;
    return frz_oracle_rideshare_rebuild(cfg, s);
}

/* transitions/movement.py:56-116 for one agent: the best of {stay, N, E, S, W(, NW, NE, SE, SW)} by Euclidean distance
 * to the goal (first minimum wins), or the whole displacement with fast travel; cost = L1 (5 directions) / L2 (9) */
void frz_oracle_rs_move(const frz_rideshare_cfg* cfg, const int32_t vec[4], int32_t move[2], float* cost) {
    static const int32_t dirs[9][2] = {{0, 0}, {-1, 0}, {0, 1}, {1, 0}, {0, -1}, {-1, -1}, {-1, 1}, {1, 1}, {1, -1}};
    const int ndirs = cfg->use_diagonal_travel ? 9 : 5;
    move[0] = move[1] = 0;
    if (cfg->use_fast_travel) {
        move[0] = vec[2] - vec[0];
        move[1] = vec[3] - vec[1];
    } else {
        float best = INFINITY;
        for (int k = 0; k < ndirs; ++k) {
            const float dy = (float)(vec[0] + dirs[k][0] - vec[2]), dx = (float)(vec[1] + dirs[k][1] - vec[3]);
            const float dist = sqrtf(dy * dy + dx * dx);
            if (dist < best) {
                best = dist;
                move[0] = dirs[k][0];
                move[1] = dirs[k][1];
            }
        }
    }
    if (vec[0] == NONE) move[0] = 0; /* best_moves[starts == -100] = 0, per component (:83) */
    if (vec[1] == NONE) move[1] = 0;
    const float my = (float)move[0], mx = (float)move[1];
    *cost = ndirs == 9 ? sqrtf(my * my + mx * mx) : fabsf(my) + fabsf(mx);
}

/* squared pre-move distance between the two points of a task vector (sqrt is monotonic, zero iff zero); returns 0 when the
 * vector is all -100, i.e. the reference's distance is +inf (passenger_state.py:48-49, passenger_exit.py:43-44) */
static int vec_dist2(const int32_t v[4], int64_t* d2) {
    const int64_t dy = (int64_t)v[0] - v[2], dx = (int64_t)v[1] - v[3];
    *d2 = dy * dy + dx * dx;
    return !(v[0] == NONE && v[1] == NONE && v[2] == NONE && v[3] == NONE);
}

/* transitions/passenger_state.py:48-98 for env b.  target[a] = slot of agent a's passenger inside the env (or NONE).  Distances
 * come from the PRE-move vectors.  Accept conflicts: while any passenger is claimed by more than one accepting agent, per env
 * only the closest of ALL contested agents keeps its claim (first index on ties) and every other contested agent loses;
 * uncontested accepts survive (:54-74). */
static void rs_passenger_state(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, int64_t b, int32_t nm, const int* accept,
                               const int* pick, const int* target, int32_t vec[][4]) {
    const int A = cfg->num_agents;
    int64_t dist2[FRZ_MAX_AGENTS];
    int has_vec[FRZ_MAX_AGENTS];
    for (int a = 0; a < A; ++a) has_vec[a] = vec_dist2(vec[a], &dist2[a]);
    int accept_target[FRZ_MAX_AGENTS];
    for (int a = 0; a < A; ++a) accept_target[a] = accept[a] ? target[a] : NONE;
    for (;;) {
        int contested[FRZ_MAX_AGENTS], any = 0, winner = 0;
        for (int a = 0; a < A; ++a) {
            contested[a] = 0;
            if (accept_target[a] == NONE) continue;
            for (int o = 0; o < A; ++o) contested[a] |= o != a && accept_target[o] == accept_target[a];
            any |= contested[a];
        }
        if (!any) break;
        int found = 0;
        for (int a = 0; a < A; ++a) /* argmin over the row with non-contested entries at +inf; row of all inf -> index 0 */
            if (contested[a] && has_vec[a] && (!found || dist2[a] < dist2[winner])) winner = a, found = 1;
        for (int a = 0; a < A; ++a)
            if (contested[a] && !(found && a == winner) && !(!found && a == 0)) accept_target[a] = NONE;
    }
    for (int a = 0; a < A; ++a)
        if (accept_target[a] != NONE) {
            int32_t* p = prow(s, cfg, b, accept_target[a]);
            p[PSTATE] = 1, p[PACCEPTED] = nm, p[PDRIVER] = a;
        }
    for (int a = 0; a < A; ++a)
        if (pick[a] && target[a] != NONE && has_vec[a] && dist2[a] == 0) { /* distance < 1e-6: the agent already stood on the passenger */
            int32_t* p = prow(s, cfg, b, target[a]);
            p[PSTATE] = 2, p[PPICKED] = nm;
        }
}

/* transitions/passenger_exit.py:22-56 for env b: drops succeed at distance 0; fares paid; rows removed in order */
static void rs_passenger_exit(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, int64_t b, const int* drop, const int* target,
                              int32_t vec[][4], int32_t* fares) {
    const int A = cfg->num_agents;
    uint8_t removed[FRZ_MAX_PASSENGERS];
    memset(removed, 0, sizeof(removed));
    for (int a = 0; a < A; ++a) {
        int64_t d2;
        const int has = vec_dist2(vec[a], &d2);
        fares[a] = 0;
        if (drop[a] && target[a] != NONE && has && d2 == 0) {
            fares[a] = prow(s, cfg, b, target[a])[PFARE];
            removed[target[a]] = 1;
        }
    }
    int kept = 0;
    for (int k = 0; k < s->passenger_count[b]; ++k)
        if (!removed[k]) {
            if (kept != k) memcpy(prow(s, cfg, b, kept), prow(s, cfg, b, k), PCOLS * sizeof(int32_t));
            ++kept;
        }
    s->passenger_count[b] = kept;
}

/* Single transitions on the slot table, for the known-answer vectors recorded from the reference's own transition tests
 * (tests/golden/ka_rideshare.npz): masks uint8 [B][A], targets int32 [B][A] = slot inside the env or -100, vectors int32 [B][A][4] */
int frz_oracle_rs_passenger_state(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, const uint8_t* accepts, const uint8_t* picks,
                                  const int32_t* targets, const int32_t* vectors, const int32_t* timesteps) {
    const int A = cfg->num_agents;
    for (int64_t b = 0; b < cfg->parallel_envs; ++b) {
        int accept[FRZ_MAX_AGENTS], pick[FRZ_MAX_AGENTS], target[FRZ_MAX_AGENTS];
        int32_t vec[FRZ_MAX_AGENTS][4];
        for (int a = 0; a < A; ++a) {
            accept[a] = accepts[b * A + a], pick[a] = picks[b * A + a], target[a] = targets[b * A + a];
            memcpy(vec[a], vectors + (b * A + a) * 4, sizeof(vec[a]));
        }
        rs_passenger_state(cfg, s, b, timesteps[b], accept, pick, target, vec);
    }
    return FRZ_OK;
}

int frz_oracle_rs_passenger_exit(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, const uint8_t* drops, const int32_t* targets,
                                 const int32_t* vectors, int32_t* fares_out) {
    const int A = cfg->num_agents;
    for (int64_t b = 0; b < cfg->parallel_envs; ++b) {
        int drop[FRZ_MAX_AGENTS], target[FRZ_MAX_AGENTS];
        int32_t vec[FRZ_MAX_AGENTS][4];
        for (int a = 0; a < A; ++a) {
            drop[a] = drops[b * A + a], target[a] = targets[b * A + a];
            memcpy(vec[a], vectors + (b * A + a) * 4, sizeof(vec[a]));
        }
        rs_passenger_exit(cfg, s, b, drop, target, vec, fares_out + b * A);
    }
    return FRZ_OK;
}

int frz_oracle_rs_passenger_entry(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, const int32_t* schedule, const int32_t* timesteps) {
    for (int64_t b = 0; b < cfg->parallel_envs; ++b) passenger_entry(cfg, s, schedule, b, timesteps[b]);
    return FRZ_OK;
}

/* One ParallelEnv.step(): utils/conversions.py:59-99 -> utils/env.py:203-242 -> rideshare.py:248-365 -> rebuild */
int frz_oracle_rideshare_step(const frz_rideshare_cfg* cfg, frz_oracle_rideshare_bufs* s, const int32_t* schedule, const int32_t* actions) {
    const int64_t B = cfg->parallel_envs, cap = B * cfg->max_passengers;
    const int A = cfg->num_agents;
    { /* utils/env.py:211-213 early-out + utils/conversions.py:87-90 (see the wildfire oracle) */
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
    for (int64_t b = 0; b < B; ++b) {
        const int32_t nm = s->num_moves[b];
        int noop[FRZ_MAX_AGENTS], accept[FRZ_MAX_AGENTS], pick[FRZ_MAX_AGENTS], drop[FRZ_MAX_AGENTS], target[FRZ_MAX_AGENTS];
        int32_t vec[FRZ_MAX_AGENTS][4], move[FRZ_MAX_AGENTS][2], fares[FRZ_MAX_AGENTS];
        float cost[FRZ_MAX_AGENTS];
        /* (1) rideshare.py:256-300 — action decode through the agent's stored action mapping, task vectors from the
         * state BEFORE movement */
        for (int a = 0; a < A; ++a) {
            const int32_t idx = actions[((int64_t)a * B + b) * 2 + 0], act = actions[((int64_t)a * B + b) * 2 + 1];
            noop[a] = act == -1, accept[a] = act == 0, pick[a] = act == 1, drop[a] = act == 2;
            target[a] = NONE;
            vec[a][0] = vec[a][1] = vec[a][2] = vec[a][3] = NONE;
            if (!noop[a]) {
                const int64_t* off = s->agent_offsets + (int64_t)a * (B + 1);
                if (idx < 0 || idx >= off[b + 1] - off[b]) {
                    s->error_flags[0] |= FRZ_ERR_BAD_ACTION_INDEX; /* the reference reads a garbage row here */
                    accept[a] = pick[a] = drop[a] = 0;
                } else {
                    target[a] = (int)s->agent_map_values[(int64_t)a * cap + off[b] + idx];
                }
            }
            if (accept[a] || pick[a] || drop[a]) {
                const int32_t* p = prow(s, cfg, b, target[a]);
                vec[a][0] = s->agents[(b * A + a) * 2 + 0], vec[a][1] = s->agents[(b * A + a) * 2 + 1];
                vec[a][2] = drop[a] ? p[PYD] : p[PY], vec[a][3] = drop[a] ? p[PXD] : p[PX];
            }
        }
        /* (2) movement: agents first, then riding passengers follow their driver (transitions/movement.py:98-114) */
        for (int a = 0; a < A; ++a) {
            frz_oracle_rs_move(cfg, vec[a], move[a], &cost[a]);
            s->agents[(b * A + a) * 2 + 0] += move[a][0];
            s->agents[(b * A + a) * 2 + 1] += move[a][1];
        }
        for (int k = 0; k < s->passenger_count[b]; ++k) {
            int32_t* p = prow(s, cfg, b, k);
            if (p[PSTATE] == 2) {
                /* best_moves[env, driver]: a riding passenger that was never accepted (driver -1, only reachable through a
                 * pick the action space does not offer) follows the LAST agent, as Python's negative index does (:107-108) */
                const int driver = p[PDRIVER] < 0 ? A + p[PDRIVER] : p[PDRIVER];
                p[PY] += move[driver][0];
                p[PX] += move[driver][1];
            }
        }
        /* (3) transitions/passenger_state.py:48-98 on the PRE-move vectors, then (4) transitions/passenger_exit.py:22-56 */
        rs_passenger_state(cfg, s, b, nm, accept, pick, target, vec);
        rs_passenger_exit(cfg, s, b, drop, target, vec, fares);
        /* (5) entry of the passengers scheduled for the NEXT timestep (rideshare.py:308) */
        passenger_entry(cfg, s, schedule, b, nm + 1);

        /* (6) rewards (rideshare.py:310-363) on the post-transition table, num_moves not yet incremented */
        const int n = s->passenger_count[b];
        float global = 0.0f;
        if (cfg->use_waiting_costs) {
            /* `global_rewards[envs] += cost` is an index_put WITHOUT accumulation: with several passengers of one env in the
             * index list only the LAST one (table order) takes effect, once per statement (:323-333) */
            int last[3] = {-1, -1, -1};
            int unaccepted = 0;
            for (int k = 0; k < n; ++k) {
                const int st = prow(s, cfg, b, k)[PSTATE];
                if (st >= 0 && st <= 2) last[st] = k;
                unaccepted += st == 0;
            }
            const int since[3] = {PENTERED, PACCEPTED, PPICKED};
            for (int st = 0; st < 3; ++st)
                if (last[st] >= 0) {
                    const int32_t wait = nm - prow(s, cfg, b, last[st])[since[st]];
                    global = global + (wait >= cfg->wait_limit[st] ? 1.0f : 0.0f) * cfg->general_wait_cost;
                }
            if (last[0] >= 0) {
                const int32_t wait = nm - prow(s, cfg, b, last[0])[PENTERED];
                global = global + (wait >= cfg->long_wait_time ? 1.0f : 0.0f) * cfg->long_wait_cost;
            }
            const int slots = A * cfg->pool_limit; /* :336-339 unserved cost */
            global = global + ((float)(unaccepted >= slots - n ? 1 : 0) * -0.5f) * (float)(slots - n);
        }
        for (int a = 0; a < A; ++a) {
            int accepted = 0;
            for (int k = 0; k < n; ++k) accepted += prow(s, cfg, b, k)[PDRIVER] == a;
            float r = 0.0f;
            r = r + (accepted > cfg->pool_limit ? cfg->pool_limit_cost : 0.0f);
            r = r + (float)noop[a] * cfg->noop_cost;
            r = r + (float)(actions[((int64_t)a * B + b) * 2 + 1] == 0) * cfg->accept_cost; /* accept ACTION, won or not */
            r = r + (fares[a] > 0 ? (float)fares[a] - cfg->drop_cost : 0.0f);
            float dr = cost[a] * cfg->move_cost;
            if (cfg->use_variable_move_cost) dr = dr / (float)(accepted + 1);
            r = r + dr;
            r = r + global;
            s->rewards[(int64_t)a * B + b] = r;
        }
        /* utils/env.py:228-235 */
        s->num_moves[b] += 1;
        for (int a = 0; a < A; ++a) {
            if (cfg->max_steps >= 0) s->truncations[(int64_t)a * B + b] = s->num_moves[b] >= cfg->max_steps;
            if (cfg->track_cumulative_rewards)
                s->cumulative_rewards[(int64_t)a * B + b] = s->cumulative_rewards[(int64_t)a * B + b] + s->rewards[(int64_t)a * B + b];
        }
    }
    return frz_oracle_rideshare_rebuild(cfg, s);
}

/* uniform member of OneOf([Discrete(1, start=state_t) for visible task t] + [noop]) (spaces/actions.py:10-50) */
void frz_oracle_rideshare_random_policy(const frz_rideshare_cfg* cfg, const frz_oracle_rideshare_bufs* s, uint64_t seed, uint64_t step,
                                        int32_t* actions) {
    const int64_t B = cfg->parallel_envs, cap = B * cfg->max_passengers;
    for (int64_t i = 0; i < (int64_t)cfg->num_agents * B; ++i) {
        const int a = (int)(i / B);
        const int64_t b = i % B;
        const int n = s->agent_task_count[i];
        const uint32_t ctr[4] = {(uint32_t)a, (uint32_t)(b + cfg->first_env_index), (uint32_t)step, (uint32_t)(step >> 32)};
        const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        uint32_t out[4];
        frz_oracle_philox4x32_10(ctr, key, out);
        const int j = (int)(((uint64_t)out[0] * (uint64_t)(n + 1)) >> 32);
        const int64_t off = s->agent_offsets[(int64_t)a * (B + 1) + b];
        actions[i * 2 + 0] = j;
        actions[i * 2 + 1] = j < n ? s->agent_task_states[(int64_t)a * cap + off + j] : -1;
    }
}
