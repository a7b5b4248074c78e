/*
 * frz_oracle_rng.c — CPU restatement of the random streams.  TEST INFRASTRUCTURE ONLY (see frz_oracle.h).
 *
 * MT19937: the reference's RandomGenerator (utils/random_generator.py:49-146) keeps one torch CPU generator state
 * per env; torch's CPU generator is the standard MT19937 (Matsumoto & Nishimura 1998, init_genrand(seed)) and
 * torch.rand(float32) maps ONE 32-bit output x per element to (x & 0xFFFFFF) * 2^-24, elements in row-major order.
 * Pinned by tests/golden/mt19937_torch.npz (torch.Generator().manual_seed(s); torch.rand(n, generator=g)) and by the
 * published first outputs of init_genrand(5489): 3499211612, 581869302, 3890346734.
 *
 * Philox4x32-10: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11); pinned by the
 * Random123 known-answer vectors in tests/test_oracle_rng.py.
 */
#include "frz_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

#define MT_N 624
#define MT_M 397

void frz_oracle_mt19937_seed(uint32_t* mt_state, int32_t* mt_index, const int32_t* seeds, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        uint32_t* mt = mt_state + b * MT_N;
        mt[0] = (uint32_t)seeds[b];
        for (int j = 1; j < MT_N; ++j) mt[j] = 1812433253u * (mt[j - 1] ^ (mt[j - 1] >> 30)) + (uint32_t)j;
        mt_index[b] = 0; /* number of outputs consumed from the NEXT generation, lazily twisted word by word */
    }
}

/* One output.  The block twist of the textbook generator updates mt[] in place for i = 0..623 in order; doing
 * the same update for word i just before it is read (same order, same operands) yields the identical stream. */
static inline uint32_t mt_next(uint32_t* mt, int32_t* index) {
    const int i = *index;
    const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % MT_N] & 0x7fffffffu);
    uint32_t v = mt[(i + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    mt[i] = v;
    *index = (i + 1) % MT_N;
    v ^= v >> 11;
    v ^= (v << 7) & 0x9d2c5680u;
    v ^= (v << 15) & 0xefc60000u;
    v ^= v >> 18;
    return v;
}

/* RandomGenerator.generate(), unbuffered multi-seed branch (utils/random_generator.py:106-114): env b draws
 * events*count consecutive floats; output is transposed to [events][B][count]. */
void frz_oracle_mt19937_generate(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events, int64_t count, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        uint32_t* mt = mt_state + b * MT_N;
        for (int64_t e = 0; e < events; ++e)
            for (int64_t k = 0; k < count; ++k) {
                const uint32_t x = mt_next(mt, &mt_index[b]);
                out[(e * B + b) * count + k] = (float)(x & 0xFFFFFFu) * (1.0f / 16777216.0f);
            }
    }
}

void frz_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* FRZ_RNG_PHILOX stream definition: key = (seed, "FRZ" tag), counter = (draw / 4, step, stream, 0);
 * float = (word >> 8) * 2^-24 with word = output[draw % 4]. */
float frz_oracle_philox_uniform(int32_t seed, uint32_t step, uint32_t draw, uint32_t stream) {
    const uint32_t ctr[4] = {draw >> 2, step, stream, 0u};
    const uint32_t key[2] = {(uint32_t)seed, 0x46525A00u};
    uint32_t out[4];
    frz_oracle_philox4x32_10(ctr, key, out);
    return (float)(out[draw & 3u] >> 8) * (1.0f / 16777216.0f);
}

/* The randomness tensors a FRZ_RNG_PHILOX wildfire step consumes (include/frz.h), step = num_moves before the step,
 * key = (seed, 0x46525A00).  A 128-bit Philox block (word 0 least significant) is read as five 24-bit uniforms:
 *   draw u of the step = bits [24k, 24k + 24) of block (u / 5, step, 0, 0), k = u % 5, float = field * 2^-24
 *   field event e of cell c  = draw e * HW + c            agent event e of agent a = draw 3 * HW + e * A + a */
static float philox_draw24(int32_t seed, uint32_t step, uint32_t u) {
    const uint32_t ctr[4] = {u / 5u, step, 0u, 0u};
    const uint32_t key[2] = {(uint32_t)seed, 0x46525A00u};
    uint32_t out[4];
    frz_oracle_philox4x32_10(ctr, key, out);
    const uint32_t sh = 24u * (u % 5u);
    const uint64_t lo = (uint64_t)out[0] | ((uint64_t)out[1] << 32), hi = (uint64_t)out[2] | ((uint64_t)out[3] << 32);
    uint64_t bits;
    if (sh < 64u) {
        bits = lo >> sh;
        if (sh > 40u) bits |= hi << (64u - sh);
    } else {
        bits = hi >> (sh - 64u);
    }
    return (float)(uint32_t)(bits & 0xFFFFFFull) * (1.0f / 16777216.0f);
}

void frz_oracle_wildfire_philox_randomness(const frz_wildfire_cfg* cfg, const int32_t* seeds, const int32_t* num_moves, float* field,
                                           float* agent) {
    const int64_t B = cfg->parallel_envs;
    const int32_t HW = cfg->grid_height * cfg->grid_width, A = cfg->num_agents;
    for (int64_t b = 0; b < B; ++b) {
        const uint32_t step = (uint32_t)num_moves[b];
        for (int32_t e = 0; e < 3; ++e)
            for (int32_t c = 0; c < HW; ++c) field[(e * B + b) * HW + c] = philox_draw24(seeds[b], step, (uint32_t)(e * HW + c));
        for (int32_t e = 0; e < 5; ++e)
            for (int32_t a = 0; a < A; ++a) agent[(e * B + b) * A + a] = philox_draw24(seeds[b], step, (uint32_t)(3 * HW + e * A + a));
    }
}

/* Uniform random policy over OneOf([task]*n + [noop]) (spaces/actions.py:23-41): member j = floor(u32 * (n+1) / 2^32)
 * with u32 = word agent % 4 of Philox(counter = (agent / 4, 0, step lo, step hi), key = (seed lo ^ env seed, seed hi)). */
void frz_oracle_wildfire_random_policy(const frz_wildfire_cfg* cfg, const int32_t* agent_task_count, const int64_t* env_task_count,
                                       const int32_t* env_seeds, uint64_t seed, uint64_t step, int32_t* actions) {
    const int64_t B = cfg->parallel_envs;
    for (int64_t i = 0; i < (int64_t)cfg->num_agents * B; ++i) {
        const int32_t n = cfg->show_bad_actions ? (int32_t)env_task_count[i % B] : agent_task_count[i];
        const uint32_t agent = (uint32_t)(i / B); /* one block serves four agents: agent a draws word a % 4 of block a / 4 */
        const uint32_t ctr[4] = {agent >> 2, 0u, (uint32_t)step, (uint32_t)(step >> 32)};
        const uint32_t key[2] = {(uint32_t)seed ^ (uint32_t)env_seeds[i % B], (uint32_t)(seed >> 32)};
        uint32_t out[4];
        frz_oracle_philox4x32_10(ctr, key, out);
        const int32_t j = (int32_t)(((uint64_t)out[agent & 3u] * (uint64_t)(n + 1)) >> 32);
        actions[i * 2 + 0] = j < n ? j : n;
        actions[i * 2 + 1] = j < n ? 0 : -1;
    }
}

/* envs/wildfire/baselines/strongest.py:32-62 / weakest.py: the scripted "always fight the strongest (weakest) fire" agents.
 * observation = (obs, {'agent_action_mapping': mapping}); candidates = intensity (task column 3) of the env's FIRST len(mapping[b])
 * task rows (:47-48 index the padded task tensor by position); empty mapping everywhere -> all [-1, -1] (:41-43); empty in
 * this env -> [-1, -1] (:50-52); ties broken uniformly (:54-57; the reference uses torch's global generator, the build's
 * stream is word 0 of Philox(counter (first_env + b, 0, step lo, step hi), key (seed lo, seed hi)), member
 * floor(u32 * ties / 2^32)); no suppressant (self[:, 3] == 0) -> second component -1 (:62). */
void frz_oracle_wildfire_extreme_policy(const int64_t* task_values, const int64_t* task_offsets, const int64_t* map_offsets,
                                        const int64_t* map_lengths, const float* obs_self, int64_t B, int weakest, uint64_t seed, uint64_t step,
                                        int64_t first_env, int32_t* actions) {
    const int any_mapping = map_offsets[B] - map_offsets[0] > 0;
    for (int64_t b = 0; b < B; ++b) {
        int32_t idx = -1, act = -1;
        const int64_t n = map_lengths[b];
        if (any_mapping && n > 0) {
            const int64_t* rows = task_values + task_offsets[b] * 4;
            int64_t best = rows[3];
            for (int64_t k = 1; k < n; ++k) {
                const int64_t v = rows[k * 4 + 3];
                if (weakest ? v < best : v > best) best = v;
            }
            int64_t ties = 0;
            for (int64_t k = 0; k < n; ++k) ties += rows[k * 4 + 3] == best;
            const uint32_t ctr[4] = {(uint32_t)(b + first_env), 0u, (uint32_t)step, (uint32_t)(step >> 32)};
            const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
            uint32_t out[4];
            frz_oracle_philox4x32_10(ctr, key, out);
            int64_t pick = (int64_t)(((uint64_t)out[0] * (uint64_t)ties) >> 32);
            for (int64_t k = 0; k < n; ++k)
                if (rows[k * 4 + 3] == best && pick-- == 0) {
                    idx = (int32_t)k;
                    break;
                }
            act = 0;
        }
        if (any_mapping && obs_self[b * 4 + 3] == 0.0f) act = -1;
        actions[b * 2 + 0] = idx;
        actions[b * 2 + 1] = act;
    }
}


/* envs/rideshare/baselines/{greedy_Tfocus,greedy_Tglobal,fifo_Tfocus,fifo_Tglobal}.py, observe(), reproduced as written.
 * kind 0 greedy_Tfocus (:48-115), 1 greedy_Tglobal (:48-99), 2 fifo_Tfocus (:36-80), 3 fifo_Tglobal (:36-74).
 * Task rows are the observation's (y, x, y_dest, x_dest, accepted_by, riding_by, fare, entered_step) (rideshare.py:397-467).
 *   - no task in any env -> [-1, -1] everywhere; no mapped task in this env -> [-1, -1];
 *   - candidates are the env's FIRST n task rows, n = len(agent_action_mapping[b]);
 *   - greedy key: trip length + my distance to the passenger, float32; the agents build MovementTransition with fast_travel=True,
 *     so distance() (transitions/movement.py:56-86) is |dy| + |dx| on a 4-connected grid and sqrt(dy^2 + dx^2) with diagonals;
 *     greedy_Tglobal stores the key into an int64 buffer (empty_like of the mapping): truncated toward zero;
 *   - fifo key: entered_step as float32;
 *   - the Tfocus agents overwrite the key of every row that is not accepted, when the env shows an accepted row, and then of
 *     every row that is not riding, when it shows a riding row, with FLT_MAX (greedy) / +inf (fifo): an observation holding both
 *     kinds ends with every key equal (greedy_Tfocus asserts on such observations and on two accepted rows; this restatement
 *     carries on with the arithmetic);
 *   - the answer is drawn uniformly among the rows whose key equals the minimum (torch.randint on the global generator in the
 *     reference; here word 0 of Philox(counter (first_env + b, 0, step lo, step hi), key (seed lo, seed hi)), member
 *     floor(u32 * ties / 2^32), or forced_pick[b] when given, which is how the recorded reference draws are replayed);
 *   - second component by the chosen row: riding 2, accepted 1, otherwise 0. */
void frz_oracle_rideshare_task_policy(const int32_t* task_values, const int64_t* task_offsets, const int64_t* task_lengths,
                                      const int64_t* map_lengths, const int32_t* obs_self, int64_t B, int kind, int diagonal, uint64_t seed,
                                      uint64_t step, int64_t first_env, const int64_t* forced_pick, int64_t* ties_out, int32_t* actions) {
    int64_t total = 0;
    for (int64_t b = 0; b < B; ++b) total += task_lengths[b];
    const int greedy = kind < 2, focus = (kind & 1) == 0;
    const float masked = greedy ? FLT_MAX : INFINITY;
    for (int64_t b = 0; b < B; ++b) {
        int32_t idx = -1, act = -1;
        int64_t ties = 0;
        const int64_t count = task_lengths[b];
        const int64_t n = map_lengths[b] < count ? map_lengths[b] : count;
        if (total > 0 && n > 0) {
            const int32_t* rows = task_values + task_offsets[b] * 8;
            int any_accepted = 0, any_riding = 0;
            for (int64_t k = 0; k < count; ++k) {
                any_accepted |= rows[k * 8 + 4] >= 0;
                any_riding |= rows[k * 8 + 5] >= 0;
            }
            float* keys = (float*)malloc((size_t)n * sizeof(float));
            for (int64_t k = 0; k < n; ++k) {
                const int32_t* r = rows + k * 8;
                float key;
                if (greedy) {
                    const float ty = (float)(r[2] - r[0]), tx = (float)(r[3] - r[1]);
                    const float my = (float)(r[0] - obs_self[b * 4 + 0]), mx = (float)(r[1] - obs_self[b * 4 + 1]);
                    const float trip = diagonal ? sqrtf(ty * ty + tx * tx) : fabsf(ty) + fabsf(tx);
                    const float mine = diagonal ? sqrtf(my * my + mx * mx) : fabsf(my) + fabsf(mx);
                    key = trip + mine;
                } else {
                    key = (float)r[7];
                }
                if (focus) {
                    if (any_accepted && !(r[4] >= 0)) key = masked;
                    if (any_riding && !(r[5] >= 0)) key = masked;
                }
                if (kind == 1) key = (float)(int64_t)key;
                keys[k] = key;
            }
            float best = keys[0];
            for (int64_t k = 1; k < n; ++k)
                if (keys[k] < best) best = keys[k];
            for (int64_t k = 0; k < n; ++k) ties += keys[k] == best;
            int64_t pick;
            if (forced_pick) {
                pick = forced_pick[b];
            } else {
                const uint32_t ctr[4] = {(uint32_t)(b + first_env), 0u, (uint32_t)step, (uint32_t)(step >> 32)};
                const uint32_t key2[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
                uint32_t out[4];
                frz_oracle_philox4x32_10(ctr, key2, out);
                pick = (int64_t)(((uint64_t)out[0] * (uint64_t)ties) >> 32);
            }
            for (int64_t k = 0; k < n; ++k)
                if (keys[k] == best && pick-- == 0) {
                    idx = (int32_t)k;
                    break;
                }
            free(keys);
            if (idx >= 0) act = rows[idx * 8 + 5] >= 0 ? 2 : (rows[idx * 8 + 4] >= 0 ? 1 : 0);
        }
        if (ties_out) ties_out[b] = ties;
        actions[b * 2 + 0] = idx;
        actions[b * 2 + 1] = act;
    }
}

/* envs/cybersecurity/baselines/patched.py:33-71 (kind 0 PatchedAttackerBaseline), exploited.py:39-88 (1 ExploitedAttackerBaseline),
 * patched.py:96-152 (2 PatchedDefenderBaseline), exploited.py:120-167 (3 ExploitedDefenderBaseline), camp.py:32-60
 * (4 CampDefenderBaseline), observe(), reproduced as written.  The agents are stateful: target_node / time_focused / actions
 * persist between calls (int32 [B], [B], [B][2]).
 *   - an action mapping without any element (the agent is absent everywhere) answers [-100, -1] and leaves the state alone;
 *   - the candidates are `observation['tasks'][:, 0]`: for env b the row_len values tasks[b * env_stride + k * elem_stride].
 *     On the env's [B, N, F] task observation that is the F features of node 0 (env_stride N * F, elem_stride 1, row_len F),
 *     which is what the reference computes; (env_stride N * F, elem_stride F, row_len N) is feature 0 of every node;
 *   - key: 0 the value, minimum; 1 the value with (subnetwork_states - 1) replaced by -100, maximum; 2 the value with -100
 *     and 0 replaced by 1000, minimum; 3 the value, maximum.  One uniform draw per env among the positions holding the
 *     extreme (torch.randint in the reference; forced_pick[b] replays it, else word 0 of Philox(counter (first_env + b, 0,
 *     step lo, step hi), key (seed lo, seed hi)) -> floor(u32 * ties / 2^32));
 *   - attackers take the new target when they have none (-1); defenders take it whenever the row holds no -100 ("the last
 *     action was a monitor");
 *   - absent = self[:, 1] == 0; attackers: present with a target -> [target, 0] and time_focused + 1; absent -> -1.
 *     defenders: present with a target -> 0 (move) when self[:, 2] != target else -2 (patch) and time_focused + 1;
 *     absent -> -1; present without a target and without a monitored row -> -3 (monitor);
 *   - time_focused reaching 3 clears the target and the counter;
 *   - camp: target = camp_target (agent index % nodes); not there -> 0, absent -> -1, there -> -2 (also when absent: the
 *     fills are applied in that order). */
void frz_oracle_cyber_focus_policy(const int64_t* tasks, int64_t env_stride, int64_t elem_stride, int32_t row_len, const float* obs_self,
                                   int32_t self_width, int64_t B, int kind, int32_t subnetwork_states, int32_t camp_target,
                                   int64_t mapping_numel, uint64_t seed, uint64_t step, int64_t first_env, const int64_t* forced_pick,
                                   int64_t* ties_out, int32_t* target_node, int32_t* time_focused, int32_t* actions) {
    for (int64_t b = 0; b < B; ++b) {
        if (ties_out) ties_out[b] = 0;
        if (mapping_numel == 0) {
            actions[b * 2 + 0] = -100;
            actions[b * 2 + 1] = -1;
            continue;
        }
        const float* me = obs_self + b * self_width;
        const int absent = me[1] == 0.0f;
        int32_t a1 = actions[b * 2 + 1];
        if (kind == 4) {
            const int at = me[2] == (float)camp_target;
            if (!at) a1 = 0;
            if (absent) a1 = -1;
            if (at) a1 = -2;
            actions[b * 2 + 0] = camp_target;
            actions[b * 2 + 1] = a1;
            continue;
        }
        const int64_t* row = tasks + b * env_stride;
        const int want_max = kind == 1 || kind == 3;
        int64_t best = 0, ties = 0;
        int monitored = 1;
        for (int32_t k = 0; k < row_len; ++k) {
            const int64_t x = row[k * elem_stride];
            monitored &= x != -100;
            const int64_t key = kind == 1 ? (x == subnetwork_states - 1 ? -100 : x) : (kind == 2 ? ((x == -100 || x == 0) ? 1000 : x) : x);
            if (k == 0 || (want_max ? key > best : key < best)) best = key, ties = 1;
            else if (key == best) ++ties;
        }
        int64_t pick;
        if (forced_pick) {
            pick = forced_pick[b];
        } else {
            const uint32_t ctr[4] = {(uint32_t)(b + first_env), 0u, (uint32_t)step, (uint32_t)(step >> 32)};
            const uint32_t key2[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
            uint32_t out[4];
            frz_oracle_philox4x32_10(ctr, key2, out);
            pick = (int64_t)(((uint64_t)out[0] * (uint64_t)ties) >> 32);
        }
        int32_t fresh = 0;
        for (int32_t k = 0; k < row_len; ++k) {
            const int64_t x = row[k * elem_stride];
            const int64_t key = kind == 1 ? (x == subnetwork_states - 1 ? -100 : x) : (kind == 2 ? ((x == -100 || x == 0) ? 1000 : x) : x);
            if (key == best && pick-- == 0) {
                fresh = k;
                break;
            }
        }
        if (ties_out) ties_out[b] = ties;
        int32_t target = target_node[b], focused = time_focused[b];
        const int defender = kind >= 2;
        if (defender ? monitored : target == -1) target = fresh;
        const int targeted = target != -1 && !absent, targetless = target == -1 && !absent;
        if (!defender) {
            if (targeted) a1 = 0;
            if (absent) a1 = -1;
            if (targeted) ++focused;
        } else {
            const int at = me[2] == (float)target;
            if (targeted && !at) a1 = 0;
            if (absent) a1 = -1;
            if (targeted && at) a1 = -2;
            if (targetless && !monitored) a1 = -3;
            if (targeted && at) ++focused;
        }
        actions[b * 2 + 0] = target;
        actions[b * 2 + 1] = a1;
        if (focused >= 3) target = -1, focused = 0;
        target_node[b] = target;
        time_focused[b] = focused;
    }
}

/* CPU-baseline aid (bench.py): n_steps random-policy steps of one env batch, entirely in C — per step the uniform random policy
 * (frz_oracle_wildfire_random_policy), the step's Philox randomness (frz_oracle_wildfire_philox_randomness) and
 * frz_oracle_wildfire_step: the loop tests/ and bench.py otherwise drive from Python, one call per episode so that many host
 * threads can each step their own shard without meeting on the interpreter lock.  actions int32 [A][B][2], field float32
 * [3][B][H*W], agent float32 [5][B][A]: caller-provided scratch.  Returns the number of steps taken, or a negative error. */
/* the cybersecurity twin: n_steps x (uniform random policy of spaces/actions.py:11-99 on the stream of frz_cybersecurity_random_policy, the
 * step's FRZ_RNG_PHILOX randomness, frz_oracle_cybersecurity_step) — the loop `env.step({a: action_space(a).sample_nested()})` of
 * baselines/random.py:20 that frz_cybersecurity_rollout enqueues as one launch */
int frz_oracle_cybersecurity_rollout(const frz_cybersecurity_cfg* cfg, frz_oracle_cybersecurity_bufs* s, const int32_t* env_seeds, uint64_t policy_seed,
                                     uint64_t first_step, int32_t n_steps, int32_t* actions, float* network, float* agent) {
    for (int32_t t = 0; t < n_steps; ++t) {
        frz_oracle_cybersecurity_random_policy(cfg, s->agent_task_count, s->location, env_seeds, policy_seed, first_step + (uint64_t)t, actions);
        frz_oracle_cybersecurity_philox_randomness(cfg, env_seeds, s->num_moves, network, agent);
        const int rc = frz_oracle_cybersecurity_step(cfg, s, actions, network, agent);
        if (rc != 0) return rc < 0 ? rc : -rc;
    }
    return n_steps;
}

int frz_oracle_wildfire_rollout(const frz_wildfire_cfg* cfg, frz_oracle_wildfire_bufs* s, const int32_t* env_seeds, uint64_t policy_seed,
                                uint64_t first_step, int32_t n_steps, int32_t* actions, float* field, float* agent) {
    for (int32_t t = 0; t < n_steps; ++t) {
        frz_oracle_wildfire_random_policy(cfg, s->agent_task_count, s->env_task_count, env_seeds, policy_seed, first_step + (uint64_t)t, actions);
        frz_oracle_wildfire_philox_randomness(cfg, env_seeds, s->num_moves, field, agent);
        const int rc = frz_oracle_wildfire_step(cfg, s, actions, field, agent);
        if (rc != 0) return rc < 0 ? rc : -rc;
    }
    return n_steps;
}
