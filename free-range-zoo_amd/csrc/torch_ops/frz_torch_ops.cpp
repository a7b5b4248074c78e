// frz_torch_ops.cpp — PyTorch custom ops (namespace `frz`) over the C-ABI of include/frz.h.
//
// The product is libfrz_hip.so behind plain C entry points; this shim only registers them with the PyTorch dispatcher
// (TORCH_LIBRARY schemas with mutable-alias annotations, implementation on the CUDA dispatch key, which is what ROCm builds of
// PyTorch use for HIP tensors), so that `torch.ops.frz.wildfire_step(...)` etc. exist as BASELINE.json's north star asks.  Every op
//   * takes the env's device arena (uint8 tensor, declared mutated: `Tensor(a!)`) and the opaque env handle frz_<domain>_create returned,
//   * launches on the CURRENT HIP stream of the arena's device, never synchronises, never allocates,
//   * raises (TORCH_CHECK) on shape / dtype / device mismatches and on negative FRZ_E_* codes.
// No kernel lives here; the ctypes binding (free-range-zoo_amd/_capi.py) remains the torch-free path to the same entry points.
#include <ATen/core/Tensor.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <cstring>

#include "../../../include/frz.h"

namespace {

void* current_stream(const at::Tensor& on) { return c10::hip::getCurrentHIPStream(on.device().index()).stream(); }

void ok(int code, const char* what) { TORCH_CHECK(code == FRZ_OK, what, " failed with FRZ code ", code); }

void check_arena(const at::Tensor& arena) {
    TORCH_CHECK(arena.is_cuda() && arena.scalar_type() == at::kByte && arena.is_contiguous(), "arena must be a contiguous uint8 tensor on the GPU");
}
void inside(const at::Tensor& arena, const void* p, const char* what) {
    const char* lo = static_cast<const char*>(arena.data_ptr());
    TORCH_CHECK(static_cast<const char*>(p) >= lo && static_cast<const char*>(p) < lo + arena.numel(), what, ": the handle is not bound to this arena");
}
// Every op starts here: `handle` is a live handle of the expected domain (the library keeps a registry: a stale or foreign integer is
// refused, not dereferenced), it is bound to THIS arena, and the sizes the op works with are the handle's own — the agents / envs / units
// arguments of the schemas are checked against them, never trusted.  The guard makes the arena's device current for the launch.
struct Env {
    void* handle;
    int64_t A, B, U;  // agents, envs, units (cells / nodes / passenger slots per env)
    c10::hip::HIPGuardMasqueradingAsCUDA guard;
};
enum Kind { kWildfire = 1, kCybersecurity = 2, kRideshare = 3 };
Env checked(const at::Tensor& arena, int64_t handle, int kind, const char* what) {
    check_arena(arena);
    void* h = reinterpret_cast<void*>(handle);
    TORCH_CHECK(frz_handle_kind(h) == kind, what, ": not a live ", (kind == kWildfire ? "wildfire" : (kind == kCybersecurity ? "cybersecurity" : "rideshare")),
                " env handle of this library");
    int64_t A = 0, B = 0, U = 0;
    ok(frz_handle_shape(h, &A, &B, &U), "frz_handle_shape");
    const void* bound = nullptr;
    if (kind == kWildfire) {
        frz_wildfire_bufs bufs;
        ok(frz_wildfire_get_bufs(static_cast<frz_wildfire_env*>(h), &bufs), "frz_wildfire_get_bufs");
        bound = bufs.error_flags;
    } else if (kind == kCybersecurity) {
        frz_cybersecurity_bufs bufs;
        ok(frz_cybersecurity_get_bufs(static_cast<frz_cybersecurity_env*>(h), &bufs), "frz_cybersecurity_get_bufs");
        bound = bufs.error_flags;
    } else {
        frz_rideshare_bufs bufs;
        ok(frz_rideshare_get_bufs(static_cast<frz_rideshare_env*>(h), &bufs), "frz_rideshare_get_bufs");
        bound = bufs.error_flags;
    }
    inside(arena, bound, what);
    return Env{h, A, B, U, c10::hip::HIPGuardMasqueradingAsCUDA(arena.device())};
}
void same_sizes(const Env& e, int64_t agents, int64_t envs, int64_t units, const char* what) {
    TORCH_CHECK(agents == e.A && envs == e.B && (units < 0 || units == e.U), what, ": sizes (", agents, ", ", envs, ", ", units, ") are not the handle's (", e.A,
                ", ", e.B, ", ", e.U, ")");
}
const int32_t* actions_ptr(const at::Tensor& arena, const at::Tensor& actions, const Env& e) {
    TORCH_CHECK(actions.is_cuda() && actions.device() == arena.device() && actions.scalar_type() == at::kInt && actions.is_contiguous() &&
                    actions.numel() == e.A * e.B * 2,
                "actions must be a contiguous int32 [A, B, 2] tensor on the env's device");
    return actions.data_ptr<int32_t>();
}
const float* floats_or_null(const at::Tensor& arena, const c10::optional<at::Tensor>& t, int64_t numel, const char* what) {
    if (!t.has_value()) return nullptr;
    TORCH_CHECK(t->is_cuda() && t->device() == arena.device() && t->scalar_type() == at::kFloat && t->is_contiguous() && t->numel() == numel, what,
                ": expected a contiguous float32 tensor of ", numel, " elements on the env's device");
    return t->data_ptr<float>();
}

// ---------------------------------------------------------------------------------------------------- wildfire
void wildfire_reset(at::Tensor arena, int64_t handle) {
    const Env e = checked(arena, handle, kWildfire, "wildfire_reset");
    ok(frz_wildfire_reset(static_cast<frz_wildfire_env*>(e.handle), current_stream(arena)), "frz_wildfire_reset");
}
void wildfire_reset_reseed(at::Tensor arena, int64_t handle, int64_t seed_increment) {
    const Env e = checked(arena, handle, kWildfire, "wildfire_reset_reseed");
    ok(frz_wildfire_reset_reseed(static_cast<frz_wildfire_env*>(e.handle), (int32_t)seed_increment, current_stream(arena)), "frz_wildfire_reset_reseed");
}
void wildfire_rebuild(at::Tensor arena, int64_t handle) {
    const Env e = checked(arena, handle, kWildfire, "wildfire_rebuild");
    ok(frz_wildfire_rebuild(static_cast<frz_wildfire_env*>(e.handle), current_stream(arena)), "frz_wildfire_rebuild");
}
void wildfire_step(at::Tensor arena, int64_t handle, const at::Tensor& actions, int64_t rng_mode, const c10::optional<at::Tensor>& field_randomness,
                   const c10::optional<at::Tensor>& agent_randomness, int64_t agents, int64_t envs, int64_t cells) {
    const Env e = checked(arena, handle, kWildfire, "wildfire_step");
    same_sizes(e, agents, envs, cells, "wildfire_step");
    ok(frz_wildfire_step(static_cast<frz_wildfire_env*>(e.handle), actions_ptr(arena, actions, e), (int)rng_mode,
                         floats_or_null(arena, field_randomness, 3 * e.B * e.U, "field_randomness"),
                         floats_or_null(arena, agent_randomness, 5 * e.B * e.A, "agent_randomness"), current_stream(arena)),
       "frz_wildfire_step");
}
void wildfire_random_policy(const at::Tensor& arena, int64_t handle, int64_t policy_seed, int64_t policy_step, at::Tensor actions_out, int64_t agents,
                            int64_t envs) {
    const Env e = checked(arena, handle, kWildfire, "wildfire_random_policy");
    same_sizes(e, agents, envs, -1, "wildfire_random_policy");
    ok(frz_wildfire_random_policy(static_cast<frz_wildfire_env*>(e.handle), (uint64_t)policy_seed, (uint64_t)policy_step,
                                  const_cast<int32_t*>(actions_ptr(arena, actions_out, e)), current_stream(arena)),
       "frz_wildfire_random_policy");
}
void wildfire_step_random_policy(at::Tensor arena, int64_t handle, int64_t policy_seed, int64_t policy_step, at::Tensor actions_out, int64_t rng_mode,
                                 int64_t agents, int64_t envs) {
    const Env e = checked(arena, handle, kWildfire, "wildfire_step_random_policy");
    same_sizes(e, agents, envs, -1, "wildfire_step_random_policy");
    ok(frz_wildfire_step_random_policy(static_cast<frz_wildfire_env*>(e.handle), (uint64_t)policy_seed, (uint64_t)policy_step,
                                       const_cast<int32_t*>(actions_ptr(arena, actions_out, e)), (int)rng_mode, nullptr, nullptr, current_stream(arena)),
       "frz_wildfire_step_random_policy");
}

// ---------------------------------------------------------------------------------------------------- cybersecurity
void cybersecurity_reset(at::Tensor arena, int64_t handle) {
    const Env e = checked(arena, handle, kCybersecurity, "cybersecurity_reset");
    ok(frz_cybersecurity_reset(static_cast<frz_cybersecurity_env*>(e.handle), current_stream(arena)), "frz_cybersecurity_reset");
}
void cybersecurity_rebuild(at::Tensor arena, int64_t handle) {
    const Env e = checked(arena, handle, kCybersecurity, "cybersecurity_rebuild");
    ok(frz_cybersecurity_rebuild(static_cast<frz_cybersecurity_env*>(e.handle), current_stream(arena)), "frz_cybersecurity_rebuild");
}
void cybersecurity_step(at::Tensor arena, int64_t handle, const at::Tensor& actions, int64_t rng_mode, const c10::optional<at::Tensor>& network_randomness,
                        const c10::optional<at::Tensor>& agent_randomness, int64_t agents, int64_t envs, int64_t nodes) {
    const Env e = checked(arena, handle, kCybersecurity, "cybersecurity_step");
    same_sizes(e, agents, envs, nodes, "cybersecurity_step");
    ok(frz_cybersecurity_step(static_cast<frz_cybersecurity_env*>(e.handle), actions_ptr(arena, actions, e), (int)rng_mode,
                              floats_or_null(arena, network_randomness, e.B * e.U, "network_randomness"),
                              floats_or_null(arena, agent_randomness, e.B * e.A, "agent_randomness"), current_stream(arena)),
       "frz_cybersecurity_step");
}
void cybersecurity_random_policy(const at::Tensor& arena, int64_t handle, int64_t policy_seed, int64_t policy_step, at::Tensor actions_out, int64_t agents,
                                 int64_t envs) {
    const Env e = checked(arena, handle, kCybersecurity, "cybersecurity_random_policy");
    same_sizes(e, agents, envs, -1, "cybersecurity_random_policy");
    ok(frz_cybersecurity_random_policy(static_cast<frz_cybersecurity_env*>(e.handle), (uint64_t)policy_seed, (uint64_t)policy_step,
                                       const_cast<int32_t*>(actions_ptr(arena, actions_out, e)), current_stream(arena)),
       "frz_cybersecurity_random_policy");
}
void cybersecurity_step_random_policy(at::Tensor arena, int64_t handle, int64_t policy_seed, int64_t policy_step, at::Tensor actions_out, int64_t rng_mode,
                                      int64_t agents, int64_t envs) {
    const Env e = checked(arena, handle, kCybersecurity, "cybersecurity_step_random_policy");
    same_sizes(e, agents, envs, -1, "cybersecurity_step_random_policy");
    ok(frz_cybersecurity_step_random_policy(static_cast<frz_cybersecurity_env*>(e.handle), (uint64_t)policy_seed, (uint64_t)policy_step,
                                            const_cast<int32_t*>(actions_ptr(arena, actions_out, e)), (int)rng_mode, nullptr, nullptr,
                                            current_stream(arena)),
       "frz_cybersecurity_step_random_policy");
}

// ---------------------------------------------------------------------------------------------------- rollouts (frz_rollout_spec)
// n steps of the rollout loop as one op: every tensor of the spec is an argument (so the dispatcher sees what is read and what is
// written), checked for device / dtype / contiguity / size before its pointer goes into the spec.
template <typename T>
T* tape_or_null(const at::Tensor& arena, const c10::optional<at::Tensor>& t, at::ScalarType dtype, int64_t numel, const char* what) {
    if (!t.has_value()) return nullptr;
    TORCH_CHECK(t->is_cuda() && t->device() == arena.device() && t->scalar_type() == dtype && t->is_contiguous() && t->numel() == numel, what,
                ": expected a contiguous tensor of ", numel, " elements of the documented dtype on the env's device");
    return static_cast<T*>(t->data_ptr());
}
frz_rollout_spec make_spec(const at::Tensor& arena, const Env& e, int64_t units_a, int64_t units_b, int64_t list_block_bytes, int64_t steps, int64_t rng_mode,
                           int64_t flags, int64_t seed_increment, int64_t seed_stride, int64_t policy_seed, int64_t first_step,
                           const c10::optional<at::Tensor>& action_tape, const c10::optional<at::Tensor>& randomness_a,
                           const c10::optional<at::Tensor>& randomness_b, const c10::optional<at::Tensor>& actions_out, bool record_actions,
                           const c10::optional<at::Tensor>& reward_tape, const c10::optional<at::Tensor>& done_tape,
                           const c10::optional<at::Tensor>& list_record, const c10::optional<at::Tensor>& metrics) {
    TORCH_CHECK(steps >= 0, "rollout: steps must not be negative");  // (0 steps: a reset-only rollout, as through the C-ABI)
    frz_rollout_spec spec;
    std::memset(&spec, 0, sizeof(spec));
    spec.n_steps = (int32_t)steps;
    spec.rng_mode = (int32_t)rng_mode;
    spec.flags = (uint32_t)flags;
    spec.seed_increment = (int32_t)seed_increment;
    spec.seed_stride = (uint32_t)seed_stride;
    spec.policy_seed = (uint64_t)policy_seed;
    spec.first_step = (uint64_t)first_step;
    spec.action_tape = tape_or_null<const int32_t>(arena, action_tape, at::kInt, steps * e.A * e.B * 2, "action_tape");
    spec.randomness_tape_a = tape_or_null<const float>(arena, randomness_a, at::kFloat, steps * units_a, "randomness_a");
    spec.randomness_tape_b = tape_or_null<const float>(arena, randomness_b, at::kFloat, steps * units_b, "randomness_b");
    TORCH_CHECK(steps == 0 || rng_mode != FRZ_RNG_INJECTED || (spec.randomness_tape_a && spec.randomness_tape_b), "rollout: FRZ_RNG_INJECTED needs both randomness tapes");
    spec.record_actions = record_actions ? 1 : 0;
    spec.actions_out = tape_or_null<int32_t>(arena, actions_out, at::kInt, (record_actions ? steps : 1) * e.A * e.B * 2, "actions_out");
    TORCH_CHECK(spec.action_tape || spec.actions_out, "rollout: the in-kernel policy needs actions_out");
    spec.reward_tape = tape_or_null<float>(arena, reward_tape, at::kFloat, steps * e.A * e.B, "reward_tape");
    spec.done_tape = tape_or_null<uint8_t>(arena, done_tape, at::kByte, steps * 2 * e.B, "done_tape");
    spec.list_record = tape_or_null<uint8_t>(arena, list_record, at::kByte, (steps > 0 ? steps - 1 : 0) * list_block_bytes, "list_record");
    spec.metrics = tape_or_null<double>(arena, metrics, at::kDouble, e.A + 2, "metrics");
    return spec;
}
void wildfire_rollout(at::Tensor arena, int64_t handle, int64_t steps, int64_t rng_mode, int64_t flags, int64_t seed_increment, int64_t seed_stride,
                      int64_t policy_seed, int64_t first_step, const c10::optional<at::Tensor>& action_tape, const c10::optional<at::Tensor>& randomness_a,
                      const c10::optional<at::Tensor>& randomness_b, const c10::optional<at::Tensor>& actions_out, bool record_actions,
                      const c10::optional<at::Tensor>& reward_tape, const c10::optional<at::Tensor>& done_tape, const c10::optional<at::Tensor>& list_record,
                      const c10::optional<at::Tensor>& metrics, const c10::optional<at::Tensor>& obs_tape, const c10::optional<at::Tensor>& state_tape) {
    const Env e = checked(arena, handle, kWildfire, "wildfire_rollout");
    frz_wildfire_env* const env = static_cast<frz_wildfire_env*>(e.handle);
    void* block = nullptr;
    int64_t block_bytes = 0;
    ok(frz_wildfire_list_block(env, &block, &block_bytes), "frz_wildfire_list_block");
    frz_rollout_spec spec = make_spec(arena, e, 3 * e.B * e.U, 5 * e.B * e.A, block_bytes, steps, rng_mode, flags, seed_increment, seed_stride, policy_seed,
                                      first_step, action_tape, randomness_a, randomness_b, actions_out, record_actions, reward_tape, done_tape, list_record,
                                      metrics);
    {   // observation / state tapes: byte tensors of the sizes the library's own block queries give (v5)
        void *obs = nullptr, *cells = nullptr, *agents = nullptr;
        int64_t obs_bytes = 0, others_offset = 0, cells_bytes = 0, agents_bytes = 0;
        ok(frz_wildfire_obs_block(env, &obs, &obs_bytes, &others_offset), "frz_wildfire_obs_block");
        ok(frz_wildfire_state_block(env, &cells, &cells_bytes, &agents, &agents_bytes), "frz_wildfire_state_block");
        const int64_t per_step = (flags & FRZ_ROLLOUT_OBS_COMPACT) ? e.A * e.B * 4 : obs_bytes;
        spec.obs_tape = tape_or_null<uint8_t>(arena, obs_tape, at::kByte, steps * per_step, "obs_tape");
        spec.state_tape = tape_or_null<uint8_t>(arena, state_tape, at::kByte, steps * (cells_bytes + agents_bytes), "state_tape");
    }
    ok(frz_wildfire_rollout(static_cast<frz_wildfire_env*>(e.handle), &spec, current_stream(arena)), "frz_wildfire_rollout");
}
void cybersecurity_rollout(at::Tensor arena, int64_t handle, int64_t steps, int64_t rng_mode, int64_t flags, int64_t seed_increment, int64_t seed_stride,
                           int64_t policy_seed, int64_t first_step, const c10::optional<at::Tensor>& action_tape,
                           const c10::optional<at::Tensor>& randomness_a, const c10::optional<at::Tensor>& randomness_b,
                           const c10::optional<at::Tensor>& actions_out, bool record_actions, const c10::optional<at::Tensor>& reward_tape,
                           const c10::optional<at::Tensor>& done_tape, const c10::optional<at::Tensor>& list_record, const c10::optional<at::Tensor>& metrics,
                           const c10::optional<at::Tensor>& obs_tape, const c10::optional<at::Tensor>& state_tape) {
    const Env e = checked(arena, handle, kCybersecurity, "cybersecurity_rollout");
    frz_cybersecurity_env* const env = static_cast<frz_cybersecurity_env*>(e.handle);
    void* block = nullptr;
    int64_t block_bytes = 0;
    ok(frz_cybersecurity_list_block(env, &block, &block_bytes), "frz_cybersecurity_list_block");
    frz_rollout_spec spec = make_spec(arena, e, e.B * e.U, e.B * e.A, block_bytes, steps, rng_mode, flags, seed_increment, seed_stride, policy_seed,
                                      first_step, action_tape, randomness_a, randomness_b, actions_out, record_actions, reward_tape, done_tape, list_record,
                                      metrics);
    {
        void *obs = nullptr, *rows = nullptr, *presence = nullptr;
        int64_t obs_bytes = 0, rows_bytes = 0, presence_bytes = 0;
        ok(frz_cybersecurity_obs_block(env, &obs, &obs_bytes), "frz_cybersecurity_obs_block");
        ok(frz_cybersecurity_state_block(env, &rows, &rows_bytes, &presence, &presence_bytes), "frz_cybersecurity_state_block");
        spec.obs_tape = tape_or_null<uint8_t>(arena, obs_tape, at::kByte, steps * obs_bytes, "obs_tape");
        spec.state_tape = tape_or_null<uint8_t>(arena, state_tape, at::kByte, steps * ((rows_bytes + presence_bytes + 255) / 256 * 256), "state_tape");
    }
    ok(frz_cybersecurity_rollout(static_cast<frz_cybersecurity_env*>(e.handle), &spec, current_stream(arena)), "frz_cybersecurity_rollout");
}

// ---------------------------------------------------------------------------------------------------- rideshare
void rideshare_reset(at::Tensor arena, int64_t handle) {
    const Env e = checked(arena, handle, kRideshare, "rideshare_reset");
    ok(frz_rideshare_reset(static_cast<frz_rideshare_env*>(e.handle), current_stream(arena)), "frz_rideshare_reset");
}
void rideshare_rebuild(at::Tensor arena, int64_t handle) {
    const Env e = checked(arena, handle, kRideshare, "rideshare_rebuild");
    ok(frz_rideshare_rebuild(static_cast<frz_rideshare_env*>(e.handle), current_stream(arena)), "frz_rideshare_rebuild");
}
void rideshare_step(at::Tensor arena, int64_t handle, const at::Tensor& actions, int64_t agents, int64_t envs) {
    const Env e = checked(arena, handle, kRideshare, "rideshare_step");
    same_sizes(e, agents, envs, -1, "rideshare_step");
    ok(frz_rideshare_step(static_cast<frz_rideshare_env*>(e.handle), actions_ptr(arena, actions, e), current_stream(arena)), "frz_rideshare_step");
}
void rideshare_random_policy(const at::Tensor& arena, int64_t handle, int64_t policy_seed, int64_t policy_step, at::Tensor actions_out, int64_t agents,
                             int64_t envs) {
    const Env e = checked(arena, handle, kRideshare, "rideshare_random_policy");
    same_sizes(e, agents, envs, -1, "rideshare_random_policy");
    ok(frz_rideshare_random_policy(static_cast<frz_rideshare_env*>(e.handle), (uint64_t)policy_seed, (uint64_t)policy_step,
                                   const_cast<int32_t*>(actions_ptr(arena, actions_out, e)), current_stream(arena)),
       "frz_rideshare_random_policy");
}
void rideshare_step_random_policy(at::Tensor arena, int64_t handle, int64_t policy_seed, int64_t policy_step, at::Tensor actions_out, int64_t agents,
                                  int64_t envs) {
    const Env e = checked(arena, handle, kRideshare, "rideshare_step_random_policy");
    same_sizes(e, agents, envs, -1, "rideshare_step_random_policy");
    ok(frz_rideshare_step_random_policy(static_cast<frz_rideshare_env*>(e.handle), (uint64_t)policy_seed, (uint64_t)policy_step,
                                        const_cast<int32_t*>(actions_ptr(arena, actions_out, e)), current_stream(arena)),
       "frz_rideshare_step_random_policy");
}

// ---------------------------------------------------------------------------------------------------- per-env MT19937 streams
void mt19937_seed(at::Tensor state, at::Tensor index, const at::Tensor& seeds, const c10::optional<at::Tensor>& batch_indices) {
    const int64_t B = index.numel();
    TORCH_CHECK(state.is_cuda() && state.scalar_type() == at::kInt && state.is_contiguous() && state.numel() == 624 * B, "state must be int32 [624, B]");
    TORCH_CHECK(index.scalar_type() == at::kInt && index.is_contiguous() && seeds.scalar_type() == at::kInt && seeds.is_contiguous() && seeds.numel() == B,
                "index / seeds must be contiguous int32 [B]");
    const int32_t* which = nullptr;
    int64_t n = 0;
    if (batch_indices.has_value()) {
        TORCH_CHECK(batch_indices->scalar_type() == at::kInt && batch_indices->is_contiguous() && batch_indices->device() == state.device(),
                    "batch_indices must be a contiguous int32 tensor on the same device");
        which = batch_indices->data_ptr<int32_t>();
        n = batch_indices->numel();
    }
    ok(frz_mt19937_seed(reinterpret_cast<uint32_t*>(state.data_ptr<int32_t>()), index.data_ptr<int32_t>(), seeds.data_ptr<int32_t>(), which, n, B,
                        current_stream(state)),
       "frz_mt19937_seed");
}
void mt19937_generate(at::Tensor state, at::Tensor index, at::Tensor out, int64_t events, int64_t count) {
    const int64_t B = index.numel();
    TORCH_CHECK(state.is_cuda() && state.scalar_type() == at::kInt && state.is_contiguous() && state.numel() == 624 * B, "state must be int32 [624, B]");
    TORCH_CHECK(out.device() == state.device() && out.scalar_type() == at::kFloat && out.is_contiguous() && out.numel() == events * B * count,
                "out must be a contiguous float32 [events, B, count] tensor on the same device");
    ok(frz_mt19937_generate(reinterpret_cast<uint32_t*>(state.data_ptr<int32_t>()), index.data_ptr<int32_t>(), out.data_ptr<float>(), events, count, B,
                            current_stream(state)),
       "frz_mt19937_generate");
}

}  // namespace

TORCH_LIBRARY(frz, m) {
    // `arena` holds every array of the env (state, outputs, RNG state): the ops mutate it in place and return nothing; the typed views the
    // Python env hands out alias it.  `handle` = the value frz_<domain>_create returned (an opaque host pointer).
    m.def("wildfire_reset(Tensor(a!) arena, int handle) -> ()");
    m.def("wildfire_reset_reseed(Tensor(a!) arena, int handle, int seed_increment) -> ()");
    m.def("wildfire_rebuild(Tensor(a!) arena, int handle) -> ()");
    m.def("wildfire_step(Tensor(a!) arena, int handle, Tensor actions, int rng_mode, Tensor? field_randomness, Tensor? agent_randomness, int agents, "
          "int envs, int cells) -> ()");
    m.def("wildfire_random_policy(Tensor arena, int handle, int policy_seed, int policy_step, Tensor(b!) actions_out, int agents, int envs) -> ()");
    m.def("wildfire_step_random_policy(Tensor(a!) arena, int handle, int policy_seed, int policy_step, Tensor(b!) actions_out, int rng_mode, int agents, "
          "int envs) -> ()");
    m.def("wildfire_rollout"
          "(Tensor(a!) arena, int handle, int steps, int rng_mode, int flags, int seed_increment, int seed_stride, int policy_seed, int first_step, "
          "Tensor? action_tape, Tensor? randomness_a, Tensor? randomness_b, Tensor(b!)? actions_out, bool record_actions, Tensor(c!)? reward_tape, "
          "Tensor(d!)? done_tape, Tensor(e!)? list_record, Tensor(f!)? metrics, Tensor(g!)? obs_tape, Tensor(h!)? state_tape) -> ()");
    m.def("cybersecurity_rollout"
          "(Tensor(a!) arena, int handle, int steps, int rng_mode, int flags, int seed_increment, int seed_stride, int policy_seed, int first_step, "
          "Tensor? action_tape, Tensor? randomness_a, Tensor? randomness_b, Tensor(b!)? actions_out, bool record_actions, Tensor(c!)? reward_tape, "
          "Tensor(d!)? done_tape, Tensor(e!)? list_record, Tensor(f!)? metrics, Tensor(g!)? obs_tape, Tensor(h!)? state_tape) -> ()");
    m.def("cybersecurity_reset(Tensor(a!) arena, int handle) -> ()");
    m.def("cybersecurity_rebuild(Tensor(a!) arena, int handle) -> ()");
    m.def("cybersecurity_step(Tensor(a!) arena, int handle, Tensor actions, int rng_mode, Tensor? network_randomness, Tensor? agent_randomness, int agents, "
          "int envs, int nodes) -> ()");
    m.def("cybersecurity_random_policy(Tensor arena, int handle, int policy_seed, int policy_step, Tensor(b!) actions_out, int agents, int envs) -> ()");
    m.def("cybersecurity_step_random_policy(Tensor(a!) arena, int handle, int policy_seed, int policy_step, Tensor(b!) actions_out, int rng_mode, int agents, "
          "int envs) -> ()");
    m.def("rideshare_reset(Tensor(a!) arena, int handle) -> ()");
    m.def("rideshare_rebuild(Tensor(a!) arena, int handle) -> ()");
    m.def("rideshare_step(Tensor(a!) arena, int handle, Tensor actions, int agents, int envs) -> ()");
    m.def("rideshare_random_policy(Tensor arena, int handle, int policy_seed, int policy_step, Tensor(b!) actions_out, int agents, int envs) -> ()");
    m.def("rideshare_step_random_policy(Tensor(a!) arena, int handle, int policy_seed, int policy_step, Tensor(b!) actions_out, int agents, int envs) -> ()");
    m.def("mt19937_seed(Tensor(a!) state, Tensor(b!) index, Tensor seeds, Tensor? batch_indices) -> ()");
    m.def("mt19937_generate(Tensor(a!) state, Tensor(b!) index, Tensor(c!) out, int events, int count) -> ()");
}

TORCH_LIBRARY_IMPL(frz, CUDA, m) {  // HIP tensors dispatch on the CUDA key in ROCm builds of PyTorch
    m.impl("wildfire_reset", &wildfire_reset);
    m.impl("wildfire_reset_reseed", &wildfire_reset_reseed);
    m.impl("wildfire_rebuild", &wildfire_rebuild);
    m.impl("wildfire_step", &wildfire_step);
    m.impl("wildfire_random_policy", &wildfire_random_policy);
    m.impl("wildfire_step_random_policy", &wildfire_step_random_policy);
    m.impl("wildfire_rollout", &wildfire_rollout);
    m.impl("cybersecurity_rollout", &cybersecurity_rollout);
    m.impl("cybersecurity_reset", &cybersecurity_reset);
    m.impl("cybersecurity_rebuild", &cybersecurity_rebuild);
    m.impl("cybersecurity_step", &cybersecurity_step);
    m.impl("cybersecurity_random_policy", &cybersecurity_random_policy);
    m.impl("cybersecurity_step_random_policy", &cybersecurity_step_random_policy);
    m.impl("rideshare_reset", &rideshare_reset);
    m.impl("rideshare_rebuild", &rideshare_rebuild);
    m.impl("rideshare_step", &rideshare_step);
    m.impl("rideshare_random_policy", &rideshare_random_policy);
    m.impl("rideshare_step_random_policy", &rideshare_step_random_policy);
    m.impl("mt19937_seed", &mt19937_seed);
    m.impl("mt19937_generate", &mt19937_generate);
}
