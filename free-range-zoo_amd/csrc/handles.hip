// handles.hip — the live env handles of this library: frz_handle_kind / frz_handle_shape (include/frz.h) let a binding that receives a handle
// as a plain integer (the torch.ops.frz.* shim) check it — and the sizes it is about to trust — before anything is launched with it.
#include "frz_device.h"

#include "../../include/frz.h"

#include <mutex>
#include <unordered_map>

namespace frz {

namespace {
struct Entry {
    int kind;
    int64_t agents, envs, units;
};
std::mutex g_lock;
std::unordered_map<const void*, Entry>& table() {
    static std::unordered_map<const void*, Entry> t;
    return t;
}
}  // namespace

void handle_register(const void* handle, int kind, int64_t agents, int64_t envs, int64_t units) {
    std::lock_guard<std::mutex> guard(g_lock);
    table()[handle] = Entry{kind, agents, envs, units};
}
void handle_unregister(const void* handle) {
    std::lock_guard<std::mutex> guard(g_lock);
    table().erase(handle);
}

}  // namespace frz

extern "C" {

int frz_handle_kind(const void* handle) {
    std::lock_guard<std::mutex> guard(frz::g_lock);
    const auto it = frz::table().find(handle);
    return it == frz::table().end() ? 0 : it->second.kind;
}

int frz_handle_shape(const void* handle, int64_t* agents, int64_t* envs, int64_t* units) {
    std::lock_guard<std::mutex> guard(frz::g_lock);
    const auto it = frz::table().find(handle);
    if (it == frz::table().end() || !agents || !envs || !units) return FRZ_E_INVALID;
    *agents = it->second.agents, *envs = it->second.envs, *units = it->second.units;
    return FRZ_OK;
}

}  // extern "C"
