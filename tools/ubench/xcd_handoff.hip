// Micro-benchmark (diagnostic, not part of the product): what an inter-workgroup hand-off costs on gfx950 when the two workgroups sit
// on the SAME XCD (one L2 between them) and when they sit on different XCDs, for every cache-scope pair of the store and the polling load.
//   1. `where`: which XCD / CU every workgroup of a 256-workgroup launch lands on (HW_REG_XCC_ID, HW_REG_HW_ID).
//   2. `pingpong<LD, ST>`: workgroup a stores ping = i, workgroup b polls it and stores pong = i, a polls pong; N round trips timed with
//      the constant-rate wall clock.  LD / ST: 0 = no scope bits, 1 = sc0, 2 = sc1 (agent scope: what the product's granules use),
//      3 = sc0 sc1.  A poll that gives up after 2^18 reads is reported as "never seen" (a stale line in a cache the scope does not bypass).
//   3. `fanin<LD, ST>`: 32 workgroups of one XCD (or of all XCDs) each post a granule, every one of them polls all 32 (a wave's lanes):
//      the shape of the product's batch-totals hand-off, per step, 50 steps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; }   // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ uint32_t hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); }          // HW_REG_HW_ID

template <int MODE>
__device__ __forceinline__ uint32_t ld(const uint32_t* p) {
    uint32_t v;
    if constexpr (MODE == 0) asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (MODE == 1) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (MODE == 2) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (MODE == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__device__ __forceinline__ void st(uint32_t* p, uint32_t v) {
    if constexpr (MODE == 0) asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
    if constexpr (MODE == 1) asm volatile("global_store_dword %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
    if constexpr (MODE == 2) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    if constexpr (MODE == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

__global__ void __launch_bounds__(64) where(uint32_t* out) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc_id();
        out[2 * blockIdx.x + 1] = hw_id();
    }
}

constexpr int SPIN_MAX = 1 << 18;

// flags: ping at word 0, pong at word 64 (different 128-byte lines); result: [0] wall-clock ticks, [1] failures, [2] xcc of a, [3] xcc of b
template <int LD, int ST>
__global__ void __launch_bounds__(64) pingpong(uint32_t* flags, unsigned long long* result, int a, int b, int n, uint32_t base) {
    const int wg = blockIdx.x;
    if (threadIdx.x != 0 || (wg != a && wg != b)) return;
    uint32_t* ping = flags;
    uint32_t* pong = flags + 64;
    unsigned long long fails = 0;
    if (wg == a) {
        result[2] = xcc_id();
        const unsigned long long t0 = wall_clock64();
        for (int i = 1; i <= n; ++i) {
            st<ST>(ping, base + i);
            int spins = 0;
            while (ld<LD>(pong) != base + i && ++spins < SPIN_MAX) {}
            fails += spins >= SPIN_MAX;
            if (spins >= SPIN_MAX) break;
        }
        result[0] = wall_clock64() - t0;
        result[1] = fails;
        st<3>(ping, 0xffffffffu);  // release b whatever happened
    } else {
        result[3] = xcc_id();
        for (int i = 1; i <= n; ++i) {
            int spins = 0;
            uint32_t v;
            while ((v = ld<LD>(ping)) != base + i && v != 0xffffffffu && ++spins < SPIN_MAX) {}
            if (v == 0xffffffffu) break;
            if (spins >= SPIN_MAX) { if (LD != 3) { while (ld<3>(ping) != 0xffffffffu && ++spins < 4 * SPIN_MAX) {} } break; }
            st<ST>(pong, base + i);
        }
    }
}

// `members` workgroups (ids listed in `who`) each post {step} into their slot (one 128-byte line each) and poll everybody's slot, `steps` times.
// result[k] = wall-clock ticks of member k; result[members] = failures.
template <int LD, int ST>
__global__ void __launch_bounds__(64) fanin(uint32_t* slots, unsigned long long* result, const int* who, int members, int steps, uint32_t base) {
    int me = -1;
    for (int k = 0; k < members; ++k) me = who[k] == (int)blockIdx.x ? k : me;
    if (me < 0) return;
    const int lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    unsigned long long fails = 0;
    for (int s = 1; s <= steps; ++s) {
        if (lane == 0) st<ST>(slots + 32 * me, base + s);
        int spins = 0;
        bool ok;
        do {
            const uint32_t v = lane < members ? ld<LD>(slots + 32 * lane) : base + s;
            ok = __all((int32_t)(v - (base + s)) >= 0);
        } while (!ok && ++spins < SPIN_MAX);
        if (!ok) { fails = 1; break; }
    }
    if (lane == 0) {
        result[me] = wall_clock64() - t0;
        if (fails) atomicAdd(result + members, 1ull);
    }
}

static const char* NAMES[4] = {"-", "sc0", "sc1", "sc0sc1"};

template <int LD, int ST>
void run_pair(uint32_t* flags, unsigned long long* result, int a, int b, int n, double tick_ns, uint32_t& base, const char* what) {
    CK(hipMemset(result, 0, 64));
    CK(hipMemset(flags, 0, 32 * 128 * 4));
    hipLaunchKernelGGL((pingpong<LD, ST>), dim3(256), dim3(64), 0, 0, flags, result, a, b, n, base);
    CK(hipDeviceSynchronize());
    unsigned long long h[4];
    CK(hipMemcpy(h, result, sizeof(h), hipMemcpyDeviceToHost));
    base += n + 16;
    if (h[1]) printf("  %-10s load %-6s store %-6s  xcd %llu -> %llu : never seen\n", what, NAMES[LD], NAMES[ST], h[2], h[3]);
    else printf("  %-10s load %-6s store %-6s  xcd %llu -> %llu : %.0f ns per round trip (%.0f ns one way)\n", what, NAMES[LD], NAMES[ST], h[2], h[3],
                h[0] * tick_ns / n, h[0] * tick_ns / n / 2);
}

template <int LD, int ST>
void run_fanin(uint32_t* slots, unsigned long long* result, int* who_dev, const std::vector<int>& who, int steps, double tick_ns, uint32_t& base, const char* what) {
    CK(hipMemcpy(who_dev, who.data(), who.size() * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemset(result, 0, 8 * 80));
    CK(hipMemset(slots, 0, 32 * 128 * 4));
    hipLaunchKernelGGL((fanin<LD, ST>), dim3(256), dim3(64), 0, 0, slots, result, who_dev, (int)who.size(), steps, base);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(who.size() + 1);
    CK(hipMemcpy(h.data(), result, h.size() * 8, hipMemcpyDeviceToHost));
    base += steps + 16;
    const unsigned long long worst = *std::max_element(h.begin(), h.end() - 1);
    if (h.back()) printf("  %-22s load %-6s store %-6s : never seen\n", what, NAMES[LD], NAMES[ST]);
    else printf("  %-22s load %-6s store %-6s : %.0f ns per all-to-all step (%zu workgroups)\n", what, NAMES[LD], NAMES[ST], worst * tick_ns / steps, who.size());
}

int main() {
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const double tick_ns = 1e6 / rate_khz;
    uint32_t* where_dev;
    CK(hipMalloc(&where_dev, 2 * 256 * 4));
    hipLaunchKernelGGL(where, dim3(256), dim3(64), 0, 0, where_dev);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> w(512);
    CK(hipMemcpy(w.data(), where_dev, 2048, hipMemcpyDeviceToHost));
    printf("workgroup -> XCD of a 256-workgroup launch (first 32): ");
    for (int i = 0; i < 32; ++i) printf("%u ", w[2 * i]);
    bool round_robin = true;
    for (int i = 0; i < 256; ++i) round_robin &= w[2 * i] == (uint32_t)(i % 8);
    printf("\nworkgroup i on XCD i %% 8 for all 256: %s\n", round_robin ? "yes" : "no");
    printf("HW_ID of workgroups 0, 8, 16, 24 (same XCD): %08x %08x %08x %08x\n", w[1], w[17], w[33], w[49]);

    uint32_t* flags;
    unsigned long long* result;
    int* who_dev;
    CK(hipMalloc(&flags, 32 * 128 * 4));
    CK(hipMemset(flags, 0, 32 * 128 * 4));
    CK(hipMalloc(&result, 8 * 80));
    CK(hipMalloc(&who_dev, 64 * 4));
    uint32_t base = 16;
    const int n = 2000;
    // same XCD: workgroups 0 and 8 (if round robin); different XCDs: 0 and 1
    int same_b = 8, other_b = 1;
    for (int i = 1; i < 256; ++i) if (w[2 * i] == w[0] && w[2 * i + 1] != w[1]) { same_b = i; break; }
    for (int i = 1; i < 256; ++i) if (w[2 * i] != w[0]) { other_b = i; break; }
    printf("ping-pong, %d round trips, wall clock %.1f MHz:\n", n, rate_khz / 1e3);
#define PAIRS(LD, ST) run_pair<LD, ST>(flags, result, 0, same_b, n, tick_ns, base, "same XCD"); run_pair<LD, ST>(flags, result, 0, other_b, n, tick_ns, base, "other XCD");
    PAIRS(2, 2) PAIRS(3, 3) PAIRS(1, 0) PAIRS(1, 1) PAIRS(1, 2) PAIRS(2, 0) PAIRS(2, 1) PAIRS(0, 2)
    // fan-in: the 32 workgroups of XCD 0, then 32 workgroups spread over all XCDs
    std::vector<int> local, spread;
    for (int i = 0; i < 256 && local.size() < 32; ++i) if (w[2 * i] == w[0]) local.push_back(i);
    for (int i = 0; i < 32; ++i) spread.push_back(i);
    printf("all-to-all among 32 workgroups, 50 steps:\n");
#define FAN(LD, ST) run_fanin<LD, ST>(flags, result, who_dev, local, 50, tick_ns, base, "32 WGs of one XCD"); run_fanin<LD, ST>(flags, result, who_dev, spread, 50, tick_ns, base, "32 WGs over 8 XCDs");
    FAN(2, 2) FAN(1, 0) FAN(1, 1) FAN(1, 2) FAN(3, 3)
    return 0;
}
