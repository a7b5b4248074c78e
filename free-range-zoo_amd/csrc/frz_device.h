// frz_device.h — device-side building blocks shared by the gfx950 env-step kernels.
//
// Written for CDNA4 only: 64-lane wavefronts, one environment per lane, 256-thread workgroups (one wave per SIMD).
// Compile with -ffp-contract=off: reward / probability arithmetic must round once per operation, like the
// reference's eager float32 ops.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace frz {

constexpr int kBlock = 256;  // threads (= environments) per workgroup
constexpr int kWaves = kBlock / 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. SC'11).  The FRZ_RNG_PHILOX stream definitions live with each domain's step entry
// point in include/frz.h (and oracle/frz_oracle_rng.c); float = (word >> 8) * 2^-24.
// ------------------------------------------------------------------------------------------------------------
struct Philox4 {
    uint32_t w[4];
};

#ifndef FRZ_PHILOX_ROUNDS
#define FRZ_PHILOX_ROUNDS 10  // (an experiment build may lower it: tools/dbg/ab_persist.py, profiles/r04_experiments.txt)
#endif
__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < FRZ_PHILOX_ROUNDS; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;  // one 32x32->64 multiply each
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

__device__ __forceinline__ float u32_to_unit_float(uint32_t word) { return (float)(word >> 8) * (1.0f / 16777216.0f); }

// A 128-bit Philox block read as five 24-bit uniforms (w[0] least significant; the top 8 bits of w[3] are unused):
// field k = bits [24k, 24k + 24), float = field * 2^-24.  Used where a step needs many draws per env (wildfire).
template <int K>
__device__ __forceinline__ float philox_unit24(const Philox4& w) {
    static_assert(K >= 0 && K < 5, "five 24-bit fields per block");
    uint32_t v;
    if constexpr (K == 0) v = w.w[0] & 0xFFFFFFu;
    else if constexpr (K == 1) v = __builtin_amdgcn_alignbit(w.w[1], w.w[0], 24) & 0xFFFFFFu;
    else if constexpr (K == 2) v = __builtin_amdgcn_alignbit(w.w[2], w.w[1], 16) & 0xFFFFFFu;
    else if constexpr (K == 3) v = w.w[2] >> 8;
    else v = w.w[3] & 0xFFFFFFu;
    return (float)v * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float philox_unit24(const Philox4& w, int k) {
    const uint64_t lo = (uint64_t)w.w[0] | ((uint64_t)w.w[1] << 32), hi = (uint64_t)w.w[2] | ((uint64_t)w.w[3] << 32);
    const int sh = 24 * k;
    const uint64_t bits = sh < 64 ? ((lo >> sh) | (sh > 40 ? hi << (64 - sh) : 0ull)) : (hi >> (sh - 64));
    return (float)(uint32_t)(bits & 0xFFFFFFull) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------------------------------
// Wavefront / workgroup prefix sums of per-lane counts (variable-length task lists).
// Counts are packed four 16-bit channels per 64-bit word: a 256-env workgroup with <= 64 tasks per env cannot
// overflow a channel (256 * 64 = 16384 < 65536).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_inclusive_scan(uint64_t v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = __shfl_up(v, d, 64);
        if (lane >= d) v += up;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// Inter-workgroup hand-off granule: one naturally aligned 8-byte {tag, value} word written by ONE agent-scope
// store and read by agent-scope (L1-bypassing) loads.  The tag carries the launch epoch, so a granule is
// self-validating and needs neither fences nor clearing between launches (MI355X_MICROARCH.md, hand-off R2).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void granule_store(uint64_t* slot, uint32_t tag, uint32_t value) {
    __hip_atomic_store(slot, ((uint64_t)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t granule_load(const uint64_t* slot) {
    return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Spin (bounded) until the granule carries `tag`; returns its value, sets *timed_out on the bound.
__device__ __forceinline__ uint32_t granule_wait(const uint64_t* slot, uint32_t tag, bool* timed_out) {
    constexpr int kSpinBound = 1 << 22;  // ~seconds; every spin in this library is bounded
    for (int spin = 0; spin < kSpinBound; ++spin) {
        const uint64_t g = granule_load(slot);
        if ((uint32_t)(g >> 32) == tag) return (uint32_t)g;
        __builtin_amdgcn_s_sleep(2);
    }
    *timed_out = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Write-through stores.  A plain store leaves its line dirty in the XCD's L2 and the end-of-kernel release writes all
// of them back at once: for a step launch that writes ~20 MB in its last microseconds that tail is on the critical path
// of the next launch.  A store marked sc0 sc1 goes through to memory as it is issued (the line stays valid in L2 for
// the next step's loads), so the write-back is spread over the kernel: 10.6 -> 9.9 us per launch on the wildfire bench
// kernel.  Worth it only where a wavefront writes WHOLE lines ([rows][B] rows, offsets arrays): record-strided stores
// (16-byte pieces of different lines per lane) need the L2 to merge them and get slower (cybersecurity task rows:
// 11.3 -> 13.7 us when they were written through too).  A relaxed system-scope atomic store is exactly such a store
// to the compiler (same instruction, sc0 sc1 set, no fence, its vmcnt bookkeeping intact).
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void store_through(T* p, T v) {
    static_assert(sizeof(T) == 1 || sizeof(T) == 2 || sizeof(T) == 4 || sizeof(T) == 8, "one store instruction");
    if constexpr (sizeof(T) == 1)
        __hip_atomic_store(reinterpret_cast<uint8_t*>(p), __builtin_bit_cast(uint8_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if constexpr (sizeof(T) == 2)
        __hip_atomic_store(reinterpret_cast<uint16_t*>(p), __builtin_bit_cast(uint16_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if constexpr (sizeof(T) == 4)
        __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else
        __hip_atomic_store(reinterpret_cast<uint64_t*>(p), __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// An element of a [rows][B] row: reads like a T, assignment is a write-through store
template <typename T>
struct RowRef {
    T* p;
    __device__ __forceinline__ operator T() const { return *p; }
    __device__ __forceinline__ const RowRef& operator=(T v) const {
#ifdef FRZ_ROWS_PLAIN  // (experiment builds: the rows left to the L2's write-back)
        *p = v;
#else
        store_through(p, v);
#endif
        return *this;
    }
    __device__ __forceinline__ const RowRef& operator=(const RowRef& o) const { return *this = (T)o; }
    __device__ __forceinline__ const RowRef& operator+=(T v) const { return *this = (T)(*p + v); }
};
template <typename T>
struct RowRef<const T> {
    const T* p;
    __device__ __forceinline__ operator T() const { return *p; }
};

// mt19937.hip: frz_mt19937_generate_pair gated on an env object's batch totals (host-callable; see the definition)
int mt19937_generate_pair_gated(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events, int64_t count, float* out2, int64_t events2,
                                int64_t count2, int64_t B, const uint32_t* epoch, const uint32_t* totals, int channel, int stride, void* stream);


// handles.hip: the registry behind frz_handle_kind / frz_handle_shape (kind 1 wildfire, 2 cybersecurity, 3 rideshare; units = cells / nodes /
// passenger slots per env)
void handle_register(const void* handle, int kind, int64_t agents, int64_t envs, int64_t units);
void handle_unregister(const void* handle);

}  // namespace frz
