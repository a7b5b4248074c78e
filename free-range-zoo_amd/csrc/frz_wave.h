// frz_wave.h — building blocks of the "one environment per wavefront" kernels (rideshare.hip, wildfire_grid.hip): the 64 lanes of a
// wavefront hold the slots / cells of ONE env, per-agent quantities sit in lanes 0..A-1 of the same wavefront ("agent lanes"), and
// what crosses between the two views is a ballot, a v_writelane, a ds_bpermute or an LDS word — never a workgroup barrier.
//
// A CU has ONE scalar unit for its four SIMDs and a SIMD issues roughly one instruction per four cycles whatever its kind, so per-agent
// work is written as vector arithmetic over the agent lanes (all agents at once), not as a scalar loop over the agents.
#pragma once

#include "frz_device.h"

#include <type_traits>
#include <utility>

namespace frz {

// element `index` of an array whose base is wave-uniform, addressed as (scalar base) + (32-bit byte offset): one instruction instead of
// 64-bit address arithmetic per access (the callers bound every array addressed this way below 4 GiB)
template <typename T>
__device__ __forceinline__ T& at32(T* base, uint32_t index) {
    using Byte = std::conditional_t<std::is_const_v<T>, const char, char>;
    return *reinterpret_cast<T*>(reinterpret_cast<Byte*>(base) + (uint64_t)(index * (uint32_t)sizeof(T)));
}

__device__ __forceinline__ int lane_rank(uint64_t m) {  // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// value held by another lane.  A ds_bpermute reads 0 from lanes a divergent branch has switched off: call it with every lane active.
__device__ __forceinline__ int from_lane(int src_lane, int value) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, value); }
__device__ __forceinline__ int read_lane(int value, int lane) { return __builtin_amdgcn_readlane(value, lane); }
__device__ __forceinline__ int last_bit(uint64_t m) { return 63 - __builtin_clzll(m); }
__device__ __forceinline__ int first_bit(uint64_t m) { return __builtin_ctzll(m); }

// vec with lane LANE replaced by a wave-uniform value.  gfx950 needs two wait states between a vector instruction that writes a scalar
// register (a ballot's v_cmp) and a vector instruction that reads it; the compiler inserts them for its own instructions but does not
// look inside an asm statement, hence the s_nop.
template <int LANE>
__device__ __forceinline__ uint32_t write_lane_c(uint32_t vec, uint32_t scalar) {
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(scalar), "n"(LANE));
    return vec;
}
template <int LANE>
__device__ __forceinline__ void write_lane_c(uint32_t& lo, uint32_t& hi, uint64_t scalar) {  // the two halves of a ballot
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
        : "+v"(lo), "+v"(hi)
        : "s"((uint32_t)scalar), "s"((uint32_t)(scalar >> 32)), "n"(LANE));
}
// `value` in the lanes whose bit of a wave-uniform mask is set, 0.0f elsewhere: the mask is the instruction's lane predicate (one
// vector instruction; written as `(mask >> lane) & 1` the compiler shifts a 64-bit value per lane)
__device__ __forceinline__ float where_lane(uint64_t mask, float value) {
    float out;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(out) : "v"(value), "s"(mask));
    return out;
}
template <typename F, int... Is>
__device__ __forceinline__ void for_each_index(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}

// LDS traffic of one wavefront is executed in order; this keeps the compiler from reordering it across a cross-lane hand-over
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// position of the n-th (0-based) set bit of a mask of W 32-bit words, -1 when the mask has no more than n bits: a descent on population
// counts, per lane, without divergence
template <int W>
__device__ __forceinline__ int select_nth(const uint32_t (&mask)[W], int n) {
    int word = 0, rem = n;
#pragma unroll
    for (int i = 0; i + 1 < W; ++i) {
        const int c = __popc(mask[i]);
        const bool up = word == i && rem >= c;
        rem -= up ? c : 0;
        word += up ? 1 : 0;
    }
    uint32_t w = mask[0];
#pragma unroll
    for (int i = 1; i < W; ++i) w = word == i ? mask[i] : w;
    const bool found = n >= 0 && rem < __popc(w);
    int pos = 32 * word;
#pragma unroll
    for (int width = 16; width >= 1; width >>= 1) {
        const int c = __popc(w & ((1u << width) - 1u));
        const bool up = rem >= c;
        rem -= up ? c : 0;
        w = up ? w >> width : w;
        pos += up ? width : 0;
    }
    return found ? pos : -1;
}
template <int W>
__device__ __forceinline__ int popc_words(const uint32_t (&mask)[W]) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) n += __popc(mask[i]);
    return n;
}

// The env a wavefront owns (ENVS = wavefronts per workgroup), -1 past the batch.  Workgroups are dealt round-robin over the 8 XCDs
// (blockIdx % 8 share one); the mapping gives every run of 8 * ENVS consecutive envs (for ENVS = 4: one 128-byte line of each
// [rows][B] array) to workgroups of ONE XCD, so that its L2 merges their 4-byte pieces into whole lines.  Speed only: any bijection is
// correct.  The result is the same in every lane and said so explicitly (readfirstlane), so that everything derived from it stays in
// scalar registers.
template <int ENVS>
__device__ __forceinline__ int env_of_wave(int64_t B) {
    const uint32_t nblocks = gridDim.x, blk = blockIdx.x;
    uint32_t quad = blk;
    const uint32_t group = blk >> 6;
    if ((group + 1u) * 64u <= nblocks) {
        const uint32_t l = blk & 63u;
        quad = (group << 6) + ((l & 7u) << 3) + (l >> 3);
    }
    const int b = __builtin_amdgcn_readfirstlane((int)(quad * ENVS + (threadIdx.x >> 6)));
    return b < B ? b : -1;
}

}  // namespace frz
