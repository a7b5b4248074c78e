// mt19937.hip — per-environment MT19937 streams on gfx950, seed-identical to the reference's CPU RandomGenerator
// (free_range_zoo/utils/random_generator.py:49-146: one torch CPU generator state per env; torch's CPU generator is
// MT19937 init_genrand(seed), and float32 rand = (u32 & 0xFFFFFF) * 2^-24, one output per element).
//
// Layout: state word j of env b at mt_state[j * B + b] (env innermost): when all envs sit at the same stream
// position — the normal case, every env draws the same number of floats per step — a wavefront reads 256
// contiguous bytes per word.  The 624-word block twist is done lazily, one word per draw, in the same in-place order
// as the textbook generator, so only 3 words are read and 1 written per output.
#include "frz_device.h"

#include "../../include/frz.h"

namespace {

constexpr int kN = 624, kM = 397;

__global__ void __launch_bounds__(frz::kBlock) mt_seed_kernel(uint32_t* mt_state, int32_t* mt_index, const int32_t* seeds,
                                                                const int32_t* batch_indices, int64_t n, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * frz::kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t b = batch_indices ? batch_indices[i] : i;
    if (b < 0 || b >= B) return;
    uint32_t v = (uint32_t)seeds[b];
    mt_state[b] = v;
    for (int j = 1; j < kN; ++j) {
        v = 1812433253u * (v ^ (v >> 30)) + (uint32_t)j;
        mt_state[(int64_t)j * B + b] = v;
    }
    mt_index[b] = 0;
}

// RandomGenerator.generate(), unbuffered per-env branch (random_generator.py:106-114): env b draws events*count
// consecutive floats; output [events][B][count].
__global__ void __launch_bounds__(frz::kBlock) mt_generate_kernel(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events,
                                                                    int64_t count, int64_t B) {
    const int64_t b = (int64_t)blockIdx.x * frz::kBlock + threadIdx.x;
    if (b >= B) return;
    int i = mt_index[b];
    uint32_t cur = mt_state[(int64_t)i * B + b];
    for (int64_t e = 0; e < events; ++e) {
        for (int64_t k = 0; k < count; ++k) {
            const int i1 = i + 1 == kN ? 0 : i + 1;
            const int im = i + kM >= kN ? i + kM - kN : i + kM;
            const uint32_t next = mt_state[(int64_t)i1 * B + b];
            const uint32_t far = mt_state[(int64_t)im * B + b];
            const uint32_t y = (cur & 0x80000000u) | (next & 0x7fffffffu);
            uint32_t v = far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            mt_state[(int64_t)i * B + b] = v;
            v ^= v >> 11;
            v ^= (v << 7) & 0x9d2c5680u;
            v ^= (v << 15) & 0xefc60000u;
            v ^= v >> 18;
            out[(e * B + b) * count + k] = (float)(v & 0xFFFFFFu) * (1.0f / 16777216.0f);
            // word i1 is re-read from memory next iteration unless it is the word just written (i1 == i only if kN == 1)
            cur = next;
            i = i1;
            if (i == 0) cur = mt_state[b];  // wrapped: word 0 was rewritten earlier in this generation
        }
    }
    mt_index[b] = i;
}

}  // namespace

extern "C" {

int frz_mt19937_seed(uint32_t* mt_state, int32_t* mt_index, const int32_t* seeds, const int32_t* batch_indices, int64_t n, int64_t B,
                     void* stream) {
    if (!mt_state || !mt_index || !seeds || B <= 0) return FRZ_E_INVALID;
    if (!batch_indices) n = B;
    if (n <= 0) return FRZ_OK;
    const int blocks = (int)((n + frz::kBlock - 1) / frz::kBlock);
    hipLaunchKernelGGL(mt_seed_kernel, dim3(blocks), dim3(frz::kBlock), 0, static_cast<hipStream_t>(stream), mt_state, mt_index, seeds,
                       batch_indices, n, B);
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

int frz_mt19937_generate(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events, int64_t count, int64_t B, void* stream) {
    if (!mt_state || !mt_index || !out || B <= 0 || events < 0 || count < 0) return FRZ_E_INVALID;
    if (events == 0 || count == 0) return FRZ_OK;
    const int blocks = (int)((B + frz::kBlock - 1) / frz::kBlock);
    hipLaunchKernelGGL(mt_generate_kernel, dim3(blocks), dim3(frz::kBlock), 0, static_cast<hipStream_t>(stream), mt_state, mt_index, out,
                       events, count, B);
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

}  // extern "C"
