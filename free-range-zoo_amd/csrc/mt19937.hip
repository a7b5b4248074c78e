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
// consecutive floats; output [events][B][count].  A second tensor (events2 x count2, may be empty) continues the same
// streams in the same launch: the step kernels draw field randomness then agent randomness (wildfire.py:409-410,
// cybersecurity.py:304-315) and one launch serves both.
//
// The 624-word block twist is done lazily, one word per draw.  Word i's twist reads words i, i+1 and i+397 (mod 624); a
// batch of kBatch <= 227 consecutive draws reads no word it rewrites, so all of its 2*kBatch+1 loads are issued before the
// first use (one memory round trip per batch instead of one per draw).
constexpr int kBatch = 16;

__device__ __forceinline__ int wrap(int j) { return j >= kN ? j - kN : j; }

// gate (optional): the epoch word and the two batch-totals slots of an env object's scan workspace + the channel that counts the envs not
// yet terminated (the next channel counts those not yet truncated).  When either is zero the step launch that follows is a no-op
// (utils/env.py:211-213: the reference returns before step_environment draws anything), so the streams are left where they are.
struct MtGate {
    const uint32_t* epoch;
    const uint32_t* totals;
    int32_t channel, stride;
};

__global__ void __launch_bounds__(frz::kBlock) mt_generate_kernel(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events,
                                                                    int64_t count, float* out2, int64_t events2, int64_t count2, int64_t B,
                                                                    const MtGate gate) {
    const int64_t b = (int64_t)blockIdx.x * frz::kBlock + threadIdx.x;
    if (b >= B) return;
    if (gate.epoch) {
        const uint32_t* left = gate.totals + ((*gate.epoch + 1u) & 1u) * gate.stride + gate.channel;  // what the previous launch left
        if (left[0] == 0u || left[1] == 0u) return;
    }
    int i = mt_index[b];
    const int64_t n1 = events * count, total = n1 + events2 * count2;
    for (int64_t u0 = 0; u0 < total; u0 += kBatch) {
        const int n = (int)(total - u0 < kBatch ? total - u0 : kBatch);
        uint32_t w[kBatch + 1], far[kBatch];
#pragma unroll
        for (int k = 0; k <= kBatch; ++k) w[k] = k <= n ? mt_state[(int64_t)wrap(i + k) * B + b] : 0u;
#pragma unroll
        for (int k = 0; k < kBatch; ++k) far[k] = k < n ? mt_state[(int64_t)wrap(wrap(i + k) + kM) * B + b] : 0u;
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if (k < n) {
                const uint32_t y = (w[k] & 0x80000000u) | (w[k + 1] & 0x7fffffffu);
                uint32_t v = far[k] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                mt_state[(int64_t)wrap(i + k) * B + b] = v;
                v ^= v >> 11;
                v ^= (v << 7) & 0x9d2c5680u;
                v ^= (v << 15) & 0xefc60000u;
                v ^= v >> 18;
                const float r = (float)(v & 0xFFFFFFu) * (1.0f / 16777216.0f);
                const int64_t u = u0 + k;
                if (u < n1) {
                    out[((u / count) * B + b) * count + u % count] = r;
                } else {
                    const int64_t u2 = u - n1;
                    out2[((u2 / count2) * B + b) * count2 + u2 % count2] = r;
                }
            }
        }
        i = wrap(i + n);
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
                       events, count, static_cast<float*>(nullptr), (int64_t)0, (int64_t)1, B, MtGate{});
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

int frz_mt19937_generate_pair(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events, int64_t count, float* out2, int64_t events2,
                              int64_t count2, int64_t B, void* stream) {
    if (!mt_state || !mt_index || !out || !out2 || B <= 0 || events <= 0 || count <= 0 || events2 <= 0 || count2 <= 0) return FRZ_E_INVALID;
    const int blocks = (int)((B + frz::kBlock - 1) / frz::kBlock);
    hipLaunchKernelGGL(mt_generate_kernel, dim3(blocks), dim3(frz::kBlock), 0, static_cast<hipStream_t>(stream), mt_state, mt_index, out,
                       events, count, out2, events2, count2, B, MtGate{});
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

}  // extern "C"

namespace frz {

// frz_mt19937_generate_pair for the step launches that stage their draws (library-internal): nothing is drawn when the env batch is
// frozen, exactly as the kernels that advance the streams themselves behave
int mt19937_generate_pair_gated(uint32_t* mt_state, int32_t* mt_index, float* out, int64_t events, int64_t count, float* out2, int64_t events2,
                                int64_t count2, int64_t B, const uint32_t* epoch, const uint32_t* totals, int channel, int stride, void* stream) {
    if (!mt_state || !mt_index || !out || !out2 || B <= 0 || events <= 0 || count <= 0 || events2 <= 0 || count2 <= 0) return FRZ_E_INVALID;
    const int blocks = (int)((B + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(mt_generate_kernel, dim3(blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), mt_state, mt_index, out, events,
                       count, out2, events2, count2, B, MtGate{epoch, totals, channel, stride});
    return hipGetLastError() == hipSuccess ? FRZ_OK : FRZ_E_LAUNCH;
}

}  // namespace frz
