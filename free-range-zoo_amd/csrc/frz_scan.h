// frz_scan.h — launch-wide exclusive prefix sums of per-environment counts in ONE pass (one env per lane).
//
// The open action / observation lists of the reference are jagged tensors in env-major order: env b's segment starts at
// the sum of the counts of all envs < b.  Each workgroup (256 envs = one chunk) scans its envs with wavefront shuffles
// (counts packed four 16-bit channels per 64-bit word), combines its four waves through LDS, publishes the chunk sums
// as epoch-tagged 8-byte granules and reads the sums of the preceding chunks of its window (kRound chunks) plus the
// inclusive prefix the previous window's last chunk published.  The batch totals of every channel are left for the next
// launch (freeze detection, batch-global quirks).  One chunk per workgroup: chunk = blockIdx.x while the grid is
// co-resident (at most one workgroup per CU), otherwise chunks are handed out in arrival order (scan_take_chunk), so a
// chunk only waits on chunks whose workgroups have started.  See wildfire_roles.hip for the same scheme written inline.
#pragma once

#include "frz_device.h"

namespace frz {

constexpr int kTotalsStride = 32;  // uint32 words per totals slot (channels 0..31)
constexpr int kRound = 256;        // look-back window: a chunk sums at most kRound - 1 granules per channel + one prefix granule

// workspace words shared by every launch of one env object (zero-filled once by the caller)
struct ScanWorkspace {
    uint32_t* epoch;   // [1]; word 32 of the same 256-byte block is the chunk ticket counter
    uint32_t* totals;  // [2][kTotalsStride]
    uint64_t* agg;     // [nchunks][nch_total]
    uint64_t* prefix;  // [nchunks][nch_total]
};

template <int NCH, int BITS = 16>
struct ScanShared {
    uint64_t wave_scan[kWaves][(NCH + 64 / BITS - 1) / (64 / BITS)];
    uint32_t wave_live[kWaves][2];
    uint32_t reduce[kWaves][32];
    uint32_t prefix[32];
};

struct ScanLaunch {
    uint32_t epoch, tag;
    const uint32_t* prev;  // totals of the previous launch
    uint32_t* cur;         // totals of this launch (written by the last chunk)
};

__device__ __forceinline__ ScanLaunch scan_begin(const ScanWorkspace& ws) {
    ScanLaunch l;
    l.epoch = *ws.epoch;  // written by the previous launch; rewritten only after every workgroup of this one has read it
    l.tag = l.epoch + 1u;  // never 0 on a zero-filled workspace
    l.prev = ws.totals + ((l.epoch + 1u) & 1u) * kTotalsStride;
    l.cur = ws.totals + (l.epoch & 1u) * kTotalsStride;
    return l;
}

// The chunk this workgroup processes.  ticketed: arrival order (one atomic per workgroup; the holder of the last ticket
// re-arms the counter for the next launch: every ticket of this launch is out by then).
__device__ __forceinline__ int scan_take_chunk(const ScanWorkspace& ws, int nchunks, bool ticketed, int* lds_word) {
    if (!ticketed) return (int)blockIdx.x;
    if (threadIdx.x == 0) {
        uint32_t* const counter = ws.epoch + 32;
        const uint32_t t = atomicAdd(counter, 1u);
        if (t == (uint32_t)nchunks - 1u) atomicExch(counter, 0u);
        *lds_word = (int)t;
    }
    __syncthreads();
    return *lds_word;
}

// Counts are packed in BITS-wide fields over a 256-env chunk: cnt[ch] < 2^BITS / 256 per env (BITS = 16: up to 255 tasks per env, four
// channels per 64-bit word; BITS = 32: large wildfire grids, two channels per word).
// On return excl[ch] = sum of cnt[ch] over all envs of the launch that precede this lane's env.
// Channels nch and nch + 1 of the granules / totals carry the number of envs with live0 / live1 set.
template <int NCH, int BITS = 16>
__device__ __forceinline__ void scan_chunk(ScanShared<NCH, BITS>& sh, const ScanWorkspace& ws, const ScanLaunch& l, const uint32_t (&cnt)[NCH],
                                           bool live0, bool live1, int nch, int chunk, int nchunks, uint32_t (&excl)[NCH],
                                           uint32_t* err) {
    static_assert(BITS == 16 || BITS == 32, "field width");
    constexpr int PER = 64 / BITS;  // channels per packed word
    constexpr uint64_t FIELD = BITS == 16 ? 0xFFFFull : 0xFFFFFFFFull;
    constexpr int PW = (NCH + PER - 1) / PER;
    constexpr int NCHP = NCH + 2 <= 8 ? 8 : (NCH + 2 <= 16 ? 16 : 32);
    static_assert(NCH + 2 <= 32, "too many scan channels");
    // (role-split kernels call this from the first kBlock threads of a larger workgroup while the others run scan_chunk_passive)
    const int tid = threadIdx.x & (kBlock - 1), lane = lane_id(), wave = wave_id() & (kWaves - 1);
    const int nch_total = nch + 2;

    uint64_t packed[PW], incl[PW];
#pragma unroll
    for (int w = 0; w < PW; ++w) packed[w] = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) packed[ch / PER] |= ((uint64_t)cnt[ch] & FIELD) << (BITS * (ch % PER));
#pragma unroll
    for (int w = 0; w < PW; ++w) incl[w] = wave_inclusive_scan(packed[w]);
    const uint32_t n0 = (uint32_t)__popcll(__ballot(live0)), n1 = (uint32_t)__popcll(__ballot(live1));
    __syncthreads();
    if (lane == 63) {
#pragma unroll
        for (int w = 0; w < PW; ++w) sh.wave_scan[wave][w] = incl[w];
        sh.wave_live[wave][0] = n0;
        sh.wave_live[wave][1] = n1;
    }
    __syncthreads();
    uint64_t base[PW], total[PW];
#pragma unroll
    for (int w = 0; w < PW; ++w) {
        base[w] = 0;
        total[w] = 0;
#pragma unroll
        for (int j = 0; j < kWaves; ++j) {
            const uint64_t t = sh.wave_scan[j][w];
            base[w] += j < wave ? t : 0ull;
            total[w] += t;
        }
    }

    // publish this chunk's sums
    uint32_t mine = 0;  // channel `tid`
    if (tid < nch_total) {
        if (tid < nch) {
            uint64_t word = total[0];
#pragma unroll
            for (int w = 1; w < PW; ++w) word = (tid / PER) == w ? total[w] : word;
            mine = (uint32_t)((word >> (BITS * (tid % PER))) & FIELD);
        } else {
#pragma unroll
            for (int j = 0; j < kWaves; ++j) mine += sh.wave_live[j][tid - nch];
        }
        granule_store(ws.agg + (int64_t)chunk * nch_total + tid, l.tag, mine);
    }

    // sums of the preceding chunks of this round + inclusive prefix of the previous round's last chunk
    const int round_first = chunk & ~(kRound - 1);
    bool timed_out = false;
    uint32_t acc = 0;
    {
        const int ch = tid & (NCHP - 1), slot = tid / NCHP;
        constexpr int PP = kBlock / NCHP, UNR = 8;
        for (int first = round_first; first < chunk; first += PP * UNR) {
            uint32_t part = 0;
            for (int spin = 0;; ++spin) {  // bounded
                bool all = true;
                part = 0;
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    // unconditional loads (lanes without a predecessor read granule 0 and ignore it): in flight together
                    const int pred = first + u * PP + slot;
                    const bool valid = pred < chunk && ch < nch_total;
                    const uint64_t g = granule_load(ws.agg + (valid ? (int64_t)pred * nch_total + ch : (int64_t)0));
                    all = all && (!valid || (uint32_t)(g >> 32) == l.tag);
                    part += valid ? (uint32_t)g : 0u;
                }
                if (all) break;
                if (spin >= (1 << 22)) {
                    timed_out = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            acc += part;
        }
        if (round_first > 0 && tid < nch_total) acc += granule_wait(ws.prefix + (int64_t)(round_first - 1) * nch_total + tid, l.tag, &timed_out);
#pragma unroll
        for (int dd = NCHP; dd < 64; dd <<= 1) acc += __shfl_xor(acc, dd, 64);
        if (lane < NCHP) sh.reduce[wave][lane] = acc;
    }
    __syncthreads();
    if (tid < nch_total) {
        uint32_t s = 0;
#pragma unroll
        for (int j = 0; j < kWaves; ++j) s += sh.reduce[j][tid];
        sh.prefix[tid] = s;
        const bool round_last = (chunk & (kRound - 1)) == kRound - 1 || chunk == nchunks - 1;
        if (round_last) {
            granule_store(ws.prefix + (int64_t)chunk * nch_total + tid, l.tag, s + mine);
            if (chunk == nchunks - 1) l.cur[tid] = s + mine;  // batch totals, read by the next launch
        }
    }
    __syncthreads();
    if (timed_out) *err |= 2u;  // FRZ_ERR_SCAN_TIMEOUT

#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int w = ch / PER, shf = BITS * (ch % PER);
        excl[ch] = sh.prefix[ch] + (uint32_t)(((base[w] + incl[w] - packed[w]) >> shf) & FIELD);
    }
}

// The wavefronts of a role-split workgroup that do not scan: scan_chunk's four workgroup barriers, nothing else
__device__ __forceinline__ void scan_chunk_passive_front() {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ void scan_chunk_passive_back() {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
}

// The workgroup that owns the last chunk finished its look-back only after every other chunk published, i.e. after
// every workgroup of the launch read the epoch: it advances it for the next launch.
__device__ __forceinline__ void scan_end(const ScanWorkspace& ws, const ScanLaunch& l, int chunk, int nchunks) {
    if (chunk == nchunks - 1 && threadIdx.x == 0) *ws.epoch = l.epoch + 1u;
}

}  // namespace frz
