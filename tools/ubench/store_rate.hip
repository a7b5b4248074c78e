// Micro-benchmark (diagnostic, not part of the product): per-CU store issue rate on gfx950.
// One 256-thread workgroup per CU; every lane stores NR values per "env" in four patterns:
//   0: dword, SoA rows  [r][B]          (256 B contiguous per wave-instruction)
//   1: dwordx4, record [B][NR]           (lane writes 16 B at stride NR*4)
//   2: dwordx2, record
//   3: dword SoA but only 1 wave per workgroup active (no contention inside the CU)
// Prints shader cycles per wave-instruction and bytes/clk/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NR = 64;

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, unsigned long long* cyc, int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    const bool on = MODE != 3 || threadIdx.x < 64;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (on) {
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int r = 0; r < NR; ++r) out[(size_t)r * B + b] = r + b;
        } else if (MODE == 1) {
            uint4* p = reinterpret_cast<uint4*>(out + (size_t)b * NR);
#pragma unroll
            for (int r = 0; r < NR / 4; ++r) p[r] = make_uint4(r, b, r, b);
        } else if (MODE == 2) {
            uint2* p = reinterpret_cast<uint2*>(out + (size_t)b * NR);
#pragma unroll
            for (int r = 0; r < NR / 2; ++r) p[r] = make_uint2(r, b);
        } else if (MODE == 4) {  // x4, lanes contiguous (1 KB per wave-instruction), planes of [B] uint4
            uint4* p = reinterpret_cast<uint4*>(out);
#pragma unroll
            for (int r = 0; r < NR / 4; ++r) p[(size_t)r * B + b] = make_uint4(r, b, r, b);
        } else if (MODE == 5) {  // 24-byte records [B][6], three x2 per record; NR/6 planes (uses 60 of 64 rows)
#pragma unroll
            for (int r = 0; r < NR / 6; ++r) {
                uint2* p = reinterpret_cast<uint2*>(out + ((size_t)r * B + b) * 6);
                p[0] = make_uint2(r, b); p[1] = make_uint2(r, b); p[2] = make_uint2(r, b);
            }
        } else if (MODE == 6) {  // 32-byte rows [B][8] as two x4; NR/8 planes
#pragma unroll
            for (int r = 0; r < NR / 8; ++r) {
                uint4* p = reinterpret_cast<uint4*>(out + ((size_t)r * B + b) * 8);
                p[0] = make_uint4(r, b, r, b); p[1] = make_uint4(r, b, r, b);
            }
        }
    }
    const unsigned long long ti = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): stores retired from the wave's point of view
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = ti - t0; cyc[blockIdx.x * 2 + 1] = t2 - t0; }
}

template <int MODE>
void run(const char* name, int blocks) {
    const int B = blocks * 256;
    uint32_t* out; unsigned long long* cyc;
    CK(hipMalloc(&out, (size_t)B * NR * 4)); CK(hipMalloc(&cyc, blocks * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) k<MODE><<<blocks, 256>>>(out, cyc, B);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) k<MODE><<<blocks, 256>>>(out, cyc, B);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 2);
    CK(hipMemcpy(h.data(), cyc, blocks * 16, hipMemcpyDeviceToHost));
    double a = 0, c = 0; for (int i = 0; i < blocks; ++i) { a += h[2 * i]; c += h[2 * i + 1]; }
    a /= blocks; c /= blocks;
    const int waves = MODE == 3 ? 1 : 4;
    const int instr = MODE == 1 || MODE == 4 || MODE == 6 ? NR / 4 : (MODE == 2 ? NR / 2 : (MODE == 5 ? NR / 6 * 3 : NR));
    const double bytes = (double)waves * 64 * (MODE == 5 ? NR / 6 * 6 : NR) * 4;
    printf("%-28s blocks=%5d  kernel %.2f us  issue cycles %.0f  done cycles %.0f  -> %.1f clk/instr/wave (issue), %.1f B/clk/CU (done)\n", name, blocks,
           ms * 100.0, a, c, a / instr, bytes / c);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    for (int blocks : {256, 1024}) {
        run<0>("dword SoA [r][B]", blocks);
        run<1>("dwordx4 record [B][NR]", blocks);
        run<2>("dwordx2 record [B][NR]", blocks);
        run<3>("dword SoA, 1 wave/WG", blocks);
        run<4>("dwordx4 planes [r][B]x16B", blocks);
        run<5>("24B records 3 x dwordx2", blocks);
        run<6>("32B rows 2 x dwordx4", blocks);
    }
    return 0;
}
