// Calibration for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 in THIS project's access pattern
// (one dword per lane, rows of a [rows][B] block; MI355X_MICROARCH.md: widths other than 16 B/lane are uncalibrated).
// read_rows reads NR rows of B dwords (known bytes = NR * B * 4) and writes B dwords; write_rows writes NR rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int NR = 64;

__global__ void __launch_bounds__(256) read_rows(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) acc += in[(size_t)r * B + b];
    out[b] = acc;
}
__global__ void __launch_bounds__(256) write_rows(uint32_t* __restrict__ out, int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int r = 0; r < NR; ++r) out[(size_t)r * B + b] = r + b;
}

int main() {
    const int B = 65536 * 4;
    uint32_t *a, *o;
    CK(hipMalloc(&a, (size_t)B * NR * 4)); CK(hipMalloc(&o, (size_t)B * 4));
    CK(hipMemset(a, 1, (size_t)B * NR * 4));
    for (int i = 0; i < 5; ++i) {
        read_rows<<<B / 256, 256>>>(a, o, B);
        write_rows<<<B / 256, 256>>>(a, B);
    }
    CK(hipDeviceSynchronize());
    printf("known bytes per launch: read_rows reads %zu writes %zu; write_rows writes %zu\n", (size_t)B * NR * 4, (size_t)B * 4, (size_t)B * NR * 4);
    return 0;
}
