// Does a wave64 VALU instruction whose upper 32 (or 48) lanes are masked off issue faster on gfx950?  One wave per SIMD (W=1) and two (W=2).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int ACTIVE>
__global__ __launch_bounds__(64) void k(double *out, double a, double b, long long *cyc, int slot)
{
    const int lane = threadIdx.x;
    if (lane >= ACTIVE) return;
    double x0 = a + lane * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7, y = b;
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
    long long t0 = clock64();
    for (int i = 0; i < N; i++)
    {
        x0 = __builtin_fma(x0, y, y); x1 = __builtin_fma(x1, y, y); x2 = __builtin_fma(x2, y, y); x3 = __builtin_fma(x3, y, y);
        x4 = __builtin_fma(x4, y, y); x5 = __builtin_fma(x5, y, y); x6 = __builtin_fma(x6, y, y); x7 = __builtin_fma(x7, y, y);
        asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0\n v_xor_b32 %0, %0, %2\n v_xor_b32 %1, %1, %3\n v_xor_b32 %2, %2, %0\n v_xor_b32 %3, %3, %1"
                     : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + i0 + i1 + i2 + i3;
    if (lane == 0 && blockIdx.x == 0) cyc[slot] = t1 - t0;
}
int main()
{
    double *out; long long *cyc, h[16] = {0};
    hipMalloc(&out, 8192 * 64 * 8); hipMalloc(&cyc, 128); hipMemset(cyc, 0, 128);
    // W=1: 1024 blocks of 64 (one wave per SIMD); W=2: 2048 blocks
    for (int rep = 0; rep < 2; rep++)
    {
        k<64><<<1024, 64>>>(out, 0.5, 0.999, cyc, 0); k<32><<<1024, 64>>>(out, 0.5, 0.999, cyc, 1); k<16><<<1024, 64>>>(out, 0.5, 0.999, cyc, 2);
        k<64><<<2048, 64>>>(out, 0.5, 0.999, cyc, 3); k<32><<<2048, 64>>>(out, 0.5, 0.999, cyc, 4); k<16><<<2048, 64>>>(out, 0.5, 0.999, cyc, 5);
        k<32><<<4096, 64>>>(out, 0.5, 0.999, cyc, 6); k<16><<<4096, 64>>>(out, 0.5, 0.999, cyc, 7);
    }
    hipDeviceSynchronize(); hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
        #define TIMEIT(A, BLK, label) hipEventRecord(e0); k<A><<<BLK, 64>>>(out, 0.5, 0.999, cyc, 15); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("%-28s wall %8.1f us\n", label, ms * 1e3);
        TIMEIT(64, 1024, "W=1 64 lanes (1024 waves)") TIMEIT(64, 2048, "W=2 64 lanes (2048 waves)") TIMEIT(64, 4096, "W=4 64 lanes (4096 waves)")
        TIMEIT(32, 2048, "W=2 32 lanes") TIMEIT(32, 4096, "W=4 32 lanes") TIMEIT(16, 4096, "W=4 16 lanes") TIMEIT(16, 8192, "W=8 16 lanes")
    }
    const char *names[] = {"W=1, 64 lanes", "W=1, 32 lanes", "W=1, 16 lanes", "W=2, 64 lanes", "W=2, 32 lanes", "W=2, 16 lanes", "W=4, 32 lanes", "W=4, 16 lanes"};
    for (int i = 0; i < 8; i++) printf("%-16s %8.2f cycles per iteration (8 fp64 fma + 8 v_xor_b32) of ONE wave\n", names[i], (double)h[i] / N);
    return 0;
}
