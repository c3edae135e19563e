// Dependent-chain latencies of the fp64 primitives the l-QR pivot step is made of (one wave, one SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
__device__ __forceinline__ double rdlane(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <int MODE>
__global__ void k(double *out, double a, double b, long long *cyc)
{
    double x = a + threadIdx.x * 1e-9, y = b;
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++)
    {
        if (MODE == 0) x = __builtin_fma(x, y, y);                // dependent fma
        if (MODE == 1) x = __builtin_fmax(x, y) + 0.0 * x;       // (placeholder)
        if (MODE == 2) x = sqrt(x + 1.0);                        // IEEE sqrt sequence
        if (MODE == 3) x = y / (x + 2.0);                        // IEEE division sequence
        if (MODE == 4) x = rdlane(x, 5) + y;                     // readlane round trip + add
        if (MODE == 5)                                           // DPP mov pair + max (one butterfly step)
        {
            int lo = __double2loint(x), hi = __double2hiint(x);
            lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false);
            hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
            x  = __builtin_fmax(x, __hiloint2double(hi, lo)) + y;
        }
        if (MODE == 6) x = x * y;                                 // dependent mul
        if (MODE == 7) x = x + y;                                 // dependent add
    }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
template <int MODE>
__global__ void k2(double *out, double a, double b, long long *cyc) // 4 independent chains (throughput)
{
    double x0 = a + threadIdx.x * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y = b;
    long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < N; i++)
    {
        x0 = __builtin_fma(x0, y, y); x1 = __builtin_fma(x1, y, y); x2 = __builtin_fma(x2, y, y); x3 = __builtin_fma(x3, y, y);
    }
    long long t1 = clock64();
    out[threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) cyc[8] = t1 - t0;
}
int main()
{
    double *out; long long *cyc, h[16] = {0};
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8); hipMemset(cyc, 0, 128);
    for (int rep = 0; rep < 2; rep++) {
    k<0><<<1, 64>>>(out, 0.5, 0.999, cyc); k<2><<<1, 64>>>(out, 0.5, 0.999, cyc); k<3><<<1, 64>>>(out, 0.5, 0.999, cyc);
    k<4><<<1, 64>>>(out, 0.5, 0.999, cyc); k<5><<<1, 64>>>(out, 0.5, 0.999, cyc); k<6><<<1, 64>>>(out, 0.5, 0.999, cyc); k<7><<<1, 64>>>(out, 0.5, 0.999, cyc);
    k2<0><<<1, 64>>>(out, 0.5, 0.999, cyc); }
    hipDeviceSynchronize(); hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    const char *names[] = {"dependent v_fma_f64", "-", "IEEE sqrt(x+1) (+1 add)", "IEEE y/(x+2) (+1 add)", "readlane x2 + add", "dpp butterfly step (2 dpp mov + fmax + add)", "dependent v_mul_f64", "dependent v_add_f64", "4 independent fma chains (per 4 fma)"};
    for (int i = 0; i < 9; i++) if (h[i]) printf("%-48s %8.1f cycles per iteration\n", names[i], (double)h[i] / N);
    return 0;
}
