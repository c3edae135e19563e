// Does v_mfma_f64_16x16x4_f64 accumulate like an ascending chain of fused multiply-adds (the contract of oracle/lexlse_oracle.h)?
// D = A(16x4) * B(4x16) + C on random data with wide exponent spread, compared bitwise with candidate orders on the host.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef double double4v __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, const double *C, double *D)
{
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)]; // A[row l&15][k = l>>4]
    const double b = B[(l >> 4) * 16 + (l & 15)]; // B[k = l>>4][col l&15]
    double4v c;
    for (int r = 0; r < 4; r++) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)]; // row = (lane>>4) + 4*reg, col = lane&15
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
static double rnd() { return (drand48() - 0.5) * std::ldexp(1.0, (int)(drand48() * 20) - 10); }
int main()
{
    double hA[64], hB[64], hC[256], hD[256], *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    long match[5] = {0, 0, 0, 0, 0}, total = 0;
    srand48(12345);
    for (int trial = 0; trial < 200; trial++)
    {
        for (double &v : hA) v = rnd();
        for (double &v : hB) v = rnd();
        for (double &v : hC) v = rnd();
        hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 2048, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC, dD);
        hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++)
            {
                const double *a = hA + i * 4, c = hC[i * 16 + j];
                double b[4];
                for (int kk = 0; kk < 4; kk++) b[kk] = hB[kk * 16 + j];
                double cand[5];
                cand[0] = std::fma(a[3], b[3], std::fma(a[2], b[2], std::fma(a[1], b[1], std::fma(a[0], b[0], c)))); // ascending chain from C
                cand[1] = std::fma(a[0], b[0], std::fma(a[1], b[1], std::fma(a[2], b[2], std::fma(a[3], b[3], c)))); // descending chain from C
                cand[2] = (std::fma(a[1], b[1], a[0] * b[0]) + std::fma(a[3], b[3], a[2] * b[2])) + c;                // pairwise tree, C last
                cand[3] = std::fma(a[3], b[3], std::fma(a[2], b[2], std::fma(a[1], b[1], a[0] * b[0]))) + c;          // chain from 0, C last
                cand[4] = (double)((long double)a[0] * b[0] + (long double)a[1] * b[1] + (long double)a[2] * b[2] + (long double)a[3] * b[3] + (long double)c); // one rounding (approx.)
                for (int q = 0; q < 5; q++) match[q] += std::memcmp(&cand[q], &hD[i * 16 + j], 8) == 0;
                total++;
            }
    }
    const char *names[] = {"ascending fma chain from C", "descending fma chain from C", "pairwise tree + C", "fma chain from 0, then + C", "single rounding (80-bit host sum)"};
    for (int q = 0; q < 5; q++) printf("%-36s matches %ld of %ld results\n", names[q], match[q], total);
    return 0;
}
