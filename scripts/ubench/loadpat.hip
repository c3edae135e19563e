// How fast can one wavefront per SIMD (1024 on the chip, four problems each) pull the rows of ONE LEVEL of its IK problems (12 rows of 41
// columns at a 480-byte column stride) out of HBM?  Cycles per level for several lane -> address patterns, all waves loading at once.
//   0: 16 B per lane, 3 consecutive lanes = 48 contiguous bytes (6 rows of a column), columns in consecutive lane triples   [lqr_qtol]
//   1: 16 B per lane, 6 consecutive lanes = 96 contiguous bytes (12 rows of a column)
//   2: 16 B per lane, lane = column (480-byte lane stride), 6 loads per 16 columns                                         [lqr_quad]
//   3: 8 B per lane, 12 consecutive lanes = 96 contiguous bytes
//   4: 16 B per lane, fully contiguous: the whole problem block of the wave (all levels), 1 KB per instruction
//   5: as 0 through LDS-DMA (global_load_lds_dwordx4)
//   6: as 1 through LDS-DMA
// usage: loadpat <mode> [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int N = 40, CAP = 60, PST = CAP * (N + 1); // doubles per problem

template <int MODE>
__global__ __launch_bounds__(64) void k(const double *in, double *out, unsigned long long *cyc, int levels)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x, g = lane >> 4, gl = lane & 15;
    const double *inw = in + (size_t)blockIdx.x * 4 * PST;
    double acc = 0.0;
    unsigned long long t_issue = 0, t_total = 0;
    for (int k = 0; k < levels; k++)
    {
        const int F = 12 * k;
        const unsigned long long t0 = clock64();
        unsigned long long t1;
        if (MODE == 0 || MODE == 1)
        {
            constexpr int PC = MODE == 0 ? 3 : 6; // 16-byte pieces per contiguous run
            constexpr int NI = MODE == 0 ? 16 : 16;
            d2 v[NI];
#pragma unroll
            for (int i = 0; i < NI; i++)
            {
                int col, m, h = 0;
                if (MODE == 0)
                {
                    h      = i / 8;
                    int ch = 16 * (i % 8) + gl;
                    ch     = ch < 123 ? ch : 122;
                    col = ch / 3, m = ch - 3 * col;
                }
                else
                {
                    int ch = 16 * i + gl;
                    ch     = ch < 246 ? ch : 245;
                    col = ch / 6, m = ch - 6 * col;
                }
                v[i] = *reinterpret_cast<const d2 *>(inw + g * PST + col * CAP + F + 6 * h + 2 * m);
            }
            t1 = clock64();
#pragma unroll
            for (int i = 0; i < NI; i++) acc += v[i].x + v[i].y;
        }
        else if (MODE == 2)
        {
            d2 v[18];
#pragma unroll
            for (int s = 0; s < 3; s++)
            {
                int col = 16 * s + gl - 7;
                col     = col < 0 ? 0 : (col > N ? N : col);
#pragma unroll
                for (int r = 0; r < 6; r++) v[6 * s + r] = *reinterpret_cast<const d2 *>(inw + g * PST + col * CAP + F + 2 * r);
            }
            t1 = clock64();
#pragma unroll
            for (int i = 0; i < 18; i++) acc += v[i].x + v[i].y;
        }
        else if (MODE == 3)
        {
            double v[32];
#pragma unroll
            for (int i = 0; i < 32; i++)
            {
                int ch = 16 * i + gl;
                ch     = ch < 492 ? ch : 491;
                const int col = ch / 12, m = ch - 12 * col;
                v[i] = inw[g * PST + col * CAP + F + m];
            }
            t1 = clock64();
#pragma unroll
            for (int i = 0; i < 32; i++) acc += v[i];
        }
        else if (MODE == 4)
        {
            // a quarter of the whole block per "level": 4 * 19680 B = 78720 B = 76.9 KB -> 20 instructions of 1 KB per level
            d2 v[20];
#pragma unroll
            for (int i = 0; i < 20; i++)
            {
                int idx = (k * 20 + i) * 64 + lane; // 16-byte units
                idx     = idx < 4 * PST / 2 ? idx : 4 * PST / 2 - 1;
                v[i]    = *reinterpret_cast<const d2 *>(inw + 2 * idx);
            }
            t1 = clock64();
#pragma unroll
            for (int i = 0; i < 20; i++) acc += v[i].x + v[i].y;
        }
        else
        {
            constexpr int PC = MODE == 5 ? 3 : 6;
#pragma unroll
            for (int i = 0; i < 16; i++)
            {
                int col, m, h = 0;
                if (MODE == 5)
                {
                    h      = i / 8;
                    int ch = 16 * (i % 8) + gl;
                    ch     = ch < 123 ? ch : 122;
                    col = ch / 3, m = ch - 3 * col;
                }
                else
                {
                    int ch = 16 * i + gl;
                    ch     = ch < 246 ? ch : 245;
                    col = ch / 6, m = ch - 6 * col;
                }
                __builtin_amdgcn_global_load_lds(inw + g * PST + col * CAP + F + 6 * h + 2 * m, (__attribute__((address_space(3))) void *)(sm + 128 * i), 16, 0, 0);
            }
            t1 = clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += sm[lane] + sm[lane + 1024];
        }
        asm volatile("" : "+v"(acc));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = clock64();
        t_issue += t1 - t0;
        t_total += t2 - t0;
        // some arithmetic between the levels (as the kernel has): ~2000 cycles
        for (int it = 0; it < 60; it++)
        {
            acc = __builtin_fma(acc, 1.0000001, 1e-9); acc = __builtin_fma(acc, 0.9999999, 1e-9); acc = __builtin_fma(acc, 1.0000001, 1e-9); acc = __builtin_fma(acc, 0.9999999, 1e-9);
            acc = __builtin_fma(acc, 1.0000001, 1e-9); acc = __builtin_fma(acc, 0.9999999, 1e-9); acc = __builtin_fma(acc, 1.0000001, 1e-9); acc = __builtin_fma(acc, 0.9999999, 1e-9);
        }
    }
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0)
    {
        cyc[2 * blockIdx.x]     = t_issue;
        cyc[2 * blockIdx.x + 1] = t_total;
    }
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0, rounds = argc > 2 ? atoi(argv[2]) : 5;
    const int batch = 4096, NB = 4;
    const size_t bytes = (size_t)batch * PST * 8;
    double *in[NB], *out;
    unsigned long long *cyc;
    std::vector<double> h((size_t)batch * PST);
    for (size_t i = 0; i < h.size(); i++) h[i] = (double)((i * 2654435761u) & 0xffff) * 1e-5;
    for (int b = 0; b < NB; b++)
    {
        hipMalloc(&in[b], bytes + 4096);
        hipMemcpy(in[b], h.data(), bytes, hipMemcpyHostToDevice);
    }
    hipMalloc(&out, 1024 * 64 * 8);
    hipMalloc(&cyc, 1024 * 2 * 8);
    std::vector<unsigned long long> hc(2048);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int r = 0; r < rounds; r++)
    {
        const double *p = in[r % NB];
        hipEventRecord(e0);
        const int levels = 4;
        switch (mode)
        {
        case 0: hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 40960, 0, p, out, cyc, levels); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), 40960, 0, p, out, cyc, levels); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(1024), dim3(64), 40960, 0, p, out, cyc, levels); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(1024), dim3(64), 40960, 0, p, out, cyc, levels); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(1024), dim3(64), 40960, 0, p, out, cyc, levels); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(1024), dim3(64), 40960, 0, p, out, cyc, levels); break;
        default: hipLaunchKernelGGL(k<6>, dim3(1024), dim3(64), 40960, 0, p, out, cyc, levels); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hc.data(), cyc, 2048 * 8, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> is(1024), to(1024);
        for (int i = 0; i < 1024; i++) is[i] = hc[2 * i], to[i] = hc[2 * i + 1];
        std::sort(is.begin(), is.end());
        std::sort(to.begin(), to.end());
        printf("mode %d round %d: kernel %.1f us; per level: issue %llu cycles, issue+wait %llu cycles (median over waves; max %llu)\n", mode, r, ms * 1e3,
               is[512] / levels, to[512] / levels, to[1023] / levels);
    }
    return 0;
}
