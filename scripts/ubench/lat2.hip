// Single-wave (one wave on an idle SIMD) dependent-chain latencies and issue costs of the primitives the four-problems-per-
// wavefront l-QR kernel (lqr_quad_impl.h) is built from.  cycles = s_memtime ticks per iteration of the measured pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
extern "C" __device__ double lexls_update_dpp_f64(double, double, int, int, int, bool) __asm("llvm.amdgcn.update.dpp.f64");
template <int R> __device__ __forceinline__ double bc(double v) { return lexls_update_dpp_f64(0.0, v, 0x150 + R, 0xf, 0xf, true); }
__device__ __forceinline__ double vmax(double x, double y) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
template <int CTRL> __device__ __forceinline__ double dpp_max(double v)
{
    const int lo2 = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi2 = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return vmax(v, __hiloint2double(hi2, lo2));
}
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, double a, double b, long long *cyc, const double *g)
{
    __shared__ double sm[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) sm[i] = a + i;
    __syncthreads();
    double x = a + lane * 1e-9, y = b, z = a * 0.5, w = b * 0.25;
    double h[12];
    for (int r = 0; r < 12; r++) h[r] = a + r * 1e-3 + lane;
    int ix = lane;
    long long t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N; i++)
    {
        if (MODE == 0) x = __builtin_fma(x, y, y);
        if (MODE == 1) x = __builtin_fma(bc<3>(x), y, y);                    // v_mov_b64_dpp row_newbcast + fma, dependent
        if (MODE == 2) x = sqrt(x + 1.0);
        if (MODE == 3) x = y / (x + 2.0);
        if (MODE == 4) { x = dpp_max<0xB1>(x); x = dpp_max<0x4E>(x); x = dpp_max<0x141>(x); x = dpp_max<0x140>(x); x += y; } // 16-lane f64 max butterfly
        if (MODE == 5)                                                         // 16-lane u32 min butterfly
        {
            ix = min(ix, __builtin_amdgcn_update_dpp(ix, ix, 0xB1, 0xf, 0xf, false));
            ix = min(ix, __builtin_amdgcn_update_dpp(ix, ix, 0x4E, 0xf, 0xf, false));
            ix = min(ix, __builtin_amdgcn_update_dpp(ix, ix, 0x141, 0xf, 0xf, false));
            ix = min(ix, __builtin_amdgcn_update_dpp(ix, ix, 0x140, 0xf, 0xf, false));
            ix += lane;
        }
        if (MODE == 6)                                                         // ds_bpermute of a double (2 x b32), dependent
        {
            const int src = ((lane & 48) + 5) << 2;
            const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(x));
            const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(x));
            x = __hiloint2double(hi, lo) + y;
        }
        if (MODE == 7)                                                         // LDS round trip: one lane per group writes 12 doubles (b128), all read one
        {
            if ((lane & 15) == 5)
            {
                double2 *p = reinterpret_cast<double2 *>(sm + (lane >> 4) * 16);
                p[0] = make_double2(x, h[1]); p[1] = make_double2(h[2], h[3]); p[2] = make_double2(h[4], h[5]);
                p[3] = make_double2(h[6], h[7]); p[4] = make_double2(h[8], h[9]); p[5] = make_double2(h[10], h[11]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            asm volatile("" ::: "memory");
            x = sm[(lane >> 4) * 16 + ((lane + i) & 7)] + y;
            asm volatile("" ::: "memory");
        }
        if (MODE == 8) x = vmax(x, y) + z;                                    // v_max_f64 + add dependent
        if (MODE == 9) x = x * y;
        if (MODE == 10)                                                        // 12-term dot with DPP broadcasts (chain) — e from x
        {
            double t = 0.0;
            t = __builtin_fma(bc<1>(x), h[1], t); t = __builtin_fma(bc<2>(x), h[2], t); t = __builtin_fma(bc<3>(x), h[3], t);
            t = __builtin_fma(bc<4>(x), h[4], t); t = __builtin_fma(bc<5>(x), h[5], t); t = __builtin_fma(bc<6>(x), h[6], t);
            t = __builtin_fma(bc<7>(x), h[7], t); t = __builtin_fma(bc<8>(x), h[8], t); t = __builtin_fma(bc<9>(x), h[9], t);
            t = __builtin_fma(bc<10>(x), h[10], t); t = __builtin_fma(bc<11>(x), h[11], t);
            x = t * 1e-3 + y;
        }
        if (MODE == 11)                                                        // 24 independent fma (issue rate, one wave)
        {
#pragma unroll
            for (int r = 0; r < 12; r++) h[r] = __builtin_fma(h[r], y, z);
#pragma unroll
            for (int r = 0; r < 12; r++) h[r] = __builtin_fma(h[r], w, y);
        }
        if (MODE == 12)                                                        // 12 independent v_mov_b64_dpp + 12 fma
        {
            h[0] = __builtin_fma(bc<0>(z), h[0], y); h[1] = __builtin_fma(bc<1>(z), h[1], y); h[2] = __builtin_fma(bc<2>(z), h[2], y);
            h[3] = __builtin_fma(bc<3>(z), h[3], y); h[4] = __builtin_fma(bc<4>(z), h[4], y); h[5] = __builtin_fma(bc<5>(z), h[5], y);
            h[6] = __builtin_fma(bc<6>(z), h[6], y); h[7] = __builtin_fma(bc<7>(z), h[7], y); h[8] = __builtin_fma(bc<8>(z), h[8], y);
            h[9] = __builtin_fma(bc<9>(z), h[9], y); h[10] = __builtin_fma(bc<10>(z), h[10], y); h[11] = __builtin_fma(bc<11>(z), h[11], y);
        }
        if (MODE == 13)                                                        // 12 uniform-per-group ds_read_b64 + 12 fma (independent)
        {
            const double *p = sm + (lane >> 4) * 64 + (i & 7);
#pragma unroll
            for (int r = 0; r < 12; r++) h[r] = __builtin_fma(p[r * 2], h[r], y);
        }
        if (MODE == 14)                                                        // dependent LDS read chain (pointer chase)
        {
            ix = (int)sm[(ix & 63) + 64 * (i & 1)] & 1023;
        }
        if (MODE == 15)                                                        // dependent global load chain, L2-resident (64 KB table)
        {
            ix = (int)g[(ix & 8191)] & 8191;
        }
        if (MODE == 16)                                                        // cndmask select of 12 doubles (slot select) — issue cost
        {
            const bool c = (lane + i) & 1;
#pragma unroll
            for (int r = 0; r < 12; r++) h[r] = c ? h[r] : h[(r + 1) % 12];
        }
        if (MODE == 17)                                                        // fused asm v_fmac_f64_dpp chain of 11 (one slot)
        {
            double t = 0.0;
            asm volatile("s_nop 1\n"
                         "v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %0, %1, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %0, %1, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %0, %1, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %0, %1, %6 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %0, %1, %7 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %0, %1, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %0, %1, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %0, %1, %10 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %0, %1, %11 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %0, %1, %12 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(t) : "v"(x), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(h[4]), "v"(h[5]), "v"(h[6]), "v"(h[7]), "v"(h[8]), "v"(h[9]), "v"(h[10]), "v"(h[11]));
            x = t * 1e-3 + y;
        }
        if (MODE == 18)                                                        // 12 independent fused v_fmac_f64_dpp (issue)
        {
            asm volatile("s_nop 1\n"
                         "v_fmac_f64_dpp %0, %12, %13 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %1, %12, %13 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %2, %12, %13 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %3, %12, %13 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %4, %12, %13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %5, %12, %13 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %6, %12, %13 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %7, %12, %13 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %8, %12, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %9, %12, %13 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %10, %12, %13 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %11, %12, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]), "+v"(h[8]), "+v"(h[9]), "+v"(h[10]), "+v"(h[11])
                         : "v"(z), "v"(w));
        }
        if (MODE == 19) x = x + y;
    }
    long long t1 = clock64();
    double s = x + ix;
    for (int r = 0; r < 12; r++) s += h[r];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
template <int M> void run(double *out, long long *cyc, const double *g) { k<M><<<1, 64>>>(out, 0.5, 0.999, cyc, g); k<M><<<1, 64>>>(out, 0.5, 0.999, cyc, g); }
int main()
{
    double *out, *g; long long *cyc, h[32] = {0};
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 32 * 8); hipMemset(cyc, 0, 256);
    hipMalloc(&g, 8192 * 8);
    { double hg[8192]; for (int i = 0; i < 8192; i++) hg[i] = (double)((i * 977 + 13) & 8191); hipMemcpy(g, hg, sizeof(hg), hipMemcpyHostToDevice); }
    run<0>(out, cyc, g); run<1>(out, cyc, g); run<2>(out, cyc, g); run<3>(out, cyc, g); run<4>(out, cyc, g); run<5>(out, cyc, g); run<6>(out, cyc, g);
    run<7>(out, cyc, g); run<8>(out, cyc, g); run<9>(out, cyc, g); run<10>(out, cyc, g); run<11>(out, cyc, g); run<12>(out, cyc, g); run<13>(out, cyc, g);
    run<14>(out, cyc, g); run<15>(out, cyc, g); run<16>(out, cyc, g); run<17>(out, cyc, g); run<18>(out, cyc, g); run<19>(out, cyc, g);
    hipDeviceSynchronize(); hipMemcpy(h, cyc, 256, hipMemcpyDeviceToHost);
    const char *names[] = {"dependent v_fma_f64", "dependent bcast(v_mov_b64_dpp)+fma", "IEEE sqrt(x+1) (+add)", "IEEE y/(x+2) (+add)", "16-lane f64 max butterfly (+add)",
                           "16-lane u32 min butterfly (+add)", "ds_bpermute double (+add)", "LDS: 6xb128 write by 1 lane/group -> read (+add)", "v_max_f64 + add", "dependent v_mul_f64",
                           "11-term bcast dot chain (+mul,add)", "24 independent fma", "12 x (bcast mov + fma) independent", "12 x (ds_read_b64 uniform + fma)", "dependent ds_read chain",
                           "dependent global load chain (L2)", "12-double cndmask select", "fused asm v_fmac_f64_dpp 11-chain (+mul,add)", "12 fused v_fmac_f64_dpp independent", "dependent v_add_f64"};
    for (int i = 0; i < 20; i++) printf("%-52s %8.1f cycles per iteration\n", names[i], (double)h[i] / N);
    return 0;
}
