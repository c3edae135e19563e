// Does v_fmac_f64 take a DPP row broadcast on gfx950?  d += bcast(x of lane L of the 16-lane row) * y, against the two-instruction form.
// hipcc --offload-arch=gfx950 -O3 -o fmac_dpp fmac_dpp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double *o, const double *a, const double *b, long long *cyc)
{
    const int l = threadIdx.x;
    double d = o[l], x = a[l], y = b[l];
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(x), "v"(y));
    o[l] = d;
    // issue cost: 256 dependent-free fmac_dpp on 8 accumulators vs mov_dpp + fma
    double acc[8];
    for (int i = 0; i < 8; i++) acc[i] = x + i;
    long long t0 = clock64();
    for (int it = 0; it < 32; it++)
        for (int i = 0; i < 8; i++) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(x), "v"(y));
    long long t1 = clock64();
    // hazard test: a VGPR written by the PREVIOUS instruction is read through DPP (the ISA asks for two wait states between them; does the
    // hardware interlock?)  1024 rounds, every round checked against the two-instruction form computed afterwards
    double hz = 0.0, hzref = 0.0;
    for (int it = 0; it < 1024; it++)
    {
        double p = x + it, q = y - it, d1, accf = 1.0 + it;
        asm volatile("v_fma_f64 %0, %2, %3, %2\n\tv_fmac_f64_dpp %1, %0, %3 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "=&v"(d1), "+v"(accf) : "v"(p), "v"(q));
        hz += accf;
        const double d2 = __builtin_fma(p, q, p);
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(d2), 0x157, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(d2), 0x157, 0xf, 0xf, false);
        hzref += __builtin_fma(__hiloint2double(hi, lo), q, 1.0 + it);
    }
    o[128 + l] = hz - hzref;
    double s = 0;
    for (int i = 0; i < 8; i++) s += acc[i];
    o[64 + l] = s;
    if (l == 0) cyc[0] = t1 - t0;
}
int main()
{
    double h[192], ha[64], hb[64], *d, *a, *b;
    long long *c, hc = 0;
    for (int i = 0; i < 64; i++) h[i] = 1.0 + i, ha[i] = 0.5 * i - 3, hb[i] = 2.0 + 0.25 * i;
    hipMalloc(&d, sizeof(h)), hipMalloc(&a, sizeof(ha)), hipMalloc(&b, sizeof(hb)), hipMalloc(&c, 8);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice), hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice), hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, b, c);
    double r[192];
    hipMemcpy(r, d, sizeof(r), hipMemcpyDeviceToHost), hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++)
    {
        const double want = fma(ha[(i & ~15) + 3], hb[i], h[i]);
        if (r[i] != want) bad++;
    }
    int hzbad = 0;
    for (int i = 0; i < 64; i++) hzbad += r[128 + i] != 0.0;
    printf("DPP read of a register written by the previous instruction: %s (%d lanes differ)\n", hzbad ? "HAZARD (stale value read)" : "interlocked / correct", hzbad);
    printf("v_fmac_f64_dpp row_newbcast: %s (%d of 64 lanes differ); 256 independent-accumulator instructions: %lld cycles = %.1f per instruction\n", bad ? "WRONG" : "correct", bad, hc, hc / 256.0);
    return bad != 0;
}
