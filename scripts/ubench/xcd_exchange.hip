// All-to-all record exchange between G one-wavefront workgroups, the hand-off of the large path's in-launch pivot loop
// (lqr_large.hip, fast_level_persist): every workgroup publishes a tagged 16-byte record, polls the G records of the step until all carry
// the step's tag, repeats.  Measured: ns per step (a) with the workgroups spread over all XCDs and system-scope accesses (sc0 sc1: what the
// kernel does now), (b) with all G workgroups on ONE XCD (the launch is 8 x larger; workgroups on other XCDs leave at once) and accesses that
// only have to be coherent in that XCD's L2 (sc0 / sc1 / sc0 sc1).
// hipcc --offload-arch=gfx950 -O3 -o xcd_exchange xcd_exchange.hip && ./xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define STRIDE 16 // records 256 bytes apart
template <int M> __device__ __forceinline__ void st16(void *p, u32x4 v)
{
    if (M == 0) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (M == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (M == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (M == 3 || M == 4 || M == 6 || M == 8) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (M == 9) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (M == 10) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (M == 11) // payload as a plain store, then the tag word as an atomic exchange behind it (same wave, same line: in order at the L2?)
    {
        unsigned old;
        u32x4 w = v;
        w.w = 0;
        asm volatile("global_store_dwordx3 %1, %2, off\n\tglobal_atomic_swap %0, %3, %4, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(__builtin_shufflevector(w, w, 0, 1, 2)), "v"((char *)p + 12), "v"(v.w) : "memory");
    }
    if (M == 12) // two 64-bit exchanges: {x, y} then {z, tag}
    {
        unsigned long long o1, o2;
        unsigned long long lo = ((unsigned long long)v.y << 32) | v.x, hi = ((unsigned long long)v.w << 32) | v.z;
        asm volatile("global_atomic_swap_x2 %0, %2, %3, off sc0\n\tglobal_atomic_swap_x2 %1, %4, %5, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(o1), "=&v"(o2) : "v"(p), "v"(lo), "v"((char *)p + 8), "v"(hi) : "memory");
    }
    if (M == 5 || M == 7) // the tag word travels as an agent-scope atomic exchange (executed in the L2)
    {
        unsigned old;
        asm volatile("global_atomic_swap %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"((char *)p + 12), "v"(v.w) : "memory");
    }
}
template <int M> __device__ __forceinline__ u32x4 ld16(const void *p)
{
    u32x4 v;
    if (M == 0) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (M == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (M == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (M == 3) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (M == 4) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (M == 5 || M == 8 || M == 9 || M == 10 || M == 11 || M == 12) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (M == 6 || M == 7) // poll = atomic OR of zero with return: served by the L2
    {
        unsigned w, z = 0;
        asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"((const char *)p + 12), "v"(z) : "memory");
        v.x = v.y = v.z = 0, v.w = w;
    }
    return v;
}
__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
struct Ctl
{
    unsigned claim, abort, pad[14];
    unsigned perxcc[16];
};
// one_xcd < 0: every workgroup takes part (ids = blockIdx); else only those on that XCD, ids claimed in arrival order
template <int M>
__global__ __launch_bounds__(64) void exchange(u32x4 *rec, Ctl *ctl, int G, int steps, int one_xcd, long long *out, unsigned *xcc_of)
{
    const unsigned lane = threadIdx.x;
    unsigned id         = blockIdx.x;
    const unsigned x    = xcc_id();
    if (one_xcd >= 0)
    {
        if (lane == 0) atomicAdd(&ctl->perxcc[x], 1u);
        if ((int)x != one_xcd) return;
        unsigned c = 0;
        if (lane == 0) c = atomicAdd(&ctl->claim, 1u);
        id = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
        if (id >= (unsigned)G) return;
    }
    else if (id >= (unsigned)G)
        return;
    if (lane == 0) xcc_of[id] = x;
    long long t0 = 0;
    for (int k = 0; k < steps; k++)
    {
        if (k == steps / 4) t0 = (long long)wall_clock64();
        const unsigned tag = (unsigned)k + 1u;
        if (lane == 0)
        {
            u32x4 q;
            q.x = id, q.y = tag * 3u, q.z = 7u, q.w = tag;
            st16<M>(rec + ((size_t)(k & 1) * G + id) * STRIDE, q);
        }
        bool ok = false;
        for (unsigned spin = 0; spin < (1u << 16); spin++)
        {
            const u32x4 q   = ld16<M>(rec + ((size_t)(k & 1) * G + (lane < (unsigned)G ? lane : 0u)) * STRIDE);
            const bool mine = lane >= (unsigned)G || (q.w == tag && (M < 8 || q.y == tag * 3u));
            if (__ballot(!mine) == 0ull)
            {
                ok = true;
                break;
            }
            if ((spin & 255u) == 255u && __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
        if (!ok)
        {
            if (lane == 0) __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    const long long t1 = (long long)wall_clock64();
    if (lane == 0 && id == 0)
    {
        out[0] = t1 - t0;
        out[1] = steps - steps / 4;
    }
}
template <int M> static void run(const char *name, int G, int one_xcd, int steps)
{
    u32x4 *rec;
    Ctl *ctl;
    long long *out;
    unsigned *xo;
    hipMalloc(&rec, sizeof(u32x4) * STRIDE * 2 * 256);
    hipMalloc(&ctl, sizeof(Ctl));
    hipMalloc(&out, 16);
    hipMalloc(&xo, 4 * 256);
    hipMemset(rec, 0, sizeof(u32x4) * STRIDE * 2 * 256);
    hipMemset(ctl, 0, sizeof(Ctl));
    hipMemset(out, 0, 16);
    hipMemset(xo, 0xff, 4 * 256);
    const int grid = one_xcd >= 0 ? 8 * G : G;
    hipLaunchKernelGGL(exchange<M>, dim3(grid), dim3(64), 0, 0, rec, ctl, G, steps, one_xcd, out, xo);
    hipError_t e = hipDeviceSynchronize();
    long long h[2];
    Ctl hc;
    unsigned hx[256];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
    hipMemcpy(hx, xo, 4 * 256, hipMemcpyDeviceToHost);
    int nx[16] = {0};
    for (int i = 0; i < G; i++)
        if (hx[i] < 16) nx[hx[i]]++;
    printf("%-44s G=%3d  %s  %8.1f ns per step   (workgroups per XCD:", name, G, (e != hipSuccess || hc.abort || h[1] == 0) ? "FAILED" : "ok    ", h[1] ? h[0] * 10.0 / h[1] : 0.0);
    for (int i = 0; i < 8; i++) printf(" %d", nx[i]);
    if (one_xcd >= 0)
    {
        printf(" | launched per XCD:");
        for (int i = 0; i < 8; i++) printf(" %u", hc.perxcc[i]);
    }
    printf(")\n");
    fflush(stdout);
    hipFree(rec), hipFree(ctl), hipFree(out), hipFree(xo);
}
int main()
{
    const int steps = 4000;
    for (int G : {2, 8, 32, 64})
    {
        run<0>("all XCDs, sc0 sc1", G, -1, steps);
        run<1>("all XCDs, sc1", G, -1, steps);
        run<12>("all XCDs, two 64-bit swaps, sc1 load", G, -1, steps);
    }
    for (int G : {2, 16, 32, 48, 64})
    {
        run<0>("one XCD, sc0 sc1", G, 0, steps);
        run<1>("one XCD, sc1", G, 0, steps);
        run<2>("one XCD, sc0", G, 0, steps);
        run<3>("one XCD, no scope bits", G, 0, steps);
        run<4>("one XCD, plain store, buffer_inv sc1 + load", G, 0, steps);
        run<5>("one XCD, atomic swap, sc1 load", G, 0, steps);
        run<8>("one XCD, plain store, sc1 load", G, 0, steps);
        run<9>("one XCD, sc0 store, sc1 load", G, 0, steps);
        run<10>("one XCD, nt store, sc1 load", G, 0, steps);
        run<11>("one XCD, store x3 + atomic swap tag, sc1 load", G, 0, steps);
        run<12>("one XCD, two 64-bit swaps, sc1 load", G, 0, steps);
    }
    run<2>("all XCDs, sc0 (expected to fail or stall)", 8, -1, 400);
    return 0;
}
