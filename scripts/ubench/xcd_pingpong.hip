// Does a plain global store leave the CU at once, or only when the wavefront issues its next vector-memory instruction?  Two one-wavefront
// workgroups on ONE XCD: the initiator stores a tagged granule, then does D cycles of work without vector-memory instructions, then polls for
// the reply; the responder polls and replies at once.  If the store leaves at once the round trip is max(D, 2 L); if it is held back it is D + 2 L.
// hipcc --offload-arch=gfx950 -O3 -o xcd_pingpong xcd_pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int M> __device__ __forceinline__ void st16(void *p, u32x4 v)
{
    if (M == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (M == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int M> __device__ __forceinline__ u32x4 ld16(const void *p)
{
    u32x4 v;
    if (M == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (M == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
template <int M> __global__ __launch_bounds__(64) void pp(u32x4 *rec, unsigned *claim, int steps, int delay, int samexcd, long long *out)
{
    const unsigned x = xcc_id();
    int role         = -1;
    if (samexcd)
    {
        if (x != 0) return;
        unsigned c = 0;
        if (threadIdx.x == 0) c = atomicAdd(claim, 1u);
        role = __builtin_amdgcn_readfirstlane((int)c);
    }
    else
        role = blockIdx.x == 0 ? 0 : (blockIdx.x == 1 ? 1 : 2);
    if (role > 1) return;
    u32x4 *ping = rec, *pong = rec + 64;
    long long total = 0;
    for (int k = 1; k <= steps; k++)
    {
        if (role == 0)
        {
            const long long t0 = clock64();
            u32x4 q;
            q.x = q.y = q.z = 0, q.w = (unsigned)k;
            if (threadIdx.x == 0) st16<M>(ping, q);
            while (clock64() - t0 < delay) __builtin_amdgcn_s_sleep(1);
            for (unsigned spin = 0; spin < (1u << 20); spin++)
                if (ld16<M>(pong).w == (unsigned)k) break;
            total += clock64() - t0;
        }
        else
        {
            for (unsigned spin = 0; spin < (1u << 20); spin++)
                if (ld16<M>(ping).w == (unsigned)k) break;
            u32x4 q;
            q.x = q.y = q.z = 0, q.w = (unsigned)k;
            if (threadIdx.x == 0) st16<M>(pong, q);
        }
    }
    if (role == 0 && threadIdx.x == 0) out[0] = total / steps;
}
template <int M> static void run(const char *name, int samexcd, int delay)
{
    u32x4 *rec;
    unsigned *claim;
    long long *out, h = 0;
    hipMalloc(&rec, 4096), hipMalloc(&claim, 4), hipMalloc(&out, 8);
    hipMemset(rec, 0, 4096), hipMemset(claim, 0, 4), hipMemset(out, 0, 8);
    hipLaunchKernelGGL(pp<M>, dim3(16), dim3(64), 0, 0, rec, claim, 2000, delay, samexcd, out);
    hipDeviceSynchronize();
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("%-40s delay %5d cycles: round trip %6lld cycles\n", name, delay, h);
    hipFree(rec), hipFree(claim), hipFree(out);
}
int main()
{
    for (int d : {0, 500, 1000, 2000, 4000})
    {
        run<0>("one XCD, plain store / sc1 load", 1, d);
        run<1>("one XCD, sc0 sc1 store / load", 1, d);
        run<1>("two XCDs, sc0 sc1 store / load", 0, d);
    }
    return 0;
}
