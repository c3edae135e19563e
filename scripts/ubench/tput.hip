// Issue-throughput of the instructions the wave kernels are made of, whole chip busy (W waves per SIMD).
// Reports ns per wave-instruction per SIMD (VALU/SALU) or per CU (LDS); divide by the clock period for cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 256
#define UNR 32
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, double a, double b, int lanesel)
{
    __shared__ double sm[512];
    const int lane = threadIdx.x;
    sm[lane] = a + lane; sm[lane + 64] = b; sm[lane + 128] = a; sm[lane + 192] = b;
    __syncthreads();
    double x0 = a + lane * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7, y = b;
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
    int acc = 0;
    for (int it = 0; it < ITER; it++)
    {
#pragma unroll
        for (int u = 0; u < UNR / 8; u++)
        {
            if (MODE == 0) // 8 independent v_fma_f64
            {
                x0 = __builtin_fma(x0, y, y); x1 = __builtin_fma(x1, y, y); x2 = __builtin_fma(x2, y, y); x3 = __builtin_fma(x3, y, y);
                x4 = __builtin_fma(x4, y, y); x5 = __builtin_fma(x5, y, y); x6 = __builtin_fma(x6, y, y); x7 = __builtin_fma(x7, y, y);
            }
            if (MODE == 1) // 8 v_readlane_b32 (independent), results consumed by s_add
            {
                int s;
                asm volatile("v_readlane_b32 %0, %1, 3\n s_add_i32 %2, %2, %0\n v_readlane_b32 %0, %1, 5\n v_readlane_b32 %0, %1, 7\n v_readlane_b32 %0, %1, 9\n"
                             "v_readlane_b32 %0, %1, 11\n v_readlane_b32 %0, %1, 13\n v_readlane_b32 %0, %1, 15\n v_readlane_b32 %0, %1, 17\n"
                             : "=&s"(s), "+v"(i0), "+s"(acc));
            }
            if (MODE == 2) // 8 v_mov_b32 (32-bit VALU)
            {
                asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7\n v_mov_b32 %0, %5\n v_mov_b32 %1, %6\n v_mov_b32 %2, %7\n v_mov_b32 %3, %4\n"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(lane), "v"(lanesel), "v"(acc), "v"(it));
            }
            if (MODE == 3) // 8 v_mov_b32 dpp (quad_perm)
            {
                asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
            }
            if (MODE == 4) // 8 v_mul_f64 independent
            {
                x0 = x0 * y; x1 = x1 * y; x2 = x2 * y; x3 = x3 * y; x4 = x4 * y; x5 = x5 * y; x6 = x6 * y; x7 = x7 * y;
            }
            if (MODE == 5 || MODE == 6 || MODE == 7) // 8 ds_read_b128, uniform address: all lanes / 16 lanes / 1 lane
            {
                const bool on = MODE == 5 ? true : (MODE == 6 ? lane < 16 : lane == lanesel);
                if (on)
                {
                    const double2 *p = reinterpret_cast<const double2 *>(sm) + (it & 7);
                    double2 v0 = p[0], v1 = p[8], v2 = p[16], v3 = p[24], v4 = p[32], v5 = p[40], v6 = p[48], v7 = p[56];
                    asm volatile("" ::"v"(v0.x), "v"(v1.x), "v"(v2.x), "v"(v3.x), "v"(v4.x), "v"(v5.x), "v"(v6.x), "v"(v7.x));
                    asm volatile("" ::"v"(v0.y), "v"(v1.y), "v"(v2.y), "v"(v3.y), "v"(v4.y), "v"(v5.y), "v"(v6.y), "v"(v7.y));
                }
            }
            if (MODE == 8) // 8 ds_read_b64, lane-linear (conflict-free)
            {
                const double *p = sm + lane + (it & 7);
                double v0 = p[0], v1 = p[64], v2 = p[128], v3 = p[192], v4 = p[256], v5 = p[320], v6 = p[384], v7 = p[440];
                asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
            }
            if (MODE == 9) // 8 ds_read_b64, uniform address
            {
                const double *p = sm + (it & 7);
                double v0 = p[0], v1 = p[64], v2 = p[128], v3 = p[192], v4 = p[256], v5 = p[320], v6 = p[384], v7 = p[440];
                asm volatile("" ::"v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
            }
            if (MODE == 10) // 8 ds_write_b128 from one lane
            {
                if (lane == lanesel)
                {
                    double2 *p = reinterpret_cast<double2 *>(sm) + (it & 7);
                    const double2 v = make_double2(x0, x1);
                    p[0] = v; p[8] = v; p[16] = v; p[24] = v; p[32] = v; p[40] = v; p[48] = v; p[56] = v;
                }
            }
            if (MODE == 11) // 8 v_cndmask_b32
            {
                asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %0, vcc\n v_cndmask_b32 %3, %0, %1, vcc\n"
                             "v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %0, vcc\n v_cndmask_b32 %3, %0, %1, vcc\n"
                             : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) :: "vcc");
            }
            if (MODE == 12) // 8 s_add (SALU)
            {
                asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 2\n s_add_i32 %0, %0, 3\n s_add_i32 %0, %0, 4\n s_add_i32 %0, %0, 5\n s_add_i32 %0, %0, 6\n s_add_i32 %0, %0, 7\n s_add_i32 %0, %0, 8\n" : "+s"(acc));
            }
            if (MODE == 13) // 8 v_max_f64 independent
            {
                x0 = __builtin_fmax(x0, y); x1 = __builtin_fmax(x1, y); x2 = __builtin_fmax(x2, y); x3 = __builtin_fmax(x3, y);
                x4 = __builtin_fmax(x4, y); x5 = __builtin_fmax(x5, y); x6 = __builtin_fmax(x6, y); x7 = __builtin_fmax(x7, y);
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        }
    }
    out[blockIdx.x * 64 + lane] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + i0 + i1 + i2 + i3 + acc;
}
template <int MODE>
void run(const char *name, double *out, int wpsimd, bool per_cu)
{
    const int blocks = 256 * 4 * wpsimd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(out, 0.5, 0.999, 5);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) k<MODE><<<blocks, 64>>>(out, 0.5, 0.999, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double instr_per_unit = (double)ITER * UNR * wpsimd * (per_cu ? 4 : 1);
    printf("%-44s W=%d  %8.3f ms  -> %7.3f ns per instruction per %s\n", name, wpsimd, ms, ms * 1e6 / instr_per_unit, per_cu ? "CU" : "SIMD");
}
int main()
{
    double *out; hipMalloc(&out, 256 * 4 * 8 * 64 * 8);
    for (int w : {1, 4})
    {
        run<0>("v_fma_f64 (8 independent)", out, w, false);
        run<4>("v_mul_f64", out, w, false);
        run<13>("v_max_f64", out, w, false);
        run<1>("v_readlane_b32", out, w, false);
        run<2>("v_mov_b32", out, w, false);
        run<3>("v_mov_b32 dpp quad_perm", out, w, false);
        run<11>("v_cndmask_b32", out, w, false);
        run<12>("s_add_i32", out, w, false);
        run<5>("ds_read_b128 uniform, 64 lanes", out, w, true);
        run<6>("ds_read_b128 uniform, 16 lanes", out, w, true);
        run<7>("ds_read_b128 uniform, 1 lane", out, w, true);
        run<8>("ds_read_b64 lane-linear", out, w, true);
        run<9>("ds_read_b64 uniform", out, w, true);
        run<10>("ds_write_b128, 1 lane", out, w, true);
    }
    return 0;
}
