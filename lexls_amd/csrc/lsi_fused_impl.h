// The resident active-set iterations of a lock-step LexLSI batch as ONE launch: every instance's wavefront runs
//     l-QR of its equality problem (lqr_wave_body, rows gathered by reference, levels above the changed one read back)
//     -> step + ratio test (lsi_iterate_step) -> removal search if nothing blocked the step (sensitivity_sweep_body)
//     -> working-set change, counters, next problem (lsi_iterate_finish)
// until the instance stops (lexlsi.h:1144-1265 per iteration) — the bodies of the three kernels the driver otherwise launches per stage
// (lexls_lsi_capi.hip: enqueue_resident), unchanged, so the trajectories are the same bits.  What the single launch removes: three kernel
// prologues / epilogues per stage (4.3 us each: the duration of a launch whose instances all skip), the lock step itself (an instance no
// longer waits for the slowest one of its stage) and the host's chunked enqueue-and-poll.  One wavefront per instance; the phases hand
// their results over through HBM exactly as the separate kernels do, with a device-scope fence between them.
#pragma once
#include "lqr_small_impl.h"
#include "lexls_sweep_impl.h"
#include "lexls_lsi_device.h"

#include <cstring>

namespace lexls
{
    namespace
    {
        /// everything the launch carries, as ONE kernel parameter: it sits at offset 0 of the kernel-argument segment, where the phases (functions of
        /// their own, see below) read it through the segment's address — scalar loads of uniform data, as in the three separate kernels, instead of
        /// a copy of two argument structs in scratch memory behind a by-reference parameter
        struct FusedArgs
        {
            LseArgs a;
            ResidentArgs ra;
            const int32_t *obj_index;
            double tolW, tolC;
            uint32_t img_doubles;
            int scan_up, count;
        };
        /// the kernel hands the phases the address of its argument segment (as an integer: function arguments travel in vector registers); a phase
        /// makes it a wave-uniform pointer into the constant address space again, so that its reads of the arguments are scalar loads
        typedef const FusedArgs __attribute__((address_space(4))) *FusedArgsPtr;
        __device__ __forceinline__ const FusedArgs &fused_args(unsigned long long kernarg)
        {
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kernarg);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(kernarg >> 32));
            return *(const FusedArgs *)(FusedArgsPtr)(((unsigned long long)hi << 32) | lo);
        }

        // the three phases as functions of their own (not inlined): each keeps the register allocation it has as a kernel — inlined into one
        // body they cost 330 spilled scalar registers (two argument structs live across everything) and ran slower than the three launches
        template <int NC, int MD, bool EXACT>
        __device__ __noinline__ void fused_phase_lqr(unsigned long long kernarg, uint32_t b)
        {
            const FusedArgs &fa = fused_args(kernarg);
            lqr_wave_body<NC, MD, EXACT, true, false>(fa.a, fa.img_doubles, 0u, b);
        }
        template <int SMD>
        __device__ __noinline__ void fused_phase_sweep(unsigned long long kernarg, uint32_t b)
        {
            const FusedArgs &fa = fused_args(kernarg);
            sensitivity_sweep_body<SMD>(fa.a, fa.obj_index, 0, fa.tolW, fa.tolC, fa.scan_up, b);
        }
        __device__ __noinline__ StepVerdict fused_phase_step(unsigned long long kernarg, uint32_t b) { return lsi_iterate_step(fused_args(kernarg).ra, b, 0u); }
        __device__ __noinline__ void fused_phase_finish(unsigned long long kernarg, uint32_t b, StepVerdict verdict)
        {
            lsi_iterate_finish<true>(fused_args(kernarg).ra, b, 0u, verdict);
        }

        /// what one phase wrote to HBM is read by the next one of the SAME wavefront: its stores must have left the wavefront and the vector
        /// cache must not serve lines it held before them
        __device__ __forceinline__ void fused_phase_fence()
        {
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef LEXLS_FUSED_AGENT_FENCE
            __threadfence();
#endif
        }

        template <int NC, int MD, bool EXACT, int SMD>
        __global__ __launch_bounds__(64, 1) void lsi_fused_kernel(FusedArgs fa)
        {
            const uint32_t b                = blockIdx.x;
            const unsigned long long kernarg = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr(); // (FusedArgs is the only parameter: offset 0)
            // Diagnostic build only (-DLEXLS_FUSED_STAMPS): shader-clock totals of the phases of this instance's iterations, left in the multiplier
            // buffer (which the product build's last removal search owns): l-QR, step, removal search, finish, iterations, searches run
#ifdef LEXLS_FUSED_STAMPS
            long long fst[4] = {0, 0, 0, 0}, fst0 = clock64(), fn_it = 0, fn_sw = 0;
#define FUSED_STAMP(i) { const long long t_ = clock64(); fst[i] += t_ - fst0; fst0 = t_; }
#else
#define FUSED_STAMP(i)
#endif
            for (int it = 0; it < fa.count; it++)
            {
                if (!fa.ra.alive[b]) break; // (wave-uniform; lsi_iterate_body clears it when the instance stops, at the latest after max_factorizations)
                fused_phase_lqr<NC, MD, EXACT>(kernarg, b);
                fused_phase_fence();
                FUSED_STAMP(0)
                const StepVerdict verdict = fused_phase_step(kernarg, b);
                fused_phase_fence();
                FUSED_STAMP(1)
                if (verdict.blk_obj < 0) // the removal search only behind a step that nothing blocked (lexlsi.h:1181-1232; the per-stage launches run it
                {                        // speculatively for every instance and ignore it for the blocked ones)
                    fused_phase_sweep<SMD>(kernarg, b);
                    fused_phase_fence();
#ifdef LEXLS_FUSED_STAMPS
                    fn_sw++;
#endif
                }
                FUSED_STAMP(2)
                fused_phase_finish(kernarg, b, verdict);
                fused_phase_fence();
                FUSED_STAMP(3)
#ifdef LEXLS_FUSED_STAMPS
                fn_it++;
#endif
            }
#ifdef LEXLS_FUSED_STAMPS
            if (threadIdx.x == 0)
            {
                double *o = fa.a.lambda + (size_t)b * (fa.a.nVar + fa.a.cap);
                for (int i = 0; i < 4; i++) o[i] = (double)fst[i];
                o[4] = (double)fn_it;
                o[5] = (double)fn_sw;
            }
#endif
#undef FUSED_STAMP
        }

        template <int NC, int MD, bool EXACT>
        hipError_t launch_lsi_fused_t(const LseArgs &a, uint32_t sweep_level_dim, const int32_t *d_obj_index, double tolW, double tolC, bool scan_up, const void *resident_args,
                                      size_t resident_args_bytes, int count, hipStream_t s)
        {
            FusedArgs fa;
            if (resident_args_bytes != sizeof(fa.ra)) return hipErrorInvalidValue;
            std::memcpy(&fa.ra, resident_args, sizeof(fa.ra));
            fa.a           = a;
            fa.obj_index   = d_obj_index;
            fa.tolW        = tolW;
            fa.tolC        = tolC;
            fa.img_doubles = wave_img_doubles<MD>(a);
            fa.scan_up     = scan_up ? 1 : 0;
            fa.count       = count;
            size_t lds      = wave_lds_bytes<NC, MD>(a, fa.img_doubles);
            const size_t l2 = sweep_lds_bytes(a);
            const size_t l3 = resident_lds_per_wave(fa.ra.sh.SD, fa.ra.sh.total);
            lds             = lds > l2 ? lds : l2;
            lds             = lds > l3 ? lds : l3;
            if (lds > 64 * 1024) return hipErrorNotSupported; // (the caller falls back to the three launches per stage)
            if (sweep_level_dim <= 12)
                hipLaunchKernelGGL((lsi_fused_kernel<NC, MD, EXACT, 12>), dim3(a.batch), dim3(64), lds, s, fa);
            else
                hipLaunchKernelGGL((lsi_fused_kernel<NC, MD, EXACT, SWEEP_MD>), dim3(a.batch), dim3(64), lds, s, fa);
            return hipGetLastError();
        }
    } // namespace
} // namespace lexls

#define LEXLS_LSI_FUSED_INSTANCE(NAME, NC, MD, EXACT)                                                                                                             \
    namespace lexls                                                                                                                                               \
    {                                                                                                                                                             \
        hipError_t NAME(const LseArgs &a, uint32_t sweep_level_dim, const int32_t *d_obj_index, double tolW, double tolC, bool scan_up, const void *resident_args,  \
                        size_t resident_args_bytes, int count, hipStream_t s)                                                                                     \
        {                                                                                                                                                         \
            return launch_lsi_fused_t<NC, MD, EXACT>(a, sweep_level_dim, d_obj_index, tolW, tolC, scan_up, resident_args, resident_args_bytes, count, s);          \
        }                                                                                                                                                         \
    }
