// TWO (or one) PROBLEMS PER WAVEFRONT, TWO (or four) WAVEFRONTS PER SIMD, Gauss step on the matrix cores: x-only, tolerance-contract
// lexicographic-QR kernel for IK-sized batches — the bench kernel since round 4 (lexlse.h:117-506 factorize() + :1015-1045 solve()).
//
// Contract (T) of include/lexls_hip.h (BASELINE north_star): column permutation, ranks and first columns EXACTLY those of the reference
// algorithm (first-maximum column pivoting on down-dated norms — compared BY VALUE, ties by position —, rank test on the fresh squared norm
// against tol_linear_dependence), x within 1e-10.
//
// Why another mapping.  lqr_qtol (four problems per wavefront, ONE wavefront per SIMD) spends half of its cycles exposed to latencies that
// nothing hides, and four in five of its vector instructions are not fma's.  4096 problems on 1024 SIMDs are four problems per SIMD however
// they are cut; here they are cut as 2 wavefronts x 2 problems (LP = 32 lanes per problem; LP = 64: 4 wavefronts x 1 problem), so that
//   * a second wavefront issues while the first waits for its pivot chain (LDS broadcast, 1/sqrt, butterfly);
//   * 32-bit vector instructions (selects, DPP moves, index work) issue every 2 cycles instead of every 4 (MI355X_MICROARCH.md, constants);
//   * the Gauss elimination of a level's rows by the finished levels (lexlse.h:431-471: two thirds of the model's flops) leaves the vector
//     unit: C[12 x 48] -= L[12 x 12] N_e[12 x 48] per finished level e is 3 k-steps x (1..3) column tiles of v_mfma_f64_16x16x4_f64 with one
//     problem spread over the whole wavefront.  No DPP row broadcasts, no multiplier chain: the finished levels are kept REDUCED
//     (N_e = R_e^-1 [T_e | rhs_e], identity in the pivot columns), so a row's multipliers are its own entries in the pivot columns.
// The wavefront alternates between two modes that exchange data through LDS:
//   "LP mode"     lane l of a problem's LP lanes, slot s  <->  the column whose position in the reference's permuted order (lexlse.h:222-232)
//                 was Fc + LP s + l when the level started (free columns are contiguous from Fc on: compaction for free).  Householder QR with
//                 column pivoting of the level, all MD rows of a column in the lane's registers.
//   "serial mode" the wavefront works on ONE problem at a time in the matrix-core layout (lane = (column c of a 16-position tile, row group g),
//                 register v = row g + 4 v): level loads, Gauss step.
//
// Level k of a problem:
//   1. its rows were requested one level ahead by LDS-DMA (global_load_lds_dwordx4: 16-byte pieces of columns straight into LDS, [column][MD]);
//   2. serial mode: C <- rows by POSITION; for every finished level e (ascending): multipliers = C's entries at e's pivot positions (through
//      an LDS scratch: C layout -> A-operand layout), C -= L_e N_e on the matrix cores; C -> LDS by position;
//   3. LP mode: 12 pivot steps.  Per step: decision (max butterfly over the down-dated norms, then a min butterfly over the positions of the
//      lanes that hold the maximum), the winner's lane stores its column to the problem's broadcast slot, every lane reads it back, fresh norm
//      / rank test / 1/sqrt in every lane, reflector in raw form H a = a + q (w.a) w, row j normalised by 1/R_jj, and a JORDAN step on the
//      rows above (a[i] -= U_i[p] U_j): after the last pivot the level block IS N_k — no triangular solve, no image of R;
//   4. N_k goes to LDS, indexed by a per-level local column index (byte k of a column's index word, as in lqr_quad).
// solve(): x_pivots(k) = rhs'_k - N_k x_later, k descending (lexlse.h:1015-1045 on the reduced rows), x kept by physical column.
//
// Shapes: every level of every problem has exactly MD rows (LseArgs::uniform_dim), no fixed variables, no regularization, n + 1 <= 48,
// cap even, 16-byte aligned input, n <= 47.  Anything else takes the other kernels.
#pragma once
#include "lqr_wave_common.h"

#include <cstdlib>

namespace lexls
{
    namespace
    {
        typedef double mf_d2 __attribute__((ext_vector_type(2)));
        typedef double mf_d4 __attribute__((ext_vector_type(4)));
        typedef unsigned mf_u2 __attribute__((ext_vector_type(2)));
        typedef unsigned mf_u4 __attribute__((ext_vector_type(4)));

        __device__ __forceinline__ void mf_lds_fence()
        {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            asm volatile("" ::: "memory");
        }

        /// 1 / x to ~2^-50 (v_rcp_f64 + one Newton step): the factor of a rank-one update, whose own rounding is of that size
        __device__ __forceinline__ double mf_rcp1(double x)
        {
            const double y = __builtin_amdgcn_rcp(x);
            return dfma(y, dfma(-x, y, 1.0), y);
        }

        /// maximum over the LP lanes of a problem, in every lane: butterfly inside the 16-lane DPP rows, then v_permlane16_swap / v_permlane32_swap
        /// (both operands the same value: whichever rows the instruction exchanges, the two results hold the two partners of every lane)
        template <int LP>
        __device__ __forceinline__ double mf_grp_max(double v)
        {
            v = dpp_max<0xB1>(v);  // quad_perm [1,0,3,2]
            v = dpp_max<0x4E>(v);  // quad_perm [2,3,0,1]
            v = dpp_max<0x141>(v); // row_half_mirror
            v = dpp_max<0x140>(v); // row_mirror
            if constexpr (LP >= 32)
            {
                const mf_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
                const mf_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
                v              = vmax(__hiloint2double((int)hi.x, (int)lo.x), __hiloint2double((int)hi.y, (int)lo.y));
            }
            if constexpr (LP >= 64)
            {
                const mf_u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
                const mf_u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
                v              = vmax(__hiloint2double((int)hi.x, (int)lo.x), __hiloint2double((int)hi.y, (int)lo.y));
            }
            return v;
        }
        template <int LP>
        __device__ __forceinline__ unsigned mf_grp_minu(unsigned v)
        {
            v = dpp_minu<0xB1>(v);
            v = dpp_minu<0x4E>(v);
            v = dpp_minu<0x141>(v);
            v = dpp_minu<0x140>(v);
            if constexpr (LP >= 32)
            {
                const mf_u2 r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
                v             = r.x < r.y ? r.x : r.y;
            }
            if constexpr (LP >= 64)
            {
                const mf_u2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
                v             = r.x < r.y ? r.x : r.y;
            }
            return v;
        }

        template <int CTRL>
        __device__ __forceinline__ int mf_dpp_maxi(int v)
        {
            const int o = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); // folds into v_max_i32_dpp
            return o > v ? o : v;
        }
        /// signed maximum over the LP lanes of a problem, in every lane
        template <int LP>
        __device__ __forceinline__ int mf_grp_maxi(int v)
        {
            v = mf_dpp_maxi<0xB1>(v);
            v = mf_dpp_maxi<0x4E>(v);
            v = mf_dpp_maxi<0x141>(v);
            v = mf_dpp_maxi<0x140>(v);
            if constexpr (LP >= 32)
            {
                const mf_u2 r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
                v             = (int)r.x > (int)r.y ? (int)r.x : (int)r.y;
            }
            if constexpr (LP >= 64)
            {
                const mf_u2 r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
                v             = (int)r.x > (int)r.y ? (int)r.x : (int)r.y;
            }
            return v;
        }

        /// f(integral_constant<int, I>) for I = B, B+1, ... while pred(I) holds
        template <int B, int E, class P, class F>
        __device__ __forceinline__ void mf_for_each_while(P &&pred, F &&f)
        {
            if constexpr (B < E)
            {
                if (pred(std::integral_constant<int, B>{}))
                {
                    f(std::integral_constant<int, B>{});
                    mf_for_each_while<B + 1, E>(pred, f);
                }
            }
        }

        constexpr double kMfSentinel = -1.0e300; // below every down-dated norm; stays there under further down-dates
        constexpr int kMfMaxObj      = 8;

        // per-level phase stamps of the diagnostic build (-DLEXLS_WAVE_STAMPS): lambda[11 + 4 k + {0 Gauss step, 1 level start, 2 Householder, 3 level end}]
#ifdef LEXLS_WAVE_STAMPS
#define MF_LSTAMP(ph)                                                                                     \
    {                                                                                                     \
        const unsigned long long t_ = clock64();                                                          \
        if (gl == 0 && live) a.lambda[(size_t)b * (n + cap) + 11 + 4 * k + (ph)] = (double)(t_ - lst_t0); \
        lst_t0 = t_;                                                                                      \
    }
#define MF_GSTAMP(i)                                     \
    {                                                    \
        const unsigned long long t_ = clock64();         \
        gst[i] += t_ - gst_t0;                           \
        gst_t0 = t_;                                     \
    }
#define MF_CSTAMP(i, val)                                                                              \
    if constexpr (NS == 1)                                                                             \
    {                                                                                                  \
        int dummy_;                                                                                    \
        asm volatile("v_mov_b32 %1, %2\n\ts_memtime %0" : "=s"(cst[i]), "=v"(dummy_) : "v"(val));   \
    }
#define MF_CSTAMP_COLLECT                                                              \
    if constexpr (NS == 1)                                                             \
    {                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             \
        cacc[0] += cst[1] - cst[0]; cacc[1] += cst[2] - cst[1]; cacc[2] += cst[3] - cst[2]; \
        cacc[3] += cst[4] - cst[3]; cacc[4] += cst[5] - cst[4]; cacc[5] += 1;          \
    }
#else
#define MF_LSTAMP(ph)
#define MF_GSTAMP(i)
#define MF_CSTAMP(i, val)
#define MF_CSTAMP_COLLECT
#endif

        /// LP lanes per problem (32: two problems per wavefront, two wavefronts per SIMD; 64: one problem, four wavefronts per SIMD);
        /// NV: the number of variables when the instantiation serves ONE n (0: taken from the arguments)
        template <int LP, int MD, int NV>
        __global__ __launch_bounds__(256, (LP == 64 ? 4 : (LP == 32 ? 2 : 1))) void lqr_mfma_kernel(LseArgs a, uint32_t nd_doubles, uint32_t group_bytes, uint32_t stagger)
        {
            static_assert((LP == 16 || LP == 32 || LP == 64) && (MD == 12 || MD == 16), "mappings / level sizes served");
            constexpr int G   = 64 / LP;            // problems per wavefront
            constexpr int NSM = (48 + LP - 1) / LP; // slots of a lane
            constexpr int NT  = 3;                  // 16-position tiles of the matrix-core layout (n + 1 <= 48)
            constexpr int HP  = MD / 2;             // 16-byte pieces per column of a level
            constexpr int CB  = 8 * MD;             // bytes per column of a level
            constexpr int NV4 = MD / 4;             // accumulator registers that hold rows of the level = k-steps of a finished level
            extern __shared__ double smem[];
            typedef __attribute__((address_space(3))) char lds_char;
            const int lds0 = (int)(unsigned)(size_t)(lds_char *)smem;
            auto D   = [&](int off) -> __attribute__((address_space(3))) double & { return *(__attribute__((address_space(3))) double *)(size_t)(unsigned)off; };
            auto D2  = [&](int off) -> __attribute__((address_space(3))) mf_d2 & { return *(__attribute__((address_space(3))) mf_d2 *)(size_t)(unsigned)off; };
            auto B8  = [&](int off) -> __attribute__((address_space(3))) uint8_t & { return *(__attribute__((address_space(3))) uint8_t *)(size_t)(unsigned)off; };
            auto U32 = [&](int off) -> __attribute__((address_space(3))) unsigned & { return *(__attribute__((address_space(3))) unsigned *)(size_t)(unsigned)off; };
            auto U64 = [&](int off) -> __attribute__((address_space(3))) unsigned long long & { return *(__attribute__((address_space(3))) unsigned long long *)(size_t)(unsigned)off; };

            const int lane    = threadIdx.x & 63;
            const int wv      = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            const uint32_t wq = blockIdx.x * 4u + (uint32_t)wv; // this wavefront's group of G problems
            if (wq * (uint32_t)G >= a.batch) return;
            const int g    = lane / LP; // problem inside the wavefront (LP mode)
            const int gl   = lane % LP;
            const int c16  = lane & 15; // serial mode: column inside a 16-position tile
            const int gg   = lane >> 4; //              row group (register v holds row gg + 4 v)
            const int n    = NV ? NV : (int)a.nVar;
            const int cap  = (int)a.cap;
            const int nObj = (int)a.nObj;
            const uint32_t b  = wq * (uint32_t)G + (uint32_t)g;
            const uint32_t bb = b < a.batch ? b : a.batch - 1u;
            const bool live   = b < a.batch && !(a.skip && a.skip[bb]);
            const uint32_t pstride = (uint32_t)cap * (uint32_t)(n + 1);
            const double *inw      = a.in + (size_t)wq * (uint32_t)G * pstride; // wave-uniform base
            const uint32_t poff    = (bb - wq * (uint32_t)G) * pstride;

            // ---- level 0: lane = column (its position layout is the identity), straight from HBM ----
            double blk[NSM][MD];
#pragma unroll
            for (int s = 0; s < NSM; s++)
            {
                const int P       = LP * s + gl;
                const int c       = P <= n ? P : n;
                const mf_d2 *src2 = reinterpret_cast<const mf_d2 *>(inw + (poff + (uint32_t)(c * cap)));
#pragma unroll
                for (int r = 0; r < MD / 2; r++)
                {
                    const mf_d2 v     = src2[r];
                    blk[s][2 * r]     = v.x;
                    blk[s][2 * r + 1] = v.y;
                }
            }

            // ---- LDS carve-up of a problem's slice (byte offsets; launch_mfma_t computes group_bytes) ----
            //   [0, o_pf)   the reduced rows N_e of the finished levels (row q of level e: S_e doubles at noff_e + q S_e), each followed by its
            //               inverse map (S_e bytes: column index in N_e -> physical column)
            //   o_pf        [column][MD]: the level loaded ahead / scratch of the Gauss step ([pivot][16 rows], then [position][MD]) / x by physical
            //               column and by column index during solve()
            //   o_bc        MD doubles + 16 bytes: broadcast slot of the pivot steps (the winner's column, its position)
            //   o_mx        2 doubles, always zero: what an operand that does not apply reads
            //   o_phys      48 B: physical column at each position (column_permutations go straight to HBM)
            //   o_meta      per level {Fc | rank << 8 | S << 16, offset of N_e (doubles)}
            //   o_emap      per physical column: byte k = its column index in N_k
            const int pfb    = CB * (n + 1) > 128 * MD ? CB * (n + 1) : 128 * MD;
            const int o_pf   = 8 * (int)nd_doubles;
            const int o_bc   = o_pf + pfb;
            const int o_mx   = o_bc + CB + 16;
            const int o_phys = o_mx + 16;
            const int o_meta = o_phys + 48;
            const int o_emap = o_meta + 8 * nObj;
            const int my = lds0 + (wv * G + g) * (int)group_bytes; // LP mode: this lane's problem

            for (int i = gl; i < 48; i += LP) B8(my + o_phys + i) = (uint8_t)i;
            for (int i = gl; i <= n; i += LP) D(my + o_emap + 8 * i) = 0.0;
            for (int i = gl; i < MD + 4; i += LP) D(my + o_bc + 8 * i) = 0.0; // (o_mx: the slice's zero words)
            mf_lds_fence();

            // ---- a level's rows by LDS-DMA: 16-byte pieces, piece t = 64 i + lane -> column t / HP, rows 2 (t % HP) .. +1; LDS image [column][MD] ----
            const int CH = (n + 1) * HP;
            uint32_t pieceoff[5];
            bool piecein[5];
#pragma unroll
            for (int i = 0; i < 5; i++)
            {
                const int t   = 64 * i + lane;
                const int tc  = t < CH ? t : CH - 1;
                const int col = tc / HP, m = tc - col * HP;
                pieceoff[i]   = (uint32_t)(col * cap + 2 * m);
                piecein[i]    = t < CH;
            }
            // The requests are inline assembly (cdna_hip_programming.md 5.7: M0 written in the statement that reads it): the compiler does not
            // count them, so no s_waitcnt of its own waits for a level that is still on its way; the kernel waits itself (vmcnt(0)) before the
            // Gauss step reads the buffer.  `tie`: a value that must be complete before the requests are issued (operand dependency)
            auto prefetch_level = [&](auto pp, int Frow, int tie) __attribute__((always_inline)) {
                constexpr int p   = decltype(pp)::value;
                const uint32_t pb = wq * (uint32_t)G + (uint32_t)p;
                const double *src = inw + (size_t)((pb < a.batch ? pb : a.batch - 1u) - wq * (uint32_t)G) * pstride + Frow; // wave-uniform
                const int dst     = lds0 + (wv * G + p) * (int)group_bytes + o_pf;
#pragma unroll
                for (int i = 0; i < 5; i++)
                    if (64 * i < CH) // wave-uniform
                    {
                        uint32_t po = 8u * pieceoff[i];
                        asm volatile("" : "+v"(po) : "v"(tie));
                        if (piecein[i])
                        {
                            unsigned keep;
                            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                                         : "=&s"(keep)
                                         : "v"(po), "s"(dst + 1024 * i), "s"(src)
                                         : "memory");
                        }
                    }
            };

            // The wavefronts that share a SIMD run the same program on the same schedule: left alone they reach every LDS round trip of the pivot
            // chain together and wait together.  Every second wavefront slot starts `stagger` x 64 cycles late
            if (stagger)
            {
                const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((4 - 1) << 11)); // HW_REG_HW_ID, bits [3:0] = wave slot of the SIMD
                if (hwid & 1u)
                    for (unsigned i = 0; i < stagger; i++) __builtin_amdgcn_s_sleep(1);
            }

            int pos[NSM]; // current position of the column held in slot s
            int pc[NSM];  // its physical column
#pragma unroll
            for (int s = 0; s < NSM; s++)
            {
                pos[s] = LP * s + gl;
                pc[s]  = LP * s + gl <= n ? LP * s + gl : n;
            }

            int ColIndex  = 0; // per problem (uniform inside its LP lanes), like everything below
            int TotalRank = 0;
            int noff      = 0; // doubles
            bool exh      = false;
            bool have_next = false; // the rows of the level about to start are in flight / in LDS (per problem)
#ifdef LEXLS_WAVE_STAMPS
            unsigned long long lst_t0 = clock64();
            const unsigned long long lst_t00 = lst_t0;
            unsigned long long gst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gst_t0 = 0;
            unsigned long long cacc[6] = {0, 0, 0, 0, 0, 0}, cst[6] = {0, 0, 0, 0, 0, 0};
#endif

            for (int k = 0; k < nObj; k++)
            {
                const bool work = live && !exh; // x only: once the columns are exhausted nothing below matters
                const int Fc    = ColIndex;
                int rank        = 0;
                if (__ballot(work) == 0ull)
                {
                    if (gl == 0) U64(my + o_meta + 8 * k) = (unsigned long long)((unsigned)Fc | ((unsigned)(n + 1 - Fc) << 16)) | ((unsigned long long)(unsigned)noff << 32);
                    continue;
                }
                const int F = k * MD;

                // =====================================================================================
                // serial mode: Gauss elimination of this level's rows by the finished levels (lexlse.h:431-471) on the matrix cores.
                // The problems of the wavefront go through every stage together (straight-line code: their LDS round trips and matrix
                // instructions overlap); what does not apply reads a zero (the slice's zero word), never a branch around a matrix instruction.
                // N_e is stored NEGATED: C += M N'_e with the multipliers M taken as they stand.
                // =====================================================================================
                if (k > 0)
                {
                    int wk[G], Fck[G], sp[G];
                    for_each_index<0, G>([&](auto pp) __attribute__((always_inline)) {
                        constexpr int p = decltype(pp)::value;
                        wk[p]           = __builtin_amdgcn_readlane((int)work, p * LP);
                        Fck[p]          = __builtin_amdgcn_readlane(ColIndex, p * LP);
                        sp[p]           = lds0 + (wv * G + p) * (int)group_bytes;
                        // a level whose predecessor could have exhausted the columns was not requested ahead
                        if (__builtin_amdgcn_readlane((int)(work && !have_next), p * LP)) prefetch_level(pp, F, 0);
                    });
#ifdef LEXLS_WAVE_STAMPS
                    gst_t0 = clock64();
#endif
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    MF_GSTAMP(0)
                    mf_d4 C[G][NT];
                    unsigned long long emw[G][NT];
                    int mxs[G][kMfMaxObj - 1], nos[G][kMfMaxObj - 1]; // the finished levels' records (scalar registers): read together, ahead of their use
                    {
                        unsigned long long mtv[G][kMfMaxObj - 1];
#pragma unroll
                        for (int p = 0; p < G; p++)
#pragma unroll
                            for (int e = 0; e < kMfMaxObj - 1; e++) mtv[p][e] = e < k ? U64(sp[p] + o_meta + 8 * e) : 0ull;
#pragma unroll
                        for (int p = 0; p < G; p++)
#pragma unroll
                            for (int e = 0; e < kMfMaxObj - 1; e++)
                            {
                                mxs[p][e] = __builtin_amdgcn_readfirstlane((int)(unsigned)mtv[p][e]);
                                nos[p][e] = __builtin_amdgcn_readfirstlane((int)(unsigned)(mtv[p][e] >> 32));
                            }
                    }
#pragma unroll
                    for (int p = 0; p < G; p++)
#pragma unroll
                        for (int t = 0; t < NT; t++)
                        {
                            const int P  = 16 * t + c16;
                            const int ph = P < n ? (int)B8(sp[p] + o_phys + P) : n;
                            emw[p][t]    = U64(sp[p] + o_emap + 8 * ph);
#pragma unroll
                            for (int v = 0; v < 4; v++)
                            {
                                const double rd = D((v < NV4 && P <= n && wk[p]) ? sp[p] + o_pf + ph * CB + 8 * (gg + 4 * v) : sp[p] + o_mx);
                                C[p][t][v]      = rd;
                            }
                        }
                    MF_GSTAMP(1)
                    for_each_index<0, kMfMaxObj - 1>([&](auto ec) __attribute__((always_inline)) {
                        constexpr int e = decltype(ec)::value;
                        if (e >= k) return; // wave-uniform
                        int Fce[G], re[G], Se[G], Noffe[G];
                        int tmin = NT;
#pragma unroll
                        for (int p = 0; p < G; p++)
                        {
                            const int mx = mxs[p][e];
                            Noffe[p]     = nos[p][e];
                            Fce[p]       = mx & 0xff;
                            re[p]        = wk[p] ? (mx >> 8) & 0xff : 0;
                            Se[p]        = (mx >> 16) & 0xff;
                            const int tm = (Fce[p] + re[p]) >> 4;
                            tmin         = (re[p] > 0 && tm < tmin) ? tm : tmin;
                        }
                        if (tmin >= NT) return; // nobody has pivots at this level (wave-uniform)
                        // B operand = N'_e, lane (column c16, k = gg): its reads do not depend on C — issued first, they run beside the previous
                        // level's matrix instructions.  k-step ks <-> pivots 4 ks .. 4 ks + 3
                        double Bop[G][NV4][NT];
#pragma unroll
                        for (int p = 0; p < G; p++)
                        {
                            const int vb = sp[p] + 8 * Noffe[p] + gg * (8 * Se[p]);
#pragma unroll
                            for (int t = 0; t < NT; t++)
                            {
                                const int P     = 16 * t + c16;
                                const int j     = (int)(((unsigned)(emw[p][t] >> (8 * (e & 3) + 32 * (e >> 2)))) & 0xffu);
                                const bool vt   = P >= Fce[p] + re[p] && P <= n;
                                const int jaddr = vb + 8 * j;
#pragma unroll
                                for (int ks = 0; ks < NV4; ks++)
                                {
                                    const bool valid = vt && 4 * ks + gg < re[p];
                                    Bop[p][ks][t]    = D(valid ? jaddr + ks * (32 * Se[p]) : sp[p] + o_mx);
                                }
                            }
                        }
                        // Problem by problem: entries at e's pivot positions -> scratch [pivot q][16 rows] -> A operands (lane (row c16, k = gg)) -> its
                        // matrix instructions.  The scratch round trip of one problem runs behind the matrix instructions of the other, here and
                        // across the levels e: the matrix pipe is the only thing that stays in line
                        auto run_tiles = [&](auto pp, auto t0c, const double (&Ao)[NV4]) __attribute__((always_inline)) {
                            constexpr int p = decltype(pp)::value, T0 = decltype(t0c)::value;
#pragma unroll
                            for (int ks = 0; ks < NV4; ks++)
#pragma unroll
                                for (int t = T0; t < NT; t++) C[p][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ao[ks], Bop[p][ks][t], C[p][t], 0, 0, 0);
                        };
                        MF_GSTAMP(5)
                        for_each_index<0, G>([&](auto pp) __attribute__((always_inline)) {
                            constexpr int p = decltype(pp)::value;
#pragma unroll
                            for (int t = 0; t < NT; t++)
                            {
                                const int q = 16 * t + c16 - Fce[p];
                                if (q >= 0 && q < re[p])
                                {
#pragma unroll
                                    for (int v = 0; v < NV4; v++) D(sp[p] + o_pf + q * 128 + 8 * (gg + 4 * v)) = C[p][t][v];
                                }
                            }
                            mf_lds_fence(); // (lanes read what OTHER lanes stored: without the fence the compiler may move a lane's load above the store)
                            double Ao[NV4];
#pragma unroll
                            for (int ks = 0; ks < NV4; ks++)
                            {
                                const int q = 4 * ks + gg;
                                Ao[ks]      = D((q < re[p] && c16 < MD) ? sp[p] + o_pf + q * 128 + 8 * c16 : sp[p] + o_mx);
                            }
                            MF_GSTAMP(6)
                            // tiles in front of every problem's pivots of this level are not touched (wave-uniform choice of the first tile)
                            if (tmin == 0)
                                run_tiles(pp, std::integral_constant<int, 0>{}, Ao);
                            else if (tmin == 1)
                                run_tiles(pp, std::integral_constant<int, 1>{}, Ao);
                            else
                                run_tiles(pp, std::integral_constant<int, 2>{}, Ao);
                            MF_GSTAMP(7)
                        });
                        MF_GSTAMP(2)
                        MF_GSTAMP(3)
                    });
                    // the eliminated rows, by position: [position][MD]
#pragma unroll
                    for (int p = 0; p < G; p++)
#pragma unroll
                        for (int t = 0; t < NT; t++)
                        {
                            const int P = 16 * t + c16;
                            if (wk[p] && P >= Fck[p] && P <= n)
                            {
#pragma unroll
                                for (int v = 0; v < NV4; v++) D(sp[p] + o_pf + P * CB + 8 * (gg + 4 * v)) = C[p][t][v];
                            }
                        }
                    MF_GSTAMP(4)
                    mf_lds_fence();
                }
                MF_LSTAMP(0)

                // =====================================================================================
                // LP mode: position layout of the level
                // =====================================================================================
                const int fcmin = [&]() {
                    int m = 0x7fff;
                    for_each_index<0, G>([&](auto pp) __attribute__((always_inline)) {
                        constexpr int p = decltype(pp)::value;
                        const int w = __builtin_amdgcn_readlane((int)work, p * LP), f = __builtin_amdgcn_readlane(ColIndex, p * LP);
                        m           = (w && f < m) ? f : m;
                    });
                    return m;
                }();
                const int ns = (n + 1 - fcmin + LP - 1) / LP; // live slots (wave-uniform)
                if (k > 0)
                {
#pragma unroll
                    for (int s = 0; s < NSM; s++)
                        if (s < ns)
                        {
                            const int P      = Fc + LP * s + gl;
                            const bool valid = work && P <= n;
                            const int Pc     = valid ? P : n;
                            pc[s]            = Pc < n ? (int)B8(my + o_phys + Pc) : n;
                            pos[s]           = valid ? P : 0x3fff;
#pragma unroll
                            for (int r = 0; r < MD; r += 2)
                            {
                                const mf_d2 v = D2(my + o_pf + Pc * CB + 8 * r);
                                blk[s][r]     = valid ? v.x : 0.0;
                                blk[s][r + 1] = valid ? v.y : 0.0;
                            }
                        }
                }
                else
                {
#pragma unroll
                    for (int s = 0; s < NSM; s++)
                    {
                        const bool valid = work && LP * s + gl <= n;
                        pos[s]           = valid ? LP * s + gl : 0x3fff;
#pragma unroll
                        for (int r = 0; r < MD; r++) blk[s][r] = valid ? blk[s][r] : 0.0;
                    }
                }
                // down-dated norms start as the squared column norms of the eliminated rows (lexlse.h:193-196)
                double nrm[NSM];
#pragma unroll
                for (int s = 0; s < NSM; s++)
                {
                    double t0 = 0.0, t1 = 0.0;
#pragma unroll
                    for (int r = 0; r < MD; r += 2)
                    {
                        t0 = dfma(blk[s][r], blk[s][r], t0);
                        t1 = dfma(blk[s][r + 1], blk[s][r + 1], t1);
                    }
                    nrm[s] = (s < ns && pos[s] < n) ? t0 + t1 : kMfSentinel;
                }
                // the next level's rows: requested now (the block is in registers, the scratch is free again), unless this level can exhaust the
                // columns
                {
                    const bool pf = work && (k + 1 < nObj) && (Fc + MD < n);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the DMA writes what the reads above read)
                    int tie = __double2hiint(nrm[0]);
#pragma unroll
                    for (int s = 1; s < NSM; s++) tie ^= __double2hiint(nrm[s]);
                    for_each_index<0, G>([&](auto pp) __attribute__((always_inline)) {
                        constexpr int p = decltype(pp)::value;
                        if (__builtin_amdgcn_readlane((int)pf, p * LP)) prefetch_level(pp, F + MD, tie);
                    });
                    have_next = pf;
                }
                MF_LSTAMP(1)

                // =====================================================================================
                // Householder QR with column pivoting of the level (lexlse.h:182-268) + Jordan step: the block ends as N_k.
                // What is uniform per problem lives in scalar registers as lane masks / readlane results: the decision costs the vector unit one
                // maximum, one LDS atomic and one compare; everything a stopped level must not do is skipped under its lanes' exec bits.
                // =====================================================================================
                unsigned permw[(MD + 3) / 4]; // column_permutations of this level's pivots, four per word (uniform per problem)
#pragma unroll
                for (int i = 0; i < (MD + 3) / 4; i++) permw[i] = 0u;
                // (a problem without work has no candidates: one lane with a zero norm keeps its "maximum" unique, so that it never asks for
                // the tie path below)
                if (!work) nrm[0] = gl == 0 ? 0.0 : kMfSentinel;
                auto factor_level = [&](auto nsc) __attribute__((always_inline)) {
                    constexpr int NS = decltype(nsc)::value;
                    bool go = work;
                    mf_for_each_while<0, MD>(
                        [&](auto jc) __attribute__((always_inline)) { return (decltype(jc)::value % 4 != 0) || __ballot(go) != 0ull; },
                        [&](auto jc) __attribute__((always_inline)) {
                            constexpr int j = decltype(jc)::value;
                            const bool act  = go;
                            // ---- decision: first maximum of the down-dated norms (lexlse.h:205-206).  Fast path on the norms' HIGH WORDS (sign, exponent,
                            //      20 mantissa bits; signed integer order = the order of non-negative doubles): one v_max_i32_dpp per butterfly
                            //      stage.  A unique lane with the largest high word holds the largest norm.  Anything else — two lanes with that
                            //      high word, or no non-negative norm at all — takes the exact path below: maximum by VALUE over the whole
                            //      double, the smallest position among equals.  (Every problem of the wavefront has exactly one such lane in the
                            //      usual case, whether it still works or not: the test is one population count.) ----
                            int hl = __double2hiint(nrm[0]);
#pragma unroll
                            for (int s = 1; s < NS; s++) hl = __double2hiint(nrm[s]) > hl ? __double2hiint(nrm[s]) : hl;
                            MF_CSTAMP(0, hl)
                            const int mh = mf_grp_maxi<LP>(hl);
                            MF_CSTAMP(1, mh)
                            bool iswin[NS];
                            unsigned long long mk = 0ull, two = 0ull;
#pragma unroll
                            for (int s = 0; s < NS; s++)
                            {
                                iswin[s]                    = __double2hiint(nrm[s]) == mh;
                                const unsigned long long ms = __builtin_amdgcn_uicmp((unsigned)__double2hiint(nrm[s]), (unsigned)mh, 32 /* == */);
                                two |= mk & ms;
                                mk |= ms;
                            }
                            const unsigned long long negm = __builtin_amdgcn_sicmp(mh, 0, 40 /* signed < */);
                            if ((two | negm) != 0ull || __builtin_popcountll(mk) != G) // (wave-uniform, rare)
                            {
                                double mv = kMfSentinel;
#pragma unroll
                                for (int s = 0; s < NS; s++) mv = vmax(mv, nrm[s]);
                                mv          = mf_grp_max<LP>(mv);
                                unsigned pk = 0xffffu;
#pragma unroll
                                for (int s = 0; s < NS; s++)
                                {
                                    iswin[s]         = nrm[s] == mv && nrm[s] > 0.5 * kMfSentinel;
                                    const unsigned c = iswin[s] ? (unsigned)pos[s] : 0xffffu;
                                    pk               = c < pk ? c : pk;
                                }
                                const unsigned wp = mf_grp_minu<LP>(pk);
#pragma unroll
                                for (int s = 0; s < NS; s++) iswin[s] = iswin[s] && (unsigned)pos[s] == wp;
                            }
                            // ---- the winner's column and position to every lane of the problem (a problem that has stopped may write: nobody uses it) ----
#pragma unroll
                            for (int s = 0; s < NS; s++)
                                if (iswin[s])
                                {
#pragma unroll
                                    for (int r = 0; r < MD; r += 2) D2(my + o_bc + 8 * r) = mf_d2{blk[s][r], blk[s][r + 1]};
                                    U32(my + o_bc + CB) = (unsigned)pos[s];
                                }
                            mf_lds_fence(); // (lanes read what another lane stored: without the fence the compiler may move a lane's load above the store)
                            double w[MD];
#ifdef LEXLS_MFMA_DPP_BCAST
                            // one 8-byte read per lane (lane l of every 16-lane row reads entry l), then row broadcasts: an eighth of the LDS traffic of
                            // the all-lanes-read-everything form, twelve more vector instructions
                            {
                                const double wl_ = D(my + o_bc + 8 * ((lane & 15) < MD ? (lane & 15) : 0));
                                for_each_index<0, MD>([&](auto rr) __attribute__((always_inline)) { w[decltype(rr)::value] = gbc<decltype(rr)::value>(wl_); });
                            }
#else
#pragma unroll
                            for (int r = 0; r < MD; r += 2)
                            {
                                const mf_d2 v = D2(my + o_bc + 8 * r);
                                w[r]          = v.x;
                                w[r + 1]      = v.y;
                            }
#endif
                            const int ppos = (int)U32(my + o_bc + CB);
                            MF_CSTAMP(2, ppos)
                            mf_lds_fence();
                            const double c0 = w[j];
                            // The chain to the next decision first: the raw dot products w . a_s (they need nothing but the column) beside the tail norm,
                            // then fresh norm -> 1/sqrt -> R_js -> down-dated norms; what only the rows' update needs (1 / (c0 - beta), the rank-one
                            // update, the Jordan step) comes behind a scheduling barrier, in the shadow of the next step's butterfly and LDS round trip
                            double dwv[NS];
                            double t0 = 0.0, t1 = 0.0, t2 = 0.0; // tail norm in three partial sums, fresh norm = c0^2 + tail (lexlse.h:210-211, :241)
#pragma unroll
                            for (int r = j + 1; r < MD; r++)
                            {
                                if ((r - j) % 3 == 1) t0 = dfma(w[r], w[r], t0);
                                if ((r - j) % 3 == 2) t1 = dfma(w[r], w[r], t1);
                                if ((r - j) % 3 == 0) t2 = dfma(w[r], w[r], t2);
                            }
#pragma unroll
                            for (int s = 0; s < NS; s++)
                            {
                                double d0 = 0.0, d1 = 0.0;
#pragma unroll
                                for (int r = j + 1; r < MD; r++)
                                {
                                    if ((r - j) & 1)
                                        d0 = dfma(w[r], blk[s][r], d0);
                                    else
                                        d1 = dfma(w[r], blk[s][r], d1);
                                }
                                dwv[s] = dfma(c0, blk[s][j], d0 + d1);
                            }
                            const double fresh = dfma(c0, c0, (t0 + t1) + t2);
                            MF_CSTAMP(3, __double2hiint(fresh))
                            const bool cont    = act && !(fresh < a.tol); // rank test on the squared norm (lexlse.h:214)
                            if (cont) // a level that has stopped leaves its block, norms and maps alone (its lanes sit out)
                            {
                                // 1 / sqrt(fresh): v_rsq_f64 and one coupled iteration (g -> sqrt, h -> 1 / (2 sqrt)).  ~2^-45: the reflector need not be
                                // orthogonal to rounding — N_k = (G A_P)^-1 G A_rest for ANY row transformation G applied to all columns alike; what
                                // matters is that the pivot column's image is (beta, 0, ..) to a relative 1e-13, and the norms' down-date to 1e-13
                                double gq, hq;
                                {
                                    const double y = __builtin_amdgcn_rsq(fresh);
                                    gq             = fresh * y;
                                    hq             = 0.5 * y;
                                    const double r = dfma(-hq, gq, 0.5);
                                    gq             = dfma(gq, r, gq);
                                    hq             = dfma(hq, r, hq);
                                }
                                const bool neg    = c0 >= 0.0; // beta = -sign(c0) sqrt(fresh)
                                const double ibet = (neg ? -2.0 : 2.0) * hq; // 1 / beta
                                double tv[NS];
#pragma unroll
                                for (int s = 0; s < NS; s++)
                                {
                                    tv[s]  = dwv[s] * ibet;               // R_js = (w . a_s) / beta
                                    nrm[s] = dfma(-tv[s], tv[s], nrm[s]); // lexlse.h:262-266
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                const double beta = neg ? -gq : gq;
                                const double rden = mf_rcp1(c0 - beta);
                                // every column: rows below a_s += gs w, row j normalised, Jordan step on the rows above
#pragma unroll
                                for (int s = 0; s < NS; s++)
                                {
                                    const double t  = tv[s];
                                    const double gs = (t - blk[s][j]) * rden; // a_s[r] += gs w[r] (= a_s - tau v v.a_s, lexlse.h:243-246)
                                    const double u  = t * ibet;               // R_js / R_jj
#pragma unroll
                                    for (int r = j + 1; r < MD; r++) blk[s][r] = dfma(gs, w[r], blk[s][r]);
                                    blk[s][j] = u;
#pragma unroll
                                    for (int i = 0; i < j; i++) blk[s][i] = dfma(-w[i], u, blk[s][i]);
                                }
                                // column "swap": the position map (lexlse.h:222-232); the pivot column leaves the candidates
#pragma unroll
                                for (int s = 0; s < NS; s++)
                                {
                                    pos[s] = pos[s] == ColIndex ? ppos : pos[s];
                                    pos[s] = iswin[s] ? ColIndex : pos[s];
                                    nrm[s] = iswin[s] ? kMfSentinel : nrm[s];
                                }
                                permw[j / 4] |= (unsigned)ppos << (8 * (j % 4));
                                ColIndex += 1;
                                rank += 1;
                            }
                            MF_CSTAMP(4, __double2hiint(nrm[0]))
                            MF_CSTAMP(5, __double2hiint(blk[0][MD - 1]))
                            MF_CSTAMP_COLLECT
                            const bool full = cont && ColIndex == n;
                            exh             = exh || full;
                            go              = cont && !full;
                        });
                };
                if (NSM > 2 && ns > 2)
                    factor_level(std::integral_constant<int, (NSM > 2 ? 3 : 1)>{});
                else if (NSM > 1 && ns > 1)
                    factor_level(std::integral_constant<int, (NSM > 1 ? 2 : 1)>{});
                else
                    factor_level(std::integral_constant<int, 1>{});
                MF_LSTAMP(2)

                // =====================================================================================
                // level end: N'_k = minus the first `rank` rows of the columns behind the pivots, by local column index; maps; column_permutations
                // =====================================================================================
                {
                    const int S = n + 1 - Fc - rank;
                    bool same = true; // every working problem of the wavefront has the same rank (the usual case): the row loop is scalar
                    int rk0   = 0;
#pragma unroll
                    for (int p = 0; p < G; p++)
                    {
                        const int w_ = __builtin_amdgcn_readlane((int)work, p * LP), r_ = __builtin_amdgcn_readlane(rank, p * LP);
                        if (w_)
                        {
                            same = same && (rk0 == 0 || r_ == rk0 || r_ == 0);
                            rk0  = r_ > rk0 ? r_ : rk0;
                        }
                    }
#pragma unroll
                    for (int s = 0; s < NSM; s++)
                        if (s < ns)
                        {
                            const int P0     = Fc + LP * s + gl;
                            const bool mv    = work && P0 <= n;
                            const int j      = pos[s] - (Fc + rank);
                            const bool free_ = mv && j >= 0 && rank > 0;
                            if (free_)
                            {
                                const int na = my + 8 * (noff + j);
                                if (same)
                                {
#pragma unroll
                                    for (int q = 0; q < MD; q++)
                                        if (q < rk0) D(na + 8 * q * S) = -blk[s][q]; // (wave-uniform bound)
                                }
                                else
                                {
#pragma unroll
                                    for (int q = 0; q < MD; q++)
                                        if (q < rank) D(na + 8 * q * S) = -blk[s][q];
                                }
                                B8(my + 8 * (noff + rank * S) + j) = (uint8_t)pc[s];
                            }
                            if (mv && j >= 0) B8(my + o_emap + 8 * pc[s] + k) = (uint8_t)j;
                            if (mv && P0 < n) B8(my + o_phys + pos[s]) = (uint8_t)pc[s];
                        }
                    if (gl < rank)
                    {
                        unsigned wsel = permw[0];
#pragma unroll
                        for (int i = 1; i < (MD + 3) / 4; i++) wsel = (gl >> 2) == i ? permw[i] : wsel;
                        a.perm[(size_t)b * n + Fc + gl] = (wsel >> (8 * (gl & 3))) & 0xffu;
                    }
                    if (gl == 0) U64(my + o_meta + 8 * k) = (unsigned long long)((unsigned)Fc | ((unsigned)rank << 8) | ((unsigned)S << 16)) | ((unsigned long long)(unsigned)noff << 32);
                    noff += (work && rank > 0) ? rank * S + ((S + 7) >> 3) : 0;
                    TotalRank += rank;
                }
                mf_lds_fence();
                MF_LSTAMP(3)
            }

            // ---- solve(): x_pivots(k) = rhs'_k - N_k x_later, k descending (lexlse.h:1015-1045); lane q <-> row q of a level; x by physical column
            //      in the (idle) level buffer.  A DMA request that nobody consumed (the columns ran out under it) must have landed first ----
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LEXLS_WAVE_STAMPS
            {
                const unsigned long long t_ = clock64();
                if (gl == 0 && live) a.lambda[(size_t)b * (n + cap) + 8] = (double)(t_ - lst_t00);
                lst_t0 = t_;
            }
#endif
            const int o_x  = o_pf;       // x by physical column
            const int o_xl = o_pf + 512; // x by column index in N_k (the operand of a level's product)
            for (int i = gl; i < 64 + 56; i += LP) D(my + o_x + 8 * i) = 0.0; // (x by physical column and, from o_xl on, by column index: finite everywhere)
            mf_lds_fence();
            for (int k = nObj; k--;)
            {
                const unsigned long long mt = U64(my + o_meta + 8 * k);
                const int Fck = (int)(mt & 0xffu), rk = live ? (int)((mt >> 8) & 0xffu) : 0, Sk = (int)((mt >> 16) & 0xffu);
                const int nofk = (int)(unsigned)(mt >> 32), invk = 8 * (nofk + rk * Sk);
                int rmax = 0, smax = 0;
                for_each_index<0, G>([&](auto pp) __attribute__((always_inline)) {
                    constexpr int p = decltype(pp)::value;
                    const int r_ = __builtin_amdgcn_readlane(rk, p * LP), s_ = __builtin_amdgcn_readlane(rk > 0 ? Sk : 0, p * LP);
                    rmax = r_ > rmax ? r_ : rmax;
                    smax = s_ > smax ? s_ : smax;
                });
                if (rmax == 0) continue;
                // x of the columns behind this level's pivots, by their column index in N_k (zero beyond a problem's own count)
                for (int j = gl; j < smax - 1; j += LP)
                {
                    const bool in = rk > 0 && j < Sk - 1;
                    const int ph  = in ? (int)B8(my + invk + j) : 0;
                    const double xv = D(my + o_x + 8 * ph);
                    D(my + o_xl + 8 * j) = in ? xv : 0.0;
                }
                mf_lds_fence();
                const bool row = gl < rk;
                const int rowa = my + 8 * (nofk + (row ? gl : 0) * Sk);
                double s0 = row ? -D(rowa + 8 * (Sk - 1)) : 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0; // the right-hand side is the last column of N_k (rows stored negated)
                for (int j = 0; j < smax - 1; j += 4)
                {
                    // (reads beyond a row's own columns stay inside the slice and meet zeros of x)
                    const double n0 = D(rowa + 8 * j), n1 = D(rowa + 8 * j + 8), n2 = D(rowa + 8 * j + 16), n3 = D(rowa + 8 * j + 24);
                    const mf_d2 xa = D2(my + o_xl + 8 * j), xb = D2(my + o_xl + 8 * j + 16);
                    s0 = dfma((row && j < Sk - 1 ? n0 : 0.0), xa.x, s0);
                    s1 = dfma((row && j + 1 < Sk - 1 ? n1 : 0.0), xa.y, s1);
                    s2 = dfma((row && j + 2 < Sk - 1 ? n2 : 0.0), xb.x, s2);
                    s3 = dfma((row && j + 3 < Sk - 1 ? n3 : 0.0), xb.y, s3);
                }
                if (row) D(my + o_x + 8 * (int)B8(my + o_phys + Fck + gl)) = (s0 + s1) + (s2 + s3);
                mf_lds_fence();
            }
#ifdef LEXLS_WAVE_STAMPS
            {
                const unsigned long long t_ = clock64();
                if (gl == 0 && live) a.lambda[(size_t)b * (n + cap) + 9] = (double)(t_ - lst_t0);
                lst_t0 = t_;
                if (gl == 0 && live)
                    for (int i_ = 0; i_ < 8; i_++) a.lambda[(size_t)b * (n + cap) + 32 + i_] = (double)gst[i_];
                if (gl == 0 && live)
                    for (int i_ = 0; i_ < 6; i_++) a.lambda[(size_t)b * (n + cap) + 44 + i_] = (double)cacc[i_];
            }
#endif
            // ---- results ----
            if (live)
            {
                for (int P = gl; P < n; P += LP)
                {
                    a.x[(size_t)b * n + P] = D(my + o_x + 8 * P); // x of the variable (physical column) P: x = P x already applied (lexlse.h:1044)
                    if (P >= TotalRank) a.perm[(size_t)b * n + P] = (uint32_t)P; // (the pivots' entries were stored level by level)
                }
                if (gl < nObj)
                {
                    const unsigned long long mt   = U64(my + o_meta + 8 * gl);
                    a.fcol[(size_t)b * nObj + gl] = (uint32_t)(mt & 0xffu);
                    a.rank[(size_t)b * nObj + gl] = (uint32_t)((mt >> 8) & 0xffu);
                }
                if (gl == 0) a.totalrank[b] = (uint32_t)TotalRank;
            }
        }

        /// exact worst case of the reduced rows and their inverse maps: max over rank distributions (rank_k <= md, sum <= n) of
        /// sum_k rank_k S_k + ceil(S_k / 8), S_k = n + 1 - Fc_k - rank_k
        inline uint32_t mfma_nd_doubles(uint32_t n, uint32_t nObj, uint32_t md)
        {
            // best[fc]: the most doubles the levels from the current one on can need when the current one starts at column fc
            uint32_t best[65], next[65];
            for (uint32_t fc = 0; fc <= n; fc++) next[fc] = 0;
            for (uint32_t k = nObj; k--;)
            {
                for (uint32_t fc = 0; fc <= n; fc++)
                {
                    uint32_t m = 0;
                    for (uint32_t r = 0; r <= md && fc + r <= n; r++)
                    {
                        const uint32_t S = n + 1 - fc - r;
                        const uint32_t v = r * S + (r ? (S + 7) / 8 : 0) + next[fc + r]; // the rows and their inverse map
                        m                = v > m ? v : m;
                    }
                    best[fc] = m;
                }
                for (uint32_t fc = 0; fc <= n; fc++) next[fc] = best[fc];
            }
            return (next[0] + 1) & ~1u;
        }

        template <int MD>
        inline size_t mfma_group_bytes(uint32_t n, uint32_t nObj)
        {
            const size_t pfb = 8u * MD * (size_t)(n + 1) > 128u * MD ? 8u * MD * (size_t)(n + 1) : 128u * MD;
            const size_t raw = 8 * (size_t)mfma_nd_doubles(n, nObj, MD) + pfb + 8 * MD + 16 + 16 + 48 + 8 * (size_t)nObj + 8 * (size_t)(n + 1);
            // (no padding against bank conflicts between the problems of a wavefront: the LDS serves a wave's 8- and 16-byte accesses in lane groups
            // that never mix the two halves of the wavefront, MI355X_MICROARCH.md LDS table)
            return (raw + 15) & ~(size_t)15;
        }

        template <int LP, int MD, int NV>
        hipError_t launch_mfma_t(const LseArgs &a, hipStream_t s)
        {
            constexpr uint32_t G = 64 / LP;
            const uint32_t nd    = mfma_nd_doubles(a.nVar, a.nObj, MD);
            const size_t gbytes  = mfma_group_bytes<MD>(a.nVar, a.nObj);
            const size_t lds     = 4 * G * gbytes;
            // (two / four wavefronts per SIMD are the point of the mapping: the workgroups of a CU must fit its LDS together)
            if (lds * (LP == 64 ? 4 : (LP == 32 ? 2 : 1)) > kMaxLdsBytes || a.nObj > (uint32_t)kMfMaxObj || a.nVar + 1 > 48u || a.nVar < 1u || (NV && a.nVar != (uint32_t)NV)) return hipErrorInvalidValue;
            if (a.uniform_dim != (uint32_t)MD || (a.cap & 1u) || (reinterpret_cast<uintptr_t>(a.in) & 15u) || a.nfixed || a.reg_type != 0) return hipErrorInvalidValue;
            if (lds > 64 * 1024)
            {
                static size_t granted[64] = {0}; // per device: the attribute is set once, not per launch
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
                if (dev < 0 || granted[dev] < lds)
                {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lqr_mfma_kernel<LP, MD, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (e != hipSuccess) return e;
                    if (dev >= 0) granted[dev] = lds;
                }
            }
            const uint32_t blocks = (a.batch + 4u * G - 1u) / (4u * G);
            const char *st_env     = std::getenv("LEXLS_MFMA_STAGGER"); // x 64 cycles; diagnostic knob
            const uint32_t stagger = st_env ? (uint32_t)std::atoi(st_env) : (LP == 32 ? 8u : 0u);
            hipLaunchKernelGGL((lqr_mfma_kernel<LP, MD, NV>), dim3(blocks), dim3(256), lds, s, a, nd, (uint32_t)gbytes, stagger);
            return hipGetLastError();
        }
    } // namespace
} // namespace lexls

#define LEXLS_MFMA_INSTANCE(NAME, LP, MD, NV) \
    namespace lexls { hipError_t NAME(const LseArgs &a, hipStream_t s) { return launch_mfma_t<LP, MD, NV>(a, s); } \
                      size_t NAME##_lds(uint32_t nVar, uint32_t nObj) { return 4 * (64 / LP) * mfma_group_bytes<MD>(nVar, nObj); } }
