// FOUR PROBLEMS PER WAVEFRONT, x-only, TOLERANCE-CONTRACT lexicographic-QR kernel for IK-sized batches — the bench kernel since round 3
// (lexlse.h:117-506 factorize() + :1015-1045 solve()).
//
// Contract (BASELINE north_star): column permutation, ranks and first columns EXACTLY those of the reference algorithm (first-maximum
// column pivoting on down-dated norms, rank test on the fresh squared norm against tol_linear_dependence), x within 1e-10.  Unlike
// lqr_quad_impl.h (same wavefront mapping, bit-identical to oracle/lexlse_oracle.h, still behind lexls_lse_set_kernel_policy(h, 4)) the
// VALUES are not held to the oracle's summation order, which buys:
//   * the reflector in RAW form.  With w = [c0 - beta; column tail] (= (c0 - beta) [1; essential part], lexlse.h:239-248) and
//     q = 1 / ((c0 - beta) beta):   H a = a + q (w.a) w.   The dot products w.a run on the raw pivot column BESIDE the sqrt / reciprocal
//     chain instead of behind it, and no essential part / tau is ever formed or broadcast (22 row broadcasts per pivot gone);
//   * NORMALISED images: a finished pivot row is kept as [R T | rhs] / R_jj.  The Gauss elimination of a later level's rows
//     (lexlse.h:431-471) then needs no multiplier product (a[r][P] -= a[r][c'] U'[c'][P]) and the back-substitution (lexlse.h:1015-1045) no
//     division.  The sign of beta cancels in U', so identity reflectors (tail == 0, last row of a level) need no special case;
//   * the search for pivot j+1 (butterflies on the down-dated norms, lexlse.h:205-206, :262-266) is issued right after row j of the block is
//     final, in front of the rank-one update of the rows below — latency chain and fma stream overlap;
//   * coalesced level loads: a level's rows arrive as 48-byte pieces of columns in consecutive lanes (a third of the line look-ups of the
//     column-per-lane load), are requested ONE LEVEL AHEAD (spread over the pivot steps of the level in front) and turned into the
//     position layout through a 2-KB LDS staging block per problem, half the rows at a time;
//   * triangular images (the part of [R_q T_q] below the diagonal is never read): 6,880 instead of 8,512 bytes per IK problem — what
//     makes room for the staging block at one wavefront per SIMD (4 x 10 KB per wavefront, 160 KB per CU).
// Mapping, position layout, index words: as in lqr_quad_impl.h (one problem per 16-lane DPP row, slot s of lane l = position 16 s + l - SIG
// when the level started, finished pivots in static lanes, row broadcasts with v_mov_b64_dpp row_newbcast).
//
// Shapes: every level of every problem has exactly MD rows (checked by the host: LseArgs::uniform_dim), no fixed variables, no
// regularization, n + 1 + SIG <= 16 NS, cap even, 16-byte aligned input.  Anything else takes the bit-exact kernels.
#pragma once
#include "lqr_quad_impl.h"

#include <cstdlib>

#ifndef QT_WPB
#define QT_WPB 4 // wavefronts per workgroup of lqr_qtol (1 or 4)
#endif

namespace lexls
{
    namespace
    {
        /// 1 / x to full double precision for normal x (v_rcp_f64 + two Newton steps); no scaling: |x| in [1e-290, 1e290]
        __device__ __forceinline__ double qt_rcp(double x)
        {
            double y = __builtin_amdgcn_rcp(x);
            double e = dfma(-x, y, 1.0);
            y        = dfma(y, e, y);
            e        = dfma(-x, y, 1.0);
            return dfma(y, e, y);
        }

        /// 1 / x to ~2^-50 (v_rcp_f64 + one Newton step): the factor of a rank-one update, whose own rounding is of that size
        __device__ __forceinline__ double qt_rcp1(double x)
        {
            const double y = __builtin_amdgcn_rcp(x);
            return dfma(y, dfma(-x, y, 1.0), y);
        }

        /// sqrt(x) for normal x (v_rsq_f64, one coupled iteration, two correction steps — the unscaled core of the library routine)
        __device__ __forceinline__ double qt_sqrt(double x)
        {
            const double y = __builtin_amdgcn_rsq(x);
            double g       = x * y;
            double h       = 0.5 * y;
            const double r = dfma(-h, g, 0.5);
            g              = dfma(g, r, g);
            h              = dfma(h, r, h);
            double d       = dfma(-g, g, x);
            g              = dfma(d, h, g);
            d              = dfma(-g, g, x);
            return dfma(d, h, g);
        }

        /// a double whose high word is `hi` and whose low word is that of v (sentinel norms: any low word gives a huge negative number)
        __device__ __forceinline__ double qt_with_hi(double v, int hi) { return __hiloint2double(hi, __double2loint(v)); }

#ifdef LEXLS_WAVE_STAMPS_FINE // = the S0 whose pivot steps are stamped (0: level 0 of the IK shape, 2: its last level)
#define FSTAMP(i) if constexpr (S0 == LEXLS_WAVE_STAMPS_FINE) STAMP(i)
#else
#define FSTAMP(i)
#endif
// Chain stamps (-DLEXLS_QTOL_CHAIN=<S0>): s_memtime behind an instruction that depends on the named value — the time at which that value is
// READY, without draining anything (the plain stamps wait for lgkmcnt(0) and disturb the chain they measure).  Seven points per pivot step.
#ifdef LEXLS_QTOL_CHAIN
#define CSTAMP(i, val)                                                                                     \
    if constexpr (S0 == LEXLS_QTOL_CHAIN)                                                                  \
    {                                                                                                      \
        int dummy_;                                                                                        \
        asm volatile("v_mov_b32 %1, %2\n\ts_memtime %0" : "=s"(ct[i]), "=v"(dummy_) : "v"(val));         \
    }
#define CSTAMP_COLLECT                                                                                     \
    if constexpr (S0 == LEXLS_QTOL_CHAIN)                                                                  \
    {                                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        if (cvalid)                                                                                        \
        {                                                                                                  \
            cacc[2] += ct[2] - cp1; cacc[3] += ct[3] - ct[2]; cacc[4] += ct[4] - ct[3];                     \
            cacc[5] += ct[5] - ct[4]; cacc[6] += ct[6] - ct[5]; cacc[0] += ct[0] - ct[6];                   \
        }                                                                                                  \
        cacc[1] += ct[1] - ct[0];                                                                          \
        cp1    = ct[1];                                                                                    \
        cvalid = true;                                                                                     \
    }
#else
#define CSTAMP(i, val)
#define CSTAMP_COLLECT
#endif
// per-level phase stamps of the diagnostic build: lambda[11 + 4 k + {0 load, 1 eliminate, 2 Householder, 3 level end}]
#ifdef LEXLS_WAVE_STAMPS
#define LSTAMP(ph)                                                                                   \
    {                                                                                                \
        const unsigned long long t_ = clock64();                                                     \
        if (gl == 0 && live) a.lambda[(size_t)b * (n + cap) + 11 + 4 * k + (ph)] = (double)(t_ - lst_t0); \
        lst_t0 = t_;                                                                                 \
    }
#else
#define LSTAMP(ph)
#endif
        /// f(integral_constant<int, I>) for I = B, B+1, ... while pred(I) holds: the first failing test leaves the whole remainder behind one branch
        template <int B, int E, class P, class F>
        __device__ __forceinline__ void qt_for_each_while(P &&pred, F &&f)
        {
            if constexpr (B < E)
            {
                if (pred(std::integral_constant<int, B>{}))
                {
                    f(std::integral_constant<int, B>{});
                    qt_for_each_while<B + 1, E>(pred, f);
                }
            }
        }

        typedef double qt_d2 __attribute__((ext_vector_type(2))); // 16 bytes as a native vector (the level pieces stay in registers)


        // ---- the level-ahead pieces live in FIXED accumulation registers a[184:255], managed by inline assembly -----------------------
        // A piece register that the compiler allocates must have ONE load site (two sites meet in register copies that wait for the loads in
        // flight: s_waitcnt vmcnt(0) in the middle of the pivot loop), but the requests have to be spread over the whole level in front —
        // Householder phase included, which exists in one instantiation per number of live slots — because the memory system takes what
        // every wave of the chip asks for at once at HBM speed only (scripts/ubench/loadpat.hip: ~8-10 k cycles per level and wave when
        // nothing else is done meanwhile).  So the compiler never sees these registers: piece t is a[184 + 4 t : 187 + 4 t], loaded and
        // stored by the statements below; the build checks that no other instruction of the code object names a184 .. a255
        // (scripts/check_qtol_regs.py, run by the Makefile).
#define QT_PF_LIST(X) X(0, 184, 187) X(1, 188, 191) X(2, 192, 195) X(3, 196, 199) X(4, 200, 203) X(5, 204, 207) X(6, 208, 211) X(7, 212, 215) X(8, 216, 219) \
    X(9, 220, 223) X(10, 224, 227) X(11, 228, 231) X(12, 232, 235) X(13, 236, 239) X(14, 240, 243) X(15, 244, 247) X(16, 248, 251) X(17, 252, 255)
        /// piece T <- 16 bytes at base (wave-uniform, in scalar registers) + byte offset (per lane): no vector address arithmetic per request
        template <int T>
        __device__ __forceinline__ void qt_pf_load(const double *base, uint32_t byte_offset)
        {
#define QT_PF_LOAD(t, lo, hi) \
    if constexpr (T == t) asm volatile("global_load_dwordx4 a[" #lo ":" #hi "], %0, %1" ::"v"(byte_offset), "s"(base) : "memory", "a" #lo, "a" #hi);
            QT_PF_LIST(QT_PF_LOAD)
#undef QT_PF_LOAD
        }
        template <int T>
        __device__ __forceinline__ void qt_pf_store_lds(int lds_byte_address)
        {
#define QT_PF_STORE(t, lo, hi) \
    if constexpr (T == t) asm volatile("ds_write_b128 %0, a[" #lo ":" #hi "]" ::"v"(lds_byte_address) : "memory");
            QT_PF_LIST(QT_PF_STORE)
#undef QT_PF_STORE
        }

        constexpr int kQtSentinelHi = (int)0xFFE00000; // -2^1023 * 1.x: below every down-dated norm, finite whatever the low word

        /// NV: the number of variables when the instantiation serves ONE n (0: taken from the arguments) — the piece counts of the level loads and
        /// the layout tests then fold at compile time
        template <int NS, int MD, int SIG, int NV>
        __global__ __launch_bounds__(64 * QT_WPB) void lqr_qtol_kernel(LseArgs a, uint32_t img_doubles, uint32_t group_bytes, uint32_t stagger)
        {
            static_assert(NS >= 1 && NS <= 4 && MD <= 16 && (MD % 4) == 0, "shape limits of the row layout / two row parts of even size");
            constexpr int NH  = 2;        // row parts of the staging transposition
            constexpr int RP  = MD / NH;  // rows per part
            constexpr int HP  = RP / 2;   // 16-byte pieces per column and part
            constexpr int kHandoffStride = 8 * MD + 16; // bytes between the lanes' hand-off slots: 16-byte aligned, b128 stores of 8 lanes on 32 banks
            constexpr int NIH = NS * HP;  // load instructions per part (16 NS columns x HP pieces / 16 lanes)
#ifndef LEXLS_QTOL_PF_STEPS
#define LEXLS_QTOL_PF_STEPS 9
#endif
            constexpr int PF_STEPS = LEXLS_QTOL_PF_STEPS < MD ? LEXLS_QTOL_PF_STEPS : MD; // pivot steps over which a level's requests are spread
            extern __shared__ double smem[];
            // LDS is addressed by 32-bit byte addresses turned into address-space-3 pointers directly (base of the dynamic block folded into the
            // slice offset once): an access through a generic pointer costs an extra add of the block's base (zero) per access
            typedef __attribute__((address_space(3))) char lds_char;
            const int lds0 = (int)(unsigned)(size_t)(lds_char *)smem;
            const int lane = threadIdx.x & 63;
            // QT_WPB wavefronts per workgroup (independent of each other; one per SIMD): a quarter of the workgroups to dispatch
            const uint32_t wq = QT_WPB > 1 ? blockIdx.x * QT_WPB + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : blockIdx.x; // this wavefront's quad of problems (in scalar registers)
            if (wq * 4u >= a.batch) return; // (a wavefront beyond the batch: nothing of it is waited for)
            const int g    = lane >> 4; // row = problem inside the wave
            const int gl   = lane & 15;
            const int n    = NV ? NV : (int)a.nVar;
            const int cap  = (int)a.cap;
            const int nObj = (int)a.nObj;
            const uint32_t b  = wq * 4u + (uint32_t)g;
            const uint32_t bb = b < a.batch ? b : a.batch - 1u; // rows beyond the batch idle on a valid address
            const bool live   = b < a.batch && !(a.skip && a.skip[bb]);
            const uint32_t pstride = (uint32_t)cap * (uint32_t)(n + 1);
            const double *inw      = a.in + (size_t)wq * 4u * pstride; // wave-uniform base; lane offsets stay 32-bit
            const uint32_t poff    = (bb - wq * 4u) * pstride;

            // ---- the first level's rows are requested before anything else (every wave of the chip asks for its first level at once: the HBM serves
            //      this burst at its full rate, and nothing can be computed before it lands).  Its position layout is the identity, so lane = column
            //      loads the block directly, no staging; for such a burst this pattern is also the fastest of those measured
            //      (scripts/ubench/loadpat.hip: 7.7-8.4 k cycles per level against 9.5 k for the 48-byte pieces) ----
            double blk[NS][MD]; // the level block, position layout
#pragma unroll
            for (int s = 0; s < NS; s++)
            {
                const int P       = 16 * s + gl - SIG;
                const int c       = (P >= 0 && P <= n) ? P : 0;
                const qt_d2 *src2 = reinterpret_cast<const qt_d2 *>(inw + (poff + (uint32_t)(c * cap)));
#pragma unroll
                for (int r = 0; r < MD / 2; r++)
                {
                    const qt_d2 v     = src2[r];
                    blk[s][2 * r]     = v.x;
                    blk[s][2 * r + 1] = v.y;
                }
            }

            // ---- LDS carve-up of this row's slice (byte offsets; launch_qtol_t computes group_bytes) ----
            const int o_img   = lds0 + (int)((QT_WPB > 1 ? (threadIdx.x >> 6) * 4u : 0u) + (uint32_t)g) * (int)group_bytes;
            const int o_xs    = o_img + 8 * (int)img_doubles; // 16*NS : x by position (zero until the back-substitution: also the "U" of a position that is no pivot yet)
            const int o_ex    = o_xs + 8 * 16 * NS;           // MD    : dump slots (one dword per lane) of byte stores that do not apply
            const int o_phys  = o_ex + 8 * MD;                // 64 B  : physical column at each position
            const int o_perm  = o_phys + 64;                  // 64 B  : column_permutations
            const int o_meta  = o_perm + 64;                  // kQuadMaxObj x {first column, rank, image offset, image width}
            const int o_emap  = o_meta + 16 * kQuadMaxObj;    // 16 NS x 8 B: byte k of entry j = column index of PHYSICAL column j in the image of level k
            const int o_stage = o_emap + 8 * 16 * NS;         // max((n + 1) RP, 16 MD) doubles: staging block of the level loads; 16 hand-off slots of the pivot steps
            auto D   = [&](int off) -> __attribute__((address_space(3))) double & { return *(__attribute__((address_space(3))) double *)(size_t)(unsigned)off; };
            auto D2  = [&](int off) -> __attribute__((address_space(3))) qt_d2 & { return *(__attribute__((address_space(3))) qt_d2 *)(size_t)(unsigned)off; };
            auto B8  = [&](int off) -> __attribute__((address_space(3))) uint8_t & { return *(__attribute__((address_space(3))) uint8_t *)(size_t)(unsigned)off; };
            auto U32 = [&](int off) -> __attribute__((address_space(3))) uint32_t & { return *(__attribute__((address_space(3))) uint32_t *)(size_t)(unsigned)off; };
            auto U64 = [&](int off) -> __attribute__((address_space(3))) unsigned long long & { return *(__attribute__((address_space(3))) unsigned long long *)(size_t)(unsigned)off; };
            typedef unsigned qt_u4 __attribute__((ext_vector_type(4)));
            auto U4 = [&](int off) -> __attribute__((address_space(3))) qt_u4 & { return *(__attribute__((address_space(3))) qt_u4 *)(size_t)(unsigned)off; };

#pragma unroll
            for (int s = 0; s < 4; s++) B8(o_phys + 16 * s + gl) = (uint8_t)(16 * s + gl);
#pragma unroll
            for (int s = 0; s < NS; s++)
            {
                D(o_emap + 8 * (16 * s + gl)) = 0.0;
                D(o_xs + 8 * (16 * s + gl))   = 0.0;
            }
            quad_lds_fence();

            // The four waves of a CU (one per SIMD) start a little apart: every wave of the chip is in the same phase of the same level otherwise,
            // and each level's rows are asked for by all 1024 waves at once — a burst the HBM serves at its full rate while every SIMD waits
            if (stagger)
            {
                const unsigned simd = __builtin_amdgcn_s_getreg((4 << 0) | (4 << 6) | ((2 - 1) << 11)); // HW_REG_HW_ID, bits [5:4] = SIMD
                for (unsigned i = 0; i < simd * stagger; i++) __builtin_amdgcn_s_sleep(8);
            }

            // ---- level loads: pieces of 16 bytes, HP consecutive pieces = RP rows of one column, columns in consecutive lanes ----
            const int CH = (n + 1) * HP; // pieces per problem, level and row part
            static_assert(NH * NIH <= 18, "eighteen piece registers");
            // piece t = (row part t / NIH, instruction t % NIH) of the level whose first row is Frow -> its fixed registers
            // byte offsets of this lane's pieces inside a level, computed once (the division by HP is not repeated per request)
            uint32_t pieceoff[NH * NIH];
            for_each_index<0, NH * NIH>([&](auto tt) __attribute__((always_inline)) {
                constexpr int t = decltype(tt)::value, h = t / NIH, i = t % NIH;
                int ch        = 16 * i + gl;
                ch            = ch < CH ? ch : CH - 1; // lanes past the end repeat the last piece (same bytes to the same LDS address)
                const int col = ch / HP, m = ch - col * HP;
                pieceoff[t]   = 8u * (poff + (uint32_t)(col * cap + h * RP + 2 * m)); // bytes
            });
            auto prefetch_piece = [&](auto tt, int Frow) __attribute__((always_inline)) {
                constexpr int t = decltype(tt)::value, i = t % NIH;
                if (16 * i < CH) // wave-uniform
                    qt_pf_load<t>(inw + Frow, pieceoff[t]);
            };

            int rp[NS];         // slot s, lane l: LDS byte address of the (triangular) image row of pivot position c = 16 s + l - SIG
            int rq[NS];         // slot s, lane l: v_perm selector that picks the byte of pivot position c's LEVEL out of a column's index word
            unsigned long long em[NS]; // the index word of the column held in slot s
            int pos[NS];        // current position of the column held in slot s
            int pc[NS];         // its physical column
#pragma unroll
            for (int s = 0; s < NS; s++)
            {
                rp[s]  = o_xs; // (a position that is not a pivot yet "reads" zeros of the x block: see the elimination)
                rq[s]  = 0x0c0c0c00;
                em[s]  = 0ull;
                pos[s] = 0;
                pc[s]  = 0;
            }

            int ColIndex   = 0; // per row (uniform inside a row), like everything below
            int TotalRank  = 0;
            int imgoff     = 0; // doubles
            bool exh       = false;
            bool have_next = false; // the pieces of the level about to start are already in flight / in registers (wave-uniform)
            STAMP_DECL
            STAMP(0)
#ifdef LEXLS_QTOL_CHAIN
            unsigned long long cacc[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
#ifdef LEXLS_WAVE_STAMPS
            unsigned long long lst_t0 = clock64();
#endif

            for (int k = 0; k < nObj; k++)
            {
                const bool work = live && !exh; // x only: once the columns are exhausted nothing below matters
                const int Fc    = ColIndex;
                int rank        = 0;
                if (__ballot(work) == 0ull)
                {
                    if (gl == 0)
                    {
                        U32(o_meta + 16 * k)      = (uint32_t)Fc;
                        U32(o_meta + 16 * k + 4)  = 0u;
                        U32(o_meta + 16 * k + 8)  = (uint32_t)imgoff;
                        U32(o_meta + 16 * k + 12) = (uint32_t)(n + 1 - Fc);
                    }
                    continue;
                }
                const int F = k * MD;
                if (k > 0 && !have_next) // a level whose predecessor could have exhausted the columns: all pieces at once
                    for_each_index<0, NH * NIH>([&](auto tt) __attribute__((always_inline)) { prefetch_piece(tt, F); });

                // =====================================================================================
                // position layout of the level; staged pieces -> block
                // =====================================================================================
#pragma unroll
                for (int s = 0; s < NS; s++)
                {
                    const int P  = 16 * s + gl - SIG;
                    const int ph = (int)B8(o_phys + (P >= 0 && P < n ? P : 0)); // (unconditional reads: the three slots' look-ups go out together)
                    pc[s]        = (P >= 0 && P < n) ? ph : (P == n ? n : 0);
                    pos[s]       = (P >= 0 && P <= n) ? P : 0x3fffff;
                }
#pragma unroll
                for (int s = 0; s < NS; s++) em[s] = U64(o_emap + 8 * pc[s]);
                if (k > 0)
                for_each_index<0, NH>([&](auto hh) __attribute__((always_inline)) {
                    constexpr int h = decltype(hh)::value;
                    // the pieces come out of their fixed registers (requested during the level in front, or just now)
                    if constexpr (h == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    for_each_index<0, NIH>([&](auto ii) __attribute__((always_inline)) {
                        constexpr int i = decltype(ii)::value;
                        if (16 * i < CH)
                        {
                            int ch = 16 * i + gl;
                            ch     = ch < CH ? ch : CH - 1;
                            qt_pf_store_lds<h * NIH + i>(o_stage + 16 * ch);
                        }
                    });
                    quad_lds_fence();
#pragma unroll
                    for (int s = 0; s < NS; s++)
                    {
#pragma unroll
                        for (int m = 0; m < HP; m++)
                        {
                            const qt_d2 v            = D2(o_stage + 16 * (pc[s] * HP + m));
                            blk[s][h * RP + 2 * m]     = v.x;
                            blk[s][h * RP + 2 * m + 1] = v.y;
                        }
                    }
                    quad_lds_fence();
                });
                STAMP(1)
                LSTAMP(0)
                // the next level's rows: requested while this level is factorised (two pieces per pivot step, below), unless this level can
                // exhaust the columns
                const bool prefetch = (k + 1 < nObj) && (rows_min(work ? Fc : 0x3fffffff) + MD < n);
                have_next           = prefetch;

                // =====================================================================================
                // Gauss elimination of these rows by every finished pivot c' (lexlse.h:431-471, left-looking, normalised pivot rows)
                // =====================================================================================
                const int Fcmax = rows_max(work ? Fc : 0);
                {
                    // U'[c'][P] of this lane's columns is fetched one pivot ahead of its use (the read depends on a row-broadcast address)
                    auto fetch_u = [&](auto cc, double (&u)[NS]) __attribute__((always_inline)) {
                        constexpr int C  = decltype(cc)::value;
                        constexpr int sc = (C + SIG) / 16, lc = (C + SIG) % 16;
                        const int rowp   = gbci<lc>(rp[sc]);
                        const int selq   = gbci<lc>(rq[sc]);
#pragma unroll
                        for (int s = sc; s < NS; s++)
                        {
                            const unsigned e = __builtin_amdgcn_perm((unsigned)(em[s] >> 32), (unsigned)em[s], (unsigned)selq);
                            u[s]             = D(rowp + (int)(e << 3));
                        }
                    };
                    double ua[NS], ub[NS]; // even / odd steps (no copies between the steps)
#pragma unroll
                    for (int s = 0; s < NS; s++) ua[s] = ub[s] = 0.0;
                    if (Fcmax > 0) fetch_u(std::integral_constant<int, 0>{}, ua);
                    qt_for_each_while<0, 16 * NS - SIG>(
                        [&](auto cc) __attribute__((always_inline)) { return decltype(cc)::value < Fcmax; }, // wave-uniform
                        [&](auto cc) __attribute__((always_inline)) {
                            constexpr int C  = decltype(cc)::value;
                            constexpr int sc = (C + SIG) / 16, lc = (C + SIG) % 16;
                            double(&ucur)[NS]  = (C & 1) ? ub : ua;
                            double(&unext)[NS] = (C & 1) ? ua : ub;
                            // a row of the wavefront whose own pivots end before Fcmax meets the zeros of the x block as "U'": every fma adds a zero product
                            double lr[MD];
                            for_each_index<0, MD>([&](auto rr) {
                                constexpr int r = decltype(rr)::value;
                                lr[r]           = gbc<lc>(blk[sc][r]);
                            });
#pragma unroll
                            for (int r = 0; r < MD; r++) blk[sc][r] = dfma(-lr[r], ucur[sc], blk[sc][r]);
                            // (the next step's U' behind the first slot's work: its address chain does not stall the step's start)
                            if constexpr (C + 1 < 16 * NS - SIG)
                            {
                                if (C + 1 < Fcmax) fetch_u(std::integral_constant<int, C + 1>{}, unext);
                            }
#pragma unroll
                            for (int s = sc + 1; s < NS; s++)
                            {
#pragma unroll
                                for (int r = 0; r < MD; r++) blk[s][r] = dfma(-lr[r], ucur[s], blk[s][r]);
                            }
                        });
                }
                STAMP(7)
                LSTAMP(1)

                // =====================================================================================
                // Householder QR with column pivoting of the level (lexlse.h:182-268), slots S0 .. NS-1.
                // Straight-line per pivot, ONE basic block: a row that has stopped keeps executing on data nobody reads again, its bookkeeping
                // frozen by selects.  Order inside a step = the dependency chain, with everything that is not on it placed in its shadows:
                //   read the pivot column (LDS, slot of the winning lane)  ->  tail / fresh norm  ->  1/sqrt  ->  row j of the block,
                //   down-dated norms  ->  DECISION for pivot j+1 (local best, two butterflies)  ->  next read.
                // Beside the chain: the raw dot products (beside the 1/sqrt), the reciprocal of c0 - beta and the rank-one update of the rows
                // below (beside the butterflies), then every lane stores the column of its local best slot in its own LDS slot — the store
                // does not wait for the decision, only the next step's read address does.
                // =====================================================================================
                auto factor_level = [&](auto s0c) __attribute__((always_inline)) {
                    constexpr int S0 = decltype(s0c)::value;
                    constexpr int SL = NS - S0; // live slots
                    double nrm[NS];
#pragma unroll
                    for (int s = S0; s < NS; s++)
                    {
                        double t = 0.0;
#pragma unroll
                        for (int r = 0; r < MD; r++) t = dfma(blk[s][r], blk[s][r], t); // lexlse.h:193-196
                        nrm[s] = sel(pos[s] >= ColIndex && pos[s] < n, t, qt_with_hi(t, kQtSentinelHi));
                    }
                    bool go = work;
#ifdef LEXLS_QTOL_CHAIN
                    unsigned long long ct[7] = {0, 0, 0, 0, 0, 0, 0}, cp1 = 0;
                    bool cvalid = false;
#endif
                    // Pivot decision: first maximum (by position) of the down-dated norms (lexlse.h:205-206) as ONE f64 max butterfly: the low twelve
                    // bits of a candidate's norm are replaced by 4095 - (position << 6 | slot << 4 | lane) for the comparison, so that equal
                    // norms order by position and the winner's identity comes out of the maximum itself.  (Norms that agree in their upper
                    // 52 - 12 mantissa bits also order by position: a 2^-40 window in which the reference's own choice depends on its summation
                    // order.  The norms themselves stay untouched.)
                    int cur_lbs = S0, nxt_lbs = S0;     // slot of the lane's local best candidate: for this pivot / the next one
                    bool cur_ispl = false, nxt_ispl = false; // this lane holds the pivot column
                    unsigned cur_w = 0, nxt_w = 0;      // the winner's 12-bit key (position << 6 | slot << 4 | lane)
                    double pbest = 0.0;                 // the lane's local best, packed
                    auto decide_local = [&]() __attribute__((always_inline)) {
#pragma unroll
                        for (int s = S0; s < NS; s++)
                        {
                            const int kinv  = (0xFFF - ((s << 4) | gl)) - (pos[s] << 6);
                            const double pv = __hiloint2double(__double2hiint(nrm[s]), (__double2loint(nrm[s]) & ~0xFFF) | kinv);
                            pbest           = s == S0 ? pv : vmax(pbest, pv);
                        }
                        nxt_lbs = ((0xFFF - (__double2loint(pbest) & 0xFFF)) >> 4) & 3;
                    };
                    auto decide_finish = [&](double m) __attribute__((always_inline)) {
                        const int mlo = __double2loint(m);
                        nxt_w         = (unsigned)(0xFFF - (mlo & 0xFFF));
                        nxt_ispl      = ((__double2loint(pbest) ^ mlo) & 0xFFF) == 0;
                    };
                    // prologue: decision for pivot 0, every lane's best column to its hand-off slot
                    {
                        decide_local();
                        decide_finish(row_max16(pbest));
                        double colv[MD];
#pragma unroll
                        for (int r = 0; r < MD; r++) colv[r] = blk[S0][r];
#pragma unroll
                        for (int s = S0 + 1; s < NS; s++)
                        {
                            const bool pick = nxt_lbs == s;
#pragma unroll
                            for (int r = 0; r < MD; r++) colv[r] = sel(pick, blk[s][r], colv[r]);
                        }
#pragma unroll
                        for (int r = 0; r < MD; r += 2) D2(o_stage + gl * kHandoffStride + 8 * r) = qt_d2{colv[r], colv[r + 1]};
                        cur_lbs = nxt_lbs, cur_ispl = nxt_ispl, cur_w = nxt_w;
                    }
                    // the winner's column is read AHEAD: as soon as a step knows the next pivot's lane, the read is issued — the rest of the step (rank-one
                    // update of the other columns) runs while it is on its way
                    double coln[MD];
                    auto fetch_column = [&](auto jjc, unsigned w) __attribute__((always_inline)) {
                        constexpr int ce0 = decltype(jjc)::value & ~1;
                        quad_lds_fence();
                        const int src = o_stage + (int)(w & 15u) * kHandoffStride;
#pragma unroll
                        for (int r = ce0; r < MD; r += 2)
                        {
                            const qt_d2 v   = D2(src + 8 * r);
                            coln[r]         = v.x;
                            coln[r + 1]     = v.y;
                        }
                    };
                    fetch_column(std::integral_constant<int, 0>{}, cur_w);

                    int pf_issued = 0; // pieces of the next level requested so far (wave-uniform)
                    qt_for_each_while<0, MD>(
                        [&](auto jc) __attribute__((always_inline)) { return (decltype(jc)::value % 4 != 0) || __ballot(go) != 0ull; }, // tested every fourth step: no row of the wavefront has work left -> ONE branch leaves the level
                        [&](auto cnt) __attribute__((always_inline)) {
                        constexpr int j   = decltype(cnt)::value;
                        constexpr int ce  = j & ~1;       // first (even) row of this step's hand-off
                        constexpr int cen = (j + 1) & ~1; // ... of the next step's
                        // the next level's pieces: PF_PER per pivot step from the first step on (what an early end leaves over is requested behind the loop)
                        {
                            constexpr int TOT = NH * NIH, PF_PER = (TOT + PF_STEPS - 1) / PF_STEPS;
                            constexpr int lo = (j * PF_PER < TOT ? j * PF_PER : TOT), hi = ((j + 1) * PF_PER < TOT ? (j + 1) * PF_PER : TOT);
                            if (prefetch) for_each_index<lo, hi>([&](auto tt) __attribute__((always_inline)) { prefetch_piece(tt, F + MD); });
                            pf_issued = hi;
                        }
                        const bool act = go;
                        CSTAMP(0, (int)cur_w)
                        double col[MD];
#pragma unroll
                        for (int r = ce; r < MD; r++) col[r] = coln[r];
                        FSTAMP(2)
                        CSTAMP(1, __double2loint(col[MD - 1]))
                        CSTAMP_COLLECT
                        const double c0 = col[j];
                        // tail norm in three partial sums, fresh norm = c0^2 + tail (lexlse.h:210-211, :241)
                        double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
                        for (int r = j + 1; r < MD; r++)
                        {
                            if ((r - j) % 3 == 1) t0 = dfma(col[r], col[r], t0);
                            if ((r - j) % 3 == 2) t1 = dfma(col[r], col[r], t1);
                            if ((r - j) % 3 == 0) t2 = dfma(col[r], col[r], t2);
                        }
                        const double tailSq = (t0 + t1) + t2;
                        const double fresh  = dfma(c0, c0, tailSq);
                        CSTAMP(2, __double2loint(fresh))
                        const bool cont     = act && !(fresh < a.tol); // rank test on the squared norm (lexlse.h:214); no branch: a stopped row runs on
                        go                  = cont;
                        // 1 / sqrt(fresh): v_rsq_f64 and two coupled iterations (g -> sqrt, h -> 1 / (2 sqrt))
                        double g, h;
                        {
                            const double y = __builtin_amdgcn_rsq(fresh);
                            g              = fresh * y;
                            h              = 0.5 * y;
                            double r       = dfma(-h, g, 0.5);
                            g              = dfma(g, r, g);
                            h              = dfma(h, r, h);
#ifndef LEXLS_QTOL_ONE_NEWTON
                            r              = dfma(-h, g, 0.5);
                            g              = dfma(g, r, g);
                            h              = dfma(h, r, h);
#endif
                        }
                        const bool neg    = c0 >= 0.0;       // beta = -sign(c0) sqrt(fresh)
                        const double beta = neg ? -g : g;
                        const double ibet = (neg ? -2.0 : 2.0) * h; // 1 / beta
                        CSTAMP(3, __double2loint(ibet))
                        const double rden = qt_rcp1(c0 - beta);     // (for the rows below; not on the chain to the next decision)
                        // raw dot products col . a of every live column (beside the chain above)
                        double dw[NS];
#pragma unroll
                        for (int s = S0; s < NS; s++)
                        {
                            double d0 = 0.0, d1 = 0.0;
#pragma unroll
                            for (int r = j + 1; r < MD; r++)
                            {
                                if ((r - j) & 1)
                                    d0 = dfma(col[r], blk[s][r], d0);
                                else
                                    d1 = dfma(col[r], blk[s][r], d1);
                            }
                            dw[s] = dfma(c0, blk[s][j], d0 + d1);
                        }
                        FSTAMP(3)
                        // row j of the block: R_js = (col . a_s) / beta (final after this reflector); norm down-date (lexlse.h:262-266); the pivot
                        // column leaves the candidates; the row is kept normalised by 1 / R_jj.  gs: a_s[r] += gs col[r] for the rows below,
                        // gs = (R_js - a_s[j]) / (c0 - beta)   (= a_s - tau v v.a_s, lexlse.h:243-246)
                        double gs[NS];
#pragma unroll
                        for (int s = S0; s < NS; s++)
                        {
                            const double t = dw[s] * ibet;
                            gs[s]          = (t - blk[s][j]) * rden;
                            nrm[s]         = dfma(-t, t, nrm[s]);
                            nrm[s]         = sel(cont && cur_ispl && cur_lbs == s, qt_with_hi(nrm[s], kQtSentinelHi), nrm[s]);
                            blk[s][j]      = t * ibet;
                        }
                        CSTAMP(4, __double2hiint(nrm[NS - 1]))
                        // column "swap": update the position map (lexlse.h:222-232)
                        const int ppos = (int)(cur_w >> 6);
#pragma unroll
                        for (int s = S0; s < NS; s++)
                        {
                            const bool front = cont && pos[s] == ColIndex;
                            pos[s]           = sel(front, ppos, pos[s]);
                            pos[s]           = sel(cont && cur_ispl && cur_lbs == s, ColIndex, pos[s]);
                        }
                        B8(sel(cont && cur_ispl, o_perm + ColIndex, o_ex + 4 * gl)) = (uint8_t)ppos; // (o_ex: dump slots, one bank per lane — same-address stores of many lanes serialise)
                        ColIndex += cont ? 1 : 0;
                        rank += cont ? 1 : 0;
                        const bool full = cont && ColIndex == n;
                        exh             = exh || full;
                        go              = go && !full;
                        FSTAMP(4)
                        if constexpr (j + 1 < MD && SL == 1)
                        {
                            // one live slot: the lane's best column IS the slot — update it (in the stalls of the butterfly), store it
                            decide_local();
                            double m = pbest;
                            auto upd_rows = [&](auto qq) __attribute__((always_inline)) {
                                constexpr int q = decltype(qq)::value;
#pragma unroll
                                for (int r = j + 1; r < MD; r++)
                                    if ((r - j - 1) % 4 == q) blk[S0][r] = dfma(gs[S0], col[r], blk[S0][r]);
                            };
                            m = dpp_max<0xB1>(m);
                            upd_rows(std::integral_constant<int, 0>{});
                            m = dpp_max<0x4E>(m);
                            upd_rows(std::integral_constant<int, 1>{});
                            m = dpp_max<0x141>(m);
                            upd_rows(std::integral_constant<int, 2>{});
                            m = dpp_max<0x140>(m);
                            upd_rows(std::integral_constant<int, 3>{});
#pragma unroll
                            for (int r = cen; r < MD; r += 2) D2(o_stage + gl * kHandoffStride + 8 * r) = qt_d2{blk[S0][r], blk[S0][r + 1]};
                            decide_finish(m);
                            fetch_column(std::integral_constant<int, j + 1>{}, nxt_w);
                        }
                        else if constexpr (j + 1 < MD)
                        {
                            // The decision for the next pivot: local part, then the four butterfly stages.  In the stalls of the stages: the lane's
                            // best column (for the next hand-off) is picked out of the slots BEFORE the rank-one update, updated on its own and
                            // stored — the store does not wait for the decision, only the next step's read address does
                            decide_local();
                            double colv[MD];
                            double gsb = gs[S0];
                            double m   = pbest;
                            auto pick_rows = [&](auto qq) __attribute__((always_inline)) {
                                constexpr int q = decltype(qq)::value;
#pragma unroll
                                for (int r = cen; r < MD; r++)
                                    if ((r - cen) % 4 == q)
                                    {
                                        colv[r] = blk[S0][r];
#pragma unroll
                                        for (int s = S0 + 1; s < NS; s++) colv[r] = sel(nxt_lbs == s, blk[s][r], colv[r]);
                                    }
                            };
                            m = dpp_max<0xB1>(m);
                            pick_rows(std::integral_constant<int, 0>{});
                            m = dpp_max<0x4E>(m);
                            pick_rows(std::integral_constant<int, 1>{});
                            m = dpp_max<0x141>(m);
                            pick_rows(std::integral_constant<int, 2>{});
                            m = dpp_max<0x140>(m);
                            pick_rows(std::integral_constant<int, 3>{});
#pragma unroll
                            for (int s = S0 + 1; s < NS; s++) gsb = sel(nxt_lbs == s, gs[s], gsb);
#pragma unroll
                            for (int r = (cen > j + 1 ? cen : j + 1); r < MD; r++) colv[r] = dfma(gsb, col[r], colv[r]);
#pragma unroll
                            for (int r = cen; r < MD; r += 2) D2(o_stage + gl * kHandoffStride + 8 * r) = qt_d2{colv[r], colv[r + 1]};
                            decide_finish(m);
                            fetch_column(std::integral_constant<int, j + 1>{}, nxt_w);
                        }
                        CSTAMP(5, (int)nxt_w)
                        // rows below of every live column
                        if constexpr (!(j + 1 < MD && SL == 1))
                        {
#pragma unroll
                            for (int s = S0; s < NS; s++)
                            {
#pragma unroll
                                for (int r = j + 1; r < MD; r++) blk[s][r] = dfma(gs[s], col[r], blk[s][r]);
                            }
                        }
                        cur_lbs = nxt_lbs, cur_ispl = nxt_ispl, cur_w = nxt_w;
                        CSTAMP(6, __double2loint(blk[NS - 1][MD - 1]))
                        FSTAMP(5)
                    });
                    if (prefetch)
                        for_each_index<0, NH * NIH>([&](auto tt) __attribute__((always_inline)) {
                            if (decltype(tt)::value >= pf_issued) prefetch_piece(tt, F + MD);
                        });
                    (void)SL;
                };
                {
                    const int s0 = (rows_min(work ? Fc : 0x3fffffff) + SIG) >> 4;
                    if (NS > 3 && s0 >= 3)
                        factor_level(std::integral_constant<int, (NS > 3 ? 3 : 0)>{});
                    else if (NS > 2 && s0 >= 2)
                        factor_level(std::integral_constant<int, (NS > 2 ? 2 : 0)>{});
                    else if (NS > 1 && s0 >= 1)
                        factor_level(std::integral_constant<int, (NS > 1 ? 1 : 0)>{});
                    else
                        factor_level(std::integral_constant<int, 0>{});
                }
                STAMP(5)
                LSTAMP(2)

                // =====================================================================================
                // level end: triangular image [R_k T_k | rhs_k] / diag in end-of-level position order, maps
                // =====================================================================================
                const int wk   = n + 1 - Fc;
                const int dump = o_stage + 8 * gl; // stores that do not apply go to a dump slot of the lane's own (the staging block is idle here): no divergent regions, no bank conflicts
                int roff[MD];          // byte offset of image row p (triangular packing), the same for every slot
#pragma unroll
                for (int p = 0; p < MD; p++) roff[p] = 8 * (p * wk - p * (p + 1) / 2);
#pragma unroll
                for (int s = 0; s < NS; s++)
                {
                    const int P0  = 16 * s + gl - SIG;
                    const bool mv = work && P0 <= n && P0 >= Fc; // columns that were live in this level (the RHS included)
                    const int e   = pos[s] - Fc;                 // index of this column in the level's image: end-of-level position order
                    const int lim = mv ? (e < rank - 1 ? e : rank - 1) : -1; // rows 0 .. lim of the column go to the image (p < rank, p <= e)
                    const int base = o_img + 8 * (imgoff + e);
                    if (mv)
                    {
#pragma unroll
                        for (int p = 0; p < MD; p++) D(sel(p <= lim, base + roff[p], dump)) = blk[s][p];
                        B8(o_emap + 8 * pc[s] + k) = (uint8_t)e;
                        if (P0 < n) B8(o_phys + pos[s]) = (uint8_t)pc[s];
                    }
                    // pivot position P0 of this level: its image row and the selector of this level's index byte
                    const bool piv = work && P0 >= Fc && P0 < Fc + rank;
                    const int p    = P0 - Fc;
                    rp[s]          = sel(piv, o_img + 8 * (imgoff + p * wk - p * (p + 1) / 2), rp[s]);
                    rq[s]          = sel(piv, 0x0c0c0c00 | k, rq[s]);
                }
                STAMP(6)
                LSTAMP(3)
                if (gl == 0)
                {
                    U32(o_meta + 16 * k)      = (uint32_t)Fc;
                    U32(o_meta + 16 * k + 4)  = (uint32_t)rank;
                    U32(o_meta + 16 * k + 8)  = (uint32_t)imgoff;
                    U32(o_meta + 16 * k + 12) = (uint32_t)wk;
                }
                quad_lds_fence();
                imgoff += wk * rank - rank * (rank - 1) / 2;
                TotalRank += rank;
            }

            // ---- solve(): block back-substitution on the normalised images (lexlse.h:1015-1045); lane p <-> row p of a level.  Straight-line
            // per level: what does not apply reads a zero (position 16 NS - 1 of the x block is never written) or goes to the dump slot ----
            const int o_zero = o_xs + 8 * (16 * NS - 1);
            for (int k = nObj; k--;)
            {
                const qt_u4 mt = U4(o_meta + 16 * k); // {first column, rank, image offset, image width}
                const int rank = live ? (int)mt.y : 0;
                const int rmax = rows_max(rank);
                if (rmax == 0) continue;
                const int Fc = (int)mt.x, ok = (int)mt.z, wk = (int)mt.w;
                const int c0   = Fc + rank;
                const int acc  = rank > 0 ? TotalRank - c0 : 0;
                const int amax = rows_max(acc);
                const int p    = gl < rank ? gl : 0;
                const int row  = o_img + 8 * (ok + p * wk - p * (p + 1) / 2);
                double col[MD];
#pragma unroll
                for (int j = 1; j < MD; j++) col[j] = D(sel(j < rank && gl < j, row + 8 * j, o_zero));
                double sv = D(sel(gl < rank, row + 8 * (n - Fc), o_zero));
                // rhs'_k - T'_k x_later (lexlse.h:1029-1033): the column at final position c sits at its level-k index inside the image;
                // sixteen solved positions per trip — lane j looks up index and x of position c0 + base + j, the row-broadcast hands them out
                for (int base = 0; base < amax; base += 16)
                {
                    const bool have = base + gl < acc;
                    const int c     = have ? c0 + base + gl : 16 * NS - 1;
                    const int ph    = (int)B8(o_phys + c);
                    const int offv  = have ? (int)B8(o_emap + 8 * ph + k) : 0;
                    const double xv = D(o_xs + 8 * c); // (zero where the position does not apply)
                    double uj[16]; // all sixteen reads in flight before the first is used (issued a few at a time, each group pays the LDS latency)
                    for_each_index<0, 16>([&](auto jj) {
                        constexpr int j = decltype(jj)::value;
                        uj[j]           = D(row + 8 * gbci<j>(offv));
                    });
                    __builtin_amdgcn_sched_barrier(0);
                    double s0 = 0.0, s1 = 0.0; // two chains
                    for_each_index<0, 16>([&](auto jj) {
                        constexpr int j = decltype(jj)::value;
                        if (j & 1)
                            s1 = dfma(-uj[j], gbc<j>(xv), s1);
                        else
                            s0 = dfma(-uj[j], gbc<j>(xv), s0);
                    });
                    sv += s0 + s1;
                }
                sv = sel(gl < rank, sv, 0.0);
                for_each_index<1, MD>([&](auto jj) {
                    constexpr int j = MD - decltype(jj)::value; // MD-1 .. 1
                    sv = dfma(-col[j], gbc<j>(sv), sv); // unit diagonal; col[j] is zero at and below the diagonal and beyond the rank
                });
                D(sel(gl < rank, o_xs + 8 * (Fc + gl), o_stage + 8 * gl)) = sv;
                quad_lds_fence();
            }
            STAMP(9)
            // ---- results ----
            if (live)
            {
#pragma unroll
                for (int s = 0; s < NS; s++)
                {
                    const int P = 16 * s + gl; // every position once, whatever the layout offset
                    if (P < n)
                    {
                        a.x[(size_t)b * n + B8(o_phys + P)] = D(o_xs + 8 * P); // x = P x: the variable at position P (lexlse.h:1044)
                        a.perm[(size_t)b * n + P]           = (P < TotalRank) ? (uint32_t)B8(o_perm + P) : (uint32_t)P;
                    }
                }
                if (gl < nObj)
                {
                    a.fcol[(size_t)b * nObj + gl] = U32(o_meta + 16 * gl);
                    a.rank[(size_t)b * nObj + gl] = U32(o_meta + 16 * gl + 4);
                }
                if (gl == 0) a.totalrank[b] = (uint32_t)TotalRank;
            }
            STAMP(10)
            STAMP_WRITE
#ifdef LEXLS_QTOL_CHAIN
            if (lane == 0)
                for (int i_ = 0; i_ < 7; i_++) a.lambda[(size_t)b * (n + cap) + 30 + i_] = (double)cacc[i_];
#endif
        }

        /// exact worst case of the triangular images: sum_k ((n+1-Fc_k) rank_k - rank_k (rank_k - 1) / 2) over rank distributions with rank_k <= md
        inline uint32_t qtol_image_doubles(uint32_t n, uint32_t nObj, uint32_t md)
        {
            uint32_t fc = 0, total = 0;
            for (uint32_t k = 0; k < nObj && fc < n; k++)
            {
                const uint32_t r = md < n - fc ? md : n - fc;
                total += (n + 1 - fc) * r - r * (r - 1) / 2;
                fc += r;
            }
            return (total + 1) & ~1u;
        }

        template <int NS, int MD>
        inline size_t qtol_group_bytes(uint32_t n, uint32_t nObj)
        {
            // staging block: the level pieces (half the rows at a time) or the sixteen hand-off slots of the pivot steps, whichever is larger
            const size_t stage = 8 * (size_t)(n + 1) * (MD / 2) > 16u * (8 * MD + 16) ? 8 * (size_t)(n + 1) * (MD / 2) : 16u * (8 * MD + 16);
            const size_t raw   = 8 * ((size_t)qtol_image_doubles(n, nObj, MD) + 16 * NS + MD) + 64 + 64 + 16 * kQuadMaxObj + 8 * 16 * NS + stage;
            // rounded up to an ODD multiple of 128 bytes: the four problems of a wavefront read the same relative addresses of their slices at
            // the same time; slices half a bank row apart do not collide
            return ((raw + 127) / 256) * 256 + 128;
        }

        template <int NS, int MD, int SIG, int NV>
        hipError_t launch_qtol_t(const LseArgs &a, hipStream_t s)
        {
            const uint32_t img  = qtol_image_doubles(a.nVar, a.nObj, MD);
            const size_t gbytes = qtol_group_bytes<NS, MD>(a.nVar, a.nObj);
            const size_t lds    = 4 * QT_WPB * gbytes;
            if (lds > kMaxLdsBytes || a.nObj > (uint32_t)kQuadMaxObj || a.nVar + 1 + SIG > 16u * NS || a.nVar > 63u || (NV && a.nVar != (uint32_t)NV)) return hipErrorInvalidValue;
            if (a.uniform_dim != (uint32_t)MD || (a.cap & 1u) || (reinterpret_cast<uintptr_t>(a.in) & 15u) || a.nfixed || a.reg_type != 0) return hipErrorInvalidValue;
            if (lds > 64 * 1024)
            {
                static size_t granted[64] = {0}; // per device: the attribute is set once, not per launch
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
                if (dev < 0 || granted[dev] < lds)
                {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lqr_qtol_kernel<NS, MD, SIG, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (e != hipSuccess) return e;
                    if (dev >= 0) granted[dev] = lds;
                }
            }
            const uint32_t blocks = (a.batch + 4u * QT_WPB - 1u) / (4u * QT_WPB);
            static const uint32_t stagger = std::getenv("LEXLS_QTOL_STAGGER") ? (uint32_t)std::atoi(std::getenv("LEXLS_QTOL_STAGGER")) : 0u; // x 512 cycles per SIMD index
            hipLaunchKernelGGL((lqr_qtol_kernel<NS, MD, SIG, NV>), dim3(blocks), dim3(64 * QT_WPB), lds, s, a, img, (uint32_t)gbytes, stagger);
            return hipGetLastError();
        }
    } // namespace
} // namespace lexls

#define LEXLS_QTOL_INSTANCE(NAME, NS, MD, SIG, NV) \
    namespace lexls { hipError_t NAME(const LseArgs &a, hipStream_t s) { return launch_qtol_t<NS, MD, SIG, NV>(a, s); } \
                      size_t NAME##_lds(uint32_t nVar, uint32_t nObj) { return 4 * QT_WPB * qtol_group_bytes<NS, MD>(nVar, nObj); } }
