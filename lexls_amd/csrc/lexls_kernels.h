// Device-side argument block shared by every kernel of liblexls_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lexls
{
    // ConstraintActivationType values (reference typedefs.h:69-76)
    enum : uint8_t
    {
        CTR_INACTIVE           = 0,
        CTR_ACTIVE_LB          = 1,
        CTR_ACTIVE_UB          = 2,
        CTR_ACTIVE_EQ          = 3,
        CORRECT_SIGN_OF_LAMBDA = 4
    };

    /// One batch of equality problems resident in HBM.  All per-problem arrays are dense, problem-major.
    struct LseArgs
    {
        uint32_t batch, nVar, nObj, cap; // cap = sum(maxObjDim) = leading dimension of one problem
        uint32_t ldp;                    // padded (odd) leading dimension of the LDS image
        double tol;                      // tol_linear_dependence (compared with a SQUARED norm, lexlse.h:214)
        const double *in;                // batch x cap x (nVar+1)  problem data (read-only)
        double *fac;                     // batch x cap x (nVar+1)  factor
        double *x;                       // batch x nVar
        double *hh;                      // batch x cap
        uint32_t *perm;                  // batch x nVar
        uint32_t *rank;                  // batch x nObj
        uint32_t *fcol;                  // batch x nObj
        uint32_t *totalrank;             // batch
        const uint32_t *dims;            // batch x nObj
        uint32_t uniform_dim;            // != 0: EVERY level of EVERY problem has exactly this many rows (host-side knowledge; 0 = ragged / unknown)
        const uint32_t *nfixed;          // batch (NULL: no fixed variables anywhere)
        const uint32_t *fixed_idx;       // batch x nVar
        const double *fixed_val;         // batch x nVar
        uint8_t *fixed_type;             // batch x nVar
        uint8_t *ctr_type;               // batch x cap
        double *v;                       // batch x cap
        double *lambda;                  // batch x (nVar + cap)
        int32_t *sens;                   // batch x 3
        double *maxabs;                  // batch
        double *scratch;                 // batch x 2 x nVar x nVar (least-norm only, may be NULL)
        uint32_t reg_type;               // LexLS::RegularizationType (0 = none); != 0 is served by the generic kernel only
        uint32_t reg_cg_iters;           // max_number_of_CG_iterations (typedefs.h:111), CG variants only
        double reg_variable;             // variable_regularization_factor (typedefs.h:116), 0 = constant factors
        const double *reg_factor;        // batch x nObj regularization factors (lexlse.h:1477)
        double *reg_scratch;             // batch x reg_scratch_doubles(nVar): null-space basis + work matrices (lexls_regularize.h)
        double *reg_mu;                  // batch x reg_mu_doubles(nVar, nObj, cap): X_mu, X_mu_rhs, residual_mu of reg_type 7 (else NULL)
        const uint8_t *skip;             // batch flags: non-zero = leave this problem untouched (NULL: none); lock-step LSI batches
        // Row gather fused into the load of lqr_wave_kernel (NULL: `in` holds the assembled problems).  Row r of problem b is row
        // row_src[b*cap+r] of the resident constraint data: element j at g_cdata[b*g_per + row_src + j*ld], right-hand side at column
        // nVar (+1 when the top bit of row_ld is set): the layout lexls_lse_gather_problem documents (include/lexls_hip.h)
        const double *g_cdata;
        uint64_t g_per;
        const uint32_t *g_row_src, *g_row_ld;
        // Prefix reuse (the reference refactorizes everything every iteration, README.md:14): a LexLSI iteration changes ONE
        // row of ONE level, so the levels above it factorize to what they did before.  resume_state (batch x resume_state_bytes(nObj), NULL: off)
        // receives what the register-resident wave kernel needs to pick a factorization up again; resume_level[b] = K > 0 (NULL: nobody)
        // says that levels 0 .. K-1 of problem b — rows, dimensions, fixed variables — are those of its previous factorization by that
        // kernel into the SAME factor buffer: they are read back instead of factorized, and only their elimination of the rows from level K
        // on is redone.  Same instructions on the same operands for everything that is recomputed: results identical bit for bit.
        const int32_t *resume_level;
        uint8_t *resume_state;
    };
    /// per problem: the position of every physical column after each level (nObj x 64 bytes) and after the last one (64 bytes)
    __host__ __device__ inline size_t resume_state_bytes(uint32_t nObj) { return 64 * ((size_t)nObj + 1); }
} // namespace lexls
