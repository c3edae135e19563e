/* lexls_hip.h — C ABI of the MI355X-native lexicographic-QR core (liblexls_hip.so).
 *
 * The reference (jrl-umi3218/lexls) has no FFI: its boundary for this path is the C++ class
 * LexLS::internal::LexLSE (include/lexls/lexlse.h:33-2886), one object = one problem, all buffers
 * owned by the object.  This ABI is that class turned into a handle over a BATCH of independent
 * problems of the same capacity (the solver's intended workload: successive IK-style instances),
 * each entry point citing the member function it replaces.  Plain pointers and sizes only; no C++
 * or torch types cross the boundary; errors are status codes + lexls_last_error(), never exceptions.
 *
 * Data layout (identical to the reference's storage, lexlse.h:85): one problem is a column-major
 * cap x (nVar+1) array "LOD" with leading dimension cap = sum(maxObjDim); rows [0, nCtr) hold the
 * stacked levels [A_k | b_k] (column nVar = right-hand side).  A batch is `batch` such arrays
 * back to back.  Index = uint32_t, RealScalar = double (typedefs.h:16-17).
 *
 * Host pointers are named h_*, device pointers d_*.  All device work is enqueued on the handle's
 * HIP stream (default: the null stream); calls that return data to the host synchronise that stream.
 */
#ifndef LEXLS_HIP_H
#define LEXLS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lexls_lse_s *lexls_lse_t;

enum lexls_status
{
    LEXLS_OK              = 0,
    LEXLS_ERR_INVALID     = 1, /* bad argument / call order (the reference throws LexLS::Exception) */
    LEXLS_ERR_HIP         = 2, /* a HIP runtime call failed (message has the HIP error string)      */
    LEXLS_ERR_UNSUPPORTED = 3, /* feature of the reference that has no device path yet              */
    LEXLS_ERR_NO_DEVICE   = 4  /* no usable GPU: there is NO CPU fallback in this library           */
};

/* which device array lexls_lse_device_ptr returns */
enum lexls_array
{
    LEXLS_ARRAY_X = 0,      /* double   batch x nVar            solution                              */
    LEXLS_ARRAY_FACTOR,     /* double   batch x cap x (nVar+1)  factor ("lexqr")                      */
    LEXLS_ARRAY_HH,         /* double   batch x cap             Householder scalars                   */
    LEXLS_ARRAY_PERM,       /* uint32   batch x nVar            column_permutations                   */
    LEXLS_ARRAY_RANK,       /* uint32   batch x nObj            obj_info[k].rank                      */
    LEXLS_ARRAY_FIRST_COL,  /* uint32   batch x nObj            obj_info[k].first_col_index           */
    LEXLS_ARRAY_TOTAL_RANK, /* uint32   batch                   TotalRank                             */
    LEXLS_ARRAY_V,          /* double   batch x cap             residuals (get_v)                     */
    LEXLS_ARRAY_LAMBDA,     /* double   batch x (nVar+cap)      [lambda_fixed; lambda] of the last sensitivity call */
    LEXLS_ARRAY_INPUT       /* double   batch x cap x (nVar+1)  library-owned input buffer            */
};

/* replaces LexLS::Exception::what() (typedefs.h:300-314): no exception crosses the ABI — every entry point returns a status code and
 * leaves the message of the last failure here */
const char *lexls_last_error(void);
int lexls_version(void);

/* number of visible HIP devices; LEXLS_ERR_NO_DEVICE if none */
int lexls_device_count(int *count);

/* ---- lifetime -------------------------------------------------------------------------------- */

/* replaces LexLSE::LexLSE(nVar,nObj,ObjDim) / resize() (lexlse.h:50-103): allocates every device
 * buffer once for `batch` problems of capacity maxObjDim[k] rows per level. */
int lexls_lse_create(lexls_lse_t *out, int device, uint32_t batch, uint32_t nVar, uint32_t nObj, const uint32_t *h_maxObjDim);
int lexls_lse_destroy(lexls_lse_t h);
/* all later work of this handle runs on `hip_stream` (a hipStream_t; NULL = null stream) */
int lexls_lse_set_stream(lexls_lse_t h, void *hip_stream);
int lexls_lse_synchronize(lexls_lse_t h);
/* Deferred synchronisation (lock-step LexLSI batches issue ~10 small copies per active-set round): while on, the set_* / gather / get_*
 * calls only ENQUEUE their copies on the handle's stream.  The caller then owes two things the default mode does not ask for: host input
 * arrays stay untouched, and host output arrays are only read, after the next lexls_lse_synchronize(); and the arrays should be pinned
 * (hipHostMalloc) — with pageable memory the runtime stages the copies and nothing is gained. */
int lexls_lse_set_deferred_sync(lexls_lse_t h, int on);

/* ---- problem definition ----------------------------------------------------------------------- */

/* replaces setParameters (lexlse.h:1467); only REGULARIZATION_NONE has a device path (typedefs.h:122) */
int lexls_lse_set_tolerance(lexls_lse_t h, double tol_linear_dependence);
/* replaces setObjDim (lexlse.h:1426): h_dims is nObj values (per_problem = 0, same for the whole
 * batch) or batch x nObj values (ragged batch, per_problem = 1); each dims[k] <= maxObjDim[k] */
int lexls_lse_set_obj_dim(lexls_lse_t h, const uint32_t *h_dims, int per_problem);
/* replaces setFixedVariablesCount + fixVariable(s) (lexlse.h:1381-1419, :1449): per problem the
 * number of fixed variables, then (batch x nVar, first nfixed[b] entries used) index / value /
 * ConstraintActivationType.  h_nfixed == NULL clears all fixed variables. */
int lexls_lse_set_fixed(lexls_lse_t h, const uint32_t *h_nfixed, const uint32_t *h_index, const double *h_value, const uint8_t *h_type);
/* activation types of the fixed variables only (batch x nVar bytes, fixVariable order, lexlse.h:1381-1419); unlike lexls_lse_set_fixed
 * it keeps the factorization valid — the types only matter to ObjectiveSensitivity (lexlse.h:866-987 marks CORRECT_SIGN_OF_LAMBDA) */
int lexls_lse_set_fixed_type(lexls_lse_t h, const uint8_t *h_type);
/* replaces setCtrType (lexlse.h:1548): batch x cap ConstraintActivationType bytes, row order of LOD */
int lexls_lse_set_ctr_type(lexls_lse_t h, const uint8_t *h_types);
/* lock-step batches (batched LexLSI): problems whose flag is non-zero are left untouched by the next
 * factorize / factorize_solve calls (their previous results stay valid).  NULL clears the mask. */
int lexls_lse_set_skip(lexls_lse_t h, const uint8_t *h_skip);
/* replaces setProblem / setData (lexlse.h:1511-1530): copies batch x cap x (nVar+1) doubles H2D */
int lexls_lse_set_problem_host(lexls_lse_t h, const double *h_lod);
/* zero-copy variant: the caller's device buffer becomes the (read-only) input of later factorizations */
int lexls_lse_set_problem_device(lexls_lse_t h, const double *d_lod);
/* Device-side assembly of the equality problems of a LexLSI iteration — replaces the row copies of Objective::formLexLSE
 * (objective.h:434-494).  set_constraint_data keeps, per problem, `per_problem` doubles of constraint data
 * resident on the device (the objectives' [A | lb | ub] blocks back to back, each column-major — the flat layout of lexls_lsi_solve);
 * gather_problem then builds every non-skipped problem's LOD from two batch x cap uint32 arrays: row r of the LOD is
 * [ data[src + j*ld], j < nVar | data[src + (nVar + ub)*ld] ] with src = h_row_src[r], ld = h_row_ld[r] & 0x7fffffff,
 * ub = h_row_ld[r] >> 31 (0: right-hand side = lb, 1: ub — objective.h:472-486); rows with h_row_ld[r] == 0 are left alone. */
int lexls_lse_set_constraint_data(lexls_lse_t h, const double *h_data, uint64_t per_problem);
int lexls_lse_gather_problem(lexls_lse_t h, const uint32_t *h_row_src, const uint32_t *h_row_ld);

/* One copy each way per active-set round (lock-step LexLSI batches).  The handle keeps the small per-round arrays in two device slabs;
 * lexls_lse_round_layout gives the byte offset of each array inside them so that a host block of the same layout can be moved with one
 * copy instead of one per array:
 *   in slab  : dims (u32 batch x nObj) | nfixed (u32 batch) | fixed_idx (u32 batch x nVar) | fixed_val (f64 batch x nVar) | skip (u8 batch)
 *              | obj_index (i32 batch) | row_src, row_ld (u32 batch x cap) | fixed_type (u8 batch x nVar) | ctr_type (u8 batch x cap)
 *   out slab : x (f64 batch x nVar) | total_rank (u32 batch) | found (i32 batch x 3) | max_abs (f64 batch)
 * upload_round  = set_obj_dim(per_problem) + set_fixed + set_ctr_type + set_skip + the obj_index of a later sensitivity_resident
 *                 (+ gather_problem when gather != 0), with the same argument checks;
 * download_round: h_out receives the out slab (out_bytes), h_types the tail of the in slab from fixed_type on
 *                 (in_bytes - fixed_type bytes: fixed_type, then ctr_type at offset ctr_type - fixed_type); either may be NULL. */
typedef struct lexls_round_layout
{
    uint64_t in_bytes, dims, nfixed, fixed_idx, fixed_val, skip, obj_index, row_src, row_ld, fixed_type, ctr_type;
    uint64_t out_bytes, x, total_rank, found, max_abs;
} lexls_round_layout;
int lexls_lse_round_layout(lexls_lse_t h, lexls_round_layout *out);
int lexls_lse_upload_round(lexls_lse_t h, const void *h_in, int gather);
int lexls_lse_download_round(lexls_lse_t h, void *h_out, void *h_types);

/* ---- the hot path ------------------------------------------------------------------------------ */

/* replaces factorize() (lexlse.h:117-506): factor, Householder scalars, pivots, ranks on device */
int lexls_lse_factorize(lexls_lse_t h);
/* replaces solve() (lexlse.h:1015-1045); needs a factorization */
int lexls_lse_solve(lexls_lse_t h);
/* factorize()+solve() in one launch per batch; keep_factor = 0 skips writing the factor to HBM
 * (x, ranks and pivots only — the "x-only" traffic variant of SURVEY.md section 8(d)) */
int lexls_lse_factorize_solve(lexls_lse_t h, int keep_factor);
/* replaces solveLeastNorm_1() (lexlse.h:1052-1131, Givens sweep); needs a factorization */
int lexls_lse_solve_least_norm(lexls_lse_t h);
/* replaces setParameters(regularization_type, variable_regularization_factor) + setRegularizationFactor (lexlse.h:1467, :1477;
 * dispatch lexlse.h:277-411).  type: LexLS::RegularizationType — NONE 0, TIKHONOV 1, R 3, R_NO_Z 4, RT_NO_Z 5, TIKHONOV_2 8, TEST 9
 * the CGLS variants TIKHONOV_CG 2, RT_NO_Z_CG 6, and TIKHONOV_1 7 (the type the reference marks experimental: regularize_tikhonov_1_test
 * lexlse.h:1774-1886; its ObjectiveSensitivity then returns the multipliers of the regularized problem, :647-651; generic kernel only;
 * by-products: lexls_lse_get_mu).  h_factors: one factor per level (per_problem == 0) or batch x nObj (per_problem != 0); NULL = all zero. */
int lexls_lse_set_regularization(lexls_lse_t h, int type, const double *h_factors, int per_problem, double variable_factor);
/* max_number_of_CG_iterations (typedefs.h:111, default 10): iteration cap of the two CGLS variants */
int lexls_lse_set_cg_iterations(lexls_lse_t h, uint32_t max_iterations);
/* replaces solveLeastNorm_3() (lexlse.h:1222-1277): least-norm solution from the null-space basis accumulated by the Tikhonov family
 * (regularization type 1, 2, 3 or 8, normally with all factors zero) */
int lexls_lse_solve_least_norm_3(lexls_lse_t h);
/* replaces solveLeastNorm_2() (lexlse.h:1138-1213, normal equations of the free variables + Cholesky); needs a factorization */
int lexls_lse_solve_least_norm_2(lexls_lse_t h);
/* replaces get_v() (lexlse.h:1560-1582); needs a factorization */
int lexls_lse_residual(lexls_lse_t h);
/* replaces bool ObjectiveSensitivity(ObjIndex, CtrIndex2Remove, ObjIndex2Remove, tolWrong, tolCorrect,
 * maxAbsValue) (lexlse.h:611-762) incl. findDescentDirection (:935-987) and its mutation of ctr_type.
 * h_obj_index: one level per problem (batch values; a negative value skips that problem), or NULL
 * with `obj_index_all` applied to every problem.  Results: lexls_lse_get_sensitivity. */
int lexls_lse_sensitivity(lexls_lse_t h, const int32_t *h_obj_index, int32_t obj_index_all, double tol_wrong_sign_lambda, double tol_correct_sign_lambda);
/* While on, lexls_lse_sensitivity(_resident) does what LexLSI's removal search does with one call per level (lexlsi.h:1121-1132): starting
 * at the given level it goes on to the next ones until a level reports a wrong-sign multiplier or the last level is done — one launch; the
 * outputs are those of the level it stopped at.  Off by default (= the reference's one-level call). */
int lexls_lse_set_sensitivity_scan(lexls_lse_t h, int on);
/* same, with the per-problem objective indices already on the device (the obj_index array of lexls_lse_upload_round) */
int lexls_lse_sensitivity_resident(lexls_lse_t h, double tol_wrong_sign_lambda, double tol_correct_sign_lambda);

/* ---- results (synchronise the stream, then D2H) ------------------------------------------------- */
int lexls_lse_get_x(lexls_lse_t h, double *h_x);                    /* get_x()      lexlse.h:1587 */
int lexls_lse_get_factor(lexls_lse_t h, double *h_lod);             /* get_lexqr()  lexlse.h:1626 */
int lexls_lse_get_hh_scalars(lexls_lse_t h, double *h_hh);
int lexls_lse_get_permutation(lexls_lse_t h, uint32_t *h_perm);
int lexls_lse_get_ranks(lexls_lse_t h, uint32_t *h_rank, uint32_t *h_first_col, uint32_t *h_total_rank); /* getRank/getTotalRank :1503,:1603 */
int lexls_lse_get_v(lexls_lse_t h, double *h_v);                    /* get_v()      lexlse.h:1560 */
/* replaces get_X_mu() / get_X_mu_rhs() / get_residual_mu() (lexlse.h:1636-1650), filled with regularization type 7 only: h_x_mu and
 * h_x_mu_rhs are batch x nObj x nVar (column k of the reference's nVar x nObj matrix is contiguous), h_residual_mu is batch x cap;
 * NULL = not wanted.  X_mu_rhs columns are written by lexls_lse_sensitivity (initialize_rhs, :1921-1959). */
int lexls_lse_get_mu(lexls_lse_t h, double *h_x_mu, double *h_x_mu_rhs, double *h_residual_mu);
int lexls_lse_get_lambda(lexls_lse_t h, double *h_lambda);          /* getWorkspace() after ObjectiveSensitivity, lexlse.h:1621 */
/* h_found_ctr_obj: batch x 3 int32 {found, CtrIndex2Remove, ObjIndex2Remove}; h_max_abs: batch */
int lexls_lse_get_sensitivity(lexls_lse_t h, int32_t *h_found_ctr_obj, double *h_max_abs);
int lexls_lse_get_ctr_type(lexls_lse_t h, uint8_t *h_types);
int lexls_lse_get_fixed_type(lexls_lse_t h, uint8_t *h_types);   /* batch x nVar: activation types of the fixed variables incl. the CORRECT_SIGN_OF_LAMBDA marks (lexlse.h:866-987) */
/* raw device pointer of one of the handle's arrays (enum lexls_array), for zero-copy consumers */
int lexls_lse_device_ptr(lexls_lse_t h, int which, void **d_ptr);

/* name of the kernel variant the last factorize/factorize_solve call dispatched to (diagnostics) */
const char *lexls_lse_last_kernel(lexls_lse_t h);
/* Kernel policy — which CONTRACT a solve is held to, and which kernel family serves it.
 *   Contracts: (B) bit-identical to the arithmetic contract of oracle/lexlse_oracle.h (pivots, ranks, Householder scalars, factor, x, multipliers);
 *              (T) BASELINE north_star's: column permutation, ranks and first columns exact, x (and factor MAGNITUDES) within 1e-10
 *                  (relative to max(1, |x|_inf)) — on problems whose own solution is determined that well.  An ill-conditioned problem
 *                  (tiny pivots above the rank tolerance, rows / columns scaled over many decades), whose x moves by more than ~1e-11 when
 *                  its DATA move by one ulp, is solved to a small multiple of that sensitivity instead (scripts/soak_qtol.py: 21 k
 *                  random batches, 92 such problems beyond 1e-10, at most 20 x their one-ulp sensitivity — 47 x once levels of eight rows
 *                  joined the soak, 91 x on lqr_mfma's soak; the soaks' bound is 100 x, a random one-ulp perturbation being a LOWER
 *                  estimate of what rounding does to such a problem —; pivots and ranks exact in all).
 *                  PIVOT RULE under (T).  The reference takes the first maximum of the down-dated column norms (lexlse.h:205-206).  lqr_mfma compares
 *                  the norms by VALUE (whole doubles; equal values: the smallest position) — the reference's rule on this kernel's own norms.
 *                  lqr_qtol compares them in ONE max butterfly on a packed key whose low 12 mantissa bits carry the position: two candidates whose
 *                  norms agree in their upper 40 mantissa bits (relative difference below 2^-40 = 9.1e-13) are ordered BY POSITION, whatever their
 *                  last 12 bits say.  Exact ties (duplicated columns) are ordered as the reference orders them by either kernel; norms that differ by
 *                  1e-11 relative or more are ordered by value by both (tests/test_gpu_qtol.py, tests/test_gpu_mfma.py: near-tie cases).
 *   policy 0 = automatic dispatch.  (T) for x-only solves whose levels ALL have 12 rows (or ALL 8: round 4), no fixed variables, no regularization, n <= 40 (the IK shape of
 *              BASELINE configs[2]/[3] and its smaller relatives):
 *              lqr_qtol, the bench kernel — and for problems beyond one CU's LDS (the step-per-pivot path with the trailing update on the matrix
 *              cores; there the reflector of a row that exactly repeats a row of an earlier level may come out with the opposite SIGN — that row of R
 *              and its essential part are negated, x and everything else agree: consumers of get_lexqr / hh scalars that need sign parity with the
 *              ordered-chain arithmetic ask for policy 5).  (B) for everything else: with the factor kept the register-resident wave kernel while
 *              the batch fits one round of it, beyond that the four-per-wavefront kernel's factor-keeping form (the left-looking wave kernel
 *              where that does not fit); x-only solves of other small shapes the four-per-wavefront kernel.  LEXLS_QTOL=0 in the environment keeps
 *              every small-shape solve on (B).
 *   1 = only the generic one-workgroup-per-problem kernel (B);  2 = automatic, but never the left-looking / four-per-wavefront kernels (B);
 *   3 = the left-looking wave kernel whenever the shape allows it, whatever the batch size (B);
 *   4 = the bit-exact four-problems-per-wavefront kernel whenever the shape allows it (x-only solves; else as 3) (B);
 *   5 = automatic with (B) everywhere: small shapes as under LEXLS_QTOL=0, large problems on the bit-exact multi-launch path (ordered chains, two
 *       launches per pivot) — the policy for factor / sign parity;
 *   6 = the tolerance-contract kernel lqr_qtol wherever it serves (T), else as 0;
 *   7 = the matrix-core tolerance-contract kernel lqr_mfma (lexls_amd/csrc/lqr_mfma_impl.h: two problems per wavefront, two wavefronts per SIMD, the
 *       Gauss step of lexlse.h:431-471 on v_mfma_f64_16x16x4_f64, finished levels kept in reduced form) wherever it serves (T) — x-only solves
 *       whose levels all have 12 rows, n + 1 <= 48, no fixed variables / regularization, two workgroups' LDS slices per CU (n = 40: up to 5 levels) —
 *       else as 6;  8 = the same kernel with one problem per wavefront (four wavefronts per SIMD);  9 = with four problems per wavefront (the IK
 *       shape only).  Automatic dispatch (0) takes lqr_qtol first — the faster one on MI355X (41 us against 57 us per 4096 IK problems) — and
 *       lqr_mfma for the shapes lqr_qtol's slices do not hold.
 *   Policy 0 is therefore NOT bit-exact for those x-only solves; a caller that needs (B) everywhere sets policy 5 (per handle) or runs under
 *   LEXLS_QTOL=0 (whole process; read at every factorization, so it may be changed between solves). */
int lexls_lse_set_kernel_policy(lexls_lse_t h, int policy);

/* ---- prefix reuse ------------------------------------------------------------------------------
 * The reference refactorizes the whole hierarchy in every LexLSI iteration although one row of one level changed (README.md:14 "No update
 * mechanism ... each iteration of the solver performs a full decomposition"; the loop at lexlsi.h:1144-1172 calls factorize() on the whole
 * problem).  Here a factorization can pick up the previous one of the same handle: with
 *     lexls_lse_set_prefix_reuse(h, 1)
 * every factor-keeping factorization by the register-resident wave kernel (IK-sized problems: nVar + 1 <= 64 columns, <= 64 rows — the kernel of
 * a LexLSI stage; lexls_lse_prefix_reuse_ready(h) says whether the last one was) also leaves the position map after each level, and
 *     lexls_lse_set_resume_levels(h, levels[batch])
 * tells the NEXT factorization that, for problem b, levels 0 .. levels[b]-1 — their rows, dimensions and the fixed variables — are exactly
 * those of its previous factorization, which still sits in the factor buffer.  Those levels are read back instead of factorized (no pivot
 * search, no reflectors: the dependent chains that make up most of the kernel's time); their elimination of the rows from level levels[b] on is
 * redone with the same instructions on the same operands.  Factor, permutation, ranks, Householder scalars and x are IDENTICAL, bit for bit, to
 * a full factorization (tests/test_gpu_prefix_reuse.py).  levels[b] = 0: factorize everything.  The levels are consumed by that factorization.
 * The caller vouches for "unchanged": the kernel does not compare rows.  Regularization, x-only solves and the other kernels ignore the request
 * (full factorization) and leave nothing to resume from. */
int lexls_lse_set_prefix_reuse(lexls_lse_t h, int enable);
int lexls_lse_prefix_reuse_ready(lexls_lse_t h);
int lexls_lse_set_resume_levels(lexls_lse_t h, const int32_t *h_levels);

/* ---- inequality problems: the reference's LexLSI active-set driver (lexlsi.h), kept on the host -------------
 * The driver is host C++ (include/lexls/lexlsi.h, same logic as the reference's lexlsi.h/objective.h/workingset.h);
 * every factorize / solve / ObjectiveSensitivity it issues goes to the HIP kernels above.  Call sequence = the
 * reference's MEX front end (interfaces/matlab-octave/lexlsi.cpp:527-625).
 *   h_dims[nObj]; h_types[nObj]: 0 general, 1 simple bounds (objective 0 only, typedefs.h:60-64);
 *   h_data: objectives back to back, column-major: general dim x (nVar+2) = [A lb ub], simple dim x 2 = [lb ub];
 *   h_var_index[dims[0]]: 0-based variable indices of a simple-bounds objective (else NULL);
 *   h_active_guess: sum(dims) ConstraintActivationType bytes or NULL; h_x0: nVar or NULL;
 *   h_params9: {max_number_of_factorizations, tol_linear_dependence, tol_wrong_sign_lambda, tol_correct_sign_lambda,
 *               tol_feasibility, cycling_handling_enabled, cycling_max_counter, cycling_relax_step,
 *               deactivate_first_wrong_sign} or NULL for the defaults of typedefs.h:268-294;
 *   outputs: h_x[nVar]; h_info6 = {status, iterations, activations, deactivations, factorizations, total_rank};
 *            h_active[sum(dims)] final working set; h_v[sum(dims)] constraint violations (either may be NULL). */
int lexls_lsi_solve(int device, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types, const double *h_data,
                    const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0, const double *h_params9, double *h_x,
                    int32_t *h_info6, uint8_t *h_active, double *h_v);
/* A batch of LexLSI problems of ONE structure (same nVar, dims, types; different data), advanced in LOCK STEP: every
 * active-set round issues one batched factorize+solve and one batched ObjectiveSensitivity per LexLSE level for all the
 * instances that need it (BASELINE configs[4]).  Arrays are the per-problem arrays of lexls_lsi_solve, back to back
 * (h_var_index: batch x dims[0]; h_active_guess / h_x0 may be NULL); h_rounds2 (may be NULL) receives
 * {factorize+solve stages, sensitivity stages} actually issued to the device.  The instances are split into g groups (default: 2 from 512
 * instances on, else 1; LEXLS_LSI_GROUPS=g overrides) that take turns: one group's stage runs on the GPU while the host advances the other
 * groups' active sets. */
int lexls_lsi_batch_solve(int device, uint32_t batch, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types,
                          const double *h_data, const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0,
                          const double *h_params9, double *h_x, int32_t *h_info6, uint8_t *h_active, double *h_v, int32_t *h_rounds2);
/* lexls_lsi_batch_solve with the regularization inputs of lexls_lsi_solve_ex: h_reg_factors = one factor per objective, shared by the batch,
 * or NULL; h_params with nparams == 9 or 12 (+ regularization_type, variable_regularization_factor, max_number_of_CG_iterations).
 * Regularized batches run their factorizations on the generic kernel. */
int lexls_lsi_batch_solve_ex(int device, uint32_t batch, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types,
                             const double *h_data, const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0,
                             const double *h_reg_factors, const double *h_params, uint32_t nparams, double *h_x, int32_t *h_info6,
                             uint8_t *h_active, double *h_v, int32_t *h_rounds2);
/* The same batch as an object that outlives one solve — the way the reference uses LexLSI (constructed and sized once, lexlsi.h:56-112, then
 * fed successive problems): lexls_lsi_batch_create makes the device buffers, pinned blocks, streams and the host worker pool for `batch`
 * problems of the structure (nVar, dims, types); every lexls_lsi_batch_run solves `batch` new problems of that structure (arguments as
 * lexls_lsi_batch_solve_ex).  lexls_lsi_batch_solve(_ex) = create + run + destroy; a serving loop saves the 6-8 ms of create per call. */
typedef struct lexls_lsi_batch_s *lexls_lsi_batch_t;
int lexls_lsi_batch_create(lexls_lsi_batch_t *out, int device, uint32_t batch, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types);
int lexls_lsi_batch_run(lexls_lsi_batch_t b, const double *h_data, const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0,
                        const double *h_v0 /* batch x sum(dims) initial residuals (set_v0, lexlsi.cpp:571-588) or NULL */, const double *h_reg_factors, const double *h_params, uint32_t nparams, double *h_x, int32_t *h_info6, uint8_t *h_active,
                        double *h_v, int32_t *h_rounds2);
/* How a run executes: phase 1 of every instance on the host; from then on the instance's active-set iterations are resident on the
 * device (LEXLS_LSI_RESIDENT=0: host logic, lock-step stages).  Where the batch's shape has a persistent instantiation (the register-resident l-QR shapes:
 * nVar + 1 <= 41 with levels of up to 12 rows, nVar + 1 <= 64 with levels of up to 16 — except 42..48 columns), everything behind the first resident
 * stage is ONE launch: per instance l-QR -> step -> removal search behind an unblocked step -> working-set change, until the instance stops
 * (LEXLS_LSI_NO_FUSED=1, read per run: three launches per lock-step stage instead; same results bit for bit).
 * of the last lexls_lsi_batch_run: {factorize+solve stages, sensitivity stages, stages whose iteration step ran on the device, groups}.
 * The step of an iteration (A*dx, ratio test, update of x / v / A*x: lexlsi.h:987-1029, :1234-1240) runs on the device
 * next to the equality solve when the batch is created with LEXLS_LSI_DEVICE_STEP=1 in the environment; off by default. */
int lexls_lsi_batch_stats(lexls_lsi_batch_t b, int32_t *h_stats4);
int lexls_lsi_batch_destroy(lexls_lsi_batch_t b);
/* lexls_lsi_solve plus what the MEX front end also passes (interfaces/matlab-octave/lexlsi.cpp:527-625): h_v0 = initial residuals,
 * sum(dims) doubles (set_v0 per objective) or NULL; h_reg_factors = one regularization factor per objective or NULL; h_params with
 * nparams == 9 (as lexls_lsi_solve) or 12: + regularization_type, variable_regularization_factor, max_number_of_CG_iterations. */
int lexls_lsi_solve_ex(int device, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types, const double *h_data,
                       const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0, const double *h_v0,
                       const double *h_reg_factors, const double *h_params, uint32_t nparams, double *h_x, int32_t *h_info6, uint8_t *h_active,
                       double *h_v);
/* The fifth output of the MEX front end, `[x, info, v, as, d] = lexlsi(...)` (interfaces/matlab-octave/lexlsi.cpp:739-770,
 * formDebugStructure :77-260): lexls_lsi_solve_ex with the working-set log on (ParametersLexLSI::log_working_set_enabled), followed by the
 * getter sequence of :752-762.  Every pointer of `debug` may be NULL.  total = sum(dims); rows = row capacity of the equality solver
 * (total, minus dims[0] when objective 0 holds simple bounds); nObjL = its number of levels. */
typedef struct lexls_lsi_debug
{
    double *lambda;       /* total x nObj, column-major: getLambda() lexlsi.h:552-605, objectives stacked, user's constraint order   */
    double *lexqr, *data; /* rows x (nVar+1), column-major, ld = rows: get_lexqr() :632, get_data() :637 of the last equality problem */
    double *x_star;       /* nVar: get_xStar() :519                                                                                */
    int32_t *active_ctr;  /* total x 3: (obj_index, ctr_index, ctr_type) in working-set order, getActiveCtr_order() :703            */
    int32_t *log;         /* max_log x 5: (obj_index, ctr_index, ctr_type, cycling_detected, rank) per change, getWorkingSetLog()  */
    double *log_alpha;    /* max_log: alpha_or_lambda of the entry                                                                 */
    uint32_t max_log;
    double *x_mu, *x_mu_rhs, *residual_mu; /* REGULARIZATION_TIKHONOV_1 only (:617-630): nObjL x nVar (column k contiguous) twice, rows */
    uint32_t *counts;     /* 4: rows, nObjL, number of active constraints, number of log entries (entries beyond max_log are dropped) */
} lexls_lsi_debug;
int lexls_lsi_solve_debug(int device, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types, const double *h_data,
                          const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0, const double *h_v0,
                          const double *h_reg_factors, const double *h_params, uint32_t nparams, double *h_x, int32_t *h_info6, uint8_t *h_active,
                          double *h_v, const lexls_lsi_debug *debug);
/* the same on a hierarchy file in the reference's .dat format (tools.h:261-453); h_solution receives the file's
 * `#Solution` block when present (may be NULL).  one_based: simple-bound indices in the file are 1-based. */
int lexls_lsi_solve_dat(int device, const char *path, int one_based, int use_active_guess, int use_x_guess, double *h_x, int32_t *h_info6,
                        double *h_solution);

#ifdef __cplusplus
}
#endif
#endif /* LEXLS_HIP_H */
