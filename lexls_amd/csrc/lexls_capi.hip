// C ABI of liblexls_hip (include/lexls_hip.h): handle management, H2D/D2H plumbing and kernel
// dispatch.  There is deliberately NO CPU fallback: without a usable HIP device every entry point
// that needs one fails with LEXLS_ERR_NO_DEVICE / LEXLS_ERR_HIP.
#include "../../include/lexls_hip.h"
#include "lexls_kernels.h"
#include "lexls_launch.h"
#include "lexls_regularize.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace lexls;

namespace
{
    thread_local std::string g_err;

    int fail(int code, const std::string &msg)
    {
        g_err = msg;
        return code;
    }

#define HIP_TRY(expr)                                                                                                   \
    do                                                                                                                  \
    {                                                                                                                   \
        hipError_t e_ = (expr);                                                                                         \
        if (e_ != hipSuccess) return fail(LEXLS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

#define CHECK_HANDLE(h) \
    if (!(h)) return fail(LEXLS_ERR_INVALID, "null handle")
} // namespace

struct lexls_lse_s
{
    int device;
    hipStream_t stream;
    uint32_t batch, nVar, nObj, cap, max_rows, max_level_dim, min_level_dim;
    int force_generic;
    std::vector<uint32_t> maxdim, level_max;
    void *d_large_state;
    void *d_large_ws; // work space of the fast large path
    size_t large_ws_bytes;
    double *d_norms;
    double tol;
    bool dims_set, has_fixed, factor_valid, factor_in_hbm;
    uint64_t x_epoch, factor_epoch; // x_epoch == factor_epoch: d_x holds the basic solution of the current factor (lexls_lse_solve has nothing to do)
    const char *last_kernel;

    double *d_in_owned;
    bool deferred_sync;       // lexls_lse_set_deferred_sync: copies are enqueued, not waited for
    uint32_t *h_dims_pinned;
    hipEvent_t dims_event;
    bool dims_event_pending;
    double *d_cdata;          // resident constraint data of the batch (lexls_lse_set_constraint_data)
    uint64_t cdata_per_problem;
    uint32_t *d_row_src, *d_row_ld;
    const double *d_in;
    double *d_fac, *d_x, *d_hh, *d_v, *d_lambda, *d_maxabs, *d_scratch, *d_fixed_val;
    uint32_t *d_perm, *d_rank, *d_fcol, *d_totalrank, *d_dims, *d_nfixed, *d_fixed_idx;
    uint8_t *d_fixed_type, *d_ctr_type, *d_skip;
    bool has_skip;
    bool fused_gather = false; // this round's rows are read by reference inside lqr_wave_kernel (lexls_internal_round_resident)
    // prefix reuse (lexls_lse_set_prefix_reuse): the register-resident wave kernel leaves what a later factorization needs to read levels back
    int32_t *d_resume_level = nullptr; // batch
    uint8_t *d_resume_state = nullptr; // batch x resume_state_bytes(nObj)
    bool resume_enabled = false;
    bool resume_valid   = false; // the LAST factorization of this handle left that state (same kernel, factor kept, no regularization)
    bool resume_armed   = false; // d_resume_level holds the levels for the NEXT factorization
    int32_t *d_sens, *d_objidx;
    uint32_t reg_type;    // LexLS::RegularizationType, 0 = none
    uint32_t reg_cg_iters;
    double reg_variable;
    double *d_reg_factor, *d_reg_scratch, *d_reg_mu;
    bool sens_scan; // lexls_lse_set_sensitivity_scan
    char *d_round_in, *d_round_out; // the per-round arrays live in two slabs (lexls_lse_round_layout): one copy each way per round
    lexls_round_layout lay;

    LseArgs args() const
    {
        LseArgs a;
        a.batch = batch;
        a.nVar  = nVar;
        a.nObj  = nObj;
        a.cap   = cap;
        a.ldp   = odd_ld(max_rows);
        a.tol   = tol;
        a.in    = d_in;
        a.fac   = d_fac;
        a.x     = d_x;
        a.hh    = d_hh;
        a.perm  = d_perm;
        a.rank  = d_rank;
        a.fcol  = d_fcol;
        a.totalrank  = d_totalrank;
        a.dims       = d_dims;
        a.uniform_dim = (dims_set && min_level_dim == max_level_dim) ? max_level_dim : 0u;
        a.nfixed     = has_fixed ? d_nfixed : nullptr;
        a.fixed_idx  = d_fixed_idx;
        a.fixed_val  = d_fixed_val;
        a.fixed_type = d_fixed_type;
        a.ctr_type   = d_ctr_type;
        a.v          = d_v;
        a.lambda     = d_lambda;
        a.sens       = d_sens;
        a.maxabs     = d_maxabs;
        a.scratch    = d_scratch;
        a.skip       = has_skip ? d_skip : nullptr;
        a.reg_type     = reg_type;
        a.reg_cg_iters = reg_cg_iters;
        a.reg_variable = reg_variable;
        a.reg_factor   = d_reg_factor;
        a.reg_scratch  = d_reg_scratch;
        a.reg_mu       = d_reg_mu;
        a.g_cdata      = fused_gather ? d_cdata : nullptr;
        a.g_per        = cdata_per_problem;
        a.g_row_src    = d_row_src;
        a.g_row_ld     = d_row_ld;
        a.resume_state = resume_enabled ? d_resume_state : nullptr;
        a.resume_level = (resume_enabled && resume_armed && resume_valid) ? d_resume_level : nullptr;
        return a;
    }
    size_t problem_elems() const { return (size_t)cap * (nVar + 1); }
};

extern "C"
{
    const char *lexls_last_error(void) { return g_err.c_str(); }
    /* internal: lets the other translation units of the library report through lexls_last_error() */
    void lexls_internal_set_error(const char *msg) { g_err = msg ? msg : ""; }
    int lexls_version(void) { return 100; }

    int lexls_device_count(int *count)
    {
        int n        = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (count) *count = (e == hipSuccess) ? n : 0;
        if (e != hipSuccess || n == 0) return fail(LEXLS_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
        return LEXLS_OK;
    }

    int lexls_lse_create(lexls_lse_t *out, int device, uint32_t batch, uint32_t nVar, uint32_t nObj, const uint32_t *h_maxObjDim)
    {
        if (!out || !h_maxObjDim || batch == 0 || nVar == 0 || nObj == 0) return fail(LEXLS_ERR_INVALID, "lexls_lse_create: bad argument");
        int ndev = 0;
        if (lexls_device_count(&ndev) != LEXLS_OK) return LEXLS_ERR_NO_DEVICE;
        if (device < 0 || device >= ndev) return fail(LEXLS_ERR_INVALID, "lexls_lse_create: device index out of range");
        HIP_TRY(hipSetDevice(device));

        lexls_lse_s *h = new (std::nothrow) lexls_lse_s();
        if (!h) return fail(LEXLS_ERR_INVALID, "out of host memory");
        h->device = device;
        h->stream = nullptr;
        h->batch  = batch;
        h->nVar   = nVar;
        h->nObj   = nObj;
        h->maxdim.assign(h_maxObjDim, h_maxObjDim + nObj);
        h->cap = 0;
        for (uint32_t k = 0; k < nObj; k++) h->cap += h_maxObjDim[k];
        if (h->cap == 0)
        {
            delete h;
            return fail(LEXLS_ERR_INVALID, "lexls_lse_create: zero capacity");
        }
        h->max_rows    = h->cap;
        h->max_level_dim = 0;
        h->min_level_dim = 0;
        h->force_generic = 0;
        if (const char *e = std::getenv("LEXLS_KERNEL_POLICY")) h->force_generic = std::atoi(e); // diagnostic default of lexls_lse_set_kernel_policy
        h->tol         = 1e-12; // typedefs.h:120
        h->dims_set    = false;
        h->has_fixed   = false;
        h->factor_valid = h->factor_in_hbm = false;
        h->last_kernel = "";
        h->d_in_owned  = nullptr;
        h->d_cdata     = nullptr;
        h->reg_type    = 0;
        h->reg_cg_iters = 10; // typedefs.h:170
        h->reg_variable = 0.0;
        h->d_reg_factor = h->d_reg_scratch = h->d_reg_mu = nullptr;
        h->deferred_sync = false;
        h->h_dims_pinned = nullptr;
        h->dims_event = nullptr;
        h->dims_event_pending = false;
        h->cdata_per_problem = 0;
        h->d_round_in = h->d_round_out = nullptr;
        h->d_in        = nullptr;
        h->d_scratch   = nullptr;

        const size_t B = batch, n = nVar, cap = h->cap;
        hipError_t e   = hipSuccess;
        auto alloc     = [&](void **p, size_t bytes) {
            if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 8);
        };
        // the small per-round arrays are carved out of two slabs so that a lock-step driver moves each with ONE copy
        {
            auto up = [](uint64_t v) { return (v + 255) & ~uint64_t(255); };
            lexls_round_layout &L = h->lay;
            uint64_t o   = 0;
            L.dims       = o, o = up(o + 4 * B * nObj);
            L.nfixed     = o, o = up(o + 4 * B);
            L.fixed_idx  = o, o = up(o + 4 * B * n);
            L.fixed_val  = o, o = up(o + 8 * B * n);
            L.skip       = o, o = up(o + B);
            L.obj_index  = o, o = up(o + 4 * B);
            L.row_src    = o, o = up(o + 4 * B * cap);
            L.row_ld     = o, o = up(o + 4 * B * cap);
            L.fixed_type = o, o = up(o + B * n);
            L.ctr_type   = o, o = up(o + B * cap);
            L.in_bytes   = o;
            o            = 0;
            L.x          = o, o = up(o + 8 * B * n);
            L.total_rank = o, o = up(o + 4 * B);
            L.found      = o, o = up(o + 4 * B * 3);
            L.max_abs    = o, o = up(o + 8 * B);
            L.out_bytes  = o;
            alloc((void **)&h->d_round_in, L.in_bytes);
            alloc((void **)&h->d_round_out, L.out_bytes);
            if (e == hipSuccess) e = hipMemset(h->d_round_in, 0, L.in_bytes);
            if (e == hipSuccess) e = hipMemset(h->d_round_out, 0, L.out_bytes);
            if (e == hipSuccess)
            {
                char *in = h->d_round_in, *out_ = h->d_round_out;
                h->d_dims       = (uint32_t *)(in + L.dims);
                h->d_nfixed     = (uint32_t *)(in + L.nfixed);
                h->d_fixed_idx  = (uint32_t *)(in + L.fixed_idx);
                h->d_fixed_val  = (double *)(in + L.fixed_val);
                h->d_skip       = (uint8_t *)(in + L.skip);
                h->d_objidx     = (int32_t *)(in + L.obj_index);
                h->d_row_src    = (uint32_t *)(in + L.row_src);
                h->d_row_ld     = (uint32_t *)(in + L.row_ld);
                h->d_fixed_type = (uint8_t *)(in + L.fixed_type);
                h->d_ctr_type   = (uint8_t *)(in + L.ctr_type);
                h->d_x          = (double *)(out_ + L.x);
                h->d_totalrank  = (uint32_t *)(out_ + L.total_rank);
                h->d_sens       = (int32_t *)(out_ + L.found);
                h->d_maxabs     = (double *)(out_ + L.max_abs);
            }
        }
        alloc((void **)&h->d_fac, 8 * B * h->problem_elems());
        alloc((void **)&h->d_hh, 8 * B * cap);
        alloc((void **)&h->d_v, 8 * B * cap);
        alloc((void **)&h->d_lambda, 8 * B * (n + cap));
        alloc((void **)&h->d_perm, 4 * B * n);
        alloc((void **)&h->d_rank, 4 * B * nObj);
        alloc((void **)&h->d_fcol, 4 * B * nObj);
        if (e != hipSuccess)
        {
            lexls_lse_destroy(h);
            return fail(LEXLS_ERR_HIP, std::string("lexls_lse_create: ") + hipGetErrorString(e));
        }
        // default dims = capacities (LexLSE(nVar,nObj,ObjDim) constructor semantics, lexlse.h:50-54)
        *out = h;
        return lexls_lse_set_obj_dim(h, h_maxObjDim, 0);
    }

    int lexls_lse_destroy(lexls_lse_t h)
    {
        if (!h) return LEXLS_OK;
        (void)hipSetDevice(h->device);
        void *ptrs[] = {h->d_in_owned, h->d_fac, h->d_hh, h->d_v, h->d_lambda, h->d_scratch, h->d_perm, h->d_rank, h->d_fcol, h->d_round_in, h->d_round_out,
                        h->d_large_state, h->d_large_ws, h->d_norms, h->d_cdata, h->d_reg_factor, h->d_reg_scratch, h->d_reg_mu, h->d_resume_level, h->d_resume_state};
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
        if (h->h_dims_pinned) (void)hipHostFree(h->h_dims_pinned);
        if (h->dims_event) (void)hipEventDestroy(h->dims_event);
        delete h;
        return LEXLS_OK;
    }

    int lexls_lse_set_stream(lexls_lse_t h, void *hip_stream)
    {
        CHECK_HANDLE(h);
        h->stream = static_cast<hipStream_t>(hip_stream);
        return LEXLS_OK;
    }

    int lexls_lse_synchronize(lexls_lse_t h)
    {
        CHECK_HANDLE(h);
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipStreamSynchronize(h->stream));
        return LEXLS_OK;
    }

    /// A round whose rows are read by reference inside the register-resident wave kernel (fused_gather) leaves `in` unassembled.  Anything that
    /// can send the next factorization to ANOTHER kernel (kernel policy, regularization type) assembles the rows first.
    static hipError_t materialize_fused_gather(lexls_lse_t h)
    {
        if (!h->fused_gather) return hipSuccess;
        hipError_t e = hipSetDevice(h->device);
        if (e != hipSuccess) return e;
        h->fused_gather = false;
        LseArgs a       = h->args();
        e = launch_gather_rows(a, h->d_cdata, h->cdata_per_problem, h->d_row_src, h->d_row_ld, h->d_in_owned, h->stream);
        h->d_in = h->d_in_owned;
        return e;
    }

    int lexls_lse_set_regularization(lexls_lse_t h, int type, const double *h_factors, int per_problem, double variable_factor)
    {
        CHECK_HANDLE(h);
        if ((uint32_t)type != h->reg_type) HIP_TRY(materialize_fused_gather(h));
        switch (type)
        {
        case 0: case 1: case 2: case 3: case 4: case 5: case 6: case 7: case 8: case 9: break;
        default: return fail(LEXLS_ERR_INVALID, "set_regularization: unknown regularization type");
        }
        HIP_TRY(hipSetDevice(h->device));
        h->factor_valid = false;
        h->reg_type     = (uint32_t)type;
        h->reg_variable = variable_factor;
        if (type == 0) return LEXLS_OK;
        const size_t B = h->batch, nObj = h->nObj;
        std::vector<double> f(B * nObj, 0.0);
        if (h_factors)
            for (size_t b = 0; b < B; b++)
                for (size_t k = 0; k < nObj; k++) f[b * nObj + k] = per_problem ? h_factors[b * nObj + k] : h_factors[k];
        if (!h->d_reg_factor) HIP_TRY(hipMalloc((void **)&h->d_reg_factor, 8 * B * nObj));
        if (!h->d_reg_scratch)
        {
            const size_t bytes = 8 * B * reg_scratch_doubles(h->nVar);
            HIP_TRY(hipMalloc((void **)&h->d_reg_scratch, bytes));
            HIP_TRY(hipMemsetAsync(h->d_reg_scratch, 0, bytes, h->stream));
        }
        if (type == 7 && !h->d_reg_mu) // X_mu, X_mu_rhs, residual_mu of the reference's experimental type (lexlse.h:96-99)
        {
            const size_t bytes = 8 * B * reg_mu_doubles(h->nVar, h->nObj, h->cap);
            HIP_TRY(hipMalloc((void **)&h->d_reg_mu, bytes));
            HIP_TRY(hipMemsetAsync(h->d_reg_mu, 0, bytes, h->stream));
        }
        HIP_TRY(hipMemcpyAsync(h->d_reg_factor, f.data(), 8 * B * nObj, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream)); // f is a temporary
        return LEXLS_OK;
    }

    int lexls_lse_set_cg_iterations(lexls_lse_t h, uint32_t max_iterations)
    {
        CHECK_HANDLE(h);
        h->reg_cg_iters = max_iterations;
        h->factor_valid = false;
        return LEXLS_OK;
    }

    int lexls_lse_set_deferred_sync(lexls_lse_t h, int on)
    {
        CHECK_HANDLE(h);
        if (!on && h->deferred_sync) // leaving the mode: everything enqueued so far completes first
        {
            HIP_TRY(hipSetDevice(h->device));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        h->deferred_sync = on != 0;
        return LEXLS_OK;
    }

    int lexls_lse_set_tolerance(lexls_lse_t h, double tol)
    {
        CHECK_HANDLE(h);
        if (tol != h->tol) h->factor_valid = false; // a factor / solution computed with the old tolerance must not be served any more
        h->tol = tol;
        return LEXLS_OK;
    }

    int lexls_lse_set_obj_dim(lexls_lse_t h, const uint32_t *h_dims, int per_problem)
    {
        CHECK_HANDLE(h);
        if (!h_dims) return fail(LEXLS_ERR_INVALID, "set_obj_dim: null dims");
        const size_t nd = (size_t)h->batch * h->nObj;
        std::vector<uint32_t> d_tmp;
        uint32_t *d = nullptr;
        if (h->deferred_sync) // the copy below is not waited for: its source must outlive this call and be DMA-able
        {
            HIP_TRY(hipSetDevice(h->device));
            if (!h->h_dims_pinned) HIP_TRY(hipHostMalloc((void **)&h->h_dims_pinned, 4 * nd, hipHostMallocDefault));
            else if (h->dims_event_pending) HIP_TRY(hipEventSynchronize(h->dims_event)); // the previous copy out of this buffer is done
            d = h->h_dims_pinned;
        }
        else
        {
            d_tmp.resize(nd);
            d = d_tmp.data();
        }
        uint32_t max_rows = 0, max_level = 0, min_level = 0xffffffffu;
        h->level_max.assign(h->nObj, 0);
        for (uint32_t b = 0; b < h->batch; b++)
        {
            uint32_t m = 0;
            for (uint32_t k = 0; k < h->nObj; k++)
            {
                const uint32_t v = per_problem ? h_dims[(size_t)b * h->nObj + k] : h_dims[k];
                if (v > h->maxdim[k]) return fail(LEXLS_ERR_INVALID, "set_obj_dim: dimension exceeds the capacity given at creation");
                d[(size_t)b * h->nObj + k] = v;
                m += v;
                if (v > max_level) max_level = v;
                if (v < min_level) min_level = v;
                if (v > h->level_max[k]) h->level_max[k] = v;
            }
            if (m > max_rows) max_rows = m;
        }
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMemcpyAsync(h->d_dims, d, 4 * nd, hipMemcpyHostToDevice, h->stream));
        if (h->deferred_sync)
        {
            if (!h->dims_event) HIP_TRY(hipEventCreateWithFlags(&h->dims_event, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(h->dims_event, h->stream));
            h->dims_event_pending = true;
        }
        else
            HIP_TRY(hipStreamSynchronize(h->stream)); // d is a temporary
        h->max_rows      = max_rows ? max_rows : 1;
        h->max_level_dim = max_level;
        h->min_level_dim = min_level;
        h->dims_set     = true;
        h->factor_valid = false;
        return LEXLS_OK;
    }

    int lexls_lse_set_fixed(lexls_lse_t h, const uint32_t *h_nfixed, const uint32_t *h_index, const double *h_value, const uint8_t *h_type)
    {
        CHECK_HANDLE(h);
        HIP_TRY(hipSetDevice(h->device));
        h->factor_valid = false;
        if (!h_nfixed)
        {
            h->has_fixed = false;
            return LEXLS_OK;
        }
        if (!h_index || !h_value) return fail(LEXLS_ERR_INVALID, "set_fixed: null index/value");
        bool any = false;
        for (uint32_t b = 0; b < h->batch; b++)
        {
            if (h_nfixed[b] > h->nVar) return fail(LEXLS_ERR_INVALID, "Cannot fix more than nVar variables"); // lexlse.h:1453
            for (uint32_t k = 0; k < h_nfixed[b]; k++)
                if (h_index[(size_t)b * h->nVar + k] >= h->nVar) return fail(LEXLS_ERR_INVALID, "set_fixed: variable index out of range");
            any = any || h_nfixed[b] > 0;
        }
        const size_t B = h->batch, n = h->nVar;
        HIP_TRY(hipMemcpyAsync(h->d_nfixed, h_nfixed, 4 * B, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->d_fixed_idx, h_index, 4 * B * n, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->d_fixed_val, h_value, 8 * B * n, hipMemcpyHostToDevice, h->stream));
        if (h_type)
            HIP_TRY(hipMemcpyAsync(h->d_fixed_type, h_type, B * n, hipMemcpyHostToDevice, h->stream));
        else
            HIP_TRY(hipMemsetAsync(h->d_fixed_type, CTR_ACTIVE_UB, B * n, h->stream)); // default of fixVariable, lexlse.h:1381
        if (!h->deferred_sync) HIP_TRY(hipStreamSynchronize(h->stream));
        h->has_fixed = any;
        return LEXLS_OK;
    }

    int lexls_lse_set_fixed_type(lexls_lse_t h, const uint8_t *h_type)
    {
        CHECK_HANDLE(h);
        if (!h_type) return fail(LEXLS_ERR_INVALID, "set_fixed_type: null");
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMemcpyAsync(h->d_fixed_type, h_type, (size_t)h->batch * h->nVar, hipMemcpyHostToDevice, h->stream));
        if (!h->deferred_sync) HIP_TRY(hipStreamSynchronize(h->stream));
        return LEXLS_OK; // activation types only matter to the dual solve: the factorization stays valid
    }

    int lexls_lse_set_ctr_type(lexls_lse_t h, const uint8_t *h_types)
    {
        CHECK_HANDLE(h);
        if (!h_types) return fail(LEXLS_ERR_INVALID, "set_ctr_type: null");
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMemcpyAsync(h->d_ctr_type, h_types, (size_t)h->batch * h->cap, hipMemcpyHostToDevice, h->stream));
        if (!h->deferred_sync) HIP_TRY(hipStreamSynchronize(h->stream));
        return LEXLS_OK;
    }

    int lexls_lse_set_skip(lexls_lse_t h, const uint8_t *h_skip)
    {
        CHECK_HANDLE(h);
        if (!h_skip)
        {
            h->has_skip = false;
            return LEXLS_OK;
        }
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMemcpyAsync(h->d_skip, h_skip, (size_t)h->batch, hipMemcpyHostToDevice, h->stream));
        if (!h->deferred_sync) HIP_TRY(hipStreamSynchronize(h->stream));
        h->has_skip = true;
        return LEXLS_OK;
    }

    int lexls_lse_set_problem_host(lexls_lse_t h, const double *h_lod)
    {
        CHECK_HANDLE(h);
        if (!h_lod) return fail(LEXLS_ERR_INVALID, "set_problem_host: null");
        HIP_TRY(hipSetDevice(h->device));
        const size_t bytes = 8 * (size_t)h->batch * h->problem_elems();
        if (!h->d_in_owned) HIP_TRY(hipMalloc((void **)&h->d_in_owned, bytes));
        HIP_TRY(hipMemcpyAsync(h->d_in_owned, h_lod, bytes, hipMemcpyHostToDevice, h->stream));
        if (!h->deferred_sync) HIP_TRY(hipStreamSynchronize(h->stream));
        h->d_in         = h->d_in_owned;
        h->fused_gather = false;
        h->factor_valid = false;
        return LEXLS_OK;
    }

    int lexls_lse_set_constraint_data(lexls_lse_t h, const double *h_data, uint64_t per_problem)
    {
        CHECK_HANDLE(h);
        if (!h_data || per_problem == 0) return fail(LEXLS_ERR_INVALID, "set_constraint_data: null / empty");
        HIP_TRY(hipSetDevice(h->device));
        const size_t bytes = 8 * (size_t)h->batch * per_problem;
        if (h->d_cdata && h->cdata_per_problem != per_problem)
        {
            HIP_TRY(hipFree(h->d_cdata));
            h->d_cdata = nullptr;
        }
        if (!h->d_cdata) HIP_TRY(hipMalloc((void **)&h->d_cdata, bytes));
        h->cdata_per_problem = per_problem;
        HIP_TRY(hipMemcpyAsync(h->d_cdata, h_data, bytes, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        return LEXLS_OK;
    }

    int lexls_lse_gather_problem(lexls_lse_t h, const uint32_t *h_row_src, const uint32_t *h_row_ld)
    {
        CHECK_HANDLE(h);
        if (!h_row_src || !h_row_ld) return fail(LEXLS_ERR_INVALID, "gather_problem: null");
        if (!h->d_cdata) return fail(LEXLS_ERR_INVALID, "gather_problem: call lexls_lse_set_constraint_data first");
        const size_t B = h->batch, cap = h->cap;
        // every element a row refers to must lie inside the problem's resident block (the kernel does not check)
        for (size_t i = 0; i < B * cap; i++)
        {
            const uint64_t ld = h_row_ld[i] & 0x7fffffffu;
            if (ld && (uint64_t)h_row_src[i] + (uint64_t)(h->nVar + 1) * ld >= h->cdata_per_problem)
                return fail(LEXLS_ERR_INVALID, "gather_problem: row reference outside the constraint data");
        }
        HIP_TRY(hipSetDevice(h->device));
        if (!h->d_in_owned)
        {
            HIP_TRY(hipMalloc((void **)&h->d_in_owned, 8 * B * h->problem_elems()));
            HIP_TRY(hipMemsetAsync(h->d_in_owned, 0, 8 * B * h->problem_elems(), h->stream));
        }
        HIP_TRY(hipMemcpyAsync(h->d_row_src, h_row_src, 4 * B * cap, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->d_row_ld, h_row_ld, 4 * B * cap, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(launch_gather_rows(h->args(), h->d_cdata, h->cdata_per_problem, h->d_row_src, h->d_row_ld, h->d_in_owned, h->stream));
        if (!h->deferred_sync) HIP_TRY(hipStreamSynchronize(h->stream)); // the host arrays may be reused by the caller
        h->d_in         = h->d_in_owned;
        h->fused_gather = false;
        h->factor_valid = false;
        return LEXLS_OK;
    }

    int lexls_lse_round_layout(lexls_lse_t h, lexls_round_layout *out)
    {
        CHECK_HANDLE(h);
        if (!out) return fail(LEXLS_ERR_INVALID, "round_layout: null");
        *out = h->lay;
        return LEXLS_OK;
    }

    /* internal: the resident constraint data (lexls_lse_set_constraint_data), read by the lock-step driver's step kernel */
    const double *lexls_internal_cdata(lexls_lse_t h) { return h ? h->d_cdata : nullptr; }

    /* internal: the device copy of the in slab (lexls_lse_round_layout) — the lock-step driver's resident iterations write the next
     * equality problem's dimensions, fixed variables, types and row references there themselves */
    char *lexls_internal_round_in(lexls_lse_t h) { return h ? h->d_round_in : nullptr; }
    /* internal: the in slab was written ON THE DEVICE (same stream): gather the rows it names.  The host does not know this round's
     * dimensions, so kernel choice and LDS budgets follow the capacities given at creation (every kernel takes smaller problems). */
    int lexls_internal_round_resident(lexls_lse_t h, int has_fixed)
    {
        CHECK_HANDLE(h);
        if (!h->d_cdata || !h->d_in_owned) return fail(LEXLS_ERR_INVALID, "round_resident: needs resident constraint data and one uploaded round");
        HIP_TRY(hipSetDevice(h->device));
        uint32_t max_level = 0;
        h->level_max.assign(h->maxdim.begin(), h->maxdim.end());
        for (uint32_t v : h->maxdim) max_level = v > max_level ? v : max_level;
        h->max_rows      = h->cap ? h->cap : 1;
        h->max_level_dim = max_level;
        h->min_level_dim = 0; // this round's dimensions are known to the device only
        h->dims_set      = true;
        h->has_fixed     = has_fixed != 0;
        h->has_skip      = true;
        h->factor_valid  = false;
        h->fused_gather  = false;
        h->d_in          = h->d_in_owned;
        // the register-resident wave kernel reads the rows by reference itself (one launch and one pass over the problems less)
        // (a forced left-looking / four-per-wavefront policy is honoured; otherwise the register-resident kernel at every batch size: on the
        // ragged problems of a lock-step LSI stage, with the gather fused, it beats the four-per-wavefront kernel + gather launch also beyond
        // one round — 4096 instances, warm-started ~30 iterations: 37.1 ms vs 39.1 ms)
        // 42..48 columns are the exception: their register-resident form is the 64-column instantiation, and the four-per-wavefront kernel
        // behind a gather launch is ahead of it at every batch size (1024 instances, n = 47, 5 x 12, cold: 21.3 -> 17.1 ms)
        const bool wide_slot = h->nVar + 1 > 41 && h->nVar + 1 <= 48 && h->max_level_dim <= 12;
        const int ll = h->force_generic == 3 ? 1 : (h->force_generic == 4 ? 2 : (wide_slot && h->force_generic == 0 ? 0 : -1));
        if (h->force_generic != 1 && h->reg_type == 0 && wave_kernel_supports(h->args(), h->max_rows, h->max_level_dim, h->has_fixed) &&
            wave_dispatch_is_register_resident(h->args(), h->max_level_dim, h->has_fixed, ll))
            h->fused_gather = true;
        else
            HIP_TRY(launch_gather_rows(h->args(), h->d_cdata, h->cdata_per_problem, h->d_row_src, h->d_row_ld, h->d_in_owned, h->stream));
        return LEXLS_OK;
    }

    /* internal (the lock-step LexLSI driver): ALL remaining resident iterations in one persistent launch — per instance l-QR (rows gathered by
     * reference, levels above the changed one read back) -> removal sweep -> lsi_iterate_body, until the instance stops (at most `count`
     * iterations).  Same preparation and the same bookkeeping as lexls_internal_round_resident + lexls_internal_arm_resume +
     * lexls_lse_factorize_solve(h, 1) + lexls_lse_sensitivity_resident per stage.  Returns 1 (nothing done, nothing changed) when the shape has no
     * persistent instantiation: the driver then enqueues the stage's three kernels as before. */
    int lexls_internal_resident_fused(lexls_lse_t h, int has_fixed, int count, double tolW, double tolC, const void *resident_args, size_t resident_args_bytes)
    {
        CHECK_HANDLE(h);
        if (!h->d_cdata || !h->d_in_owned) return fail(LEXLS_ERR_INVALID, "resident_fused: needs resident constraint data and one uploaded round");
        if (h->force_generic != 0 || h->reg_type != 0 || std::getenv("LEXLS_LSI_NO_FUSED")) return 1;
        HIP_TRY(hipSetDevice(h->device));
        // what lexls_internal_round_resident would set — on a copy of the fields first: nothing changes when the launch is not taken
        uint32_t max_level = 0;
        for (uint32_t v : h->maxdim) max_level = v > max_level ? v : max_level;
        if (h->nVar + 1 > 41 && h->nVar + 1 <= 48 && max_level <= 12) return 1; // (42..48 columns: the four-per-wavefront kernel behind a gather launch, as lexls_internal_round_resident decides)
        {
            lexls_lse_s probe    = *h; // (plain fields and pointers; the vectors are copied, the probe owns nothing it frees)
            probe.max_rows       = h->cap ? h->cap : 1;
            probe.max_level_dim  = max_level;
            probe.min_level_dim  = 0;
            probe.dims_set       = true;
            probe.has_fixed      = has_fixed != 0;
            probe.has_skip       = true;
            probe.fused_gather   = true;
            probe.d_in           = h->d_in_owned;
            probe.resume_armed   = h->resume_enabled;
            const LseArgs pa     = probe.args();
            const char *variant  = "";
            if (!wave_kernel_supports(pa, probe.max_rows, max_level, probe.has_fixed)) return 1;
            const hipError_t e = launch_lsi_fused(pa, max_level, probe.has_fixed, h->d_objidx, tolW, tolC, h->sens_scan, resident_args, resident_args_bytes, count, h->stream, &variant);
            if (e == hipErrorNotSupported) return 1;
            HIP_TRY(e);
            h->last_kernel = variant;
        }
        h->level_max.assign(h->maxdim.begin(), h->maxdim.end());
        h->max_rows      = h->cap ? h->cap : 1;
        h->max_level_dim = max_level;
        h->min_level_dim = 0;
        h->dims_set      = true;
        h->has_fixed     = has_fixed != 0;
        h->has_skip      = true;
        h->fused_gather  = true;
        h->d_in          = h->d_in_owned;
        h->resume_armed  = false;
        h->resume_valid  = h->resume_enabled && h->d_resume_state != nullptr;
        h->factor_valid  = true;
        h->factor_epoch++;
        h->x_epoch       = h->factor_epoch;
        h->factor_in_hbm = true;
        return LEXLS_OK;
    }

    static int upload_round(lexls_lse_t h, const void *h_in, int gather, bool trusted);
    int lexls_lse_upload_round(lexls_lse_t h, const void *h_in, int gather) { return upload_round(h, h_in, gather, false); }
    /* internal (not in include/lexls_hip.h): the lock-step LexLSI driver of this library fills the block itself — variable indices and
     * row references come from its own working sets — and skips the per-element argument checks */
    int lexls_internal_upload_round_trusted(lexls_lse_t h, const void *h_in, int gather) { return upload_round(h, h_in, gather, true); }

    static int upload_round(lexls_lse_t h, const void *h_in, int gather, bool trusted)
    {
        CHECK_HANDLE(h);
        if (!h_in) return fail(LEXLS_ERR_INVALID, "upload_round: null");
        const lexls_round_layout &L = h->lay;
        const char *in           = static_cast<const char *>(h_in);
        const uint32_t *dims     = reinterpret_cast<const uint32_t *>(in + L.dims);
        const uint32_t *nfixed   = reinterpret_cast<const uint32_t *>(in + L.nfixed);
        const uint32_t *fixedidx = reinterpret_cast<const uint32_t *>(in + L.fixed_idx);
        const uint8_t *skip      = reinterpret_cast<const uint8_t *>(in + L.skip);
        const uint32_t *row_src  = reinterpret_cast<const uint32_t *>(in + L.row_src);
        const uint32_t *row_ld   = reinterpret_cast<const uint32_t *>(in + L.row_ld);
        // the host-side checks and bookkeeping of set_obj_dim / set_fixed / gather_problem (skipped problems included: a later
        // sensitivity call may still serve them, and the kernels' LDS budget follows the largest problem of the batch)
        uint32_t max_rows = 0, max_level = 0, min_level = 0xffffffffu;
        bool any_fixed = false;
        h->level_max.assign(h->nObj, 0);
        if (gather && !h->d_cdata) return fail(LEXLS_ERR_INVALID, "upload_round: call lexls_lse_set_constraint_data first");
        for (uint32_t b = 0; b < h->batch; b++)
        {
            uint32_t m = 0;
            for (uint32_t k = 0; k < h->nObj; k++)
            {
                const uint32_t v = dims[(size_t)b * h->nObj + k];
                if (v > h->maxdim[k]) return fail(LEXLS_ERR_INVALID, "upload_round: dimension exceeds the capacity given at creation");
                m += v;
                if (v > max_level) max_level = v;
                if (v < min_level) min_level = v;
                if (v > h->level_max[k]) h->level_max[k] = v;
            }
            if (m > max_rows) max_rows = m;
            if (nfixed[b] > h->nVar) return fail(LEXLS_ERR_INVALID, "Cannot fix more than nVar variables"); // lexlse.h:1453
            any_fixed = any_fixed || nfixed[b] > 0;
            if (trusted) continue;
            for (uint32_t k = 0; k < nfixed[b]; k++)
                if (fixedidx[(size_t)b * h->nVar + k] >= h->nVar) return fail(LEXLS_ERR_INVALID, "upload_round: variable index out of range");
            if (gather && !skip[b])
                for (size_t i = (size_t)b * h->cap; i < (size_t)(b + 1) * h->cap; i++)
                {
                    const uint64_t ld = row_ld[i] & 0x7fffffffu;
                    if (ld && (uint64_t)row_src[i] + (uint64_t)(h->nVar + 1) * ld >= h->cdata_per_problem)
                        return fail(LEXLS_ERR_INVALID, "upload_round: row reference outside the constraint data");
                }
        }
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMemcpyAsync(h->d_round_in, h_in, L.in_bytes, hipMemcpyHostToDevice, h->stream));
        h->fused_gather  = false;
        h->max_rows      = max_rows ? max_rows : 1;
        h->max_level_dim = max_level;
        h->min_level_dim = min_level;
        h->dims_set      = true;
        h->has_fixed     = any_fixed;
        h->has_skip      = false; // kernels look at the mask only when some problem is masked (the one-launch-per-level large kernel needs "none")
        for (uint32_t b = 0; b < h->batch && !h->has_skip; b++) h->has_skip = skip[b] != 0;
        h->factor_valid  = false;
        if (gather)
        {
            if (!h->d_in_owned)
            {
                HIP_TRY(hipMalloc((void **)&h->d_in_owned, 8 * (size_t)h->batch * h->problem_elems()));
                HIP_TRY(hipMemsetAsync(h->d_in_owned, 0, 8 * (size_t)h->batch * h->problem_elems(), h->stream));
            }
            HIP_TRY(launch_gather_rows(h->args(), h->d_cdata, h->cdata_per_problem, h->d_row_src, h->d_row_ld, h->d_in_owned, h->stream));
            h->d_in = h->d_in_owned;
        }
        if (!h->deferred_sync) HIP_TRY(hipStreamSynchronize(h->stream));
        return LEXLS_OK;
    }

    int lexls_lse_download_round(lexls_lse_t h, void *h_out, void *h_types)
    {
        CHECK_HANDLE(h);
        HIP_TRY(hipSetDevice(h->device));
        const lexls_round_layout &L = h->lay;
        if (h_out) HIP_TRY(hipMemcpyAsync(h_out, h->d_round_out, L.out_bytes, hipMemcpyDeviceToHost, h->stream));
        if (h_types) HIP_TRY(hipMemcpyAsync(h_types, h->d_round_in + L.fixed_type, L.in_bytes - L.fixed_type, hipMemcpyDeviceToHost, h->stream));
        if (!h->deferred_sync) HIP_TRY(hipStreamSynchronize(h->stream));
        return LEXLS_OK;
    }

    int lexls_lse_set_problem_device(lexls_lse_t h, const double *d_lod)
    {
        CHECK_HANDLE(h);
        if (!d_lod) return fail(LEXLS_ERR_INVALID, "set_problem_device: null");
        h->d_in         = d_lod;
        h->fused_gather = false;
        h->factor_valid = false;
        return LEXLS_OK;
    }

    /// Which tolerance-contract kernels a factorization of this handle may take (launch_lqr_wave's `tolerance`): policies 6 / 7 / 8 name one;
    /// automatic dispatch (policy 0) takes them wherever they serve unless the process runs under LEXLS_QTOL=0 (read at every call, so that a
    /// caller may change it between solves); every other policy stays on the bit-exact kernels
    static int tolerance_mode(lexls_lse_t h)
    {
        if (h->fused_gather) return 0;
        if (h->force_generic >= 6 && h->force_generic <= 9) return h->force_generic;
        if (h->force_generic != 0) return 0;
        const char *e = std::getenv("LEXLS_QTOL");
        return (e && std::atoi(e) == 0) ? 0 : 1;
    }

    /// opportunistic_solve: the caller only asked for the factor; kernels that produce x on the way at no extra launch do (the wave kernels
    /// always, the generic kernel on request), so that a later lexls_lse_solve of the same factor has nothing left to do
    static int run_lqr(lexls_lse_t h, bool write_factor, bool do_solve, bool opportunistic_solve = false)
    {
        bool solved = true;
        CHECK_HANDLE(h);
        if (!h->d_in) return fail(LEXLS_ERR_INVALID, "no problem data: call lexls_lse_set_problem_host/device first");
        HIP_TRY(hipSetDevice(h->device));
        const char *variant = "";
        const LseArgs a     = h->args();
        const bool shape_kernels = h->force_generic != 1;
        // (the regularization family lives in the register-resident wave kernel's REG instantiations and in the generic kernel)
        if (shape_kernels && wave_kernel_supports(a, h->max_rows, h->max_level_dim, h->has_fixed))
        {
            const int ll = h->fused_gather ? -1 : (h->force_generic == 2 ? -1 : (h->force_generic == 3 ? 1 : (h->force_generic == 4 ? 2 : 0)));
            // the tolerance-contract kernel (lqr_qtol_impl.h): automatic dispatch and policy 6; LEXLS_QTOL=0 keeps every solve bit-exact
            HIP_TRY(launch_lqr_wave(a, h->max_level_dim, write_factor, h->has_fixed, ll, h->stream, &variant, tolerance_mode(h))); // always solves as well
        }
        else if (shape_kernels && h->force_generic != 2 && h->max_rows > 64 && h->max_level_dim <= 16 && a.nObj <= 16 &&
                 deep_kernel_supports(a, h->max_level_dim, write_factor, h->has_fixed))
        {
            // deep hierarchies (more than 64 rows in all): the left-looking kernels, whose LDS holds pivot rows only (x-only solves of the IK
            // shape: the tolerance-contract kernel under the same rules as above — it reads a level's rows when the level starts, too)
            HIP_TRY(launch_lqr_wave(a, h->max_level_dim, write_factor, h->has_fixed, 2, h->stream, &variant, tolerance_mode(h)));
        }
        else if (shape_kernels && h->reg_type == 0 && !generic_fits_lds(a, h->max_rows) && large_kernel_supports(a, h->max_level_dim, h->has_fixed))
        {
            if (h->force_generic == 5) // the bit-exact multi-launch path (ordered chains: parity tests, reference for the fast path)
            {
                if (!h->d_large_state) HIP_TRY(hipMalloc(&h->d_large_state, large_state_bytes(h->batch)));
                if (!h->d_norms) HIP_TRY(hipMalloc((void **)&h->d_norms, 8 * (size_t)h->batch * h->nVar));
                HIP_TRY(launch_lqr_large(a, h->level_max.data(), h->max_rows, h->d_large_state, h->d_norms, h->stream));
                variant = "lqr_large<multi-launch>";
            }
            else
            {
                uint32_t md = 0;
                for (uint32_t v : h->level_max) md = v > md ? v : md;
                const size_t need = large_fast_workspace_bytes(h->batch, h->nVar, h->cap, md);
                if (need > h->large_ws_bytes)
                {
                    if (h->d_large_ws) HIP_TRY(hipFree(h->d_large_ws));
                    h->d_large_ws = nullptr;
                    HIP_TRY(hipMalloc(&h->d_large_ws, need));
                    h->large_ws_bytes = need;
                }
                HIP_TRY(launch_lqr_large_fast(a, h->level_max.data(), h->max_rows, h->d_large_ws, h->stream));
                variant = "lqr_large<step-per-pivot,mfma>";
            }
            if (do_solve) HIP_TRY(launch_solve_generic(a, h->stream, h->force_generic != 5)); // (the step-per-pivot path's contract allows reciprocals)
            solved       = do_solve;
            write_factor = true;
        }
        else
        {
            solved = do_solve || opportunistic_solve;
            HIP_TRY(launch_lqr_generic(a, h->max_rows, write_factor, solved, h->stream, &variant));
        }
        // prefix reuse: the levels handed over are consumed; the state is there for the next factorization iff this one was the register-resident
        // wave kernel keeping its factor (every other kernel ignores both pointers and factorizes everything)
        h->resume_armed = false;
        h->resume_valid = a.resume_state != nullptr && write_factor && h->reg_type == 0 && std::strncmp(variant, "lqr_wave<", 9) == 0;
        h->last_kernel   = variant;
        h->factor_valid  = true;
        h->factor_epoch++;
        if (solved) h->x_epoch = h->factor_epoch;
        h->factor_in_hbm = write_factor || std::strstr(variant, "hbm") != nullptr;
        return LEXLS_OK;
    }

    int lexls_lse_factorize(lexls_lse_t h) { return run_lqr(h, true, false, true); }
    int lexls_lse_factorize_solve(lexls_lse_t h, int keep_factor) { return run_lqr(h, keep_factor != 0, true); }

    static int need_factor(lexls_lse_t h, const char *who)
    {
        CHECK_HANDLE(h);
        if (!h->factor_valid || !h->factor_in_hbm) return fail(LEXLS_ERR_INVALID, std::string(who) + ": needs a factorization whose factor was kept");
        return LEXLS_OK;
    }

    int lexls_lse_solve(lexls_lse_t h)
    {
        if (int rc = need_factor(h, "lexls_lse_solve")) return rc;
        if (h->x_epoch == h->factor_epoch) return LEXLS_OK; // the factorization kernel left the basic solution in place
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(launch_solve_generic(h->args(), h->stream, std::strstr(h->last_kernel, "step-per-pivot") != nullptr)); // (same x as that path's factorize_solve)
        h->x_epoch = h->factor_epoch;
        return LEXLS_OK;
    }

    int lexls_lse_solve_least_norm(lexls_lse_t h)
    {
        if (int rc = need_factor(h, "lexls_lse_solve_least_norm")) return rc;
        HIP_TRY(hipSetDevice(h->device));
        if (!h->d_scratch) HIP_TRY(hipMalloc((void **)&h->d_scratch, 8 * (size_t)h->batch * 2 * h->nVar * h->nVar));
        HIP_TRY(launch_leastnorm(h->args(), h->stream));
        h->x_epoch = 0; // d_x now holds a least-norm solution
        return LEXLS_OK;
    }

    int lexls_lse_solve_least_norm_2(lexls_lse_t h)
    {
        if (int rc = need_factor(h, "lexls_lse_solve_least_norm_2")) return rc;
        HIP_TRY(hipSetDevice(h->device));
        if (!h->d_scratch) HIP_TRY(hipMalloc((void **)&h->d_scratch, 8 * (size_t)h->batch * 2 * h->nVar * h->nVar));
        HIP_TRY(launch_leastnorm2(h->args(), h->stream));
        h->x_epoch = 0; // d_x now holds a least-norm solution
        return LEXLS_OK;
    }

    int lexls_lse_solve_least_norm_3(lexls_lse_t h)
    {
        if (int rc = need_factor(h, "lexls_lse_solve_least_norm_3")) return rc;
        if (h->reg_type != 1 && h->reg_type != 2 && h->reg_type != 8 && h->reg_type != 3 && h->reg_type != 7)
            return fail(LEXLS_ERR_INVALID, "lexls_lse_solve_least_norm_3: needs a factorization with a regularization type that accumulates the null-space basis (lexlse.h:1217-1221)");
        HIP_TRY(hipSetDevice(h->device));
        if (!h->d_scratch) HIP_TRY(hipMalloc((void **)&h->d_scratch, 8 * (size_t)h->batch * 2 * h->nVar * h->nVar));
        HIP_TRY(launch_leastnorm3(h->args(), h->stream));
        h->x_epoch = 0; // d_x now holds a least-norm solution
        return LEXLS_OK;
    }

    int lexls_lse_residual(lexls_lse_t h)
    {
        if (int rc = need_factor(h, "lexls_lse_residual")) return rc;
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(launch_residual(h->args(), h->stream));
        return LEXLS_OK;
    }

    int lexls_lse_sensitivity(lexls_lse_t h, const int32_t *h_obj_index, int32_t obj_index_all, double tolW, double tolC)
    {
        if (int rc = need_factor(h, "lexls_lse_sensitivity")) return rc;
        HIP_TRY(hipSetDevice(h->device));
        const int32_t *d_obj = nullptr;
        if (h_obj_index)
        {
            HIP_TRY(hipMemcpyAsync(h->d_objidx, h_obj_index, 4 * (size_t)h->batch, hipMemcpyHostToDevice, h->stream));
            if (!h->deferred_sync) HIP_TRY(hipStreamSynchronize(h->stream));
            d_obj = h->d_objidx;
        }
        else if (obj_index_all < 0 || (uint32_t)obj_index_all >= h->nObj)
        {
            return fail(LEXLS_ERR_INVALID, "ObjIndex >= nObj");
        }
        HIP_TRY(launch_sensitivity(h->args(), d_obj, obj_index_all, tolW, tolC, h->stream, h->sens_scan, h->max_level_dim));
        return LEXLS_OK;
    }

    int lexls_lse_set_sensitivity_scan(lexls_lse_t h, int on)
    {
        CHECK_HANDLE(h);
        h->sens_scan = on != 0;
        return LEXLS_OK;
    }

    int lexls_lse_sensitivity_resident(lexls_lse_t h, double tolW, double tolC)
    {
        if (int rc = need_factor(h, "lexls_lse_sensitivity_resident")) return rc;
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(launch_sensitivity(h->args(), h->d_objidx, 0, tolW, tolC, h->stream, h->sens_scan, h->max_level_dim));
        return LEXLS_OK;
    }

    static int download(lexls_lse_t h, void *dst, const void *src, size_t bytes)
    {
        CHECK_HANDLE(h);
        if (!dst) return fail(LEXLS_ERR_INVALID, "null output pointer");
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
        if (!h->deferred_sync) HIP_TRY(hipStreamSynchronize(h->stream));
        return LEXLS_OK;
    }

    int lexls_lse_get_x(lexls_lse_t h, double *h_x) { return h ? download(h, h_x, h->d_x, 8 * (size_t)h->batch * h->nVar) : fail(LEXLS_ERR_INVALID, "null handle"); }
    int lexls_lse_get_factor(lexls_lse_t h, double *h_lod)
    {
        if (int rc = need_factor(h, "lexls_lse_get_factor")) return rc;
        return download(h, h_lod, h->d_fac, 8 * (size_t)h->batch * h->problem_elems());
    }
    int lexls_lse_get_hh_scalars(lexls_lse_t h, double *h_hh) { return h ? download(h, h_hh, h->d_hh, 8 * (size_t)h->batch * h->cap) : fail(LEXLS_ERR_INVALID, "null handle"); }
    int lexls_lse_get_permutation(lexls_lse_t h, uint32_t *h_perm)
    {
        return h ? download(h, h_perm, h->d_perm, 4 * (size_t)h->batch * h->nVar) : fail(LEXLS_ERR_INVALID, "null handle");
    }
    int lexls_lse_get_ranks(lexls_lse_t h, uint32_t *h_rank, uint32_t *h_first_col, uint32_t *h_total_rank)
    {
        CHECK_HANDLE(h);
        int rc = LEXLS_OK;
        if (h_rank) rc = download(h, h_rank, h->d_rank, 4 * (size_t)h->batch * h->nObj);
        if (!rc && h_first_col) rc = download(h, h_first_col, h->d_fcol, 4 * (size_t)h->batch * h->nObj);
        if (!rc && h_total_rank) rc = download(h, h_total_rank, h->d_totalrank, 4 * (size_t)h->batch);
        return rc;
    }
    int lexls_lse_get_mu(lexls_lse_t h, double *h_x_mu, double *h_x_mu_rhs, double *h_residual_mu)
    {
        CHECK_HANDLE(h);
        if (h->reg_type != 7 || !h->d_reg_mu) return fail(LEXLS_ERR_INVALID, "lexls_lse_get_mu: X_mu / X_mu_rhs / residual_mu exist with REGULARIZATION_TIKHONOV_1 (7) only");
        HIP_TRY(hipSetDevice(h->device));
        const size_t n = h->nVar, nObj = h->nObj, cap = h->cap, per = reg_mu_doubles(h->nVar, h->nObj, h->cap);
        if (h_x_mu) HIP_TRY(hipMemcpy2DAsync(h_x_mu, 8 * nObj * n, h->d_reg_mu, 8 * per, 8 * nObj * n, h->batch, hipMemcpyDeviceToHost, h->stream));
        if (h_x_mu_rhs) HIP_TRY(hipMemcpy2DAsync(h_x_mu_rhs, 8 * nObj * n, h->d_reg_mu + nObj * n, 8 * per, 8 * nObj * n, h->batch, hipMemcpyDeviceToHost, h->stream));
        if (h_residual_mu) HIP_TRY(hipMemcpy2DAsync(h_residual_mu, 8 * cap, h->d_reg_mu + 2 * nObj * n, 8 * per, 8 * cap, h->batch, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        return LEXLS_OK;
    }
    int lexls_lse_get_v(lexls_lse_t h, double *h_v) { return h ? download(h, h_v, h->d_v, 8 * (size_t)h->batch * h->cap) : fail(LEXLS_ERR_INVALID, "null handle"); }
    int lexls_lse_get_lambda(lexls_lse_t h, double *h_lambda)
    {
        return h ? download(h, h_lambda, h->d_lambda, 8 * (size_t)h->batch * (h->nVar + h->cap)) : fail(LEXLS_ERR_INVALID, "null handle");
    }
    int lexls_lse_get_sensitivity(lexls_lse_t h, int32_t *h_found_ctr_obj, double *h_max_abs)
    {
        CHECK_HANDLE(h);
        int rc = LEXLS_OK;
        if (h_found_ctr_obj) rc = download(h, h_found_ctr_obj, h->d_sens, 4 * (size_t)h->batch * 3);
        if (!rc && h_max_abs) rc = download(h, h_max_abs, h->d_maxabs, 8 * (size_t)h->batch);
        return rc;
    }
    int lexls_lse_get_fixed_type(lexls_lse_t h, uint8_t *h_types) { return h ? download(h, h_types, h->d_fixed_type, (size_t)h->batch * h->nVar) : fail(LEXLS_ERR_INVALID, "null handle"); }
    int lexls_lse_get_ctr_type(lexls_lse_t h, uint8_t *h_types) { return h ? download(h, h_types, h->d_ctr_type, (size_t)h->batch * h->cap) : fail(LEXLS_ERR_INVALID, "null handle"); }

    int lexls_lse_device_ptr(lexls_lse_t h, int which, void **d_ptr)
    {
        CHECK_HANDLE(h);
        if (!d_ptr) return fail(LEXLS_ERR_INVALID, "null output pointer");
        switch (which)
        {
        case LEXLS_ARRAY_X: *d_ptr = h->d_x; break;
        case LEXLS_ARRAY_FACTOR: *d_ptr = h->d_fac; break;
        case LEXLS_ARRAY_HH: *d_ptr = h->d_hh; break;
        case LEXLS_ARRAY_PERM: *d_ptr = h->d_perm; break;
        case LEXLS_ARRAY_RANK: *d_ptr = h->d_rank; break;
        case LEXLS_ARRAY_FIRST_COL: *d_ptr = h->d_fcol; break;
        case LEXLS_ARRAY_TOTAL_RANK: *d_ptr = h->d_totalrank; break;
        case LEXLS_ARRAY_V: *d_ptr = h->d_v; break;
        case LEXLS_ARRAY_LAMBDA: *d_ptr = h->d_lambda; break;
        case LEXLS_ARRAY_INPUT:
            if (!h->d_in_owned)
            {
                HIP_TRY(hipSetDevice(h->device));
                HIP_TRY(hipMalloc((void **)&h->d_in_owned, 8 * (size_t)h->batch * h->problem_elems()));
            }
            *d_ptr  = h->d_in_owned;
            h->d_in = h->d_in_owned;
            h->fused_gather = false;
            break;
        default: return fail(LEXLS_ERR_INVALID, "unknown array id");
        }
        return LEXLS_OK;
    }

    const char *lexls_lse_last_kernel(lexls_lse_t h) { return h ? h->last_kernel : ""; }

    int lexls_lse_set_prefix_reuse(lexls_lse_t h, int enable)
    {
        CHECK_HANDLE(h);
        HIP_TRY(hipSetDevice(h->device));
        if (enable && !h->d_resume_state)
        {
            HIP_TRY(hipMalloc((void **)&h->d_resume_state, (size_t)h->batch * resume_state_bytes(h->nObj)));
            HIP_TRY(hipMalloc((void **)&h->d_resume_level, 4 * (size_t)h->batch));
            HIP_TRY(hipMemsetAsync(h->d_resume_level, 0, 4 * (size_t)h->batch, h->stream));
        }
        h->resume_enabled = enable != 0;
        h->resume_valid   = false;
        h->resume_armed   = false;
        return LEXLS_OK;
    }
    int lexls_lse_prefix_reuse_ready(lexls_lse_t h) { return (h && h->resume_enabled && h->resume_valid) ? 1 : 0; }
    int lexls_lse_set_resume_levels(lexls_lse_t h, const int32_t *h_levels)
    {
        CHECK_HANDLE(h);
        if (!h->resume_enabled) return fail(LEXLS_ERR_INVALID, "set_resume_levels: lexls_lse_set_prefix_reuse(h, 1) first");
        if (!h_levels) return fail(LEXLS_ERR_INVALID, "set_resume_levels: levels is NULL");
        if (!h->resume_valid) return fail(LEXLS_ERR_INVALID, "set_resume_levels: the last factorization left nothing to resume from (another kernel, no factor kept, or none yet)");
        for (uint32_t b = 0; b < h->batch; b++)
            if (h_levels[b] < 0 || (uint32_t)h_levels[b] > h->nObj) return fail(LEXLS_ERR_INVALID, "set_resume_levels: a level is outside 0 .. nObj");
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMemcpyAsync(h->d_resume_level, h_levels, 4 * (size_t)h->batch, hipMemcpyHostToDevice, h->stream));
        if (!h->deferred_sync) HIP_TRY(hipStreamSynchronize(h->stream));
        h->resume_armed = true;
        return LEXLS_OK;
    }
    /* internal (the lock-step LexLSI driver): the device array its resident iterations write the levels into, and the promise that it holds
     * them for the next factorization */
    int32_t *lexls_internal_resume_levels(lexls_lse_t h) { return (h && h->resume_enabled) ? h->d_resume_level : nullptr; }
    void lexls_internal_arm_resume(lexls_lse_t h)
    {
        if (h && h->resume_enabled) h->resume_armed = true;
    }

    int lexls_lse_set_kernel_policy(lexls_lse_t h, int force_generic)
    {
        CHECK_HANDLE(h);
        if (force_generic != h->force_generic) HIP_TRY(materialize_fused_gather(h)); // another kernel may read `in`: the rows named by the round must be there
        h->force_generic = force_generic;
        return LEXLS_OK;
    }
}
