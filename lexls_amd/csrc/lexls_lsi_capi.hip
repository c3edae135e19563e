// C ABI for inequality problems: the host-side active-set driver (include/lexls/lexlsi.h) instantiated over
// the HIP-backed equality solver (include/lexls/lexlse.h).  The equality solves happen inside the lexls_lse_* calls the driver
// issues; lock-step batches additionally run the step of an iteration (A*dx, ratio test, state update — SURVEY 8(f) item 1) in the
// kernel below, next to the equality solve, on the constraint data that is resident for the row gather anyway.
#include <lexls/lexls.h>
#include <lexls/lsi_runner.h>

#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

using namespace LexLS;

extern "C" int lexls_internal_upload_round_trusted(lexls_lse_t h, const void *h_in, int gather); // lexls_capi.hip
extern "C" const double *lexls_internal_cdata(lexls_lse_t h);                                        // lexls_capi.hip
extern "C" char *lexls_internal_round_in(lexls_lse_t h);                                             // lexls_capi.hip
extern "C" int lexls_internal_round_resident(lexls_lse_t h, int has_fixed);                          // lexls_capi.hip
extern "C" int32_t *lexls_internal_resume_levels(lexls_lse_t h);                                     // lexls_capi.hip
extern "C" void lexls_internal_arm_resume(lexls_lse_t h);                                            // lexls_capi.hip
extern "C" int lexls_internal_resident_fused(lexls_lse_t h, int has_fixed, int count, double tolW, double tolC, const void *resident_args, size_t resident_args_bytes); // lexls_capi.hip
#include "lqr_wave_common.h" // wave_max

namespace
{

    /// p[0..9): see lexls_lsi_solve; p[9..12) (only read when nparams >= 12): regularization_type, variable_regularization_factor,
    /// max_number_of_CG_iterations (typedefs.h:185-187)
    ParametersLexLSI unpack(const double *p, uint32_t nparams = 9)
    {
        ParametersLexLSI par;
        if (p)
        {
            par.max_number_of_factorizations = static_cast<Index>(p[0]);
            par.tol_linear_dependence        = p[1];
            par.tol_wrong_sign_lambda        = p[2];
            par.tol_correct_sign_lambda      = p[3];
            par.tol_feasibility              = p[4];
            par.cycling_handling_enabled     = p[5] != 0;
            par.cycling_max_counter          = static_cast<Index>(p[6]);
            par.cycling_relax_step           = p[7];
            par.deactivate_first_wrong_sign  = p[8] != 0;
            if (nparams >= 12)
            {
                par.regularization_type            = static_cast<RegularizationType>(static_cast<int>(p[9]));
                par.variable_regularization_factor = p[10];
                par.max_number_of_CG_iterations    = static_cast<Index>(p[11]);
            }
        }
        return par;
    }
} // namespace

// ---------------------------------------------------------------------------------------------------
// Lock-step batch: B driver instances share ONE batched device handle.  SlotLSE is the equality-solver
// facade each instance sees: its setters stage the instance's problem in the batch's host arrays, its
// factorize()/solve()/ObjectiveSensitivity() return what the batch call of this round already computed.
// ---------------------------------------------------------------------------------------------------
namespace
{
    void hip_check(int rc)
    {
        if (rc != LEXLS_OK) throw Exception(std::string("liblexls_hip: ") + lexls_last_error());
    }

} // namespace
#include "lexls_lsi_device.h" // StepShape / StepArgs / lsi_step_kernel, ResidentArgs / lsi_iterate_kernel (shared with the persistent iteration kernel)
namespace
{

    /// host array in pinned memory (hipHostMalloc): the per-round copies of a lock-step batch are enqueued, not waited for
    /// (lexls_lse_set_deferred_sync), so their sources / destinations must be DMA-able and stable until the round's synchronize
    template <class T>
    struct Pinned
    {
        T *p     = NULL;
        size_t n = 0;
        Pinned() {}
        Pinned(const Pinned &)            = delete;
        Pinned &operator=(const Pinned &) = delete;
        ~Pinned()
        {
            if (p) (void)hipHostFree(p);
        }
        void assign(size_t n_, T v)
        {
            if (p) (void)hipHostFree(p);
            p = NULL;
            if (hipHostMalloc((void **)&p, (n_ ? n_ : 1) * sizeof(T), hipHostMallocDefault) != hipSuccess) throw Exception("hipHostMalloc failed (lock-step LSI batch)");
            n = n_;
            std::fill(p, p + n, v);
        }
        T *data() { return p; }
        T &operator[](size_t i) { return p[i]; }
        T *begin() { return p; }
        T *end() { return p + n; }
    };

    /// typed window into a pinned block (the per-round arrays of a batch sit in blocks laid out like the handle's device slabs)
    template <class T>
    struct View
    {
        T *p     = NULL;
        size_t n = 0;
        void bind(void *base, uint64_t offset, size_t n_, T v)
        {
            p = reinterpret_cast<T *>(static_cast<char *>(base) + offset);
            n = n_;
            std::fill(p, p + n, v);
        }
        T *data() { return p; }
        T &operator[](size_t i) { return p[i]; }
        T *begin() { return p; }
        T *end() { return p + n; }
    };

    struct BatchCtx
    {
        lexls_lse_t h = NULL;
        hipStream_t stream = NULL; // every group of a lock-step batch has its own stream: group A's kernels run while group B's host logic does
        hipStream_t stream_sens = NULL; // the sensitivity kernel of a stage serves other instances than its l-QR kernel: they run side by side
        hipEvent_t ev_uploaded = NULL, ev_sens_done = NULL;
        bool stage_fs = false, stage_sens = false; // what the stage in flight serves
        uint32_t B = 0, n = 0, nObjL = 0, cap = 0;
        size_t pstride = 0;
        std::vector<uint32_t> maxdim, rank, totalrank;
        std::vector<double> x;
        lexls_round_layout lay;
        Pinned<char> in_block, out_block; // pinned mirrors of the handle's round slabs: ONE copy each per stage
        View<uint32_t> dims, nfixed, fixed_idx, row_src, row_ld, tr_dl; // row_src/row_ld: B x cap, where each LOD row comes from (device gather)
        View<double> fixed_val, maxabs, x_dl;
        double *lod = NULL; // B x cap x (n+1), PINNED: host-staging fallback, uploaded every active-set round
        View<uint8_t> fixed_type, ctr_type, skip;
        View<int32_t> sens, objidx;
        std::vector<double> reg_factor;        // B x nObjL regularization factors (host copy; uploaded when they change)
        int reg_type = 0;                      // LexLS::RegularizationType shared by the batch
        double reg_variable = 0.0;
        uint32_t reg_cg_iters = 10;
        std::atomic<bool> reg_dirty{false};
        bool gather = false;                   // constraint data resident on the device: only row references travel per round
        // ---- step of an iteration on the device (lsi_step_kernel) ----
        bool device_step = false;
        StepShape shape;
        double *d_state = NULL, *d_state_in = NULL, *d_res = NULL;
        uint32_t *d_var = NULL;
        uint8_t *d_wset = NULL;
        Pinned<double> state_host, res_host;       // B x SD (hand-over staging, final download), B x 4
        Pinned<uint8_t> wset_host;                 // [mode B | ctr_state B x total | inact_pos (u16) B x total]
        size_t wset_bytes = 0, wset_state = 0, wset_pos = 0;
        std::vector<uint8_t> on_device;            // per instance: x / v / A x live on the device
        std::atomic<bool> handover{false};         // some instance put its state into state_host for the next stage
        bool stage_step = false;
        bool spec_sens  = false; // every factorization is followed by its removal search in the same stage (results used if the step is not blocked)
        // ---- resident iterations (lsi_iterate_kernel): x / v / A x, the working sets and the counters of an instance live on the device ----
        bool resident = false; // buffers exist (the structure allows it)
        StepShape rshape;
        uint32_t r_off = 0;
        double *d_rstate = NULL;
        uint32_t *d_rvar = NULL;
        char *d_rws      = NULL; // one slab: ctr_state | alive | act | inact | inact_pos | na | info | finished
        size_t rws_bytes = 0, r_alive = 0, r_act = 0, r_inact = 0, r_ipos = 0, r_na = 0, r_info = 0, r_fin = 0;
        Pinned<char> rws_host;
        Pinned<double> rstate_host;
        Pinned<uint32_t> fin_host;
        std::vector<uint8_t> is_resident; // per instance: handed over to the device
        uint32_t n_resident  = 0;
        int rounds_resident  = 0;
        std::vector<int32_t> iterations_at_handover;
        int rounds_fs_at_handover = 0, rounds_sens_at_handover = 0;
        bool fused_all = false, fused_refused = false; // the rest of the resident iterations is one persistent launch / the shape has none
        int rounds_fs = 0, rounds_sens = 0, rounds_step = 0;
        double t_enqueue = 0, t_wait = 0; // seconds, reported when LEXLS_LSI_TIMING is set
        static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

        void create(int device, uint32_t B_, uint32_t n_, uint32_t nObjL_, const uint32_t *maxdim_, bool gather_)
        {
            gather = gather_;
            B     = B_;
            n     = n_;
            nObjL = nObjL_;
            maxdim.assign(maxdim_, maxdim_ + nObjL);
            cap = 0;
            for (uint32_t k = 0; k < nObjL; k++) cap += maxdim[k];
            pstride = (size_t)cap * (n + 1);
            hip_check(lexls_lse_create(&h, device, B, n, nObjL, maxdim.data()));
            // prefix reuse in the resident iterations: LEXLS_LSI_PREFIX_REUSE=0 factorizes everything in every iteration
            if (!(std::getenv("LEXLS_LSI_PREFIX_REUSE") && std::atoi(std::getenv("LEXLS_LSI_PREFIX_REUSE")) == 0)) hip_check(lexls_lse_set_prefix_reuse(h, 1));
            if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&stream_sens, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&ev_uploaded, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_sens_done, hipEventDisableTiming) != hipSuccess)
                throw Exception("hipStreamCreate / hipEventCreate failed (lock-step LSI batch)");
            hip_check(lexls_lse_set_stream(h, stream));
            hip_check(lexls_lse_round_layout(h, &lay));
            in_block.assign(lay.in_bytes, 0);
            out_block.assign(lay.out_bytes, 0);
            if (!gather) need_staging();
            reset();
        }

        /// host staging of whole problems (pinned, B x cap x (n+1)): only for runs without the device-side gather
        void need_staging()
        {
            if (lod) return;
            if (hipHostMalloc((void **)&lod, 8 * (size_t)B * pstride, hipHostMallocDefault) != hipSuccess) throw Exception("hipHostMalloc failed for the LSI staging buffer");
            std::memset(lod, 0, 8 * (size_t)B * pstride);
        }

        /// buffers of the device-side step for a batch of this structure (once per batch object)
        void create_step(const StepShape &sh)
        {
            shape = sh;
            const size_t SD = sh.SD, total = sh.total;
            wset_state = ((size_t)B + 15) & ~size_t(15);
            wset_pos   = (wset_state + (size_t)B * total + 15) & ~size_t(15);
            wset_bytes = wset_pos + 2 * (size_t)B * total;
            if (hipMalloc((void **)&d_state, 8 * B * SD) != hipSuccess || hipMalloc((void **)&d_state_in, 8 * B * SD) != hipSuccess ||
                hipMalloc((void **)&d_res, 8 * (size_t)B * 4) != hipSuccess || hipMalloc((void **)&d_var, 4 * (size_t)B * (sh.dim0 ? sh.dim0 : 1)) != hipSuccess ||
                hipMalloc((void **)&d_wset, wset_bytes) != hipSuccess)
                throw Exception("hipMalloc failed (device-side LSI step)");
            state_host.assign((size_t)B * SD, 0.0);
            res_host.assign((size_t)B * 4, 0.0);
            wset_host.assign(wset_bytes, 0);
            on_device.assign(B, 0);
            device_step = true;
        }
        /// buffers of the resident iterations for a batch of this structure (once per batch object)
        void create_resident(const StepShape &sh, uint32_t off)
        {
            rshape = sh;
            r_off  = off;
            auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
            const size_t total = sh.total;
            size_t o = 0;
            o        = up(o + (size_t)B * total); // ctr_state at 0
            r_alive = o, o = up(o + B);
            r_act = o, o = up(o + 2 * (size_t)B * total);
            r_inact = o, o = up(o + 2 * (size_t)B * total);
            r_ipos = o, o = up(o + 2 * (size_t)B * total);
            r_na = o, o = up(o + 2 * (size_t)B * STEP_MAX_OBJ);
            r_info = o, o = up(o + 4 * (size_t)B * 8);
            r_fin = o, o = up(o + 16);
            rws_bytes = o;
            if (hipMalloc((void **)&d_rstate, 8 * (size_t)B * sh.SD) != hipSuccess || hipMalloc((void **)&d_rws, rws_bytes) != hipSuccess ||
                hipMalloc((void **)&d_rvar, 4 * (size_t)B * (sh.dim0 ? sh.dim0 : 1)) != hipSuccess)
                throw Exception("hipMalloc failed (resident LSI iterations)");
            rws_host.assign(rws_bytes, 0);
            rstate_host.assign((size_t)B * sh.SD, 0.0);
            fin_host.assign(4, 0u);
            is_resident.assign(B, 0);
            resident = true;
        }
        uint8_t *r_ctr_state(uint32_t b) { return reinterpret_cast<uint8_t *>(rws_host.data()) + (size_t)b * rshape.total; }
        int32_t *r_info_of(uint32_t b) { return reinterpret_cast<int32_t *>(rws_host.data() + r_info) + (size_t)b * 8; }

        /// instance b (its equality problem of a regular iteration is formed and staged in the in block) leaves the host: state, working
        /// sets in list order (workingset.h) and counters go into the hand-over slabs
        template <class LSI>
        void hand_over(uint32_t b, const LSI &inst)
        {
            const StepShape &sh = rshape;
            double *st          = rstate_host.data() + (size_t)b * sh.SD;
            const dVectorType &x = inst.get_x();
            for (uint32_t j = 0; j < sh.n; j++) st[j] = x(j);
            char *base    = rws_host.data();
            uint8_t *cs   = reinterpret_cast<uint8_t *>(base) + (size_t)b * sh.total;
            uint16_t *act = reinterpret_cast<uint16_t *>(base + r_act) + (size_t)b * sh.total;
            uint16_t *ina = reinterpret_cast<uint16_t *>(base + r_inact) + (size_t)b * sh.total;
            uint16_t *ip  = reinterpret_cast<uint16_t *>(base + r_ipos) + (size_t)b * sh.total;
            uint16_t *na  = reinterpret_cast<uint16_t *>(base + r_na) + (size_t)b * STEP_MAX_OBJ;
            std::memset(cs, 0, sh.total);
            const std::vector<internal::Objective> &obj = inst.getObjectives();
            for (uint32_t k = 0; k < sh.nObj; k++)
            {
                const uint32_t first = sh.first[k];
                const dVectorType &v = obj[k].get_v(), &ax = obj[k].get_Ax();
                for (uint32_t i = 0; i < sh.dim[k]; i++)
                {
                    st[sh.n + first + i]            = v(i);
                    st[sh.n + sh.total + first + i] = ax(i);
                }
                na[k] = static_cast<uint16_t>(obj[k].getActiveCtrCount());
                for (Index a = 0; a < obj[k].getActiveCtrCount(); a++)
                {
                    act[first + a]                          = static_cast<uint16_t>(obj[k].getActiveCtrIndex(a));
                    cs[first + obj[k].getActiveCtrIndex(a)] = static_cast<uint8_t>(obj[k].getActiveCtrType(a));
                }
                for (Index i = 0; i < obj[k].getInactiveCtrCount(); i++)
                {
                    ina[first + i]                            = static_cast<uint16_t>(obj[k].getInactiveCtrIndex(i));
                    ip[first + obj[k].getInactiveCtrIndex(i)] = static_cast<uint16_t>(i);
                }
            }
            int32_t *info = r_info_of(b);
            info[0]       = static_cast<int32_t>(inst.getStatus());
            info[1]       = static_cast<int32_t>(inst.getIterationsCount());
            info[2]       = static_cast<int32_t>(inst.getActivationsCount());
            info[3]       = static_cast<int32_t>(inst.getDeactivationsCount());
            info[4]       = static_cast<int32_t>(inst.getFactorizationsCount());
            info[5]       = static_cast<int32_t>(totalrank[b]);
            info[6] = info[7] = 0; // prefix reuse: levels read back, summed over the resident factorizations; their number
            reinterpret_cast<uint8_t *>(base + r_alive)[b] = 1;
            is_resident[b] = 1;
            skip[b]        = 0; // its staged equality problem is served by the first resident stage, followed by its removal sweep
            objidx[b]      = 0;
        }

        ResidentArgs resident_args(int32_t max_factorizations)
        {
            ResidentArgs ra;
            std::memset(&ra, 0, sizeof(ra));
            void *d_out = NULL;
            hip_check(lexls_lse_device_ptr(h, LEXLS_ARRAY_X, &d_out)); // x is the head of the out slab (lexls_lse_round_layout)
            char *out = static_cast<char *>(d_out), *in = lexls_internal_round_in(h);
            ra.sh     = rshape;
            ra.B = B, ra.cap = cap, ra.nObjL = nObjL, ra.off = r_off;
            ra.max_factorizations = max_factorizations;
            ra.cdata     = lexls_internal_cdata(h);
            ra.var       = d_rvar;
            ra.x_lse     = reinterpret_cast<const double *>(out + lay.x);
            ra.totalrank = reinterpret_cast<const uint32_t *>(out + lay.total_rank);
            ra.sens      = reinterpret_cast<const int32_t *>(out + lay.found);
            ra.state     = d_rstate;
            ra.ctr_state = reinterpret_cast<uint8_t *>(d_rws);
            ra.alive     = reinterpret_cast<uint8_t *>(d_rws + r_alive);
            ra.act       = reinterpret_cast<uint16_t *>(d_rws + r_act);
            ra.inact     = reinterpret_cast<uint16_t *>(d_rws + r_inact);
            ra.inact_pos = reinterpret_cast<uint16_t *>(d_rws + r_ipos);
            ra.na        = reinterpret_cast<uint16_t *>(d_rws + r_na);
            ra.info      = reinterpret_cast<int32_t *>(d_rws + r_info);
            ra.finished  = reinterpret_cast<uint32_t *>(d_rws + r_fin);
            ra.dims      = reinterpret_cast<uint32_t *>(in + lay.dims);
            ra.nfixed    = reinterpret_cast<uint32_t *>(in + lay.nfixed);
            ra.fixed_idx = reinterpret_cast<uint32_t *>(in + lay.fixed_idx);
            ra.fixed_val = reinterpret_cast<double *>(in + lay.fixed_val);
            ra.skip      = reinterpret_cast<uint8_t *>(in + lay.skip);
            ra.objidx    = reinterpret_cast<int32_t *>(in + lay.obj_index);
            ra.row_src   = reinterpret_cast<uint32_t *>(in + lay.row_src);
            ra.row_ld    = reinterpret_cast<uint32_t *>(in + lay.row_ld);
            ra.fixed_type = reinterpret_cast<uint8_t *>(in + lay.fixed_type);
            ra.ctr_type   = reinterpret_cast<uint8_t *>(in + lay.ctr_type);
            ra.resume     = lexls_internal_resume_levels(h);
            return ra;
        }

        /// the handed-over instances start: slabs up, then `count` whole iterations are enqueued (nothing is waited for)
        void begin_resident()
        {
            *reinterpret_cast<uint32_t *>(rws_host.data() + r_fin) = 0u;
            if (hipMemcpyAsync(d_rws, rws_host.data(), rws_bytes, hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipMemcpyAsync(d_rstate, rstate_host.data(), 8 * (size_t)B * rshape.SD, hipMemcpyHostToDevice, stream) != hipSuccess)
                throw Exception("hipMemcpyAsync failed (resident hand-over)");
            rounds_resident = 0;
            fused_all       = false;
            rounds_fs_at_handover   = rounds_fs;
            rounds_sens_at_handover = rounds_sens;
            iterations_at_handover.assign(B, 0);
            for (uint32_t b = 0; b < B; b++) iterations_at_handover[b] = reinterpret_cast<const int32_t *>(rws_host.data() + r_info)[(size_t)b * 8 + 1];
        }
        void enqueue_resident(int count, double tolW, double tolC, int32_t max_factorizations)
        {
            const double t0        = now();
            const ResidentArgs ra  = resident_args(max_factorizations);
            for (int i = 0; i < count && !fused_all; i++)
            {
                if (rounds_resident == 0)
                    hip_check(lexls_internal_upload_round_trusted(h, in_block.data(), 1)); // the problems the host formed last
                else
                {
                    // every iteration that is left, of every instance, in ONE persistent launch (lsi_fused_impl.h) where the shape has one: an
                    // instance runs l-QR -> removal sweep -> iteration until it stops, at most max_factorizations times
                    if (!fused_refused)
                    {
                        const int rc = lexls_internal_resident_fused(h, rshape.dim0 ? 1 : 0, max_factorizations > 0 ? max_factorizations : 1, tolW, tolC, &ra, sizeof(ra));
                        if (rc == LEXLS_OK)
                        {
                            fused_all = true;
                            rounds_resident++;
                            rounds_fs++;
                            rounds_sens++;
                            break;
                        }
                        if (rc != 1) hip_check(rc);
                        fused_refused = true;
                    }
                    hip_check(lexls_internal_round_resident(h, rshape.dim0 ? 1 : 0)); // the problems lsi_iterate_kernel formed
                    lexls_internal_arm_resume(h);                                     // ... and the levels it found unchanged
                }
                hip_check(lexls_lse_factorize_solve(h, 1));
                hip_check(lexls_lse_sensitivity_resident(h, tolW, tolC)); // speculative: used when the step is not blocked
                hipLaunchKernelGGL(lsi_iterate_kernel, dim3((B + 3) / 4), dim3(256), 4 * resident_lds_per_wave(rshape.SD, rshape.total), stream, ra);
                if (hipGetLastError() != hipSuccess) throw Exception("lsi_iterate_kernel launch failed");
                rounds_resident++;
                rounds_fs++;
                rounds_sens++;
            }
            if (hipMemcpyAsync(fin_host.data(), d_rws + r_fin, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) throw Exception("hipMemcpyAsync failed (finished count)");
            t_enqueue += now() - t0;
        }
        /// waits for what is enqueued; true when every handed-over instance has stopped
        bool resident_done()
        {
            const double t0 = now();
            hip_check(lexls_lse_synchronize(h));
            t_wait += now() - t0;
            return fin_host[0] >= n_resident;
        }
        void download_resident()
        {
            if (hipMemcpyAsync(rws_host.data(), d_rws, rws_bytes, hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipMemcpyAsync(rstate_host.data(), d_rstate, 8 * (size_t)B * rshape.SD, hipMemcpyDeviceToHost, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess)
                throw Exception("download of the resident state failed");
            if (fused_all && std::getenv("LEXLS_FUSED_STAMPS_DUMP")) // (a -DLEXLS_FUSED_STAMPS build leaves its phase clocks in the multiplier buffer)
            {
                std::vector<double> lam((size_t)B * (n + cap));
                hip_check(lexls_lse_get_lambda(h, lam.data()));
                hip_check(lexls_lse_synchronize(h));
                double sum[6] = {0, 0, 0, 0, 0, 0}, most[6] = {0, 0, 0, 0, 0, 0};
                for (uint32_t b = 0; b < B; b++)
                {
                    const double *o = lam.data() + (size_t)b * (n + cap);
                    for (int i = 0; i < 6; i++) sum[i] += o[i];
                    if (o[4] > most[4])
                        for (int i = 0; i < 6; i++) most[i] = o[i];
                }
                std::fprintf(stderr, "persistent launch, cycles per iteration [l-QR | step | removal search (per iteration) | finish], iterations, searches: all instances %.0f | %.0f | %.0f | %.0f, %.0f, %.0f; the longest-running one %.0f | %.0f | %.0f | %.0f, %.0f, %.0f\n",
                             sum[0] / sum[4], sum[1] / sum[4], sum[2] / sum[4], sum[3] / sum[4], sum[4], sum[5], most[0] / most[4], most[1] / most[4], most[2] / most[4], most[3] / most[4], most[4], most[5]);
            }
            if (fused_all) // the persistent launch: the stages it ran = the iterations of the instance that ran longest (statistics only)
            {
                int32_t most = 0;
                for (uint32_t b = 0; b < B; b++)
                    if (is_resident[b])
                    {
                        const int32_t d = reinterpret_cast<const int32_t *>(rws_host.data() + r_info)[(size_t)b * 8 + 1] - iterations_at_handover[b];
                        most            = d > most ? d : most;
                    }
                rounds_resident = most;
                rounds_fs       = rounds_fs_at_handover + most;
                rounds_sens     = rounds_sens_at_handover + most;
            }
        }
        uint8_t *mode() { return wset_host.data(); }
        uint8_t *ctr_state(uint32_t b) { return wset_host.data() + wset_state + (size_t)b * shape.total; }
        uint16_t *inact_pos(uint32_t b) { return reinterpret_cast<uint16_t *>(wset_host.data() + wset_pos) + (size_t)b * shape.total; }

        /// per-solve state: what a freshly created context holds (a context serves many lexls_lsi_batch_run calls)
        void reset()
        {
            dims.bind(in_block.data(), lay.dims, (size_t)B * nObjL, 0);
            nfixed.bind(in_block.data(), lay.nfixed, B, 0);
            fixed_idx.bind(in_block.data(), lay.fixed_idx, (size_t)B * n, 0);
            fixed_val.bind(in_block.data(), lay.fixed_val, (size_t)B * n, 0.0);
            skip.bind(in_block.data(), lay.skip, B, 0);
            objidx.bind(in_block.data(), lay.obj_index, B, -1);
            row_src.bind(in_block.data(), lay.row_src, (size_t)B * cap, 0);
            row_ld.bind(in_block.data(), lay.row_ld, (size_t)B * cap, 0);
            fixed_type.bind(in_block.data(), lay.fixed_type, (size_t)B * n, static_cast<uint8_t>(CTR_ACTIVE_UB));
            ctr_type.bind(in_block.data(), lay.ctr_type, (size_t)B * cap, static_cast<uint8_t>(CTR_INACTIVE));
            x_dl.bind(out_block.data(), lay.x, (size_t)B * n, 0.0);
            tr_dl.bind(out_block.data(), lay.total_rank, B, 0);
            sens.bind(out_block.data(), lay.found, (size_t)B * 3, 0);
            maxabs.bind(out_block.data(), lay.max_abs, B, 0.0);
            if (lod) std::memset(lod, 0, 8 * (size_t)B * pstride);
            x.assign((size_t)B * n, 0.0);
            rank.assign((size_t)B * nObjL, 0);
            totalrank.assign(B, 0);
            reg_factor.assign((size_t)B * nObjL, 0.0);
            rounds_fs = rounds_sens = rounds_step = 0;
            t_enqueue = t_wait = 0.0;
            if (device_step)
            {
                std::fill(wset_host.begin(), wset_host.end(), 0);
                std::fill(on_device.begin(), on_device.end(), 0);
                handover.store(false);
                stage_step = false;
            }
            stage_fs = stage_sens = false;
            if (resident)
            {
                std::fill(rws_host.begin(), rws_host.end(), 0);
                std::fill(is_resident.begin(), is_resident.end(), 0);
                n_resident      = 0;
                rounds_resident = 0;
            }
        }
        ~BatchCtx()
        {
            if (h) lexls_lse_destroy(h);
            void *dev[] = {d_state, d_state_in, d_res, d_var, d_wset, d_rstate, d_rvar, d_rws};
            for (void *q : dev)
                if (q) (void)hipFree(q);
            if (stream) (void)hipStreamDestroy(stream);
            if (stream_sens) (void)hipStreamDestroy(stream_sens);
            if (ev_uploaded) (void)hipEventDestroy(ev_uploaded);
            if (ev_sens_done) (void)hipEventDestroy(ev_sens_done);
            if (lod) (void)hipHostFree(lod);
        }

        /// Enqueue ONE stage on this group's stream: a batched factorize+solve for the instances with skip == 0 (if serve_fs) and a batched
        /// ObjectiveSensitivity for the instances with objidx >= 0 (if serve_sens) — disjoint sets of instances.  Nothing is waited for.
        void enqueue_stage(bool serve_fs, bool serve_sens, bool use_step, bool x_needed, double tolW, double tolC)
        {
            stage_step = false;
            const double t0 = now();
            stage_fs   = serve_fs;
            stage_sens = serve_sens;
            if (serve_fs)
            {
                if (reg_type != 0 && reg_dirty.exchange(false)) // the factors are the same every round: uploaded once (this call synchronises)
                {
                    hip_check(lexls_lse_set_cg_iterations(h, reg_cg_iters));
                    hip_check(lexls_lse_set_regularization(h, reg_type, reg_factor.data(), 1, reg_variable));
                }
                // dims, fixed variables, types, skip flags, sensitivity levels and row references: one copy (+ the gather kernel)
                hip_check(lexls_internal_upload_round_trusted(h, in_block.data(), gather ? 1 : 0));
                if (serve_sens && hipEventRecord(ev_uploaded, stream) != hipSuccess) throw Exception("hipEventRecord failed");
                if (!gather) hip_check(lexls_lse_set_problem_host(h, lod));
                hip_check(lexls_lse_factorize_solve(h, 1));
                rounds_fs++;
                stage_step = use_step;
                if (use_step) rounds_step++;
                if (use_step) // the step of the iteration, right behind its equality solve (same stream)
                {
                    if (hipMemcpyAsync(d_wset, wset_host.data(), wset_bytes, hipMemcpyHostToDevice, stream) != hipSuccess) throw Exception("hipMemcpyAsync failed (working sets)");
                    if (handover.exchange(false) &&
                        hipMemcpyAsync(d_state_in, state_host.data(), 8 * (size_t)B * shape.SD, hipMemcpyHostToDevice, stream) != hipSuccess)
                        throw Exception("hipMemcpyAsync failed (state hand-over)");
                    void *d_x = NULL;
                    hip_check(lexls_lse_device_ptr(h, LEXLS_ARRAY_X, &d_x));
                    StepArgs sa;
                    sa.sh        = shape;
                    sa.B         = B;
                    sa.cdata     = lexls_internal_cdata(h);
                    sa.var       = d_var;
                    sa.x_lse     = static_cast<const double *>(d_x);
                    sa.state     = d_state;
                    sa.state_in  = d_state_in;
                    sa.mode      = d_wset;
                    sa.ctr_state = d_wset + wset_state;
                    sa.inact_pos = reinterpret_cast<const uint16_t *>(d_wset + wset_pos);
                    sa.res       = d_res;
                    hipLaunchKernelGGL(lsi_step_kernel, dim3((B + 3) / 4), dim3(256), 8 * (size_t)shape.SD * 4, stream, sa);
                    if (hipGetLastError() != hipSuccess ||
                        hipMemcpyAsync(res_host.data(), d_res, 8 * (size_t)B * 4, hipMemcpyDeviceToHost, stream) != hipSuccess)
                        throw Exception("lsi_step_kernel launch / result copy failed");
                }
            }
            if (serve_sens)
            {
                if (serve_fs && !spec_sens)
                {
                    // disjoint instances (a problem is either re-factorised or asked for multipliers): the two kernels are both
                    // latency-bound at these batch sizes and share the chip — second stream, joined again before the download
                    if (hipStreamWaitEvent(stream_sens, ev_uploaded, 0) != hipSuccess) throw Exception("hipStreamWaitEvent failed");
                    hip_check(lexls_lse_set_stream(h, stream_sens));
                    hip_check(lexls_lse_sensitivity_resident(h, tolW, tolC));
                    hip_check(lexls_lse_set_stream(h, stream));
                    if (hipEventRecord(ev_sens_done, stream_sens) != hipSuccess || hipStreamWaitEvent(stream, ev_sens_done, 0) != hipSuccess)
                        throw Exception("hipEventRecord / hipStreamWaitEvent failed");
                }
                else if (serve_fs)
                    hip_check(lexls_lse_sensitivity_resident(h, tolW, tolC)); // behind the l-QR kernel: it reads the factors just made
                else
                    hip_check(lexls_lse_sensitivity(h, objidx.data(), 0, tolW, tolC));
                rounds_sens++;
            }
            // x / total rank / sensitivity verdicts in one copy.  (The CORRECT_SIGN_OF_LAMBDA marks ObjectiveSensitivity leaves on the
            // device, lexlse.h:866-987, only matter between the levels of ONE removal search — which is one launch here,
            // lexls_lse_set_sensitivity_scan — so they never have to come back: the next equality problem sets every row's type anew.)
            if (x_needed || !serve_fs)
                hip_check(lexls_lse_download_round(h, out_block.data(), NULL));
            else
            {
                // every equality solve of this stage feeds a device-side step: x stays on the device, only the tail of the out slab
                // (total ranks, sensitivity verdicts) comes back
                void *d_out = NULL;
                hip_check(lexls_lse_device_ptr(h, LEXLS_ARRAY_X, &d_out)); // x is the head of the out slab (lexls_lse_round_layout)
                if (hipMemcpyAsync(out_block.data() + lay.total_rank, static_cast<char *>(d_out) + lay.total_rank, lay.out_bytes - lay.total_rank, hipMemcpyDeviceToHost,
                                   stream) != hipSuccess)
                    throw Exception("hipMemcpyAsync failed (results without x)");
            }
            t_enqueue += now() - t0;
        }

        /// wait for the stage in flight (the ONE synchronisation of a stage); its results are taken over per instance, on the worker pool
        void finish_stage()
        {
            const double t0 = now();
            hip_check(lexls_lse_synchronize(h));
            t_wait += now() - t0;
        }
        void take_solution(uint32_t b)
        {
            std::copy(x_dl.begin() + (size_t)b * n, x_dl.begin() + (size_t)(b + 1) * n, x.begin() + (size_t)b * n);
            totalrank[b] = tr_dl[b];
        }
    };

    class SlotLSE
    {
    public:
        SlotLSE() : c(NULL), b(0), nVarFixed(0), nVarFixedInit(0) {}
        void bind(BatchCtx *ctx, uint32_t slot)
        {
            c = ctx;
            b = slot;
            x.resize(c->n);
            first_row.assign(c->nObjL, 0);
        }
        void resize(Index nVar_, Index nObj_, Index *maxObjDim)
        {
            if (!c) throw Exception("SlotLSE: not bound to a batch");
            if (nVar_ != c->n || nObj_ != c->nObjL) throw Exception("SlotLSE: shape differs from the batch");
            for (Index k = 0; k < nObj_; k++)
                if (maxObjDim[k] != c->maxdim[k]) throw Exception("SlotLSE: capacity differs from the batch");
        }
        void setParameters(const ParametersLexLSE &p)
        {
            tol = p.tol_linear_dependence; // tolerance and regularization type of the batch handle are set once by the batch driver
        }
        void setRegularizationFactor(Index ObjIndex, RealScalar factor)
        {
            double &f = c->reg_factor[(size_t)b * c->nObjL + ObjIndex];
            if (f != factor)
            {
                f = factor;
                c->reg_dirty.store(true);
            }
        }
        void setObjDim(Index *ObjDim_)
        {
            Index r = 0;
            for (Index k = 0; k < c->nObjL; k++)
            {
                c->dims[(size_t)b * c->nObjL + k] = ObjDim_[k];
                first_row[k]                      = r;
                r += ObjDim_[k];
            }
            nVarFixedInit = 0;
            if (c->gather) std::fill(c->row_ld.begin() + (size_t)b * c->cap, c->row_ld.begin() + (size_t)(b + 1) * c->cap, 0u);
        }
        /// row `CtrIndex` of the LOD = row of the resident constraint data (Objective::formLexLSE); false: not available, send numbers
        bool setCtrIndexed(Index CtrIndex, size_t first_element, Index ld, unsigned use_ub)
        {
            if (!c->gather) return false;
            c->row_src[(size_t)b * c->cap + CtrIndex] = static_cast<uint32_t>(first_element);
            c->row_ld[(size_t)b * c->cap + CtrIndex]  = static_cast<uint32_t>(ld) | (use_ub ? 0x80000000u : 0u);
            return true;
        }
        void setFixedVariablesCount(Index nf)
        {
            if (nf > c->n) throw Exception("Cannot fix more than nVar variables");
            nVarFixed    = nf;
            c->nfixed[b] = nf;
        }
        void fixVariable(Index VarIndex, RealScalar VarValue, ConstraintActivationType type = CTR_ACTIVE_UB)
        {
            const size_t o   = (size_t)b * c->n + nVarFixedInit++;
            c->fixed_idx[o]  = VarIndex;
            c->fixed_val[o]  = VarValue;
            c->fixed_type[o] = static_cast<uint8_t>(type);
        }
        void setCtrType(Index ObjIndex, Index CtrIndex, ConstraintActivationType type) { c->ctr_type[(size_t)b * c->cap + first_row[ObjIndex] + CtrIndex] = static_cast<uint8_t>(type); }
        void setCtrStrided(Index CtrIndex, const RealScalar *row, Index stride, RealScalar rhs)
        {
            double *L = c->lod + (size_t)b * c->pstride;
            for (Index j = 0; j < c->n; j++) L[CtrIndex + (size_t)j * c->cap] = row[(size_t)j * stride];
            L[CtrIndex + (size_t)c->n * c->cap] = rhs;
        }
        // served by the batch call of this round
        void factorize() {}
        void solve()
        {
            for (Index i = 0; i < c->n; i++) x(i) = c->x[(size_t)b * c->n + i];
        }
        bool ObjectiveSensitivity(Index, Index &CtrIndex2Remove, int &ObjIndex2Remove, RealScalar, RealScalar, RealScalar &maxAbsValue)
        {
            const int32_t *s3 = &c->sens[(size_t)b * 3];
            maxAbsValue       = c->maxabs[b];
            if (s3[0])
            {
                CtrIndex2Remove = static_cast<Index>(s3[1]);
                ObjIndex2Remove = s3[2];
            }
            return s3[0] != 0;
        }
        void ObjectiveSensitivity(Index, RealScalar, RealScalar, std::vector<ConstraintInfo> &) { throw Exception("not available in lock-step batches"); }
        const dVectorType &get_x() const { return x; }
        Index getTotalRank() const { return c->totalrank[b]; }
        Index getDim(Index k) const { return c->dims[(size_t)b * c->nObjL + k]; }
        Index getFixedVariablesCount() const { return nVarFixed; }
        const dVectorType &getWorkspace() const { return x; }
        const dMatrixType &get_lexqr() { throw Exception("not available in lock-step batches"); }
        const dMatrixType &get_data() { throw Exception("not available in lock-step batches"); }

    private:
        BatchCtx *c;
        uint32_t b;
        Index nVarFixed, nVarFixedInit;
        double tol = 1e-12;
        std::vector<Index> first_row;
        dVectorType x;
    };

    typedef internal::LexLSI_T<SlotLSE> SlotLSI;

    /// One instance's side of the device-side step (LexLSI_T::StepHook): posts the working set of the equality problem just formed,
    /// hands x / v / A x over to the device the first time, and reads the ratio test's verdict back.
    struct SlotStep : SlotLSI::StepHook
    {
        BatchCtx *c = NULL;
        uint32_t b  = 0;
        void prepare(const dVectorType &x, const std::vector<internal::Objective> &obj) override
        {
            const StepShape &sh = c->shape;
            uint8_t *cs         = c->ctr_state(b);
            uint16_t *ip        = c->inact_pos(b);
            std::memset(cs, 0, sh.total);
            for (uint32_t k = 0; k < sh.nObj; k++)
            {
                const uint32_t first = sh.first[k];
                for (Index a = 0; a < obj[k].getActiveCtrCount(); a++) cs[first + obj[k].getActiveCtrIndex(a)] = static_cast<uint8_t>(obj[k].getActiveCtrType(a));
                for (Index i = 0; i < obj[k].getInactiveCtrCount(); i++) ip[first + obj[k].getInactiveCtrIndex(i)] = static_cast<uint16_t>(i);
            }
            if (!c->on_device[b])
            {
                double *st = c->state_host.data() + (size_t)b * sh.SD;
                for (uint32_t j = 0; j < sh.n; j++) st[j] = x(j);
                for (uint32_t k = 0; k < sh.nObj; k++)
                {
                    const dVectorType &v = obj[k].get_v(), &ax = obj[k].get_Ax();
                    for (uint32_t i = 0; i < sh.dim[k]; i++)
                    {
                        st[sh.n + sh.first[k] + i]            = v(i);
                        st[sh.n + sh.total + sh.first[k] + i] = ax(i);
                    }
                }
                c->on_device[b] = 1;
                c->mode()[b]    = 2;
                c->handover.store(true);
            }
            else
                c->mode()[b] = 1;
        }
        bool blocking(Index &ObjIndex, Index &CtrIndex, ConstraintActivationType &CtrType, RealScalar &alpha) override
        {
            const double *r = c->res_host.data() + (size_t)b * 4;
            alpha           = r[0];
            if (r[1] < 0.0) return false;
            ObjIndex = static_cast<Index>(r[1]);
            CtrIndex = static_cast<Index>(r[2]);
            CtrType  = static_cast<ConstraintActivationType>(static_cast<int>(r[3]));
            return true;
        }
    };

    /// Persistent host worker pool for the per-instance work of a lock-step batch (the instances are independent; each touches only
    /// its own LexLSI object and its own slot of the staging arrays).  Created once per lexls_lsi_batch_solve call: the active-set
    /// rounds are short (~1 ms of host work for 1024 instances), so threads must not be spawned per round.
    class WorkerPool
    {
    public:
        explicit WorkerPool(uint32_t workers)
        {
            if (const char *e = std::getenv("LEXLS_POOL_SPIN_US")) spin_seconds = 1e-6 * std::atof(e); // diagnostic: 0 = sleep at once
            for (uint32_t i = 0; i < workers; i++) th.emplace_back([this]() { loop(); });
        }
        ~WorkerPool()
        {
            {
                std::lock_guard<std::mutex> lk(m);
                stop.store(true);
            }
            cv_start.notify_all();
            for (auto &t : th) t.join();
        }
        static uint32_t default_workers(uint32_t batch)
        {
            const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
            const uint32_t nt = std::min<uint32_t>(std::min<uint32_t>(hw, 16u), batch / 64);
            return nt > 1 ? nt - 1 : 0; // the calling thread works too
        }
        /// f(b) for b in [0, count); returns when all are done; rethrows the first exception.
        /// The stages of a batch solve follow each other every ~50 us: a condition-variable wake-up per stage would cost more than
        /// the stage's host work, so idle workers spin on the generation counter for a while (spin_seconds) before they go to sleep.
        /// light: a job of a few microseconds per element (copies, releases): when the workers have gone to sleep (the GPU ran for
        /// milliseconds meanwhile) waking sixteen threads through the condition variable costs more than running it here
        void run(uint32_t count_, const std::function<void(uint32_t)> &f, bool light = false)
        {
            while (pending.load(std::memory_order_acquire) != 0) relax(); // (a prewake() still being acknowledged)
            if (th.empty() || count_ < 128 || (light && sleepers.load() != 0))
            {
                for (uint32_t b = 0; b < count_; b++) f(b);
                return;
            }
            job   = &f;
            count = count_;
            next.store(0);
            err = nullptr;
            pending.store(static_cast<uint32_t>(th.size()), std::memory_order_relaxed);
            gen.fetch_add(1); // seq_cst with the sleepers' increment / generation check below: one side always sees the other
            if (sleepers.load() != 0)
            {
                std::lock_guard<std::mutex> lk(m); // a worker between its last check and its wait holds m: it sees the new generation
                cv_start.notify_all();
            }
            work();
            while (pending.load(std::memory_order_acquire) != 0) relax(); // every worker acknowledges every generation
            job = nullptr;
            if (err) std::rethrow_exception(err);
        }

        /// wakes sleeping workers ahead of a run() that is about to come (they spin again for spin_seconds): the wake-up latency of the
        /// condition variable (~0.1-0.3 ms for the last of sixteen threads) then overlaps what the caller does in between
        void prewake()
        {
            if (th.empty() || sleepers.load() == 0) return;
            while (pending.load(std::memory_order_acquire) != 0) relax();
            static const std::function<void(uint32_t)> nothing = [](uint32_t) {};
            job   = &nothing;
            count = 0;
            next.store(0);
            pending.store(static_cast<uint32_t>(th.size()), std::memory_order_relaxed);
            gen.fetch_add(1);
            {
                std::lock_guard<std::mutex> lk(m);
                cv_start.notify_all();
            }
        }

    private:
        double spin_seconds = 300e-6;
        static void relax()
        {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
        }
        static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
        void work()
        {
            const uint32_t chunk = 16;
            for (;;)
            {
                const uint32_t b0 = next.fetch_add(chunk);
                if (b0 >= count) break;
                const uint32_t b1 = std::min(count, b0 + chunk);
                try
                {
                    for (uint32_t b = b0; b < b1; b++) (*job)(b);
                }
                catch (...)
                {
                    std::lock_guard<std::mutex> lk(m);
                    if (!err) err = std::current_exception();
                }
            }
        }
        void loop()
        {
            uint64_t seen = 0;
            for (;;)
            {
                uint32_t spins = 0;
                double t_idle  = 0.0;
                while (gen.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_relaxed))
                {
                    relax();
                    if ((++spins & 255u) != 0) continue;
                    const double t = now();
                    if (t_idle == 0.0) t_idle = t;
                    if (t - t_idle < spin_seconds) continue;
                    std::unique_lock<std::mutex> lk(m);
                    sleepers.fetch_add(1);
                    cv_start.wait(lk, [&]() { return stop.load() || gen.load() != seen; });
                    sleepers.fetch_sub(1);
                }
                if (stop.load()) return;
                seen = gen.load(std::memory_order_acquire);
                work();
                pending.fetch_sub(1, std::memory_order_release);
            }
        }
        std::vector<std::thread> th;
        std::mutex m;
        std::condition_variable cv_start;
        const std::function<void(uint32_t)> *job = nullptr;
        uint32_t count                           = 0;
        std::atomic<uint32_t> next{0}, pending{0}, sleepers{0};
        std::atomic<uint64_t> gen{0};
        std::atomic<bool> stop{false};
        std::exception_ptr err;
    };
} // namespace

/// A lock-step batch that outlives one solve (the reference constructs a LexLSI once and feeds it successive problems, lexlsi.h:56-112):
/// device buffers, pinned blocks, streams and the worker pool are made once; every run() re-reads the problem data.
struct lexls_lsi_batch_s
{
    int device;
    uint32_t batch, nVar, nObj, off;
    std::vector<uint32_t> dims;
    std::vector<int32_t> types;
    size_t per_data = 0, total = 0;
    bool gather = false;
    uint32_t nGroups = 1;
    std::vector<std::unique_ptr<BatchCtx>> grp;
    std::vector<uint32_t> lo, group_of;
    std::unique_ptr<WorkerPool> pool;
    bool resident_ok = false;
    double t_create = 0.0;
    int32_t last_stats[4] = {0, 0, 0, 0}; // of the last run: factorize+solve stages, sensitivity stages, stages with the step on the device, groups

    lexls_lsi_batch_s(int device_, uint32_t batch_, uint32_t nVar_, uint32_t nObj_, const uint32_t *h_dims, const int32_t *h_types)
    : device(device_), batch(batch_), nVar(nVar_), nObj(nObj_)
    {
        if (batch == 0 || nObj == 0) throw Exception("lexls_lsi_batch_solve: empty batch");
        dims.assign(h_dims, h_dims + nObj);
        types.assign(h_types, h_types + nObj);
        off = (h_types[0] == 1) ? 1 : 0;
        if (nObj - off == 0) throw Exception("Problems consisting of one level of simple bounds are not supported."); // lexlsi.cpp:417
        for (uint32_t k = 0; k < nObj; k++)
        {
            per_data += (size_t)h_dims[k] * (h_types[k] == 1 ? 2 : nVar + 2);
            total += h_dims[k];
        }
        const double t_begin = BatchCtx::now();
        // The instances can be split into groups that take turns: while one group's stage runs on the GPU (its own stream), the host
        // advances the active-set logic of the other one.  Every stage carries fixed costs (one copy each way, launches, one
        // synchronisation) that a split multiplies, so it pays for large batches only.  Measured on MI355X, cold
        // solve of n = 40, 5 x 12, seconds with 1 / 2 / 3 groups: 512 instances 0.034 / 0.030 / 0.040; 1024: 0.039 / 0.034 / 0.045
        // (256 instances, an earlier state of the driver: 0.044 / 0.047).  LEXLS_LSI_GROUPS overrides the number.
        // (With resident iterations — the default, see below — there is no host work per stage left to overlap: one group.)
        {
            const char *want_res = std::getenv("LEXLS_LSI_RESIDENT"), *want_step = std::getenv("LEXLS_LSI_DEVICE_STEP");
            const bool host_logic = (want_res && std::atoi(want_res) == 0) || (want_step && std::atoi(want_step) != 0);
            nGroups               = (host_logic && batch >= 512) ? 2u : 1u;
        }
        if (const char *e = std::getenv("LEXLS_LSI_GROUPS")) nGroups = std::max(1, std::atoi(e));
        nGroups = std::min(nGroups, batch);
        gather = per_data < 0x7fffffffull && !std::getenv("LEXLS_LSI_HOST_STAGING"); // (diagnostic switch: assemble on the host, stage over PCIe)
        grp.resize(nGroups);
        lo.assign(nGroups + 1, 0);
        for (uint32_t g = 0; g < nGroups; g++) lo[g + 1] = lo[g] + batch / nGroups + (g < batch % nGroups ? 1u : 0u);
        for (uint32_t g = 0; g < nGroups; g++)
        {
            grp[g].reset(new BatchCtx());
            BatchCtx &ctx = *grp[g];
            ctx.create(device, lo[g + 1] - lo[g], nVar, nObj - off, h_dims + off, gather);
            hip_check(lexls_lse_set_deferred_sync(ctx.h, 1)); // every per-round array of BatchCtx is pinned and only touched between stages
            hip_check(lexls_lse_set_sensitivity_scan(ctx.h, 1)); // the removal search of an iteration in ONE sensitivity stage (all its levels)
            // The removal search runs speculatively behind every factorization (a third fewer stages: one synchronisation per active-set
            // iteration instead of two).  With round 1's 83-us level-by-level search it lost (1024 instances cold 0.040 s vs 0.036 s); since the
            // search is one 36-us sweep (sensitivity_sweep_kernel) it wins: warm-started ~30 iterations 0.027-0.028 s vs 0.029-0.034 s.
            // LEXLS_LSI_SPECULATIVE_SENS=0 restores the two-stage form.
            ctx.spec_sens = true;
            if (const char *e = std::getenv("LEXLS_LSI_SPECULATIVE_SENS")) ctx.spec_sens = std::atoi(e) != 0;
        }
        // the step of an iteration can run on the device when the constraint data is resident
        StepShape sh;
        std::memset(&sh, 0, sizeof(sh));
        sh.n = nVar, sh.nObj = nObj, sh.total = (uint32_t)total, sh.SD = nVar + 2 * (uint32_t)total, sh.per_data = per_data, sh.dim0 = off ? h_dims[0] : 0;
        // Off by default: measured on MI355X (scripts/lsi_ab.sh; 1024 / 4096 instances of n = 40, 5 x 12) it is a wash — cold 0.052 s
        // vs 0.049 s at 1024, 0.117 s vs 0.120 s at 4096: the host's share of a stage is parallel and small, the extra copy + kernel +
        // copy of a stage is not free.  LEXLS_LSI_DEVICE_STEP=1 turns it on.
        const char *want_step = std::getenv("LEXLS_LSI_DEVICE_STEP");
        bool step_ok = want_step && std::atoi(want_step) != 0 && gather && nObj <= STEP_MAX_OBJ && 8 * (size_t)sh.SD * 4 <= 48 * 1024;
        if (step_ok)
        {
            uint64_t o = 0;
            uint32_t f = 0;
            for (uint32_t k = 0; k < nObj; k++)
            {
                sh.dim[k] = h_dims[k], sh.simple[k] = h_types[k] == 1, sh.first[k] = f, sh.off[k] = o;
                if (h_dims[k] > 65535) step_ok = false;
                o += (uint64_t)h_dims[k] * (h_types[k] == 1 ? 2 : nVar + 2);
                f += h_dims[k];
            }
        }
        if (step_ok)
            for (uint32_t g = 0; g < nGroups; g++) grp[g]->create_step(sh);
        // Resident iterations (lsi_iterate_kernel), the default where the structure allows it: after phase 1 the instances leave the host —
        // a stage is row gather + l-QR + removal sweep + step / working-set change / next problem, all enqueued, and the host only polls
        // how many instances have stopped.  LEXLS_LSI_RESIDENT=0 keeps the active-set logic on the host (one synchronisation per stage).
        const char *want_res = std::getenv("LEXLS_LSI_RESIDENT");
        resident_ok = !(want_res && std::atoi(want_res) == 0) && !step_ok && gather && nObj <= STEP_MAX_OBJ && 4 * resident_lds_per_wave(sh.SD, (uint32_t)total) <= 48 * 1024 && total <= 65535;
        if (resident_ok)
        {
            uint64_t o = 0;
            uint32_t f = 0;
            for (uint32_t k = 0; k < nObj; k++)
            {
                sh.dim[k] = h_dims[k], sh.simple[k] = h_types[k] == 1, sh.first[k] = f, sh.off[k] = o;
                if (h_types[k] == 1 && k != 0) resident_ok = false; // (the driver itself only takes a simple-bounds objective first, lexlsi.h:402-405)
                o += (uint64_t)h_dims[k] * (h_types[k] == 1 ? 2 : nVar + 2);
                f += h_dims[k];
            }
            if (o > 0xffffffffull) resident_ok = false;
        }
        if (resident_ok)
            for (uint32_t g = 0; g < nGroups; g++) grp[g]->create_resident(sh, off);
        group_of.resize(batch);
        for (uint32_t g = 0; g < nGroups; g++)
            for (uint32_t b = lo[g]; b < lo[g + 1]; b++) group_of[b] = g;
        pool.reset(new WorkerPool(WorkerPool::default_workers(batch)));
        t_create = BatchCtx::now() - t_begin;
    }

    void run(const double *h_data, const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0, const double *h_v0,
             const double *h_reg_factors, const ParametersLexLSI &par, double *h_x, int32_t *h_info6, uint8_t *h_active, double *h_v, int32_t *h_rounds2)
    {
        if (!h_data || !h_x) throw Exception("lexls_lsi_batch_run: null data / x");
        if (par.deactivate_first_wrong_sign)
        {
            // The lock-step stages ask the device for ONE removal candidate per instance; this option (lexlsi.h:1089-1103) wants every
            // wrong-sign multiplier of the first level that has one, read back per iteration.  Such a batch runs its instances one after
            // the other through the single-problem driver — same results as lexls_lsi_solve_ex on each, every equality problem on the GPU.
            int32_t fs = 0;
            for (uint32_t b = 0; b < batch; b++)
            {
                runner::LsiProblem p = {nVar, nObj, dims.data(), types.data(), h_data + (size_t)b * per_data, h_var_index ? h_var_index + (size_t)b * dims[0] : NULL,
                                        h_active_guess ? h_active_guess + (size_t)b * total : NULL, h_x0 ? h_x0 + (size_t)b * nVar : NULL,
                                        h_v0 ? h_v0 + (size_t)b * total : NULL, h_reg_factors};
                internal::LexLSI lsi;
                lsi.getLexLSE().setDevice(device);
                lsi.getLexLSE().setSensitivityScan(true);
                lsi.setSensitivityScansAllLevels(true);
                runner::setup(lsi, p, par);
                lsi.solve();
                runner::LsiInfo info;
                runner::collect(lsi, p, h_x + (size_t)b * nVar, &info, h_active ? h_active + (size_t)b * total : NULL, h_v ? h_v + (size_t)b * total : NULL);
                if (h_info6) std::memcpy(h_info6 + (size_t)b * 6, &info, sizeof(info));
                fs += info.factorizations;
            }
            last_stats[0] = fs;
            last_stats[1] = last_stats[2] = 0;
            last_stats[3] = 1;
            if (h_rounds2) h_rounds2[0] = fs, h_rounds2[1] = 0;
            return;
        }
        const uint32_t *h_dims = dims.data();
        const int32_t *h_types = types.data();
        WorkerPool &pool       = *this->pool;
        const double t_begin   = BatchCtx::now();
        pool.prewake(); // (the workers went to sleep between two solves; they are needed in ~0.1 ms)
        // Cycling handling relaxes bounds in the host copy of the constraint data (cycling.h:32-65, objective.h:774-790): such a run
        // assembles its problems on the host from that copy instead of gathering rows of the resident (unrelaxed) device copy
        const bool run_gather = gather && !par.cycling_handling_enabled;
        for (uint32_t g = 0; g < nGroups; g++)
        {
            BatchCtx &ctx = *grp[g];
            ctx.gather    = run_gather;
            if (!run_gather) ctx.need_staging();
            ctx.reset();
            hip_check(lexls_lse_set_tolerance(ctx.h, par.tol_linear_dependence));
            ctx.reg_type     = static_cast<int>(par.regularization_type);
            ctx.reg_variable = par.variable_regularization_factor;
            ctx.reg_cg_iters = par.max_number_of_CG_iterations;
            ctx.reg_dirty.store(ctx.reg_type != 0);
            if (ctx.reg_type == 0) hip_check(lexls_lse_set_regularization(ctx.h, 0, NULL, 0, 0.0));
            if (run_gather && ctx.device_step)
            {
                ctx.shape.tol_feasibility = par.tol_feasibility;
                if (ctx.shape.dim0)
                {
                    if (!h_var_index) throw Exception("lexls_lsi_batch_run: a simple-bounds objective needs variable indices");
                    if (hipMemcpyAsync(ctx.d_var, h_var_index + (size_t)lo[g] * ctx.shape.dim0, 4 * (size_t)ctx.B * ctx.shape.dim0, hipMemcpyHostToDevice, ctx.stream) != hipSuccess ||
                        hipStreamSynchronize(ctx.stream) != hipSuccess)
                        throw Exception("upload of the variable indices failed");
                }
            }
        }
        // the constraint data goes to the device (16 MB for 1024 IK instances: ~0.4 ms) while the worker pool builds the instances' host
        // objects and runs their phase 1: nothing of that touches the handles; joined before the first stage
        int upload_rc = LEXLS_OK;
        std::string upload_err;
        std::thread uploader;
        if (run_gather)
            uploader = std::thread([&]() {
                for (uint32_t g = 0; g < nGroups && upload_rc == LEXLS_OK; g++)
                {
                    upload_rc = lexls_lse_set_constraint_data(grp[g]->h, h_data + (size_t)lo[g] * per_data, per_data);
                    if (upload_rc != LEXLS_OK) upload_err = lexls_last_error();
                }
            });
        struct Joiner // (the setup below may throw)
        {
            std::thread &t;
            ~Joiner()
            {
                if (t.joinable()) t.join();
            }
        } joiner{uploader};
        // whole iterations on the device: plain runs only (cycling handling edits the host's bounds; a regularized equality problem has
        // per-run factors the host posts)
        const bool run_resident = run_gather && resident_ok && grp[0]->resident && par.regularization_type == REGULARIZATION_NONE &&
                                  par.max_number_of_factorizations < 0x7fffffff;
        if (run_resident)
            for (uint32_t g = 0; g < nGroups; g++)
            {
                BatchCtx &ctx               = *grp[g];
                ctx.rshape.tol_feasibility = par.tol_feasibility;
                if (ctx.rshape.dim0)
                {
                    if (!h_var_index) throw Exception("lexls_lsi_batch_run: a simple-bounds objective needs variable indices");
                    if (hipMemcpyAsync(ctx.d_rvar, h_var_index + (size_t)lo[g] * ctx.rshape.dim0, 4 * (size_t)ctx.B * ctx.rshape.dim0, hipMemcpyHostToDevice, ctx.stream) != hipSuccess ||
                        hipStreamSynchronize(ctx.stream) != hipSuccess)
                        throw Exception("upload of the variable indices failed");
                }
            }
        const bool run_step = run_gather && grp[0]->device_step;
        std::vector<SlotStep> hooks(run_step ? batch : 0);
        const double t_ctx = BatchCtx::now() - t_begin;
        std::vector<std::unique_ptr<SlotLSI>> lsi(batch);
        std::vector<runner::LsiProblem> prob(batch);
        pool.run(batch, [&](uint32_t b) {
            const uint32_t g = group_of[b];
            lsi[b].reset(new SlotLSI());
            lsi[b]->getLexLSE().bind(grp[g].get(), b - lo[g]);
            prob[b] = {nVar,
                       nObj,
                       h_dims,
                       h_types,
                       h_data + (size_t)b * per_data,
                       h_var_index ? h_var_index + (size_t)b * h_dims[0] : NULL,
                       h_active_guess ? h_active_guess + (size_t)b * total : NULL,
                       h_x0 ? h_x0 + (size_t)b * nVar : NULL,
                       h_v0 ? h_v0 + (size_t)b * total : NULL,
                       h_reg_factors};
            runner::setup(*lsi[b], prob[b], par);
            lsi[b]->setSensitivityScansAllLevels(true);
            if (run_step)
            {
                hooks[b].c = grp[g].get();
                hooks[b].b = b - lo[g];
                lsi[b]->setStepHook(&hooks[b]);
            }
            lsi[b]->begin();
        });
        if (uploader.joinable()) uploader.join();
        if (upload_rc != LEXLS_OK) throw Exception(std::string("liblexls_hip: ") + upload_err);
        const double t_setup = BatchCtx::now() - t_begin;
        double t_host        = 0.0;

        // One stage of group g serves every pending factorize+solve of the group in one call and every pending ObjectiveSensitivity in
        // one call (different instances), both only enqueued.  Between two stages every instance of the group runs ONE job on the worker
        // pool: take over the results of the stage that just finished (if it was served), advance its active-set logic, and post what it
        // needs next into the group's round block.
        std::vector<std::atomic<uint32_t>> wants(nGroups); // bit 0: somebody alive, bit 1: a factorize+solve, bit 2: a sensitivity, bit 3: a device-side step, bit 4: a solve whose x the host needs
        auto turn = [&](uint32_t g) {
            BatchCtx &ctx = *grp[g];
            wants[g].store(0);
            const double t0 = BatchCtx::now();
            pool.run(lo[g + 1] - lo[g], [&](uint32_t k) {
                SlotLSI &inst        = *lsi[lo[g] + k];
                if (run_resident && ctx.is_resident[k]) return; // waits for the others to leave phase 1
                const bool served_fs = ctx.stage_fs && !ctx.skip[k], served_sens = ctx.stage_sens && ctx.skip[k] && ctx.objidx[k] >= 0;
                const bool has_spec  = served_fs && ctx.stage_sens && ctx.objidx[k] == 0; // its removal search ran right behind its l-QR
                if (run_step) ctx.mode()[k] = 0; // (the hook raises it again when the instance posts an iteration's equality problem)
                if (served_fs) ctx.take_solution(k);
                if (served_fs || served_sens) inst.advance();
                if (has_spec && !inst.finished() && inst.need() == SlotLSI::NEED_SENSITIVITY && inst.needLevel() == 0) inst.advance(); // step not blocked: use it
                const bool alive = !inst.finished();
                if (run_resident && alive && inst.atIterationSolve())
                {
                    // phase 1 is over and the equality problem of a regular iteration is staged: from here on the instance iterates on the
                    // device (its staged problem is served by the first resident stage)
                    ctx.hand_over(k, inst);
                    ctx.skip[k]   = 1;
                    ctx.objidx[k] = -1;
                    return;
                }
                const bool fs    = alive && inst.need() == SlotLSI::NEED_FACTORIZE_SOLVE;
                const bool se    = alive && inst.need() == SlotLSI::NEED_SENSITIVITY;
                const bool spec  = fs && ctx.spec_sens;
                ctx.skip[k]      = fs ? 0 : 1;
                ctx.objidx[k]    = se ? static_cast<int32_t>(inst.needLevel()) : (spec ? 0 : -1);
                const bool dstep = run_step && fs && ctx.mode()[k] != 0;
                const uint32_t w = (alive ? 1u : 0u) | (fs ? 2u : 0u) | ((se || spec) ? 4u : 0u) | (dstep ? 8u : 0u) | ((fs && !dstep) ? 16u : 0u);
                if (w & ~wants[g].load(std::memory_order_relaxed)) wants[g].fetch_or(w, std::memory_order_relaxed);
            });
            t_host += BatchCtx::now() - t0;
        };
        auto enqueue = [&](uint32_t g) -> bool { // false when no instance of the group is alive any more
            BatchCtx &ctx    = *grp[g];
            const uint32_t w = wants[g].load();
            ctx.stage_fs = ctx.stage_sens = false;
            if (!(w & 1u)) return false;
            if (!(w & 6u)) throw Exception("lexls_lsi_batch_solve: an instance is alive but requests nothing");
            ctx.enqueue_stage((w & 2u) != 0, (w & 4u) != 0, (w & 8u) != 0, (w & 16u) != 0, par.tol_wrong_sign_lambda, par.tol_correct_sign_lambda);
            return true;
        };
        auto finish = [&](uint32_t g) {
            grp[g]->finish_stage();
            turn(g);
        };

        std::vector<char> alive(nGroups, 0);
        bool any = false;
        for (uint32_t g = 0; g < nGroups; g++)
        {
            turn(g); // nothing served yet: only posts the first requests
            any = (alive[g] = enqueue(g)) || any;
        }
        while (any)
        {
            any = false;
            for (uint32_t g = 0; g < nGroups; g++)
                if (alive[g])
                {
                    finish(g);                // the other groups' stages keep the GPU busy meanwhile
                    alive[g] = enqueue(g);
                    any      = any || alive[g];
                }
        }

        if (run_resident)
        {
            std::vector<char> going(nGroups, 0);
            bool more = false;
            for (uint32_t g = 0; g < nGroups; g++)
            {
                BatchCtx &ctx  = *grp[g];
                ctx.n_resident = 0;
                for (uint32_t k = 0; k < ctx.B; k++)
                {
                    const bool r  = ctx.is_resident[k] != 0;
                    ctx.skip[k]   = r ? 0 : 1;
                    ctx.objidx[k] = r ? 0 : -1;
                    ctx.n_resident += r ? 1u : 0u;
                }
                if (ctx.n_resident)
                {
                    ctx.begin_resident();
                    going[g] = 1;
                    more     = true;
                }
            }
            // stages are enqueued in chunks; after each chunk ONE word comes back (instances that have stopped).  Stages past an
            // instance's end skip it in every kernel; a chunk that turns out not to be needed costs a few launches of early-exit kernels
            const int chunk = 8;
            bool freed = false;
            while (more)
            {
                more = false;
                for (uint32_t g = 0; g < nGroups; g++)
                    if (going[g]) grp[g]->enqueue_resident(chunk, par.tol_wrong_sign_lambda, par.tol_correct_sign_lambda, static_cast<int32_t>(par.max_number_of_factorizations));
                if (!freed) // the handed-over instances' host objects (a thousand LexLSI instances, dozens of vectors each) are not needed any
                {           // more: they are freed now, while the GPU works on the first chunk, instead of on the caller's time at the end
                    freed = true;
                    pool.run(batch, [&](uint32_t b) {
                        if (grp[group_of[b]]->is_resident[b - lo[group_of[b]]]) lsi[b].reset();
                    });
                }
                for (uint32_t g = 0; g < nGroups; g++)
                    if (going[g])
                    {
                        if (grp[g]->resident_done()) going[g] = 0;
                        more = more || going[g];
                    }
            }
            for (uint32_t g = 0; g < nGroups; g++)
                if (grp[g]->n_resident) grp[g]->download_resident();
        }
        if (run_step) // x and v of the instances whose state lives on the device
            for (uint32_t g = 0; g < nGroups; g++)
            {
                BatchCtx &ctx = *grp[g];
                if (hipMemcpyAsync(ctx.state_host.data(), ctx.d_state, 8 * (size_t)ctx.B * ctx.shape.SD, hipMemcpyDeviceToHost, ctx.stream) != hipSuccess ||
                    hipStreamSynchronize(ctx.stream) != hipSuccess)
                    throw Exception("download of the final state failed");
            }
        bool every_instance_resident = run_resident; // (then the job below is four small copies per instance)
        for (uint32_t g = 0; g < nGroups && every_instance_resident; g++)
            for (uint32_t k = 0; k < grp[g]->B && every_instance_resident; k++) every_instance_resident = grp[g]->is_resident[k] != 0;
        pool.run(batch, [&](uint32_t b) {
            if (run_resident)
            {
                BatchCtx &ctx    = *grp[group_of[b]];
                const uint32_t k = b - lo[group_of[b]];
                if (ctx.is_resident[k]) // x, v, working set and counters as the device left them (its host object is gone already)
                {
                    const double *st = ctx.rstate_host.data() + (size_t)k * ctx.rshape.SD;
                    std::copy(st, st + nVar, h_x + (size_t)b * nVar);
                    if (h_v) std::copy(st + nVar, st + nVar + total, h_v + (size_t)b * total);
                    if (h_active) std::copy(ctx.r_ctr_state(k), ctx.r_ctr_state(k) + total, h_active + (size_t)b * total);
                    if (h_info6) std::memcpy(h_info6 + (size_t)b * 6, ctx.r_info_of(k), 6 * sizeof(int32_t));
                    return;
                }
            }
            runner::LsiInfo info;
            runner::collect(*lsi[b], prob[b], h_x + (size_t)b * nVar, &info, h_active ? h_active + (size_t)b * total : NULL,
                            h_v ? h_v + (size_t)b * total : NULL);
            if (h_info6) std::memcpy(h_info6 + (size_t)b * 6, &info, sizeof(info));
            if (run_step)
            {
                BatchCtx &ctx    = *grp[group_of[b]];
                const uint32_t k = b - lo[group_of[b]];
                if (ctx.on_device[k])
                {
                    const double *st = ctx.state_host.data() + (size_t)k * ctx.shape.SD;
                    std::copy(st, st + nVar, h_x + (size_t)b * nVar);
                    if (h_v) std::copy(st + nVar, st + nVar + total, h_v + (size_t)b * total);
                }
            }
        }, every_instance_resident);
        int rounds_fs = 0, rounds_sens = 0, rounds_step = 0;
        double t_enq = 0.0, t_wait = 0.0;
        for (uint32_t g = 0; g < nGroups; g++)
        {
            rounds_fs += grp[g]->rounds_fs;
            rounds_sens += grp[g]->rounds_sens;
            rounds_step += grp[g]->rounds_step + grp[g]->rounds_resident;
            t_enq += grp[g]->t_enqueue;
            t_wait += grp[g]->t_wait;
        }
        if (std::getenv("LEXLS_LSI_TIMING") && run_resident) // prefix reuse: what the lock-step stages could and what a per-instance loop would save
        {
            long sumK = 0, cnt = 0, worstK = 0, worstN = -1;
            for (uint32_t g = 0; g < nGroups; g++)
                for (uint32_t k = 0; k < grp[g]->B; k++)
                    if (grp[g]->is_resident[k])
                    {
                        const int32_t *inf = grp[g]->r_info_of(k);
                        sumK += inf[6], cnt += inf[7];
                        if (inf[7] > worstN) worstN = inf[7], worstK = inf[6];
                    }
            std::fprintf(stderr, "lexls_lsi_batch_solve: prefix reuse: %ld resident factorizations behind a working-set change, %.2f levels read back on average; the instance with the most (%ld): %.2f\n",
                         cnt, cnt ? (double)sumK / cnt : 0.0, worstN, worstN > 0 ? (double)worstK / worstN : 0.0);
        }
        if (std::getenv("LEXLS_LSI_TIMING"))
            std::fprintf(stderr, "lexls_lsi_batch_solve: setup = %.4f s reset / constraint upload + %.4f s LexLSI objects (batch created in %.4f s)\n", t_ctx, t_setup - t_ctx, t_create),
            std::fprintf(stderr, "lexls_lsi_batch_solve: total %.4f s = setup %.4f + enqueue %.4f + wait for the GPU %.4f + host logic %.4f + rest %.4f (%u groups, %d+%d stages, %d with the step on the device)\n",
                         BatchCtx::now() - t_begin, t_setup, t_enq, t_wait, t_host, BatchCtx::now() - t_begin - t_setup - t_enq - t_wait - t_host, nGroups,
                         rounds_fs, rounds_sens, rounds_step);
        if (h_rounds2)
        {
            h_rounds2[0] = rounds_fs;
            h_rounds2[1] = rounds_sens;
        }
        last_stats[0] = rounds_fs, last_stats[1] = rounds_sens, last_stats[2] = rounds_step, last_stats[3] = (int32_t)nGroups;
        bool any_left = false;
        for (uint32_t b = 0; b < batch && !any_left; b++) any_left = lsi[b] != nullptr;
        if (any_left) pool.run(batch, [&](uint32_t b) { lsi[b].reset(); }); // a thousand LexLSI objects (dozens of vectors each): freed in parallel, not serially on return
    }
};

extern "C"
{
    void lexls_internal_set_error(const char *msg);

    int lexls_lsi_batch_solve(int device, uint32_t batch, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types,
                              const double *h_data, const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0,
                              const double *h_params9, double *h_x, int32_t *h_info6, uint8_t *h_active, double *h_v, int32_t *h_rounds2)
    {
        return lexls_lsi_batch_solve_ex(device, batch, nVar, nObj, h_dims, h_types, h_data, h_var_index, h_active_guess, h_x0, NULL, h_params9, 9, h_x,
                                        h_info6, h_active, h_v, h_rounds2);
    }

    int lexls_lsi_batch_create(lexls_lsi_batch_t *out, int device, uint32_t batch, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types)
    {
        try
        {
            if (!out || !h_dims || !h_types) throw Exception("lexls_lsi_batch_create: null argument");
            *out = new lexls_lsi_batch_s(device, batch, nVar, nObj, h_dims, h_types);
            return LEXLS_OK;
        }
        catch (const std::exception &e)
        {
            lexls_internal_set_error(e.what());
            return LEXLS_ERR_INVALID;
        }
    }

    int lexls_lsi_batch_stats(lexls_lsi_batch_t b, int32_t *h_stats4)
    {
        if (!b || !h_stats4)
        {
            lexls_internal_set_error("lexls_lsi_batch_stats: null argument");
            return LEXLS_ERR_INVALID;
        }
        std::memcpy(h_stats4, b->last_stats, sizeof(b->last_stats));
        return LEXLS_OK;
    }

    int lexls_lsi_batch_destroy(lexls_lsi_batch_t b)
    {
        delete b;
        return LEXLS_OK;
    }

    int lexls_lsi_batch_run(lexls_lsi_batch_t b, const double *h_data, const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0,
                            const double *h_v0, const double *h_reg_factors, const double *h_params, uint32_t nparams, double *h_x, int32_t *h_info6, uint8_t *h_active,
                            double *h_v, int32_t *h_rounds2)
    {
        try
        {
            if (!b) throw Exception("lexls_lsi_batch_run: null handle");
            if (h_params && nparams != 9 && nparams != 12) throw Exception("lexls_lsi_batch_solve_ex: 9 or 12 parameters expected");
            b->run(h_data, h_var_index, h_active_guess, h_x0, h_v0, h_reg_factors, unpack(h_params, nparams), h_x, h_info6, h_active, h_v, h_rounds2);
            return LEXLS_OK;
        }
        catch (const std::exception &e)
        {
            lexls_internal_set_error(e.what());
            return LEXLS_ERR_INVALID;
        }
    }

    int lexls_lsi_batch_solve_ex(int device, uint32_t batch, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types,
                                 const double *h_data, const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0,
                                 const double *h_reg_factors, const double *h_params, uint32_t nparams, double *h_x, int32_t *h_info6,
                                 uint8_t *h_active, double *h_v, int32_t *h_rounds2)
    {
        lexls_lsi_batch_t b = NULL;
        int rc              = lexls_lsi_batch_create(&b, device, batch, nVar, nObj, h_dims, h_types);
        if (rc == LEXLS_OK) rc = lexls_lsi_batch_run(b, h_data, h_var_index, h_active_guess, h_x0, NULL, h_reg_factors, h_params, nparams, h_x, h_info6, h_active, h_v, h_rounds2);
        lexls_lsi_batch_destroy(b);
        return rc;
    }

    int lexls_lsi_solve(int device, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types, const double *h_data,
                        const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0, const double *h_params9, double *h_x,
                        int32_t *h_info6, uint8_t *h_active, double *h_v)
    {
        try
        {
            runner::LsiProblem p = {nVar, nObj, h_dims, h_types, h_data, h_var_index, h_active_guess, h_x0};
            internal::LexLSI lsi;
            lsi.getLexLSE().setDevice(device);
            lsi.getLexLSE().setSensitivityScan(true); // the removal search of an iteration in one device call
            lsi.setSensitivityScansAllLevels(true);
            runner::setup(lsi, p, unpack(h_params9));
            lsi.solve();
            runner::LsiInfo info;
            runner::collect(lsi, p, h_x, &info, h_active, h_v);
            if (h_info6) std::memcpy(h_info6, &info, sizeof(info));
            return LEXLS_OK;
        }
        catch (const std::exception &e)
        {
            lexls_internal_set_error(e.what());
            return LEXLS_ERR_INVALID;
        }
    }

    int lexls_lsi_solve_ex(int device, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types, const double *h_data,
                           const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0, const double *h_v0,
                           const double *h_reg_factors, const double *h_params, uint32_t nparams, double *h_x, int32_t *h_info6, uint8_t *h_active,
                           double *h_v)
    {
        try
        {
            if (h_params && nparams != 9 && nparams != 12) throw Exception("lexls_lsi_solve_ex: 9 or 12 parameters expected");
            runner::LsiProblem p = {nVar, nObj, h_dims, h_types, h_data, h_var_index, h_active_guess, h_x0, h_v0, h_reg_factors};
            internal::LexLSI lsi;
            lsi.getLexLSE().setDevice(device);
            lsi.getLexLSE().setSensitivityScan(true); // the removal search of an iteration in one device call
            lsi.setSensitivityScansAllLevels(true);
            runner::setup(lsi, p, unpack(h_params, nparams));
            lsi.solve();
            runner::LsiInfo info;
            runner::collect(lsi, p, h_x, &info, h_active, h_v);
            if (h_info6) std::memcpy(h_info6, &info, sizeof(info));
            return LEXLS_OK;
        }
        catch (const std::exception &e)
        {
            lexls_internal_set_error(e.what());
            return LEXLS_ERR_INVALID;
        }
    }

    int lexls_lsi_solve_debug(int device, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types, const double *h_data,
                              const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0, const double *h_v0,
                              const double *h_reg_factors, const double *h_params, uint32_t nparams, double *h_x, int32_t *h_info6, uint8_t *h_active,
                              double *h_v, const lexls_lsi_debug *debug)
    {
        try
        {
            if (h_params && nparams != 9 && nparams != 12) throw Exception("lexls_lsi_solve_debug: 9 or 12 parameters expected");
            if (!debug) throw Exception("lexls_lsi_solve_debug: debug is NULL (use lexls_lsi_solve_ex)");
            runner::LsiProblem p = {nVar, nObj, h_dims, h_types, h_data, h_var_index, h_active_guess, h_x0, h_v0, h_reg_factors};
            ParametersLexLSI par        = unpack(h_params, nparams);
            par.log_working_set_enabled = true;
            internal::LexLSI lsi;
            lsi.getLexLSE().setDevice(device);
            lsi.getLexLSE().setSensitivityScan(true);
            lsi.setSensitivityScansAllLevels(true);
            runner::setup(lsi, p, par);
            lsi.solve();
            runner::LsiInfo info;
            runner::collect(lsi, p, h_x, &info, h_active, h_v);
            if (h_info6) std::memcpy(h_info6, &info, sizeof(info));
            const runner::LsiDebug d = {debug->lambda, debug->lexqr, debug->data, debug->x_star, debug->active_ctr, debug->log, debug->log_alpha, debug->max_log,
                                        debug->x_mu, debug->x_mu_rhs, debug->residual_mu, debug->counts};
            runner::collect_debug(lsi, p, par, d);
            return LEXLS_OK;
        }
        catch (const std::exception &e)
        {
            lexls_internal_set_error(e.what());
            return LEXLS_ERR_INVALID;
        }
    }

    int lexls_lsi_solve_dat(int device, const char *path, int one_based, int use_active_guess, int use_x_guess, double *h_x, int32_t *h_info6,
                            double *h_solution)
    {
        try
        {
            tools::Hierarchy h;
            tools::HierarchyFileProcessor().import(path, h);
            runner::FlatHierarchy f;
            runner::flatten(h, one_based != 0, use_active_guess != 0, use_x_guess != 0, f);
            internal::LexLSI lsi;
            lsi.getLexLSE().setDevice(device);
            lsi.getLexLSE().setSensitivityScan(true); // the removal search of an iteration in one device call
            lsi.setSensitivityScansAllLevels(true);
            runner::setup(lsi, f.problem, ParametersLexLSI());
            lsi.solve();
            runner::LsiInfo info;
            runner::collect(lsi, f.problem, h_x, &info, NULL, NULL);
            if (h_info6) std::memcpy(h_info6, &info, sizeof(info));
            if (h_solution)
                for (Index i = 0; i < h.solution.size(); i++) h_solution[i] = h.solution(i);
            return LEXLS_OK;
        }
        catch (const std::exception &e)
        {
            lexls_internal_set_error(e.what());
            return LEXLS_ERR_INVALID;
        }
    }
}
