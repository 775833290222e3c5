// C ABI (include/ccv_mppi.h) over the gfx950 kernels of mppi_kernels.h.
// Host side only orchestrates: allocate once, build the window coefficients, launch, copy u* back.
// There is deliberately no CPU fallback: every entry point fails with CCV_MPPI_ERR_NO_DEVICE / _HIP.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "fast_trig.h"
#include "mppi_kernels.h"
#include "mppi_launch.h"
#include "mppi_update.h"
#include "mppi_resident.h"

using namespace ccv;

struct ccv_mppi_handle {
    ccv_mppi_config cfg{};
    int udim = 0, K = 0, H = 0, R = 0, pitch = 0, nchunks = 0, nblocks = 0;
    int nparts_last = 0;   // number of partial columns the last cost evaluation produced (fused: workgroups, else: chunks)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // device buffers
    double* d_nominal = nullptr;
    const double* pending_vec = nullptr;   // deferred apply_partials: u* = pending_vec[1..] / pending_vec[0] (see flush_pending)
    // device-resident loop: the update of a tick is launched together with the next tick's prologue (k_finalize_advance);
    // anything else that needs u* / the statistics first gets a plain k_finalize (flush_pending)
    bool fin_pending = false;
    FinalizeArgs fin_args{};
    double* d_u = nullptr;
    void* d_arena = nullptr;           // one allocation behind u, z, xs, ys, cost, w, partial (2 MB-aligned pieces)
    float* d_z = nullptr;              // the fused iteration stores the normals in place of the controls (mppi_kernels.h)
    double* d_nom_used = nullptr;      // ... and the warm start they were drawn around
    bool controls_in_z = false;        // d_u is stale: the controls of the last iteration are (d_z, d_nom_used)
    double* d_xs = nullptr;
    double* d_ys = nullptr;
    double* d_cost = nullptr;
    double* d_w = nullptr;
    double* d_partial = nullptr;
    double* d_statpart = nullptr;
    double* d_vec = nullptr;
    double* d_stats = nullptr;
    double* d_cmin = nullptr;
    unsigned long long* d_dbg = nullptr;   // -DCCV_DIAG builds only (mppi_diag.h): the kernels' stamp buffer
    // device-resident closed loop (mppi_resident.h)
    ResidentFrame* d_frame = nullptr;
    double* d_path = nullptr;    // [2][n_path]: x then y
    double* d_trace = nullptr;   // [kTraceRows][6]
    static constexpr int kTraceRows = 8192;
    int n_path = 0;
    double path_resolution = 0.0;
    bool have_pose = false;
    int64_t res_steps = 0;                 // k_advance launches since the pose was set
    double res_angle_abs[3] = {0, 0, 0};   // conservative bounds on |yaw|, |roll|, |pitch| of the resident pose (fast_trig_safe)
    // direct exchange of the partial vectors between the devices of a node (k_finalize_exchange, mppi_kernels.h)
    ExchangeBox* d_box = nullptr;                   // this device's box (peers write into it)
    ExchangeBox* box_peer[kMaxRanks] = {nullptr};   // every rank's box as mapped here ([xchg_rank] = d_box)
    bool box_opened[kMaxRanks] = {false};           // mapped with hipIpcOpenMemHandle (to be closed)
    double* d_xvec = nullptr;                       // reduced [sum w, sum w*u]
    int32_t* h_xflag = nullptr;                     // "a peer timed out" flag: pinned, host-mapped memory the update kernel writes
    int32_t* d_xflag = nullptr;                     // ... and its device address (sticky until the exchange is released)
    double xchg_timeout_s = 10.0;                   // (what the message says)
    int xchg_world = 0, xchg_rank = 0;
    bool xchg_connected = false;
    bool box_fine_grained = false;                  // the box is fine-grained (device-coherent) memory
    uint32_t xchg_nonce = 0;                        // this rank's contribution to the sequence base (rank 0's is used)
    uint32_t xchg_base = 0;                         // sequence numbers start here: a restarted job does not match old packets
    unsigned long long xchg_seq = 0;
    unsigned long long xchg_timeout_ticks = 1000000000ull;   // 10 s of the 100 MHz clock (CCV_MPPI_EXCHANGE_TIMEOUT_MS: tests)
    // queue-depth throttle for the asynchronous entry points: beyond a few dozen iterations in flight the HIP runtime's
    // enqueue path slows down several-fold (measured: 12 us/call at depth <= 64, 90 us/call at depth 512), so every
    // kThrottleEvery-th enqueue records an event and waits for the one recorded kThrottleSlots marks earlier
    static constexpr int kThrottleEvery = 16, kThrottleSlots = 3;
    hipEvent_t throttle_ev[kThrottleSlots] = {nullptr, nullptr, nullptr};
    bool throttle_used[kThrottleSlots] = {false, false, false};
    uint64_t enqueued = 0;
    bool throttle = true;   // CCV_MPPI_THROTTLE=0 disables (experiments)
    double* d_scratch = nullptr;  // read-back staging
    size_t scratch_bytes = 0;
    // pinned host staging
    double* h_pin = nullptr;
    size_t pin_doubles = 0;
    // result mailbox of the blocking calls (FinalizeArgs::mail): pinned host-mapped memory the update kernel writes
    unsigned long long* h_mail = nullptr;
    unsigned long long* d_mail = nullptr;   // its device address
    uint32_t mail_seq = 0;
    bool want_mail = false;      // the next plain k_finalize launch posts its result (set by the blocking entry points)
    bool mail_pending = false;   // ... and that launch is in flight: fetch_result() polls the mailbox
    bool use_mail = true;        // CCV_MPPI_MAILBOX=0: copy + stream synchronisation instead (experiments)
    // stage-wise state
    bool have_controls = false, have_rollout = false, have_weights = false;
    double st_x0[5] = {0, 0, 0, 0, 0};
    double st_dt = 0.1;
    // kernel selection (experiments): CCV_MPPI_KERNEL=v1 -> one-sample-per-lane k_rollout_cost,
    // CCV_MPPI_WINDOW=scalar -> its scalar-load window variant; default = k_rollout_pc
    int lds_window = 1;
    int coop = 1;
    bool solo = false;   // fused iterations run k_rollout_solo (one wave per 64 samples) instead of coop's kernel
    bool wide_turn = false;   // this launch: diff drive beyond |w|max dt = pi/4 -> the full-range sin / cos instantiation
    bool fast_clamp_allowed = true;   // clampd_fast (mppi_kernels.h) unless CCV_MPPI_FAST_CLAMP=0
    int prio_rotate = 0, cu_count = 256;   // pc_rotate_priority (mppi_rollout_pc.h)
    int prune = 0;                         // pc_prune_window (mppi_rollout_pc.h)
    double inj_absmax[CCV_MPPI_MAX_UDIM] = {0, 0, 0, 0, 0};   // largest |control| per dimension in the buffer (sampled: clamp bound)
    double nom_absmax[CCV_MPPI_MAX_UDIM] = {0, 0, 0, 0, 0};   // largest |u*| per dimension a caller has put there (ccv_mppi_set_nominal)
    // timing
    bool timing = false;
    int timing_every = 1;     // record events on every n-th iteration only
    int64_t timing_count = 0;
    std::vector<hipEvent_t> ev;  // triples: rollout kernel begin, rollout kernel end, end of the launch sequence
    hipEvent_t ev_kernel_start = nullptr, ev_kernel_stop = nullptr;   // set for the duration of a timed launch
    size_t ev_used = 0;
    double t_roll_sum = 0.0, t_iter_sum = 0.0;
    int64_t t_n = 0;
    float last_iter_us = 0.f, last_roll_us = 0.f;
    std::string err;
};

namespace {

const char* kVersion = "ccv_mppi_hip 0.1 (gfx950)";

int fail(ccv_mppi_handle* h, int code, const char* what, hipError_t e = hipSuccess) {
    if (h) {
        h->err = what;
        if (e != hipSuccess) {
            h->err += ": ";
            h->err += hipGetErrorString(e);
        }
    }
    return code;
}

#define HIP_TRY(h, call)                                                            \
    do {                                                                            \
        hipError_t e__ = (call);                                                    \
        if (e__ != hipSuccess) return fail((h), CCV_MPPI_ERR_HIP, #call, e__);      \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// The caller's current device is put back when an entry point that had to switch to the handle's device returns (on error
// paths too): a process that drives several devices must not find its current device changed behind its back.
struct DeviceGuard {
    int prev = -1, mine = -1;
    explicit DeviceGuard(int device) : mine(device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != mine) (void)hipSetDevice(prev);
    }
};

int ensure_scratch(ccv_mppi_handle* h, size_t bytes) {
    if (bytes <= h->scratch_bytes) return CCV_MPPI_OK;
    if (h->d_scratch) HIP_TRY(h, hipFree(h->d_scratch));
    h->d_scratch = nullptr;
    h->scratch_bytes = 0;
    HIP_TRY(h, hipMalloc(&h->d_scratch, bytes));
    h->scratch_bytes = bytes;
    return CCV_MPPI_OK;
}

void fill_args(const ccv_mppi_handle* h, RolloutArgs& A, const double* x0, double dt, double yaw_ref0, uint64_t seed,
               uint64_t iter) {
    const ccv_mppi_config& c = h->cfg;
    std::memset(&A, 0, sizeof(A));
    const int nx = c.model == CCV_MPPI_FULL_BODY ? 5 : 3;
    for (int i = 0; i < nx; ++i) A.x0[i] = x0[i];
    A.dt = dt;
    A.yaw_ref0 = yaw_ref0;
    A.sigma = c.control_noise;
    A.lambda = c.lambda;
    A.v_ref = c.v_ref;
    for (int d = 0; d < CCV_MPPI_MAX_UDIM; ++d) {
        A.umin[d] = c.u_min[d];
        A.umax[d] = c.u_max[d];
    }
    // two-instruction clamps (clampd_fast, mppi_kernels.h) give the reference's values when no control is NaN and no pair of
    // bounds is out of order; the kernel adds its own test of the warm start
    A.fast_clamp = std::isfinite(c.control_noise) ? 1 : 0;
    for (int d = 0; d < udim_of(c.model); ++d)
        if (!(c.u_min[d] <= c.u_max[d])) A.fast_clamp = 0;
    if (!h->fast_clamp_allowed) A.fast_clamp = 0;   // (CCV_MPPI_FAST_CLAMP=0: tests, experiments)
    const bool roll_off = (c.flags & CCV_MPPI_FLAG_ROLL_OFF) != 0;
    A.w_path = c.path_weight;
    A.w_v = c.v_weight;
    A.w_zmp = roll_off ? 0.0 : c.zmp_weight;        // fb:43-46
    A.w_rollv = roll_off ? 0.0 : c.roll_v_weight;
    A.w_back = c.back_weight;
    A.w_yaw = c.yaw_weight;
    // fb.h:212-216, fb:86-91
    const double upper_body_height = 0.8075, upper_body_width = 0.208, mass = 60.0;
    const double base2CoM = upper_body_height / 2;
    A.fb_mass = mass;
    A.fb_L = base2CoM;
    A.fb_Ixx = (mass * (upper_body_width * upper_body_width + upper_body_height * upper_body_height)) / 12 + mass * base2CoM * base2CoM;
    A.fb_gz = -9.8;  // fb.h:30
    A.inv_dt = 1.0 / dt;
    A.seed_lo = (uint32_t)seed;
    A.seed_hi = (uint32_t)(seed >> 32);
    A.iter_lo = (uint32_t)iter;
    A.iter_hi = (uint32_t)(iter >> 32);
    A.K = h->K;
    A.pitch = h->pitch;
    A.H = h->H;
    A.k_offset = c.sample_offset;
    A.steer_off = (c.flags & CCV_MPPI_FLAG_STEER_OFF) ? 1 : 0;
    A.nominal = h->d_nominal;
    A.u = h->d_u;
    A.z = h->d_z;
    A.nominal_used = h->d_nom_used;
    A.xs = h->d_xs;
    A.ys = h->d_ys;
    A.cost = h->d_cost;
    A.w = h->d_w;
    A.partial = h->d_partial;
    A.statpart = h->d_statpart;
    A.nparts = h->nblocks;
    A.fuse_update = 0;
    A.prio_rotate = h->prio_rotate;
    A.pending_vec = nullptr;
    A.nominal_w = h->d_nominal;
    A.stats_w = h->d_stats;
    A.cu_count = h->cu_count;
    A.prune = h->prune;
    A.dbg = h->d_dbg;
}

// Window coefficients relative to the current pose: |p - r_j|^2 = |p|^2 + a_j px + b_j py + c_j
void fill_window(const ccv_mppi_handle* h, Window& W, const double* x0, const double* x_ref, const double* y_ref) {
    for (int j = 0; j < h->H; ++j) {
        const double xl = x_ref[j] - x0[0], yl = y_ref[j] - x0[1];
        W.a[j] = -2.0 * xl;
        W.b[j] = -2.0 * yl;
        W.c[j] = xl * xl + yl * yl;
    }
}

// mode: MODE_FUSED / MODE_ROLLOUT / MODE_COST (mppi_rollout_pc.h).  The kernels live in translation units of their own
// (mppi_launch.h); a timed fused launch carries its events on the dispatch itself.
void launch_rollout_model(const ccv_mppi_handle* h, const RolloutArgs& A, const Window& W, int mode) {
    const int model = h->cfg.model;
    const bool timed = mode == MODE_FUSED && h->ev_kernel_start;
    const LaunchAt at{h->stream, timed ? h->ev_kernel_start : nullptr, timed ? h->ev_kernel_stop : nullptr};
    if (h->solo && h->coop && mode == MODE_FUSED) {
        // one wave per 64 samples (mppi_rollout_solo.h): K provides two or more such waves per SIMD
        launch_rollout_solo(model, h->wide_turn, at, A, W);
        return;
    }
    if (h->coop == 3) {   // four-wave kernel (mppi_rollout_r4.h)
        launch_rollout_r4(model, mode, h->wide_turn, at, A, W);
        return;
    }
    if (model != CCV_MPPI_FULL_BODY && h->coop == 2) {   // three-wave kernel (mppi_rollout_r3.h); not built for full body
        launch_rollout_r3(model, mode, at, A, W);
        return;
    }
    if (h->coop) {   // two-wave kernel (mppi_rollout_pc.h)
        launch_rollout_pc(model, mode, at, A, W);
        return;
    }
    // plain one-sample-per-lane kernel: the path of unbounded headings (fast_trig_safe) and CCV_MPPI_KERNEL=v1
    // (timed launch: events recorded around it -- the fallback must deliver kernel times too)
    const LaunchAt plain{h->stream, nullptr, nullptr};
    if (mode == MODE_FUSED && h->ev_kernel_start) (void)hipEventRecord(h->ev_kernel_start, h->stream);
    launch_rollout_plain(model, mode == MODE_FUSED, h->lds_window != 0, plain, A, W);
    if (mode == MODE_FUSED && h->ev_kernel_stop) (void)hipEventRecord(h->ev_kernel_stop, h->stream);
}

// A deferred ccv_mppi_apply_partials_enqueue is normally consumed by the next fused rollout launch (pc_stage_nominal);
// anything else that reads the warm start first gets it materialised here.
int flush_pending(ccv_mppi_handle* h) {
    if (h->fin_pending) {
        hipLaunchKernelGGL(k_finalize, dim3(finalize_blocks(h->fin_args.R)), dim3(kBlock), 0, h->stream, h->fin_args);
        h->fin_pending = false;
        HIP_TRY(h, hipGetLastError());
    }
    if (!h->pending_vec) return CCV_MPPI_OK;
    hipLaunchKernelGGL(k_apply_partials, dim3(1), dim3(kBlock), 0, h->stream, h->pending_vec, h->d_nominal, h->d_stats, h->R);
    h->pending_vec = nullptr;
    HIP_TRY(h, hipGetLastError());
    return CCV_MPPI_OK;
}

// The fused kernels store the normals, not the controls; whoever needs the controls as an array (the stage-wise calls after a
// fused iteration, the unfused update) gets them re-derived into d_u first: the samplers' arithmetic, the same bits.
int materialize_controls(ccv_mppi_handle* h) {
    if (!h->controls_in_z) return CCV_MPPI_OK;
    MaterializeArgs M;
    M.z = h->d_z;
    M.nominal_used = h->d_nom_used;
    M.u = h->d_u;
    M.sigma = h->cfg.control_noise;
    for (int d = 0; d < CCV_MPPI_MAX_UDIM; ++d) {
        M.umin[d] = h->cfg.u_min[d];
        M.umax[d] = h->cfg.u_max[d];
    }
    M.K = h->K;
    M.pitch = h->pitch;
    M.R = h->R;
    M.udim = h->udim;
    M.zero_dim = (h->cfg.model == CCV_MPPI_FULL_BODY && (h->cfg.flags & CCV_MPPI_FLAG_STEER_OFF)) ? 2 : -1;
    hipLaunchKernelGGL(k_materialize_controls, dim3((h->K + kBlock - 1) / kBlock, h->R), dim3(kBlock), 0, h->stream, M);
    HIP_TRY(h, hipGetLastError());
    h->controls_in_z = false;
    return CCV_MPPI_OK;
}

// The cooperative kernels use a branch-free sin/cos that is valid for |angle| <= kFastTrigLimit.  Every heading a sample can
// reach is bounded by the start angle plus (H-1) steps at the largest control magnitude, so the decision is made here,
// once per call; anything else (huge or non-finite angles, unbounded injected controls) runs the plain
// one-sample-per-lane kernel with OCML's sincos.
// Returns kTrigUnsafe (plain kernel), kTrigSafe, or kTrigWide: diff drive, fused iteration, four-wave or one-wave kernel,
// every heading inside the range but a turn per step beyond pi/4 -- the instantiation that evaluates sin / cos of every
// heading in full (as the steering model's does) instead of advancing them by the step's turn.
enum : int { kTrigUnsafe = 0, kTrigSafe = 1, kTrigWide = 2 };
int fast_trig_safe(const ccv_mppi_handle* h, const RolloutArgs& A, int mode) {
    const int ud = h->udim;
    double umax[CCV_MPPI_MAX_UDIM];
    for (int d = 0; d < ud; ++d) {
        umax[d] = mode == MODE_FUSED ? std::fmax(std::fabs(h->cfg.u_min[d]), std::fabs(h->cfg.u_max[d])) : h->inj_absmax[d];
    }
    const double steps = (double)(h->H - 1) * std::fabs(A.dt);
    double bound = std::fabs(A.x0[2]) + steps * umax[1];
    if (h->cfg.model != CCV_MPPI_DIFF_DRIVE) bound += umax[2];
    if (h->cfg.model == CCV_MPPI_FULL_BODY) {
        bound = std::fmax(bound, std::fabs(A.x0[3]) + steps * umax[3]);
        bound = std::fmax(bound, std::fabs(A.x0[4]) + steps * umax[4]);
    }
    // diff drive advances (sin, cos) of the heading by the step's turn angle: needs |w| dt <= pi/4 (fast_trig.h)
    bool wide = false;
    if (h->cfg.model == CCV_MPPI_DIFF_DRIVE && !(umax[1] * std::fabs(A.dt) <= kSmallTurnLimit)) {
        if (!(mode == MODE_FUSED && (h->coop == 3 || h->solo))) return kTrigUnsafe;   // (the stage-wise and the experiment kernels have no wide form)
        wide = true;
    }
    // full body: the same for yaw, roll and pitch, and the direction angle itself is evaluated without range reduction
    if (h->cfg.model == CCV_MPPI_FULL_BODY) {
        if (!(umax[1] * std::fabs(A.dt) <= kSmallTurnLimit) || !(umax[3] * std::fabs(A.dt) <= kSmallTurnLimit) ||
            !(umax[4] * std::fabs(A.dt) <= kSmallTurnLimit) || !(umax[2] <= kSmallTurnLimit))
            return kTrigUnsafe;
        // ... and divides by dt through its reciprocal (div_uniform, mppi_kernels.h): nothing may overflow or vanish on the way
        if (!(std::fabs(A.dt) >= 1.0e-100 && std::fabs(A.dt) <= 1.0e100) || !(umax[0] <= 1.0e100) || !(umax[3] <= 1.0e100)) return kTrigUnsafe;
    }
    if (!(bound <= kFastTrigLimit)) return kTrigUnsafe;   // (also for NaN)
    return wide ? kTrigWide : kTrigSafe;
}

int launch_rollout(ccv_mppi_handle* h, const RolloutArgs& A_in, const Window& W, int mode) {
    RolloutArgs A = A_in;
    const int saved = h->coop;
    const int trig = h->coop ? fast_trig_safe(h, A, mode) : kTrigUnsafe;
    if (h->coop && trig == kTrigUnsafe) h->coop = 0;
    h->wide_turn = trig == kTrigWide;
    struct Restore { ccv_mppi_handle* h; int v; ~Restore() { h->coop = v; h->wide_turn = false; } } restore{h, saved};
    // the production kernel also reduces its workgroup's share of sum w and sum w*u (no second pass over the controls);
    // the underflow-safe MIN_SHIFT mode needs the global minimum first and keeps the separate update kernels
    A.fuse_update = (h->coop && mode != MODE_ROLLOUT && !(h->cfg.flags & CCV_MPPI_FLAG_MIN_SHIFT)) ? 1 : 0;
    if (h->fin_pending) {   // (the kernel reads the warm start)
        const double* keep = h->pending_vec;
        h->pending_vec = nullptr;
        int rc = flush_pending(h);
        h->pending_vec = keep;
        if (rc) return rc;
    }
    if (h->pending_vec) {
        if (h->coop && mode == MODE_FUSED) {   // the kernel divides while it stages u* (and writes it back)
            A.pending_vec = h->pending_vec;
            h->pending_vec = nullptr;
        } else {
            int rc = flush_pending(h);
            if (rc) return rc;
        }
    }
    if (mode != MODE_ROLLOUT) h->nparts_last = A.fuse_update ? h->nblocks : 0;
    if (mode != MODE_FUSED) {   // the stage-wise kernels read the controls as an array
        if (int rc = materialize_controls(h)) return rc;
    }
    launch_rollout_model(h, A, W, mode);
    HIP_TRY(h, hipGetLastError());
    if (mode == MODE_FUSED) h->controls_in_z = h->coop != 0;   // (the plain kernel writes u itself)
    return CCV_MPPI_OK;
}

int launch_sample(ccv_mppi_handle* h, const RolloutArgs& A) {
    if (int rc = flush_pending(h)) return rc;
    launch_sample(h->cfg.model, h->stream, A);
    HIP_TRY(h, hipGetLastError());
    return CCV_MPPI_OK;
}

// weights -> [sum w, sum w*u] (-> u* when `normalise`); vec_out may be a caller-owned device buffer.
int launch_update(ccv_mppi_handle* h, bool normalise, double* vec_out, bool exchange = false, bool defer = false) {
    if (h->fin_pending) {
        if (int rc = flush_pending(h)) return rc;
    }
    int nparts = h->nparts_last;
    if (nparts == 0) {
        // not fused (one-sample-per-lane fallback kernel or MIN_SHIFT): stream w and the controls once more
        if (int rc = materialize_controls(h)) return rc;
        if (h->cfg.flags & CCV_MPPI_FLAG_MIN_SHIFT) {
            hipLaunchKernelGGL(k_min_cost, dim3(1), dim3(1024), 0, h->stream, h->d_cost, h->K, h->d_cmin);
            hipLaunchKernelGGL(k_reweight, dim3((h->K + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, h->d_cost, h->d_cmin,
                               h->cfg.lambda, h->K, h->d_w);
        }
        UpdateArgs U;
        U.u = h->d_u;
        U.w = h->d_w;
        U.cost = h->d_cost;
        U.partial = h->d_partial;
        U.statpart = h->d_statpart;
        U.K = h->K;
        U.pitch = h->pitch;
        U.R = h->R;
        U.nchunks = h->nchunks;
        hipLaunchKernelGGL(k_update_partials, dim3(h->nchunks, h->R + 1), dim3(kBlock), 0, h->stream, U);
        nparts = h->nchunks;
    }
    FinalizeArgs F;
    F.partial = h->d_partial;
    F.statpart = h->d_statpart;
    F.nominal = h->d_nominal;
    F.vec = vec_out ? vec_out : h->d_vec;
    F.stats = h->d_stats;
    F.R = h->R;
    F.nchunks = nparts;
    F.normalise = normalise ? 1 : 0;
    F.mail = nullptr;
    F.mail_seq = 0;
    const bool post = h->want_mail && h->use_mail && normalise && !exchange && !defer && h->d_mail;
    h->want_mail = false;
    if (post) {
        if (++h->mail_seq == 0) h->mail_seq = 1;
        F.mail = h->d_mail;
        F.mail_seq = h->mail_seq;
        h->mail_pending = true;
    }
    if (exchange) {
        ExchangeArgs X;
        for (int r = 0; r < kMaxRanks; ++r) X.peer[r] = h->box_peer[r];
        X.local = h->d_box;
        X.reduced = h->d_xvec;
        ++h->xchg_seq;
        X.seq = (uint32_t)(((unsigned long long)h->xchg_base + h->xchg_seq) % 0xFFFFFFFFull) + 1u;   // 1 .. 2^32-1, never 0 (the box starts zeroed)
        X.timeout_flag = h->d_xflag;
        X.world = h->xchg_world;
        X.rank = h->xchg_rank;
        X.parity = (int)(h->xchg_seq & 1);
        X.timeout_ticks = h->xchg_timeout_ticks;   // a peer that never arrives yields NaN and a flag, not a hang
        hipLaunchKernelGGL(k_finalize_exchange, dim3(finalize_blocks(h->R)), dim3(kBlock), 0, h->stream, F, X);
        HIP_TRY(h, hipGetLastError());
        h->pending_vec = h->d_xvec;   // u* = reduced[1..] / reduced[0]: deferred like ccv_mppi_apply_partials_enqueue
        return CCV_MPPI_OK;
    }
    if (defer && nparts == h->nblocks && normalise) {   // fused partials, plain update: launched with the next tick's prologue
        h->fin_args = F;
        h->fin_pending = true;
        return CCV_MPPI_OK;
    }
    hipLaunchKernelGGL(k_finalize, dim3(finalize_blocks(h->R)), dim3(kBlock), 0, h->stream, F);
    HIP_TRY(h, hipGetLastError());
    return CCV_MPPI_OK;
}

int timing_begin(ccv_mppi_handle* h, size_t& slot) {
    slot = h->ev_used;
    if (h->ev.size() < slot + 3) {
        for (int i = 0; i < 3; ++i) {
            hipEvent_t e;
            HIP_TRY(h, hipEventCreate(&e));
            h->ev.push_back(e);
        }
    }
    h->ev_used += 3;
    return CCV_MPPI_OK;
}

int timing_collect(ccv_mppi_handle* h) {
    if (h->ev_used == 0) return CCV_MPPI_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (size_t s = 0; s + 3 <= h->ev_used; s += 3) {
        float a = 0.f, b = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&a, h->ev[s], h->ev[s + 1]));
        HIP_TRY(h, hipEventElapsedTime(&b, h->ev[s], h->ev[s + 2]));
        h->t_roll_sum += (double)a * 1000.0;
        h->t_iter_sum += (double)b * 1000.0;
        h->t_n += 1;
        h->last_roll_us = a * 1000.f;
        h->last_iter_us = b * 1000.f;
    }
    h->ev_used = 0;
    return CCV_MPPI_OK;
}

int check_iter_args(ccv_mppi_handle* h, const double* x0, double dt, const double* x_ref, const double* y_ref) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (!x0 || !x_ref || !y_ref) return fail(h, CCV_MPPI_ERR_INVALID_ARG, "null pointer argument");
    if (!(dt == dt)) return fail(h, CCV_MPPI_ERR_INVALID_ARG, "dt is NaN");
    return CCV_MPPI_OK;
}

// the fused iteration: sample+rollout+cost kernel, then the weighted update
// (resident: the pose and the window are taken from h->d_frame on the device; x0 then carries only the bounds on the pose
//  angles that fast_trig_safe() needs, and x_ref / y_ref are not read)
int enqueue_iteration(ccv_mppi_handle* h, const double* x0, double dt, const double* x_ref, const double* y_ref,
                      double yaw_ref0, uint64_t seed, uint64_t iter, bool normalise, double* vec_out, bool resident = false,
                      bool exchange = false) {
    RolloutArgs A;
    Window W;
    fill_args(h, A, x0, dt, yaw_ref0, seed, iter);
    if (resident) {
        std::memset(&W, 0, sizeof(W));
        A.frame = h->d_frame;
        if (!h->coop || fast_trig_safe(h, A, MODE_FUSED) == kTrigUnsafe)
            return fail(h, CCV_MPPI_ERR_STATE, "the resident loop needs the cooperative kernels and bounded pose angles");
    } else {
        fill_window(h, W, x0, x_ref, y_ref);
    }
    A.store_u = 1;
    A.store_xy = (h->cfg.flags & CCV_MPPI_FLAG_NO_STATE_STORE) ? 0 : 1;
    A.do_cost = 1;
    size_t slot = 0;
    const bool timed = h->timing && (h->timing_count++ % h->timing_every) == 0;
    if (timed) {
        int rc = timing_begin(h, slot);
        if (rc) return rc;
    }
    if (timed) {
        h->ev_kernel_start = h->ev[slot];
        h->ev_kernel_stop = h->ev[slot + 1];
    }
    int rc = launch_rollout(h, A, W, MODE_FUSED);
    h->ev_kernel_start = h->ev_kernel_stop = nullptr;
    if (rc) return rc;
    rc = launch_update(h, normalise, vec_out, exchange, /*defer=*/resident && normalise && !vec_out && !exchange && !timed);
    if (rc) return rc;
    if (timed) HIP_TRY(h, hipEventRecord(h->ev[slot + 2], h->stream));
    if (h->throttle && ++h->enqueued % ccv_mppi_handle::kThrottleEvery == 0) {
        const int ts = (int)((h->enqueued / ccv_mppi_handle::kThrottleEvery) % ccv_mppi_handle::kThrottleSlots);
        if (h->throttle_used[ts]) HIP_TRY(h, hipEventSynchronize(h->throttle_ev[ts]));
        HIP_TRY(h, hipEventRecord(h->throttle_ev[ts], h->stream));
        h->throttle_used[ts] = true;
    }
    std::memcpy(h->st_x0, A.x0, sizeof(h->st_x0));
    h->st_dt = dt;
    h->have_controls = h->have_rollout = h->have_weights = true;
    return CCV_MPPI_OK;
}

// The blocking calls' result: the update kernel has been told to post u* and the statistics into the pinned mailbox
// (FinalizeArgs::mail); poll until every packet carries this call's sequence number.  Costs the PCIe write latency after the
// kernel's last store instead of two copy-engine transfers and a stream synchronisation (C2: 72 -> ~50 us per blocking
// iteration; the reference defaults, K = 1 000, H = 15: 34 -> ~25 us).  A kernel that never posts (a fault, a lost device)
// is found by the stream query / synchronisation the poll falls back to, so the call returns an error instead of spinning.
int wait_mail(ccv_mppi_handle* h, const size_t n_slots) {
    const uint32_t seq = h->mail_seq;
    volatile unsigned long long* m = h->h_mail;
    const auto t0 = std::chrono::steady_clock::now();
    size_t next = 0;
    bool synced = false;
    for (unsigned spin = 0;; ++spin) {
        while (next < n_slots && (uint32_t)m[2 * next] == seq && (uint32_t)m[2 * next + 1] == seq) ++next;
        if (next == n_slots) return CCV_MPPI_OK;
        if (synced) return fail(h, CCV_MPPI_ERR_HIP, "the update kernel finished without posting its result");
        asm volatile("" ::: "memory");
        if ((spin & 1023u) == 1023u) {
            // a long kernel (K in the millions) or a stuck one: stop burning a core after a millisecond and let the runtime wait
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > 1.0e-3) {
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                synced = true;   // (one more sweep: everything the kernel posted is visible now)
            }
        }
    }
}

int fetch_result(ccv_mppi_handle* h, double* u_opt_out, ccv_mppi_stats* stats) {
    if (int rc = flush_pending(h)) return rc;
    const size_t n = (size_t)h->R;
    if (h->mail_pending) {
        h->mail_pending = false;
        if (int rc = wait_mail(h, n + (stats ? 4 : 0))) return rc;
        for (size_t i = 0; i < n + (stats ? 4u : 0u); ++i) {
            const unsigned long long hi = h->h_mail[2 * i], lo = h->h_mail[2 * i + 1];
            const unsigned long long bits = (hi & 0xFFFFFFFF00000000ull) | (lo >> 32);
            std::memcpy(&h->h_pin[i], &bits, sizeof(double));
        }
    } else {
        // one D2H of [u* | stats] through pinned memory, then a stream sync
        HIP_TRY(h, hipMemcpyAsync(h->h_pin, h->d_nominal, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->h_pin + n, h->d_stats, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    int nonfinite = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!std::isfinite(h->h_pin[i])) nonfinite = 1;
        if (u_opt_out) u_opt_out[i] = h->h_pin[i];
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->sum_w = h->h_pin[n + 0];
        stats->min_cost = h->h_pin[n + 1];
        stats->max_cost = h->h_pin[n + 2];
        stats->n_zero_weight = (int64_t)h->h_pin[n + 3];
        stats->nonfinite = nonfinite;
        if (h->timing) {
            int rc = timing_collect(h);
            if (rc) return rc;
            stats->device_us = h->last_iter_us;
            stats->rollout_us = h->last_roll_us;
        }
    }
    return CCV_MPPI_OK;
}


// ---- direct exchange: what travels between the ranks at set-up, and the boxes this process owns -------------------------
struct ExchangeBlob {
    hipIpcMemHandle_t ipc;
    int32_t fine_grained;   // the box is fine-grained memory (coherent across devices)
    uint32_t nonce;         // rank 0's is the base of the sequence numbers
    int32_t pid;
    int32_t device;         // ordinal inside that process
    char bus[24];           // PCI bus id of the device that holds the box
};

// Boxes created by THIS process: hipIpcOpenMemHandle refuses a handle of the opening process itself, so a process that
// drives several handles (several devices from one process, or several shards on one device) maps them directly.
struct OwnBox {
    ExchangeBlob blob;
    ExchangeBox* box;
};
std::mutex g_box_mutex;
std::vector<OwnBox> g_boxes;

void exchange_release(ccv_mppi_handle* h) {
    for (int r = 0; r < kMaxRanks; ++r) {
        if (h->box_opened[r] && h->box_peer[r]) (void)hipIpcCloseMemHandle(h->box_peer[r]);
        h->box_opened[r] = false;
        h->box_peer[r] = nullptr;
    }
    if (h->d_box) {
        {
            std::lock_guard<std::mutex> lock(g_box_mutex);
            g_boxes.erase(std::remove_if(g_boxes.begin(), g_boxes.end(), [&](const OwnBox& b) { return b.box == h->d_box; }), g_boxes.end());
        }
        (void)hipFree(h->d_box);
    }
    if (h->d_xvec) (void)hipFree(h->d_xvec);
    if (h->h_xflag) (void)hipHostFree(h->h_xflag);
    h->h_xflag = nullptr;
    if (h->pending_vec == h->d_xvec) h->pending_vec = nullptr;
    h->d_box = nullptr;
    h->d_xvec = nullptr;
    h->d_xflag = nullptr;
    h->xchg_connected = false;
    h->xchg_world = h->xchg_rank = 0;
}

// After a synchronisation: did the exchange kernel give up waiting for a peer?  The flag lives in pinned host-mapped memory
// (the kernel stores to it once, system scope, in the rare case): reading it costs no copy and no extra synchronisation.
// Sticky: the controls are NaN from then on; releasing the exchange (ccv_mppi_destroy, or a failed set-up) frees it and a
// new ccv_mppi_exchange_create starts from a cleared one.
int exchange_check(ccv_mppi_handle* h) {
    if (!h->h_xflag) return CCV_MPPI_OK;
    if (*static_cast<volatile int32_t*>(h->h_xflag)) {
        char msg[256];
        std::snprintf(msg, sizeof(msg), "direct exchange: a peer's partial vector did not arrive within %.3g s; the controls are NaN "
                                        "from that iteration on (destroy the handles and set the exchange up again)", h->xchg_timeout_s);
        return fail(h, CCV_MPPI_ERR_TIMEOUT, msg);
    }
    return CCV_MPPI_OK;
}
}  // namespace

extern "C" {

const char* ccv_mppi_version(void) { return kVersion; }

int ccv_mppi_udim(int model) {
    if (model < CCV_MPPI_DIFF_DRIVE || model > CCV_MPPI_FULL_BODY) return CCV_MPPI_ERR_INVALID_ARG;
    return udim_of(model);
}

const char* ccv_mppi_last_error(const ccv_mppi_handle* h) { return h ? h->err.c_str() : "null handle"; }

int ccv_mppi_create(const ccv_mppi_config* cfg, ccv_mppi_handle** out) {
    if (!cfg || !out) return CCV_MPPI_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->abi_version != CCV_MPPI_ABI_VERSION) return CCV_MPPI_ERR_INVALID_ARG;
    if (cfg->model < CCV_MPPI_DIFF_DRIVE || cfg->model > CCV_MPPI_FULL_BODY) return CCV_MPPI_ERR_INVALID_ARG;
    if (cfg->num_samples < 1 || cfg->horizon < 3 || cfg->horizon > CCV_MPPI_MAX_HORIZON) return CCV_MPPI_ERR_INVALID_ARG;
    if (cfg->sample_offset < 0) return CCV_MPPI_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CCV_MPPI_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return CCV_MPPI_ERR_NO_DEVICE;
    ccv_mppi_handle* h = new (std::nothrow) ccv_mppi_handle();
    if (!h) return CCV_MPPI_ERR_ALLOC;
    h->cfg = *cfg;
    h->udim = udim_of(cfg->model);
    h->K = cfg->num_samples;
    h->H = cfg->horizon;
    h->R = (h->H - 1) * h->udim;
    h->pitch = round_up(h->K, 64);
    h->nchunks = (h->K + kChunk - 1) / kChunk;
    h->nblocks = (h->K + kPcSamples - 1) / kPcSamples;
    const char* env = getenv("CCV_MPPI_WINDOW");
    h->lds_window = !(env && std::strcmp(env, "scalar") == 0);
    const char* kenv = getenv("CCV_MPPI_KERNEL");
    h->coop = !(kenv && std::strcmp(kenv, "v1") == 0) && h->lds_window;
    int cus = 256;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    // diff-drive, steering: the four-wave kernel (noise / dynamics / distance / store wave, mppi_rollout_r4.h; round 2: -6 %
    // against the three-wave kernel at C2 and, unlike it, the same time in every process at C3).  Full body: its dynamics batch
    // needs 250 VGPRs, so the four-wave kernel is built for one wave per SIMD there, one workgroup per CU -- used up to that many
    // blocks of 64 samples (round 3; the reference's own K = 10 000 is 157 blocks), the two-wave kernel up to four per CU.
    // CCV_MPPI_KERNEL=pc / r3 / r4 force one where built (experiments, tests)
    if (h->coop && h->cfg.model != CCV_MPPI_FULL_BODY) h->coop = 3;
    if (h->coop && h->cfg.model == CCV_MPPI_FULL_BODY && h->nblocks <= cus) h->coop = 3;
    if (h->coop && kenv && std::strcmp(kenv, "r4") == 0) h->coop = 3;
    if (h->coop && kenv && std::strcmp(kenv, "r3") == 0 && h->cfg.model != CCV_MPPI_FULL_BODY) h->coop = 2;
    if (h->coop && kenv && std::strcmp(kenv, "pc") == 0) h->coop = 1;
    // More blocks of 64 samples than the multi-wave kernels can hold at once (4 workgroups per CU): one wave does
    // everything for its samples (mppi_rollout_solo.h) -- the SIMDs are kept busy by independent waves then, and the
    // hand-off between the waves of a workgroup is pure loss.  Measured on 256 CUs (kernel us, multi-wave vs one-wave):
    // diff drive K = 65 536: 45 vs 56; 98 304: 85 vs 80; 131 072: 106 vs 88; 524 288: 348 vs 295; steering 131 072: 135 vs
    // 117; full body 65 536: 169 vs 192; 98 304: 316 vs 291; 131 072 (C4): 374 vs 335.  CCV_MPPI_KERNEL=solo forces it.
    {
        // (round 2, four-wave kernel against one-wave kernel, diff drive, kernel us: K = 81 920 61.0 vs 64.2; 98 304 66.0 vs 64.3;
        //  131 072 75.2 vs 71.0; 196 608 104 vs 99; steering 131 072 90.1 vs 84.0 -- the switch sits at five blocks per CU there)
        h->solo = h->coop && !kenv && h->nblocks > (h->cfg.model == CCV_MPPI_FULL_BODY ? 4 : 5) * cus;
        if (h->coop && kenv && std::strcmp(kenv, "solo") == 0) h->solo = true;
    }
    // wave priorities (pc_rotate_priority): measured -4 us on the three-wave kernel (C2), -3 % on the two-wave one (C4), and
    // with four levels -5 us on the four-wave kernel (43.4 -> 38.3 us at C2)
    // four-wave kernel: (rank + level[role] + b) mod 4 with the roles' levels per model -- noise / dynamics / distance / store
    // (r4_rotate_priority, mppi_rollout_pc.h: where the numbers are)
    auto levels = [](int noise, int dynamics, int distance, int store) { return 16 + (noise | dynamics << 2 | distance << 4 | store << 6); };
    const int kR4PrioLevels = h->cfg.model == CCV_MPPI_DIFF_DRIVE ? levels(3, 2, 1, 0)
                              : h->cfg.model == CCV_MPPI_STEERING_DIFF_DRIVE ? levels(2, 3, 1, 0) : levels(0, 1, 2, 3);
    h->prio_rotate = h->coop == 3 ? kR4PrioLevels : h->coop ? 1 : 0;
    if (const char* pv = std::getenv("CCV_MPPI_PRIO")) {   // 0: off; 2 .. 5: the formula schedules; 16 + digits: a level table
        const int v = std::atoi(pv);
        h->prio_rotate = v == 0 ? 0 : (((v >= 2 && v <= 5) || (v >= 16 && v < 16 + 256)) && h->coop == 3) ? v : h->prio_rotate;
    }

    // Exact window pruning in the distance loop (pc_prune_window).  Measured on one box, kernel us off -> on: diff drive
    // K = 65 536 49.0 -> 42.7, steering 61.7 -> 57.3 (three-wave kernels).  CCV_MPPI_PRUNE=0/1 forces it (experiments;
    // results do not depend on it, tested).
    // (not for windows of 16 points or fewer -- the reference default H = 15: the test costs a block about what the whole loop
    //  over such a window does; per iteration 14.6 -> 14.2 us (dd), 16.0 -> 15.3 (sd), 21.3 -> 21.0 (fb) without it)
    h->prune = (h->coop && h->H > 16) ? 1 : 0;
    if (const char* pv = std::getenv("CCV_MPPI_PRUNE")) h->prune = std::strcmp(pv, "0") != 0;
    if (const char* pv = std::getenv("CCV_MPPI_FAST_CLAMP")) h->fast_clamp_allowed = std::strcmp(pv, "0") != 0;

    auto bail = [&](int code, const char* what, hipError_t e) {
        fail(h, code, what, e);
        std::fprintf(stderr, "ccv_mppi_create: %s\n", h->err.c_str());
        ccv_mppi_destroy(h);
        return code;
    };
    hipError_t e;
    if ((e = hipSetDevice(cfg->device)) != hipSuccess) return bail(CCV_MPPI_ERR_NO_DEVICE, "hipSetDevice", e);
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && ncu > 0) h->cu_count = ncu;
    }
    if ((e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(CCV_MPPI_ERR_HIP, "hipStreamCreate", e);
    h->stream = h->own_stream;
    const size_t P = (size_t)h->pitch;
    // The large arrays are pieces of ONE allocation, each starting on a 2 MB boundary: one mapping, one set of large page
    // fragments, seven allocator calls less per handle.
    const size_t nparts_max = (size_t)(h->nblocks > h->nchunks ? h->nblocks : h->nchunks);
    struct Piece { void** p; size_t bytes; } pieces[] = {
        {(void**)&h->d_z, (size_t)h->R * P * sizeof(float)},
        {(void**)&h->d_xs, (size_t)h->H * P * sizeof(double)},
        {(void**)&h->d_ys, (size_t)h->H * P * sizeof(double)},
        {(void**)&h->d_u, (size_t)h->R * P * sizeof(double)},
        {(void**)&h->d_cost, P * sizeof(double)},
        {(void**)&h->d_w, P * sizeof(double)},
        {(void**)&h->d_partial, (size_t)(h->R + 1) * nparts_max * sizeof(double)},
    };
    constexpr size_t kPieceAlign = (size_t)2 << 20;
    size_t arena_bytes = 0;
    for (const Piece& pc : pieces) arena_bytes += (pc.bytes + kPieceAlign - 1) / kPieceAlign * kPieceAlign;
    if ((e = hipMalloc(&h->d_arena, arena_bytes)) != hipSuccess) return bail(CCV_MPPI_ERR_ALLOC, "hipMalloc", e);
    if ((e = hipMemset(h->d_arena, 0, arena_bytes)) != hipSuccess) return bail(CCV_MPPI_ERR_HIP, "hipMemset", e);
    {
        size_t at = 0;
        for (const Piece& pc : pieces) {
            *pc.p = static_cast<char*>(h->d_arena) + at;
            at += (pc.bytes + kPieceAlign - 1) / kPieceAlign * kPieceAlign;
        }
    }
    struct { double** p; size_t n; } allocs[] = {
        {&h->d_nominal, (size_t)(CCV_MPPI_MAX_HORIZON + 8) * CCV_MPPI_MAX_UDIM},   // padded: read 4 at a time
        {&h->d_nom_used, (size_t)(CCV_MPPI_MAX_HORIZON + 8) * CCV_MPPI_MAX_UDIM},
        {&h->d_statpart, nparts_max * 3},
        {&h->d_vec, (size_t)h->R + 1},
        {&h->d_stats, 4},
        {&h->d_cmin, 1},
    };
    for (auto& a : allocs) {
        if ((e = hipMalloc(a.p, a.n * sizeof(double))) != hipSuccess) return bail(CCV_MPPI_ERR_ALLOC, "hipMalloc", e);
        if ((e = hipMemset(*a.p, 0, a.n * sizeof(double))) != hipSuccess) return bail(CCV_MPPI_ERR_HIP, "hipMemset", e);
    }
#if defined(CCV_DIAG)
    {
        const size_t dbg_bytes = (size_t)(kDiagHeader + kDiagSlots * kDiagBlocks) * sizeof(unsigned long long);
        if ((e = hipMalloc(&h->d_dbg, dbg_bytes)) != hipSuccess) return bail(CCV_MPPI_ERR_ALLOC, "hipMalloc", e);
        if ((e = hipMemset(h->d_dbg, 0, dbg_bytes)) != hipSuccess) return bail(CCV_MPPI_ERR_HIP, "hipMemset", e);
    }
#endif
    if (const char* tv = std::getenv("CCV_MPPI_THROTTLE")) h->throttle = std::strcmp(tv, "0") != 0;
    for (hipEvent_t& te : h->throttle_ev)
        if ((e = hipEventCreateWithFlags(&te, hipEventDisableTiming)) != hipSuccess) return bail(CCV_MPPI_ERR_HIP, "hipEventCreate", e);
    h->pin_doubles = (size_t)h->R + 16;
    if ((e = hipHostMalloc(&h->h_pin, h->pin_doubles * sizeof(double), hipHostMallocDefault)) != hipSuccess)
        return bail(CCV_MPPI_ERR_ALLOC, "hipHostMalloc", e);
    {
        const size_t mail_bytes = ((size_t)h->R + 4) * 2 * sizeof(unsigned long long);
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&h->h_mail), mail_bytes, hipHostMallocMapped)) != hipSuccess)
            return bail(CCV_MPPI_ERR_ALLOC, "hipHostMalloc(mailbox)", e);
        std::memset(h->h_mail, 0, mail_bytes);   // (sequence numbers start at 1: nothing in a fresh box is taken for a packet)
        if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_mail), h->h_mail, 0)) != hipSuccess)
            return bail(CCV_MPPI_ERR_HIP, "hipHostGetDevicePointer(mailbox)", e);
        if (const char* mv = std::getenv("CCV_MPPI_MAILBOX")) h->use_mail = std::strcmp(mv, "0") != 0;
    }
    if ((e = hipDeviceSynchronize()) != hipSuccess) return bail(CCV_MPPI_ERR_HIP, "hipDeviceSynchronize", e);
    *out = h;
    return CCV_MPPI_OK;
}

int ccv_mppi_destroy(ccv_mppi_handle* h) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    // everything below belongs to the handle's device; the caller's current device is put back afterwards
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->own_stream && h->own_stream != h->stream) (void)hipStreamSynchronize(h->own_stream);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->throttle_ev)
        if (e) (void)hipEventDestroy(e);
    exchange_release(h);
    void* bufs[] = {h->d_arena, h->d_nominal, h->d_nom_used, h->d_statpart, h->d_vec, h->d_stats,
                    h->d_cmin, h->d_scratch, h->d_frame, h->d_path, h->d_trace, h->d_dbg};   // (u, z, xs, ys, cost, w, partial: the arena)
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (h->h_pin) (void)hipHostFree(h->h_pin);
    if (h->h_mail) (void)hipHostFree(h->h_mail);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    if (prev >= 0 && prev != h->cfg.device) (void)hipSetDevice(prev);
    delete h;
    return CCV_MPPI_OK;
}

int ccv_mppi_set_stream(ccv_mppi_handle* h, void* hip_stream) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (int rc = flush_pending(h)) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    for (bool& u : h->throttle_used) u = false;   // marks recorded on the old stream are complete (synchronised above)
    return CCV_MPPI_OK;
}


#if defined(CCV_DIAG)
// diagnostic builds: the stamps of the first `nblocks` workgroups of the last launch, kDiagSlots values each (mppi_diag.h)
extern "C" int ccv_mppi_debug_blocks(ccv_mppi_handle* h, unsigned long long* out, int nblocks) {
    if (!h || !out || nblocks < 0 || nblocks > kDiagBlocks) return CCV_MPPI_ERR_INVALID_ARG;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out, h->d_dbg + kDiagHeader, (size_t)nblocks * kDiagSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return CCV_MPPI_OK;
}
#endif

int ccv_mppi_synchronize(ccv_mppi_handle* h) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (int rc = flush_pending(h)) return rc;   // (after this the caller may free the partials buffer)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return exchange_check(h);
}

int ccv_mppi_set_nominal(ccv_mppi_handle* h, const double* u) {
    if (!h || !u) return CCV_MPPI_ERR_INVALID_ARG;
    h->pending_vec = nullptr;   // overwritten anyway
    if (int rc = flush_pending(h)) return rc;   // (a deferred update of the resident loop must not land on top of it)
    HIP_TRY(h, hipMemcpyAsync(h->d_nominal, u, (size_t)h->R * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    // (the resident loop's plant integrates u*[0]: its angle bounds must cover what the caller put there; NaN sticks)
    for (int d = 0; d < CCV_MPPI_MAX_UDIM; ++d) h->nom_absmax[d] = 0.0;
    for (int n = 0; n < h->R; ++n) {
        const int d = n % h->udim;
        if (!(std::fabs(u[n]) <= h->nom_absmax[d])) h->nom_absmax[d] = std::fabs(u[n]);
    }
    return CCV_MPPI_OK;
}

int ccv_mppi_get_nominal(ccv_mppi_handle* h, double* u) {
    if (!h || !u) return CCV_MPPI_ERR_INVALID_ARG;
    if (int rc = flush_pending(h)) return rc;
    HIP_TRY(h, hipMemcpyAsync(u, h->d_nominal, (size_t)h->R * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return exchange_check(h);
}

int ccv_mppi_iterate(ccv_mppi_handle* h, const double* x0, double dt, const double* x_ref, const double* y_ref,
                     double yaw_ref0, uint64_t seed, uint64_t iter, double* u_opt_out, ccv_mppi_stats* stats) {
    int rc = check_iter_args(h, x0, dt, x_ref, y_ref);
    if (rc) return rc;
    h->want_mail = !(stats && h->timing);   // (a timed call synchronises for its events anyway)
    rc = enqueue_iteration(h, x0, dt, x_ref, y_ref, yaw_ref0, seed, iter, true, nullptr);
    h->want_mail = false;
    if (rc) return rc;
    return fetch_result(h, u_opt_out, stats);
}

int ccv_mppi_iterate_enqueue(ccv_mppi_handle* h, const double* x0, double dt, const double* x_ref, const double* y_ref,
                             double yaw_ref0, uint64_t seed, uint64_t iter) {
    int rc = check_iter_args(h, x0, dt, x_ref, y_ref);
    if (rc) return rc;
    return enqueue_iteration(h, x0, dt, x_ref, y_ref, yaw_ref0, seed, iter, true, nullptr);
}

int ccv_mppi_partials_size(const ccv_mppi_handle* h) { return h ? h->R + 1 : CCV_MPPI_ERR_INVALID_ARG; }

int ccv_mppi_iterate_partials_enqueue(ccv_mppi_handle* h, const double* x0, double dt, const double* x_ref,
                                      const double* y_ref, double yaw_ref0, uint64_t seed, uint64_t iter,
                                      double* dev_partials) {
    int rc = check_iter_args(h, x0, dt, x_ref, y_ref);
    if (rc) return rc;
    if (!dev_partials) return fail(h, CCV_MPPI_ERR_INVALID_ARG, "dev_partials is null");
    if (h->cfg.flags & CCV_MPPI_FLAG_MIN_SHIFT)
        return fail(h, CCV_MPPI_ERR_INVALID_ARG, "MIN_SHIFT needs a cross-device min; not supported with partials");
    return enqueue_iteration(h, x0, dt, x_ref, y_ref, yaw_ref0, seed, iter, false, dev_partials);
}

int ccv_mppi_apply_partials_enqueue(ccv_mppi_handle* h, const double* dev_partials) {
    if (!h || !dev_partials) return CCV_MPPI_ERR_INVALID_ARG;
    // deferred: the next fused rollout launch on this handle forms u* = V / S while it stages the warm start (one kernel
    // launch less per iteration); any other reader of the warm start triggers k_apply_partials first (flush_pending)
    int rc = flush_pending(h);
    if (rc) return rc;
    h->pending_vec = dev_partials;
    return CCV_MPPI_OK;
}

// ---- direct exchange between the devices of a node ------------------------------------------------------------------

int ccv_mppi_exchange_handle_bytes(void) { return (int)sizeof(ExchangeBlob); }

int ccv_mppi_exchange_create(ccv_mppi_handle* h, int32_t world, int32_t rank, void* ipc_handle_out) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (!ipc_handle_out || world < 1 || world > kMaxRanks || rank < 0 || rank >= world)
        return fail(h, CCV_MPPI_ERR_INVALID_ARG, "exchange: 1 <= world <= 8, 0 <= rank < world");
    if (h->cfg.flags & CCV_MPPI_FLAG_MIN_SHIFT)
        return fail(h, CCV_MPPI_ERR_INVALID_ARG, "MIN_SHIFT needs a cross-device min; not supported with partials");
    if (h->d_box) return fail(h, CCV_MPPI_ERR_STATE, "exchange already created");
    const DeviceGuard guard(h->cfg.device);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    // Fine-grained memory: a peer's stores become visible to a kernel that is already running on the owner.  Ordinary
    // (coarse-grained) device memory guarantees that only inside one device, so it is accepted as a fall-back only when
    // every rank's box lives on this same device (ccv_mppi_exchange_connect checks; a one-device rehearsal).
    void* box = nullptr;
    ExchangeBlob blob;
    std::memset(&blob, 0, sizeof(blob));
    hipError_t e = hipExtMallocWithFlags(&box, sizeof(ExchangeBox), hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&blob.ipc, box);
    blob.fine_grained = e == hipSuccess ? 1 : 0;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (box) (void)hipFree(box);
        box = nullptr;
        HIP_TRY(h, hipMalloc(&box, sizeof(ExchangeBox)));
        e = hipIpcGetMemHandle(&blob.ipc, box);
        if (e != hipSuccess) {
            (void)hipFree(box);
            return fail(h, CCV_MPPI_ERR_HIP, "hipIpcGetMemHandle failed: no peer mapping on this system", e);
        }
    }
    h->d_box = static_cast<ExchangeBox*>(box);
    auto undo = [&](int code, const char* what, hipError_t err) {
        exchange_release(h);
        return fail(h, code, what, err);
    };
    if ((e = hipMemset(box, 0, sizeof(ExchangeBox))) != hipSuccess) return undo(CCV_MPPI_ERR_HIP, "hipMemset(box)", e);
    if ((e = hipMalloc(&h->d_xvec, (size_t)(h->R + 1) * sizeof(double))) != hipSuccess) return undo(CCV_MPPI_ERR_ALLOC, "hipMalloc(xvec)", e);
    if ((e = hipMemset(h->d_xvec, 0, (size_t)(h->R + 1) * sizeof(double))) != hipSuccess) return undo(CCV_MPPI_ERR_HIP, "hipMemset(xvec)", e);
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&h->h_xflag), sizeof(int32_t), hipHostMallocMapped)) != hipSuccess)
        return undo(CCV_MPPI_ERR_ALLOC, "hipHostMalloc(xflag)", e);
    *h->h_xflag = 0;
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_xflag), h->h_xflag, 0)) != hipSuccess)
        return undo(CCV_MPPI_ERR_HIP, "hipHostGetDevicePointer(xflag)", e);
    if ((e = hipDeviceSynchronize()) != hipSuccess) return undo(CCV_MPPI_ERR_HIP, "hipDeviceSynchronize", e);
    blob.pid = (int32_t)getpid();
    blob.device = h->cfg.device;
    if (hipDeviceGetPCIBusId(blob.bus, (int)sizeof(blob.bus), h->cfg.device) != hipSuccess) std::snprintf(blob.bus, sizeof(blob.bus), "dev%d", h->cfg.device);
    blob.bus[sizeof(blob.bus) - 1] = 0;
    // sequence base: a job that is started again must not take the packets an earlier one left in a peer's box for its own
    const uint64_t now = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    blob.nonce = (uint32_t)(now ^ (now >> 29) ^ ((uint64_t)blob.pid * 0x9E3779B97F4A7C15ull >> 17));
    h->xchg_nonce = blob.nonce;
    if (const char* tv = std::getenv("CCV_MPPI_EXCHANGE_TIMEOUT_MS")) {
        const long ms = std::atol(tv);
        if (ms > 0) h->xchg_timeout_ticks = (unsigned long long)ms * 100000ull;
    }
    h->xchg_timeout_s = (double)h->xchg_timeout_ticks * 1.0e-8;   // 100 MHz ticks
    h->box_fine_grained = blob.fine_grained != 0;
    h->xchg_world = world;
    h->xchg_rank = rank;
    h->xchg_seq = 0;
    {
        std::lock_guard<std::mutex> lock(g_box_mutex);
        g_boxes.push_back(OwnBox{blob, h->d_box});
    }
    std::memcpy(ipc_handle_out, &blob, sizeof(blob));
    return CCV_MPPI_OK;
}

int ccv_mppi_exchange_connect(ccv_mppi_handle* h, const void* ipc_handles) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (!ipc_handles) return fail(h, CCV_MPPI_ERR_INVALID_ARG, "ipc_handles is null");
    if (!h->d_box) return fail(h, CCV_MPPI_ERR_STATE, "ccv_mppi_exchange_create first");
    if (h->xchg_connected) return fail(h, CCV_MPPI_ERR_STATE, "exchange already connected");
    const DeviceGuard guard(h->cfg.device);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const ExchangeBlob* blobs = static_cast<const ExchangeBlob*>(ipc_handles);
    ExchangeBlob mine;
    std::memcpy(&mine, &blobs[h->xchg_rank], sizeof(mine));   // (the caller's buffer need not be aligned)
    auto undo = [&](int code, const char* what, hipError_t err) {
        for (int r = 0; r < kMaxRanks; ++r) {
            if (h->box_opened[r] && h->box_peer[r]) (void)hipIpcCloseMemHandle(h->box_peer[r]);
            h->box_opened[r] = false;
            h->box_peer[r] = nullptr;
        }
        return fail(h, code, what, err);
    };
    for (int r = 0; r < h->xchg_world; ++r) {
        ExchangeBlob peer;
        std::memcpy(&peer, reinterpret_cast<const char*>(ipc_handles) + (size_t)r * sizeof(ExchangeBlob), sizeof(peer));
        peer.bus[sizeof(peer.bus) - 1] = 0;
        if (r == h->xchg_rank) {
            h->box_peer[r] = h->d_box;
            continue;
        }
        if ((!peer.fine_grained || !mine.fine_grained) && std::strcmp(peer.bus, mine.bus) != 0)
            return undo(CCV_MPPI_ERR_STATE, "direct exchange refused: a box in coarse-grained memory would be polled across devices "
                                            "(fine-grained allocation or its IPC export failed); use the all-reduce path", hipSuccess);
        // a box of this very process (several handles driven by one process) is used as it is
        ExchangeBox* local = nullptr;
        if (peer.pid == (int32_t)getpid()) {
            std::lock_guard<std::mutex> lock(g_box_mutex);
            for (const OwnBox& b : g_boxes)
                if (std::memcmp(&b.blob.ipc, &peer.ipc, sizeof(peer.ipc)) == 0 && b.blob.nonce == peer.nonce) local = b.box;
        }
        if (local) {
            if (peer.device != h->cfg.device) {
                const hipError_t pe = hipDeviceEnablePeerAccess(peer.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return undo(CCV_MPPI_ERR_HIP, "hipDeviceEnablePeerAccess", pe);
                (void)hipGetLastError();
            }
            h->box_peer[r] = local;
            continue;
        }
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, peer.ipc, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return undo(CCV_MPPI_ERR_HIP, "hipIpcOpenMemHandle", e);
        h->box_peer[r] = static_cast<ExchangeBox*>(p);
        h->box_opened[r] = true;
        // touch the mapping through the runtime first: a mapping that cannot be used fails here with an error code
        // instead of faulting in a kernel
        unsigned long long probe = 0;
        if ((e = hipMemcpy(&probe, p, sizeof(probe), hipMemcpyDeviceToHost)) != hipSuccess) return undo(CCV_MPPI_ERR_HIP, "peer box not readable", e);
    }
    ExchangeBlob first;
    std::memcpy(&first, ipc_handles, sizeof(first));
    h->xchg_base = first.nonce;
    h->xchg_connected = true;
    return CCV_MPPI_OK;
}

int ccv_mppi_exchange_info(const ccv_mppi_handle* h, int32_t* world, int32_t* rank, int32_t* fine_grained, int32_t* connected) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (world) *world = h->xchg_world;
    if (rank) *rank = h->xchg_rank;
    if (fine_grained) *fine_grained = (h->d_box && h->box_fine_grained) ? 1 : 0;
    if (connected) *connected = h->xchg_connected ? 1 : 0;
    return CCV_MPPI_OK;
}

int ccv_mppi_iterate_exchange_enqueue(ccv_mppi_handle* h, const double* x0, double dt, const double* x_ref,
                                      const double* y_ref, double yaw_ref0, uint64_t seed, uint64_t iter) {
    int rc = check_iter_args(h, x0, dt, x_ref, y_ref);
    if (rc) return rc;
    if (!h->xchg_connected) return fail(h, CCV_MPPI_ERR_STATE, "ccv_mppi_exchange_create / _connect first");
    return enqueue_iteration(h, x0, dt, x_ref, y_ref, yaw_ref0, seed, iter, false, nullptr, false, true);
}

// ---- device-resident closed loop (mppi_resident.h) ------------------------------------------------------------------

int ccv_mppi_resident_set_path(ccv_mppi_handle* h, const double* path_x, const double* path_y, int32_t n_path,
                               double resolution) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (!path_x || !path_y || n_path < 1 || !(resolution > 0.0)) return fail(h, CCV_MPPI_ERR_INVALID_ARG, "path: null, empty or resolution <= 0");
    HIP_TRY(h, hipStreamSynchronize(h->stream));   // (a queued k_advance may still read the old path)
    if (h->d_path) HIP_TRY(h, hipFree(h->d_path));
    h->d_path = nullptr;
    h->n_path = 0;
    HIP_TRY(h, hipMalloc(&h->d_path, (size_t)2 * n_path * sizeof(double)));
    HIP_TRY(h, hipMemcpy(h->d_path, path_x, (size_t)n_path * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_path + n_path, path_y, (size_t)n_path * sizeof(double), hipMemcpyHostToDevice));
    if (!h->d_frame) {
        HIP_TRY(h, hipMalloc(&h->d_frame, sizeof(ResidentFrame)));
        HIP_TRY(h, hipMemset(h->d_frame, 0, sizeof(ResidentFrame)));
        HIP_TRY(h, hipMalloc(&h->d_trace, (size_t)ccv_mppi_handle::kTraceRows * 6 * sizeof(double)));
        HIP_TRY(h, hipMemset(h->d_trace, 0, (size_t)ccv_mppi_handle::kTraceRows * 6 * sizeof(double)));
    }
    h->n_path = n_path;
    h->path_resolution = resolution;
    return CCV_MPPI_OK;
}

int ccv_mppi_resident_set_pose(ccv_mppi_handle* h, const double* state) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (!state) return fail(h, CCV_MPPI_ERR_INVALID_ARG, "state is null");
    if (!h->d_frame) return fail(h, CCV_MPPI_ERR_STATE, "ccv_mppi_resident_set_path first");
    const int nx = h->cfg.model == CCV_MPPI_FULL_BODY ? 5 : 3;
    // pose and step counter: the head of the frame
    struct { double x0[5]; double yaw_ref0; int32_t index, steps; } head{};
    for (int i = 0; i < nx; ++i) head.x0[i] = state[i];
    static_assert(offsetof(ResidentFrame, W) == sizeof(head), "frame head layout");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(h->d_frame, &head, sizeof(head), hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i) h->res_angle_abs[i] = std::fabs(head.x0[2 + i]);
    h->res_steps = 0;
    h->have_pose = true;
    return CCV_MPPI_OK;
}

namespace {
int resident_step(ccv_mppi_handle* h, double dt, uint64_t seed, uint64_t iter, int32_t advance, bool normalise, double* vec_out,
                  bool exchange = false) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    // dt is the stride of the window index (dd:160-163): as ccv_mppi_calc_ref_path, only 0 <= dt < inf is defined
    const double stride = h->cfg.v_ref * dt / h->path_resolution;
    if (!(dt >= 0.0) || !std::isfinite(dt)) return fail(h, CCV_MPPI_ERR_INVALID_ARG, "resident step: dt must be finite and not negative");
    if (!h->d_frame || !h->have_pose) return fail(h, CCV_MPPI_ERR_STATE, "ccv_mppi_resident_set_path and _set_pose first");
    if (!std::isfinite(stride) || stride < 0.0 || stride * h->H > 2.0e9)
        return fail(h, CCV_MPPI_ERR_INVALID_ARG, "resident step: v_ref * dt / resolution is not a usable window stride");
    // Bounds on |yaw|, |roll|, |pitch| of the resident pose, which the host never sees: the command u*[0] is a weighted mean
    // of clamped samples (or what ccv_mppi_set_nominal put there), and the plant takes an angle modulo 2 pi once it leaves
    // +-kAngleRebase (rebase_angle), so the bounds stay below kAngleRebase + one step for a loop of any length.
    const ccv_mppi_config& c = h->cfg;
    auto lim = [&](int d) {
        const double a = std::fmax(std::fabs(c.u_min[d]), std::fabs(c.u_max[d]));
        const double b = std::fmax(h->inj_absmax[d], h->nom_absmax[d]);
        return (b == b) ? std::fmax(a, b) : b;   // NaN sticks
    };
    double nb[3] = {h->res_angle_abs[0], h->res_angle_abs[1], h->res_angle_abs[2]};
    if (advance) {
        auto step = [&](double bound, int d) {
            const double after = bound + lim(d) * dt;
            return after <= kAngleRebase ? after : (after == after ? kAngleRebase : after);   // (beyond it the plant re-bases: |angle| <= pi)
        };
        nb[0] = step(nb[0], 1);
        if (c.model == CCV_MPPI_FULL_BODY) {
            nb[1] = step(nb[1], 3);
            nb[2] = step(nb[2], 4);
        }
    }
    // everything that can refuse the step is checked BEFORE k_advance moves the pose
    {
        RolloutArgs chk;
        const double bounds[5] = {0.0, 0.0, nb[0], nb[1], nb[2]};
        fill_args(h, chk, bounds, dt, 0.0, seed, iter);
        // k_advance itself takes sin / cos of the OLD heading (+ the steering command)
        const double heading_bound = h->res_angle_abs[0] + (c.model == CCV_MPPI_DIFF_DRIVE ? 0.0 : lim(2));
        if (!h->coop || fast_trig_safe(h, chk, MODE_FUSED) == kTrigUnsafe || !(heading_bound <= kFastTrigLimit))
            return fail(h, CCV_MPPI_ERR_STATE, "the resident loop needs the cooperative kernels and bounded pose angles / commands");
    }
    const bool fuse = h->fin_pending && !h->pending_vec;   // the last tick's update is still to be launched: together with this prologue
    if (advance && !fuse) {
        if (int rc = flush_pending(h)) return rc;   // the command is u*[0]: a deferred division has to happen now
    }
    AdvanceArgs V;
    V.frame = h->d_frame;
    V.path_x = h->d_path;
    V.path_y = h->d_path + h->n_path;
    V.nominal = h->d_nominal;
    V.trace = h->d_trace;
    V.dt = dt;
    V.v_ref = h->cfg.v_ref;
    V.resolution = h->path_resolution;
    V.n_path = h->n_path;
    V.H = h->H;
    V.model = h->cfg.model;
    V.advance = advance ? 1 : 0;
    V.trace_cap = ccv_mppi_handle::kTraceRows;
    if (fuse) {
        hipLaunchKernelGGL(k_finalize_advance, dim3(finalize_blocks(h->fin_args.R) + 1), dim3(kBlock), 0, h->stream, h->fin_args, V);
        h->fin_pending = false;
    } else {
        if (int rc = flush_pending(h)) return rc;
        hipLaunchKernelGGL(k_advance, dim3(1), dim3(kAdvanceThreads), 0, h->stream, V);
    }
    HIP_TRY(h, hipGetLastError());
    h->res_steps += 1;
    for (int i = 0; i < 3; ++i) h->res_angle_abs[i] = nb[i];
    const double bounds[5] = {0.0, 0.0, nb[0], nb[1], nb[2]};
    return enqueue_iteration(h, bounds, dt, nullptr, nullptr, 0.0, seed, iter, normalise, vec_out, true, exchange);
}
}  // namespace

int ccv_mppi_resident_step_enqueue(ccv_mppi_handle* h, double dt, uint64_t seed, uint64_t iter, int32_t advance) {
    return resident_step(h, dt, seed, iter, advance, true, nullptr);
}

int ccv_mppi_resident_step_partials_enqueue(ccv_mppi_handle* h, double dt, uint64_t seed, uint64_t iter, int32_t advance,
                                            double* dev_partials) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (!dev_partials) return fail(h, CCV_MPPI_ERR_INVALID_ARG, "dev_partials is null");
    if (h->cfg.flags & CCV_MPPI_FLAG_MIN_SHIFT)
        return fail(h, CCV_MPPI_ERR_INVALID_ARG, "MIN_SHIFT needs a cross-device min; not supported with partials");
    return resident_step(h, dt, seed, iter, advance, false, dev_partials);
}

int ccv_mppi_resident_step_exchange_enqueue(ccv_mppi_handle* h, double dt, uint64_t seed, uint64_t iter, int32_t advance) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (!h->xchg_connected) return fail(h, CCV_MPPI_ERR_STATE, "ccv_mppi_exchange_create / _connect first");
    return resident_step(h, dt, seed, iter, advance, false, nullptr, true);
}

int ccv_mppi_resident_read(ccv_mppi_handle* h, double* state, int32_t* current_index, double* x_ref, double* y_ref,
                           double* yaw_ref0, int64_t* steps) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (!h->d_frame || !h->have_pose) return fail(h, CCV_MPPI_ERR_STATE, "ccv_mppi_resident_set_path and _set_pose first");
    ResidentFrame F;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(&F, h->d_frame, sizeof(F), hipMemcpyDeviceToHost));
    const int nx = h->cfg.model == CCV_MPPI_FULL_BODY ? 5 : 3;
    if (state) for (int i = 0; i < nx; ++i) state[i] = F.x0[i];
    if (current_index) *current_index = F.index;
    if (x_ref) for (int i = 0; i < h->H; ++i) x_ref[i] = F.x_ref[i];
    if (y_ref) for (int i = 0; i < h->H; ++i) y_ref[i] = F.y_ref[i];
    if (yaw_ref0) *yaw_ref0 = F.yaw_ref0;
    if (steps) *steps = F.steps;
    return CCV_MPPI_OK;
}

int ccv_mppi_resident_read_trace(ccv_mppi_handle* h, int32_t max_rows, double* rows, int32_t* n_rows) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (!rows || !n_rows || max_rows < 0) return fail(h, CCV_MPPI_ERR_INVALID_ARG, "rows / n_rows null or max_rows < 0");
    if (!h->d_frame || !h->have_pose) return fail(h, CCV_MPPI_ERR_STATE, "ccv_mppi_resident_set_path and _set_pose first");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    // the last min(steps, capacity, max_rows) launches, oldest first
    const int64_t cap = ccv_mppi_handle::kTraceRows;
    const int64_t have = h->res_steps < cap ? h->res_steps : cap;
    const int64_t n = have < max_rows ? have : max_rows;
    std::vector<double> ring((size_t)cap * 6);
    HIP_TRY(h, hipMemcpy(ring.data(), h->d_trace, ring.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) {
        const int64_t step = h->res_steps - n + i;
        std::memcpy(rows + i * 6, ring.data() + (step % cap) * 6, 6 * sizeof(double));
    }
    *n_rows = (int32_t)n;
    return CCV_MPPI_OK;
}

// ---- stage-wise -------------------------------------------------------------------------------------------------

int ccv_mppi_sample(ccv_mppi_handle* h, uint64_t seed, uint64_t iter) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    RolloutArgs A;
    const double zero[5] = {0, 0, 0, 0, 0};
    fill_args(h, A, zero, 0.1, 0.0, seed, iter);
    int rc = launch_sample(h, A);
    if (rc) return rc;
    // (no host wait: the stage-wise calls hand nothing back to the host before ccv_mppi_update -- sampling(),
    //  predict_States() and calc_Weights() are void in the reference -- so they only enqueue; the stream keeps their order,
    //  ccv_mppi_update and every read-back wait for what they return)
    for (int d = 0; d < h->udim; ++d) h->inj_absmax[d] = std::fmax(std::fabs(h->cfg.u_min[d]), std::fabs(h->cfg.u_max[d]));
    h->controls_in_z = false;
    h->have_controls = true;
    h->have_rollout = h->have_weights = false;
    return CCV_MPPI_OK;
}

int ccv_mppi_inject_controls(ccv_mppi_handle* h, const double* u_samples) {
    if (!h || !u_samples) return CCV_MPPI_ERR_INVALID_ARG;
    // [K][(H-1)][u_dim] -> rows n = t*u_dim + d of pitch doubles
    std::vector<double> tmp((size_t)h->R * h->pitch, 0.0);
    for (int d = 0; d < CCV_MPPI_MAX_UDIM; ++d) h->inj_absmax[d] = 0.0;
    for (int i = 0; i < h->K; ++i)
        for (int n = 0; n < h->R; ++n) {
            const double v = u_samples[(size_t)i * h->R + n];
            tmp[(size_t)n * h->pitch + i] = v;
            const int d = n % h->udim;
            if (!(std::fabs(v) <= h->inj_absmax[d])) h->inj_absmax[d] = std::fabs(v);   // NaN sticks
        }
    HIP_TRY(h, hipMemcpyAsync(h->d_u, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->controls_in_z = false;
    h->have_controls = true;
    h->have_rollout = h->have_weights = false;
    return CCV_MPPI_OK;
}

int ccv_mppi_rollout(ccv_mppi_handle* h, const double* x0, double dt) {
    if (!h || !x0) return CCV_MPPI_ERR_INVALID_ARG;
    if (!h->have_controls) return fail(h, CCV_MPPI_ERR_STATE, "ccv_mppi_rollout before ccv_mppi_sample/inject_controls");
    RolloutArgs A;
    Window W;
    std::memset(&W, 0, sizeof(W));
    fill_args(h, A, x0, dt, 0.0, 0, 0);
    A.store_u = 0;
    A.store_xy = 1;
    A.do_cost = 0;
    int rc = launch_rollout(h, A, W, MODE_ROLLOUT);
    if (rc) return rc;
    std::memcpy(h->st_x0, A.x0, sizeof(h->st_x0));
    h->st_dt = dt;
    h->have_rollout = true;
    h->have_weights = false;
    return CCV_MPPI_OK;
}

int ccv_mppi_weights(ccv_mppi_handle* h, const double* x_ref, const double* y_ref, double yaw_ref0) {
    if (!h || !x_ref || !y_ref) return CCV_MPPI_ERR_INVALID_ARG;
    if (!h->have_rollout) return fail(h, CCV_MPPI_ERR_STATE, "ccv_mppi_weights before ccv_mppi_rollout");
    RolloutArgs A;
    Window W;
    fill_args(h, A, h->st_x0, h->st_dt, yaw_ref0, 0, 0);
    fill_window(h, W, h->st_x0, x_ref, y_ref);
    A.store_u = 0;
    A.store_xy = 0;
    A.do_cost = 1;
    // the rollout is recomputed from the stored controls (bit-identical to the stored states) and scored
    int rc = launch_rollout(h, A, W, MODE_COST);
    if (rc) return rc;
    // sum of weights (calc_Weights normalises, dd:222) without touching u*
    rc = launch_update(h, false, nullptr);
    if (rc) return rc;
    h->have_weights = true;
    return CCV_MPPI_OK;
}

int ccv_mppi_update(ccv_mppi_handle* h, double* u_opt_out, ccv_mppi_stats* stats) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    if (!h->have_weights) return fail(h, CCV_MPPI_ERR_STATE, "ccv_mppi_update before ccv_mppi_weights");
    h->want_mail = !(stats && h->timing);
    int rc = launch_update(h, true, nullptr);
    h->want_mail = false;
    if (rc) return rc;
    return fetch_result(h, u_opt_out, stats);
}

// ---- read-back --------------------------------------------------------------------------------------------------

int ccv_mppi_read_candidates(ccv_mppi_handle* h, int32_t first, int32_t count, int32_t stride, double* xy_out) {
    if (!h || !xy_out || first < 0 || count < 0 || stride < 1) return CCV_MPPI_ERR_INVALID_ARG;
    if (h->cfg.flags & CCV_MPPI_FLAG_NO_STATE_STORE) return fail(h, CCV_MPPI_ERR_STATE, "state buffer disabled (NO_STATE_STORE)");
    if (!h->have_rollout) return fail(h, CCV_MPPI_ERR_STATE, "no rollout yet");
    if (count == 0) return CCV_MPPI_OK;
    if ((int64_t)first + (int64_t)(count - 1) * stride >= h->K) return fail(h, CCV_MPPI_ERR_INVALID_ARG, "candidate range exceeds num_samples");
    const size_t n = (size_t)count * h->H * 2;
    int rc = ensure_scratch(h, n * sizeof(double));
    if (rc) return rc;
    hipLaunchKernelGGL(k_gather_xy, dim3((count * h->H + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, h->d_xs, h->d_ys,
                       h->pitch, h->H, first, count, stride, h->d_scratch);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(xy_out, h->d_scratch, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CCV_MPPI_OK;
}

int ccv_mppi_read_top_candidates(ccv_mppi_handle* h, int32_t count, int32_t* sample_out, double* weight_out, double* xy_out) {
    if (!h || !sample_out || count < 0) return CCV_MPPI_ERR_INVALID_ARG;
    if (!h->have_weights) return fail(h, CCV_MPPI_ERR_STATE, "no weights yet");
    if (count > h->K) return fail(h, CCV_MPPI_ERR_INVALID_ARG, "count exceeds num_samples");
    if (xy_out && (h->cfg.flags & CCV_MPPI_FLAG_NO_STATE_STORE)) return fail(h, CCV_MPPI_ERR_STATE, "state buffer disabled (NO_STATE_STORE)");
    if (count == 0) return CCV_MPPI_OK;
    // scratch: [count] indices (as 8-byte slots) | [count] weights | [count][H][2] states
    const size_t n_xy = xy_out ? (size_t)count * h->H * 2 : 0;
    int rc = ensure_scratch(h, ((size_t)count * 2 + n_xy) * sizeof(double));
    if (rc) return rc;
    int* d_idx = reinterpret_cast<int*>(h->d_scratch);
    double* d_wsel = h->d_scratch + count;
    double* d_xy = h->d_scratch + 2 * (size_t)count;
    hipLaunchKernelGGL(k_top_weights, dim3(1), dim3(kTopBlock), 0, h->stream, h->d_w, h->K, count, d_idx, d_wsel);
    HIP_TRY(h, hipGetLastError());
    std::vector<int> idx(count);
    std::vector<double> wsel(count);
    HIP_TRY(h, hipMemcpyAsync(idx.data(), d_idx, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(wsel.data(), d_wsel, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    // descending weight (NaN first), ties by sample index
    std::vector<int> order(count);
    for (int i = 0; i < count; ++i) order[i] = i;
    auto key = [&](int i) { return (unsigned long long)*reinterpret_cast<const long long*>(&wsel[i]); };
    std::sort(order.begin(), order.end(), [&](int a, int b) { return key(a) != key(b) ? key(a) > key(b) : idx[a] < idx[b]; });
    std::vector<int> sorted_idx(count);
    for (int i = 0; i < count; ++i) {
        sorted_idx[i] = idx[order[i]];
        sample_out[i] = sorted_idx[i];
        if (weight_out) weight_out[i] = wsel[order[i]];
    }
    if (xy_out) {
        HIP_TRY(h, hipMemcpyAsync(d_idx, sorted_idx.data(), (size_t)count * sizeof(int), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_gather_xy_list, dim3((count * h->H + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, h->d_xs, h->d_ys,
                           h->pitch, h->H, d_idx, count, d_xy);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(xy_out, d_xy, n_xy * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return CCV_MPPI_OK;
}

static int check_range(ccv_mppi_handle* h, int32_t first, int32_t count, const void* out) {
    if (!h || !out || first < 0 || count < 0) return CCV_MPPI_ERR_INVALID_ARG;
    if ((int64_t)first + count > h->K) return fail(h, CCV_MPPI_ERR_INVALID_ARG, "range exceeds num_samples");
    return CCV_MPPI_OK;
}

int ccv_mppi_read_costs(ccv_mppi_handle* h, int32_t first, int32_t count, double* out) {
    int rc = check_range(h, first, count, out);
    if (rc) return rc;
    if (!h->have_weights) return fail(h, CCV_MPPI_ERR_STATE, "no costs yet");
    HIP_TRY(h, hipMemcpyAsync(out, h->d_cost + first, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CCV_MPPI_OK;
}

int ccv_mppi_read_weights(ccv_mppi_handle* h, int32_t first, int32_t count, double* out) {
    int rc = check_range(h, first, count, out);
    if (rc) return rc;
    if (!h->have_weights) return fail(h, CCV_MPPI_ERR_STATE, "no weights yet");
    if (count == 0) return CCV_MPPI_OK;
    if ((rc = flush_pending(h)) != CCV_MPPI_OK) return rc;   // (sum w)
    rc = ensure_scratch(h, (size_t)count * sizeof(double));
    if (rc) return rc;
    hipLaunchKernelGGL(k_normalise_weights, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, h->d_w, h->d_stats,
                       first, count, h->d_scratch);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out, h->d_scratch, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CCV_MPPI_OK;
}

int ccv_mppi_read_controls(ccv_mppi_handle* h, int32_t first, int32_t count, double* out) {
    int rc = check_range(h, first, count, out);
    if (rc) return rc;
    if (!h->have_controls) return fail(h, CCV_MPPI_ERR_STATE, "no controls yet");
    if (count == 0) return CCV_MPPI_OK;
    std::vector<double> tmp((size_t)h->R * count);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->controls_in_z) {
        // the fused iteration kept the normals and the warm start they were drawn around: u = clamp(double(z) * sigma + u*[n]),
        // the samplers' operations (this file is compiled with -ffp-contract=off: a multiply and an add, as on the device)
        std::vector<float> zt((size_t)h->R * count);
        std::vector<double> nom((size_t)h->R);
        HIP_TRY(h, hipMemcpy2D(zt.data(), (size_t)count * sizeof(float), h->d_z + first, (size_t)h->pitch * sizeof(float),
                               (size_t)count * sizeof(float), (size_t)h->R, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(nom.data(), h->d_nom_used, (size_t)h->R * sizeof(double), hipMemcpyDeviceToHost));
        const bool steer_off = h->cfg.model == CCV_MPPI_FULL_BODY && (h->cfg.flags & CCV_MPPI_FLAG_STEER_OFF);
        for (int n = 0; n < h->R; ++n) {
            const int d = n % h->udim;
            const double lo = h->cfg.u_min[d], hi = h->cfg.u_max[d], sigma = h->cfg.control_noise;
            for (int i = 0; i < count; ++i) {
                const double prod = (double)zt[(size_t)n * count + i] * sigma;
                double v = prod + nom[n];
                v = v < lo ? lo : (v > hi ? hi : v);
                if (steer_off && d == 2) v = 0.0;
                tmp[(size_t)n * count + i] = v;
            }
        }
    } else
    HIP_TRY(h, hipMemcpy2D(tmp.data(), (size_t)count * sizeof(double), h->d_u + first, (size_t)h->pitch * sizeof(double),
                           (size_t)count * sizeof(double), (size_t)h->R, hipMemcpyDeviceToHost));
    for (int i = 0; i < count; ++i)
        for (int n = 0; n < h->R; ++n) out[(size_t)i * h->R + n] = tmp[(size_t)n * count + i];
    return CCV_MPPI_OK;
}

// ---- measurement ------------------------------------------------------------------------------------------------

int ccv_mppi_timing_enable(ccv_mppi_handle* h, int32_t on) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    int rc = timing_collect(h);
    if (rc) return rc;
    h->timing = on != 0;
    h->timing_every = on > 1 ? on : 1;   // on = n > 1: sample every n-th iteration
    h->timing_count = 0;
    return CCV_MPPI_OK;
}

int ccv_mppi_timing_read(ccv_mppi_handle* h, double* rollout_us_sum, double* iter_us_sum, int64_t* n_iters, int32_t reset) {
    if (!h) return CCV_MPPI_ERR_INVALID_ARG;
    int rc = timing_collect(h);
    if (rc) return rc;
    if (rollout_us_sum) *rollout_us_sum = h->t_roll_sum;
    if (iter_us_sum) *iter_us_sum = h->t_iter_sum;
    if (n_iters) *n_iters = h->t_n;
    if (reset) {
        h->t_roll_sum = h->t_iter_sum = 0.0;
        h->t_n = 0;
    }
    return CCV_MPPI_OK;
}

}  // extern "C"
