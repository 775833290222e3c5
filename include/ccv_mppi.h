/*
 * ccv_mppi.h -- C ABI of the MI355X-native MPPI hot path (libccv_mppi_hip.so).
 *
 * Drop-in boundary for the four hot methods of the reference controller classes of
 * YoshikiMaekawa2000/ccv_mppi_path_tracker.  The reference has no FFI/plugin interface
 * (SURVEY.md 8b): the seam is the bodies of the private methods
 *
 *     sampling()                    src/diff_drive_mppi.cpp:81-102   src/steering_diff_drive_mppi.cpp:97-118   src/full_body_mppi.cpp:491-520
 *     predict_States()              src/diff_drive_mppi.cpp:111-124  src/steering_diff_drive_mppi.cpp:127-140  src/full_body_mppi.cpp:454-489
 *     calc_Weights()                src/diff_drive_mppi.cpp:212-223  src/steering_diff_drive_mppi.cpp:228-239  src/full_body_mppi.cpp:426-443
 *     determine_OptimalSolution()   src/diff_drive_mppi.cpp:225-246  src/steering_diff_drive_mppi.cpp:241-264  src/full_body_mppi.cpp:308-333
 *
 * called once each, in this order, from run() (src/diff_drive_mppi.cpp:352-358).  INTEGRATION.md
 * shows the patch a maintainer applies to those method bodies.
 *
 * Conventions (mirroring the reference):
 *   - plain C, no exceptions; every function returns an int status (0 = OK, <0 = error) and never aborts;
 *   - an opaque handle owns all device memory, allocated once in ccv_mppi_create() (the reference allocates
 *     everything once in the constructor, src/diff_drive_mppi.cpp:36-46) and never resized;
 *   - the caller owns all host arrays; pointers are host pointers unless a parameter name starts with `dev_`;
 *   - one handle = one caller thread; a call that hands data back to the host (ccv_mppi_iterate, ccv_mppi_update, the
 *     read-backs, ccv_mppi_get_nominal) returns when that data is complete; calls named *_enqueue and the stage-wise calls
 *     that return nothing to the host (ccv_mppi_sample, ccv_mppi_rollout, ccv_mppi_weights -- void methods in the
 *     reference) only enqueue work on the handle's stream, which keeps the order; ccv_mppi_synchronize waits for all of it;
 *   - all floating point data is IEEE double (the reference computes in double throughout);
 *   - there is NO CPU fallback: without a usable HIP device every call fails with CCV_MPPI_ERR_NO_DEVICE.
 */
#ifndef CCV_MPPI_H_
#define CCV_MPPI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCV_MPPI_ABI_VERSION 1
#define CCV_MPPI_MAX_UDIM 5
#define CCV_MPPI_MAX_HORIZON 128 /* window coefficients travel in the kernel-argument segment */

/* status codes */
#define CCV_MPPI_OK 0
#define CCV_MPPI_ERR_INVALID_ARG (-1)
#define CCV_MPPI_ERR_NO_DEVICE (-2)
#define CCV_MPPI_ERR_HIP (-3)
#define CCV_MPPI_ERR_STATE (-4) /* stage-wise calls in the wrong order */
#define CCV_MPPI_ERR_ALLOC (-5)
#define CCV_MPPI_ERR_TIMEOUT (-6) /* direct exchange: a peer's partial vector never arrived (reported at the next synchronisation) */

/* controller model; u_dim = 2 / 3 / 5, control order = declaration order of the reference
 * (dd: v,w  sd: v,w,steer  fb: v,w,direction,roll_v,pitch_v) */
#define CCV_MPPI_DIFF_DRIVE 0
#define CCV_MPPI_STEERING_DIFF_DRIVE 1
#define CCV_MPPI_FULL_BODY 2

/* flags */
#define CCV_MPPI_FLAG_ROLL_OFF 0x1   /* src/full_body_mppi.cpp:43-46: zmp_weight = roll_v_weight = 0 */
#define CCV_MPPI_FLAG_STEER_OFF 0x2  /* src/full_body_mppi.cpp:517: direction forced to 0 after the draw */
#define CCV_MPPI_FLAG_MIN_SHIFT 0x4  /* NOT reference behaviour: w = exp(-(c - min c)/lambda) (underflow-safe) */
#define CCV_MPPI_FLAG_NO_STATE_STORE 0x8 /* skip the K x H x,y state buffer (read_candidates then fails) */

typedef struct ccv_mppi_config {
    int32_t abi_version;        /* CCV_MPPI_ABI_VERSION */
    int32_t model;              /* CCV_MPPI_DIFF_DRIVE ... */
    int32_t num_samples;        /* K on THIS device (param "num_samples", dd:19) */
    int32_t horizon;            /* H: number of states, H-1 control steps (param "horizon", dd:18); 3..CCV_MPPI_MAX_HORIZON */
    int32_t sample_offset;      /* global id of local sample 0 when K is sharded over devices (0 otherwise) */
    int32_t device;             /* HIP device ordinal */
    int32_t flags;              /* CCV_MPPI_FLAG_* */
    int32_t reserved;
    double control_noise;       /* sigma, one value for every control dimension (dd:20) */
    double lambda;              /* dd:21 */
    double v_ref;               /* dd:28 */
    double u_min[CCV_MPPI_MAX_UDIM]; /* clamp bounds per control dimension (dd:22-26, sd:23-28, fb:13-26) */
    double u_max[CCV_MPPI_MAX_UDIM];
    double path_weight;         /* dd:33 */
    double v_weight;            /* dd:34 ("control_weight" in dd/sd, "v_weight" in fb:35) */
    double zmp_weight;          /* fb:36 */
    double roll_v_weight;       /* fb:37 */
    double back_weight;         /* fb:38 */
    double yaw_weight;          /* fb:39 */
} ccv_mppi_config;

typedef struct ccv_mppi_stats {
    double sum_w;        /* sum_i exp(-cost_i/lambda) (unnormalised; 0 => u* is NaN exactly like dd:222) */
    double min_cost;
    double max_cost;
    int64_t n_zero_weight; /* samples whose weight underflowed to exactly 0 */
    int32_t nonfinite;   /* 1 if any component of the returned controls is NaN/Inf */
    int32_t reserved;
    float device_us;     /* device time of the last iteration's kernels (hipEvent), 0 if not measured */
    float rollout_us;    /* device time of the dominant kernel (sample+rollout+cost) */
} ccv_mppi_stats;

typedef struct ccv_mppi_handle ccv_mppi_handle;

/* ---- lifetime ------------------------------------------------------------------------------------ */
/* Replaces the allocation part of the constructors (dd:36-46, sd:38-48, fb:72-84). */
int ccv_mppi_create(const ccv_mppi_config* cfg, ccv_mppi_handle** out);
int ccv_mppi_destroy(ccv_mppi_handle* h);
/* Launch on a caller-owned HIP stream (hipStream_t passed as void*); NULL restores the handle's own stream. */
int ccv_mppi_set_stream(ccv_mppi_handle* h, void* hip_stream);
const char* ccv_mppi_last_error(const ccv_mppi_handle* h);
const char* ccv_mppi_version(void);
int ccv_mppi_udim(int model);

/* ---- warm start: optimal_solution controls, layout [(H-1)][u_dim] (dd.h:100) -------- */
int ccv_mppi_set_nominal(ccv_mppi_handle* h, const double* u);
int ccv_mppi_get_nominal(ccv_mppi_handle* h, double* u);

/* ---- one whole iteration: sampling + predict_States + calc_Weights + determine_OptimalSolution ---- */
/* x0: (x, y, yaw[, roll, pitch]) = current_pose_/current_state_ (dd:115-117, fb:458-464); dt: dt_ (dd:347);
 * x_ref/y_ref: the H window points written by calc_RefPath() (dd:156-181); yaw_ref0: yaw_ref_[0] (fb:408);
 * seed/iter: counter-based noise key (the reference reseeds mt19937 from random_device every call, dd:83-84);
 * u_opt_out: new optimal_solution controls [(H-1)][u_dim]; stats may be NULL. Blocking. */
int ccv_mppi_iterate(ccv_mppi_handle* h, const double* x0, double dt, const double* x_ref, const double* y_ref,
                     double yaw_ref0, uint64_t seed, uint64_t iter, double* u_opt_out, ccv_mppi_stats* stats);
/* Same work, enqueued on the stream without any host synchronisation; u* stays resident on the device as the
 * next call's warm start.  Read it back with ccv_mppi_get_nominal(). */
int ccv_mppi_iterate_enqueue(ccv_mppi_handle* h, const double* x0, double dt, const double* x_ref,
                             const double* y_ref, double yaw_ref0, uint64_t seed, uint64_t iter);
int ccv_mppi_synchronize(ccv_mppi_handle* h);

/* ---- stage-wise mirrors of the four reference methods (reference call order; parity tests) -------- */
int ccv_mppi_sample(ccv_mppi_handle* h, uint64_t seed, uint64_t iter);                 /* sampling() */
/* Parity hook: overwrite the sample controls with caller data [K][(H-1)][u_dim] (already clamped, e.g. the
 * output of the reference's own sampling()); replaces ccv_mppi_sample for that iteration. */
int ccv_mppi_inject_controls(ccv_mppi_handle* h, const double* u_samples);
int ccv_mppi_rollout(ccv_mppi_handle* h, const double* x0, double dt);                  /* predict_States() */
int ccv_mppi_weights(ccv_mppi_handle* h, const double* x_ref, const double* y_ref, double yaw_ref0); /* calc_Weights() */
int ccv_mppi_update(ccv_mppi_handle* h, double* u_opt_out, ccv_mppi_stats* stats);     /* determine_OptimalSolution() */

/* ---- read-back ------------------------------------------------------------------------------------ */
/* Feeds publish_CandidatePath() (dd:265-294) without a K x H device-to-host copy: samples first, first+stride, ...
 * (count of them); xy_out layout [count][H][2]. */
int ccv_mppi_read_candidates(ccv_mppi_handle* h, int32_t first, int32_t count, int32_t stride, double* xy_out);
/* The `count` samples with the largest weights of the last cost evaluation, in descending order of weight (ties: lower
 * sample index first; NaN weights first): their indices, optionally their unnormalised weights [count] and their rollouts
 * [count][H][2].  Selection (radix select) and gather run on the device; only the selected rows cross PCIe.  This is
 * what publish_CandidatePath() (dd:265-294) can sensibly show of K = 65 536 candidates. */
int ccv_mppi_read_top_candidates(ccv_mppi_handle* h, int32_t count, int32_t* sample_out, double* weight_out, double* xy_out);
int ccv_mppi_read_costs(ccv_mppi_handle* h, int32_t first, int32_t count, double* out);
/* normalised weights w_i / sum_w, i.e. the reference's weights_ (dd:222) */
int ccv_mppi_read_weights(ccv_mppi_handle* h, int32_t first, int32_t count, double* out);
/* sample controls, layout [count][(H-1)][u_dim] */
int ccv_mppi_read_controls(ccv_mppi_handle* h, int32_t first, int32_t count, double* out);

/* ---- K sharded over several devices (one handle per device/process) ------------------ */
/* Number of doubles in the per-device partial vector: 1 + (H-1)*u_dim = [sum w, sum w*u[t][d] ...]. */
int ccv_mppi_partials_size(const ccv_mppi_handle* h);
/* sample+rollout+cost+local reduction; leaves the unnormalised partials in dev_partials (DEVICE memory of this
 * handle's device, e.g. a torch tensor) without touching u*.  The caller all-reduces (sum) dev_partials across
 * devices (RCCL) on the same stream, then calls ccv_mppi_apply_partials on every device. No host sync. */
int ccv_mppi_iterate_partials_enqueue(ccv_mppi_handle* h, const double* x0, double dt, const double* x_ref,
                                      const double* y_ref, double yaw_ref0, uint64_t seed, uint64_t iter,
                                      double* dev_partials);
/* u* = partials[1:] / partials[0] becomes the resident warm start (no host sync).  The division is deferred into the
 * next ccv_mppi_iterate*_enqueue on this handle (one kernel launch less per iteration); dev_partials must stay valid and
 * unchanged until then, or until ccv_mppi_synchronize / ccv_mppi_get_nominal, which perform it at once. */
int ccv_mppi_apply_partials_enqueue(ccv_mppi_handle* h, const double* dev_partials);

/* Same, without a collective-library call per iteration, for the devices of ONE node (world <= 8): each handle owns a
 * small box in its HBM that the peers map (hipIpc over xGMI); the update kernel writes this device's partial vector
 * straight into every peer's box, waits for the peers' vectors in its own and adds them in rank order (identical bits on
 * every device); the division is deferred into the next rollout as above.  Set-up: every process calls _create, the
 * processes exchange the returned handles (ccv_mppi_exchange_handle_bytes() bytes each, e.g. torch.distributed
 * all_gather), then every process calls _connect with all of them in rank order.  Every rank must then issue the same
 * sequence of ccv_mppi_iterate_exchange_enqueue calls; a peer that does not arrive within 10 s yields NaN controls, not a
 * hang, and the next ccv_mppi_synchronize / ccv_mppi_get_nominal on that handle returns CCV_MPPI_ERR_TIMEOUT (the *_enqueue
 * calls themselves cannot know).  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 on hosts whose driver only supports dmabuf IPC.
 * The box is fine-grained (device-coherent) memory; where that cannot be allocated or exported it falls back to ordinary
 * device memory, which _connect accepts only if every rank's box lives on the same physical device (a one-device
 * rehearsal) and refuses with CCV_MPPI_ERR_STATE otherwise -- the caller then takes the all-reduce path above.
 * Handles of one process (one process driving several devices) connect to each other directly, without hipIpc.
 * The blob a rank hands out also carries a nonce; rank 0's is the base of the packet sequence numbers, so a job that is
 * started again does not mistake packets of an earlier one for its own. */
int ccv_mppi_exchange_handle_bytes(void);
int ccv_mppi_exchange_create(ccv_mppi_handle* h, int32_t world, int32_t rank, void* ipc_handle_out);
int ccv_mppi_exchange_connect(ccv_mppi_handle* h, const void* ipc_handles);
/* What was set up (any pointer may be NULL): world, rank, whether this handle's box is fine-grained memory, whether
 * _connect has succeeded. */
int ccv_mppi_exchange_info(const ccv_mppi_handle* h, int32_t* world, int32_t* rank, int32_t* fine_grained, int32_t* connected);
int ccv_mppi_iterate_exchange_enqueue(ccv_mppi_handle* h, const double* x0, double dt, const double* x_ref,
                                      const double* y_ref, double yaw_ref0, uint64_t seed, uint64_t iter);

/* ---- device-resident closed loop -------------------------------------------------- */
/* The per-tick prologue of run() on the device: the whole reference path and the pose live in HBM, and every step does
 *   (advance != 0) pose <- pose advanced for dt by the command u*[0] of the previous step (the Euler model of
 *                  predict_NextState(), dd:104-109 / sd:120-125 / fb:445-452: the closed-loop plant, what the robot does
 *                  between two ticks; ccv_mppi_plant_step() in ccv_mppi_host.h is the same arithmetic on the host)
 *   get_CurrentIndex() (dd:126-140) + calc_RefPath() (dd:156-181) from that pose, then the iteration itself
 * with no host data in between: a closed loop costs two kernel launches per tick (the update of a tick is launched together
 * with the prologue of the next one) and no PCIe traffic.  Window, index
 * and pose are bit-identical to ccv_mppi_calc_ref_path() / ccv_mppi_plant_step() on the host; yaw_ref[0] (read by fb:408
 * only) comes from the device atan2 and may differ from libm's in the last place.  v_ref and the horizon are the
 * handle's; `resolution` is the spacing of the path poses (resolution_, dd:160).  Needs the default (cooperative)
 * kernels.
 * dt must be finite and not negative (it is the stride of the window index, dd:160-163; the reference's behaviour for anything
 * else is undefined): CCV_MPPI_ERR_INVALID_ARG otherwise, as from ccv_mppi_calc_ref_path().
 * The plant takes yaw / roll / pitch modulo 2 pi once they leave +-1e4 rad (the real node reads them from tf in [-pi, pi]),
 * identically in ccv_mppi_plant_step(), so a loop of any length stays inside the range of the kernels' branch-free
 * sin/cos.  A step is refused with CCV_MPPI_ERR_STATE -- before the pose is moved -- when the pose angles (as set by
 * _set_pose) or the commands (clamp bounds, and whatever ccv_mppi_set_nominal put into u*) can leave that range (1e5 rad)
 * within one horizon. */
int ccv_mppi_resident_set_path(ccv_mppi_handle* h, const double* path_x, const double* path_y, int32_t n_path,
                               double resolution);
/* state: (x, y, yaw[, roll, pitch]); also restarts the step counter and the trace */
int ccv_mppi_resident_set_pose(ccv_mppi_handle* h, const double* state);
int ccv_mppi_resident_step_enqueue(ccv_mppi_handle* h, double dt, uint64_t seed, uint64_t iter, int32_t advance);
/* K sharded over devices: as ccv_mppi_iterate_partials_enqueue; every device advances the same pose with the same u* */
int ccv_mppi_resident_step_partials_enqueue(ccv_mppi_handle* h, double dt, uint64_t seed, uint64_t iter, int32_t advance,
                                            double* dev_partials);
/* ... or as ccv_mppi_iterate_exchange_enqueue (direct exchange between the devices of one node) */
int ccv_mppi_resident_step_exchange_enqueue(ccv_mppi_handle* h, double dt, uint64_t seed, uint64_t iter, int32_t advance);
/* Synchronises; any output pointer may be NULL.  x_ref / y_ref: H values, the window of the last step. */
int ccv_mppi_resident_read(ccv_mppi_handle* h, double* state, int32_t* current_index, double* x_ref, double* y_ref,
                           double* yaw_ref0, int64_t* steps);
/* The poses of the last steps, oldest first: rows of (x, y, yaw, roll, pitch, current_index); at most max_rows and at
 * most the 8192 most recent.  This is what record_state.py:118-139 logs from tf. */
int ccv_mppi_resident_read_trace(ccv_mppi_handle* h, int32_t max_rows, double* rows, int32_t* n_rows);

/* ---- measurement ----------------------------------------------------------------------------------- */
/* on = 1: every iteration records hipEvents around the dominant kernel and the whole launch sequence; on = n > 1: every
 * n-th iteration only (keeps the event overhead out of a throughput measurement); 0: off.
 * ccv_mppi_timing_read returns the accumulated device times of the recorded iterations since the last reset (it synchronises). */
int ccv_mppi_timing_enable(ccv_mppi_handle* h, int32_t on);
int ccv_mppi_timing_read(ccv_mppi_handle* h, double* rollout_us_sum, double* iter_us_sum, int64_t* n_iters,
                         int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* CCV_MPPI_H_ */
