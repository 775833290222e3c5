// HIP kernels of the MPPI hot path for gfx950 (MI355X).  Hand-written, wave64, one sample per lane.
//
//   k_rollout_cost   sampling() + predict_States() + calc_Cost()/calc_MinDistance() + exp   (dd:81-124,183-221;
//                    sd:97-140,199-237; fb:394-424,445-520)
//   k_update_partials / k_finalize / k_apply_partials    calc_Weights() normalisation + determine_OptimalSolution()
//                    (dd:216-237, sd:232-255, fb:437-326)  -- in mppi_update.h
//   k_sample         stand-alone sampling() for the stage-wise ABI
//   k_gather_xy      strided read-back feeding publish_CandidatePath() (dd:265-294)
//
// Data layout in HBM (all fp64, k fastest => every store/load of a wave is one contiguous 512-byte run):
//   u    [(H-1)*u_dim][pitch]   row n = t*u_dim + d  clamped sample controls (sample[i].v_[t] ... in the reference); written by
//                               the stage-wise calls and the plain kernel only -- the fused iteration stores, in its place,
//   z    [(H-1)*u_dim][pitch]   fp32: the N(0,1) variate each control was made from, with the warm start it was made around
//                               (nominal_used): u = clamp(double(z) * sigma + u*[n]) is re-derived, bit for bit, where it is
//                               needed (epilogue, ccv_mppi_read_controls, k_materialize_controls) -- half the bytes to store and
//                               to re-read
//   xs,ys[H][pitch]             sample[i].x_[t], sample[i].y_[t]
//   cost [pitch], w [pitch]     per-sample cost and unnormalised weight exp(-cost/lambda)
//   nominal [(H-1)*u_dim]       optimal_solution controls (resident warm start)
// pitch = K rounded up to 64.  Compiled with -ffp-contract=off: a*b+c is fused only where fma() is written.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "../../include/ccv_mppi.h"
#include "noise_spec.h"

#if defined(CCV_DIAG)
#include "mppi_diag.h"   // in-kernel time stamps of the diagnostic builds (tools/ablate.py, tools/stamps_r4.py)
#else
#define CCV_DIAG_STAMP(A, slot) do {} while (0)   // a product build contains nothing of the diagnostics
#define CCV_DIAG_STAMP_VALUE(A, slot, value) do {} while (0)
#endif

namespace ccv {

constexpr int kMaxH = CCV_MPPI_MAX_HORIZON;
constexpr int kTU = 8;        // time steps per register block (kTU*u_dim is a multiple of 4 normals for every model)
constexpr int kBlock = 256;   // 4 waves: one per SIMD of a CU
constexpr int kChunk = 2048;  // samples per k_update_partials block
constexpr int kPcSamples = 64;   // samples per block of the cooperative kernels: one per lane (mppi_rollout_*.h)

enum : int { MODE_FUSED = 0, MODE_ROLLOUT = 1, MODE_COST = 2 };
//   MODE_FUSED    device Philox noise, controls + states stored, cost + weight   (ccv_mppi_iterate*)
//   MODE_ROLLOUT  controls read from HBM, states stored, no cost                  (ccv_mppi_rollout)
//   MODE_COST     controls read from HBM, nothing stored but cost + weight        (ccv_mppi_weights)

enum : int { SRC_PHILOX = 0, SRC_BUFFER = 1 };

__host__ __device__ constexpr int udim_of(int model) { return model == 0 ? 2 : (model == 1 ? 3 : 5); }

// Window coefficients: squared distance to window point j is |p|^2 + a_j*px + b_j*py + c_j with p relative to the
// current pose.  Travels in the kernel-argument segment (wave-uniform => scalar loads).
struct Window {
    double a[kMaxH];
    double b[kMaxH];
    double c[kMaxH];
};

// Device-resident closed loop (k_advance below): the pose, the window built from it and its distance coefficients stay
// in HBM; a rollout launched with RolloutArgs::frame set takes x0, yaw_ref0 and the window from here instead of from its
// kernel arguments.
struct ResidentFrame {
    double x0[5];      // x, y, yaw[, roll, pitch]
    double yaw_ref0;   // yaw_ref[0] of calc_RefPath() (fb:408 is its only reader)
    int32_t index;     // current_index_ (get_CurrentIndex())
    int32_t steps;     // k_advance launches so far
    Window W;
    double x_ref[kMaxH], y_ref[kMaxH];
};

struct RolloutArgs {
    double x0[5];
    double dt;
    double yaw_ref0;
    double sigma, lambda, v_ref;
    double umin[5], umax[5];
    double w_path, w_v, w_zmp, w_rollv, w_back, w_yaw;
    double fb_mass, fb_L, fb_Ixx, fb_gz;  // fb.h:212-216, fb:86-91, fb.h:30
    double inv_dt;                        // 1 / dt, correctly rounded on the host: div_uniform() below
    uint32_t seed_lo, seed_hi, iter_lo, iter_hi;
    int32_t K, pitch, H, k_offset;
    int32_t steer_off, store_u, store_xy, do_cost;
    const double* __restrict__ nominal;
    double* u;
    float* z;                 // fused iteration: the normals, in place of u (see the layout comment at the top)
    double* nominal_used;     // ... and the warm start they were made around (written by workgroup 0)
    double* xs;
    double* ys;
    double* cost;
    double* w;
    double* partial;    // fused update: [(R+1)][nparts] per-workgroup sums of w and w*u (null: not fused)
    double* statpart;   // [nparts][3]: min cost, max cost, zero-weight count
    int32_t nparts, fuse_update;
    int32_t prio_rotate, cu_count;   // k_rollout_pc / k_rollout_r3: see pc_rotate_priority()
    int32_t prune, fast_clamp;       // exact pruning of the window in the distance loop (pc_prune_window); 0 = off.  fast_clamp: clampd_fast() allowed (host side: sigma finite, umin <= umax)
    // deferred ccv_mppi_apply_partials_enqueue (K sharded over devices): when set, the warm start is pending_vec[1..] /
    // pending_vec[0]; every workgroup forms it while staging u* in LDS and workgroup 0 writes it (and sum w) back
    const double* pending_vec;
    double* nominal_w;
    double* stats_w;
    const ResidentFrame* frame;   // device-resident pose and window (null: x0, yaw_ref0 and the Window argument)
    unsigned long long* dbg;   // diagnostic builds only (-DCCV_DIAG, mppi_diag.h): the stamp buffer; null otherwise
};

// Every 64-byte line of the RolloutArgs block of the kernel arguments, requested at once.  The host wrote the block into
// device memory a moment ago and every launch starts with an invalidated scalar cache, so the first use of a field is a miss
// all the way to HBM (0.4 us on an idle chip, about 1 us on a busy one) -- and the compiler loads the fields where they are
// first needed: three to four of those latencies in a row before a wave has done anything (entry -> first arguments ->
// pointers -> loop bounds).  One dword of each line, all in flight together, then one wait: the lines are in the scalar
// cache, every later load of a field hits.
__device__ __forceinline__ void touch_rollout_args() {
    typedef const uint32_t __attribute__((address_space(4))) * ConstArgWords;
    ConstArgWords ka = (ConstArgWords)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t t = 0;
#pragma unroll
    for (int off = 0; off < (int)sizeof(RolloutArgs); off += 64) t |= ka[off / 4];
    asm volatile("" ::"s"(t));
}

// the kernel arguments with the resident pose substituted (wave-uniform scalar loads)
__device__ __forceinline__ RolloutArgs with_resident_pose(const RolloutArgs& Ak) {
    RolloutArgs A = Ak;
    if (Ak.frame) {
#pragma unroll
        for (int i = 0; i < 5; ++i) A.x0[i] = Ak.frame->x0[i];
        A.yaw_ref0 = Ak.frame->yaw_ref0;
    }
    return A;
}

// window coefficients -> LDS, padded to a multiple of 4 points with c = +inf (never the minimum)
template <class SH>
__device__ __forceinline__ void stage_window(const RolloutArgs& A, const Window& Wk, SH& sh, int nthreads,
                                             const int tid = threadIdx.x) {   // (tid: 0 .. nthreads-1 over the staging threads)
    const int H = A.H, H4 = (H + 3) & ~3;
    if (A.frame) {
        const Window& W = A.frame->W;
        for (int j = tid; j < H4; j += nthreads) {
            sh.ab[j] = j < H ? make_double2(W.a[j], W.b[j]) : make_double2(0.0, 0.0);
            sh.c[j] = j < H ? W.c[j] : INFINITY;
        }
    } else {
        for (int j = tid; j < H4; j += nthreads) {
            sh.ab[j] = j < H ? make_double2(Wk.a[j], Wk.b[j]) : make_double2(0.0, 0.0);
            sh.c[j] = j < H ? Wk.c[j] : INFINITY;
        }
    }
}

template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// x / d for a wave-uniform divisor d whose reciprocal r = RN(1 / d) was formed once: q = RN(x r), the exact remainder
// x - q d by FMA, one correction -- the correctly rounded quotient (Markstein's theorem for a faithful q and a correctly
// rounded r), in 3 instructions instead of the ~10 of an fp64 division.  The full-body ZMP term divides by dt twice per
// sample-step (fb:469, fb:479-481).  (Its third division, by mass * gravity_.z (fb:601), stays a division: one more pair of
// kernel-argument registers pushes the one-wave full-body kernel from 232 to 256 VGPRs with 17 spilled -- SGPR spills live
// in VGPR lanes -- and costs more than the division: C4 287 -> 294 us, measured.)  Valid while nothing overflows or
// vanishes on the way (the host admits the kernels that use it only for 1e-100 <= |dt| <= 1e100 and control bounds below
// 1e100, fast_trig_safe(); anything else runs the plain kernel with true divisions); a zero quotient may lose its sign.
__device__ __forceinline__ double div_uniform(const double x, const double d, const double r) {
    const double q = x * r;
    return fma(fma(-q, d, x), r, q);
}

// dd:62-67
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
// The same value in two instructions instead of six (two compares, four selects) -- for every v that is not NaN and bounds
// with lo <= hi (NaN bounds pass v through in both forms).  The reference's clamp passes a NaN control through (dd:98-99: both
// comparisons are false), min / max would replace it by a bound: the kernels use this form only where the host has checked
// sigma and the bounds and the kernel itself has found no NaN in the warm start (RolloutArgs::fast_clamp; a control is
// double(z) * sigma + u* with a finite z).  (-0 against a bound of +0 comes out as +0 here and -0 there: equal values.)
__device__ __forceinline__ double clampd_fast(double v, double lo, double hi) { return __builtin_fmin(__builtin_fmax(v, lo), hi); }
template <bool FAST>
__device__ __forceinline__ double clampd_as(double v, double lo, double hi) {
    if constexpr (FAST) return clampd_fast(v, lo, hi);
    else return clampd(v, lo, hi);
}

// min over the H window points of (a_j*px + b_j*py + c_j) for NV trajectory points held in registers.
// 2 FMA + 1 MIN per (point, window point): the O(K*H^2) core (calc_MinDistance, dd:183-192).
template <int NV, bool LDSWIN>
__device__ __forceinline__ void window_min(const double (&px)[kTU], const double (&py)[kTU], double (&m)[kTU], int H,
                                           const Window& W, const double2* s_ab, const double* s_c) {
#pragma unroll 2
    for (int j = 0; j < H; ++j) {
        double a, b, c;
        if constexpr (LDSWIN) {
            const double2 ab = s_ab[j];
            a = ab.x;
            b = ab.y;
            c = s_c[j];
        } else {
            a = W.a[j];
            b = W.b[j];
            c = W.c[j];
        }
        static_for<NV>([&](auto I) {
            constexpr int i = decltype(I)::value;
            m[i] = fmin(m[i], fma(a, px[i], fma(b, py[i], c)));
        });
    }
}

template <int MODEL, int SRC, bool LDSWIN>
__global__ __launch_bounds__(kBlock) void k_rollout_cost(const RolloutArgs A, const Window W) {
    constexpr int UD = udim_of(MODEL);
    __shared__ double2 s_ab[LDSWIN ? kMaxH : 1];
    __shared__ double s_c[LDSWIN ? kMaxH : 1];
    const int H = A.H;
    if constexpr (LDSWIN) {
        for (int j = threadIdx.x; j < H; j += kBlock) {
            s_ab[j] = make_double2(W.a[j], W.b[j]);
            s_c[j] = W.c[j];
        }
        __syncthreads();
    }
    const int k = blockIdx.x * kBlock + threadIdx.x;
    const bool live = k < A.K;
    const int kk = live ? k : A.K - 1;
    const size_t pitch = (size_t)A.pitch;
    const double dt = A.dt;
    const uint32_t kg = (uint32_t)(A.k_offset + kk);

    double x = A.x0[0], y = A.x0[1], yaw = A.x0[2];
    double roll = A.x0[3], pitchang = A.x0[4];
    double cost = 0.0;
    if constexpr (MODEL == CCV_MPPI_FULL_BODY) {
        // fb:408 -- identical for every sample
        cost += A.w_yaw * (yaw - A.yaw_ref0) * (yaw - A.yaw_ref0);
    }
    // full-body carry from step t-1 (the ZMP term of index t-1 needs controls t-1 and t, fb:468-486)
    double p_v = 0.0, p_rv = 0.0, p_sdir = 0.0, p_cdir = 1.0, p_c2 = 0.0, p_c3 = 0.0, p_ac = 0.0;
    const double fb_mgz = A.fb_mass * A.fb_gz;    // (mass*gravity_).z
    const double fb_den = A.fb_mass * A.fb_gz;    // mass*(gravity_-accel).dot(z); accel.z == 0 (fb:475,601)

    for (int t0 = 0; t0 < H; t0 += kTU) {
        double px[kTU], py[kTU];
        float zq[4] = {0.f, 0.f, 0.f, 0.f};
        static_for<kTU>([&](auto TT) {
            constexpr int tt = decltype(TT)::value;
            const int t = t0 + tt;
            px[tt] = x - A.x0[0];
            py[tt] = y - A.x0[1];
            if (t < H) {
                if (A.store_xy && live) {
                    A.xs[(size_t)t * pitch + k] = x;
                    A.ys[(size_t)t * pitch + k] = y;
                }
                if (t < H - 1) {
                    double u[UD];
                    static_for<UD>([&](auto D) {
                        constexpr int d = decltype(D)::value;
                        constexpr int nloc = tt * UD + d;
                        const size_t row = (size_t)(t * UD + d);
                        if constexpr (SRC == SRC_PHILOX) {
                            if constexpr ((nloc & 3) == 0) {
                                const Philox4 r = philox4x32_10(kg, (uint32_t)((t0 * UD + nloc) >> 2), A.iter_lo,
                                                                A.iter_hi, A.seed_lo, A.seed_hi);
                                box_muller_f32(r.x, r.y, zq[0], zq[1]);
                                box_muller_f32(r.z, r.w, zq[2], zq[3]);
                            }
                            // libstdc++ normal_distribution: ret * stddev + mean (dd:96-97), then clamp (dd:98-99)
                            double v = (double)zq[nloc & 3] * A.sigma + A.nominal[row];
                            v = clampd(v, A.umin[d], A.umax[d]);
                            if constexpr (MODEL == CCV_MPPI_FULL_BODY && d == 2) {
                                if (A.steer_off) v = 0.0;  // fb:517
                            }
                            u[d] = v;
                            if (A.store_u && live) A.u[row * pitch + k] = v;
                        } else {
                            u[d] = A.u[row * pitch + kk];
                        }
                    });
                    // ---- cost terms that do not need the window ----
                    if (A.do_cost) {
                        if constexpr (MODEL != CCV_MPPI_FULL_BODY) {
                            cost += A.w_v * ((u[0] - A.v_ref) * (u[0] - A.v_ref));  // dd:204-206
                        } else {
                            if (t < H - 2) {  // fb:409
                                cost += A.w_v * (u[0] - A.v_ref) * (u[0] - A.v_ref);          // fb:413
                                if (u[0] < 0.0) cost += A.w_back * u[0] * u[0];               // fb:420
                            }
                            if (t >= 1) {  // finish index t-1 <= H-3: ZMP (fb:468-485, 597-603) and roll-rate terms
                                const double drive_accel = (u[0] - p_v) / dt;                  // fb:469
                                const double ay = drive_accel * p_sdir + p_ac * p_cdir;        // fb:473
                                const double hgdot_x = (A.fb_Ixx * u[3] - A.fb_Ixx * p_rv) / dt;  // fb:479-481
                                const double mo_x = (p_c2 * fb_mgz + p_c3 * (A.fb_mass * ay)) - hgdot_x;  // fb:600
                                const double zmp_y = mo_x / fb_den;                            // fb:601
                                cost += A.w_zmp * zmp_y * zmp_y;                               // fb:416
                                cost += A.w_rollv * (u[3] - p_rv) * (u[3] - p_rv);             // fb:418
                            }
                        }
                    }
                    // ---- dynamics: explicit Euler (dd:104-109, sd:120-125, fb:445-452) ----
                    double hd = yaw;
                    if constexpr (MODEL != CCV_MPPI_DIFF_DRIVE) hd = yaw + u[2];
                    double sn, cs;
                    sincos(hd, &sn, &cs);
                    if constexpr (MODEL == CCV_MPPI_FULL_BODY) {
                        if (A.do_cost) {
                            double sd_, cd_, sr_, cr_;
                            sincos(u[2], &sd_, &cd_);
                            sincos(roll, &sr_, &cr_);
                            p_sdir = sd_;
                            p_cdir = cd_;
                            p_c2 = -A.fb_L * sr_;                      // CoM.y (fb:482)
                            p_c3 = A.fb_L * cos(pitchang) * cr_;       // CoM.z
                            p_ac = u[0] * u[1];                        // fb:471
                            p_v = u[0];
                            p_rv = u[3];
                        }
                    }
                    x = x + u[0] * cs * dt;
                    y = y + u[0] * sn * dt;
                    yaw = yaw + u[1] * dt;
                    if constexpr (MODEL == CCV_MPPI_FULL_BODY) {
                        roll = roll + u[3] * dt;
                        pitchang = pitchang + u[4] * dt;
                    }
                } else {
                    if constexpr (MODEL != CCV_MPPI_FULL_BODY) {
                        // t == H-1: the reference reads control index H-1, one past the end (dd:199,204); defined
                        // semantics: that element is 0.0
                        if (A.do_cost) cost += A.w_v * ((0.0 - A.v_ref) * (0.0 - A.v_ref));
                    }
                }
            }
        });
        if (A.do_cost) {
            // states that reach the path cost: all H for dd/sd (dd:199), the first H-2 for fb (fb:409)
            const int nstates = (MODEL == CCV_MPPI_FULL_BODY) ? H - 2 : H;
            const int nv = min(kTU, nstates - t0);
            if (nv > 0) {
                double m[kTU];
#pragma unroll
                for (int i = 0; i < kTU; ++i) m[i] = INFINITY;
                switch (nv) {
                    case 8: window_min<8, LDSWIN>(px, py, m, H, W, s_ab, s_c); break;
                    case 7: window_min<7, LDSWIN>(px, py, m, H, W, s_ab, s_c); break;
                    case 6: window_min<6, LDSWIN>(px, py, m, H, W, s_ab, s_c); break;
                    case 5: window_min<5, LDSWIN>(px, py, m, H, W, s_ab, s_c); break;
                    case 4: window_min<4, LDSWIN>(px, py, m, H, W, s_ab, s_c); break;
                    case 3: window_min<3, LDSWIN>(px, py, m, H, W, s_ab, s_c); break;
                    case 2: window_min<2, LDSWIN>(px, py, m, H, W, s_ab, s_c); break;
                    default: window_min<1, LDSWIN>(px, py, m, H, W, s_ab, s_c); break;
                }
#pragma unroll
                for (int i = 0; i < kTU; ++i) {
                    if (i < nv) {
                        // d^2 = |p|^2 + min_j(...), gate d <= 100 (dd:185), cost += path_weight*d*d (dd:206)
                        double d2 = m[i] + fma(px[i], px[i], py[i] * py[i]);
                        d2 = d2 < 1.0e4 ? fmax(d2, 0.0) : 1.0e4;   // NaN -> gate value, as `distance < min_distance` (dd:189) is false for NaN
                        cost += A.w_path * d2;
                    }
                }
            }
        }
    }
    if (A.do_cost && live) {
        A.cost[k] = cost;
        A.w[k] = exp(-cost / A.lambda);  // dd:219 (no min-cost shift)
    }
}

// stand-alone sampling(): one thread per (sample, Philox call) -> 4 consecutive rows of u.
template <int MODEL>
__global__ __launch_bounds__(kBlock) void k_sample(const RolloutArgs A) {
    constexpr int UD = udim_of(MODEL);
    const int k = blockIdx.x * kBlock + threadIdx.x;
    const int call = blockIdx.y;
    if (k >= A.K) return;
    const int ntot = (A.H - 1) * UD;
    const Philox4 r = philox4x32_10((uint32_t)(A.k_offset + k), (uint32_t)call, A.iter_lo, A.iter_hi, A.seed_lo, A.seed_hi);
    float z[4];
    box_muller_f32(r.x, r.y, z[0], z[1]);
    box_muller_f32(r.z, r.w, z[2], z[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = call * 4 + i;
        if (n < ntot) {
            const int d = n % UD;
            double v = (double)z[i] * A.sigma + A.nominal[n];
            v = clampd(v, A.umin[d], A.umax[d]);
            if (MODEL == CCV_MPPI_FULL_BODY && d == 2 && A.steer_off) v = 0.0;
            A.u[(size_t)n * A.pitch + k] = v;
        }
    }
}

// ---- cross-lane reductions (DPP) ---------------------------------------------------------------------------------

// Cross-lane moves of an fp64 value as two DPP moves (ALU latency) instead of ds_bpermute (an LDS round trip, ~250
// cycles per butterfly step with its wait): a 64-lane reduction drops from ~1500 to ~150 cycles.
template <int CTRL>
__device__ __forceinline__ double dpp_move(const double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]: lane ^ 1
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]: lane ^ 2
constexpr int kDppHalfMirror = 0x141;   // row_half_mirror: lane i <-> 7 - i within 8
constexpr int kDppMirror = 0x140;       // row_mirror: lane i <-> 15 - i within 16

__device__ __forceinline__ double lane_value(const double v, const int lane) {   // wave-uniform result
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// Fixed-shape reductions over the 64 lanes (every lane of a 16-lane row ends with the row's result; the four rows are
// combined as (r0 op r1) op (r2 op r3)); the result is wave-uniform.
#define CCV_WAVE_REDUCE(name, OP)                                                         \
    __device__ __forceinline__ double name(double v) {                                    \
        v = OP(v, dpp_move<kDppXor1>(v));                                                 \
        v = OP(v, dpp_move<kDppXor2>(v));                                                 \
        v = OP(v, dpp_move<kDppHalfMirror>(v));                                           \
        v = OP(v, dpp_move<kDppMirror>(v));                                               \
        return OP(OP(lane_value(v, 0), lane_value(v, 16)), OP(lane_value(v, 32), lane_value(v, 48))); \
    }
__device__ __forceinline__ double op_add(const double a, const double b) { return a + b; }
__device__ __forceinline__ double op_min(const double a, const double b) { return fmin(a, b); }
__device__ __forceinline__ double op_max(const double a, const double b) { return fmax(a, b); }
CCV_WAVE_REDUCE(wave_sum, op_add)
CCV_WAVE_REDUCE(wave_min, op_min)
CCV_WAVE_REDUCE(wave_max, op_max)
#undef CCV_WAVE_REDUCE

// Four 64-lane fp32 reductions at once, each one DPP-modified VALU instruction per step (the compiler does not fold a DPP
// move into v_min / v_max and puts a canonicalising v_max in front of every fminf): 24 instructions + 4 v_readlane for
// four results, against ~25 per fp64 reduction above.  The four chains are interleaved, so a step's DPP read of a register
// is four instructions behind its write (the DPP hazard needs two wait states; the leading s_nop covers the producer of
// the inputs, which the compiler's hazard recogniser cannot see into from outside the asm).  row_bcast:15 / :31 carry a
// row's result into the next row(s); lane 63 ends with the result of all 64 lanes.  NaN operands are ignored (IEEE
// minNum / maxNum); all 64 lanes must be active.
#define CCV_RED4_STEP(O0, O1, O2, O3, CTRL)                                                           \
    O0 " %0, %0, %0 " CTRL "\n\t" O1 " %1, %1, %1 " CTRL "\n\t" O2 " %2, %2, %2 " CTRL "\n\t" O3 " %3, %3, %3 " CTRL "\n\t"
#define CCV_RED4(NAME, O0, O1, O2, O3)                                                                \
    __device__ __forceinline__ void NAME(float& a, float& b, float& c, float& d) {                    \
        asm volatile("s_nop 1\n\t"                                                                    \
                     CCV_RED4_STEP(O0, O1, O2, O3, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")  \
                     CCV_RED4_STEP(O0, O1, O2, O3, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")  \
                     CCV_RED4_STEP(O0, O1, O2, O3, "row_half_mirror row_mask:0xf bank_mask:0xf")      \
                     CCV_RED4_STEP(O0, O1, O2, O3, "row_mirror row_mask:0xf bank_mask:0xf")           \
                     CCV_RED4_STEP(O0, O1, O2, O3, "row_bcast:15 row_mask:0xa bank_mask:0xf")         \
                     CCV_RED4_STEP(O0, O1, O2, O3, "row_bcast:31 row_mask:0xc bank_mask:0xf")         \
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));                                           \
        a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));                         \
        b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));                         \
        c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63));                         \
        d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));                         \
    }
CCV_RED4(wave_min_max_min_max_f32, "v_min_f32_dpp", "v_max_f32_dpp", "v_min_f32_dpp", "v_max_f32_dpp")
CCV_RED4(wave_min4_f32, "v_min_f32_dpp", "v_min_f32_dpp", "v_min_f32_dpp", "v_min_f32_dpp")
#undef CCV_RED4
#undef CCV_RED4_STEP

}  // namespace ccv
