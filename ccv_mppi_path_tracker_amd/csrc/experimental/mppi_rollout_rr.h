// EXPERIMENT (not built into the product; tools/ablate.py rr3=-DCCV_USE_RR,-DCCV_RR_WAVES=3): k_rollout_rr, "round-robin
// time blocks".  Same-device A/B against the shipped k_rollout_pc (profiles/README.md): equal at K = 65 536 (62.7 vs 64.0 us),
// 6-11 % slower at K >= 131 072 and for the full-body model, and it needs LDS spin-waits where k_rollout_pc needs only
// workgroup barriers -- so it is not shipped.
//
// One sample per lane.  A workgroup is kRrWaves waves that own the same 64 samples; the horizon is cut into time blocks of
// kTU = 8 steps and block b belongs to wave b % kRrWaves, which does ALL of the block's work:
//
//   N(b)  noise     Philox4x32-10 + Box-Muller -> clamped controls -> registers + HBM              [sampling()]
//   D(b)  dynamics  heading recurrence, 8 independent branch-free sin/cos, Euler positions, control / ZMP cost terms,
//                   x,y -> HBM                                   [predict_States(), non-window part of calc_Cost()]
//   C(b)  distance  squared distance of the block's 8 states to every point of the reference window, 2 FMA + 1 MIN per
//                   pair -- the O(K*H^2) core                                                 [calc_MinDistance()]
//
// The only thing a block needs from its predecessor is the lane's end state (x, y, yaw [, roll, pitch, ZMP carry]):
// D(b) waits on an LDS counter until D(b-1) has published it, everything else (N(b) before, C(b) after) overlaps with the
// other waves' blocks.  So the serial chain of a sample is the dynamics stage alone (about a fifth of the work), four
// waves (one per SIMD of the CU) work on four consecutive blocks at any time, there are no workgroup barriers in the
// time loop and no arithmetic is duplicated: per sample the operations are exactly those of k_rollout_cost, only the
// order in which one sample's cost terms are added differs.
//
// Why: at K = 65 536 one-sample-per-lane is exactly one wave per SIMD (1024 waves / 1024 SIMDs) and every stall is
// exposed (first version: 75 us per iteration; 41 us per 65 536 samples once 8 waves share a SIMD).
#pragma once
#include "../fast_trig.h"
#include "../mppi_kernels.h"

namespace ccv {

#ifndef CCV_RR_WAVES
#define CCV_RR_WAVES 4
#endif
constexpr int kRrWaves = CCV_RR_WAVES;
constexpr int kRrRedRows = 7;   // rows per batch of the fused update's LDS transpose
constexpr int kRrSamples = 64;

enum : int { MODE_FUSED = 0, MODE_ROLLOUT = 1, MODE_COST = 2 };
//   MODE_FUSED    device Philox noise, controls + states stored, cost + weight   (ccv_mppi_iterate*)
//   MODE_ROLLOUT  controls read from HBM, states stored, no cost                  (ccv_mppi_rollout)
//   MODE_COST     controls read from HBM, nothing stored but cost + weight        (ccv_mppi_weights)

template <int D>
__device__ __forceinline__ double arg5(const double (&v)[5]) { return v[D]; }

// lane state handed from block b to block b+1 (through LDS)
template <int MODEL>
struct RrState {
    double x, y, yaw;
    double roll, pitch;                                  // full body only
    double p_v, p_rv, p_sdir, p_cdir, p_c2, p_c3, p_ac;  // full body: step t-1 quantities for the ZMP term (fb:468-486)
};
template <int MODEL>
constexpr int kRrStateWords = MODEL == CCV_MPPI_FULL_BODY ? 12 : 3;

template <int MODEL>
struct RrShared {
    double2 ab[kMaxH + 4];                                 // window coefficients, padded to a multiple of 4 points
    double c[kMaxH + 4];
    double st[2][kRrStateWords<MODEL>][kRrSamples];        // end state of block b in slot b & 1
    double cost[kRrWaves][kRrSamples];
    double red[kRrWaves][kRrRedRows * (kRrSamples + 1)];   // fused update: per-wave transpose buffer
    int ready;                                             // number of blocks whose end state has been published
};

// ---------------------------------------------------------------------------------------------------------------
// N(b): controls of the 8 steps of a full block, in one basic block so the independent Philox / Box-Muller chains
// interleave.
// ---------------------------------------------------------------------------------------------------------------
template <int MODEL, int MODE>
__device__ __forceinline__ void rr_noise(const RolloutArgs& A, const int b, const int k, const int kk, const bool live,
                                         const uint32_t kg, double (&u)[kTU][udim_of(MODEL)]) {
    constexpr int UD = udim_of(MODEL);
    constexpr bool FB = MODEL == CCV_MPPI_FULL_BODY;
    constexpr int NCALL = kTU * UD / 4;
    const int t0 = b * kTU;
    const size_t pitch = (size_t)A.pitch;
    if constexpr (MODE == MODE_FUSED) {
        static_for<NCALL>([&](auto CC) {
            constexpr int c = decltype(CC)::value;
            const int n0 = t0 * UD + 4 * c;
            const double4 nom = *reinterpret_cast<const double4*>(A.nominal + n0);   // wave-uniform warm start u*[n0..n0+3]
            float z[4];
#if defined(CCV_ABL_NO_NOISE)
            z[0] = z[1] = z[2] = z[3] = (float)(kg & 1023u) * 1e-3f - 0.5f;
#else
            const Philox4 r = philox4x32_10(kg, (uint32_t)(n0 >> 2), A.iter_lo, A.iter_hi, A.seed_lo, A.seed_hi);
            box_muller_f32(r.x, r.y, z[0], z[1]);
            box_muller_f32(r.z, r.w, z[2], z[3]);
#endif
            const double mean[4] = {nom.x, nom.y, nom.z, nom.w};
            static_for<4>([&](auto II) {
                constexpr int i = decltype(II)::value;
                constexpr int nloc = 4 * c + i;
                constexpr int tt = nloc / UD, d = nloc % UD;
                // libstdc++ normal_distribution: ret * stddev + mean (dd:96-97), then clamp (dd:98-99)
                double v = (double)z[i] * A.sigma + mean[i];
                v = clampd(v, arg5<d>(A.umin), arg5<d>(A.umax));
                if constexpr (FB && d == 2) {
                    if (A.steer_off) v = 0.0;   // fb:517
                }
                u[tt][d] = v;
#if !defined(CCV_ABL_NO_STORE)
                if (live) A.u[(size_t)(n0 + i) * pitch + k] = v;
#endif
            });
        });
    } else {
#pragma unroll
        for (int tt = 0; tt < kTU; ++tt)
#pragma unroll
            for (int d = 0; d < UD; ++d) u[tt][d] = A.u[(size_t)((t0 + tt) * UD + d) * pitch + kk];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// D(b) for a full block: the serial link of the chain.  Same arithmetic as the step-by-step path, arranged in batches so
// the 8 independent sin/cos evaluations interleave.  (The host only selects this kernel when every reachable angle is
// inside the branch-free sin/cos range: ccv_mppi_capi.hip fast_trig_safe().)
// ---------------------------------------------------------------------------------------------------------------
template <int MODEL, int MODE>
__device__ __forceinline__ void rr_dynamics(const RolloutArgs& A, RrState<MODEL>& S, double& cost, const int b, const int k,
                                            const bool live, const double (&u)[kTU][udim_of(MODEL)], double (&px)[kTU],
                                            double (&py)[kTU]) {
    constexpr bool FB = MODEL == CCV_MPPI_FULL_BODY;
    constexpr bool COST = MODE != MODE_ROLLOUT;
    const int H = A.H;
    const int t0 = b * kTU;
    const size_t pitch = (size_t)A.pitch;
    const double dt = A.dt;
    // ---- 2. heading (roll, pitch) recurrences: yaw[t+1] = yaw[t] + w[t]*dt (dd:108, fb:449-451)
    double yawv[kTU + 1], rollv[FB ? kTU + 1 : 1], pitchv[FB ? kTU + 1 : 1];
    yawv[0] = S.yaw;
    if constexpr (FB) {
        rollv[0] = S.roll;
        pitchv[0] = S.pitch;
    }
#pragma unroll
    for (int tt = 0; tt < kTU; ++tt) {
        yawv[tt + 1] = yawv[tt] + u[tt][1] * dt;
        if constexpr (FB) {
            rollv[tt + 1] = rollv[tt] + u[tt][3] * dt;
            pitchv[tt + 1] = pitchv[tt] + u[tt][4] * dt;
        }
    }
    // (the host only selects this kernel when every reachable angle is inside the branch-free sin/cos range:
    //  ccv_mppi_capi.hip fast_trig_safe())
    double hd[kTU];
#pragma unroll
    for (int tt = 0; tt < kTU; ++tt) {
        hd[tt] = yawv[tt];
        if constexpr (MODEL != CCV_MPPI_DIFF_DRIVE) hd[tt] = yawv[tt] + u[tt][2];
    }
    // ---- 3. sin/cos of the 8 headings (independent chains)
    double sn[kTU], cs[kTU];
#pragma unroll
    for (int tt = 0; tt < kTU; ++tt) {
#if defined(CCV_ABL_NO_SINCOS)
        sn[tt] = hd[tt] * 0.5; cs[tt] = 1.0 - hd[tt] * 0.25;
#else
        fast_sincos(hd[tt], sn[tt], cs[tt]);
#endif
    }
    // ---- 4. cost terms that do not need the window
    if constexpr (COST) {
        if constexpr (!FB) {
#pragma unroll
            for (int tt = 0; tt < kTU; ++tt) cost += A.w_v * ((u[tt][0] - A.v_ref) * (u[tt][0] - A.v_ref));   // dd:204-206
        } else {
            const double mgz = A.fb_mass * A.fb_gz;   // (mass*gravity_).z
#pragma unroll
            for (int tt = 0; tt < kTU; ++tt) {
                const int t = t0 + tt;
                if (t < H - 2) {                                                      // fb:409
                    cost += A.w_v * (u[tt][0] - A.v_ref) * (u[tt][0] - A.v_ref);      // fb:413
                    if (u[tt][0] < 0.0) cost += A.w_back * u[tt][0] * u[tt][0];       // fb:420
                }
                if (t >= 1) {   // finish index t-1: ZMP (fb:468-485, 597-603) and roll-rate terms
                    const double drive_accel = (u[tt][0] - S.p_v) / dt;                          // fb:469
                    const double ay = drive_accel * S.p_sdir + S.p_ac * S.p_cdir;                // fb:473
                    const double hgdot_x = (A.fb_Ixx * u[tt][3] - A.fb_Ixx * S.p_rv) / dt;       // fb:479-481
                    const double mo_x = (S.p_c2 * mgz + S.p_c3 * (A.fb_mass * ay)) - hgdot_x;    // fb:600
                    const double zmp_y = mo_x / mgz;                                             // fb:601
                    cost += A.w_zmp * zmp_y * zmp_y;                                             // fb:416
                    cost += A.w_rollv * (u[tt][3] - S.p_rv) * (u[tt][3] - S.p_rv);               // fb:418
                }
                double sd_, cd_, sr_, cr_, sp_, cp_;
                fast_sincos(u[tt][2], sd_, cd_);
                fast_sincos(rollv[tt], sr_, cr_);
                fast_sincos(pitchv[tt], sp_, cp_);
                S.p_sdir = sd_;
                S.p_cdir = cd_;
                S.p_c2 = -A.fb_L * sr_;             // CoM.y (fb:482)
                S.p_c3 = A.fb_L * cp_ * cr_;        // CoM.z
                S.p_ac = u[tt][0] * u[tt][1];       // fb:471
                S.p_v = u[tt][0];
                S.p_rv = u[tt][3];
            }
        }
    }
    // ---- 5. positions (dd:106-107), (x,y) - pose kept in registers for C(b), x,y -> HBM
    double x = S.x, y = S.y;
#pragma unroll
    for (int tt = 0; tt < kTU; ++tt) {
        px[tt] = x - A.x0[0];
        py[tt] = y - A.x0[1];
        if constexpr (MODE != MODE_COST) {
#if !defined(CCV_ABL_NO_STORE)
            if (A.store_xy && live) {
                A.xs[(size_t)(t0 + tt) * pitch + k] = x;
                A.ys[(size_t)(t0 + tt) * pitch + k] = y;
            }
#endif
        }
        x = x + u[tt][0] * cs[tt] * dt;
        y = y + u[tt][0] * sn[tt] * dt;
    }
    S.x = x;
    S.y = y;
    S.yaw = yawv[kTU];
    if constexpr (FB) {
        S.roll = rollv[kTU];
        S.pitch = pitchv[kTU];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// N + D of the last, ragged block (fewer than 8 steps with controls): step by step with guards.
// ---------------------------------------------------------------------------------------------------------------
template <int MODEL, int MODE>
__device__ __forceinline__ void rr_tail_block(const RolloutArgs& A, RrState<MODEL>& S, double& cost, const int b, const int k,
                                              const int kk, const bool live, const uint32_t kg, double (&px)[kTU],
                                              double (&py)[kTU]) {
    constexpr bool FULL = false;
    constexpr int UD = udim_of(MODEL);
    constexpr bool FB = MODEL == CCV_MPPI_FULL_BODY;
    constexpr bool COST = MODE != MODE_ROLLOUT;
    const int H = A.H;
    const int t0 = b * kTU;
    const size_t pitch = (size_t)A.pitch;
    const double dt = A.dt;
    float zq[4] = {0.f, 0.f, 0.f, 0.f};
    double4 nom = make_double4(0.0, 0.0, 0.0, 0.0);
    static_for<kTU>([&](auto TT) {
        constexpr int tt = decltype(TT)::value;
        const int t = t0 + tt;
        px[tt] = S.x - A.x0[0];
        py[tt] = S.y - A.x0[1];
        if (FULL || t < H) {
            if constexpr (MODE != MODE_COST) {
#if !defined(CCV_ABL_NO_STORE)
                if (A.store_xy && live) {
                    A.xs[(size_t)t * pitch + k] = S.x;
                    A.ys[(size_t)t * pitch + k] = S.y;
                }
#endif
            }
            if (FULL || t < H - 1) {
                double u[UD];
                static_for<UD>([&](auto D) {
                    constexpr int d = decltype(D)::value;
                    constexpr int nloc = tt * UD + d;
                    const int n = t0 * UD + nloc;   // row = step*UD + dim
                    if constexpr (MODE == MODE_FUSED) {
                        if constexpr ((nloc & 3) == 0) {
                            // warm start u*[n .. n+3]: wave-uniform load, in flight while the Philox rounds run
                            nom = *reinterpret_cast<const double4*>(A.nominal + n);
#if defined(CCV_ABL_NO_NOISE)
                            zq[0] = zq[1] = zq[2] = zq[3] = (float)(kg & 1023u) * 1e-3f - 0.5f;
#else
                            const Philox4 r = philox4x32_10(kg, (uint32_t)(n >> 2), A.iter_lo, A.iter_hi, A.seed_lo, A.seed_hi);
                            box_muller_f32(r.x, r.y, zq[0], zq[1]);
                            box_muller_f32(r.z, r.w, zq[2], zq[3]);
#endif
                        }
                        constexpr int q = nloc & 3;
                        const double mean = q == 0 ? nom.x : q == 1 ? nom.y : q == 2 ? nom.z : nom.w;
                        // libstdc++ normal_distribution: ret * stddev + mean (dd:96-97), then clamp (dd:98-99)
                        double v = (double)zq[q] * A.sigma + mean;
                        v = clampd(v, arg5<d>(A.umin), arg5<d>(A.umax));
                        if constexpr (FB && d == 2) {
                            if (A.steer_off) v = 0.0;   // fb:517
                        }
                        u[d] = v;
#if !defined(CCV_ABL_NO_STORE)
                        if (live) A.u[(size_t)n * pitch + k] = v;
#endif
                    } else {
                        u[d] = A.u[(size_t)n * pitch + kk];
                    }
                });
                // ---- cost terms that do not need the window ----
                if constexpr (COST) {
                    if constexpr (!FB) {
                        cost += A.w_v * ((u[0] - A.v_ref) * (u[0] - A.v_ref));   // dd:204-206
                    } else {
                        if (t < H - 2) {                                          // fb:409
                            cost += A.w_v * (u[0] - A.v_ref) * (u[0] - A.v_ref);  // fb:413
                            if (u[0] < 0.0) cost += A.w_back * u[0] * u[0];       // fb:420
                        }
                        if (t >= 1) {   // finish index t-1 <= H-3: ZMP (fb:468-485, 597-603) and roll-rate terms
                            const double mgz = A.fb_mass * A.fb_gz;                               // (mass*gravity_).z
                            const double drive_accel = (u[0] - S.p_v) / dt;                       // fb:469
                            const double ay = drive_accel * S.p_sdir + S.p_ac * S.p_cdir;         // fb:473
                            const double hgdot_x = (A.fb_Ixx * u[3] - A.fb_Ixx * S.p_rv) / dt;    // fb:479-481
                            const double mo_x = (S.p_c2 * mgz + S.p_c3 * (A.fb_mass * ay)) - hgdot_x;  // fb:600
                            const double zmp_y = mo_x / mgz;                                      // fb:601 (accel.z == 0)
                            cost += A.w_zmp * zmp_y * zmp_y;                                      // fb:416
                            cost += A.w_rollv * (u[3] - S.p_rv) * (u[3] - S.p_rv);                // fb:418
                        }
                    }
                }
                // ---- dynamics: explicit Euler (dd:104-109, sd:120-125, fb:445-452) ----
                double hd = S.yaw;
                if constexpr (MODEL != CCV_MPPI_DIFF_DRIVE) hd = S.yaw + u[2];
                double sn, cs;
#if defined(CCV_ABL_NO_SINCOS)
                sn = hd * 0.5; cs = 1.0 - hd * 0.25;
#else
                fast_sincos(hd, sn, cs);
#endif
                if constexpr (FB && COST) {
                    double sd_, cd_, sr_, cr_, sp_, cp_;
                    fast_sincos(u[2], sd_, cd_);
                    fast_sincos(S.roll, sr_, cr_);
                    fast_sincos(S.pitch, sp_, cp_);
                    S.p_sdir = sd_;
                    S.p_cdir = cd_;
                    S.p_c2 = -A.fb_L * sr_;                    // CoM.y (fb:482)
                    S.p_c3 = A.fb_L * cp_ * cr_;               // CoM.z
                    S.p_ac = u[0] * u[1];                      // fb:471
                    S.p_v = u[0];
                    S.p_rv = u[3];
                }
                S.x = S.x + u[0] * cs * dt;
                S.y = S.y + u[0] * sn * dt;
                S.yaw = S.yaw + u[1] * dt;
                if constexpr (FB) {
                    S.roll = S.roll + u[3] * dt;
                    S.pitch = S.pitch + u[4] * dt;
                }
            } else {
                if constexpr (!FB && COST) {
                    // t == H-1: the reference reads control index H-1, one past the end (dd:199,204): defined as 0.0 (Q1)
                    cost += A.w_v * ((0.0 - A.v_ref) * (0.0 - A.v_ref));
                }
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------------------------
// C(b): min over the H window points of (a_j px + b_j py + c_j) for NV states held in registers, then the path cost.
// Straight fp64 FMA/MIN; the window is padded with c = +inf to a multiple of 4 points, so the loop body is four points
// with no remainder (LDS broadcast reads; the compiler's canonicalising max in front of fmin() is paid once per four
// minima).
// ---------------------------------------------------------------------------------------------------------------
template <int NV, int MODEL>
__device__ __forceinline__ void rr_distance(const RolloutArgs& A, const RrShared<MODEL>& sh, const double (&px)[kTU],
                                            const double (&py)[kTU], double& cost) {
    const int H4 = (A.H + 3) & ~3;
    double m[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) m[i] = INFINITY;
    for (int j = 0; j < H4; j += 4) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const double2 ab = sh.ab[j + jj];
            const double c = sh.c[j + jj];
#pragma unroll
            for (int i = 0; i < NV; ++i) m[i] = fmin(m[i], fma(ab.x, px[i], fma(ab.y, py[i], c)));
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        // d^2 = |p|^2 + min_j(...), gate d <= 100 (dd:185), cost += path_weight*d*d (dd:206)
        double d2 = m[i] + fma(px[i], px[i], py[i] * py[i]);
        d2 = d2 < 1.0e4 ? fmax(d2, 0.0) : 1.0e4;   // NaN -> gate value, as `distance < min_distance` (dd:189) is false for NaN
        cost += A.w_path * d2;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused first half of determine_OptimalSolution() (dd:228-237): this workgroup's share of sum_i w_i and
// sum_i w_i * u_i[t][d] for every control row, so that the K x (H-1) x u_dim controls are not streamed from HBM a second
// time by a separate kernel -- they are re-read here, by the CU that wrote them (L2 / Infinity Cache hits).
// Rows are dealt to the waves 7 at a time; a wave reduces its 7 rows through LDS: every lane drops w*u for each row,
// then lane (r, q) adds 16 of the 64 entries of row r and two shuffles finish the row.  Fixed order => reproducible.
// ---------------------------------------------------------------------------------------------------------------
template <int MODEL>
__device__ __forceinline__ void rr_partial_update(const RolloutArgs& A, RrShared<MODEL>& sh, const double wgt, const double total,
                                                  const int lane, const int wv, const int kk, const bool live) {
    constexpr int UD = udim_of(MODEL);
    constexpr int RB = kRrRedRows;
    constexpr int STRIDE = kRrSamples + 1;   // padded row: lanes (r, q) hit different banks
    const int R = (A.H - 1) * UD;
    const size_t pitch = (size_t)A.pitch;
    double* buf = sh.red[wv];
    const int rr = lane >> 2, q = lane & 3;
    for (int base = wv * RB; base < R; base += kRrWaves * RB) {
        const int nrows = min(RB, R - base);
        for (int r = 0; r < nrows; ++r) buf[r * STRIDE + lane] = wgt * A.u[(size_t)(base + r) * pitch + kk];
        __builtin_amdgcn_wave_barrier();
        double acc = 0.0;
        if (rr < nrows) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += buf[rr * STRIDE + q * 16 + i];
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (rr < nrows && q == 0) A.partial[(size_t)(base + rr) * A.nparts + blockIdx.x] = acc;
        __builtin_amdgcn_wave_barrier();
    }
    if (wv == 0) {
        const double sw = wave_sum(wgt);
        const double mn = wave_min(live ? total : INFINITY);
        const double mx = wave_max(live ? total : -INFINITY);
        const double nz = wave_sum((live && wgt == 0.0) ? 1.0 : 0.0);
        if (lane == 0) {
            A.partial[(size_t)R * A.nparts + blockIdx.x] = sw;
            A.statpart[blockIdx.x * 3 + 0] = mn;
            A.statpart[blockIdx.x * 3 + 1] = mx;
            A.statpart[blockIdx.x * 3 + 2] = nz;
        }
    }
}

template <int MODEL, int MODE>
__global__ __launch_bounds__(kRrWaves * 64, MODEL == CCV_MPPI_FULL_BODY ? 2 : kRrWaves) void k_rollout_rr(const RolloutArgs A, const Window W) {
    constexpr bool FB = MODEL == CCV_MPPI_FULL_BODY;
    constexpr bool COST = MODE != MODE_ROLLOUT;
    constexpr int UD = udim_of(MODEL);
    constexpr int NW = kRrStateWords<MODEL>;
    __shared__ RrShared<MODEL> sh;
    const int H = A.H;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (COST) {
        const int H4 = (H + 3) & ~3;
        for (int j = threadIdx.x; j < H4; j += kRrWaves * 64) {
            sh.ab[j] = j < H ? make_double2(W.a[j], W.b[j]) : make_double2(0.0, 0.0);
            sh.c[j] = j < H ? W.c[j] : INFINITY;
        }
    }
    if (threadIdx.x == 0) sh.ready = 0;
    const int k = blockIdx.x * kRrSamples + lane;
    const bool live = k < A.K;
    const int kk = live ? k : A.K - 1;
    const uint32_t kg = (uint32_t)(A.k_offset + kk);
    double cost = 0.0;
    if constexpr (FB && COST) {
        if (wv == 0) cost += A.w_yaw * (A.x0[2] - A.yaw_ref0) * (A.x0[2] - A.yaw_ref0);   // fb:408
    }
    const int nblocks = (H + kTU - 1) / kTU;
    const int nstates = FB ? H - 2 : H;   // states that reach the path cost (dd:199 / fb:409)
    __syncthreads();
    for (int b = wv; b < nblocks; b += kRrWaves) {
        const bool full = b * kTU + kTU <= H - 1;   // all 8 steps carry controls
        double u[kTU][UD];
        if (full) rr_noise<MODEL, MODE>(A, b, k, kk, live, kg, u);   // N(b): no dependence on the state
        // ---- wait for the end state of block b-1 (published by the wave that owns it)
        while (__hip_atomic_load(&sh.ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < b) __builtin_amdgcn_s_sleep(1);
        RrState<MODEL> S;
        if (b == 0) {
            S.x = A.x0[0];
            S.y = A.x0[1];
            S.yaw = A.x0[2];
            S.roll = A.x0[3];
            S.pitch = A.x0[4];
            S.p_v = S.p_rv = S.p_sdir = S.p_c2 = S.p_c3 = S.p_ac = 0.0;
            S.p_cdir = 1.0;
        } else {
            const double(*st)[kRrSamples] = sh.st[(b - 1) & 1];
            S.x = st[0][lane];
            S.y = st[1][lane];
            S.yaw = st[2][lane];
            if constexpr (FB) {
                S.roll = st[3][lane];
                S.pitch = st[4][lane];
                S.p_v = st[5][lane];
                S.p_rv = st[6][lane];
                S.p_sdir = st[7][lane];
                S.p_cdir = st[8][lane];
                S.p_c2 = st[9][lane];
                S.p_c3 = st[10][lane];
                S.p_ac = st[11][lane];
            }
        }
        // ---- D(b)
        double px[kTU], py[kTU];
        if (full) rr_dynamics<MODEL, MODE>(A, S, cost, b, k, live, u, px, py);
        else rr_tail_block<MODEL, MODE>(A, S, cost, b, k, kk, live, kg, px, py);
        if (b + 1 < nblocks) {
            double(*st)[kRrSamples] = sh.st[b & 1];
            st[0][lane] = S.x;
            st[1][lane] = S.y;
            st[2][lane] = S.yaw;
            if constexpr (FB) {
                st[3][lane] = S.roll;
                st[4][lane] = S.pitch;
                st[5][lane] = S.p_v;
                st[6][lane] = S.p_rv;
                st[7][lane] = S.p_sdir;
                st[8][lane] = S.p_cdir;
                st[9][lane] = S.p_c2;
                st[10][lane] = S.p_c3;
                st[11][lane] = S.p_ac;
            }
            // publish: the LDS writes above are ordered before the counter update (release, workgroup scope)
            if (lane == 0) __hip_atomic_store(&sh.ready, b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // ---- C(b)
        if constexpr (COST) {
#if defined(CCV_ABL_NO_DIST)
            const int nv = 0;
#else
            const int nv = min(kTU, nstates - b * kTU);
#endif
            if (nv == kTU) rr_distance<kTU, MODEL>(A, sh, px, py, cost);
            else if (nv > 0) {
                switch (nv) {
                    case 7: rr_distance<7, MODEL>(A, sh, px, py, cost); break;
                    case 6: rr_distance<6, MODEL>(A, sh, px, py, cost); break;
                    case 5: rr_distance<5, MODEL>(A, sh, px, py, cost); break;
                    case 4: rr_distance<4, MODEL>(A, sh, px, py, cost); break;
                    case 3: rr_distance<3, MODEL>(A, sh, px, py, cost); break;
                    case 2: rr_distance<2, MODEL>(A, sh, px, py, cost); break;
                    default: rr_distance<1, MODEL>(A, sh, px, py, cost); break;
                }
            }
        }
    }
    if constexpr (COST) {
        sh.cost[wv][lane] = cost;
        __syncthreads();
        double total = sh.cost[0][lane];
#pragma unroll
        for (int q = 1; q < kRrWaves; ++q) total += sh.cost[q][lane];
        const double wgt = live ? exp(-total / A.lambda) : 0.0;   // dd:219 (no min-cost shift)
        if (wv == 0 && live) {
            A.cost[k] = total;
            A.w[k] = wgt;
        }
        if (A.fuse_update) rr_partial_update<MODEL>(A, sh, wgt, total, lane, wv, kk, live);
    }
}

// aliases so that the C-ABI translation unit can launch either kernel
constexpr int kPcWaves = kRrWaves;
constexpr int kPcSamples = kRrSamples;
#define k_rollout_pc k_rollout_rr

}  // namespace ccv
