// Branch-free fp64 sin/cos for the rollout (predict_NextState, dd:106-107).
//
// OCML's sincos() carries a large-argument (Payne-Hanek) branch, which splits the unrolled time block into basic blocks
// and stops the scheduler from interleaving the eight independent heading evaluations of a block.  This version is
// straight-line code: Cody-Waite reduction by pi/2 in three FMA steps (exact products for |n| < 2^20), then the fdlibm
// kernel polynomials (Sun Microsystems' __kernel_sin/__kernel_cos minimax coefficients, |error| < 2^-58 on [-pi/4, pi/4]).
// Valid for |x| <= kFastTrigLimit; callers test the whole wave with fast_trig_ok() and fall back to sincos() otherwise.
// Within 1 ulp of the correctly rounded result for the sine, 1.3 ulp for the cosine (0.8 ulp of the evaluation + rounding).
#pragma once
#include <hip/hip_runtime.h>

#include "noise_spec.h"   // CCV_KEEP_ORDER

namespace ccv {

constexpr double kFastTrigLimit = 1.0e5;

// The closed-loop plant (k_advance, ccv_mppi_plant_step) integrates yaw / roll / pitch without bound, where the real node
// reads them from tf in [-pi, pi].  An angle that leaves +-kAngleRebase is taken modulo 2 pi (two-step Cody-Waite, the
// same operations on host and device), so a loop of any length stays inside the branch-free sin/cos range.
constexpr double kAngleRebase = 1.0e4;
__host__ __device__ inline double rebase_angle(double a) {
    if (!(fabs(a) > kAngleRebase)) return a;   // (NaN: unchanged)
    const double n = rint(a * 1.59154943091895345608e-01);     // 1 / (2 pi)
    a = fma(-n, 6.28318530717958623200e+00, a);               // 2 pi, rounded to double
    return fma(-n, 2.44929359829470641435e-16, a);            // 2 pi - the above
}

__device__ __forceinline__ bool fast_trig_ok(double x) { return fabs(x) <= kFastTrigLimit; }   // false for NaN/Inf

__device__ __forceinline__ void fast_sincos(double x, double& s, double& c) {
    const double fn = __builtin_rint(x * 6.36619772367581382433e-01);   // n = nearest integer to x * 2/pi
    double r = fma(-fn, 1.57079632673412561417e+00, x);                  // pi/2, first 33 bits: exact product
    r = fma(-fn, 6.07710050630396597660e-11, r);                         // next 33 bits
    r = fma(-fn, 2.02226624879595063154e-21, r);                         // tail
    const double z = r * r;
    // sin(r) = r + r^3 * (S1 + z*(S2 + ... ))
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sr = fma(z * r, ps, r);
    // cos(r) = 1 - z/2 + z^2 * (C1 + z*(C2 + ...))
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    // cos r = 1 + z (-1/2 + z pc) in two fused steps: within 0.8 ulp on |r| <= pi/4 (fdlibm's compensated sum: 0.5 ulp, seven
    // operations); the host restatement (ccv_mppi_host.cpp spec_sincos) does the same
    const double cr = fma(z, fma(z, pc, -0.5), 1.0);
    // quadrant
    const int q = (int)fn;
    const double sa = (q & 1) ? cr : sr;
    const double ca = (q & 1) ? sr : cr;
    s = (q & 2) ? -sa : sa;
    c = ((q + 1) & 2) ? -ca : ca;
}

// N independent evaluations of fast_sincos, stage by stage: the same arithmetic per element, arranged so that the N
// serial chains (reduction, two polynomials) interleave in program order.
template <int N>
__device__ __forceinline__ void fast_sincos_n(const double (&x)[N], double (&s)[N], double (&c)[N]) {
    double fn[N], r[N], z[N], ps[N], pc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) fn[i] = __builtin_rint(x[i] * 6.36619772367581382433e-01);
    CCV_KEEP_ORDER();
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = fma(-fn[i], 1.57079632673412561417e+00, x[i]);
    CCV_KEEP_ORDER();
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = fma(-fn[i], 6.07710050630396597660e-11, r[i]);
    CCV_KEEP_ORDER();
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = fma(-fn[i], 2.02226624879595063154e-21, r[i]);
    CCV_KEEP_ORDER();
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = r[i] * r[i];
    CCV_KEEP_ORDER();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ps[i] = fma(z[i], 1.58969099521155010221e-10, -2.50507602534068634195e-08);
        pc[i] = fma(z[i], -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    }
    CCV_KEEP_ORDER();
#define CCV_STAGE2(cs_, cc_)                                         \
    _Pragma("unroll") for (int i = 0; i < N; ++i) {                  \
        ps[i] = fma(z[i], ps[i], cs_);                               \
        pc[i] = fma(z[i], pc[i], cc_);                               \
    }                                                                \
    CCV_KEEP_ORDER();
    CCV_STAGE2(2.75573137070700676789e-06, -2.75573143513906633035e-07)
    CCV_STAGE2(-1.98412698298579493134e-04, 2.48015872894767294178e-05)
    CCV_STAGE2(8.33333333332248946124e-03, -1.38888888888741095749e-03)
    CCV_STAGE2(-1.66666666666666324348e-01, 4.16666666666666019037e-02)
#undef CCV_STAGE2
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double sr = fma(z[i] * r[i], ps[i], r[i]);
        const double cr = fma(z[i], fma(z[i], pc[i], -0.5), 1.0);
        const int q = (int)fn[i];
        const double sa = (q & 1) ? cr : sr;
        const double ca = (q & 1) ? sr : cr;
        s[i] = (q & 2) ? -sa : sa;
        c[i] = ((q + 1) & 2) ? -ca : ca;
    }
}

// sin and cos of N small angles |r| <= pi/4: the kernel polynomials of fast_sincos without range reduction and quadrant
// logic, stage by stage.  Used to ADVANCE a heading's (sin, cos) by the step's turn angle instead of evaluating them from
// the accumulated heading: (s, c) <- (s cos d + c sin d, c cos d - s sin d).
constexpr double kSmallTurnLimit = 7.85398163397448279e-01;   // pi/4
template <int N>
__device__ __forceinline__ void kernel_sincos_n(const double (&r)[N], double (&s)[N], double (&c)[N]) {
    double z[N], ps[N], pc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = r[i] * r[i];
    CCV_KEEP_ORDER();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ps[i] = fma(z[i], 1.58969099521155010221e-10, -2.50507602534068634195e-08);
        pc[i] = fma(z[i], -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    }
    CCV_KEEP_ORDER();
#define CCV_STAGE2(cs_, cc_)                                         \
    _Pragma("unroll") for (int i = 0; i < N; ++i) {                  \
        ps[i] = fma(z[i], ps[i], cs_);                               \
        pc[i] = fma(z[i], pc[i], cc_);                               \
    }                                                                \
    CCV_KEEP_ORDER();
    CCV_STAGE2(2.75573137070700676789e-06, -2.75573143513906633035e-07)
    CCV_STAGE2(-1.98412698298579493134e-04, 2.48015872894767294178e-05)
    CCV_STAGE2(8.33333333332248946124e-03, -1.38888888888741095749e-03)
    CCV_STAGE2(-1.66666666666666324348e-01, 4.16666666666666019037e-02)
#undef CCV_STAGE2
#pragma unroll
    for (int i = 0; i < N; ++i) {
        s[i] = fma(z[i] * r[i], ps[i], r[i]);
        // cos r = 1 + z (-1/2 + z pc) in two fused steps, as in fast_sincos
        c[i] = fma(z[i], fma(z[i], pc[i], -0.5), 1.0);
    }
}

}  // namespace ccv
