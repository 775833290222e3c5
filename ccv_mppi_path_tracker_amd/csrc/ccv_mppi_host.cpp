// Host prologue of the MPPI path (include/ccv_mppi_host.h): window builder, synthetic paths, kinematic plant.
// Plain C++; no device work.  Reference locations are cited in the header.
#include "../../include/ccv_mppi_host.h"

#include <cmath>

#include "../../include/ccv_mppi.h"

namespace {

// nearest pose of the path, accepted only inside a 100 m gate; 0 when nothing qualifies (dd:126-140)
int nearest_index(const double* px, const double* py, int n, double x, double y) {
    int best = 0;
    double best_d = 100.0;
    for (int i = 0; i < n; ++i) {
        const double ex = x - px[i], ey = y - py[i];
        const double d = std::sqrt(ex * ex + ey * ey);
        if (d < best_d) {
            best_d = d;
            best = i;
        }
    }
    return best;
}

// sin and cos as the device computes them (csrc/fast_trig.h fast_sincos, restated operation by operation: three-step
// Cody-Waite reduction by pi/2, fdlibm kernel polynomials, quadrant selection), so that a pose advanced here and one
// advanced by k_advance on the device are the same bits.  |x| <= 1e5 as on the device; at most 1 ulp from libm.
// csrc/fast_trig.h rebase_angle(), restated (this file is compiled without the HIP headers)
double rebase_angle(double a) {
    if (!(std::fabs(a) > 1.0e4)) return a;
    const double n = std::rint(a * 1.59154943091895345608e-01);
    a = std::fma(-n, 6.28318530717958623200e+00, a);
    return std::fma(-n, 2.44929359829470641435e-16, a);
}

void spec_sincos(double x, double& s, double& c) {
    const double fn = std::rint(x * 6.36619772367581382433e-01);
    double r = std::fma(-fn, 1.57079632673412561417e+00, x);
    r = std::fma(-fn, 6.07710050630396597660e-11, r);
    r = std::fma(-fn, 2.02226624879595063154e-21, r);
    const double z = r * r;
    double ps = std::fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = std::fma(z, ps, 2.75573137070700676789e-06);
    ps = std::fma(z, ps, -1.98412698298579493134e-04);
    ps = std::fma(z, ps, 8.33333333332248946124e-03);
    ps = std::fma(z, ps, -1.66666666666666324348e-01);
    const double sr = std::fma(z * r, ps, r);
    double pc = std::fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = std::fma(z, pc, -2.75573143513906633035e-07);
    pc = std::fma(z, pc, 2.48015872894767294178e-05);
    pc = std::fma(z, pc, -1.38888888888741095749e-03);
    pc = std::fma(z, pc, 4.16666666666666019037e-02);
    const double cr = std::fma(z, std::fma(z, pc, -0.5), 1.0);
    const int q = static_cast<int>(fn);
    const double sa = (q & 1) ? cr : sr;
    const double ca = (q & 1) ? sr : cr;
    s = (q & 2) ? -sa : sa;
    c = ((q + 1) & 2) ? -ca : ca;
}

}  // namespace

extern "C" {

int ccv_mppi_calc_ref_path(const double* path_x, const double* path_y, int32_t n_path, double cur_x, double cur_y,
                           double v_ref, double dt, double resolution, int32_t horizon, double* x_ref, double* y_ref,
                           double* yaw_ref) {
    if (!path_x || !path_y || !x_ref || !y_ref || !yaw_ref || n_path < 1 || horizon < 2) return CCV_MPPI_ERR_INVALID_ARG;
    // The window index start + i * stride is the truncation of a double (dd:160-163).  The node takes dt from its clock
    // (dd:346-348); a negative or non-finite dt, v_ref or resolution makes that index negative or undefined (the reference
    // then reads outside path_): refused here, and in the same way by the device-resident prologue.  dt == 0 (two ticks
    // inside one clock tick) is defined -- stride 0, the window is H copies of the nearest pose, exactly what v_ref = 0
    // gives -- and admitted.
    {
        const double stride_chk = v_ref * dt / resolution;
        if (!(dt >= 0.0) || !std::isfinite(stride_chk) || stride_chk < 0.0 || stride_chk * horizon > 2.0e9) return CCV_MPPI_ERR_INVALID_ARG;
    }
    const int start = nearest_index(path_x, path_y, n_path, cur_x, cur_y);
    // window stride in path indices; the index is the truncation of a double (dd:160-163)
    const double stride = v_ref * dt / resolution;
    const int last = n_path - 1;
    for (int i = 0; i < horizon; ++i) {
        const int idx = static_cast<int>(start + i * stride);
        const int src = idx < n_path ? idx : last;  // past the end: repeat the final pose (dd:169-172)
        x_ref[i] = path_x[src];
        y_ref[i] = path_y[src];
    }
    for (int i = 0; i + 1 < horizon; ++i) yaw_ref[i] = std::atan2(y_ref[i + 1] - y_ref[i], x_ref[i + 1] - x_ref[i]);
    return start;
}

int ccv_mppi_path_cosine(const double* A, const double* omega, const double* delta, double resolution,
                         double course_length, double init_x, double init_y, double* path_x, double* path_y,
                         int32_t cap) {
    if (!A || !omega || !delta || !path_x || !path_y || !(resolution > 0.0)) return CCV_MPPI_ERR_INVALID_ARG;
    int n = 0;
    // the arc parameter is accumulated, not multiplied, so the pose count follows the reference's rounding
    for (double s = 0.0; s < course_length && n < cap; s += resolution, ++n) {
        double y = A[0] * std::cos(2 * M_PI * omega[0] * s + delta[0]) + A[1] * std::cos(2 * M_PI * omega[1] * s + delta[1]) +
                   A[2] * std::cos(2 * M_PI * omega[2] * s + delta[2]) + init_y;
        y -= A[0] + A[1] + A[2];
        path_x[n] = init_x + s;
        path_y[n] = y;
    }
    return n;
}

int ccv_mppi_path_dkan(double resolution, double* path_x, double* path_y, int32_t cap) {
    if (!path_x || !path_y || !(resolution > 0.0)) return CCV_MPPI_ERR_INVALID_ARG;
    static const double corner[4][2] = {{0.0, 0.0}, {17.7, 0.0}, {17.7, 8.0}, {0.0, 8.0}};
    int n = 0;
    for (int leg = 0; leg < 3; ++leg) {
        const double dx = corner[leg + 1][0] - corner[leg][0];
        const double dy = corner[leg + 1][1] - corner[leg][1];
        const double len = std::sqrt(dx * dx + dy * dy);
        for (double s = 0.0; s < len; s += resolution) {
            if (n >= cap) return n;
            path_x[n] = corner[leg][0] + s * dx / len;
            path_y[n] = corner[leg][1] + s * dy / len;
            ++n;
        }
    }
    return n;
}

int ccv_mppi_plant_step(int32_t model, double* state, const double* u, double dt) {
    if (!state || !u || model < CCV_MPPI_DIFF_DRIVE || model > CCV_MPPI_FULL_BODY) return CCV_MPPI_ERR_INVALID_ARG;
    const double heading = model == CCV_MPPI_DIFF_DRIVE ? state[2] : state[2] + u[2];
    if (!(std::fabs(heading) <= 1.0e5)) return CCV_MPPI_ERR_INVALID_ARG;   // (the range spec_sincos is specified for)
    double sn, cs;
    spec_sincos(heading, sn, cs);
    state[0] = state[0] + u[0] * cs * dt;
    state[1] = state[1] + u[0] * sn * dt;
    // (angles beyond +-1e4 rad are taken modulo 2 pi, exactly as k_advance does on the device: csrc/fast_trig.h)
    state[2] = rebase_angle(state[2] + u[1] * dt);
    if (model == CCV_MPPI_FULL_BODY) {
        state[3] = rebase_angle(state[3] + u[3] * dt);
        state[4] = rebase_angle(state[4] + u[4] * dt);
    }
    return CCV_MPPI_OK;
}

}  // extern "C"
