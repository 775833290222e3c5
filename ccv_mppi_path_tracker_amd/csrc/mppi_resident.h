// Device-resident closed loop: the per-tick prologue of the reference's run() loop on the GPU, so that consecutive MPPI
// iterations need nothing from the host but their launches.
//
//   k_advance   (optional) the commanded motion u*[0] applied to the pose for one period -- the kinematic plant of the
//               closed-loop harness, the Euler model of predict_NextState() (dd:104-109, sd:120-125, fb:445-452 pose part)
//               get_CurrentIndex()   nearest path pose inside the 100 m gate      dd:126-140  sd:142-156  fb:335-349
//               calc_RefPath()       window of H poses, stride v_ref*dt/resolution dd:156-181  sd:172-197  fb:365-392
//               + the distance coefficients of the window relative to the pose (fill_window() in ccv_mppi_capi.hip)
//
// One workgroup; the path (a few hundred to a few thousand poses) is scanned by its 1024 threads and the (distance, index)
// pairs are reduced so that the result is the one the reference's serial scan gives: the lowest index among the poses at
// the smallest distance, 0 when none is inside the gate.  x_ref / y_ref / index and the coefficients are bit-identical to
// ccv_mppi_calc_ref_path() + fill_window() on the host (same operations, no contraction); the plant uses fast_sincos(),
// which ccv_mppi_plant_step() restates operation by operation on the host; yaw_ref0 comes from the device atan2 and may
// differ from libm's in the last place (only fb:408 reads it).
#pragma once
#include "fast_trig.h"
#include "mppi_update.h"

namespace ccv {

struct AdvanceArgs {
    ResidentFrame* frame;
    const double* path_x;
    const double* path_y;
    const double* nominal;   // u* (row n = t*u_dim + d): the command is rows 0..u_dim-1
    double* trace;           // [trace_cap][6]: pose (5) and index after each launch (ring), or null
    double dt, v_ref, resolution;
    int32_t n_path, H, model, advance, trace_cap;
};

constexpr int kAdvanceThreads = 1024;   // one workgroup; a path of a few thousand poses is one or two batches of loads per thread

// NT threads of one workgroup; cmd: the command u*[0][0 .. u_dim) (read only when A.advance)
template <int NT>
__device__ __forceinline__ void advance_body(const AdvanceArgs& A, const double* cmd) {
    __shared__ double s_d[NT / 64];
    __shared__ int s_i[NT / 64];
    __shared__ int s_start;
    ResidentFrame& F = *A.frame;
    // ---- pose (every thread computes it: wave-uniform, no hand-off)
    double x = F.x0[0], y = F.x0[1], yaw = F.x0[2], roll = F.x0[3], pitch = F.x0[4];
    if (A.advance) {
        const double v = cmd[0], w = cmd[1];
        const double heading = A.model == CCV_MPPI_DIFF_DRIVE ? yaw : yaw + cmd[2];
        double sn, cs;
        fast_sincos(heading, sn, cs);
        x = x + v * cs * A.dt;
        y = y + v * sn * A.dt;
        yaw = rebase_angle(yaw + w * A.dt);
        if (A.model == CCV_MPPI_FULL_BODY) {
            roll = rebase_angle(roll + cmd[3] * A.dt);
            pitch = rebase_angle(pitch + cmd[4] * A.dt);
        }
    }
    // ---- get_CurrentIndex(): strict '<' against a running minimum that starts at the 100 m gate
    double best_d = 100.0;
    int best_i = -1;
    constexpr int kBatch = 4;   // loads in flight per thread: the scan is a chain of memory latencies otherwise
    for (int i0 = threadIdx.x; i0 < A.n_path; i0 += NT * kBatch) {
        double qx[kBatch], qy[kBatch];
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            const int i = min(i0 + b * NT, A.n_path - 1);
            qx[b] = A.path_x[i];
            qy[b] = A.path_y[i];
        }
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            const int i = i0 + b * NT;
            const double ex = x - qx[b], ey = y - qy[b];
            const double d = sqrt(ex * ex + ey * ey);
            if (i < A.n_path && d < best_d) {   // (ascending i within a thread: the first of equal distances stays)
                best_d = d;
                best_i = i;
            }
        }
    }
    // (distance, index) minimum, lexicographic: the smallest distance, and among equal distances the smallest index --
    // what the serial scan's strict '<' keeps.  Two wave reductions per level (DPP), one LDS hand-off between the levels.
    auto lexmin = [](double d, int i, double& d_out, int& i_out) {
        const double dm = wave_min(d);                                   // (no candidate: d = 100, the gate)
        const double im = wave_min((i >= 0 && d == dm) ? (double)i : 1.0e300);
        d_out = dm;
        i_out = im < 1.0e299 ? (int)im : -1;
    };
    double wd;
    int wi;
    lexmin(best_d, best_i, wd, wi);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        s_d[wave] = wd;
        s_i[wave] = wi;
    }
    __syncthreads();   // (also: every thread has read the old pose)
    if (wave == 0) {
        constexpr int NW = NT / 64;
        const double d2 = lane < NW ? s_d[lane] : 100.0;
        const int i2 = lane < NW ? s_i[lane] : -1;
        double fd;
        int fi;
        lexmin(d2, i2, fd, fi);
        if (lane == 0) s_start = fi < 0 ? 0 : fi;
    }
    __syncthreads();
    const int start = s_start;
    // ---- calc_RefPath(): the index is the truncation of a double; past the end the final pose repeats
    const double stride = A.v_ref * A.dt / A.resolution;
    for (int i = threadIdx.x; i < A.H; i += NT) {
        const int idx = (int)(start + i * stride);   // (the host admits 0 < dt < inf only: idx >= 0)
        const int src = idx < A.n_path ? idx : A.n_path - 1;
        const double xr = A.path_x[src], yr = A.path_y[src];
        F.x_ref[i] = xr;
        F.y_ref[i] = yr;
        const double xl = xr - x, yl = yr - y;
        F.W.a[i] = -2.0 * xl;
        F.W.b[i] = -2.0 * yl;
        F.W.c[i] = xl * xl + yl * yl;
    }
    if (threadIdx.x == 0) {
        const int i1 = (int)(start + 1 * stride), i0 = (int)(start + 0 * stride);
        const int s1 = i1 < A.n_path ? i1 : A.n_path - 1, s0 = i0 < A.n_path ? i0 : A.n_path - 1;
        F.yaw_ref0 = atan2(A.path_y[s1] - A.path_y[s0], A.path_x[s1] - A.path_x[s0]);
        F.x0[0] = x;
        F.x0[1] = y;
        F.x0[2] = yaw;
        F.x0[3] = roll;
        F.x0[4] = pitch;
        F.index = start;
        const int n = F.steps;
        F.steps = n + 1;
        if (A.trace) {
            double* t = A.trace + (size_t)(n % A.trace_cap) * 6;
            t[0] = x;
            t[1] = y;
            t[2] = yaw;
            t[3] = roll;
            t[4] = pitch;
            t[5] = (double)start;
        }
    }
}

__global__ __launch_bounds__(kAdvanceThreads) void k_advance(const AdvanceArgs A) { advance_body<kAdvanceThreads>(A, A.nominal); }

// The update of tick i and the prologue of tick i+1 in ONE launch (the closed loop then costs two launches per tick, not
// three): blocks 0 .. finalize_blocks(R)-1 are k_finalize; one more block forms the command u*[0][d] = V_d / S from the same
// partial sums in the same order (the same bits the finalize waves write into the warm start -- it cannot wait for them)
// and runs the prologue with it.
__global__ __launch_bounds__(kBlock) void k_finalize_advance(const FinalizeArgs F, const AdvanceArgs A) {
    if ((int)blockIdx.x < finalize_blocks(F.R)) {
        finalize_rows(F);
        return;
    }
    __shared__ double cmd[CCV_MPPI_MAX_UDIM + 3];
    if (A.advance) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, ud = udim_of(A.model);
        for (int d = wv; d < ud; d += kBlock / 64) {
            double s, v;
            lane_partial_sum2(F.partial + (size_t)F.R * F.nchunks, F.partial + (size_t)d * F.nchunks, F.nchunks, lane, s, v);
            s = wave_sum(s);
            v = wave_sum(v);
            if (lane == 0) cmd[d] = v / s;
        }
    }
    __syncthreads();
    advance_body<kBlock>(A, cmd);
}

}  // namespace ccv
