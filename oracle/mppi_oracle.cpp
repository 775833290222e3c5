// =============================================================================
// TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the MPPI hot path of
// YoshikiMaekawa2000/ccv_mppi_path_tracker.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this; the product (libccv_mppi_hip.so)
// never does.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
//  and cannot be compiled in this image (it needs ROS/tf/catkin and
// Eigen, none of which is installed; writing stand-in headers is not allowed).
// This file therefore restates the algorithm from a reading of the sources; each
// function cites the reference file:line it follows.  libstdc++'s <random>
// (std::mt19937 + std::normal_distribution, the only third-party arithmetic on the
// path besides Eigen 3-vectors) IS present and is used directly, not restated.
//
// Abbreviations: dd = src/diff_drive_mppi.cpp, sd = src/steering_diff_drive_mppi.cpp,
// fb = src/full_body_mppi.cpp, *.h = include/ccv_mppi_path_tracker/*_mppi.h.
//
// Defined semantics for the reference's out-of-bounds index H-1 on the control
// vectors (dd:199-204, dd:228-236, sd:215-220, sd:244-254, fb:311-325):
// the control vectors here carry one extra "phantom" element that reads 0.0.
// =============================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "philox_normal.h"

namespace {

enum Model { DIFF = 0, STEER = 1, FULLBODY = 2 };

struct OrcConfig {
    int32_t model;       // 0 dd, 1 sd, 2 fb
    int32_t K;           // num_samples_
    int32_t H;           // horizon_ (number of states; H-1 control steps)
    int32_t roll_off;    // fb:43-46
    int32_t steer_off;   // fb:517
    int32_t pad_;
    double sigma;        // control_noise_
    double lambda;       // lambda_
    double v_ref;
    double u_min[5];     // per control dim, declaration order of the reference
    double u_max[5];
    double path_weight, v_weight, zmp_weight, roll_v_weight, back_weight, yaw_weight;
};

inline int udim_of(int model) { return model == DIFF ? 2 : (model == STEER ? 3 : 5); }

// dd.h:20-50, sd.h:21-54, fb.h:34-65 -- one heap vector per quantity per sample (AoS of vectors).
struct RobotStates {
    std::vector<double> x_, y_, yaw_, roll_, pitch_;   // states [H]
    std::vector<double> u_[5];                         // controls [H-1] + phantom (Q1)
    std::vector<double> zmp_x_, zmp_y_;                // fb only [H-2]
    void init(int model, int H) {
        x_.assign(H, 0.0); y_.assign(H, 0.0); yaw_.assign(H, 0.0);
        if (model == FULLBODY) { roll_.assign(H, 0.0); pitch_.assign(H, 0.0); }
        for (int d = 0; d < udim_of(model); ++d) u_[d].assign(H, 0.0);  // H-1 real + 1 phantom
        if (model == FULLBODY) { zmp_x_.assign(std::max(H - 2, 1), 0.0); zmp_y_.assign(std::max(H - 2, 1), 0.0); }
    }
};

// Minimal 3-vector for the Eigen operations fb uses (fb:475-483, fb:597-603): cross, dot, +, -, scalar*, /scalar, all
// coefficient-wise in the obvious order.  dot is the one reduction: Eigen (3.3.x, the system package of the reference's
// Ubuntu 20.04 / Noetic; version unpinned, not vendored) sums a Vector3d product as (a0 b0 + a1 b1) + a2 b2 when the
// redux is vectorised (x86-64: one SSE2 packet of two doubles, then the odd element) and as a0 b0 + (a1 b1 + a2 b2) in
// a non-vectorised build.  The reference only ever dots with z = (0, 0, 1) (fb:590 `sumF.dot(n)`, fb:601), where two of
// the three products are exact zeros and both orders give the same bits; the first order is written here.
struct V3 {
    double x, y, z;
    V3 cross(const V3& b) const { return {y * b.z - z * b.y, z * b.x - x * b.z, x * b.y - y * b.x}; }
    double dot(const V3& b) const { return (x * b.x + y * b.y) + z * b.z; }
};
inline V3 operator+(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, const V3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(const V3& a, double s) { return {a.x / s, a.y / s, a.z / s}; }

struct Oracle {
    OrcConfig c;
    int udim;
    std::vector<RobotStates> sample;
    RobotStates optimal;
    std::vector<double> weights, costs;
    std::vector<double> x_ref, y_ref;
    double yaw_ref0 = 0.0;
    double sum_w = 0.0;
    double dt = 0.1;
    double x0[5] = {0, 0, 0, 0, 0};
    // fb constants: fb.h:30, fb.h:212-216, fb:86-91
    double mass = 60.0;
    double base2CoM;
    double Ixx, Iyy, Izz;
    V3 gravity{0.0, 0.0, -9.8};

    explicit Oracle(const OrcConfig& cfg) : c(cfg), udim(udim_of(cfg.model)) {
        // dd:36-46, sd:38-48, fb:72-84
        sample.resize(c.K);
        for (auto& s : sample) s.init(c.model, c.H);
        optimal.init(c.model, c.H);
        weights.assign(c.K, 0.0);
        costs.assign(c.K, 0.0);
        x_ref.assign(c.H, 0.0);
        y_ref.assign(c.H, 0.0);
        if (c.roll_off) { c.zmp_weight = 0.0; c.roll_v_weight = 0.0; }  // fb:43-46
        const double upper_body_height = 0.8075, upper_body_depth = 0.208, upper_body_width = 0.208;
        base2CoM = upper_body_height / 2;  // fb:86
        Ixx = (mass * (upper_body_width * upper_body_width + upper_body_height * upper_body_height)) / 12 + mass * base2CoM * base2CoM;
        Iyy = (mass * (upper_body_height * upper_body_height + upper_body_depth * upper_body_depth)) / 12 + mass * base2CoM * base2CoM;
        Izz = (mass * (upper_body_depth * upper_body_depth + upper_body_width * upper_body_width)) / 12;
    }

    // dd:62-67, sd:78-83, fb:522-526
    static void clamp(double& val, double mn, double mx) {
        if (val < mn) val = mn;
        else if (val > mx) val = mx;
    }

    // dd:81-102, sd:97-118, fb:491-520.  The reference seeds std::mt19937 from
    // std::random_device every call (dd:83-84); here the 32-bit seed is an argument.
    void sampling_mt19937(uint32_t seed) {
        std::mt19937 mt(seed);
        for (int t = 0; t < c.H - 1; ++t) {
            // fresh distribution objects per t (dd:89-90): the polar method's cached
            // second variate never crosses a time step.
            std::normal_distribution<> norm[5] = {
                std::normal_distribution<>(optimal.u_[0][t], c.sigma),
                std::normal_distribution<>(optimal.u_[1][t], c.sigma),
                std::normal_distribution<>(udim > 2 ? optimal.u_[2][t] : 0.0, c.sigma),
                std::normal_distribution<>(udim > 3 ? optimal.u_[3][t] : 0.0, c.sigma),
                std::normal_distribution<>(udim > 4 ? optimal.u_[4][t] : 0.0, c.sigma)};
            for (int i = 0; i < c.K; ++i) {
                for (int d = 0; d < udim; ++d) sample[i].u_[d][t] = norm[d](mt);   // draws first (dd:96-97, fb:506-510)
                for (int d = 0; d < udim; ++d) clamp(sample[i].u_[d][t], c.u_min[d], c.u_max[d]);
                if (c.model == FULLBODY && c.steer_off) sample[i].u_[2][t] = 0.0;  // fb:517
            }
        }
    }

    // Counter-based sampling (the repo's own spec, not the reference's RNG): same
    // clamp / steer_off handling, mean + sigma*z with libstdc++'s "ret*stddev+mean" shape.
    void sampling_philox(uint64_t seed, uint64_t iter, uint32_t k_offset) {
        for (int t = 0; t < c.H - 1; ++t)
            for (int i = 0; i < c.K; ++i) {
                for (int d = 0; d < udim; ++d) {
                    const double z = (double)orc_noise::normal_at(seed, iter, k_offset + (uint32_t)i, (uint32_t)(t * udim + d));
                    double v = z * c.sigma + optimal.u_[d][t];
                    clamp(v, c.u_min[d], c.u_max[d]);
                    sample[i].u_[d][t] = v;
                }
                if (c.model == FULLBODY && c.steer_off) sample[i].u_[2][t] = 0.0;
            }
    }

    // dd:104-109, sd:120-125, fb:445-452
    void predict_next(RobotStates& s, int t) const {
        const double heading = (c.model == DIFF) ? s.yaw_[t] : s.yaw_[t] + s.u_[2][t];
        s.x_[t + 1] = s.x_[t] + s.u_[0][t] * std::cos(heading) * dt;
        s.y_[t + 1] = s.y_[t] + s.u_[0][t] * std::sin(heading) * dt;
        s.yaw_[t + 1] = s.yaw_[t] + s.u_[1][t] * dt;
        if (c.model == FULLBODY) {
            s.roll_[t + 1] = s.roll_[t] + s.u_[3][t] * dt;
            s.pitch_[t + 1] = s.pitch_[t] + s.u_[4][t] * dt;
        }
    }

    // fb:597-603
    V3 zmp_from_model(const V3& CoM, const V3& accel, const V3& HGdot) const {
        const V3 z{0.0, 0.0, 1.0};
        const V3 M_O = CoM.cross(mass * gravity) - CoM.cross(mass * accel) - HGdot;
        return z.cross(M_O) / (mass * (gravity - accel).dot(z));
    }

    // dd:111-124, sd:127-140, fb:454-489 (publish_CandidatePath is ROS plumbing, out of scope)
    void predict_states(const double* x0_in, double dt_in) {
        dt = dt_in;
        std::memcpy(x0, x0_in, sizeof(double) * 5);
        for (int i = 0; i < c.K; ++i) {
            RobotStates& s = sample[i];
            s.x_[0] = x0[0]; s.y_[0] = x0[1]; s.yaw_[0] = x0[2];
            if (c.model == FULLBODY) { s.roll_[0] = x0[3]; s.pitch_[0] = x0[4]; }
            for (int t = 0; t < c.H - 1; ++t) predict_next(s, t);
            if (c.model != FULLBODY) continue;
            for (int t = 0; t < c.H - 2; ++t) {   // fb:468-486
                const double v = s.u_[0][t], w = s.u_[1][t], dir = s.u_[2][t];
                const double drive_accel = (s.u_[0][t + 1] - v) / dt;
                const double ac = v * w;
                const double ax = drive_accel * std::cos(dir) - ac * std::sin(dir);
                const double ay = drive_accel * std::sin(dir) + ac * std::cos(dir);
                const V3 accel{ax, ay, 0.0};
                const V3 next_omega{s.u_[3][t + 1], s.u_[4][t + 1], s.u_[1][t + 1]};
                const V3 omega{s.u_[3][t], s.u_[4][t], w};
                const V3 HG_next{Ixx * next_omega.x, Iyy * next_omega.y, Izz * next_omega.z};  // I_O diagonal (fb:87-91)
                const V3 HG{Ixx * omega.x, Iyy * omega.y, Izz * omega.z};
                const V3 HG_dot = (HG_next - HG) / dt;
                const V3 CoM{base2CoM * std::sin(s.pitch_[t]), -base2CoM * std::sin(s.roll_[t]),
                             base2CoM * std::cos(s.pitch_[t]) * std::cos(s.roll_[t])};
                const V3 zmp = zmp_from_model(CoM, accel, HG_dot);
                s.zmp_x_[t] = zmp.x;
                s.zmp_y_[t] = zmp.y;
            }
        }
    }

    // dd:183-192, sd:199-208, fb:394-403
    double min_distance(double x, double y) const {
        double min_d = 100.0;
        for (int j = 0; j < c.H; ++j) {
            const double d = std::sqrt(std::pow(x - x_ref[j], 2) + std::pow(y - y_ref[j], 2));
            if (d < min_d) min_d = d;
        }
        return min_d;
    }

    // The reference's signatures take their arguments BY VALUE: calc_Cost(RobotStates sample) copies the
    // 5 (dd) / 6 (sd) / 12 (fb) heap vectors of a sample per call (dd.h:130, fb.h:151), and
    // calc_MinDistance(double, double, std::vector<double> x_ref, std::vector<double> y_ref) copies the window twice per
    // (sample, t) (dd.h:140, dd:183; fb:411 calls it twice per t).  Same values, the reference's memory behaviour: this is
    // what bench.py times as the reference-shaped CPU baseline.  (noinline: the copies must really happen.)
    bool by_value = false;
    __attribute__((noinline)) double min_distance_by_value(double x, double y, std::vector<double> xr, std::vector<double> yr) const {
        double min_d = 100.0;
        for (int j = 0; j < c.H; ++j) {
            const double d = std::sqrt(std::pow(x - xr[j], 2) + std::pow(y - yr[j], 2));
            if (d < min_d) min_d = d;
        }
        return min_d;
    }
    __attribute__((noinline)) double calc_cost_by_value(RobotStates s) const {
        double cost = 0.0;
        if (c.model != FULLBODY) {
            for (int t = 0; t < c.H; ++t) {
                const double d = min_distance_by_value(s.x_[t], s.y_[t], x_ref, y_ref);
                const double v_cost = (s.u_[0][t] - c.v_ref) * (s.u_[0][t] - c.v_ref);
                cost += c.path_weight * d * d + c.v_weight * v_cost;
            }
            return cost;
        }
        cost += c.yaw_weight * (s.yaw_[0] - yaw_ref0) * (s.yaw_[0] - yaw_ref0);
        for (int t = 0; t < c.H - 2; ++t) {
            cost += c.path_weight * min_distance_by_value(s.x_[t], s.y_[t], x_ref, y_ref) * min_distance_by_value(s.x_[t], s.y_[t], x_ref, y_ref);
            cost += c.v_weight * (s.u_[0][t] - c.v_ref) * (s.u_[0][t] - c.v_ref);
            cost += c.zmp_weight * s.zmp_y_[t] * s.zmp_y_[t];
            cost += c.roll_v_weight * (s.u_[3][t + 1] - s.u_[3][t]) * (s.u_[3][t + 1] - s.u_[3][t]);
            if (s.u_[0][t] < 0.0) cost += c.back_weight * s.u_[0][t] * s.u_[0][t];
        }
        return cost;
    }

    // dd:194-210, sd:210-226 (t runs to H-1 inclusive: Q1 phantom control) ; fb:404-424
    double calc_cost(const RobotStates& s) const {
        double cost = 0.0;
        if (c.model != FULLBODY) {
            for (int t = 0; t < c.H; ++t) {
                const double d = min_distance(s.x_[t], s.y_[t]);
                const double v_cost = (s.u_[0][t] - c.v_ref) * (s.u_[0][t] - c.v_ref);
                cost += c.path_weight * d * d + c.v_weight * v_cost;
            }
            return cost;
        }
        cost += c.yaw_weight * (s.yaw_[0] - yaw_ref0) * (s.yaw_[0] - yaw_ref0);
        for (int t = 0; t < c.H - 2; ++t) {
            cost += c.path_weight * min_distance(s.x_[t], s.y_[t]) * min_distance(s.x_[t], s.y_[t]);
            cost += c.v_weight * (s.u_[0][t] - c.v_ref) * (s.u_[0][t] - c.v_ref);
            cost += c.zmp_weight * s.zmp_y_[t] * s.zmp_y_[t];
            cost += c.roll_v_weight * (s.u_[3][t + 1] - s.u_[3][t]) * (s.u_[3][t + 1] - s.u_[3][t]);
            if (s.u_[0][t] < 0.0) cost += c.back_weight * s.u_[0][t] * s.u_[0][t];
        }
        return cost;
    }

    // dd:212-223, sd:228-239, fb:426-443 (calc_RefPath is the host prologue: orc_calc_ref_path)
    void calc_weights(const double* xr, const double* yr, double yaw0) {
        std::copy(xr, xr + c.H, x_ref.begin());
        std::copy(yr, yr + c.H, y_ref.begin());
        yaw_ref0 = yaw0;
        double sum = 0.0;
        for (int i = 0; i < c.K; ++i) {
            const double cost = by_value ? calc_cost_by_value(sample[i]) : calc_cost(sample[i]);
            costs[i] = cost;
            weights[i] = std::exp(-cost / c.lambda);
            sum += weights[i];
        }
        sum_w = sum;
        for (int i = 0; i < c.K; ++i) weights[i] /= sum;
    }

    // dd:225-237, sd:241-255, fb:308-326 (t runs to H-1 inclusive: the phantom slot
    // receives sum(w*0) and is never read by sampling)
    void determine_optimal() {
        for (int t = 0; t < c.H; ++t)
            for (int d = 0; d < udim; ++d) {
                double acc = 0.0;
                for (int i = 0; i < c.K; ++i) acc += weights[i] * sample[i].u_[d][t];
                optimal.u_[d][t] = acc;
            }
    }
};


// =============================================================================
// Full-body state estimator: imuCallback fb:199-237, wrenchCallback fb:115-156, get_CurrentState
// fb:528-567, calc_true_ZMP fb:569-596, constants fb:49-63, fb.h:30-32,205-226.  O(1) host math per tick.
// Third-party pieces on this path that are ABSENT from /root/reference and restated from their published source (ros/geometry
// `tf`, Noetic 1.13.x, include/tf/LinearMath/Matrix3x3.h and Quaternion.h -- version unpinned by the reference's package.xml):
//   tf::Matrix3x3(const Quaternion&) -> setRotation():  d = |q|^2, s = 2/d, the nine products in its order
//   tf::Matrix3x3::getRPY() -> getEulerYPR(yaw, pitch, roll, solution 1): pitch = -asin(m[2][0]) unless |m[2][0]| >= 1
//   tf::Matrix3x3 * tf::Vector3: row dot products, each x*x' + y*y' + z*z' left to right
//   tf::getYaw(): plumbing, the pose's yaw arrives as a number here
// No fixture of the reference pins any of it: parity unpinned, like the rest of this file.
// =============================================================================
struct FbEstimator {
    // fb.h:205-216, fb:86-91
    double mass = 60.0, alpha = 0.3, g = -9.81;
    double base2CoM, Ixx, Iyy, Izz;
    V3 gravity{0.0, 0.0, -9.8};
    // imuCallback outputs
    double imu_roll = 0.0, imu_pitch = 0.0, imu_yaw = 0.0;
    double accel_x = 0.0, accel_y = 0.0, accel_z = 0.0;
    V3 ang_vel{0.0, 0.0, 0.0};
    // force sensors, order of force_sensor_topic_ (fb:49-56) = order of contactPositions (fb:63)
    V3 force[6] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    V3 contact[6] = {{0.0, 0.225, 0.075},    {0.0, -0.225, 0.075},   {0.245, 0.167, -0.003},
                     {0.245, -0.167, -0.004}, {-0.245, -0.167, -0.004}, {-0.245, 0.167, -0.003}};   // fb:57-63
    V3 last_HG{0.0, 0.0, 0.0}, true_ZMP{0.0, 0.0, 0.0};
    double state[5] = {0, 0, 0, 0, 0}, zmp_x = 0.0, zmp_y = 0.0;   // current_state_ (fb:74-76)

    FbEstimator() {
        const double upper_body_height = 0.8075, upper_body_depth = 0.208, upper_body_width = 0.208;
        base2CoM = upper_body_height / 2;
        Ixx = (mass * (upper_body_width * upper_body_width + upper_body_height * upper_body_height)) / 12 + mass * base2CoM * base2CoM;
        Iyy = (mass * (upper_body_height * upper_body_height + upper_body_depth * upper_body_depth)) / 12 + mass * base2CoM * base2CoM;
        Izz = (mass * (upper_body_depth * upper_body_depth + upper_body_width * upper_body_width)) / 12;
    }

    static V3 rotate(const double* m, const V3& v) {   // tf::Matrix3x3 (row-major m[9]) * tf::Vector3
        return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z};
    }

    // fb:199-237.  basis: rotation ROBOT_FRAME <- IMU_FRAME (the tf lookup of fb:218), row-major
    void imu(const double* q, const double* w, const double* a, const double* basis) {
        ang_vel = {w[0], w[1], w[2]};
        // tf::Matrix3x3(imu_orientation_).getRPY(imu_roll_, imu_pitch_, imu_yaw_)
        const double x = q[0], y = q[1], z = q[2], ww = q[3];
        const double d = x * x + y * y + z * z + ww * ww;
        const double s = 2.0 / d;
        const double xs = x * s, ys = y * s, zs = z * s;
        const double wx = ww * xs, wy = ww * ys, wz = ww * zs;
        const double xx = x * xs, xy = x * ys, xz = x * zs;
        const double yy = y * ys, yz = y * zs, zz = z * zs;
        const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
        (void)wz;
        if (std::fabs(m20) >= 1.0) {
            imu_yaw = 0.0;
            const double delta = std::atan2(m21, m22);
            if (m20 < 0.0) { imu_pitch = M_PI / 2.0; imu_roll = delta; }
            else { imu_pitch = -M_PI / 2.0; imu_roll = delta; }
        } else {
            imu_pitch = -std::asin(m20);
            imu_roll = std::atan2(m21 / std::cos(imu_pitch), m22 / std::cos(imu_pitch));
            imu_yaw = std::atan2(m10 / std::cos(imu_pitch), m00 / std::cos(imu_pitch));
        }
        const V3 acc = rotate(basis, V3{a[0], a[1], a[2]});   // fb:219-226
        accel_x = acc.x;
        accel_y = acc.y;
        accel_z = acc.z;
        accel_x -= g * std::sin(imu_pitch);   // fb:233
    }

    // fb:115-156: the two wheel sensors are rotated into the robot frame, the four caster sensors are taken as they are
    void wrench(int idx, const double* f, const double* basis) {
        V3 v{f[0], f[1], f[2]};
        if ((idx == 0 || idx == 1) && basis) v = rotate(basis, v);
        force[idx] = v;
    }

    // fb:569-596
    int calc_true_zmp() {
        const V3 n{0.0, 0.0, 1.0};
        V3 sumF{0.0, 0.0, 0.0}, sumM{0.0, 0.0, 0.0};
        for (int i = 0; i < 6; ++i) {
            if (force[i].z > 0.0) {
                sumF = sumF + force[i];
                sumM = sumM + contact[i].cross(force[i]);
            }
        }
        const double denom = sumF.dot(n);
        if (std::fabs(denom) < 1e-6) return 0;   // "denom is too small": true_ZMP keeps its value
        const V3 numerator = n.cross(sumM);
        true_ZMP = alpha * (numerator / denom) + (1 - alpha) * true_ZMP;
        return 1;
    }

    // fb:597-603
    V3 zmp_from_model(const V3& CoM, const V3& accel, const V3& HGdot) const {
        const V3 z{0.0, 0.0, 1.0};
        const V3 M_O = CoM.cross(mass * gravity) - CoM.cross(mass * accel) - HGdot;
        return z.cross(M_O) / (mass * (gravity - accel).dot(z));
    }

    // fb:528-567 (use_gazebo_pose_ / tf branch: the pose arrives as numbers)
    void get_current_state(double px, double py, double yaw, double dt) {
        state[0] = px;
        state[1] = py;
        state[2] = yaw;
        state[3] = imu_roll;
        state[4] = imu_pitch;
        // sin and cos of one angle through ONE libm call, written out: g++ merges std::sin(a) / std::cos(a) into sincos(a) by
        // itself, clang (the host mirror's compiler) does not, and glibc's sincos differs from its sin / cos in the last place
        // about once in 300 arguments -- both sides call it explicitly
        double s_pitch, c_pitch, s_roll, c_roll;
        ::sincos(imu_pitch, &s_pitch, &c_pitch);
        ::sincos(imu_roll, &s_roll, &c_roll);
        const V3 CoM{base2CoM * s_pitch, -base2CoM * s_roll, base2CoM * c_pitch * c_roll};
        const V3 accel{accel_x, accel_y, 0.0};
        const V3 H_G{Ixx * ang_vel.x, Iyy * ang_vel.y, Izz * ang_vel.z};
        const V3 H_Gdot = (H_G - last_HG) / dt;
        last_HG = H_G;
        const V3 ZMP = zmp_from_model(CoM, accel, H_Gdot);
        zmp_x = alpha * ZMP.x + (1 - alpha) * zmp_x;
        zmp_y = alpha * ZMP.y + (1 - alpha) * zmp_y;
    }
};

}  // namespace

extern "C" {

void* orc_create(const OrcConfig* cfg) { return new Oracle(*cfg); }
void orc_destroy(void* h) { delete static_cast<Oracle*>(h); }
int orc_udim(void* h) { return static_cast<Oracle*>(h)->udim; }

// nominal controls, layout [(H-1)][u_dim]
void orc_set_nominal(void* h, const double* u) {
    Oracle* o = static_cast<Oracle*>(h);
    for (int t = 0; t < o->c.H - 1; ++t)
        for (int d = 0; d < o->udim; ++d) o->optimal.u_[d][t] = u[t * o->udim + d];
}
void orc_get_nominal(void* h, double* u) {
    Oracle* o = static_cast<Oracle*>(h);
    for (int t = 0; t < o->c.H - 1; ++t)
        for (int d = 0; d < o->udim; ++d) u[t * o->udim + d] = o->optimal.u_[d][t];
}
void orc_sampling_mt19937(void* h, uint32_t seed) { static_cast<Oracle*>(h)->sampling_mt19937(seed); }
void orc_sampling_philox(void* h, uint64_t seed, uint64_t iter, uint32_t k_offset) {
    static_cast<Oracle*>(h)->sampling_philox(seed, iter, k_offset);
}
// sample controls, layout [K][(H-1)][u_dim]
void orc_set_controls(void* h, const double* u) {
    Oracle* o = static_cast<Oracle*>(h);
    const int T1 = o->c.H - 1, ud = o->udim;
    for (int i = 0; i < o->c.K; ++i)
        for (int t = 0; t < T1; ++t)
            for (int d = 0; d < ud; ++d) o->sample[i].u_[d][t] = u[((size_t)i * T1 + t) * ud + d];
}
void orc_get_controls(void* h, double* u) {
    Oracle* o = static_cast<Oracle*>(h);
    const int T1 = o->c.H - 1, ud = o->udim;
    for (int i = 0; i < o->c.K; ++i)
        for (int t = 0; t < T1; ++t)
            for (int d = 0; d < ud; ++d) u[((size_t)i * T1 + t) * ud + d] = o->sample[i].u_[d][t];
}
void orc_predict_states(void* h, const double* x0, double dt) { static_cast<Oracle*>(h)->predict_states(x0, dt); }
void orc_calc_weights(void* h, const double* xr, const double* yr, double yaw0) {
    static_cast<Oracle*>(h)->calc_weights(xr, yr, yaw0);
}
void orc_determine_optimal(void* h) { static_cast<Oracle*>(h)->determine_optimal(); }
// 1: calc_Cost / calc_MinDistance take their arguments by value, as the reference's signatures do
void orc_set_by_value(void* h, int on) { static_cast<Oracle*>(h)->by_value = on != 0; }
void orc_get_costs(void* h, double* out) {
    Oracle* o = static_cast<Oracle*>(h);
    std::copy(o->costs.begin(), o->costs.end(), out);
}
void orc_get_weights(void* h, double* out) {
    Oracle* o = static_cast<Oracle*>(h);
    std::copy(o->weights.begin(), o->weights.end(), out);
}
double orc_get_sum_w(void* h) { return static_cast<Oracle*>(h)->sum_w; }
// which: 0 x, 1 y, 2 yaw, 3 roll, 4 pitch -> out[K][H];  5 zmp_x, 6 zmp_y -> out[K][H-2]
void orc_get_states(void* h, int which, double* out) {
    Oracle* o = static_cast<Oracle*>(h);
    for (int i = 0; i < o->c.K; ++i) {
        const RobotStates& s = o->sample[i];
        const std::vector<double>* v = which == 0 ? &s.x_ : which == 1 ? &s.y_ : which == 2 ? &s.yaw_ : which == 3 ? &s.roll_
                                     : which == 4 ? &s.pitch_ : which == 5 ? &s.zmp_x_ : &s.zmp_y_;
        const int n = which < 5 ? o->c.H : o->c.H - 2;
        for (int t = 0; t < n; ++t) out[(size_t)i * n + t] = (*v)[t];
    }
}

// One whole reference iteration in the reference's call order (dd:352-358).
// rng: 0 = mt19937(seed32), 1 = philox(seed, iter, k_offset), 2 = keep injected controls
void orc_iterate(void* h, int rng, uint64_t seed, uint64_t iter, uint32_t k_offset, const double* x0, double dt,
                 const double* xr, const double* yr, double yaw0, double* u_opt_out) {
    Oracle* o = static_cast<Oracle*>(h);
    if (rng == 0) o->sampling_mt19937((uint32_t)seed);
    else if (rng == 1) o->sampling_philox(seed, iter, k_offset);
    o->predict_states(x0, dt);
    o->calc_weights(xr, yr, yaw0);
    o->determine_optimal();
    if (u_opt_out) orc_get_nominal(h, u_opt_out);
}

// Host prologue: dd:126-140 + dd:156-181 (sd:142-156,172-197; fb:335-349,365-392).
// Returns current_index_.  yaw_ref has H entries; entry H-1 is left untouched (Q11).
int orc_calc_ref_path(const double* path_x, const double* path_y, int n_path, double cur_x, double cur_y, double v_ref,
                      double dt, double resolution, int H, double* x_ref, double* y_ref, double* yaw_ref) {
    int index0 = 0;
    double min_distance = 100.0;
    for (int i = 0; i < n_path; ++i) {
        const double distance = std::sqrt(std::pow(cur_x - path_x[i], 2) + std::pow(cur_y - path_y[i], 2));
        if (distance < min_distance) { min_distance = distance; index0 = i; }
    }
    const double step = v_ref * dt / resolution;
    for (int i = 0; i < H; ++i) {
        const int index = index0 + i * step;   // int + double -> truncation (Q12)
        if (index < n_path) { x_ref[i] = path_x[index]; y_ref[i] = path_y[index]; }
        else { x_ref[i] = path_x[n_path - 1]; y_ref[i] = path_y[n_path - 1]; }
    }
    for (int i = 0; i < H - 1; ++i) yaw_ref[i] = std::atan2(y_ref[i + 1] - y_ref[i], x_ref[i + 1] - x_ref[i]);
    return index0;
}

// reference_path_creator.cpp:37-56 ("sin" branch).  Returns number of poses written (<= cap).
int orc_path_cosine(double A1, double A2, double A3, double om1, double om2, double om3, double d1, double d2, double d3,
                    double resolution, double course_length, double init_x, double init_y, double* px, double* py, int cap) {
    int n = 0;
    for (double s = 0.0; s < course_length; s += resolution) {
        if (n >= cap) break;
        double y = A1 * std::cos(2 * M_PI * om1 * s + d1) + A2 * std::cos(2 * M_PI * om2 * s + d2) + A3 * std::cos(2 * M_PI * om3 * s + d3) + init_y;
        y -= A1 + A2 + A3;
        px[n] = init_x + s;
        py[n] = y;
        ++n;
    }
    return n;
}

// dkan_path_creator.cpp:11-35 (corner poses), :37-51 (segment walk), :62-64 (three segments)
int orc_path_dkan(double resolution, double* px, double* py, int cap) {
    const double cx[4] = {0.0, 17.7, 17.7, 0.0};
    const double cy[4] = {0.0, 0.0, 8.0, 8.0};
    int n = 0;
    for (int seg = 0; seg < 3; ++seg) {
        const double dx = cx[seg + 1] - cx[seg], dy = cy[seg + 1] - cy[seg];
        for (double s = 0.0; s < std::sqrt(dx * dx + dy * dy); s += resolution) {
            if (n >= cap) return n;
            px[n] = cx[seg] + s * dx / std::sqrt(dx * dx + dy * dy);
            py[n] = cy[seg] + s * dy / std::sqrt(dx * dx + dy * dy);
            ++n;
        }
    }
    return n;
}

// full-body estimator (fb:115-156,199-237,528-596)
void* orc_fbest_create() { return new FbEstimator(); }
void orc_fbest_destroy(void* h) { delete static_cast<FbEstimator*>(h); }
void orc_fbest_imu(void* h, const double* quat_xyzw, const double* ang_vel, const double* lin_acc, const double* basis9) {
    static_cast<FbEstimator*>(h)->imu(quat_xyzw, ang_vel, lin_acc, basis9);
}
void orc_fbest_wrench(void* h, int sensor, const double* force, const double* basis9) {
    static_cast<FbEstimator*>(h)->wrench(sensor, force, basis9);
}
int orc_fbest_true_zmp(void* h) { return static_cast<FbEstimator*>(h)->calc_true_zmp(); }
void orc_fbest_state(void* h, double x, double y, double yaw, double dt) { static_cast<FbEstimator*>(h)->get_current_state(x, y, yaw, dt); }
// out: state (5), zmp_x, zmp_y, true_ZMP (3), imu rpy (3), accel (3)
void orc_fbest_read(void* h, double* out16) {
    const FbEstimator* e = static_cast<FbEstimator*>(h);
    for (int i = 0; i < 5; ++i) out16[i] = e->state[i];
    out16[5] = e->zmp_x; out16[6] = e->zmp_y;
    out16[7] = e->true_ZMP.x; out16[8] = e->true_ZMP.y; out16[9] = e->true_ZMP.z;
    out16[10] = e->imu_roll; out16[11] = e->imu_pitch; out16[12] = e->imu_yaw;
    out16[13] = e->accel_x; out16[14] = e->accel_y; out16[15] = e->accel_z;
}

// noise-spec probes for tests
void orc_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    orc_noise::U4 c{{ctr[0], ctr[1], ctr[2], ctr[3]}};
    orc_noise::U4 o = orc_noise::philox4x32_10(c, key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = o.w[i];
}
void orc_normal_pair(uint32_t a, uint32_t b, float* z) { orc_noise::normal_pair(a, b, &z[0], &z[1]); }
void orc_normals(uint64_t seed, uint64_t iter, uint32_t k0, uint32_t nk, uint32_t n_per, float* out) {
    for (uint32_t k = 0; k < nk; ++k)
        for (uint32_t n = 0; n < n_per; ++n) out[(size_t)k * n_per + n] = orc_noise::normal_at(seed, iter, k0 + k, n);
}

}  // extern "C"
