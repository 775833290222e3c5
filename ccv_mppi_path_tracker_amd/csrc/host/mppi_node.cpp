// Host mirror of the reference controller classes (include/ccv_mppi_node.hpp).  No ROS; the hot methods are calls into
// the C ABI.  Reference line numbers are given next to each member in the header.
#include "../../../include/ccv_mppi_node.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>

#include "../../../include/ccv_mppi_host.h"

namespace ccv_mppi_node {

namespace {
constexpr double kDeg = M_PI / 180.0;
}

double MPPIBase::param(const ParamMap& p, const char* name, double dflt) {
    auto it = p.find(name);
    return it == p.end() ? dflt : it->second;
}

MPPIBase::MPPIBase(int model, const ParamMap& p, int /*device*/) : model_(model), udim_(ccv_mppi_udim(model)) {}

MPPIBase::~MPPIBase() {
    if (handle_) ccv_mppi_destroy(handle_);
}

void MPPIBase::create_handle(const ccv_mppi_config& cfg) {
    last_status_ = ccv_mppi_create(&cfg, &handle_);
    if (last_status_ != CCV_MPPI_OK) {
        handle_ = nullptr;
        throw std::runtime_error("ccv_mppi_create failed (status " + std::to_string(last_status_) +
                                 "): the MI355X path has no CPU fallback");
    }
    // the constructors size the window and the warm start once (dd:36-46)
    x_ref_.assign(horizon_, 0.0);
    y_ref_.assign(horizon_, 0.0);
    yaw_ref_.assign(horizon_, 0.0);
    optimal_solution.assign((size_t)(horizon_ - 1) * udim_, 0.0);
}

void MPPIBase::pathCallback(const Path& msg) {
    path_ = msg;
    path_uploaded_ = false;
    if (!path_received_) path_received_ = true;
}

void MPPIBase::set_CurrentState(const RobotState& s) { current_state_ = s; }

int MPPIBase::get_CurrentIndex() {
    // shares the implementation with calc_RefPath(); kept as a method because the reference exposes it
    std::vector<double> xr(horizon_), yr(horizon_), yw(horizon_);
    return ccv_mppi_calc_ref_path(path_.x.data(), path_.y.data(), (int32_t)path_.size(), current_state_.x, current_state_.y, v_ref_,
                                  dt_, resolution_, horizon_, xr.data(), yr.data(), yw.data());
}

void MPPIBase::calc_RefPath() {
    // (a negative return is a status code -- dt not positive and finite, an unusable stride: the window is NOT filled then;
    //  the index of the last good window stays and the caller must not iterate against the stale window)
    const int rc = ccv_mppi_calc_ref_path(path_.x.data(), path_.y.data(), (int32_t)path_.size(), current_state_.x,
                                          current_state_.y, v_ref_, dt_, resolution_, horizon_, x_ref_.data(), y_ref_.data(),
                                          yaw_ref_.data());
    if (rc < 0) {
        last_status_ = rc;
        return;
    }
    current_index_ = rc;
}

void MPPIBase::sampling() { last_status_ = ccv_mppi_sample(handle_, seed_, iteration_); }

void MPPIBase::predict_States() {
    const double x0[5] = {current_state_.x, current_state_.y, current_state_.yaw, current_state_.roll, current_state_.pitch};
    last_status_ = ccv_mppi_rollout(handle_, x0, dt_);
}

void MPPIBase::calc_Weights() {
    last_status_ = CCV_MPPI_OK;
    calc_RefPath();
    if (last_status_ != CCV_MPPI_OK) return;   // no window: no weights, and run_once() publishes no command
    last_status_ = ccv_mppi_weights(handle_, x_ref_.data(), y_ref_.data(), yaw_ref_[0]);
}

void MPPIBase::determine_OptimalSolution() {
    last_status_ = ccv_mppi_update(handle_, optimal_solution.data(), &last_stats_);
}

void MPPIBase::publish_CmdVel() {
    cmd_vel_.linear_x = optimal_solution[0];    // optimal_solution.v_[0]
    cmd_vel_.angular_z = optimal_solution[1];   // optimal_solution.w_[0]
}

bool MPPIBase::run_once(double dt) {
    if (!path_received_) return false;
    dt_ = dt;   // the reference overwrites dt_ with the measured loop period every pass (dd:346-348)
    update_state();
    if (use_fused_ && device_prologue_) {
        // the window is built where it is used: pose in, one launch sequence, u* (and the window, for get_ref_path) out
        const double x0[5] = {current_state_.x, current_state_.y, current_state_.yaw, current_state_.roll, current_state_.pitch};
        last_status_ = CCV_MPPI_OK;
        if (!path_uploaded_) {
            last_status_ = ccv_mppi_resident_set_path(handle_, path_.x.data(), path_.y.data(), (int32_t)path_.size(), resolution_);
            path_uploaded_ = last_status_ == CCV_MPPI_OK;
        }
        if (last_status_ == CCV_MPPI_OK) last_status_ = ccv_mppi_resident_set_pose(handle_, x0);
        if (last_status_ == CCV_MPPI_OK) last_status_ = ccv_mppi_resident_step_enqueue(handle_, dt_, seed_, iteration_, 0);
        if (last_status_ == CCV_MPPI_OK) last_status_ = ccv_mppi_get_nominal(handle_, optimal_solution.data());
        int32_t idx = 0;
        if (last_status_ == CCV_MPPI_OK)
            last_status_ = ccv_mppi_resident_read(handle_, nullptr, &idx, x_ref_.data(), y_ref_.data(), &yaw_ref_[0], nullptr);
        current_index_ = idx;
        // yaw_ref_[1..] is not needed by the iteration (fb:408 reads element 0 only); publish_RefPath()'s mirror derives
        // it from the window like calc_RefPath() does (dd:176-179)
        for (int i = 1; i + 1 < horizon_; ++i) yaw_ref_[i] = std::atan2(y_ref_[i + 1] - y_ref_[i], x_ref_[i + 1] - x_ref_[i]);
    } else if (use_fused_) {
        last_status_ = CCV_MPPI_OK;
        calc_RefPath();
        const double x0[5] = {current_state_.x, current_state_.y, current_state_.yaw, current_state_.roll, current_state_.pitch};
        if (last_status_ == CCV_MPPI_OK)   // (refused window: the tick is refused like on the device-prologue path)
            last_status_ = ccv_mppi_iterate(handle_, x0, dt_, x_ref_.data(), y_ref_.data(), yaw_ref_[0], seed_, iteration_,
                                            optimal_solution.data(), &last_stats_);
    } else {
        sampling();
        if (last_status_ == CCV_MPPI_OK) predict_States();
        if (last_status_ == CCV_MPPI_OK) calc_Weights();
        if (last_status_ == CCV_MPPI_OK) determine_OptimalSolution();
    }
    ++iteration_;
    if (last_status_ != CCV_MPPI_OK) return false;
    publish_CmdVel();
    publish_CmdPos();
    return true;
}

std::vector<double> MPPIBase::candidate_path(int count, int stride) {
    std::vector<double> out((size_t)count * horizon_ * 2);
    last_status_ = ccv_mppi_read_candidates(handle_, 0, count, stride, out.data());
    return out;
}

std::vector<double> MPPIBase::best_candidate_paths(int count, std::vector<int32_t>* samples) {
    // the `count` highest-weight samples of the last iteration, selected and gathered on the device
    std::vector<double> out((size_t)count * horizon_ * 2);
    std::vector<int32_t> idx((size_t)count);
    last_status_ = ccv_mppi_read_top_candidates(handle_, count, idx.data(), nullptr, out.data());
    if (samples) *samples = idx;
    return out;
}

std::vector<double> MPPIBase::optimal_path() {
    // re-rolls the optimal controls through the plant model (dd:295-312)
    std::vector<double> out((size_t)(horizon_ - 1) * 3);
    double s[5] = {current_state_.x, current_state_.y, current_state_.yaw, current_state_.roll, current_state_.pitch};
    for (int i = 0; i < horizon_ - 1; ++i) {
        out[(size_t)i * 3 + 0] = s[0];
        out[(size_t)i * 3 + 1] = s[1];
        out[(size_t)i * 3 + 2] = s[2];
        double u[5] = {0, 0, 0, 0, 0};
        for (int d = 0; d < udim_; ++d) u[d] = optimal_solution[(size_t)i * udim_ + d];
        ccv_mppi_plant_step(model_, s, u, dt_);
    }
    return out;
}

// ---- diff drive ------------------------------------------------------------------------------------------------
DiffDriveMPPI::DiffDriveMPPI(const ParamMap& p, int device) : MPPIBase(CCV_MPPI_DIFF_DRIVE, p, device) {
    // dd:17-34 (note: the weight of the speed term is read from "control_weight")
    dt_ = param(p, "dt", 0.1);
    horizon_ = (int)param(p, "horizon", 15);
    num_samples_ = param(p, "num_samples", 1000.0);
    control_noise_ = param(p, "control_noise", 0.5);
    lambda_ = param(p, "lambda", 1.0);
    v_max_ = param(p, "v_max", 1.2);
    w_max_ = param(p, "w_max", 2.0);
    v_min_ = param(p, "v_min", -1.2);
    w_min_ = param(p, "w_min", -2.0);
    pitch_offset_ = param(p, "pitch_offset", 3.0 * kDeg);
    v_ref_ = param(p, "v_ref", 0.8);
    resolution_ = param(p, "resolution", 0.1);
    exploration_noise_ = param(p, "exploration_noise", 0.5);
    path_weight_ = param(p, "path_weight", 1.0);
    v_weight_ = param(p, "control_weight", 1.0);
    ccv_mppi_config c{};
    c.abi_version = CCV_MPPI_ABI_VERSION;
    c.model = model_;
    c.num_samples = (int32_t)num_samples_;
    c.horizon = horizon_;
    c.device = device;
    c.control_noise = control_noise_;
    c.lambda = lambda_;
    c.v_ref = v_ref_;
    c.u_min[0] = v_min_; c.u_max[0] = v_max_;
    c.u_min[1] = w_min_; c.u_max[1] = w_max_;
    c.path_weight = path_weight_;
    c.v_weight = v_weight_;
    create_handle(c);
}

void DiffDriveMPPI::publish_CmdPos() {
    cmd_pos_.steer_l = 0.0;
    cmd_pos_.steer_r = 0.0;
    cmd_pos_.fore = pitch_offset_;
    cmd_pos_.rear = pitch_offset_;
    cmd_pos_.roll = 0.0;
}

// ---- steering diff drive -----------------------------------------------------------------------------------------
SteeringDiffDriveMPPI::SteeringDiffDriveMPPI(const ParamMap& p, int device) : MPPIBase(CCV_MPPI_STEERING_DIFF_DRIVE, p, device) {
    // sd:18-36
    dt_ = param(p, "dt", 0.1);
    horizon_ = (int)param(p, "horizon", 15);
    num_samples_ = param(p, "num_samples", 10000.0);
    control_noise_ = param(p, "control_noise", 0.5);
    lambda_ = param(p, "lambda", 1.0);
    v_max_ = param(p, "v_max", 1.2);
    w_max_ = param(p, "w_max", 1.0);
    steer_max_ = param(p, "steer_max", 30.0 * kDeg);
    v_min_ = param(p, "v_min", -1.2);
    w_min_ = param(p, "w_min", -1.0);
    steer_min_ = param(p, "steer_min", -30.0 * kDeg);
    pitch_offset_ = param(p, "pitch_offset", 3.0 * kDeg);
    v_ref_ = param(p, "v_ref", 0.8);
    resolution_ = param(p, "resolution", 0.1);
    exploration_noise_ = param(p, "exploration_noise", 0.1);
    path_weight_ = param(p, "path_weight", 1.0);
    v_weight_ = param(p, "control_weight", 1.0);
    ccv_mppi_config c{};
    c.abi_version = CCV_MPPI_ABI_VERSION;
    c.model = model_;
    c.num_samples = (int32_t)num_samples_;
    c.horizon = horizon_;
    c.device = device;
    c.control_noise = control_noise_;
    c.lambda = lambda_;
    c.v_ref = v_ref_;
    c.u_min[0] = v_min_; c.u_max[0] = v_max_;
    c.u_min[1] = w_min_; c.u_max[1] = w_max_;
    c.u_min[2] = steer_min_; c.u_max[2] = steer_max_;
    c.path_weight = path_weight_;
    c.v_weight = v_weight_;
    create_handle(c);
}

void SteeringDiffDriveMPPI::publish_CmdPos() {
    // inner / outer wheel angles of the steered differential drive (sd:275-291)
    const double v = optimal_solution[0], w = optimal_solution[1], steer = optimal_solution[2];
    const double R = std::fabs(v / w);
    const double steer_in = std::atan2(R * std::sin(steer), R * std::cos(steer) - tread_ / 2.0);
    const double steer_out = std::atan2(R * std::sin(steer), R * std::cos(steer) + tread_ / 2.0);
    if (w > 0.0) {
        cmd_pos_.steer_l = steer_in;
        cmd_pos_.steer_r = steer_out;
    } else {
        cmd_pos_.steer_l = steer_out;
        cmd_pos_.steer_r = steer_in;
    }
    cmd_pos_.fore = pitch_offset_;
    cmd_pos_.rear = pitch_offset_;
    cmd_pos_.roll = 0.0;
}

// ---- full body -----------------------------------------------------------------------------------------------------
FullBodyMPPI::FullBodyMPPI(const ParamMap& p, int device) : MPPIBase(CCV_MPPI_FULL_BODY, p, device) {
    // fb:8-46
    dt_ = param(p, "dt", 0.1);
    horizon_ = (int)param(p, "horizon", 15);
    num_samples_ = param(p, "num_samples", 10000.0);
    control_noise_ = param(p, "control_noise", 0.5);
    lambda_ = param(p, "lambda", 1.0);
    v_max_ = param(p, "v_max", 1.2);
    w_max_ = param(p, "w_max", 1.0);
    steer_max_ = param(p, "steer_max", 30.0 * kDeg);
    roll_max_ = param(p, "roll_max", 30.0 * kDeg);
    pitch_max_ = param(p, "pitch_max", 15.0 * kDeg);
    roll_v_max_ = param(p, "roll_v_max", 30.0 * kDeg);
    pitch_v_max_ = param(p, "pitch_v_max", 15.0 * kDeg);
    v_min_ = param(p, "v_min", -3.0);
    w_min_ = param(p, "w_min", -1.0);
    steer_min_ = param(p, "steer_min", -30.0 * kDeg);
    roll_min_ = param(p, "roll_min", -30.0 * kDeg);
    pitch_min_ = param(p, "pitch_min", -15.0 * kDeg);
    roll_v_min_ = param(p, "roll_v_min", -30.0 * kDeg);
    pitch_v_min_ = param(p, "pitch_v_min", -15.0 * kDeg);
    pitch_offset_ = param(p, "pitch_offset", 0.0);
    v_ref_ = param(p, "v_ref", 1.2);
    resolution_ = param(p, "resolution", 0.1);
    exploration_noise_ = param(p, "exploration_noise", 0.1);
    path_weight_ = param(p, "path_weight", 1.0);
    v_weight_ = param(p, "v_weight", 1.0);
    zmp_weight_ = param(p, "zmp_weight", 1.0);
    roll_v_weight_ = param(p, "roll_v_weight", 1.0);
    back_weight_ = param(p, "back_weight", 1.0);
    yaw_weight_ = param(p, "yaw_weight", 1.0);
    roll_off_ = param(p, "roll_off", 0.0) != 0.0;
    steer_off_ = param(p, "steer_off", 0.0) != 0.0;
    ccv_mppi_config c{};
    c.abi_version = CCV_MPPI_ABI_VERSION;
    c.model = model_;
    c.num_samples = (int32_t)num_samples_;
    c.horizon = horizon_;
    c.device = device;
    c.flags = (roll_off_ ? CCV_MPPI_FLAG_ROLL_OFF : 0) | (steer_off_ ? CCV_MPPI_FLAG_STEER_OFF : 0);
    c.control_noise = control_noise_;
    c.lambda = lambda_;
    c.v_ref = v_ref_;
    c.u_min[0] = v_min_; c.u_max[0] = v_max_;
    c.u_min[1] = w_min_; c.u_max[1] = w_max_;
    c.u_min[2] = steer_min_; c.u_max[2] = steer_max_;
    c.u_min[3] = roll_v_min_; c.u_max[3] = roll_v_max_;
    c.u_min[4] = pitch_v_min_; c.u_max[4] = pitch_v_max_;
    c.path_weight = path_weight_;
    c.v_weight = v_weight_;
    c.zmp_weight = zmp_weight_;
    c.roll_v_weight = roll_v_weight_;
    c.back_weight = back_weight_;
    c.yaw_weight = yaw_weight_;
    create_handle(c);
}

// ---- full-body state estimator ------------------------------------------------------------------------------------
// The arithmetic the reference delegates to tf and Eigen is written out: tf::Matrix3x3(q) (setRotation), getRPY()
// (getEulerYPR, first solution), Matrix3x3 * Vector3 (tf LinearMath, ros/geometry noetic); Eigen cross / dot / +,- / scalar.
namespace {
constexpr double kMass = 60.0;                 // fb.h:216
constexpr double kAlpha = 0.3;                 // fb.h:218: low-pass weight
constexpr double kG = -9.81;                   // fb.h:32 (gravity compensation of the IMU acceleration)
constexpr double kGravityZ = -9.8;             // fb.h:30 gravity_ (the ZMP model)
// contactPositions (fb:57-63), in the order of force_sensor_topic_ (fb:49-56)
constexpr double kContact[6][3] = {{0.0, 0.225, 0.075},    {0.0, -0.225, 0.075},   {0.245, 0.167, -0.003},
                                   {0.245, -0.167, -0.004}, {-0.245, -0.167, -0.004}, {-0.245, 0.167, -0.003}};

inline void cross3(const double a[3], const double b[3], double out[3]) {
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}
inline void rotate3(const double m[9], const double v[3], double out[3]) {
    for (int r = 0; r < 3; ++r) out[r] = m[3 * r] * v[0] + m[3 * r + 1] * v[1] + m[3 * r + 2] * v[2];
}
}  // namespace

FullBodyStateEstimator::FullBodyStateEstimator() {
    // fb.h:212-216, fb:86-91
    const double upper_body_height = 0.8075, upper_body_depth = 0.208, upper_body_width = 0.208, mass = kMass;
    base2CoM = upper_body_height / 2;
    I_O[0] = (mass * (upper_body_width * upper_body_width + upper_body_height * upper_body_height)) / 12 + mass * base2CoM * base2CoM;
    I_O[1] = (mass * (upper_body_height * upper_body_height + upper_body_depth * upper_body_depth)) / 12 + mass * base2CoM * base2CoM;
    I_O[2] = (mass * (upper_body_depth * upper_body_depth + upper_body_width * upper_body_width)) / 12;
}

void FullBodyStateEstimator::imuCallback(const Imu& msg, const Rotation& imu_to_robot) {
    for (int i = 0; i < 3; ++i) filterd_imu_angular_velocity_[i] = msg.angular_velocity[i];   // fb:209-211 (the filter is commented out)
    // tf::Matrix3x3(imu_orientation_).getRPY(imu_roll_, imu_pitch_, imu_yaw_)  (fb:216-217)
    const double qx = msg.orientation[0], qy = msg.orientation[1], qz = msg.orientation[2], qw = msg.orientation[3];
    const double d = qx * qx + qy * qy + qz * qz + qw * qw;
    const double s = 2.0 / d;
    const double xs = qx * s, ys = qy * s, zs = qz * s;
    const double wx = qw * xs, wy = qw * ys, wz = qw * zs;
    const double xx = qx * xs, xy = qx * ys, xz = qx * zs;
    const double yy = qy * ys, yz = qy * zs, zz = qz * zs;
    const double r00 = 1.0 - (yy + zz), r10 = xy + wz, r20 = xz - wy, r21 = yz + wx, r22 = 1.0 - (xx + yy);
    if (std::fabs(r20) >= 1.0) {   // pitch at +-90 degrees: yaw is taken as 0
        imu_yaw_ = 0.0;
        imu_roll_ = std::atan2(r21, r22);
        imu_pitch_ = r20 < 0.0 ? M_PI / 2.0 : -M_PI / 2.0;
    } else {
        imu_pitch_ = -std::asin(r20);
        const double cp = std::cos(imu_pitch_);
        imu_roll_ = std::atan2(r21 / cp, r22 / cp);
        imu_yaw_ = std::atan2(r10 / cp, r00 / cp);
    }
    double acc[3];
    rotate3(imu_to_robot.m, msg.linear_acceleration, acc);   // fb:218-226
    accel_x = acc[0];
    accel_y = acc[1];
    accel_z = acc[2];
    accel_x -= kG * std::sin(imu_pitch_);                    // fb:233
    imu_received_ = true;
}

void FullBodyStateEstimator::wrenchCallback(int sensor, const double force[3], const Rotation& wheel_to_robot) {
    if (sensor < 0 || sensor >= 6) return;
    if (sensor <= 1) rotate3(wheel_to_robot.m, force, force_sensor_data_[sensor]);   // fb:121-148
    else for (int i = 0; i < 3; ++i) force_sensor_data_[sensor][i] = force[i];
}

void FullBodyStateEstimator::gazeboStatesCallback(double x, double y, double yaw) {
    gazebo_pose_[0] = x;
    gazebo_pose_[1] = y;
    gazebo_pose_[2] = yaw;
}

bool FullBodyStateEstimator::calc_true_ZMP() {
    double sumF[3] = {0.0, 0.0, 0.0}, sumM[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < 6; ++i) {
        const double* f = force_sensor_data_[i];
        if (f[2] > 0.0) {   // only sensors in contact (fb:580)
            double m[3];
            cross3(kContact[i], f, m);
            for (int k = 0; k < 3; ++k) {
                sumF[k] += f[k];
                sumM[k] += m[k];
            }
        }
    }
    const double denom = (sumF[0] * 0.0 + sumF[1] * 0.0) + sumF[2] * 1.0;   // sumF.dot(n), n = (0, 0, 1)
    if (std::fabs(denom) < 1e-6) return false;                            // fb:588-592
    const double n[3] = {0.0, 0.0, 1.0};
    double num[3];
    cross3(n, sumM, num);
    for (int k = 0; k < 3; ++k) true_ZMP[k] = kAlpha * (num[k] / denom) + (1 - kAlpha) * true_ZMP[k];   // fb:595
    return true;
}

void FullBodyStateEstimator::computeZMPfromModel(const double CoM[3], const double accel[3], const double HGdot[3], double zmp[3]) const {
    const double z[3] = {0.0, 0.0, 1.0};
    const double mg[3] = {kMass * 0.0, kMass * 0.0, kMass * kGravityZ};
    const double ma[3] = {kMass * accel[0], kMass * accel[1], kMass * accel[2]};
    double c1[3], c2[3], M_O[3], num[3];
    cross3(CoM, mg, c1);
    cross3(CoM, ma, c2);
    for (int k = 0; k < 3; ++k) M_O[k] = c1[k] - c2[k] - HGdot[k];
    const double gd[3] = {0.0 - accel[0], 0.0 - accel[1], kGravityZ - accel[2]};
    const double den = kMass * ((gd[0] * z[0] + gd[1] * z[1]) + gd[2] * z[2]);
    cross3(z, M_O, num);
    for (int k = 0; k < 3; ++k) zmp[k] = num[k] / den;
}

void FullBodyStateEstimator::get_CurrentState(double dt_) {
    current_state_.x = gazebo_pose_[0];  // use_gazebo_pose_ (fb:546-550; the tf branch delivers the same three numbers)
    current_state_.y = gazebo_pose_[1];
    current_state_.yaw = gazebo_pose_[2];
    current_state_.roll = imu_roll_;     // fb:552-553
    current_state_.pitch = imu_pitch_;
    // (sin and cos of one angle through one sincos call, as the oracle's restatement: glibc's sincos and its sin / cos differ in
    //  the last place now and then, and which of them a compiler emits for std::sin(a), std::cos(a) is its own choice)
    double s_pitch, c_pitch, s_roll, c_roll;
    ::sincos(imu_pitch_, &s_pitch, &c_pitch);
    ::sincos(imu_roll_, &s_roll, &c_roll);
    const double CoM[3] = {base2CoM * s_pitch, -base2CoM * s_roll, base2CoM * c_pitch * c_roll};
    const double accel[3] = {accel_x, accel_y, 0.0};
    double H_G[3], H_Gdot[3], zmp[3];
    for (int k = 0; k < 3; ++k) {
        H_G[k] = I_O[k] * filterd_imu_angular_velocity_[k];   // I_O is diagonal (fb:87-91)
        H_Gdot[k] = (H_G[k] - last_HG[k]) / dt_;
        last_HG[k] = H_G[k];
    }
    computeZMPfromModel(CoM, accel, H_Gdot, zmp);
    zmp_x_ = kAlpha * zmp[0] + (1 - kAlpha) * zmp_x_;   // fb:565-566
    zmp_y_ = kAlpha * zmp[1] + (1 - kAlpha) * zmp_y_;
}

void FullBodyMPPI::get_CurrentState() {
    est_.get_CurrentState(dt_);
    current_state_ = est_.current_state_;
}

void FullBodyMPPI::update_state() {
    if (!est_.imu_received_) return;   // (a caller that feeds set_CurrentState() directly keeps doing so)
    est_.calc_true_ZMP();              // fb:623
    get_CurrentState();                // fb:625
}

void FullBodyMPPI::publish_CmdPos() {
    const double v = optimal_solution[0], w = optimal_solution[1], direction = optimal_solution[2], roll_v = optimal_solution[3];
    if (steer_off_) {
        cmd_pos_.steer_l = 0.0;
        cmd_pos_.steer_r = 0.0;
    } else {
        const double R = std::fabs(v / w);
        const double steer_in = std::atan2(R * std::sin(direction), R * std::cos(direction) - tread_ / 2.0);
        const double steer_out = std::atan2(R * std::sin(direction), R * std::cos(direction) + tread_ / 2.0);
        if (w > 0.0) {
            cmd_pos_.steer_l = steer_in;
            cmd_pos_.steer_r = steer_out;
        } else {
            cmd_pos_.steer_l = steer_out;
            cmd_pos_.steer_r = steer_in;
        }
    }
    // roll command: integrate the commanded roll rate one period, then clamp (fb:266-269)
    cmd_pos_.roll = current_state_.roll + roll_v * dt_;
    if (cmd_pos_.roll > roll_max_) cmd_pos_.roll = roll_max_;
    else if (cmd_pos_.roll < roll_min_) cmd_pos_.roll = roll_min_;
    if (roll_off_) cmd_pos_.roll = 0.0;
    cmd_pos_.fore = pitch_offset_;
    cmd_pos_.rear = pitch_offset_;
}

}  // namespace ccv_mppi_node

// ---- C access ---------------------------------------------------------------------------------------------------
struct ccv_mppi_node_t {
    ccv_mppi_node::MPPIBase* impl = nullptr;
};

extern "C" {

int ccv_mppi_node_create(int model, const char* const* names, const double* values, int n, int device, ccv_mppi_node_t** out) {
    if (!out || n < 0 || (n > 0 && (!names || !values))) return CCV_MPPI_ERR_INVALID_ARG;
    *out = nullptr;
    ccv_mppi_node::ParamMap p;
    for (int i = 0; i < n; ++i) p[names[i]] = values[i];
    try {
        ccv_mppi_node::MPPIBase* impl = nullptr;
        if (model == CCV_MPPI_DIFF_DRIVE) impl = new ccv_mppi_node::DiffDriveMPPI(p, device);
        else if (model == CCV_MPPI_STEERING_DIFF_DRIVE) impl = new ccv_mppi_node::SteeringDiffDriveMPPI(p, device);
        else if (model == CCV_MPPI_FULL_BODY) impl = new ccv_mppi_node::FullBodyMPPI(p, device);
        else return CCV_MPPI_ERR_INVALID_ARG;
        *out = new ccv_mppi_node_t{impl};
        return CCV_MPPI_OK;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "ccv_mppi_node_create: %s\n", e.what());
        return CCV_MPPI_ERR_NO_DEVICE;
    }
}

int ccv_mppi_node_destroy(ccv_mppi_node_t* node) {
    if (!node) return CCV_MPPI_ERR_INVALID_ARG;
    delete node->impl;
    delete node;
    return CCV_MPPI_OK;
}

int ccv_mppi_node_set_path(ccv_mppi_node_t* node, const double* x, const double* y, int n) {
    if (!node || !x || !y || n < 1) return CCV_MPPI_ERR_INVALID_ARG;
    ccv_mppi_node::Path p;
    p.x.assign(x, x + n);
    p.y.assign(y, y + n);
    node->impl->pathCallback(p);
    return CCV_MPPI_OK;
}

int ccv_mppi_node_set_state(ccv_mppi_node_t* node, const double* s) {
    if (!node || !s) return CCV_MPPI_ERR_INVALID_ARG;
    ccv_mppi_node::RobotState st;
    st.x = s[0]; st.y = s[1]; st.yaw = s[2]; st.roll = s[3]; st.pitch = s[4];
    node->impl->set_CurrentState(st);
    return CCV_MPPI_OK;
}

int ccv_mppi_node_set_seed(ccv_mppi_node_t* node, uint64_t seed) {
    if (!node) return CCV_MPPI_ERR_INVALID_ARG;
    node->impl->set_Seed(seed);
    return CCV_MPPI_OK;
}

int ccv_mppi_node_set_fused(ccv_mppi_node_t* node, int fused) {
    if (!node) return CCV_MPPI_ERR_INVALID_ARG;
    node->impl->use_fused_ = fused != 0;
    return CCV_MPPI_OK;
}

int ccv_mppi_node_set_device_prologue(ccv_mppi_node_t* node, int on) {
    if (!node) return CCV_MPPI_ERR_INVALID_ARG;
    node->impl->device_prologue_ = on != 0;
    return CCV_MPPI_OK;
}

int ccv_mppi_node_run_once(ccv_mppi_node_t* node, double dt, double* cmd) {
    if (!node || !cmd) return CCV_MPPI_ERR_INVALID_ARG;
    const bool produced = node->impl->run_once(dt);
    if (node->impl->last_status_ != CCV_MPPI_OK) return node->impl->last_status_;
    if (!produced) return 0;
    cmd[0] = node->impl->cmd_vel_.linear_x;
    cmd[1] = node->impl->cmd_vel_.angular_z;
    cmd[2] = node->impl->cmd_pos_.steer_l;
    cmd[3] = node->impl->cmd_pos_.steer_r;
    cmd[4] = node->impl->cmd_pos_.fore;
    cmd[5] = node->impl->cmd_pos_.rear;
    cmd[6] = node->impl->cmd_pos_.roll;
    return 1;
}

int ccv_mppi_node_get_optimal(ccv_mppi_node_t* node, double* u_out) {
    if (!node || !u_out) return CCV_MPPI_ERR_INVALID_ARG;
    std::memcpy(u_out, node->impl->optimal_solution.data(), node->impl->optimal_solution.size() * sizeof(double));
    return CCV_MPPI_OK;
}

int ccv_mppi_node_get_ref_path(ccv_mppi_node_t* node, double* out) {
    if (!node || !out) return CCV_MPPI_ERR_INVALID_ARG;
    for (int i = 0; i < node->impl->horizon(); ++i) {
        out[i * 3 + 0] = node->impl->x_ref_[i];
        out[i * 3 + 1] = node->impl->y_ref_[i];
        out[i * 3 + 2] = node->impl->yaw_ref_[i];
    }
    return CCV_MPPI_OK;
}

namespace {
ccv_mppi_node::FullBodyMPPI* full_body(ccv_mppi_node_t* node) {
    return node ? dynamic_cast<ccv_mppi_node::FullBodyMPPI*>(node->impl) : nullptr;
}
ccv_mppi_node::Rotation rotation_of(const double* basis9) {
    ccv_mppi_node::Rotation r;
    if (basis9) std::memcpy(r.m, basis9, sizeof(r.m));
    return r;
}
void read_estimator(const ccv_mppi_node::FullBodyStateEstimator& est, const ccv_mppi_node::RobotState& st, double* out) {
    out[0] = st.x; out[1] = st.y; out[2] = st.yaw; out[3] = st.roll; out[4] = st.pitch;
    out[5] = est.zmp_x_; out[6] = est.zmp_y_;
    for (int k = 0; k < 3; ++k) out[7 + k] = est.true_ZMP[k];
    out[10] = est.imu_roll_; out[11] = est.imu_pitch_; out[12] = est.imu_yaw_;
    out[13] = est.accel_x; out[14] = est.accel_y; out[15] = est.accel_z;
}
}  // namespace

int ccv_mppi_node_fb_imu(ccv_mppi_node_t* node, const double* q, const double* w, const double* a, const double* basis9) {
    ccv_mppi_node::FullBodyMPPI* fb = full_body(node);
    if (!fb || !q || !w || !a) return CCV_MPPI_ERR_INVALID_ARG;
    ccv_mppi_node::Imu m;
    std::memcpy(m.orientation, q, sizeof(m.orientation));
    std::memcpy(m.angular_velocity, w, sizeof(m.angular_velocity));
    std::memcpy(m.linear_acceleration, a, sizeof(m.linear_acceleration));
    fb->imuCallback(m, rotation_of(basis9));
    return CCV_MPPI_OK;
}

int ccv_mppi_node_fb_wrench(ccv_mppi_node_t* node, int sensor, const double* force3, const double* basis9) {
    ccv_mppi_node::FullBodyMPPI* fb = full_body(node);
    if (!fb || !force3 || sensor < 0 || sensor >= 6) return CCV_MPPI_ERR_INVALID_ARG;
    fb->wrenchCallback(sensor, force3, rotation_of(basis9));
    return CCV_MPPI_OK;
}

int ccv_mppi_node_fb_pose(ccv_mppi_node_t* node, double x, double y, double yaw) {
    ccv_mppi_node::FullBodyMPPI* fb = full_body(node);
    if (!fb) return CCV_MPPI_ERR_INVALID_ARG;
    fb->gazeboStatesCallback(x, y, yaw);
    return CCV_MPPI_OK;
}

int ccv_mppi_node_fb_update_state(ccv_mppi_node_t* node, double dt) {
    ccv_mppi_node::FullBodyMPPI* fb = full_body(node);
    if (!fb || !(dt == dt)) return CCV_MPPI_ERR_INVALID_ARG;
    fb->set_dt(dt);
    const bool ok = fb->calc_true_ZMP();
    fb->get_CurrentState();
    return ok ? 1 : 0;
}

int ccv_mppi_node_fb_read(ccv_mppi_node_t* node, double* out) {
    ccv_mppi_node::FullBodyMPPI* fb = full_body(node);
    if (!fb || !out) return CCV_MPPI_ERR_INVALID_ARG;
    read_estimator(fb->estimator(), fb->current_state(), out);
    return CCV_MPPI_OK;
}

// ---- the estimator on its own (no device) ----
struct ccv_mppi_fb_estimator_t {
    ccv_mppi_node::FullBodyStateEstimator impl;
};

int ccv_mppi_fb_estimator_create(ccv_mppi_fb_estimator_t** out) {
    if (!out) return CCV_MPPI_ERR_INVALID_ARG;
    *out = new (std::nothrow) ccv_mppi_fb_estimator_t();
    return *out ? CCV_MPPI_OK : CCV_MPPI_ERR_ALLOC;
}

int ccv_mppi_fb_estimator_destroy(ccv_mppi_fb_estimator_t* e) {
    if (!e) return CCV_MPPI_ERR_INVALID_ARG;
    delete e;
    return CCV_MPPI_OK;
}

int ccv_mppi_fb_estimator_imu(ccv_mppi_fb_estimator_t* e, const double* q, const double* w, const double* a, const double* basis9) {
    if (!e || !q || !w || !a) return CCV_MPPI_ERR_INVALID_ARG;
    ccv_mppi_node::Imu m;
    std::memcpy(m.orientation, q, sizeof(m.orientation));
    std::memcpy(m.angular_velocity, w, sizeof(m.angular_velocity));
    std::memcpy(m.linear_acceleration, a, sizeof(m.linear_acceleration));
    e->impl.imuCallback(m, rotation_of(basis9));
    return CCV_MPPI_OK;
}

int ccv_mppi_fb_estimator_wrench(ccv_mppi_fb_estimator_t* e, int sensor, const double* force3, const double* basis9) {
    if (!e || !force3 || sensor < 0 || sensor >= 6) return CCV_MPPI_ERR_INVALID_ARG;
    e->impl.wrenchCallback(sensor, force3, rotation_of(basis9));
    return CCV_MPPI_OK;
}

int ccv_mppi_fb_estimator_update(ccv_mppi_fb_estimator_t* e, double x, double y, double yaw, double dt) {
    if (!e || !(dt == dt)) return CCV_MPPI_ERR_INVALID_ARG;
    e->impl.gazeboStatesCallback(x, y, yaw);
    const bool ok = e->impl.calc_true_ZMP();   // fb:623
    e->impl.get_CurrentState(dt);              // fb:625
    return ok ? 1 : 0;
}

int ccv_mppi_fb_estimator_read(ccv_mppi_fb_estimator_t* e, double* out) {
    if (!e || !out) return CCV_MPPI_ERR_INVALID_ARG;
    read_estimator(e->impl, e->impl.current_state_, out);
    return CCV_MPPI_OK;
}

int ccv_mppi_node_get_optimal_path(ccv_mppi_node_t* node, double* out) {
    if (!node || !out) return CCV_MPPI_ERR_INVALID_ARG;
    const std::vector<double> p = node->impl->optimal_path();
    std::memcpy(out, p.data(), p.size() * sizeof(double));
    return CCV_MPPI_OK;
}

}  // extern "C"
