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
    current_index_ = ccv_mppi_calc_ref_path(path_.x.data(), path_.y.data(), (int32_t)path_.size(), current_state_.x,
                                            current_state_.y, v_ref_, dt_, resolution_, horizon_, x_ref_.data(), y_ref_.data(),
                                            yaw_ref_.data());
}

void MPPIBase::sampling() { last_status_ = ccv_mppi_sample(handle_, seed_, iteration_); }

void MPPIBase::predict_States() {
    const double x0[5] = {current_state_.x, current_state_.y, current_state_.yaw, current_state_.roll, current_state_.pitch};
    last_status_ = ccv_mppi_rollout(handle_, x0, dt_);
}

void MPPIBase::calc_Weights() {
    calc_RefPath();
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
        calc_RefPath();
        const double x0[5] = {current_state_.x, current_state_.y, current_state_.yaw, current_state_.roll, current_state_.pitch};
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

int ccv_mppi_node_get_optimal_path(ccv_mppi_node_t* node, double* out) {
    if (!node || !out) return CCV_MPPI_ERR_INVALID_ARG;
    const std::vector<double> p = node->impl->optimal_path();
    std::memcpy(out, p.data(), p.size() * sizeof(double));
    return CCV_MPPI_OK;
}

}  // extern "C"
