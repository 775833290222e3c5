// ccv_mppi_node.hpp -- ROS-free host mirror of the three reference controller classes.
//
// The reference's "API" is the class/topic surface of its ROS nodes.  ROS is not installed in this image,
// so these classes keep the reference's class names, method names, member names and parameter names/defaults, replace
// the ROS message types by plain structs, and implement the four hot methods
//     sampling(), predict_States(), calc_Weights(), determine_OptimalSolution()
// by calls into the C ABI of include/ccv_mppi.h.  A maintainer of the ROS package applies the same four bodies to the
// real node.  Reference locations:
//     DiffDriveMPPI          include/ccv_mppi_path_tracker/diff_drive_mppi.h:52-155, src/diff_drive_mppi.cpp
//     SteeringDiffDriveMPPI  include/ccv_mppi_path_tracker/steering_diff_drive_mppi.h:56-163, src/steering_diff_drive_mppi.cpp
//     FullBodyMPPI           include/ccv_mppi_path_tracker/full_body_mppi.h:68-231, src/full_body_mppi.cpp
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "ccv_mppi.h"

namespace ccv_mppi_node {

// stand-ins for the message types the hot path touches (only the fields it reads or writes)
struct Path {                       // nav_msgs::Path: poses[i].pose.position.{x,y}
    std::vector<double> x, y;
    size_t size() const { return x.size(); }
};
struct Twist {                      // geometry_msgs::Twist: linear.x, angular.z
    double linear_x = 0.0, angular_z = 0.0;
};
struct CmdPoseByRadian {            // ccv_dynamixel_msgs::CmdPoseByRadian
    double steer_l = 0.0, steer_r = 0.0, fore = 0.0, rear = 0.0, roll = 0.0;
};
struct RobotState {                 // current_pose_ (dd/sd) or current_state_ (fb): x, y, yaw[, roll, pitch]
    double x = 0.0, y = 0.0, yaw = 0.0, roll = 0.0, pitch = 0.0;
};
struct Imu {                        // sensor_msgs::Imu: orientation (x, y, z, w), angular_velocity, linear_acceleration
    double orientation[4] = {0.0, 0.0, 0.0, 1.0};
    double angular_velocity[3] = {0.0, 0.0, 0.0};
    double linear_acceleration[3] = {0.0, 0.0, 0.0};
};
struct Rotation {                   // tf::Matrix3x3 of a looked-up transform (row-major); identity by default
    double m[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
};

// nh_.param(name, var, default): parameters come from a string->double map (what the ROS parameter server would hold)
using ParamMap = std::map<std::string, double>;

// Common machinery of the three controllers: everything the reference duplicates verbatim in its three classes.
class MPPIBase {
public:
    virtual ~MPPIBase();
    MPPIBase(const MPPIBase&) = delete;
    MPPIBase& operator=(const MPPIBase&) = delete;

    // ---- subscriber side ----
    void pathCallback(const Path& msg);            // dd:48-52
    void set_CurrentState(const RobotState& s);    // what get_Transform() (dd:314-329) / get_CurrentState() (fb:528-567) produce
    void set_Seed(uint64_t seed) { seed_ = seed; }

    // ---- one pass of the body of run() (dd:346-361): false while no path has been received ----
    bool run_once(double dt);
    // what run() does between measuring dt_ and sampling(): nothing for dd / sd (get_Transform() is tf plumbing, its
    // result arrives through set_CurrentState); fb: calc_true_ZMP() + get_CurrentState() (fb:623-625)
    virtual void update_state() {}

    // ---- the hot methods, in the reference's call order (dd:352-358) ----
    void sampling();                     // dd:81-102   -> ccv_mppi_sample
    void predict_States();               // dd:111-124  -> ccv_mppi_rollout
    void calc_Weights();                 // dd:212-223  -> calc_RefPath() + ccv_mppi_weights
    void determine_OptimalSolution();    // dd:225-246  -> ccv_mppi_update
    // host prologue
    int get_CurrentIndex();              // dd:126-140
    void calc_RefPath();                 // dd:156-181
    // command post-processing
    void publish_CmdVel();               // dd:248-253
    virtual void publish_CmdPos() = 0;   // dd:255-263 / sd:273-296 / fb:246-275

    // ---- results ("published" values) ----
    Twist cmd_vel_;
    CmdPoseByRadian cmd_pos_;
    std::vector<double> x_ref_, y_ref_, yaw_ref_;       // ref_path_ (dd:142-154)
    std::vector<double> optimal_solution;                // controls [(H-1)][u_dim]  (RobotStates optimal_solution, dd.h:100)
    ccv_mppi_stats last_stats_{};
    int last_status_ = CCV_MPPI_OK;
    std::vector<double> candidate_path(int count, int stride);   // publish_CandidatePath() feed (dd:265-294): [count][H][2]
    std::vector<double> best_candidate_paths(int count, std::vector<int32_t>* samples = nullptr);   // same, the top-weight ones
    std::vector<double> optimal_path();                           // publish_OptimalPath() (dd:295-312): [H-1][3] x,y,yaw

    const RobotState& current_state() const { return current_state_; }
    void set_dt(double dt) { dt_ = dt; }
    int horizon() const { return horizon_; }
    int num_samples() const { return (int)num_samples_; }
    int udim() const { return udim_; }
    bool use_fused_ = true;   // run_once(): one fused device iteration (default) or the four stage-wise calls
    // run_once() with use_fused_: get_CurrentIndex() + calc_RefPath() on the device (ccv_mppi_resident_*): the path is
    // uploaded once by pathCallback(), a tick sends the measured pose and receives u* and the window it was planned on
    bool device_prologue_ = false;

protected:
    MPPIBase(int model, const ParamMap& params, int device);
    static double param(const ParamMap& p, const char* name, double dflt);
    void create_handle(const ccv_mppi_config& cfg);

    int model_;
    int udim_;
    ccv_mppi_handle* handle_ = nullptr;
    // parameters (names as in the reference)
    int horizon_ = 15;
    double num_samples_ = 1000.0;
    double control_noise_ = 0.5, exploration_noise_ = 0.5, lambda_ = 1.0;
    double v_max_ = 1.2, w_max_ = 2.0, steer_max_ = 0.0, v_min_ = -1.2, w_min_ = -2.0, steer_min_ = 0.0;
    double v_ref_ = 0.8, path_weight_ = 1.0, v_weight_ = 1.0;
    double dt_ = 0.1, resolution_ = 0.1, pitch_offset_ = 0.0;
    double tread_ = 0.501, wheel_radius_ = 0.1435;
    // state
    Path path_;
    bool path_uploaded_ = false;   // device_prologue_: the current path_ is in HBM
    RobotState current_state_;
    bool path_received_ = false;
    int current_index_ = 0;
    uint64_t seed_ = 42, iteration_ = 0;
};

class DiffDriveMPPI : public MPPIBase {
public:
    explicit DiffDriveMPPI(const ParamMap& params = {}, int device = 0);
    void publish_CmdPos() override;   // dd:255-263
};

class SteeringDiffDriveMPPI : public MPPIBase {
public:
    explicit SteeringDiffDriveMPPI(const ParamMap& params = {}, int device = 0);
    void publish_CmdPos() override;   // sd:273-296
};

// Full-body state estimator: the O(1)-per-tick sensor math of the reference's FullBodyMPPI, with the
// reference's method and member names, usable without a device.  tf / Eigen arithmetic is written out (csrc/host/mppi_node.cpp).
class FullBodyStateEstimator {
public:
    FullBodyStateEstimator();
    // imuCallback() fb:199-237: roll / pitch / yaw of the IMU link from its orientation (tf::Matrix3x3::getRPY), the
    // acceleration rotated into the robot frame by `imu_to_robot` (the tf lookup of fb:218) with the gravity term of fb:233
    void imuCallback(const Imu& msg, const Rotation& imu_to_robot = Rotation());
    // wrenchCallback() fb:115-156: sensor = index into force_sensor_topic_ (fb:49-56: left wheel, right wheel, front-left,
    // front-right, back-left, back-right caster); the two wheel sensors are rotated into the robot frame (fb:121-148)
    void wrenchCallback(int sensor, const double force[3], const Rotation& wheel_to_robot = Rotation());
    void gazeboStatesCallback(double x, double y, double yaw);   // fb:188-196: gazebo_pose_ (x, y, tf::getYaw of the orientation)
    bool calc_true_ZMP();                     // fb:569-596; false = "denom is too small", true_ZMP keeps its value (fb:588-592)
    void get_CurrentState(double dt_);        // fb:528-567: current_state_ from the pose, the IMU angles and the model ZMP (alpha = 0.3 low-pass)
    RobotState current_state_;
    double zmp_x_ = 0.0, zmp_y_ = 0.0;        // current_state_.zmp_x_[0], zmp_y_[0]  (debug topic zmp_y, fb:630-631)
    double true_ZMP[3] = {0.0, 0.0, 0.0};     // (debug topic true_zmp, fb:632-633)
    double imu_roll_ = 0.0, imu_pitch_ = 0.0, imu_yaw_ = 0.0;
    double accel_x = 0.0, accel_y = 0.0, accel_z = 0.0;
    bool imu_received_ = false;
private:
    void computeZMPfromModel(const double CoM[3], const double accel[3], const double HGdot[3], double zmp[3]) const;   // fb:597-603
    double filterd_imu_angular_velocity_[3] = {0.0, 0.0, 0.0};
    double force_sensor_data_[6][3] = {};
    double gazebo_pose_[3] = {0.0, 0.0, 0.0};
    double last_HG[3] = {0.0, 0.0, 0.0};
    double base2CoM = 0.0, I_O[3] = {0.0, 0.0, 0.0};   // fb:86-91 (diagonal inertia)
};

class FullBodyMPPI : public MPPIBase {
public:
    explicit FullBodyMPPI(const ParamMap& params = {}, int device = 0);
    void publish_CmdPos() override;   // fb:246-275

    // ---- state estimator: the callbacks and run()'s prologue of the reference, on plain structs ----
    void imuCallback(const Imu& msg, const Rotation& imu_to_robot = Rotation()) { est_.imuCallback(msg, imu_to_robot); }      // fb:199-237
    void wrenchCallback(int sensor, const double force[3], const Rotation& wheel_to_robot = Rotation()) { est_.wrenchCallback(sensor, force, wheel_to_robot); }   // fb:115-156
    void gazeboStatesCallback(double x, double y, double yaw) { est_.gazeboStatesCallback(x, y, yaw); }                         // fb:188-196
    bool calc_true_ZMP() { return est_.calc_true_ZMP(); }   // fb:569-596
    void get_CurrentState();                                // fb:528-567 -> current_state_
    void update_state() override;                           // fb:623-625, once the first IMU message has arrived
    const FullBodyStateEstimator& estimator() const { return est_; }
private:
    FullBodyStateEstimator est_;
    double roll_max_, roll_min_, pitch_max_, pitch_min_, roll_v_max_, roll_v_min_, pitch_v_max_, pitch_v_min_;
    double zmp_weight_, roll_v_weight_, back_weight_, yaw_weight_;
    bool roll_off_, steer_off_;
};

}  // namespace ccv_mppi_node

// ---- plain C access to the classes above (ctypes / tests) ----------------------------------------------------------
extern "C" {
typedef struct ccv_mppi_node_t ccv_mppi_node_t;
/* model: CCV_MPPI_DIFF_DRIVE ...; params: n (name, value) pairs */
int ccv_mppi_node_create(int model, const char* const* names, const double* values, int n, int device, ccv_mppi_node_t** out);
int ccv_mppi_node_destroy(ccv_mppi_node_t* node);
int ccv_mppi_node_set_path(ccv_mppi_node_t* node, const double* x, const double* y, int n);
int ccv_mppi_node_set_state(ccv_mppi_node_t* node, const double* state5);
int ccv_mppi_node_set_seed(ccv_mppi_node_t* node, uint64_t seed);
int ccv_mppi_node_set_fused(ccv_mppi_node_t* node, int fused);
int ccv_mppi_node_set_device_prologue(ccv_mppi_node_t* node, int on);   /* window built on the device (fused mode only) */
/* one run() pass; returns 1 if a command was produced, 0 while waiting for the path, <0 on error.
 * cmd_out: linear.x, angular.z, steer_l, steer_r, fore, rear, roll */
int ccv_mppi_node_run_once(ccv_mppi_node_t* node, double dt, double* cmd_out7);
int ccv_mppi_node_get_optimal(ccv_mppi_node_t* node, double* u_out);      /* [(H-1)][u_dim] */
int ccv_mppi_node_get_ref_path(ccv_mppi_node_t* node, double* xyyaw_out); /* [H][3] */
int ccv_mppi_node_get_optimal_path(ccv_mppi_node_t* node, double* xyyaw_out); /* [H-1][3] */
/* full-body state estimator of a node (fb:115-156,188-237,528-596); CCV_MPPI_ERR_INVALID_ARG for a node of another model.
 * quat: (x, y, z, w); basis9: row-major rotation into the robot frame or NULL for identity; sensor: 0..5 as fb:49-56 */
int ccv_mppi_node_fb_imu(ccv_mppi_node_t* node, const double* quat_xyzw, const double* angular_velocity3,
                         const double* linear_acceleration3, const double* basis9);
int ccv_mppi_node_fb_wrench(ccv_mppi_node_t* node, int sensor, const double* force3, const double* basis9);
int ccv_mppi_node_fb_pose(ccv_mppi_node_t* node, double x, double y, double yaw);
/* calc_true_ZMP() + get_CurrentState() for a loop period dt (what run_once does itself once IMU data has arrived);
 * returns 1, or 0 when the true-ZMP denominator was too small (fb:588-592) */
int ccv_mppi_node_fb_update_state(ccv_mppi_node_t* node, double dt);
/* out16: current_state_ (x, y, yaw, roll, pitch), zmp_x, zmp_y, true_ZMP (3), imu roll / pitch / yaw, accel x / y / z */
int ccv_mppi_node_fb_read(ccv_mppi_node_t* node, double* out16);
/* the same estimator on its own (no device needed): arguments as above */
typedef struct ccv_mppi_fb_estimator_t ccv_mppi_fb_estimator_t;
int ccv_mppi_fb_estimator_create(ccv_mppi_fb_estimator_t** out);
int ccv_mppi_fb_estimator_destroy(ccv_mppi_fb_estimator_t* e);
int ccv_mppi_fb_estimator_imu(ccv_mppi_fb_estimator_t* e, const double* quat_xyzw, const double* angular_velocity3,
                              const double* linear_acceleration3, const double* basis9);
int ccv_mppi_fb_estimator_wrench(ccv_mppi_fb_estimator_t* e, int sensor, const double* force3, const double* basis9);
int ccv_mppi_fb_estimator_update(ccv_mppi_fb_estimator_t* e, double x, double y, double yaw, double dt);   /* 1, or 0: denom too small */
int ccv_mppi_fb_estimator_read(ccv_mppi_fb_estimator_t* e, double* out16);
}
