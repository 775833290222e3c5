// translation unit: the two-wave rollout kernel (mppi_rollout_pc.h), diff drive and steering (experiments, CCV_MPPI_KERNEL=pc);
// full body: k_pc_fb.hip
#include "mppi_launch.h"
#include "mppi_rollout_pc.h"

namespace ccv {

template <int MODEL>
static void launch_pc_model(int mode, const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    const dim3 grid = blocks_of_64(A), block(kPcWaves * 64);
    if (mode == MODE_FUSED) launch_at(k_rollout_pc<MODEL, MODE_FUSED>, grid, block, at, A, W);
    else if (mode == MODE_ROLLOUT) launch_at(k_rollout_pc<MODEL, MODE_ROLLOUT>, grid, block, at, A, W);
    else launch_at(k_rollout_pc<MODEL, MODE_COST>, grid, block, at, A, W);
}
void launch_rollout_pc_fb(int mode, const LaunchAt& at, const RolloutArgs& A, const Window& W);   // k_pc_fb.hip

void launch_rollout_pc(int model, int mode, const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    if (model == CCV_MPPI_DIFF_DRIVE) launch_pc_model<CCV_MPPI_DIFF_DRIVE>(mode, at, A, W);
    else if (model == CCV_MPPI_STEERING_DIFF_DRIVE) launch_pc_model<CCV_MPPI_STEERING_DIFF_DRIVE>(mode, at, A, W);
    else launch_rollout_pc_fb(mode, at, A, W);
}

}  // namespace ccv
