// translation unit: the three-wave rollout kernel (mppi_rollout_r3.h), diff drive and steering -- the four-wave kernel's
// cross-check (CCV_MPPI_KERNEL=r3)
#include "mppi_launch.h"
#include "mppi_rollout_r3.h"

namespace ccv {

template <int MODEL>
static void launch_r3_model(int mode, const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    const dim3 grid = blocks_of_64(A), block(kR3Waves * 64);
    if (mode == MODE_FUSED) launch_at(k_rollout_r3<MODEL, MODE_FUSED>, grid, block, at, A, W);
    else if (mode == MODE_ROLLOUT) launch_at(k_rollout_r3<MODEL, MODE_ROLLOUT>, grid, block, at, A, W);
    else launch_at(k_rollout_r3<MODEL, MODE_COST>, grid, block, at, A, W);
}

void launch_rollout_r3(int model, int mode, const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    if (model == CCV_MPPI_DIFF_DRIVE) launch_r3_model<CCV_MPPI_DIFF_DRIVE>(mode, at, A, W);
    else launch_r3_model<CCV_MPPI_STEERING_DIFF_DRIVE>(mode, at, A, W);
}

}  // namespace ccv
