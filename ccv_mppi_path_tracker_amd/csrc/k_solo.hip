// translation unit: the one-wave rollout kernel (mppi_rollout_solo.h), diff drive and steering; full body: k_solo_fb.hip
#include "mppi_launch.h"
#include "mppi_rollout_solo.h"

namespace ccv {

void launch_rollout_solo_fb(const LaunchAt& at, const RolloutArgs& A, const Window& W);   // k_solo_fb.hip

void launch_rollout_solo(int model, bool wide, const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    const dim3 grid = blocks_of_64(A), block(kPcSamples);
    if (model == CCV_MPPI_DIFF_DRIVE) {
        if (wide) launch_at(k_rollout_solo<CCV_MPPI_DIFF_DRIVE, MODE_FUSED, true>, grid, block, at, A, W);
        else launch_at(k_rollout_solo<CCV_MPPI_DIFF_DRIVE, MODE_FUSED>, grid, block, at, A, W);
    } else if (model == CCV_MPPI_STEERING_DIFF_DRIVE) {
        launch_at(k_rollout_solo<CCV_MPPI_STEERING_DIFF_DRIVE, MODE_FUSED>, grid, block, at, A, W);
    } else {
        launch_rollout_solo_fb(at, A, W);
    }
}

}  // namespace ccv
