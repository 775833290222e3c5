// translation unit: the dense two-wave rollout kernel (mppi_rollout_d2.h), full body -- BASELINE config C4's kernel
#include "mppi_launch.h"
#include "mppi_rollout_d2.h"

namespace ccv {

void launch_rollout_d2_fb(const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    launch_at(k_rollout_d2<CCV_MPPI_FULL_BODY, MODE_FUSED>, blocks_of_64(A), dim3(2 * kPcSamples), at, A, W);
}

}  // namespace ccv
