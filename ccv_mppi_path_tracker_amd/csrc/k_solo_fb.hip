// translation unit: the one-wave rollout kernel (mppi_rollout_solo.h), full body -- the kernel of BASELINE config C4
#include "mppi_launch.h"
#include "mppi_rollout_solo.h"

namespace ccv {

void launch_rollout_solo_fb(const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    launch_at(k_rollout_solo<CCV_MPPI_FULL_BODY, MODE_FUSED>, blocks_of_64(A), dim3(kPcSamples), at, A, W);
}

}  // namespace ccv
