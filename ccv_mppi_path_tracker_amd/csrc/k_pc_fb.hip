// translation unit: the two-wave rollout kernel (mppi_rollout_pc.h), full body -- its production kernel up to four blocks
// of 64 samples per CU, and its stage-wise modes
#include "mppi_launch.h"
#include "mppi_rollout_pc.h"

namespace ccv {

void launch_rollout_pc_fb(int mode, const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    constexpr int MODEL = CCV_MPPI_FULL_BODY;
    const dim3 grid = blocks_of_64(A), block(kPcWaves * 64);
    if (mode == MODE_FUSED) launch_at(k_rollout_pc<MODEL, MODE_FUSED>, grid, block, at, A, W);
    else if (mode == MODE_ROLLOUT) launch_at(k_rollout_pc<MODEL, MODE_ROLLOUT>, grid, block, at, A, W);
    else launch_at(k_rollout_pc<MODEL, MODE_COST>, grid, block, at, A, W);
}

}  // namespace ccv
