// translation unit: the four-wave rollout kernel (mppi_rollout_r4.h), full body -- its kernel up to two blocks of 64 samples per
// CU (the reference's own operating point, K = 10 000, H = 15), and its stage-wise modes
#include "mppi_launch.h"
#include "mppi_rollout_r4.h"

namespace ccv {

void launch_rollout_r4_fb(int mode, const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    constexpr int MODEL = CCV_MPPI_FULL_BODY;
    const dim3 grid = blocks_of_64(A), block(kR4Waves * 64);
    const bool tail = (A.H - 1) % kTU >= kPartialMin;
    if (mode == MODE_FUSED && tail) launch_at(k_rollout_r4<MODEL, MODE_FUSED, false, true>, grid, block, at, A, W);
    else if (mode == MODE_FUSED) launch_at(k_rollout_r4<MODEL, MODE_FUSED>, grid, block, at, A, W);
    else if (mode == MODE_ROLLOUT) launch_at(k_rollout_r4<MODEL, MODE_ROLLOUT>, grid, block, at, A, W);
    else launch_at(k_rollout_r4<MODEL, MODE_COST>, grid, block, at, A, W);
}

}  // namespace ccv
