// translation unit: the four-wave rollout kernel (mppi_rollout_r4.h), diff drive and steering (full body: k_r4_fb.hip)
#include "mppi_launch.h"
#include "mppi_rollout_r4.h"

namespace ccv {

template <int MODEL>
static void launch_r4_model(int mode, bool wide, const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    const dim3 grid = blocks_of_64(A), block(kR4Waves * 64);
    // the horizon's tail: kPartialMin .. 7 control steps in the last block -> the instantiation with the masked batch producer
    const bool tail = (A.H - 1) % kTU >= kPartialMin;
    if constexpr (MODEL == CCV_MPPI_DIFF_DRIVE) {
        if (mode == MODE_FUSED && wide) {
            if (tail) launch_at(k_rollout_r4<MODEL, MODE_FUSED, true, true>, grid, block, at, A, W);
            else launch_at(k_rollout_r4<MODEL, MODE_FUSED, true, false>, grid, block, at, A, W);
            return;
        }
    }
    if (mode == MODE_FUSED && tail) launch_at(k_rollout_r4<MODEL, MODE_FUSED, false, true>, grid, block, at, A, W);
    else if (mode == MODE_FUSED) launch_at(k_rollout_r4<MODEL, MODE_FUSED>, grid, block, at, A, W);
    else if (mode == MODE_ROLLOUT) launch_at(k_rollout_r4<MODEL, MODE_ROLLOUT>, grid, block, at, A, W);
    else launch_at(k_rollout_r4<MODEL, MODE_COST>, grid, block, at, A, W);
}

void launch_rollout_r4_fb(int mode, const LaunchAt& at, const RolloutArgs& A, const Window& W);   // k_r4_fb.hip

void launch_rollout_r4(int model, int mode, bool wide, const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    if (model == CCV_MPPI_DIFF_DRIVE) launch_r4_model<CCV_MPPI_DIFF_DRIVE>(mode, wide, at, A, W);
    else if (model == CCV_MPPI_STEERING_DIFF_DRIVE) launch_r4_model<CCV_MPPI_STEERING_DIFF_DRIVE>(mode, false, at, A, W);
    else launch_rollout_r4_fb(mode, at, A, W);
}

}  // namespace ccv
