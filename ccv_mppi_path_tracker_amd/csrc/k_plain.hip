// translation unit: the plain one-sample-per-lane kernel (k_rollout_cost, OCML sincos: the path of unbounded headings and the
// experiment baseline CCV_MPPI_KERNEL=v1) and the stand-alone sampling() kernel
#include "mppi_launch.h"

namespace ccv {

template <int MODEL>
static void launch_plain_model(bool philox, bool lds_window, const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    const dim3 grid((unsigned)((A.K + kBlock - 1) / kBlock)), block(kBlock);
    if (philox) {
        if (lds_window) launch_at(k_rollout_cost<MODEL, SRC_PHILOX, true>, grid, block, at, A, W);
        else launch_at(k_rollout_cost<MODEL, SRC_PHILOX, false>, grid, block, at, A, W);
    } else {
        if (lds_window) launch_at(k_rollout_cost<MODEL, SRC_BUFFER, true>, grid, block, at, A, W);
        else launch_at(k_rollout_cost<MODEL, SRC_BUFFER, false>, grid, block, at, A, W);
    }
}

void launch_rollout_plain(int model, bool philox, bool lds_window, const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    if (model == CCV_MPPI_DIFF_DRIVE) launch_plain_model<CCV_MPPI_DIFF_DRIVE>(philox, lds_window, at, A, W);
    else if (model == CCV_MPPI_STEERING_DIFF_DRIVE) launch_plain_model<CCV_MPPI_STEERING_DIFF_DRIVE>(philox, lds_window, at, A, W);
    else launch_plain_model<CCV_MPPI_FULL_BODY>(philox, lds_window, at, A, W);
}

void launch_sample(int model, hipStream_t stream, const RolloutArgs& A) {
    const int R = (A.H - 1) * udim_of(model);
    const dim3 grid((unsigned)((A.K + kBlock - 1) / kBlock), (unsigned)((R + 3) / 4)), block(kBlock);
    if (model == CCV_MPPI_DIFF_DRIVE) hipLaunchKernelGGL((k_sample<CCV_MPPI_DIFF_DRIVE>), grid, block, 0, stream, A);
    else if (model == CCV_MPPI_STEERING_DIFF_DRIVE) hipLaunchKernelGGL((k_sample<CCV_MPPI_STEERING_DIFF_DRIVE>), grid, block, 0, stream, A);
    else hipLaunchKernelGGL((k_sample<CCV_MPPI_FULL_BODY>), grid, block, 0, stream, A);
}

}  // namespace ccv
