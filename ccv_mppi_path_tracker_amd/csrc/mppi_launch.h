// Launchers of the rollout kernel families.  Every family is a translation unit of its own (k_r4.hip, k_r3.hip, k_pc.hip,
// k_r4_fb.hip, k_pc_fb.hip, k_solo.hip, k_solo_fb.hip, k_plain.hip): hipcc spends over a minute on all instantiations in one file, the
// units compile side by side (build.py).  ccv_mppi_capi.hip -- the C ABI -- selects the family and calls these.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "mppi_kernels.h"

namespace ccv {

// where a rollout kernel is launched: the stream and, for a timed launch, the events attached to the dispatch itself
// (kernel begin / end timestamps: hipExtLaunchKernelGGL); null events = a plain launch
struct LaunchAt {
    hipStream_t stream;
    hipEvent_t ev_start, ev_stop;
};

// mode: MODE_FUSED / MODE_ROLLOUT / MODE_COST (mppi_rollout_pc.h).  K, H, ... come from the arguments themselves.
void launch_rollout_r4(int model, int mode, bool wide, const LaunchAt& at, const RolloutArgs& A, const Window& W);   // all models
void launch_rollout_r3(int model, int mode, const LaunchAt& at, const RolloutArgs& A, const Window& W);              // dd, sd
void launch_rollout_pc(int model, int mode, const LaunchAt& at, const RolloutArgs& A, const Window& W);              // all models
void launch_rollout_solo(int model, bool wide, const LaunchAt& at, const RolloutArgs& A, const Window& W);           // fused only
// the plain one-sample-per-lane kernel: philox = device noise (fused iteration) or controls read from the buffer
void launch_rollout_plain(int model, bool philox, bool lds_window, const LaunchAt& at, const RolloutArgs& A, const Window& W);
void launch_sample(int model, hipStream_t stream, const RolloutArgs& A);

template <class KERNEL>
inline void launch_at(KERNEL kernel, const dim3 grid, const dim3 block, const LaunchAt& at, const RolloutArgs& A, const Window& W) {
    if (at.ev_start) hipExtLaunchKernelGGL(kernel, grid, block, 0, at.stream, at.ev_start, at.ev_stop, 0, A, W);
    else hipLaunchKernelGGL(kernel, grid, block, 0, at.stream, A, W);
}
inline dim3 blocks_of_64(const RolloutArgs& A) { return dim3((unsigned)((A.K + 63) / 64)); }

}  // namespace ccv
