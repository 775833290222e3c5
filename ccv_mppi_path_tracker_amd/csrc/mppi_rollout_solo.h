// One wave per 64 samples, everything in that wave: k_rollout_solo.
//
// The two- and three-wave kernels (mppi_rollout_pc.h, mppi_rollout_r3.h) split the work of 64 samples over several waves
// because at K = 65 536 there is only one block of 64 samples per SIMD, and a lone wave leaves the SIMD idle in every
// dependent chain.  Their price is the hand-off: a barrier per block of 8 time steps, and the waves of a workgroup wait for
// each other whenever a neighbour on their SIMD delays one of them (full body, K = 131 072: 29 % of the wave cycles are
// barrier waits).  Once K provides two or more blocks of 64 samples per SIMD the SIMD is kept busy by independent waves
// anyway, and the hand-off is pure loss: this kernel runs the same building blocks -- the batched branch-free producer
// (pc_produce_batched), the software-pipelined distance loop (pc_consume), the fused update epilogue -- in ONE wave per 64
// samples, producer and distance phase alternating in program order, the state in registers, no barrier in the time loop.
// The full-body model is the one that profits (its 230 VGPRs allow two waves per SIMD whatever the kernel shape).
#pragma once
#include "mppi_rollout_pc.h"

namespace ccv {

template <int MODEL>
struct SoloShared {
    static constexpr bool kStage = false;
    static constexpr int kPBuf = 1;                        // produced and consumed by the same wave, one after the other
    double2 ab[kMaxH + 4];                                 // window coefficients, padded to a multiple of 4 points
    double c[kMaxH + 4];
    double p[1][kTU][2][kPcSamples];                       // (x,y) - pose of the 8 states of a block; epilogue: transpose buffer
    alignas(32) double nom[(kMaxH + 8) * udim_of(MODEL)];  // warm start u*
};

template <int MODEL, int MODE, bool WIDE = false>   // WIDE: see pc_produce_batched (diff drive beyond |w|max dt = pi/4)
__global__ __launch_bounds__(kPcSamples, 2) void k_rollout_solo(const RolloutArgs Ak, const Window Wk) {
    static_assert(MODE == MODE_FUSED, "the stage-wise modes use k_rollout_pc");
    constexpr bool FB = MODEL == CCV_MPPI_FULL_BODY;
    __shared__ SoloShared<MODEL> sh;
    static_assert(sizeof(sh.p) >= kUpdRB * (kPcSamples + 2) * sizeof(double), "epilogue buffer");
    const RolloutArgs A = with_resident_pose(Ak);
    const int H = A.H;
    const int lane = threadIdx.x;
    stage_window(A, Wk, sh, kPcSamples);
    // two-instruction clamps (clampd_fast, mppi_kernels.h): the host has checked sigma and the bounds, this wave the warm start;
    // a NaN anywhere takes every block through the compare-and-select instantiation
    const bool fast_clamp = A.fast_clamp && __builtin_amdgcn_ballot_w64(pc_stage_nominal<MODEL>(A, sh, kPcSamples)) == 0ull;
    const int k = blockIdx.x * kPcSamples + lane;
    const bool live = k < A.K;
    const int kk = live ? k : A.K - 1;
    const uint32_t kg = (uint32_t)(A.k_offset + kk);
    double cost = 0.0;
    if constexpr (FB) cost += A.w_yaw * (A.x0[2] - A.yaw_ref0) * (A.x0[2] - A.yaw_ref0);   // fb:408
    const int nblocks = (H + kTU - 1) / kTU;
    const int nstates = FB ? H - 2 : H;   // states that reach the path cost (dd:199 / fb:409)
    int prune_on = 1;
    PcState<MODEL> S;
    S.x = A.x0[0];
    S.y = A.x0[1];
    S.yaw = A.x0[2];
    S.roll = A.x0[3];
    S.pitch = A.x0[4];
    S.p_v = S.p_rv = S.p_sdir = S.p_c2 = S.p_c3 = S.p_ac = 0.0;
    S.p_cdir = 1.0;
    fast_sincos(A.x0[2], S.sn, S.cs);
    __syncthreads();   // (one wave: the staged window and warm start are visible to all its lanes)
    // Wave priorities: two (or more) of these waves share a SIMD and run the same code in step; left alone they contend for
    // the same pipe at the same time.  Level (-rank - b) mod 4 -- the workgroup's dispatch rank on its CU, rotating with the
    // time block -- lets one run ahead for a block, then the other: C4 203.3 -> 191.1 us on one box (gpurun_out/r3by; the
    // other rotations tried there: (rank - b) 192.2, (rank - b / 2) 191.8, (rank + b) 194.0, two levels instead of four 195,
    // a constant level per wave: no gain, levels by phase (noise + dynamics high / distance low): +7 us).  CCV_MPPI_PRIO=0: off.
    const int prio_rank = A.prio_rotate ? (int)blockIdx.x / A.cu_count : 0;
    for (int b = 0; b < nblocks; ++b) {
        if (A.prio_rotate) pc_set_priority((-prio_rank - b) & 3);
        // ---------------- states and controls of steps 8b .. 8b+7
        bool done = false;
        const int nctl = min(kTU, H - 1 - b * kTU);   // steps of this block that carry controls
        if (nctl == kTU) {
            if (fast_clamp)
                done = pc_produce_batched<MODEL, MODE, SoloShared<MODEL>, false, true, WIDE>(A, sh, S, cost, b, lane, k, kk, live, kg);
            else if constexpr (!FB)   // (a second instantiation, so that a NaN in the warm start gives the multi-wave kernels'
                                      //  bits; full body has no registers for it -- its NaN case takes pc_produce below, whose
                                      //  sin / cos differ from the block path's in the last place)
                done = pc_produce_batched<MODEL, MODE, SoloShared<MODEL>, false, false, WIDE>(A, sh, S, cost, b, lane, k, kk, live, kg);
        } else if (nctl >= kPartialMin) {   // the horizon's last block, partly filled (C4: 7 of its 8 steps)
            if (fast_clamp)
                done = pc_produce_batched<MODEL, MODE, SoloShared<MODEL>, false, true, WIDE, true>(A, sh, S, cost, b, lane, k, kk, live, kg, nullptr, nctl);
            else if constexpr (!FB)
                done = pc_produce_batched<MODEL, MODE, SoloShared<MODEL>, false, false, WIDE, true>(A, sh, S, cost, b, lane, k, kk, live, kg, nullptr, nctl);
        }
        if (!done) pc_produce<MODEL, MODE, false>(A, sh, S, cost, b, lane, k, kk, live, kg);
        // ---------------- their distance to the window
        const int nv = min(kTU, nstates - b * kTU);
        switch (nv) {
            case 8: pc_consume<8, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
            case 7: pc_consume<7, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
            case 6: pc_consume<6, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
            case 5: pc_consume<5, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
            case 4: pc_consume<4, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
            case 3: pc_consume<3, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
            case 2: pc_consume<2, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
            case 1: pc_consume<1, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
            default: break;
        }
    }
    if (A.prio_rotate) __builtin_amdgcn_s_setprio(0);
    // ---------------- weights and this wave's share of the update (dd:216-237)
    using Rows = UpdRowsT<kTU * udim_of(MODEL), 1>;
    const int R = (H - 1) * udim_of(MODEL);
    UpdT<MODE> upd[kUpdCH];
    const Rows rows{R, 0};
    const int mcount = A.fuse_update ? rows.count() : 0;
    if (mcount > 0) pc_update_fetch(A, upd, rows, 0, mcount, kk);   // (in flight during the exp below)
    const double total = cost;
    const double wgt = live ? exp(-total / A.lambda) : 0.0;   // dd:219 (no min-cost shift)
    if (live) {
        A.cost[k] = total;
        A.w[k] = wgt;
    }
    if (A.fuse_update) {
        pc_reduce_rows<kUpdRB, MODEL>(A, sh, &sh.p[0][0][0][0], upd, rows, mcount, wgt, lane, kk, fast_clamp);
        pc_block_stats(A, R, wgt, total, live, lane);
    }
}

}  // namespace ccv
