// Three-wave rollout kernel for gfx950 (MI355X): one workgroup = 64 samples = producer wave + distance wave + store wave.
//
// Issuing a 512-byte global store costs the issuing wave 45-120 cycles on this chip (tools/microbench/store_issue.hip:
// the CU's store path moves ~10 B/cycle for one wave, ~30 B/cycle with eight), and the rollout stores 198 rows per
// workgroup: in k_rollout_pc (mppi_rollout_pc.h) that is ~11 % of the producer's time, on the critical chain of the
// workgroup.  Here a third wave does nothing but stores:
//
//   wave 0 (producer)   sampling + dynamics + control costs of time block s; the rollout state stays in its registers
//                       from block to block; clamped controls and (x, y) go to LDS, double buffered
//   wave 1 (distance)   min over the window points of the squared distance for the states of block s-1 (the O(K T^2) part)
//   wave 2 (store)      controls and states of block s-1: LDS -> HBM; it is the wave that waits on the store path
//
// The hand-off between the waves is a pair of LDS sequence numbers per direction, not a barrier (round 2): the producer
// publishes "block b is in LDS"; the distance and the store wave publish "block b is in my registers" as soon as they have
// LOADED it, i.e. before the distance loop and before the (slow) global stores.  The producer only ever waits for the
// loads of block b-2 before it overwrites that buffer, so neither the distance loop nor a store stall of block b-1 holds up
// block b+1: with a barrier per block the producer stood at the barrier for 16-18 % of the loop.
// With three waves per workgroup the chip also holds three waves per SIMD at K = 65 536.
// The epilogue (weights, fused sum w*u partials) deals the control rows to all three waves.  Arithmetic, noise and the
// summation order inside a row are those of k_rollout_pc; only the per-sample cost is the sum of the producer's and the
// distance wave's parts.
#pragma once
#include "mppi_rollout_pc.h"

namespace ccv {

constexpr int kR3Waves = 3;
// states of a block whose distance the distance wave computes; the store wave would take the rest.  8: measured best
// (6 + 2 is 4 us slower at K = 65 536: with eight waves per CU storing, the store wave is busy most of a block time)
constexpr int kR3CStates = 8;
constexpr int kR3RB = 12;   // rows per LDS transpose batch in the epilogue: 3 waves x 12 x 65 doubles fit p + ab + c

// What the producer stages for the store wave besides the positions: the fp32 normals of the block (2 x 4 KB at u_dim = 2,
// 2 x 6 KB at u_dim = 3) -- they are what is stored (mppi_kernels.h, layout comment: z in place of u).
template <int MODEL>
struct R3Shared {
    static constexpr bool kStage = true;
    static constexpr int kPBuf = 2;
    // (p, ab, c are contiguous and are reused as the epilogue's transpose buffers)
    double p[2][kTU][2][kPcSamples];                       // absolute (x,y) of the 8 states of a block, double buffered
    double2 ab[kMaxH + 4];                                 // window coefficients, padded to a multiple of 4 points
    double c[kMaxH + 4];
    double cost[kR3Waves][kPcSamples];
    alignas(32) double nom[(kMaxH + 8) * udim_of(MODEL)];  // warm start u*
    float zs[2][kTU * udim_of(MODEL)][kPcSamples];         // normals of a block, double buffered
    // hand-off sequence numbers: [0] blocks the producer has finished writing, [1] / [2] blocks the distance / store wave has
    // taken into registers
    int seq[4];
};

// first part of the hand-off: see pc_publish / pc_wait_for in mppi_rollout_pc.h

// First chunk of the epilogue's re-read for this kernel's row dealing (units of kR3RB rows, wave w owns units w, w+3, ...):
// with the row of load i a compile-time distance from the wave's first row, an address costs one scalar multiply and
// one vector add instead of the ~12 scalar instructions of the generic clamped form (60 loads per wave: ~1.3 us).
template <class T, int RB = kR3RB, int NW = kR3Waves>
__device__ __forceinline__ void r3_update_fetch0(const RolloutArgs& A, T (&v)[kUpdCH], const int wv, const int mcount,
                                                 const int kk) {
    const size_t pitch = (size_t)A.pitch;
    const T* p0;
    if constexpr (std::is_same<T, float>::value) p0 = A.z + kk + (size_t)(wv * RB) * pitch;   // fused: the normals
    else p0 = A.u + kk + (size_t)(wv * RB) * pitch;
#pragma unroll
    for (int i = 0; i < kUpdCH; ++i) {
        const size_t rows_ahead = (size_t)((i / RB) * (NW * RB) + i % RB);   // constant after unrolling
        v[i] = 0;
        if (i < mcount) v[i] = p0[rows_ahead * pitch];
    }
}

template <int MODEL, int MODE>
__global__ __launch_bounds__(kR3Waves * 64, 3) void k_rollout_r3(const RolloutArgs Ak, const Window Wk) {
    constexpr bool FB = MODEL == CCV_MPPI_FULL_BODY;
    constexpr bool COST = MODE != MODE_ROLLOUT;
    __shared__ R3Shared<MODEL> sh;
    static_assert(sizeof(sh.p) + sizeof(sh.ab) + sizeof(sh.c) >= kR3Waves * kR3RB * (kPcSamples + 2) * sizeof(double), "epilogue buffers");
    const RolloutArgs A = with_resident_pose(Ak);
    const int H = A.H;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (COST) stage_window(A, Wk, sh, kR3Waves * 64);
    if constexpr (MODE == MODE_FUSED) pc_stage_nominal<MODEL>(A, sh, kR3Waves * 64);
    const int k = blockIdx.x * kPcSamples + lane;
    const bool live = k < A.K;
    const int kk = live ? k : A.K - 1;
    const uint32_t kg = (uint32_t)(A.k_offset + kk);
    double cost = 0.0;
    const int nblocks = (H + kTU - 1) / kTU;
    const int nstates = FB ? H - 2 : H;   // states that reach the path cost (dd:199 / fb:409)
    if (threadIdx.x < 4) sh.seq[threadIdx.x] = 0;
    __syncthreads();
    int* const seq_ready = &sh.seq[0];
    int* const seq_dist = &sh.seq[1];
    int* const seq_store = &sh.seq[2];
    if (wv == 0) {
        // ---------------- producer: all time blocks, state in registers
        if constexpr (FB && COST) cost += A.w_yaw * (A.x0[2] - A.yaw_ref0) * (A.x0[2] - A.yaw_ref0);   // fb:408
        PcState<MODEL> S;
        S.x = A.x0[0];
        S.y = A.x0[1];
        S.yaw = A.x0[2];
        S.roll = A.x0[3];
        S.pitch = A.x0[4];
        S.p_v = S.p_rv = S.p_sdir = S.p_c2 = S.p_c3 = S.p_ac = 0.0;
        S.p_cdir = 1.0;
        fast_sincos(A.x0[2], S.sn, S.cs);
        for (int b = 0; b < nblocks; ++b) {
            pc_rotate_priority(A, b);
            if (b >= 2) {   // the buffers of block b last held block b-2: both readers must have taken it
                pc_wait_for(seq_dist, b - 1);
                pc_wait_for(seq_store, b - 1);
            }
            const int nctl = min(kTU, H - 1 - b * kTU);   // steps of this block that carry controls
            if (nctl == kTU) pc_produce_batched<MODEL, MODE>(A, sh, S, cost, b, lane, k, kk, live, kg);
            else if (nctl >= kPartialMin) pc_produce_batched<MODEL, MODE, R3Shared<MODEL>, false, false, false, true>(A, sh, S, cost, b, lane, k, kk, live, kg, nullptr, nctl);
            else pc_produce<MODEL, MODE, false>(A, sh, S, cost, b, lane, k, kk, live, kg);   // (a short tail, or the final state only)
            pc_publish(seq_ready, b + 1);
        }
    } else if (wv == 1) {
        // ---------------- distance wave: the states of block b as soon as the producer has published it
        int prune_on = 1;
        for (int b = 0; b < nblocks; ++b) {
            pc_rotate_priority(A, b + 1);
            pc_wait_for(seq_ready, b + 1);
            bool taken = false;
            if constexpr (COST) {
                const int nv = min(kR3CStates, nstates - b * kTU);
                taken = nv > 0;
                switch (nv) {
                    case 8: pc_consume<8, MODEL>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 7: pc_consume<7, MODEL>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 6: pc_consume<6, MODEL>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 5: pc_consume<5, MODEL>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 4: pc_consume<4, MODEL>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 3: pc_consume<3, MODEL>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 2: pc_consume<2, MODEL>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 1: pc_consume<1, MODEL>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    default: break;
                }
            }
            if (!taken) pc_publish(seq_dist, b + 1);   // (nothing of this block reaches the path cost)
        }
    } else {
        // ---------------- store wave: controls (sampled here: MODE_FUSED) and states (not in MODE_COST) of block b, LDS ->
        // registers -> HBM.  Everything is read from LDS first and the buffer handed back before the first store issues:
        // a 512-byte store costs this wave 45-120 cycles, and the producer must not wait for 32 of them.
        constexpr int UD = udim_of(MODEL);
        const size_t pitch = (size_t)A.pitch;
        for (int b = 0; b < nblocks; ++b) {
            pc_rotate_priority(A, b + 1);
            pc_wait_for(seq_ready, b + 1);
            const int t0 = b * kTU;
            float zv[kTU * UD];
            double xv[kTU], yv[kTU];
            if constexpr (MODE == MODE_FUSED) {
#pragma unroll
                for (int r = 0; r < kTU * UD; ++r) zv[r] = sh.zs[b & 1][r][lane];
            }
            if constexpr (MODE != MODE_COST) {
#pragma unroll
                for (int tt = 0; tt < kTU; ++tt) {
                    xv[tt] = sh.p[b & 1][tt][0][lane];
                    yv[tt] = sh.p[b & 1][tt][1][lane];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            pc_publish(seq_store, b + 1);
            if constexpr (MODE == MODE_FUSED) {
                const int nrows = min(kTU, H - 1 - t0) * UD;   // control steps t < H-1
                static_for<kTU * UD>([&](auto RR) {
                    constexpr int r = decltype(RR)::value;
                    // rows are padded to a multiple of 64 samples (pitch): lanes past K write their padding slot
                    if (r < nrows) A.z[(size_t)(t0 * UD + r) * pitch + k] = zv[r];
                });
            }
            if constexpr (MODE != MODE_COST) {
                if (A.store_xy) {
                    const int nst = min(kTU, H - t0);           // states t < H
#pragma unroll
                    for (int tt = 0; tt < kTU; ++tt) {
                        if (tt < nst) {
                            CCV_STATE_STORE(&A.xs[(size_t)(t0 + tt) * pitch + k], xv[tt]);
                            CCV_STATE_STORE(&A.ys[(size_t)(t0 + tt) * pitch + k], yv[tt]);
                        }
                    }
                }
            }
        }
        // the other two waves re-read the control rows in the epilogue: all of this wave's stores are acknowledged before
        // the barrier below
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (A.prio_rotate) __builtin_amdgcn_s_setprio(0);
    if constexpr (COST) {
        using Rows = UpdRowsT<kR3RB, kR3Waves>;
        const int R = (H - 1) * udim_of(MODEL);
        UpdT<MODE> upd[kUpdCH];
        const Rows rows{R, wv};
        const int mcount = A.fuse_update ? rows.count() : 0;
        sh.cost[wv][lane] = cost;
        // the one barrier of the kernel: every wave is through its loop (the store wave with all its stores acknowledged),
        // the three cost parts are in LDS, and p / ab / c are dead
        pc_barrier_lds();
        // the re-read of this wave's share of the controls is issued before anything else (see pc_update_fetch)
        if (mcount > 0) r3_update_fetch0(A, upd, wv, mcount, kk);
        const double total = (sh.cost[0][lane] + sh.cost[1][lane]) + sh.cost[2][lane];
        const double wgt = live ? exp(-total / A.lambda) : 0.0;   // dd:219 (no min-cost shift)
        if (wv == 0 && live) {
            A.cost[k] = total;
            A.w[k] = wgt;
        }
        if (A.fuse_update) {
            // p, ab, c are dead (the loop's last barrier): a private transpose buffer per wave
            double* buf = &sh.p[0][0][0][0] + wv * (kR3RB * (kPcSamples + 2));
            pc_reduce_rows<kR3RB, MODEL>(A, sh, buf, upd, rows, mcount, wgt, lane, kk);
            if (wv == kR3Waves - 1) pc_block_stats(A, R, wgt, total, live, lane);   // (the wave with the fewest rows)
        }
    }
}

}  // namespace ccv
