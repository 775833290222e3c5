// Two waves per 64 samples at FOUR waves per SIMD: k_rollout_d2 (full body, fused iteration, K beyond a few blocks per CU).
//
// The one-wave kernel (mppi_rollout_solo.h) needs 256 registers for the full-body model, so a SIMD holds two of its waves --
// and K = 131 072 (C4) provides exactly two per SIMD.  A wave issues at most one vector instruction every 8 cycles
// (tools/microbench/valu_rates.hip), in practice every 11: two waves leave a third of the SIMD's issue slots empty, and the
// instruction count of the model (544 per sample-step) does the rest.  The four-wave kernel's split (noise / dynamics / distance /
// store) keeps four waves on a SIMD but its dynamics wave carries half of the model's instructions: the longest role sets a
// workgroup's pace, and K = 131 072 would need two rounds of workgroups.
// Here the 64 samples of a workgroup are shared by TWO waves of at most 128 registers each, eight workgroups per CU, all 2 048
// of C4 resident at once:
//   wave 0  dynamics: controls from the normals, rotations, ZMP and control-cost terms, positions      (~250 instructions / step)
//   wave 1  the normals (Philox + Box-Muller, stored by itself) and the distance to the window         (~150 + ~140 / step)
// To fit 128 registers the dynamics run in sub-batches of FOUR steps (d2_produce: pc_produce_batched's full-body arithmetic
// step for step -- the rotation chains restart at the block's first step as they do there, and are carried from the first
// half of a block to the second in the state -- so the results are the other kernels' bits), and to fit eight workgroups into
// 160 KB of LDS the hand-off buffers hold four steps as well: two halves of normals (wave 1 -> wave 0, 2 x 5 KB) and ONE buffer of
// four positions (wave 0 -> wave 1, 4 KB; wave 1 moves them to registers at once and runs the pruned distance loop over all eight
// states of a block when it has both halves).  The warm start is not staged: it is read through the scalar cache where it is
// used (wave-uniform addresses).  Sequence numbers in LDS, no barrier in the time loop (as in mppi_rollout_r4.h).
//
// Launched by the host when: full body, fused iteration, the horizon's last block is a batch ((H - 1) mod 8 >= kPartialMin),
// no deferred division pending (K sharded over devices: RolloutArgs::pending_vec), more than kD2MinBlocksPerCu blocks of 64
// samples per CU.  Everything else keeps its kernel.
#pragma once
#include "mppi_rollout_pc.h"

namespace ccv {

constexpr int kD2NT = 4;      // steps per hand-off
constexpr int kD2SB = 1;      // steps per sub-batch of the dynamics (registers: 128 per wave; two steps spill 25)
constexpr int kD2RB = 10;     // epilogue: rows per transpose batch (a multiple of u_dim = 5; 2 x 10 x 66 doubles fit the loop's buffers)

template <int MODEL>
struct D2Shared {
    static constexpr bool kStage = false;                  // (the waves store what they make themselves)
    static constexpr int kPBuf = 1;
    static constexpr int kZRows = kD2NT * udim_of(MODEL);
    double2 ab[kMaxH + 4];                                 // window coefficients, padded to a multiple of 4 points
    double c[kMaxH + 4];
    // time loop: zs[h] = the normals of a block's half h (row = step-in-half * u_dim + d), p = (x, y) - pose of four states;
    // epilogue: one transpose buffer per wave over the same bytes
    union {
        struct {
            float zs[2][kZRows][kPcSamples];
            double p[kD2NT][2][kPcSamples];
        } loop;
        double tr[2][kD2RB * (kPcSamples + 2)];
    };
    double cost[kPcSamples];                               // wave 0's cost terms, for wave 1's total
    int seq[4];                                            // 0: halves of normals published  1: halves of positions published  2: halves taken
    int nan_seen[2];
};
static_assert(sizeof(D2Shared<CCV_MPPI_FULL_BODY>) <= 20 * 1024, "eight workgroups per CU");

// what the epilogue needs of the warm start (pc_reduce_rows reads sh.nom): the array the launch was made around
struct D2Nominal {
    const double* __restrict__ nom;
};

// ---------------------------------------------------------------------------------------------------------------
// dynamics of steps T0 .. T0+kD2SB-1 of block b: pc_produce_batched's full-body path, kD2SB steps wide.
// PARTIAL: the horizon's last block, nctl (< 8) of its steps carry controls -- computed in full, masked where it leaves.
// Before the positions go to LDS the wave waits for `taken` to reach `taken_value` (wave 1 has the previous half in registers).
// ---------------------------------------------------------------------------------------------------------------
// T0 (the sub-batch's first step inside the block) is a run-time value: the sub-batches of a block are the iterations of a real
// loop -- unrolled, the compiler moves loads and common subexpressions of later sub-batches to the front and 100-290 registers
// spill (measured for sub-batches of 4, 2 and 1 steps alike).
template <int MODEL, bool FASTCLAMP, bool PARTIAL>
__device__ __forceinline__ void d2_produce(const RolloutArgs& A, D2Shared<MODEL>& sh, PcState<MODEL>& S, double& cost, const int b,
                                           const int T0, const int lane, const int k, const int nctl, int* const taken,
                                           const int taken_value) {
    static_assert(MODEL == CCV_MPPI_FULL_BODY, "full body only");
    constexpr int UD = udim_of(MODEL), NT = kD2SB;
    const int HALF = T0 / kD2NT, TH = T0 % kD2NT;   // TH: first step inside the half
    const int H = A.H;
    const int t0 = b * kTU;
    const int R = (H - 1) * UD;
    const size_t pitch = (size_t)A.pitch;
    const double dt = A.dt;
    // ---- 1. controls: libstdc++ normal_distribution, ret * stddev + mean (dd:96-97), then clamp (dd:98-99)
    double u[NT][UD];
    static_for<NT * UD>([&](auto II) {
        constexpr int i = decltype(II)::value;
        constexpr int tt = i / UD, d = i % UD;
        const int row = (t0 + T0) * UD + i;   // wave-uniform: a scalar load
        // (rows past the horizon's end: the normal is drawn and dropped, the warm start reads as 0 -- pc_produce_batched, PARTIAL)
        const double mean = (!PARTIAL || row < R) ? A.nominal[PARTIAL ? min(row, R - 1) : row] : 0.0;
        u[tt][d] = pc_control_from_normal_at<MODEL, d, FASTCLAMP>(A, sh.loop.zs[HALF][TH * UD + i][lane], mean);
    });
    // ---- 2. heading, roll, pitch recurrences (fb:449-451)
    double yawv[NT + 1], rollv[NT + 1], pitchv[NT + 1];
    yawv[0] = S.yaw;
    rollv[0] = S.roll;
    pitchv[0] = S.pitch;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        yawv[tt + 1] = yawv[tt] + u[tt][1] * dt;
        rollv[tt + 1] = rollv[tt] + u[tt][3] * dt;
        pitchv[tt + 1] = pitchv[tt] + u[tt][4] * dt;
    }
    // ---- 3. sin / cos: full evaluations at the block's first step, rotations by the short polynomials after it
    double sn[NT], cs[NT], fb_sd[NT], fb_cd[NT], fb_sr[NT], fb_cr[NT], fb_cp[NT];
    {
        double turn[NT], st_[NT], ct_[NT], dir[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            turn[tt] = u[tt][1] * dt;
            dir[tt] = u[tt][2];
        }
        kernel_sincos_n<NT>(turn, st_, ct_);
        kernel_sincos_n<NT>(dir, fb_sd, fb_cd);
        double sy, cy;
        if (T0 == 0) {
            fast_sincos(yawv[0], sy, cy);
        } else {
            sy = S.r_sy;
            cy = S.r_cy;
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            sn[tt] = fma(sy, fb_cd[tt], cy * fb_sd[tt]);
            cs[tt] = fma(cy, fb_cd[tt], -(sy * fb_sd[tt]));
            const double s2 = fma(sy, ct_[tt], cy * st_[tt]);
            const double c2 = fma(cy, ct_[tt], -(sy * st_[tt]));
            sy = s2;
            cy = c2;
        }
        S.r_sy = sy;
        S.r_cy = cy;
        double rinc[NT], pinc[NT], sri[NT], cri[NT], spi[NT], cpi[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            rinc[tt] = u[tt][3] * dt;
            pinc[tt] = u[tt][4] * dt;
        }
        kernel_sincos_n<NT>(rinc, sri, cri);
        kernel_sincos_n<NT>(pinc, spi, cpi);
        double sr, cr, sp, cp;
        if (T0 == 0) {
            fast_sincos(rollv[0], sr, cr);
            fast_sincos(pitchv[0], sp, cp);
        } else {
            sr = S.r_sr;
            cr = S.r_cr;
            sp = S.r_sp;
            cp = S.r_cp;
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            fb_sr[tt] = sr;
            fb_cr[tt] = cr;
            fb_cp[tt] = cp;
            const double s2 = fma(sr, cri[tt], cr * sri[tt]);
            const double c2 = fma(cr, cri[tt], -(sr * sri[tt]));
            sr = s2;
            cr = c2;
            const double s3 = fma(sp, cpi[tt], cp * spi[tt]);
            const double c3 = fma(cp, cpi[tt], -(sp * spi[tt]));
            sp = s3;
            cp = c3;
        }
        S.r_sr = sr;
        S.r_cr = cr;
        S.r_sp = sp;
        S.r_cp = cp;
    }
    // ---- 4. cost terms that do not need the window (fb:409-420, 468-486, 597-603)
    {
        const double mgz = A.fb_mass * A.fb_gz;   // (mass*gravity_).z
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int t = t0 + T0 + tt;
            {                                                                     // fb:409: t < H - 2
                const double cv = A.w_v * (u[tt][0] - A.v_ref) * (u[tt][0] - A.v_ref);       // fb:413
                const double cb = A.w_back * u[tt][0] * u[tt][0];                            // fb:420
                const bool in = t < H - 2;
                cost += in ? cv : 0.0;
                cost += (in && u[tt][0] < 0.0) ? cb : 0.0;
            }
            {   // t >= 1: finish index t-1: ZMP and roll-rate terms
                const double drive_accel = div_uniform(u[tt][0] - S.p_v, dt, A.inv_dt);      // fb:469
                const double ay = drive_accel * S.p_sdir + S.p_ac * S.p_cdir;                // fb:473
                const double hgdot_x = div_uniform(A.fb_Ixx * u[tt][3] - A.fb_Ixx * S.p_rv, dt, A.inv_dt);   // fb:479-481
                const double mo_x = (S.p_c2 * mgz + S.p_c3 * (A.fb_mass * ay)) - hgdot_x;    // fb:600
                const double zmp_y = mo_x / mgz;                                             // fb:601
                const double cz = A.w_zmp * zmp_y * zmp_y;                                   // fb:416
                const double cr = A.w_rollv * (u[tt][3] - S.p_rv) * (u[tt][3] - S.p_rv);     // fb:418
                const bool in = t >= 1 && (!PARTIAL || T0 + tt < nctl);   // (index t-1 <= H-3)
                cost += in ? cz : 0.0;
                cost += in ? cr : 0.0;
            }
            S.p_sdir = fb_sd[tt];
            S.p_cdir = fb_cd[tt];
            S.p_c2 = -A.fb_L * fb_sr[tt];                 // CoM.y (fb:482)
            S.p_c3 = A.fb_L * fb_cp[tt] * fb_cr[tt];      // CoM.z
            S.p_ac = u[tt][0] * u[tt][1];                 // fb:471
            S.p_v = u[tt][0];
            S.p_rv = u[tt][3];
        }
    }
    // ---- 5. positions (fb:445-447): to wave 1 through LDS (relative to the pose), to HBM
    double x = S.x, y = S.y;
    double xv[NT], yv[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        xv[tt] = x;
        yv[tt] = y;
        x = x + u[tt][0] * cs[tt] * dt;
        y = y + u[tt][0] * sn[tt] * dt;
    }
    if (TH == 0) pc_wait_for(taken, taken_value);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        sh.loop.p[TH + tt][0][lane] = xv[tt] - A.x0[0];
        sh.loop.p[TH + tt][1][lane] = yv[tt] - A.x0[1];
    }
    if (A.store_xy) {   // (padded rows: no `live` predicate, as in pc_produce_batched)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            if (!PARTIAL || T0 + tt <= nctl) {   // states t <= H-1
                CCV_STATE_STORE(&A.xs[(size_t)(t0 + T0 + tt) * pitch + k], xv[tt]);
                CCV_STATE_STORE(&A.ys[(size_t)(t0 + T0 + tt) * pitch + k], yv[tt]);
            }
        }
    }
    S.x = x;
    S.y = y;
    S.yaw = yawv[NT];
    S.roll = rollv[NT];
    S.pitch = pitchv[NT];
}

// the 20 normals of half `HALF` of block b (Philox calls 5 HALF .. 5 HALF + 4 of the block): to LDS for wave 0, to HBM
template <int MODEL, int HALF>
__device__ __forceinline__ void d2_noise(const RolloutArgs& A, D2Shared<MODEL>& sh, const int b, const int lane, const int k,
                                         const uint32_t kg) {
    constexpr int UD = udim_of(MODEL), NZ = kD2NT * UD, NC = NZ / 4;
    static_assert(NZ % 4 == 0, "whole Philox calls per half");
    float z[NZ];
    pc_block_normals<MODEL, HALF * NC, NC>(A, b, kg, z);
    const int R = (A.H - 1) * UD;
    const int row0 = (b * kTU + HALF * kD2NT) * UD;
    const size_t pitch = (size_t)A.pitch;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
        sh.loop.zs[HALF][i][lane] = z[i];
        if (row0 + i < R) A.z[(size_t)(row0 + i) * pitch + k] = z[i];   // (wave-uniform; padded rows: no `live` predicate)
    }
}

template <int MODEL, int MODE>
__global__ __launch_bounds__(2 * kPcSamples, 4) void k_rollout_d2(const RolloutArgs Ak, const Window Wk) {
    static_assert(MODEL == CCV_MPPI_FULL_BODY && MODE == MODE_FUSED, "full body, fused iteration");
    constexpr int UD = udim_of(MODEL);
    __shared__ D2Shared<MODEL> sh;
    const RolloutArgs A = with_resident_pose(Ak);
    const int H = A.H;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int R = (H - 1) * UD;
    const int nblocks = (H + kTU - 1) / kTU;
    const int nstates = H - 2;   // states that reach the path cost (fb:409)
    const int k = blockIdx.x * kPcSamples + lane;
    const bool live = k < A.K;
    const int kk = live ? k : A.K - 1;
    const uint32_t kg = (uint32_t)(A.k_offset + kk);
    int* const seq_noise = &sh.seq[0];
    int* const seq_pos = &sh.seq[1];
    int* const seq_taken = &sh.seq[2];
    // ---------------- prologue: window to LDS, a NaN in the warm start found, its copy for the read-back kept (workgroup 0)
    stage_window(A, Wk, sh, 2 * kPcSamples);
    bool bad = false;
    for (int j = threadIdx.x; j < R; j += 2 * kPcSamples) {
        const double v = A.nominal[j];
        bad |= v != v;
        if (blockIdx.x == 0) A.nominal_used[j] = v;   // (the normals are stored, not the controls: kept with them)
    }
    if (threadIdx.x < 4) sh.seq[threadIdx.x] = 0;
    const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;
    if (lane == 0) sh.nan_seen[wv] = wave_bad ? 1 : 0;
    __syncthreads();
    const bool fast_clamp = A.fast_clamp && __builtin_amdgcn_readfirstlane(sh.nan_seen[0] | sh.nan_seen[1]) == 0;
    double cost = 0.0;
#ifdef D2_NO_W0
    if (false) {
#else
    if (wv == 0) {
#endif
        // ---------------- dynamics wave
        cost += A.w_yaw * (A.x0[2] - A.yaw_ref0) * (A.x0[2] - A.yaw_ref0);   // fb:408
        PcState<MODEL> S;
        S.x = A.x0[0];
        S.y = A.x0[1];
        S.yaw = A.x0[2];
        S.roll = A.x0[3];
        S.pitch = A.x0[4];
        S.p_v = S.p_rv = S.p_sdir = S.p_c2 = S.p_c3 = S.p_ac = 0.0;
        S.p_cdir = 1.0;
        S.r_sy = S.r_cy = S.r_sr = S.r_cr = S.r_sp = S.r_cp = 0.0;
        for (int b = 0; b < nblocks; ++b) {
            const int nctl = min(kTU, H - 1 - b * kTU);   // steps of this block that carry controls (the host: 8, or 4 .. 7 in the last)
#pragma clang loop unroll(disable)
            for (int T0 = 0; T0 < kTU; T0 += kD2SB) {
                const int hn = 2 * b + T0 / kD2NT;         // this half's number
                if (T0 % kD2NT == 0) pc_wait_for(seq_noise, hn + 1);
                if (nctl == kTU) {
                    if (fast_clamp) d2_produce<MODEL, true, false>(A, sh, S, cost, b, T0, lane, k, nctl, seq_taken, hn);
                    else d2_produce<MODEL, false, false>(A, sh, S, cost, b, T0, lane, k, nctl, seq_taken, hn);
                } else {
                    if (fast_clamp) d2_produce<MODEL, true, true>(A, sh, S, cost, b, T0, lane, k, nctl, seq_taken, hn);
                    else d2_produce<MODEL, false, true>(A, sh, S, cost, b, T0, lane, k, nctl, seq_taken, hn);
                }
                if ((T0 + kD2SB) % kD2NT == 0) pc_publish(seq_pos, hn + 1);
            }
        }
        sh.cost[lane] = cost;
    } else if (
#ifdef D2_NO_W1
    false
#else
    true
#endif
    ) {
        // ---------------- noise + distance wave
        int prune_on = 1;
        d2_noise<MODEL, 0>(A, sh, 0, lane, k, kg);
        pc_publish(seq_noise, 1);
        d2_noise<MODEL, 1>(A, sh, 0, lane, k, kg);
        pc_publish(seq_noise, 2);
        for (int b = 0; b < nblocks; ++b) {
            double px[kTU], py[kTU];
            const bool more = b + 1 < nblocks;   // (the host admits horizons whose every block carries controls)
            auto half = [&](auto HALF_) {
                constexpr int HALF = decltype(HALF_)::value;
                const int hn = 2 * b + HALF;
                pc_wait_for(seq_pos, hn + 1);
#pragma unroll
                for (int tt = 0; tt < kD2NT; ++tt) {
                    px[HALF * kD2NT + tt] = sh.loop.p[tt][0][lane];
                    py[HALF * kD2NT + tt] = sh.loop.p[tt][1][lane];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the loads have returned before the buffer is given back
                pc_publish(seq_taken, hn + 1);
                // the normals of the next block's half: wave 0 has read zs[HALF] of this block (it published the positions made of it)
                if (more) {
                    d2_noise<MODEL, HALF>(A, sh, b + 1, lane, k, kg);
                    pc_publish(seq_noise, hn + 3);
                }
            };
            half(std::integral_constant<int, 0>{});
            half(std::integral_constant<int, 1>{});
            const int nv = min(kTU, nstates - b * kTU);
            if (nv == kTU) {
                pc_consume_at<kTU, MODEL, D2Shared<MODEL>, true>(A, sh, cost, px, py, lane, &prune_on);
            } else if (nv > 0) {
                // the horizon's last states: one at a time (their count varies; once per launch)
                for (int i = 0; i < nv; ++i) {
                    double qx[1], qy[1];
                    qx[0] = px[0];
                    qy[0] = py[0];
#pragma unroll
                    for (int j = 1; j < kTU; ++j) {
                        if (j == i) {
                            qx[0] = px[j];
                            qy[0] = py[j];
                        }
                    }
                    pc_consume_at<1, MODEL, D2Shared<MODEL>, true>(A, sh, cost, qx, qy, lane, &prune_on);
                }
            }
        }
    }
    // ---------------- weights and the workgroup's share of the update (dd:216-237): rows dealt to the two waves by time block
    __syncthreads();   // wave 0's cost is in LDS; the loop's buffers are free
    using Rows = UpdRowsT<kTU * UD, 2>;
    const Rows rows{R, wv};
    UpdT<MODE> upd[kUpdCH];
    const int mcount = A.fuse_update ? rows.count() : 0;
    if (mcount > 0) pc_update_fetch(A, upd, rows, 0, mcount, kk);   // (in flight during the exp below)
    // wave 1 holds the distance terms, wave 0 everything else: wave 1 forms the total and hands it back, so that both waves
    // weigh with the same bits
    if (wv == 1) sh.cost[lane] = sh.cost[lane] + cost;
    pc_barrier_lds();
    const double total = sh.cost[lane];
    const double wgt = live ? exp(-total / A.lambda) : 0.0;   // dd:219 (no min-cost shift)
    if (wv == 1 && live) {
        A.cost[k] = total;
        A.w[k] = wgt;
    }
    if (A.fuse_update) {
        const D2Nominal nomv{A.nominal};
        pc_reduce_rows<kD2RB, MODEL, true>(A, nomv, &sh.tr[wv][0], upd, rows, mcount, wgt, lane, kk, fast_clamp);
        if (wv == 1) pc_block_stats(A, R, wgt, total, live, lane);
    }
}

}  // namespace ccv
