// Four-wave rollout kernel for gfx950 (MI355X): one workgroup = 64 samples = noise wave + dynamics wave + distance wave +
// store wave.
//
// The three-wave kernel's producer (mppi_rollout_r3.h) is cut in two:
//
//   wave 0 (noise)      the fp32 normals of time block b (Philox4x32-10 + Box-Muller: integer / fp32 work only) -> LDS
//   wave 1 (dynamics)   controls from the normals, dynamics and control costs of block b-1; (x, y) -> LDS
//   wave 2 (distance)   min over the window points of the squared distance for the states of block b-2
//   wave 3 (store)      normals and states of block b-2: LDS -> HBM
//
// Why a fourth wave -- not for more overlap: at K = 65 536 the kernel's time is the time its instructions take to issue,
// whichever wave issues them (moving work between the waves changes nothing, profiles/
// r02_ab_same_box.txt r4c, r5j).  But (1) the hardware deals the waves of consecutive workgroups to the SIMDs of a CU in a
// fixed rotation (tools/microbench/wave_placement.hip): with four workgroups of four waves per CU every SIMD receives wave
// 0, 1, 2 and 3 of four different workgroups, one wave of each role, in every process -- with three waves per workgroup
// that depends on the dispatch, and the steering workload ran at 52 or 61 us by process; and (2) the noise and the dynamics
// wave are through their loops early, which lets the epilogue's re-read start before the kernel's barrier, and the noise
// wave needs nothing that is staged, which lets block 0's normals be made beside the staging.
// Hand-offs are LDS sequence numbers as in the three-wave kernel.  Arithmetic, noise and the summation order inside a row
// are those of k_rollout_pc / k_rollout_r3: the results are bit-identical.
#pragma once
#include "mppi_rollout_r3.h"

namespace ccv {

constexpr int kR4Waves = 4;
// rows per LDS transpose batch in the epilogue: a multiple of u_dim that fits the loop's buffers (12 at u_dim 2 and 3; 6: +1.4 us
// at C2), 15 for full body
template <int MODEL>
constexpr int kR4RB = MODEL == CCV_MPPI_FULL_BODY ? 15 : 12;

// The control rows of the epilogue, dealt to the four waves as contiguous ranges (first row a multiple of u_dim, so that a
// row's control dimension is a compile-time function of its position in the range: pc_reduce_rows).
//   waves 0, 1 (noise, dynamics)   the rows of all time blocks but the last three, half each: these waves are through their
//                                  loops one to two block times before the workgroup is, the rows have long been stored, and
//                                  their re-read is ISSUED BEFORE the kernel's barrier -- its latency (most of these rows
//                                  have left the L2 by then) passes while the distance and the store wave finish
//   waves 2, 3 (distance, store)   the rows of the last three blocks, half each, read after the barrier (L2 hits)
struct R4Rows {
    static constexpr int BR = 60;   // (what pc_reduce_rows asserts: ranges start at multiples of 2, 3 and 5)
    int first, n;
    __device__ __forceinline__ int count() const { return n; }
    __device__ __forceinline__ int row(const int m) const { return first + m; }
};
template <int UD>
__device__ __forceinline__ R4Rows r4_rows(const int H, const int nblocks, const int wv, int& nb_early) {
    const int R = (H - 1) * UD;
    nb_early = max(0, nblocks - 3);   // (round 2: 2: +0.8 us, 4: +0.3 us at C2; round 3, after the layout fix: 2: +0.4, 4: +-0.1)
    if (nb_early == 0) {
        // a horizon of three blocks or fewer (the reference default H = 15 is two): no row can be re-read early, so all of
        // them are dealt to ALL FOUR waves and read after the barrier -- with the early / late split the noise and the
        // dynamics wave had nothing and the other two a batch more each (H = 15: 14 + 14 rows in 2 batches -> 8 + 8 + 8 + 4 in 1)
        const int per = ((R / UD + kR4Waves - 1) / kR4Waves) * UD;
        const int first = min(R, wv * per);
        return R4Rows{first, min(per, R - first)};
    }
    const int r_early = min(R, nb_early * kTU * UD);
    const int half_e = ((r_early / UD + 1) / 2) * UD, half_l = (((R - r_early) / UD + 1) / 2) * UD;
    switch (wv) {
        case 0: return R4Rows{0, half_e};
        case 1: return R4Rows{half_e, r_early - half_e};
        case 2: return R4Rows{r_early, half_l};
        default: return R4Rows{r_early + half_l, R - r_early - half_l};
    }
}
// first chunk of a range: row i of the chunk is i rows past the range's first one (one scalar add per address)
template <class T>
__device__ __forceinline__ void r4_fetch0(const RolloutArgs& A, T (&v)[kUpdCH], const R4Rows& rows, const int kk) {
    const size_t pitch = (size_t)A.pitch;
    const char* p0;   // (uniform row pointer + 32-bit lane offset: scalar-base loads)
    if constexpr (std::is_same<T, float>::value) p0 = reinterpret_cast<const char*>(A.z + (size_t)rows.first * pitch);   // fused: the normals
    else p0 = reinterpret_cast<const char*>(A.u + (size_t)rows.first * pitch);
    const uint32_t koff = (uint32_t)kk * (uint32_t)sizeof(T);
#pragma unroll
    for (int i = 0; i < kUpdCH; ++i) {
        v[i] = 0;
        if (i < rows.n) v[i] = *reinterpret_cast<const T*>(p0 + (size_t)i * (pitch * sizeof(T)) + koff);
    }
}

template <int MODEL>
struct R4Shared {
    static constexpr bool kStage = true;
    static constexpr int kPBuf = 2;
    // (p, ab, c, zs are contiguous and are reused as the epilogue's transpose buffers)
    double p[2][kTU][2][kPcSamples];                       // absolute (x,y) of the 8 states of a block, double buffered
    double2 ab[kMaxH + 4];                                 // window coefficients, padded to a multiple of 4 points
    double c[kMaxH + 4];
    float zs[2][kTU * udim_of(MODEL)][kPcSamples];         // normals of a block, double buffered
    double cost[kR4Waves][kPcSamples];
    alignas(32) double nom[(kMaxH + 8) * udim_of(MODEL)];  // warm start u*
    // hand-off sequence numbers: [0] blocks whose normals are in LDS, [1] blocks whose states are in LDS, [2] / [3] blocks the
    // distance / store wave has taken into registers, [4] blocks whose stores to HBM are all acknowledged
    int seq[8];
    int nan_seen[kR4Waves];   // per wave: a thread of it staged a NaN of the warm start (see clampd_fast)
};

// The normals of time block b -> LDS (noise wave).  The same grouping of the Philox calls as pc_produce_batched (4 | 3 + 3):
// a normal does not depend on it.
template <int MODEL, class SH>
__device__ __forceinline__ void r4_noise_block(const RolloutArgs& A, SH& sh, const int b, const int lane, const uint32_t kg) {
    constexpr int UD = udim_of(MODEL), NCALL = kTU * UD / 4;
    auto group = [&](auto C0_, auto CN_) {
        constexpr int C0 = decltype(C0_)::value, CN = decltype(CN_)::value;
        float z[4 * CN];
        pc_block_normals<MODEL, C0, CN>(A, b, kg, z);
#pragma unroll
        for (int i = 0; i < 4 * CN; ++i) sh.zs[b & 1][4 * C0 + i][lane] = z[i];
    };
    using std::integral_constant;
    if constexpr (NCALL == 4) {
        group(integral_constant<int, 0>{}, integral_constant<int, 4>{});
    } else if constexpr (NCALL == 6) {
        group(integral_constant<int, 0>{}, integral_constant<int, 3>{});
        group(integral_constant<int, 3>{}, integral_constant<int, 3>{});
    } else {
        static_assert(NCALL == 10, "full body: 4 + 3 + 3 Philox calls");
        group(integral_constant<int, 0>{}, integral_constant<int, 4>{});
        group(integral_constant<int, 4>{}, integral_constant<int, 3>{});
        group(integral_constant<int, 7>{}, integral_constant<int, 3>{});
    }
}

// One Philox call's four normals of time block b -> LDS (the prologue: block 0's calls are dealt to all four waves, which have
// nothing else to do until the window and the warm start have arrived).  The same words, the same Box-Muller arithmetic:
// a normal does not depend on how the calls are grouped.
template <int MODEL, class SH>
__device__ __forceinline__ void r4_noise_call(const RolloutArgs& A, SH& sh, const int b, const int call, const int lane, const uint32_t kg) {
    constexpr int UD = udim_of(MODEL);
    const Philox4 r = philox4x32_10(kg, (uint32_t)((b * kTU * UD) >> 2) + (uint32_t)call, A.iter_lo, A.iter_hi, A.seed_lo, A.seed_hi);
    float z[4];
    box_muller_f32(r.x, r.y, z[0], z[1]);
    box_muller_f32(r.z, r.w, z[2], z[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) sh.zs[b & 1][4 * call + i][lane] = z[i];
}

// Staging of the window coefficients and the warm start in two halves, so that ALL global loads of the prologue are in flight
// together (one memory latency instead of two in a row: the kernel arguments were written by the host a moment ago and the
// warm start by the previous update kernel -- neither is in a cache) and something useful can be done in between.
template <int MODEL, int NT>
struct R4Staged {
    static constexpr int kPerThread = ((kMaxH - 1 + kTU) * udim_of(MODEL) + NT - 1) / NT;
    double wa, wb, wc;
    double nv[kPerThread];
    double S;
};
template <int MODEL, int NT>
__device__ __forceinline__ void r4_stage_issue(const RolloutArgs& A, const Window& Wk, const int tid, R4Staged<MODEL, NT>& L) {
    static_assert(NT >= kMaxH + 4, "one window point per thread");
    const int H = A.H, R = (H - 1) * udim_of(MODEL);
    L.wa = L.wb = 0.0;
    L.wc = INFINITY;
    if (tid < H) {
        if (A.frame) {   // (a wave-uniform choice: the resident loop's window, or the copy in the kernel arguments)
            const Window& W = A.frame->W;
            L.wa = W.a[tid];
            L.wb = W.b[tid];
            L.wc = W.c[tid];
        } else {
            L.wa = Wk.a[tid];
            L.wb = Wk.b[tid];
            L.wc = Wk.c[tid];
        }
    }
    const double* src = A.pending_vec ? A.pending_vec + 1 : A.nominal;
    L.S = A.pending_vec ? A.pending_vec[0] : 1.0;
#pragma unroll
    for (int i = 0; i < R4Staged<MODEL, NT>::kPerThread; ++i) {
        const int j = tid + i * NT;
        L.nv[i] = j < R ? src[j] : 0.0;
    }
}
// ... and the LDS half (what stage_window + pc_stage_nominal write, the same values); returns whether a value this thread
// staged is NaN (clampd_fast)
template <int MODEL, int NT, class SH>
__device__ __forceinline__ bool r4_stage_commit(const RolloutArgs& A, SH& sh, const int tid, const R4Staged<MODEL, NT>& L) {
    const int H = A.H, H4 = (H + 3) & ~3, R = (H - 1) * udim_of(MODEL);
    if (tid < H4) {
        sh.ab[tid] = make_double2(L.wa, L.wb);
        sh.c[tid] = L.wc;
    }
    bool bad = false;
#pragma unroll
    for (int i = 0; i < R4Staged<MODEL, NT>::kPerThread; ++i) {
        const int j = tid + i * NT;
        if (j < R + kTU * udim_of(MODEL)) {   // (zeros past the last row: a partial last block reads them, pc_produce_batched)
            // K sharded over devices: the all-reduced [sum w, sum w*u] has not been divided yet -- pc_stage_nominal's division,
            // bit for bit
            const double v = (A.pending_vec && j < R) ? L.nv[i] / L.S : L.nv[i];
            sh.nom[j] = v;
            bad |= v != v;
            if (blockIdx.x == 0 && j < R) {
                if (A.pending_vec) A.nominal_w[j] = v;
                A.nominal_used[j] = v;   // (the normals are stored, not the controls: kept with them)
            }
        }
    }
    if (A.pending_vec && blockIdx.x == 0 && tid == 0) A.stats_w[0] = L.S;
    return bad;
}

// The per-lane sample indices, made afresh where a role (or the epilogue) needs them: five values that are live from the
// first instruction to the last would otherwise be spilled at 128 registers (the asm statement keeps the compiler from
// merging the copies back into one).
struct R4Lane {
    int lane, k, kk;
    bool live;
    uint32_t kg;
};
__device__ __forceinline__ R4Lane r4_lane(const RolloutArgs& A) {
    R4Lane L;
    L.lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (all 64 lanes are active here)
    asm volatile("" : "+v"(L.lane));
    L.k = (int)blockIdx.x * kPcSamples + L.lane;
    L.live = L.k < A.K;
    L.kk = L.live ? L.k : A.K - 1;
    L.kg = (uint32_t)(A.k_offset + L.kk);
    return L;
}

// WIDE (diff drive, fused iteration): full-range sin / cos of every heading, for |w|max dt > pi/4 (pc_produce_batched)
// TAIL (fused iteration): the horizon's last block carries kPartialMin .. 7 control steps and is made as a masked batch
// (pc_produce_batched, PARTIAL) -- an instantiation of its own, chosen by the launcher from H: with the masked producer merely
// present in the dynamics wave's loop, the kernel that never runs it (C2: H - 1 = 6 * 8 + 1) was 1.3 us slower.
// Full body (round 3): the same four roles with one wave per SIMD -- its dynamics batch needs 250 registers and spills some thirty
// at a 256 cap; alone on its SIMD a wave may have 512 -- i.e. one workgroup per CU: the kernel for K up to one block of 64
// samples per CU, where the reference's own operating point lies (K = 10 000: 157 blocks on 256 CUs; in the two-wave kernel every
// wave was alone on its SIMD there too, and its producer made the 40 normals of a block itself).
template <int MODEL, int MODE, bool WIDE = false, bool TAIL = false>
__global__ __launch_bounds__(kR4Waves * 64, MODEL == CCV_MPPI_FULL_BODY ? 1 : 4) void k_rollout_r4(const RolloutArgs Ak, const Window Wk) {
    constexpr bool FB = MODEL == CCV_MPPI_FULL_BODY;
    constexpr bool COST = MODE != MODE_ROLLOUT;
    constexpr int UD = udim_of(MODEL);
    static_assert(!WIDE || (MODEL == CCV_MPPI_DIFF_DRIVE && MODE == MODE_FUSED), "the wide-turn form exists for the fused diff-drive iteration");
    static_assert(!TAIL || MODE == MODE_FUSED, "the stage-wise modes carry the masked producer anyway");
    __shared__ R4Shared<MODEL> sh;
    static_assert(offsetof(R4Shared<MODEL>, zs) + sizeof(sh.zs) >= kR4Waves * kR4RB<MODEL> * (kPcSamples + 2) * sizeof(double), "epilogue buffers");
    touch_rollout_args();
    const RolloutArgs A = with_resident_pose(Ak);
    // The prologue runs in all sixteen waves of a CU at once and SIMD arbitration is oldest first: the workgroup dispatched last
    // to a CU was through it 1.5 us after the first (stamps: staging barrier passed at 1.4 / 1.7 / 2.0 / 3.0 us by dispatch
    // rank) and carried that lag to the end of the kernel, which ends with the slowest.  Until the loops' rotation takes over,
    // the younger workgroup has the higher priority.
    if (A.prio_rotate) pc_set_priority((int)blockIdx.x / A.cu_count);
    const int H = A.H;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv == 0) CCV_DIAG_STAMP(A, 0);
    const int nblocks = (H + kTU - 1) / kTU;
    const int nstates = FB ? H - 2 : H;   // states that reach the path cost (dd:199 / fb:409)
    // blocks whose controls are made as a batch: their normals come from the noise wave -- all eight steps' worth, also for a
    // last block that uses fewer but at least kPartialMin (pc_produce_batched, PARTIAL); a shorter tail is the dynamics wave's
    // own, step by step (pc_produce)
    const int nfull = MODE == MODE_FUSED ? (TAIL ? (H - 1 + kTU - kPartialMin) / kTU : (H - 1) / kTU) : 0;
    int* const seq_noise = &sh.seq[0];
    // ---------------- prologue: ONE barrier.  Every thread issues its loads of the window and the warm start, the four waves
    // make the normals of time block 0 meanwhile -- its Philox calls dealt round (a lone wave needs as long for four
    // interleaved calls as the memory latency lasts; four waves are through theirs well inside it) -- and the dynamics wave
    // its own set-up; then the staged values go to LDS.  The barrier publishes all of it, block 0's normals included.
    // (sequence numbers: zero, but "normals of block 0 are in LDS"; seq[5..7] are not used any more)
    if (threadIdx.x < 8) sh.seq[threadIdx.x] = (threadIdx.x == 0 && nfull > 0) ? 1 : 0;
    PcState<MODEL> S;          // (the dynamics wave's rollout state)
    bool bad_nominal = false;
    if constexpr (MODE == MODE_FUSED) {
        R4Staged<MODEL, kR4Waves * 64> staged;
        r4_stage_issue<MODEL>(A, Wk, (int)threadIdx.x, staged);
        if (nfull > 0) {
            constexpr int NCALL = kTU * UD / 4;
            const R4Lane L = r4_lane(A);
            for (int c = wv; c < NCALL; c += kR4Waves) r4_noise_call<MODEL>(A, sh, 0, c, L.lane, L.kg);
        }
        if (wv == 0) CCV_DIAG_STAMP(A, 3);
        if (wv == 1) {
            S.x = A.x0[0];
            S.y = A.x0[1];
            S.yaw = A.x0[2];
            S.roll = A.x0[3];
            S.pitch = A.x0[4];
            S.p_v = S.p_rv = S.p_sdir = S.p_c2 = S.p_c3 = S.p_ac = 0.0;
            S.p_cdir = 1.0;
            fast_sincos(A.x0[2], S.sn, S.cs);
            CCV_DIAG_STAMP(A, 1);
        }
        bad_nominal = r4_stage_commit<MODEL>(A, sh, (int)threadIdx.x, staged);
        // a NaN in the warm start: every wave says what its threads saw, every wave reads all four words after the barrier
        const bool wave_bad = __builtin_amdgcn_ballot_w64(bad_nominal) != 0ull;
        if (r4_lane(A).lane == 0) sh.nan_seen[wv] = wave_bad ? 1 : 0;
    } else {
        if constexpr (COST) stage_window(A, Wk, sh, kR4Waves * 64);
        if (wv == 1) {
            S.x = A.x0[0];
            S.y = A.x0[1];
            S.yaw = A.x0[2];
            S.roll = A.x0[3];
            S.pitch = A.x0[4];
            S.p_v = S.p_rv = S.p_sdir = S.p_c2 = S.p_c3 = S.p_ac = 0.0;
            S.p_cdir = 1.0;
            fast_sincos(A.x0[2], S.sn, S.cs);
        }
    }
    __syncthreads();
    if (wv == 1) CCV_DIAG_STAMP(A, 2);
    // two-instruction clamps (clampd_fast): the host has checked sigma and the bounds, the staging found no NaN in u*
    const bool fast_clamp = MODE == MODE_FUSED && A.fast_clamp &&
                            __builtin_amdgcn_readfirstlane(sh.nan_seen[0] | sh.nan_seen[1] | sh.nan_seen[2] | sh.nan_seen[3]) == 0;
    int* const seq_ready = &sh.seq[1];
    int* const seq_dist = &sh.seq[2];
    int* const seq_store = &sh.seq[3];
    int* const seq_stored = &sh.seq[4];
    // the epilogue's rows of this wave (see R4Rows)
    int nb_early;
    R4Rows rows = r4_rows<UD>(H, nblocks, wv, nb_early);
    if (!(COST && A.fuse_update)) rows.n = 0;
    UpdT<MODE> upd[kUpdCH];
    // Preconditions of the early re-read (noise and dynamics wave, before the kernel's barrier): the rows were stored by ANOTHER
    // wave of this workgroup; they are visible to this wave's loads because (1) the store wave has waited for their
    // acknowledgement (counted vmcnt) before it raised seq_stored, (2) all waves of a workgroup share one CU's vector L1, which
    // is write-through, and none of these rows has been loaded by this CU before (nothing stale can sit in L1).  (2) does not
    // hold in threadgroup-split mode (tgsplit: a workgroup's waves on different CUs): the kernel must not be built or launched
    // with -mtgsplit / amdgpu-tgsplit.  hipcc's default is off, and the build (build.py) does not set it.
    // Controls are re-derived from the normals with the clamp form of the launch (fast_clamp: v_max / v_min); the rows of a
    // step-by-step tail (pc_produce: compare-and-select) can differ from that in ONE way -- the sign of a zero at a bound of
    // +-0 -- which no sum, cost or comparison can see (clampd_fast, mppi_kernels.h).
    auto early_fetch = [&]() {
        if constexpr (MODE == MODE_FUSED) {   // (stage-wise cost call: fp64 controls, 120 registers -- fetched after the barrier)
            if (rows.n > 0 && nb_early > 0) {   // (nb_early == 0: every wave reads its rows after the barrier)
                pc_wait_for(seq_stored, nb_early);   // the rows are in HBM / L2
                r4_fetch0(A, upd, rows, r4_lane(A).kk);
            }
        }
    };
    if (wv == 0) {
        // ---------------- noise wave
        if constexpr (MODE == MODE_FUSED) {
            const R4Lane L = r4_lane(A);
            const int lane = L.lane;
            const uint32_t kg = L.kg;
            for (int b = 1; b < nfull; ++b) {   // (block 0: above, beside the staging)
                r4_rotate_priority(A, b, 0);
                if (b >= 2) {   // zs[b & 1] last held block b-2: the dynamics wave is through it, the store wave has loaded it
                    pc_wait_for(seq_ready, b - 1);
                    pc_wait_for(seq_store, b - 1);
                }
                r4_noise_block<MODEL>(A, sh, b, lane, kg);
                pc_publish(seq_noise, b + 1);
            }
            CCV_DIAG_STAMP(A, 15);
        }
        if constexpr (COST) sh.cost[0][r4_lane(A).lane] = 0.0;
        if (A.prio_rotate) __builtin_amdgcn_s_setprio(0);
        early_fetch();
    } else if (wv == 1) {
        // ---------------- dynamics wave: all time blocks, state in registers
        const R4Lane L = r4_lane(A);
        const int lane = L.lane, k = L.k, kk = L.kk;
        const bool live = L.live;
        const uint32_t kg = L.kg;
        double cost = 0.0;
        if constexpr (FB && COST) cost += A.w_yaw * (A.x0[2] - A.yaw_ref0) * (A.x0[2] - A.yaw_ref0);   // fb:408
        for (int b = 0; b < nblocks; ++b) {
            r4_rotate_priority(A, b, 1);
            if (b >= 2) {   // the buffers of block b last held block b-2: both readers must have taken it
                pc_wait_for(seq_dist, b - 1);
                pc_wait_for(seq_store, b - 1);
            }
            const int nctl = min(kTU, H - 1 - b * kTU);   // steps of this block that carry controls
            if (b < nfull) {
                pc_wait_for(seq_noise, b + 1);
                if (b == 0) CCV_DIAG_STAMP(A, 4);
                // (two instantiations each: a NaN in the warm start is rare, but its results are to be the other kernels' bits too)
                if (!TAIL || nctl == kTU) {
                    if (fast_clamp)
                        pc_produce_batched<MODEL, MODE, R4Shared<MODEL>, true, true, WIDE>(A, sh, S, cost, b, lane, k, kk, live, kg);
                    else
                        pc_produce_batched<MODEL, MODE, R4Shared<MODEL>, true, false, WIDE>(A, sh, S, cost, b, lane, k, kk, live, kg);
                } else if constexpr (TAIL) {
                    // (six steps -- the reference's default horizon, H = 15 -- as an instantiation of their own: the masked batch
                    //  computes all eight and drops two; same box, kernel: full body K = 10 000 17.8 -> 17.3 us, steering
                    //  K = 1 000 14.3 -> 13.9 us per iteration, diff drive unchanged)
                    if (fast_clamp && nctl == 6)
                        pc_produce_batched<MODEL, MODE, R4Shared<MODEL>, true, true, WIDE, true, 6>(A, sh, S, cost, b, lane, k, kk, live, kg, nullptr, nctl);
                    else if (fast_clamp)
                        pc_produce_batched<MODEL, MODE, R4Shared<MODEL>, true, true, WIDE, true>(A, sh, S, cost, b, lane, k, kk, live, kg, nullptr, nctl);
                    else
                        pc_produce_batched<MODEL, MODE, R4Shared<MODEL>, true, false, WIDE, true>(A, sh, S, cost, b, lane, k, kk, live, kg, nullptr, nctl);
                }
            } else if (MODE != MODE_FUSED && nctl == kTU) {
                pc_produce_batched<MODEL, MODE>(A, sh, S, cost, b, lane, k, kk, live, kg);
            } else if (MODE != MODE_FUSED && nctl >= kPartialMin) {
                pc_produce_batched<MODEL, MODE, R4Shared<MODEL>, false, false, false, true>(A, sh, S, cost, b, lane, k, kk, live, kg, nullptr, nctl);
            } else {
                pc_produce<MODEL, MODE, false>(A, sh, S, cost, b, lane, k, kk, live, kg);   // (a short tail, or the final state only)
            }
            pc_publish(seq_ready, b + 1);
            if (b == 0) CCV_DIAG_STAMP(A, 5);
        }
        CCV_DIAG_STAMP(A, 6);
        if constexpr (COST) sh.cost[1][lane] = cost;
        if (A.prio_rotate) __builtin_amdgcn_s_setprio(0);
        early_fetch();
    } else if (wv == 2) {
        // ---------------- distance wave: the states of block b as soon as the dynamics wave has published it
        // (round 3 tried handing every second one of the last blocks to the noise wave, which is idle from two thirds of the
        //  kernel on: +1.5 us -- the loop is priced by the instructions all four waves of a SIMD issue, not by this wave's share)
        int prune_on = 1;
        const int lane = r4_lane(A).lane;
        double cost = 0.0;
        for (int b = 0; b < nblocks; ++b) {
            r4_rotate_priority(A, b, 2);
            pc_wait_for(seq_ready, b + 1);
            bool taken = false;
            if constexpr (COST) {
                const int nv = min(kR3CStates, nstates - b * kTU);
                taken = nv > 0;
                switch (nv) {
                    case 8: pc_consume<8, MODEL, R4Shared<MODEL>, true>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 7: pc_consume<7, MODEL, R4Shared<MODEL>, true>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 6: pc_consume<6, MODEL, R4Shared<MODEL>, true>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 5: pc_consume<5, MODEL, R4Shared<MODEL>, true>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 4: pc_consume<4, MODEL, R4Shared<MODEL>, true>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 3: pc_consume<3, MODEL, R4Shared<MODEL>, true>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 2: pc_consume<2, MODEL, R4Shared<MODEL>, true>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    case 1: pc_consume<1, MODEL, R4Shared<MODEL>, true>(A, sh, cost, b, lane, 0, &prune_on, seq_dist, b + 1); break;
                    default: break;
                }
            }
            if (!taken) pc_publish(seq_dist, b + 1);   // (nothing of this block reaches the path cost)
            if (b == 0) CCV_DIAG_STAMP(A, 14);
        }
        CCV_DIAG_STAMP(A, 7);
        if constexpr (COST) sh.cost[2][lane] = cost;
    } else {
        // ---------------- store wave: normals (MODE_FUSED) and states (not in MODE_COST) of block b, LDS -> registers ->
        // HBM.  Everything is read from LDS first and the buffer handed back before the first store issues.
        const size_t pitch = (size_t)A.pitch;
        const R4Lane L = r4_lane(A);
        const int lane = L.lane;
        const uint32_t koff4 = (uint32_t)L.k * 4u, koff8 = (uint32_t)L.k * 8u;
        for (int b = 0; b < nblocks; ++b) {
            r4_rotate_priority(A, b, 3);
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
            // (the row pitch in bytes stays below 4 GB by construction: 32-bit lane offsets)
            if constexpr (MODE == MODE_FUSED) {
                const int nrows = min(kTU, H - 1 - t0) * UD;   // control steps t < H-1
                char* const zrow = reinterpret_cast<char*>(A.z + (size_t)(t0 * UD) * pitch);
                static_for<kTU * UD>([&](auto RR) {
                    constexpr int r = decltype(RR)::value;
                    // rows are padded to a multiple of 64 samples (pitch): lanes past K write their padding slot
                    if (r < nrows) pc_store_f32(zrow + (size_t)r * (pitch * 4), koff4, zv[r]);
                });
            }
            if constexpr (MODE != MODE_COST) {
                if (A.store_xy) {
                    const int nst = min(kTU, H - t0);           // states t < H
                    char* const xrow = reinterpret_cast<char*>(A.xs + (size_t)t0 * pitch);
                    char* const yrow = reinterpret_cast<char*>(A.ys + (size_t)t0 * pitch);
#pragma unroll
                    for (int tt = 0; tt < kTU; ++tt) {
                        if (tt < nst) {
                            pc_store_f64_stream(xrow + (size_t)tt * (pitch * 8), koff8, xv[tt]);
                            pc_store_f64_stream(yrow + (size_t)tt * (pitch * 8), koff8, yv[tt]);
                        }
                    }
                }
            }
            if constexpr (MODE == MODE_FUSED) {
                // all stores of the blocks before this one are acknowledged once no more than this block's own are
                // outstanding (vector-memory operations complete in order); a partial block waits for everything
                constexpr int kRowsZ = kTU * UD, kRowsAll = kTU * UD + 2 * kTU;
                static_assert(kRowsAll <= 63, "vmcnt is a 6-bit counter");
                if (t0 + kTU <= H - 1 && A.store_xy) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRowsAll) : "memory");
                else if (t0 + kTU <= H - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRowsZ) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pc_publish(seq_stored, b);
            }
        }
        // the other waves re-read the rows of normals in the epilogue: all of this wave's stores are acknowledged before the
        // barrier below
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (COST) sh.cost[3][lane] = 0.0;
    }
    // (the epilogue at equal priorities: youngest-first there, as in the prologue, was measured 1.4 us SLOWER -- the epilogue is
    //  bound by LDS bandwidth, and strict priorities only serialise it)
    if (A.prio_rotate) __builtin_amdgcn_s_setprio(0);
    if constexpr (COST) {
        const int R = (H - 1) * UD;
        const int mcount = rows.n;
        const R4Lane L = r4_lane(A);
        const int lane = L.lane, k = L.k, kk = L.kk;
        const bool live = L.live;
        // the one barrier of the kernel: every wave is through its loop (the store wave with all its stores acknowledged),
        // the cost parts are in LDS, and p / ab / c / zs are dead
        pc_barrier_lds();
        if (wv == 0) CCV_DIAG_STAMP(A, 8);
        // A horizon of one or two time blocks (the reference default H = 15): every normal of it is still in sh.zs -- block b in
        // zs[b & 1] -- so the rows come from there instead of from memory (0.7 us of load latency on the epilogue's chain).  The
        // transpose buffers overlap zs: every wave has its rows in registers before any of them is written (second barrier).
        bool from_lds = false;
        if constexpr (MODE == MODE_FUSED) {
            from_lds = nblocks <= 2;
            if (from_lds) {
                constexpr int kMaxLdsRows = 4 * UD;   // (rows per wave there: at most ceil(15 / 4) steps' worth)
                static_assert(kMaxLdsRows <= kUpdCH, "first chunk");
#pragma unroll
                for (int i = 0; i < kMaxLdsRows; ++i) {
                    const int row = rows.first + min(i, max(mcount - 1, 0));
                    upd[i] = sh.zs[(row / (kTU * UD)) & 1][row % (kTU * UD)][lane];
                }
                pc_barrier_lds();
            }
        }
        if (!from_lds && (MODE != MODE_FUSED || wv >= 2 || nb_early == 0) && mcount > 0) r4_fetch0(A, upd, rows, kk);
        const double total = ((sh.cost[0][lane] + sh.cost[1][lane]) + sh.cost[2][lane]) + sh.cost[3][lane];
        const double wgt = live ? exp(-total / A.lambda) : 0.0;   // dd:219 (no min-cost shift)
        if (wv == 0 && live) {
            A.cost[k] = total;
            A.w[k] = wgt;
        }
        if (wv == 0) CCV_DIAG_STAMP_VALUE(A, 9, wgt);
        if (wv == 2) CCV_DIAG_STAMP_VALUE(A, 11, wgt);
        if (A.fuse_update) {
            double* buf = &sh.p[0][0][0][0] + wv * (kR4RB<MODEL> * (kPcSamples + 2));
            pc_reduce_rows<kR4RB<MODEL>, MODEL, true>(A, sh, buf, upd, rows, mcount, wgt, lane, kk, fast_clamp);
            if (wv == kR4Waves - 1) pc_block_stats(A, R, wgt, total, live, lane);   // (the wave with the fewest rows)
        }
        if (wv == 0) CCV_DIAG_STAMP(A, 10);
        if (wv == 2) CCV_DIAG_STAMP(A, 12);
        if (wv == 3) CCV_DIAG_STAMP(A, 13);
    }
}

}  // namespace ccv
