// k_rollout_pc: the two-wave sample + rollout + cost kernel ("producer / consumer" time blocks) -- the production kernel
// of the full-body model -- and the building blocks (pc_produce*, pc_consume, pc_reduce_rows, ...) it shares with the
// three-wave kernel of the diff-drive and steering models (mppi_rollout_r3.h).
//
// One sample per lane.  A workgroup is TWO waves that own the same 64 samples and alternate roles over time blocks
// of kTU = 8 steps:
//
//   step s:   wave (s & 1)        PRODUCES block s   : Philox noise -> clamped controls -> HBM, Euler rollout with sincos,
//                                                      x,y -> HBM, control / ZMP cost terms; hands (x,y) of its 8 states
//                                                      and the end state to the other wave through LDS
//             wave ((s - 1) & 1)  CONSUMES block s-1 : squared distance of its 8 states to every point of the reference
//                                                      window (2 FMA + 1 MIN per pair, the O(K*H^2) core) -> path cost
//             barrier
//
// Why: at K = 65 536 a plain one-sample-per-lane grid is exactly one wave per SIMD (1024 waves on 1024 SIMDs) and
// every stall is exposed -- the wave that waits for a store slot, an LDS broadcast or a Philox multiply leaves its
// SIMD idle (measured 75 us for one iteration vs 41 us per 65 536 samples once 8 waves share a SIMD).  Splitting the
// sample's work by time block gives 2048 waves (2 per SIMD) with complementary instruction mixes (integer/trig/stores
// vs straight fp64 FMA/MIN) and no duplicated arithmetic: per sample the operations are exactly those of
// k_rollout_cost; only the order in which one sample's cost terms are added differs.
#pragma once
#include "fast_trig.h"
#include "mppi_kernels.h"

namespace ccv {

constexpr int kPcWaves = 2;


// element d of a 5-entry kernel-argument array for a compile-time d
template <int D>
__device__ __forceinline__ double arg5(const double (&v)[5]) { return v[D]; }

// lane state handed from the producer of block b to the producer of block b+1 (through LDS)
template <int MODEL>
struct PcState {
    double x, y, yaw;
    double roll, pitch;                                 // full body only
    double p_v, p_rv, p_sdir, p_cdir, p_c2, p_c3, p_ac;  // full body: step t-1 quantities for the ZMP term (fb:468-486)
    double sn, cs;                                      // diff drive: sin / cos of yaw, advanced by rotation (pc_produce_batched)
};
template <int MODEL>
constexpr int kPcStateWords = MODEL == CCV_MPPI_FULL_BODY ? 12 : (MODEL == CCV_MPPI_DIFF_DRIVE ? 5 : 3);

template <int MODEL>
struct PcShared {
    // kStage: the producer hands its normals (sh.zs) and states to a store wave through LDS instead of storing them itself
    // (mppi_rollout_r3.h, mppi_rollout_r4.h); here it stores them to HBM directly
    static constexpr bool kStage = false;
    static constexpr int kPBuf = 2;              // buffers of p: the block being produced and the one being consumed
    double2 ab[kMaxH + 4];                                 // window coefficients, padded to a multiple of 4 points
    double c[kMaxH + 4];
    double p[2][kTU][2][kPcSamples];                       // (x,y) - pose of the 8 states of a block, double buffered
    // producer -> next producer.  One buffer is enough: the producer of block s reads it first thing and writes it last,
    // the other wave reads it only after the barrier that ends the step
    double st[kPcStateWords<MODEL>][kPcSamples];
    double cost[kPcWaves][kPcSamples];
    alignas(32) double nom[(kMaxH + 8) * udim_of(MODEL)];  // warm start u* (see pc_stage_nominal)
    // full body: normals made ahead by each wave for the block it produces next (pc_noise_ahead)
    float ahead[MODEL == CCV_MPPI_FULL_BODY ? kPcWaves : 1][MODEL == CCV_MPPI_FULL_BODY ? 16 : 1][kPcSamples];
};

// The warm start u* is staged in LDS once per workgroup.  Read straight from memory inside the time loop (scalar or
// vector loads) every block of 8 steps exposes one cache-miss latency -- ~1500 cycles, measured -- on the producer chain.
// Returns whether one of the values this thread staged is NaN (see clampd_fast).
template <int MODEL, class SH>
__device__ __forceinline__ bool pc_stage_nominal(const RolloutArgs& A, SH& sh, const int nthreads,
                                                 const int tid = threadIdx.x) {   // (tid: 0 .. nthreads-1)
    const int R = (A.H - 1) * udim_of(MODEL);
    bool bad = false;
    if (A.pending_vec) {
        // K sharded over devices: the all-reduced [sum w, sum w*u] has not been divided yet -- do it here (the division
        // k_apply_partials would do, bit for bit) instead of spending a kernel launch on 100 quotients
        const double S = A.pending_vec[0];
        for (int j = tid; j < R + kTU * udim_of(MODEL); j += nthreads) {
            const double v = j < R ? A.pending_vec[1 + j] / S : 0.0;
            sh.nom[j] = v;
            bad |= v != v;
            if (blockIdx.x == 0 && j < R) {
                A.nominal_w[j] = v;
                A.nominal_used[j] = v;
            }
        }
        if (blockIdx.x == 0 && tid == 0) A.stats_w[0] = S;
        return bad;
    }
    for (int j = tid; j < R + kTU * udim_of(MODEL); j += nthreads) {
        const double v = j < R ? A.nominal[j] : 0.0;
        sh.nom[j] = v;
        bad |= v != v;
        if (blockIdx.x == 0 && j < R) A.nominal_used[j] = v;   // (the normals are stored, not the controls: kept with them)
    }
    return bad;
}

// The control of row n = t * u_dim + d from its normal: the samplers' arithmetic (double(z) * sigma + mean, clamp, steer_off),
// so the same bits as the value the rollout used.  D = n % u_dim is a template argument: the clamp bounds then are scalar
// kernel arguments (a run-time dimension would cost an integer division and two LDS reads with their waits per value).
template <int MODEL, int D, bool FASTCLAMP = false>
__device__ __forceinline__ double pc_control_from_normal_at(const RolloutArgs& A, const float z, const double nominal) {
    double v = (double)z * A.sigma + nominal;
    v = clampd_as<FASTCLAMP>(v, arg5<D>(A.umin), arg5<D>(A.umax));
    if constexpr (MODEL == CCV_MPPI_FULL_BODY && D == 2) {
        if (A.steer_off) v = 0.0;   // fb:517
    }
    return v;
}
template <int MODEL, int D, bool FASTCLAMP = false, class SH>
__device__ __forceinline__ double pc_control_from_normal(const RolloutArgs& A, const SH& sh, const float z, const int n) {
    return pc_control_from_normal_at<MODEL, D, FASTCLAMP>(A, z, sh.nom[n]);
}

// The candidate states x, y are written once and not read again by this kernel: streaming (non-temporal) stores keep
// them from displacing the controls, which the epilogue re-reads, from L2 and from piling up as dirty lines that the
// end-of-kernel write-back has to drain.
#define CCV_STATE_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))

// Stores with a scalar base: address = (uniform 64-bit row pointer) + (32-bit lane offset).  The compiler hoists the
// zero-extension of the lane offset out of the loop and then adds 64-bit vector addresses (one v_lshl_add_u64 per store,
// 198 per workgroup at C2); written out, a store issues no vector instruction for its address.  The compiler does not count
// these among the outstanding vector-memory operations: its own s_waitcnt vmcnt(n) then waits for more than it needs to, never
// for less (the counter covers them and operations complete in order), and the four-wave kernel's store wave counts its own.
// The row pitch in bytes stays below 4 GB by construction (32-bit lane offsets).  Used by the four-wave kernel's store wave
// (C2 -0.3 us, C3 -1.2 us); in the one-wave kernels the row pointers cost scalar registers they do not have (full body:
// 97 spilled SGPRs, 229.7 -> 232.0 us), so those keep the compiler's addressing.
__device__ __forceinline__ void pc_store_f32(char* const row, const uint32_t lane_off, const float v) {
    asm volatile("global_store_dword %0, %1, %2" ::"v"(lane_off), "v"(v), "s"(row) : "memory");
}
__device__ __forceinline__ void pc_store_f64_stream(char* const row, const uint32_t lane_off, const double v) {   // (non-temporal)
    asm volatile("global_store_dwordx2 %0, %1, %2 nt" ::"v"(lane_off), "v"(v), "s"(row) : "memory");
}

// Wave arbitration on a SIMD is by priority, then age: with every workgroup at priority 0 the workgroup dispatched first
// to a CU runs almost unimpeded and the last one gets the left-over issue slots -- measured at K = 65 536, four
// workgroups per CU: 26 / 33 / 41 / 45 us for the same work, and the kernel ends with the slowest.  Alternating which
// half of a CU's workgroups is favoured, once per time block, evens them out (28 / 33 / 39 / 40 us; kernel -4 us).
// rank = position of the workgroup in its CU's dispatch order (workgroups go round-robin over the CUs).
__device__ __forceinline__ void pc_set_priority(const int p) {   // (s_setprio takes an immediate)
    switch (p) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}
// prio_rotate 1: two levels, alternating halves (two- and three-wave kernels); 2: four levels, level = (rank + s) mod 4 (the
// four-wave kernel, mppi_rollout_r4.h: its waves call this with s = time block + role)
__device__ __forceinline__ void pc_rotate_priority(const RolloutArgs& A, const int s) {
    if (!A.prio_rotate) return;
    const int rank = (int)blockIdx.x / A.cu_count;
    if (A.prio_rotate == 1) {
        if ((rank ^ s) & 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    } else {
        pc_set_priority((rank + s) & 3);
    }
}

// four-wave kernel: the level from the workgroup's dispatch rank, the time block and the wave's role.
//   prio_rotate >= 16: (rank + level[role] + b) mod 4 with level[] = the four base-4 digits of prio_rotate - 16 (digit 0: noise
//   wave, 1: dynamics, 2: distance, 3: store) -- the default, with one table per model (ccv_mppi_capi.hip: kR4PrioLevels);
//   2: level[role] = role, i.e. (rank + role + b); 5: (role - rank - b); 3, 4: the other two sign combinations (CCV_MPPI_PRIO).
// What matters, measured at C2 on one box (gpurun_out/r3bn, r3bj): no priorities 36.6 us; the prologue's rank priorities only
// 35.3; rank only, constant 34.5; (rank + role) 34.5; (rank + b) 32.4; (rank + role + b) 31.7; (rank + role - b) 32.9 -- the
// rotation with the time block is worth 3 us, its direction 1, the role term 0.7.  And WHICH role sits on which level another
// 0.7 us at C2 and 2 us at C3 (all 24 assignments, gpurun_out/r3br): diff drive noise 3 / dynamics 2 / distance 1 / store 0
// 31.9 us against 32.6 for 0 / 1 / 2 / 3, steering 2 / 3 / 1 / 0 39.7 against 41.7 (and 42.1 with diff drive's) -- the roles'
// blocks differ in length between the models, and the best interleaving with them.
__device__ __forceinline__ void r4_rotate_priority(const RolloutArgs& A, const int b, const int role) {
    if (!A.prio_rotate) return;
    const int rank = (int)blockIdx.x / A.cu_count;
    if (A.prio_rotate >= 16) {
        pc_set_priority((rank + (((A.prio_rotate - 16) >> (2 * role)) & 3) + b) & 3);
        return;
    }
    switch (A.prio_rotate) {
        case 2: pc_set_priority((rank + role + b) & 3); break;
        case 3: pc_set_priority((rank + role - b) & 3); break;
        case 4: pc_set_priority((role - rank + b) & 3); break;
        default: pc_set_priority((role - rank - b) & 3); break;
    }
}

// LDS sequence numbers for hand-offs without a barrier (mppi_rollout_r3.h): written by one wave, polled by another.  The
// LDS operations of a wave execute in order, so a number written after the data (or after the loads have returned) needs
// no fence -- and must not get one: a release at workgroup scope would also wait for the wave's global stores.  Relaxed
// workgroup-scope atomics compile to plain ds_write_b32 / ds_read_b32; the empty asm statements keep the compiler from
// moving other memory accesses across them.
__device__ __forceinline__ void pc_publish(int* flag, const int value) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void pc_wait_for(int* flag, const int value) {
    asm volatile("" ::: "memory");
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < value)
        __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// Workgroup barrier for the block hand-off.  The two waves exchange data through LDS only, so the barrier waits for LDS
// (lgkmcnt) and not, as __syncthreads() does, for the acknowledgement of every control / state store still in flight.
__device__ __forceinline__ void pc_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------------------
// producer: steps t0 .. t0+7 of one sample (sampling + predict_NextState + control costs).  FULL: every step of the
// block carries controls (t0 + 8 <= H - 1), so the body is branch-free.
// ---------------------------------------------------------------------------------------------------------------
template <int MODEL, int MODE, bool FULL, class SH>
__device__ __forceinline__ void pc_produce(const RolloutArgs& A, SH& sh, PcState<MODEL>& S, double& cost,
                                           const int b, const int lane, const int k, const int kk, const bool live,
                                           const uint32_t kg) {
    constexpr int UD = udim_of(MODEL);
    constexpr bool FB = MODEL == CCV_MPPI_FULL_BODY;
    constexpr bool COST = MODE != MODE_ROLLOUT;
    const int H = A.H;
    const int t0 = b * kTU;
    const size_t pitch = (size_t)A.pitch;
    const double dt = A.dt;
    float zq[4] = {0.f, 0.f, 0.f, 0.f};
    double4 nom = make_double4(0.0, 0.0, 0.0, 0.0);
    static_for<kTU>([&](auto TT) {
        constexpr int tt = decltype(TT)::value;
        const int t = t0 + tt;
        if constexpr (SH::kStage) {   // absolute: the store wave writes them out, the distance wave subtracts the pose
            sh.p[b & (SH::kPBuf - 1)][tt][0][lane] = S.x;
            sh.p[b & (SH::kPBuf - 1)][tt][1][lane] = S.y;
        } else {
            sh.p[b & (SH::kPBuf - 1)][tt][0][lane] = S.x - A.x0[0];
            sh.p[b & (SH::kPBuf - 1)][tt][1][lane] = S.y - A.x0[1];
        }
        if (FULL || t < H) {
            if constexpr (MODE != MODE_COST && !SH::kStage) {
                if (A.store_xy && live) {
                    CCV_STATE_STORE(&A.xs[(size_t)t * pitch + k], S.x);
                    CCV_STATE_STORE(&A.ys[(size_t)t * pitch + k], S.y);
                }
            }
            if (FULL || t < H - 1) {
                double u[UD];
                static_for<UD>([&](auto D) {
                    constexpr int d = decltype(D)::value;
                    constexpr int nloc = tt * UD + d;
                    const int n = t0 * UD + nloc;   // row = step*UD + dim
                    if constexpr (MODE == MODE_FUSED) {
                        if constexpr ((nloc & 3) == 0) {
                            // warm start u*[n .. n+3]: wave-uniform load, in flight while the Philox rounds run
                            nom = *reinterpret_cast<const double4*>(&sh.nom[n]);
                            const Philox4 r = philox4x32_10(kg, (uint32_t)(n >> 2), A.iter_lo, A.iter_hi, A.seed_lo, A.seed_hi);
                            box_muller_f32(r.x, r.y, zq[0], zq[1]);
                            box_muller_f32(r.z, r.w, zq[2], zq[3]);
                        }
                        constexpr int q = nloc & 3;
                        const double mean = q == 0 ? nom.x : q == 1 ? nom.y : q == 2 ? nom.z : nom.w;
                        // libstdc++ normal_distribution: ret * stddev + mean (dd:96-97), then clamp (dd:98-99)
                        double v = (double)zq[q] * A.sigma + mean;
                        v = clampd(v, arg5<d>(A.umin), arg5<d>(A.umax));
                        if constexpr (FB && d == 2) {
                            if (A.steer_off) v = 0.0;   // fb:517
                        }
                        u[d] = v;
                        if constexpr (SH::kStage) {
                            sh.zs[b & 1][nloc][lane] = zq[q];
                        } else {
                            if (live) A.z[(size_t)n * pitch + k] = zq[q];
                        }
                    } else {
                        u[d] = A.u[(size_t)n * pitch + kk];
                    }
                });
                // ---- cost terms that do not need the window ----
                if constexpr (COST) {
                    if constexpr (!FB) {
                        const double dv = u[0] - A.v_ref;
                        cost = fma(A.w_v * dv, dv, cost);   // dd:204-206 (the arithmetic of pc_produce_batched)
                    } else {
                        if (t < H - 2) {                                          // fb:409
                            cost += A.w_v * (u[0] - A.v_ref) * (u[0] - A.v_ref);  // fb:413
                            if (u[0] < 0.0) cost += A.w_back * u[0] * u[0];       // fb:420
                        }
                        if (t >= 1) {   // finish index t-1 <= H-3: ZMP (fb:468-485, 597-603) and roll-rate terms
                            const double mgz = A.fb_mass * A.fb_gz;                               // (mass*gravity_).z
                            const double drive_accel = div_uniform(u[0] - S.p_v, dt, A.inv_dt);   // fb:469
                            const double ay = drive_accel * S.p_sdir + S.p_ac * S.p_cdir;         // fb:473
                            const double hgdot_x = div_uniform(A.fb_Ixx * u[3] - A.fb_Ixx * S.p_rv, dt, A.inv_dt);   // fb:479-481
                            const double mo_x = (S.p_c2 * mgz + S.p_c3 * (A.fb_mass * ay)) - hgdot_x;  // fb:600
                            const double zmp_y = mo_x / mgz;                                      // fb:601 (accel.z == 0)
                            cost += A.w_zmp * zmp_y * zmp_y;                                      // fb:416
                            cost += A.w_rollv * (u[3] - S.p_rv) * (u[3] - S.p_rv);                // fb:418
                        }
                    }
                }
                // ---- dynamics: explicit Euler (dd:104-109, sd:120-125, fb:445-452) ----
                double hd = S.yaw;
                if constexpr (MODEL != CCV_MPPI_DIFF_DRIVE) hd = S.yaw + u[2];
                double sn, cs;
                fast_sincos(hd, sn, cs);
                if constexpr (FB && COST) {
                    double sd_, cd_, sr_, cr_, sp_, cp_;
                    fast_sincos(u[2], sd_, cd_);
                    fast_sincos(S.roll, sr_, cr_);
                    fast_sincos(S.pitch, sp_, cp_);
                    S.p_sdir = sd_;
                    S.p_cdir = cd_;
                    S.p_c2 = -A.fb_L * sr_;                    // CoM.y (fb:482)
                    S.p_c3 = A.fb_L * cp_ * cr_;               // CoM.z
                    S.p_ac = u[0] * u[1];                      // fb:471
                    S.p_v = u[0];
                    S.p_rv = u[3];
                }
                if constexpr (FB) {
                    S.x = S.x + u[0] * cs * dt;
                    S.y = S.y + u[0] * sn * dt;
                } else {   // (the arithmetic of pc_produce_batched)
                    const double step = u[0] * dt;
                    S.x = fma(step, cs, S.x);
                    S.y = fma(step, sn, S.y);
                }
                S.yaw = S.yaw + u[1] * dt;
                if constexpr (FB) {
                    S.roll = S.roll + u[3] * dt;
                    S.pitch = S.pitch + u[4] * dt;
                }
            } else {
                if constexpr (!FB && COST) {
                    // t == H-1: the reference reads control index H-1, one past the end (dd:199,204): defined as 0.0 (Q1)
                    const double dv = 0.0 - A.v_ref;
                    cost = fma(A.w_v * dv, dv, cost);
                }
            }
        }
    });
}

// The 4*CN normals of Philox calls C0 .. C0+CN-1 of time block b (call c holds normals 4c .. 4c+3 of the block): CN Philox
// blocks with their rounds interleaved, then the 2*CN Box-Muller pairs stage by stage.
template <int MODEL, int C0, int CN>
__device__ __forceinline__ void pc_block_normals(const RolloutArgs& A, const int b, const uint32_t kg, float (&z)[4 * CN]) {
    constexpr int UD = udim_of(MODEL);
    uint32_t c0[CN], c1[CN], c2[CN], c3[CN];
#pragma unroll
    for (int i = 0; i < CN; ++i) {
        c0[i] = kg;
        c1[i] = (uint32_t)((b * kTU * UD) >> 2) + (uint32_t)(C0 + i);
        c2[i] = A.iter_lo;
        c3[i] = A.iter_hi;
    }
    philox4x32_10_n<CN>(c0, c1, c2, c3, A.seed_lo, A.seed_hi);
    uint32_t ba[2 * CN], bb[2 * CN];
    float z0[2 * CN], z1[2 * CN];
#pragma unroll
    for (int i = 0; i < CN; ++i) {
        ba[2 * i] = c0[i];
        bb[2 * i] = c1[i];
        ba[2 * i + 1] = c2[i];
        bb[2 * i + 1] = c3[i];
    }
    box_muller_f32_n<2 * CN>(ba, bb, z0, z1);
#pragma unroll
    for (int i = 0; i < 2 * CN; ++i) {
        z[2 * i] = z0[i];
        z[2 * i + 1] = z1[i];
    }
}

// Full body, two-wave kernel: the producer (10 Philox calls, 20 Box-Muller pairs, 32 sin/cos per block) takes 1.5x as
// long as the distance phase of its partner, which then idles.  The partner therefore makes the first kPcAheadCalls
// Philox calls' normals of the block IT will produce next and parks them in LDS (its own slot: no synchronisation).
constexpr int kPcAheadCalls = 4;
template <int MODEL>
__device__ __forceinline__ void pc_noise_ahead(const RolloutArgs& A, float (*slot)[kPcSamples], const int b, const int lane,
                                               const uint32_t kg) {
    float z[4 * kPcAheadCalls];
    pc_block_normals<MODEL, 0, kPcAheadCalls>(A, b, kg, z);
#pragma unroll
    for (int i = 0; i < 4 * kPcAheadCalls; ++i) slot[i][lane] = z[i];
}

// ---------------------------------------------------------------------------------------------------------------
// producer, fast path for a block whose 8 steps all carry controls: the same arithmetic as pc_produce, arranged in
// batches so that the independent chains of the 8 steps (Philox rounds, Box-Muller, sin/cos) sit in ONE basic block and
// interleave -- a lone wave then issues back to back instead of waiting out each chain's latency.
// ---------------------------------------------------------------------------------------------------------------
// ZLDS (four-wave kernel): the block's normals are already in sh.zs, made by the noise wave.
// WIDE (diff drive): sin / cos of every heading evaluated in full (fast_sincos_n, as steering does) instead of advanced by
// the step's turn -- for loop periods beyond |w|max dt = pi/4, where the short polynomials of the rotation form are not
// valid (the node measures dt, dd:346-348: one slow tick must not cost a different, twice as slow kernel).
// PARTIAL: the horizon's last block when fewer than 8 of its steps carry controls (nctl = H - 1 - 8 b of them, 1 .. 7; the
// block's states are t0 .. t0 + nctl, the last one t = H - 1).  The whole batch is computed as for a full block -- the normals
// past row (H-1) u_dim are drawn and dropped, the warm start reads as 0 there -- and only what leaves the block is masked:
// stores, cost terms (dd / sd: the Q1 phantom term at t = H - 1), the step-range conditions of the full-body terms.  The
// reference defaults run H = 15: six of fourteen control steps used to go through the step-by-step path (pc_produce), whose
// chains (Philox -> Box-Muller -> sin/cos -> position, one step after the other) a lone wave waits out one by one.
// (worth it from kPartialMin steps on: a block of one or two steps is quicker step by step than as a batch of eight -- C2's
//  H - 1 = 49 leaves one step, and batching it cost the four-wave kernel 1.6 us)
constexpr int kPartialMin = 4;
// NCTL > 0 (with PARTIAL): the number of control steps is known at compile time -- every mask folds, and what the masked steps
// would have computed falls away as dead code (the reference's default horizon H = 15 leaves a last block of six: a quarter of
// the batch).  The steps that remain are computed as before, bit for bit.
template <int MODEL, int MODE, class SH, bool ZLDS = false, bool FASTCLAMP = false, bool WIDE = false, bool PARTIAL = false,
          int NCTL = 0>
__device__ __forceinline__ bool pc_produce_batched(const RolloutArgs& A, SH& sh, PcState<MODEL>& S, double& cost,
                                                   const int b, const int lane, const int k, const int kk, const bool live,
                                                   const uint32_t kg,
                                                   const float (*ahead)[kPcSamples] = nullptr,   // pc_noise_ahead's slot
                                                   const int nctl_in = kTU) {
    static_assert(NCTL == 0 || (PARTIAL && NCTL < kTU), "a compile-time step count is a partial block's");
    const int nctl = NCTL > 0 ? NCTL : (PARTIAL ? nctl_in : kTU);   // steps of this block that carry controls (wave-uniform)
    constexpr int UD = udim_of(MODEL);
    constexpr bool FB = MODEL == CCV_MPPI_FULL_BODY;
    constexpr bool COST = MODE != MODE_ROLLOUT;
    constexpr int NCALL = kTU * UD / 4;
    const int H = A.H;
    const int t0 = b * kTU;
    const size_t pitch = (size_t)A.pitch;
    const double dt = A.dt;
    // ---- 1. controls of the 8 steps
    double u[kTU][UD];
    if constexpr (MODE == MODE_FUSED) {
        // Philox calls in groups (rounds interleaved within a group, then the group's Box-Muller pairs stage by stage):
        // 4 | 3+3 | 5+5, or for full body 4 (possibly made ahead by the partner wave) + 3 + 3
        auto group = [&](auto C0_, auto CN_, const bool from_lds) {
            constexpr int C0 = decltype(C0_)::value, CN = decltype(CN_)::value;
            float z[4 * CN];
            double nomv[4 * CN];   // warm start u* of this group from LDS (broadcast reads), in flight under the Philox rounds
#pragma unroll
            for (int i = 0; i < 4 * CN; ++i) nomv[i] = sh.nom[t0 * UD + 4 * C0 + i];
            CCV_KEEP_ORDER();
            if constexpr (ZLDS) {
#pragma unroll
                for (int i = 0; i < 4 * CN; ++i) z[i] = sh.zs[b & 1][4 * C0 + i][lane];
            } else if (from_lds) {
#pragma unroll
                for (int i = 0; i < 4 * CN; ++i) z[i] = ahead[4 * C0 + i][lane];
            } else {
                pc_block_normals<MODEL, C0, CN>(A, b, kg, z);
            }
            static_for<4 * CN>([&](auto II) {
                constexpr int i = decltype(II)::value;
                constexpr int nloc = 4 * C0 + i;
                constexpr int tt = nloc / UD, d = nloc % UD;
                // libstdc++ normal_distribution: ret * stddev + mean (dd:96-97), then clamp (dd:98-99)
                double v = (double)z[i] * A.sigma + nomv[i];
                v = clampd_as<FASTCLAMP>(v, arg5<d>(A.umin), arg5<d>(A.umax));
                if constexpr (FB && d == 2) {
                    if (A.steer_off) v = 0.0;   // fb:517
                }
                u[tt][d] = v;
                if constexpr (SH::kStage) {
                    if constexpr (!ZLDS) sh.zs[b & 1][nloc][lane] = z[i];   // (ZLDS: the noise wave has put it there)
                } else {
                    // no `live` predicate: rows are padded to a multiple of 64 samples (pitch), lanes past K write their
                    // padding slot -- a branch per store would cut this block into pieces the scheduler cannot interleave
                    if (!PARTIAL || tt < nctl) A.z[(size_t)(t0 * UD + nloc) * pitch + k] = z[i];
                }
            });
        };
        using std::integral_constant;
        if constexpr (NCALL == 4) {
            group(integral_constant<int, 0>{}, integral_constant<int, 4>{}, false);
        } else if constexpr (NCALL == 6) {
            group(integral_constant<int, 0>{}, integral_constant<int, 3>{}, false);
            group(integral_constant<int, 3>{}, integral_constant<int, 3>{}, false);
        } else {
            static_assert(NCALL == 10 && kPcAheadCalls == 4, "full body: 4 + 3 + 3 Philox calls");
            group(integral_constant<int, 0>{}, integral_constant<int, 4>{}, ahead != nullptr);
            group(integral_constant<int, 4>{}, integral_constant<int, 3>{}, false);
            group(integral_constant<int, 7>{}, integral_constant<int, 3>{}, false);
        }
    } else {
#pragma unroll
        for (int tt = 0; tt < kTU; ++tt)
#pragma unroll
            for (int d = 0; d < UD; ++d) u[tt][d] = (!PARTIAL || tt < nctl) ? A.u[(size_t)((t0 + tt) * UD + d) * pitch + kk] : 0.0;
    }
    // ---- 2. heading (roll, pitch) recurrences: yaw[t+1] = yaw[t] + w[t]*dt (dd:108, fb:449-451)
    double yawv[kTU + 1], rollv[FB ? kTU + 1 : 1], pitchv[FB ? kTU + 1 : 1];
    yawv[0] = S.yaw;
    if constexpr (FB) {
        rollv[0] = S.roll;
        pitchv[0] = S.pitch;
    }
#pragma unroll
    for (int tt = 0; tt < kTU; ++tt) {
        yawv[tt + 1] = yawv[tt] + u[tt][1] * dt;
        if constexpr (FB) {
            rollv[tt + 1] = rollv[tt] + u[tt][3] * dt;
            pitchv[tt + 1] = pitchv[tt] + u[tt][4] * dt;
        }
    }
    // (the host only selects this kernel when every reachable angle is inside the branch-free sin/cos range:
    //  ccv_mppi_capi.hip fast_trig_safe())
    double hd[kTU];
#pragma unroll
    for (int tt = 0; tt < kTU; ++tt) {
        hd[tt] = yawv[tt];
        if constexpr (MODEL != CCV_MPPI_DIFF_DRIVE) hd[tt] = yawv[tt] + u[tt][2];
    }
    // ---- 3. sin/cos of the 8 headings
    double sn[kTU], cs[kTU];
    double fb_sd[FB ? kTU : 1], fb_cd[FB ? kTU : 1], fb_sr[FB ? kTU : 1], fb_cr[FB ? kTU : 1], fb_cp[FB ? kTU : 1];   // full body: direction, roll, pitch
    if constexpr (MODEL == CCV_MPPI_DIFF_DRIVE && !WIDE) {
        // diff drive: the heading only ever changes by the step's turn w*dt, so its (sin, cos) are ADVANCED by that angle --
        // eight short independent polynomial pairs (no range reduction, no quadrant logic: the host admits this kernel
        // only for |w| dt <= pi/4) and a chain of eight 2x2 rotations, ~25 fp64 operations per step instead of ~40.
        // Rounding differs from sin(yaw) / cos(yaw) of the accumulated yaw by a few ulp per step.
        double turn[kTU], sd[kTU], cd[kTU];
#pragma unroll
        for (int tt = 0; tt < kTU; ++tt) turn[tt] = u[tt][1] * dt;
        kernel_sincos_n<kTU>(turn, sd, cd);
        double s_ = S.sn, c_ = S.cs;
#pragma unroll
        for (int tt = 0; tt < kTU; ++tt) {
            sn[tt] = s_;
            cs[tt] = c_;
            const double s2 = fma(s_, cd[tt], c_ * sd[tt]);
            const double c2 = fma(c_, cd[tt], -(s_ * sd[tt]));
            s_ = s2;
            c_ = c2;
        }
        if constexpr (NCTL == 0) {
            S.sn = s_;
            S.cs = c_;
        }
    } else if constexpr (FB) {
        // full body needs sin / cos of four angles per step: heading = yaw + direction, direction, roll, pitch.  Every one
        // of them either is small (|direction| <= pi/4: a clamped control) or changes by a small step (yaw, roll, pitch:
        // rate * dt; the host admits this kernel only for |rate| dt <= pi/4, fast_trig_safe()), so per block one full
        // evaluation of sin / cos(yaw), (roll), (pitch) at the block's first step is advanced by rotations, and the short
        // polynomials (no range reduction, no quadrant logic) do the rest: 32 batched short evaluations + 3 full ones per
        // block instead of 32 full ones.  sin / cos(yaw + direction) comes from the addition theorem.  Rounding differs
        // from the direct evaluation by a few ulp per step; the chains restart from the accumulated angles every block.
        double turn[kTU], st_[kTU], ct_[kTU], dir[kTU];
#pragma unroll
        for (int tt = 0; tt < kTU; ++tt) {
            turn[tt] = u[tt][1] * dt;
            dir[tt] = u[tt][2];
        }
        kernel_sincos_n<kTU>(turn, st_, ct_);
        kernel_sincos_n<kTU>(dir, fb_sd, fb_cd);
        double sy, cy;
        fast_sincos(yawv[0], sy, cy);
#pragma unroll
        for (int tt = 0; tt < kTU; ++tt) {
            sn[tt] = fma(sy, fb_cd[tt], cy * fb_sd[tt]);
            cs[tt] = fma(cy, fb_cd[tt], -(sy * fb_sd[tt]));
            const double s2 = fma(sy, ct_[tt], cy * st_[tt]);
            const double c2 = fma(cy, ct_[tt], -(sy * st_[tt]));
            sy = s2;
            cy = c2;
        }
        if constexpr (COST) {
            double rinc[kTU], pinc[kTU], sri[kTU], cri[kTU], spi[kTU], cpi[kTU];
#pragma unroll
            for (int tt = 0; tt < kTU; ++tt) {
                rinc[tt] = u[tt][3] * dt;
                pinc[tt] = u[tt][4] * dt;
            }
            kernel_sincos_n<kTU>(rinc, sri, cri);
            kernel_sincos_n<kTU>(pinc, spi, cpi);
            double sr, cr, sp, cp;
            fast_sincos(rollv[0], sr, cr);
            fast_sincos(pitchv[0], sp, cp);
#pragma unroll
            for (int tt = 0; tt < kTU; ++tt) {
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
        }
    } else {
        // independent chains, evaluated stage by stage in two groups of four
        constexpr int SG = 4;
#pragma unroll
        for (int g = 0; g < kTU / SG; ++g) {
            double xin[SG], so[SG], co[SG];
#pragma unroll
            for (int i = 0; i < SG; ++i) xin[i] = hd[g * SG + i];
            fast_sincos_n<SG>(xin, so, co);
#pragma unroll
            for (int i = 0; i < SG; ++i) {
                sn[g * SG + i] = so[i];
                cs[g * SG + i] = co[i];
            }
        }
    }
    // ---- 4. cost terms that do not need the window
    if constexpr (COST) {
        if constexpr (!FB) {
#pragma unroll
            for (int tt = 0; tt < kTU; ++tt) {
                // dd:204-206: v_weight (v - v_ref)^2, the last factor riding on the addition
                if constexpr (PARTIAL) {
                    // t == H-1: the reference reads control index H-1, one past the end (dd:199,204): defined as 0.0 (Q1)
                    const double dv = (tt < nctl ? u[tt][0] : 0.0) - A.v_ref;
                    cost = tt <= nctl ? fma(A.w_v * dv, dv, cost) : cost;
                } else {
                    const double dv = u[tt][0] - A.v_ref;
                    cost = fma(A.w_v * dv, dv, cost);
                }
            }
        } else {
            const double mgz = A.fb_mass * A.fb_gz;   // (mass*gravity_).z
#pragma unroll
            for (int tt = 0; tt < kTU; ++tt) {
                const int t = t0 + tt;
                // the two step-range conditions are wave-uniform: selects, not branches, keep the block in one piece
                {                                                                     // fb:409: t < H - 2
                    const double cv = A.w_v * (u[tt][0] - A.v_ref) * (u[tt][0] - A.v_ref);       // fb:413
                    const double cb = A.w_back * u[tt][0] * u[tt][0];                            // fb:420
                    const bool in = t < H - 2;
                    cost += in ? cv : 0.0;
                    cost += (in && u[tt][0] < 0.0) ? cb : 0.0;
                }
                {   // t >= 1: finish index t-1: ZMP (fb:468-485, 597-603) and roll-rate terms
                    const double drive_accel = div_uniform(u[tt][0] - S.p_v, dt, A.inv_dt);      // fb:469
                    const double ay = drive_accel * S.p_sdir + S.p_ac * S.p_cdir;                // fb:473
                    const double hgdot_x = div_uniform(A.fb_Ixx * u[tt][3] - A.fb_Ixx * S.p_rv, dt, A.inv_dt);   // fb:479-481
                    const double mo_x = (S.p_c2 * mgz + S.p_c3 * (A.fb_mass * ay)) - hgdot_x;    // fb:600
                    const double zmp_y = mo_x / mgz;                                             // fb:601
                    const double cz = A.w_zmp * zmp_y * zmp_y;                                   // fb:416
                    const double cr = A.w_rollv * (u[tt][3] - S.p_rv) * (u[tt][3] - S.p_rv);     // fb:418
                    const bool in = t >= 1 && (!PARTIAL || tt < nctl);   // (index t-1 <= H-3)
                    cost += in ? cz : 0.0;
                    cost += in ? cr : 0.0;
                }
                if (NCTL == 0 || tt < NCTL) {   // (a compile-time count: nothing reads the state past the last control step)
                    S.p_sdir = fb_sd[tt];
                    S.p_cdir = fb_cd[tt];
                    S.p_c2 = -A.fb_L * fb_sr[tt];                 // CoM.y (fb:482)
                    S.p_c3 = A.fb_L * fb_cp[tt] * fb_cr[tt];      // CoM.z
                    S.p_ac = u[tt][0] * u[tt][1];       // fb:471
                    S.p_v = u[tt][0];
                    S.p_rv = u[tt][3];
                }
            }
        }
    }
    // ---- 5. positions (dd:106-107), hand-off of (x,y) - pose to the consumer, x,y -> HBM
    double x = S.x, y = S.y;
    double xv[kTU], yv[kTU];
#pragma unroll
    for (int tt = 0; tt < kTU; ++tt) {
        xv[tt] = x;
        yv[tt] = y;
        if (NCTL == 0 || tt <= NCTL) {   // (a compile-time count: the states past t = H - 1 are nobody's)
            if constexpr (SH::kStage) {
                sh.p[b & (SH::kPBuf - 1)][tt][0][lane] = x;
                sh.p[b & (SH::kPBuf - 1)][tt][1][lane] = y;
            } else {
                sh.p[b & (SH::kPBuf - 1)][tt][0][lane] = x - A.x0[0];
                sh.p[b & (SH::kPBuf - 1)][tt][1][lane] = y - A.x0[1];
            }
        }
        if constexpr (FB) {
            // (the full-body kernels keep the reference's form: the one-wave kernel sits at 256 registers, and every
            //  rearrangement of this block tried so far -- fused products here or in the cost terms, weights switched instead
            //  of 64-bit selects -- either spills or costs it 4 - 14 us of C4's 206, profiles/README.md round 3)
            x = x + u[tt][0] * cs[tt] * dt;
            y = y + u[tt][0] * sn[tt] * dt;
        } else {
            const double step = u[tt][0] * dt;   // dd:106-107: x + v cos(yaw) dt -- one product shared, the other fused
            x = fma(step, cs[tt], x);
            y = fma(step, sn[tt], y);
        }
    }
    if constexpr (MODE != MODE_COST && !SH::kStage) {
        if (A.store_xy) {   // one wave-uniform branch for the 16 stores (padded rows: no `live` predicate, as above)
#pragma unroll
            for (int tt = 0; tt < kTU; ++tt) {
                if (!PARTIAL || tt <= nctl) {   // states t <= H-1
                    CCV_STATE_STORE(&A.xs[(size_t)(t0 + tt) * pitch + k], xv[tt]);
                    CCV_STATE_STORE(&A.ys[(size_t)(t0 + tt) * pitch + k], yv[tt]);
                }
            }
        }
    }
    if constexpr (NCTL > 0) return true;   // (the horizon's last block: the state is not carried any further)
    S.x = x;
    S.y = y;
    S.yaw = yawv[kTU];
    if constexpr (FB) {
        S.roll = rollv[kTU];
        S.pitch = pitchv[kTU];
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Exact pruning of the window for one wave and one block of states.  f_j(p) = a_j px + b_j py + c_j (= |p - r_j|^2 - |p|^2)
// is linear in p, so is the difference of two of them: window point j cannot be the nearest one for ANY position p of the
// wave if some other point r DOMINATES it over a set that contains them all,
//     min over the box B of (f_j - f_r)(p)  =  (c_j - c_r) + min((a_j - a_r) x) + min((b_j - b_r) y)  >  0,
// i.e. B lies strictly on r's side of the bisector of r_j and r.  B is the bounding box of the wave's NV x 64 positions;
// the candidates r are the window points nearest to B's four corners.  Lanes take the role of window points; the
// survivors are contiguous along a path, so the distance loop runs over their hull.  On the launch workload 31 % of the
// (state, point) pairs remain (tools/prune_study.py; all-pairs dominance over the same boxes: 30 %, the hull of the true
// nearest points: 24 %), against 56 % for the round-1 test (one bound M = min_j max_B f_j against min_B f_j, which takes
// the two extrema at different corners).
// Nothing here assumes a path-like window: any point excluded is excluded by a valid inequality, the rest costs time only.
// Rounding: the box and the choice of r are made in fp32 (any r is a valid dominator; the box is rounded outwards), the
// inequality itself in fp64 against T = 1e-9 * (sum of the magnitudes that enter it) -- the rounding errors of the test and
// of the loop's own f_j(p), f_r(p) are below 1e-15 of that sum.  Everything unordered (NaN / infinite positions or
// coefficients) fails the comparison and keeps the point; a lane whose own position is not finite ends at the 100 m
// gate value whatever the loop covers.
// ---------------------------------------------------------------------------------------------------------------
template <int NV, int NHALF, class SH>   // NHALF: window points per lane (2 when the window is longer than 64)
__device__ __forceinline__ void pc_prune_window(const RolloutArgs& A, const SH& sh, const double (&px)[NV], const double (&py)[NV],
                                                const int lane, int& jb, int& je) {
    // ---- bounding box of the NV x 64 positions
    float xlo = (float)px[0], ylo = (float)py[0];
    float xhi = xlo, yhi = ylo;
#pragma unroll
    for (int i = 1; i < NV; ++i) {
        const float fx = (float)px[i], fy = (float)py[i];
        xlo = fminf(xlo, fx);
        xhi = fmaxf(xhi, fx);
        ylo = fminf(ylo, fy);
        yhi = fmaxf(yhi, fy);
    }
    wave_min_max_min_max_f32(xlo, xhi, ylo, yhi);
    // outwards: |x - (float)x| <= 2^-24 |x|, or 2^-126 where fp32 goes subnormal / flushes
    const double bx0 = (double)xlo - (fabs((double)xlo) * 1.2e-7 + 1.0e-37), bx1 = (double)xhi + (fabs((double)xhi) * 1.2e-7 + 1.0e-37);
    const double by0 = (double)ylo - (fabs((double)ylo) * 1.2e-7 + 1.0e-37), by1 = (double)yhi + (fabs((double)yhi) * 1.2e-7 + 1.0e-37);
    const double X = fmax(fabs(bx0), fabs(bx1)), Y = fmax(fabs(by0), fabs(by1));
    // ---- this lane's window points: j = lane, and lane + 64 when the window is longer than 64
    double2 ab[NHALF];
    double c[NHALF], tj[NHALF];
    float v[4][NHALF];
    bool valid[NHALF];
#pragma unroll
    for (int hh = 0; hh < NHALF; ++hh) {
        const int j = lane + 64 * hh;
        valid[hh] = j < A.H;
        ab[hh] = valid[hh] ? sh.ab[j] : make_double2(0.0, 0.0);
        c[hh] = valid[hh] ? sh.c[j] : INFINITY;
        tj[hh] = 1.0e-9 * (fabs(c[hh]) + (fabs(ab[hh].x) * X + fabs(ab[hh].y) * Y));
        // f_j at the four corners, fp32: only to choose the dominators
        const float a32 = (float)ab[hh].x, b32 = (float)ab[hh].y, c32 = (float)c[hh];
        v[0][hh] = __builtin_fmaf(a32, xlo, __builtin_fmaf(b32, ylo, c32));
        v[1][hh] = __builtin_fmaf(a32, xlo, __builtin_fmaf(b32, yhi, c32));
        v[2][hh] = __builtin_fmaf(a32, xhi, __builtin_fmaf(b32, ylo, c32));
        v[3][hh] = __builtin_fmaf(a32, xhi, __builtin_fmaf(b32, yhi, c32));
    }
    float best[4], mine[4];
    unsigned long long upper[4];   // lanes whose better candidate is lane + 64
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        upper[k] = 0ull;
        mine[k] = v[k][0];
        if constexpr (NHALF == 2) {
            const bool up = v[k][1] < v[k][0];
            upper[k] = __ballot(up);
            mine[k] = up ? v[k][1] : v[k][0];
        }
        best[k] = mine[k];
    }
    wave_min4_f32(best[0], best[1], best[2], best[3]);
    int jr[4];
    bool ok[4];
    double2 abr[4];
    double cr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long at = __ballot(mine[k] == best[k]);   // (empty when every value is NaN)
        ok[k] = at != 0ull;
        const int r = ok[k] ? __builtin_ctzll(at) : 0;
        jr[k] = r + (int)((upper[k] >> r) & 1ull) * 64;
        abr[k] = sh.ab[jr[k]];   // wave-uniform address: broadcast reads, all four in flight together
        cr[k] = sh.c[jr[k]];
    }
    // ---- dominance of this lane's points by each of the four
    unsigned long long keep[2] = {0ull, 0ull};
#pragma unroll
    for (int hh = 0; hh < NHALF; ++hh) {
        bool kp = valid[hh];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double da = ab[hh].x - abr[k].x, db = ab[hh].y - abr[k].y, dc = c[hh] - cr[k];
            const double L = dc + (fmin(da * bx0, da * bx1) + fmin(db * by0, db * by1));
            const double T = tj[hh] + 1.0e-9 * (fabs(cr[k]) + (fabs(abr[k].x) * X + fabs(abr[k].y) * Y));
            kp = kp && !(ok[k] && L > T);
        }
        keep[hh] = __ballot(kp);
    }
    if (keep[0] | keep[1]) {
        const int lo = keep[0] ? __builtin_ctzll(keep[0]) : 64 + __builtin_ctzll(keep[1]);
        const int hi = keep[1] ? 127 - __builtin_clzll(keep[1]) : 63 - __builtin_clzll(keep[0]);
        jb = lo & ~3;
        je = (hi | 3) + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// consumer: min over the H window points of (a_j px + b_j py + c_j) for the NV states of one block, then the path cost.
// Straight fp64 FMA/MIN; four window points per iteration so the LDS broadcast reads are covered and the compiler's
// canonicalising max in front of fmin() is paid once per four minima.
// ---------------------------------------------------------------------------------------------------------------
// LEAN (four-wave kernel, 128 VGPRs): two window points per register set instead of four and the running minimum taken
// point pair by point pair -- 48 registers less; the same minima, hence the same bits.
template <int NV, int MODEL, class SH, bool LEAN = false>
__device__ __forceinline__ void pc_consume(const RolloutArgs& A, const SH& sh, double& cost, const int b, const int lane,
                                           const int i0 = 0,          // states i0 .. i0+NV-1 of block b
                                           int* prune_on = nullptr,   // wave-uniform switch of the window pruning (below)
                                           int* taken_flag = nullptr, const int taken_value = 0) {   // see below
    const int H4 = (A.H + 3) & ~3;   // the window is padded with c = +inf: four points per iteration, no remainder
    double px[NV], py[NV], m[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        px[i] = sh.p[b & (SH::kPBuf - 1)][i0 + i][0][lane];
        py[i] = sh.p[b & (SH::kPBuf - 1)][i0 + i][1][lane];
        if constexpr (SH::kStage) {   // staged positions are absolute
            px[i] -= A.x0[0];
            py[i] -= A.x0[1];
        }
        m[i] = INFINITY;
    }
    if (taken_flag) {
        // three-wave kernel: the positions of the block are in registers -- tell the producer that the LDS buffer is free
        // (mppi_rollout_r3.h).  The wait makes sure the loads above have returned before the number is written.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pc_publish(taken_flag, taken_value);
    }
    // software pipeline: the coefficients of points j+4..j+7 are read from LDS (broadcast reads) before the ~100 fp64
    // instructions on points j..j+3 issue, so no iteration waits out the LDS latency.  sh.ab / sh.c carry 4 spare
    // entries past H4, so the last iteration's read-ahead stays inside the arrays.
    double2 ab0[4], ab1[4];
    double c0[4], c1[4];
    auto fetch = [&](double2(&ab)[4], double(&c)[4], const int j) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            ab[jj] = sh.ab[j + jj];
            c[jj] = sh.c[j + jj];
        }
    };
    auto points4 = [&](const double2(&ab)[4], const double(&c)[4]) {
        double f[4][NV];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < NV; ++i) f[jj][i] = fma(ab[jj].x, px[i], fma(ab[jj].y, py[i], c[jj]));
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            // the three minima over freshly computed values need no canonicalising v_max; the loop-carried one would get
            // one per iteration from the compiler (it cannot see through the phi that m is already quiet), so it is
            // issued directly: v_min_f64 of two quiet operands
            const double t = fmin(fmin(f[0][i], f[1][i]), fmin(f[2][i], f[3][i]));
            asm("v_min_f64 %0, %1, %2" : "=v"(m[i]) : "v"(m[i]), "v"(t));
        }
    };
    // ---- exact pruning of the window (pc_prune_window below): the loop runs over the hull [jb, je) of the window points
    // that can be the nearest one for some sample of this wave
    int jb = 0, je = H4;
    if (A.prune && (prune_on == nullptr || *prune_on)) {
        if (H4 > 64) pc_prune_window<NV, 2>(A, sh, px, py, lane, jb, je);
        else pc_prune_window<NV, 1>(A, sh, px, py, lane, jb, je);
        // the samples of a wave fan out with time: once a block keeps more than 3/4 of the window the later ones will too,
        // and the wave stops testing for the rest of the launch
        if (prune_on && 4 * (je - jb) > 3 * H4) *prune_on = 0;
    }
    if constexpr (LEAN) {
        double2 qa[2], qb[2];
        double ca[2], cb[2];
        auto fetch2 = [&](double2(&ab)[2], double(&c)[2], const int j) {
            ab[0] = sh.ab[j];
            c[0] = sh.c[j];
            ab[1] = sh.ab[j + 1];
            c[1] = sh.c[j + 1];
        };
        auto points2 = [&](const double2(&ab)[2], const double(&c)[2]) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const double f0 = fma(ab[0].x, px[i], fma(ab[0].y, py[i], c[0]));
                const double f1 = fma(ab[1].x, px[i], fma(ab[1].y, py[i], c[1]));
                const double t = fmin(f0, f1);
                asm("v_min_f64 %0, %1, %2" : "=v"(m[i]) : "v"(m[i]), "v"(t));
            }
        };
        // the first pair of points starts the minimum (no +inf to initialise, no minimum against it: the hull is never empty,
        // je - jb is a multiple of 4, and min(+inf, t) = t for every t the loop can produce but NaN, which ends at the gate
        // value either way); the read-ahead ends inside the arrays' padding
        auto first2 = [&](const double2(&ab)[2], const double(&c)[2]) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const double f0 = fma(ab[0].x, px[i], fma(ab[0].y, py[i], c[0]));
                const double f1 = fma(ab[1].x, px[i], fma(ab[1].y, py[i], c[1]));
                m[i] = fmin(f0, f1);
            }
        };
        fetch2(qa, ca, jb);
        fetch2(qb, cb, jb + 2);
        __builtin_amdgcn_sched_barrier(0);
        first2(qa, ca);
        fetch2(qa, ca, jb + 4);
        __builtin_amdgcn_sched_barrier(0);
        points2(qb, cb);
        for (int j = jb + 4; j < je; j += 4) {
            fetch2(qb, cb, j + 2);
            __builtin_amdgcn_sched_barrier(0);
            points2(qa, ca);
            fetch2(qa, ca, j + 4);
            __builtin_amdgcn_sched_barrier(0);
            points2(qb, cb);
        }
    } else {
    fetch(ab0, c0, jb);
    for (int j = jb; j < je; j += 8) {   // two register sets in ping-pong: no copies
        fetch(ab1, c1, j + 4);
        __builtin_amdgcn_sched_barrier(0);   // keep the read-ahead above the arithmetic (the scheduler would sink it)
        points4(ab0, c0);
        if (j + 4 >= je) break;
        fetch(ab0, c0, j + 8);
        __builtin_amdgcn_sched_barrier(0);
        points4(ab1, c1);
    }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        // d^2 = |p|^2 + min_j(...), gate d <= 100 (dd:185), cost += path_weight*d*d (dd:206)
        double d2 = m[i] + fma(px[i], px[i], py[i] * py[i]);
        if constexpr (MODEL == CCV_MPPI_FULL_BODY) {   // (kept as it was: the one-wave full-body kernel is at its register limit)
            d2 = d2 < 1.0e4 ? fmax(d2, 0.0) : 1.0e4;   // NaN -> gate value, as `distance < min_distance` (dd:189) is false for NaN
            cost += A.w_path * d2;
        } else {
            // the same value in two instructions instead of four (compare, max, two selects): min(d2, 1e4) first -- of a NaN
            // and a number v_min_f64 returns the number (d2 is the result of arithmetic: never a signalling NaN), which is
            // the gate value -- then max(., 0); and the weight rides on the addition
            double g;
            asm("v_min_f64 %0, %1, %2" : "=v"(g) : "v"(d2), "s"(1.0e4));
            asm("v_max_f64 %0, %1, 0" : "=v"(d2) : "v"(g));
            cost = fma(A.w_path, d2, cost);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused first half of determine_OptimalSolution() (dd:228-237): this workgroup's share of sum_i w_i and
// sum_i w_i * u_i[t][d] for every control row, so that the K x (H-1) x u_dim controls are not streamed from HBM a second
// time by a separate kernel -- they are re-read here, by the CU that wrote them (L2 / Infinity Cache hits).
// Rows are dealt to the two waves; a wave reduces 15 rows at a time through LDS: every lane drops w*u for each row,
// then lane (r, q) adds 16 of the 64 entries of row r and two shuffles finish the row.  Fixed order => reproducible.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kUpdRB = 15;                  // rows per LDS batch: 15 * 66 doubles fit one wave's half of sh.p
constexpr int kUpdCH = 4 * kUpdRB;          // rows whose loads are in flight together (120 VGPRs)

// Control rows dealt to the waves of a workgroup in units of BR rows: wave w of NW owns units w, w+NW, ...
// m = 0 .. count()-1 enumerates a wave's rows.  k_rollout_pc: BR = rows of one time block, NW = 2, so that every wave
// re-reads exactly the rows it stored itself (program order makes them visible; no vector-memory wait at the barriers).
template <int BR_, int NW_>
struct UpdRowsT {
    static constexpr int BR = BR_, NW = NW_;
    int R, wv;
    __device__ __forceinline__ int count() const {
        int n = 0;
        for (int r0 = wv * BR; r0 < R; r0 += NW * BR) n += min(BR, R - r0);
        return n;
    }
    __device__ __forceinline__ int row(const int m) const { return ((m / BR) * NW + wv) * BR + m % BR; }
};
template <int MODEL>
using UpdRows = UpdRowsT<kTU * udim_of(MODEL), kPcWaves>;

// The re-read of this workgroup's controls: the loads of a whole chunk of rows are issued back to back, so the chunk
// pays one memory latency, not one per batch.  Rows past the end are clamped (loaded, never used).
// T = float: the fused iteration stored the normals (A.z); T = double: the stage-wise calls hold the controls themselves (A.u)
template <int MODE>
using UpdT = std::conditional_t<MODE == MODE_FUSED, float, double>;
template <class T, class ROWS>
__device__ __forceinline__ void pc_update_fetch(const RolloutArgs& A, T (&v)[kUpdCH], const ROWS& rows, const int m0,
                                                const int mcount, const int kk) {
    const size_t pitch = (size_t)A.pitch;
#pragma unroll
    for (int i = 0; i < kUpdCH; ++i) {
        const size_t at = (size_t)rows.row(min(m0 + i, mcount - 1)) * pitch + kk;
        if constexpr (std::is_same<T, float>::value) v[i] = A.z[at];
        else v[i] = A.u[at];
    }
}

// sum_k w_k * u_k[row] over the 64 samples of the workgroup for this wave's rows -> A.partial[row][workgroup].
// RB rows at a time through the wave-private LDS buffer `buf` (RB * 65 doubles): every lane drops w*u for each row, then
// lane (r, q) adds 16 of the 64 entries of row r and two shuffles finish the row.  (The caller has fetched the first
// chunk into v.)
// LOOP (four-wave kernel): the batches of a chunk are a real loop whose body always works on v[0 .. RB-1]; after a batch the
// register array is shifted down by RB (48 moves).  Unrolled, the epilogue is 2-3 KB of straight-line code per wave that runs
// once, and the instruction cache is cold for every line of it: tools/stamps_r4.py found 1.6 us per batch of 12 rows, the time
// of ~20 line fetches, not of 150 instructions.  (The one- and two-wave kernels go round their unrolled chunk several times.)
template <int RB, int MODEL, bool LOOP = false, class SH, class T, class ROWS>
__device__ __forceinline__ void pc_reduce_rows(const RolloutArgs& A, const SH& sh, double* buf, T (&v)[kUpdCH], const ROWS& rows,
                                               const int mcount, const double wgt, const int lane, const int kk,
                                               const bool fast_clamp = false) {   // (wave-uniform: see clampd_fast)
    static_assert(RB <= 16 && kUpdCH % RB == 0, "batch size");
    // A row is 64 products with one slot of padding after the first 32 (sample k at column k + (k >> 5)) and one at the end:
    // lane (r, q) then reads its 16 entries of row r from banks that no other lane of its half-wave touches (with 65-double
    // rows the lanes q and q + 2 of a row met in the same bank: every ds_read_b64 of the sums took two passes)
    constexpr int STRIDE = kPcSamples + 2;
    const int rr = lane >> 2, q = lane & 3;
    const int col_w = lane + (lane >> 5), col_r = q * 16 + (q >> 1);
    // one batch: products of rows base .. base+RB-1 (held in v[V0 .. V0+RB-1]) -> LDS -> row sums -> partial
    auto batch = [&](auto V0_, const int base) {
        constexpr int V0 = decltype(V0_)::value;
        const int nrows = min(RB, mcount - base);
        // the warm start of the batch's rows first, all reads in flight together: read row by row between the LDS writes
        // below, every row waits out an LDS round trip (the compiler keeps LDS reads behind LDS writes: 12 x lgkmcnt(0) per
        // batch, ~1.6 us per batch with sixteen waves of a CU in their epilogues at once)
        double nomv[RB];
        if constexpr (std::is_same<T, float>::value) {
#pragma unroll
            for (int r = 0; r < RB; ++r) nomv[r] = sh.nom[rows.row(min(base + r, mcount - 1))];
        }
        auto products = [&](auto FAST_) {
            constexpr bool FAST = decltype(FAST_)::value;
            static_for<RB>([&](auto RR) {
                constexpr int r = decltype(RR)::value;
                if constexpr (std::is_same<T, float>::value) {
                    // the row's control dimension: rows are dealt in units of ROWS::BR (a multiple of u_dim), chunks of kUpdCH
                    // (= 60: a multiple of 2, 3 and 5) and batches of RB, so it only depends on the position inside the batch
                    constexpr int UD = udim_of(MODEL);
                    static_assert(ROWS::BR % UD == 0 && kUpdCH % UD == 0 && (!LOOP || RB % UD == 0), "row dealing vs control dimension");
                    constexpr int d = (V0 + r) % UD;
                    // (rows past the end: clamped above, never summed)
                    buf[r * STRIDE + col_w] = wgt * pc_control_from_normal_at<MODEL, d, FAST>(A, v[V0 + r], nomv[r]);
                } else {
                    buf[r * STRIDE + col_w] = wgt * v[V0 + r];
                }
            });
        };
        if (std::is_same<T, float>::value && fast_clamp) products(std::true_type{});
        else products(std::false_type{});
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS writes have landed (wave-private buffer)
        __builtin_amdgcn_wave_barrier();
        double acc = 0.0;
        if (rr < nrows) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += buf[rr * STRIDE + col_r + i];
        }
        acc += dpp_move<kDppXor1>(acc);   // the four lanes of a row are one quad
        acc += dpp_move<kDppXor2>(acc);
        if (rr < nrows && q == 0) A.partial[(size_t)rows.row(base + rr) * A.nparts + blockIdx.x] = acc;
        __builtin_amdgcn_wave_barrier();
    };
    for (int chunk0 = 0; chunk0 < mcount; chunk0 += kUpdCH) {
        if (chunk0 != 0) pc_update_fetch(A, v, rows, chunk0, mcount, kk);
        if constexpr (LOOP) {
#pragma clang loop unroll(disable)
            for (int base = chunk0; base < min(mcount, chunk0 + kUpdCH); base += RB) {
                batch(std::integral_constant<int, 0>{}, base);
#pragma unroll
                for (int i = 0; i + RB < kUpdCH; ++i) v[i] = v[i + RB];
            }
        } else {
            static_for<kUpdCH / RB>([&](auto BB) {
                constexpr int bb = decltype(BB)::value;
                if (chunk0 + bb * RB < mcount) batch(std::integral_constant<int, bb * RB>{}, chunk0 + bb * RB);
            });
        }
    }
}

// sum of weights + cost statistics of the workgroup (one wave)
__device__ __forceinline__ void pc_block_stats(const RolloutArgs& A, const int R, const double wgt, const double total, const bool live,
                                               const int lane) {
    const double sw = wave_sum(wgt);
    const double mn = wave_min(live ? total : INFINITY);
    const double mx = wave_max(live ? total : -INFINITY);
    const double nz = wave_sum((live && wgt == 0.0) ? 1.0 : 0.0);
    if (lane == 0) {
        A.partial[(size_t)R * A.nparts + blockIdx.x] = sw;
        A.statpart[blockIdx.x * 3 + 0] = mn;
        A.statpart[blockIdx.x * 3 + 1] = mx;
        A.statpart[blockIdx.x * 3 + 2] = nz;
    }
}

template <int MODEL, class T>
__device__ __forceinline__ void pc_partial_update(const RolloutArgs& A, PcShared<MODEL>& sh, T (&v)[kUpdCH],
                                                  const UpdRows<MODEL>& rows, const int mcount, const double wgt,
                                                  const double total, const int lane, const int wv, const int kk,
                                                  const bool live) {
    // sh.p: 8 * 2 * 64 = 1024 doubles per wave, free after the last consume (the caller has passed the barrier)
    pc_reduce_rows<kUpdRB, MODEL>(A, sh, &sh.p[wv][0][0][0], v, rows, mcount, wgt, lane, kk);
    if (wv == 0) pc_block_stats(A, (A.H - 1) * udim_of(MODEL), wgt, total, live, lane);
}

template <int MODEL, int MODE>
__global__ __launch_bounds__(kPcWaves * 64, 2) void k_rollout_pc(const RolloutArgs Ak, const Window Wk) {
    constexpr bool FB = MODEL == CCV_MPPI_FULL_BODY;
    constexpr bool COST = MODE != MODE_ROLLOUT;
    __shared__ PcShared<MODEL> sh;
    const RolloutArgs A = with_resident_pose(Ak);
    const int H = A.H;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (COST) stage_window(A, Wk, sh, kPcWaves * 64);
    if constexpr (MODE == MODE_FUSED) pc_stage_nominal<MODEL>(A, sh, kPcWaves * 64);
    const int k = blockIdx.x * kPcSamples + lane;
    const bool live = k < A.K;
    const int kk = live ? k : A.K - 1;
    const uint32_t kg = (uint32_t)(A.k_offset + kk);
    double cost = 0.0;
    if constexpr (FB && COST) {
        if (wv == 0) cost += A.w_yaw * (A.x0[2] - A.yaw_ref0) * (A.x0[2] - A.yaw_ref0);   // fb:408
    }
    const int nblocks = (H + kTU - 1) / kTU;
    const int nstates = FB ? H - 2 : H;   // states that reach the path cost (dd:199 / fb:409)
    int prune_on = 1;
    constexpr bool AHEAD = FB && MODE == MODE_FUSED;   // pc_noise_ahead
    __syncthreads();
    for (int s = 0; s <= nblocks; ++s) {
        pc_rotate_priority(A, s);
        if (s < nblocks && (s & 1) == wv) {
            // ---------------- produce block s
            PcState<MODEL> S;
            if (s == 0) {
                S.x = A.x0[0];
                S.y = A.x0[1];
                S.yaw = A.x0[2];
                S.roll = A.x0[3];
                S.pitch = A.x0[4];
                S.p_v = S.p_rv = S.p_sdir = S.p_c2 = S.p_c3 = S.p_ac = 0.0;
                S.p_cdir = 1.0;
                fast_sincos(A.x0[2], S.sn, S.cs);
            } else {
                const double(*st)[kPcSamples] = sh.st;
                S.x = st[0][lane];
                S.y = st[1][lane];
                S.yaw = st[2][lane];
                if constexpr (MODEL == CCV_MPPI_DIFF_DRIVE) {
                    S.sn = st[3][lane];
                    S.cs = st[4][lane];
                }
                if constexpr (FB) {
                    S.roll = st[3][lane];
                    S.pitch = st[4][lane];
                    S.p_v = st[5][lane];
                    S.p_rv = st[6][lane];
                    S.p_sdir = st[7][lane];
                    S.p_cdir = st[8][lane];
                    S.p_c2 = st[9][lane];
                    S.p_c3 = st[10][lane];
                    S.p_ac = st[11][lane];
                }
            }
            const int nctl = min(kTU, H - 1 - s * kTU);   // steps of this block that carry controls
            const float(*ahead)[kPcSamples] = (AHEAD && s >= 1) ? sh.ahead[wv] : nullptr;
            if (nctl == kTU) pc_produce_batched<MODEL, MODE>(A, sh, S, cost, s, lane, k, kk, live, kg, ahead);
            else if (nctl >= kPartialMin) pc_produce_batched<MODEL, MODE, PcShared<MODEL>, false, false, false, true>(A, sh, S, cost, s, lane, k, kk, live, kg, ahead, nctl);
            else pc_produce<MODEL, MODE, false>(A, sh, S, cost, s, lane, k, kk, live, kg);   // (a short tail, or the final state only)
            double(*st)[kPcSamples] = sh.st;
            st[0][lane] = S.x;
            st[1][lane] = S.y;
            st[2][lane] = S.yaw;
            if constexpr (MODEL == CCV_MPPI_DIFF_DRIVE) {
                st[3][lane] = S.sn;
                st[4][lane] = S.cs;
            }
            if constexpr (FB) {
                st[3][lane] = S.roll;
                st[4][lane] = S.pitch;
                st[5][lane] = S.p_v;
                st[6][lane] = S.p_rv;
                st[7][lane] = S.p_sdir;
                st[8][lane] = S.p_cdir;
                st[9][lane] = S.p_c2;
                st[10][lane] = S.p_c3;
                st[11][lane] = S.p_ac;
            }
        }
        if constexpr (COST) {
            if (s >= 1 && ((s - 1) & 1) == wv) {
                // ---------------- consume block s-1
                const int b = s - 1;
                const int nv = min(kTU, nstates - b * kTU);
                if (nv == kTU) pc_consume<kTU, MODEL>(A, sh, cost, b, lane, 0, &prune_on);
                else if (nv > 0) {
                    switch (nv) {
                        case 7: pc_consume<7, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
                        case 6: pc_consume<6, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
                        case 5: pc_consume<5, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
                        case 4: pc_consume<4, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
                        case 3: pc_consume<3, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
                        case 2: pc_consume<2, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
                        default: pc_consume<1, MODEL>(A, sh, cost, b, lane, 0, &prune_on); break;
                    }
                }
            }
        }
        if constexpr (AHEAD) {
            // the wave that is not producing now produces block s+1 next: part of that block's normals, made in its idle time
            if (((s + 1) & 1) == wv && s + 1 < nblocks && H - 1 - (s + 1) * kTU >= kPartialMin)   // (a block made as a batch, full or partial)
                pc_noise_ahead<MODEL>(A, sh.ahead[wv], s + 1, lane, kg);
        }
        pc_barrier_lds();
    }
    if (A.prio_rotate) __builtin_amdgcn_s_setprio(0);
    if constexpr (COST) {
        UpdT<MODE> upd[kUpdCH];
        const UpdRows<MODEL> rows{(H - 1) * udim_of(MODEL), wv};
        const int mcount = A.fuse_update ? rows.count() : 0;
        // start the re-read of this wave's controls before anything else (see pc_update_fetch)
        if (mcount > 0) pc_update_fetch(A, upd, rows, 0, mcount, kk);
        sh.cost[wv][lane] = cost;
        pc_barrier_lds();   // also: everyone is done with sh.p
        const double total = sh.cost[0][lane] + sh.cost[1][lane];
        const double wgt = live ? exp(-total / A.lambda) : 0.0;   // dd:219 (no min-cost shift)
        if (wv == 0 && live) {
            A.cost[k] = total;
            A.w[k] = wgt;
        }
        if (A.fuse_update) pc_partial_update<MODEL>(A, sh, upd, rows, mcount, wgt, total, lane, wv, kk, live);
    }
}

}  // namespace ccv
