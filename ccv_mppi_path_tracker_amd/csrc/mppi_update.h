// The small kernels around the rollout: weighted update (k_update_partials / k_finalize / k_finalize_exchange /
// k_apply_partials), re-derivation of the controls from the stored normals, MIN_SHIFT re-weighting and the read-back
// helpers.  Not templates: included by ONE translation unit only (ccv_mppi_capi.hip); the rollout kernels' translation
// units (k_*.hip) include mppi_kernels.h alone.
#pragma once
#include "mppi_kernels.h"

namespace ccv {

// u = clamp(double(z) * sigma + u*[n]) for every sample and row, from the normals the fused iteration stored: for the
// stage-wise calls after a fused iteration and for the unfused update (MIN_SHIFT).  Same operations as the samplers.
struct MaterializeArgs {
    const float* z;
    const double* nominal_used;
    double* u;
    double sigma;
    double umin[5], umax[5];
    int32_t K, pitch, R, udim, zero_dim;   // zero_dim: the control dimension that steer_off forces to 0 (fb:517), or -1
};
__global__ __launch_bounds__(kBlock) void k_materialize_controls(const MaterializeArgs A) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    const int n = blockIdx.y;
    if (k >= A.K) return;
    const int d = n % A.udim;
    double v = (double)A.z[(size_t)n * A.pitch + k] * A.sigma + A.nominal_used[n];
    v = clampd(v, A.umin[d], A.umax[d]);
    if (d == A.zero_dim) v = 0.0;
    A.u[(size_t)n * A.pitch + k] = v;
}

// ---- weighted update -------------------------------------------------------------------------------------------
struct UpdateArgs {
    const double* u;
    const double* w;
    const double* cost;
    double* partial;   // [(R+1)][nchunks]
    double* statpart;  // [nchunks][3]: min cost, max cost, zero-weight count
    int32_t K, pitch, R, nchunks;
};

// grid (nchunks, R+1).  Row n < R: sum_k w_k*u[n][k] over this chunk; row R: sum_k w_k (+ cost stats).
// Fixed reduction order => bitwise reproducible (no atomics).
__global__ __launch_bounds__(kBlock) void k_update_partials(const UpdateArgs A) {
    __shared__ double red[4][4];
    const int row = blockIdx.y;
    const int chunk = blockIdx.x;
    const int base = chunk * kChunk;
    const bool wrow = row == A.R;
    const double* urow = A.u + (size_t)row * A.pitch;
    double acc = 0.0, mn = INFINITY, mx = -INFINITY, nz = 0.0;
#pragma unroll
    for (int i = 0; i < kChunk / (2 * kBlock); ++i) {
        const int k = base + i * 2 * kBlock + threadIdx.x * 2;
        if (k + 1 < A.K) {
            const double2 wv = *reinterpret_cast<const double2*>(A.w + k);
            if (wrow) {
                acc += wv.x;
                acc += wv.y;
                const double2 cv = *reinterpret_cast<const double2*>(A.cost + k);
                mn = fmin(mn, fmin(cv.x, cv.y));
                mx = fmax(mx, fmax(cv.x, cv.y));
                nz += (wv.x == 0.0 ? 1.0 : 0.0) + (wv.y == 0.0 ? 1.0 : 0.0);
            } else {
                const double2 uv = *reinterpret_cast<const double2*>(urow + k);
                acc += wv.x * uv.x;
                acc += wv.y * uv.y;
            }
        } else if (k < A.K) {
            const double wv = A.w[k];
            if (wrow) {
                acc += wv;
                const double cv = A.cost[k];
                mn = fmin(mn, cv);
                mx = fmax(mx, cv);
                nz += (wv == 0.0 ? 1.0 : 0.0);
            } else {
                acc += wv * urow[k];
            }
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    acc = wave_sum(acc);
    if (wrow) {
        mn = wave_min(mn);
        mx = wave_max(mx);
        nz = wave_sum(nz);
    }
    if (lane == 0) {
        red[0][wid] = acc;
        red[1][wid] = mn;
        red[2][wid] = mx;
        red[3][wid] = nz;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        A.partial[(size_t)row * A.nchunks + chunk] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        if (wrow) {
            A.statpart[chunk * 3 + 0] = fmin(fmin(red[1][0], red[1][1]), fmin(red[1][2], red[1][3]));
            A.statpart[chunk * 3 + 1] = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
            A.statpart[chunk * 3 + 2] = (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]);
        }
    }
}

struct FinalizeArgs {
    const double* partial;   // [(R+1)][nchunks]
    const double* statpart;  // [nchunks][3]
    double* nominal;         // [R]      (mode 0)
    double* vec;             // [1+R]    unnormalised [sum w, sum w*u] (always written)
    double* stats;           // [4]      sum_w, min cost, max cost, zero-weight count
    int32_t R, nchunks, normalise;
    // Blocking calls (ccv_mppi_iterate, ccv_mppi_update): the result also goes straight into a mailbox in pinned host
    // memory, slot n = u*[n] for n < R, slots R .. R+3 = the four statistics.  A value travels as two self-validating 8-byte
    // packets {32 data bits, 32-bit sequence number}, each one atomic store (the exchange's packet format, below): the host
    // polls until every packet carries this call's number -- no copy engine, no stream synchronisation, no fence.  Null: off.
    unsigned long long* mail;
    uint32_t mail_seq;       // never 0
};

__device__ __forceinline__ void mail_post(const FinalizeArgs& A, const int slot, const double value) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(value);
    unsigned long long* dst = A.mail + 2 * (size_t)slot;
    __hip_atomic_store(dst + 0, (bits & 0xFFFFFFFF00000000ull) | A.mail_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 1, (bits << 32) | A.mail_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One wave per row n: lanes read the chunk partials of the row (fixed order => bitwise reproducible), wave-reduce them,
// and re-derive S = sum w the same way, so no cross-block hand-off is needed.  u*[n] = V_n / S
// (== sum_i (w_i/S) u_i of dd:222,234 up to rounding; S == 0 gives NaN exactly as dd:222 does).
// sums of two rows of n partials each (lane l takes columns l, l+64, ...): up to 1024 columns per pass, all 32 loads of a
// lane issued before the first add (one memory latency for both rows, not one per row)
__device__ __forceinline__ void lane_partial_sum2(const double* row_a, const double* row_b, int n, int lane, double& sum_a,
                                                  double& sum_b) {
    double acc_a = 0.0, acc_b = 0.0;
    for (int c0 = 0; c0 < n; c0 += 1024) {
        double va[16], vb[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = min(c0 + lane + 64 * i, n - 1);   // (clamped: the loads carry no branch)
            va[i] = row_a[c];
            vb[i] = row_b[c];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool in = c0 + lane + 64 * i < n;
            acc_a += in ? va[i] : 0.0;
            acc_b += in ? vb[i] : 0.0;
        }
    }
    sum_a = acc_a;
    sum_b = acc_b;
}

// min / max cost and the zero-weight count over the per-workgroup statistics: a wave of its own (n == R + 1), so that its
// loads run beside the row reductions instead of after one of them
__device__ __forceinline__ void finalize_cost_stats(const FinalizeArgs& A, const int lane) {
    double mn = INFINITY, mx = -INFINITY, nz = 0.0;
    for (int c0 = 0; c0 < A.nchunks; c0 += 1024) {
        double a[16], b[16], z[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {   // all loads first: one memory latency per 1024 partials
            const int c = min(c0 + lane + 64 * i, A.nchunks - 1);
            a[i] = A.statpart[c * 3 + 0];
            b[i] = A.statpart[c * 3 + 1];
            z[i] = A.statpart[c * 3 + 2];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool in = c0 + lane + 64 * i < A.nchunks;
            mn = fmin(mn, in ? a[i] : INFINITY);
            mx = fmax(mx, in ? b[i] : -INFINITY);
            nz += in ? z[i] : 0.0;
        }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    nz = wave_sum(nz);
    if (lane == 0) {
        A.stats[1] = mn;
        A.stats[2] = mx;
        A.stats[3] = nz;
        if (A.mail) {
            mail_post(A, A.R + 1, mn);
            mail_post(A, A.R + 2, mx);
            mail_post(A, A.R + 3, nz);
        }
    }
}
constexpr int finalize_blocks(int R) { return (R + 2 + kBlock / 64 - 1) / (kBlock / 64); }   // waves: R rows, sum w, statistics

__device__ __forceinline__ void finalize_rows(const FinalizeArgs& A) {
    const int lane = threadIdx.x & 63;
    // rows 0..R-1: one wave each; wave R: sum w; wave R + 1: the cost statistics
    const int n = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (n > A.R) {
        if (n == A.R + 1) finalize_cost_stats(A, lane);
        return;
    }
    const int nrow = n < A.R ? n : A.R;
    // S = sum w and this wave's row are fetched together
    double s, v;
    lane_partial_sum2(A.partial + (size_t)A.R * A.nchunks, A.partial + (size_t)nrow * A.nchunks, A.nchunks, lane, s, v);
    s = wave_sum(s);
    v = wave_sum(v);
    if (n < A.R && lane == 0) {
        A.vec[1 + n] = v;
        if (A.normalise) {
            const double q = v / s;
            A.nominal[n] = q;
            if (A.mail) mail_post(A, n, q);
        }
    }
    if (n == A.R && lane == 0) {
        A.vec[0] = s;
        A.stats[0] = s;
        if (A.mail) mail_post(A, A.R, s);
    }
}
__global__ __launch_bounds__(kBlock) void k_finalize(const FinalizeArgs A) { finalize_rows(A); }

// ---- K sharded over the GPUs of one node without a collective library call ---------------------------
// The exchanged message is 1 + (H-1)*u_dim doubles (<= 3.2 KB): far below the size at which a ring all-reduce pays, and a
// collective kernel launch between two rollout launches costs more than the transfer.  Instead every device owns an
// ExchangeBox in its HBM that all peers have mapped (hipIpc, xGMI peer access).  The wave of k_finalize_exchange that owns
// a row writes its value straight into slot [rank][row] of every peer's box, waits for the peers' values of the same row
// in its own box and adds them in rank order -- the same order on every device, so all devices hold the same bits.
// A value travels as two self-validating 8-byte packets {32 data bits, 32-bit sequence number} (each an atomic store):
// the receiver needs no flag and the sender no fence -- a packet is either the old one or the new one.  Two parities
// alternate: a slot is rewritten two exchanges later, which a peer can only reach after it has received this device's
// next packets, i.e. after this device's launch that read the slot has finished (stream order).
constexpr int kMaxRanks = 8;
constexpr int kMaxVec = 1 + (kMaxH - 1) * CCV_MPPI_MAX_UDIM;
struct ExchangeBox {
    unsigned long long pkt[2][kMaxRanks][kMaxVec][2];   // [parity][source rank][slot: 0 = sum w, 1 + row][high / low half]
};
struct ExchangeArgs {
    ExchangeBox* peer[kMaxRanks];   // every rank's box as mapped on this device (peer[rank] = the local one)
    ExchangeBox* local;
    double* reduced;                // [1 + R]: sum over ranks, rank order (becomes the deferred warm-start update)
    uint32_t seq;                   // this exchange's sequence number, never 0
    int32_t world, rank, parity;
    unsigned long long timeout_ticks;   // s_memrealtime ticks (100 MHz) to wait for the peers; then `reduced` is NaN
    int32_t* timeout_flag;              // set to 1 when that happens (the host reports it at the next synchronisation)
};

__device__ __forceinline__ unsigned long long load_system(const unsigned long long* p) {   // past every cache
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

__global__ __launch_bounds__(kBlock) void k_finalize_exchange(const FinalizeArgs A, const ExchangeArgs X) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);   // rows 0..R-1; wave R: sum w; wave R + 1: cost statistics
    if (n > A.R) {
        if (n == A.R + 1) finalize_cost_stats(A, lane);   // (this device's samples only)
        return;
    }
    const int nrow = n < A.R ? n : A.R;
    double s, v;
    lane_partial_sum2(A.partial + (size_t)A.R * A.nchunks, A.partial + (size_t)nrow * A.nchunks, A.nchunks, lane, s, v);
    s = wave_sum(s);
    v = wave_sum(v);
    // ---- this wave's value into slot [rank] of every peer's box: lane d writes to rank d
    const int slot = n < A.R ? 1 + n : 0;
    const double mine = n < A.R ? v : s;
    if (lane < X.world) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
        unsigned long long* dst = &X.peer[lane]->pkt[X.parity][X.rank][slot][0];
        __hip_atomic_store(dst + 0, (bits & 0xFFFFFFFF00000000ull) | X.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + 1, (bits << 32) | X.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (lane == 0) {   // (this device's share)
        A.vec[slot] = mine;
        if (n == A.R) A.stats[0] = s;
    }
    // ---- the peers' values of the same slot: lane r polls rank r's two packets in the local box
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long* src = &X.local->pkt[X.parity][lane < X.world ? lane : 0][slot][0];
    unsigned long long hi = 0, lo = 0;
    bool arrived = lane >= X.world;
    while (true) {
        if (!arrived) {
            hi = load_system(src + 0);
            lo = load_system(src + 1);
            arrived = (uint32_t)hi == X.seq && (uint32_t)lo == X.seq;
        }
        if (__all(arrived)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > X.timeout_ticks) break;
        __builtin_amdgcn_s_sleep(4);
    }
    const bool ok = __all(arrived);
    const double theirs = __longlong_as_double((long long)((hi & 0xFFFFFFFF00000000ull) | (lo >> 32)));
    double acc = 0.0;
    for (int r = 0; r < X.world; ++r) acc += lane_value(theirs, r);   // rank order on every device
    if (lane == 0) {
        X.reduced[slot] = ok ? acc : __builtin_nan("");
        if (!ok && X.timeout_flag) __hip_atomic_store(X.timeout_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (pinned host memory)
    }
}

// After the cross-device all-reduce of [sum w, sum w*u]: u* = V / S on every device.
__global__ __launch_bounds__(kBlock) void k_apply_partials(const double* vec, double* nominal, double* stats, int R) {
    const double S = vec[0];
    if (threadIdx.x == 0) stats[0] = S;
    for (int n = threadIdx.x; n < R; n += kBlock) nominal[n] = vec[1 + n] / S;
}

// ---- optional underflow-safe weights (CCV_MPPI_FLAG_MIN_SHIFT; not reference behaviour) --------------------------
__global__ __launch_bounds__(1024) void k_min_cost(const double* cost, int K, double* out_min) {
    __shared__ double red[16];
    double mn = INFINITY;
    for (int k = threadIdx.x; k < K; k += 1024) mn = fmin(mn, cost[k]);
    mn = wave_min(mn);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = red[0];
        for (int i = 1; i < 16; ++i) r = fmin(r, red[i]);
        *out_min = r;
    }
}
__global__ __launch_bounds__(kBlock) void k_reweight(const double* cost, const double* cmin, double lambda, int K, double* w) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k < K) w[k] = exp(-(cost[k] - *cmin) / lambda);
}

// ---- read-back helpers -----------------------------------------------------------------------------------------
// out[c][t][2] = (xs[t][first + c*stride], ys[t][...])
__global__ __launch_bounds__(kBlock) void k_gather_xy(const double* xs, const double* ys, int pitch, int H, int first,
                                                      int count, int stride, double* out) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= count * H) return;
    const int c = idx / H, t = idx % H;
    const size_t src = (size_t)t * pitch + first + (size_t)c * stride;
    out[(size_t)idx * 2 + 0] = xs[src];
    out[(size_t)idx * 2 + 1] = ys[src];
}

// ---- top-N candidates by weight (publish_CandidatePath() feed, dd:265-294: at K = 65 536 rviz can only draw a few) ----
// One workgroup.  Radix select on the weights' bit patterns (w >= 0, so the IEEE order is the integer order; NaN sorts
// above everything and is reported first, as a reader of a NaN iteration should see): eight 8-bit passes find the N-th
// largest key T and how many samples equal to T belong to the answer; a last pass writes the sample indices, "greater
// than T" first and then the lowest-index "equal to T" ones -- positions come from block-wide prefix sums in index
// order, so the output is the same on every run.  The host sorts the N pairs.
constexpr int kTopBlock = 1024;
__global__ __launch_bounds__(kTopBlock) void k_top_weights(const double* w, int K, int N, int* idx_out, double* w_out) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_remaining;
    __shared__ int wsum[kTopBlock / 64][2];
    __shared__ int s_base[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) {
        s_prefix = 0ull;
        s_remaining = N;
    }
    __syncthreads();
    for (int pass = 7; pass >= 0; --pass) {
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        for (int i = tid; i < K; i += kTopBlock) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(w[i]);
            const bool match = pass == 7 || (key >> (8 * (pass + 1))) == prefix;
            if (match) atomicAdd(&hist[(unsigned int)(key >> (8 * pass)) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int rem = s_remaining, b = 255;
            for (; b > 0; --b) {
                if ((int)hist[b] >= rem) break;
                rem -= (int)hist[b];
            }
            s_remaining = rem;                 // how many of bin b (and, after the last pass, of key T) are still wanted
            s_prefix = (prefix << 8) | (unsigned long long)b;
        }
        __syncthreads();
    }
    const unsigned long long T = s_prefix;
    const int n_equal = s_remaining, n_greater = N - n_equal;
    if (tid == 0) s_base[0] = s_base[1] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < K; i0 += kTopBlock) {
        const int i = i0 + tid;
        const unsigned long long key = i < K ? (unsigned long long)__double_as_longlong(w[i]) : 0ull;
        const int fg = (i < K && key > T) ? 1 : 0, fe = (i < K && key == T) ? 1 : 0;
        // exclusive prefix sums over the block in index order: wave ballots, then a scan of the 16 wave totals
        const unsigned long long bg = __ballot(fg), be = __ballot(fe);
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        const int pg = __popcll(bg & below), pe = __popcll(be & below);
        if (lane == 0) {
            wsum[wv][0] = __popcll(bg);
            wsum[wv][1] = __popcll(be);
        }
        __syncthreads();
        int og = s_base[0], oe = s_base[1];
        for (int j = 0; j < wv; ++j) {
            og += wsum[j][0];
            oe += wsum[j][1];
        }
        if (fg) {
            idx_out[og + pg] = i;
            w_out[og + pg] = w[i];
        }
        if (fe && oe + pe < n_equal) {
            idx_out[n_greater + oe + pe] = i;
            w_out[n_greater + oe + pe] = w[i];
        }
        __syncthreads();
        if (tid == 0) {
            int tg = 0, te = 0;
            for (int j = 0; j < kTopBlock / 64; ++j) {
                tg += wsum[j][0];
                te += wsum[j][1];
            }
            s_base[0] += tg;
            s_base[1] += te;
        }
        __syncthreads();
    }
}

// gather of listed samples: out[c][t] = (x, y) of sample idx[c] at step t
__global__ __launch_bounds__(kBlock) void k_gather_xy_list(const double* xs, const double* ys, int pitch, int H, const int* idx,
                                                           int count, double* out) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= count * H) return;
    const int c = j / H, t = j % H;
    const size_t src = (size_t)t * pitch + idx[c];
    out[(size_t)j * 2 + 0] = xs[src];
    out[(size_t)j * 2 + 1] = ys[src];
}

__global__ __launch_bounds__(kBlock) void k_normalise_weights(const double* w, const double* stats, int first, int count,
                                                             double* out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < count) out[i] = w[first + i] / stats[0];
}

}  // namespace ccv
