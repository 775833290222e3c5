// Counter-based control noise for the MI355X MPPI path (DESIGN.md "Noise spec").
//
// Replaces std::mt19937 + std::normal_distribution of the reference's sampling()
// (src/diff_drive_mppi.cpp:83-97): that generator is a single serial stream with a
// data-dependent number of engine words per variate (polar rejection), which cannot be
// drawn one-sample-per-lane.  Here normal number n = t*u_dim + d of global sample k in
// iteration `iter` is a pure function of (seed, iter, k, n):
//
//   words = Philox4x32-10(counter = {k, n/4, iter_lo, iter_hi}, key = {seed_lo, seed_hi})
//   (z[4c+0], z[4c+1]) = box_muller_f32(words[0], words[1]);  (z[4c+2], z[4c+3]) = box_muller_f32(words[2], words[3])
//
// box_muller_f32 uses only integer ops and correctly rounded IEEE fp32 add/mul/fma/sqrt with
// fixed polynomial coefficients, so a CPU restatement (oracle/philox_normal.h, test-only)
// reproduces every bit.  Build with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ccv {

// The machine scheduler re-serialises interleaved chains to save registers; a scheduling barrier between the stages of
// the N-wide helpers keeps the stage order (nothing else is affected: it emits no instruction).
#define CCV_KEEP_ORDER() __builtin_amdgcn_sched_barrier(0)

struct Philox4 {
    uint32_t x, y, z, w;
};

// a ^ b ^ c in one instruction (gfx950: v_bitop3_b32 with the truth table of the 3-input xor; the compiler does not form
// it by itself): two of them per Philox round instead of four v_xor_b32
__device__ __forceinline__ uint32_t xor3(const uint32_t a, const uint32_t b, const uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply per lane pair (v_mad_u64_u32) instead of mul_lo + mul_hi
        const unsigned long long pa = (unsigned long long)c0 * 0xD2511F53ull;
        const unsigned long long pb = (unsigned long long)c2 * 0xCD9E8D57ull;
        const uint32_t n0 = xor3((uint32_t)(pb >> 32), c1, k0);
        const uint32_t n2 = xor3((uint32_t)(pa >> 32), c3, k1);
        c1 = (uint32_t)pb;
        c3 = (uint32_t)pa;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// N independent Philox blocks with the rounds interleaved (round r of every block before round r+1 of any): the same
// arithmetic as philox4x32_10 per block, written in the order a lone wave should issue it -- the multiplies of the N
// blocks are independent, so their latencies overlap instead of adding up.  The compiler keeps this order.
template <int N>
__device__ __forceinline__ void philox4x32_10_n(uint32_t (&c0)[N], uint32_t (&c1)[N], uint32_t (&c2)[N], uint32_t (&c3)[N],
                                                uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long pa[N], pb[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            pa[i] = (unsigned long long)c0[i] * 0xD2511F53ull;
            pb[i] = (unsigned long long)c2[i] * 0xCD9E8D57ull;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t n0 = xor3((uint32_t)(pb[i] >> 32), c1[i], k0);
            const uint32_t n2 = xor3((uint32_t)(pa[i] >> 32), c3[i], k1);
            c1[i] = (uint32_t)pb[i];
            c3[i] = (uint32_t)pa[i];
            c0[i] = n0;
            c2[i] = n2;
        }
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
        CCV_KEEP_ORDER();
    }
}

// Correctly rounded sqrt for the radius.  The argument is -2 ln u1 with u1 in [2^-32, 1 - 2^-32], i.e. in [4.6e-10, 44.4]:
// the generic expansion's scaling of tiny arguments and its zero / infinity test are dead weight there (16 instructions per
// root).  What is left is its core: the hardware root (within 1 ulp) and the two residual tests that move it to the
// neighbour where needed -- the same bits as __builtin_sqrtf under -fhip-fp32-correctly-rounded-divide-sqrt, checked for
// every one of the 2^32 radius words and for every float in [2^-40, 2^7] by tools/microbench/sqrt_check.hip.
__device__ __forceinline__ float sqrt_cr_radius(const float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    float r = r_dn <= 0.0f ? s_dn : s;
    r = r_up > 0.0f ? s_up : r;
    return r;
}

// Quadrant rotation of Box-Muller by sign bits: z0 = r * (+-ca), z1 = r * (+-sa) with the cosine negative in quadrants 1, 2
// (bit 31 of b + 2^30) and the sine in quadrants 2, 3 (bit 31 of b); flipping the product's sign bit is the same as negating
// a factor.
__device__ __forceinline__ void bm_rotate(const uint32_t b, const float r, const float sn, const float cs, float& z0, float& z1) {
    const bool odd = (b & 0x40000000u) != 0u;
    const float ca = odd ? sn : cs, sa = odd ? cs : sn;
    // x ^ (y & 0x80000000): truth table 0xf0 ^ (0xcc & 0xaa) = 0x78 (operands count 0xf0, 0xcc, 0xaa)
    z0 = __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(r * ca), b + 0x40000000u, 0x80000000u, 0x78));
    z1 = __uint_as_float(__builtin_amdgcn_bitop3_b32(__float_as_uint(r * sa), b, 0x80000000u, 0x78));
}

// -log2 polynomial, sin/cos polynomials: tools/fit_normal_polys.py
#define CCV_Q0 0x1.715476p+0f
#define CCV_Q1 -0x1.715476p-1f
#define CCV_Q2 0x1.ec73e0p-2f
#define CCV_Q3 -0x1.715946p-2f
#define CCV_Q4 0x1.26cfb8p-2f
#define CCV_Q5 -0x1.e9df04p-3f
#define CCV_Q6 0x1.ba9caap-3f
#define CCV_Q7 -0x1.a548fcp-3f
#define CCV_Q8 0x1.f702acp-4f

// (a, b) -> two N(0,1) variates at fp32 resolution.  r = sqrt(-2 ln u1), u1 = max(a,1)/2^32;
// theta = (pi/2)*(quadrant + centred 30-bit fraction of b).
__device__ __forceinline__ void box_muller_f32(uint32_t a, uint32_t b, float& z0, float& z1) {
    const uint32_t a1 = a == 0u ? 1u : a;
    const int lz = __builtin_clz(a1);
    const uint32_t m = a1 << lz;                      // u1 = m * 2^(-32-lz)
    const bool fold = m > 0xB504F333u;                // mantissa > sqrt(2): use m/2^32 in [sqrt(.5),1), bump exponent
    // t = mantissa - 1 from the integer: keeps full relative precision as u1 -> 1
    const float t = fold ? -((float)(0u - m) * 0x1p-32f) : (float)(m - 0x80000000u) * 0x1p-31f;
    const float L0 = (float)(1 + lz - (fold ? 1 : 0));
    float q = CCV_Q8;
    q = __builtin_fmaf(q, t, CCV_Q7);
    q = __builtin_fmaf(q, t, CCV_Q6);
    q = __builtin_fmaf(q, t, CCV_Q5);
    q = __builtin_fmaf(q, t, CCV_Q4);
    q = __builtin_fmaf(q, t, CCV_Q3);
    q = __builtin_fmaf(q, t, CCV_Q2);
    q = __builtin_fmaf(q, t, CCV_Q1);
    q = __builtin_fmaf(q, t, CCV_Q0);
    const float L = __builtin_fmaf(-t, q, L0);       // -log2(u1) >= 0
    const float r = sqrt_cr_radius(L * 0x1.62e430p+0f);   // correctly rounded

    const uint32_t quad = b >> 30;
    const int32_t f = (int32_t)(b & 0x3FFFFFFFu) - (1 << 29);
    const float al = (float)f * 0x1.921fb6p-30f;     // [-pi/4, pi/4)
    const float w = al * al;
    float s = 0x1.6dbc3ep-19f, c = 0x1.9a6a98p-16f;
    s = __builtin_fmaf(s, w, -0x1.a013a2p-13f);
    c = __builtin_fmaf(c, w, -0x1.6c0df8p-10f);
    s = __builtin_fmaf(s, w, 0x1.11110ep-7f);
    c = __builtin_fmaf(c, w, 0x1.55554cp-5f);
    s = __builtin_fmaf(s, w, -0x1.555556p-3f);
    c = __builtin_fmaf(c, w, -0x1.000000p-1f);
    const float sn = __builtin_fmaf(al * w, s, al);
    const float cs = __builtin_fmaf(w, c, 1.0f);
    // rotate by quadrant: (cos,sin)(theta) for theta = quad*pi/2 + al
    (void)quad;
    bm_rotate(b, r, sn, cs, z0, z1);
}

// N independent Box-Muller pairs, stage by stage (the same per-element arithmetic as box_muller_f32, bit for bit): the
// two serial polynomial chains of every pair advance together, N-way instruction-level parallelism for a lone wave.
template <int N>
__device__ __forceinline__ void box_muller_f32_n(const uint32_t (&a)[N], const uint32_t (&b)[N], float (&z0)[N], float (&z1)[N]) {
    float t[N], L0[N], q[N], al[N], w[N], s[N], c[N], r[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t a1 = a[i] == 0u ? 1u : a[i];
        const int lz = __builtin_clz(a1);
        const uint32_t m = a1 << lz;
        const bool fold = m > 0xB504F333u;
        t[i] = fold ? -((float)(0u - m) * 0x1p-32f) : (float)(m - 0x80000000u) * 0x1p-31f;
        L0[i] = (float)(1 + lz - (fold ? 1 : 0));
        const int32_t f = (int32_t)(b[i] & 0x3FFFFFFFu) - (1 << 29);
        al[i] = (float)f * 0x1.921fb6p-30f;
        w[i] = al[i] * al[i];
        q[i] = CCV_Q8;
        s[i] = 0x1.6dbc3ep-19f;
        c[i] = 0x1.9a6a98p-16f;
    }
    CCV_KEEP_ORDER();
#define CCV_STAGE(var, x, coef)                                                                   \
    _Pragma("unroll") for (int i = 0; i < N; ++i) var[i] = __builtin_fmaf(var[i], x[i], coef); \
    CCV_KEEP_ORDER();
    CCV_STAGE(q, t, CCV_Q7)
    CCV_STAGE(s, w, -0x1.a013a2p-13f)
    CCV_STAGE(c, w, -0x1.6c0df8p-10f)
    CCV_STAGE(q, t, CCV_Q6)
    CCV_STAGE(s, w, 0x1.11110ep-7f)
    CCV_STAGE(c, w, 0x1.55554cp-5f)
    CCV_STAGE(q, t, CCV_Q5)
    CCV_STAGE(s, w, -0x1.555556p-3f)
    CCV_STAGE(c, w, -0x1.000000p-1f)
    CCV_STAGE(q, t, CCV_Q4)
    CCV_STAGE(q, t, CCV_Q3)
    CCV_STAGE(q, t, CCV_Q2)
    CCV_STAGE(q, t, CCV_Q1)
    CCV_STAGE(q, t, CCV_Q0)
#undef CCV_STAGE
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float L = __builtin_fmaf(-t[i], q[i], L0[i]);
        r[i] = sqrt_cr_radius(L * 0x1.62e430p+0f);
    }
    CCV_KEEP_ORDER();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float sn = __builtin_fmaf(al[i] * w[i], s[i], al[i]);
        const float cs = __builtin_fmaf(w[i], c[i], 1.0f);
        bm_rotate(b[i], r[i], sn, cs, z0[i], z1[i]);
    }
}

}  // namespace ccv
