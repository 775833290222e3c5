// Exhaustive check of the two Box-Muller shortcuts of noise_spec.h (round 2) against the plain formulation:
//   (1) sqrt_cr_radius(x) == __builtin_sqrtf(x) (correctly rounded under -fhip-fp32-correctly-rounded-divide-sqrt) for the
//       radius argument of every one of the 2^32 words a, and for every float in [2^-40, 2^7];
//   (2) box_muller_f32(a, b) == the formulation with the quadrant's compare / select / negate, for all 2^32 words a with
//       b = hash(a), and for all 2^32 words b with a = hash(b).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I../../ccv_mppi_path_tracker_amd/csrc \
//         -o sqrt_check sqrt_check.hip && ./sqrt_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "noise_spec.h"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// the formulation before the shortcuts (radius by __builtin_sqrtf, quadrant by compares)
__device__ __forceinline__ void box_muller_plain(uint32_t a, uint32_t b, float& z0, float& z1, float& xr) {
    const uint32_t a1 = a == 0u ? 1u : a;
    const int lz = __builtin_clz(a1);
    const uint32_t m = a1 << lz;
    const bool fold = m > 0xB504F333u;
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
    const float L = __builtin_fmaf(-t, q, L0);
    xr = L * 0x1.62e430p+0f;
    const float r = __builtin_sqrtf(xr);
    const uint32_t quad = b >> 30;
    const int32_t f = (int32_t)(b & 0x3FFFFFFFu) - (1 << 29);
    const float al = (float)f * 0x1.921fb6p-30f;
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
    const float ca = (quad & 1u) ? sn : cs;
    const float sa = (quad & 1u) ? cs : sn;
    const float cq = (quad == 1u || quad == 2u) ? -ca : ca;
    const float sq = (quad >= 2u) ? -sa : sa;
    z0 = r * cq;
    z1 = r * sq;
}

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// mode 0: a sweeps, b = hash(a); mode 1: b sweeps, a = hash(b); mode 2: sqrt over float bit patterns [lo, hi)
__global__ __launch_bounds__(256) void k_check(int mode, unsigned long long first, unsigned long long count, unsigned long long* bad,
                                               float* minx, float* maxx) {
    unsigned long long nbad = 0;
    float mn = 1e30f, mx = 0.0f;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256) {
        const uint32_t v = (uint32_t)(first + i);
        if (mode == 2) {
            const float x = __uint_as_float(v);
            if (__float_as_uint(ccv::sqrt_cr_radius(x)) != __float_as_uint(__builtin_sqrtf(x))) ++nbad;
        } else {
            const uint32_t a = mode == 0 ? v : hash32(v), b = mode == 0 ? hash32(v) : v;
            float p0, p1, xr, n0, n1;
            box_muller_plain(a, b, p0, p1, xr);
            ccv::box_muller_f32(a, b, n0, n1);
            if (__float_as_uint(p0) != __float_as_uint(n0) || __float_as_uint(p1) != __float_as_uint(n1)) ++nbad;
            if (__float_as_uint(ccv::sqrt_cr_radius(xr)) != __float_as_uint(__builtin_sqrtf(xr))) ++nbad;
            mn = fminf(mn, xr);
            mx = fmaxf(mx, xr);
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    if (mode != 2) {
        atomicMin((unsigned*)minx, __float_as_uint(mn));   // (positive floats order like their bit patterns)
        atomicMax((unsigned*)maxx, __float_as_uint(mx));
    }
}

int main() {
    unsigned long long* bad;
    float* mm;
    CHECK(hipMalloc(&bad, 8));
    CHECK(hipMalloc(&mm, 8));
    int fails = 0;
    for (int mode = 0; mode < 3; ++mode) {
        CHECK(hipMemset(bad, 0, 8));
        const float init[2] = {1e30f, 0.0f};
        CHECK(hipMemcpy(mm, init, 8, hipMemcpyHostToDevice));
        unsigned long long first = 0, count = 1ull << 32;
        if (mode == 2) {
            const float lo = 0x1p-40f, hi = 0x1p7f;
            unsigned ulo, uhi;
            memcpy(&ulo, &lo, 4); memcpy(&uhi, &hi, 4);
            first = ulo; count = (unsigned long long)uhi - ulo + 1;
        }
        hipLaunchKernelGGL(k_check, dim3(256 * 64), dim3(256), 0, 0, mode, first, count, bad, mm, mm + 1);
        CHECK(hipDeviceSynchronize());
        unsigned long long h;
        float hm[2];
        CHECK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hm, mm, 8, hipMemcpyDeviceToHost));
        if (mode == 2) printf("sqrt_cr_radius vs __builtin_sqrtf over %llu floats in [2^-40, 2^7]: %llu mismatches\n", count, h);
        else printf("box_muller_f32 vs plain formulation, %s sweeps all 2^32 words: %llu mismatches; radius argument in [%.3e, %.4f]\n",
                    mode == 0 ? "a" : "b", h, hm[0], hm[1]);
        fails += h != 0;
    }
    printf(fails ? "FAILED\n" : "OK\n");
    return fails ? 1 : 0;
}
