// Do an fp64 VALU stream and an fp32 / int32 VALU stream from two waves of one SIMD overlap on gfx950?
// 512-thread workgroups: waves 0-3 run stream A, waves 4-7 stream B (one of each per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o valu_mix valu_mix.hip && ./valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum { F64 = 0, F32 = 1, I32 = 2, MAD64 = 3, NONE = 4 };

template <int KIND>
__device__ __forceinline__ double stream(int iters, int l) {
    if constexpr (KIND == F64) {
        double x0 = l, x1 = l + 1, x2 = l + 2, x3 = l + 3, x4 = l + 4, x5 = l + 5, x6 = l + 6, x7 = l + 7;
        const double m = 0.999999, a = 1e-9;
#pragma unroll 16
        for (int i = 0; i < iters; ++i) {
            x0 = fma(x0, m, a); x1 = fma(x1, m, a); x2 = fma(x2, m, a); x3 = fma(x3, m, a);
            x4 = fma(x4, m, a); x5 = fma(x5, m, a); x6 = fma(x6, m, a); x7 = fma(x7, m, a);
        }
        return x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else if constexpr (KIND == F32) {
        float x0 = l, x1 = l + 1, x2 = l + 2, x3 = l + 3, x4 = l + 4, x5 = l + 5, x6 = l + 6, x7 = l + 7;
        const float m = 0.9999f, a = 1e-6f;
#pragma unroll 16
        for (int i = 0; i < iters; ++i) {
            x0 = fmaf(x0, m, a); x1 = fmaf(x1, m, a); x2 = fmaf(x2, m, a); x3 = fmaf(x3, m, a);
            x4 = fmaf(x4, m, a); x5 = fmaf(x5, m, a); x6 = fmaf(x6, m, a); x7 = fmaf(x7, m, a);
        }
        return x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    } else if constexpr (KIND == I32) {
        unsigned x0 = l, x1 = l + 1, x2 = l + 2, x3 = l + 3, x4 = l + 4, x5 = l + 5, x6 = l + 6, x7 = l + 7;
#pragma unroll 16
        for (int i = 0; i < iters; ++i) {
            x0 = (x0 ^ 0x9E3779B9u) + i; x1 = (x1 ^ 0x9E3779B9u) + i; x2 = (x2 ^ 0x9E3779B9u) + i; x3 = (x3 ^ 0x9E3779B9u) + i;
            x4 = (x4 ^ 0x9E3779B9u) + i; x5 = (x5 ^ 0x9E3779B9u) + i; x6 = (x6 ^ 0x9E3779B9u) + i; x7 = (x7 ^ 0x9E3779B9u) + i;
        }
        return (double)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
    } else if constexpr (KIND == MAD64) {
        unsigned x0 = l, x1 = l + 1, x2 = l + 2, x3 = l + 3, x4 = l + 4, x5 = l + 5, x6 = l + 6, x7 = l + 7;
#pragma unroll 16
        for (int i = 0; i < iters; ++i) {
            x0 = __umulhi(x0, 0xD2511F53u) ^ i; x1 = __umulhi(x1, 0xD2511F53u) ^ i; x2 = __umulhi(x2, 0xD2511F53u) ^ i;
            x3 = __umulhi(x3, 0xD2511F53u) ^ i; x4 = __umulhi(x4, 0xD2511F53u) ^ i; x5 = __umulhi(x5, 0xD2511F53u) ^ i;
            x6 = __umulhi(x6, 0xD2511F53u) ^ i; x7 = __umulhi(x7, 0xD2511F53u) ^ i;
        }
        return (double)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
    }
    return 0.0;
}

template <int KA, int KB>
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* cyc, int iters) {
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const bool isA = wv < 4;
    if ((isA && KA == NONE) || (!isA && KB == NONE)) return;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const double r = isA ? stream<KA>(iters, l) : stream<KB>(iters, l);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(r) : "memory");
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (l == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int KA, int KB>
void run(const char* name, double* out, unsigned long long* cyc, int ops_a, int ops_b) {
    const int blocks = 256, iters = 4000;
    std::vector<unsigned long long> h(blocks * 8);
    hipMemset(cyc, 0, blocks * 64);
    for (int rep = 0; rep < 3; ++rep) k<KA, KB><<<blocks, 512>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, blocks * 64, hipMemcpyDeviceToHost);
    double a = 0, b = 0; int na = 0, nb = 0;
    for (int i = 0; i < blocks; ++i)
        for (int w = 0; w < 8; ++w) {
            if (!h[i * 8 + w]) continue;
            if (w < 4) { a += h[i * 8 + w]; na++; } else { b += h[i * 8 + w]; nb++; }
        }
    printf("%-28s", name);
    if (na) printf("  A: %6.2f cyc/instr", a / na / (iters * (double)ops_a));
    if (nb) printf("  B: %6.2f cyc/instr", b / nb / (iters * (double)ops_b));
    printf("\n");
}

int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 64);
    run<F64, NONE>("f64 alone", out, cyc, 8, 8);
    run<F32, NONE>("f32 alone", out, cyc, 8, 8);
    run<I32, NONE>("i32 (xor,add) alone", out, cyc, 16, 16);
    run<MAD64, NONE>("mul_hi (+xor) alone", out, cyc, 16, 16);
    run<F64, F64>("f64 | f64", out, cyc, 8, 8);
    run<F32, F32>("f32 | f32", out, cyc, 8, 8);
    run<F64, F32>("f64 | f32", out, cyc, 8, 8);
    run<F64, I32>("f64 | i32", out, cyc, 8, 16);
    run<F64, MAD64>("f64 | mul_hi", out, cyc, 8, 16);
    run<F32, MAD64>("f32 | mul_hi", out, cyc, 8, 16);
    return 0;
}
