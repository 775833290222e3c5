// v_mfma_f64_16x16x4_f64 on gfx950: (1) lane maps and rounding against an fma chain, (2) cycles per instruction
// back-to-back on one SIMD, (3) whether a second wave's fp64 VALU stream on the same SIMD runs concurrently.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64 mfma_f64.hip && ./mfma_f64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A, const double* B, double* D) {   // A[16][4], B[4][16], D[16][16]
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];
    const double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(64 * v) + l] = c[v];   // raw: register v of lane l
}

template <int MODE>   // 0: MFMA waves only, 1: FMA waves only, 2: both; waves 0-3 MFMA, waves 4-7 FMA (one each per SIMD)
__global__ __launch_bounds__(512) void k_rate(double* out, unsigned long long* cyc, int iters) {
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const bool mf = wv < 4;
    if ((MODE == 0 && !mf) || (MODE == 1 && mf)) return;
    unsigned long long t0, t1;
    double r = 0;
    if (mf) {
        double a = 1.0 + l * 1e-3, b = 0.5 + l * 1e-4;
        d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        r = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        double x0 = l, x1 = l + 1, x2 = l + 2, x3 = l + 3, x4 = l + 4, x5 = l + 5, x6 = l + 6, x7 = l + 7;
        const double m = 0.999999, a = 1e-9;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int i = 0; i < iters; ++i) {
            x0 = fma(x0, m, a); x1 = fma(x1, m, a); x2 = fma(x2, m, a); x3 = fma(x3, m, a);
            x4 = fma(x4, m, a); x5 = fma(x5, m, a); x6 = fma(x6, m, a); x7 = fma(x7, m, a);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (l == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

int main() {
    // ---- layout + rounding ----
    std::vector<double> A(64), B(64), D(256);
    srand(1);
    for (auto& x : A) x = (rand() / (double)RAND_MAX - 0.5) * 3.0;
    for (auto& x : B) x = (rand() / (double)RAND_MAX - 0.5) * 3.0;
    double *dA, *dB, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    int bad_map = 0, exact_fwd = 0, exact_rev = 0;
    double maxrel = 0;
    for (int v = 0; v < 4; ++v)
        for (int l = 0; l < 64; ++l) {
            const int col = l & 15, row = (l >> 4) + 4 * v;
            double fwd = 0, rev = 0, ref = 0;
            for (int k = 0; k < 4; ++k) fwd = fma(A[row * 4 + k], B[k * 16 + col], fwd);
            for (int k = 3; k >= 0; --k) rev = fma(A[row * 4 + k], B[k * 16 + col], rev);
            long double acc = 0;
            for (int k = 0; k < 4; ++k) acc += (long double)A[row * 4 + k] * B[k * 16 + col];
            ref = (double)acc;
            const double got = D[64 * v + l];
            if (fabs(got - ref) > 1e-12) bad_map++;
            exact_fwd += got == fwd;
            exact_rev += got == rev;
            maxrel = fmax(maxrel, fabs(got - ref));
        }
    printf("layout col=l&15,row=(l>>4)+4v: mismatches %d / 256; == fma chain k=0..3: %d, k=3..0: %d; max |err| %.3g\n",
           bad_map, exact_fwd, exact_rev, maxrel);
    // ---- rates ----
    const int blocks = 256, iters = 2000;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 512 * 8); hipMalloc(&cyc, blocks * 8 * 8);
    std::vector<unsigned long long> h(blocks * 8);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(cyc, 0, blocks * 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k_rate<0><<<blocks, 512>>>(out, cyc, iters);
            if (mode == 1) k_rate<1><<<blocks, 512>>>(out, cyc, iters);
            if (mode == 2) k_rate<2><<<blocks, 512>>>(out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, blocks * 64, hipMemcpyDeviceToHost);
        double mf = 0, fm = 0; int nm = 0, nf = 0;
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < 8; ++w) {
                if (!h[b * 8 + w]) continue;
                if (w < 4) { mf += h[b * 8 + w]; nm++; } else { fm += h[b * 8 + w]; nf++; }
            }
        // s_memtime counts at 100 MHz on gfx950 -> report ns per instruction and kernel time
        printf("mode %d: kernel %.1f us;", mode, ms * 1e3);
        if (nm) printf("  mfma: %.2f ns/instr (memtime ticks %.0f)", mf / nm * 10.0 / (iters * 4.0), mf / nm);
        if (nf) printf("  fma64: %.2f ns/instr", fm / nf * 10.0 / (iters * 8.0));
        printf("\n");
    }
    return 0;
}
