// Micro-benchmark of the O(K*H^2) core (calc_MinDistance, dd:183-192): cycles per (trajectory point, window point)
// pair for several formulations of the inner loop.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <type_traits>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int TU = 8;
constexpr int REP = 64;
constexpr int MAXH = 128;

struct Win { double a[MAXH], b[MAXH], c[MAXH]; };

template <int N, class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ double vmin64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// V0: plain fmin, LDS reads at use (what the first kernel did)
__device__ __forceinline__ void v0(const double (&px)[TU], const double (&py)[TU], double (&m)[TU], int H, const double2* sab, const double* sc, const Win&) {
#pragma unroll 2
    for (int j = 0; j < H; ++j) {
        const double2 ab = sab[j]; const double c = sc[j];
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; m[i] = fmin(m[i], fma(ab.x, px[i], fma(ab.y, py[i], c))); });
    }
}
// V1: asm min + rotation (prefetch next from LDS)
__device__ __forceinline__ void v1(const double (&px)[TU], const double (&py)[TU], double (&m)[TU], int H, const double2* sab, const double* sc, const Win&) {
    double a = sab[0].x, b = sab[0].y, c = sc[0];
    for (int j = 0; j < H; ++j) {
        const int jn = (j + 1 < H) ? j + 1 : j;
        const double2 abn = sab[jn]; const double cn = sc[jn];
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
        double t[TU];
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; t[i] = fma(b, py[i], c); });
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; t[i] = fma(a, px[i], t[i]); });
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; m[i] = vmin64(m[i], t[i]); });
        a = abn.x; b = abn.y; c = cn;
    }
}
// V2: asm min, two points per iteration, prefetch the next two
__device__ __forceinline__ void v2(const double (&px)[TU], const double (&py)[TU], double (&m)[TU], int H, const double2* sab, const double* sc, const Win&) {
    double a0 = sab[0].x, b0 = sab[0].y, c0 = sc[0], a1 = sab[1].x, b1 = sab[1].y, c1 = sc[1];
    for (int j = 0; j < H; j += 2) {   // H even here
        const int jn = (j + 2 < H) ? j + 2 : j;
        const double2 abn0 = sab[jn], abn1 = sab[jn + 1]; const double cn0 = sc[jn], cn1 = sc[jn + 1];
        asm volatile("" : "+v"(a0), "+v"(b0), "+v"(c0), "+v"(a1), "+v"(b1), "+v"(c1));
        double t[TU], s[TU];
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; t[i] = fma(b0, py[i], c0); });
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; s[i] = fma(b1, py[i], c1); });
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; t[i] = fma(a0, px[i], t[i]); });
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; s[i] = fma(a1, px[i], s[i]); });
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; m[i] = vmin64(m[i], t[i]); });
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; m[i] = vmin64(m[i], s[i]); });
        a0 = abn0.x; b0 = abn0.y; c0 = cn0; a1 = abn1.x; b1 = abn1.y; c1 = cn1;
    }
}
// V3: coefficients straight from the kernel-argument segment (scalar loads), asm min
__device__ __forceinline__ void v3(const double (&px)[TU], const double (&py)[TU], double (&m)[TU], int H, const double2*, const double*, const Win& W) {
#pragma unroll 2
    for (int j = 0; j < H; ++j) {
        const double a = W.a[j], b = W.b[j], c = W.c[j];
        double t[TU];
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; t[i] = fma(b, py[i], c); });
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; t[i] = fma(a, px[i], t[i]); });
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; m[i] = vmin64(m[i], t[i]); });
    }
}
// V4: plain fmin but 4 points per iteration (amortises the canonicalising max), LDS
__device__ __forceinline__ void v4(const double (&px)[TU], const double (&py)[TU], double (&m)[TU], int H, const double2* sab, const double* sc, const Win&) {
#pragma unroll 4
    for (int j = 0; j < H; ++j) {
        const double2 ab = sab[j]; const double c = sc[j];
        sfor<TU>([&](auto I) { constexpr int i = decltype(I)::value; m[i] = fmin(m[i], fma(ab.x, px[i], fma(ab.y, py[i], c))); });
    }
}

// V5: the 24 instructions of one window point as ONE asm block (8 FMA, 8 FMA, 8 MIN: dependent ops 8 issue slots apart, no
// canonicalising max), next point's coefficients prefetched from LDS by rotation
#define TRIPLE_ASM(a, b, c)                                                                                              \
    asm volatile(                                                                                                        \
        "v_fma_f64 %8, %24, %16, %26\n\tv_fma_f64 %9, %24, %17, %26\n\tv_fma_f64 %10, %24, %18, %26\n\tv_fma_f64 %11, %24, %19, %26\n\t" \
        "v_fma_f64 %12, %24, %20, %26\n\tv_fma_f64 %13, %24, %21, %26\n\tv_fma_f64 %14, %24, %22, %26\n\tv_fma_f64 %15, %24, %23, %26\n\t" \
        "v_fma_f64 %8, %25, %27, %8\n\tv_fma_f64 %9, %25, %28, %9\n\tv_fma_f64 %10, %25, %29, %10\n\tv_fma_f64 %11, %25, %30, %11\n\t"   \
        "v_fma_f64 %12, %25, %31, %12\n\tv_fma_f64 %13, %25, %32, %13\n\tv_fma_f64 %14, %25, %33, %14\n\tv_fma_f64 %15, %25, %34, %15\n\t" \
        "v_min_f64 %0, %0, %8\n\tv_min_f64 %1, %1, %9\n\tv_min_f64 %2, %2, %10\n\tv_min_f64 %3, %3, %11\n\t"                         \
        "v_min_f64 %4, %4, %12\n\tv_min_f64 %5, %5, %13\n\tv_min_f64 %6, %6, %14\n\tv_min_f64 %7, %7, %15"                            \
        : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]), "=&v"(t[0]),   \
          "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])                      \
        : "v"(py[0]), "v"(py[1]), "v"(py[2]), "v"(py[3]), "v"(py[4]), "v"(py[5]), "v"(py[6]), "v"(py[7]), "v"(b), "v"(a), \
          "v"(c), "v"(px[0]), "v"(px[1]), "v"(px[2]), "v"(px[3]), "v"(px[4]), "v"(px[5]), "v"(px[6]), "v"(px[7]))
__device__ __forceinline__ void v5(const double (&px)[TU], const double (&py)[TU], double (&m)[TU], int H, const double2* sab, const double* sc, const Win&) {
    double a = sab[0].x, b = sab[0].y, c = sc[0];
    double t[TU];
    for (int j = 0; j < H; ++j) {
        const int jn = (j + 1 < H) ? j + 1 : j;
        const double2 abn = sab[jn]; const double cn = sc[jn];
        TRIPLE_ASM(a, b, c);
        a = abn.x; b = abn.y; c = cn;
    }
}
// V6: as V5, two points per iteration (prefetch distance = 48 instructions)
__device__ __forceinline__ void v6(const double (&px)[TU], const double (&py)[TU], double (&m)[TU], int H, const double2* sab, const double* sc, const Win&) {
    double a0 = sab[0].x, b0 = sab[0].y, c0 = sc[0], a1 = sab[1].x, b1 = sab[1].y, c1 = sc[1];
    double t[TU];
    for (int j = 0; j < H; j += 2) {
        const int jn = (j + 2 < H) ? j + 2 : j;
        const double2 abn0 = sab[jn], abn1 = sab[jn + 1]; const double cn0 = sc[jn], cn1 = sc[jn + 1];
        TRIPLE_ASM(a0, b0, c0);
        TRIPLE_ASM(a1, b1, c1);
        a0 = abn0.x; b0 = abn0.y; c0 = cn0; a1 = abn1.x; b1 = abn1.y; c1 = cn1;
    }
}

// V7: plain fmin, 4 points per group, the NEXT group's coefficients are fetched from LDS while the current group is
// evaluated (ping-pong register sets; the empty asm pins the current set so the loads cannot be re-materialised at use)
#define PIN4(ab, c) asm volatile("" : "+v"(ab[0].x), "+v"(ab[0].y), "+v"(ab[1].x), "+v"(ab[1].y), "+v"(ab[2].x), "+v"(ab[2].y), \
                                      "+v"(ab[3].x), "+v"(ab[3].y), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]))
__device__ __forceinline__ void group4(const double (&px)[TU], const double (&py)[TU], double (&m)[TU], const double2 (&ab)[4], const double (&c)[4]) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int i = 0; i < TU; ++i) m[i] = fmin(m[i], fma(ab[jj].x, px[i], fma(ab[jj].y, py[i], c[jj])));
}
__device__ __forceinline__ void v7(const double (&px)[TU], const double (&py)[TU], double (&m)[TU], int H, const double2* sab, const double* sc, const Win&) {
    const int H4 = (H + 3) & ~3;   // caller's arrays are padded
    double2 a0[4], a1[4]; double c0[4], c1[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { a0[jj] = sab[jj]; c0[jj] = sc[jj]; }
    for (int j = 0; j < H4; j += 8) {
        const int j1 = (j + 4 < H4) ? j + 4 : j;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { a1[jj] = sab[j1 + jj]; c1[jj] = sc[j1 + jj]; }
        PIN4(a0, c0);
        group4(px, py, m, a0, c0);
        const int j2 = (j + 8 < H4) ? j + 8 : j1;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { a0[jj] = sab[j2 + jj]; c0[jj] = sc[j2 + jj]; }
        PIN4(a1, c1);
        if (j + 4 < H4) group4(px, py, m, a1, c1);
    }
}

// ---- fp32 SCREEN forms: what a screening pass over the same pairs costs before any refinement ----
// V8: 2 v_fma_f32 + 1 v_min_f32 per pair, 4 points per group (the minimum an fp32 screen can be)
// V9: the same with packed math: two states per v_pk_fma_f32
// V10: V8 + the cheapest bookkeeping that lets a refinement find its candidates: per group of 4 points and state the
//      group minimum, a compare, a select of the group index, and the runner-up group minimum (med3) to validate it
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void v8(const float (&px)[TU], const float (&py)[TU], float (&m)[TU], int H, const float2* sab, const float* sc) {
    const int H4 = (H + 3) & ~3;
    for (int j = 0; j < H4; j += 4) {
        float2 ab[4]; float c[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { ab[jj] = sab[j + jj]; c[jj] = sc[j + jj]; }
#pragma unroll
        for (int i = 0; i < TU; ++i) {
            float f[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) f[jj] = __builtin_fmaf(ab[jj].x, px[i], __builtin_fmaf(ab[jj].y, py[i], c[jj]));
            const float t = fminf(fminf(f[0], f[1]), fminf(f[2], f[3]));
            asm("v_min_f32 %0, %1, %2" : "=v"(m[i]) : "v"(m[i]), "v"(t));
        }
    }
}
__device__ __forceinline__ void v9(const float (&px)[TU], const float (&py)[TU], float (&m)[TU], int H, const float2* sab, const float* sc) {
    const int H4 = (H + 3) & ~3;
    f32x2 qx[TU / 2], qy[TU / 2];
#pragma unroll
    for (int i = 0; i < TU / 2; ++i) { qx[i] = f32x2{px[2 * i], px[2 * i + 1]}; qy[i] = f32x2{py[2 * i], py[2 * i + 1]}; }
    for (int j = 0; j < H4; j += 4) {
        float2 ab[4]; float c[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { ab[jj] = sab[j + jj]; c[jj] = sc[j + jj]; }
#pragma unroll
        for (int i = 0; i < TU / 2; ++i) {
            f32x2 f[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x2 a2{ab[jj].x, ab[jj].x}, b2{ab[jj].y, ab[jj].y}, c2{c[jj], c[jj]};
                f[jj] = __builtin_elementwise_fma(a2, qx[i], __builtin_elementwise_fma(b2, qy[i], c2));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float t = fminf(fminf(f[0][h], f[1][h]), fminf(f[2][h], f[3][h]));
                asm("v_min_f32 %0, %1, %2" : "=v"(m[2 * i + h]) : "v"(m[2 * i + h]), "v"(t));
            }
        }
    }
}
__device__ __forceinline__ void v10(const float (&px)[TU], const float (&py)[TU], float (&m)[TU], float (&m2)[TU], int (&g)[TU], int H,
                                    const float2* sab, const float* sc) {
    const int H4 = (H + 3) & ~3;
    for (int j = 0; j < H4; j += 4) {
        float2 ab[4]; float c[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { ab[jj] = sab[j + jj]; c[jj] = sc[j + jj]; }
#pragma unroll
        for (int i = 0; i < TU; ++i) {
            float f[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) f[jj] = __builtin_fmaf(ab[jj].x, px[i], __builtin_fmaf(ab[jj].y, py[i], c[jj]));
            const float t = fminf(fminf(f[0], f[1]), fminf(f[2], f[3]));
            g[i] = t < m[i] ? j : g[i];
            m2[i] = __builtin_amdgcn_fmed3f(t, m[i], m2[i]);
            asm("v_min_f32 %0, %1, %2" : "=v"(m[i]) : "v"(m[i]), "v"(t));
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void k32(double* out, int H, unsigned long long* cyc, const Win W) {
    __shared__ float2 sab[MAXH];
    __shared__ float sc[MAXH];
    for (int j = threadIdx.x; j < H + 8; j += 256) { sab[j] = make_float2((float)W.a[j], (float)W.b[j]); sc[j] = (float)W.c[j]; }
    __syncthreads();
    float px[TU], py[TU], m[TU], m2[TU];
    int g[TU];
#pragma unroll
    for (int i = 0; i < TU; ++i) { px[i] = 0.01f * threadIdx.x + i; py[i] = 0.02f * threadIdx.x - i; m[i] = 1e30f; m2[i] = 1e30f; g[i] = 0; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
        if (V == 8) v8(px, py, m, H, sab, sc);
        else if (V == 9) v9(px, py, m, H, sab, sc);
        else v10(px, py, m, m2, g, H, sab, sc);
#pragma unroll
        for (int i = 0; i < TU; ++i) px[i] += 1e-9f * m[i];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < TU; ++i) s += m[i] + m2[i] + g[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int V>
__global__ __launch_bounds__(256) void k(double* out, int H, unsigned long long* cyc, const Win W) {
    __shared__ double2 sab[MAXH];
    __shared__ double sc[MAXH];
    for (int j = threadIdx.x; j < H + 8; j += 256) { sab[j] = make_double2(W.a[j], W.b[j]); sc[j] = W.c[j]; }
    __syncthreads();
    double px[TU], py[TU], m[TU];
#pragma unroll
    for (int i = 0; i < TU; ++i) { px[i] = 0.01 * threadIdx.x + i; py[i] = 0.02 * threadIdx.x - i; m[i] = 1e300; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
        if (V == 0) v0(px, py, m, H, sab, sc, W);
        else if (V == 1) v1(px, py, m, H, sab, sc, W);
        else if (V == 2) v2(px, py, m, H, sab, sc, W);
        else if (V == 3) v3(px, py, m, H, sab, sc, W);
        else if (V == 5) v5(px, py, m, H, sab, sc, W);
        else if (V == 7) v7(px, py, m, H, sab, sc, W);
        else if (V == 6) v6(px, py, m, H, sab, sc, W);
        else v4(px, py, m, H, sab, sc, W);
#pragma unroll
        for (int i = 0; i < TU; ++i) px[i] += 1e-9 * m[i];   // dependency between repetitions
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < TU; ++i) s += m[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int V>
void run(const char* name, int wps, int H, const Win& W) {
    constexpr bool F32 = V >= 8;
    const int blocks = 256 * wps;
    double* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * 256));
    CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto launch = [&]() {
        if constexpr (F32) hipLaunchKernelGGL(k32<V>, dim3(blocks), dim3(256), 0, 0, out, H, cyc, W);
        else hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, H, cyc, W);
    };
    for (int w = 0; w < 2; ++w) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hc; CHECK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
    const double pairs_per_wave = (double)REP * H * TU;
    const double us = ms * 1000.0 / reps;
    printf("%-26s waves/SIMD %d: kernel %7.1f us  %.2f ns per pair per SIMD  | wave 0: %.2f ticks per pair\n", name, wps, us,
           us * 1000.0 / (pairs_per_wave * wps), (double)hc / pairs_per_wave);
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
    Win W;
    for (int j = 0; j < MAXH; ++j) { W.a[j] = -2.0 * 0.12 * j; W.b[j] = 0.3 * j; W.c[j] = 0.01 * j * j; }
    const int H = 50;
    for (int wps : {1, 2, 3, 4}) {
        run<0>("V0 fmin, load-at-use", wps, H, W);
        run<4>("V4 fmin, unroll 4", wps, H, W);
        run<1>("V1 asm min + rotate", wps, H, W);
        run<2>("V2 asm min, 2 pts/iter", wps, H, W);
        run<3>("V3 scalar loads", wps, H, W);
        run<5>("V5 asm block/pt", wps, H, W);
        run<7>("V7 fmin, 4/grp, ping-pong", wps, H, W);
        run<6>("V6 asm block, 2 pts/iter", wps, H, W);
        run<8>("V8 fp32 screen", wps, H, W);
        run<9>("V9 fp32 screen, packed", wps, H, W);
        run<10>("V10 fp32 screen + group idx", wps, H, W);
        printf("\n");
    }
    return 0;
}
