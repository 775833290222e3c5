// Micro-benchmarks behind DESIGN.md "Measured instruction rates": issue cost of the fp64 / int ops the rollout
// kernel is made of, per wave and per SIMD occupancy.  Build: hipcc --offload-arch=gfx950 -O3 -o fp64_rates fp64_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NCH = 8;      // independent chains per lane
constexpr int ITERS = 2048; // loop iterations, each NCH ops

enum Op { FMA64, MIN64, MIN64_ASM, ADD64, MUL64, MAX64_ASM, CMPSEL64, FMA32, MAD64_32, MULHI32, XOR32, SQRT64, RCP64, FMA64_MIN_ASM };

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, unsigned long long* cyc) {
    double v[NCH];
    float f[NCH];
    unsigned int u[NCH];
    unsigned long long w[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) { v[i] = a + i + threadIdx.x * 1e-3; f[i] = (float)v[i]; u[i] = threadIdx.x * 2654435761u + i; w[i] = u[i]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (OP == FMA64) v[i] = __builtin_fma(v[i], a, b);
            else if (OP == MIN64) v[i] = __builtin_fmin(v[i] + 0.0 * it, b + i);   // includes whatever the compiler adds
            else if (OP == MIN64_ASM) asm volatile("v_min_f64 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b));
            else if (OP == MAX64_ASM) asm volatile("v_max_f64 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b));
            else if (OP == ADD64) asm volatile("v_add_f64 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(b));
            else if (OP == MUL64) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(a));
            else if (OP == CMPSEL64) v[i] = (v[i] < b + it) ? v[i] : b + it;
            else if (OP == FMA32) f[i] = __builtin_fmaf(f[i], (float)a, (float)b);
            else if (OP == MAD64_32) w[i] = (unsigned long long)(unsigned int)w[i] * 0xD2511F53ull + (w[i] >> 32);
            else if (OP == MULHI32) u[i] = __umulhi(u[i], 0xD2511F53u) ^ u[i];
            else if (OP == XOR32) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(0x9E3779B9u + it));
            else if (OP == SQRT64) asm volatile("v_sqrt_f64 %0, %1" : "=v"(v[i]) : "v"(v[i]));
            else if (OP == RCP64) asm volatile("v_rcp_f64 %0, %1" : "=v"(v[i]) : "v"(v[i]));
            else if (OP == FMA64_MIN_ASM) { double g = __builtin_fma(a, v[i], __builtin_fma(b, v[(i + 1) % NCH], a)); asm volatile("v_min_f64 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(g)); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; float fs = 0; unsigned int us = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) { s += v[i]; fs += f[i]; us ^= u[i] ^ (unsigned int)w[i]; }
    out[blockIdx.x * 256 + threadIdx.x] = s + fs + us;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_simd, int ops_per_iter_elem = 1) {
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 wave per SIMD)
    double* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * 256));
    CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 0.5, cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 0.5, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hc; CHECK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
    const double us = ms * 1000.0 / reps;
    const double wave_ops_per_simd = (double)ITERS * NCH * waves_per_simd * ops_per_iter_elem;
    // memtime ticks at 100 MHz (s_memtime = shader clock per the guide: "tick = shader cycle")
    printf("%-14s waves/SIMD %d : %8.1f us  -> %6.2f ns per wave-op per SIMD  (memtime delta %llu ticks => %.2f ticks/op one wave)\n",
           name, waves_per_simd, us, us * 1000.0 / wave_ops_per_simd, hc, (double)hc / (ITERS * NCH * ops_per_iter_elem));
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
    for (int wps : {1, 2, 4}) {
        run<FMA64>("v_fma_f64", wps);
        run<MIN64_ASM>("v_min_f64", wps);
        run<MAX64_ASM>("v_max_f64", wps);
        run<ADD64>("v_add_f64", wps);
        run<MUL64>("v_mul_f64", wps);
        run<MIN64>("fmin+add(C)", wps);
        run<CMPSEL64>("cmp+sel(C)", wps);
        run<FMA64_MIN_ASM>("2fma+min", wps, 3);
        run<FMA32>("v_fma_f32", wps);
        run<MAD64_32>("mad_u64_u32(C)", wps);
        run<MULHI32>("mulhi+xor(C)", wps);
        run<XOR32>("v_xor_b32", wps);
        run<SQRT64>("v_sqrt_f64", wps);
        run<RCP64>("v_rcp_f64", wps);
        printf("\n");
    }
    return 0;
}
