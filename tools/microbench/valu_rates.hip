// Issue cost of single VALU instructions on gfx950, one wave per SIMD and four: cycles (s_memtime ticks) per instruction for
// eight independent chains.  Question behind it: is v_pk_fma_f32 one issue slot (two FMAs for the price of one) or two?
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { FMA32 = 0, PKFMA32, PKMUL32, FMA64, MUL64, MADU64, CNDMASK, CVT, BITOP3, MIN64, CNDMASK64, MOV, ADDU32, ADD64, CMP64, FMAAK32, MUL32, CND_E64_VCC, CMP_CND_VCC, CMP_CND_SGPR, CND_E32_NODEP, NKIND };
static const char* names[NKIND] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_fma_f64", "v_mul_f64", "v_mad_u64_u32", "v_cndmask_b32", "v_cvt_f64_f32",
                                   "v_bitop3_b32", "v_min_f64", "v_cndmask e64 sgpr", "v_mov_b32", "v_add_u32", "v_add_f64", "v_cmp_lt_f64", "v_fmaak_f32", "v_mul_f32", "cndmask e64 vcc", "cmp+cndmask vcc", "cmp+cndmask sgpr", "cndmask vcc nodep"};

template <int KIND>
__device__ __forceinline__ double stream(int iters, int l) {
    if constexpr (KIND == FMA32) {
        float x[8]; for (int j = 0; j < 8; ++j) x[j] = l + j;
        const float m = 0.9999f, a = 1e-6f;
#pragma unroll 8
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(a));
        return x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7];
    } else if constexpr (KIND == PKFMA32 || KIND == PKMUL32) {
        f32x2 x[8]; for (int j = 0; j < 8; ++j) x[j] = f32x2{(float)(l + j), (float)(l - j)};
        const f32x2 m = {0.9999f, 0.9998f}, a = {1e-6f, 2e-6f};
#pragma unroll 8
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (KIND == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(a));
                else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(m));
            }
        float r = 0; for (int j = 0; j < 8; ++j) r += x[j][0] + x[j][1];
        return r;
    } else if constexpr (KIND == FMA64 || KIND == MUL64 || KIND == MIN64 || KIND == ADD64 || KIND == CMP64) {
        double x[8]; for (int j = 0; j < 8; ++j) x[j] = l + j;
        const double m = 0.999999, a = 1e-9;
#pragma unroll 8
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (KIND == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(a));
                else if constexpr (KIND == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[j]) : "v"(m));
                else if constexpr (KIND == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[j]) : "v"(m));
                else if constexpr (KIND == CMP64) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : "+v"(x[j]) : "v"(m) : "vcc");
                else asm volatile("v_min_f64 %0, %0, %1" : "+v"(x[j]) : "v"(m));
            }
        return x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7];
    } else if constexpr (KIND == MADU64) {
        unsigned long long x[8]; for (int j = 0; j < 8; ++j) x[j] = l + j;
        const unsigned m = 0xD2511F53u;
#pragma unroll 8
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned lo = (unsigned)x[j];
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[j]) : "v"(lo), "v"(m) : "vcc");
            }
        return (double)(x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[4] ^ x[5] ^ x[6] ^ x[7]);
    } else if constexpr (KIND == CNDMASK || KIND == BITOP3 || KIND == CNDMASK64 || KIND == MOV || KIND == ADDU32 || KIND == FMAAK32 || KIND == MUL32) {
        unsigned x[8]; for (int j = 0; j < 8; ++j) x[j] = l + j;
        const unsigned m = 0x9E3779B9u + l;
        unsigned long long mask = 0x5555555555555555ull + iters;
        asm volatile("s_mov_b64 vcc, %0" :: "s"(mask) : "vcc");
#pragma unroll 8
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(m));
                else if constexpr (KIND == CNDMASK64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "s"(mask));
                else if constexpr (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(x[j]) : "v"(m));
                else if constexpr (KIND == ADDU32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[j]) : "v"(m));
                else if constexpr (KIND == FMAAK32) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7ff972" : "+v"(x[j]) : "v"(m));
                else if constexpr (KIND == MUL32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(m));
                else asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(x[j]) : "v"(m));
            }
        return (double)(x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7]);
    } else if constexpr (KIND == CND_E64_VCC || KIND == CMP_CND_VCC || KIND == CMP_CND_SGPR || KIND == CND_E32_NODEP) {
        unsigned x[8]; for (int j = 0; j < 8; ++j) x[j] = l + j;
        unsigned y[8]; for (int j = 0; j < 8; ++j) y[j] = 0;
        const unsigned m = 0x9E3779B9u + l;
        unsigned long long mask = 0x5555555555555555ull + iters;
        asm volatile("s_mov_b64 vcc, %0" :: "s"(mask) : "vcc");
#pragma unroll 8
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (KIND == CND_E64_VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(m));
                else if constexpr (KIND == CMP_CND_VCC) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(m) : "vcc");
                else if constexpr (KIND == CMP_CND_SGPR) asm volatile("v_cmp_lt_u32_e64 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "s"(mask));
                else asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(y[j]) : "v"(x[j]), "v"(m));
            }
        return (double)(x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7] + y[0] + y[1] + y[2] + y[3] + y[4] + y[5] + y[6] + y[7]);
    } else if constexpr (KIND == CVT) {
        double x[8]; float y[8]; for (int j = 0; j < 8; ++j) y[j] = l + j;
#pragma unroll 8
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[j]) : "v"(y[j]));
        return x[0] + x[1] + x[2] + x[3] + x[4] + x[5] + x[6] + x[7];
    }
    return 0.0;
}

template <int KIND, int THREADS>
__global__ __launch_bounds__(THREADS) void k(double* out, unsigned long long* cyc, int iters) {
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const double r = stream<KIND>(iters, l);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(r) : "memory");
    out[blockIdx.x * THREADS + threadIdx.x] = r;
    if (l == 0) cyc[blockIdx.x * 16 + wv] = t1 - t0;
}

template <int KIND, int THREADS>
double run(double* out, unsigned long long* cyc) {
    const int blocks = 256, iters = 20000;
    std::vector<unsigned long long> h(blocks * 16);
    hipMemset(cyc, 0, blocks * 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, THREADS><<<blocks, THREADS>>>(out, cyc, iters);
    hipEventRecord(e0);
    k<KIND, THREADS><<<blocks, THREADS>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), cyc, blocks * 128, hipMemcpyDeviceToHost);
    double a = 0; int n = 0;
    for (auto v : h) if (v) { a += v; n++; }
    const double per_wave = iters * 8.0;
    printf("  %2d waves/SIMD: %6.2f ticks/instr/wave, %7.3f ns per instruction per SIMD", THREADS / 256, a / n / per_wave, ms * 1e6 / (per_wave * (THREADS / 256)));
    return ms;
}

template <int KIND>
void both(double* out, unsigned long long* cyc) {
    printf("%-16s", names[KIND]);
    run<KIND, 256>(out, cyc);
    run<KIND, 1024>(out, cyc);
    printf("\n");
}

int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 128);
    both<FMA32>(out, cyc); both<PKFMA32>(out, cyc); both<PKMUL32>(out, cyc); both<FMA64>(out, cyc); both<MUL64>(out, cyc); both<MIN64>(out, cyc);
    both<MADU64>(out, cyc); both<CNDMASK>(out, cyc); both<CNDMASK64>(out, cyc); both<BITOP3>(out, cyc); both<CVT>(out, cyc);
    both<MOV>(out, cyc); both<ADDU32>(out, cyc); both<ADD64>(out, cyc); both<CMP64>(out, cyc); both<FMAAK32>(out, cyc); both<MUL32>(out, cyc);
    both<CND_E64_VCC>(out, cyc); both<CND_E32_NODEP>(out, cyc); both<CMP_CND_VCC>(out, cyc); both<CMP_CND_SGPR>(out, cyc);
    return 0;
}
