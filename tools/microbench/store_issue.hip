// What does a wave pay to ISSUE a global store on gfx950?  Each wave stores `n` rows of its 64 lanes (row pitch 512 KiB,
// like the control matrix of the rollout kernel at K = 65 536) and measures s_memtime around the issue loop and around
// issue + completion.  Variants: 8 or 16 bytes per lane, 64-bit vector address or scalar base + 32-bit lane offset.
//   hipcc --offload-arch=gfx950 -O3 -o store_issue store_issue.hip && ./store_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int V>
__global__ __launch_bounds__(128) void k(double* buf, unsigned long long* out, size_t pitch, int nrows) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t col = ((size_t)blockIdx.x * 2 + wv) * 128 + (V >= 2 ? 2 * lane : lane);   // this wave's columns
    double* p = buf + col;
    const double x = (double)lane, y = x + 0.5;
    unsigned long long t0, t1, t2;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 16
    for (int r = 0; r < nrows; ++r) {
        double* q = p + (size_t)r * pitch;
        if (V == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(q), "v"(x) : "memory");
        if (V == 1) {   // scalar row base + lane offset
            double* rowbase = buf + (size_t)r * pitch;
            const unsigned off = (unsigned)(col * 8);
            asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(x), "s"(rowbase) : "memory");
        }
        if (V == 2) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 v = {x, y};
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(q), "v"(v) : "memory");
        }
        if (V == 3) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 v = {x, y};
            double* rowbase = buf + (size_t)r * pitch;
            const unsigned off = (unsigned)(col * 8);
            asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(off), "v"(v), "s"(rowbase) : "memory");
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    if (lane == 0) {
        out[(blockIdx.x * 2 + wv) * 2 + 0] = t1 - t0;
        out[(blockIdx.x * 2 + wv) * 2 + 1] = t2 - t0;
    }
}

template <int V>
void run(const char* name, double* buf, unsigned long long* out, int blocks, int nrows) {
    const size_t pitch = 65536 * 2;   // doubles: 1 MiB rows, room for 16 B per lane
    std::vector<unsigned long long> h(blocks * 4);
    for (int rep = 0; rep < 3; ++rep) k<V><<<blocks, 128>>>(buf, out, pitch, nrows);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < blocks * 2; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
    const int bytes = (V >= 2 ? 16 : 8) * 64;
    printf("%-34s blocks %4d rows %3d: issue %6.1f cyc/store, issue+complete %6.1f cyc/store (%5.2f B/cyc/wave)\n", name, blocks, nrows,
           a / (blocks * 2) / nrows, b / (blocks * 2) / nrows, bytes / (b / (blocks * 2) / nrows));
}

int main() {
    double* buf; unsigned long long* out;
    hipMalloc(&buf, (size_t)65536 * 2 * 8 * 128);   // 128 rows of 1 MiB
    hipMalloc(&out, 4096 * 4 * 8);
    for (int blocks : {256, 512, 1024}) {   // 1, 2, 4 workgroups of 2 waves per CU
        for (int nrows : {16, 64}) {
            run<0>("dwordx2, 64-bit vaddr", buf, out, blocks, nrows);
            run<1>("dwordx2, saddr + voffset", buf, out, blocks, nrows);
            run<2>("dwordx4, 64-bit vaddr", buf, out, blocks, nrows);
            run<3>("dwordx4, saddr + voffset", buf, out, blocks, nrows);
        }
    }
    return 0;
}
