// How fast can the rollout's store pattern be written at all?  K lanes (one sample each, k fastest) write R rows of 8 bytes
// per lane at a row pitch of K doubles -- the u / xs / ys layout of DESIGN.md section 4 -- with no arithmetic, against a plain
// linear fill of the same number of bytes.  Build: hipcc --offload-arch=gfx950 -O3 -o hbm_write hbm_write.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_rows(double* out, int K, int R) {
    const int k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    // the rows of a block of samples are dealt to the block's waves
    for (int r = w; r < R; r += WAVES) {
        double* p = out + (size_t)r * K + k;
        const double v = (double)(r + k);
        if (NT) __builtin_nontemporal_store(v, p);
        else *p = v;
    }
}

typedef double d2 __attribute__((ext_vector_type(2)));
template <int NT>
__global__ __launch_bounds__(256) void k_fill(d2* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const d2 v = {(double)i, 1.0};
        if (NT) __builtin_nontemporal_store(v, out + i);
        else out[i] = v;
    }
}

template <class F>
float time_us(F&& launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

int main() {
    const int K = 65536, R = 98 + 100;   // C2: 98 control rows + 2 x 50 state rows
    const size_t bytes = (size_t)K * R * 8;
    double* buf; CHECK(hipMalloc(&buf, bytes * 4));   // four copies: consecutive launches do not hit the same lines
    int which = 0;
    auto dst = [&]() { which = (which + 1) & 3; return buf + (size_t)which * K * R; };
    printf("bytes per launch: %.1f MB (K = %d, %d rows)\n", bytes / 1e6, K, R);
    float t;
    t = time_us([&]() { hipLaunchKernelGGL((k_rows<0, 1>), dim3(K / 64), dim3(64), 0, 0, dst(), K, R); }, 50);
    printf("rows, 1 wave per 64 samples, plain stores      : %7.1f us  %6.2f TB/s\n", t, bytes / t / 1e6);
    t = time_us([&]() { hipLaunchKernelGGL((k_rows<1, 1>), dim3(K / 64), dim3(64), 0, 0, dst(), K, R); }, 50);
    printf("rows, 1 wave per 64 samples, nontemporal       : %7.1f us  %6.2f TB/s\n", t, bytes / t / 1e6);
    t = time_us([&]() { hipLaunchKernelGGL((k_rows<0, 3>), dim3(K / 64), dim3(192), 0, 0, dst(), K, R); }, 50);
    printf("rows, 3 waves per 64 samples, plain stores     : %7.1f us  %6.2f TB/s\n", t, bytes / t / 1e6);
    t = time_us([&]() { hipLaunchKernelGGL((k_rows<1, 3>), dim3(K / 64), dim3(192), 0, 0, dst(), K, R); }, 50);
    printf("rows, 3 waves per 64 samples, nontemporal      : %7.1f us  %6.2f TB/s\n", t, bytes / t / 1e6);
    t = time_us([&]() { hipLaunchKernelGGL((k_fill<0>), dim3(256 * 8), dim3(256), 0, 0, (d2*)dst(), bytes / 16); }, 50);
    printf("linear fill, 16 B per lane, plain              : %7.1f us  %6.2f TB/s\n", t, bytes / t / 1e6);
    t = time_us([&]() { hipLaunchKernelGGL((k_fill<1>), dim3(256 * 8), dim3(256), 0, 0, (d2*)dst(), bytes / 16); }, 50);
    printf("linear fill, 16 B per lane, nontemporal        : %7.1f us  %6.2f TB/s\n", t, bytes / t / 1e6);
    const size_t big = (size_t)1 << 30;
    double* bb; CHECK(hipMalloc(&bb, big));
    t = time_us([&]() { hipLaunchKernelGGL((k_fill<0>), dim3(256 * 16), dim3(256), 0, 0, (d2*)bb, big / 16); }, 10);
    printf("linear fill of 1 GiB, plain                    : %7.1f us  %6.2f TB/s\n", t, big / t / 1e6);
    t = time_us([&]() { hipLaunchKernelGGL((k_fill<1>), dim3(256 * 16), dim3(256), 0, 0, (d2*)bb, big / 16); }, 10);
    printf("linear fill of 1 GiB, nontemporal              : %7.1f us  %6.2f TB/s\n", t, big / t / 1e6);
    return 0;
}
