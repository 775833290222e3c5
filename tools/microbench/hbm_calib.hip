// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 with THIS code base's access pattern: 8 bytes per lane,
// k-fastest (a wave instruction moves one contiguous 512-byte run).  Writes / reads a known byte count so that the
// counter-to-bytes factor can be read off (MI355X_MICROARCH.md, HBM: other widths than 16 B/lane are uncalibrated).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void calib_store8(double* p, size_t n, double v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + i;
}
__global__ void calib_load8(const double* p, size_t n, double* out) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 12345.678) out[0] = s;
}

int main() {
    const size_t n = (size_t)64 << 20;   // 64 Mi doubles = 512 MiB: larger than the 256 MiB Infinity Cache
    double *p, *o;
    CHECK(hipMalloc(&p, n * 8));
    CHECK(hipMalloc(&o, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(calib_store8, dim3(2048), dim3(256), 0, 0, p, n, 1.0);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("store8: %zu bytes in %.1f us = %.2f TB/s\n", n * 8, ms * 1e3, n * 8 / (ms * 1e-3) / 1e12);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(calib_load8, dim3(2048), dim3(256), 0, 0, p, n, o);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("load8 : %zu bytes in %.1f us = %.2f TB/s\n", n * 8, ms * 1e3, n * 8 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
