// How long until a flag stored by one workgroup is seen by polling waves of other workgroups of the same launch
// (different XCDs), gfx950?  Block 0 waits ~20 us, stores the flag (device-scope relaxed atomic) and its s_memrealtime;
// every other block polls with one lane (device-scope relaxed atomic load, s_sleep between polls) and records when it saw it.
//   hipcc --offload-arch=gfx950 -O3 -o flag_latency flag_latency.hip && ./flag_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k(unsigned int* flag, unsigned long long* t, int sleep_arg) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) {}   // 20 us at 100 MHz
        if (threadIdx.x == 0) {
            t[0] = __builtin_amdgcn_s_memrealtime();
            __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    bool seen = false;
    for (int spin = 0; spin < (1 << 18) && !seen; ++spin) {
        unsigned int f = 0;
        if (lane == 0) f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = __builtin_amdgcn_readfirstlane(f) != 0u;
        if (!seen) {
            if (sleep_arg == 1) __builtin_amdgcn_s_sleep(8);
            if (sleep_arg == 2) __builtin_amdgcn_s_sleep(32);
            if (sleep_arg == 3) __builtin_amdgcn_s_sleep(127);
        }
    }
    if (threadIdx.x == 0) t[blockIdx.x] = seen ? __builtin_amdgcn_s_memrealtime() : 0ull;
}

int main() {
    unsigned int* flag; unsigned long long* t;
    for (int fine = 0; fine < 2; ++fine) {
        if (fine) hipExtMallocWithFlags((void**)&flag, 64, hipDeviceMallocFinegrained); else hipMalloc(&flag, 64);
        hipMalloc(&t, 4096 * 8);
        for (int blocks : {2, 34, 257}) {
            for (int sl = 0; sl < 4; ++sl) {
                hipMemset(flag, 0, 64); hipMemset(t, 0, 4096 * 8);
                k<<<blocks, 192>>>(flag, t, sl);
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(blocks);
                hipMemcpy(h.data(), t, blocks * 8, hipMemcpyDeviceToHost);
                double mx = 0, sum = 0; int n = 0, miss = 0;
                for (int i = 1; i < blocks; ++i) {
                    if (!h[i]) { miss++; continue; }
                    const double d = ((double)h[i] - (double)h[0]) / 100.0;
                    mx = std::max(mx, d); sum += d; n++;
                }
                printf("%s memory, %3d polling blocks, sleep %d: seen after mean %.2f us, max %.2f us%s\n", fine ? "fine-grained  " : "coarse-grained",
                       blocks - 1, sl, n ? sum / n : -1.0, mx, miss ? " (some never saw it)" : "");
            }
        }
        hipFree(flag); hipFree(t);
    }
    return 0;
}
