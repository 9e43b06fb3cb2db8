// Workgroup dispatch rate on gfx950: an (almost) empty kernel with the tile kernel's launch shape -- 256 threads, 18.5 KiB of LDS,
// 64 VGPRs' worth of occupancy -- over 14 311 / 60 781 workgroups, and the same work done by 2 048 persistent workgroups.
// hipcc --offload-arch=gfx950 -O3 -o build/ubench_dispatch tools/ubench_dispatch.hip && ./build/ubench_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LDSW, int TOUCH>
__global__ __launch_bounds__(256, 8) void k_empty(uint32_t *out, uint32_t ntiles) {
    __shared__ uint32_t s[LDSW];
    if (TOUCH) { for (int i = threadIdx.x; i < LDSW; i += 256) s[i] = 0; __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.x] = TOUCH ? s[blockIdx.x % LDSW] : blockIdx.x;
}
template <int LDSW, int TOUCH>
__global__ __launch_bounds__(256, 8) void k_persist(uint32_t *out, uint32_t ntiles) {
    __shared__ uint32_t s[LDSW];
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (TOUCH) { for (int i = threadIdx.x; i < LDSW; i += 256) s[i] = 0; __syncthreads(); }
        if (threadIdx.x == 0) out[t] = TOUCH ? s[t % LDSW] : t;
        if (TOUCH) __syncthreads();
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1000.0f;
}
int main() {
    uint32_t *out; CK(hipMalloc(&out, 1 << 20));
    for (uint32_t nt : {2048u, 14311u, 60781u}) {
        printf("tiles %6u:  launch-per-tile  no LDS %.1f us   18.5 KiB declared+cleared %.1f us   4 KiB %.1f us | 2048 persistent: 18.5 KiB cleared %.1f us, untouched %.1f us\n", nt,
               timeit([&] { hipLaunchKernelGGL((k_empty<16, 0>), dim3(nt), dim3(256), 0, 0, out, nt); }, 50),
               timeit([&] { hipLaunchKernelGGL((k_empty<4736, 1>), dim3(nt), dim3(256), 0, 0, out, nt); }, 50),
               timeit([&] { hipLaunchKernelGGL((k_empty<1024, 1>), dim3(nt), dim3(256), 0, 0, out, nt); }, 50),
               timeit([&] { hipLaunchKernelGGL((k_persist<4736, 1>), dim3(2048), dim3(256), 0, 0, out, nt); }, 50),
               timeit([&] { hipLaunchKernelGGL((k_persist<4736, 0>), dim3(2048), dim3(256), 0, 0, out, nt); }, 50));
    }
    return 0;
}
