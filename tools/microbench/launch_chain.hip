// What a dependent launch costs on this box: N back-to-back launches on one stream of (a) an empty kernel, (b) a kernel that
// reads its arguments and does one load -> store, (c) a kernel with two DEPENDENT loads (pointer chase through L2) and a store --
// 20 workgroups of 256 threads, as the small-batch path's launches.      hipcc --offload-arch=gfx950 -O3 launch_chain.hip -o launch_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_empty() {}
__global__ void k_one(const float* __restrict__ a, float* __restrict__ o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + 1.f;
}
__global__ void k_two(const int* __restrict__ idx, const float* __restrict__ a, float* __restrict__ o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[idx[i]] + 1.f;
}
__global__ void k_three(const int* __restrict__ idx, const int* __restrict__ idx2, const float* __restrict__ a, float* __restrict__ o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[idx2[idx[i]]] + 1.f;
}
int main() {
    const int n = 20 * 256, N = 2000;
    float *a, *o; int *idx, *idx2;
    hipMalloc(&a, n * 4); hipMalloc(&o, n * 4); hipMalloc(&idx, n * 4); hipMalloc(&idx2, n * 4);
    std::vector<int> h(n); for (int i = 0; i < n; ++i) h[i] = (i * 37) % n;
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(idx2, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(a, 0, n * 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 5; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, s);
            for (int i = 0; i < N; ++i) {
                if (which == 0) hipLaunchKernelGGL(k_empty, dim3(20), dim3(256), 0, s);
                else if (which == 1) hipLaunchKernelGGL(k_one, dim3(20), dim3(256), 0, s, (const float*)a, o, n);
                else if (which == 2) hipLaunchKernelGGL(k_two, dim3(20), dim3(256), 0, s, (const int*)idx, (const float*)a, o, n);
                else if (which == 3) hipLaunchKernelGGL(k_three, dim3(20), dim3(256), 0, s, (const int*)idx, (const int*)idx2, (const float*)a, o, n);
                else hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s);
            }
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.2f us per launch\n", which == 0 ? "empty, 20 workgroups" : which == 1 ? "load -> store" : which == 2 ? "load -> load -> store"
                            : which == 3 ? "load -> load -> load -> store" : "empty, 256 workgroups", ms * 1000 / N);
        }
    }
    return 0;
}
