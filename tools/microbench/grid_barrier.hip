// What a chip-wide barrier costs inside ONE kernel on this box, beside the dependent launch of launch_chain.hip: G co-resident
// workgroups of 256 threads run N rounds of {read what ANOTHER workgroup wrote last round -> store -> barrier}.  The barrier is a
// monotonic counter: release add by one thread per workgroup, acquire spin (agent scope: the compiler writes back / invalidates
// the XCD's L2 around it, which is what makes the other XCDs' stores visible).  Spins are bounded: a lost workgroup ends the
// kernel with an error flag instead of hanging the box.      hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { *err = 1; break; }
        }
    }
    __syncthreads();
    return true;
}

// MODE 0: barrier alone; 1: load (neighbour's value of last round) -> store -> barrier; 2: load -> load (pointer chase) -> store
// STRIDE 8: eight times the workgroups are launched and only those with blockIdx % 8 == 0 take part -- workgroups go to the XCDs
// round-robin, so the participants share ONE XCD (one L2); xcc[] records HW_REG_XCC_ID of every participant.
template <int MODE, int STRIDE>
__global__ void __launch_bounds__(256) k_rounds(float* buf, const int* __restrict__ idx, unsigned* ctr, int* err, int rounds, int* xcc) {
    if (blockIdx.x % STRIDE) return;
    const int G = gridDim.x / STRIDE, wg = blockIdx.x / STRIDE, tid = threadIdx.x, n = G * 256;
    if (tid == 0) xcc[wg] = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20);
    for (int it = 0; it < rounds; ++it) {
        if (MODE >= 1) {
            const float* src = buf + (it & 1) * n;
            float* dst = buf + ((it + 1) & 1) * n;
            int j = ((wg + 1) % G) * 256 + tid;
            if (MODE == 2) j = idx[j];
            // (volatile: the values change under the kernel; after the acquire the caches hold no stale line)
            const float v = *(volatile const float*)(src + j);
            dst[wg * 256 + tid] = v + 1.f;
        }
        grid_barrier(ctr, (unsigned)(it + 1) * G, err);
        if (*(volatile int*)err) return;
    }
}

int main() {
    const int N = 2000;
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int G : {6, 24, 32, 64, 128, 256}) {
      for (int stride : {1, 8}) {
        if (stride == 8 && G > 32) continue;
        const int n = G * 256;
        float* buf; int* idx; unsigned* ctr; int* err; int* xcc;
        hipMalloc(&buf, 2 * n * 4); hipMalloc(&idx, n * 4); hipMalloc(&ctr, 4); hipMalloc(&err, 4); hipMalloc(&xcc, G * 4);
        std::vector<int> h(n); for (int i = 0; i < n; ++i) h[i] = (i * 37 + 11) % n;
        hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 3; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipMemsetAsync(buf, 0, 2 * n * 4, s); hipMemsetAsync(ctr, 0, 4, s); hipMemsetAsync(err, 0, 4, s);
                hipEventRecord(e0, s);
                if (stride == 1) {
                    if (mode == 0) hipLaunchKernelGGL((k_rounds<0, 1>), dim3(G), dim3(256), 0, s, buf, (const int*)idx, ctr, err, N, xcc);
                    else if (mode == 1) hipLaunchKernelGGL((k_rounds<1, 1>), dim3(G), dim3(256), 0, s, buf, (const int*)idx, ctr, err, N, xcc);
                    else hipLaunchKernelGGL((k_rounds<2, 1>), dim3(G), dim3(256), 0, s, buf, (const int*)idx, ctr, err, N, xcc);
                } else {
                    if (mode == 0) hipLaunchKernelGGL((k_rounds<0, 8>), dim3(8 * G), dim3(256), 0, s, buf, (const int*)idx, ctr, err, N, xcc);
                    else if (mode == 1) hipLaunchKernelGGL((k_rounds<1, 8>), dim3(8 * G), dim3(256), 0, s, buf, (const int*)idx, ctr, err, N, xcc);
                    else hipLaunchKernelGGL((k_rounds<2, 8>), dim3(8 * G), dim3(256), 0, s, buf, (const int*)idx, ctr, err, N, xcc);
                }
                hipEventRecord(e1, s); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
            std::vector<float> out(n); hipMemcpy(out.data(), buf + (N & 1) * n, n * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            if (mode >= 1) for (int i = 0; i < n; ++i) bad += out[i] != (float)N;
            std::vector<int> hx(G); hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost);
            int xmask = 0; for (int i = 0; i < G; ++i) xmask |= 1 << (hx[i] & 15);
            printf("G=%3d %s XCDs 0x%02x %s: %.2f us per round%s%s\n", G, stride == 1 ? "chip-wide" : "one XCD  ", xmask, mode == 0 ? "barrier alone" : mode == 1 ? "load -> store -> barrier" : "load -> load -> store -> barrier",
                   ms * 1000 / N, herr ? "  (SPIN BOUND HIT)" : "", bad ? "  (STALE VALUES READ)" : "");
        }
        hipFree(buf); hipFree(idx); hipFree(ctr); hipFree(err); hipFree(xcc);
      }
    }
    return 0;
}
