// Microbenchmark: how fast can every CU stream the SAME buffer (weights-like) out of L2 with 16-byte
// per-lane loads, as a function of the bytes in flight per wave and of the buffer size?
//   hipcc --offload-arch=gfx950 -O3 -o l2_stream l2_stream.hip && ./l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(512, 2) void stream_kernel(const u32x4* __restrict__ buf, size_t n_frag, int reps,
                                                        unsigned* out, int rot_frags) {
    // fragment = 64 lanes x 16 B = 1 KiB; the 8 waves of a workgroup read 8 consecutive fragments per step
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t steps = n_frag / 8;
    const size_t rot = rot_frags ? ((size_t)(blockIdx.x >> 3) * rot_frags) % steps : 0;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (size_t s0 = 0; s0 < steps; s0 += U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                size_t s = s0 + u + rot;
                if (s >= steps) s -= steps;
                v[u] = buf[(s * 8 + w) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

template <int U>
static double run(const u32x4* buf, size_t bytes, int grid, int rot, unsigned* out) {
    const size_t n_frag = bytes / 1024;
    const int reps = (int)(((size_t)64 << 20) / bytes) + 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stream_kernel<U>, dim3(grid), dim3(512), 0, 0, buf, n_frag, 1, out, rot);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(stream_kernel<U>, dim3(grid), dim3(512), 0, 0, buf, n_frag, reps, out, rot);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)(n_frag / 8 / U * U) * 8 * 1024 * reps * grid;
    return total / (ms * 1e-3) / 1e9;       // GB/s aggregate over all workgroups
}

int main() {
    const size_t max_bytes = (size_t)64 << 20;
    u32x4* buf; unsigned* out;
    hipMalloc(&buf, max_bytes); hipMalloc(&out, 4);
    hipMemset(buf, 1, max_bytes);
    const size_t sizes[] = {(size_t)1 << 20, (size_t)3500 << 10, (size_t)20 << 20, (size_t)64 << 20};
    for (int grid : {256, 512}) {
        for (size_t bytes : sizes) {
            for (int rot : {0, 37}) {
                double g1 = run<1>(buf, bytes, grid, rot, out), g2 = run<2>(buf, bytes, grid, rot, out),
                       g4 = run<4>(buf, bytes, grid, rot, out), g8 = run<8>(buf, bytes, grid, rot, out),
                       g16 = run<16>(buf, bytes, grid, rot, out);
                printf("grid %3d buf %6.1f MiB rot %2d :  U=1 %7.0f  U=2 %7.0f  U=4 %7.0f  U=8 %7.0f  U=16 %7.0f GB/s aggregate"
                       "  (per CU at U=8: %5.1f GB/s)\n", grid, bytes / 1048576.0, rot, g1, g2, g4, g8, g16, g8 / 256);
            }
        }
    }
    return 0;
}
