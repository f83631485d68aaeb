// Microbenchmark: sustained dense bf16 MFMA rate of the whole chip over seconds (power / clock limited),
// against the 2.5 PFLOP/s headline peak that a 30 ms burst reaches (issue_rate.hip: 2462 TFLOP/s).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_sustained mfma_sustained.hip && ./mfma_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned seed) {
    // operands with non-trivial bit patterns (zeros would toggle nothing and draw less power)
    u32x4 fa, fb;
    for (int i = 0; i < 4; ++i) { fa[i] = 0x3f803f80u ^ ((threadIdx.x * 2654435761u + i * 40503u + seed) & 0x007f007fu); fb[i] = fa[i] ^ 0x00150015u; }
    f32x4 a16[8];
    f32x16 a32[2];
    for (int i = 0; i < 8; ++i) a16[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) a32[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                a16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), a16[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a32[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), a32[i & 1], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a16[i][0] + a16[i][3];
    for (int i = 0; i < 2; ++i) s += a32[i][0] + a32[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE>
void run(const char* name, double flop_per_iter_per_wave) {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 1 << 20;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 0, 0, out, iters, (unsigned)rep);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double tflops = flop_per_iter_per_wave * iters * 8 * 256 / (ms * 1e-3) / 1e12;
        printf("%-28s launch %d: %8.1f ms  %7.0f TFLOP/s\n", name, rep, ms, tflops);
    }
}

int main() {
    run<0>("mfma_f32_16x16x32_bf16", 8 * 16384.0);
    run<1>("mfma_f32_32x32x16_bf16", 4 * 32768.0);
    return 0;
}
