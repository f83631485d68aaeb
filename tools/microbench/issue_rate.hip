// Microbenchmark of the CDNA4 SIMD issue model: cycles per instruction of dependent / independent VALU,
// packed-fp32 VALU, bf16 MFMA, and MFMA interleaved with VALU, at 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int N = 65536;

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
    float a[8];
    f32x2 p[8];
    f32x4 acc[8];
    u32x4 fa = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, fb = fa;
    for (int i = 0; i < 8; ++i) { a[i] = seed + i; p[i] = f32x2{seed + i, seed - i}; acc[i] = f32x4{0, 0, 0, 0}; }
    const float c0 = seed * 0.5f, c1 = seed * 0.25f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < N; ++it) {
        if (MODE == 0) {            // 8 independent chains of scalar fma
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], c0, c1);
        } else if (MODE == 1) {     // 1 dependent chain of 8 scalar fma
#pragma unroll
            for (int i = 0; i < 8; ++i) a[0] = __builtin_fmaf(a[0], c0, c1);
        } else if (MODE == 2) {     // 8 independent packed fma
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], (f32x2)(c0), (f32x2)(c1));
        } else if (MODE == 3) {     // 8 independent MFMA
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[i], 0, 0, 0);
        } else if (MODE == 4) {     // 8 x (MFMA + 3 independent scalar fma)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a[(3 * i) & 7] = __builtin_fmaf(a[(3 * i) & 7], c0, c1);
                a[(3 * i + 1) & 7] = __builtin_fmaf(a[(3 * i + 1) & 7], c0, c1);
                a[(3 * i + 2) & 7] = __builtin_fmaf(a[(3 * i + 2) & 7], c0, c1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 5) {     // 8 x (MFMA + 3 packed fma)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                p[(3 * i) & 7] = __builtin_elementwise_fma(p[(3 * i) & 7], (f32x2)(c0), (f32x2)(c1));
                p[(3 * i + 1) & 7] = __builtin_elementwise_fma(p[(3 * i + 1) & 7], (f32x2)(c0), (f32x2)(c1));
                p[(3 * i + 2) & 7] = __builtin_elementwise_fma(p[(3 * i + 2) & 7], (f32x2)(c0), (f32x2)(c1));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 6) {     // 8 x (MFMA + 1 scalar fma)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a[i] = __builtin_fmaf(a[i], c0, c1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 7) {     // 8 x (MFMA + 6 scalar fma)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 6; ++j) a[(6 * i + j) & 7] = __builtin_fmaf(a[(6 * i + j) & 7], c0, c1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 8) {     // 1 dependent chain of 8 packed fma
#pragma unroll
            for (int i = 0; i < 8; ++i) p[0] = __builtin_elementwise_fma(p[0], (f32x2)(c0), (f32x2)(c1));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int insts_per_iter, int mfma_per_iter = 0) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 8);
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int threads = 256 * waves_per_simd;     // one workgroup per CU, waves spread over the 4 SIMDs
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c = 0;
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        // wall clock: ns per loop iteration of one wave, and (for the MFMA modes) the whole-chip MFMA rate
        const double ns_iter = ms * 1e6 / N;
        const double tflops = mfma_per_iter * 16384.0 * N * (threads / 64) * 256 / (ms * 1e-3) / 1e12;
        printf("%-36s %d wave/SIMD: %7.2f ticks/iter (%5.2f/instr)  wall %7.1f ns/iter  tick = %.3f ns  MFMA %.0f TFLOP/s\n", name,
               waves_per_simd, (double)c / N, (double)c / N / insts_per_iter, ns_iter, ns_iter / ((double)c / N), tflops);
    }
}

int main() {
    run<0>("8 independent v_fma_f32", 8);
    run<1>("8 dependent v_fma_f32", 8);
    run<2>("8 independent v_pk_fma_f32", 8);
    run<8>("8 dependent v_pk_fma_f32", 8);
    run<3>("8 independent mfma_16x16x32_bf16", 8, 8);
    run<6>("8 x (mfma + 1 v_fma_f32)", 16, 8);
    run<4>("8 x (mfma + 3 v_fma_f32)", 32, 8);
    run<7>("8 x (mfma + 6 v_fma_f32)", 56, 8);
    run<5>("8 x (mfma + 3 v_pk_fma_f32)", 32, 8);
    return 0;
}
