// Microbenchmark: board power of single-resource stress loops, each held for ~3 s so that rocm-smi can
// sample it (tools/microbench/power_modes.sh):  mfma | lds (ds_read_b128) | l2 (16-byte loads of a 3 MiB
// buffer by every CU) | valu (v_fma_f32) | idle-ish (s_sleep).  Prints achieved rate per mode.
//   hipcc --offload-arch=gfx950 -O3 -o power_modes power_modes.hip && ./power_modes <mode> <seconds>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ZB: the B operand's columns 8..15 (lanes with lane & 8) are zeros -- what padding token slots would cost if they were zeroed
template <int ZB>
__global__ __launch_bounds__(512) void k_mfma(float* out, int iters) {
    u32x4 fa, fb;
    for (int i = 0; i < 4; ++i) { fa[i] = 0x3f803f80u ^ ((threadIdx.x * 2654435761u + i * 40503u) & 0x007f007fu); fb[i] = fa[i] ^ 0x00150015u; }
    if (ZB && (threadIdx.x & 8)) fb = u32x4{0, 0, 0, 0};
    f32x4 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            a[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), a[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// the other MFMA shapes at the same occupancy (round 3): 32x32x16 bf16 (half the A/B operand reads per FLOP), 16x16x16 bf16
// (the half k-step of fused.hip), 16x16x32 f16 (BESO_PREC_FP16)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_mfma32(float* out, int iters) {
    u32x4 fa, fb;
    for (int i = 0; i < 4; ++i) { fa[i] = 0x3f803f80u ^ ((threadIdx.x * 2654435761u + i * 40503u) & 0x007f007fu); fb[i] = fa[i] ^ 0x00150015u; }
    f32x16 a[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) a[i][j] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), a[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i) s += a[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k_mfma16(float* out, int iters) {
    u32x4 fa, fb;
    for (int i = 0; i < 4; ++i) { fa[i] = 0x3f803f80u ^ ((threadIdx.x * 2654435761u + i * 40503u) & 0x007f007fu); fb[i] = fa[i] ^ 0x00150015u; }
    const s16x4 ha = __builtin_bit_cast(s16x4, uint2{fa[0], fa[1]}), hb = __builtin_bit_cast(s16x4, uint2{fb[0], fb[1]});
    f32x4 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ha, hb, a[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k_mfmaf16(float* out, int iters) {
    u32x4 fa, fb;
    for (int i = 0; i < 4; ++i) { fa[i] = 0x3c003c00u ^ ((threadIdx.x * 2654435761u + i * 40503u) & 0x03ff03ffu); fb[i] = fa[i] ^ 0x01550155u; }
    f32x4 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            a[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa), __builtin_bit_cast(f16x8, fb), a[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void k_lds(float* out, int iters) {
    extern __shared__ u32x4 lds[];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = u32x4{(unsigned)i * 2654435761u, (unsigned)i, 7u * i, 3u * i};
    __syncthreads();
    u32x4 acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= lds[((it + j * 8 + w) & 127) * 64 + lane];     // 128 KiB window, lane-linear b128
    }
    out[blockIdx.x * 512 + threadIdx.x] = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
}

__global__ __launch_bounds__(512) void k_l2(float* out, const u32x4* __restrict__ buf, int n_frag, int iters) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    int s = 0;
    for (int it = 0; it < iters; ++it) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = buf[(size_t)((s + j) * 8 + w) * 64 + lane]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
        s += 8;
        if ((s + 8) * 8 > n_frag) s = 0;
    }
    out[blockIdx.x * 512 + threadIdx.x] = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
}

__global__ __launch_bounds__(512) void k_valu(float* out, int iters, float c0, float c1) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], c0, c1);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "mfma";
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    float* out; u32x4* buf;
    (void)hipMalloc(&out, 256 * 512 * 4);
    const size_t bytes = (size_t)3 << 20;
    (void)hipMalloc(&buf, bytes);
    (void)hipMemset(buf, 0x5a, bytes);
    (void)hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const auto t0 = std::chrono::steady_clock::now();
    double units = 0;
    int launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        if (!strcmp(mode, "mfma")) { hipLaunchKernelGGL(k_mfma<0>, dim3(256), dim3(512), 0, 0, out, 1 << 17); units += 8.0 * 16384 * (1 << 17) * 8 * 256; }
        else if (!strcmp(mode, "mfmaz")) { hipLaunchKernelGGL(k_mfma<1>, dim3(256), dim3(512), 0, 0, out, 1 << 17); units += 8.0 * 16384 * (1 << 17) * 8 * 256; }
        else if (!strcmp(mode, "mfma32")) { hipLaunchKernelGGL(k_mfma32, dim3(256), dim3(512), 0, 0, out, 1 << 17); units += 4.0 * 32768 * (1 << 17) * 8 * 256; }
        else if (!strcmp(mode, "mfma16")) { hipLaunchKernelGGL(k_mfma16, dim3(256), dim3(512), 0, 0, out, 1 << 17); units += 8.0 * 8192 * (1 << 17) * 8 * 256; }
        else if (!strcmp(mode, "mfmaf16")) { hipLaunchKernelGGL(k_mfmaf16, dim3(256), dim3(512), 0, 0, out, 1 << 17); units += 8.0 * 16384 * (1 << 17) * 8 * 256; }
        else if (!strcmp(mode, "lds")) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 131072, 0, out, 1 << 15); units += 16.0 * 1024 * (1 << 15) * 8 * 256; }
        else if (!strcmp(mode, "l2")) { hipLaunchKernelGGL(k_l2, dim3(256), dim3(512), 0, 0, out, buf, (int)(bytes / 1024), 1 << 13); units += 8.0 * 1024 * (1 << 13) * 8 * 256; }
        else if (!strcmp(mode, "valu")) { hipLaunchKernelGGL(k_valu, dim3(256), dim3(512), 0, 0, out, 1 << 19, 1.0001f, 0.5f); units += 8.0 * 64 * (1 << 19) * 8 * 256; }
        (void)hipDeviceSynchronize();
        ++launches;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const char* unit = !strncmp(mode, "mfma", 4) ? "TFLOP/s" : (!strcmp(mode, "valu") ? "T lane-fma/s" : "TB/s");
    printf("%s: %d launches in %.2f s -> %.1f %s\n", mode, launches, dt, units / dt / 1e12, unit);
    return 0;
}
