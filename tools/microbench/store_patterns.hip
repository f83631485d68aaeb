// Store-request granularity on MI355X: the same bytes written as row-major [rows][ld] bf16 activations by 512-thread workgroups,
// with different lane -> (row, 8-byte piece) maps.  Why: the training kernels store accumulator-layout data, lane (n, g) of a wave
// = row n, piece g -- sixteen DIFFERENT rows per quarter-wave.  hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns
//   mode 0: lane (n = lane & 15, g = lane >> 4): row n, piece g                      (the MFMA accumulator layout)
//   mode 1: lane -> row lane >> 2, piece lane & 3                                    (a row's four pieces in adjacent lanes: 32 B runs)
//   mode 2: lane -> row lane >> 4, piece lane & 15                                   (a row's 128 B in a quarter-wave)
//   mode 3: as 2 with 16-byte pieces (row lane >> 3, piece lane & 7)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512) void store_kernel(uint2* __restrict__ out, int rows_per_wg, int ld8, int mode, int reps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // a workgroup owns rows_per_wg rows; wave w owns pieces [12 w, 12 w + 12) of every row (96 B: three row tiles), ld8 pieces per row
    for (int rep = 0; rep < reps; ++rep) {
        for (int r0 = 0; r0 < rows_per_wg; r0 += 16) {
            const size_t row_base = (size_t)blockIdx.x * rows_per_wg + r0;
            if (mode == 0) {
                for (int i = 0; i < 3; ++i) {
                    const int row = lane & 15, piece = 12 * w + 4 * i + (lane >> 4);
                    out[(row_base + row) * ld8 + piece] = make_uint2(lane, rep);
                }
            } else if (mode == 1) {
                for (int i = 0; i < 3; ++i) {
                    const int row = lane >> 2, piece = 12 * w + 4 * i + (lane & 3);
                    out[(row_base + row) * ld8 + piece] = make_uint2(lane, rep);
                }
            } else if (mode == 2) {
                // 16 rows x 12 pieces = 192 pieces = 3 instructions of 64: lanes walk (row, piece) with piece fastest
                for (int i = 0; i < 3; ++i) {
                    const int idx = 64 * i + lane, row = idx / 12, piece = 12 * w + idx % 12;
                    out[(row_base + row) * ld8 + piece] = make_uint2(lane, rep);
                }
            } else {
                // 16-byte pieces: 16 rows x 6 = 96 -> 1.5 instructions
                for (int i = 0; i < 2; ++i) {
                    const int idx = 64 * i + lane;
                    if (idx < 96) {
                        const int row = idx / 6, p16 = 6 * w + idx % 6;
                        ((uint4*)out)[(row_base + row) * (ld8 / 2) + p16] = make_uint4(lane, rep, 0, 0);
                    }
                }
            }
        }
    }
}
int main(int argc, char** argv) {
    const int rows = 11264 * 32, ld8 = 360;          // [360448][1440] bf16 = 1.04 GB (unique addresses: beyond the 256 MB MALL)
    const int rows_per_wg = 48, reps = 1;
    uint2* out;
    hipMalloc(&out, (size_t)rows * ld8 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        // a wave covers 96 pieces... the workgroup's 8 waves cover 96 pieces x 8 = 768 B of each row; grid over row blocks x column groups
        for (int it = 0; it < 2; ++it) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(store_kernel, dim3(rows / rows_per_wg), dim3(512), 0, 0, out, rows_per_wg, ld8, mode, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)rows * 96 * 8 * reps;      // 8 waves x 12 pieces x 8 B per row
        printf("mode %d: %.3f ms, %.2f TB/s of stores (%.0f MB)\n", mode, ms, bytes / ms * 1e-9, bytes * 1e-6);
    }
    return 0;
}
