/*
 * beso_hip_debug.h -- development interface of libbeso_hip_dev.so (`python -m beso_amd.build --dev`: the same sources
 * compiled with -DBESO_DEV_API=1).  NOT part of the product library: libbeso_hip.so exports none of these, and nothing in
 * the package calls them.  Used by tools/ (phase stamps of the fused kernels) and by the operand-layout test of the training
 * GEMM.  These entry points keep process-wide state; do not use them from concurrent threads.
 */
#ifndef BESO_HIP_DEBUG_H
#define BESO_HIP_DEBUG_H

#include "beso_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: the entry points declared here are its whole dynamic symbol table */
#pragma GCC visibility push(default)

/* Install a device buffer of `capacity_u64` uint64 slots; workgroup 0 of the fused kernels (built with
 * -DBESO_FUSED_STAMPS=1) appends {phase id, shader clock} pairs to it (NULL / 0 switches it off).               */
void beso_debug_set_stamps(void* device_buf, int capacity_u64);

/* Operand layouts of the training GEMM: C[M][N] (fp32, ldc) = sum_k A(m,k) B(n,k);
 * a_kslow / b_kslow = 1: the operand is stored [K][ld] (contraction index slow), 0: [rows][ld] (k contiguous).
 * Supported pairs: (0,0), (0,1), (1,1).  splits > 1 accumulates split-K partial sums into a ZEROED C.
 * bf16 only: (1, W) / (W, 1) with W = 2 or 3 run the PANEL-OWNING weight-gradient tiles of the training step (both operands
 * contraction-major; (1, W): tiles of 128 rows x all N <= 128 W columns, (W, 1): all M <= 128 W rows x 128 columns); C must be
 * contiguous (ldc == N) and, for splits > 1 (row ranges), followed by splits x round_up(M N, 4) floats of slab space.           */
int beso_debug_gemm(int precision, int a_kslow, int b_kslow, const void* A, int lda, const void* B, int ldb, float* C,
                    int ldc, int M, int N, int K, int splits, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* BESO_HIP_DEBUG_H */
