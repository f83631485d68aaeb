// Training-step path (train.hip): loss forward + parameter gradients.
#pragma once
#include "common.h"

namespace beso {

int    train_validate(const beso_config* c, int batch, int t);
size_t train_workspace_bytes(const beso_config* c, int batch, int t, int precision);
size_t train_grad_floats(const beso_config* c);
int    train_loss_grad(const beso_config* c, const float* const* params, int n_params, float* grads_flat, int precision,
                       const float* state, const float* action, const float* goal, const float* noise, const float* sigma,
                       float* loss_out, int batch, int t, int flags, float embed_pdrop, float attn_pdrop, float resid_pdrop,
                       float goal_drop, uint32_t seed, float grad_scale, void* workspace, size_t workspace_bytes, hipStream_t s, hipStream_t early_stream,
                       hipStream_t loss_stream, hipError_t* err, int* err_line);
int    train_goal_mask(float* mask, size_t n, float goal_drop, uint32_t seed, hipStream_t s, hipError_t* err, int* err_line);
int    train_early_layer(const beso_config* c);
void   train_early_range(const beso_config* c, size_t* begin, size_t* end);
#if BESO_DEV_API
int    train_debug_gemm(int precision, int a_kslow, int b_kslow, const void* A, int lda, const void* B, int ldb, float* C,
                        int ldc, int M, int N, int K, int splits, hipStream_t s, hipError_t* err, int* err_line);
#endif

}  // namespace beso
