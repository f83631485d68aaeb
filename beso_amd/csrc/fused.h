// Fused score-network path (kernels specialised for the shipped model shapes).  See fused.hip.
#pragma once
#include "common.h"

namespace beso {

// The sampler loop inside layers_kernel (K8 fused into K7): one record per network evaluation of the launch -- the sigma it
// is evaluated at (the same for every sample) and the update that follows it in the head (BESO_STEP_* with the coefficients
// of beso_sampler_step).  Travels as a kernel argument: no table in device memory, nothing to copy, graph-capturable.
constexpr int kMaxLoopEvals = 128;
// mode: BESO_STEP_* | kStepAddNoise (euler_ancestral: behind the update, x += noise[evaluation] * c2 -- the caller's randn of
// that step, gc_sampling.py:246-247)
constexpr int kStepAddNoise = 0x100;
struct StepRec { float sigma, c0, c1; int mode; float c2; };
struct SampleSteps {
    int n;                         // evaluations of this launch; 0: one plain forward (per-sample sigma, out <- denoised)
    int pad[3];
    StepRec rec[kMaxLoopEvals];          // 20 B each: 2.5 KiB of the 4 KiB kernel argument
};
constexpr int kLoopMaxElems = 512;   // action-window elements per workgroup the loop can carry (one per thread)

size_t fused_packed_bytes(const Layout& lay, int precision);
int    fused_pack(const Layout& lay, const float* const* params, char* packed, int precision, hipStream_t s);
int    fused_level(const Layout& lay, const FwdArgs& a, int precision);   // 0 none, 1 MLP block, 2 whole layers
int    fused_layer_edges(const Layout& lay);        // bit 0: fused_layers embeds, bit 1: it runs the head
int    fused_layers(const Layout& lay, const char* packed, const FwdArgs& a, float* x, int* fused_edges, int precision,
                    hipStream_t s, const SampleSteps* steps = nullptr);
bool   fused_can_loop(const Layout& lay, const FwdArgs& a, int precision);   // the whole sampler loop can run inside one launch
int    fused_mlp_block(const Layout& lay, const char* packed, int layer, float* x, int M, hipStream_t s);
bool   fused_has_lin_blocks(const Layout& lay, int precision);
int    fused_lin_block(const Layout& lay, const char* packed, int layer, int which, float* x, void* buf, int ld, int M,
                       hipStream_t s);
int    fused_lin_x3(const Layout& lay, const char* packed, int layer, int next_layer, float* x, const float* y, int ld_y,
                    float* qkv_next, int M, hipStream_t s);
int    fused_lin_tail(const Layout& lay, const char* packed, int layer, float* x, const void* y, int ld_y, void* qkv_next,
                      int M, hipStream_t s);
// training forward through the tail block (train.hip): per-step fragment image of the weights + one launch per layer
bool   fused_train_supported(const Layout& lay);
size_t fused_train_image_bytes(const Layout& lay);
int    fused_train_pack(const Layout& lay, const float* const* params, char* img, hipStream_t s);
int    fused_train_tail(const Layout& lay, const char* img, int layer, int M, const float* x_in, const void* y, int ld_y,
                        float* x_mid, float* x_out, float* st2, void* xn2, void* h, void* g, float* st1n, void* xn1n,
                        void* qkvn, hipStream_t s);
// training forward of ALL layers as one launch (train_fwd_kernel): the kept activations go to the training workspace `ws`
// at byte offsets x_mid .. g of layer 0 plus l * stride; the last layer's tail runs on the compact action rows (its
// attention output to `ya`)
struct TrainWholeBufs {
    const float* x0; char* ws;
    size_t x_mid, x_out, st1, st2, xn1, qkv, y, xn2, h, g, stride, ya;
    int t;                         // steps of the window (action tokens per sample)
    float p_attn; uint32_t seed;   // attention dropout (the per-op kernels' mask)
    float p_resid;                 // dropout on the out-projection and MLP outputs (sites 4 l + 1, 4 l + 2: EpiResid's mask)
    int x_bf16;                    // 1: the kept residuals x_mid (every layer) and x_out (all but the last layer) go out as bf16
                                   // [rows][D] at the start of their fp32 buffers (p_resid = 0 only: nothing reads them but
                                   // the LayerNorm backward, which takes them with TrainLnBwd::x_bf16)
};
bool   fused_train_whole_supported(const Layout& lay, int T, int t);
size_t fused_train_whole_image_bytes(const Layout& lay);
int    fused_train_whole_pack(const Layout& lay, const float* const* params, char* img, hipStream_t s);
int    fused_train_whole(const Layout& lay, const char* img, int batch, int T, const TrainWholeBufs& a, hipStream_t s);
// training backward: the data-gradient GEMMs in the transposed formulation (train_dgrad_kernel; per-step image of the
// transposed weights).  which: 0 q|k|v, 1 out-projection (bf16 out), 2 FC1, 3 FC2 + GELU' (dh + FC1 bias column sums)
bool   fused_train_dgrad_supported(const Layout& lay);
size_t fused_train_dgrad_image_bytes(const Layout& lay);
int    fused_train_dgrad_pack(const Layout& lay, const float* const* params, char* img, hipStream_t s);
// ln (which = 0 or 2): the LayerNorm backward behind this data gradient runs as the kernel's epilogue (dxn is not written)
struct TrainLnBwd {
    const float* x; const float* stats; const float* gamma;     // LayerNorm input [M][D], (mean, rstd) [M][2], gamma [D]
    const float* dres_in; float* dres_out; void* dxb;           // residual gradient in (or nullptr) / out, its bf16 copy
    float* part;                                                // [fused_train_dgrad_blocks(M)][3][D]: dgamma, dbeta, bias partial sums
    float p; uint32_t seed, site; int skip_mod;                 // dropout of the branch behind the LayerNorm (ln_bwd_kernel's arguments)
    int x_bf16 = 0;                                             // x holds bf16 rows (TrainWholeBufs::x_bf16)
};
// FC2 + GELU' -> FC1 -> LayerNorm-2 backward -> out-projection data gradients of a layer in one launch (dh [M][4D], ln.dxb = dym
// and dy [M][D] are written; colsum: slab [fused_train_dgrad_blocks(M)][4 D])
bool   fused_train_mlp_bwd_supported(const Layout& lay);
int    fused_train_mlp_bwd(const Layout& lay, const char* img, int layer, int M, const void* dyo, const void* h, void* dh, float* colsum,
                           void* dy, const TrainLnBwd& ln, hipStream_t s);
int    fused_train_dgrad(const Layout& lay, const char* img, int layer, int which, int M, const void* in, float* out32, void* out16,
                         const void* h, void* dh, float* colsum, hipStream_t s, const TrainLnBwd* ln = nullptr);      // colsum (which = 3): slab [fused_train_dgrad_blocks(M)][4 D]
int    fused_train_dgrad_blocks(int M);
int    fused_train_bias_reduce(const float* const* slabs, float* const* outs, const int* blocks, int n, int N, hipStream_t s);
void   fused_set_stamps(void* buf, int cap);     // development builds (BESO_DEV_API): phase stamps of workgroup 0

// The fp16-operand build of layers_kernel (fused_f16.hip = fused.hip compiled with BESO_OPERAND_F16 = 1): BESO_PREC_FP16.
// Same image layout and sizes with fp16 weight fragments; `precision` arguments take BESO_PREC_BF16 ("the plain mode").
size_t fused_packed_bytes_f16(const Layout& lay, int precision);
int    fused_pack_f16(const Layout& lay, const float* const* params, char* packed, int precision, hipStream_t s);
int    fused_level_f16(const Layout& lay, const FwdArgs& a, int precision);
int    fused_layer_edges_f16(const Layout& lay);
int    fused_layers_f16(const Layout& lay, const char* packed, const FwdArgs& a, float* x, int* fused_edges, int precision,
                        hipStream_t s, const SampleSteps* steps = nullptr);
bool   fused_can_loop_f16(const Layout& lay, const FwdArgs& a, int precision);

}  // namespace beso
