// Internal definitions shared by the HIP translation units of libbeso_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <type_traits>
#include "../../include/beso_hip.h"
#ifndef BESO_DEV_API
#define BESO_DEV_API 0     // 1: the development build (libbeso_hip_dev.so, include/beso_hip_debug.h)
#endif

// Compile-time variants -- ablations that produce wrong results for timing, A/B switches of measured-and-decided choices
// (the `#ifndef BESO_... / #define` blocks in fused.hip and train.hip document each) -- exist in VARIANT builds only
// (tools/variants.py passes -DBESO_VARIANTS=1): the product sources refuse every override, so the shipped binary is the one
// configuration the tests run.
#ifndef BESO_VARIANTS
#define BESO_VARIANTS 0
#endif
#if !BESO_VARIANTS
#if defined(BESO_ABL_MASK) || defined(BESO_ZERO_PAD) || defined(BESO_FUSED_ABLATE) || defined(BESO_FUSED_WLOAD) || \
    defined(BESO_RED_PAD) || defined(BESO_FUSED_STAMPS) || defined(BESO_GELU_SCALAR) || defined(BESO_FC1_PF) || \
    defined(BESO_LAT_PF1) || defined(BESO_V_TR) || defined(BESO_LONG_PAIRED) || defined(BESO_LONG_PF1) || \
    defined(BESO_TGEMM_WAVES) || defined(BESO_TGEMM_NOSTORE) || defined(BESO_TRAIN_FWD_ABL) || defined(BESO_WG_SETS) || \
    defined(BESO_KEEP_HYBRID)
#error "compile-time variants of the kernels need -DBESO_VARIANTS=1 (tools/variants.py); the product build takes none"
#endif
#endif

namespace beso {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int kWave = 64;
constexpr int kTileMN = 128;   // generic GEMM block tile (M and N)
constexpr int kTileKBytes = 128;  // bytes of K per LDS row per stage (64 bf16 / 32 fp32)
constexpr int kHeadHidden = 100;  // action_pred hidden width when linear_output is False (score_gpts.py:187)

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ inline size_t round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

// bf16 <-> fp32, round-to-nearest-even
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

template <typename T> struct Act;
template <> struct Act<uint16_t> {            // bf16 activations / GEMM operands
    static constexpr int kPerChunk = 8;        // elements per 16-byte chunk
    __device__ static __forceinline__ uint16_t from(float f) { return f2bf(f); }
    __device__ static __forceinline__ float to(uint16_t v) { return bf2f(v); }
};
template <> struct Act<float> {
    static constexpr int kPerChunk = 4;
    __device__ static __forceinline__ float from(float f) { return f; }
    __device__ static __forceinline__ float to(float v) { return v; }
};

// bf16 outputs: the same exact-erf GELU through the transcendental-free fit of the fused path (fused.hip
// gelu_fast2, tools/fit_gelu.py): max |error| 1.9e-4, an order of magnitude under the bf16 rounding applied
// next, at 11 VALU instructions instead of the ~40 of erff -- the epilogue of fc1 was costlier than its MFMAs.
__device__ __forceinline__ float gelu_poly(float v) {
    const float vc = __builtin_amdgcn_fmed3f(v, -4.0f, 4.0f);
    const float s = vc * vc;
    float p = fmaf(s, 2.277972093e-08f, -1.598515742e-06f);
    p = fmaf(p, s, 4.795382804e-05f);
    p = fmaf(p, s, -0.0008139993719f);
    p = fmaf(p, s, 0.00877231165f);
    p = fmaf(p, s, -0.06457294506f);
    p = fmaf(p, s, 0.3978832308f);
    return v * fmaf(vc, p, 0.5f);
}
// Derivative of the same GELU, d/dv [v Phi(v)] = Phi(v) + v phi(v), as ONE odd polynomial: GELU'(v) - 1/2 = vc R(vc^2),
// vc = clamp(v, +-4), R of degree 7 fitted to the exact derivative (tools/fit_gelu.py --grad): max |error| 2.7e-4 over all v
// (beyond the clamp the true derivative is within 5e-4 of its value at +-4), below the bf16 rounding of the product it enters.
// 10 VALU operations and no transcendental (round 5: the fitted Phi + v phi(v) through v_exp_f32, ~17: the GELU' epilogue
// was 16 % of train_mlp_bwd_kernel's cycles, VALU bound).  Training step, bf16 mode; the fp32 mode uses erff / expf.
#define BESO_GELU_GRAD_C0 0.7967216041f
#define BESO_GELU_GRAD_C1 -0.2620297414f
#define BESO_GELU_GRAD_C2 0.05591474429f
#define BESO_GELU_GRAD_C3 -0.007687413836f
#define BESO_GELU_GRAD_C4 0.0006876406287f
#define BESO_GELU_GRAD_C5 -3.845913275e-05f
#define BESO_GELU_GRAD_C6 1.213786018e-06f
#define BESO_GELU_GRAD_C7 -1.641942029e-08f
__device__ __forceinline__ float gelu_grad_poly(float v) {
    const float vc = __builtin_amdgcn_fmed3f(v, -4.0f, 4.0f);
    const float s = vc * vc;
    float p = fmaf(s, BESO_GELU_GRAD_C7, BESO_GELU_GRAD_C6);
    p = fmaf(p, s, BESO_GELU_GRAD_C5);
    p = fmaf(p, s, BESO_GELU_GRAD_C4);
    p = fmaf(p, s, BESO_GELU_GRAD_C3);
    p = fmaf(p, s, BESO_GELU_GRAD_C2);
    p = fmaf(p, s, BESO_GELU_GRAD_C1);
    p = fmaf(p, s, BESO_GELU_GRAD_C0);
    return fmaf(vc, p, 0.5f);
}

// One sampler update on one element, in the reference's operation order (gc_sampling.py:921-923 DDIM, :205-210 Euler,
// :296-310 Heun; torch evaluates every product, quotient and sum as its own rounded fp32 operation, so nothing here may
// be contracted into an fma).  Shared by sampler_step_kernel (the step-by-step form) and the sampler loop inside
// layers_kernel, which therefore agree bit for bit.
//   DDIM          x <- c0*x - c1*den                                    next input: x
//   EULER         d = (x - den)/c0;  x <- x + d*c1                      next input: x
//   HEUN_PREDICT  d = (x - den)/c0;  aux <- d;  x2 <- x + d*c1          next input: x2   (x unchanged)
//   HEUN_CORRECT  d2 = (x2 - den)/c0;  x <- x + ((aux + d2)/2)*c1       next input: x
// Returns the value the mode writes to `out` (x for DDIM / EULER / HEUN_CORRECT, x2 for HEUN_PREDICT).
__device__ __forceinline__ float sampler_update(int mode, float xv, float x2v, float dv, float& aux, float c0, float c1) {
#pragma clang fp contract(off)
    if (mode == BESO_STEP_DDIM) {
        const float a = c0 * xv, b = c1 * dv;
        return a - b;
    }
    if (mode == BESO_STEP_EULER) {
        const float d = (xv - dv) / c0;
        const float s = d * c1;
        return xv + s;
    }
    if (mode == BESO_STEP_HEUN_PREDICT) {
        const float d = (xv - dv) / c0;
        aux = d;
        const float s = d * c1;
        return xv + s;
    }
    const float d2 = (x2v - dv) / c0;
    const float dp = (aux + d2) / 2.0f;
    const float s = dp * c1;
    return xv + s;
}

// Training-mode dropout: keep-scale of one element from a counter-based hash of (seed, site, element index) -- the
// backward pass recomputes the mask.  p = 0 never reaches this function.  (train.hip; the attention core of the one-launch
// training forward in fused.hip draws the same mask as attn_small_kernel.)
__device__ __forceinline__ float drop_scale(uint32_t seed, uint32_t site, size_t idx, float p, float inv_keep) {
    uint32_t h = (uint32_t)idx * 0x9E3779B1u + (uint32_t)(idx >> 32) * 0x7FEB352Du + site * 0x85EBCA77u + seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return ((float)(h >> 8) * (1.0f / 16777216.0f)) >= p ? inv_keep : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The same sum on the VALU: four DPP steps inside the 16-lane rows (quad_perm, row_ror) and gfx950's two lane-swap
// instructions across the rows, instead of six ds_bpermute_b32 round trips through the LDS pipe -- for kernels whose time is
// the latency chain of their reductions (LayerNorm backward).  Every lane receives the total.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
    auto dpp = [](float x, auto CTRL) {
        return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), decltype(CTRL)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x124>{});     // row_ror:4
    v += dpp(v, std::integral_constant<int, 0x128>{});     // row_ror:8
    const u32x2_t a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float r = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const u32x2_t b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r), __float_as_uint(r), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ---------------------------------------------------------------------------------------------
// Packed-weight image.  Offsets in bytes from the start of the image; every section 256-B aligned.
// GEMM operands are [Np][Kp] (torch Linear layout = B^T, k contiguous), Np = ceil128(N),
// Kp = ceil64(K), zero padded, element type = bf16 or fp32 by precision.
// ---------------------------------------------------------------------------------------------
struct LayerOff {
    size_t ln1_w, ln1_b, ln2_w, ln2_b;   // fp32 [D]
    size_t b_qkv, b_proj, b_fc1, b_fc2;  // fp32 [Np]
    size_t w_qkv, w_proj, w_fc1, w_fc2;  // elem [Np][Kp]
};

constexpr int kMaxLayers = 32;

struct Layout {
    int D, H, hd, L, G, W, obs, act, seq_size, linear_output;
    int Kd, Kh;          // padded K of D and 4D
    int Nqkv, Nd, Nh;    // padded N of 3D, D, 4D
    int elem_bytes;      // 2 (bf16, bf16x3) or 4 (fp32)
    size_t pos_emb, tok_w, tok_b, sig_w, sig_b, act_w, act_b, lnf_w, lnf_b;
    size_t head_w0, head_b0, head_w1, head_b1;   // linear_output: w0=[act,D], b0=[act]; else Linear(D,100),Linear(100,act)
    LayerOff layer[kMaxLayers];
    size_t fused;        // image of the fused path (fused.hip), 0 bytes when unsupported
    size_t total;
};

// Workspace carve-up for one forward over Mtok = Bv*T tokens.
struct Workspace {
    size_t x;      // fp32 [Mtok][D]       residual stream
    size_t xn;     // elem [Mtok][Kd]      LayerNorm output (GEMM operand)
    size_t qkv;    // elem [Mtok][3D]
    size_t y;      // elem [Mtok][Kd]      attention output
    size_t h;      // elem [Mtok][Kh]      MLP hidden
    size_t den;    // fp32 [B][t][act]     denoised (sampler scratch)
    size_t x2;     // fp32 [B][t][act]     Heun predictor state
    size_t d1;     // fp32 [B][t][act]     Heun first derivative d
    size_t sig;    // fp32 [B]             per-step sigma vector of beso_sample
    size_t small;  // fp32 [(H + 1)][kSmallProjRows][D]   small-batch path, few token rows: x_mid + the out-projection's per-head slabs
    size_t fused;  // scratch of the fused path
    size_t total;
};

int  validate_config(const beso_config* cfg);
bool make_layout(const beso_config* cfg, int precision, Layout* out);
bool make_workspace(const beso_config* cfg, const Layout& lay, int batch, int t, int precision,
                    int cfg_guidance, Workspace* out);

// ---- kernel launchers (each enqueues on `s`, returns hipError_t) -------------------------------
struct FwdArgs {
    const float* state; const float* action; const float* goal; const float* sigma;
    float* out;
    float* aux = nullptr;   // sampler loop only: [B][t][act] scratch (Heun's first slope between its two evaluations)
    const float* noise = nullptr;   // sampler loop, euler_ancestral: [evaluations of the launch][B][t][act], the steps' randn
    int batch;        // real batch B
    int vbatch;       // virtual batch: B or 2B (classifier-free guidance)
    int t;            // observations in the window
    int T;            // tokens per sample
    int precondition; // 1: GCDenoiser (c_in / c_out / c_skip), 0: raw inner model
    int uncond_from;  // virtual samples >= uncond_from use zero goals
    float cond_lambda;
    float sigma_data;
    int plan = 0;     // BESO_PLAN_* bits of the call's flags: which kernels run (never what they compute)
};

hipError_t launch_pack_matrix(const float* src, int rows, int cols, void* dst, int rows_p,
                              int cols_p, int precision, hipStream_t s);
constexpr int kScaleMax = 4;              // tensors per beso_scale_rows launch
hipError_t launch_scale_rows(const float* const* src, float* const* dst, const float* const* mean, const float* const* den,
                             const long long* rows, const int* cols, int n, hipStream_t s);
hipError_t launch_embed(const Layout& lay, const char* packed, const FwdArgs& a, float* x, hipStream_t s, bool per_row = false);
hipError_t launch_layernorm(const float* x, const float* w, const float* b, void* out, int rows,
                            int D, int ld_out, int precision, hipStream_t s);
hipError_t launch_attention(const void* qkv, void* y, int vbatch, int T, int D, int H, int ld_y,
                            int precision, hipStream_t s);
hipError_t launch_head(const Layout& lay, const char* packed, const FwdArgs& a, const float* x, hipStream_t s);
enum { EPI_BIAS_STORE = 0, EPI_BIAS_GELU_STORE = 1, EPI_BIAS_RESID = 2 };
hipError_t launch_gemm(int precision, int epi, const void* A, int lda, const void* Wt, int ldw,
                       const float* bias, void* out, int ldo,
                       int n_store, int M, int Np, int Kp, hipStream_t s);
hipError_t launch_sampler_step(int mode, float* out, float* aux, const float* x, const float* x2, const float* den,
                               float c0, float c1, size_t n, hipStream_t s, float* sig_next = nullptr, float sigma_next = 0.f,
                               int n_sig = 0);

// small.hip: the chip-wide small-batch path (five short launches per layer)
// token rows up to which the library takes it by itself (measured, kitchen: bf16 190 ... 228 us for 1 ... 16 samples against
// 257 ... 264 us of the one-launch kernel's latency instance, equal at 32 samples; fp32 641 ... 1060 us for 1 ... 93 samples
// against 1620 ... 1960 us of the per-op kernels)
constexpr int kSmallRows = 448, kSmallRowsF32 = 4096;
constexpr int kSmallProjRows = 96;          // token rows up to which the out-projection rides in the attention launch (small.hip, round 6)
constexpr size_t kSmallMinLDD = 500000;      // bf16: layers x embed_dim^2 from which the path beats the one-launch kernel
bool small_supported(const Layout& lay, int precision);
bool small_wanted(const Layout& lay, const FwdArgs& a, int precision);
int  forward_small(const Layout& lay, const Workspace& ws, const char* packed, int precision, const FwdArgs& a, char* wsp,
                   hipStream_t s, hipError_t* err);

// profile hooks (api.hip)
void profile_begin(int site, hipStream_t s);
void profile_end(int site, hipStream_t s);

hipError_t launch_gather_windows(const float* observations, const float* actions, const int* seq_len, int n_traj,
                                 int t_max, int obs, int act, const int* slice_traj, const int* slice_start,
                                 long long n_slices, const long long* batch_slices, const long long* draws, int batch,
                                 int window, int goal_len, int goal_mode, int min_future_sep, float* obs_out,
                                 float* act_out, float* goal_out, hipStream_t s);
hipError_t launch_log_logistic(const double* u, float* out, size_t n, double loc, double scale, double lo, double hi, hipStream_t s);
hipError_t launch_adam_ema(const void* chunks, int n_chunks, float* m, float* v, float* ema, float lr, float beta1,
                           float beta2, float eps, float wd, int decoupled, int step, float ema_decay, hipStream_t s);

}  // namespace beso
