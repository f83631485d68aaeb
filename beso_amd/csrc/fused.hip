// Fused score-network path for the shipped model shapes (bf16 MFMA inputs, fp32 everything else).
//
// MLP block  x <- x + W2 * GELU(W1 * LN2(x) + b1) + b2   (score_gpts.py:105-114)  as ONE kernel:
//   * a workgroup = 8 waves (2 per SIMD, <= 256 VGPRs) owns a tile of NTT*16 tokens; the fp32 residual
//     tile lives in MFMA accumulators for the whole kernel, split across the waves BY FEATURE
//     (wave w owns output-feature row tiles [w*RPW, (w+1)*RPW));
//   * everything is computed transposed, Y^T = W^T X^T: weights are the MFMA A operand (rows =
//     output features) and are read straight from L2 into registers in a pre-packed fragment order
//     (1 KiB per wave-instruction, lane-linear), activations are the B operand (columns = tokens);
//   * the D(col = token, row = 4*(lane>>4)+reg) accumulator layout of one GEMM IS the B-operand layout
//     of the next one up to a permutation of the contraction index, and the weights are packed with
//     that permutation (slot (g,j) of k-step kk <-> index 32kk + 16(j>>2) + 4g + (j&3)); so the GELU
//     output goes accumulator -> v_cvt_pk_bf16_f32 -> B fragment with no transpose;
//   * LN2's gamma/beta are folded into W1/b1 at pack time, the kernel only normalises;
//   * LayerNorm statistics and the B fragments (normalised x, GELU(h)) are exchanged between the
//     waves through LDS in lane-linear 1 KiB fragments (conflict-free ds_read_b128 / ds_write_b128).
#include "fused.h"

namespace beso {

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

namespace {

constexpr int kWaves = 8;
constexpr int kChunkTiles = 2 * kWaves;     // hidden row tiles per chunk: 2 per wave = one FC2 k-step per wave

struct FusedDims {
    int D, FT, RPW, KS, HT, NCH, KS2p;   // features, feature tiles, row tiles per wave, k-steps of D,
                                          // hidden row tiles, hidden chunks, padded hidden k-steps
    size_t w1_bytes, b1_bytes, w2_bytes, b2_bytes, layer_bytes;
};

bool fused_dims(const Layout& lay, FusedDims* d) {
    d->D = lay.D;
    if (lay.D % 8 != 0) return false;
    d->FT = (lay.D + 15) / 16;
    d->RPW = (d->FT + kWaves - 1) / kWaves;
    d->KS = (lay.D + 31) / 32;
    if (2 * d->KS < d->FT) return false;
    d->HT = (4 * lay.D) / 16;
    if ((4 * lay.D) % 32 != 0) return false;
    d->NCH = (d->HT + kChunkTiles - 1) / kChunkTiles;
    d->KS2p = d->NCH * kWaves;
    d->w1_bytes = round_up_sz((size_t)d->NCH * kChunkTiles * d->KS * 1024, 256);
    d->b1_bytes = round_up_sz((size_t)d->NCH * kChunkTiles * 16 * sizeof(float), 256);
    d->w2_bytes = round_up_sz((size_t)d->RPW * kWaves * d->KS2p * 1024, 256);
    d->b2_bytes = round_up_sz((size_t)d->RPW * kWaves * 16 * sizeof(float), 256);
    d->layer_bytes = d->w1_bytes + d->b1_bytes + d->w2_bytes + d->b2_bytes;
    return true;
}

bool shape_has_kernel(const FusedDims& d) {
    // instantiated (RPW, KS): kitchen D=360 -> (3, 12); block-push D=240 -> (2, 8)
    return (d.RPW == 3 && d.KS == 12) || (d.RPW == 2 && d.KS == 8);
}

// ---------------------------------------------------------------------------------------------
// pack kernels
// ---------------------------------------------------------------------------------------------
// dst[((R*kt + kk)*64 + lane)*8 + j] = bf16( src[16R + (lane&15)][32kk + 16(j>>2) + 4(lane>>4) + (j&3)] * colscale[col] )
__global__ void pack_mfma_a_kernel(const float* __restrict__ src, int rows, int cols, const float* __restrict__ colscale,
                                   uint16_t* __restrict__ dst, int rt, int kt) {
    size_t total = (size_t)rt * kt * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        size_t tile = i >> 9;
        int kk = (int)(tile % kt), R = (int)(tile / kt);
        int r = 16 * R + (lane & 15);
        int c = 32 * kk + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
        float v = 0.f;
        if (r < rows && c < cols) {
            v = src[(size_t)r * cols + c];
            if (colscale) v *= colscale[c];
        }
        dst[i] = f2bf(v);
    }
}

// out[r] = b[r] + sum_c W[r][c] * beta[c]   (LayerNorm beta folded into the following Linear's bias)
__global__ void fold_bias_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ beta,
                                 float* __restrict__ out, int rows, int cols, int rows_p) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows_p) return;
    float acc = 0.f;
    if (r < rows) {
        acc = b[r];
        for (int c = 0; c < cols; ++c) acc = fmaf(W[(size_t)r * cols + c], beta[c], acc);
    }
    out[r] = acc;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}

// exact-erf GELU (nn.GELU() default).  erf by Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7:
// 2 transcendentals (rcp, exp2) + ~11 plain VALU ops.
__device__ __forceinline__ float gelu_fast(float v) {
    const float x = v * 0.70710678118654752440f;
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(x * x * -1.4426950408889634f);   // exp(-x^2)
    const float erf_abs = fmaf(-p, e, 1.0f);
    const float erf_v = copysignf(erf_abs, x);
    return 0.5f * v * (1.0f + erf_v);
}

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                   0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// MLP block kernel.  RPW: output-feature row tiles per wave; KS: k-steps (32 wide) over D;
// NTT: token tiles (16 tokens) per workgroup.
// LDS: xnT [NTT][KS] KiB | hT [NTT][8] KiB | red [2][8][NTT*16] floats
// ---------------------------------------------------------------------------------------------
template <int RPW, int KS, int NTT>
__global__ __launch_bounds__(512, 2) void mlp_block_kernel(float* __restrict__ x, const u32x4* __restrict__ w1p,
                                                           const float* __restrict__ b1f,
                                                           const u32x4* __restrict__ w2p,
                                                           const float* __restrict__ b2p, int M, int D, int HT,
                                                           int KS2p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    u32x4* xnT = (u32x4*)lds;                                   // [(t*KS + kk)*64 + lane]
    u32x4* hT = (u32x4*)(lds + (size_t)NTT * KS * 1024);        // [(t*8 + kl)*64 + lane]
    float* red = (float*)(lds + (size_t)NTT * (KS + 8) * 1024);  // [2][8][NTT*16]
    constexpr int MT = NTT * 16;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * MT;

    // ---- residual slice -> accumulators
    f32x4 acc[RPW][NTT];
    bool fvalid[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int f0 = 16 * (w * RPW + i) + 4 * g;
        fvalid[i] = f0 < D;
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
            const int tok = m0 + t * 16 + n;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (fvalid[i] && tok < M) v = *(const f32x4*)(x + (size_t)tok * D + f0);
            acc[i][t] = v;
        }
    }

    // ---- LayerNorm statistics (two-pass, fp32), partial sums exchanged through LDS
    float mean[NTT], rstd[NTT];
    const float invD = 1.0f / (float)D;
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < RPW; ++i) s += (acc[i][t][0] + acc[i][t][1]) + (acc[i][t][2] + acc[i][t][3]);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (g == 0) red[(0 * kWaves + w) * MT + t * 16 + n] = s;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < kWaves; ++ww) s += red[(0 * kWaves + ww) * MT + t * 16 + n];
        mean[t] = s * invD;
    }
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            if (fvalid[i]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = acc[i][t][r] - mean[t]; q = fmaf(d, d, q); }
            }
        }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        if (g == 0) red[(1 * kWaves + w) * MT + t * 16 + n] = q;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        float q = 0.f;
#pragma unroll
        for (int ww = 0; ww < kWaves; ++ww) q += red[(1 * kWaves + ww) * MT + t * 16 + n];
        rstd[t] = 1.0f / sqrtf(q * invD + 1e-5f);
    }

    // ---- normalised x as bf16 B fragments -> LDS; then add the FC2 bias to the residual
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int Rf = w * RPW + i;
        if ((Rf >> 1) < KS) {
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                uint2 pk = make_uint2(0u, 0u);
                if (fvalid[i]) {
                    const float a = rstd[t], b = -mean[t] * rstd[t];
                    pk.x = pack_bf16x2(fmaf(acc[i][t][0], a, b), fmaf(acc[i][t][1], a, b));
                    pk.y = pack_bf16x2(fmaf(acc[i][t][2], a, b), fmaf(acc[i][t][3], a, b));
                }
                uint2* dst = (uint2*)(xnT + ((size_t)t * KS + (Rf >> 1)) * 64 + lane) + (Rf & 1);
                *dst = pk;
            }
        }
        const f32x4 bias = *(const f32x4*)(b2p + 16 * Rf + 4 * g);
#pragma unroll
        for (int t = 0; t < NTT; ++t) acc[i][t] += bias;
    }
    __syncthreads();

    // ---- hidden chunks: FC1 (+bias, GELU) -> hT -> FC2 accumulate
    const int n_chunks = (HT + kChunkTiles - 1) / kChunkTiles;
    for (int c = 0; c < n_chunks; ++c) {
        const int tiles_here = min(kChunkTiles, HT - c * kChunkTiles);
        if (2 * w < tiles_here) {
            const int R0 = c * kChunkTiles + 2 * w;
            f32x4 h0[NTT], h1[NTT];
            const f32x4 bias0 = *(const f32x4*)(b1f + 16 * R0 + 4 * g);
            const f32x4 bias1 = *(const f32x4*)(b1f + 16 * (R0 + 1) + 4 * g);
#pragma unroll
            for (int t = 0; t < NTT; ++t) { h0[t] = bias0; h1[t] = bias1; }
            const u32x4* a0p = w1p + (size_t)R0 * KS * 64 + lane;
            const u32x4* a1p = a0p + (size_t)KS * 64;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const u32x4 a0 = a0p[kk * 64];
                const u32x4 a1 = a1p[kk * 64];
#pragma unroll
                for (int t = 0; t < NTT; ++t) {
                    const u32x4 b = xnT[((size_t)t * KS + kk) * 64 + lane];
                    h0[t] = mfma_bf16(a0, b, h0[t]);
                    h1[t] = mfma_bf16(a1, b, h1[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                u32x4 hb;
                hb[0] = pack_bf16x2(gelu_fast(h0[t][0]), gelu_fast(h0[t][1]));
                hb[1] = pack_bf16x2(gelu_fast(h0[t][2]), gelu_fast(h0[t][3]));
                hb[2] = pack_bf16x2(gelu_fast(h1[t][0]), gelu_fast(h1[t][1]));
                hb[3] = pack_bf16x2(gelu_fast(h1[t][2]), gelu_fast(h1[t][3]));
                hT[((size_t)t * kWaves + w) * 64 + lane] = hb;
            }
        }
        __syncthreads();
        const int ksteps = tiles_here >> 1;
        for (int kl = 0; kl < ksteps; ++kl) {
            u32x4 a[RPW];
#pragma unroll
            for (int i = 0; i < RPW; ++i) a[i] = w2p[((size_t)(w * RPW + i) * KS2p + (c * kWaves + kl)) * 64 + lane];
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                const u32x4 b = hT[((size_t)t * kWaves + kl) * 64 + lane];
#pragma unroll
                for (int i = 0; i < RPW; ++i) acc[i][t] = mfma_bf16(a[i], b, acc[i][t]);
            }
        }
        __syncthreads();
    }

    // ---- residual tile back to HBM
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int f0 = 16 * (w * RPW + i) + 4 * g;
        if (!fvalid[i]) continue;
#pragma unroll
        for (int t = 0; t < NTT; ++t) {
            const int tok = m0 + t * 16 + n;
            if (tok < M) *(f32x4*)(x + (size_t)tok * D + f0) = acc[i][t];
        }
    }
}

constexpr int kNTT = 6;   // 96 tokens per workgroup

template <int RPW, int KS>
hipError_t launch_mlp_block(float* x, const char* base, const FusedDims& d, int M, hipStream_t s) {
    const size_t lds_bytes = (size_t)kNTT * (KS + 8) * 1024 + 2 * kWaves * kNTT * 16 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)mlp_block_kernel<RPW, KS, kNTT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = (M + kNTT * 16 - 1) / (kNTT * 16);
    (void)hipGetLastError();
    hipLaunchKernelGGL((mlp_block_kernel<RPW, KS, kNTT>), dim3(grid), dim3(512), lds_bytes, s, x,
                       (const u32x4*)base, (const float*)(base + d.w1_bytes),
                       (const u32x4*)(base + d.w1_bytes + d.b1_bytes),
                       (const float*)(base + d.w1_bytes + d.b1_bytes + d.w2_bytes), M, d.D, d.HT, d.KS2p);
    return hipGetLastError();
}

}  // namespace

// ---------------------------------------------------------------------------------------------
size_t fused_packed_bytes(const Layout& lay, int precision) {
    FusedDims d;
    if (precision != BESO_PREC_BF16 || !fused_dims(lay, &d) || !shape_has_kernel(d)) return 0;
    return d.layer_bytes * lay.L;
}

size_t fused_workspace_bytes(const Layout&, int, int, int) { return 0; }

#define FTRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return BESO_ERR_HIP; } while (0)

int fused_pack(const Layout& lay, const float* const* p, char* packed, int precision, hipStream_t s) {
    FusedDims d;
    if (precision != BESO_PREC_BF16 || !fused_dims(lay, &d) || !shape_has_kernel(d)) return BESO_OK;
    const int D = lay.D;
    for (int l = 0; l < lay.L; ++l) {
        // parameter order (include/beso_hip.h): 3 leading tensors, then 16 per block:
        // ln1.w ln1.b ln2.w ln2.b key.w key.b query.w query.b value.w value.b proj.w proj.b fc1.w fc1.b fc2.w fc2.b
        const float* const* q = p + 3 + 16 * l;
        const float *ln2w = q[2], *ln2b = q[3], *f1w = q[12], *f1b = q[13], *f2w = q[14], *f2b = q[15];
        char* base = packed + lay.fused + (size_t)l * d.layer_bytes;
        const int rt1 = d.NCH * kChunkTiles, rt2 = d.RPW * kWaves;
        (void)hipGetLastError();
        hipLaunchKernelGGL(pack_mfma_a_kernel, dim3(1024), dim3(256), 0, s, f1w, 4 * D, D, ln2w, (uint16_t*)base, rt1,
                           d.KS);
        hipLaunchKernelGGL(fold_bias_kernel, dim3((rt1 * 16 + 255) / 256), dim3(256), 0, s, f1w, f1b, ln2b,
                           (float*)(base + d.w1_bytes), 4 * D, D, rt1 * 16);
        hipLaunchKernelGGL(pack_mfma_a_kernel, dim3(1024), dim3(256), 0, s, f2w, D, 4 * D, (const float*)nullptr,
                           (uint16_t*)(base + d.w1_bytes + d.b1_bytes), rt2, d.KS2p);
        FTRY(hipGetLastError());
        FTRY(launch_pack_matrix(f2b, 1, D, base + d.w1_bytes + d.b1_bytes + d.w2_bytes, 1, rt2 * 16, -1, s));
    }
    return BESO_OK;
}

bool fused_supported(const Layout& lay, const FwdArgs&, int precision) {
    FusedDims d;
    return precision == BESO_PREC_BF16 && lay.fused != lay.total && fused_dims(lay, &d) && shape_has_kernel(d);
}

int fused_mlp_block(const Layout& lay, const char* packed, int layer, float* x, int M, hipStream_t s) {
    FusedDims d;
    if (!fused_dims(lay, &d)) return BESO_ERR_UNSUPPORTED;
    const char* base = packed + lay.fused + (size_t)layer * d.layer_bytes;
    hipError_t e;
    if (d.RPW == 3 && d.KS == 12) e = launch_mlp_block<3, 12>(x, base, d, M, s);
    else if (d.RPW == 2 && d.KS == 8) e = launch_mlp_block<2, 8>(x, base, d, M, s);
    else return BESO_ERR_UNSUPPORTED;
    return e == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

int forward_fused(const Layout&, const Workspace&, const char*, int, const FwdArgs&, char*, hipStream_t) {
    return BESO_ERR_UNSUPPORTED;   // orchestration lives in api.hip (forward_generic with use_fused_mlp)
}

}  // namespace beso
