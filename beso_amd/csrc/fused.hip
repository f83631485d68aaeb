// Fused score-network path for the shipped model shapes (bf16 MFMA inputs, fp32 everything else).
//
// Structure (all kernels here share it):
//   * a workgroup = 8 waves (2 per SIMD, <= 256 VGPRs) owns a tile of NTT*16 = 96 token slots; the fp32
//     residual stream of the tile lives in MFMA accumulators, split across the waves BY FEATURE
//     (wave w owns output-feature row tiles [w*RPW, (w+1)*RPW)), for the whole kernel;
//   * every GEMM is computed transposed, Y^T = W^T X^T: weights are the MFMA A operand (rows = output
//     features), read straight from L2 into registers in a pre-packed fragment order (1 KiB per
//     wave-instruction, lane-linear); activations are the B operand (columns = tokens);
//   * the D(col = token, row = 4*(lane>>4)+reg) accumulator layout of one GEMM IS the B-operand layout
//     of the next one up to a permutation of the contraction index, and the weights are packed with
//     that permutation (slot (g,j) of k-step kk <-> index 32kk + 16(j>>2) + 4g + (j&3)); so e.g. the
//     GELU output goes accumulator -> v_cvt_pk_bf16_f32 -> B fragment with no transpose;
//   * LayerNorm gamma/beta are folded into the following Linear at pack time; kernels only normalise;
//   * B fragments (normalised x, GELU(h), attention output) and the LayerNorm partial sums are
//     exchanged between the waves through LDS in lane-linear 1 KiB fragments (conflict-free b128).
//
// Kernels:
//   mlp_block_kernel   x <- x + W2 GELU(W1 LN2(x) + b1) + b2                       (score_gpts.py:105-114)
//   layers_kernel      [token embedding ->] for l in [l0, l1): x <- x + proj(attn(LN1 x)); x <- x + mlp(LN2 x)
//                      [-> ln_f, action head, preconditioning, CFG]        (:50-115, :272-358; score_wrappers.py:81-96)
//                      attention per PAIR of (virtual) heads: one QKV GEMM -> q,k,v (bf16) of one head in LDS,
//                      the other head's accumulators parked in registers -> scores, causal softmax and P.V
//                      on the matrix pipe, one sample per wave -> y_h^T B fragments -> the head's slice of
//                      the out-projection accumulated into the residual.
//
// What bounds these kernels: they run at the board's power cap, VALU instructions do
// not hide under MFMAs beyond about one per MFMA, and every workgroup streams all weights out of L2 --
// so the code below counts VALU instructions, LDS/L2 bytes and padding MFMAs, not stalls.
#include <stdlib.h>
#include <atomic>
#include "fused.h"

// BESO_OPERAND_F16 = 1 (fused_f16.hip): the same kernels with fp16 GEMM operands -- v_mfma_f32_16x16x32_f16 runs at the bf16
// rate and carries three more mantissa bits (the accumulators, the residual stream, LayerNorm, softmax, GELU stay fp32 as in
// every mode).  That build holds layers_kernel only (BESO_PREC_FP16: the shapes with the one-launch kernel), under its own
// entry-point names.
#ifndef BESO_OPERAND_F16
#define BESO_OPERAND_F16 0
#endif
#if BESO_OPERAND_F16
#define fused_packed_bytes fused_packed_bytes_f16
#define fused_pack fused_pack_f16
#define fused_level fused_level_f16
#define fused_layer_edges fused_layer_edges_f16
#define fused_layers fused_layers_f16
#define fused_can_loop fused_can_loop_f16
#endif

namespace beso {

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

namespace {

constexpr int kWaves = 8;
constexpr int kBlock = 64 * kWaves;        // threads of a workgroup of the tile kernels (a constant, not blockDim.x: reading the
                                          // launch size costs the implicit-argument pointer -- two SGPRs carried, and spilled, across layers_kernel)
constexpr int kChunkTiles = 2 * kWaves;   // hidden row tiles per MLP chunk: 2 per wave = one FC2 k-step per wave
constexpr int kNTT = 6;                   // token tiles (16 tokens) per workgroup
constexpr int kMT = kNTT * 16;            // 96 token slots
constexpr int kSPW = 8;                   // samples per workgroup in layers_kernel
constexpr int kLongNT = 5;                // token tiles of the one-sample-per-workgroup instance (sequences up to 80 tokens)
constexpr int kHDP = 64;                  // padded head dim of the attention phase
constexpr int kQKVRow = kHDP + 8;         // bf16 elements per q/k/v row in LDS: 144 B, 16-B aligned, conflict-free b128 reads
constexpr int kQKVRows = kMT + 8;         // rows per part: a sample's 16-row MFMA window may run 8 rows past the last slot
constexpr int kQKVBytes = 3 * kQKVRows * kQKVRow * 2;   // 44928

struct FusedDims {
    int D, FT, RPW, KS, HT, NCH, KS2p, H, hd;
    int HG, Hv, hdv;                      // HG real heads form one 'virtual head' of hdv = HG*hd <= 64 dims; Hv = H / HG
    bool attn;                            // attention phase available (hd <= 64, even Hv, and 8*block_size <= 96 or seq1)
    int seq1;                             // long sequences (8 samples do not fit a tile, one does in 5 token tiles): one sample per
                                          // workgroup, the attention core runs a query tile per wave (layers_kernel CORE = 1)
    // per-layer image: [w1 | b1 | w2 | b2 | wqkv | bqkv | wproj | bproj]
    // (32-bit: the struct is a kernel argument and every field the kernel touches costs SGPRs)
    uint32_t w1_bytes, b1_bytes, w2_bytes, b2_bytes, wqkv_bytes, bqkv_bytes, wproj_bytes, bproj_bytes, layer_bytes;
    uint32_t o_b1, o_w2, o_b2, o_wqkv, o_bqkv, o_wproj, o_bproj;
    // shapes without the attention phase (long sequences): q/k/v and proj as MLP-block style kernels, natural
    // [q | k | v] row order, one part = 8*RPW row tiles
    int lin;
    uint32_t o_wqkv_lin, o_bqkv_lin, o_wproj_lin, o_bproj_lin, part_bytes;
    // per-model image behind the L per-layer images (fp32): embeddings transposed to [in][Dp], head with ln_f folded
    int Dp, obs, act, seq, G, L, head_fused;
    uint32_t g_tokT, g_tokb, g_actT, g_actb, g_sigw, g_sigb, g_pos, g_headw, g_headb, global_bytes;
    uint32_t g_tokA, g_actA;      // tok_emb / action_emb weights as split-bf16 A fragments: [hi: 8*RPW KiB | lo: 8*RPW KiB]
    // BF16X3 image: a second copy of the L per-layer images holding the LOW halves of the split-bf16 weight fragments,
    // x3_delta bytes behind the first (= L * layer_bytes + global_bytes); same offsets inside, biases not repeated
    uint32_t x3_delta;
};

bool fused_dims(const Layout& lay, FusedDims* d) {
    d->D = lay.D;
    d->H = lay.H;
    d->hd = lay.hd;
    if (lay.D % 8 != 0) return false;
    d->FT = (lay.D + 15) / 16;
    d->RPW = (d->FT + kWaves - 1) / kWaves;
    d->KS = (lay.D + 31) / 32;
    if (2 * d->KS < d->FT || (d->KS & 1)) return false;
    if ((4 * lay.D) % 32 != 0) return false;
    d->HT = (4 * lay.D) / 16;
    d->NCH = (d->HT + kChunkTiles - 1) / kChunkTiles;
    d->KS2p = d->NCH * kWaves;
    const int T = 1 + lay.G + 2 * lay.W;
    // The attention phase works on 64-wide 'virtual heads' in pairs.  Small heads are grouped: HG consecutive
    // heads are HG*hd consecutive rows of the q/k/v weights, i.e. exactly one wider head as far as the GEMMs
    // and the LDS layout go; only the attention core tells them apart (masked operands).  Block-push:
    // 12 heads of 20 -> 4 virtual heads of 60, the kitchen geometry.
    d->HG = 1;
    for (int g = 3; g >= 2; --g)
        if (lay.hd * g <= kHDP && lay.H % (2 * g) == 0) { d->HG = g; break; }
    d->Hv = lay.H / d->HG;
    d->hdv = lay.hd * d->HG;
    const bool heads_ok = d->hdv <= kHDP && lay.hd % 4 == 0 && d->Hv % 2 == 0;
    d->seq1 = (heads_ok && d->HG == 1 && kSPW * T > kMT && T <= 16 * kLongNT && d->RPW == 4 && d->KS == 16 && lay.D == 16 * kWaves * d->RPW) ? 1 : 0;
    d->attn = heads_ok && (kSPW * T <= kMT || d->seq1);
    const size_t rt2 = (size_t)d->RPW * kWaves;
    d->w1_bytes = (size_t)d->NCH * kChunkTiles * d->KS * 1024;
    d->b1_bytes = round_up_sz((size_t)d->NCH * kChunkTiles * 16 * sizeof(float), 256);
    d->w2_bytes = rt2 * d->KS2p * 1024;
    d->b2_bytes = round_up_sz(rt2 * 16 * sizeof(float), 256);
    d->wqkv_bytes = d->attn ? (size_t)d->Hv * 12 * d->KS * 1024 : 0;
    d->bqkv_bytes = d->attn ? round_up_sz((size_t)d->Hv * 3 * kHDP * sizeof(float), 256) : 0;
    d->wproj_bytes = d->attn ? rt2 * (2 * d->Hv) * 1024 : 0;
    d->bproj_bytes = d->attn ? round_up_sz(rt2 * 16 * sizeof(float), 256) : 0;
    d->o_b1 = d->w1_bytes;
    d->o_w2 = d->o_b1 + d->b1_bytes;
    d->o_b2 = d->o_w2 + d->w2_bytes;
    d->o_wqkv = d->o_b2 + d->b2_bytes;
    d->o_bqkv = d->o_wqkv + d->wqkv_bytes;
    d->o_wproj = d->o_bqkv + d->bqkv_bytes;
    d->o_bproj = d->o_wproj + d->wproj_bytes;
    d->layer_bytes = d->o_bproj + d->bproj_bytes;
    d->lin = (d->attn && !d->seq1) ? 0 : 1;      // (the long shapes keep both forms: whole layers, and the block kernels)
    d->part_bytes = (uint32_t)(rt2 * d->KS * 1024);
    if (d->lin) {
        d->o_wqkv_lin = d->layer_bytes;
        d->o_bqkv_lin = d->o_wqkv_lin + 3 * d->part_bytes;
        d->o_wproj_lin = d->o_bqkv_lin + (uint32_t)round_up_sz(3 * rt2 * 16 * sizeof(float), 256);
        d->o_bproj_lin = d->o_wproj_lin + d->part_bytes;
        d->layer_bytes = d->o_bproj_lin + (uint32_t)round_up_sz(rt2 * 16 * sizeof(float), 256);
    } else {
        d->o_wqkv_lin = d->o_bqkv_lin = d->o_wproj_lin = d->o_bproj_lin = 0;
    }
    d->Dp = d->RPW * kWaves * 16;
    d->obs = lay.obs; d->act = lay.act; d->seq = lay.seq_size; d->G = lay.G; d->L = lay.L;
    d->head_fused = lay.linear_output && lay.act <= 16;
    size_t cur = 0;
    auto carve = [&](size_t n_floats) { size_t o = cur; cur = round_up_sz(cur + n_floats * sizeof(float), 256); return o; };
    d->g_tokT = carve((size_t)lay.obs * d->Dp); d->g_tokb = carve(d->Dp);
    d->g_actT = carve((size_t)lay.act * d->Dp); d->g_actb = carve(d->Dp);
    d->g_sigw = carve(d->Dp); d->g_sigb = carve(d->Dp);
    d->g_pos = carve((size_t)lay.seq_size * d->Dp);
    d->g_headw = carve((size_t)16 * d->Dp); d->g_headb = carve(16);
    d->g_tokA = carve((size_t)2 * d->RPW * kWaves * 256); d->g_actA = carve((size_t)2 * d->RPW * kWaves * 256);
    d->global_bytes = cur;
    d->x3_delta = (uint32_t)((size_t)lay.L * d->layer_bytes + d->global_bytes);
    return true;
}

// The kernels run the last k-step of the K = D contractions as a HALF k-step (mfma_op_half): the shapes
// they are used for must have at most 16 real indices there.
// (KS = 12: D = 360 and KS = 8: D = 240 qualify; the KS = 16 instance serves D = 512, a full tail.)
constexpr bool kt16(int KS) { return KS == 12 || KS == 8; }

bool shape_has_kernel(const FusedDims& d) {
    // instantiated (RPW, KS): kitchen D=360 -> (3, 12); block-push D=240 -> (2, 8); long-horizon D=512 -> (4, 16), MLP block only
    if (kt16(d.KS) && d.D > 32 * (d.KS - 1) + 16) return false;
    return (d.RPW == 3 && d.KS == 12) || (d.RPW == 2 && d.KS == 8) || (d.RPW == 4 && d.KS == 16);
}

// ---------------------------------------------------------------------------------------------
// pack kernels.  A-operand fragment order:
//   dst[((R*kt + kk)*64 + lane)*8 + j] = bf16( M[16R + (lane&15)][32kk + 16(j>>2) + 4(lane>>4) + (j&3)] )
// ---------------------------------------------------------------------------------------------
// Fragment (R, kk) lives at tile index ((R/grp)*kt + kk)*grp + R%grp: the `grp` row tiles a workgroup
// consumes together in one k-step are contiguous (grp KiB), so a k-step's loads of all 8 waves spread
// over the L2 channels instead of striding by a whole row of k-steps.
// part: 0 = bf16(v); 1 = bf16(v - bf16(v)), the low half of the split-bf16 pair of the BF16X3 mode (hi + lo = v to 2^-16)
#if BESO_OPERAND_F16
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ uint16_t f2op(float v) { return __builtin_bit_cast(uint16_t, (_Float16)v); }     // v_cvt_f16_f32 (RNE)
__device__ __forceinline__ float op2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
#else
__device__ __forceinline__ uint16_t f2op(float v) { return f2bf(v); }
__device__ __forceinline__ float op2f(uint16_t h) { return bf2f(h); }
#endif
__device__ __forceinline__ uint16_t f2op_part(float v, int part) {
    const uint16_t h = f2op(v);
    return part ? f2op(v - op2f(h)) : h;
}

__global__ void pack_mfma_a_kernel(const float* __restrict__ src, int rows, int cols, const float* __restrict__ colscale,
                                   uint16_t* __restrict__ dst, int rt, int kt, int grp, int part) {
    size_t total = (size_t)rt * kt * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        size_t tile = i >> 9;
        int rin = (int)(tile % grp);
        int kk = (int)((tile / grp) % kt);
        int R = (int)(tile / ((size_t)grp * kt)) * grp + rin;
        int r = 16 * R + (lane & 15);
        int c = 32 * kk + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
        float v = 0.f;
        if (r < rows && c < cols) {
            v = src[(size_t)r * cols + c];
            if (colscale) v *= colscale[c];
        }
        dst[i] = f2op_part(v, part);
    }
}

// q/k/v weights of all heads, two heads per k-step group: tile index = ((h/2)*kt + kk)*24 + (h%2)*12 + part*4 + R4,
// rows of head h padded hd -> 64;
// LayerNorm-1 gamma folded in.  part 0 = query, 1 = key, 2 = value.
__global__ void pack_qkv_kernel(const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
                                const float* __restrict__ gamma, uint16_t* __restrict__ dst, int D, int H, int hd,
                                int kt, int split) {
    size_t total = (size_t)H * 12 * kt * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        size_t tile = i >> 9;                    // (pair*kt + kk)*24 + (h&1)*12 + rt12
        int rt12 = (int)(tile % 12);
        int kk = (int)((tile / 24) % kt);
        int h = 2 * (int)(tile / ((size_t)24 * kt)) + (int)((tile / 12) & 1);
        int R4 = rt12 & 3, part = rt12 >> 2;
        int d = 16 * R4 + (lane & 15);
        int c = 32 * kk + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
        float v = 0.f;
        if (d < hd && c < D) {
            const float* w = part == 0 ? wq : (part == 1 ? wk : wv);
            v = w[(size_t)(h * hd + d) * D + c] * gamma[c];
        }
        dst[i] = f2op_part(v, split);
    }
}

// out[(h*3 + part)*64 + d] = b_part[h*hd + d] + sum_c W_part[h*hd + d][c] * beta[c]   (0 for d >= hd)
// One wave per output: lanes stride over the row of W (coalesced), butterfly sum.
__global__ void fold_qkv_bias_kernel(const float* __restrict__ wq, const float* __restrict__ wk,
                                     const float* __restrict__ wv, const float* __restrict__ bq,
                                     const float* __restrict__ bk, const float* __restrict__ bv,
                                     const float* __restrict__ beta, float* __restrict__ out, int D, int H, int hd) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= H * 3 * kHDP) return;
    int d = i % kHDP, part = (i / kHDP) % 3, h = i / (3 * kHDP);
    float acc = 0.f;
    if (d < hd) {
        const float* w = part == 0 ? wq : (part == 1 ? wk : wv);
        const float* b = part == 0 ? bq : (part == 1 ? bk : bv);
        const int r = h * hd + d;
        for (int c = lane; c < D; c += 64) acc = fmaf(w[(size_t)r * D + c], beta[c], acc);
        acc = wave_sum(acc) + b[r];
    }
    if (lane == 0) out[i] = acc;
}

// out-projection: k-step (2h + kk) covers head h, head dims 32kk .. 32kk+31 (zero for d >= hd)
__global__ void pack_proj_kernel(const float* __restrict__ wp, uint16_t* __restrict__ dst, int D, int H, int hd, int rt,
                                 int part) {
    const int kt = 2 * H;
    size_t total = (size_t)rt * kt * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        size_t tile = i >> 9;                    // kk*rt + R
        int R = (int)(tile % rt), kk = (int)(tile / rt);
        int o = 16 * R + (lane & 15);
        int h = kk >> 1;
        int d = 32 * (kk & 1) + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
        float v = 0.f;
        if (o < D && d < hd) v = wp[(size_t)o * D + h * hd + d];
        dst[i] = f2op_part(v, part);
    }
}

// dst[c][f] = src[f][c] * (scale ? scale[f] : 1) for f < rows, c < cols; zero padded to [cols][rows_p]
__global__ void transpose_pad_kernel(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst, int rows_p) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cols * rows_p) return;
    int c = i / rows_p, f = i % rows_p;
    dst[i] = f < rows ? src[(size_t)f * cols + c] : 0.f;
}

// head with ln_f folded:  Wh[a][f] = W[a][f] * gamma[f] (padded to [16][Dp]),  bh[a] = b[a] + sum_f W[a][f] * beta[f]
__global__ void pack_head_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float* __restrict__ Wh, float* __restrict__ bh, int act,
                                 int D, int Dp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 16 * Dp) {
        int a = i / Dp, f = i % Dp;
        Wh[i] = (a < act && f < D) ? W[(size_t)a * D + f] * gamma[f] : 0.f;
    }
    if (i < 16) {
        float acc = 0.f;
        if (i < act) {
            acc = b[i];
            for (int f = 0; f < D; ++f) acc = fmaf(W[(size_t)i * D + f], beta[f], acc);
        }
        bh[i] = acc;
    }
}

// out[r] = b[r] + sum_c W[r][c] * beta[c]   (LayerNorm beta folded into the following Linear's bias); one wave per row
__global__ void fold_bias_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ beta,
                                 float* __restrict__ out, int rows, int cols, int rows_p) {
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (r >= rows_p) return;
    float acc = 0.f;
    if (r < rows) {
        for (int c = lane; c < cols; c += 64) acc = fmaf(W[(size_t)r * cols + c], beta[c], acc);
        acc = wave_sum(acc) + b[r];
    }
    if (lane == 0) out[r] = acc;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#ifndef BESO_ABL_MASK
#define BESO_ABL_MASK 0              // timing experiments only (results are wrong): 1 GELU = identity, 2 no MLP-loop
#endif                               // barriers, 4 no embedding, 8 LayerNorm statistics skipped, 16 no attention core,
                                     // 32 no LDS refills of the activation fragments
// All-reduce over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) on gfx950's lane-swap instructions: two
// VALU swaps instead of the two ds_bpermute_b32 that __shfl_xor(., 16) / (., 32) compile to -- those go through the LDS
// pipe and sit on the serial chain of the attention core (max -> exp -> sum -> 1/sum).  Same pairing as the shuffles:
// (r0 op r1) op (r2 op r3), bit-identical results.
template <bool IS_MAX>
__device__ __forceinline__ float rows_allreduce(float v) {
    const u32x2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float r = IS_MAX ? fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])) : __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const u32x2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r), __float_as_uint(r), false, false);
    return IS_MAX ? fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1])) : __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// two fp32 values -> one dword of GEMM operands (RNE): v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32
__device__ __forceinline__ uint32_t pack_op2(float lo, float hi) {
    f32x2 v = {lo, hi};
#if BESO_OPERAND_F16
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
#else
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
#endif
}

// GELU(v) = v * Phi(v) with the exact-erf Phi of nn.GELU() (score_gpts.py:107), evaluated without
// transcendentals:  Phi(v) - 1/2 = vc * P(vc^2),  vc = clamp(v, +-4),  P a degree-6 minimax polynomial
// fitted to erf (tools/fit_gelu.py): max |error| of GELU over all v is 1.9e-4, an order of magnitude
// below the bf16 rounding (2^-9 relative) that the result receives next.  Two values per call on the
// packed-fp32 pipe (v_pk_mul_f32 / v_pk_fma_f32, constants broadcast from SGPRs): 11 VALU ops per PAIR
// -- VALU issue is what bounds the MLP phase next to the MFMAs.
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 v) {
    if (BESO_ABL_MASK & 1) return v;
    f32x2 vc;
    vc.x = __builtin_amdgcn_fmed3f(v.x, -4.0f, 4.0f);
    vc.y = __builtin_amdgcn_fmed3f(v.y, -4.0f, 4.0f);
    const f32x2 s = vc * vc;
    f32x2 p = __builtin_elementwise_fma(s, (f32x2)(2.277972093e-08f), (f32x2)(-1.598515742e-06f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(4.795382804e-05f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(-0.0008139993719f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(0.00877231165f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(-0.06457294506f));
    p = __builtin_elementwise_fma(p, s, (f32x2)(0.3978832308f));
    const f32x2 u = __builtin_elementwise_fma(vc, p, (f32x2)(0.5f));      // Phi(v)
    return v * u;
}

__device__ __forceinline__ f32x4 mfma_op(const u32x4& a, const u32x4& b, const f32x4& c) {
#if BESO_OPERAND_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                   0, 0, 0);
#endif
}

// v_mfma_f32_16x16x16 on four operands per lane (the attention core's P.V; the half k-step below)
__device__ __forceinline__ f32x4 mfma_op16(const uint2& a, const uint2& b, const f32x4& c) {
#if BESO_OPERAND_F16
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
#endif
}

// Half k-step: contraction over the FIRST 16 indices of a k-step only.  The low 8 bytes of a lane's
// A / B fragment (slots j = 0..3) are exactly the operands of the 16x16x16 MFMA for those indices
// (index 32kk + 4g + j), so a k-step whose upper 16 indices are all padding (D = 360: indices 352..359 of
// 352..383 are real) costs half an MFMA's ENERGY instead of a whole one's: the instruction occupies the matrix pipe as long
// as the full shape does (tools/microbench/power_modes: 145 G instructions/s either way, 1025 W against 1239 W), and the
// kernel runs at the board's power cap.  Measured against full-shape MFMAs on the zero-padded indices (round 3, same-box
// A/B): 0.2 % on the bf16 forward, -0.4 % in BF16X3.  Mixing the two shapes on one accumulator has a price: mixed_chain_pad.
__device__ __forceinline__ f32x4 mfma_op_half(const u32x4& a, const u32x4& b, const f32x4& c) {
    return mfma_op16(make_uint2(a[0], a[1]), make_uint2(b[0], b[1]), c);
}

// MI355X hazard the compiler does not pad (tools/check_mfma_chains.py): an MFMA whose SrcC is exactly the
// vDst of an MFMA of ANOTHER shape (here 16x16x32 -> the 16x16x16 of a half k-step) must not issue within about ten wait
// states of it -- LLVM's recogniser files "same accumulator" under the interlocked back-to-back case and inserts nothing,
// and the consumer then reads the accumulator before the producer has written it (run-to-run different results, found in
// one long-horizon instance where the scheduler had put the two MFMAs two instructions apart).  Called between the last
// full k-step and the half k-step of a phase: every full-shape MFMA stays above, every half-shape one below, ten wait
// states (40 clocks, once per phase) in between.
// Empty token slots (kitchen: 8 of a workgroup's 96, long-horizon: 13 of 80) are given exact zeros as GEMM B operands
// (layernorm_to_lds, mlp_phase): the matrix pipe draws 44 % less dynamic power on zero columns
// (tools/microbench/power_modes mfmaz: 1018 W against 1240 W with half of B's columns zero), and a kernel at the board's
// power cap runs as fast as its energy allows -- same-box A/B, round 3: kitchen B = 4096 -0.9 % (0.8106 vs 0.8177 ms),
// long-horizon Euler-100 -1 ... -2 %.  Only the instances that run at the power cap AND have such slots carry the two
// compare-and-select instructions: the block-push shape fills its tiles (8 x 12 = 96), the latency instances are bound
// by a lone workgroup's weight stream (+0.3 % with the mask).
#ifndef BESO_ZERO_PAD
#define BESO_ZERO_PAD 1              // 0: A/B builds
#endif
#ifndef BESO_KEEP_HYBRID
#define BESO_KEEP_HYBRID 1           // odd token-tile counts: h and GELU(h) leave as 16-byte pieces for the tile pairs, 8-byte ones for the last (0: all 8-byte)
#endif
#ifndef BESO_TRAIN_FWD_ABL
#define BESO_TRAIN_FWD_ABL 0         // timing experiments on train_fwd_kernel (results wrong): stores left out -- 1 x_mid / x_out,
#endif                               // 2 LayerNorm outputs + statistics, 4 q|k|v and y, 8 h, 16 GELU(h); 128 / 256: rows folded (Rows::row)
__host__ __device__ constexpr bool zero_pad_instance(int RPW, int NT) { return BESO_ZERO_PAD && RPW != 2 && NT >= 5; }
__device__ __forceinline__ void mixed_chain_pad() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 4\n\ts_nop 4");
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------
// One GEMM phase of the transposed formulation:  acc[r][t] += sum_kk A(r,kk) * B(t,kk)
//   A(r,kk) = fragment r + kk*a_ks of `a`   weights, L2 -> registers (WPtr: uniform base + lane offset)
//   B(t,kk) = b[t*b_ts + kk*b_ks]  activations, LDS -> registers (pointer already offset by the lane)
// Two k-steps per iteration with named even/odd weight-fragment registers; the loads that refill a
// register set are issued right behind the MFMAs that consumed it, so weight fragments are ~1.5
// k-steps and LDS fragments half a k-step ahead of their use without copies.  The sched_barriers pin that order
// (left alone, the scheduler sinks the loads next to their consumers and serialises on vmcnt(0)).
// aE/aO must already hold k-steps 0 and 1 (prefetch_a), which lets the caller issue them early.
// ksteps must be even.
// ---------------------------------------------------------------------------------------------
#ifndef BESO_FUSED_ABLATE
#define BESO_FUSED_ABLATE 0          // 1: every weight-fragment load hits the same few KiB (timing experiment only)
#endif
#if BESO_FUSED_ABLATE == 1
#define ABL_KS(x) 0
#define ABL_PTR(base, off) (base)
#else
#define ABL_KS(x) (x)
#define ABL_PTR(base, off) ((base).adv(off))
#endif

#ifndef BESO_FUSED_WLOAD
#define BESO_FUSED_WLOAD 0           // 0: plain loads, 1: non-temporal (measured 11 % SLOWER: 1.39 vs 1.26 ms)
#endif
__device__ __forceinline__ u32x4 wload(const u32x4* p) {
#if BESO_FUSED_WLOAD == 1
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// Address of a wave's weight fragments: a buffer resource over a wave-uniform base (SGPRs: kernel
// arguments, wave index), a uniform byte offset (SGPR: k-step / chunk / head stepping is scalar arithmetic)
// and the lane's byte offset inside a 1 KiB fragment (one VGPR).  `buffer_load_dwordx4 v, v_lo, s[rsrc], s_off
// offen` needs no per-load VALU address arithmetic; with per-lane 64-bit pointers (`global_load ... v[a:a+1]`)
// every step was 2-3 VALU instructions plus two VGPRs, which then got spilled.
struct WPtr {
    __amdgpu_buffer_rsrc_t rs;
    uint32_t so;         // uniform byte offset
    uint32_t lo;         // lane * 16
    __device__ __forceinline__ u32x4 at(int frag) const {
        return __builtin_amdgcn_raw_buffer_load_b128(rs, lo, so + (uint32_t)frag * 1024u, 0);
    }
    __device__ __forceinline__ WPtr adv(size_t frags) const { return WPtr{rs, so + (uint32_t)frags * 1024u, lo}; }
};
// 0x00020000: raw buffer, 32-bit data format field as CK / rocPRIM set it for gfx9; range checking is unused
__device__ __forceinline__ WPtr wptr(const u32x4* uniform_base, int lane) {
    return WPtr{__builtin_amdgcn_make_buffer_rsrc((void*)uniform_base, 0, 0x7fffffff, 0x00020000), 0u, (uint32_t)lane * 16u};
}

template <int R>
__device__ __forceinline__ void prefetch_a(u32x4 (&aE)[R], u32x4 (&aO)[R], WPtr a, int a_ks) {
#pragma unroll
    for (int r = 0; r < R; ++r) { aE[r] = a.at(r); aO[r] = a.at(r + ABL_KS(a_ks)); }
}

template <int R, int NT, bool TAIL16 = false, int NTA = NT>      // NTA: token tiles the accumulator array holds (>= NT used)
__device__ __forceinline__ void gemm_phase(f32x4 (&acc)[R][NTA], u32x4 (&aE)[R], u32x4 (&aO)[R],
                                           WPtr a, int a_ks, const u32x4* b, int b_ts,
                                           int b_ks, int ksteps) {
    // ONE set of B fragments: each half of it is refilled for the next k-step as soon as the MFMAs that
    // read it have been issued (prefetch distance = half a k-step of MFMAs, enough for LDS latency).
    constexpr int H1 = (NT + 1) / 2;
    u32x4 bf[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bf[t] = b[t * b_ts];
    for (int kk = 0; kk < ksteps; kk += 2) {
        // ---- k-step kk (even fragments)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < H1; ++t)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][t] = mfma_op(aE[r], bf[t], acc[r][t]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < H1; ++t) if (!(BESO_ABL_MASK & 32)) bf[t] = b[t * b_ts + (kk + 1) * b_ks];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = H1; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][t] = mfma_op(aE[r], bf[t], acc[r][t]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = H1; t < NT; ++t) if (!(BESO_ABL_MASK & 32)) bf[t] = b[t * b_ts + (kk + 1) * b_ks];
        if (kk + 2 < ksteps) {
#pragma unroll
            for (int r = 0; r < R; ++r) aE[r] = a.at(r + ABL_KS((kk + 2) * a_ks));
        }
        // ---- k-step kk+1 (odd fragments); TAIL16: the last k-step of the phase is a half k-step
        const bool tail = TAIL16 && kk + 2 >= ksteps;
        __builtin_amdgcn_sched_barrier(0);
        if (tail) {
            mixed_chain_pad();
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r][t] = mfma_op_half(aO[r], bf[t], acc[r][t]);
            break;
        }
#pragma unroll
        for (int t = 0; t < H1; ++t)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][t] = mfma_op(aO[r], bf[t], acc[r][t]);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 2 < ksteps && !(BESO_ABL_MASK & 32)) {
#pragma unroll
            for (int t = 0; t < H1; ++t) bf[t] = b[t * b_ts + (kk + 2) * b_ks];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = H1; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][t] = mfma_op(aO[r], bf[t], acc[r][t]);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 2 < ksteps && !(BESO_ABL_MASK & 32)) {
#pragma unroll
            for (int t = H1; t < NT; ++t) bf[t] = b[t * b_ts + (kk + 2) * b_ks];
        }
        if (kk + 3 < ksteps) {
#pragma unroll
            for (int r = 0; r < R; ++r) aO[r] = a.at(r + ABL_KS((kk + 3) * a_ks));
        }
    }
}

// Variant for the phases with short k-steps (2-3 row tiles x 3-6 token tiles per wave): a ring of PFA
// k-steps of weight fragments (refilled PFA k-steps ahead: the L2 round trip is longer than one short
// k-step of MFMAs) and two sets of B fragments (a whole k-step of LDS prefetch distance).
// ksteps must be a multiple of PFA, PFA even.
template <int R, int PFA>
__device__ __forceinline__ void prefetch_ring(u32x4 (&ar)[PFA][R], WPtr a, int a_ks) {
#pragma unroll
    for (int p = 0; p < PFA; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r) ar[p][r] = a.at(r + ABL_KS(p * a_ks));
}

template <int R, int NT, int PFA, bool TAIL16 = false>
__device__ __forceinline__ void gemm_phase_ring(f32x4 (&acc)[R][NT], u32x4 (&ar)[PFA][R],
                                                WPtr a, int a_ks, const u32x4* b, int b_ts,
                                                int b_ks, int ksteps) {
    static_assert(PFA % 2 == 0, "B fragments alternate between two sets");
    u32x4 bb[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bb[0][t] = b[t * b_ts];
#pragma unroll
    for (int t = 0; t < NT; ++t) bb[1][t] = b[t * b_ts + b_ks];
    for (int k0 = 0; k0 < ksteps; k0 += PFA) {
#pragma unroll
        for (int p = 0; p < PFA; ++p) {
            const int kk = k0 + p;
            __builtin_amdgcn_sched_barrier(0);
            if (TAIL16 && kk + 1 >= ksteps) {                // the last k-step of the phase is a half k-step
                mixed_chain_pad();
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r][t] = mfma_op_half(ar[p][r], bb[p & 1][t], acc[r][t]);
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r][t] = mfma_op(ar[p][r], bb[p & 1][t], acc[r][t]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (BESO_FUSED_ABLATE != 2 && !(BESO_ABL_MASK & 32) && kk + 2 < ksteps) {
#pragma unroll
                for (int t = 0; t < NT; ++t) bb[p & 1][t] = b[t * b_ts + (kk + 2) * b_ks];
            }
            if (BESO_FUSED_ABLATE != 3 && kk + PFA < ksteps) {
#pragma unroll
                for (int r = 0; r < R; ++r) ar[p][r] = a.at(r + ABL_KS((kk + PFA) * a_ks));
            }
        }
    }
}

// gemm_phase_ring whose weight ring never drains (round 6, the training step's data-gradient kernels): the refills of the
// LAST PFA k-steps fetch the first PFA k-steps of the NEXT phase's image (`nxt`, fragment stride nxt_ks; both images behind the
// same buffer resource), so the next phase starts with its ring in flight instead of one L2 round trip (~0.8 us: a lone
// workgroup per CU has nothing to hide it behind) -- and those requests are OLDER than whatever the epilogue in between stores
// (a wave's memory operations retire in order: a ring requested behind a store burst waits for the burst).  The ring must be
// full on entry (prefetch_ring, or the phase before).  No next phase: nxt.lo carries bit 31 -- beyond the resource's range, the
// loads are issued (static vmcnt bookkeeping) and return zeros without touching memory.  ksteps a multiple of PFA, PFA even.
template <int R, int NT, int PFA>
__device__ __forceinline__ void gemm_phase_ring_cont(f32x4 (&acc)[R][NT], u32x4 (&ar)[PFA][R], WPtr a, int a_ks, const u32x4* b,
                                                     int b_ts, int b_ks, int ksteps, WPtr nxt, int nxt_ks) {
    static_assert(PFA % 2 == 0, "B fragments alternate between two sets");
    u32x4 bb[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bb[0][t] = b[t * b_ts];
#pragma unroll
    for (int t = 0; t < NT; ++t) bb[1][t] = b[t * b_ts + b_ks];
    for (int k0 = 0; k0 < ksteps; k0 += PFA) {
        const bool tail = k0 + PFA >= ksteps;                 // (uniform) this block's refills belong to the next phase
        const uint32_t so0 = tail ? nxt.so : a.so + (uint32_t)((k0 + PFA) * a_ks) * 1024u;
        const uint32_t ksb = (uint32_t)(tail ? nxt_ks : a_ks) * 1024u;
        const uint32_t lo = tail ? nxt.lo : a.lo;
#pragma unroll
        for (int p = 0; p < PFA; ++p) {
            const int kk = k0 + p;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r][t] = mfma_op(ar[p][r], bb[p & 1][t], acc[r][t]);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < ksteps) {
#pragma unroll
                for (int t = 0; t < NT; ++t) bb[p & 1][t] = b[t * b_ts + (kk + 2) * b_ks];
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                ar[p][r] = __builtin_amdgcn_raw_buffer_load_b128(a.rs, lo, so0 + (uint32_t)p * ksb + (uint32_t)r * 1024u, 0);
        }
    }
}
__device__ __forceinline__ WPtr wptr_none(WPtr a) { return WPtr{a.rs, 0u, a.lo | 0x80000000u}; }

// ---------------------------------------------------------------------------------------------
// BF16X3: split-bf16 arithmetic on the same MFMA (`PX` = 1 instances of the phases).  Every GEMM operand is a pair
// (hi, lo) of bf16 values with hi = bf16(v), lo = bf16(v - hi): hi + lo = v to 2^-16 relative, and
//   acc += A_lo B_hi + A_hi B_lo + A_hi B_hi          (fp32 accumulate; the lo*lo term is below the representation error)
// -- three v_mfma_f32_16x16x32_bf16 per fragment pair, fp32-class results from the bf16 matrix pipe at a third of its
// rate (the fp32-input MFMA runs at 1/16).  Weight fragments: the low image lies x3_delta bytes behind the bf16 image
// (FusedDims); activation fragments: the low fragment `b_lo` u32x4 behind the high one in LDS.
// ---------------------------------------------------------------------------------------------
struct SplitPair { uint32_t hi, lo; };
__device__ __forceinline__ SplitPair split_op2(float a, float b) {
    SplitPair p;
    p.hi = pack_op2(a, b);
#if BESO_OPERAND_F16
    const f32x2 back = __builtin_convertvector(__builtin_bit_cast(f16x2, p.hi), f32x2);
    p.lo = pack_op2(a - back.x, b - back.y);
#else
    p.lo = pack_op2(a - __uint_as_float(p.hi << 16), b - __uint_as_float(p.hi & 0xffff0000u));
#endif
    return p;
}

// acc[r][t] += sum_kk A(r,kk) B(t,kk) as gemm_phase_ring, operands split: a ring of PFA k-steps of weight fragments
// (a lone workgroup's weight stream is latency bound: bytes in flight per wave are what sets its rate) and two sets of
// B fragments.  Every refill is unconditional (a load under a branch makes the compiler drain every load in flight
// behind it); the surplus ones at the end of the phase are made free.  ksteps even; TAIL16: the last k-step is a half k-step.
template <int R, int NT, int PFA, bool TAIL16 = false, int NTA = NT>
__device__ __forceinline__ void gemm_x3(f32x4 (&acc)[R][NTA], WPtr a, int a_ks, uint32_t x3_delta, const u32x4* b, int b_lo,
                                        int b_ts, int b_ks, int ksteps) {
    static_assert(PFA % 2 == 0, "B fragments alternate between two sets");
    const WPtr al{a.rs, a.so + x3_delta, a.lo};
    u32x4 ah[PFA][R], alo[PFA][R], bh[2][NT], bl[2][NT];
    auto load_a = [&](int p, int kk) {
        // refills past the last k-step go out with a lane offset beyond the buffer's range: the load instruction
        // is issued (the vmcnt bookkeeping stays static) but returns zeros without touching memory
        const uint32_t vo = kk < ksteps ? a.lo : (a.lo | 0x80000000u);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            ah[p][r] = __builtin_amdgcn_raw_buffer_load_b128(a.rs, vo, a.so + (uint32_t)(r + kk * a_ks) * 1024u, 0);
            alo[p][r] = __builtin_amdgcn_raw_buffer_load_b128(al.rs, vo, al.so + (uint32_t)(r + kk * a_ks) * 1024u, 0);
        }
    };
    auto load_b = [&](int q, int kk) {
        kk = min(kk, ksteps - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) { bh[q][t] = b[t * b_ts + kk * b_ks]; bl[q][t] = b[b_lo + t * b_ts + kk * b_ks]; }
    };
#pragma unroll
    for (int p = 0; p < PFA; ++p) load_a(p, p);
    load_b(0, 0);
    load_b(1, 1);
    for (int k0 = 0; k0 < ksteps; k0 += PFA) {
#pragma unroll
        for (int p = 0; p < PFA; ++p) {
            const int kk = k0 + p;
            if (kk < ksteps) {
                // the small terms first; R*NT independent accumulators between two MFMAs on the same one
                if (TAIL16 && kk + 1 >= ksteps) {
                    mixed_chain_pad();
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r][t] = mfma_op_half(alo[p][r], bh[p & 1][t], acc[r][t]);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r][t] = mfma_op_half(ah[p][r], bl[p & 1][t], acc[r][t]);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r][t] = mfma_op_half(ah[p][r], bh[p & 1][t], acc[r][t]);
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r][t] = mfma_op(alo[p][r], bh[p & 1][t], acc[r][t]);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r][t] = mfma_op(ah[p][r], bl[p & 1][t], acc[r][t]);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r][t] = mfma_op(ah[p][r], bh[p & 1][t], acc[r][t]);
                }
            }
            load_b(p & 1, kk + 2);
            load_a(p, kk + PFA);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// shared pieces of the kernels
// ---------------------------------------------------------------------------------------------
constexpr int kKCc = kChunkTiles / 2;      // FC2 k-steps per hidden chunk (= kKC below)
#ifndef BESO_RED_PAD
#define BESO_RED_PAD 4               // floats of padding per token in the LayerNorm statistics exchange (0: A/B)
#endif
// LayerNorm statistics exchange: per token kWaves (sum, sum of squares) pairs + padding -- with 16 floats per token the 16
// lanes of a row hit 4 banks groups four times over (reads of 16 B at a 64-B stride); 20 floats spread them over all 64 banks
constexpr int kRedTok = 2 * kWaves + BESO_RED_PAD;
constexpr int kXsBytes = 2048;        // the workgroup's action windows (n_real x t x act <= 512 floats): input of every evaluation
// (every instance stages them: the bound is what fused_level admits -- kSPW samples x (kMT / kSPW tokens >= 2 t) x 4 kEmbActK
//  action dims, or one long-sequence sample of 16 kLongNT / 2 steps)
struct LdsMap {            // byte offsets inside the dynamic LDS block
    int xnT, u, red, tab, xs, total;
    int cfgc;              // long-sequence instance: the conditional pass's head outputs (classifier-free pairs run as two passes)
};
__host__ __device__ constexpr LdsMap lds_map(int KS, bool mlp_only = false, int NT = kNTT) {
    // u is the phase-local region: attention (q/k/v 3*104*72*2 = 44928 | yT 12288) or MLP (hT 6*8 KiB = 49152);
    // the MLP-block kernel needs only the latter (which lets D = 512, KS = 16: 96 KiB of xnT, fit in 160 KiB)
    LdsMap m{};
    m.xnT = 0;
    m.u = NT * KS * 1024;                  // (NT < kNTT: instances that only ever touch the first NT token tiles)
    // phase-local region; the long-sequence instance (NT = kLongNT) keeps q/k/v of two heads: 2 x 3 x 16 NT rows of 144 B
    m.red = m.u + (mlp_only ? kNTT * kKCc * 1024 : (NT == kLongNT ? 2 * 3 * 16 * kLongNT * kQKVRow * 2 : 59648));
    m.tab = m.red + kRedTok * kMT * 4;
    m.xs = m.tab + 512;
    m.cfgc = m.xs + kXsBytes;
    m.total = m.cfgc + (NT == kLongNT ? kXsBytes : 0);
    return m;
}

// LDS of the BF16X3 instances (NT token tiles): every B-fragment region twice (hi | lo), q/k/v of one head as fp32 rows
// (the attention core runs on the exact-fp32 MFMA there).  yT sits behind both phase-local regions, so that its
// never-rewritten entries (padding tokens) keep the zeros of the prologue; the pad rows of q/k/v alias hT, i.e. pairs
// of finite bf16 values = finite fp32 values, and only ever meet zero probabilities.
constexpr int kQKVRowF = kHDP + 4;         // fp32 elements per q/k/v row: 272 B, conflict-free b128 reads down a column of rows
struct LdsMapX3 {
    int xn_lo, u, qkv_rows, yT, y_lo, h_lo, red, tab, xs, total;     // *_lo: distance hi -> lo fragment in u32x4 units; rest bytes
};
__host__ __device__ constexpr LdsMapX3 lds_map_x3(int KS, int NT) {
    LdsMapX3 m{};
    const int xn_bytes = NT * KS * 1024, h_bytes = NT * kKCc * 1024, y_bytes = NT * 2 * 1024;
    m.xn_lo = xn_bytes / 16;
    m.h_lo = h_bytes / 16;
    m.y_lo = y_bytes / 16;
    m.u = 2 * xn_bytes;
    m.qkv_rows = 16 * NT + 8;
    const int qkv_bytes = 3 * m.qkv_rows * kQKVRowF * 4;
    const int front = qkv_bytes > 2 * h_bytes ? qkv_bytes : 2 * h_bytes;
    m.yT = (front + 1023) / 1024 * 1024;                          // relative to u
    const int need = m.yT + 2 * y_bytes, head = kWaves * kMT * 16 * 4;
    m.red = m.u + (need > head ? need : head);
    m.tab = m.red + kRedTok * kMT * 4;
    m.xs = m.tab + 512;
    m.total = m.xs + kXsBytes;
    return m;
}

// Optional phase timing (development aid): when a stamp buffer is installed (beso_debug_set_stamps),
// thread 0 of workgroup 0 appends {phase id, s_memtime} pairs.
#ifndef BESO_FUSED_STAMPS
#define BESO_FUSED_STAMPS 0          // build with -DBESO_FUSED_STAMPS=1 for tools/phase_stamps.py
#endif
struct Stamps {
    unsigned long long* buf;
    int cap;
    int n;
};
// KIND: which kernels of a -DBESO_FUSED_STAMPS=<kind> build record (every launch starts at the head of the buffer, so a build
// stamps one family): 1 the forward kernels, 2 train_mlp_bwd_kernel, 3 train_dgrad_kernel (tools/train_stamps.py)
template <int KIND = 1>
__device__ __forceinline__ void stamp(Stamps& st, int id) {
    // lane 0 of every wave of workgroup 0 records into its own eighth of the buffer
    if (BESO_FUSED_STAMPS == KIND && st.buf && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && st.n + 2 <= st.cap / 8) {
        unsigned long long* b = st.buf + (size_t)(threadIdx.x >> 6) * (st.cap / 8);
        b[st.n] = (unsigned long long)id;
        // ids >= 100 record the constant-rate (100 MHz) counter instead of the shader clock: the pair gives the
        // core frequency the kernel actually ran at
        b[st.n + 1] = id >= 100 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
        st.n += 2;
    }
}

// Hooks of the training-forward instance of the block kernels (train_tail_kernel): the phases below can apply the
// LayerNorm affine themselves (training weights are not gamma-folded) and store what the backward pass keeps -- LayerNorm
// statistics and output, the FC1 pre-activation h and GELU(h) -- row-major [token][feature] as the per-op training kernels
// do.  The default types switch all of it off at compile time: the inference instances are unchanged.
// Which row of the kept [rows][features] buffers a token slot of the tile maps to: slot < n ? base + (tab ? tab[slot] : slot)
// : none.  tab = nullptr: the tile holds consecutive rows (train_tail_kernel; the compact action rows of the last layer in
// train_fwd_kernel); tab = SlotTabs::row_of_slot: the action-tokens-first slot order of the one-launch kernels.
struct Rows {
    const unsigned char* tab; int base, n;
    __device__ __forceinline__ int row(int slot) const {
        int r = slot < n ? base + (tab ? (int)tab[slot] : slot) : -1;
        // (timing experiments, results wrong: every workgroup's kept rows folded onto the first 64 / 4096 rows of each buffer --
        //  the same store instructions onto an L2-resident / a page-local footprint; profiles/r06_store_wave_probe.txt)
        if ((BESO_TRAIN_FWD_ABL & 128) && r >= 0) r &= 63;
        if ((BESO_TRAIN_FWD_ABL & 256) && r >= 0) r &= 4095;
        return r;
    }
};
// Two 8-byte row pieces of the accumulator layout -> one 16-byte store per lane.  a = the lane's four bf16 features (f0 .. f0 + 3,
// f0 = 16 tile + 4 g) of the token `tok_a` of token tile t, b = the same features of `tok_b` of tile t + 1.  After the two swaps
// (v_permlane16_swap: the odd 16-lane rows of the first operand <-> the even rows of the second) a lane of an even group g holds
// [its own a | group g + 1's a] = features f0 .. f0 + 7 of tok_a, a lane of an odd group [group g - 1's b | its own b] =
// features f0 - 4 .. f0 + 3 of tok_b: every lane stores 16 bytes to ONE row.  ld (bf16 elements per row) is a multiple of 8.
__device__ __forceinline__ void store_pair16(uint16_t* __restrict__ base, int ld, int tok_a, int tok_b, int f0, int g, uint2 a, uint2 b,
                                             bool on) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
    const u32x2_t sx = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
    const u32x2_t sy = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
    const int tok = (g & 1) ? tok_b : tok_a, f = f0 - 4 * (g & 1);
    if (tok >= 0 && f < ld && on) *(u32x4*)(base + (size_t)tok * ld + f) = u32x4{sx[0], sy[0], sx[1], sy[1]};
}
struct LnPlain { static constexpr bool on = false; };
struct LnTrain {
    static constexpr bool on = true;
    const float* gamma; const float* beta;    // fp32, zero padded to the tile's feature count
    float* stats;                             // [M][2] (mean, rstd)
    uint16_t* xn;                             // [M][D] bf16, LayerNorm output with the affine applied
    Rows rows; int D;
};
struct MlpPlain { static constexpr bool on = false; };
struct MlpTrain {
    static constexpr bool on = true;
    uint16_t* h; uint16_t* g;                 // [M][ld] bf16: FC1 pre-activation (with bias), GELU of it
    Rows rows; int ld;                        // ld = 4 D = number of real hidden features
};
// Attention phase of the training forward: q | k | v rows and the attention output go out row-major for the backward pass
// (attn_small_kernel<BWD> recomputes the probabilities from the kept q/k/v), dropout on the probabilities with the mask of
// the per-op training kernels (hash of (seed, site, ((b H + h) T + i) T + j)).
struct AttnPlain { static constexpr bool on = false; };
struct AttnTrain {
    static constexpr bool on = true;
    uint16_t* qkv;                            // [M][3 D] bf16, columns [q | k | v], heads in natural order
    uint16_t* y;                              // [rows][D] bf16
    Rows rows, rows_y;                        // all token rows; rows of y (all, or the compact action rows of the last layer)
    int D, s0, H;                             // first sample of the workgroup, real heads
    float p, inv_keep; uint32_t seed, site;
};

template <int RPW>
struct Tile {
    f32x4 acc[RPW][kNTT];
    bool fvalid[RPW];
};

template <int RPW, int NT = kNTT>
__device__ __forceinline__ void load_x_tile(Tile<RPW>& T, const float* __restrict__ x, int D, int m0, int m_end,
                                            int w, int n, int g) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int f0 = 16 * (w * RPW + i) + 4 * g;
        T.fvalid[i] = f0 < D;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tok = m0 + t * 16 + n;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (T.fvalid[i] && tok < m_end) v = *(const f32x4*)(x + (size_t)tok * D + f0);
            T.acc[i][t] = v;
        }
    }
}

template <int RPW, int NT = kNTT>
__device__ __forceinline__ void store_x_tile(const Tile<RPW>& T, float* __restrict__ x, int D, int m0, int m_end,
                                             int w, int n, int g) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int f0 = 16 * (w * RPW + i) + 4 * g;
        if (!T.fvalid[i]) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tok = m0 + t * 16 + n;
            if (tok < m_end) *(f32x4*)(x + (size_t)tok * D + f0) = T.acc[i][t];
        }
    }
}

// The same through a slot -> row map (train_fwd_kernel: action tokens first, or the compact action rows of the last layer).
template <int RPW, int NT>
__device__ __forceinline__ void load_x_rows(Tile<RPW>& T, const float* __restrict__ x, int D, const Rows& rows, int w, int lane) {
    const int n = lane & 15, g = lane >> 4;
    int row[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) row[t] = rows.row(16 * t + n);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int f0 = 16 * (w * RPW + i) + 4 * g;
        T.fvalid[i] = f0 < D;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (T.fvalid[i] && row[t] >= 0) v = *(const f32x4*)(x + (size_t)row[t] * D + f0);
            T.acc[i][t] = v;
        }
    }
}
// Residual add behind a dropout (nn.Dropout(resid_pdrop) on the out-projection and on the MLP output, score_gpts.py:79,109,
// 113-114): T holds the branch output (+ bias) alone, the residual it is added to lies in memory (the previous kept x), and
// element (row, f) of the branch keeps drop_scale(seed, site, row D + f) -- the per-op forward's EpiResid and the backward
// kernels' epilogues evaluate the same hash.  rows_x: where the residual's rows lie; rows_m: the rows the mask is indexed by
// (the same, or the compact action rows of the last layer).
template <int RPW, int NT>
__device__ __forceinline__ void resid_dropout_add(Tile<RPW>& T, const float* __restrict__ x, int D, const Rows& rows_x,
                                                  const Rows& rows_m, float p, float inv_keep, uint32_t seed, uint32_t site,
                                                  int w, int lane) {
    asm volatile("" : "+v"(lane));
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int f0 = 16 * (w * RPW + i) + 4 * g;
        if (f0 >= D) continue;                                   // (padding features stay exact zeros: zero weights, zero bias)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int rx = rows_x.row(16 * t + n), rm = rows_m.row(16 * t + n);
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if (rx >= 0) xv = *(const f32x4*)(x + (size_t)rx * D + f0);
            if (rm >= 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) T.acc[i][t][j] *= drop_scale(seed, site, (size_t)rm * D + f0 + j, p, inv_keep);
            }
            T.acc[i][t] += xv;
        }
    }
}
template <int RPW, int NT>
__device__ __forceinline__ void set_bias_rows(Tile<RPW>& T, const float* __restrict__ bias, int w, int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const f32x4 bv = *(const f32x4*)(bias + 16 * (w * RPW + i) + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t) T.acc[i][t] = bv;
    }
}
// ... as bf16 rows [rows][D] (the one-launch training forward without residual dropout: the kept x_mid / x_out are read by
// the LayerNorm backward alone, whose use of them -- x_hat = (x - mean) rstd inside two row sums -- does not need more than
// the 8 significant bits the other kept activations have; half the bytes of what was 22 % of the forward's stores)
template <int RPW, int NT>
__device__ __forceinline__ void store_x_rows_bf16(const Tile<RPW>& T, uint16_t* __restrict__ x, int D, const Rows& rows, int w, int lane) {
    asm volatile("" : "+v"(lane));
    const int n = lane & 15, g = lane >> 4;
    int row[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) row[t] = rows.row(16 * t + n);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int f0 = 16 * (w * RPW + i) + 4 * g;
        if (f0 >= D) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (row[t] >= 0 && !(BESO_TRAIN_FWD_ABL & 1))
                *(uint2*)(x + (size_t)row[t] * D + f0) = make_uint2(pack_op2(T.acc[i][t][0], T.acc[i][t][1]), pack_op2(T.acc[i][t][2], T.acc[i][t][3]));
    }
}
template <int RPW, int NT>
__device__ __forceinline__ void store_x_rows(const Tile<RPW>& T, float* __restrict__ x, int D, const Rows& rows, int w, int lane) {
    asm volatile("" : "+v"(lane));
    const int n = lane & 15, g = lane >> 4;
    int row[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) row[t] = rows.row(16 * t + n);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int f0 = 16 * (w * RPW + i) + 4 * g;
        if (f0 >= D) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (row[t] >= 0 && !(BESO_TRAIN_FWD_ABL & 1)) *(f32x4*)(x + (size_t)row[t] * D + f0) = T.acc[i][t];
    }
}

// LayerNorm statistics of every token of the tile over the feature slices of all 8 waves: per-wave
// partial (sum, sum of squares) in fp32, exchanged through LDS behind ONE barrier; var = E[x^2] - mean^2
// (the residual stream is O(1..10) with |mean| << std, and the result is rounded to bf16 next: the
// cancellation is far below that rounding).  Invalid (padding) features hold exact zeros.
template <int RPW, int NW, int NT = kNTT>         // NT: the first NT token tiles only
__device__ __forceinline__ void ln_stats(const Tile<RPW>& T, float* red, int D, int w, int lane, float (&mean)[NT],
                                         float (&rstd)[NT], Stamps& st) {
    // (every multiply-add below is written out: which products the compiler contracts must not depend on the instance this is
    // inlined into -- the instances of the kernel agree bit for bit)
#pragma clang fp contract(off)
    const int n = lane & 15, row = lane >> 4;            // (token tiles are reduced in pairs; an odd last one pairs with zeros)
    const float invD = 1.0f / (float)D;
    // Cross-lane part of the reduction (over the four 16-lane rows g) on gfx950's lane-swap instructions:
    //   v_permlane32_swap(s, q) + add : rows {0,1} = s(g) + s(g+2), rows {2,3} = q(g) + q(g+2)
    //   v_permlane16_swap(r_t, r_t+1) + add : row 0 = S_t, row 1 = S_t+1, row 2 = Q_t, row 3 = Q_t+1
    // so every lane ends up with one finished (token, statistic) and writes it: red[token][wave][stat].
#pragma unroll
    for (int tp = 0; tp < (NT + 1) / 2; ++tp) {
        float r[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = 2 * tp + h < NT ? 2 * tp + h : NT - 1;
            float s = 0.f, q = 0.f;
            if (2 * tp + h < NT) {
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float v = T.acc[i][t][k]; s += v; q = fmaf(v, v, q); }
            }
            }
            const u32x2 x = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(q), false, false);
            r[h] = __uint_as_float(x[0]) + __uint_as_float(x[1]);
        }
        const u32x2 y = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[0]), __float_as_uint(r[1]), false, false);
        const float v = __uint_as_float(y[0]) + __uint_as_float(y[1]);
        const int tok = (2 * tp + (row & 1)) * 16 + n;      // (odd NT: the last pair's second token tile is tile NT < kNTT, unused)
        red[tok * kRedTok + w * 2 + (row >> 1)] = v;
    }
    stamp(st, 30);
    __syncthreads();
    stamp(st, 31);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4* pr = (const f32x4*)(red + (size_t)(t * 16 + n) * kRedTok);
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < NW / 2; ++k) { const f32x4 v = pr[k]; s += v[0] + v[2]; q += v[1] + v[3]; }
        mean[t] = s * invD;
        rstd[t] = __builtin_amdgcn_rsqf(fmaxf(fmaf(-mean[t], mean[t], q * invD), 0.f) + 1e-5f);     // v_rsq_f32 (1 ulp); 1/sqrtf is ~35 VALU ops
    }
}

// LayerNorm without affine (gamma/beta are folded into the consumer): (x - mean) * rstd as bf16 B
// fragments into xnT.  Ends with a barrier; `bias` (the residual-add bias of the block's last Linear) is
// added to the residual after the normalised copy has been taken.  A wave's row tiles (Rf, Rf+1) with Rf
// even are the two halves of one k-step fragment and go out as one 16-byte LDS write per lane.
// NOTE: the red[] buffer is re-used by the next LayerNorm; the barrier at the end of this function (and
// the phases in between) orders the reads above against those writes.
// PX = 1 (BF16X3): the normalised values go out as split-bf16 pairs, the low fragment `lo_off` u32x4 behind the high one.
template <int RPW, int KS, int NW, bool ADD_BIAS = true, int NT = kNTT, int PX = 0, class LX = LnPlain>
__device__ __forceinline__ void layernorm_to_lds(Tile<RPW>& T, u32x4* xnT, float* red, int D, int w, int lane,
                                                 const float* __restrict__ bias, Stamps& st, int lo_off = 0,
                                                 const LX lx = LX{}, int n_valid = 1 << 30) {
    // `lane` is made opaque at the top of every phase: otherwise the per-lane address arithmetic of ALL
    // phases is hoisted out of the layer loop and kept live across it (46 spilled VGPRs).
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4;
    float mean[NT], rstd[NT];
    if (BESO_ABL_MASK & 8) {
#pragma unroll
        for (int t = 0; t < NT; ++t) { mean[t] = 0.01f * lane; rstd[t] = 0.5f; }
    } else ln_stats<RPW, NW, NT>(T, red, D, w, lane, mean, rstd, st);
    // empty token slots of the last tile (slots >= n_valid): rstd = 0, i.e. exact zeros as B operands (zero_pad_instance);
    // nothing a real token reads changes
    if (zero_pad_instance(RPW, NT)) rstd[NT - 1] = 16 * (NT - 1) + (lane & 15) < n_valid ? rstd[NT - 1] : 0.f;
    // Padding features (>= D) are NOT masked here: their xnT entries only ever meet the zero-padded
    // contraction columns of the packed QKV / FC1 weights, and (0 - mean) * rstd is finite.
    f32x4 gam[LX::on ? RPW : 1], bet[LX::on ? RPW : 1];
    if constexpr (LX::on) {
        const int n = lane & 15;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            gam[i] = *(const f32x4*)(lx.gamma + 16 * (w * RPW + i) + 4 * g);
            bet[i] = *(const f32x4*)(lx.beta + 16 * (w * RPW + i) + 4 * g);
        }
        if (w == 0 && g == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tok = lx.rows.row(16 * t + n);
                if (tok >= 0 && !(BESO_TRAIN_FWD_ABL & 2)) *(float2*)(lx.stats + 2 * (size_t)tok) = make_float2(mean[t], rstd[t]);
            }
        }
    }
    auto half = [&](int i, int t) {
        const float a = rstd[t], b = -mean[t] * rstd[t];
        uint2 pk;
        if constexpr (LX::on) {
            // gamma / beta applied here (zero for padding features: their xnT entries are exact zeros), and the
            // operand-typed copy for the weight gradients goes out row-major
            pk.x = pack_op2(fmaf(fmaf(T.acc[i][t][0], a, b), gam[i][0], bet[i][0]), fmaf(fmaf(T.acc[i][t][1], a, b), gam[i][1], bet[i][1]));
            pk.y = pack_op2(fmaf(fmaf(T.acc[i][t][2], a, b), gam[i][2], bet[i][2]), fmaf(fmaf(T.acc[i][t][3], a, b), gam[i][3], bet[i][3]));
            // (the kept copy leaves in keep_xn below: 16-byte pieces, two token tiles per store)
            if constexpr (NT % 2 != 0) {
                const int tok = lx.rows.row(16 * t + (lane & 15)), f0 = 16 * (w * RPW + i) + 4 * g;
                if (tok >= 0 && f0 < lx.D && !(BESO_TRAIN_FWD_ABL & 2)) *(uint2*)(lx.xn + (size_t)tok * lx.D + f0) = pk;
            }
        } else {
            pk.x = pack_op2(fmaf(T.acc[i][t][0], a, b), fmaf(T.acc[i][t][1], a, b));
            pk.y = pack_op2(fmaf(T.acc[i][t][2], a, b), fmaf(T.acc[i][t][3], a, b));
        }
        return pk;
    };
    auto half_x3 = [&](int i, int t, uint2& hi, uint2& lo) {
        const float a = rstd[t], b = -mean[t] * rstd[t];
        const SplitPair p0 = split_op2(fmaf(T.acc[i][t][0], a, b), fmaf(T.acc[i][t][1], a, b));
        const SplitPair p1 = split_op2(fmaf(T.acc[i][t][2], a, b), fmaf(T.acc[i][t][3], a, b));
        hi = make_uint2(p0.hi, p1.hi);
        lo = make_uint2(p0.lo, p1.lo);
    };
    // training instances with an even tile count: the operand-typed copy for the weight gradients, row-major, as 16-byte pieces
    // (store_pair16: the lane groups of two token tiles exchange halves)
    uint2 kept[LX::on && NT % 2 == 0 ? NT : 1];
    auto keep_xn = [&](int i) {
        if constexpr (LX::on && NT % 2 == 0) {
            const int n = lane & 15, f0 = 16 * (w * RPW + i) + 4 * g;
#pragma unroll
            for (int t = 0; t < NT; t += 2)
                store_pair16(lx.xn, lx.D, lx.rows.row(16 * t + n), lx.rows.row(16 * (t + 1) + n), f0, g, kept[t], kept[t + 1],
                             !(BESO_TRAIN_FWD_ABL & 2));
        }
    };
    auto write_pair = [&](int i) {          // row tiles i, i+1 of this wave: Rf = w*RPW + i is even
        const int ks = (w * RPW + i) >> 1;
        if constexpr (LX::on && NT % 2 == 0) {
            if (ks >= KS) return;
            uint2 hi2[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                kept[t] = half(i, t); hi2[t] = half(i + 1, t);
                xnT[((size_t)t * KS + ks) * 64 + lane] = u32x4{kept[t].x, kept[t].y, hi2[t].x, hi2[t].y};
            }
            keep_xn(i);
#pragma unroll
            for (int t = 0; t < NT; ++t) kept[t] = hi2[t];
            keep_xn(i + 1);
            return;
        }
        if (ks < KS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (PX) {
                    uint2 h0, l0, h1, l1;
                    half_x3(i, t, h0, l0);
                    half_x3(i + 1, t, h1, l1);
                    xnT[((size_t)t * KS + ks) * 64 + lane] = u32x4{h0.x, h0.y, h1.x, h1.y};
                    xnT[((size_t)t * KS + ks) * 64 + lane + lo_off] = u32x4{l0.x, l0.y, l1.x, l1.y};
                } else {
                    const uint2 lo = half(i, t), hi = half(i + 1, t);
                    xnT[((size_t)t * KS + ks) * 64 + lane] = u32x4{lo.x, lo.y, hi.x, hi.y};
                }
            }
        }
    };
    auto write_single = [&](int i) {
        const int Rf = w * RPW + i;
        if constexpr (LX::on && NT % 2 == 0) {
            if ((Rf >> 1) >= KS) return;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                kept[t] = half(i, t);
                *((uint2*)(xnT + ((size_t)t * KS + (Rf >> 1)) * 64 + lane) + (Rf & 1)) = kept[t];
            }
            keep_xn(i);
            return;
        }
        if ((Rf >> 1) < KS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (PX) {
                    uint2 h0, l0;
                    half_x3(i, t, h0, l0);
                    *((uint2*)(xnT + ((size_t)t * KS + (Rf >> 1)) * 64 + lane) + (Rf & 1)) = h0;
                    *((uint2*)(xnT + ((size_t)t * KS + (Rf >> 1)) * 64 + lane + lo_off) + (Rf & 1)) = l0;
                } else {
                    *((uint2*)(xnT + ((size_t)t * KS + (Rf >> 1)) * 64 + lane) + (Rf & 1)) = half(i, t);
                }
            }
        }
    };
    if constexpr (RPW % 2 == 0) {
#pragma unroll
        for (int i = 0; i < RPW; i += 2) write_pair(i);
    } else {
        static_assert(RPW == 3, "row tiles per wave");
        if (w & 1) { write_single(0); write_pair(1); }
        else { write_pair(0); write_single(2); }
    }
    if constexpr (ADD_BIAS) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const f32x4 bv = *(const f32x4*)(bias + 16 * (w * RPW + i) + 4 * g);
#pragma unroll
            for (int t = 0; t < NT; ++t) T.acc[i][t] += bv;
        }
    }
    stamp(st, 34);
    __syncthreads();
}

// Token slots of a workgroup tile.  The GEMMs, LayerNorm and GELU do not care which token sits in which of
// the 96 slots; attention and the edges do.  layers_kernel puts the ACTION tokens of all samples first
// (slots [0, n_samples*t)) and the sigma / goal / state tokens behind them, so that the last layer -- whose
// MLP, out-projection and LayerNorm-2 only matter for the action tokens (the head reads nothing else,
// score_gpts.py:344-354) -- can run them on the first two token tiles only.  q/k/v rows in LDS stay in natural
// order (sample-major, position order) for the attention core; the tables translate.
struct SlotTabs {
    unsigned char sp_of_slot[kMT];    // (sample << PS) | position of the token in a slot; 0xFF = empty slot  (PS = 4; 7 in the
                                      // one-sample-per-workgroup instance, whose positions run to 79)
    unsigned char row_of_slot[kMT];   // natural row (sample*Tn + position) of a slot; empty slots map to themselves
    unsigned char slot_of_row[kMT];   // inverse
};
__device__ __forceinline__ void build_slot_tabs(SlotTabs* tb, int n_samples, int Tn, int t_win, int G, bool actions_first,
                                                int pshift = 4) {
    const int slot = threadIdx.x;
    if (slot < kMT) {
        const int nv = n_samples * Tn;
        unsigned char sp = 0xFF;
        int row = slot;
        if (slot < nv) {
            int sl, p;
            if (actions_first) {
                const int na = n_samples * t_win, no = Tn - t_win;
                if (slot < na) { sl = slot / t_win; p = G + 2 + 2 * (slot - sl * t_win); }
                else {
                    const int q = slot - na;
                    sl = q / no;
                    const int r = q - sl * no;
                    p = r <= G ? r : G + 1 + 2 * (r - G - 1);
                }
            } else { sl = slot / Tn; p = slot - sl * Tn; }
            sp = (unsigned char)((sl << pshift) | p);
            row = sl * Tn + p;
        }
        tb->sp_of_slot[slot] = sp;
        tb->row_of_slot[slot] = (unsigned char)row;
        tb->slot_of_row[row] = (unsigned char)slot;
    }
}

// Inputs / outputs of the network edges when they are fused into layers_kernel.
// One forward: out = D(state, action, goal, sigma).  Sampler loop (SampleSteps::n > 0): `action` holds x_T, every
// evaluation's sigma comes from its step record (`sigma` is not read), `out` receives the sample (it may alias `action`).
struct EdgeArgs {
    const float* state;    // [B][t][obs]
    const float* action;   // [B][t][act]
    const float* goal;     // [B][G][obs]
    const float* sigma;    // [B]
    float* out;            // [B][t][act]
    float* aux;            // [B][t][act] scratch of the sampler loop (Heun's first slope), or nullptr
    const float* noise;    // [evaluations][B][t][act]: the randn of every ancestral step of the launch, or nullptr
    int B, t, precondition, uncond_all, two;   // two: classifier-free pair (cond, uncond) = virtual samples (2b, 2b+1)
    float cond_lambda, sigma_data;
};

constexpr int kEmbObsK = 8, kEmbActK = 3;     // k-steps (4 inputs each) of the fused embedding GEMMs: obs <= 32, act <= 12
static_assert(kSPW * ((kMT / kSPW - 1) / 2) * 4 * kEmbActK * 4 <= kXsBytes && ((16 * kLongNT - 1) / 2) * 4 * kEmbActK * 4 <= kXsBytes,
              "xs holds the action windows of every shape fused_level admits (T = 1 + G + 2 t tokens per sample, kSPW T <= kMT or "
              "one sample of T <= 16 kLongNT; act <= 4 kEmbActK); fused_layers checks the call's own sizes as well");

// Which real sample and which conditioning a virtual sample stands for.
__device__ __forceinline__ void sample_of(const EdgeArgs& e, int vb, int& b, bool& uncond) {
    if (e.two) { b = vb >> 1; uncond = vb & 1; }
    else { b = vb; uncond = e.uncond_all != 0; }
}

// K1 fused: the residual tile is built in registers from (state, action, goal, sigma):
//   token 0: sigma_emb(log(sigma)/4); 1..G: tok_emb(goal)+pos; then tok_emb(state_i)+pos, action_emb(action_i*c_in)+pos
// (score_gpts.py:284-337, score_wrappers.py:96).  Weights come transposed ([in][Dp] fp32).
// Written WITHOUT divergent control flow around loads: every load is unconditional from a clamped
// (always valid) address and its value is selected afterwards -- a load inside a per-lane `if` is waited
// for at the end of its block, which serialised ~120 cache-cold round trips (70 kcycles per workgroup).
// PX = 0 (bf16 instances): the two embedding GEMMs run as split-bf16 products on the bf16 MFMA (operands (hi, lo) pairs, three
// MFMAs per pair: 2^-16 relative, three orders below the bf16 rounding of the layers that follow) -- one k-step of
// 32 inputs for tok_emb and a half k-step for action_emb instead of eleven k-steps of the 1/16-rate fp32 MFMA: the
// embedding's matrix-pipe time drops from 6.3 k to 1.3 k cycles per wave.  PX = 1 (BF16X3) keeps the exact-fp32 form.
// The noisy actions are read from `xs` (LDS: the action windows of the workgroup's real samples, [n_real][t][act], staged by
// the kernel -- in the sampler loop the previous evaluation's update left the next input there); sigma_u > 0: the
// evaluation's sigma, the same for every sample (sampler loop), else sigma[b].
template <int RPW, int PX = 1, int PS = 4>
__device__ __forceinline__ void embed_tile(Tile<RPW>& T, const EdgeArgs& e, const FusedDims& d, const char* gw, int s0,
                                           int n_samples, int Tn, int w, int lane, const SlotTabs* tb, const float* xs,
                                           float sigma_u, Stamps& st) {
#pragma clang fp contract(off)          // (explicit fmas only: the instances of the kernel agree bit for bit)
    asm volatile("" : "+v"(lane));
    const int n = lane & 15, g = lane >> 4;
    const int G = d.G, Dp = d.Dp;
    const float* tokT = (const float*)(gw + d.g_tokT);
    const float* actT = (const float*)(gw + d.g_actT);
    const float* src[kNTT];
    float scale[kNTT], sg[kNTT];
    int kind[kNTT], prow[kNTT];       // kind: 0 none, 1 tok_emb input (state / goal), 2 action, 3 sigma, 4 zeroed goal
    int xoff[kNTT];                   // offset in xs of the action window row a slot's token belongs to (valid for every slot)
    const int last = s0 + n_samples - 1;
    int b0; bool un0;
    sample_of(e, s0, b0, un0);
#pragma unroll
    for (int t = 0; t < kNTT; ++t) {
        const int sp = tb->sp_of_slot[t * 16 + n];
        const bool live = sp != 0xFF;
        const int sl = live ? sp >> PS : 0, p = live ? sp & ((1 << PS) - 1) : 0;
        int b; bool un;
        sample_of(e, min(s0 + sl, last), b, un);
        sg[t] = sigma_u > 0.f ? sigma_u : e.sigma[b];
        const int idx = p - G - 1, i = max(idx, 0) >> 1;
        xoff[t] = ((b - b0) * e.t + i) * d.act;
        const bool is_sig = p == 0, is_goal = p >= 1 && p <= G, is_act = idx >= 0 && (idx & 1);
        kind[t] = !live ? 0 : is_sig ? 3 : is_goal ? (un ? 4 : 1) : is_act ? 2 : 1;
        prow[t] = is_sig ? 0 : is_goal ? p - 1 : G + i;
        const float* ps = e.state + ((size_t)b * e.t + i) * d.obs;
        const float* pg = e.goal + ((size_t)b * G + max(p - 1, 0)) * d.obs;
        src[t] = kind[t] == 1 ? (is_goal ? pg : ps) : tokT;      // tokT: any readable address
        scale[t] = 1.f;
        if (kind[t] == 2 && e.precondition) scale[t] = __builtin_amdgcn_rsqf(fmaf(sg[t], sg[t], e.sigma_data * e.sigma_data));   // c_in
    }
    // tok_emb over states / goals and action_emb over the (pre-conditioned) noisy actions, exact fp32 on the
    // matrix pipe: X^T[f][tok] += W^T[f][c] * in[tok][c], four input features per v_mfma_f32_16x16x4_f32
    // (A: lane (f = lane&15, c = 4kk + g) from the transposed weights; B: lane (tok = lane&15, c = 4kk + g)
    // gathered from the inputs, zero for tokens of another kind).  Every operand of both GEMMs is requested
    // up front; kEmbObsK / kEmbActK bound obs / act (fused_level).
    stamp(st, 40);
    if constexpr (PX == 0) {
        // A: weight fragments (rows = output features) of this wave's row tiles, hi and lo images
        const u32x4* tokA = (const u32x4*)(gw + d.g_tokA);
        const u32x4* actA = (const u32x4*)(gw + d.g_actA);
        constexpr int RT = RPW * kWaves;
        u32x4 ath[RPW], atl[RPW], aah[RPW], aal[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            ath[i] = tokA[(size_t)(w * RPW + i) * 64 + lane]; atl[i] = tokA[(size_t)(RT + w * RPW + i) * 64 + lane];
            aah[i] = actA[(size_t)(w * RPW + i) * 64 + lane]; aal[i] = actA[(size_t)(RT + w * RPW + i) * 64 + lane];
        }
        // B: the lane's eight (tok_emb) / four (action_emb) inputs of every token tile -- contraction slot j of lane group g
        // is input 16 (j >> 2) + 4 g + (j & 3), as everywhere; clamped addresses, values selected afterwards
        float vt[kNTT][8], va[kNTT][4];
#pragma unroll
        for (int t = 0; t < kNTT; ++t) {
            const float* pt = src[t];
#pragma unroll
            for (int j = 0; j < 8; ++j) vt[t][j] = pt[min(16 * (j >> 2) + 4 * g + (j & 3), d.obs - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) va[t][j] = xs[xoff[t] + min(4 * g + j, d.act - 1)];
        }
        stamp(st, 41);
        float lsig[kNTT];
#pragma unroll
        for (int t = 0; t < kNTT; ++t) lsig[t] = logf(sg[t]) / 4.0f;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int f0 = 16 * (w * RPW + i) + 4 * g;
            T.fvalid[i] = f0 < d.D;
            const f32x4 bt = *(const f32x4*)((const float*)(gw + d.g_tokb) + f0);
            const f32x4 ba = *(const f32x4*)((const float*)(gw + d.g_actb) + f0);
            const f32x4 sw = *(const f32x4*)((const float*)(gw + d.g_sigw) + f0);
            const f32x4 sb = *(const f32x4*)((const float*)(gw + d.g_sigb) + f0);
#pragma unroll
            for (int t = 0; t < kNTT; ++t) {
                const int k = kind[t];
                const f32x4 pos = *(const f32x4*)((const float*)(gw + d.g_pos) + (size_t)prow[t] * Dp + f0);
                const f32x4 sig = __builtin_elementwise_fma(sw, (f32x4)(lsig[t]), sb);
                const f32x4 lin = (k == 2 ? ba : bt) + pos;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                T.acc[i][t] = k == 0 ? zero : k == 3 ? sig : lin;
            }
        }
        stamp(st, 42);
#pragma unroll
        for (int t = 0; t < kNTT; ++t) {
            u32x4 bh, bl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 16 * (q >> 1) + 4 * g + 2 * (q & 1);
                const float x0 = (kind[t] == 1 && c0 < d.obs) ? vt[t][2 * q] : 0.f;
                const float x1 = (kind[t] == 1 && c0 + 1 < d.obs) ? vt[t][2 * q + 1] : 0.f;
                const SplitPair p = split_op2(x0, x1);
                bh[q] = p.hi; bl[q] = p.lo;
            }
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                T.acc[i][t] = mfma_op(atl[i], bh, T.acc[i][t]);
                T.acc[i][t] = mfma_op(ath[i], bl, T.acc[i][t]);
                T.acc[i][t] = mfma_op(ath[i], bh, T.acc[i][t]);
            }
        }
        // (the action embedding's K = act <= 16 is a half k-step: the other shape of MFMA on the same accumulators)
        mixed_chain_pad();
#pragma unroll
        for (int t = 0; t < kNTT; ++t) {
            u32x4 ch = {0, 0, 0, 0}, cl = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c0 = 4 * g + 2 * q;
                const float x0 = (kind[t] == 2 && c0 < d.act) ? va[t][2 * q] * scale[t] : 0.f;
                const float x1 = (kind[t] == 2 && c0 + 1 < d.act) ? va[t][2 * q + 1] * scale[t] : 0.f;
                const SplitPair p = split_op2(x0, x1);
                ch[q] = p.hi; cl[q] = p.lo;
            }
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                T.acc[i][t] = mfma_op_half(aal[i], ch, T.acc[i][t]);
                T.acc[i][t] = mfma_op_half(aah[i], cl, T.acc[i][t]);
                T.acc[i][t] = mfma_op_half(aah[i], ch, T.acc[i][t]);
            }
        }
        return;
    }
    float aT[kEmbObsK][RPW], bT[kEmbObsK][kNTT], aA[kEmbActK][RPW], bA[kEmbActK][kNTT];
    // A token's source row has obs (state / goal) or act (action) elements: a row is only ever read by the
    // pass it belongs to -- every other token reads the (always long enough) weight table instead.
    auto load = [&](const float* wT, int n_in, int want, int kk, float (&a_)[RPW], float (&b_)[kNTT]) {
        const int cl = min(4 * kk + g, n_in - 1);
#pragma unroll
        for (int i = 0; i < RPW; ++i) a_[i] = wT[(size_t)cl * Dp + 16 * (w * RPW + i) + n];
#pragma unroll
        for (int t = 0; t < kNTT; ++t) b_[t] = want == 2 ? xs[xoff[t] + cl] : src[t][cl];
    };
    auto mask = [&](int n_in, int want, int kk, float (&a_)[RPW], float (&b_)[kNTT]) {
        const bool cv = 4 * kk + g < n_in;
#pragma unroll
        for (int i = 0; i < RPW; ++i) a_[i] = cv ? a_[i] : 0.f;
#pragma unroll
        for (int t = 0; t < kNTT; ++t) b_[t] = (cv && kind[t] == want) ? b_[t] * scale[t] : 0.f;
    };
#pragma unroll
    for (int kk = 0; kk < kEmbObsK; ++kk) load(tokT, d.obs, 1, kk, aT[kk], bT[kk]);
#pragma unroll
    for (int kk = 0; kk < kEmbActK; ++kk) load(actT, d.act, 2, kk, aA[kk], bA[kk]);
    stamp(st, 41);
    // biases, positions, sigma token: the accumulators start from them
    float lsig[kNTT];
#pragma unroll
    for (int t = 0; t < kNTT; ++t) lsig[t] = logf(sg[t]) / 4.0f;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int f0 = 16 * (w * RPW + i) + 4 * g;
        T.fvalid[i] = f0 < d.D;
        const f32x4 bt = *(const f32x4*)((const float*)(gw + d.g_tokb) + f0);
        const f32x4 ba = *(const f32x4*)((const float*)(gw + d.g_actb) + f0);
        const f32x4 sw = *(const f32x4*)((const float*)(gw + d.g_sigw) + f0);
        const f32x4 sb = *(const f32x4*)((const float*)(gw + d.g_sigb) + f0);
#pragma unroll
        for (int t = 0; t < kNTT; ++t) {
            const int k = kind[t];
            const f32x4 pos = *(const f32x4*)((const float*)(gw + d.g_pos) + (size_t)prow[t] * Dp + f0);
            const f32x4 sig = __builtin_elementwise_fma(sw, (f32x4)(lsig[t]), sb);
            const f32x4 lin = (k == 2 ? ba : bt) + pos;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            T.acc[i][t] = k == 0 ? zero : k == 3 ? sig : lin;
        }
    }
    stamp(st, 42);
#pragma unroll
    for (int kk = 0; kk < kEmbObsK; ++kk) {
        mask(d.obs, 1, kk, aT[kk], bT[kk]);
#pragma unroll
        for (int t = 0; t < kNTT; ++t)
#pragma unroll
            for (int i = 0; i < RPW; ++i)
                T.acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aT[kk][i], bT[kk][t], T.acc[i][t], 0, 0, 0);
    }
#pragma unroll
    for (int kk = 0; kk < kEmbActK; ++kk) {
        mask(d.act, 2, kk, aA[kk], bA[kk]);
#pragma unroll
        for (int t = 0; t < kNTT; ++t)
#pragma unroll
            for (int i = 0; i < RPW; ++i)
                T.acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aA[kk][i], bA[kk][t], T.acc[i][t], 0, 0, 0);
    }
}

// K7 fused: ln_f on the action tokens, action_pred (ln_f folded into it), c_out / c_skip and the
// classifier-free combination (score_gpts.py:341-354, score_wrappers.py:96, classifier_free_sampler.py:49).
// Every wave reduces its 48 features of every action token to `act` partial sums; LDS sums the waves.
// NT: token tiles that can hold action tokens -- with the action tokens first in the tile (layers_kernel's peeled last
// layer) those are the first NTL tiles and the statistics / partial sums of the other tiles are never read.
// Sampler loop (ls.mode >= 0): the thread that owns an element of the action window applies the step's update to the denoised
// value it has just computed (sampler_update: the step-by-step kernel's arithmetic, K8 of SURVEY 2.1) and leaves the input of
// the next evaluation in xs.  DDIM / Euler carry x in xs alone; a Heun step parks x (in `out`) and its first slope (in
// `aux`) in global memory between its two evaluations -- written and read back by the same thread.  The caller's barrier
// behind this function orders the xs write against the next embed.
struct LoopState {
    int mode;              // -1: a single forward (out <- denoised); else the BESO_STEP_* update of this evaluation (| kStepAddNoise)
    float c0, c1, sigma;   // the step's coefficients; sigma of this evaluation (> 0: uniform over the batch)
    bool last;             // the last evaluation of the launch: x goes out
    float c2;              // kStepAddNoise: sigma_up
    int ev;                // index of the evaluation inside the launch (its slab of EdgeArgs::noise)
};
template <int RPW, int NT = kNTT>
__device__ __forceinline__ void head_tile(const Tile<RPW>& T, const EdgeArgs& e, const FusedDims& d, const char* gw,
                                          float* red, float* part, int s0, int n_samples, int Tn, int w, int lane,
                                          const SlotTabs* tb, float* xs, LoopState& ls, Stamps& st, int cfg_pass = -1,
                                          float* cfgc = nullptr) {
#pragma clang fp contract(off)          // (explicit fmas only: the instances of the kernel agree bit for bit)
    asm volatile("" : "+v"(lane));
    const int n = lane & 15, g = lane >> 4;
    const int act = d.act, Dp = d.Dp;
    float mean[NT], rstd[NT];
    ln_stats<RPW, kWaves, NT>(T, red, d.D, w, lane, mean, rstd, st);
    const float* Wh = (const float*)(gw + d.g_headw);
    // part[(w*kMT + tokl)*16 + a]
    for (int a = 0; a < act; ++a) {
        f32x4 wv[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) wv[i] = *(const f32x4*)(Wh + (size_t)a * Dp + 16 * (w * RPW + i) + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                if (T.fvalid[i]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s = fmaf((T.acc[i][t][r] - mean[t]) * rstd[t], wv[i][r], s);
                }
            }
            s = rows_allreduce<false>(s);
            if (g == 0) part[((size_t)w * kMT + t * 16 + n) * 16 + a] = s;
        }
    }
    __syncthreads();
    // one thread per (real sample slot, step i, action dim): element `it` of the workgroup's action windows
    // cfg_pass >= 0 (long-sequence instance): the classifier-free pair runs as TWO passes of the workgroup's one sample --
    // pass 0 (conditional) leaves its head outputs in `cfgc`, pass 1 (unconditional) combines and applies the update
    const int G = d.G, per = (e.two && cfg_pass < 0) ? 2 : 1;
    const int n_real = n_samples / per;
    const float* bh = (const float*)(gw + d.g_headb);
    int b0; bool un0;
    sample_of(e, s0, b0, un0);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));            // (the sampler loop: nothing of this block may be hoisted out of it and kept live)
    for (int it = tid; it < n_real * e.t * act; it += kBlock) {
        const int a = it % act, i = (it / act) % e.t, sr = it / (act * e.t);
        const int sl = sr * per;
        int b; bool un;
        sample_of(e, s0 + sl, b, un);
        // slot of the action token of step i (conditional / only sample); the unconditional copy is sample sl+1
        const int tok_c = tb->slot_of_row[sl * Tn + G + 2 + 2 * i];
        float fc = bh[a], fu = bh[a];
#pragma unroll
        for (int ww = 0; ww < kWaves; ++ww) fc += part[((size_t)ww * kMT + tok_c) * 16 + a];
        if (cfg_pass >= 0) {
            if (cfg_pass == 0) { cfgc[it] = fc; continue; }
            fu = fc;
            fc = cfgc[it];
        } else if (e.two) {
            const int tok_u = tb->slot_of_row[(sl + 1) * Tn + G + 2 + 2 * i];
#pragma unroll
            for (int ww = 0; ww < kWaves; ++ww) fu += part[((size_t)ww * kMT + tok_u) * 16 + a];
        }
        const float sg = ls.sigma > 0.f ? ls.sigma : e.sigma[b];
        const float av = xs[it];                       // the evaluation's input action (x, or Heun's x2)
        float c_skip = 0.f, c_out = 1.f;
        if (e.precondition) {
            const float sd2 = e.sigma_data * e.sigma_data, s2 = fmaf(sg, sg, sd2);
            c_skip = sd2 / s2;
            c_out = sg * e.sigma_data / sqrtf(s2);
        }
        const float skip = av * c_skip;
        const float oc = fmaf(fc, c_out, skip);
        float r = oc;
        if (e.two) {
            const float ou = fmaf(fu, c_out, skip);
            r = fmaf(e.cond_lambda, oc - ou, ou);
        }
        float* dst = e.out + (size_t)b0 * e.t * act + it;
        if (ls.mode < 0) *dst = r;
        else {
            float* daux = e.aux + (size_t)b0 * e.t * act + it;
            float xv = av, x2v = 0.f, d1 = 0.f;
            const int mode = ls.mode & 0xff;
            if (mode == BESO_STEP_HEUN_CORRECT) { xv = *dst; x2v = av; d1 = *daux; }      // (wave-uniform)
            float o = sampler_update(mode, xv, x2v, r, d1, ls.c0, ls.c1);
            if (ls.mode & kStepAddNoise) {
                // sample_euler_ancestral (gc_sampling.py:246-247): x <- x + randn * sigma_up, the step's draw supplied by the
                // caller; two rounded operations, as BESO_STEP_ADD_NOISE of the step-by-step form
                const float nz = e.noise[((size_t)ls.ev * e.B + b0) * e.t * act + it] * ls.c2;
                o = o + nz;
            }
            xs[it] = o;
            if (mode == BESO_STEP_HEUN_PREDICT) { *dst = xv; *daux = d1; }
            else if (ls.last) *dst = o;
        }
    }
}

// GELU of one chunk's FC1 accumulators (RC row tiles x kNTT token tiles per wave) -> packed bf16 B
// fragments (RC/2 FC2 k-steps per wave), as a sequence of (RC/2)*kNTT*4 pair evaluations that can be issued
// one at a time between MFMAs.  Pair pi -> k-step pi/(4*kNTT), token tile (pi/4)%kNTT, elements 2*(pi%4),
// 2*(pi%4)+1 of the fragment's eight (0..3: even row tile's registers, 4..7: odd row tile's).
template <int RC, int NT>
__device__ __forceinline__ void gelu_pair(const f32x4 (&h)[RC][NT], float (&gq)[8], u32x4 (&hb)[RC / 2][NT], int pi) {
    const int j2 = pi / (4 * NT), t = (pi >> 2) % NT, j = (pi & 3) * 2;
    const f32x4& hv = h[2 * j2 + (j >> 2)][t];
    const f32x2 r = gelu_fast2(f32x2{hv[j & 3], hv[(j & 3) + 1]});
    gq[j] = r.x;
    gq[j + 1] = r.y;
    asm volatile("" : "+v"(gq[j]), "+v"(gq[j + 1]));      // keep the evaluation HERE (between the MFMAs), not sunk to its use
    if ((pi & 3) == 3) {
        hb[j2][t][0] = pack_op2(gq[0], gq[1]);
        hb[j2][t][1] = pack_op2(gq[2], gq[3]);
        hb[j2][t][2] = pack_op2(gq[4], gq[5]);
        hb[j2][t][3] = pack_op2(gq[6], gq[7]);
    }
}

// The same evaluation cut into 12 single-instruction slots, for two pairs (2q, 2q+1) at once: slot sigma of
// group q is step sigma/2 of pair 2q + sigma%2.  The MLP weave issues a few slots behind every MFMA: a wave's
// VALU instruction directly behind its own MFMA issues in that MFMA's shadow, a run of them does not
// (measured, tools/microbench/issue_rate.hip), and alternating two independent chains keeps the packed-fp32
// pipe (8 cycles dependent, 5.5 independent) from waiting on itself.
#ifndef BESO_GELU_SCALAR
#define BESO_GELU_SCALAR 0           // 1: GELU chain as v_fma_f32 pairs instead of v_pk_fma_f32 (A/B experiment)
#endif
struct GeluChain { f32x2 v, vc, s, p; };
template <int RC, int NT, int SIGMA>
__device__ __forceinline__ void gelu_slot(const f32x4 (&h)[RC][NT], GeluChain& c0, GeluChain& c1, float (&gq)[8],
                                          u32x4 (&hb)[RC / 2][NT]) {
    constexpr int q = SIGMA / 24, step = (SIGMA % 24) >> 1, ch = SIGMA & 1, pi = 2 * q + ch;
    constexpr int j2 = pi / (4 * NT), t = (pi >> 2) % NT, j = (pi & 3) * 2;
    GeluChain& g = ch ? c1 : c0;
    if constexpr ((BESO_ABL_MASK & 1) != 0) {
        if constexpr (step == 0) { gq[j] = h[2 * j2 + (j >> 2)][t][j & 3]; gq[j + 1] = h[2 * j2 + (j >> 2)][t][(j & 3) + 1]; }
    } else {
        if constexpr (step == 0) {
            g.v = f32x2{h[2 * j2 + (j >> 2)][t][j & 3], h[2 * j2 + (j >> 2)][t][(j & 3) + 1]};
            g.vc.x = __builtin_amdgcn_fmed3f(g.v.x, -4.0f, 4.0f);
        } else if constexpr (step == 1) g.vc.y = __builtin_amdgcn_fmed3f(g.v.y, -4.0f, 4.0f);
#if BESO_GELU_SCALAR
        // experiment: the same chain as single-lane-width ops (two per slot), kept apart so that they are not re-packed
#define BESO_G2(dst, ex, ey) do { dst.x = (ex); asm volatile("" : "+v"(dst.x)); dst.y = (ey); asm volatile("" : "+v"(dst.y)); } while (0)
        else if constexpr (step == 2) BESO_G2(g.s, g.vc.x * g.vc.x, g.vc.y * g.vc.y);
        else if constexpr (step == 3) BESO_G2(g.p, __builtin_fmaf(g.s.x, 2.277972093e-08f, -1.598515742e-06f), __builtin_fmaf(g.s.y, 2.277972093e-08f, -1.598515742e-06f));
        else if constexpr (step == 4) BESO_G2(g.p, __builtin_fmaf(g.p.x, g.s.x, 4.795382804e-05f), __builtin_fmaf(g.p.y, g.s.y, 4.795382804e-05f));
        else if constexpr (step == 5) BESO_G2(g.p, __builtin_fmaf(g.p.x, g.s.x, -0.0008139993719f), __builtin_fmaf(g.p.y, g.s.y, -0.0008139993719f));
        else if constexpr (step == 6) BESO_G2(g.p, __builtin_fmaf(g.p.x, g.s.x, 0.00877231165f), __builtin_fmaf(g.p.y, g.s.y, 0.00877231165f));
        else if constexpr (step == 7) BESO_G2(g.p, __builtin_fmaf(g.p.x, g.s.x, -0.06457294506f), __builtin_fmaf(g.p.y, g.s.y, -0.06457294506f));
        else if constexpr (step == 8) BESO_G2(g.p, __builtin_fmaf(g.p.x, g.s.x, 0.3978832308f), __builtin_fmaf(g.p.y, g.s.y, 0.3978832308f));
        else if constexpr (step == 9) BESO_G2(g.p, __builtin_fmaf(g.vc.x, g.p.x, 0.5f), __builtin_fmaf(g.vc.y, g.p.y, 0.5f));
        else if constexpr (step == 10) { gq[j] = g.v.x * g.p.x; asm volatile("" : "+v"(gq[j])); gq[j + 1] = g.v.y * g.p.y; }
#undef BESO_G2
#else
        else if constexpr (step == 2) g.s = g.vc * g.vc;
        else if constexpr (step == 3) g.p = __builtin_elementwise_fma(g.s, (f32x2)(2.277972093e-08f), (f32x2)(-1.598515742e-06f));
        else if constexpr (step == 4) g.p = __builtin_elementwise_fma(g.p, g.s, (f32x2)(4.795382804e-05f));
        else if constexpr (step == 5) g.p = __builtin_elementwise_fma(g.p, g.s, (f32x2)(-0.0008139993719f));
        else if constexpr (step == 6) g.p = __builtin_elementwise_fma(g.p, g.s, (f32x2)(0.00877231165f));
        else if constexpr (step == 7) g.p = __builtin_elementwise_fma(g.p, g.s, (f32x2)(-0.06457294506f));
        else if constexpr (step == 8) g.p = __builtin_elementwise_fma(g.p, g.s, (f32x2)(0.3978832308f));
        else if constexpr (step == 9) g.p = __builtin_elementwise_fma(g.vc, g.p, (f32x2)(0.5f));
        else if constexpr (step == 10) { const f32x2 r = g.v * g.p; gq[j] = r.x; gq[j + 1] = r.y; }
#endif
        // every slot is pinned where it is issued (otherwise the whole chain sinks to its use)
        if constexpr (step <= 1) asm volatile("" : "+v"(g.vc));
        else if constexpr (step == 2) asm volatile("" : "+v"(g.s));
        else if constexpr (step <= 9) asm volatile("" : "+v"(g.p));
        else if constexpr (step == 10) asm volatile("" : "+v"(gq[j]), "+v"(gq[j + 1]));
    }
    if constexpr (step == 11 && (pi & 3) == 3) {
        hb[j2][t][0] = pack_op2(gq[0], gq[1]);
        hb[j2][t][1] = pack_op2(gq[2], gq[3]);
        hb[j2][t][2] = pack_op2(gq[4], gq[5]);
        hb[j2][t][3] = pack_op2(gq[6], gq[7]);
    }
}

// MLP phase (xnT holds LN2(x) fragments on entry): hidden chunks of 16 row tiles (= 8 FC2 k-steps); per
// chunk FC1 (+bias) -> GELU -> hT -> FC2 accumulated into the residual.  NW waves: each owns RC = 16/NW
// row tiles of the chunk (RC/2 k-steps of hT) and RPW row tiles of the residual.  Software pipelined so
// that the VALU work hides under the matrix pipe: the GELU of chunk c is issued pair by pair between the
// MFMAs of FC2(c-1), whose operands (hT(c-1)) are still in LDS; the packed result waits in registers
// until every wave has finished reading hT(c-1).
//     FC1(0)
//     for c:  [FC2(c-1) || GELU(c)]  barrier  hT <- GELU(c)  FC1(c+1)  barrier
//     FC2(n-1)
#ifndef BESO_FC1_PF
#define BESO_FC1_PF 2
#endif
constexpr int kFc1PF = BESO_FC1_PF;      // k-steps of FC1 weight fragments in flight per wave
#ifndef BESO_LAT_PF1
#define BESO_LAT_PF1 4                   // ... in the latency instances (KS % BESO_LAT_PF1 == 0)
#endif
constexpr int kKC = kChunkTiles / 2;     // FC2 k-steps per hidden chunk
// First k-steps of chunk 0's FC1 weights of a layer (issued before the LayerNorm that precedes the phase).
template <int KS, int NW, int PF1 = kFc1PF>
__device__ __forceinline__ void mlp_prefetch(u32x4 (&a1r)[PF1][kChunkTiles / NW], const u32x4* __restrict__ w1p, int w,
                                             int lane) {
    constexpr int RC = kChunkTiles / NW;
    prefetch_ring<RC, PF1>(a1r, wptr(w1p + (size_t)(RC * w) * 64, lane), kChunkTiles);
}

template <int RPW, int KS, int NW, int NT = kNTT, int PF1 = kFc1PF, class MX = MlpPlain>   // NT: the first NT token tiles only
                                                                      // (last layer); PF1: k-steps of FC1 weights in flight per wave
__device__ __forceinline__ void mlp_phase(Tile<RPW>& T, const u32x4* xnT, u32x4* hT, const u32x4* __restrict__ w1p,
                                          const float* __restrict__ b1f, const u32x4* __restrict__ w2p, int HT,
                                          int KS2p, int w, int lane, u32x4 (&a1r)[PF1][kChunkTiles / NW],
                                          Stamps& st, const MX mx = MX{}, int n_valid = 1 << 30) {
    asm volatile("" : "+v"(lane));
    const int n_chunks = (HT + kChunkTiles - 1) / kChunkTiles;
    constexpr int RC = kChunkTiles / NW, KW = RC / 2;       // row tiles / FC2 k-steps of a chunk per wave
    constexpr int A2KS = NW * RPW;                           // fragments between FC2 k-steps
    constexpr int H1 = NT / 2;
    constexpr int PAIRS = KW * NT * 4;                     // GELU pair evaluations per chunk and wave
    constexpr int SLOTS = PAIRS * 12, MFMAS = kKC * NT * RPW;   // their instruction slots / the MFMAs they hide behind
    static_assert(RC % 2 == 0 && PAIRS % 2 == 0, "pairs are evaluated two at a time");
    // w1p: [chunk][kk][16 row tiles]; w2p: [kk2][NW*RPW row tiles]
    // (walking the chunks in a per-workgroup rotation to spread L2 channel load was measured SLOWER: 1.16 vs 1.13 ms;
    // workgroups of an XCD streaming the same weights in lockstep is what keeps them L2-resident)
    auto pc = [&](int c) { return c; };
    auto fc1_a = [&](int c) { return ABL_PTR(wptr(w1p + (size_t)(RC * w) * 64, lane), (size_t)pc(c) * KS * kChunkTiles); };
    auto fc2_a = [&](int c) { return ABL_PTR(wptr(w2p + (size_t)(w * RPW) * 64, lane), (size_t)(pc(c) * kKC) * (NW * RPW)); };
    auto fc1 = [&](int c, f32x4 (&h)[RC][NT], u32x4 (&ar)[PF1][RC]) {
        const int R0 = pc(c) * kChunkTiles + RC * w, g = lane >> 4;
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            const f32x4 bias = *(const f32x4*)(b1f + 16 * (R0 + r) + 4 * g);
#pragma unroll
            for (int t = 0; t < NT; ++t) h[r][t] = bias;
        }
        // (empty token slots: no bias either -- with their zero xnT columns the hidden activations are GELU(0) = 0,
        // zero B operands of FC2: layernorm_to_lds)
        if (zero_pad_instance(RPW, NT)) {
            if (!(16 * (NT - 1) + (lane & 15) < n_valid)) {
#pragma unroll
                for (int r = 0; r < RC; ++r) h[r][NT - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        gemm_phase_ring<RC, NT, PF1, kt16(KS)>(h, ar, fc1_a(c), kChunkTiles, xnT + lane, KS * 64, 64, KS);
    };
    auto write_hT = [&](u32x4 (&hb)[KW][NT]) {
#pragma unroll
        for (int j2 = 0; j2 < KW; ++j2)
#pragma unroll
            for (int t = 0; t < NT; ++t) hT[((size_t)t * kKC + KW * w + j2) * 64 + lane] = hb[j2][t];
    };
    // training instance: the pre-activation and GELU(h) of chunk c, row-major, for the backward pass (the fragment
    // hb[j2][t] is {row tile 2 j2: features 4g..4g+3 | row tile 2 j2 + 1: the same}, i.e. two 8-byte row pieces)
    // (round 5: the kept rows leave as 16-BYTE pieces.  The accumulator layout gives a lane 4 features = 8 bytes of a token; the
    //  lane groups g, g + 1 of two token tiles t, t + 1 exchange their halves with two v_permlane16_swap, after which a lane of an
    //  even group holds 8 consecutive features of tile t's token and a lane of an odd group those of tile t + 1's -- one store
    //  instruction where there were two.  What the forward's 876 MB of kept activations cost follows the store INSTRUCTIONS: the
    //  fp32 residuals written as bf16 -- half the bytes, the same instruction count -- changed nothing.)
    auto keep_h = [&](int c, const f32x4 (&hv)[RC][NT]) {
        if constexpr (MX::on) {
            const int n = lane & 15, g = lane >> 4;
#pragma unroll
            for (int r = 0; r < RC; ++r) {
                const int f0 = 16 * (c * kChunkTiles + RC * w + r) + 4 * g;
                // (an odd tile count: the pairs as 16-byte pieces, the last tile as 8-byte ones)
                constexpr int NTE = BESO_KEEP_HYBRID ? (NT & ~1) : (NT % 2 == 0 ? NT : 0);
                {
#pragma unroll
                    for (int t = 0; t < NTE; t += 2)
                        store_pair16(mx.h, mx.ld, mx.rows.row(16 * t + n), mx.rows.row(16 * (t + 1) + n), f0, g,
                                     make_uint2(pack_op2(hv[r][t][0], hv[r][t][1]), pack_op2(hv[r][t][2], hv[r][t][3])),
                                     make_uint2(pack_op2(hv[r][t + 1][0], hv[r][t + 1][1]), pack_op2(hv[r][t + 1][2], hv[r][t + 1][3])),
                                     !(BESO_TRAIN_FWD_ABL & 8));
                }
                {
#pragma unroll
                    for (int t = NTE; t < NT; ++t) {
                        const int tok = mx.rows.row(16 * t + n);
                        if (tok >= 0 && f0 < mx.ld && !(BESO_TRAIN_FWD_ABL & 8))
                            *(uint2*)(mx.h + (size_t)tok * mx.ld + f0) = make_uint2(pack_op2(hv[r][t][0], hv[r][t][1]),
                                                                                     pack_op2(hv[r][t][2], hv[r][t][3]));
                    }
                }
            }
        }
    };
    auto keep_g = [&](int c, const u32x4 (&hb)[KW][NT]) {
        if constexpr (MX::on) {
            const int n = lane & 15, g = lane >> 4;
#pragma unroll
            for (int j2 = 0; j2 < KW; ++j2)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int f0 = 16 * (c * kChunkTiles + RC * w + 2 * j2 + q) + 4 * g;
                    constexpr int NTE = BESO_KEEP_HYBRID ? (NT & ~1) : (NT % 2 == 0 ? NT : 0);
                    {
#pragma unroll
                        for (int t = 0; t < NTE; t += 2)
                            store_pair16(mx.g, mx.ld, mx.rows.row(16 * t + n), mx.rows.row(16 * (t + 1) + n), f0, g,
                                         make_uint2(hb[j2][t][2 * q], hb[j2][t][2 * q + 1]),
                                         make_uint2(hb[j2][t + 1][2 * q], hb[j2][t + 1][2 * q + 1]), !(BESO_TRAIN_FWD_ABL & 16));
                    }
                    {
#pragma unroll
                        for (int t = NTE; t < NT; ++t) {
                            const int tok = mx.rows.row(16 * t + n);
                            if (tok >= 0 && f0 < mx.ld && !(BESO_TRAIN_FWD_ABL & 16))
                                *(uint2*)(mx.g + (size_t)tok * mx.ld + f0) = make_uint2(hb[j2][t][2 * q], hb[j2][t][2 * q + 1]);
                        }
                    }
                }
        }
    };

    f32x4 h[RC][NT];
    u32x4 hb[KW][NT];
    u32x4 af2[2][RPW];                   // FC2 weight fragments of two k-steps, [k-step parity][row tile]
    float gq[8];
    GeluChain gc0, gc1;
    auto fc2_prefetch = [&](int c) {
        const WPtr a2 = fc2_a(c);
#pragma unroll
        for (int r = 0; r < RPW; ++r) { af2[0][r] = a2.at(r); af2[1][r] = a2.at(r + ABL_KS(A2KS)); }
    };
    // ---- prologue: FC1(0) (its first weight fragments arrive preloaded in a1r), GELU(0) (nothing to hide
    // it under yet), hT(0), FC1(1).  Every later weight request is issued one phase ahead of its use.
    static_assert(KS % PF1 == 0, "FC1 weight ring");
    fc1(0, h, a1r);                 // rows beyond HT are zero weights + zero bias: harmless for every wave
    keep_h(0, h);
    if (n_chunks > 1) prefetch_ring<RC, PF1>(a1r, fc1_a(1), kChunkTiles);
    fc2_prefetch(0);
#pragma unroll
    for (int pi = 0; pi < PAIRS; ++pi) gelu_pair<RC, NT>(h, gq, hb, pi);
    write_hT(hb);
    if (NT > 4) keep_g(0, hb);
    if (n_chunks > 1) { fc1(1, h, a1r); keep_h(1, h); }
    if (NT <= 4) keep_g(0, hb);
    stamp(st, 20);
    __syncthreads();                     // hT(0) complete
#pragma unroll 1
    for (int c = 1; c < n_chunks; ++c) {
        // lane-derived addresses are recomputed per chunk: hoisted out of this loop they stay live across the
        // weave (the register-pressure peak), get spilled, and every scratch reload is a vmcnt(0) stall
        asm volatile("" : "+v"(lane));
        const int tiles_here = min(kChunkTiles, HT - pc(c) * kChunkTiles);
        const bool fc1_active = RC * w < tiles_here;
        // waves whose rows of the NEXT chunk are all padding (hidden 1440 -> 1536: the last chunk has 10 of 16
        // row tiles) skip its FC1 (their GELU slots run on the bias-only accumulators: branching
        // around the weave costs 190 spilled VGPRs) and the hT write
        const int cn = min(c + 1, n_chunks - 1);
        const bool next_active = RC * w < min(kChunkTiles, HT - pc(cn) * kChunkTiles);
        if (next_active) prefetch_ring<RC, PF1>(a1r, fc1_a(cn), kChunkTiles);
        {
            // ---- FC2(c-1) (always a full chunk: 8 k-steps) with GELU(c) woven in, fully unrolled
            const WPtr a2 = fc2_a(c - 1);
            const u32x4* b = hT + lane;
            u32x4 bf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[t] = b[t * kKC * 64];
            static_for<0, kKC>([&](auto KK) {
                constexpr int kk = decltype(KK)::value;
                static_for<0, NT>([&](auto TT) {
                    constexpr int t = decltype(TT)::value;
                    // unit of the weave: ONE MFMA + its share of the chunk's GELU slots (2 for RPW = 3)
                    static_for<0, RPW>([&](auto RR) {
                        constexpr int r = decltype(RR)::value;
                        constexpr int unit = (kk * NT + t) * RPW + r;
                        __builtin_amdgcn_sched_barrier(0);
                        T.acc[r][t] = mfma_op(af2[kk & 1][r], bf[t], T.acc[r][t]);
                        static_for<unit * SLOTS / MFMAS, (unit + 1) * SLOTS / MFMAS>([&](auto SG) {
                            gelu_slot<RC, NT, decltype(SG)::value>(h, gc0, gc1, gq, hb);
                        });
                    });
                    __builtin_amdgcn_sched_barrier(0);
                    if (t == H1 - 1 && kk + 1 < kKC && !(BESO_ABL_MASK & 32)) {
#pragma unroll
                        for (int t2 = 0; t2 < H1; ++t2) bf[t2] = b[t2 * kKC * 64 + (kk + 1) * 64];
                    }
                    if (t == NT - 1) {
                        if (kk + 1 < kKC && !(BESO_ABL_MASK & 32)) {
#pragma unroll
                            for (int t2 = H1; t2 < NT; ++t2) bf[t2] = b[t2 * kKC * 64 + (kk + 1) * 64];
                        }
                        if (kk + 2 < kKC) {
#pragma unroll
                            for (int r = 0; r < RPW; ++r) af2[kk & 1][r] = a2.at(r + ABL_KS((kk + 2) * A2KS));
                        }
                    }
                });
            });
        }
        fc2_prefetch(c);                 // first two k-steps of FC2(c): in flight across the barriers and FC1(c+1)
        stamp(st, 21);
        if (!(BESO_ABL_MASK & 2)) __syncthreads();                 // every wave is done reading hT(c-1)
        stamp(st, 22);
        // (training instances: GELU(h) of chunk c leaves TOGETHER with h of chunk c + 1, behind FC1(c + 1) -- one burst of stores per
        //  chunk instead of two.  What the kept activations cost is neither their bytes nor their instruction count but the number of
        //  BURSTS: a wave's loads queue behind its stores' acknowledgements (in-order vmcnt), ~1.8 us per burst whatever its size.)
        //  Instances of up to four token tiles; the eight-sample instance is at its 256 registers already and measured 1.3 %
        //  slower with hb kept alive across FC1: 12.16 -> 12.32 ms per 8192-sample step.)
        constexpr bool kLateG = NT <= 4;
        if (fc1_active) { write_hT(hb); if (!kLateG) keep_g(pc(c), hb); }
        if (c + 1 < n_chunks && next_active) { fc1(c + 1, h, a1r); keep_h(pc(c + 1), h); }
        if (kLateG && fc1_active) keep_g(pc(c), hb);
        stamp(st, 23);
        if (!(BESO_ABL_MASK & 2)) __syncthreads();                 // hT(c) complete
        stamp(st, 24);
    }
    {
        // ---- FC2 of the last chunk; k-steps are consumed in pairs: an odd tail reads a stale (finite) hT
        // slot against zero weights
        const int c = n_chunks - 1;
        const int tiles_here = min(kChunkTiles, HT - pc(c) * kChunkTiles);
        gemm_phase<RPW, NT, false, kNTT>(T.acc, af2[0], af2[1], fc2_a(c), A2KS, hT + lane, kKC * 64, 64, ((tiles_here >> 1) + 1) & ~1);
    }
    stamp(st, 25);
    __syncthreads();
}

// A operand of Y^T = V^T P^T (v_mfma_f32_16x16x16_bf16: lane (d = 16 dt + n, key group g) holds V[row0 + 4g + r][d], r = 0..3)
// with gfx950's transpose read: the 16 lanes of a group address the [4 keys][16 dims] block row by row (lane i: key
// 4g + i/4, dims 4(i%4)..+3, 8 bytes) and each receives column i of it -- four ds_read_b64_tr_b16 per sample and head
// instead of sixteen 16-bit reads and their shifts (V stays row-major in LDS: 144-byte rows).
#ifndef BESO_V_TR
#define BESO_V_TR 1                  // 0: the 16-bit gather (A/B)
#endif
__device__ __forceinline__ uint2 v_frag(const uint16_t* qkv, int row0, int dt, int n, int g) {
#if BESO_V_TR
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    const uint16_t* p = qkv + ((size_t)2 * kQKVRows + row0 + 4 * g + (n >> 2)) * kQKVRow + 16 * dt + 4 * (n & 3);
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p)));
#else
    const uint16_t* vb = qkv + ((size_t)2 * kQKVRows + row0 + 4 * g) * kQKVRow + n + 16 * dt;
    uint2 va;
    va.x = (uint32_t)vb[0] | ((uint32_t)vb[kQKVRow] << 16);
    va.y = (uint32_t)vb[2 * kQKVRow] | ((uint32_t)vb[3 * kQKVRow] << 16);
    return va;
#endif
}

// First two k-steps of the first head pair's QKV weights of a layer (issued before the LayerNorm that precedes
// the phase): even / odd fragments of gemm_phase.
template <int KS>
__device__ __forceinline__ void attn_prefetch(u32x4 (&qE)[3], u32x4 (&qO)[3], const u32x4* __restrict__ wqkv, int w, int lane) {
    prefetch_a<3>(qE, qO, wptr(wqkv + (size_t)(3 * w) * 64, lane), 24);
}

// Attention phase (xnT holds LN1(x) fragments on entry).  Tokens of the tile are in natural order:
// slot = sample*Tn + position, n_samples*Tn valid slots.
//
// Heads are processed in PAIRS: the q/k/v rows of two heads are 24 row tiles = 3 per wave over all 6 token
// tiles, so every QKV weight fragment is loaded by exactly one wave (one head at a time is 12 row tiles =
// 1.5 per wave: the 4x2 wave split it forces loads every fragment twice, +25 % of the layer's weight bytes).
// LDS holds q/k/v of ONE head: waves 0-3 (head A's rows) write theirs at once, waves 4-7 keep head B's
// accumulators in registers until head A's attention core is done.
//   QKV(A,B)  write(A) | bar | core(A) | bar | proj(A), write(B) | bar | core(B) | bar | proj(B)
// Weight fragments are requested one phase ahead of their use (the L2 round trip hides behind the barriers
// and the core): qE/qO (first two k-steps of the pair's QKV weights) arrive preloaded and are refilled for
// the next pair before core(B); each head's projection weights (two k-steps: all of them) are requested
// before the barrier that precedes its core.
// HG > 1: a virtual head is HG real heads of `hd` dims side by side (FusedDims); H counts virtual heads.
// CORE = 1 (one sample of up to 16 NTQ tokens per workgroup): the core of a head runs on waves 0 .. NTQ-1, wave qt owning
// query tile qt against key tiles 0 .. qt (causal), softmax over all of its keys in registers.
template <int RPW, int KS, int HG, int NTP = kNTT, int NTQ = kNTT, int CORE = 0, class AX = AttnPlain>   // NTP: token tiles that receive the out-projection (last layer:
                                                                     // action tokens only); NTQ: token tiles that hold tokens at all
__device__ __forceinline__ void attn_phase(Tile<RPW>& T, const u32x4* xnT, unsigned char* u,
                                           const u32x4* __restrict__ wqkv, const float* __restrict__ bqkv,
                                           const u32x4* __restrict__ wproj, int H, int hd, int Tn, int n_samples,
                                           int w, int lane, const SlotTabs* tb, u32x4 (&qE)[3], u32x4 (&qO)[3], Stamps& st,
                                           const AX ax = AX{}) {
    asm volatile("" : "+v"(lane));
    uint16_t* qkv = (uint16_t*)u;                         // [3][kQKVRows][kQKVRow] bf16
    u32x4* yT = (u32x4*)(u + kQKVBytes);                  // [(t*2 + kk)*64 + lane]
    const int wa = w & 3, hsel = w >> 2;                  // this wave's rows: tiles 3wa..3wa+2 of head 2*pair + hsel
    const float scale_log2e = 1.4426950408889634f * __builtin_amdgcn_rsqf((float)hd);
    auto qkv_a = [&](int pair) { return ABL_PTR(wptr(wqkv + (size_t)(3 * w) * 64, lane), (size_t)pair * KS * 24); };   // [pair][kk][24 row tiles]
    auto proj_a = [&](int h) { return ABL_PTR(wptr(wproj + (size_t)(w * RPW) * 64, lane), (size_t)(2 * h) * (kWaves * RPW)); };   // [2h+kk][row tiles]
    // `ln` (= lane) is re-made opaque in every pair iteration: the LDS addresses below are loop invariant
    // and would otherwise be hoisted out of the pair loop and spilled (24 VGPRs).
    int ln = lane;
    auto write_qkv = [&](const f32x4 (&qa)[3][NTQ], int vh) {      // vh: the (virtual) head these rows belong to
        const int n = ln & 15, g = ln >> 4;
        int row[NTQ];                                     // natural q/k/v row of this lane's token in each token tile
#pragma unroll
        for (int t = 0; t < NTQ; ++t) row[t] = tb->row_of_slot[t * 16 + n];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int rt = 3 * wa + i, part = rt >> 2, d0 = (rt & 3) * 16 + 4 * g;
            uint16_t* dst = qkv + (size_t)part * kQKVRows * kQKVRow + d0;
#pragma unroll
            for (int t = 0; t < NTQ; ++t) {
                uint2 pk;
                pk.x = pack_op2(qa[i][t][0], qa[i][t][1]);
                pk.y = pack_op2(qa[i][t][2], qa[i][t][3]);
                *(uint2*)(dst + (size_t)row[t] * kQKVRow) = pk;
                if constexpr (AX::on) {                   // kept for the backward pass: [row][part D + head dims]
                    const int grow = ax.rows.row(t * 16 + n);
                    if (grow >= 0 && d0 < HG * hd && !(BESO_TRAIN_FWD_ABL & 4))
                        *(uint2*)(ax.qkv + (size_t)grow * (3 * ax.D) + part * ax.D + vh * (HG * hd) + d0) = pk;
                }
            }
        }
    };
    // ---- attention core on the matrix pipe, one sample per wave (score_gpts.py:69-73):
    //   S^T[j][i] = sum_d K[j][d] Q[i][d]        2 x v_mfma_f32_16x16x32_bf16 (A = K rows, B = Q rows)
    //   D layout: lane (i = lane&15, g) holds keys j = 4g + r  ->  causal mask, softmax over j =
    //   in-lane over r + two xor-shuffles over g; the unnormalised probabilities are already the B
    //   operand (k = 4g..4g+3) of v_mfma_f32_16x16x16_bf16 for
    //   Y^T[d][i] = sum_j V[j][d] P[i][j]          4 x (A = V^T gathered with 16-bit LDS reads)
    //   whose D layout is the B fragment of the out-projection (same k permutation as the weights).
    // token slot of (sample w, position lane & 15): looked up ONCE (it was an LDS round trip at the tail of every core's
    // serial chain); clamped index, used only where position < Tn
    const int my_tok = tb->slot_of_row[min(w, n_samples - 1) * Tn + min(lane & 15, Tn - 1)];
    // training instance: dropout keep-scales of this lane's four probabilities (query n, keys 4g..4g+3) of real head rh, and
    // the attention output of (sample w, query n), dims d0..d0+3 of virtual head vh, row-major
    auto drop4 = [&](float (&e)[4], int rh, int n, int g) {
        if constexpr (AX::on) {
            if (ax.p > 0.f) {
                const size_t base = (((size_t)(ax.s0 + w) * ax.H + rh) * Tn + n) * Tn + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] *= drop_scale(ax.seed, ax.site, base + r, ax.p, ax.inv_keep);
            }
        }
    };
    auto keep_y = [&](const auto& yb, int vh, int tok, int g) {
        if constexpr (AX::on) {
            const int grow = ax.rows_y.row(tok);
            if (grow >= 0 && !(BESO_TRAIN_FWD_ABL & 4)) {
                uint16_t* dst = ax.y + (size_t)grow * ax.D + vh * (HG * hd) + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    if (16 * dt + 4 * g < HG * hd) *(uint2*)(dst + 16 * dt) = make_uint2(yb[dt >> 1][2 * (dt & 1)], yb[dt >> 1][2 * (dt & 1) + 1]);
            }
        }
    };
    auto core = [&](int vh) {
    const int n = ln & 15, g = ln >> 4;
    if constexpr (CORE == 1) {
        // ---- long sequence, one sample: S^T tiles [key tile kt][query tile w] for kt <= w, same operand roles and D
        // layout as below (lane (i, g) holds keys 16 kt + 4 g + r of query 16 w + i), one softmax over the wave's
        // 4 (w + 1) scores per lane, Y^T accumulated over the key tiles
        static_assert(HG == 1, "grouped heads are a short-sequence layout");
        if (w < NTQ && !(BESO_ABL_MASK & 16)) {
            const uint16_t* qb = qkv + ((size_t)0 * kQKVRows + 16 * w + n) * kQKVRow + 8 * g;
            const u32x4 q0 = *(const u32x4*)qb, q1 = *(const u32x4*)(qb + 32);
            float e[NTQ][4];
            float m = -INFINITY;
            const int query = 16 * w + n;
#pragma unroll
            for (int kt = 0; kt < NTQ; ++kt) {
                if (kt <= w) {                                             // wave-uniform
                    const uint16_t* kb = qkv + ((size_t)1 * kQKVRows + 16 * kt + n) * kQKVRow + 8 * g;
                    f32x4 sT = {0.f, 0.f, 0.f, 0.f};
                    sT = mfma_op(*(const u32x4*)kb, q0, sT);
                    sT = mfma_op(*(const u32x4*)(kb + 32), q1, sT);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = 16 * kt + 4 * g + r;
                        e[kt][r] = (key <= query && key < Tn) ? sT[r] * scale_log2e : -INFINITY;
                        m = fmaxf(m, e[kt][r]);
                    }
                }
            }
            m = rows_allreduce<true>(m);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NTQ; ++kt) {
                if (kt <= w) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { e[kt][r] = __builtin_amdgcn_exp2f(e[kt][r] - m); sum += e[kt][r]; }
                }
            }
            sum = rows_allreduce<false>(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
            f32x4 y[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) y[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NTQ; ++kt) {
                if (kt <= w) {
                    const uint2 pb = make_uint2(pack_op2(e[kt][0], e[kt][1]), pack_op2(e[kt][2], e[kt][3]));
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const uint2 va = v_frag(qkv, 16 * kt, dt, n, g);
                        y[dt] = mfma_op16(va, pb, y[dt]);
                    }
                }
            }
            // (tokens in natural order: query 16 w + n sits in slot 16 w + n)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 yb;
                yb[0] = pack_op2(y[2 * kk][0] * inv, y[2 * kk][1] * inv);
                yb[1] = pack_op2(y[2 * kk][2] * inv, y[2 * kk][3] * inv);
                yb[2] = pack_op2(y[2 * kk + 1][0] * inv, y[2 * kk + 1][1] * inv);
                yb[3] = pack_op2(y[2 * kk + 1][2] * inv, y[2 * kk + 1][3] * inv);
                yT[((size_t)w * 2 + kk) * 64 + ln] = yb;
            }
        }
    } else if constexpr (HG > 1) {
        // Grouped heads: the same fragments, but head h only sees its own dims.  S_h: the Q fragment with the
        // other heads' dims zeroed (dword granular: hd is a multiple of 4); Y rows are taken from the head they
        // belong to (lane-group granular for the same reason), already normalised by that head's 1/sum.
        if (w < n_samples && !(BESO_ABL_MASK & 16)) {
            const uint16_t* qb = qkv + ((size_t)0 * kQKVRows + w * Tn + n) * kQKVRow + 8 * g;
            const uint16_t* kb = qkv + ((size_t)1 * kQKVRows + w * Tn + n) * kQKVRow + 8 * g;
            u32x4 qf[2], kf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) { qf[kk] = *(const u32x4*)(qb + 32 * kk); kf[kk] = *(const u32x4*)(kb + 32 * kk); }
            uint2 va[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) va[dt] = v_frag(qkv, w * Tn, dt, n, g);
            f32x4 y[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) y[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < HG; ++h) {
                const int lo_d = h * hd, hi_d = lo_d + hd;
                f32x4 sT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    // (a k-half none of whose 32 dims belong to head h contributes exact zeros: skipped -- block-push, three
                    // heads of 20 in 64: heads 0 and 2 live in one half each, 4 instead of 6 MFMAs per sample and virtual head)
                    if (!(32 * kk < hi_d && 32 * kk + 32 > lo_d)) continue;          // wave-uniform
                    u32x4 qm;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int d0 = 32 * kk + 8 * g + 2 * m;
                        qm[m] = (d0 >= lo_d && d0 < hi_d) ? qf[kk][m] : 0u;
                    }
                    sT = mfma_op(kf[kk], qm, sT);
                }
                float e[4], mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = (4 * g + r <= n) ? sT[r] * scale_log2e : -INFINITY;
                    mx = fmaxf(mx, e[r]);
                }
                mx = rows_allreduce<true>(mx);
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(e[r] - mx); sum += e[r]; }
                sum = rows_allreduce<false>(sum);
                const float inv = __builtin_amdgcn_rcpf(sum);
                if constexpr (AX::on) drop4(e, vh * HG + h, n, g);
                const uint2 pb = make_uint2(pack_op2(e[0], e[1]), pack_op2(e[2], e[3]));
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    if (16 * dt < hi_d && 16 * dt + 16 > lo_d) {             // wave-uniform: tile dt holds rows of head h
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        const f32x4 yh = mfma_op16(va[dt], pb, z);
                        const int d0 = 16 * dt + 4 * g;                        // this lane's rows d0 .. d0+3
                        const bool mine = d0 >= lo_d && d0 < hi_d;
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[dt][r] = mine ? yh[r] * inv : y[dt][r];
                    }
                }
            }
            if (n < Tn) {
                const int tok = my_tok;                               // token slot of (sample w, position n)
                u32x4 ybk[AX::on ? 2 : 1];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x4 yb;
                    yb[0] = pack_op2(y[2 * kk][0], y[2 * kk][1]);
                    yb[1] = pack_op2(y[2 * kk][2], y[2 * kk][3]);
                    yb[2] = pack_op2(y[2 * kk + 1][0], y[2 * kk + 1][1]);
                    yb[3] = pack_op2(y[2 * kk + 1][2], y[2 * kk + 1][3]);
                    yT[((size_t)(tok >> 4) * 2 + kk) * 64 + (g << 4) + (tok & 15)] = yb;
                    if constexpr (AX::on) ybk[kk] = yb;
                }
                if constexpr (AX::on) keep_y(ybk, vh, tok, g);
            }
        }
    } else {
    if (w < n_samples && !(BESO_ABL_MASK & 16)) {
        const uint16_t* qb = qkv + ((size_t)0 * kQKVRows + w * Tn + n) * kQKVRow + 8 * g;
        const uint16_t* kb = qkv + ((size_t)1 * kQKVRows + w * Tn + n) * kQKVRow + 8 * g;
        f32x4 sT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            sT = mfma_op(*(const u32x4*)(kb + 32 * kk), *(const u32x4*)(qb + 32 * kk), sT);
        float e[4], m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e[r] = (4 * g + r <= n) ? sT[r] * scale_log2e : -INFINITY;     // (q k^T)/sqrt(hd), in log2 units
            m = fmaxf(m, e[r]);
        }
        m = rows_allreduce<true>(m);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(e[r] - m); sum += e[r]; }   // exp2(-inf) = 0
        sum = rows_allreduce<false>(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        if constexpr (AX::on) drop4(e, vh, n, g);
        uint2 pb = make_uint2(pack_op2(e[0], e[1]), pack_op2(e[2], e[3]));
        f32x4 y[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const uint2 va = v_frag(qkv, w * Tn, dt, n, g);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            y[dt] = mfma_op16(va, pb, z);
        }
        if (n < Tn) {
            const int tok = my_tok;                               // token slot of (sample w, position n)
            u32x4 ybk[AX::on ? 2 : 1];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 yb;
                yb[0] = pack_op2(y[2 * kk][0] * inv, y[2 * kk][1] * inv);
                yb[1] = pack_op2(y[2 * kk][2] * inv, y[2 * kk][3] * inv);
                yb[2] = pack_op2(y[2 * kk + 1][0] * inv, y[2 * kk + 1][1] * inv);
                yb[3] = pack_op2(y[2 * kk + 1][2] * inv, y[2 * kk + 1][3] * inv);
                yT[((size_t)(tok >> 4) * 2 + kk) * 64 + (g << 4) + (tok & 15)] = yb;
                if constexpr (AX::on) ybk[kk] = yb;
            }
            if constexpr (AX::on) keep_y(ybk, vh, tok, g);
        }
    }
    }
    };

    for (int pair = 0; pair < H / 2; ++pair) {
        const int hA = 2 * pair, hB = hA + 1;
        asm volatile("" : "+v"(ln));
        const int g = ln >> 4;
        stamp(st, 10);
        u32x4 aE[RPW], aO[RPW];
        f32x4 qa[3][NTQ];
        // ---- q, k, v of both heads for all tokens of the tile
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f32x4 bv = *(const f32x4*)(bqkv + ((hA + hsel) * 12 + 3 * wa + i) * 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < NTQ; ++t) qa[i][t] = bv;
        }
        gemm_phase<3, NTQ, kt16(KS)>(qa, qE, qO, qkv_a(pair), 24, xnT + lane, KS * 64, 64, KS);
        prefetch_a<RPW>(aE, aO, proj_a(hA), kWaves * RPW);
        if (hsel == 0) write_qkv(qa, hA);
        stamp(st, 11);
        __syncthreads();
        stamp(st, 12);
        core(hA);
        stamp(st, 16);
        __syncthreads();
        stamp(st, 17);
        // ---- head A's slice of the out-projection, accumulated into the residual; head B's q/k/v to LDS
        gemm_phase<RPW, NTP, false, kNTT>(T.acc, aE, aO, proj_a(hA), kWaves * RPW, yT + lane, 2 * 64, 64, 2);
        prefetch_a<RPW>(aE, aO, proj_a(hB), kWaves * RPW);
        if (hsel == 1) write_qkv(qa, hB);
        if (pair + 1 < H / 2) prefetch_a<3>(qE, qO, qkv_a(pair + 1), 24);   // qa's registers are free from here
        stamp(st, 13);
        __syncthreads();
        stamp(st, 14);
        core(hB);
        stamp(st, 15);
        __syncthreads();
        stamp(st, 18);
        gemm_phase<RPW, NTP, false, kNTT>(T.acc, aE, aO, proj_a(hB), kWaves * RPW, yT + lane, 2 * 64, 64, 2);
        // no barrier needed here: the next writes to qkv/yT happen behind the next pair's barriers
    }
    __syncthreads();
}

// Attention phase of the long-sequence instance (one sample of up to 16 NT tokens per workgroup, tokens in natural
// order; layers_kernel CORE = 1).  q/k/v of BOTH heads of a pair sit in LDS ([head][q | k | v][16 NT rows][kQKVRow]), so the
// pair costs three barriers and its ten (head, query tile) cores run at once:
//     QKV(A, B)  write(A, B) | bar | cores(A, B) | bar | proj(A), proj(B) | bar
// wave (head = w & 1, query tile qt): S^T tiles [key tile kt][query tile qt] for kt <= qt on v_mfma_f32_16x16x32_bf16 (the
// short core's operand roles and D layout: lane (i, g) holds keys 16 kt + 4 g + r of query 16 qt + i), ONE softmax over the
// 4 (qt + 1) scores of a lane, Y^T accumulated over the key tiles on v_mfma_f32_16x16x16_bf16.  Work is 1 .. NT key tiles
// per query tile; the eight waves take {4}, {4}, {3}, {3}, {2, 0}, {2, 0}, {1}, {1} (NT = 5: at most five tile units each).
// The normalised Y^T of query tile qt goes out as the out-projection's B fragments INTO the q rows of that tile -- read by
// this wave only, and already in its registers -- which is what lets two heads fit: 2 x 34.5 KiB + nothing for y.
template <int RPW, int KS, int NT>
__device__ __forceinline__ void attn_phase_long(Tile<RPW>& T, const u32x4* xnT, unsigned char* u, const u32x4* __restrict__ wqkv,
                                                const float* __restrict__ bqkv, const u32x4* __restrict__ wproj, int H, int hd,
                                                int Tn, int w, int lane, u32x4 (&qE)[3], u32x4 (&qO)[3], Stamps& st) {
    static_assert(NT == 5, "the wave -> query tile table below");
    asm volatile("" : "+v"(lane));
    constexpr int kRows = 16 * NT, kHead = 3 * kRows * kQKVRow;        // bf16 elements of one head's q | k | v
    constexpr int kYT = 16 * kQKVRow * 2 / 16;                        // u32x4 between the y fragments of two token tiles (= 16 q rows)
    uint16_t* qkv = (uint16_t*)u;
    const int wa = w & 3, hsel = w >> 2;                  // this wave's q/k/v rows: tiles 3wa..3wa+2 of head 2*pair + hsel
    const float scale_log2e = 1.4426950408889634f * __builtin_amdgcn_rsqf((float)hd);
    auto qkv_a = [&](int pair) { return wptr(wqkv + (size_t)(3 * w) * 64, lane).adv((size_t)pair * KS * 24); };
    auto proj_a = [&](int h) { return wptr(wproj + (size_t)(w * RPW) * 64, lane).adv((size_t)(2 * h) * (kWaves * RPW)); };
    int ln = lane;
    auto core = [&](int head, int qt) {
        const int n = ln & 15, g = ln >> 4;
        uint16_t* hq = qkv + (size_t)head * kHead;
        const uint16_t* qb = hq + ((size_t)16 * qt + n) * kQKVRow + 8 * g;
        const u32x4 q0 = *(const u32x4*)qb, q1 = *(const u32x4*)(qb + 32);
        float e[NT][4];
        float m = -INFINITY;
        const int query = 16 * qt + n;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (kt <= qt) {                                            // wave-uniform
                const uint16_t* kb = hq + ((size_t)kRows + 16 * kt + n) * kQKVRow + 8 * g;
                f32x4 sT = {0.f, 0.f, 0.f, 0.f};
                sT = mfma_op(*(const u32x4*)kb, q0, sT);
                sT = mfma_op(*(const u32x4*)(kb + 32), q1, sT);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kt + 4 * g + r;
                    e[kt][r] = (key <= query && key < Tn) ? sT[r] * scale_log2e : -INFINITY;
                    m = fmaxf(m, e[kt][r]);
                }
            }
        }
        m = rows_allreduce<true>(m);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (kt <= qt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { e[kt][r] = __builtin_amdgcn_exp2f(e[kt][r] - m); sum += e[kt][r]; }
            }
        }
        sum = rows_allreduce<false>(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        f32x4 y[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) y[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            if (kt <= qt) {
                const uint2 pb = make_uint2(pack_op2(e[kt][0], e[kt][1]), pack_op2(e[kt][2], e[kt][3]));
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    // V^T fragment through the transpose read (v_frag's addressing on this phase's row count)
                    const uint16_t* pv = hq + ((size_t)2 * kRows + 16 * kt + 4 * g + (n >> 2)) * kQKVRow + 16 * dt + 4 * (n & 3);
                    const uint2 va = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pv)));
                    y[dt] = mfma_op16(va, pb, y[dt]);
                }
            }
        }
        u32x4* yT = (u32x4*)(hq + (size_t)16 * qt * kQKVRow);          // the q rows of this query tile
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4 yb;
            yb[0] = pack_op2(y[2 * kk][0] * inv, y[2 * kk][1] * inv);
            yb[1] = pack_op2(y[2 * kk][2] * inv, y[2 * kk][3] * inv);
            yb[2] = pack_op2(y[2 * kk + 1][0] * inv, y[2 * kk + 1][1] * inv);
            yb[3] = pack_op2(y[2 * kk + 1][2] * inv, y[2 * kk + 1][3] * inv);
            yT[kk * 64 + ln] = yb;
        }
    };
#pragma unroll 1
    for (int pair = 0; pair < H / 2; ++pair) {
        const int hA = 2 * pair, hB = hA + 1;
        asm volatile("" : "+v"(ln));
        const int n = ln & 15, g = ln >> 4;
        stamp(st, 10);
        u32x4 aE[RPW], aO[RPW];
        f32x4 qa[3][NT];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f32x4 bv = *(const f32x4*)(bqkv + ((hA + hsel) * 12 + 3 * wa + i) * 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < NT; ++t) qa[i][t] = bv;
        }
        gemm_phase<3, NT, kt16(KS)>(qa, qE, qO, qkv_a(pair), 24, xnT + lane, KS * 64, 64, KS);
        prefetch_a<RPW>(aE, aO, proj_a(hA), kWaves * RPW);
        // q/k/v of this wave's head: row = token slot (natural order)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int rt = 3 * wa + i, part = rt >> 2, d0 = (rt & 3) * 16 + 4 * g;
            uint16_t* dst = qkv + (size_t)hsel * kHead + (size_t)part * kRows * kQKVRow + d0;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                *(uint2*)(dst + (size_t)(16 * t + n) * kQKVRow) =
                    make_uint2(pack_op2(qa[i][t][0], qa[i][t][1]), pack_op2(qa[i][t][2], qa[i][t][3]));
        }
        stamp(st, 11);
        __syncthreads();
        stamp(st, 12);
        if (!(BESO_ABL_MASK & 16)) {
            const int head = w & 1, slot = w >> 1;                    // slot 0: tile 4; 1: tile 3; 2: tiles 2 and 0; 3: tile 1
            core(head, slot == 0 ? 4 : slot == 1 ? 3 : slot == 2 ? 2 : 1);
            if (slot == 2) core(head, 0);
        }
        if (pair + 1 < H / 2) prefetch_a<3>(qE, qO, qkv_a(pair + 1), 24);
        stamp(st, 16);
        __syncthreads();
        stamp(st, 17);
        gemm_phase<RPW, NT, false, kNTT>(T.acc, aE, aO, proj_a(hA), kWaves * RPW, (const u32x4*)qkv + lane, kYT, 64, 2);
        prefetch_a<RPW>(aE, aO, proj_a(hB), kWaves * RPW);
        gemm_phase<RPW, NT, false, kNTT>(T.acc, aE, aO, proj_a(hB), kWaves * RPW, (const u32x4*)(qkv + kHead) + lane, kYT, 64, 2);
        stamp(st, 13);
        __syncthreads();                     // the y fragments live in the q rows the next pair overwrites
        stamp(st, 14);
    }
}

// ---------------------------------------------------------------------------------------------
// BF16X3 instances of the two phases: same decomposition (wave = feature slice of the residual tile, weights as A
// fragments from L2, activations as B fragments through LDS, accumulator -> operand chaining), split-bf16 GEMMs,
// exact GELU (erff) and the attention core on the exact-fp32 MFMA.  No software pipelining across phases: this
// mode exists for parity (north-star 1e-4), its speed is set by three MFMAs per fragment pair and two weight images.
// ---------------------------------------------------------------------------------------------
// nn.GELU(): v * Phi(v), erf form (score_gpts.py:107), to fp32-class accuracy in 14 VALU instructions (erff is ~40, and was
// 16 % of this mode's time): erf(x) = sign(x) (1 - (a1 t + .. + a5 t^5) exp(-x^2)), t = 1/(1 + p |x|)  (Abramowitz & Stegun
// 7.1.26, |error| <= 1.5e-7), with v_rcp_f32 / v_exp_f32; measured in fp32 arithmetic: max |GELU error| 4.7e-7 over
// |v| <= 8 -- thirty times below the 2^-16 relative precision of the split-bf16 operand the result becomes next.
__device__ __forceinline__ float gelu_exact(float v) {
    const float ax = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float ex = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
    const float r = fmaf(-p, ex, 1.0f);                 // erf(|v| / sqrt 2)
    const float hv = 0.5f * v;
    return fmaf(fabsf(hv), r, hv);                     // v/2 (1 + sign(v) r)
}

// MLP phase: per hidden chunk  FC1 -> GELU -> [barrier] -> hT (hi | lo) -> [barrier] -> FC2 into the residual.
template <int RPW, int KS, int NW, int NT>
__device__ __forceinline__ void mlp_phase_x3(Tile<RPW>& T, const u32x4* xnT, int xn_lo, u32x4* hT, int h_lo,
                                             const u32x4* __restrict__ w1p, const float* __restrict__ b1f,
                                             const u32x4* __restrict__ w2p, int HT, uint32_t x3_delta, int w, int lane) {
    asm volatile("" : "+v"(lane));
    const int n_chunks = (HT + kChunkTiles - 1) / kChunkTiles;
    constexpr int RC = kChunkTiles / NW, KW = RC / 2;
    constexpr int A2KS = NW * RPW;
#pragma unroll 1
    for (int c = 0; c < n_chunks; ++c) {
        asm volatile("" : "+v"(lane));
        const int g = lane >> 4;
        const int tiles_here = min(kChunkTiles, HT - c * kChunkTiles);
        const int R0 = c * kChunkTiles + RC * w;
        f32x4 h[RC][NT];
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            const f32x4 bias = *(const f32x4*)(b1f + 16 * (R0 + r) + 4 * g);
#pragma unroll
            for (int t = 0; t < NT; ++t) h[r][t] = bias;
        }
        // waves whose rows of this chunk are all padding (zero weights, zero bias) skip the GEMM: GELU(0) = 0
        if (RC * w < tiles_here)
            gemm_x3<RC, NT, 2, kt16(KS)>(h, wptr(w1p + (size_t)(RC * w) * 64, lane).adv((size_t)c * KS * kChunkTiles), kChunkTiles,
                                      x3_delta, xnT + lane, xn_lo, KS * 64, 64, KS);
        u32x4 hh[KW][NT], hl[KW][NT];
#pragma unroll
        for (int j2 = 0; j2 < KW; ++j2)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4& hv = h[2 * j2 + (q >> 1)][t];
                    const SplitPair p = split_op2(gelu_exact(hv[2 * (q & 1)]), gelu_exact(hv[2 * (q & 1) + 1]));
                    hh[j2][t][q] = p.hi;
                    hl[j2][t][q] = p.lo;
                }
        if (c > 0) __syncthreads();              // every wave is done reading hT(c-1)
#pragma unroll
        for (int j2 = 0; j2 < KW; ++j2)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                hT[((size_t)t * kKC + KW * w + j2) * 64 + lane] = hh[j2][t];
                hT[((size_t)t * kKC + KW * w + j2) * 64 + lane + h_lo] = hl[j2][t];
            }
        __syncthreads();                         // hT(c) complete
        gemm_x3<RPW, NT, 2, false, kNTT>(T.acc, wptr(w2p + (size_t)(w * RPW) * 64, lane).adv((size_t)(c * kKC) * A2KS), A2KS, x3_delta,
                                      hT + lane, h_lo, kKC * 64, 64, ((tiles_here >> 1) + 1) & ~1);
    }
    __syncthreads();
}

// Attention phase: head pairs as attn_phase; q/k/v of one head in LDS as fp32 rows [3][rows][kQKVRowF], the core
// (score_gpts.py:69-73) on v_mfma_f32_16x16x4_f32, one sample per wave:
//   S^T[j][i] = sum_d K[j][d] Q[i][d]     16 MFMAs: step (kk4, r) contracts d = 16 kk4 + 4 g + r, A and B read as one
//                                         float4 per lane and kk4 (any assignment of d to contraction slots is a sum)
//   softmax over j as in the bf16 core (D layout: lane (i = lane&15, g) holds keys 4g + r)
//   Y^T[d][i] = sum_j V[j][d] P[i][j]     16 MFMAs: step r contracts j = 4 g + r, so B is the probability register r as
//                                         it stands and A = V[4g + r][16 dt + n]
// and Y (normalised) goes out as split-bf16 B fragments of the out-projection.
template <int RPW, int KS, int HG, int NTP, int NTQ>
__device__ __forceinline__ void attn_phase_x3(Tile<RPW>& T, const u32x4* xnT, int xn_lo, unsigned char* u, int qkv_rows,
                                              int y_off, int y_lo, const u32x4* __restrict__ wqkv,
                                              const float* __restrict__ bqkv, const u32x4* __restrict__ wproj, int H, int hd,
                                              int Tn, int n_samples, uint32_t x3_delta, int w, int lane, const SlotTabs* tb) {
    asm volatile("" : "+v"(lane));
    float* qkv = (float*)u;                               // [3][qkv_rows][kQKVRowF]
    u32x4* yT = (u32x4*)(u + y_off);                      // [(t*2 + kk)*64 + lane], low fragments y_lo behind
    const int wa = w & 3, hsel = w >> 2;
    const float scale_log2e = 1.4426950408889634f * __builtin_amdgcn_rsqf((float)hd);
    auto qkv_a = [&](int pair) { return wptr(wqkv + (size_t)(3 * w) * 64, lane).adv((size_t)pair * KS * 24); };
    auto proj_a = [&](int h) { return wptr(wproj + (size_t)(w * RPW) * 64, lane).adv((size_t)(2 * h) * (kWaves * RPW)); };
    int ln = lane;
    auto write_qkv = [&](const f32x4 (&qa)[3][NTQ]) {
        const int n = ln & 15, g = ln >> 4;
#pragma unroll
        for (int t = 0; t < NTQ; ++t) {
            const int row = tb->row_of_slot[t * 16 + n];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int rt = 3 * wa + i, part = rt >> 2, d0 = (rt & 3) * 16 + 4 * g;
                *(f32x4*)(qkv + ((size_t)part * qkv_rows + row) * kQKVRowF + d0) = qa[i][t];
            }
        }
    };
    auto core = [&]() {
        const int n = ln & 15, g = ln >> 4;
        if (w >= n_samples) return;
        const float* qb = qkv + ((size_t)0 * qkv_rows + w * Tn + n) * kQKVRowF + 4 * g;
        const float* kb = qkv + ((size_t)1 * qkv_rows + w * Tn + n) * kQKVRowF + 4 * g;
        const float* vb = qkv + ((size_t)2 * qkv_rows + w * Tn + 4 * g) * kQKVRowF + n;
        f32x4 q4[4], k4[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { q4[kk] = *(const f32x4*)(qb + 16 * kk); k4[kk] = *(const f32x4*)(kb + 16 * kk); }
        f32x4 y[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) y[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < HG; ++h) {
            // HG > 1: head h of the group sees its own dims [lo_d, hi_d) only (hd % 4 == 0: whole float4s)
            const int lo_d = h * hd, hi_d = lo_d + hd;
            f32x4 sT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (HG > 1 && (16 * kk >= hi_d || 16 * kk + 16 <= lo_d)) continue;      // wave-uniform
                const int d0 = 16 * kk + 4 * g;
                const bool in = HG == 1 || (d0 >= lo_d && d0 < hi_d);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sT = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[kk][r], in ? q4[kk][r] : 0.f, sT, 0, 0, 0);
            }
            float e[4], mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = (4 * g + r <= n) ? sT[r] * scale_log2e : -INFINITY;       // (q k^T)/sqrt(hd), causal, log2 units
                mx = fmaxf(mx, e[r]);
            }
            mx = rows_allreduce<true>(mx);
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(e[r] - mx); sum += e[r]; }
            sum = rows_allreduce<false>(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                if (HG > 1 && (16 * dt >= hi_d || 16 * dt + 16 <= lo_d)) continue;      // wave-uniform
                f32x4 yh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    yh = __builtin_amdgcn_mfma_f32_16x16x4f32(vb[(size_t)r * kQKVRowF + 16 * dt], e[r], yh, 0, 0, 0);
                const int d0 = 16 * dt + 4 * g;
                const bool mine = HG == 1 || (d0 >= lo_d && d0 < hi_d);
#pragma unroll
                for (int r = 0; r < 4; ++r) y[dt][r] = mine ? yh[r] * inv : y[dt][r];
            }
        }
        if (n < Tn) {
            const int tok = tb->slot_of_row[w * Tn + n];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const SplitPair p0 = split_op2(y[2 * kk][0], y[2 * kk][1]), p1 = split_op2(y[2 * kk][2], y[2 * kk][3]);
                const SplitPair p2 = split_op2(y[2 * kk + 1][0], y[2 * kk + 1][1]), p3 = split_op2(y[2 * kk + 1][2], y[2 * kk + 1][3]);
                const u32x4 yh = {p0.hi, p1.hi, p2.hi, p3.hi}, yl = {p0.lo, p1.lo, p2.lo, p3.lo};
                const size_t at = ((size_t)(tok >> 4) * 2 + kk) * 64 + (g << 4) + (tok & 15);
                yT[at] = yh;
                yT[at + y_lo] = yl;
            }
        }
    };
#pragma unroll 1
    for (int pair = 0; pair < H / 2; ++pair) {
        const int hA = 2 * pair, hB = hA + 1;
        asm volatile("" : "+v"(ln));
        const int g = ln >> 4;
        f32x4 qa[3][NTQ];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f32x4 bv = *(const f32x4*)(bqkv + ((hA + hsel) * 12 + 3 * wa + i) * 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < NTQ; ++t) qa[i][t] = bv;
        }
        gemm_x3<3, NTQ, 2, kt16(KS)>(qa, qkv_a(pair), 24, x3_delta, xnT + lane, xn_lo, KS * 64, 64, KS);
        if (hsel == 0) write_qkv(qa);
        __syncthreads();
        core();
        __syncthreads();
        gemm_x3<RPW, NTP, 2, false, kNTT>(T.acc, proj_a(hA), kWaves * RPW, x3_delta, yT + lane, y_lo, 2 * 64, 64, 2);
        if (hsel == 1) write_qkv(qa);
        __syncthreads();
        core();
        __syncthreads();
        gemm_x3<RPW, NTP, 2, false, kNTT>(T.acc, proj_a(hB), kWaves * RPW, x3_delta, yT + lane, y_lo, 2 * 64, 64, 2);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
template <int RPW, int KS, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void mlp_block_kernel(float* __restrict__ x, const char* __restrict__ lw,
                                                           FusedDims d, int M, unsigned long long* stamps, int cap) {
    Stamps st{stamps, cap, 0};
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr LdsMap L = lds_map(KS, true);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * kMT;
    Tile<RPW> T;
    load_x_tile<RPW>(T, x, d.D, m0, M, w, n, g);
    u32x4 a1r[kFc1PF][kChunkTiles / NW];
    mlp_prefetch<KS, NW>(a1r, (const u32x4*)lw, w, lane);
    layernorm_to_lds<RPW, KS, NW>(T, (u32x4*)(lds + L.xnT), (float*)(lds + L.red), d.D, w, lane, (const float*)(lw + d.o_b2), st);
    mlp_phase<RPW, KS, NW>(T, (const u32x4*)(lds + L.xnT), (u32x4*)(lds + L.u), (const u32x4*)lw,
                       (const float*)(lw + d.o_b1), (const u32x4*)(lw + d.o_w2), d.HT, d.KS2p, w, lane, a1r, st);
    store_x_tile<RPW>(T, x, d.D, m0, M, w, n, g);
}

// LN1 + q/k/v projection of a 96-token tile for shapes whose sequences are too long for the fused attention
// phase (score_gpts.py:58-66 after :113's ln1): qkv[M][3D] bf16, rows [q | k | v], for the attention kernel.
template <int RPW, int KS>
__global__ __launch_bounds__(512, 2) void qkv_block_kernel(const float* __restrict__ x, const char* __restrict__ lw,
                                                           FusedDims d, int M, uint16_t* __restrict__ qkv,
                                                           unsigned long long* stamps, int cap) {
    Stamps st{stamps, cap, 0};
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr LdsMap L = lds_map(KS, true);
    int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * kMT;
    const u32x4* xnT = (const u32x4*)(lds + L.xnT);
    {
        Tile<RPW> T;
        load_x_tile<RPW>(T, x, d.D, m0, M, w, n, g);
        layernorm_to_lds<RPW, KS, kWaves, false>(T, (u32x4*)(lds + L.xnT), (float*)(lds + L.red), d.D, w, lane, nullptr, st);
    }
    const size_t ldq = (size_t)3 * d.D;
#pragma unroll 1
    for (int part = 0; part < 3; ++part) {
        asm volatile("" : "+v"(lane));
        const int gg = lane >> 4, nn = lane & 15;
        f32x4 qa[RPW][kNTT];
        const float* bq = (const float*)(lw + d.o_bqkv_lin) + (size_t)part * (kWaves * RPW * 16);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const f32x4 bv = *(const f32x4*)(bq + 16 * (w * RPW + i) + 4 * gg);
#pragma unroll
            for (int t = 0; t < kNTT; ++t) qa[i][t] = bv;
        }
        u32x4 aE[RPW], aO[RPW];
        const WPtr a = wptr((const u32x4*)(lw + d.o_wqkv_lin + (size_t)part * d.part_bytes) + (size_t)(w * RPW) * 64, lane);
        prefetch_a<RPW>(aE, aO, a, kWaves * RPW);
        gemm_phase<RPW, kNTT, kt16(KS)>(qa, aE, aO, a, kWaves * RPW, xnT + lane, KS * 64, 64, KS);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int f0 = 16 * (w * RPW + i) + 4 * gg;
            if (f0 < d.D) {
#pragma unroll
                for (int t = 0; t < kNTT; ++t) {
                    const int tok = m0 + 16 * t + nn;
                    if (tok < M) {
                        uint2 pk;
                        pk.x = pack_op2(qa[i][t][0], qa[i][t][1]);
                        pk.y = pack_op2(qa[i][t][2], qa[i][t][3]);
                        *(uint2*)(qkv + (size_t)tok * ldq + (size_t)part * d.D + f0) = pk;
                    }
                }
            }
        }
    }
}

// Out-projection of the attention output + residual add on a 96-token tile (score_gpts.py:79, :113): the
// y rows (bf16, row-major) are re-read as B fragments (two 8-byte pieces per lane and fragment).
template <int RPW, int KS>
__global__ __launch_bounds__(512, 2) void proj_block_kernel(float* __restrict__ x, const char* __restrict__ lw, FusedDims d,
                                                            int M, const uint16_t* __restrict__ y, int ld_y,
                                                            unsigned long long* stamps, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr LdsMap L = lds_map(KS, true);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * kMT;
    u32x4* yT = (u32x4*)(lds + L.xnT);
    for (int f = w; f < kNTT * KS; f += kWaves) {
        const int t = f / KS, kk = f - t * KS;
        const int tok = m0 + 16 * t + n;
        uint2 lo = make_uint2(0u, 0u), hi = lo;
        if (tok < M) {
            const uint16_t* row = y + (size_t)tok * ld_y + 32 * kk + 4 * g;
            lo = *(const uint2*)row;
            hi = *(const uint2*)(row + 16);
        }
        yT[(size_t)f * 64 + lane] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
    Tile<RPW> T;
    load_x_tile<RPW>(T, x, d.D, m0, M, w, n, g);
    u32x4 aE[RPW], aO[RPW];
    const WPtr a = wptr((const u32x4*)(lw + d.o_wproj_lin) + (size_t)(w * RPW) * 64, lane);
    prefetch_a<RPW>(aE, aO, a, kWaves * RPW);
    __syncthreads();
    gemm_phase<RPW, kNTT, kt16(KS)>(T.acc, aE, aO, a, kWaves * RPW, (const u32x4*)yT + lane, KS * 64, 64, KS);
    const float* bp = (const float*)(lw + d.o_bproj_lin);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const f32x4 bv = *(const f32x4*)(bp + 16 * (w * RPW + i) + 4 * g);
#pragma unroll
        for (int t = 0; t < kNTT; ++t) T.acc[i][t] += bv;
    }
    store_x_tile<RPW>(T, x, d.D, m0, M, w, n, g);
}

// Everything of a layer BEHIND its attention, plus the front of the next layer, for the long-sequence shapes, on one
// 96-token tile whose residual stays in registers throughout:
//     x += proj(y) + b_proj                      (score_gpts.py:79, :113)
//     x += fc2(GELU(fc1(LN2 x)))                 (:105-114)
//     qkv_next = [q | k | v](LN1' x)             (:58-66 of layer + 1; skipped behind the last layer)
// i.e. proj_block_kernel, mlp_block_kernel and the next layer's qkv_block_kernel without the two store / reload round
// trips of the fp32 residual between them (105 MB per layer at 17 k tokens) and two launches fewer per layer.
template <int RPW, int KS>
__global__ __launch_bounds__(512, 2) void tail_block_kernel(float* __restrict__ x, const char* __restrict__ lw,
                                                            const char* __restrict__ lw_next, FusedDims d, int M,
                                                            const uint16_t* __restrict__ y, int ld_y,
                                                            uint16_t* __restrict__ qkv, unsigned long long* stamps, int cap) {
    Stamps st{stamps, cap, 0};
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr LdsMap L = lds_map(KS, true);
    int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * kMT;
    Tile<RPW> T;
    {
        // ---- out-projection + residual (as proj_block_kernel)
        const int n = lane & 15, g = lane >> 4;
        u32x4* yT = (u32x4*)(lds + L.xnT);
        for (int f = w; f < kNTT * KS; f += kWaves) {
            const int t = f / KS, kk = f - t * KS;
            const int tok = m0 + 16 * t + n;
            uint2 lo = make_uint2(0u, 0u), hi = lo;
            if (tok < M) {
                const uint16_t* row = y + (size_t)tok * ld_y + 32 * kk + 4 * g;
                lo = *(const uint2*)row;
                hi = *(const uint2*)(row + 16);
            }
            yT[(size_t)f * 64 + lane] = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
        load_x_tile<RPW>(T, x, d.D, m0, M, w, n, g);
        u32x4 aE[RPW], aO[RPW];
        const WPtr a = wptr((const u32x4*)(lw + d.o_wproj_lin) + (size_t)(w * RPW) * 64, lane);
        prefetch_a<RPW>(aE, aO, a, kWaves * RPW);
        __syncthreads();
        gemm_phase<RPW, kNTT, kt16(KS)>(T.acc, aE, aO, a, kWaves * RPW, (const u32x4*)yT + lane, KS * 64, 64, KS);
        const float* bp = (const float*)(lw + d.o_bproj_lin);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const f32x4 bv = *(const f32x4*)(bp + 16 * (w * RPW + i) + 4 * g);
#pragma unroll
            for (int t = 0; t < kNTT; ++t) T.acc[i][t] += bv;
        }
    }
    {
        // ---- LN2 + MLP (as mlp_block_kernel); the barrier inside the LayerNorm statistics is what orders the writes
        // of xnT behind every wave's reads of yT above (same LDS region)
        u32x4 a1r[kFc1PF][kChunkTiles / kWaves];
        mlp_prefetch<KS, kWaves>(a1r, (const u32x4*)lw, w, lane);
        layernorm_to_lds<RPW, KS, kWaves>(T, (u32x4*)(lds + L.xnT), (float*)(lds + L.red), d.D, w, lane,
                                          (const float*)(lw + d.o_b2), st);
        mlp_phase<RPW, KS, kWaves>(T, (const u32x4*)(lds + L.xnT), (u32x4*)(lds + L.u), (const u32x4*)lw,
                                   (const float*)(lw + d.o_b1), (const u32x4*)(lw + d.o_w2), d.HT, d.KS2p, w, lane, a1r, st);
    }
    if (lw_next != nullptr) {
        // ---- the next layer's LN1 from the registers, then the residual leaves (its accumulators are free for q/k/v)
        layernorm_to_lds<RPW, KS, kWaves, false>(T, (u32x4*)(lds + L.xnT), (float*)(lds + L.red), d.D, w, lane, nullptr, st);
    }
    {
        asm volatile("" : "+v"(lane));
        store_x_tile<RPW>(T, x, d.D, m0, M, w, lane & 15, lane >> 4);
    }
    if (lw_next == nullptr) return;
    const u32x4* xnT = (const u32x4*)(lds + L.xnT);
    const size_t ldq = (size_t)3 * d.D;
#pragma unroll 1
    for (int part = 0; part < 3; ++part) {
        asm volatile("" : "+v"(lane));
        const int gg = lane >> 4, nn = lane & 15;
        f32x4 qa[RPW][kNTT];
        const float* bq = (const float*)(lw_next + d.o_bqkv_lin) + (size_t)part * (kWaves * RPW * 16);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const f32x4 bv = *(const f32x4*)(bq + 16 * (w * RPW + i) + 4 * gg);
#pragma unroll
            for (int t = 0; t < kNTT; ++t) qa[i][t] = bv;
        }
        u32x4 aE[RPW], aO[RPW];
        const WPtr a = wptr((const u32x4*)(lw_next + d.o_wqkv_lin + (size_t)part * d.part_bytes) + (size_t)(w * RPW) * 64, lane);
        prefetch_a<RPW>(aE, aO, a, kWaves * RPW);
        gemm_phase<RPW, kNTT, kt16(KS)>(qa, aE, aO, a, kWaves * RPW, xnT + lane, KS * 64, 64, KS);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int f0 = 16 * (w * RPW + i) + 4 * gg;
            if (f0 < d.D) {
#pragma unroll
                for (int t = 0; t < kNTT; ++t) {
                    const int tok = m0 + 16 * t + nn;
                    if (tok < M) {
                        uint2 pk;
                        pk.x = pack_op2(qa[i][t][0], qa[i][t][1]);
                        pk.y = pack_op2(qa[i][t][2], qa[i][t][3]);
                        *(uint2*)(qkv + (size_t)tok * ldq + (size_t)part * d.D + f0) = pk;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Training forward: the tail block with everything the backward pass keeps (train.hip, loss_grad_e).  One launch replaces
// out-projection (+ residual), LN2, FC1 (+ GELU), FC2 (+ residual) of a layer and LN1 + the q/k/v projection of the next
// one -- six launches of the per-op training forward -- on a 96-token tile whose residual stays in registers; outputs
// are the per-op kernels' buffers in their formats: x_mid, x_out fp32 [M][D]; LayerNorm statistics [M][2]; LayerNorm
// outputs, h, GELU(h), q|k|v as bf16 row-major.  Weights come from the per-step training image (fragment order, LayerNorm
// affine NOT folded: the kernel applies gamma / beta).  bf16 operands, no dropout on the proj / MLP outputs (resid_pdrop
// = 0; with it the residual adds are no longer plain accumulations), any sequence length (there is no attention in here).
// ---------------------------------------------------------------------------------------------
struct TrainImg {            // byte offsets inside one layer's training image
    uint32_t o_w1, o_b1, o_w2, o_b2, o_wqkv, o_bqkv, o_wproj, o_bproj, o_ln1w, o_ln1b, o_ln2w, o_ln2b, part_bytes, layer_bytes;
};
static TrainImg train_img(const FusedDims& d) {
    TrainImg t;
    const uint32_t rt2 = (uint32_t)d.RPW * kWaves, vec = (uint32_t)round_up_sz((size_t)rt2 * 16 * sizeof(float), 256);
    uint32_t cur = 0;
    auto carve = [&](uint32_t bytes) { uint32_t o = cur; cur = (uint32_t)round_up_sz((size_t)cur + bytes, 256); return o; };
    t.part_bytes = rt2 * d.KS * 1024;
    t.o_w1 = carve(d.w1_bytes); t.o_b1 = carve(d.b1_bytes);
    t.o_w2 = carve(d.w2_bytes); t.o_b2 = carve(vec);
    t.o_wqkv = carve(3 * t.part_bytes); t.o_bqkv = carve(3 * vec);
    t.o_wproj = carve(t.part_bytes); t.o_bproj = carve(vec);
    t.o_ln1w = carve(vec); t.o_ln1b = carve(vec); t.o_ln2w = carve(vec); t.o_ln2b = carve(vec);
    t.layer_bytes = cur;
    return t;
}

struct TrainTailArgs {
    const float* x_in; const uint16_t* y; int ld_y;
    float* x_mid; float* x_out; float* st2; uint16_t* xn2; uint16_t* h; uint16_t* g;
    float* st1n; uint16_t* xn1n; uint16_t* qkvn;       // next layer (all null: nothing follows)
};

template <int RPW, int KS>
__global__ __launch_bounds__(512, 2) void train_tail_kernel(const char* __restrict__ lw, const char* __restrict__ lw_next,
                                                            FusedDims d, TrainImg ti, int M, TrainTailArgs a) {
    Stamps st{nullptr, 0, 0};
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr LdsMap L = lds_map(KS, true);
    int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * kMT;
    Tile<RPW> T;
    {
        // ---- x_mid = x_in + proj(y) + b_proj
        const int n = lane & 15, g = lane >> 4;
        u32x4* yT = (u32x4*)(lds + L.xnT);
        for (int f = w; f < kNTT * KS; f += kWaves) {
            const int t = f / KS, kk = f - t * KS;
            const int tok = m0 + 16 * t + n;
            uint2 lo = make_uint2(0u, 0u), hi = lo;
            if (tok < M) {
                // (row pieces beyond D are zero: the padded contraction columns of the weights are zeros too, but the
                // operand must be finite)
                const int c0 = 32 * kk + 4 * g;
                const uint16_t* row = a.y + (size_t)tok * a.ld_y + c0;
                if (c0 < d.D) lo = *(const uint2*)row;
                if (c0 + 16 < d.D) hi = *(const uint2*)(row + 16);
            }
            yT[(size_t)f * 64 + lane] = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
        load_x_tile<RPW>(T, a.x_in, d.D, m0, M, w, n, g);
        u32x4 aE[RPW], aO[RPW];
        const WPtr wp = wptr((const u32x4*)(lw + ti.o_wproj) + (size_t)(w * RPW) * 64, lane);
        prefetch_a<RPW>(aE, aO, wp, kWaves * RPW);
        __syncthreads();
        gemm_phase<RPW, kNTT, kt16(KS)>(T.acc, aE, aO, wp, kWaves * RPW, (const u32x4*)yT + lane, KS * 64, 64, KS);
        const float* bp = (const float*)(lw + ti.o_bproj);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const f32x4 bv = *(const f32x4*)(bp + 16 * (w * RPW + i) + 4 * g);
#pragma unroll
            for (int t = 0; t < kNTT; ++t) T.acc[i][t] += bv;
        }
        store_x_tile<RPW>(T, a.x_mid, d.D, m0, M, w, n, g);
    }
    {
        // ---- LN2 (statistics and affine output kept), MLP (h and GELU(h) kept), x_out
        u32x4 a1r[kFc1PF][kChunkTiles / kWaves];
        mlp_prefetch<KS, kWaves>(a1r, (const u32x4*)(lw + ti.o_w1), w, lane);
        const Rows rows{nullptr, m0, min(kMT, M - m0)};
        const LnTrain lx{(const float*)(lw + ti.o_ln2w), (const float*)(lw + ti.o_ln2b), a.st2, a.xn2, rows, d.D};
        layernorm_to_lds<RPW, KS, kWaves, true, kNTT, 0, LnTrain>(T, (u32x4*)(lds + L.xnT), (float*)(lds + L.red), d.D, w, lane,
                                                                   (const float*)(lw + ti.o_b2), st, 0, lx);
        const MlpTrain mx{a.h, a.g, rows, 4 * d.D};
        mlp_phase<RPW, KS, kWaves, kNTT, kFc1PF, MlpTrain>(T, (const u32x4*)(lds + L.xnT), (u32x4*)(lds + L.u),
                                                            (const u32x4*)(lw + ti.o_w1), (const float*)(lw + ti.o_b1),
                                                            (const u32x4*)(lw + ti.o_w2), d.HT, d.KS2p, w, lane, a1r, st, mx);
    }
    if (lw_next != nullptr) {
        const LnTrain lx{(const float*)(lw_next + ti.o_ln1w), (const float*)(lw_next + ti.o_ln1b), a.st1n, a.xn1n,
                         Rows{nullptr, m0, min(kMT, M - m0)}, d.D};
        layernorm_to_lds<RPW, KS, kWaves, false, kNTT, 0, LnTrain>(T, (u32x4*)(lds + L.xnT), (float*)(lds + L.red), d.D, w, lane,
                                                                    nullptr, st, 0, lx);
    }
    {
        asm volatile("" : "+v"(lane));
        store_x_tile<RPW>(T, a.x_out, d.D, m0, M, w, lane & 15, lane >> 4);
    }
    if (lw_next == nullptr) return;
    const u32x4* xnT = (const u32x4*)(lds + L.xnT);
    const size_t ldq = (size_t)3 * d.D;
#pragma unroll 1
    for (int part = 0; part < 3; ++part) {
        asm volatile("" : "+v"(lane));
        const int gg = lane >> 4, nn = lane & 15;
        f32x4 qa[RPW][kNTT];
        const float* bq = (const float*)(lw_next + ti.o_bqkv) + (size_t)part * (kWaves * RPW * 16);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const f32x4 bv = *(const f32x4*)(bq + 16 * (w * RPW + i) + 4 * gg);
#pragma unroll
            for (int t = 0; t < kNTT; ++t) qa[i][t] = bv;
        }
        u32x4 aE[RPW], aO[RPW];
        const WPtr wq = wptr((const u32x4*)(lw_next + ti.o_wqkv + (size_t)part * ti.part_bytes) + (size_t)(w * RPW) * 64, lane);
        prefetch_a<RPW>(aE, aO, wq, kWaves * RPW);
        gemm_phase<RPW, kNTT, kt16(KS)>(qa, aE, aO, wq, kWaves * RPW, xnT + lane, KS * 64, 64, KS);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int f0 = 16 * (w * RPW + i) + 4 * gg;
            if (f0 < d.D) {
#pragma unroll
                for (int t = 0; t < kNTT; ++t) {
                    const int tok = m0 + 16 * t + nn;
                    if (tok < M)
                        *(uint2*)(a.qkvn + (size_t)tok * ldq + (size_t)part * d.D + f0) =
                            make_uint2(pack_op2(qa[i][t][0], qa[i][t][1]), pack_op2(qa[i][t][2], qa[i][t][3]));
                }
            }
        }
    }
}

// The per-step training image of ALL layers in one launch: every segment is either a matrix in A-fragment order
// (pack_mfma_a_kernel's layout) or a zero-padded fp32 vector; a workgroup serves one segment (table in the kernel argument).
struct TrainPackSeg { const float* src; uint32_t dst; int rows, cols, rt, kt, grp, first_block, tr; };   // rt = 0: vector of `rows` floats padded to `cols`; tr = 1: the matrix is src^T (src is [cols][rows]); 2, 3, 4: the attention phase's q/k/v, out-projection and bias layouts
constexpr int kTrainPackSegs = 96;                       // 13 per layer; the table travels as a kernel argument (< 4 KiB)
struct TrainPackTable { TrainPackSeg seg[kTrainPackSegs]; int n, blocks; };
__global__ void train_pack_kernel(TrainPackTable t, char* __restrict__ img) {
    __shared__ int which;
    if (threadIdx.x == 0) {
        int lo = 0, hi = t.n - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t.seg[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
        which = lo;
    }
    __syncthreads();
    const TrainPackSeg& g = t.seg[which];
    const int nb = (which + 1 < t.n ? t.seg[which + 1].first_block : t.blocks) - g.first_block;
    const size_t start = (size_t)(blockIdx.x - g.first_block) * blockDim.x + threadIdx.x, stride = (size_t)nb * blockDim.x;
    if (g.rt == 0) {
        float* dst = (float*)(img + g.dst);
        if (g.tr == 4) {
            // q / k / v bias (part = grp) in the attention phase's head layout: dst[(hv 3 + part) 64 + d] = src[hv hdv + d] (hdv = rows)
            for (size_t i = start; i < (size_t)g.cols; i += stride) {
                const int hv = (int)(i >> 6), dd = (int)(i & 63);
                dst[((size_t)hv * 3 + g.grp) * kHDP + dd] = dd < g.rows ? g.src[(size_t)hv * g.rows + dd] : 0.f;
            }
            return;
        }
        for (size_t i = start; i < (size_t)g.cols; i += stride) dst[i] = i < (size_t)g.rows ? g.src[i] : 0.f;
        return;
    }
    // Matrices: a thread writes the EIGHT bf16 of one lane of a fragment tile (16 bytes) per iteration -- elements j = 0..3 are
    // columns c0 .. c0 + 3, j = 4..7 columns c0 + 16 .. c0 + 19 of one source row: two 16-byte reads where the source is laid
    // out along c (32-bit index arithmetic: one element per thread with 64-bit divisions took 35 us per image of the step)
    uint16_t* dst = (uint16_t*)(img + g.dst);
    const uint32_t total8 = (uint32_t)g.rt * (uint32_t)g.kt * 64u * (g.tr == 2 ? 4u : 1u);
    const bool vec_ok = ((uintptr_t)g.src & 15) == 0;
    auto put8 = [&](uint32_t i8, uint32_t dst8, bool row_ok, const float* row, int c0, int ncols, size_t cstride) {
        // row: &src[row][0] (cstride 1) or &src[0][row] (cstride = rows: the matrix is stored transposed)
        float v[8];
        if (row_ok && cstride == 1 && vec_ok && c0 + 20 <= ncols && (ncols & 3) == 0) {
            const f32x4 lo = *(const f32x4*)(row + c0), hi = *(const f32x4*)(row + c0 + 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + 16 * (j >> 2) + (j & 3);
                v[j] = (row_ok && c < ncols) ? row[(size_t)c * cstride] : 0.f;
            }
        }
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (uint32_t)f2bf(v[2 * j]) | ((uint32_t)f2bf(v[2 * j + 1]) << 16);
        *(u32x4*)(dst + (size_t)dst8 * 8) = o;
    };
    const uint32_t start8 = (uint32_t)start, stride8 = (uint32_t)stride;
    if (g.tr == 2) {
        // one part (grp: 0 query, 1 key, 2 value) of the head-pair q/k/v image (pack_qkv_kernel's layout, no gamma):
        // rt = virtual heads, rows = their width hdv, cols = D
        for (uint32_t i8 = start8; i8 < total8; i8 += stride8) {
            const uint32_t lane = i8 & 63, tile = i8 >> 6;
            const uint32_t R4 = tile & 3, kk = (tile >> 2) % (uint32_t)g.kt, h = tile / (4u * (uint32_t)g.kt);
            const int dd = 16 * (int)R4 + (int)(lane & 15), c0 = 32 * (int)kk + 4 * (int)(lane >> 4);
            const uint32_t dtile = ((h >> 1) * (uint32_t)g.kt + kk) * 24 + (h & 1) * 12 + (uint32_t)g.grp * 4 + R4;
            put8(i8, dtile * 64 + lane, dd < g.rows, g.src + ((size_t)h * g.rows + (dd < g.rows ? dd : 0)) * g.cols, c0, g.cols, 1);
        }
        return;
    }
    if (g.tr == 3) {
        // out-projection per head k-step (pack_proj_kernel's layout): rows = D, cols = hdv, kt = 2 x virtual heads
        for (uint32_t i8 = start8; i8 < total8; i8 += stride8) {
            const uint32_t lane = i8 & 63, tile = i8 >> 6;
            const uint32_t R = tile % (uint32_t)g.rt, kk = tile / (uint32_t)g.rt;
            const int o = 16 * (int)R + (int)(lane & 15), h = (int)(kk >> 1), d0 = 32 * (int)(kk & 1) + 4 * (int)(lane >> 4);
            // (a head's hdv columns of row o: columns past hdv are zeros, not the next head's)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int dd = d0 + 16 * (j >> 2) + (j & 3);
                v[j] = (o < g.rows && dd < g.cols) ? g.src[(size_t)o * g.rows + h * g.cols + dd] : 0.f;
            }
            u32x4 ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = (uint32_t)f2bf(v[2 * j]) | ((uint32_t)f2bf(v[2 * j + 1]) << 16);
            *(u32x4*)(dst + (size_t)i8 * 8) = ov;
        }
        return;
    }
    for (uint32_t i8 = start8; i8 < total8; i8 += stride8) {
        const uint32_t lane = i8 & 63, tile = i8 >> 6;
        const uint32_t rin = tile % (uint32_t)g.grp, kk = (tile / (uint32_t)g.grp) % (uint32_t)g.kt;
        const uint32_t R = tile / ((uint32_t)g.grp * (uint32_t)g.kt) * (uint32_t)g.grp + rin;
        const int r = 16 * (int)R + (int)(lane & 15), c0 = 32 * (int)kk + 4 * (int)(lane >> 4);
        const bool rv = r < g.rows;
        if (g.tr) put8(i8, i8, rv, g.src + (rv ? r : 0), c0, g.cols, (size_t)g.rows);
        else put8(i8, i8, rv, g.src + (size_t)(rv ? r : 0) * g.cols, c0, g.cols, 1);
    }
}

// BF16X3 for the long-sequence shapes (BASELINE config 5: D = 512, up to 67 tokens -- a sample's tokens do not fit the LDS of
// a split-bf16 workgroup twice over, so the one-launch form is out): the two-launch-per-layer form of tail_block_kernel in
// split-bf16 arithmetic, on tiles of 16 NT token rows in natural order --
//     x += proj(y) + b_proj;  x += fc2(GELU(fc1(LN2 x)))          (tail of layer l:  lw;  y = fp32 attention output)
//     qkv_next = [q | k | v](LN1' x)                              (front of layer l + 1:  lw_next; fp32 rows for the fp32
//                                                                  attention kernel, attention.hip)
// -- either half may be absent (layer 0's q/k/v come from a launch with lw = nullptr, the last layer has no lw_next).
// Phases, layouts and weight images are the one-launch kernel's (layernorm_to_lds / gemm_x3 / mlp_phase_x3; the `lin`
// images of the block kernels with their low halves x3_delta behind).  LDS: [yT | xnT] hi + lo (one region: the LayerNorm
// barrier separates the two uses), hT hi + lo, LayerNorm statistics.
template <int NT, int KS>
struct LdsMapLinX3 {
    static constexpr int a_bytes = NT * KS * 1024, a_lo = a_bytes / 16;          // one half of [yT | xnT]; lo offset in u32x4
    static constexpr int hT = 2 * a_bytes, h_bytes = NT * kKCc * 1024, h_lo = h_bytes / 16;
    static constexpr int red = hT + 2 * h_bytes, total = red + kRedTok * kMT * 4;
};
template <int RPW, int KS, int NT>
__global__ __launch_bounds__(512, 2) void lin_block_x3_kernel(float* __restrict__ x, const char* __restrict__ lw,
                                                              const char* __restrict__ lw_next, FusedDims d, int M,
                                                              const float* __restrict__ y, int ld_y, float* __restrict__ qkv) {
    Stamps st{nullptr, 0, 0};
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    using L = LdsMapLinX3<NT, KS>;
    static_assert(L::total <= 160 * 1024, "LDS of the split-bf16 block kernel");
    int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * 16 * NT;
    u32x4* aT = (u32x4*)lds;
    float* red = (float*)(lds + L::red);
    Tile<RPW> T;
    load_x_tile<RPW, NT>(T, x, d.D, m0, M, w, lane & 15, lane >> 4);
    if (lw != nullptr) {
        // (hT slots the MLP phase reads before it has written them meet zero weights: they must be finite)
        for (int i = threadIdx.x; i < 2 * L::h_bytes / 16; i += kBlock) ((u32x4*)(lds + L::hT))[i] = u32x4{0, 0, 0, 0};
        {
            // ---- out-projection + residual: the fp32 attention output as split-bf16 B fragments
            const int n = lane & 15, g = lane >> 4;
            for (int f = w; f < NT * KS; f += kWaves) {
                const int t = f / KS, kk = f - t * KS;
                const int tok = m0 + 16 * t + n;
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                if (tok < M) {
                    const float* row = y + (size_t)tok * ld_y + 32 * kk + 4 * g;
                    v0 = *(const f32x4*)row;
                    v1 = *(const f32x4*)(row + 16);
                }
                const SplitPair p0 = split_op2(v0[0], v0[1]), p1 = split_op2(v0[2], v0[3]);
                const SplitPair p2 = split_op2(v1[0], v1[1]), p3 = split_op2(v1[2], v1[3]);
                aT[(size_t)f * 64 + lane] = u32x4{p0.hi, p1.hi, p2.hi, p3.hi};
                aT[(size_t)f * 64 + lane + L::a_lo] = u32x4{p0.lo, p1.lo, p2.lo, p3.lo};
            }
            __syncthreads();
            gemm_x3<RPW, NT, 2, kt16(KS), kNTT>(T.acc, wptr((const u32x4*)(lw + d.o_wproj_lin) + (size_t)(w * RPW) * 64, lane),
                                               kWaves * RPW, d.x3_delta, aT + lane, L::a_lo, KS * 64, 64, KS);
            const float* bp = (const float*)(lw + d.o_bproj_lin);
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const f32x4 bv = *(const f32x4*)(bp + 16 * (w * RPW + i) + 4 * g);
#pragma unroll
                for (int t = 0; t < NT; ++t) T.acc[i][t] += bv;
            }
        }
        // ---- LN2 + MLP (the barrier inside the LayerNorm statistics orders the xnT writes behind every wave's yT reads)
        layernorm_to_lds<RPW, KS, kWaves, true, NT, 1>(T, aT, red, d.D, w, lane, (const float*)(lw + d.o_b2), st, L::a_lo);
        mlp_phase_x3<RPW, KS, kWaves, NT>(T, aT, L::a_lo, (u32x4*)(lds + L::hT), L::h_lo, (const u32x4*)lw,
                                          (const float*)(lw + d.o_b1), (const u32x4*)(lw + d.o_w2), d.HT, d.x3_delta, w, lane);
    }
    if (lw_next != nullptr)
        layernorm_to_lds<RPW, KS, kWaves, false, NT, 1>(T, aT, red, d.D, w, lane, nullptr, st, L::a_lo);
    if (lw != nullptr) {
        asm volatile("" : "+v"(lane));
        store_x_tile<RPW, NT>(T, x, d.D, m0, M, w, lane & 15, lane >> 4);
    }
    if (lw_next == nullptr) return;
    const size_t ldq = (size_t)3 * d.D;
#pragma unroll 1
    for (int part = 0; part < 3; ++part) {
        asm volatile("" : "+v"(lane));
        const int gg = lane >> 4, nn = lane & 15;
        f32x4 qa[RPW][NT];
        const float* bq = (const float*)(lw_next + d.o_bqkv_lin) + (size_t)part * (kWaves * RPW * 16);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const f32x4 bv = *(const f32x4*)(bq + 16 * (w * RPW + i) + 4 * gg);
#pragma unroll
            for (int t = 0; t < NT; ++t) qa[i][t] = bv;
        }
        gemm_x3<RPW, NT, 2, kt16(KS)>(qa, wptr((const u32x4*)(lw_next + d.o_wqkv_lin + (size_t)part * d.part_bytes) + (size_t)(w * RPW) * 64, lane),
                                      kWaves * RPW, d.x3_delta, aT + lane, L::a_lo, KS * 64, 64, KS);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int f0 = 16 * (w * RPW + i) + 4 * gg;
            if (f0 < d.D) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int tok = m0 + 16 * t + nn;
                    if (tok < M) *(f32x4*)(qkv + (size_t)tok * ldq + (size_t)part * d.D + f0) = qa[i][t];
                }
            }
        }
    }
}

// Whole transformer layers [l0, l1) over the tile's 8 samples; x stays in registers in between.
// NTL: token tiles that hold the action tokens of a full tile (8 samples x window, rounded up to an even count):
// in the LAST layer only those go through the out-projection, LayerNorm-2 and the MLP -- nothing else reaches
// the head (score_gpts.py:344-354; SURVEY.md section 8 parity note 7: numerically identical per row).
// SPW / NTA: samples and token tiles per workgroup.  8 / 6 is the throughput instance; 2 / 2 is the latency instance for
// small batches (rollouts, BASELINE config 1): a workgroup carries two samples in two token tiles, so a batch of B
// spreads over B/2 CUs and every phase runs a third of the MFMA / LDS / VALU work -- what is left is the L2 -> CU
// stream of the weights.  Same phases, same per-sample arithmetic (results are bit-identical between the instances).
// PX = 1: the BF16X3 instance (split-bf16 GEMMs, exact GELU, fp32 attention core): the parity mode of this kernel.
// CORE = 1: the long-sequence instance (SPW = 1: a sample of up to 16 NTA tokens per workgroup, tokens in natural order).
// LOOP = 1: the sampler-loop instance (S.n evaluations, each followed by its update in the head); LOOP = 0 is one forward and
// compiles to the loop-free code (S is not read).
template <int RPW, int KS, int HG, int NTL, int SPW = kSPW, int NTA = kNTT, int PX = 0, int CORE = 0, int LOOP = 0>
__global__ __launch_bounds__(512, 2) void layers_kernel(float* __restrict__ x, const char* __restrict__ lw0,
                                                        FusedDims d, int l0, int l1, int n_samples_total, int Tn,
                                                        EdgeArgs e, SampleSteps S, unsigned long long* stamps, int cap) {
    Stamps st{stamps, cap, 0};
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    static_assert(CORE == 0 || (SPW == 1 && PX == 0), "long-sequence instance");
    constexpr LdsMap Lb = lds_map(KS, false, CORE == 1 ? NTA : kNTT);
    constexpr LdsMapX3 X = lds_map_x3(KS, NTA);
    constexpr LdsMap L = PX ? LdsMap{0, X.u, X.red, X.tab, X.xs, X.total, 0} : Lb;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    constexpr int NTLa = NTL < NTA ? NTL : NTA;
    // the latency instances are bound by the L2 -> CU weight stream: more FC1 weight fragments in flight per wave
#ifndef BESO_LONG_PAIRED
#define BESO_LONG_PAIRED 1               // long-sequence instance: both heads of a pair in LDS, their cores at once (0: A/B)
#endif
#ifndef BESO_LONG_PF1
#define BESO_LONG_PF1 BESO_FC1_PF        // ... in the long-sequence instance
#endif
    constexpr int PF1 = CORE == 1 ? BESO_LONG_PF1 : (NTA < kNTT ? BESO_LAT_PF1 : kFc1PF);
    // (CORE = 1 with a classifier-free pair: the workgroup's one REAL sample runs as two passes -- virtual samples 2 b, 2 b + 1)
    const bool two_pass = CORE == 1 && e.two;
    const int s0 = two_pass ? 2 * (int)blockIdx.x : (int)blockIdx.x * SPW;
    const int n_samples = two_pass ? 1 : min(SPW, n_samples_total - s0);
    // action tokens first whenever both network edges are inside the kernel (otherwise x travels in natural order)
    SlotTabs* tb = (SlotTabs*)(lds + L.tab);
    float* xs = (float*)(lds + L.xs);
    // (CORE = 1: natural order -- the long-sequence core addresses slots by position, and five tiles leave nothing to peel)
    const bool actions_first = CORE == 0;
    build_slot_tabs(tb, n_samples, Tn, e.t, d.G, actions_first, CORE == 1 ? 7 : 4);
    // the action windows of the workgroup's real samples (contiguous in `action`): x_T of the sampler loop / the noisy action
    LoopState ls{-1, 0.f, 0.f, 0.f, true, 0.f, 0};
    {
        int b0; bool un0;
        sample_of(e, s0, b0, un0);
        const int n_el = (two_pass ? 1 : n_samples / (e.two ? 2 : 1)) * e.t * d.act;
        for (int i = threadIdx.x; i < n_el; i += kBlock) xs[i] = e.action[(size_t)b0 * e.t * d.act + i];
    }
    Tile<RPW> T;
    stamp(st, 100);
    const char* gw = lw0 + (size_t)d.L * d.layer_bytes;          // per-model image (embeddings, head)
    // Sampler loop (K8 fused): S.n > 0 evaluations of the network back to back, each followed by
    // its step's update in the head; samples never interact, so the workgroup needs nothing from outside between them.
    const int n_evals = LOOP ? S.n : 1;
#pragma unroll 1
    for (int ev = 0; ev < n_evals; ++ev) {
    if constexpr (LOOP) ls.sigma = S.rec[ev].sigma;
    const int n_pass = CORE == 1 ? (two_pass ? 2 : 1) : 1;
    for (int pass = 0; pass < n_pass; ++pass) {          // (one trip, known at compile time, in every instance but CORE = 1)
    const int s0p = s0 + pass;                         // the virtual sample(s) of this pass
    int tid = threadIdx.x;
    if constexpr (LOOP) asm volatile("" : "+v"(tid));      // (per evaluation: nothing thread-derived is kept live across the loop)
    // LDS that is read but never written by the phases must be finite: the attention-output fragments of
    // padding tokens, and the 8 q/k/v rows past the last token slot (a sample's 16-row window reaches them).
    // (Per evaluation: the head's partial sums are laid over them.)
    if constexpr (PX) {
        for (int i = tid; i < (L.red - L.u) / 16; i += kBlock) ((u32x4*)(lds + L.u))[i] = u32x4{0, 0, 0, 0};
    } else {
    for (int i = tid; i < kNTT * 2 * 64; i += kBlock) ((u32x4*)(lds + L.u + kQKVBytes))[i] = u32x4{0, 0, 0, 0};
    for (int i = tid; i < 3 * 8 * kQKVRow / 2; i += kBlock) {
        const int part = i / (8 * kQKVRow / 2), rem = i % (8 * kQKVRow / 2);
        ((uint32_t*)(lds + L.u))[((size_t)part * kQKVRows + kMT) * kQKVRow / 2 + rem] = 0u;
    }
    }
    __syncthreads();
    stamp(st, 1);
    if (BESO_ABL_MASK & 4) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            T.fvalid[i] = 16 * (w * RPW + i) + 4 * g < d.D;
#pragma unroll
            for (int t = 0; t < kNTT; ++t) T.acc[i][t] = f32x4{0.1f * n, 0.2f, 0.3f * g, 0.4f};
        }
    } else embed_tile<RPW, PX, CORE == 1 ? 7 : 4>(T, e, d, gw, s0p, n_samples, Tn, w, lane, tb, xs, ls.sigma, st);
    stamp(st, 43);
    // The last layer (when this launch contains it and the action tokens of the tile fit NTL token tiles) runs its
    // out-projection, LayerNorm-2 and MLP on the action-token tiles only; it is peeled off the loop -- a branch
    // between the two variants INSIDE the loop costs 150 spilled VGPRs.
    const int n_valid = n_samples * Tn;                 // token slots in use (the empty ones are behind them)
    const bool peel = actions_first && l1 == d.L && l1 > l0 && n_samples * e.t <= 16 * NTLa && (!PX || NTLa < NTA);
    const int l_loop_end = peel ? l1 - 1 : l1;
    auto layer_weights = [&](int l) {
#if BESO_FUSED_ABLATE == 4
        (void)l;
        return lw0;                                              // timing experiment: the weights of ONE layer fit in L2
#else
        return lw0 + (size_t)l * d.layer_bytes;
#endif
    };
    if constexpr (PX) {
        auto layer_x3 = [&](const char* lw, auto NTPc) {
            constexpr int NTP = decltype(NTPc)::value;
            u32x4* xnT = (u32x4*)(lds + L.xnT);
            layernorm_to_lds<RPW, KS, kWaves, true, NTA, 1>(T, xnT, (float*)(lds + L.red), d.D, w, lane,
                                                            (const float*)(lw + d.o_bproj), st, X.xn_lo);
            attn_phase_x3<RPW, KS, HG, NTP, NTA>(T, xnT, X.xn_lo, lds + L.u, X.qkv_rows, X.yT, X.y_lo,
                                                 (const u32x4*)(lw + d.o_wqkv), (const float*)(lw + d.o_bqkv),
                                                 (const u32x4*)(lw + d.o_wproj), d.Hv, d.hd, Tn, n_samples, d.x3_delta, w, lane, tb);
            layernorm_to_lds<RPW, KS, kWaves, true, NTP, 1>(T, xnT, (float*)(lds + L.red), d.D, w, lane,
                                                            (const float*)(lw + d.o_b2), st, X.xn_lo);
            mlp_phase_x3<RPW, KS, kWaves, NTP>(T, xnT, X.xn_lo, (u32x4*)(lds + L.u), X.h_lo, (const u32x4*)lw,
                                               (const float*)(lw + d.o_b1), (const u32x4*)(lw + d.o_w2), d.HT, d.x3_delta, w, lane);
        };
#pragma unroll 1
        for (int l = l0; l < l_loop_end; ++l) layer_x3(layer_weights(l), std::integral_constant<int, NTA>{});
        if (peel) layer_x3(layer_weights(l1 - 1), std::integral_constant<int, NTLa>{});
    } else {
    for (int l = l0; l < l_loop_end; ++l) {
        const char* lw = layer_weights(l);
        stamp(st, 2);
        u32x4 qE[3], qO[3];
        attn_prefetch<KS>(qE, qO, (const u32x4*)(lw + d.o_wqkv), w, lane);
        layernorm_to_lds<RPW, KS, kWaves, true, NTA>(T, (u32x4*)(lds + L.xnT), (float*)(lds + L.red), d.D, w, lane,
                                          (const float*)(lw + d.o_bproj), st, 0, LnPlain{}, n_valid);
        stamp(st, 7);
        if constexpr (CORE == 1 && BESO_LONG_PAIRED)
            attn_phase_long<RPW, KS, NTA>(T, (const u32x4*)(lds + L.xnT), lds + L.u, (const u32x4*)(lw + d.o_wqkv),
                                          (const float*)(lw + d.o_bqkv), (const u32x4*)(lw + d.o_wproj), d.Hv, d.hd, Tn, w, lane,
                                          qE, qO, st);
        else
        attn_phase<RPW, KS, HG, NTA, NTA, CORE>(T, (const u32x4*)(lds + L.xnT), lds + L.u, (const u32x4*)(lw + d.o_wqkv),
                                (const float*)(lw + d.o_bqkv), (const u32x4*)(lw + d.o_wproj), d.Hv, d.hd, Tn, n_samples, w,
                                lane, tb, qE, qO, st);
        stamp(st, 3);
        u32x4 a1r[PF1][kChunkTiles / kWaves];
        mlp_prefetch<KS, kWaves, PF1>(a1r, (const u32x4*)lw, w, lane);
        layernorm_to_lds<RPW, KS, kWaves, true, NTA>(T, (u32x4*)(lds + L.xnT), (float*)(lds + L.red), d.D, w, lane,
                                          (const float*)(lw + d.o_b2), st, 0, LnPlain{}, n_valid);
        stamp(st, 6);
        mlp_phase<RPW, KS, kWaves, NTA, PF1>(T, (const u32x4*)(lds + L.xnT), (u32x4*)(lds + L.u), (const u32x4*)lw,
                                   (const float*)(lw + d.o_b1), (const u32x4*)(lw + d.o_w2), d.HT, d.KS2p, w, lane, a1r, st, MlpPlain{}, n_valid);
    }
    if (peel) {
        const char* lw = layer_weights(l1 - 1);
        stamp(st, 2);
        u32x4 qE[3], qO[3];
        attn_prefetch<KS>(qE, qO, (const u32x4*)(lw + d.o_wqkv), w, lane);
        layernorm_to_lds<RPW, KS, kWaves, true, NTA>(T, (u32x4*)(lds + L.xnT), (float*)(lds + L.red), d.D, w, lane,
                                          (const float*)(lw + d.o_bproj), st, 0, LnPlain{}, n_valid);
        stamp(st, 7);
        attn_phase<RPW, KS, HG, NTLa, NTA, CORE>(T, (const u32x4*)(lds + L.xnT), lds + L.u, (const u32x4*)(lw + d.o_wqkv),
                                     (const float*)(lw + d.o_bqkv), (const u32x4*)(lw + d.o_wproj), d.Hv, d.hd, Tn, n_samples, w,
                                     lane, tb, qE, qO, st);
        stamp(st, 3);
        u32x4 a1r[PF1][kChunkTiles / kWaves];
        mlp_prefetch<KS, kWaves, PF1>(a1r, (const u32x4*)lw, w, lane);
        layernorm_to_lds<RPW, KS, kWaves, true, NTLa>(T, (u32x4*)(lds + L.xnT), (float*)(lds + L.red), d.D, w, lane,
                                                     (const float*)(lw + d.o_b2), st);
        stamp(st, 6);
        mlp_phase<RPW, KS, kWaves, NTLa, PF1>(T, (const u32x4*)(lds + L.xnT), (u32x4*)(lds + L.u), (const u32x4*)lw,
                                        (const float*)(lw + d.o_b1), (const u32x4*)(lw + d.o_w2), d.HT, d.KS2p, w, lane, a1r, st);
    }
    }
    stamp(st, 4);
    if constexpr (LOOP) {
        const StepRec rec = S.rec[ev];
        ls.mode = rec.mode; ls.c0 = rec.c0; ls.c1 = rec.c1; ls.c2 = rec.c2; ls.ev = ev;
        ls.last = ev + 1 == n_evals;
    }
    {
        // (peel: the action tokens are the first n_samples * t slots, i.e. inside the first NTLa token tiles)
        if constexpr (CORE == 1) head_tile<RPW, NTA>(T, e, d, gw, (float*)(lds + L.red), (float*)(lds + L.u), s0p, n_samples, Tn, w, lane, tb, xs, ls, st,
                                                     two_pass ? pass : -1, (float*)(lds + L.cfgc));
        else if (peel && NTLa < kNTT) head_tile<RPW, NTLa>(T, e, d, gw, (float*)(lds + L.red), (float*)(lds + L.u), s0, n_samples, Tn, w, lane, tb, xs, ls, st);
        else head_tile<RPW>(T, e, d, gw, (float*)(lds + L.red), (float*)(lds + L.u), s0, n_samples, Tn, w, lane, tb, xs, ls, st);
    }
    stamp(st, 5);
    if (CORE == 1 && pass + 1 < n_pass) __syncthreads();      // the conditional pass's outputs are in cfgc, its partial sums read
    }
    if (LOOP && ev + 1 < n_evals) __syncthreads();      // the head's partial sums are read, the next input is in xs
    }
    stamp(st, 101);
}

// ---------------------------------------------------------------------------------------------
// Training forward of ALL layers as one launch (train.hip, loss_grad_e): the phases of layers_kernel with the store hooks of
// train_tail_kernel plus the attention phase's (AttnTrain) -- everything the backward pass keeps goes out in the per-op
// training kernels' buffers and formats: per layer x_mid / x_out fp32, LayerNorm statistics and affine outputs, q|k|v, the
// attention output, h and GELU(h) as bf16 row-major.  The residual tile is loaded from the embedding kernel's x0 (action
// tokens first, as in layers_kernel) and never leaves the registers; the LAST layer runs its out-projection, LayerNorm-2 and
// MLP on the action-token tiles only and writes them as the COMPACT action rows (row b t + i) on which the per-op step
// continues with ln_f, head and loss.  Weights: the per-step training image (fragment order, LayerNorm affine not folded:
// the weight gradients need the affine outputs as operands); bf16 operands; attention dropout inside the core
// (attn_small_kernel's mask); dropout on the proj / MLP outputs (resid_pdrop > 0: block-push) in the RD = 1 instances.  One launch replaces 44 of the per-op
// forward at six layers; at 1024 kitchen samples the four-samples-per-workgroup instance is one workgroup per CU.
// ---------------------------------------------------------------------------------------------
struct TrainImgW { uint32_t o_ln1w, o_ln1b, o_ln2w, o_ln2b, layer_bytes; };    // behind a layer's FusedDims sections (o_b1 .. o_bproj)
static TrainImgW train_img_whole(const FusedDims& d) {
    TrainImgW t;
    const uint32_t vec = (uint32_t)round_up_sz((size_t)d.RPW * kWaves * 16 * sizeof(float), 256);
    t.o_ln1w = d.layer_bytes; t.o_ln1b = t.o_ln1w + vec; t.o_ln2w = t.o_ln1b + vec; t.o_ln2b = t.o_ln2w + vec;
    t.layer_bytes = t.o_ln2b + vec;
    return t;
}

template <int RPW, int KS, int HG, int NTL, int SPW, int NTA, int RD = 0>      // RD = 1: dropout on the proj / MLP outputs (resid_pdrop > 0)
__global__ __launch_bounds__(512, 2) void train_fwd_kernel(const char* __restrict__ img, FusedDims d, TrainImgW ti,
                                                           int n_samples_total, int Tn, TrainWholeBufs a) {
    Stamps st{nullptr, 0, 0};
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr LdsMap L = lds_map(KS);
    constexpr int NTLa = NTL < NTA ? NTL : NTA;
    constexpr int PF1 = NTA < kNTT ? BESO_LAT_PF1 : kFc1PF;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s0 = blockIdx.x * SPW;
    const int n_samples = min(SPW, n_samples_total - s0);
    SlotTabs* tb = (SlotTabs*)(lds + L.tab);
    build_slot_tabs(tb, n_samples, Tn, a.t, d.G, true);
    // LDS that is read but never written by the phases must be finite (layers_kernel): the attention-output fragments of
    // padding tokens and the 8 q/k/v rows past the last token slot -- of THIS instance's 16 NTA slots (round 6: the four-sample
    // instance in three token tiles lets the last sample's 16-row window reach row 48)
    for (int i = threadIdx.x; i < kNTT * 2 * 64; i += kBlock) ((u32x4*)(lds + L.u + kQKVBytes))[i] = u32x4{0, 0, 0, 0};
    for (int i = threadIdx.x; i < 3 * 8 * kQKVRow / 2; i += kBlock) {
        const int part = i / (8 * kQKVRow / 2), rem = i % (8 * kQKVRow / 2);
        ((uint32_t*)(lds + L.u))[((size_t)part * kQKVRows + 16 * NTA) * kQKVRow / 2 + rem] = 0u;
    }
    __syncthreads();
    const Rows rows_all{tb->row_of_slot, s0 * Tn, n_samples * Tn};     // token rows b T + position
    const Rows rows_act{nullptr, s0 * a.t, n_samples * a.t};            // compact action rows b t + i = the first slots
    Tile<RPW> T;
    load_x_rows<RPW, NTA>(T, a.x0, d.D, rows_all, w, lane);
    u32x4* xnT = (u32x4*)(lds + L.xnT);
    float* red = (float*)(lds + L.red);
    const float inv_keep = a.p_attn > 0.f ? 1.0f / (1.0f - a.p_attn) : 1.f;
    const float inv_keep_r = RD && a.p_resid > 0.f ? 1.0f / (1.0f - a.p_resid) : 1.f;
    // RD = 1.  The residual adds of a layer are MFMA accumulations onto the residual registers; with a dropout between the
    // branch and the add (block-push: resid_pdrop 0.05) the branch must exist alone first.  The residual in front of either
    // branch already lies in memory -- x0 / the previous layer's kept x_out in front of the attention, the kept x_mid in front
    // of the MLP (written by this very lane a phase earlier) -- so: once the LayerNorm has read the residual registers they are
    // set to the branch's bias, the GEMMs accumulate the branch into them, and resid_dropout_add masks the branch and adds the
    // residual back from memory (an L2 hit: 16 bytes per lane and accumulator).  Same masks as the per-op forward (EpiResid).
    auto layer = [&](int l, auto NTPc, const Rows& rows_tail, uint16_t* ybuf) {
        constexpr int NTP = decltype(NTPc)::value;             // token tiles behind the attention: all, or the action tiles
        const char* lw = img + (size_t)l * ti.layer_bytes;
        char* wl = a.ws + (size_t)l * a.stride;
        u32x4 qE[3], qO[3];
        attn_prefetch<KS>(qE, qO, (const u32x4*)(lw + d.o_wqkv), w, lane);
        const LnTrain lx1{(const float*)(lw + ti.o_ln1w), (const float*)(lw + ti.o_ln1b), (float*)(wl + a.st1),
                          (uint16_t*)(wl + a.xn1), rows_all, d.D};
        layernorm_to_lds<RPW, KS, kWaves, !RD, NTA, 0, LnTrain>(T, xnT, red, d.D, w, lane, (const float*)(lw + d.o_bproj), st, 0, lx1);
        if constexpr (RD) set_bias_rows<RPW, NTP>(T, (const float*)(lw + d.o_bproj), w, lane);
        const AttnTrain ax{(uint16_t*)(wl + a.qkv), ybuf, rows_all, rows_tail, d.D, s0, d.H, a.p_attn, inv_keep, a.seed,
                           (uint32_t)(4 * l)};
        attn_phase<RPW, KS, HG, NTP, NTA, 0, AttnTrain>(T, xnT, lds + L.u, (const u32x4*)(lw + d.o_wqkv),
                                                        (const float*)(lw + d.o_bqkv), (const u32x4*)(lw + d.o_wproj), d.Hv, d.hd,
                                                        Tn, n_samples, w, lane, tb, qE, qO, st, ax);
        if constexpr (RD)
            resid_dropout_add<RPW, NTP>(T, l == 0 ? a.x0 : (const float*)(wl - a.stride + a.x_out), d.D, rows_all, rows_tail,
                                        a.p_resid, inv_keep_r, a.seed, (uint32_t)(4 * l + 1), w, lane);
        // (the first FC1 weight fragments are requested BEFORE the kept x_mid leaves: loads issued behind a burst of stores wait
        //  for the stores' acknowledgements)
        u32x4 a1r[PF1][kChunkTiles / kWaves];
        mlp_prefetch<KS, kWaves, PF1>(a1r, (const u32x4*)lw, w, lane);
        if (!RD && a.x_bf16) store_x_rows_bf16<RPW, NTP>(T, (uint16_t*)(wl + a.x_mid), d.D, rows_tail, w, lane);
        else store_x_rows<RPW, NTP>(T, (float*)(wl + a.x_mid), d.D, rows_tail, w, lane);
        const LnTrain lx2{(const float*)(lw + ti.o_ln2w), (const float*)(lw + ti.o_ln2b), (float*)(wl + a.st2),
                          (uint16_t*)(wl + a.xn2), rows_tail, d.D};
        layernorm_to_lds<RPW, KS, kWaves, !RD, NTP, 0, LnTrain>(T, xnT, red, d.D, w, lane, (const float*)(lw + d.o_b2), st, 0, lx2);
        if constexpr (RD) set_bias_rows<RPW, NTP>(T, (const float*)(lw + d.o_b2), w, lane);
        const MlpTrain mx{(uint16_t*)(wl + a.h), (uint16_t*)(wl + a.g), rows_tail, 4 * d.D};
        mlp_phase<RPW, KS, kWaves, NTP, PF1, MlpTrain>(T, xnT, (u32x4*)(lds + L.u), (const u32x4*)lw, (const float*)(lw + d.o_b1),
                                                       (const u32x4*)(lw + d.o_w2), d.HT, d.KS2p, w, lane, a1r, st, mx);
        if constexpr (RD)
            resid_dropout_add<RPW, NTP>(T, (const float*)(wl + a.x_mid), d.D, rows_tail, rows_tail, a.p_resid, inv_keep_r, a.seed,
                                        (uint32_t)(4 * l + 2), w, lane);
        // (the last layer's x_out -- the compact action rows -- stays fp32: the final LayerNorm and the head continue on it)
        if (!RD && a.x_bf16 && l + 1 < d.L) store_x_rows_bf16<RPW, NTP>(T, (uint16_t*)(wl + a.x_out), d.D, rows_tail, w, lane);
        else store_x_rows<RPW, NTP>(T, (float*)(wl + a.x_out), d.D, rows_tail, w, lane);
    };
#pragma unroll 1
    for (int l = 0; l + 1 < d.L; ++l)
        layer(l, std::integral_constant<int, NTA>{}, rows_all, (uint16_t*)(a.ws + (size_t)l * a.stride + a.y));
    layer(d.L - 1, std::integral_constant<int, NTLa>{}, rows_act, (uint16_t*)(a.ws + a.ya));
}

// ---------------------------------------------------------------------------------------------
// Training backward: the DATA-GRADIENT GEMMs in the transposed formulation (round 4).  The 128 x 128 tile kernel of train.hip
// moves both operands of every stage through LDS (128 KiB of LDS traffic per 2.1 MFLOP stage: it is LDS-bandwidth bound at
// ~20 % of the matrix pipe, and the N = D data gradients fill only half of its workgroup slots).  Here, as in the inference
// kernels: dX^T = W^T-side A fragments straight from L2 (per-step image of the TRANSPOSED weights in fragment order) x the
// incoming gradient tile as B fragments, staged ONCE in LDS for all k-steps; a workgroup = 8 waves owns 16 NT token rows and
// all D output features (N = 4 D: chunks of 8 RPW row tiles), the accumulators leave as row-major stores.
//   parts = 1: out = in W            (FC1 data gradient K = 4 D -> fp32 dxn; out-projection data gradient K = D -> bf16 dy)
//   parts = 3: out = sum_p in[:, p D ..] W_p   (q | k | v data gradient: three K = D contractions into one accumulator)
//   gelu:      dh = (in W2) * GELU'(h) as bf16 + its column sums (FC1 bias gradient)          (N = 4 D, K = D)
// ---------------------------------------------------------------------------------------------
struct DgradArgs {
    const uint16_t* in; int ld_in;        // [M][ld_in] bf16; part p contracts over columns [p K, (p + 1) K)
    int K, parts, kt;                     // real contraction width per part; k-steps per part in the image (even, zero padded)
    int N, n_chunks;                      // real output features; chunks of 8 RPW row tiles
    float* out32; uint16_t* out16; int ld_out;
    const uint16_t* h; uint16_t* dh; float* colsum;      // GELU' epilogue (all three set): h, dh [M][N]; colsum: slab [workgroups][N]
    int M;
};

// LayerNorm backward as the EPILOGUE of a data gradient whose output is that LayerNorm's dxn (q|k|v -> LN1, FC1 -> LN2): the
// workgroup holds all D features of its 16 NT tokens in the accumulators, so the two row sums of the backward formula are an
// exchange through LDS like the forward's statistics, and dxn never exists in memory:
//     dres_out = dres_in + (dy - mean(dy) - xh mean(dy xh)) rstd,   dy = dxn gamma,  xh = (x - mean) rstd
// plus the bf16 copy for the next GEMM and the workgroup's partial sums of dgamma, dbeta and the consumer's bias gradient
// (rows [3][D] of the LayerNorm-reduction slab).  x == nullptr: off.
struct LnBwdEpi {
    const float* x; const float* stats; const float* gamma;     // LayerNorm input [M][D], (mean, rstd) [M][2], gamma [D]
    const float* dres_in; float* dres_out;                      // residual gradient: in (nullptr: none), out [M][D]
    uint16_t* dxb;                                              // bf16 copy of dres_out x keep-scale(site) [M][D]
    float* part;                                                // [workgroups][3][D]
    float p, inv_keep; uint32_t seed, site; int skip_mod;       // dropout of the branch behind this LayerNorm (p = 0: none): element
                                                                // (row, f) keeps drop_scale(seed, site, row D + f); rows with
                                                                // row % skip_mod == 0 carry none (skip_mod > 0: the sigma token)
    int x_bf16;                                                 // x is [M][D] bf16 (the one-launch forward's kept residuals, round 5)
};

// the gradient tile as B fragments: lane (n, g) of fragment (t, kk) holds columns 32 kk + 4 g .. +3 and 32 kk + 16 + 4 g .. +3
// of token m0 + 16 t + n (zeros beyond K / M: the padded k-steps of the weights are zeros too, but operands must be finite).
// bT: [t][part][kk][lane] -- a token tile's k-steps of all parts in a row: the parts are ONE contraction of parts kt k-steps
template <int NT, int kStage = 6>
__device__ __forceinline__ void stage_grad_tile(u32x4* bT, const uint16_t* __restrict__ in, int ld_in, int K, int parts, int kt,
                                                int M, int m0, int w, int lane) {
    const int n = lane & 15, g = lane >> 4;
    const int per = NT * kt, total = parts * per;
    // kStage fragments per wave in flight, values selected afterwards -- one fragment at a time is one L2 round trip per fragment
    // (18 per wave for K = 4 D: two thirds of the kernel's time).  Round 6: (i) the q|k|v data gradient stages 3 x 12 x 3 = 108
    // fragments = 13.5 per wave; six at a time were three round trips in a row (14.5 k of the kernel's 63 k cycles,
    // profiles/r06_dgrad_stamps.txt) -- its instance takes all of them at once; (ii) the tile is read through a buffer resource
    // whose range ends with row M - 1: rows past the end are zeros by the range check, a fragment is ONE 32-bit offset (its
    // second half 32 bytes on) instead of two clamped 64-bit pointers -- 14 fragments in flight cost 70 registers, not 112
    // (M ld_in < 2^30 elements: checked on the host).  Columns past K inside a row are read (the next part / row) and dropped.
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)((uint32_t)M * (uint32_t)ld_in * 2u), 0x00020000);
    for (int f0 = w; f0 < total; f0 += kWaves * kStage) {
        u32x2 lo[kStage], hi[kStage];
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
            const int f = min(f0 + u * kWaves, total - 1);
            const int p = f / per, r = f - p * per, t = r / kt, kk = r - t * kt;
            const uint32_t off = ((uint32_t)(m0 + 16 * t + n) * (uint32_t)ld_in + (uint32_t)(p * K + 32 * kk + 4 * g)) * 2u;
            lo[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0);
            hi[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, off + 32u, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
            const int f = f0 + u * kWaves;
            if (f < total) {
                const int p = f / per, r = f - p * per, t = r / kt, kk = r - t * kt;
                const int tok = m0 + 16 * t + n, c0 = 32 * kk + 4 * g;
                const u32x2 z = {0u, 0u};
                const u32x2 l2 = (tok < M && c0 < K) ? lo[u] : z, h2 = (tok < M && c0 + 16 < K) ? hi[u] : z;
                bT[((size_t)(t * parts + p) * kt + kk) * 64 + lane] = u32x4{l2.x, l2.y, h2.x, h2.y};
            }
        }
    }
}

__device__ __forceinline__ float row16_sum(float v) {       // sum over the 16 lanes of a DPP row, in every lane
    v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xf, 0xf, false));
    v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xf, 0xf, false));
    v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x124, 0xf, 0xf, false));
    v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x128, 0xf, 0xf, false));
    return v;
}

// gelu_grad_poly (common.h) on the packed-fp32 pipe, two values per call: the same fma chain, the same bits
__device__ __forceinline__ f32x2 gelu_grad2(f32x2 v) {
    f32x2 vc;
    vc.x = __builtin_amdgcn_fmed3f(v.x, -4.0f, 4.0f);
    vc.y = __builtin_amdgcn_fmed3f(v.y, -4.0f, 4.0f);
    const f32x2 s = vc * vc;
    f32x2 p = __builtin_elementwise_fma(s, (f32x2)(BESO_GELU_GRAD_C7), (f32x2)(BESO_GELU_GRAD_C6));
    p = __builtin_elementwise_fma(p, s, (f32x2)(BESO_GELU_GRAD_C5));
    p = __builtin_elementwise_fma(p, s, (f32x2)(BESO_GELU_GRAD_C4));
    p = __builtin_elementwise_fma(p, s, (f32x2)(BESO_GELU_GRAD_C3));
    p = __builtin_elementwise_fma(p, s, (f32x2)(BESO_GELU_GRAD_C2));
    p = __builtin_elementwise_fma(p, s, (f32x2)(BESO_GELU_GRAD_C1));
    p = __builtin_elementwise_fma(p, s, (f32x2)(BESO_GELU_GRAD_C0));
    return __builtin_elementwise_fma(vc, p, (f32x2)(0.5f));
}

// the chunk's h values (GELU' epilogue), requested before the GEMM: they arrive under it
template <int RPW, int NT>
__device__ __forceinline__ void load_h_tile(uint2 (&hu)[RPW][NT], const uint16_t* __restrict__ h, int N, int M, int m0, int row0,
                                            int lane) {
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int f0 = min(16 * (row0 + i) + 4 * g, N - 4);
#pragma unroll
        for (int t = 0; t < NT; ++t) hu[i][t] = *(const uint2*)(h + (size_t)min(m0 + 16 * t + n, M - 1) * N + f0);
    }
}

// dh = acc * GELU'(h) as bf16 [M][N] + its column sums over the tile's tokens of the values AS STORED (what the weight gradient
// sees) into the workgroup's row of a slab [workgroups][N] that one small launch adds up (atomics from 235 workgroups onto the
// same 1440 addresses cost 85 of this kernel's 125 us).  emit(i, t, pk): the stored pair of words (zeros outside N / M).
template <int RPW, int NT, class Emit>
__device__ __forceinline__ void gelu_bwd_epilogue(const f32x4 (&acc)[RPW][NT], const uint2 (&hu)[RPW][NT], uint16_t* __restrict__ dh,
                                                  float* __restrict__ colsum_row, int N, int M, int m0, int row0, int lane, Emit emit) {
    const int n = lane & 15, g = lane >> 4;
    // Round 6: the derivative of ALL RPW x NT pieces first, branch-free (h was loaded through clamped addresses: every value is
    // finite) -- 2 RPW NT independent packed chains the scheduler can interleave.  Under the per-piece `if (f0 < N && tok < M)`
    // of round 5 each piece was a basic block of its own: two dependent chains of eight v_pk_fma_f32 (8.4 cycles each when
    // dependent) with a wait state between every pair, nine times over -- 6.5 ... 9.5 k cycles per chunk of train_mlp_bwd_kernel
    // for ~300 instructions of arithmetic (profiles/r06_mlp_bwd_stamps.txt).  Only the store stays under the condition.
    uint2 pkv[RPW][NT];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint2 hv = hu[i][t];
            const f32x2 d01 = gelu_grad2(f32x2{__uint_as_float(hv.x << 16), __uint_as_float(hv.x & 0xffff0000u)});
            const f32x2 d23 = gelu_grad2(f32x2{__uint_as_float(hv.y << 16), __uint_as_float(hv.y & 0xffff0000u)});
            pkv[i][t].x = pack_op2(acc[i][t][0] * d01.x, acc[i][t][1] * d01.y);
            pkv[i][t].y = pack_op2(acc[i][t][2] * d23.x, acc[i][t][3] * d23.y);
        }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int f0 = 16 * (row0 + i) + 4 * g;
        f32x4 cs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tok = m0 + 16 * t + n;
            const bool ok = f0 < N && tok < M;
            const uint2 pk = ok ? pkv[i][t] : make_uint2(0u, 0u);
            if (ok && !(BESO_TRAIN_FWD_ABL & 64)) *(uint2*)(dh + (size_t)tok * N + f0) = pk;
            cs[0] += __uint_as_float(pk.x << 16); cs[1] += __uint_as_float(pk.x & 0xffff0000u);
            cs[2] += __uint_as_float(pk.y << 16); cs[3] += __uint_as_float(pk.y & 0xffff0000u);
            emit(i, t, pk);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[r] = row16_sum(cs[r]);
        if (n == 0 && f0 < N && !(BESO_TRAIN_FWD_ABL & 32)) *(f32x4*)(colsum_row + f0) = cs;
    }
}

struct NoMark { __device__ __forceinline__ void operator()(int) const {} };
// What the LayerNorm-backward epilogue reads, as registers: requested by ln_bwd_load -- which a kernel may call long before the
// accumulators exist (train_dgrad_kernel: together with its gradient tile, so that ONE memory round trip serves both; round 6:
// profiles/r06_dgrad_stamps.txt showed the epilogue's loads + the barrier behind them at 19 k of the kernel's 71 k cycles,
// a round trip of 5 us to tensors of 16 ... 130 MB) -- and consumed by ln_bwd_finish.
template <int RPW, int NT, bool X16 = false>       // X16: the LayerNorm input is known to be bf16 (two dwords per piece instead of four held)
struct LnBwdIn {
    typedef typename std::conditional<X16, u32x2, f32x4>::type XR;
    f32x4 gam[RPW], dr[RPW][NT];
    XR xr[RPW][NT];                                // the LayerNorm input as loaded (bf16 in an f32x4: two dwords used)
    float mean[NT], rstd[NT];
    bool live[NT];
    uint32_t eoff[NT];                             // element offset of (token of tile t, this lane's first feature), clamped rows
};
// Every [M][D] tensor is addressed through a buffer resource on its (uniform) base with ONE 32-bit element offset per token
// tile: 64-bit pointers per (row tile, token tile) and tensor were ~50 VGPRs of the epilogue (M D < 2^29 elements: checked on the
// host).  The incoming residual gradient is requested here too (round 5 read it inside the last loop under its per-lane
// condition: nine loads waited for one after the other).  No incoming gradient: a valid address, values dropped.
template <int RPW, int NT, bool X16>
__device__ __forceinline__ void ln_bwd_load(LnBwdIn<RPW, NT, X16>& in, const LnBwdEpi& e, int D, int M, int m0, int w, int lane) {
    const int n = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)e.x, 0, 0x7fffffff, 0x00020000);
    const uint32_t fbase = (uint32_t)(16 * w * RPW + 4 * g);      // this lane's first feature; row tile i adds 16 i
    uint32_t eoff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tok = min(m0 + 16 * t + n, M - 1);
        in.live[t] = m0 + 16 * t + n < M;
        const float2 st2 = *(const float2*)(e.stats + 2 * (size_t)tok);
        in.mean[t] = st2.x; in.rstd[t] = st2.y;
        eoff[t] = (uint32_t)tok * (uint32_t)D + fbase;
        in.eoff[t] = eoff[t];
    }
    auto eo_of = [&](int i, int t) {                       // (a padding feature tile reads column 0 of its row)
        return (int)fbase + 16 * i < D ? eoff[t] + 16u * i : eoff[t] - fbase;
    };
#pragma unroll
    for (int i = 0; i < RPW; ++i)
        in.gam[i] = (int)fbase + 16 * i < D ? *(const f32x4*)(e.gamma + fbase + 16 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
    // (the uniform choice -- bf16 or fp32 LayerNorm input -- is taken OUTSIDE the unrolled loops: a branch per (row tile, token
    //  tile) cut the epilogue into ~40 basic blocks and its registers into scratch)
    if constexpr (X16) {
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) in.xr[i][t] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, eo_of(i, t) * 2u, 0, 0);
    } else {
        if (e.x_bf16) {
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(rs_x, eo_of(i, t) * 2u, 0, 0);
                    in.xr[i][t] = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f};
                }
        } else {
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    in.xr[i][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, eo_of(i, t) * 4u, 0, 0));
        }
    }
}
// ... and the incoming residual gradient (a call of its own: train_dgrad_kernel holds the other inputs across its GEMM, and these
// 4 RPW NT registers on top of them spilled)
template <int RPW, int NT, bool X16>
__device__ __forceinline__ void ln_bwd_load_dr(LnBwdIn<RPW, NT, X16>& in, const LnBwdEpi& e, int D, int w, int lane) {
    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc((void*)(e.dres_in ? e.dres_in : e.dres_out), 0, 0x7fffffff, 0x00020000);
    const uint32_t fbase = (uint32_t)(16 * w * RPW + 4 * (lane >> 4));
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint32_t eo = (int)fbase + 16 * i < D ? in.eoff[t] + 16u * i : in.eoff[t] - fbase;
            in.dr[i][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, eo * 4u, 0, 0));
        }
}

// LnBwdEpi on the accumulators acc = dxn^T (row tiles w RPW .. of the D features x NT token tiles); red: 16 NT kRedTok floats of
// LDS that no wave reads any more once every wave has arrived here.  emit(i, t, pk): the bf16 pair of words stored in dxb (zeros
// outside D / M).
template <int RPW, int NT, bool X16, class Emit, class Mark = NoMark>
__device__ __forceinline__ void ln_bwd_finish(const f32x4 (&acc)[RPW][NT], LnBwdIn<RPW, NT, X16>& in, const LnBwdEpi& e, int D, int M,
                                              int m0, float* red, int w, int lane, Emit emit, Mark mark = Mark{}) {
    const int n = lane & 15, g = lane >> 4;
    const float invD = 1.0f / (float)D;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)e.dres_out, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_xb = __builtin_amdgcn_make_buffer_rsrc((void*)e.dxb, 0, 0x7fffffff, 0x00020000);
    const bool has_in = e.dres_in != nullptr;
    const uint32_t fbase = (uint32_t)(16 * w * RPW + 4 * g);
    // (plain local arrays from here on: through the struct, the closure below kept the small members in scratch)
    f32x4 xh[RPW][NT], dr[RPW][NT], gam[RPW];
    float mean[NT], rstd[NT];
    bool live[NT];
    uint32_t eoff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { mean[t] = in.mean[t]; rstd[t] = in.rstd[t]; live[t] = in.live[t]; eoff[t] = in.eoff[t]; }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        gam[i] = in.gam[i];
#pragma unroll
        for (int t = 0; t < NT; ++t) dr[i][t] = in.dr[i][t];
    }
    if (X16 || e.x_bf16) {
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                uint32_t ux, uy;
                if constexpr (X16) { ux = in.xr[i][t].x; uy = in.xr[i][t].y; }
                else { ux = __float_as_uint(in.xr[i][t][0]); uy = __float_as_uint(in.xr[i][t][1]); }
                const f32x4 xv = {__uint_as_float(ux << 16), __uint_as_float(ux & 0xffff0000u), __uint_as_float(uy << 16),
                                  __uint_as_float(uy & 0xffff0000u)};
                xh[i][t] = ((int)fbase + 16 * i < D && live[t]) ? (xv - mean[t]) * rstd[t] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
    } else if constexpr (!X16) {
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) xh[i][t] = in.xr[i][t];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                xh[i][t] = ((int)fbase + 16 * i < D && live[t]) ? (xh[i][t] - mean[t]) * rstd[t] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    mark(61);
    __syncthreads();                               // every wave is through its last B fragments: the region becomes `red`
    mark(62);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const f32x4 dy = acc[i][t] * gam[i], p2 = dy * xh[i][t];
            s1 += (dy[0] + dy[1]) + (dy[2] + dy[3]);
            s2 += (p2[0] + p2[1]) + (p2[2] + p2[3]);
        }
        s1 = rows_allreduce<false>(s1);
        s2 = rows_allreduce<false>(s2);
        if (g == 0) *(float2*)(red + (size_t)(16 * t + n) * kRedTok + 2 * w) = make_float2(s1, s2);
    }
    mark(63);
    __syncthreads();
    mark(64);
    f32x4 ag[RPW], ab[RPW], ac[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) { ag[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[i] = ag[i]; ac[i] = ag[i]; }
    // (dropout or none: chosen once, outside the unrolled loops)
    auto finish = [&](auto drop_tag) {
        constexpr bool DROP = decltype(drop_tag)::value;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4* pr = (const f32x4*)(red + (size_t)(16 * t + n) * kRedTok);
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int k = 0; k < kWaves / 2; ++k) { const f32x4 v = pr[k]; c1 += v[0] + v[2]; c2 += v[1] + v[3]; }
            c1 *= invD; c2 *= invD;
            const int tok = m0 + 16 * t + n;
            const bool masked = DROP && !(e.skip_mod > 0 && tok % e.skip_mod == 0);
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int f0 = (int)fbase + 16 * i;
                uint2 pk = make_uint2(0u, 0u);
                if (f0 < D && live[t]) {
                    const f32x4 go = acc[i][t], dy = go * gam[i];
                    f32x4 tot = (dy - c1 - xh[i][t] * c2) * rstd[t];
                    const uint32_t eo = eoff[t] + 16u * i;
                    if (has_in) tot += dr[i][t];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tot), rs_out, eo * 4u, 0, 0);
                    if (DROP) {
                        const size_t idx = (size_t)tok * D + f0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) tot[j] *= masked ? drop_scale(e.seed, e.site, idx + j, e.p, e.inv_keep) : 1.0f;
                    }
                    pk = make_uint2(pack_op2(tot[0], tot[1]), pack_op2(tot[2], tot[3]));
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk.x, pk.y}, rs_xb, eo * 2u, 0, 0);
                    ag[i] += go * xh[i][t]; ab[i] += go; ac[i] += tot;
                }
                emit(i, t, pk);
            }
        }
    };
    if (e.p > 0.f) finish(std::true_type{}); else finish(std::false_type{});
    mark(65);
    // the workgroup's partial sums over its tokens: 16 lanes of a row by DPP, lane n == 0 writes
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int f0 = (int)fbase + 16 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ag[i][r] = row16_sum(ag[i][r]); ab[i][r] = row16_sum(ab[i][r]); ac[i][r] = row16_sum(ac[i][r]); }
        if (n == 0 && f0 < D) {
            float* o = e.part + (size_t)blockIdx.x * 3 * D + f0;
            *(f32x4*)o = ag[i]; *(f32x4*)(o + D) = ab[i]; *(f32x4*)(o + 2 * D) = ac[i];
        }
    }
}

template <int RPW, int NT, int PFA, int MODE>       // MODE 0: plain store or the GELU' epilogue; 1 / 2: the LayerNorm-backward epilogue on an
                                                     // fp32-or-bf16 / a known-bf16 input (separate instances: a kernel that could take either
                                                     // path at run time kept both register sets alive across the GEMM and spilled)
__global__ __launch_bounds__(512, 2) void train_dgrad_kernel(const char* __restrict__ wimg, DgradArgs a, LnBwdEpi e,
                                                            unsigned long long* stamps, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int RT = RPW * kWaves;
    constexpr bool LN = MODE != 0, X16 = MODE == 2;
    Stamps st{stamps, cap, 0};
    int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * 16 * NT;
    u32x4* bT = (u32x4*)lds;                               // [part][t][kk][lane]
    stamp<3>(st, 100);
    stamp<3>(st, 70);
    // a workgroup alone on its CU streams the weights: the ring keeps PFA k-steps of fragments in flight per wave (with the
    // two of gemm_phase the kernel ran at 37 GB/s per CU: the L2 round trip is ~0.8 us, a k-step of MFMAs 0.07 us).
    // parts > 1 (q | k | v): the part images lie back to back (part_bytes = RT kt KiB) and so do a token tile's B fragments:
    // ONE ring over parts kt k-steps instead of a restart per part.  Round 6: the first chunk's ring is requested in FRONT of
    // the gradient tile's loads (its round trip runs under the staging).
    u32x4 ar[PFA][RPW];
    const WPtr wbase = wptr((const u32x4*)wimg + (size_t)w * RPW * 64, lane);
    const uint32_t chunk_bytes = (uint32_t)(a.kt * RT) * 1024u;
    prefetch_ring<RPW, PFA>(ar, wbase, RT);
    // the LayerNorm-backward epilogue's inputs travel WITH the gradient tile (one memory round trip for both; n_chunks == 1 there)
    LnBwdIn<RPW, NT, X16> lin;
    if constexpr (LN) ln_bwd_load<RPW, NT, X16>(lin, e, a.N, a.M, m0, w, lane);
    // (every fragment of the tile in flight at once where the registers allow it: kt <= 12 per part)
    if (a.parts == 3) stage_grad_tile<NT, 14>(bT, a.in, a.ld_in, a.K, a.parts, a.kt, a.M, m0, w, lane);
    else stage_grad_tile<NT, 6>(bT, a.in, a.ld_in, a.K, a.parts, a.kt, a.M, m0, w, lane);
    stamp<3>(st, 71);
    __syncthreads();
    stamp<3>(st, 72);
#pragma unroll 1
    for (int c = 0;; ++c) {
        asm volatile("" : "+v"(lane));
        f32x4 acc[RPW][NT];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n = lane & 15, g = lane >> 4;
        uint2 hu[RPW][NT];
        if (!LN && a.dh != nullptr) load_h_tile<RPW, NT>(hu, a.h, a.N, a.M, m0, c * RT + w * RPW, lane);
        {
            const WPtr wp{wbase.rs, wbase.so + (uint32_t)c * chunk_bytes, wbase.lo};
            gemm_phase_ring<RPW, NT, PFA>(acc, ar, wp, RT, (const u32x4*)bT + lane, a.parts * a.kt * 64, 64, a.parts * a.kt);
        }
        stamp<3>(st, 73);
        if constexpr (LN) {
            ln_bwd_load_dr<RPW, NT, X16>(lin, e, a.N, w, lane);
            ln_bwd_finish<RPW, NT, X16>(acc, lin, e, a.N, a.M, m0, (float*)lds, w, lane, [](int, int, uint2) {}, [&](int id) { stamp<3>(st, id); });
        } else if (a.dh != nullptr) {
            gelu_bwd_epilogue<RPW, NT>(acc, hu, a.dh, a.colsum + (size_t)blockIdx.x * a.N, a.N, a.M, m0, c * RT + w * RPW, lane,
                                       [](int, int, uint2) {});
        } else {
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int f0 = 16 * (c * RT + w * RPW + i) + 4 * g;
                if (f0 >= a.N) continue;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int tok = m0 + 16 * t + n;
                    if (tok >= a.M) continue;
                    if (a.out32) *(f32x4*)(a.out32 + (size_t)tok * a.ld_out + f0) = acc[i][t];
                    else *(uint2*)(a.out16 + (size_t)tok * a.ld_out + f0) =
                             make_uint2(pack_op2(acc[i][t][0], acc[i][t][1]), pack_op2(acc[i][t][2], acc[i][t][3]));
                }
            }
        }
        stamp<3>(st, 74);
        // (only the stand-alone FC2 + GELU' form has more than one chunk; the ring is dead across the epilogue above)
        if (LN || c + 1 >= a.n_chunks) break;
        prefetch_ring<RPW, PFA>(ar, WPtr{wbase.rs, wbase.so + (uint32_t)(c + 1) * chunk_bytes, wbase.lo}, RT);
    }
    stamp<3>(st, 101);
}

// The second half of a layer's backward in ONE kernel (round 4): dyo -> FC2 data gradient x GELU'(h) = dh -> FC1 data gradient
// -> LayerNorm-2 backward (+ residual) = dym -> out-projection data gradient = dy.  dh and dym leave for the weight gradients
// AND stay in LDS as the B fragments of the GEMM behind them (they were written and read back: 10 D of the 44 D bytes per
// token these three launches moved).  dh is produced in chunks of 8 RPW row tiles (train_dgrad_kernel's), and a chunk is RT / 2
// k-steps of FC1's contraction: FC1 accumulates chunk c while FC2 computes chunk c + 1 into the other LDS buffer.
struct MlpBwdArgs {
    const uint16_t* dyo;                                  // [M][D]
    const uint16_t* h; uint16_t* dh; float* colsum;       // [M][4 D]; slab [workgroups][4 D]
    uint16_t* dy;                                         // [M][D]
    uint32_t o_w2T, o_w1T, o_pT;                          // inside the layer's image of the transposed weights
    int D, kt_d, kt_h, n_chunks, M;
};

template <int RPW, int NT, int PFA>
__global__ __launch_bounds__(512, 2) void train_mlp_bwd_kernel(const char* __restrict__ lw, MlpBwdArgs a, LnBwdEpi e,
                                                              unsigned long long* stamps, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int RT = RPW * kWaves, KC = RT / 2;         // k-steps of FC1 per chunk of dh
    Stamps st{stamps, cap, 0};
    stamp<2>(st, 100);
    stamp<2>(st, 50);
    int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m0 = blockIdx.x * 16 * NT, N4 = 4 * a.D;
    u32x4* bT = (u32x4*)lds;                               // dyo: [t][kk][lane], kt_d k-steps; later the LayerNorm exchange
    u32x4* hT = bT + (size_t)NT * a.kt_d * 64;             // dh chunk: [2][t][KC][lane]; later dym [t][kt_d][lane]
    // Round 6 (profiles/r06_mlp_bwd_stamps.txt: the GEMMs of this kernel ran at 2.2 x the time their weight stream needs, each
    // of its nine phases starting with an empty ring behind the stores of the epilogue in front of it, each chunk with h's HBM
    // round trip in front of its ring):
    //  * ONE weight ring from FC2(0) to FC1(last): the tail of every GEMM refills the ring with the first k-steps of the next
    //    one (gemm_phase_ring_cont), so those requests are older than the dh stores between them;
    //  * h of chunk c + 1 is requested at the START of chunk c's GELU' epilogue -- behind the ring of FC1(c), in front of ~3 us
    //    of VALU work -- and h of chunk 0 with the first ring, in front of the dyo tile's own HBM round trip.
    const WPtr wb = wptr((const u32x4*)lw + (size_t)w * RPW * 64, lane);
    auto w2_of = [&](int c) { return WPtr{wb.rs, a.o_w2T + (uint32_t)(c * a.kt_d * RT) * 1024u, wb.lo}; };
    auto w1_of = [&](int c) { return WPtr{wb.rs, a.o_w1T + (uint32_t)(c * KC * RT) * 1024u, wb.lo}; };
    u32x4 ar[PFA][RPW];
    prefetch_ring<RPW, PFA>(ar, w2_of(0), RT);
    uint2 hu[RPW][NT];
    load_h_tile<RPW, NT>(hu, a.h, N4, a.M, m0, w * RPW, lane);
    stage_grad_tile<NT>(bT, a.dyo, a.D, a.D, 1, a.kt_d, a.M, m0, w, lane);
    stamp<2>(st, 51);
    __syncthreads();
    stamp<2>(st, 52);
    f32x4 acc2[RPW][NT];
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc2[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the pair of words (i, t) of this lane as half `tile & 1` of B fragment (t, tile >> 1): row tile = 16 features = half a k-step
    auto put = [&](u32x4* frag, int ks, int i, int t, uint2 pk) {
        const int tl = w * RPW + i;
        *((uint2*)(frag + ((size_t)t * ks + (tl >> 1)) * 64 + lane) + (tl & 1)) = pk;
    };
#pragma unroll 1
    for (int c = 0; c < a.n_chunks; ++c) {
        asm volatile("" : "+v"(lane));
        u32x4* hc = hT + (size_t)(c & 1) * NT * KC * 64;
        const int kc = min(KC, a.kt_h - c * KC);           // FC1 k-steps of this chunk (a multiple of PFA; <= 0: padding only)
        const bool more = c + 1 < a.n_chunks;
        {
            f32x4 acc[RPW][NT];
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const WPtr nx = kc > 0 ? w1_of(c) : (more ? w2_of(c + 1) : wptr_none(wb));
            gemm_phase_ring_cont<RPW, NT, PFA>(acc, ar, w2_of(c), RT, (const u32x4*)bT + lane, a.kt_d * 64, 64, a.kt_d, nx, RT);
            stamp<2>(st, 53);
            uint2 hn[RPW][NT];
            load_h_tile<RPW, NT>(hn, a.h, N4, a.M, m0, (c + 1) * RT + w * RPW, lane);     // (past the last chunk: clamped addresses, unused)
            gelu_bwd_epilogue<RPW, NT>(acc, hu, a.dh, a.colsum + (size_t)blockIdx.x * N4, N4, a.M, m0, c * RT + w * RPW, lane,
                                       [&](int i, int t, uint2 pk) { put(hc, KC, i, t, pk); });
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) hu[i][t] = hn[i][t];
            stamp<2>(st, 54);
        }
        __syncthreads();                                   // chunk c of dh is in LDS (and every wave is through FC1 of chunk c - 1)
        stamp<2>(st, 55);
        if (kc > 0) {
            const WPtr nx = more ? w2_of(c + 1) : wptr_none(wb);
            gemm_phase_ring_cont<RPW, NT, PFA>(acc2, ar, w1_of(c), RT, (const u32x4*)hc + lane, KC * 64, 64, kc, nx, RT);
        }
        stamp<2>(st, 56);
    }
    // LayerNorm-2 backward on dxn2 = acc2; dym also as the B fragments of the out-projection's data gradient (the first dh buffer:
    // the epilogue's two barriers lie between the last FC1 read and these writes)
    {
        LnBwdIn<RPW, NT> lin;
        ln_bwd_load<RPW, NT, false>(lin, e, a.D, a.M, m0, w, lane);
        ln_bwd_load_dr<RPW, NT, false>(lin, e, a.D, w, lane);
        ln_bwd_finish<RPW, NT, false>(acc2, lin, e, a.D, a.M, m0, (float*)bT, w, lane, [&](int i, int t, uint2 pk) { put(hT, a.kt_d, i, t, pk); },
                               [&](int id) { stamp<2>(st, id); });
    }
    stamp<2>(st, 57);
    __syncthreads();
    stamp<2>(st, 58);
    {
        asm volatile("" : "+v"(lane));
        f32x4 acc[RPW][NT];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const WPtr wp{wb.rs, a.o_pT, wb.lo};
        prefetch_ring<RPW, PFA>(ar, wp, RT);
        gemm_phase_ring<RPW, NT, PFA>(acc, ar, wp, RT, (const u32x4*)hT + lane, a.kt_d * 64, 64, a.kt_d);
        stamp<2>(st, 59);
        const int n = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int f0 = 16 * (w * RPW + i) + 4 * g;
            if (f0 >= a.D) continue;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tok = m0 + 16 * t + n;
                if (tok < a.M)
                    *(uint2*)(a.dy + (size_t)tok * a.D + f0) =
                        make_uint2(pack_op2(acc[i][t][0], acc[i][t][1]), pack_op2(acc[i][t][2], acc[i][t][3]));
            }
        }
    }
    stamp<2>(st, 60);
    stamp<2>(st, 101);
}

// out_j[f] = sum_b slab_j[b][f]: the FC1 bias gradients of several layers from their workgroup slabs, one launch
struct SlabRed { const float* slab[kMaxLayers]; float* out[kMaxLayers]; int n; };
__global__ __launch_bounds__(256) void slab_reduce_kernel(SlabRed t, int n_blocks, int N) {
    // 16 columns x 16 row groups per workgroup, eight rows in flight per thread (64 columns x 4 row groups with two in flight:
    // 115 workgroups, each thread 30 L2 round trips in a row -- 26 us for 1.4 MB per layer)
    __shared__ float red[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4, f = blockIdx.x * 16 + cx;
    const float* src = t.slab[blockIdx.y];
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (f < N) {
        int b = ry;
        for (; b + 112 < n_blocks; b += 128) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += src[(size_t)(b + 16 * u) * N + f];
        }
        for (; b < n_blocks; b += 16) a[0] += src[(size_t)b * N + f];
    }
    red[ry][cx] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (ry == 0 && f < N) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v += red[r][cx];
        t.out[blockIdx.y][f] = v;
    }
}

// phase stamps: the development build only (beso_debug_set_stamps); the product library carries no such state
#if BESO_DEV_API
unsigned long long* g_stamps = nullptr;
int g_stamps_cap = 0;
#else
constexpr unsigned long long* g_stamps = nullptr;
constexpr int g_stamps_cap = 0;
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel ON A DEVICE: the "already set" flag of a launch site
// is a bit per device (a process may drive several GPUs), written with a relaxed atomic (launch sites are reached from any
// thread; setting the attribute twice is harmless)
struct LdsAttr { std::atomic<unsigned long long> devices{0}; };
template <typename K>
hipError_t ensure_lds(K kernel, size_t bytes, LdsAttr* done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev < 64 && (done->devices.load(std::memory_order_relaxed) & bit)) return hipSuccess;
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev < 64) done->devices.fetch_or(bit, std::memory_order_relaxed);
    return e;
}

template <int RPW, int KS, int NW>
hipError_t launch_mlp_block(float* x, const char* lw, const FusedDims& d, int M, hipStream_t s) {
    constexpr LdsMap L = lds_map(KS, true);
    static LdsAttr attr;
    hipError_t e = ensure_lds(mlp_block_kernel<RPW, KS, NW>, L.total, &attr);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL((mlp_block_kernel<RPW, KS, NW>), dim3((M + kMT - 1) / kMT), dim3(64 * NW), L.total, s, x, lw, d, M, g_stamps, g_stamps_cap);
    return hipGetLastError();
}

template <int RPW, int KS>
hipError_t launch_lin_blocks(int which, float* x, const char* lw, const FusedDims& d, int M, void* buf, int ld, hipStream_t s) {
    constexpr LdsMap L = lds_map(KS, true);
    static LdsAttr attr_q, attr_p;
    hipError_t e = which == 0 ? ensure_lds(qkv_block_kernel<RPW, KS>, L.total, &attr_q)
                              : ensure_lds(proj_block_kernel<RPW, KS>, L.total, &attr_p);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    const dim3 grid((M + kMT - 1) / kMT), block(512);
    if (which == 0)
        hipLaunchKernelGGL((qkv_block_kernel<RPW, KS>), grid, block, L.total, s, (const float*)x, lw, d, M, (uint16_t*)buf,
                           g_stamps, g_stamps_cap);
    else
        hipLaunchKernelGGL((proj_block_kernel<RPW, KS>), grid, block, L.total, s, x, lw, d, M, (const uint16_t*)buf, ld,
                           g_stamps, g_stamps_cap);
    return hipGetLastError();
}

template <int RPW, int KS>
hipError_t launch_tail_block(float* x, const char* lw, const char* lw_next, const FusedDims& d, int M, const void* y, int ld_y,
                             void* qkv, hipStream_t s) {
    constexpr LdsMap L = lds_map(KS, true);
    static LdsAttr attr;
    hipError_t e = ensure_lds(tail_block_kernel<RPW, KS>, L.total, &attr);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL((tail_block_kernel<RPW, KS>), dim3((M + kMT - 1) / kMT), dim3(512), L.total, s, x, lw, lw_next, d, M,
                       (const uint16_t*)y, ld_y, (uint16_t*)qkv, g_stamps, g_stamps_cap);
    return hipGetLastError();
}

// One instance of layers_kernel: LDS attribute once, then the launch (LOOP = 1: the sampler-loop form, steps.n evaluations).
template <int RPW, int KS, int HG, int NTL, int SPW, int NTA, int PX, int CORE, int LOOP>
hipError_t launch_instance(size_t lds_bytes, int grid, float* x, const char* lw0, const FusedDims& d, int l0, int l1,
                           int n_samples, int Tn, const EdgeArgs& edge, const SampleSteps& steps, hipStream_t s) {
    static LdsAttr attr;
    hipError_t e = ensure_lds(layers_kernel<RPW, KS, HG, NTL, SPW, NTA, PX, CORE, LOOP>, lds_bytes, &attr);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL((layers_kernel<RPW, KS, HG, NTL, SPW, NTA, PX, CORE, LOOP>), dim3(grid), dim3(512), lds_bytes, s, x, lw0, d,
                       l0, l1, n_samples, Tn, edge, steps, g_stamps, g_stamps_cap);
    return hipGetLastError();
}
template <int RPW, int KS, int HG, int NTL, int SPW, int NTA, int PX, int CORE>
hipError_t launch_either(size_t lds_bytes, float* x, const char* lw0, const FusedDims& d, int l0, int l1, int n_samples, int Tn,
                         const EdgeArgs& edge, const SampleSteps& steps, hipStream_t s) {
    const int grid = (n_samples + SPW - 1) / SPW;
    if (steps.n > 0)
        return launch_instance<RPW, KS, HG, NTL, SPW, NTA, PX, CORE, 1>(lds_bytes, grid, x, lw0, d, l0, l1, n_samples, Tn, edge, steps, s);
    return launch_instance<RPW, KS, HG, NTL, SPW, NTA, PX, CORE, 0>(lds_bytes, grid, x, lw0, d, l0, l1, n_samples, Tn, edge, steps, s);
}

// samples of Tn tokens in NT token tiles: the tokens, and the last sample's 16-row attention window, must fit
// (the q/k/v rows in LDS run 8 rows past the last slot)
constexpr bool tiles_hold(int spw, int Tn, int nt) { return spw * Tn <= 16 * nt && (spw - 1) * Tn + 16 <= 16 * nt + 8; }

constexpr int kSmallBatchMax = 512;    // batches up to this size take the two-sample instance, up to twice this the four-sample one

// Which instance of layers_kernel a call runs: by batch size, or as the call's BESO_PLAN_SPW* hint says where the shape allows
// it (the instances compute the same per-sample arithmetic -- equal bits --, so the hint is a performance / test knob only).
template <int RPW, int KS, int HG, int NTL>
hipError_t launch_layers(float* x, const char* lw0, const FusedDims& d, int l0, int l1, int n_samples, int Tn,
                         const EdgeArgs& edge, const SampleSteps& steps, int precision, int plan, hipStream_t s) {
    constexpr LdsMap L = lds_map(KS);
    constexpr int kSmallSPW = 2, kSmallNT = 2, kMidSPW = 4, kMidNT = 4;
    const int want = plan & BESO_PLAN_SPW_MASK;
    const bool small_ok = kSmallSPW * Tn <= 16 * kSmallNT && (kSmallSPW - 1) * Tn + 16 <= 16 * kSmallNT;
#if !BESO_OPERAND_F16
    if (precision == BESO_PREC_BF16X3) {
        // The split-bf16 instances keep both halves of every activation fragment in LDS, which bounds the token tiles per
        // workgroup at three (145 KiB; four would need 182 KiB).  Every workgroup streams BOTH weight images (40 MB, kitchen):
        // samples per weight byte is what sets this mode's speed, so batches beyond one workgroup of two samples per CU take
        // FOUR samples in three token tiles (4 x 11 = 44 or 4 x 12 = 48 of 48 slots) -- half the workgroups and half the
        // L2 -> CU weight stream of the two-sample instance (82 GB per B = 4096 forward), at 1.5x its MFMAs per workgroup.
        // (The last layer runs on the action-token tiles only, as in every instance: half the eight-sample instance's NTL.
        // Round 3 first shipped this instance untrimmed -- the trimmed copy differed from run to run in whole workgroups --
        // until the cause was found in the half k-steps: mixed_chain_pad above.)
        constexpr int kX3SPW = 4, kX3NT = 3, kX3NTL = (NTL + 1) / 2;
        const bool four_ok = tiles_hold(kX3SPW, Tn, kX3NT);
        if (four_ok && (want == BESO_PLAN_SPW4 || (want != BESO_PLAN_SPW2 && n_samples > kSmallBatchMax) || !small_ok)) {
            constexpr LdsMapX3 X = lds_map_x3(KS, kX3NT);
            static_assert(X.total <= 160 * 1024, "LDS of the four-sample split-bf16 instance");
            return launch_either<RPW, KS, HG, kX3NTL, kX3SPW, kX3NT, 1, 0>(X.total, x, lw0, d, l0, l1, n_samples, Tn, edge, steps, s);
        }
        if (!small_ok) return hipErrorInvalidValue;
        constexpr LdsMapX3 X = lds_map_x3(KS, kSmallNT);
        return launch_either<RPW, KS, HG, NTL, kSmallSPW, kSmallNT, 1, 0>(X.total, x, lw0, d, l0, l1, n_samples, Tn, edge, steps, s);
    }
#endif
    const bool mid_ok = kMidSPW * Tn <= 16 * kMidNT && (kMidSPW - 1) * Tn + 16 <= 16 * kMidNT;
    // latency instances: the samples' tokens and the last sample's 16-row attention window must fit the token tiles
    if (small_ok && (want == BESO_PLAN_SPW2 || (!want && n_samples <= kSmallBatchMax)))
        return launch_either<RPW, KS, HG, NTL, kSmallSPW, kSmallNT, 0, 0>(L.total, x, lw0, d, l0, l1, n_samples, Tn, edge, steps, s);
    // up to one workgroup of four samples per CU: two thirds of the throughput instance's work per workgroup
    if (mid_ok && (want == BESO_PLAN_SPW4 || (!want && n_samples <= 2 * kSmallBatchMax)))
        return launch_either<RPW, KS, HG, NTL, kMidSPW, kMidNT, 0, 0>(L.total, x, lw0, d, l0, l1, n_samples, Tn, edge, steps, s);
    return launch_either<RPW, KS, HG, NTL, kSPW, kNTT, 0, 0>(L.total, x, lw0, d, l0, l1, n_samples, Tn, edge, steps, s);
}

// The long-sequence instance: one sample (Tn <= 16 NT tokens) per workgroup.
template <int RPW, int KS, int NT>
hipError_t launch_layers_long(float* x, const char* lw0, const FusedDims& d, int l0, int l1, int n_samples, int Tn,
                              const EdgeArgs& edge, const SampleSteps& steps, hipStream_t s) {
    constexpr LdsMap L = lds_map(KS, false, NT);
    static_assert(L.total <= 160 * 1024, "LDS of the long-sequence instance");
    return launch_either<RPW, KS, 1, NT, 1, NT, 0, 1>(L.total, x, lw0, d, l0, l1, n_samples, Tn, edge, steps, s);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
static bool x3_shape(const FusedDims& d) {
    // BF16X3 as an instance of layers_kernel: the shipped shapes with the fused attention phase
    return d.attn && !d.seq1 && ((d.RPW == 3 && d.KS == 12 && d.HG == 1) || (d.RPW == 2 && d.KS == 8 && d.HG == 3));
}
static bool x3_long_shape(const FusedDims& d) {
    // ... and as the split-bf16 block kernels (lin_block_x3_kernel + the fp32 attention kernel) for the long-sequence shape
    return d.seq1 && d.lin && d.RPW == 4 && d.KS == 16 && d.D == 16 * kWaves * d.RPW;
}

size_t fused_packed_bytes(const Layout& lay, int precision) {
    FusedDims d;
    if ((precision != BESO_PREC_BF16 && precision != BESO_PREC_BF16X3) || !fused_dims(lay, &d) || !shape_has_kernel(d)) return 0;
    if (precision == BESO_PREC_BF16X3)
        return (x3_shape(d) || x3_long_shape(d)) ? (size_t)d.x3_delta + (size_t)d.layer_bytes * lay.L : 0;
    return d.layer_bytes * lay.L + d.global_bytes;
}


#define FTRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return BESO_ERR_HIP; } while (0)

int fused_pack(const Layout& lay, const float* const* p, char* packed, int precision, hipStream_t s) {
    FusedDims d;
    if ((precision != BESO_PREC_BF16 && precision != BESO_PREC_BF16X3) || !fused_dims(lay, &d) || !shape_has_kernel(d)) return BESO_OK;
    if (precision == BESO_PREC_BF16X3 && !x3_shape(d) && !x3_long_shape(d)) return BESO_ERR_UNSUPPORTED;
    const int D = lay.D;
    const int n_parts = precision == BESO_PREC_BF16X3 ? 2 : 1;
    for (int half = 0; half < n_parts; ++half)               // 0: bf16(w); 1 (BF16X3 only): the low halves
    for (int l = 0; l < lay.L; ++l) {
        // parameter order (include/beso_hip.h): 3 leading tensors, then 16 per block:
        // ln1.w ln1.b ln2.w ln2.b key.w key.b query.w query.b value.w value.b proj.w proj.b fc1.w fc1.b fc2.w fc2.b
        const float* const* q = p + 3 + 16 * l;
        const float *ln1w = q[0], *ln1b = q[1], *ln2w = q[2], *ln2b = q[3];
        const float *kw = q[4], *kb = q[5], *qw = q[6], *qb = q[7], *vw = q[8], *vb = q[9], *pw = q[10], *pb = q[11];
        const float *f1w = q[12], *f1b = q[13], *f2w = q[14], *f2b = q[15];
        char* base = packed + lay.fused + (size_t)l * d.layer_bytes + (half ? (size_t)d.x3_delta : 0);
        const int rt1 = d.NCH * kChunkTiles, rt2 = d.RPW * kWaves;
        (void)hipGetLastError();
        hipLaunchKernelGGL(pack_mfma_a_kernel, dim3(1024), dim3(256), 0, s, f1w, 4 * D, D, ln2w, (uint16_t*)base, rt1,
                           d.KS, kChunkTiles, half);
        hipLaunchKernelGGL(pack_mfma_a_kernel, dim3(1024), dim3(256), 0, s, f2w, D, 4 * D, (const float*)nullptr,
                           (uint16_t*)(base + d.o_w2), rt2, d.KS2p, rt2, half);
        if (d.attn) {
            hipLaunchKernelGGL(pack_qkv_kernel, dim3(1024), dim3(256), 0, s, qw, kw, vw, ln1w,
                               (uint16_t*)(base + d.o_wqkv), D, d.Hv, d.hdv, d.KS, half);
            hipLaunchKernelGGL(pack_proj_kernel, dim3(512), dim3(256), 0, s, pw, (uint16_t*)(base + d.o_wproj), D, d.Hv,
                               d.hdv, rt2, half);
        }
        if (d.lin) {
            const float* ws3[3] = {qw, kw, vw};
            for (int part = 0; part < 3; ++part)      // q / k / v in natural row order (block kernels)
                hipLaunchKernelGGL(pack_mfma_a_kernel, dim3(1024), dim3(256), 0, s, ws3[part], D, D, ln1w,
                                   (uint16_t*)(base + d.o_wqkv_lin + (size_t)part * d.part_bytes), rt2, d.KS, rt2, half);
            hipLaunchKernelGGL(pack_mfma_a_kernel, dim3(1024), dim3(256), 0, s, pw, D, D, (const float*)nullptr,
                               (uint16_t*)(base + d.o_wproj_lin), rt2, d.KS, rt2, half);
        }
        FTRY(hipGetLastError());
        if (half) continue;                      // the low image holds weight fragments only
        hipLaunchKernelGGL(fold_bias_kernel, dim3((rt1 * 16 + 3) / 4), dim3(256), 0, s, f1w, f1b, ln2b,
                           (float*)(base + d.o_b1), 4 * D, D, rt1 * 16);
        FTRY(hipGetLastError());
        FTRY(launch_pack_matrix(f2b, 1, D, base + d.o_b2, 1, rt2 * 16, -1, s));
        if (d.attn) {
            (void)hipGetLastError();
            hipLaunchKernelGGL(fold_qkv_bias_kernel, dim3((d.Hv * 3 * kHDP + 3) / 4), dim3(256), 0, s, qw, kw, vw,
                               qb, kb, vb, ln1b, (float*)(base + d.o_bqkv), D, d.Hv, d.hdv);
            FTRY(hipGetLastError());
            FTRY(launch_pack_matrix(pb, 1, D, base + d.o_bproj, 1, rt2 * 16, -1, s));
        }
        if (d.lin) {
            const float* ws3[3] = {qw, kw, vw};
            const float* bs3[3] = {qb, kb, vb};
            (void)hipGetLastError();
            for (int part = 0; part < 3; ++part) {      // q / k / v
                hipLaunchKernelGGL(fold_bias_kernel, dim3((rt2 * 16 + 3) / 4), dim3(256), 0, s, ws3[part], bs3[part], ln1b,
                                   (float*)(base + d.o_bqkv_lin) + (size_t)part * rt2 * 16, D, D, rt2 * 16);
            }
            FTRY(hipGetLastError());
            FTRY(launch_pack_matrix(pb, 1, D, base + d.o_bproj_lin, 1, rt2 * 16, -1, s));
        }
    }
    if (d.attn) {
        // per-model image: parameter order pos_emb, tok_emb.{w,b}, blocks..., ln_f.{w,b}, sigma_emb.{w,b},
        // action_emb.{w,b}, action_pred.{w,b}
        char* gw = packed + lay.fused + (size_t)lay.L * d.layer_bytes;
        const float* const* tail = p + 3 + 16 * lay.L;
        const float *pos = p[0], *tokw = p[1], *tokb = p[2];
        const float *lnfw = tail[0], *lnfb = tail[1], *sigw = tail[2], *sigb = tail[3], *actw = tail[4], *actb = tail[5];
        (void)hipGetLastError();
        hipLaunchKernelGGL(transpose_pad_kernel, dim3((lay.obs * d.Dp + 255) / 256), dim3(256), 0, s, tokw, D, lay.obs,
                           (float*)(gw + d.g_tokT), d.Dp);
        hipLaunchKernelGGL(transpose_pad_kernel, dim3((lay.act * d.Dp + 255) / 256), dim3(256), 0, s, actw, D, lay.act,
                           (float*)(gw + d.g_actT), d.Dp);
        {
            // split-bf16 A fragments of the two embedding matrices ([D][obs], [D][act]: one k-step each)
            const int rt2 = d.RPW * kWaves;
            for (int half = 0; half < 2; ++half) {
                hipLaunchKernelGGL(pack_mfma_a_kernel, dim3(32), dim3(256), 0, s, tokw, D, lay.obs, (const float*)nullptr,
                                   (uint16_t*)(gw + d.g_tokA + (size_t)half * rt2 * 1024), rt2, 1, rt2, half);
                hipLaunchKernelGGL(pack_mfma_a_kernel, dim3(32), dim3(256), 0, s, actw, D, lay.act, (const float*)nullptr,
                                   (uint16_t*)(gw + d.g_actA + (size_t)half * rt2 * 1024), rt2, 1, rt2, half);
            }
        }
        FTRY(hipGetLastError());
        FTRY(launch_pack_matrix(tokb, 1, D, gw + d.g_tokb, 1, d.Dp, -1, s));
        FTRY(launch_pack_matrix(actb, 1, D, gw + d.g_actb, 1, d.Dp, -1, s));
        FTRY(launch_pack_matrix(sigw, 1, D, gw + d.g_sigw, 1, d.Dp, -1, s));
        FTRY(launch_pack_matrix(sigb, 1, D, gw + d.g_sigb, 1, d.Dp, -1, s));
        FTRY(launch_pack_matrix(pos, lay.seq_size, D, gw + d.g_pos, lay.seq_size, d.Dp, -1, s));
        if (d.head_fused) {
            (void)hipGetLastError();
            hipLaunchKernelGGL(pack_head_kernel, dim3((16 * d.Dp + 255) / 256), dim3(256), 0, s, tail[6], tail[7], lnfw,
                               lnfb, (float*)(gw + d.g_headw), (float*)(gw + d.g_headb), lay.act, D, d.Dp);
            FTRY(hipGetLastError());
        }
    }
    return BESO_OK;
}

// 0: no fused kernel, 1: MLP block / tail block kernels, 2: the whole network in one launch.  The call's plan hint caps it
// (BESO_PLAN_PER_OP: 0, BESO_PLAN_BLOCKS: 1): the per-op kernels are the parity reference of the fused ones in the same
// arithmetic.  (The one-launch kernel wins at every batch size: 0.29 ms flat from B = 1 to 512 against 0.60 ms (B = 1) ..
// 1.6 ms (B = 1024) for ~45 per-op launches, tools/latency.py.)
int fused_level(const Layout& lay, const FwdArgs& a, int precision) {
    FusedDims d;
    if ((precision != BESO_PREC_BF16 && precision != BESO_PREC_BF16X3) || lay.fused == lay.total || !fused_dims(lay, &d) ||
        !shape_has_kernel(d)) return 0;
    // (the one-launch kernel runs both network edges itself: the Linear(D, act) head -- the shipped configs' linear_output: True
    // -- with act <= 16; a model with the MLP head takes the block / per-op kernels)
    const bool whole = x3_shape(d) && kSPW * a.T <= kMT && d.head_fused && d.obs <= 4 * kEmbObsK && d.act <= 4 * kEmbActK;
    // BF16X3: an instance of layers_kernel, or -- the long-sequence shape -- its block-kernel form; nothing else
    if (precision == BESO_PREC_BF16X3) return whole ? 2 : (x3_long_shape(d) ? 1 : 0);
    const int cap = (a.plan & BESO_PLAN_PER_OP) ? 0 : (a.plan & BESO_PLAN_BLOCKS) ? 1 : 2;
    if (cap < 2) return cap;
    // long sequences: the whole network in one launch, a sample per workgroup (a classifier-free pair: two passes of its workgroup)
    const bool whole_long = d.seq1 && a.T <= 16 * kLongNT && d.head_fused && d.obs <= 4 * kEmbObsK && d.act <= 4 * kEmbActK;
    return (whole || whole_long) ? 2 : 1;
}

#if !BESO_OPERAND_F16
bool fused_supported(const Layout& lay, const FwdArgs& a, int precision) { return fused_level(lay, a, precision) > 0; }

int fused_mlp_block(const Layout& lay, const char* packed, int layer, float* x, int M, hipStream_t s) {
    FusedDims d;
    if (!fused_dims(lay, &d)) return BESO_ERR_UNSUPPORTED;
    const char* base = packed + lay.fused + (size_t)layer * d.layer_bytes;
    hipError_t e;
    // (a 4-wave, 512-VGPR instance <6, 12, 4> of the same phase code was measured no faster than <3, 12, 8>)
    if (d.RPW == 3 && d.KS == 12) e = launch_mlp_block<3, 12, 8>(x, base, d, M, s);
    else if (d.RPW == 2 && d.KS == 8) e = launch_mlp_block<2, 8, 8>(x, base, d, M, s);
    else if (d.RPW == 4 && d.KS == 16) e = launch_mlp_block<4, 16, 8>(x, base, d, M, s);
    else return BESO_ERR_UNSUPPORTED;
    return e == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

// LN1 + q/k/v (which = 0: buf = qkv[M][3D] bf16 out) / proj + residual (which = 1: buf = y[M][ld] bf16 in) blocks
// for shapes without the fused attention phase; false if this shape has no such kernels.
bool fused_has_lin_blocks(const Layout& lay, int precision) {
    FusedDims d;
    if ((precision != BESO_PREC_BF16 && precision != BESO_PREC_BF16X3) || lay.fused == lay.total || !fused_dims(lay, &d) ||
        !shape_has_kernel(d)) return false;
    return d.lin && d.RPW == 4 && d.KS == 16 && d.D == 16 * kWaves * d.RPW;
}

// BF16X3, long-sequence shape: [tail of `layer` (y = fp32 attention output, ld_y floats per row)] and / or [LN1 + q/k/v of
// `next_layer` into qkv_next (fp32 [M][3D])] on M token rows as one launch; layer < 0 / next_layer < 0 leaves that half out.
int fused_lin_x3(const Layout& lay, const char* packed, int layer, int next_layer, float* x, const float* y, int ld_y,
                 float* qkv_next, int M, hipStream_t s) {
    FusedDims d;
    if (!fused_dims(lay, &d) || !x3_long_shape(d)) return BESO_ERR_UNSUPPORTED;
    constexpr int NT = 3;
    using L = LdsMapLinX3<NT, 16>;
    static LdsAttr attr;
    if (ensure_lds(lin_block_x3_kernel<4, 16, NT>, L::total, &attr) != hipSuccess) return BESO_ERR_HIP;
    const char* base = packed + lay.fused;
    const char* lw = layer >= 0 ? base + (size_t)layer * d.layer_bytes : nullptr;
    const char* lwn = next_layer >= 0 ? base + (size_t)next_layer * d.layer_bytes : nullptr;
    (void)hipGetLastError();
    hipLaunchKernelGGL((lin_block_x3_kernel<4, 16, NT>), dim3((M + 16 * NT - 1) / (16 * NT)), dim3(512), L::total, s, x, lw, lwn, d, M, y,
                       ld_y, qkv_next);
    return hipGetLastError() == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

int fused_lin_block(const Layout& lay, const char* packed, int layer, int which, float* x, void* buf, int ld, int M,
                    hipStream_t s) {
    FusedDims d;
    if (!fused_dims(lay, &d) || !d.lin) return BESO_ERR_UNSUPPORTED;
    const char* base = packed + lay.fused + (size_t)layer * d.layer_bytes;
    hipError_t e;
    if (d.RPW == 4 && d.KS == 16) e = launch_lin_blocks<4, 16>(which, x, base, d, M, buf, ld, s);
    else return BESO_ERR_UNSUPPORTED;
    return e == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

// proj + residual + LN2 + MLP of `layer`, and -- unless it is the last -- LN1 + q/k/v of layer + 1 (qkv_next [M][3D] bf16),
// as one launch (shapes with the lin blocks).
int fused_lin_tail(const Layout& lay, const char* packed, int layer, float* x, const void* y, int ld_y, void* qkv_next, int M,
                   hipStream_t s) {
    FusedDims d;
    if (!fused_dims(lay, &d) || !d.lin) return BESO_ERR_UNSUPPORTED;
    const char* base = packed + lay.fused + (size_t)layer * d.layer_bytes;
    const char* next = layer + 1 < lay.L ? base + d.layer_bytes : nullptr;
    hipError_t e;
    if (d.RPW == 4 && d.KS == 16) e = launch_tail_block<4, 16>(x, base, next, d, M, y, ld_y, qkv_next, s);
    else return BESO_ERR_UNSUPPORTED;
    return e == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

// ---- training forward through the tail block (train.hip) -----------------------------------------------------------
static bool train_tail_dims(const Layout& lay, FusedDims* d) {
    if (!fused_dims(lay, d) || !shape_has_kernel(*d)) return false;
    return (d->RPW == 3 && d->KS == 12) || (d->RPW == 2 && d->KS == 8);
}

bool fused_train_supported(const Layout& lay) { FusedDims d; return train_tail_dims(lay, &d); }

size_t fused_train_image_bytes(const Layout& lay) {
    FusedDims d;
    if (!train_tail_dims(lay, &d)) return 0;
    return (size_t)train_img(d).layer_bytes * lay.L;
}

// params: the parameter list of beso_pack_weights.  Packs layers [0, L) (fragment-ordered bf16 weights, padded fp32 biases
// and LayerNorm affine parameters) into img.
int fused_train_pack(const Layout& lay, const float* const* p, char* img, hipStream_t s) {
    FusedDims d;
    if (!train_tail_dims(lay, &d)) return BESO_ERR_UNSUPPORTED;
    const TrainImg ti = train_img(d);
    if (lay.L * 13 > kTrainPackSegs) return BESO_ERR_UNSUPPORTED;
    TrainPackTable t;
    t.n = 0;
    int blocks = 0;
    const int D = lay.D, rt1 = d.NCH * kChunkTiles, rt2 = d.RPW * kWaves;
    auto mat = [&](const float* src, uint32_t dst, int rows, int cols, int rt, int kt, int grp) {
        t.seg[t.n] = TrainPackSeg{src, dst, rows, cols, rt, kt, grp, blocks, 0};
        blocks += (rt * kt * 512 + 256 * 16 - 1) / (256 * 16);
        ++t.n;
    };
    auto vec = [&](const float* src, uint32_t dst, int n, int n_pad) {
        t.seg[t.n] = TrainPackSeg{src, dst, n, n_pad, 0, 0, 0, blocks, 0};
        blocks += 1;
        ++t.n;
    };
    for (int l = 0; l < lay.L; ++l) {
        const float* const* q = p + 3 + 16 * l;        // ln1.w ln1.b ln2.w ln2.b key.w key.b query.w query.b value.w value.b proj.w proj.b fc1.w fc1.b fc2.w fc2.b
        const uint32_t base = (uint32_t)l * ti.layer_bytes;
        mat(q[12], base + ti.o_w1, 4 * D, D, rt1, d.KS, kChunkTiles);
        vec(q[13], base + ti.o_b1, 4 * D, rt1 * 16);
        mat(q[14], base + ti.o_w2, D, 4 * D, rt2, d.KS2p, rt2);
        vec(q[15], base + ti.o_b2, D, rt2 * 16);
        const float* w3[3] = {q[6], q[4], q[8]};       // query, key, value: the [q | k | v] row order of the training buffers
        const float* b3[3] = {q[7], q[5], q[9]};
        for (int part = 0; part < 3; ++part) {
            mat(w3[part], base + ti.o_wqkv + (uint32_t)part * ti.part_bytes, D, D, rt2, d.KS, rt2);
            vec(b3[part], base + ti.o_bqkv + (uint32_t)part * rt2 * 16 * (uint32_t)sizeof(float), D, rt2 * 16);
        }
        mat(q[10], base + ti.o_wproj, D, D, rt2, d.KS, rt2);
        vec(q[11], base + ti.o_bproj, D, rt2 * 16);
        vec(q[0], base + ti.o_ln1w, D, rt2 * 16); vec(q[1], base + ti.o_ln1b, D, rt2 * 16);
        vec(q[2], base + ti.o_ln2w, D, rt2 * 16); vec(q[3], base + ti.o_ln2b, D, rt2 * 16);
    }
    t.blocks = blocks;
    (void)hipGetLastError();
    hipLaunchKernelGGL(train_pack_kernel, dim3(blocks), dim3(256), 0, s, t, img);
    return hipGetLastError() == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

// The tail block of `layer` (and LN1 + q/k/v of layer + 1 when next buffers are given) on M token rows.
int fused_train_tail(const Layout& lay, const char* img, int layer, int M, const float* x_in, const void* y, int ld_y,
                     float* x_mid, float* x_out, float* st2, void* xn2, void* h, void* g, float* st1n, void* xn1n, void* qkvn,
                     hipStream_t s) {
    FusedDims d;
    if (!train_tail_dims(lay, &d)) return BESO_ERR_UNSUPPORTED;
    const TrainImg ti = train_img(d);
    const char* lw = img + (size_t)layer * ti.layer_bytes;
    const char* lw_next = qkvn != nullptr ? lw + ti.layer_bytes : nullptr;
    TrainTailArgs a{x_in, (const uint16_t*)y, ld_y, x_mid, x_out, st2, (uint16_t*)xn2, (uint16_t*)h, (uint16_t*)g,
                    st1n, (uint16_t*)xn1n, (uint16_t*)qkvn};
    const dim3 grid((M + kMT - 1) / kMT), block(512);
    hipError_t e;
    (void)hipGetLastError();
    if (d.RPW == 3) {
        constexpr LdsMap L = lds_map(12, true);
        static LdsAttr attr;
        e = ensure_lds(train_tail_kernel<3, 12>, L.total, &attr);
        if (e != hipSuccess) return BESO_ERR_HIP;
        hipLaunchKernelGGL((train_tail_kernel<3, 12>), grid, block, L.total, s, lw, lw_next, d, ti, M, a);
    } else {
        constexpr LdsMap L = lds_map(8, true);
        static LdsAttr attr;
        e = ensure_lds(train_tail_kernel<2, 8>, L.total, &attr);
        if (e != hipSuccess) return BESO_ERR_HIP;
        hipLaunchKernelGGL((train_tail_kernel<2, 8>), grid, block, L.total, s, lw, lw_next, d, ti, M, a);
    }
    return hipGetLastError() == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

// ---- training forward of all layers as one launch (train_fwd_kernel) ---------------------------------------------------
// instance by batch size as for inference: four samples in four token tiles up to 1024 samples (one workgroup per CU at
// BASELINE config 3's per-GPU share), eight in six beyond
static bool train_whole_dims(const Layout& lay, int T, int t, FusedDims* d) {
    if (!train_tail_dims(lay, d) || !d->attn || d->seq1 || d->lin) return false;
    if (!((d->RPW == 3 && d->KS == 12 && d->HG == 1) || (d->RPW == 2 && d->KS == 8 && d->HG == 3))) return false;
    if (lay.L < 1 || lay.L * 16 > kTrainPackSegs) return false;
    const int NTL = d->RPW == 3 ? 2 : 4;                    // the instances' action-token tiles (fused_layers)
    return kSPW * T <= kMT && tiles_hold(kSPW, T, kNTT) && kSPW * t <= 16 * NTL;
}

bool fused_train_whole_supported(const Layout& lay, int T, int t) { FusedDims d; return train_whole_dims(lay, T, t, &d); }

size_t fused_train_whole_image_bytes(const Layout& lay) {
    FusedDims d;
    if (!train_tail_dims(lay, &d) || !d.attn || d.lin) return 0;
    return (size_t)train_img_whole(d).layer_bytes * lay.L;
}

// params: the parameter list of beso_pack_weights.  All layers' fragment images, biases and LayerNorm parameters: one launch.
int fused_train_whole_pack(const Layout& lay, const float* const* p, char* img, hipStream_t s) {
    FusedDims d;
    if (!train_tail_dims(lay, &d) || !d.attn || d.lin || lay.L * 16 > kTrainPackSegs) return BESO_ERR_UNSUPPORTED;
    const TrainImgW ti = train_img_whole(d);
    TrainPackTable t;
    t.n = 0;
    int blocks = 0;
    const int D = lay.D, rt1 = d.NCH * kChunkTiles, rt2 = d.RPW * kWaves;
    auto seg = [&](const float* src, uint32_t dst, int rows, int cols, int rt, int kt, int grp, int tr, size_t elems) {
        t.seg[t.n++] = TrainPackSeg{src, dst, rows, cols, rt, kt, grp, blocks, tr};
        blocks += rt == 0 ? 1 : (int)((elems + 256 * 16 - 1) / (256 * 16));
    };
    for (int l = 0; l < lay.L; ++l) {
        const float* const* q = p + 3 + 16 * l;        // ln1.w ln1.b ln2.w ln2.b key.w key.b query.w query.b value.w value.b proj.w proj.b fc1.w fc1.b fc2.w fc2.b
        const uint32_t base = (uint32_t)l * ti.layer_bytes;
        seg(q[12], base, 4 * D, D, rt1, d.KS, kChunkTiles, 0, (size_t)rt1 * d.KS * 512);
        seg(q[13], base + d.o_b1, 4 * D, rt1 * 16, 0, 0, 0, 0, 0);
        seg(q[14], base + d.o_w2, D, 4 * D, rt2, d.KS2p, rt2, 0, (size_t)rt2 * d.KS2p * 512);
        seg(q[15], base + d.o_b2, D, rt2 * 16, 0, 0, 0, 0, 0);
        const float* w3[3] = {q[6], q[4], q[8]};       // query, key, value = parts 0, 1, 2 of the attention phase
        const float* b3[3] = {q[7], q[5], q[9]};
        for (int part = 0; part < 3; ++part) {
            seg(w3[part], base + d.o_wqkv, d.hdv, D, d.Hv, d.KS, part, 2, (size_t)d.Hv * d.KS * 4 * 512);
            seg(b3[part], base + d.o_bqkv, d.hdv, d.Hv * kHDP, 0, 0, part, 4, 0);
        }
        seg(q[10], base + d.o_wproj, D, d.hdv, rt2, 2 * d.Hv, 0, 3, (size_t)rt2 * 2 * d.Hv * 512);
        seg(q[11], base + d.o_bproj, D, rt2 * 16, 0, 0, 0, 0, 0);
        seg(q[0], base + ti.o_ln1w, D, rt2 * 16, 0, 0, 0, 0, 0); seg(q[1], base + ti.o_ln1b, D, rt2 * 16, 0, 0, 0, 0, 0);
        seg(q[2], base + ti.o_ln2w, D, rt2 * 16, 0, 0, 0, 0, 0); seg(q[3], base + ti.o_ln2b, D, rt2 * 16, 0, 0, 0, 0, 0);
    }
    t.blocks = blocks;
    (void)hipGetLastError();
    hipLaunchKernelGGL(train_pack_kernel, dim3(blocks), dim3(256), 0, s, t, img);
    return hipGetLastError() == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

template <int RPW, int KS, int HG, int NTL, int SPW, int NTA, int RD>
static hipError_t launch_train_fwd_rd(const char* img, const FusedDims& d, const TrainImgW& ti, int batch, int T,
                                      const TrainWholeBufs& a, hipStream_t s) {
    constexpr LdsMap L = lds_map(KS);
    static LdsAttr attr;
    hipError_t e = ensure_lds(train_fwd_kernel<RPW, KS, HG, NTL, SPW, NTA, RD>, L.total, &attr);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL((train_fwd_kernel<RPW, KS, HG, NTL, SPW, NTA, RD>), dim3((batch + SPW - 1) / SPW), dim3(512), L.total, s, img, d,
                       ti, batch, T, a);
    return hipGetLastError();
}
template <int RPW, int KS, int HG, int NTL, int SPW, int NTA>
static hipError_t launch_train_fwd(const char* img, const FusedDims& d, const TrainImgW& ti, int batch, int T,
                                   const TrainWholeBufs& a, hipStream_t s) {
    return a.p_resid > 0.f ? launch_train_fwd_rd<RPW, KS, HG, NTL, SPW, NTA, 1>(img, d, ti, batch, T, a, s)
                           : launch_train_fwd_rd<RPW, KS, HG, NTL, SPW, NTA, 0>(img, d, ti, batch, T, a, s);
}

int fused_train_whole(const Layout& lay, const char* img, int batch, int T, const TrainWholeBufs& a, hipStream_t s) {
    FusedDims d;
    if (!train_whole_dims(lay, T, a.t, &d)) return BESO_ERR_UNSUPPORTED;
    const TrainImgW ti = train_img_whole(d);
    constexpr int kMidSPW = 4, kMidNT = 4;
    const bool mid = batch <= 2 * kSmallBatchMax && tiles_hold(kMidSPW, T, kMidNT) && (kMidSPW - 1) * T + 16 <= 16 * kMidNT;
    // Round 6: four samples of <= 12 tokens in THREE token tiles (kitchen: 44 of 48 slots instead of 44 of 64 -- the VERDICT's
    // "318.7 GFLOP issued for 211 algorithmic" at 1024 samples; the last sample's 16-row attention window ends in the pad rows
    // behind slot 48, as in the split-bf16 inference instance)
    const bool mid3 = mid && tiles_hold(kMidSPW, T, 3) && kMidSPW * a.t <= 16 * 2 ;
    hipError_t e;
    if (d.RPW == 3)
        e = mid3 ? launch_train_fwd<3, 12, 1, 2, kMidSPW, 3>(img, d, ti, batch, T, a, s)
            : mid ? launch_train_fwd<3, 12, 1, 2, kMidSPW, kMidNT>(img, d, ti, batch, T, a, s)
                : launch_train_fwd<3, 12, 1, 2, kSPW, kNTT>(img, d, ti, batch, T, a, s);
    else
        e = mid3 ? launch_train_fwd<2, 8, 3, 4, kMidSPW, 3>(img, d, ti, batch, T, a, s)
            : mid ? launch_train_fwd<2, 8, 3, 4, kMidSPW, kMidNT>(img, d, ti, batch, T, a, s)
                : launch_train_fwd<2, 8, 3, 4, kSPW, kNTT>(img, d, ti, batch, T, a, s);
    return e == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

// ---- training backward: data-gradient GEMMs in the transposed formulation (train_dgrad_kernel) ------------------------
// per-layer image of the TRANSPOSED weights in A-fragment order: [q^T | k^T | v^T | proj^T] (D x D each: kt_d k-steps),
// W1^T (D x 4 D: kt_h k-steps), W2^T (4 D x D in chunks of 8 RPW row tiles)
struct TrainBwdImgW { uint32_t o_qkvT, o_pT, o_w1T, o_w2T, dd_bytes, layer_bytes; int kt_d, kt_h, n_chunks; };
static bool train_dgrad_dims(const Layout& lay, FusedDims* d, TrainBwdImgW* t) {
    if (!train_tail_dims(lay, d)) return false;
    const int RT = d->RPW * kWaves;
    if (lay.D > 16 * RT || lay.D % 8 != 0) return false;
    const int pfa = d->RPW == 3 ? 6 : 4;                         // k-steps of weight fragments in flight per wave (train_dgrad_kernel)
    t->kt_d = d->KS;                                             // 12 / 8: a multiple of pfa
    t->kt_h = ((4 * lay.D + 31) / 32 + pfa - 1) / pfa * pfa;
    t->n_chunks = (4 * lay.D / 16 + RT - 1) / RT;
    t->dd_bytes = (uint32_t)RT * t->kt_d * 1024;
    uint32_t cur = 0;
    auto carve = [&](uint32_t bytes) { uint32_t o = cur; cur = (uint32_t)round_up_sz((size_t)cur + bytes, 256); return o; };
    t->o_qkvT = carve(3 * t->dd_bytes);
    t->o_pT = carve(t->dd_bytes);
    t->o_w1T = carve((uint32_t)RT * t->kt_h * 1024);
    t->o_w2T = carve((uint32_t)t->n_chunks * RT * t->kt_d * 1024);
    t->layer_bytes = cur;
    // the staged gradient tile: 3 token tiles x k-steps KiB of LDS
    constexpr int NT = 3;
    const int lds_kib = NT * (3 * t->kt_d > t->kt_h ? 3 * t->kt_d : t->kt_h);
    return lds_kib * 1024 <= 150 * 1024 && lay.L * 6 <= kTrainPackSegs;
}
bool fused_train_dgrad_supported(const Layout& lay) { FusedDims d; TrainBwdImgW t; return train_dgrad_dims(lay, &d, &t); }
size_t fused_train_dgrad_image_bytes(const Layout& lay) {
    FusedDims d; TrainBwdImgW t;
    return train_dgrad_dims(lay, &d, &t) ? (size_t)t.layer_bytes * lay.L : 0;
}
int fused_train_dgrad_pack(const Layout& lay, const float* const* p, char* img, hipStream_t s) {
    FusedDims d; TrainBwdImgW bi;
    if (!train_dgrad_dims(lay, &d, &bi)) return BESO_ERR_UNSUPPORTED;
    TrainPackTable t;
    t.n = 0;
    int blocks = 0;
    const int D = lay.D, RT = d.RPW * kWaves;
    auto matT = [&](const float* src, uint32_t dst, int rows, int cols, int rt, int kt) {      // A[r][c] = src[c][r], src is [cols][rows]
        t.seg[t.n++] = TrainPackSeg{src, dst, rows, cols, rt, kt, RT, blocks, 1};
        blocks += (int)(((size_t)rt * kt * 512 + 256 * 16 - 1) / (256 * 16));
    };
    for (int l = 0; l < lay.L; ++l) {
        const float* const* q = p + 3 + 16 * l;        // ln1.w ln1.b ln2.w ln2.b key.w key.b query.w query.b value.w value.b proj.w proj.b fc1.w fc1.b fc2.w fc2.b
        const uint32_t base = (uint32_t)l * bi.layer_bytes;
        const float* w3[3] = {q[6], q[4], q[8]};       // query, key, value: the [q | k | v] column order of dqkv
        for (int part = 0; part < 3; ++part) matT(w3[part], base + bi.o_qkvT + (uint32_t)part * bi.dd_bytes, D, D, RT, bi.kt_d);
        matT(q[10], base + bi.o_pT, D, D, RT, bi.kt_d);
        matT(q[12], base + bi.o_w1T, D, 4 * D, RT, bi.kt_h);                      // fc1.weight [4D][D]: A[n in D][k in 4D]
        matT(q[14], base + bi.o_w2T, 4 * D, D, bi.n_chunks * RT, bi.kt_d);        // fc2.weight [D][4D]: A[n in 4D][k in D]
    }
    t.blocks = blocks;
    (void)hipGetLastError();
    hipLaunchKernelGGL(train_pack_kernel, dim3(blocks), dim3(256), 0, s, t, img);
    return hipGetLastError() == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

// which: 0 q|k|v data gradient (in = dqkv [M][3D] -> out32 [M][D]); 1 out-projection (in [M][D] -> out16 [M][D]);
//        2 FC1 (in = dh [M][4D] -> out32 [M][D]); 3 FC2 + GELU' (in = dyo [M][D], h [M][4D] -> dh [M][4D], colsum [4D])
// rows of a bias slab of fused_train_dgrad(which = 3) over M token rows; fused_train_bias_reduce adds n slabs up into n vectors
int fused_train_dgrad_blocks(int M) { return (M + 16 * 3 - 1) / (16 * 3); }
int fused_train_bias_reduce(const float* const* slabs, float* const* outs, const int* blocks, int n, int N, hipStream_t s) {
    if (n < 1) return BESO_OK;
    (void)hipGetLastError();
    // (slabs of different heights -- the compact last layer -- get launches of their own)
    int i = 0;
    while (i < n) {
        SlabRed t;
        t.n = 0;
        const int nb = blocks[i];
        while (i < n && blocks[i] == nb && t.n < kMaxLayers) { t.slab[t.n] = slabs[i]; t.out[t.n] = outs[i]; ++t.n; ++i; }
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((N + 15) / 16, t.n), dim3(256), 0, s, t, nb, N);
    }
    return hipGetLastError() == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

int fused_train_dgrad(const Layout& lay, const char* img, int layer, int which, int M, const void* in, float* out32, void* out16,
                      const void* h, void* dh, float* colsum, hipStream_t s, const TrainLnBwd* ln) {
    FusedDims d; TrainBwdImgW bi;
    if (!train_dgrad_dims(lay, &d, &bi) || which < 0 || which > 3) return BESO_ERR_UNSUPPORTED;
    constexpr int NT = 3;
    const int D = lay.D;
    const char* lw = img + (size_t)layer * bi.layer_bytes;
    DgradArgs a{};
    a.in = (const uint16_t*)in; a.M = M; a.out32 = out32; a.out16 = (uint16_t*)out16; a.ld_out = D; a.N = D; a.n_chunks = 1;
    a.parts = 1; a.h = nullptr; a.dh = nullptr; a.colsum = nullptr;
    const char* wimg;
    if (which == 0) { wimg = lw + bi.o_qkvT; a.ld_in = 3 * D; a.K = D; a.parts = 3; a.kt = bi.kt_d; }      // (the three images lie dd_bytes = RT kt_d KiB apart: back to back)
    else if (which == 1) { wimg = lw + bi.o_pT; a.ld_in = D; a.K = D; a.kt = bi.kt_d; }
    else if (which == 2) { wimg = lw + bi.o_w1T; a.ld_in = 4 * D; a.K = 4 * D; a.kt = bi.kt_h; }
    else { wimg = lw + bi.o_w2T; a.ld_in = D; a.K = D; a.kt = bi.kt_d; a.N = 4 * D; a.n_chunks = bi.n_chunks; a.ld_out = 4 * D;
           a.h = (const uint16_t*)h; a.dh = (uint16_t*)dh; a.colsum = colsum; }
    LnBwdEpi ep{};
    if (ln != nullptr) {
        if (which != 0 && which != 2) return BESO_ERR_BAD_ARG;
        ep = LnBwdEpi{ln->x, ln->stats, ln->gamma, ln->dres_in, ln->dres_out, (uint16_t*)ln->dxb, ln->part,
                      ln->p, ln->p > 0.f ? 1.0f / (1.0f - ln->p) : 1.f, ln->seed, ln->site, ln->skip_mod, ln->x_bf16};
        a.out32 = nullptr;
    }
    size_t lds_bytes = (size_t)NT * a.parts * a.kt * 1024;
    if (lds_bytes < (size_t)16 * NT * kRedTok * sizeof(float)) lds_bytes = (size_t)16 * NT * kRedTok * sizeof(float);
    const dim3 grid((M + 16 * NT - 1) / (16 * NT)), block(512);
    hipError_t e;
    (void)hipGetLastError();
    const int mode = ln == nullptr ? 0 : (ln->x_bf16 ? 2 : 1);
#define BESO_DGRAD_LAUNCH(R, P, X)                                                                                       \
    do {                                                                                                                 \
        static LdsAttr attr;                                                                                             \
        e = ensure_lds(train_dgrad_kernel<R, NT, P, X>, 150 * 1024, &attr);                                              \
        if (e != hipSuccess) return BESO_ERR_HIP;                                                                        \
        hipLaunchKernelGGL((train_dgrad_kernel<R, NT, P, X>), grid, block, lds_bytes, s, wimg, a, ep, g_stamps, g_stamps_cap); \
    } while (0)
    if (d.RPW == 3) { if (mode == 2) BESO_DGRAD_LAUNCH(3, 6, 2); else if (mode == 1) BESO_DGRAD_LAUNCH(3, 6, 1); else BESO_DGRAD_LAUNCH(3, 6, 0); }
    else { if (mode == 2) BESO_DGRAD_LAUNCH(2, 4, 2); else if (mode == 1) BESO_DGRAD_LAUNCH(2, 4, 1); else BESO_DGRAD_LAUNCH(2, 4, 0); }
#undef BESO_DGRAD_LAUNCH
    return hipGetLastError() == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

// the second half of a layer's backward (train_mlp_bwd_kernel): the launches which = 3, 2 (with ln), 1 of fused_train_dgrad in one
bool fused_train_mlp_bwd_supported(const Layout& lay) {
    FusedDims d; TrainBwdImgW bi;
    if (!train_dgrad_dims(lay, &d, &bi)) return false;
    const int KC = d.RPW * kWaves / 2;
    return bi.kt_d == KC && bi.kt_h <= bi.n_chunks * KC && 3 * 3 * KC * 1024 <= 150 * 1024;
}
int fused_train_mlp_bwd(const Layout& lay, const char* img, int layer, int M, const void* dyo, const void* h, void* dh, float* colsum,
                        void* dy, const TrainLnBwd& ln, hipStream_t s) {
    FusedDims d; TrainBwdImgW bi;
    if (!fused_train_mlp_bwd_supported(lay) || !train_dgrad_dims(lay, &d, &bi)) return BESO_ERR_UNSUPPORTED;
    constexpr int NT = 3;
    const char* lw = img + (size_t)layer * bi.layer_bytes;
    const MlpBwdArgs a{(const uint16_t*)dyo, (const uint16_t*)h, (uint16_t*)dh, colsum, (uint16_t*)dy, bi.o_w2T, bi.o_w1T, bi.o_pT,
                       lay.D, bi.kt_d, bi.kt_h, bi.n_chunks, M};
    const LnBwdEpi ep{ln.x, ln.stats, ln.gamma, ln.dres_in, ln.dres_out, (uint16_t*)ln.dxb, ln.part,
                       ln.p, ln.p > 0.f ? 1.0f / (1.0f - ln.p) : 1.f, ln.seed, ln.site, ln.skip_mod, ln.x_bf16};
    const int KC = d.RPW * kWaves / 2;
    const size_t lds_bytes = (size_t)NT * (bi.kt_d + 2 * KC) * 1024;
    const dim3 grid((M + 16 * NT - 1) / (16 * NT)), block(512);
    hipError_t e;
    (void)hipGetLastError();
    if (d.RPW == 3) {
        static LdsAttr attr;
        e = ensure_lds(train_mlp_bwd_kernel<3, NT, 6>, 150 * 1024, &attr);
        if (e != hipSuccess) return BESO_ERR_HIP;
        hipLaunchKernelGGL((train_mlp_bwd_kernel<3, NT, 6>), grid, block, lds_bytes, s, lw, a, ep, g_stamps, g_stamps_cap);
    } else {
        static LdsAttr attr;
        e = ensure_lds(train_mlp_bwd_kernel<2, NT, 4>, 150 * 1024, &attr);
        if (e != hipSuccess) return BESO_ERR_HIP;
        hipLaunchKernelGGL((train_mlp_bwd_kernel<2, NT, 4>), grid, block, lds_bytes, s, lw, a, ep, g_stamps, g_stamps_cap);
    }
    return hipGetLastError() == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

#endif   // !BESO_OPERAND_F16

// Whole network (embed -> all layers -> head) or layers only.  Returns in *fused_edges whether the token
// embedding / action head ran inside the kernel (bit 0 / bit 1).
// which network edges fused_layers runs inside the kernel (bit 0: embedding, bit 1: head); the caller runs the others
int fused_layer_edges(const Layout& lay) {
    FusedDims d;
    if (!fused_dims(lay, &d) || !d.attn) return 0;
    return 1 | (d.head_fused ? 2 : 0);
}

// The sampler loop needs both network edges inside the kernel and one action-window element per thread of a workgroup.
bool fused_can_loop(const Layout& lay, const FwdArgs& a, int precision) {
    FusedDims d;
    if (fused_level(lay, a, precision) != 2 || !fused_dims(lay, &d) || !d.attn || !d.head_fused) return false;
    return (d.seq1 ? 1 : kSPW) * a.t * lay.act <= kLoopMaxElems;
}

int fused_layers(const Layout& lay, const char* packed, const FwdArgs& a, float* x, int* fused_edges, int precision,
                 hipStream_t s, const SampleSteps* steps) {
    FusedDims d;
    if (!fused_dims(lay, &d) || !d.attn) return BESO_ERR_UNSUPPORTED;
    static const SampleSteps no_steps{};
    const SampleSteps& S = steps ? *steps : no_steps;
    const char* base = packed + lay.fused;
    EdgeArgs e;
    e.state = a.state; e.action = a.action; e.goal = a.goal; e.sigma = a.sigma; e.out = a.out;
    e.aux = a.aux;
    e.noise = a.noise;
    e.B = a.batch; e.t = a.t; e.precondition = a.precondition;
    e.two = a.vbatch > a.batch ? 1 : 0;
    e.uncond_all = (!e.two && a.uncond_from == 0) ? 1 : 0;
    e.cond_lambda = a.cond_lambda; e.sigma_data = a.sigma_data;
    // with a classifier-free pair the virtual samples are interleaved (2b, 2b+1) so that both halves of a
    // pair live in one workgroup; that ordering only exists inside the kernel, so the head must be fused too
    if (!d.head_fused) return BESO_ERR_UNSUPPORTED;
    // every instance stages its workgroup's action windows in the kXsBytes of `xs`
    if ((size_t)(d.seq1 ? 1 : kSPW) * a.t * lay.act * sizeof(float) > (size_t)kXsBytes) return BESO_ERR_UNSUPPORTED;
    if (fused_edges) *fused_edges = 3;
    hipError_t err;
    if (d.RPW == 3 && d.KS == 12 && d.HG == 1) err = launch_layers<3, 12, 1, 2>(x, base, d, 0, lay.L, a.vbatch, a.T, e, S, precision, a.plan, s);    // kitchen: 8 x 4 action tokens
    else if (d.RPW == 2 && d.KS == 8 && d.HG == 3) err = launch_layers<2, 8, 3, 4>(x, base, d, 0, lay.L, a.vbatch, a.T, e, S, precision, a.plan, s);   // block-push: 8 x 5
    else if (d.seq1 && precision == BESO_PREC_BF16) err = launch_layers_long<4, 16, kLongNT>(x, base, d, 0, lay.L, a.batch, a.T, e, S, s);   // long horizon: 1 x 67 tokens
                                                                                                             // (one workgroup per REAL sample: pairs run as two passes)
    else return BESO_ERR_UNSUPPORTED;
    return err == hipSuccess ? BESO_OK : BESO_ERR_HIP;
}

#if !BESO_OPERAND_F16 && BESO_DEV_API
void fused_set_stamps(void* buf, int cap) {
    g_stamps = (unsigned long long*)buf;
    g_stamps_cap = cap;
}
#endif

}  // namespace beso
