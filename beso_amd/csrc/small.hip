// Small batches CHIP-WIDE (round 5): the score-net forward as a chain of short launches whose grids spread every weight
// matrix over the CUs, instead of one workgroup streaming all 20 MB of weights alone.
//   reference: score_gpts.py:272-358 (DiffusionGPT.forward), :50-80 (attention), :105-115 (Block), score_wrappers.py:81-96
//
// Why: the rollout workload of the reference is B = 1 (kitchen_workspace_manager.py:286-294).  The one-launch kernel's
// latency instance puts a sample group on ONE CU, which then reads the whole packed weight image by itself: 0.25 ms per
// forward flat from B = 1 to 512 -- ~80 GB/s, 1 % of what the chip can stream, 224+ of 256 CUs idle.  Samples are few here,
// weights are many: so the WEIGHTS are what gets partitioned.  A dependent launch boundary costs ~1.6 us on this part
// (tools/microbench/launch_chain), less than an in-kernel cross-workgroup exchange (3.75 us measured), and needs no spin-wait protocol.
//
// Per layer FOUR launches (bf16 or exact-fp32 MFMA, operands straight from the GENERIC section of the packed image, torch
// layout [out][in] = k contiguous: an MFMA operand fragment is one 16-byte load per lane, no transposition anywhere):
//   sb_qkv_attn     LN1 -> q | k | v of ONE head -> that head's causal attention -> y     tile = (head, whole samples of <= 16 tokens)
//   sb_gemm_resid   x += y Wp^T + b                                             tile rows x 16 features, the 4 waves split K
//   sb_ln_gemm      LN2 -> FC1 -> exact GELU -> h                               tile rows x 64 features, 4 waves x 16 features
//   sb_gemm_resid   x += h W2^T + b
// (windows of more than 16 tokens or heads wider than 64: sb_ln_gemm for q|k|v and the per-op attention kernel instead of the
// first).  A tile is 16 token rows -- one MFMA row tile, LayerNorm with 16 threads per row -- while all of a launch's
// workgroups find a slot at once (three per CU), 32 rows beyond.  Every wave requests ALL the weight fragments of its tile at
// once, the activations of a tile come from the fp32 residual through LayerNorm into LDS (or as fragments straight from
// memory: sb_gemm_resid).  What a launch costs is memory round trips in a row and the instructions its waves issue at one
// wave per SIMD (4.6 ... 10 us each, whatever the arithmetic): DESIGN.md section 4.4 has the measurements, including the
// layers as ONE kernel with hand-overs through memory (slower: 3.75 us per hand-over) and the instruction diet that
// followed.  Weight bytes per workgroup: 46 KB (138 KB: the head-split tile; bf16, kitchen).  The embedding and the head
// are the per-op kernels.
// Round 6 (bf16): the head-split launch runs TWELVE waves -- a (part, 16-dim tile) per wave: a third of the instructions per
// wave -- and, up to 96 token rows, carries the out-projection as its epilogue: per-head partial products into H slabs that
// the FC1 launch's LayerNorm prologue adds to the residual (fixed order) -- THREE launches per layer.  The wide tiles tried for
// 41 ... 500 samples (32 x 128 FC1, 32 x 32 residual GEMMs) were measured and removed: DESIGN.md section 4.4.
// Arithmetic: bf16 mode = the per-op bf16 kernels' (bf16 operands, fp32 accumulate, fp32 LayerNorm / softmax, the fitted
// GELU); fp32 mode = exact-fp32 MFMA, two-pass LayerNorm, erff -- the per-op fp32 path's results to rounding order.
#include <algorithm>
#include <atomic>
#include "common.h"
#include "fused.h"

namespace beso {
namespace {

template <typename E> struct SbE;
template <> struct SbE<uint16_t> {
    static constexpr int KPL = 8, KSTEP = 32;          // contraction indices per 16-byte lane chunk / per fragment
    // D[feature][token] += W-fragment x activation-fragment^T: the lane ends up with 4 consecutive features of one token
    static __device__ __forceinline__ void mma(f32x4& acc, const u32x4& w, const u32x4& a) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float gelu(float v) { return gelu_poly(v); }
};
template <> struct SbE<float> {
    static constexpr int KPL = 4, KSTEP = 16;
    static __device__ __forceinline__ void mma(f32x4& acc, const u32x4& w, const u32x4& a) {
#pragma unroll
        for (int j = 0; j < 4; ++j)         // element j of every lane group: k = 4 g + j (the same assignment in both operands)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[j]), __uint_as_float(a[j]), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
};

template <typename E> __device__ __forceinline__ void store4(E* p, const f32x4& v);
template <> __device__ __forceinline__ void store4<float>(float* p, const f32x4& v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<uint16_t>(uint16_t* p, const f32x4& v) {
    uint2 u;
    u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *(uint2*)p = u;
}

constexpr int kSbRows = 32;            // token rows of a tile (two MFMA tiles)

// LayerNorm of rows [m0, m0 + rows) (eps 1e-5, biased variance, two passes over the registers: score_gpts.py:96-97), the affine
// output as operand type E into the LDS tile `at` (32 rows of pitch Kd sizeof(E) + 16); columns D .. Kd and rows past the end
// are zeros.  Eight threads per row -- or SIXTEEN in the instances with tiles of 16 token rows (few samples: with 32-row tiles half the
// threads would normalise rows of zeros, and these launches are bound by the instructions a wave issues, not by what they
// compute: ~450 of a launch's ~1,000 were this LayerNorm); rows 16 .. 31 of the tile are then neither written nor read.  Two steps: load()
// requests the rows and the affine parameters -- the caller issues it BEFORE its weight-fragment loads (a wave's loads return
// in order, and the LayerNorm is what runs first) -- finish() reduces, normalises and writes the tile.
template <typename E, int KD64, int TPR>          // TPR = threads per row: 8 (tiles of up to 32 rows) or 16 (up to 16 rows)
struct LnTile {
    static constexpr int Kd = 64 * KD64, PITCH = Kd * (int)sizeof(E) + 16, NC = Kd / (4 * TPR);
    f32x4 v[NC], g4[NC], b4[NC];
    __device__ __forceinline__ void load(const float* __restrict__ x, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, int m0, int rows, int D, int tid) {
        const int r = tid / TPR, q = tid % TPR;
        const bool rv = r < rows;
        const float* xr = x + (size_t)(m0 + (rv ? r : 0)) * D;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = 4 * q + 4 * TPR * j;
            const bool ok = c < D;
            v[j] = (rv && ok) ? *(const f32x4*)(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            g4[j] = ok ? *(const f32x4*)(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            b4[j] = ok ? *(const f32x4*)(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // Round 6 (few token rows): the residual rows PLUS the out-projection that the head-split attention launch left as one
    // partial product per head -- x_mid = (x + b_p) + slab_0 + ... + slab_{H-1}, added in this fixed order (deterministic), six
    // slabs in flight at a time.  xm != nullptr: this workgroup also writes x_mid back (one column tile per row tile does).
    __device__ __forceinline__ void load_sum(const float* __restrict__ x, const float* __restrict__ slab, size_t slab_stride, int H,
                                             const float* __restrict__ pbias, float* __restrict__ xm,
                                             const float* __restrict__ gamma, const float* __restrict__ beta, int m0, int rows,
                                             int D, int tid) {
        const int r = tid / TPR, q = tid % TPR;
        const bool rv = r < rows;
        const size_t ro = (size_t)(m0 + (rv ? r : 0)) * D;
        f32x4 pb[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = 4 * q + 4 * TPR * j;
            const bool ok = c < D;
            v[j] = (rv && ok) ? *(const f32x4*)(x + ro + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            pb[j] = ok ? *(const f32x4*)(pbias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            g4[j] = ok ? *(const f32x4*)(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            b4[j] = ok ? *(const f32x4*)(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) v[j] += pb[j];
        // (six slabs in flight at a time: every batch is one more memory round trip in front of the LayerNorm -- with three, the
        //  kitchen shape's six heads cost this launch what the saved out-projection launch had cost; the instances that take
        //  this path are a few workgroups on an empty chip, registers are free)
        constexpr int SB = 6;
        for (int h0 = 0; h0 < H; h0 += SB) {
            f32x4 sv[SB][NC];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const float* sp = slab + (size_t)min(h0 + u, H - 1) * slab_stride + ro;     // (a clamped slab is loaded and not added)
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const int c = 4 * q + 4 * TPR * j;
                    sv[u][j] = (rv && c < D) ? *(const f32x4*)(sp + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                if (h0 + u < H) {
#pragma unroll
                    for (int j = 0; j < NC; ++j) v[j] += sv[u][j];
                }
            }
        }
        if (xm != nullptr && rv) {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int c = 4 * q + 4 * TPR * j;
                if (c < D) *(f32x4*)(xm + ro + c) = v[j];
            }
        }
    }
    __device__ __forceinline__ void finish(unsigned char* at, int rows, int D, int tid) {
        const int r = tid / TPR, q = tid % TPR;
        const bool rv = r < rows;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
#pragma unroll
        for (int m = 1; m < TPR; m <<= 1) sum += __shfl_xor(sum, m, 64);
        const float mean = sum / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            if (4 * q + 4 * TPR * j < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dlt = v[j][e] - mean; sq = fmaf(dlt, dlt, sq); }
            }
        }
#pragma unroll
        for (int m = 1; m < TPR; m <<= 1) sq += __shfl_xor(sq, m, 64);
        const float rstd = 1.0f / sqrtf(sq / (float)D + 1e-5f);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = 4 * q + 4 * TPR * j;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (rv && c < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * g4[j][e] + b4[j][e];
            }
            store4<E>((E*)(at + r * PITCH) + c, o);
        }
    }
};

// The out-projection as per-head partial products (round 6, few token rows): slab[h][m][:] = y_h[m][:] Wp[:, h hd ..]^T written by
// the head-split attention launch, added up -- with the projection's bias and the residual -- by the LayerNorm prologue of
// the launch behind it, which also leaves the sum in xm.  slab == nullptr: off.
struct SbSlabs { const float* slab; size_t stride; int H; const float* pbias; float* xm; };

// out[m][n] = epi( LayerNorm(x[m][:]) . W[n][:] + bias[n] )     EPI 0: store, 1: exact GELU, store
// grid (Np / 64, ceil(M / 32)), 256 threads: wave w owns features [64 bx + 16 w, +16) of the tile's 32 rows.
template <typename E, int KD64, int EPI, int TPR, bool SUM = false>      // SUM: the slab-sum prologue (an instance of its own: its
                                                                            // registers cost the 32-row instances a workgroup per CU)
__global__ __launch_bounds__(256) void sb_ln_gemm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const E* __restrict__ W,
                                                         const float* __restrict__ bias, E* __restrict__ out, int M, int D,
                                                         int ld_out, int n_store, SbSlabs sl) {
    constexpr int Kd = 64 * KD64, KPL = SbE<E>::KPL, KSTEP = SbE<E>::KSTEP, NK = Kd / KSTEP;
    constexpr int PITCH = Kd * (int)sizeof(E) + 16;        // +16 B: the 16 rows of a fragment read land on different banks
    __shared__ __attribute__((aligned(16))) unsigned char at[kSbRows * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int ROWS = TPR == 16 ? 16 : kSbRows;         // token rows of this instance's tile
    const int m0 = blockIdx.y * ROWS, rows = min(ROWS, M - m0);
    const int n0 = (blockIdx.x * 4 + wave) * 16;
    const int li = lane & 15, lg = lane >> 4;
    // 1. the tile's rows, then every weight fragment of this wave's 16 features: all requested at once
    LnTile<E, KD64, TPR> ln;
    if constexpr (SUM) ln.load_sum(x, sl.slab, sl.stride, sl.H, sl.pbias, blockIdx.x == 0 ? sl.xm : nullptr, gamma, beta, m0, rows, D, tid);
    else ln.load(x, gamma, beta, m0, rows, D, tid);
    u32x4 wf[NK];
    {
        const E* wp = W + (size_t)(n0 + li) * Kd + KPL * lg;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) wf[ks] = *(const u32x4*)(wp + ks * KSTEP);
    }
    const f32x4 bv = *(const f32x4*)(bias + n0 + 4 * lg);
    // 2. LayerNorm of the tile's rows as operand type E into LDS
    ln.finish(at, rows, D, tid);
    __syncthreads();
    // 3. the tile's product
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const bool two = rows > 16;                            // (one sample of <= 16 tokens: the second row tile is idle)
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const u32x4 a0 = *(const u32x4*)(at + li * PITCH + (ks * KSTEP + KPL * lg) * (int)sizeof(E));
        SbE<E>::mma(acc[0], wf[ks], a0);
        if (two) {
            const u32x4 a1 = *(const u32x4*)(at + (16 + li) * PITCH + (ks * KSTEP + KPL * lg) * (int)sizeof(E));
            SbE<E>::mma(acc[1], wf[ks], a1);
        }
    }
    // 4. the lane holds features n0 + 4 lg .. +3 of token 16 rt + li
    const int n = n0 + 4 * lg;
    if (n < n_store) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int tok = 16 * rt + li;
            if (tok < rows) {
                f32x4 v = acc[rt] + bv;
                if (EPI == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = SbE<E>::gelu(v[e]);
                }
                store4<E>(out + (size_t)(m0 + tok) * ld_out + n, v);
            }
        }
    }
}

// LN1 -> q | k | v of ONE head -> causal attention of that head, for a block of whole samples (spb T <= 32 token rows):
//   y[m][h hd + d] = sum_j softmax_j( q_i . k_j / sqrt(hd), j <= i ) v_j[d]        (score_gpts.py:58-76)
// grid (H, ceil(vbatch / spb)), 256 threads.  The attention needs q, k AND v of its head and nothing else, so the q|k|v weight
// rows are split by HEAD here (3 x hd rows per workgroup: 138 KB in bf16 at the kitchen shape) and a launch boundary between
// the projections and the attention disappears -- a dependent launch is ~5 us on this part whatever it computes
// (tools/microbench/launch_chain).  Wave w owns dims [16 w, 16 w + 16) of the head in all three parts; q, k, v go to LDS as fp32;
// then 16 lanes per (sample, query) item: each lane 4 dims, the score's partial dot products summed over the item's DPP row,
// softmax and the weighted sum of v in registers (fp32, expf: the reference's operation order).
constexpr int kSbHP = 68;              // floats per q / k / v row in LDS (64 dims + 4: rows land on different banks)
template <typename E, int KD64, int TPR>
__global__ __launch_bounds__(256) void sb_qkv_attn_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const E* __restrict__ W,
                                                          const float* __restrict__ bias, E* __restrict__ y, int vbatch, int T,
                                                          int spb, int D, int hd, int ld_y, float scale) {
    constexpr int Kd = 64 * KD64, KPL = SbE<E>::KPL, KSTEP = SbE<E>::KSTEP, NK = Kd / KSTEP;
    constexpr int PITCH = Kd * (int)sizeof(E) + 16;
    constexpr int NP = sizeof(E) == 2 ? 3 : 1;             // parts whose weight fragments are in flight together
    __shared__ __attribute__((aligned(16))) unsigned char at[kSbRows * PITCH];
    __shared__ __attribute__((aligned(16))) float qs[3][kSbRows][kSbHP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, s0 = blockIdx.y * spb;
    const int ns = min(spb, vbatch - s0), rows = ns * T, m0 = s0 * T;
    const int li = lane & 15, lg = lane >> 4;
    // feature (dim of the head) of this lane's weight row; dims past hd read a valid row and are zeroed when stored
    const int dw = 16 * wave + li, dwc = dw < hd ? dw : 0;
    auto wrow = [&](int p) { return W + (size_t)(p * D + h * hd + dwc) * Kd + KPL * lg; };
    LnTile<E, KD64, TPR> ln;
    ln.load(x, gamma, beta, m0, rows, D, tid);
    u32x4 wf[NP][NK];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const E* wp = wrow(p);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) wf[p][ks] = *(const u32x4*)(wp + ks * KSTEP);
    }
    ln.finish(at, rows, D, tid);
    __syncthreads();
    const bool two = rows > 16;                            // (one sample of <= 16 tokens: the second row tile is idle)
    const int d0 = 16 * wave + 4 * lg;                     // the lane ends with dims d0 .. d0 + 3 of token 16 rt + li
    auto put = [&](int p, const f32x4 (&acc)[2]) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (d0 < hd) bv = *(const f32x4*)(bias + p * D + h * hd + d0);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x4 v = acc[rt] + bv;
            if (d0 >= hd) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *(f32x4*)&qs[p][16 * rt + li][d0] = v;
        }
    };
    if constexpr (NP == 3) {
        // bf16: the three parts' products as independent accumulator chains, k-step by k-step (one activation fragment read
        // serves all three)
        f32x4 acc[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p) { acc[p][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[p][1] = acc[p][0]; }
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const u32x4 a0 = *(const u32x4*)(at + li * PITCH + (ks * KSTEP + KPL * lg) * (int)sizeof(E));
#pragma unroll
            for (int p = 0; p < 3; ++p) SbE<E>::mma(acc[p][0], wf[p][ks], a0);
            if (two) {
                const u32x4 a1 = *(const u32x4*)(at + (16 + li) * PITCH + (ks * KSTEP + KPL * lg) * (int)sizeof(E));
#pragma unroll
                for (int p = 0; p < 3; ++p) SbE<E>::mma(acc[p][1], wf[p][ks], a1);
            }
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) put(p, acc[p]);
    } else {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            if (p > 0) {                                   // (fp32: one part's 96 registers of fragments at a time)
                const E* wp = wrow(p);
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) wf[0][ks] = *(const u32x4*)(wp + ks * KSTEP);
            }
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const u32x4 a0 = *(const u32x4*)(at + li * PITCH + (ks * KSTEP + KPL * lg) * (int)sizeof(E));
                SbE<E>::mma(acc[0], wf[0][ks], a0);
                if (two) {
                    const u32x4 a1 = *(const u32x4*)(at + (16 + li) * PITCH + (ks * KSTEP + KPL * lg) * (int)sizeof(E));
                    SbE<E>::mma(acc[1], wf[0][ks], a1);
                }
            }
            put(p, acc);
        }
    }
    __syncthreads();
    // attention: item = (sample, query row); 16 lanes of a DPP row per item, lane dq holds dims 4 dq .. +3 of q, k, v and the
    // output -- and the softmax of key dq: every lane reduces all the scores (the partial dot products summed over the row),
    // keeps the one of ITS key, and max / exp / sum / divide happen once per (query, key) with butterfly reductions over the
    // row (each step adds the same two values in both partners: all 16 lanes end with the same bits); the probabilities come
    // back through 64 bytes of LDS per item (the LayerNorm tile: every wave is past its MFMAs; a wave's LDS operations run in
    // order).  With every lane evaluating all T exponentials and quotients this section was ~700 of the kernel's ~1,500
    // instructions per wave -- and these launches are bound by what a wave issues.
    const int dq = tid & 15;
    float* pl = (float*)at + (tid >> 4) * 16;
    auto row_max = [](float v) {
        v = fmaxf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xf, 0xf, false)));
        v = fmaxf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xf, 0xf, false)));
        v = fmaxf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x141, 0xf, 0xf, false)));
        v = fmaxf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x140, 0xf, 0xf, false)));
        return v;
    };
    auto row_sum = [](float v) {
        v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xf, 0xf, false));
        v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xf, 0xf, false));
        v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x141, 0xf, 0xf, false));
        v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x140, 0xf, 0xf, false));
        return v;
    };
#pragma unroll 1
    for (int pass = 0; pass < (rows > 16 ? 2 : 1); ++pass) {
        const int item = pass * 16 + (tid >> 4);
        const bool iv = item < rows;
        const int sm = iv ? item / T : 0, qi = iv ? item - sm * T : 0, r0 = sm * T;
        const f32x4 q4 = *(const f32x4*)&qs[0][r0 + qi][4 * dq];
        float mine = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < T) {                                   // (wave-uniform)
                const f32x4 k4 = *(const f32x4*)&qs[1][r0 + j][4 * dq];
                float part = q4[0] * k4[0];
                part = fmaf(q4[1], k4[1], part); part = fmaf(q4[2], k4[2], part); part = fmaf(q4[3], k4[3], part);
                part = row_sum(part);
                mine = dq == j ? part * scale : mine;
            }
        }
        const bool live = dq <= qi;                        // (the causal bound; qi < T)
        mine = live ? mine : -INFINITY;
        const float mx = row_max(mine);
        const float ex = live ? expf(mine - mx) : 0.f;
        const float den = row_sum(ex);
        pl[dq] = ex / den;
        f32x4 p4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p4[u] = *(const f32x4*)(pl + 4 * u);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < T) {
                const f32x4 v4 = *(const f32x4*)&qs[2][r0 + j][4 * dq];
                const float pj = p4[j >> 2][j & 3];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(pj, v4[e], o[e]);
            }
        }
        if (iv && 4 * dq < hd) store4<E>(y + (size_t)(m0 + item) * ld_y + h * hd + 4 * dq, o);
    }
    // the K padding of the attention output (columns D .. ld_y of the out-projection's operand): zeros, once per row
    // (thread = (row, piece of 8 elements): D and ld_y are multiples of 8 and 64, the padding is at most 7 pieces of the 32 rows)
    if (h == 0) {
        const int r = tid >> 3, c = D + 8 * (tid & 7);
        if (r < rows && c < ld_y) {
            E* yp = y + (size_t)(m0 + r) * ld_y + c;
            *(u32x4*)yp = u32x4{0u, 0u, 0u, 0u};
            if (sizeof(E) == 4) *(u32x4*)(yp + 4) = u32x4{0u, 0u, 0u, 0u};
        }
    }
}

// x[m][n] += A[m][:] . W[n][:] + bias[n]      (out-projection: A = attention output, K = Kd; FC2: A = GELU(h), K = Kh)
// grid (ceil(D / 16), ceil(M / rpb)), 256 threads: ONE 16-feature column tile per workgroup, its four waves split the
// contraction (k-step ks goes to wave ks % 4) and add up through LDS in a fixed order; both operands as fragments straight
// from memory, all of a wave's fragments requested at once (chunks of kSbChunk k-steps: the instance that covers K / 4 in one chunk where one exists).  A's rows are padded to a multiple
// of 128 in the workspace (rows past M are never stored), its K padding holds zeros, W's is zeros.
template <typename E, int kSbChunk>      // k-steps a wave has in flight at once: 3 (K = 384 in bf16), 6 or 12
__global__ __launch_bounds__(256) void sb_gemm_resid_kernel(const E* __restrict__ A, int lda, const E* __restrict__ W, int K,
                                                            const float* __restrict__ bias, const float* xin, float* x, int M, int D,
                                                            int rpb) {           // xin: the residual read (x itself: in place)
    constexpr int KPL = SbE<E>::KPL, KSTEP = SbE<E>::KSTEP;
    __shared__ __attribute__((aligned(16))) f32x4 red[3][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * rpb, rows = min(rpb, M - m0);          // rpb: 32 token rows per workgroup, or 16 (one MFMA row tile)
    const int n0 = blockIdx.x * 16;
    const int li = lane & 15, lg = lane >> 4;
    const int nkt = K / KSTEP;
    const E* wp = W + (size_t)(n0 + li) * K + KPL * lg;
    const E* ap0 = A + (size_t)(m0 + li) * lda + KPL * lg;
    const E* ap1 = ap0 + (size_t)16 * lda;
    // (wave 0's slice of the residual and the bias: requested first, used last)
    const int n = n0 + 4 * lg;
    f32x4 xv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, bv = {0.f, 0.f, 0.f, 0.f};
    if (wave == 0 && n < D) {
        bv = *(const f32x4*)(bias + n);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
            if (16 * rt + li < rows) xv[rt] = *(const f32x4*)(xin + (size_t)(m0 + 16 * rt + li) * D + n);
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const bool two = rows > 16;                              // (one sample of <= 16 tokens: the second row tile is not fetched)
    for (int ks0 = wave; ks0 < nkt; ks0 += 4 * kSbChunk) {
        u32x4 wf[kSbChunk], a0[kSbChunk], a1[kSbChunk];
#pragma unroll
        for (int u = 0; u < kSbChunk; ++u) {
            const int ks = min(ks0 + 4 * u, nkt - 1);          // (wave-uniform; a clamped step is loaded and not used)
            wf[u] = *(const u32x4*)(wp + ks * KSTEP);
            a0[u] = *(const u32x4*)(ap0 + ks * KSTEP);
            if (two) a1[u] = *(const u32x4*)(ap1 + ks * KSTEP);
        }
#pragma unroll
        for (int u = 0; u < kSbChunk; ++u) {
            if (ks0 + 4 * u < nkt) {
                SbE<E>::mma(acc[0], wf[u], a0[u]);
                if (two) SbE<E>::mma(acc[1], wf[u], a1[u]);
            }
        }
    }
    if (wave > 0) { red[wave - 1][0][lane] = acc[0]; red[wave - 1][1][lane] = acc[1]; }
    __syncthreads();
    if (wave == 0 && n < D) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int tok = 16 * rt + li;
            if (tok < rows) {
                f32x4 v = acc[rt];
#pragma unroll
                for (int w = 0; w < 3; ++w) v += red[w][rt][lane];
                *(f32x4*)(x + (size_t)(m0 + tok) * D + n) = xv[rt] + (v + bv);
            }
        }
    }
}

// The TWELVE-WAVE form of sb_qkv_attn_kernel (round 6; bf16): wave w owns part w / 4 (q, k, v) and dims 16 (w % 4) .. +15 of the
// head -- one set of weight fragments and one accumulator chain per wave instead of three of each; 16 ROWS threads normalise the
// tile's ROWS (16 or 32) rows with sixteen threads per row; the attention's (sample, query) items are one pass of the 48 item
// slots.  These launches are bound by what a wave issues: the four-wave form is ~1,500 instructions per
// wave, this one ~600 on three times the waves -- 12.0 -> 10.1 us per launch at 64 samples, 22.1 -> 18.2 at 128
// (profiles/r06_small_mid.txt).
// PROJ (few token rows): the out-projection as the EPILOGUE -- the head's attention output goes through LDS as activation
// fragments, the workgroup multiplies it by its head's hd columns of Wp (wave w: feature tiles w and w + 12) and writes the
// partial product to slab[h]; the FC1 launch's LayerNorm prologue adds the H slabs, the bias and the residual (LnTile::load_sum).
// One dependent launch less per layer (~5 us each on this part whatever they compute); y is not written.
template <int KD64, int ROWS, bool PROJ>
__global__ __launch_bounds__(768) void sb_qkv_attn_wide_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const uint16_t* __restrict__ W,
                                                               const float* __restrict__ bias, uint16_t* __restrict__ y, int vbatch,
                                                               int T, int spb, int D, int hd, int ld_y, float scale,
                                                               const uint16_t* __restrict__ Wp, float* __restrict__ slab,
                                                               size_t slab_stride) {
    typedef uint16_t E;
    constexpr int Kd = 64 * KD64, KPL = SbE<E>::KPL, KSTEP = SbE<E>::KSTEP, NK = Kd / KSTEP;
    constexpr int PITCH = Kd * (int)sizeof(E) + 16;
    __shared__ __attribute__((aligned(16))) unsigned char at[kSbRows * PITCH];
    __shared__ __attribute__((aligned(16))) float qs[3][kSbRows][kSbHP];
    __shared__ __attribute__((aligned(16))) uint16_t yts[PROJ ? ROWS * 72 : 8];      // PROJ: the head's attention output, 144-byte rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, s0 = blockIdx.y * spb;
    const int ns = min(spb, vbatch - s0), rows = ns * T, m0 = s0 * T;
    const int li = lane & 15, lg = lane >> 4;
    const int part = wave >> 2, dt = wave & 3;
    const int dw = 16 * dt + li, dwc = dw < hd ? dw : 0;   // dims past hd read a valid row and are zeroed when stored
    static_assert(ROWS == 16 || ROWS == kSbRows, "one or two MFMA row tiles");
    LnTile<E, KD64, 16> ln;
    if (tid < 16 * ROWS) ln.load(x, gamma, beta, m0, rows, D, tid);
    u32x4 wf[NK];
    {
        const E* wp = W + (size_t)(part * D + h * hd + dwc) * Kd + KPL * lg;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) wf[ks] = *(const u32x4*)(wp + ks * KSTEP);
    }
    const int d0 = 16 * dt + 4 * lg;                       // the lane ends with dims d0 .. d0 + 3 of token 16 rt + li
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (d0 < hd) bv = *(const f32x4*)(bias + part * D + h * hd + d0);
    if (tid < 16 * ROWS) ln.finish(at, rows, D, tid);
    __syncthreads();
    const bool two = ROWS > 16 && rows > 16;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const u32x4 a0 = *(const u32x4*)(at + li * PITCH + (ks * KSTEP + KPL * lg) * (int)sizeof(E));
        SbE<E>::mma(acc[0], wf[ks], a0);
        if (two) {
            const u32x4 a1 = *(const u32x4*)(at + (16 + li) * PITCH + (ks * KSTEP + KPL * lg) * (int)sizeof(E));
            SbE<E>::mma(acc[1], wf[ks], a1);
        }
    }
    // PROJ: this wave's fragments of Wp[:, h hd ..] -- feature tiles `wave` and `wave + 12`, two k-steps of 32 dims (the head's
    // columns start at a multiple of hd = 4 n elements: 8-byte pieces; dims past hd meet zeros of the attention output) --
    // requested here, where the q | k | v fragments are dead: they arrive under the attention
    uint2 pw[2][2][2];
    if constexpr (PROJ) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int ft = wave + 12 * f;
            const uint16_t* wp = Wp + (size_t)(16 * min(ft, (D + 15) / 16 - 1) + li) * Kd + h * hd + KPL * lg;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { pw[f][ks][0] = *(const uint2*)(wp + 32 * ks); pw[f][ks][1] = *(const uint2*)(wp + 32 * ks + 4); }
        }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f32x4 v = acc[rt] + bv;
        if (d0 >= hd) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *(f32x4*)&qs[part][16 * rt + li][d0] = v;
    }
    __syncthreads();
    // attention as in sb_qkv_attn_kernel: item = (sample, query row), 16 lanes of a DPP row per item -- 48 item slots, one pass
    const int dq = tid & 15, item = tid >> 4;
    float* pl = (float*)at + item * 16;
    auto row_max = [](float v) {
        v = fmaxf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xf, 0xf, false)));
        v = fmaxf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xf, 0xf, false)));
        v = fmaxf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x141, 0xf, 0xf, false)));
        v = fmaxf(v, __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x140, 0xf, 0xf, false)));
        return v;
    };
    auto row_sum = [](float v) {
        v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0xB1, 0xf, 0xf, false));
        v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4E, 0xf, 0xf, false));
        v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x141, 0xf, 0xf, false));
        v += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x140, 0xf, 0xf, false));
        return v;
    };
    {
        const bool iv = item < rows;
        const int sm = iv ? item / T : 0, qi = iv ? item - sm * T : 0, r0 = sm * T;
        const f32x4 q4 = *(const f32x4*)&qs[0][r0 + qi][4 * dq];
        float mine = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < T) {                                   // (wave-uniform)
                const f32x4 k4 = *(const f32x4*)&qs[1][r0 + j][4 * dq];
                float part_ = q4[0] * k4[0];
                part_ = fmaf(q4[1], k4[1], part_); part_ = fmaf(q4[2], k4[2], part_); part_ = fmaf(q4[3], k4[3], part_);
                part_ = row_sum(part_);
                mine = dq == j ? part_ * scale : mine;
            }
        }
        const bool live = dq <= qi;                        // (the causal bound; qi < T)
        mine = live ? mine : -INFINITY;
        const float mx = row_max(mine);
        const float ex = live ? expf(mine - mx) : 0.f;
        const float den = row_sum(ex);
        pl[dq] = ex / den;
        f32x4 p4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p4[u] = *(const f32x4*)(pl + 4 * u);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < T) {
                const f32x4 v4 = *(const f32x4*)&qs[2][r0 + j][4 * dq];
                const float pj = p4[j >> 2][j & 3];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(pj, v4[e], o[e]);
            }
        }
        if constexpr (PROJ) {
            // y_h as bf16 rows of 64 dims (+ 8: 144-byte rows: conflict-free fragment reads); zeros past hd / rows
            constexpr int YP = 72;
            uint16_t* yt = yts;
            if (item < ROWS) {
                f32x4 ov = (iv && 4 * dq < hd) ? o : f32x4{0.f, 0.f, 0.f, 0.f};
                store4<E>(yt + item * YP + 4 * dq, ov);
            }
            __syncthreads();
            constexpr int NRT = ROWS / 16;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int ft = wave + 12 * f, n = 16 * ft + 4 * lg;
                if (16 * ft < D) {                         // (wave-uniform)
                    f32x4 pa[NRT];
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt) pa[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const u32x4 wfrag = {pw[f][ks][0].x, pw[f][ks][0].y, pw[f][ks][1].x, pw[f][ks][1].y};
#pragma unroll
                        for (int rt = 0; rt < NRT; ++rt) {
                            const u32x4 a0 = *(const u32x4*)(yt + (16 * rt + li) * YP + 32 * ks + KPL * lg);
                            SbE<E>::mma(pa[rt], wfrag, a0);
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt) {
                        const int tok = 16 * rt + li;
                        if (tok < rows && n < D) *(f32x4*)(slab + (size_t)h * slab_stride + (size_t)(m0 + tok) * D + n) = pa[rt];
                    }
                }
            }
        } else if (iv && 4 * dq < hd) store4<E>(y + (size_t)(m0 + item) * ld_y + h * hd + 4 * dq, o);
    }
    // the K padding of the attention output (columns D .. ld_y of the out-projection's operand): zeros, once per row
    if (!PROJ && h == 0 && tid < 256) {
        const int r = tid >> 3, c = D + 8 * (tid & 7);
        if (r < rows && c < ld_y) *(u32x4*)(y + (size_t)(m0 + r) * ld_y + c) = u32x4{0u, 0u, 0u, 0u};
    }
}

// compute units of the current device (cached per device; 256 on the MI355X)
int device_cus_small() {
    constexpr int kMaxDev = 64;
    static std::atomic<int> cached[kMaxDev] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

template <typename E, int KD64>
hipError_t run_layers(const Layout& lay, const Workspace& ws, const char* packed, const FwdArgs& a, char* wsp, int precision,
                      hipStream_t s) {
    float* x = (float*)(wsp + ws.x);
    E* qkv = (E*)(wsp + ws.qkv); E* y = (E*)(wsp + ws.y); E* h = (E*)(wsp + ws.h);
    const int M = a.vbatch * a.T, D = lay.D;
    // the head-split LN1 + q|k|v + attention launch: whole samples in a 32-row tile, a head in 64 dims of float4 pieces
    const bool head_fused = a.T <= 16 && lay.hd <= 64 && lay.hd % 4 == 0;
    // 16-ROW tiles (one MFMA row tile, LayerNorm with 16 threads per row, whole samples of <= 16 tokens per attention tile)
    // while every workgroup of a launch still finds a slot at once (three per CU: 149 VGPRs, 25 KB of LDS): a launch of this
    // path is as long as the instructions its waves issue, so twice the workgroups of half the rows each are faster than
    // fewer, fuller ones (kitchen, 16 samples: 11 row tiles x 24 column tiles = 264 workgroups; 3-step DDIM at 32 / 40 samples
    // 672 / 775 -> 644 / 712 us; beyond three per CU nothing more: fp32 at 64 samples 1478 us with 32-row tiles, 1519 with 16)
    const int widest = std::max((D + 15) / 16, lay.Nh / 64);
    const bool one16 = head_fused && ((M + 15) / 16) * widest <= 3 * device_cus_small();
    const int rpb = one16 ? 16 : kSbRows, rb = (M + rpb - 1) / rpb;
    const int spb = a.T <= rpb ? rpb / a.T : 0;
    constexpr bool kTwelve = sizeof(E) == 2;             // the twelve-wave head-split attention launch (bf16)
    // few token rows (bf16): the out-projection rides in the attention launch as per-head partial products, summed by the FC1
    // launch's LayerNorm prologue -- three launches per layer instead of four (Workspace::small: x_mid + H slabs of kSmallProjRows rows)
    const bool fuse_proj = kTwelve && one16 && M <= kSmallProjRows;
    float* xm = (float*)(wsp + ws.small);
    float* slab = xm + (size_t)kSmallProjRows * D;
    const size_t slab_stride = (size_t)kSmallProjRows * D;
    auto F = [&](size_t off) { return (const float*)(packed + off); };
    auto launch_resid = [&](const E* A, int K, const E* Wt, const float* bias, const float* xin) {
        const int per_wave = (K / SbE<E>::KSTEP + 3) / 4;
        const dim3 grid((D + 15) / 16, rb);
        if (per_wave <= 3) hipLaunchKernelGGL((sb_gemm_resid_kernel<E, 3>), grid, dim3(256), 0, s, A, K, Wt, K, bias, xin, x, M, D, rpb);
        else if (per_wave <= 6) hipLaunchKernelGGL((sb_gemm_resid_kernel<E, 6>), grid, dim3(256), 0, s, A, K, Wt, K, bias, xin, x, M, D, rpb);
        else hipLaunchKernelGGL((sb_gemm_resid_kernel<E, 12>), grid, dim3(256), 0, s, A, K, Wt, K, bias, xin, x, M, D, rpb);
    };
    (void)hipGetLastError();
    for (int l = 0; l < lay.L; ++l) {
        const LayerOff& o = lay.layer[l];
        hipError_t e = hipSuccess;
        if (head_fused) {
            // LN1 -> q|k|v -> attention, split by head: one launch
            if constexpr (kTwelve) {
                const dim3 grid(lay.H, (a.vbatch + spb - 1) / spb);
                if (fuse_proj)
                    hipLaunchKernelGGL((sb_qkv_attn_wide_kernel<KD64, 16, true>), grid, dim3(768), 0, s, (const float*)x, F(o.ln1_w), F(o.ln1_b),
                                       (const uint16_t*)(packed + o.w_qkv), F(o.b_qkv), (uint16_t*)y, a.vbatch, a.T, spb, D, lay.hd, lay.Kd,
                                       1.0f / sqrtf((float)lay.hd), (const uint16_t*)(packed + o.w_proj), slab, slab_stride);
                else if (one16)
                    hipLaunchKernelGGL((sb_qkv_attn_wide_kernel<KD64, 16, false>), grid, dim3(768), 0, s, (const float*)x, F(o.ln1_w), F(o.ln1_b),
                                       (const uint16_t*)(packed + o.w_qkv), F(o.b_qkv), (uint16_t*)y, a.vbatch, a.T, spb, D, lay.hd, lay.Kd,
                                       1.0f / sqrtf((float)lay.hd), nullptr, nullptr, 0);
                else
                    hipLaunchKernelGGL((sb_qkv_attn_wide_kernel<KD64, kSbRows, false>), grid, dim3(768), 0, s, (const float*)x, F(o.ln1_w),
                                       F(o.ln1_b), (const uint16_t*)(packed + o.w_qkv), F(o.b_qkv), (uint16_t*)y, a.vbatch, a.T, spb, D,
                                       lay.hd, lay.Kd, 1.0f / sqrtf((float)lay.hd), nullptr, nullptr, 0);
            } else if (one16)
                hipLaunchKernelGGL((sb_qkv_attn_kernel<E, KD64, 16>), dim3(lay.H, (a.vbatch + spb - 1) / spb), dim3(256), 0, s, (const float*)x,
                                   F(o.ln1_w), F(o.ln1_b), (const E*)(packed + o.w_qkv), F(o.b_qkv), y, a.vbatch, a.T, spb, D, lay.hd,
                                   lay.Kd, 1.0f / sqrtf((float)lay.hd));
            else
                hipLaunchKernelGGL((sb_qkv_attn_kernel<E, KD64, 8>), dim3(lay.H, (a.vbatch + spb - 1) / spb), dim3(256), 0, s, (const float*)x,
                                   F(o.ln1_w), F(o.ln1_b), (const E*)(packed + o.w_qkv), F(o.b_qkv), y, a.vbatch, a.T, spb, D, lay.hd,
                                   lay.Kd, 1.0f / sqrtf((float)lay.hd));
        } else {
            hipLaunchKernelGGL((sb_ln_gemm_kernel<E, KD64, 0, 8>), dim3(lay.Nqkv / 64, rb), dim3(256), 0, s, (const float*)x, F(o.ln1_w),
                               F(o.ln1_b), (const E*)(packed + o.w_qkv), F(o.b_qkv), qkv, M, D, 3 * D, 3 * D, SbSlabs{nullptr, 0, 0, nullptr, nullptr});
            e = launch_attention(qkv, y, a.vbatch, a.T, D, lay.H, lay.Kd, precision, s);
            if (e != hipSuccess) return e;
        }
        const SbSlabs none{nullptr, 0, 0, nullptr, nullptr};
        if (fuse_proj) {
            // x_mid = x + b_p + the H slabs (into xm), LayerNorm-2, FC1, GELU; then x = x_mid + FC2
            const SbSlabs sl{slab, slab_stride, lay.H, F(o.b_proj), xm};
            hipLaunchKernelGGL((sb_ln_gemm_kernel<E, KD64, 1, 16, true>), dim3(lay.Nh / 64, rb), dim3(256), 0, s, (const float*)x, F(o.ln2_w),
                               F(o.ln2_b), (const E*)(packed + o.w_fc1), F(o.b_fc1), h, M, D, lay.Kh, lay.Kh, sl);
            launch_resid((const E*)h, lay.Kh, (const E*)(packed + o.w_fc2), F(o.b_fc2), xm);
        } else {
            launch_resid((const E*)y, lay.Kd, (const E*)(packed + o.w_proj), F(o.b_proj), x);
            if (one16)
                hipLaunchKernelGGL((sb_ln_gemm_kernel<E, KD64, 1, 16>), dim3(lay.Nh / 64, rb), dim3(256), 0, s, (const float*)x, F(o.ln2_w),
                                   F(o.ln2_b), (const E*)(packed + o.w_fc1), F(o.b_fc1), h, M, D, lay.Kh, lay.Kh, none);
            else
                hipLaunchKernelGGL((sb_ln_gemm_kernel<E, KD64, 1, 8>), dim3(lay.Nh / 64, rb), dim3(256), 0, s, (const float*)x, F(o.ln2_w),
                                   F(o.ln2_b), (const E*)(packed + o.w_fc1), F(o.b_fc1), h, M, D, lay.Kh, lay.Kh, none);
            launch_resid((const E*)h, lay.Kh, (const E*)(packed + o.w_fc2), F(o.b_fc2), x);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

// Shapes and batches this path takes: bf16 / fp32 (the generic section of the packed image), D a multiple of 8 up to 384
// (LayerNorm tile in static LDS), whatever the per-op embedding / attention / head kernels take.
bool small_supported(const Layout& lay, int precision) {
    if (precision != BESO_PREC_BF16 && precision != BESO_PREC_FP32) return false;
    return lay.D % 8 == 0 && lay.Kd <= 384;
}

// The library's choice: up to kSmallRows token rows (where the one-launch kernel's latency instance would leave most of the
// chip idle), unless the call's plan asks for specific kernels (per-op / block forms, a samples-per-workgroup instance) --
// BESO_PLAN_SMALL asks for this path at any size.
bool small_wanted(const Layout& lay, const FwdArgs& a, int precision) {
    if (!small_supported(lay, precision)) return false;
    if (a.plan & BESO_PLAN_SMALL) return true;
    if (a.plan & (BESO_PLAN_PER_OP | BESO_PLAN_BLOCKS | BESO_PLAN_FUSED | BESO_PLAN_SPW_MASK)) return false;
    if (precision == BESO_PREC_FP32) return a.vbatch * a.T <= kSmallRowsF32;
    // bf16: against the one-launch kernel's latency instance the chain of launches wins where there are weights to spread --
    // kitchen (6 layers of 360^2: 172 vs 258 us at one sample), not block-push (4 layers of 240^2: 127 vs 114 us)
    return a.vbatch * a.T <= kSmallRows && (size_t)lay.L * lay.D * lay.D >= kSmallMinLDD;
}

int forward_small(const Layout& lay, const Workspace& ws, const char* packed, int precision, const FwdArgs& a, char* wsp,
                  hipStream_t s, hipError_t* err) {
    float* x = (float*)(wsp + ws.x);
    profile_begin(BESO_SITE_EMBED, s);
    // (one block per token row up to 2048 rows: the one-block-per-sample form walks a sample's tokens one after the other -- 19 us at
    //  64 ... 256 samples where the rows' blocks take ~8; bit-identical results)
    hipError_t e = launch_embed(lay, packed, a, x, s, a.vbatch * a.T <= 2048);
    profile_end(BESO_SITE_EMBED, s);
    if (e != hipSuccess) { *err = e; return BESO_ERR_HIP; }
    profile_begin(BESO_SITE_SMALL, s);
    const int kd64 = lay.Kd / 64;
#define SB_RUN(E)                                                                                          \
    (kd64 <= 1 ? run_layers<E, 1>(lay, ws, packed, a, wsp, precision, s)                                  \
     : kd64 == 2 ? run_layers<E, 2>(lay, ws, packed, a, wsp, precision, s)                                \
     : kd64 == 3 ? run_layers<E, 3>(lay, ws, packed, a, wsp, precision, s)                                \
     : kd64 == 4 ? run_layers<E, 4>(lay, ws, packed, a, wsp, precision, s)                                \
     : kd64 == 5 ? run_layers<E, 5>(lay, ws, packed, a, wsp, precision, s)                                \
                 : run_layers<E, 6>(lay, ws, packed, a, wsp, precision, s))
    e = precision == BESO_PREC_FP32 ? SB_RUN(float) : SB_RUN(uint16_t);
#undef SB_RUN
    profile_end(BESO_SITE_SMALL, s);
    if (e != hipSuccess) { *err = e; return BESO_ERR_HIP; }
    profile_begin(BESO_SITE_HEAD, s);
    e = launch_head(lay, packed, a, x, s);
    profile_end(BESO_SITE_HEAD, s);
    if (e != hipSuccess) { *err = e; return BESO_ERR_HIP; }
    return BESO_OK;
}

}  // namespace beso
