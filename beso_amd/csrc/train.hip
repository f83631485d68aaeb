// Training step of the score network in HIP (row f1): GCDenoiser.loss forward + the gradient of the loss
// with respect to every parameter, enqueued on one stream.
//   reference: score_wrappers.py:45-79 (loss), score_gpts.py:272-358 (network), beso_agent.py:215-248 (step)
//
// Structure (M = batch * T token rows; E = GEMM operand type: bf16, or fp32 in the parity mode):
//   forward    prep (noised action, target) -> embed (x0, feature matrix Xemb) -> L x [ LN1 -> QKV GEMM ->
//              attention (dropout on the probabilities) -> proj GEMM + residual -> LN2 -> FC1 GEMM (+GELU) ->
//              FC2 GEMM + residual ] -> ln_f -> head GEMM -> squared error.  Every GEMM input is kept.
//   backward   the same chain reversed.  All contractions run on ONE MFMA GEMM kernel (`tgemm_kernel`) whose
//              operands may be stored contraction-major ("k-slow"): dgrad reads W[out][in] as it lies
//              (contraction over `out`), wgrad reads dY[m][n] and X[m][k] as they lie (contraction over the
//              token index m).  k-slow bf16 tiles are fetched from LDS with ds_read_b64_tr_b16 (the gfx950
//              transpose read), k-slow fp32 tiles with plain ds_read_b32 (v_mfma_f32_16x16x4_f32 takes one
//              element per lane), so nothing is ever transposed through HBM.  The weight gradients of the whole
//              step run as ONE grouped launch after the data-gradient chain (every tile contracts over all token
//              rows and stores its result: no split-K, no atomics).
//   LayerNorm backward also carries the residual gradient and emits its operand-typed copy (with the
//   dropout mask of the consuming linear layer) and that layer's bias gradient; GELU' is the epilogue of the FC2
//   dgrad GEMM.  Bias / LayerNorm-affine gradients are the only accumulations (block partials + one reduction,
//   or a handful of atomics per block): the flat gradient buffer is zeroed first.
// Dropout masks come from a counter-based hash of (seed, site, element index): the backward recomputes them.
#include <string.h>
#include <atomic>
#include "common.h"
#include "fused.h"
#include "train.h"

namespace beso {
namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr uint32_t kEmbedSite = 4 * kMaxLayers + 16;     // dropout site of the embeddings (layer sites are 4 l + {0, 1, 2})
constexpr uint32_t kGoalSite = 4 * kMaxLayers + 17;      // DiffusionGPT.mask_cond: elementwise Bernoulli over goals [B,G,obs]

// mask_cond (score_gpts.py:360-371): cond * (1 - bernoulli(p)), elementwise over [B, G, obs], NO rescaling of the kept
// elements.  keep(b, g, c) = 1 iff the hash-uniform of element ((b*G + g)*obs + c) is >= p.
__device__ __forceinline__ float goal_keep(uint32_t seed, size_t idx, float p) { return drop_scale(seed, kGoalSite, idx, p, 1.0f); }

__global__ void goal_mask_kernel(float* __restrict__ mask, size_t n, float p, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        mask[i] = p > 0.f ? goal_keep(seed, i, p) : 1.f;
}

// GELU and its derivative (nn.GELU(), score_gpts.py:107): exact erf / exp in the fp32 mode, the fitted polynomial of
// the inference kernels (max |error| 1.9e-4, below the bf16 rounding of the stored value) in the bf16 mode
template <typename E> __device__ __forceinline__ float gelu_t(float v);
template <> __device__ __forceinline__ float gelu_t<float>(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
template <> __device__ __forceinline__ float gelu_t<uint16_t>(float v) { return gelu_poly(v); }
template <typename E> __device__ __forceinline__ float gelu_grad_t(float v);
template <> __device__ __forceinline__ float gelu_grad_t<float>(float v) {
    return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.39894228040143267794f * expf(-0.5f * v * v);
}
template <> __device__ __forceinline__ float gelu_grad_t<uint16_t>(float v) { return gelu_grad_poly(v); }

// ---------------------------------------------------------------------------------------------
// tgemm:  C[m][n] = sum_k Aop(m,k) * Bop(n,k)
//   AKS = false: A is [M][lda], k contiguous ("k-contig")      AKS = true: A is [K][lda], m contiguous ("k-slow")
//   BKS likewise for B ([N][ldb] / [K][ldb]).
// 128x128 block tile, 8 waves as 2x4 of 64x32, 16x16 MFMA tiles; one stage = 128 bytes of k per row
// (64 bf16 / 32 fp32); register-staged, two stages in flight; all loads predicated (zero fill), so M, N, K are
// arbitrary up to: K % (16/sizeof(E)) == 0 for a k-contig operand, rows % (16/sizeof(E)) == 0 for a k-slow one.
// blockIdx.y splits K (k_per_split, a multiple of the stage); the epilogue functor decides what a partial
// sum means (EpiAtomic accumulates).
// ---------------------------------------------------------------------------------------------
constexpr int kOpBytes = 16896;        // LDS bytes of one operand stage (k-slow images carry padding)
constexpr int kSubBytes = 2080;        // bf16 k-slow image: one 16-column subtile = 64 k-rows x 32 B + 32 B pad
constexpr int kRowBytes32 = 528;       // fp32 k-slow image: one k-row = 128 columns x 4 B + 16 B pad

// Global side of an operand: raw buffer loads with 32-bit per-lane byte offsets (computed once per tile) and the
// k advance in the scalar offset.  Nothing relies on the descriptor's range check: rows / columns outside the
// operand are read at offset 0 (their products land in outputs that are never stored), lanes past k_end are read
// at a valid address and replaced by zeros.
#ifndef BESO_TGEMM_WAVES
#define BESO_TGEMM_WAVES 8                 // 8: waves as 2 x 4, 64 x 32 each; 4: 2 x 2, 64 x 64 each (A/B builds)
#endif
constexpr int kNW = BESO_TGEMM_WAVES;
constexpr int kGT = 64 * kNW;              // threads of a tgemm workgroup
constexpr int kGL = 1024 / kGT;            // 16-byte chunks per thread, operand and stage
constexpr int kWN = kNW == 8 ? 4 : 2;      // waves along n
constexpr int kOcc = kNW == 8 ? 4 : 2;     // waves per SIMD asked of the compiler (two workgroups per CU)
template <int NCH> struct GOp { uint32_t voff[NCH]; };          // (the descriptor is rebuilt from the kernel argument at every use: a
                                           //  descriptor carried in VGPRs makes every load a waterfall loop)

template <typename E, bool KS, int NCH>
__device__ __forceinline__ GOp<NCH> make_gop(const E* __restrict__ P, int ld, int r0, int R, int tid) {
    constexpr int EPC = 16 / (int)sizeof(E);
    GOp<NCH> g;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + kGT * i;
        if (!KS) {
            const int row = c >> 3, kc = c & 7, gr = r0 + row;
            g.voff[i] = gr < R ? (uint32_t)(((size_t)gr * ld + kc * EPC) * sizeof(E)) : 0u;
        } else {
            constexpr int CPR = 128 / EPC;               // 16-byte chunks per k-row
            const int krow = c / CPR, cc = c % CPR, gc = r0 + cc * EPC;
            g.voff[i] = gc < R ? (uint32_t)(((size_t)krow * ld + gc) * sizeof(E)) : 0u;
        }
    }
    return g;
}

template <typename E, bool KS, int NCH, bool RAW = false>      // RAW: lanes past k_end are zeroed by op_lstore_masked
__device__ __forceinline__ void op_gload(const E* __restrict__ P, const GOp<NCH>& g, int ld, int k0, int k_end, int tid,
                                         u32x4 (&r)[NCH]) {
    constexpr int EPC = 16 / (int)sizeof(E);
    // P and the k offset are wave-uniform; saying so keeps the descriptor and the scalar offset in SGPRs
    const uint64_t pv = (uint64_t)P;
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pv >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pv);      // (the builtin returns int)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pu, 0, 0x7fffffff, 0x00020000);
    const bool stage_ok = k0 < k_end;                        // a whole stage past the end reads stage 0 (discarded)
    const u32x4 zero = {0u, 0u, 0u, 0u};
    if (!KS) {
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(stage_ok ? (uint32_t)k0 * (uint32_t)sizeof(E) : 0u);
        const bool ok = k0 + (tid & 7) * EPC < k_end;        // the chunk index along k is the same for all loads of a thread
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? g.voff[i] : 0u, soff, 0);
            r[i] = (RAW || ok) ? v : zero;
        }
    } else {
        constexpr int CPR = 128 / EPC;
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(stage_ok ? (uint32_t)k0 * (uint32_t)ld * (uint32_t)sizeof(E) : 0u);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const bool ok = k0 + (tid + kGT * i) / CPR < k_end;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? g.voff[i] : 0u, soff, 0);
            r[i] = (RAW || ok) ? v : zero;
        }
    }
}

// `ones`: bit i set = chunk i of this thread is replaced by (1, 0, 0, ...) -- the ONES COLUMN of the weight-gradient tiles: with
// a column of ones behind the last real column of the activation operand, the product dY^T [X | 1] carries the bias gradient
// (the column sums of dY) in that column for free (tgemm_wgrad_group_kernel)
template <typename E, bool KS, int NCH>
__device__ __forceinline__ void op_lstore(unsigned char* base, int tid, const u32x4 (&r)[NCH], uint32_t ones = 0u) {
    const u32x4 one = {sizeof(E) == 2 ? 0x3F80u : 0x3F800000u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + kGT * i;
        int off;
        if (!KS) {
            const int row = c >> 3, kc = c & 7;
            off = row * 128 + ((kc ^ (row & 7)) << 4);
        } else if (sizeof(E) == 2) {
            const int krow = c >> 4, cc = c & 15;        // 8 columns per chunk: subtile cc/2, half cc%2
            // k-row r = 8g + 4h + j of a 32-row group is kept at row 16h + 4g + j: the four 16-lane groups of one
            // transpose read (fixed h) then fetch 512 contiguous bytes (no bank conflict; with the rows in natural
            // order groups 0/1 and 2/3 were 256 B apart = the same banks: half of all LDS cycles were conflicts)
            const int prow = (krow & ~31) | ((krow & 4) << 2) | ((krow & 24) >> 1) | (krow & 3);
            off = (cc >> 1) * kSubBytes + prow * 32 + (cc & 1) * 16;
        } else {
            const int krow = c >> 5, cc = c & 31;
            off = krow * kRowBytes32 + cc * 16;
        }
        *(u32x4*)(base + off) = (ones >> i) & 1u ? one : r[i];
    }
}

// op_lstore of a RAW-loaded stage that began at k0: the zeroing of the lanes past k_end happens here, when the data
// is consumed, so that nothing touches the registers of a load in flight
template <typename E, bool KS, int NCH>
__device__ __forceinline__ void op_lstore_masked(unsigned char* base, int tid, const u32x4 (&r)[NCH], int k0, int k_end,
                                                 uint32_t ones = 0u) {
    constexpr int EPC = 16 / (int)sizeof(E), CPR = 128 / EPC, KSTAGE = 128 / (int)sizeof(E);
    if (k0 + KSTAGE <= k_end) {                 // (wave-uniform) a whole stage: nothing to zero
        op_lstore<E, KS, NCH>(base, tid, r, ones);
        return;
    }
    const u32x4 one = {sizeof(E) == 2 ? 0x3F80u : 0x3F800000u, 0u, 0u, 0u};
    u32x4 m[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const bool ok = KS ? (k0 + (tid + kGT * i) / CPR < k_end) : (k0 + (tid & 7) * EPC < k_end);
        m[i] = ok ? ((ones >> i) & 1u ? one : r[i]) : u32x4{0u, 0u, 0u, 0u};       // (rows past k_end contribute nothing)
    }
    op_lstore<E, KS, NCH>(base, tid, m);
}

// MFMA operand fragment of 16-row tile `tile` (0..7 of the block tile), half-stage s (0/1).
// k mapping (both layouts, both operands): bf16 slot j of lane group g is k = 32 s + 8 g + j; fp32 element j of
// lane group g is k = 16 s + 4 g + j.
template <typename E, bool KS>
__device__ __forceinline__ u32x4 op_frag(const unsigned char* base, int tile, int s, int lane) {
    const int i = lane & 15, g = lane >> 4;
    if (!KS) {
        const int row = tile * 16 + i, kc = s * 4 + g;
        return *(const u32x4*)(base + row * 128 + ((kc ^ (row & 7)) << 4));
    } else if (sizeof(E) == 2) {
        // ds_read_b64_tr_b16: the 16 lanes of a group cover a [4 k][16 col] block (lane i: k-row i/4, columns
        // 4(i%4)..+3) and receive column i of it, k-rows 0..3.  Two reads (h = 0, 1) = k 8g .. 8g+7 of column i.
        const unsigned char* p = base + tile * kSubBytes + (s * 32 + 4 * g + (i >> 2)) * 32 + (i & 3) * 8;   // permuted rows
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 16 * 32));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return u32x4{l2.x, l2.y, h2.x, h2.y};
    } else {
        u32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = *(const uint32_t*)(base + (s * 16 + 4 * g + j) * kRowBytes32 + (tile * 16 + i) * 4);
        return v;
    }
}

template <typename E> __device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mma16<uint16_t>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x4& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), acc, 0, 0, 0);
}

typedef unsigned char TileLds[2][2][kOpBytes];

// Workgroups are dealt to the 8 XCDs round robin by blockIdx.x, and every XCD has its own 4 MB L2.  Tiles that are
// neighbours in the logical order share an operand panel (same rows of A, next columns of B), so each XCD takes a
// CONTIGUOUS run of logical tiles: the panel is then fetched into one L2 instead of eight.
__device__ __forceinline__ int xcd_tile(int b, int nb) {
    const int per = nb >> 3, rem = nb & 7, x = b & 7, i = b >> 3;
    return x * per + (x < rem ? x : rem) + i;
}

// One 128 x 128 output tile at (m0, n0), contraction over [k_begin, k_end): 8 waves as 2 (m) x 4 (n), 64 x 32 each.
// Register-staged pipeline with two register sets (see the loop).  Loads and LDS stores are unconditional -- a
// stage past k_end is read at a valid address and stored as zeros -- so that the compiler's vmcnt bookkeeping
// stays exact.  (Measured without effect on the six-stage GEMMs of the step: persistent workgroups that keep the
// pipeline filled across tiles, 128-byte aligned leading dimensions.)
template <typename E, bool AKS, bool BKS, typename Epi>
__device__ __forceinline__ void tgemm_tile(TileLds& lds, const E* __restrict__ A, int lda, const E* __restrict__ B, int ldb,
                                           int M, int N, int m0, int n0, int k_begin, int k_end, const Epi& epi) {
    constexpr int KSTAGE = 128 / (int)sizeof(E);
    constexpr int NA = kGL;                            // 16-byte chunks of A per thread and stage
    constexpr int NI = 128 / (16 * kWN);               // 16-column MFMA tiles per wave
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / kWN, wn = wid % kWN;
    const int nk = (k_end - k_begin + KSTAGE - 1) / KSTAGE;

    f32x4 acc[4][NI];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](const unsigned char* la, const unsigned char* lb) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 bf[NI];                                 // (the two B fragments stay, the four A fragments pass through)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bf[ni] = op_frag<E, BKS>(lb, wn * NI + ni, s, lane);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const u32x4 af = op_frag<E, AKS>(la, wm * 4 + mi, s, lane);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mma16<E>(acc[mi][ni], bf[ni], af);   // D[n][m]: 4 consecutive n per lane
            }
            __builtin_amdgcn_sched_barrier(0);              // keeps the second half-stage's fragment reads out of the first (VGPRs)
        }
    };
    if (nk > 0) {
        // Two register sets, ping-pong: the stage that landed during the previous iteration is written to LDS BEFORE
        // the MFMAs of the current one while the stage after it is being fetched -- no wait for memory inside the
        // loop, LDS writes under the matrix pipe (the weight-gradient launch, hundreds of stages per tile: -29 %).
        u32x4 a0[NA], b0[kGL], a1[NA], b1[kGL];
        const GOp<NA> ga = make_gop<E, AKS, NA>(A, lda, m0, M, tid);
        const GOp<kGL> gb = make_gop<E, BKS, kGL>(B, ldb, n0, N, tid);
        uint32_t b_ones = 0u;                             // chunks of this thread that start at column N of B (the ones column)
        if constexpr (Epi::kBiasCol) {
            static_assert(BKS, "the ones column is a column of a k-slow B operand");
            constexpr int EPC = 16 / (int)sizeof(E), CPR = 128 / EPC;
            if (epi.bias) {
#pragma unroll
                for (int i = 0; i < kGL; ++i) b_ones |= (n0 + ((tid + kGT * i) % CPR) * EPC == N ? 1u : 0u) << i;
            }
        }
        auto load0 = [&](int kt) {
            op_gload<E, AKS, NA, true>(A, ga, lda, k_begin + kt * KSTAGE, k_end, tid, a0);
            op_gload<E, BKS, kGL, true>(B, gb, ldb, k_begin + kt * KSTAGE, k_end, tid, b0);
        };
        auto load1 = [&](int kt) {
            op_gload<E, AKS, NA, true>(A, ga, lda, k_begin + kt * KSTAGE, k_end, tid, a1);
            op_gload<E, BKS, kGL, true>(B, gb, ldb, k_begin + kt * KSTAGE, k_end, tid, b1);
        };
        load0(0);
        load1(1);
        op_lstore_masked<E, AKS, NA>(lds[0][0], tid, a0, k_begin, k_end);
        op_lstore_masked<E, BKS, kGL>(lds[0][1], tid, b0, k_begin, k_end, b_ones);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            load0(kt + 2);
            __builtin_amdgcn_sched_barrier(0);          // the loads are issued HERE (the scheduler sinks them below the MFMAs)
            op_lstore_masked<E, AKS, NA>(lds[1][0], tid, a1, k_begin + (kt + 1) * KSTAGE, k_end);
            op_lstore_masked<E, BKS, kGL>(lds[1][1], tid, b1, k_begin + (kt + 1) * KSTAGE, k_end, b_ones);
            compute(lds[0][0], lds[0][1]);
            __syncthreads();
            if (kt + 1 >= nk) break;
            load1(kt + 3);
            __builtin_amdgcn_sched_barrier(0);
            op_lstore_masked<E, AKS, NA>(lds[0][0], tid, a0, k_begin + (kt + 2) * KSTAGE, k_end);
            op_lstore_masked<E, BKS, kGL>(lds[0][1], tid, b0, k_begin + (kt + 2) * KSTAGE, k_end, b_ones);
            compute(lds[1][0], lds[1][1]);
            __syncthreads();
        }
    }
    int te = tid;
    asm volatile("" : "+v"(te));          // epilogue addresses are formed HERE, not hoisted above the k loop (spills)
    // The MFMA ran as D = Bfrag x Afrag^T: the lane holds C[m][n..n+3] with m = lane & 15, n = 4*(lane >> 4) + reg,
    // so every epilogue access is a 4-element vector (N % 4 == 0).
    const int le = te & 63, wme = (te >> 6) / kWN, wne = (te >> 6) % kWN;
    f32x4 cs[NI];                         // column sums of what this lane stored (epilogues with a bias gradient)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) cs[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wme * 64 + mi * 16 + (le & 15);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wne * (16 * NI) + ni * 16 + (le >> 4) * 4;
            if (m < M && n < N) {
                if constexpr (Epi::kColSum) cs[ni] += epi(m, n, acc[mi][ni]);
                else epi(m, n, acc[mi][ni]);
            }
            if constexpr (Epi::kBiasCol) {
                if (m < M && n == N && epi.bias) epi.bias[m] = acc[mi][ni][0] + (epi.add ? epi.bias[m] : 0.f);      // the ones column: sum over the rows of A's column m
            }
        }
    }
    if constexpr (Epi::kColSum) {
        // the 16 lanes with the same lane >> 4 hold the same four columns for 16 different rows: butterfly over
        // lane & 15, then one atomic per column, wave and tile (64 rows each)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) cs[ni][r] += __shfl_xor(cs[ni][r], o, 64);
            const int n = n0 + wne * (16 * NI) + ni * 16 + (le >> 4) * 4;
            if ((le & 15) == 0 && n < N) {
#pragma unroll
                for (int r = 0; r < 4; ++r) unsafeAtomicAdd(epi.colsum + n + r, cs[ni][r]);
            }
        }
    }
}

template <typename E, bool AKS, bool BKS, typename Epi>
__global__ __launch_bounds__(kGT, kOcc) void tgemm_kernel(const E* __restrict__ A, int lda, const E* __restrict__ B,
                                                       int ldb, int M, int N, int K, int k_per_split, int nt_n,
                                                       Epi epi) {
    __shared__ __attribute__((aligned(16))) TileLds lds;
    const int bt = xcd_tile(blockIdx.x, gridDim.x);
    const int tile_n = bt % nt_n, tile_m = bt / nt_n;
    const int k_begin = blockIdx.y * k_per_split;
    tgemm_tile<E, AKS, BKS, Epi>(lds, A, lda, B, ldb, M, N, tile_m * kTileMN, tile_n * kTileMN, k_begin,
                                 min(K, k_begin + k_per_split), epi);
}

// ---- epilogues: (m, n, v) = C[m][n..n+3] ------------------------------------------------------
template <typename E> struct Vec4;
template <> struct Vec4<float> {
    __device__ static __forceinline__ f32x4 load(const float* p) { return *(const f32x4*)p; }
    __device__ static __forceinline__ void store(float* p, const f32x4& v) { *(f32x4*)p = v; }
    __device__ static __forceinline__ f32x4 rounded(const f32x4& v) { return v; }
};
template <> struct Vec4<uint16_t> {
    __device__ static __forceinline__ f32x4 load(const uint16_t* p) {
        const uint2 u = *(const uint2*)p;
        return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u)};
    }
    __device__ static __forceinline__ void store(uint16_t* p, const f32x4& v) {
        uint2 u;
        u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *(uint2*)p = u;
    }
    __device__ static __forceinline__ f32x4 rounded(const f32x4& v) {
        return f32x4{bf2f(f2bf(v[0])), bf2f(f2bf(v[1])), bf2f(f2bf(v[2])), bf2f(f2bf(v[3]))};
    }
};

template <typename E> struct EpiStore {          // out = acc + bias  (fp32 and / or operand-typed copy)
    static constexpr bool kColSum = false; static constexpr bool kBiasCol = false;
    float* o32; E* oe; const float* bias; int ld;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
        if (bias) v += *(const f32x4*)(bias + n);
        const size_t i = (size_t)m * ld + n;
#ifdef BESO_TGEMM_NOSTORE               // timing experiment: the GEMM without its output traffic
        if (v[0] != 12345.678f) return;
#endif
        if (o32) *(f32x4*)(o32 + i) = v;
        if (oe) Vec4<E>::store(oe + i, v);
    }
};
template <typename E> struct EpiFc1 {            // h = acc + bias (kept for GELU'), g = GELU(h)   (score_gpts.py:105-108)
    static constexpr bool kColSum = false; static constexpr bool kBiasCol = false;
    E* h; E* g; const float* bias; int ld;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
        v += *(const f32x4*)(bias + n);
        const size_t i = (size_t)m * ld + n;
        Vec4<E>::store(h + i, v);
        Vec4<E>::store(g + i, f32x4{gelu_t<E>(v[0]), gelu_t<E>(v[1]), gelu_t<E>(v[2]), gelu_t<E>(v[3])});
    }
};
struct EpiResid {                                // x_out = x_in + dropout(acc + bias)              (:79,:109,:113-114)
    static constexpr bool kColSum = false; static constexpr bool kBiasCol = false;
    const float* xin; float* xout; const float* bias; int ld; float p, inv_keep; uint32_t seed, site;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
        v += *(const f32x4*)(bias + n);
        const size_t i = (size_t)m * ld + n;
        if (p > 0.f) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= drop_scale(seed, site, i + j, p, inv_keep);
        }
        *(f32x4*)(xout + i) = *(const f32x4*)(xin + i) + v;
    }
};
template <typename E> struct EpiGeluBwd {        // dh = dg * GELU'(h); its column sums are the FC1 bias gradient
    static constexpr bool kColSum = true; static constexpr bool kBiasCol = false;
    const E* h; E* dh; float* colsum; int ld;
    __device__ __forceinline__ f32x4 operator()(int m, int n, f32x4 v) const {
        const size_t i = (size_t)m * ld + n;
        const f32x4 hv = Vec4<E>::load(h + i);
        const f32x4 d = {v[0] * gelu_grad_t<E>(hv[0]), v[1] * gelu_grad_t<E>(hv[1]), v[2] * gelu_grad_t<E>(hv[2]),
                         v[3] * gelu_grad_t<E>(hv[3])};
        Vec4<E>::store(dh + i, d);
        return Vec4<E>::rounded(d);      // the sum is taken over the values as stored (what the weight gradient sees)
    }
};
template <typename E> struct EpiSilu {           // z = acc + bias (kept for SiLU'), a = SiLU(z): the hidden layer of the MLP action head
    static constexpr bool kColSum = false; static constexpr bool kBiasCol = false;       // (score_gpts.py:187-191: Linear(D,100) - SiLU - Linear(100,act))
    E* z; E* a; const float* bias; int ld;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
        v += *(const f32x4*)(bias + n);
        const size_t i = (size_t)m * ld + n;
        Vec4<E>::store(z + i, v);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = v[j] / (1.0f + expf(-v[j]));
        Vec4<E>::store(a + i, o);
    }
};
template <typename E> struct EpiSiluBwd {        // dz = da * SiLU'(z); column sums = bias gradient of the hidden layer
    static constexpr bool kColSum = true; static constexpr bool kBiasCol = false;
    const E* z; E* dz; float* colsum; int ld;
    __device__ __forceinline__ f32x4 operator()(int m, int n, f32x4 v) const {
        const size_t i = (size_t)m * ld + n;
        const f32x4 zv = Vec4<E>::load(z + i);
        f32x4 d;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sg = 1.0f / (1.0f + expf(-zv[j]));
            d[j] = v[j] * sg * (1.0f + zv[j] * (1.0f - sg));
        }
        Vec4<E>::store(dz + i, d);
        return Vec4<E>::rounded(d);
    }
};
#if BESO_DEV_API
struct EpiAtomic {                               // split-K partial sums accumulated with atomics (debug entry point)
    static constexpr bool kColSum = false; static constexpr bool kBiasCol = false;
    float* out; int ld;
    __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
        float* o = out + (size_t)m * ld + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(o + r, v[r]);
    }
};
inline EpiAtomic epi_atomic(float* out, int ld) { return EpiAtomic{out, ld}; }
#endif
template <typename E, bool AKS, bool BKS, typename Epi>
hipError_t tgemm(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int splits, Epi epi, hipStream_t s) {
    (void)hipGetLastError();
    constexpr int EPC = 16 / (int)sizeof(E), KSTAGE = 128 / (int)sizeof(E);
    if (M < 1 || N < 1 || K < 1) return hipErrorInvalidValue;
    if ((AKS ? M : K) % EPC != 0 || (BKS ? N : K) % EPC != 0 || lda % EPC != 0 || ldb % EPC != 0 || N % 4 != 0)
        return hipErrorInvalidValue;
    if (((uintptr_t)A | (uintptr_t)B) & 15) return hipErrorInvalidValue;
    // 32-bit byte offsets inside an operand (raw buffer loads)
    if ((size_t)(AKS ? K : M) * lda * sizeof(E) >= ((size_t)1 << 31) || (size_t)(BKS ? K : N) * ldb * sizeof(E) >= ((size_t)1 << 31))
        return hipErrorInvalidValue;
    const int nt_n = (N + kTileMN - 1) / kTileMN, nt_m = (M + kTileMN - 1) / kTileMN;
    if (splits < 1) splits = 1;
    const int kps = ((K + splits - 1) / splits + KSTAGE - 1) / KSTAGE * KSTAGE;
    splits = (K + kps - 1) / kps;
    // (64-row tiles for the GEMMs whose 128-row grid is under one round of workgroups measured 20 % SLOWER: the kernel
    //  is bound by operand bytes per FLOP through LDS, not by idle CUs)
    hipLaunchKernelGGL((tgemm_kernel<E, AKS, BKS, Epi>), dim3(nt_n * nt_m, splits), dim3(kGT), 0, s, (const E*)A, lda,
                       (const E*)B, ldb, M, N, K, kps, nt_n, epi);
    return hipGetLastError();
}

// All weight gradients of a step in ONE launch: problem p is C_p[Mo][No] = A_p^T B_p over the M token rows (both
// operands k-slow: the kept activation and the kept output gradient as they lie).  Every tile contracts over
// the whole token range and stores its result -- no split-K, no atomics (fp32 atomics sustain ~35 G/s on this
// part: a split-K version spent more time adding partial sums than multiplying), no reduction pass; a few hundred
// tiles of equal length fill the chip by themselves.
struct GProb { const void* A; const void* B; float* out; float* bias; int lda, ldb, Mo, No, tile_begin, nt_n, K;
               uint32_t slab_off; };   // bias: column sums of A (or nullptr); slab_off: this problem's floats in a split's slab (out, then bias)
constexpr int kMaxGroup = 56;                          // 6 per layer + 2: up to 9 layers per launch, more launches beyond (a 4 KiB kernel argument)
struct GTable { GProb p[kMaxGroup]; int n; };
struct EpiStoreF { static constexpr bool kColSum = false; static constexpr bool kBiasCol = true; float* out; int ld; float* bias;
    bool add;                                          // out += v (a later row window of the same product: launches in stream order)
    __device__ __forceinline__ void operator()(int m, int n, f32x4 v) const {
        f32x4* o = (f32x4*)(out + (size_t)m * ld + n);
        *o = add ? *o + v : v;
    } };

// splits > 1 (round 4): the contraction -- the TOKEN ROWS of the step -- is cut into `splits` ranges, one workgroup per (tile,
// range); range z writes its partial tile into slab z (wgrad_reduce_kernel adds the slabs up in a fixed order: deterministic,
// no atomics).  A 6-layer kitchen step has 75 output tiles for 256 CUs x 2 workgroup slots, each a latency chain over all
// 11,264 rows (176 stages): 417 us of the 2.26 ms step at 2 % of the matrix pipe.
template <typename E>
__global__ __launch_bounds__(kGT, kOcc) void tgemm_wgrad_group_kernel(GTable t, int n_tiles, int splits, float* slab,
                                                                      size_t slab_stride, int win, int n_win) {
    const int v = xcd_tile(blockIdx.x, gridDim.x);
    const int z = v / n_tiles, b = v - z * n_tiles;
    int pi = 0;
    while (pi + 1 < t.n && b >= t.p[pi + 1].tile_begin) ++pi;
    const GProb g = t.p[pi];
    const int local = b - g.tile_begin, tile_n = local % g.nt_n, tile_m = local / g.nt_n;
    __shared__ __attribute__((aligned(16))) TileLds lds;
    constexpr int KSTAGE = 128 / (int)sizeof(E);
    int k0 = 0, k1 = g.K;
    float* out = g.out; float* bias = g.bias;
    if (splits > 1) {
        const int kc = ((g.K + splits - 1) / splits + KSTAGE - 1) / KSTAGE * KSTAGE;
        k0 = min(z * kc, g.K); k1 = min(k0 + kc, g.K);
        out = slab + (size_t)z * slab_stride + g.slab_off;
        if (bias) bias = out + (size_t)g.Mo * g.No;
    }
    if (n_win > 1) {                                  // row window `win` of n_win (one launch per window)
        const int kc = ((g.K + n_win - 1) / n_win + KSTAGE - 1) / KSTAGE * KSTAGE;
        k0 = min(win * kc, g.K); k1 = min(k0 + kc, g.K);
    }
    tgemm_tile<E, true, true, EpiStoreF>(lds, (const E*)g.A, g.lda, (const E*)g.B, g.ldb, g.Mo, g.No, tile_m * kTileMN,
                                         tile_n * kTileMN, k0, k1, EpiStoreF{out, g.No, bias, n_win > 1 && win > 0});
}

// out (and bias) of every problem of a split launch = the sum of its slabs, z = 0 .. splits-1 in that order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(GTable t, const float* __restrict__ slab, size_t slab_stride,
                                                           int splits, uint32_t total) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int lo = 0, hi = t.n - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t.p[mid].slab_off <= i) lo = mid; else hi = mid - 1; }
        const GProb& g = t.p[lo];
        const uint32_t k = i - g.slab_off, no = (uint32_t)g.Mo * (uint32_t)g.No;
        float* dst = k < no ? g.out + k : ((g.bias && k < no + (uint32_t)g.Mo) ? g.bias + (k - no) : nullptr);
        if (!dst) continue;                                   // (alignment padding between two problems)
        float acc = 0.f;
        for (int z = 0; z < splits; ++z) acc += slab[(size_t)z * slab_stride + i];
        *dst = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Round 5: PANEL-OWNING weight-gradient tiles (bf16).  The 128 x 128 tiles above read every operand panel once per tile that
// shares it -- 36 tiles of an FC1 / FC2 weight gradient share 15 panels -- and rely on an XCD's 4 MB L2 to serve the
// sharers, which drift apart over the 176 ... 1408 stages of a contraction: FETCH_SIZE 1.78 GB for 0.78 GB of operands at
// 1024 kitchen samples (3.7 x at 8192), the launch bound by the fabric behind L2.  Every weight gradient of the network has
// one side of width D (or less): here a tile covers ALL of that side -- 128 x (128 W) or (128 W) x 128, W = ceil(D / 128)
// in {2, 3} -- so the LARGE operand (GELU(h), dh, dqkv: 4 D or 3 D wide) is read exactly once by exactly one workgroup, and
// only the D-wide panel (dyo, xn2, y, xn1) is shared, by all tiles of its problem at the same stage.  Kitchen: 36 tiles per
// layer, 218 per step = ONE round of one workgroup per CU (no second round to start out of step), a wave tile of 64 x 96
// (or 96 x 64): 10 fragment reads per 24 MFMAs instead of 6 per 8.  Same operand images in LDS as tgemm_tile (k-slow
// 16-column subtiles, ds_read_b64_tr_b16), same two-register-set pipeline, same epilogue and ones column.
// ---------------------------------------------------------------------------------------------
#ifndef BESO_WG_SETS
#define BESO_WG_SETS 1                     // register sets of the global -> LDS pipeline (A/B builds: 2)
#endif
template <int MT, int NT>
__device__ __forceinline__ void wgrad_panel_tile(unsigned char* lds, const uint16_t* __restrict__ A, int lda,
                                                 const uint16_t* __restrict__ B, int ldb, int M, int N, int m0, int n0,
                                                 int k_begin, int k_end, const EpiStoreF& epi) {
    typedef uint16_t E;
    static_assert(kNW == 8 && kGL == 2, "eight waves, two 16-byte chunks per thread and 128-column block");
    constexpr int KSTAGE = 64;
    constexpr int WM = MT >= NT ? 4 : 2, WN = 8 / WM;        // waves along m and n
    constexpr int MI = 8 * MT / WM, NI = 8 * NT / WN;        // 16-row MFMA tiles per wave: 4 x 6, 6 x 4 or 4 x 4
    constexpr int A_BYTES = 8 * MT * kSubBytes, STAGE = 8 * (MT + NT) * kSubBytes;      // one stage: A's subtiles, then B's
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int nk = (k_end - k_begin + KSTAGE - 1) / KSTAGE;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // One stage = two k-steps of 32.  FRAGMENT PIPELINE: the A fragment of row tile mi + 1 -- in the last group of the first k-step,
    // the B fragments and the first A fragment of the second k-step -- is requested BEFORE the MFMAs of row tile mi, so the matrix
    // pipe works on group mi while the LDS pipe fetches group mi + 1 (with the reads issued just in time the two pipes took
    // turns: LDS 1750 + MFMA 1540 clocks of a 3400-clock stage).  `head` issues the stage's first fragments; the caller puts the
    // global -> LDS traffic of the next stage between `head` and `body`.
    u32x4 bf[NI], af;
    auto head = [&](const unsigned char* st) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bf[ni] = op_frag<E, true>(st + A_BYTES, wn * NI + ni, 0, lane);
        af = op_frag<E, true>(st, wm * MI, 0, lane);
    };
    auto body = [&](const unsigned char* st) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                u32x4 an = af, bn[NI];
                if (mi + 1 < MI) an = op_frag<E, true>(st, wm * MI + mi + 1, s, lane);
                else if (s == 0) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) bn[ni] = op_frag<E, true>(st + A_BYTES, wn * NI + ni, 1, lane);
                    an = op_frag<E, true>(st, wm * MI, 1, lane);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mma16<E>(acc[mi][ni], bf[ni], af);      // D[n][m]: 4 consecutive n per lane
                af = an;
                if (mi + 1 == MI && s == 0) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) bf[ni] = bn[ni];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (nk > 0) {
        // Global side: one 16-byte chunk = (k-row, 8 columns); per-chunk byte offsets inside a stage are formed once, a column
        // outside the operand gets an offset beyond every range (reads zeros).  The k advance goes into the buffer descriptor:
        // base = first row of the stage, num_records = the bytes up to k_end -- so rows past the end of the contraction, whole
        // stages past it included, are zeros BY THE DESCRIPTOR'S RANGE CHECK: no predicates, no masked LDS stores, exact vmcnt.
        uint32_t va[MT][kGL], vb[NT][kGL];
#pragma unroll
        for (int i = 0; i < kGL; ++i) {
            const int c = tid + kGT * i, krow = c >> 4, cc = c & 15;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int gc = m0 + 128 * j + cc * 8;
                va[j][i] = gc < M ? (uint32_t)((krow * lda + gc) * 2) : 0x80000000u;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int gc = n0 + 128 * j + cc * 8;
                vb[j][i] = gc < N ? (uint32_t)((krow * ldb + gc) * 2) : 0x80000000u;
            }
        }
        uint32_t b_ones[NT];                               // chunks of this thread that start at column N of B (the ones column)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            b_ones[j] = 0u;
            if (epi.bias) {
#pragma unroll
                for (int i = 0; i < kGL; ++i) b_ones[j] |= (n0 + 128 * j + ((tid + kGT * i) & 15) * 8 == N ? 1u : 0u) << i;
            }
        }
        auto uniform_ptr = [](const void* p) {
            const uint64_t pv = (uint64_t)p;
            return (void*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pv >> 32)) << 32) |
                           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pv));
        };
        auto load = [&](int kt, u32x4 (&ra)[MT][kGL], u32x4 (&rb)[NT][kGL]) {
            const int k0 = k_begin + kt * KSTAGE;
            const int rows = k0 < k_end ? k_end - k0 : 0;
            const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(
                uniform_ptr(A + (rows ? (size_t)k0 * lda : 0)), 0, rows * lda * 2, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(
                uniform_ptr(B + (rows ? (size_t)k0 * ldb : 0)), 0, rows * ldb * 2, 0x00020000);
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < kGL; ++i) ra[j][i] = __builtin_amdgcn_raw_buffer_load_b128(rsa, va[j][i], 0, 0);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < kGL; ++i) rb[j][i] = __builtin_amdgcn_raw_buffer_load_b128(rsb, vb[j][i], 0, 0);
        };
        auto store = [&](unsigned char* st, const u32x4 (&ra)[MT][kGL], const u32x4 (&rb)[NT][kGL]) {
#pragma unroll
            for (int j = 0; j < MT; ++j) op_lstore<E, true, kGL>(st + 8 * j * kSubBytes, tid, ra[j]);
#pragma unroll
            for (int j = 0; j < NT; ++j) op_lstore<E, true, kGL>(st + A_BYTES + 8 * j * kSubBytes, tid, rb[j], b_ones[j]);
        };
#if BESO_WG_SETS == 2
        // two register sets, ping-pong (tgemm_tile's pipeline): two stages in flight
        u32x4 a0[MT][kGL], b0[NT][kGL], a1[MT][kGL], b1[NT][kGL];
        load(0, a0, b0);
        load(1, a1, b1);
        store(lds, a0, b0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            load(kt + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);          // the loads are issued HERE (the scheduler sinks them below the MFMAs)
            head(lds);
            store(lds + STAGE, a1, b1);
            body(lds);
            __syncthreads();
            // (no exit here when nk is odd: the stage past the end is zeros -- one wasted stage, but an exit in the middle of the
            //  loop makes the compiler keep TWO copies of the accumulators, 64 ... 96 VGPRs)
            load(kt + 3, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            head(lds + STAGE);
            store(lds, a0, b0);
            body(lds + STAGE);
            __syncthreads();
        }
#else
        // ONE register set (a stage is 128 B per thread): the stage that landed during the previous iteration's MFMAs is written
        // to the other LDS buffer, the stage after it requested, then the MFMAs of the current one run -- every load has a whole
        // compute phase to land
        u32x4 ar[MT][kGL], br[NT][kGL];
        load(0, ar, br);
        store(lds, ar, br);
        load(1, ar, br);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            head(lds + (kt & 1) * STAGE);
            __builtin_amdgcn_sched_barrier(0);
            store(lds + ((kt + 1) & 1) * STAGE, ar, br);      // (a stage past the end is zeros and never read)
            load(kt + 2, ar, br);
            __builtin_amdgcn_sched_barrier(0);          // the loads are issued HERE (the scheduler sinks them below the MFMAs)
            body(lds + (kt & 1) * STAGE);
            __syncthreads();
        }
#endif
    }
    int te = tid;
    asm volatile("" : "+v"(te));          // epilogue addresses are formed HERE, not hoisted above the k loop (spills)
    const int le = te & 63, wme = (te >> 6) / WN, wne = (te >> 6) % WN;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wme * (16 * MI) + mi * 16 + (le & 15);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wne * (16 * NI) + ni * 16 + (le >> 4) * 4;
            if (m < M && n < N) epi(m, n, acc[mi][ni]);
            if (m < M && n == N && epi.bias) epi.bias[m] = acc[mi][ni][0] + (epi.add ? epi.bias[m] : 0.f);      // the ones column
        }
    }
}

constexpr size_t wgrad_panel_lds(int W) { return (size_t)2 * 8 * (W + 1) * kSubBytes; }     // two stages of (1 + W) 128-column blocks

// The grouped launch on panel-owning tiles.  GProb::nt_n carries the orientation: 0 = the tile covers all No columns (128 rows
// of Mo per tile), -1 = it covers all Mo rows (128 columns of No per tile); tile_begin counts those tiles.
template <int W>
__global__ __launch_bounds__(kGT, 2) void wgrad_panel_group_kernel(GTable t, int n_tiles, int splits, float* slab,
                                                                  size_t slab_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char plds[];
    const int v = xcd_tile(blockIdx.x, gridDim.x);
    const int z = v / n_tiles, b = v - z * n_tiles;
    int pi = 0;
    while (pi + 1 < t.n && b >= t.p[pi + 1].tile_begin) ++pi;
    const GProb g = t.p[pi];
    const int local = b - g.tile_begin;
    int k0 = 0, k1 = g.K;
    float* out = g.out; float* bias = g.bias;
    if (splits > 1) {
        const int kc = ((g.K + splits - 1) / splits + 63) / 64 * 64;
        k0 = min(z * kc, g.K); k1 = min(k0 + kc, g.K);
        out = slab + (size_t)z * slab_stride + g.slab_off;
        if (bias) bias = out + (size_t)g.Mo * g.No;
    }
    const EpiStoreF epi{out, g.No, bias, false};
    if (g.nt_n == 0)
        wgrad_panel_tile<1, W>(plds, (const uint16_t*)g.A, g.lda, (const uint16_t*)g.B, g.ldb, g.Mo, g.No, local * kTileMN, 0, k0, k1, epi);
    else
        wgrad_panel_tile<W, 1>(plds, (const uint16_t*)g.A, g.lda, (const uint16_t*)g.B, g.ldb, g.Mo, g.No, 0, local * kTileMN, k0, k1, epi);
}

// W of the panel kernel for a model of width D (0: the 128 x 128 tiles stay)
static int wgrad_panel_w(int D, size_t elem_bytes) {
    if (elem_bytes != 2) return 0;
    return D > 128 && D <= 256 ? 2 : (D > 256 && D <= 384 ? 3 : 0);
}
// tiles of problem (Mo, No) on panel tiles of width W; *orient = 0 / -1 as GProb::nt_n (a problem with neither side within
// 128 W does not exist in this network: every weight gradient has a side of width <= D)
static int wgrad_panel_tiles(int Mo, int No, int W, int* orient) {
    if (No <= 128 * W) { *orient = 0; return (Mo + kTileMN - 1) / kTileMN; }
    *orient = -1;
    return (No + kTileMN - 1) / kTileMN;
}

// ---------------------------------------------------------------------------------------------
// operand-typed copies of one layer's weights (and the fused q|k|v bias) in one launch
// ---------------------------------------------------------------------------------------------
struct PackSeg { const float* src; void* dst; uint32_t first; uint32_t f32; };      // first: index of the segment's first float4 in the launch
constexpr int kPackSegs = 64;                          // 9 per layer: seven layers per launch (24 B each in the kernel argument)
struct PackTable { PackSeg seg[kPackSegs]; int n; };

template <typename E>
__global__ void pack_table_kernel(PackTable t, uint32_t total4) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
        int lo = 0, hi = t.n - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t.seg[mid].first <= i) lo = mid; else hi = mid - 1; }
        const PackSeg& g = t.seg[lo];
        const uint32_t k = i - g.first;
        const f32x4 v = *(const f32x4*)(g.src + 4 * (size_t)k);
        if (g.f32) *(f32x4*)((float*)g.dst + 4 * (size_t)k) = v;
        else Vec4<E>::store((E*)g.dst + 4 * (size_t)k, v);
    }
}

// ---------------------------------------------------------------------------------------------
// prep: noised = a + n*sigma; target = (a - c_skip*noised)/c_out            (score_wrappers.py:64-69)
// ---------------------------------------------------------------------------------------------
// ... and the step's two small initialisations (launches of their own before): the loss accumulator, and the head bias padded
// to the ap columns of the head GEMM.
__global__ void prep_kernel(const float* __restrict__ action, const float* __restrict__ noise,
                            const float* __restrict__ sigma, float* __restrict__ noised, float* __restrict__ target,
                            int per_sample, size_t n, float sigma_data, float* __restrict__ loss, float* __restrict__ b_head,
                            const float* __restrict__ head_bias, int act, int ap) {
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) *loss = 0.f;
        for (int c = threadIdx.x; c < ap; c += blockDim.x) b_head[c] = c < act ? head_bias[c] : 0.f;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float sg = sigma[i / per_sample];
        const float sd2 = sigma_data * sigma_data, den = sg * sg + sd2;
        const float c_skip = sd2 / den, c_out = sg * sigma_data / sqrtf(den);
        const float a = action[i], nz = a + noise[i] * sg;
        noised[i] = nz;
        target[i] = (a - c_skip * nz) / c_out;
    }
}

// ---------------------------------------------------------------------------------------------
// embed (training): x0 row (same arithmetic as embed_kernel of elementwise.hip) + the row of the feature
// matrix Xemb[M][Ke] whose transpose-product with dx0 yields every embedding gradient at once:
//   columns [0,obs) state/goal features | [obs,obs+act) c_in * noised action | obs+act: log(sigma)/4 |
//   +1,+2,+3: one-hot of the token kind (tok_emb.bias, action_emb.bias, sigma_emb.bias) |
//   then seq_size one-hot columns of the position row used (pos_emb).
// ---------------------------------------------------------------------------------------------
template <typename E>
__global__ void train_embed_kernel(const float* __restrict__ state, const float* __restrict__ action,
                                   const float* __restrict__ goal, const float* __restrict__ sigma,
                                   const float* __restrict__ pos, const float* __restrict__ tok_w,
                                   const float* __restrict__ tok_b, const float* __restrict__ sig_w,
                                   const float* __restrict__ sig_b, const float* __restrict__ act_w,
                                   const float* __restrict__ act_b, float* __restrict__ x, E* __restrict__ xemb,
                                   int t, int T, int G, int D, int obs, int act, int Ke, float sigma_data, float p_drop,
                                   float p_goal, uint32_t seed) {
    extern __shared__ float in_vec[];
    const int row = blockIdx.x, b = row / T, j = row % T;
    const float sg = sigma[b];
    int kind, len = 0, posrow = -1;
    const float* src = nullptr;
    float scale = 1.f;
    if (j == 0) {
        kind = 0;
    } else if (j <= G) {
        kind = 1; len = obs; posrow = j - 1; src = goal + ((size_t)b * G + (j - 1)) * obs;
    } else {
        const int idx = j - G - 1, i = idx >> 1;
        posrow = G + i;
        if ((idx & 1) == 0) { kind = 1; len = obs; src = state + ((size_t)b * t + i) * obs; }
        else {
            kind = 2; len = act; src = action + ((size_t)b * t + i) * act;
            scale = 1.0f / sqrtf(sg * sg + sigma_data * sigma_data);                      // c_in
        }
    }
    // training-mode goal masking (mask_cond, score_gpts.py:298-299) happens here: the goal tokens' inputs are zeroed
    // elementwise; the feature matrix xemb (operand of the embedding weight gradients) takes the masked values
    const bool goal_tok = j >= 1 && j <= G && p_goal > 0.f;
    for (int c = threadIdx.x; c < len; c += blockDim.x)
        in_vec[c] = src[c] * scale * (goal_tok ? goal_keep(seed, ((size_t)b * G + (j - 1)) * obs + c, p_goal) : 1.f);
    __syncthreads();
    const float lsg = logf(sg) / 4.0f;
    for (int c = threadIdx.x; c < Ke; c += blockDim.x) {
        float v = 0.f;
        if (c < obs) v = kind == 1 ? in_vec[c] : 0.f;
        else if (c < obs + act) v = kind == 2 ? in_vec[c - obs] : 0.f;
        else if (c == obs + act) v = kind == 0 ? lsg : 0.f;
        else if (c == obs + act + 1) v = kind == 1 ? 1.f : 0.f;
        else if (c == obs + act + 2) v = kind == 2 ? 1.f : 0.f;
        else if (c == obs + act + 3) v = kind == 0 ? 1.f : 0.f;
        else v = (c - (obs + act + 4)) == posrow ? 1.f : 0.f;
        xemb[(size_t)row * Ke + c] = Act<E>::from(v);
    }
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float v;
        if (kind == 0) {
            v = sig_w[d] * lsg + sig_b[d];
        } else {
            const float* w = (kind == 1 ? tok_w : act_w) + (size_t)d * len;
            float acc = 0.f;
            for (int c = 0; c < len; ++c) acc = fmaf(in_vec[c], w[c], acc);
            v = acc + (kind == 1 ? tok_b[d] : act_b[d]) + pos[(size_t)posrow * D + d];
            // self.drop(tok_emb(..) + pos) / self.drop(action_emb(..) + pos): score_gpts.py:321-325 (the sigma token has none)
            if (p_drop > 0.f) v *= drop_scale(seed, kEmbedSite, (size_t)row * D + d, p_drop, 1.0f / (1.0f - p_drop));
        }
        x[(size_t)row * D + d] = v;
    }
}

// The embedding as ONE GEMM (round 4): x0[M][D] = Xemb[M][Ke] Wcat[D][Ke]^T with the feature matrix below (the operand of the
// embedding weight gradients since round 1) and Wcat = [tok_emb.W | action_emb.W | sigma_emb.W | tok_emb.b | action_emb.b |
// sigma_emb.b | pos_emb^T] -- on the exact-fp32 MFMA (both operands fp32: fp32 products, fp32 accumulation), so the rows
// equal train_embed_kernel's to summation order.  One block per token row with a strided read of every weight took 48 us per
// 1024 kitchen samples; the feature kernel + the GEMM take 6 + 9.  (Embedding dropout keeps the old kernel: its mask is an
// elementwise epilogue this GEMM does not have.)
template <typename E>
__global__ __launch_bounds__(256) void train_feat_kernel(const float* __restrict__ state, const float* __restrict__ action,
                                                         const float* __restrict__ goal, const float* __restrict__ sigma,
                                                         E* __restrict__ xemb, float* __restrict__ xemb32, int M, int t, int T, int G,
                                                         int obs, int act, int Ke, float sigma_data, float p_goal, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * Ke; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / Ke), c = (int)(i % Ke), b = row / T, j = row % T;
        int kind, posrow = -1;
        const float* src = nullptr;
        if (j == 0) kind = 0;
        else if (j <= G) { kind = 1; posrow = j - 1; src = goal + ((size_t)b * G + (j - 1)) * obs; }
        else {
            const int idx = j - G - 1, k = idx >> 1;
            posrow = G + k;
            if ((idx & 1) == 0) { kind = 1; src = state + ((size_t)b * t + k) * obs; }
            else { kind = 2; src = action + ((size_t)b * t + k) * act; }
        }
        float v = 0.f;
        if (c < obs) {
            if (kind == 1) {
                v = src[c];
                if (j <= G && p_goal > 0.f) v *= goal_keep(seed, ((size_t)b * G + (j - 1)) * obs + c, p_goal);     // mask_cond
            }
        } else if (c < obs + act) {
            if (kind == 2) { const float sg = sigma[b]; v = src[c - obs] * (1.0f / sqrtf(sg * sg + sigma_data * sigma_data)); }     // c_in
        } else if (c == obs + act) v = kind == 0 ? logf(sigma[b]) / 4.0f : 0.f;
        else if (c == obs + act + 1) v = kind == 1 ? 1.f : 0.f;
        else if (c == obs + act + 2) v = kind == 2 ? 1.f : 0.f;
        else if (c == obs + act + 3) v = kind == 0 ? 1.f : 0.f;
        else v = (c - (obs + act + 4)) == posrow ? 1.f : 0.f;
        xemb[i] = Act<E>::from(v);
        if ((const void*)xemb32 != (const void*)xemb) xemb32[i] = v;
    }
}

__global__ void wcat_pack_kernel(const float* __restrict__ pos, const float* __restrict__ tok_w, const float* __restrict__ tok_b,
                                 const float* __restrict__ sig_w, const float* __restrict__ sig_b, const float* __restrict__ act_w,
                                 const float* __restrict__ act_b, float* __restrict__ wcat, int D, int obs, int act, int seq_size,
                                 int Ke) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D * Ke; i += gridDim.x * blockDim.x) {
        const int d = i / Ke, c = i % Ke;
        float v = 0.f;
        if (c < obs) v = tok_w[(size_t)d * obs + c];
        else if (c < obs + act) v = act_w[(size_t)d * act + (c - obs)];
        else if (c == obs + act) v = sig_w[d];
        else if (c == obs + act + 1) v = tok_b[d];
        else if (c == obs + act + 2) v = act_b[d];
        else if (c == obs + act + 3) v = sig_b[d];
        else if (c - (obs + act + 4) < seq_size) v = pos[(size_t)(c - (obs + act + 4)) * D + d];
        wcat[i] = v;
    }
}

// dWcat[Ke][D] -> the individual embedding gradients (accumulated: the flat buffer was zeroed)
__global__ void scatter_emb_kernel(const float* __restrict__ dw, float* __restrict__ g_pos, float* __restrict__ g_tokw,
                                   float* __restrict__ g_tokb, float* __restrict__ g_sigw, float* __restrict__ g_sigb,
                                   float* __restrict__ g_actw, float* __restrict__ g_actb, int D, int obs, int act,
                                   int seq_size) {
    const int n = (obs + act + 4 + seq_size) * D;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c = i / D, d = i % D;
        const float v = dw[i];
        if (c < obs) g_tokw[(size_t)d * obs + c] += v;
        else if (c < obs + act) g_actw[(size_t)d * act + (c - obs)] += v;
        else if (c == obs + act) g_sigw[d] += v;
        else if (c == obs + act + 1) g_tokb[d] += v;
        else if (c == obs + act + 2) g_actb[d] += v;
        else if (c == obs + act + 3) g_sigb[d] += v;
        else g_pos[(size_t)(c - (obs + act + 4)) * D + d] += v;
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm forward with saved statistics; one wave per row, the row in registers as float4 per lane
// (D % 4 == 0, D <= 1024).
// ---------------------------------------------------------------------------------------------
// NV = float4 slots per lane = ceil(D / 256): 1, 2 or 4

template <typename E, int kLnVec>
__global__ void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                              E* __restrict__ out, float* __restrict__ stats, int rows, int D) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    f32x4 v[kLnVec];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kLnVec; ++i) {
        const int c = (lane + i * 64) * 4;
        v[i] = c < D ? *(const f32x4*)(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < kLnVec; ++i) {
        const int c = (lane + i * 64) * 4;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; sq = fmaf(d, d, sq); }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + 1e-5f);
    if (lane == 0) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
    E* o = out + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < kLnVec; ++i) {
        const int c = (lane + i * 64) * 4;
        if (c < D) {
            const f32x4 wv = *(const f32x4*)(w + c), bv = *(const f32x4*)(b + c);
            Vec4<E>::store(o + c, (v[i] - mean) * rstd * wv + bv);
        }
    }
}

// LayerNorm backward + residual gradient:  dres <- dres + dLN(dxn)  (dres_in == nullptr: no incoming residual
// gradient), operand-typed copy dxb = dres * keep-scale(site) for the linear layer that consumes it, that layer's
// bias gradient (column sums of dxb; dbias may be nullptr) and the affine gradients -- register partials per
// wave -> LDS -> one atomic per block and feature.
template <typename E, int kLnVec>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dxn, const float* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     const float* dres_in, float* dres_out, E* __restrict__ dxb,
                                                     float* __restrict__ part, int rows, int D, int rows_per_wave, float p,
                                                     float inv_keep, uint32_t seed, uint32_t site, int skip_mod) {
    __shared__ f32x4 red[3][4][64 * kLnVec];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + wid;
    f32x4 gw[kLnVec], ag[kLnVec], ab[kLnVec], ac[kLnVec];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kLnVec; ++i) {
        const int c = (lane + i * 64) * 4;
        gw[i] = c < D ? *(const f32x4*)(gamma + c) : zero;
        ag[i] = zero; ab[i] = zero; ac[i] = zero;
    }
    const int r_begin = wave * rows_per_wave, r_end = min(rows, r_begin + rows_per_wave);
    // TWO rows per iteration: the kernel is bound by the latency chain load -> two wave reductions -> store of a row, not by
    // bytes (bf16 instead of fp32 for dxn changed nothing); with two rows in flight the four reductions interleave
    constexpr int U = kLnVec == 4 ? 1 : 2;         // (D > 512: the second row would not fit the registers of three waves per SIMD)
    for (int row0 = r_begin; row0 < r_end; row0 += U) {
        f32x4 xh[U][kLnVec], dy[U][kLnVec], go[U][kLnVec], xv[U][kLnVec];
        float mean[U], rstd[U], s1[U], s2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = min(row0 + u, r_end - 1);        // (a surplus second row re-reads the first; nothing of it is stored)
            mean[u] = stats[2 * (size_t)row]; rstd[u] = stats[2 * (size_t)row + 1];
#pragma unroll
            for (int i = 0; i < kLnVec; ++i) {
                const int c = (lane + i * 64) * 4;
                go[u][i] = c < D ? *(const f32x4*)(dxn + (size_t)row * D + c) : zero;
                xv[u][i] = c < D ? *(const f32x4*)(x + (size_t)row * D + c) : zero;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = row0 + u < r_end;
            s1[u] = 0.f; s2[u] = 0.f;
#pragma unroll
            for (int i = 0; i < kLnVec; ++i) {
                const int c = (lane + i * 64) * 4;
                xh[u][i] = c < D ? (xv[u][i] - mean[u]) * rstd[u] : zero;
                dy[u][i] = go[u][i] * gw[i];
                const f32x4 t = dy[u][i] * xh[u][i];
                s1[u] += (dy[u][i][0] + dy[u][i][1]) + (dy[u][i][2] + dy[u][i][3]);
                s2[u] += (t[0] + t[1]) + (t[2] + t[3]);
                if (live) { ag[i] += go[u][i] * xh[u][i]; ab[i] += go[u][i]; }
            }
        }
        float c1[U], c2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { c1[u] = s1[u]; c2[u] = s2[u]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { c1[u] = wave_sum_dpp(c1[u]); c2[u] = wave_sum_dpp(c2[u]); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = row0 + u;
            if (row >= r_end) break;
            const float m1 = c1[u] / (float)D, m2 = c2[u] / (float)D;
#pragma unroll
            for (int i = 0; i < kLnVec; ++i) {
                const int c = (lane + i * 64) * 4;
                if (c < D) {
                    const size_t idx = (size_t)row * D + c;
                    f32x4 tot = (dy[u][i] - m1 - xh[u][i] * m2) * rstd[u];
                    if (dres_in) tot += *(const f32x4*)(dres_in + idx);
                    *(f32x4*)(dres_out + idx) = tot;
                    f32x4 op = tot;
                    if (p > 0.f && !(skip_mod > 0 && row % skip_mod == 0)) {      // (the sigma token's embedding has no dropout)
#pragma unroll
                        for (int j = 0; j < 4; ++j) op[j] *= drop_scale(seed, site, idx + j, p, inv_keep);
                    }
                    Vec4<E>::store(dxb + idx, op);
                    ac[i] += op;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kLnVec; ++i) {
        red[0][wid][lane + i * 64] = ag[i]; red[1][wid][lane + i * 64] = ab[i]; red[2][wid][lane + i * 64] = ac[i];
    }
    __syncthreads();
    // block partials [3][D] -> part[blockIdx.x]; ln_reduce_kernel sums them for all LayerNorms of the step at once
    // (fp32 atomics from every block were the larger half of this kernel's time)
    const float* r0 = (const float*)red[0], *r1 = (const float*)red[1], *r2 = (const float*)red[2];
    constexpr int WS = 64 * kLnVec * 4;          // floats per wave slab
    float* o = part + (size_t)blockIdx.x * 3 * D;
    for (int c = threadIdx.x; c < D; c += 256) {
        o[c] = r0[c] + r0[WS + c] + r0[2 * WS + c] + r0[3 * WS + c];
        o[D + c] = r1[c] + r1[WS + c] + r1[2 * WS + c] + r1[3 * WS + c];
        o[2 * D + c] = r2[c] + r2[WS + c] + r2[2 * WS + c] + r2[3 * WS + c];
    }
}

// sums the block partials of every LayerNorm backward of the step: call z -> gamma / beta gradient and the bias
// gradient of the linear layer that consumed its output (assigned: each destination has exactly one source)
struct LnRedCall { float* dgamma; float* dbeta; float* dbias; };
struct LnRedTable { LnRedCall c[2 * kMaxLayers + 1]; int nb[2 * kMaxLayers + 1]; };    // nb: block partials call z wrote (<= nblk, the slab stride)
__global__ __launch_bounds__(256) void ln_reduce_kernel(const float* __restrict__ part, LnRedTable t, int nblk, int D,
                                                        int call0) {
    // thread (cx, ry): four consecutive columns (one 16-byte load: 16 lanes = 256 contiguous bytes of a slab row), block partials
    // ry, ry + 16, ... with EIGHT loads in flight.  Round 5's form -- one column per thread, four row groups, four loads in
    // flight -- was nblk / 16 dependent HBM round trips per thread: 375 us at 8192 samples (1878 partials per LayerNorm, 105 MB
    // of slabs) where the bytes cost ~25 us.  The sum order is fixed (row groups in order, then the 16 groups in order):
    // deterministic.  D is a multiple of 4 (the training step requires a multiple of 8).
    __shared__ f32x4 red[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4, which = blockIdx.y, call = blockIdx.z + call0;
    const int c = blockIdx.x * 64 + 4 * cx;
    const float* src = part + ((size_t)call * nblk * 3 + which) * D + c;
    nblk = t.nb[call];
    const size_t rs = (size_t)3 * D;
    f32x4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < D) {
        int b = ry;
        for (; b + 112 < nblk; b += 128) {
            // (the eight loads first, into registers of their own: written as a[u] += load the compiler issued them one at a
            //  time -- load, s_waitcnt vmcnt(0), add -- i.e. nblk / 16 dependent round trips again: 326 us at 8192 samples)
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((const f32x4*)(src + (size_t)(b + 16 * u) * rs));
            __builtin_amdgcn_sched_barrier(0);          // (nor may the scheduler fold the adds back between the loads)
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += v[u];
        }
        for (; b < nblk; b += 16) a[0] += *(const f32x4*)(src + (size_t)b * rs);
    }
    red[ry][cx] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (ry == 0 && c < D) {
        const LnRedCall k = t.c[call];
        float* dst = which == 0 ? k.dgamma : (which == 1 ? k.dbeta : k.dbias);
        if (dst) {
            f32x4 v = red[0][cx];
#pragma unroll
            for (int r = 1; r < 16; ++r) v += red[r][cx];
            *(f32x4*)(dst + c) = v;
        }
    }
}

// column sums (bias gradients): out_j[n - j*seg] += sum_m a[m][n] for n in segment j (q | k | v share one pass).
// Thread (cx, ry): 16-byte chunk cx of the row, rows ry, ry+4, ...; a wave reads 1 KB of a row at a time.
template <typename E>
__global__ __launch_bounds__(256) void colsum_kernel(const E* __restrict__ a, int ld, int rows, int cols, float* out0,
                                                     float* out1, float* out2, int seg, int rows_per_block) {
    constexpr int EPC = 16 / (int)sizeof(E);
    __shared__ float red[4][64][EPC];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + cx) * EPC;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float acc[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) acc[j] = 0.f;
    if (c < cols) {
        for (int r = r0 + ry; r < r1; r += 4) {
            const u32x4 u = *(const u32x4*)(a + (size_t)r * ld + c);
            if (sizeof(E) == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * j] += __uint_as_float(u[j] << 16);
                    acc[2 * j + 1] += __uint_as_float(u[j] & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += __uint_as_float(u[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < EPC; ++j) red[ry][cx][j] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * EPC; i += 256) {
        const int n = blockIdx.x * 64 * EPC + i;
        if (n < cols) {
            const int x = i / EPC, j = i % EPC;
            const float v = red[0][x][j] + red[1][x][j] + red[2][x][j] + red[3][x][j];
            const int sj = n / seg;
            float* o = sj == 0 ? out0 : (sj == 1 ? out1 : out2);
            unsafeAtomicAdd(o + (n - sj * seg), v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// attention of one (sample, head) per wave, forward (training: dropout on the probabilities, score_gpts.py:69-76)
// and backward.  q/k/v rows of the head in LDS as fp32; T x T scores in LDS.  qkv row layout [q | k | v].
// ---------------------------------------------------------------------------------------------
struct AttnLds { float *q, *k, *v, *dy, *P, *dP; int ldh, ldt; };
__device__ __forceinline__ AttnLds attn_carve(float* base, int T, int hd, bool bwd) {
    AttnLds a;
    a.ldh = hd + 1; a.ldt = T + 1;
    a.q = base; a.k = a.q + T * a.ldh; a.v = a.k + T * a.ldh;
    a.dy = a.v + T * a.ldh;
    a.P = a.dy + (bwd ? T * a.ldh : 0);
    a.dP = a.P + T * a.ldt;
    return a;
}
size_t attn_lds_bytes(int T, int hd, bool bwd) {
    return sizeof(float) * ((size_t)(bwd ? 4 : 3) * T * (hd + 1) + (size_t)(bwd ? 2 : 1) * T * (T + 1));
}

// P (pre-dropout probabilities) for all rows; lanes stride over rows
template <typename E>
__device__ __forceinline__ void attn_load_scores(const E* __restrict__ qkv, const AttnLds& a, int b, int h, int T,
                                                 int D, int hd, float scale, int lane) {
    const size_t ldq = (size_t)3 * D;
    // four elements per lane in flight (12 loads issued before the first is used; clamped index, no branch)
    const int n_el = T * hd;
    for (int u0 = lane; u0 < n_el; u0 += 256) {
        E qv[4], kv[4], vv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = min(u0 + 64 * j, n_el - 1), r = u / hd, d = u - r * hd;
            const E* src = qkv + ((size_t)b * T + r) * ldq + (size_t)h * hd + d;
            qv[j] = src[0]; kv[j] = src[D]; vv[j] = src[2 * D];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = u0 + 64 * j;
            if (u < n_el) {
                const int r = u / hd, d = u - r * hd;
                a.q[r * a.ldh + d] = Act<E>::to(qv[j]);
                a.k[r * a.ldh + d] = Act<E>::to(kv[j]);
                a.v[r * a.ldh + d] = Act<E>::to(vv[j]);
            }
        }
    }
    __syncthreads();
    for (int u = lane; u < T * T; u += 64) {
        const int i = u / T, j = u % T;
        float s = -INFINITY;
        if (j <= i) {
            s = 0.f;
            for (int d = 0; d < hd; ++d) s = fmaf(a.q[i * a.ldh + d], a.k[j * a.ldh + d], s);
            s *= scale;
        }
        a.P[i * a.ldt + j] = s;
    }
    __syncthreads();
    for (int i = lane; i < T; i += 64) {
        float m = -INFINITY;
        for (int j = 0; j <= i; ++j) m = fmaxf(m, a.P[i * a.ldt + j]);
        float l = 0.f;
        for (int j = 0; j <= i; ++j) { const float e = expf(a.P[i * a.ldt + j] - m); a.P[i * a.ldt + j] = e; l += e; }
        const float inv = 1.0f / l;
        for (int j = 0; j < T; ++j) a.P[i * a.ldt + j] = j <= i ? a.P[i * a.ldt + j] * inv : 0.f;
    }
    __syncthreads();
}

template <typename E>
__global__ __launch_bounds__(64) void attn_fwd_kernel(const E* __restrict__ qkv, E* __restrict__ y, int T, int D, int H,
                                                      int hd, float scale, float p, float inv_keep, uint32_t seed,
                                                      uint32_t site) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int pair = blockIdx.x, b = pair / H, h = pair % H, lane = threadIdx.x;
    const AttnLds a = attn_carve(smem_f, T, hd, false);
    attn_load_scores<E>(qkv, a, b, h, T, D, hd, scale, lane);
    if (p > 0.f) {
        for (int u = lane; u < T * T; u += 64) {
            const int i = u / T, j = u % T;
            a.P[i * a.ldt + j] *= drop_scale(seed, site, (size_t)pair * T * T + u, p, inv_keep);
        }
        __syncthreads();
    }
    for (int u = lane; u < T * hd; u += 64) {
        const int i = u / hd, d = u % hd;
        float o = 0.f;
        for (int j = 0; j <= i; ++j) o = fmaf(a.P[i * a.ldt + j], a.v[j * a.ldh + d], o);
        y[((size_t)b * T + i) * D + (size_t)h * hd + d] = Act<E>::from(o);
    }
}

template <typename E>
__global__ __launch_bounds__(64) void attn_bwd_kernel(const E* __restrict__ qkv, const E* __restrict__ dy,
                                                      E* __restrict__ dqkv, int T, int D, int H, int hd, float scale,
                                                      float p, float inv_keep, uint32_t seed, uint32_t site) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int pair = blockIdx.x, b = pair / H, h = pair % H, lane = threadIdx.x;
    const AttnLds a = attn_carve(smem_f, T, hd, true);
    attn_load_scores<E>(qkv, a, b, h, T, D, hd, scale, lane);
    for (int u0 = lane; u0 < T * hd; u0 += 256) {
        E gv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = min(u0 + 64 * j, T * hd - 1), r = u / hd, d = u - r * hd;
            gv[j] = dy[((size_t)b * T + r) * D + (size_t)h * hd + d];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = u0 + 64 * j;
            if (u < T * hd) { const int r = u / hd, d = u - r * hd; a.dy[r * a.ldh + d] = Act<E>::to(gv[j]); }
        }
    }
    __syncthreads();
    // dP[i][j] = keep-scale * sum_d dy[i][d] v[j][d]   (gradient w.r.t. the pre-dropout probability)
    for (int u = lane; u < T * T; u += 64) {
        const int i = u / T, j = u % T;
        float g = 0.f;
        if (j <= i) {
            for (int d = 0; d < hd; ++d) g = fmaf(a.dy[i * a.ldh + d], a.v[j * a.ldh + d], g);
            if (p > 0.f) g *= drop_scale(seed, site, (size_t)pair * T * T + u, p, inv_keep);
        }
        a.dP[i * a.ldt + j] = g;
    }
    __syncthreads();
    const size_t ldq = (size_t)3 * D;
    // dv[j][d] = sum_{i>=j} Pd[i][j] dy[i][d]  (Pd = P * keep-scale) -- before dP is turned into dS
    for (int u = lane; u < T * hd; u += 64) {
        const int j = u / hd, d = u % hd;
        float g = 0.f;
        for (int i = j; i < T; ++i) {
            float pd = a.P[i * a.ldt + j];
            if (p > 0.f) pd *= drop_scale(seed, site, (size_t)pair * T * T + (size_t)i * T + j, p, inv_keep);
            g = fmaf(pd, a.dy[i * a.ldh + d], g);
        }
        dqkv[((size_t)b * T + j) * ldq + 2 * (size_t)D + (size_t)h * hd + d] = Act<E>::from(g);
    }
    __syncthreads();
    // dS[i][j] = P[i][j] * (dP[i][j] - sum_j' dP[i][j'] P[i][j'])
    for (int i = lane; i < T; i += 64) {
        float dot = 0.f;
        for (int j = 0; j <= i; ++j) dot = fmaf(a.dP[i * a.ldt + j], a.P[i * a.ldt + j], dot);
        for (int j = 0; j < T; ++j)
            a.dP[i * a.ldt + j] = j <= i ? a.P[i * a.ldt + j] * (a.dP[i * a.ldt + j] - dot) * scale : 0.f;
    }
    __syncthreads();
    for (int u = lane; u < T * hd; u += 64) {
        const int r = u / hd, d = u % hd;
        float gq = 0.f, gk = 0.f;
        for (int j = 0; j <= r; ++j) gq = fmaf(a.dP[r * a.ldt + j], a.k[j * a.ldh + d], gq);
        for (int i = r; i < T; ++i) gk = fmaf(a.dP[i * a.ldt + r], a.q[i * a.ldh + d], gk);
        E* dst = dqkv + ((size_t)b * T + r) * ldq + (size_t)h * hd + d;
        dst[0] = Act<E>::from(gq);
        dst[D] = Act<E>::from(gk);
    }
}

// ---------------------------------------------------------------------------------------------
// Short sequences (T <= 16, hd <= 64, hd % 4 == 0: kitchen T = 11 / hd = 60, block-push T = 12 / hd = 20): the
// same attention, forward or backward, with lane d holding column d of q / k / v (/ dy) in registers.  All global
// loads are issued before the first use; the T x T parts run on a (4 rows x 16 columns) lane grid with float4
// dot products from LDS (row stride 68 floats: conflict-free b128 reads), the d-parallel parts read the T x T
// matrices as broadcasts.
// ---------------------------------------------------------------------------------------------
constexpr int kTP = 16, kLdh = 68, kLdt = 20;

__device__ __forceinline__ float dot_rows(const float* a, const float* b, int hd) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < hd; d += 4) acc += *(const f32x4*)(a + d) * *(const f32x4*)(b + d);
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// all-reduce over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15): every lane receives the row's sum / maximum
template <bool IS_MAX>
__device__ __forceinline__ float row16_allreduce(float v) {
    auto dpp = [](float x, auto CTRL) {
        return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), decltype(CTRL)::value, 0xf, 0xf, false));
    };
    auto op = [](float a, float b) { return IS_MAX ? fmaxf(a, b) : a + b; };
    v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));       // quad_perm [1,0,3,2]
    v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));       // quad_perm [2,3,0,1]
    v = op(v, dpp(v, std::integral_constant<int, 0x124>{}));      // row_ror:4
    v = op(v, dpp(v, std::integral_constant<int, 0x128>{}));      // row_ror:8
    return v;
}

template <typename E, bool BWD, int TT>          // TT = T rounded up to a multiple of 4 (8, 12 or 16)
__global__ __launch_bounds__(64) void attn_small_kernel(const E* __restrict__ qkv, const E* __restrict__ dy,
                                                        E* __restrict__ out, int T, int D, int H, int hd, float scale,
                                                        float p, float inv_keep, uint32_t seed, uint32_t site) {
    // q | k rows for the scores; the backward then re-uses the two buffers for dy | v (round 4: 15 -> 9.4 KiB per wave, i.e.
    // 16 instead of 10 waves per CU -- the kernel is a latency chain per (sample, head), so waves in flight are its rate)
    __shared__ __attribute__((aligned(16))) float sq[TT][kLdh], sk[TT][kLdh];
    __shared__ __attribute__((aligned(16))) float sP[TT][kLdt], sdP[BWD ? TT : 1][kLdt];
    const int pair = blockIdx.x, b = pair / H, h = pair % H, lane = threadIdx.x;
    const size_t ldq = (size_t)3 * D;
    const bool act = lane < hd;
    float qr[TT], kr[TT], vr[TT], gr[TT];
    {
        // branch-free: out-of-range rows / lanes read a clamped address and are zeroed afterwards (a load inside a
        // divergent branch is waited for before the next one is issued: 48 serial round trips)
        const int lc = act ? lane : 0;
        const E* base = qkv + (size_t)b * T * ldq + (size_t)h * hd + lc;
        const E* gbase = dy + (size_t)b * T * D + (size_t)h * hd + lc;
        E qe[TT], ke[TT], ve[TT], ge[TT];
#pragma unroll
        for (int r = 0; r < TT; ++r) {
            const int rc = r < T ? r : 0;
            qe[r] = base[rc * ldq];
            ke[r] = base[rc * ldq + D];
            ve[r] = base[rc * ldq + 2 * D];
            if (BWD) ge[r] = gbase[(size_t)rc * D];
        }
#pragma unroll
        for (int r = 0; r < TT; ++r) {
            const bool ok = act && r < T;
            qr[r] = ok ? Act<E>::to(qe[r]) : 0.f;
            kr[r] = ok ? Act<E>::to(ke[r]) : 0.f;
            vr[r] = ok ? Act<E>::to(ve[r]) : 0.f;
            gr[r] = (BWD && ok) ? Act<E>::to(ge[r]) : 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < TT; ++r) { sq[r][lane] = qr[r]; sk[r][lane] = kr[r]; }
    __syncthreads();
    // scores on the 4 x 16 lane grid: lane (il, j) holds S[4 ib + il][j] for ib = 0 .. TT/4 - 1
    const int il = lane >> 4, j = lane & 15;
    float sc[TT / 4], dpd[TT / 4];
#pragma unroll
    for (int ib = 0; ib < TT / 4; ++ib) {
        const int i = ib * 4 + il;
        const bool in = i < T && j <= i;
        sc[ib] = in ? dot_rows(sq[i], sk[j < TT ? j : 0], hd) * scale : -INFINITY;
    }
    if (BWD) {
        __syncthreads();                              // every score is read: dy | v take the buffers' place
#pragma unroll
        for (int r = 0; r < TT; ++r) { sq[r][lane] = gr[r]; sk[r][lane] = vr[r]; }
        __syncthreads();
#pragma unroll
        for (int ib = 0; ib < TT / 4; ++ib) {
            const int i = ib * 4 + il;
            dpd[ib] = (i < T && j <= i) ? dot_rows(sq[i], sk[j < TT ? j : 0], hd) : 0.f;      // dPd = dy v^T
        }
    }
    // row softmax IN the lane grid (a row's 16 columns are the 16 lanes of a DPP row): max, exp, sum, the dropout keep-scale
    // of the lane's own element and -- backward -- dS, all lanes busy (round 3: T lanes, each a serial loop over its row)
#pragma unroll
    for (int ib = 0; ib < TT / 4; ++ib) {
        const int i = ib * 4 + il;
        const bool in = i < T && j <= i;
        const float m = row16_allreduce<true>(sc[ib]);
        const float e = in ? expf(sc[ib] - m) : 0.f;
        const float l = row16_allreduce<false>(e);
        const float pr = in ? e * (1.0f / l) : 0.f;
        const float ks = (p > 0.f && in) ? drop_scale(seed, site, (size_t)pair * T * T + (size_t)i * T + j, p, inv_keep) : 1.f;
        if (BWD) {
            const float dot = row16_allreduce<false>(dpd[ib] * ks * pr);
            if (i < TT && j < kLdt) sdP[i][j] = pr * (dpd[ib] * ks - dot) * scale;       // dS
        }
        if (i < TT && j < kLdt) sP[i][j] = pr * ks;                                       // Pd
    }
    __syncthreads();
    if (!act) return;
    // the d-parallel products walk the lower triangle only: row i needs columns 0 .. i (in groups of four)
    if (!BWD) {
        E* o = out + (size_t)b * T * D + (size_t)h * hd + lane;
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            if (i < T) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c <= (i | 3) && c < TT; c += 4) {
                    const f32x4 pv = *(const f32x4*)&sP[i][c];
                    acc = fmaf(pv[0], vr[c], acc); acc = fmaf(pv[1], vr[c + 1], acc);
                    acc = fmaf(pv[2], vr[c + 2], acc); acc = fmaf(pv[3], vr[c + 3], acc);
                }
                o[(size_t)i * D] = Act<E>::from(acc);
            }
        }
    } else {
        // dq[i] = sum_j dS[i][j] k[j];  dk[j] = sum_i dS[i][j] q[i];  dv[j] = sum_i Pd[i][j] dy[i]
        float dq[TT], dk[TT], dv[TT];
#pragma unroll
        for (int r = 0; r < TT; ++r) { dq[r] = 0.f; dk[r] = 0.f; dv[r] = 0.f; }
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            if (i < T) {
#pragma unroll
                for (int c = 0; c <= (i | 3) && c < TT; c += 4) {
                    const f32x4 ds = *(const f32x4*)&sdP[i][c], pd = *(const f32x4*)&sP[i][c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dq[i] = fmaf(ds[e], kr[c + e], dq[i]);
                        dk[c + e] = fmaf(ds[e], qr[i], dk[c + e]);
                        dv[c + e] = fmaf(pd[e], gr[i], dv[c + e]);
                    }
                }
            }
        }
        E* o = out + (size_t)b * T * ldq + (size_t)h * hd + lane;
#pragma unroll
        for (int r = 0; r < TT; ++r) {
            if (r < T) {
                o[r * ldq] = Act<E>::from(dq[r]);
                o[r * ldq + D] = Act<E>::from(dk[r]);
                o[r * ldq + 2 * D] = Act<E>::from(dv[r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same backward on the matrix pipe (bf16 operands, T <= 16, hd <= 64, hd % 4 == 0; round 4).  The VALU kernel above is a
// latency chain of ~29 k cycles per (sample, head): hundreds of dependent LDS round trips for 72 kFLOP.  Here a wave's products
// are 20 MFMAs and everything between them stays in registers:
//   S^T = K Q^T and S = Q K^T, dPd^T = V dY^T and dPd = dY V^T      8 x v_mfma_f32_16x16x32_bf16: both operands of every one are
//        row chunks (lane (x, g): columns 32 kk + 8 g .. + 7 of row x), loaded from global in fragment order, used as A or B
//   the two layouts of the T x T matrices -- "T": lane (i, g) holds keys j = 4 g + r; "N": lane (j, g) holds queries i = 4 g + r
//        -- are what the second stage needs as B operands as they stand (k index 4 g + r), so softmax, dropout and dS are
//        evaluated in both (T: reductions over r and across the four lane rows; N: over the 16 lanes of a DPP row)
//   dq^T = K^T dS^T, dk^T = Q^T dS, dv^T = dY^T Pd                  12 x v_mfma_f32_16x16x16_bf16: A = the transposed operand from a
//        row-major LDS copy (ds_read_b64_tr_b16), D: lane (token, g) holds dims 16 dt + 4 g .. + 3 = one 8-byte store
// Same dropout mask as attn_small_kernel (hash of (pair T + i) T + j).  Four (sample, head) pairs per workgroup, no barriers.
// ---------------------------------------------------------------------------------------------
constexpr int kAmLd = 72;                          // halfwords per LDS row: 64 dims + pad (144 B: the transpose reads spread over banks)

__device__ __forceinline__ u32x4 row_chunk16(const uint16_t* __restrict__ base, size_t ld, int x, int c0, int T, int hd) {
    // ONE 16-byte request per chunk (a head's rows are 8-byte aligned when hd is not a multiple of 8: global memory takes the
    // unaligned dwordx4).  A chunk that straddles the end of the head (hd = 60: columns 56 .. 63) is fetched as the head's last
    // eight columns and shifted down, so nothing behind the head -- or the tensor -- is read.  hd >= 8, a multiple of 4.
    typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
    const uint16_t* p = base + (size_t)min(x, T - 1) * ld;
    const int cl = min(c0, hd - 8);
    const u32x4_a8 v = *(const u32x4_a8*)(p + cl);
    // (selects, not branches: a branch between the loads makes each of them wait for the one before)
    const bool shifted = cl != c0, half = c0 - cl == 4;               // (c0 - cl is 0, 4 or >= 8)
    const bool lo_on = x < T && c0 < hd, hi_on = x < T && c0 + 4 < hd;
    const uint32_t l0 = shifted ? (half ? v[2] : 0u) : v[0], l1 = shifted ? (half ? v[3] : 0u) : v[1];
    const uint32_t h0 = shifted ? 0u : v[2], h1 = shifted ? 0u : v[3];
    return u32x4{lo_on ? l0 : 0u, lo_on ? l1 : 0u, hi_on ? h0 : 0u, hi_on ? h1 : 0u};
}
__device__ __forceinline__ uint2 pack4_bf16(float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(2))) float f2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
    const f2_t v0 = {a, b}, v1 = {c, d};                 // v_cvt_pk_bf16_f32 (RNE)
    return make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(v0, b2_t)),
                      __builtin_bit_cast(uint32_t, __builtin_convertvector(v1, b2_t)));
}
// sum / maximum over the four lane rows (lanes x, x + 16, x + 32, x + 48), in every lane
template <bool IS_MAX>
__device__ __forceinline__ float cross_rows_allreduce(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
    const u32x2_t a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float r = IS_MAX ? fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])) : __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const u32x2_t b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r), __float_as_uint(r), false, false);
    return IS_MAX ? fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1])) : __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__global__ __launch_bounds__(256) void attn_mfma_bwd_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dy,
                                                            uint16_t* __restrict__ dqkv, int n_pairs, int T, int D, int H, int hd,
                                                            float scale, float p, float inv_keep, uint32_t seed, uint32_t site) {
    __shared__ __attribute__((aligned(16))) uint16_t sm[4][3][16][kAmLd];       // per wave: q, k, dy row-major
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, x = lane & 15, g = lane >> 4;
    const int pair = blockIdx.x * 4 + wid;
    if (pair >= n_pairs) return;                     // (wave-uniform; the kernel has no workgroup barrier)
    const int b = pair / H, h = pair % H;
    const size_t ldq = (size_t)3 * D;
    const uint16_t* qb = qkv + (size_t)b * T * ldq + (size_t)h * hd;
    const uint16_t* gb = dy + (size_t)b * T * D + (size_t)h * hd;
    const int nkk = hd > 32 ? 2 : 1;
    u32x4 qf[2], kf[2], vf[2], gf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int c0 = 32 * kk + 8 * g;
        qf[kk] = row_chunk16(qb, ldq, x, c0, T, hd);
        kf[kk] = row_chunk16(qb + D, ldq, x, c0, T, hd);
        vf[kk] = row_chunk16(qb + 2 * D, ldq, x, c0, T, hd);
        gf[kk] = row_chunk16(gb, (size_t)D, x, c0, T, hd);
    }
    uint16_t (*sq)[kAmLd] = sm[wid][0], (*sk)[kAmLd] = sm[wid][1], (*sg)[kAmLd] = sm[wid][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        *(u32x4*)&sq[x][32 * kk + 8 * g] = qf[kk];
        *(u32x4*)&sk[x][32 * kk + 8 * g] = kf[kk];
        *(u32x4*)&sg[x][32 * kk + 8 * g] = gf[kk];
    }
    auto mma32 = [](const u32x4& a, const u32x4& bb, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bb), c, 0, 0, 0);
    };
    f32x4 sT = {0.f, 0.f, 0.f, 0.f}, sN = sT, pT = sT, pN = sT;
    for (int kk = 0; kk < nkk; ++kk) {
        sT = mma32(kf[kk], qf[kk], sT);              // [j][i]: lane (i, g) holds j = 4 g + r
        sN = mma32(qf[kk], kf[kk], sN);              // [i][j]: lane (j, g) holds i = 4 g + r
        pT = mma32(vf[kk], gf[kk], pT);              // dPd, the same two layouts
        pN = mma32(gf[kk], vf[kk], pN);
    }
    // softmax, dropout keep-scales, dS -- layout T: query i = x, keys j = 4 g + r
    float dsT[4], dsN[4], pdN[4];
    {
        float e[4], ks[4], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = (4 * g + r <= x) ? sT[r] * scale : -INFINITY; mx = fmaxf(mx, e[r]); }
        mx = cross_rows_allreduce<true>(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = expf(e[r] - mx); sum += e[r]; }
        sum = cross_rows_allreduce<false>(sum);
        const float inv = 1.0f / sum;
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 4 * g + r;
            ks[r] = (p > 0.f && x < T && j <= x) ? drop_scale(seed, site, ((size_t)pair * T + x) * T + j, p, inv_keep) : 1.f;
            e[r] *= inv;
            dot = fmaf(pT[r] * ks[r], e[r], dot);
        }
        dot = cross_rows_allreduce<false>(dot);
#pragma unroll
        for (int r = 0; r < 4; ++r) dsT[r] = e[r] * (pT[r] * ks[r] - dot) * scale;
    }
    // layout N: key j = x, queries i = 4 g + r (a query's keys are the 16 lanes of this DPP row)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        const bool in = x <= i;
        const float ev = in ? sN[r] * scale : -INFINITY;
        const float mx = row16_allreduce<true>(ev);
        const float ex = expf(ev - mx);
        const float pr = ex * (1.0f / row16_allreduce<false>(ex));
        const float ks = (p > 0.f && i < T && in) ? drop_scale(seed, site, ((size_t)pair * T + i) * T + x, p, inv_keep) : 1.f;
        const float dot = row16_allreduce<false>(pN[r] * ks * pr);
        dsN[r] = pr * (pN[r] * ks - dot) * scale;
        pdN[r] = pr * ks;
    }
    const uint2 bdsT = pack4_bf16(dsT[0], dsT[1], dsT[2], dsT[3]);
    const uint2 bdsN = pack4_bf16(dsN[0], dsN[1], dsN[2], dsN[3]);
    const uint2 bpdN = pack4_bf16(pdN[0], pdN[1], pdN[2], pdN[3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                // the wave's LDS copies are written (LDS serves a wave's accesses in order)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    auto tr = [&](uint16_t (*m)[kAmLd], int dt) {   // lane (x, g): m[4 g + r][16 dt + x], r = 0 .. 3
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(&m[4 * g + (x >> 2)][16 * dt + 4 * (x & 3)]));
    };
    auto mma16b = [](const s16x4& a, const uint2& bb, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(s16x4, bb), c, 0, 0, 0);
    };
    uint16_t* ob = dqkv + ((size_t)b * T + x) * ldq + (size_t)h * hd;      // row x of this pair's dq | dk | dv
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // The results leave as 16-BYTE pieces (round 6): a lane holds dims 16 dt + 4 g .. + 3 of its row = 8 bytes; the lane groups g, g + 1
    // of the dim tiles dt, dt + 1 exchange halves (two v_permlane16_swap per tensor), after which an even group holds 8 consecutive
    // dims of tile dt and an odd group 8 of tile dt + 1 -- six requests per lane where there were twelve (a head's rows are 8-byte
    // aligned when hd is not a multiple of 8: global memory takes the unaligned dwordx4).  A piece that straddles the end of the head
    // (hd = 60: dims 56 .. 63) goes out as its first 8 bytes.
    typedef uint32_t u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
    auto put16 = [&](uint16_t* dst, int dt, uint2 a, uint2 bq) {            // a: tile dt, bq: tile dt + 1 (this lane's four dims of each)
        const u32x2_t sx = __builtin_amdgcn_permlane16_swap(a.x, bq.x, false, false);
        const u32x2_t sy = __builtin_amdgcn_permlane16_swap(a.y, bq.y, false, false);
        const int f = 16 * (dt + (g & 1)) + 4 * (g & ~1);                  // first of the lane's eight dims
        if (x < T && f + 8 <= hd) *(u32x4_a8*)(dst + f) = u32x4_a8{sx[0], sy[0], sx[1], sy[1]};
        else if (x < T && f < hd) *(uint2*)(dst + f) = make_uint2(sx[0], sy[0]);
    };
#pragma unroll
    for (int dt = 0; dt < 4; dt += 2) {
        if (16 * dt >= hd) break;                    // (wave-uniform)
        uint2 pq[2], pk[2], pv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pq[u] = pk[u] = pv[u] = make_uint2(0u, 0u);
            if (16 * (dt + u) < hd) {                // (wave-uniform)
                const f32x4 dq = mma16b(tr(sk, dt + u), bdsT, zero);     // [d][i]: lane (i, g) holds d = 16 (dt + u) + 4 g + r
                const f32x4 dk = mma16b(tr(sq, dt + u), bdsN, zero);     // [d][j]
                const f32x4 dv = mma16b(tr(sg, dt + u), bpdN, zero);
                pq[u] = pack4_bf16(dq[0], dq[1], dq[2], dq[3]);
                pk[u] = pack4_bf16(dk[0], dk[1], dk[2], dk[3]);
                pv[u] = pack4_bf16(dv[0], dv[1], dv[2], dv[3]);
            }
        }
        put16(ob, dt, pq[0], pq[1]);
        put16(ob + D, dt, pk[0], pk[1]);
        put16(ob + 2 * D, dt, pv[0], pv[1]);
    }
}

// ---------------------------------------------------------------------------------------------
// The last layer's out-projection, MLP, ln_f and head only matter on the action-token rows (nothing else reaches the
// loss: score_gpts.py:341-353), so they run on a COMPACT copy of those rows: compact row c = b*t + i <-> token row
// b*T + G + 2 + 2i.  gather: full -> compact; scatter: compact -> full with zeros on every other row (the gradient
// that enters the attention of the last layer).  Four elements per thread (D % 8 == 0).
// ---------------------------------------------------------------------------------------------
template <typename V>     // V: float or the operand type
__global__ void gather_rows_kernel(const V* __restrict__ src, V* __restrict__ dst, int n_compact, int t, int T, int G, int D) {
    const int d4 = D / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n_compact * d4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / d4), k = (int)(i % d4), b = c / t, j = c % t;
        const size_t m = (size_t)b * T + G + 2 + 2 * j;
        Vec4<V>::store(dst + (size_t)c * D + 4 * k, Vec4<V>::load(src + m * D + 4 * k));
    }
}
template <typename V>
__global__ void scatter_rows_kernel(const V* __restrict__ src, V* __restrict__ dst, int M, int t, int T, int G, int D) {
    const int d4 = D / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * d4; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / d4), k = (int)(i % d4), b = m / T, idx = m % T - 1 - G;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (idx >= 0 && (idx & 1)) v = Vec4<V>::load(src + ((size_t)b * t + (idx >> 1)) * D + 4 * k);
        Vec4<V>::store(dst + (size_t)m * D + 4 * k, v);
    }
}

// both scatters of the last layer's backward in one launch (round 6: two dependent launches of ~7 us on the step's chain)
template <typename E>
__global__ void scatter_rows2_kernel(const E* __restrict__ src_e, E* __restrict__ dst_e, const float* __restrict__ src_f,
                                     float* __restrict__ dst_f, int M, int t, int T, int G, int D) {
    const int d4 = D / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)M * d4; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / d4), k = (int)(i % d4), b = m / T, idx = m % T - 1 - G;
        f32x4 ve = {0.f, 0.f, 0.f, 0.f}, vf = ve;
        if (idx >= 0 && (idx & 1)) {
            const size_t so = ((size_t)b * t + (idx >> 1)) * D + 4 * k;
            ve = Vec4<E>::load(src_e + so);
            vf = Vec4<float>::load(src_f + so);
        }
        Vec4<E>::store(dst_e + (size_t)m * D + 4 * k, ve);
        Vec4<float>::store(dst_f + (size_t)m * D + 4 * k, vf);
    }
}

// ---------------------------------------------------------------------------------------------
// squared-error loss over the (compact) action-token rows (score_wrappers.py:70-79 with pred_last_action_only
// False: per-sample mean over (t, act), then the batch mean = the mean over all B*t*act elements), its gradient
// with respect to the prediction, operand typed, zero on the padding columns.
// ---------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                   E* __restrict__ dpred, float* __restrict__ loss, int M, int act, int ap,
                                                   float inv_count, float grad_scale, int t, int last_only) {
    __shared__ float part[4];
    float acc = 0.f;
    const size_t n = (size_t)M * ap;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t m = i / ap;                     // compact action row b*t + i: the layout of `target`
        const int a = (int)(i % ap);
        float g = 0.f;
        // pred_last_action_only: only the last step of every window is scored (score_wrappers.py:76-77)
        if (a < act && (!last_only || (int)(m % t) == t - 1)) {
            const float diff = pred[i] - target[m * act + a];
            acc = fmaf(diff, diff, acc);
            g = 2.0f * diff * inv_count * grad_scale;
        }
        dpred[i] = Act<E>::from(g);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) * inv_count);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// "the dynamic-LDS attribute of this kernel is set" per device (hipFuncSetAttribute is a driver call: once, not per step)
struct LdsAttrT { std::atomic<unsigned long long> devices{0}; };
static hipError_t ensure_lds_t(const void* kernel, size_t bytes, LdsAttrT* done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev < 64 && (done->devices.load(std::memory_order_relaxed) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev < 64) done->devices.fetch_or(bit, std::memory_order_relaxed);
    return e;
}

// compute units of the current device (cached per device; 256 on the MI355X): the row-range heuristics of the grouped
// weight-gradient launches fill THIS part's workgroup slots
static int device_cus() {
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

// Events that order the step's side streams, per calling thread AND per device (an event created on device A and
// recorded on a stream of device B is hipErrorInvalidHandle -- a thread that drives several GPUs one after the other)
enum { kEvFork = 0, kEvJoin, kEvCopies, kEvLoss, kEvEarly, kEvSideFork, kEvSideJoin, kEvWcat, kEvCount };
static hipError_t step_event(int which, hipEvent_t* out) {
    constexpr int kMaxDev = 64;
    static thread_local hipEvent_t ev[kMaxDev][kEvCount] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDev) return hipErrorInvalidDevice;
    if (!ev[dev][which]) {
        e = hipEventCreateWithFlags(&ev[dev][which], hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    *out = ev[dev][which];
    return hipSuccess;
}

// A stream of the library's own per calling thread and device: work of the step that depends on neither the caller's side
// streams nor the launch it runs beside (round 5: the small reductions of the bias / LayerNorm partial sums under the grouped
// weight-gradient launch).  Forked from and joined to the compute stream with events inside the call -- the caller never sees it.
static hipError_t step_side_stream(hipStream_t* out) {
    constexpr int kMaxDev = 64;
    static thread_local hipStream_t st[kMaxDev] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDev) return hipErrorInvalidDevice;
    if (!st[dev]) {
        e = hipStreamCreateWithFlags(&st[dev], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    *out = st[dev];
    return hipSuccess;
}

static size_t carve_t(size_t& cur, size_t bytes) {
    const size_t off = cur;
    cur = round_up_sz(cur + bytes, 256);
    return off;
}

struct TrainLayerWs {
    size_t w_qkv, b_qkv, w_proj, w_fc1, w_fc2;                    // operand-typed weight copies (b_qkv fp32 [3D])
    size_t x_mid, x_out, st1, st2, xn1, qkv, y, xn2, h, g;        // kept activations
    size_t dyo, dym, dh, dqkv;                                    // kept output gradients (operands of the weight gradients)
};
struct TrainWs {
    int M, T, Ke, ap;
    size_t noised, target, x0, xemb, stf, xf, pred, dpred, w_head, b_head, dw_cat, dw_head;
    size_t xemb32, wcat;                                            // fp32 feature matrix [M][Ke] and Wcat [D][Ke]: the embedding as one GEMM
    int Hp;                                                         // padded hidden width of the MLP head (0: linear head)
    size_t w_hid, b_hid, hz, ha, hdz, dw_hid, db_hid;
    size_t dx, dx0b, dxn, dy, ln_part;
    size_t ya, xa, dxa, dya;                                        // compact action rows of the last layer
    size_t fimg;                                                    // per-step fragment image of the weights (tail-block forward)
    size_t bimg;                                                    // ... of the transposed weights (data-gradient kernel)
    size_t b1slab;                                                  // [L][workgroups][4 D] fp32: FC1 bias sums per workgroup of that kernel
    size_t wslab; int w_splits, w_splits_panel; size_t wslab_floats;   // split weight-gradient launch: [splits][wslab_floats] partial outputs
    TrainLayerWs layer[kMaxLayers];
    size_t total;
};

static bool make_train_ws(const beso_config* c, int batch, int t, int precision, TrainWs* w) {
    memset(w, 0, sizeof(*w));
    const size_t e = precision == BESO_PREC_FP32 ? 4 : 2, f = 4;
    const int D = c->embed_dim, G = c->goal_seq_len;
    const int T = 1 + G + 2 * t;
    const size_t M = (size_t)batch * T;
    w->M = (int)M; w->T = T;
    w->Ke = round_up(c->obs_dim + c->act_dim + 4 + G + c->obs_seq_len + 1, 8);
    w->ap = round_up(c->act_dim, 16);
    size_t cur = 0;
    const size_t na = (size_t)batch * t * c->act_dim;
    w->noised = carve_t(cur, f * na); w->target = carve_t(cur, f * na);
    w->x0 = carve_t(cur, f * M * D); w->xemb = carve_t(cur, e * M * w->Ke);
    w->xemb32 = carve_t(cur, e == 4 ? 0 : f * M * w->Ke); w->wcat = carve_t(cur, f * (size_t)D * w->Ke);
    if (e == 4) w->xemb32 = w->xemb;                                // (fp32 mode: the operand-typed matrix IS the fp32 one)
    w->stf = carve_t(cur, f * M * 2); w->xf = carve_t(cur, e * M * D);
    w->pred = carve_t(cur, f * M * w->ap); w->dpred = carve_t(cur, e * M * w->ap);
    w->b_head = carve_t(cur, f * w->ap);
    w->dw_cat = carve_t(cur, f * (size_t)w->Ke * D);
    w->Hp = c->linear_output ? 0 : round_up(kHeadHidden, 8);
    {
        // linear head: w_head [ap][D]; MLP head: w_hid [Hp][D] (first layer) and w_head [ap][Hp] (second layer)
        const size_t Kh = w->Hp ? (size_t)w->Hp : (size_t)D, Ma = (size_t)batch * t;
        w->w_head = carve_t(cur, e * (size_t)w->ap * Kh); w->dw_head = carve_t(cur, f * (size_t)w->ap * Kh);
        w->w_hid = carve_t(cur, e * (size_t)w->Hp * D); w->b_hid = carve_t(cur, f * (size_t)(w->Hp + 8));
        w->dw_hid = carve_t(cur, f * (size_t)w->Hp * D); w->db_hid = carve_t(cur, f * (size_t)(w->Hp + 8));
        w->hz = carve_t(cur, e * Ma * w->Hp); w->ha = carve_t(cur, e * Ma * w->Hp); w->hdz = carve_t(cur, e * Ma * w->Hp);
    }
    w->dx = carve_t(cur, f * M * D); w->dx0b = carve_t(cur, e * M * D); w->dxn = carve_t(cur, f * M * D);
    w->dy = carve_t(cur, e * M * D);
    {
        const size_t Ma = (size_t)batch * t;
        w->ya = carve_t(cur, e * Ma * D); w->xa = carve_t(cur, f * Ma * D);
        w->dxa = carve_t(cur, f * Ma * D); w->dya = carve_t(cur, e * Ma * D);
    }
    w->ln_part = carve_t(cur, f * (size_t)(2 * c->n_layers + 1) * ((M + 15) / 16) * 3 * D);     // LayerNorm backward block partials
    {
        Layout lay;
        const bool img = precision == BESO_PREC_BF16 && make_layout(c, BESO_PREC_BF16, &lay);
        const size_t tail_b = img ? fused_train_image_bytes(lay) : 0, whole_b = img ? fused_train_whole_image_bytes(lay) : 0;
        w->fimg = carve_t(cur, tail_b > whole_b ? tail_b : whole_b);
        w->bimg = carve_t(cur, img ? fused_train_dgrad_image_bytes(lay) : 0);
        w->b1slab = carve_t(cur, img && fused_train_dgrad_supported(lay)
                                     ? f * (size_t)c->n_layers * fused_train_dgrad_blocks((int)M) * 4 * D : 0);
    }
    {
        // row ranges of the grouped weight-gradient launch: enough workgroups for two per CU (small models: few output tiles,
        // long contractions); every output of the launch once per range
        const int tD = (D + kTileMN - 1) / kTileMN, tH = (4 * D + kTileMN - 1) / kTileMN;
        const int tiles = c->n_layers * (2 * tD * tH + 4 * tD * tD) + 4;
        const int cus = device_cus();
        int sp = (2 * cus + tiles - 1) / tiles;
        if (sp > 8) sp = 8;
        if ((size_t)sp > M / (4 * (128 / e))) sp = (int)(M / (4 * (128 / e)));      // (ranges of at least four stages)
        w->w_splits = sp < 2 ? 1 : sp;
        // ... and of the panel-owning tiles (one workgroup per CU, one round): as many row ranges as keep the launch within 256
        w->w_splits_panel = 1;
        if (const int W = wgrad_panel_w(D, e)) {
            (void)W;
            const int ptiles = c->n_layers * (2 * tH + 4 * tD) + 4;
            int psp = cus / ptiles;
            if (psp > 8) psp = 8;
            if ((size_t)psp > M / (4 * 64)) psp = (int)(M / (4 * 64));
            w->w_splits_panel = psp < 2 ? 1 : psp;
        }
        const int sp_max = w->w_splits > w->w_splits_panel ? w->w_splits : w->w_splits_panel;
        const size_t Kh = w->Hp ? (size_t)w->Hp : (size_t)D;
        w->wslab_floats = train_grad_floats(c) + (size_t)w->Ke * D + (size_t)w->ap * Kh + (size_t)w->Hp * D + 8 * (size_t)(6 * c->n_layers + 8);
        w->wslab = carve_t(cur, sp_max > 1 ? f * sp_max * w->wslab_floats : 0);
    }
    for (int l = 0; l < c->n_layers; ++l) {
        TrainLayerWs& y = w->layer[l];
        y.w_qkv = carve_t(cur, e * (size_t)3 * D * D); y.b_qkv = carve_t(cur, f * (size_t)3 * D);
        y.w_proj = carve_t(cur, e * (size_t)D * D);
        y.w_fc1 = carve_t(cur, e * (size_t)4 * D * D); y.w_fc2 = carve_t(cur, e * (size_t)4 * D * D);
        y.x_mid = carve_t(cur, f * M * D); y.x_out = carve_t(cur, f * M * D);
        y.st1 = carve_t(cur, f * M * 2); y.st2 = carve_t(cur, f * M * 2);
        y.xn1 = carve_t(cur, e * M * D); y.qkv = carve_t(cur, e * M * 3 * D); y.y = carve_t(cur, e * M * D);
        y.xn2 = carve_t(cur, e * M * D); y.h = carve_t(cur, e * M * 4 * D); y.g = carve_t(cur, e * M * 4 * D);
        y.dyo = carve_t(cur, e * M * D); y.dym = carve_t(cur, e * M * D); y.dh = carve_t(cur, e * M * 4 * D);
        y.dqkv = carve_t(cur, e * M * 3 * D);
    }
    w->total = cur;
    return true;
}

int train_validate(const beso_config* c, int batch, int t) {
    int st = validate_config(c);
    if (st != BESO_OK) return st;
    if (batch < 1 || t < 1 || t > c->obs_seq_len) return BESO_ERR_BAD_SHAPE;
    if (c->embed_dim % 8 != 0) return BESO_ERR_UNSUPPORTED;        // 16-byte operand chunks, float4 LayerNorm rows
    const int T = 1 + c->goal_seq_len + 2 * t;
    if (attn_lds_bytes(T, c->embed_dim / c->n_heads, true) > 150 * 1024) return BESO_ERR_UNSUPPORTED;
    // operands are addressed with 32-bit byte offsets (raw buffer loads): the widest one is [M][4D] in fp32
    if ((size_t)batch * T * 4 * c->embed_dim * sizeof(float) >= ((size_t)1 << 31)) return BESO_ERR_BAD_SHAPE;
    return BESO_OK;
}

size_t train_workspace_bytes(const beso_config* c, int batch, int t, int precision) {
    if (train_validate(c, batch, t) != BESO_OK) return 0;
    if (precision != BESO_PREC_BF16 && precision != BESO_PREC_FP32) return 0;
    TrainWs w;
    make_train_ws(c, batch, t, precision, &w);
    return w.total;
}

// The data-parallel exchange can start before the backward pass is over: the gradients of transformer layers
// train_early_layer(c) .. L-1 and of ln_f form one contiguous range of the flat gradient buffer and are completed first.
// Their weight gradients run as a grouped launch of their own, so the split is placed where that launch is one full
// round of workgroups (256 CUs x 2): as many upper layers as fit into 512 tiles.  (An even split of the kitchen model,
// 324 + 324 tiles, costs 0.15 ms per step -- two half-filled rounds; 432 + 216 costs nothing measurable.)
int train_early_layer(const beso_config* c) {
    const int L = c->n_layers;
    if (L < 2) return 0;
    const int td = (c->embed_dim + kTileMN - 1) / kTileMN, t4 = (4 * c->embed_dim + kTileMN - 1) / kTileMN;
    const int per_layer = 4 * td * td + 2 * t4 * td;               // q, k, v, proj + fc1 + fc2
    int n = 512 / per_layer;
    if (n < 1) n = 1;
    if (n > L - 1) n = L - 1;
    return L - n;
}
void train_early_range(const beso_config* c, size_t* begin, size_t* end) {
    const size_t D = (size_t)c->embed_dim, seq = (size_t)c->goal_seq_len + c->obs_seq_len + 1;
    const size_t per_layer = 4 * D + 4 * (D * D + D) + (4 * D * D + 4 * D) + (4 * D * D + D);
    const size_t layers0 = seq * D + D * (size_t)c->obs_dim + D;
    const int l0 = train_early_layer(c);
    if (l0 < 1) { *begin = *end = 0; return; }                     // fewer than two layers: nothing is early
    *begin = layers0 + per_layer * (size_t)l0;
    *end = layers0 + per_layer * (size_t)c->n_layers + 2 * D;      // ... + ln_f weight and bias
}
size_t train_grad_floats(const beso_config* c) {
    if (validate_config(c) != BESO_OK) return 0;
    const size_t D = c->embed_dim, seq = c->goal_seq_len + c->obs_seq_len + 1;
    size_t n = seq * D + D * c->obs_dim + D;
    n += (size_t)c->n_layers * (4 * D + 4 * (D * D + D) + (4 * D * D + 4 * D) + (4 * D * D + D));
    n += 2 * D + 2 * D + D * c->act_dim + D;
    n += c->linear_output ? (size_t)c->act_dim * D + c->act_dim
                          : (size_t)kHeadHidden * D + kHeadHidden + (size_t)c->act_dim * kHeadHidden + c->act_dim;
    return n;
}

// The tail-block forward pays off once its 96-token tiles are several rounds of workgroups (measured on MI355X, kitchen:
// 8192 samples = 939 tiles 16.57 vs 17.41 ms per step; 1024 samples = 118 tiles 3.43 vs 3.39 ms -- at under one round both
// forms are bound by a lone workgroup's latency): taken from kTailMinRows token rows on, unless the call's plan hint says
// otherwise (BESO_TRAIN_PLAN_PER_OP: never, BESO_TRAIN_PLAN_TILES: always -- the two forms write the same kept activations).
constexpr int kTailMinRows = 16000;       // kitchen: 3.39 = 3.40 ms at 1024 samples (11 k rows), 4.99 vs 5.37 at 2048, 8.61 vs 8.96 at 4096
static bool tail_forward_enabled(int rows, int flags) {
    if (flags & BESO_TRAIN_PLAN_PER_OP) return false;
    return (flags & BESO_TRAIN_PLAN_TILES) || rows >= kTailMinRows;
}

#define TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { *err = _e; *err_line = __LINE__; return BESO_ERR_HIP; } } while (0)

template <typename E>
static int loss_grad_e(const beso_config* c, const float* const* p, float* gflat, int precision, const float* state,
                       const float* action, const float* goal, const float* noise, const float* sigma, float* loss_out,
                       int batch, int t, int flags, float embed_p, float attn_p, float resid_p, float goal_p, uint32_t seed,
                       float grad_scale, char* ws,
                       const TrainWs& w, hipStream_t s, hipStream_t early_stream, hipStream_t loss_stream, hipError_t* err, int* err_line) {
    const int D = c->embed_dim, H = c->n_heads, hd = D / H, L = c->n_layers, G = c->goal_seq_len;
    const int obs = c->obs_dim, act = c->act_dim, seq = G + c->obs_seq_len + 1;
    const int M = w.M, T = w.T, Ke = w.Ke, ap = w.ap, D3 = 3 * D, D4 = 4 * D;
    const int last_only = flags & BESO_TRAIN_LAST_ACTION_ONLY;
    const float scale = 1.0f / sqrtf((float)hd);
    const float attn_ik = attn_p > 0.f ? 1.0f / (1.0f - attn_p) : 1.f, resid_ik = resid_p > 0.f ? 1.0f / (1.0f - resid_p) : 1.f;
    auto F = [&](size_t off) { return (float*)(ws + off); };
    auto P = [&](size_t off) { return (E*)(ws + off); };

    // parameter / gradient pointers, order of beso_pack_weights
    const float* const* q = p;
    float* g = gflat;
    struct PG { const float* p; float* g; };
    auto take = [&](size_t n) { PG r{*q, g}; ++q; g += n; return r; };
    const PG pos = take((size_t)seq * D), tokw = take((size_t)D * obs), tokb = take(D);
    struct LayerPG { PG ln1w, ln1b, ln2w, ln2b, kw, kb, qw, qb, vw, vb, pw, pb, f1w, f1b, f2w, f2b; };
    LayerPG lp[kMaxLayers];
    for (int l = 0; l < L; ++l) {
        LayerPG& y = lp[l];
        y.ln1w = take(D); y.ln1b = take(D); y.ln2w = take(D); y.ln2b = take(D);
        y.kw = take((size_t)D * D); y.kb = take(D); y.qw = take((size_t)D * D); y.qb = take(D);
        y.vw = take((size_t)D * D); y.vb = take(D); y.pw = take((size_t)D * D); y.pb = take(D);
        y.f1w = take((size_t)D4 * D); y.f1b = take(D4); y.f2w = take((size_t)D * D4); y.f2b = take(D);
    }
    const PG lnfw = take(D), lnfb = take(D), sigw = take(D), sigb = take(D), actw = take((size_t)D * act), actb = take(D);
    // action head: Linear(D, act), or Linear(D, 100) - SiLU - Linear(100, act)   (score_gpts.py:184-191)
    const int Hh = kHeadHidden, Hp = w.Hp;
    const bool mlp_head = !c->linear_output;
    const PG h0w = mlp_head ? take((size_t)Hh * D) : PG{nullptr, nullptr}, h0b = mlp_head ? take(Hh) : PG{nullptr, nullptr};
    const PG hw = take((size_t)act * (mlp_head ? Hh : D)), hb = take(act);
    const size_t n_grad = (size_t)(g - gflat);

    // Which form the forward takes (decided here: its weight image is packed with the other per-step weight copies).
    // bf16, no dropout on the proj / MLP outputs, a shape with a fused tile kernel: everything of a layer behind its
    // attention and the LN1 + q/k/v of the next layer run as ONE launch (fused.hip: train_tail_kernel) on 96-token tiles
    // with the residual in registers -- six launches of the per-op forward below -- writing the same kept activations in
    // the same formats.  The last layer stays per-op (it continues on the compact action rows).
    Layout flay;
    const bool use_tail = sizeof(E) == 2 && resid_p == 0.f && L >= 2 && L * 13 <= 96 && make_layout(c, BESO_PREC_BF16, &flay) &&
                          fused_train_supported(flay) && fused_train_image_bytes(flay) > 0 && tail_forward_enabled(M, flags);
    // ... and where the shape has the one-launch kernel (kitchen, block-push; bf16; round 5: with or without dropout on the
    // proj / MLP outputs), ALL
    // layers run as ONE launch (fused.hip: train_fwd_kernel) -- 44 launches of the per-op forward at six layers; the call's plan
    // hints keep the other two forms reachable (BESO_TRAIN_PLAN_PER_OP, BESO_TRAIN_PLAN_TILES)
    const bool use_whole = sizeof(E) == 2 && !(flags & (BESO_TRAIN_PLAN_PER_OP | BESO_TRAIN_PLAN_TILES)) &&
                           make_layout(c, BESO_PREC_BF16, &flay) && fused_train_whole_supported(flay, T, t);

    // The data-gradient GEMMs of the backward pass in the transposed formulation (fused.hip: train_dgrad_kernel; bf16, the shapes
    // with the fused kernels); BESO_TRAIN_PLAN_PER_OP keeps the 128 x 128 tile kernel for them too
    // (they address the [M][D] ... [M][4 D] tensors with 32-bit byte offsets through buffer resources: M 4 D < 2^30 elements)
    const bool use_dgrad = sizeof(E) == 2 && !(flags & BESO_TRAIN_PLAN_PER_OP) && make_layout(c, BESO_PREC_BF16, &flay) &&
                           fused_train_dgrad_supported(flay) && (size_t)M * D < ((size_t)1 << 28);

    // The per-step weight copies depend on the parameters only, the embedding on the batch only: when the caller handed over a
    // second stream (loss_stream: idle at this point), the copies run THERE beside the gradient buffer's memset, the
    // preconditioning and the embedding on `s` -- two short chains of small kernels side by side instead of one after the
    // other (round 4: -50 us of a 2.5 ms step).  The forward waits for both.
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_copies = nullptr;
    const bool fork = loss_stream != nullptr;
    hipStream_t ps = fork ? loss_stream : s;
    if (fork) {
        TRY(step_event(kEvFork, &ev_fork));
        TRY(step_event(kEvJoin, &ev_join));
        TRY(step_event(kEvCopies, &ev_copies));
        TRY(hipEventRecord(ev_fork, s));                 // (behind the optimizer step that wrote the parameters)
        TRY(hipStreamWaitEvent(ps, ev_fork, 0));
    }
    // (loss_out and the padded head bias: prep_kernel.  Beside the store-bound forward the 37.5 MB memset measured slower; round 6:
    //  in FRONT of the forward image's pack on the side stream -- the forward waits for that stream anyway (ev_join), and the
    //  compute stream's chain of small launches in front of the forward is 9 us shorter)
    // (the embedding's concatenated weight -- parameters only -- is packed there too, first: the compute stream waits for it
    //  in front of the embedding GEMM, three small launches later)
    hipEvent_t ev_wcat = nullptr;
    const bool wcat_side = fork && embed_p == 0.f;
    if (wcat_side) {
        hipLaunchKernelGGL(wcat_pack_kernel, dim3((D * Ke + 255) / 256), dim3(256), 0, ps, pos.p, tokw.p, tokb.p, sigw.p, sigb.p,
                           actw.p, actb.p, F(w.wcat), D, obs, act, seq, Ke);
        TRY(hipGetLastError());
        TRY(step_event(kEvWcat, &ev_wcat));
        TRY(hipEventRecord(ev_wcat, ps));
    }
    TRY(hipMemsetAsync(gflat, 0, sizeof(float) * n_grad, ps));

    // (first what the forward launch needs -- its fragment image --, then the plain copies the backward pass reads)
    if (use_whole || use_tail) {
        const int pst = use_whole ? fused_train_whole_pack(flay, p, ws + w.fimg, ps) : fused_train_pack(flay, p, ws + w.fimg, ps);
        if (pst != BESO_OK) { *err = hipGetLastError(); *err_line = __LINE__; return pst; }
    }
    if (fork) TRY(hipEventRecord(ev_join, ps));
    const bool use_mlp_bwd = use_dgrad && fused_train_mlp_bwd_supported(flay);
    // the one-launch forward keeps x_mid / x_out as bf16 when nothing but the fused LayerNorm-backward epilogues read them
    // (no residual dropout: with it the forward itself adds the residual back from the kept fp32 rows)
    const int x16 = (use_whole && use_dgrad && use_mlp_bwd && resid_p == 0.f) ? 1 : 0;
    if (use_dgrad) {
        const int pst = fused_train_dgrad_pack(flay, p, ws + w.bimg, ps);
        if (pst != BESO_OK) { *err = hipGetLastError(); *err_line = __LINE__; return pst; }
    }
    // ---- operand-typed weight copies (fused q|k|v rows as in the inference image): one launch for up to seven layers
    // (nobody reads them when the forward is the one-launch kernel and the data gradients take the transposed weights)
    if (!(use_whole && use_dgrad)) {
        PackTable t;
        t.n = 0;
        uint32_t total4 = 0;
        auto flush = [&]() -> hipError_t {
            if (t.n == 0) return hipSuccess;
            hipLaunchKernelGGL(pack_table_kernel<E>, dim3((total4 + 255) / 256 > 4096 ? 4096 : (total4 + 255) / 256), dim3(256), 0,
                               ps, t, total4);
            t.n = 0; total4 = 0;
            return hipGetLastError();
        };
        for (int l = 0; l < L; ++l) {
            const TrainLayerWs& y = w.layer[l];
            const size_t e = sizeof(E);
            const uint32_t dd4 = (uint32_t)((size_t)D * D / 4), d4 = (uint32_t)(D / 4);
            if (t.n + 9 > kPackSegs) TRY(flush());
            auto seg = [&](const float* src, void* dst, uint32_t n4, uint32_t f32) { t.seg[t.n++] = PackSeg{src, dst, total4, f32}; total4 += n4; };
            seg(lp[l].qw.p, ws + y.w_qkv, dd4, 0);
            seg(lp[l].kw.p, ws + y.w_qkv + e * (size_t)D * D, dd4, 0);
            seg(lp[l].vw.p, ws + y.w_qkv + e * (size_t)2 * D * D, dd4, 0);
            seg(lp[l].pw.p, ws + y.w_proj, dd4, 0);
            seg(lp[l].f1w.p, ws + y.w_fc1, 4 * dd4, 0);
            seg(lp[l].f2w.p, ws + y.w_fc2, 4 * dd4, 0);
            seg(lp[l].qb.p, ws + y.b_qkv, d4, 1);
            seg(lp[l].kb.p, ws + y.b_qkv + sizeof(float) * D, d4, 1);
            seg(lp[l].vb.p, ws + y.b_qkv + sizeof(float) * 2 * D, d4, 1);
        }
        TRY(flush());
    }
    if (mlp_head) {
        TRY(launch_pack_matrix(h0w.p, Hh, D, ws + w.w_hid, Hp, D, precision, ps));          // zero rows Hh..Hp
        TRY(hipMemsetAsync(ws + w.b_hid, 0, sizeof(float) * Hp, ps));
        TRY(hipMemcpyAsync(ws + w.b_hid, h0b.p, sizeof(float) * Hh, hipMemcpyDeviceToDevice, ps));
        TRY(hipMemsetAsync(ws + w.db_hid, 0, sizeof(float) * Hp, ps));
        TRY(launch_pack_matrix(hw.p, act, Hh, ws + w.w_head, ap, Hp, precision, ps));       // zero columns Hh..Hp, rows act..ap
    } else {
        TRY(launch_pack_matrix(hw.p, act, D, ws + w.w_head, ap, D, precision, ps));
    }
    if (fork) TRY(hipEventRecord(ev_copies, ps));

    // ---- forward
    {
        const size_t n = (size_t)batch * t * act;
        int grid = (int)((n + 255) / 256); if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(prep_kernel, dim3(grid), dim3(256), 0, s, action, noise, sigma, F(w.noised), F(w.target),
                           t * act, n, c->sigma_data, loss_out, F(w.b_head), hb.p, act, ap);
        TRY(hipGetLastError());
        if (embed_p == 0.f) {
            // the embedding as one exact-fp32 GEMM over the feature matrix (train_feat_kernel)
            if (!wcat_side)
                hipLaunchKernelGGL(wcat_pack_kernel, dim3((D * Ke + 255) / 256), dim3(256), 0, s, pos.p, tokw.p, tokb.p, sigw.p, sigb.p,
                                   actw.p, actb.p, F(w.wcat), D, obs, act, seq, Ke);
            const size_t nf = (size_t)M * Ke;
            hipLaunchKernelGGL(train_feat_kernel<E>, dim3((unsigned)((nf + 255) / 256 > 4096 ? 4096 : (nf + 255) / 256)), dim3(256), 0, s,
                               state, (const float*)F(w.noised), goal, sigma, P(w.xemb), F(w.xemb32), M, t, T, G, obs, act, Ke,
                               c->sigma_data, goal_p, seed);
            TRY(hipGetLastError());
            if (wcat_side) TRY(hipStreamWaitEvent(s, ev_wcat, 0));
            TRY((tgemm<float, false, false>(F(w.xemb32), Ke, F(w.wcat), Ke, M, D, Ke, 1, EpiStore<float>{F(w.x0), nullptr, nullptr, D}, s)));
        } else {
        const int threads = D >= 256 ? 256 : round_up(D, 64);
        hipLaunchKernelGGL(train_embed_kernel<E>, dim3(M), dim3(threads), sizeof(float) * (size_t)(obs > act ? obs : act), s,
                           state, (const float*)F(w.noised), goal, sigma, pos.p, tokw.p, tokb.p, sigw.p, sigb.p, actw.p,
                           actb.p, F(w.x0), P(w.xemb), t, T, G, D, obs, act, Ke, c->sigma_data, embed_p, goal_p, seed);
        TRY(hipGetLastError());
        }
    }
    const int nv = D <= 256 ? 1 : (D <= 512 ? 2 : 4);
    const int Ma = batch * t;                             // compact action rows (last layer's projection, MLP, ln_f, head)
    auto ln_fwd = [&](const float* x, const float* gw, const float* gb, E* out, float* st, int rows) -> hipError_t {
        const int grid = (rows + 3) / 4;
        if (nv == 1) hipLaunchKernelGGL((ln_fwd_kernel<E, 1>), dim3(grid), dim3(256), 0, s, x, gw, gb, out, st, rows, D);
        else if (nv == 2) hipLaunchKernelGGL((ln_fwd_kernel<E, 2>), dim3(grid), dim3(256), 0, s, x, gw, gb, out, st, rows, D);
        else hipLaunchKernelGGL((ln_fwd_kernel<E, 4>), dim3(grid), dim3(256), 0, s, x, gw, gb, out, st, rows, D);
        return hipGetLastError();
    };
    auto gs_grid = [&](size_t n4) { const size_t g = (n4 + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); };
    const size_t lds_f = attn_lds_bytes(T, hd, false), lds_b = attn_lds_bytes(T, hd, true);
    const bool attn_small = T <= kTP && hd <= 64 && hd % 4 == 0;
    if (lds_f > 64 * 1024) {
        TRY(hipFuncSetAttribute((const void*)attn_fwd_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
    }
    if (lds_b > 64 * 1024) {
        TRY(hipFuncSetAttribute((const void*)attn_bwd_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    }
    if (fork) TRY(hipStreamWaitEvent(s, ev_join, 0));             // the forward's weight image is in place
    if (fork && !use_whole) TRY(hipStreamWaitEvent(s, ev_copies, 0));      // (per-op / tile forward: the plain copies too)
    if (use_whole) {
        const TrainLayerWs& y0 = w.layer[0];
        const size_t stride = L > 1 ? w.layer[1].x_mid - y0.x_mid : 0;
        const TrainWholeBufs b{F(w.x0), ws, y0.x_mid, y0.x_out, y0.st1, y0.st2, y0.xn1, y0.qkv, y0.y, y0.xn2, y0.h, y0.g,
                               stride, w.ya, t, attn_p, seed, resid_p, x16};
        profile_begin(BESO_SITE_FUSED_LAYER, s);
        const int st = fused_train_whole(flay, ws + w.fimg, batch, T, b, s);
        profile_end(BESO_SITE_FUSED_LAYER, s);
        if (st != BESO_OK) { *err = hipGetLastError(); *err_line = __LINE__; return st; }
        if (fork) TRY(hipStreamWaitEvent(s, ev_copies, 0));       // head weights and the backward's copies: ready long since
    }
    for (int l = 0; l < (use_whole ? 0 : L); ++l) {
        const TrainLayerWs& y = w.layer[l];
        const float* x_in = l == 0 ? F(w.x0) : F(w.layer[l - 1].x_out);
        const bool last = l == L - 1;
        if (!(use_tail && l > 0)) {           // (behind a tail block these arrive from the previous layer's launch)
            TRY(ln_fwd(x_in, lp[l].ln1w.p, lp[l].ln1b.p, P(y.xn1), F(y.st1), M));
            TRY((tgemm<E, false, false>(P(y.xn1), D, P(y.w_qkv), D, M, D3, D, 1,
                                        EpiStore<E>{nullptr, P(y.qkv), F(y.b_qkv), D3}, s)));
        }
        if (attn_small) {
#define ATT(TT) hipLaunchKernelGGL((attn_small_kernel<E, false, TT>), dim3(batch * H), dim3(64), 0, s, (const E*)P(y.qkv), \
                                   (const E*)nullptr, P(y.y), T, D, H, hd, scale, attn_p, attn_ik, seed, (uint32_t)(4 * l))
            if (T <= 8) ATT(8); else if (T <= 12) ATT(12); else ATT(16);
#undef ATT
        } else
            hipLaunchKernelGGL(attn_fwd_kernel<E>, dim3(batch * H), dim3(64), lds_f, s, (const E*)P(y.qkv), P(y.y), T, D, H,
                               hd, scale, attn_p, attn_ik, seed, (uint32_t)(4 * l));
        TRY(hipGetLastError());
        if (use_tail && !last) {
            const TrainLayerWs& nx = w.layer[l + 1];
            const int tst = fused_train_tail(flay, ws + w.fimg, l, M, x_in, P(y.y), D, F(y.x_mid), F(y.x_out), F(y.st2), P(y.xn2),
                                             P(y.h), P(y.g), F(nx.st1), P(nx.xn1), P(nx.qkv), s);
            if (tst != BESO_OK) { *err = hipGetLastError(); *err_line = __LINE__; return tst; }
            continue;
        }
        // the last layer continues on the compact action rows only (its buffers hold Ma rows from here on)
        const int rows = last ? Ma : M;
        const E* y_in = P(y.y);
        const float* res_in = x_in;
        if (last) {
            const size_t n4 = (size_t)Ma * (D / 4);
            hipLaunchKernelGGL(gather_rows_kernel<E>, dim3(gs_grid(n4)), dim3(256), 0, s, (const E*)P(y.y), P(w.ya), Ma, t, T, G, D);
            hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(gs_grid(n4)), dim3(256), 0, s, x_in, F(w.xa), Ma, t, T, G, D);
            TRY(hipGetLastError());
            y_in = P(w.ya); res_in = F(w.xa);
        }
        TRY((tgemm<E, false, false>(y_in, D, P(y.w_proj), D, rows, D, D, 1,
                                    EpiResid{res_in, F(y.x_mid), lp[l].pb.p, D, resid_p, resid_ik, seed, (uint32_t)(4 * l + 1)}, s)));
        TRY(ln_fwd(F(y.x_mid), lp[l].ln2w.p, lp[l].ln2b.p, P(y.xn2), F(y.st2), rows));
        TRY((tgemm<E, false, false>(P(y.xn2), D, P(y.w_fc1), D, rows, D4, D, 1, EpiFc1<E>{P(y.h), P(y.g), lp[l].f1b.p, D4}, s)));
        TRY((tgemm<E, false, false>(P(y.g), D4, P(y.w_fc2), D4, rows, D, D4, 1,
                                    EpiResid{(const float*)F(y.x_mid), F(y.x_out), lp[l].f2b.p, D, resid_p, resid_ik, seed,
                                             (uint32_t)(4 * l + 2)}, s)));
    }
    const float* x_last = F(w.layer[L - 1].x_out);       // [Ma][D]
    TRY(ln_fwd(x_last, lnfw.p, lnfb.p, P(w.xf), F(w.stf), Ma));
    if (mlp_head) {
        TRY((tgemm<E, false, false>(P(w.xf), D, P(w.w_hid), D, Ma, Hp, D, 1, EpiSilu<E>{P(w.hz), P(w.ha), F(w.b_hid), Hp}, s)));
        TRY((tgemm<E, false, false>(P(w.ha), Hp, P(w.w_head), Hp, Ma, ap, Hp, 1, EpiStore<E>{F(w.pred), nullptr, F(w.b_head), ap}, s)));
    } else {
        TRY((tgemm<E, false, false>(P(w.xf), D, P(w.w_head), D, Ma, ap, D, 1, EpiStore<E>{F(w.pred), nullptr, F(w.b_head), ap}, s)));
    }
    {
        const size_t n = (size_t)Ma * ap;
        int grid = (int)((n + 255) / 256); if (grid > 1024) grid = 1024;
        hipLaunchKernelGGL(loss_kernel<E>, dim3(grid), dim3(256), 0, s, (const float*)F(w.pred), (const float*)F(w.target),
                           P(w.dpred), loss_out, Ma, act, ap,
                           1.0f / (float)((size_t)batch * (last_only ? 1 : t) * act), grad_scale, t, last_only);
        TRY(hipGetLastError());
        if (loss_stream) {
            // the loss is final here, long before the step is: a stream of the caller's is ordered behind this point, so that
            // reading the loss there (beso_agent.py:248 `loss.item()`) does not wait for the backward pass
            hipEvent_t ev_loss = nullptr;
            TRY(step_event(kEvLoss, &ev_loss));
            TRY(hipEventRecord(ev_loss, s));
            TRY(hipStreamWaitEvent(loss_stream, ev_loss, 0));
        }
    }

    // ---- backward
    const int rpb = 128;                                  // rows per block of the column sums
    constexpr int EPC = 16 / (int)sizeof(E);
    auto colsum = [&](const E* a, int ld, int cols, int rows, float* o0, float* o1 = nullptr, float* o2 = nullptr,
                      int seg = 1 << 30) -> hipError_t {
        hipLaunchKernelGGL(colsum_kernel<E>, dim3((cols + 64 * EPC - 1) / (64 * EPC), (rows + rpb - 1) / rpb), dim3(256), 0, s, a,
                           ld, rows, cols, o0, o1 ? o1 : o0, o2 ? o2 : o0, seg, rpb);
        return hipGetLastError();
    };
    const int rpw = 4;                                    // rows per wave of the LayerNorm backward
    const int lnb_grid = (M + 4 * rpw - 1) / (4 * rpw);
    LnRedTable lrt;
    int ln_calls = 0, ln_reduced = 0;
    // (every call launches lnb_grid blocks so that the partial slabs have one shape; blocks past `rows` write zeros)
    auto ln_bwd = [&](const float* x, size_t st, const float* gamma, const float* dres_in, float* dres_out, E* dxb, int rows,
                      float* dgam, float* dbet, float* dbias, float p_site, uint32_t site, int skip_mod = 0) -> hipError_t {
        float* part = F(w.ln_part) + (size_t)ln_calls * lnb_grid * 3 * D;
        lrt.nb[ln_calls] = lnb_grid;
        lrt.c[ln_calls++] = LnRedCall{dgam, dbet, dbias};
#define LNB(NV)                                                                                                     \
        hipLaunchKernelGGL((ln_bwd_kernel<E, NV>), dim3(lnb_grid), dim3(256), 0, s, (const float*)F(w.dxn), x,                \
                           (const float*)F(st), gamma, dres_in, dres_out, dxb, part, rows, D,                               \
                           rpw, p_site, p_site > 0.f ? 1.0f / (1.0f - p_site) : 1.f, seed, site, skip_mod)
        if (nv == 1) LNB(1); else if (nv == 2) LNB(2); else LNB(4);
#undef LNB
        return hipGetLastError();
    };
    // the same LayerNorm backward as the epilogue of the data gradient in front of it (which = 0: q|k|v, 2: FC1); dxn is never
    // written (dropout at the site: the same (row, feature) hash, evaluated in the epilogue).
    auto dgrad_ln = [&](int l, int which, int rows, const E* in, const float* x, size_t st, const float* gamma, float* dres,
                        E* dxb, float* dgam, float* dbet, float* dbias, float p_site, uint32_t site, int skip_mod = 0,
                        int x_is_bf16 = 0) -> int {
        float* part = F(w.ln_part) + (size_t)ln_calls * lnb_grid * 3 * D;
        lrt.nb[ln_calls] = fused_train_dgrad_blocks(rows);
        lrt.c[ln_calls++] = LnRedCall{dgam, dbet, dbias};
        const TrainLnBwd ln{x, (const float*)F(st), gamma, dres, dres, dxb, part, p_site, seed, site, skip_mod, x_is_bf16};
        return fused_train_dgrad(flay, ws + w.bimg, l, which, rows, in, nullptr, nullptr, nullptr, nullptr, nullptr, s, &ln);
    };
    // The weight gradients are collected and run as one grouped launch after the chain of data gradients: every
    // output gradient they need stays in its own buffer until then.
    GTable gt;
    gt.n = 0;
    int g_tiles = 0;
    // bf16, 128 < D <= 384: the grouped launch runs on panel-owning tiles (wgrad_panel_group_kernel); the per-op plan keeps the
    // 128 x 128 tiles
    const int panel_w = (flags & BESO_TRAIN_PLAN_PER_OP) ? 0 : wgrad_panel_w(D, sizeof(E));
    // FC1 bias gradients of the transposed-formulation data-gradient kernel: per-workgroup sums in a slab per layer, added up
    // (assigned, not accumulated) where the LayerNorm partial sums are
    const float* b1_slabs[kMaxLayers]; float* b1_outs[kMaxLayers]; int b1_blocks[kMaxLayers]; int b1_n = 0;
    hipStream_t rs = s;                                   // stream of the partial-sum reductions (the side stream at the step's end)
    auto flush_b1 = [&]() -> int {
        const int st = fused_train_bias_reduce(b1_slabs, b1_outs, b1_blocks, b1_n, D4, rs);
        b1_n = 0;
        return st;
    };
    // launches the collected weight gradients behind everything issued on `s` so far.  (Measured and rejected, round 2: the
    // grouped launch of a layer on a side stream under the data gradients of the layers in front of it -- 3.63 vs 3.39 ms per
    // 1024-sample kitchen step, the two streams evict each other's operands from L2 / MALL.)
    uint32_t g_floats = 0;                                // floats of the collected problems in a range's slab
    // Order of the problems inside a launch: each XCD takes a contiguous run of the launch's tiles (xcd_tile), and a problem
    // whose tiles straddle two runs has its operand panels fetched into two L2s.  Units (problems sharing their B operand:
    // q | k | v of a layer) are packed into eight bins of ceil(tiles / 8), largest first, and emitted bin by bin -- kitchen:
    // six bins {FC2, FC1, out-projection} of 81 tiles and two of three q|k|v triples, against runs of 81 / 82.
    auto arrange_group = [&]() {
        const int n = gt.n;
        if (n < 3 || (flags & BESO_TRAIN_PLAN_PER_OP)) return;
        int ufirst[kMaxGroup], ucnt[kMaxGroup], utiles[kMaxGroup], order[kMaxGroup], bin_of[kMaxGroup], nu = 0;
        auto tiles_of = [&](const GProb& q) {
            if (panel_w) { int o; return wgrad_panel_tiles(q.Mo, q.No, panel_w, &o); }
            return q.nt_n * ((q.Mo + kTileMN - 1) / kTileMN);
        };
        for (int i = 0; i < n; ++i) {
            if (i > 0 && gt.p[i].B == gt.p[i - 1].B) { ++ucnt[nu - 1]; utiles[nu - 1] += tiles_of(gt.p[i]); }
            else { ufirst[nu] = i; ucnt[nu] = 1; utiles[nu] = tiles_of(gt.p[i]); ++nu; }
        }
        for (int u = 0; u < nu; ++u) {                    // (stable insertion sort, largest first)
            int j = u;
            while (j > 0 && utiles[order[j - 1]] < utiles[u]) { order[j] = order[j - 1]; --j; }
            order[j] = u;
        }
        const int cap = (g_tiles + 7) / 8;
        int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < nu; ++k) {
            const int u = order[k];
            int b = -1;
            for (int x = 0; x < 8 && b < 0; ++x) if (load[x] + utiles[u] <= cap) b = x;
            if (b < 0) { b = 0; for (int x = 1; x < 8; ++x) if (load[x] < load[b]) b = x; }
            bin_of[u] = b; load[b] += utiles[u];
        }
        GTable out;
        out.n = 0;
        int tiles = 0; uint32_t floats = 0;
        for (int b = 0; b < 8; ++b)
            for (int k = 0; k < nu; ++k) {
                const int u = order[k];
                if (bin_of[u] != b) continue;
                for (int i = ufirst[u]; i < ufirst[u] + ucnt[u]; ++i) {
                    GProb q = gt.p[i];
                    q.tile_begin = tiles; q.slab_off = floats;
                    tiles += tiles_of(q);
                    floats += (uint32_t)round_up(q.Mo * q.No + (q.bias ? q.Mo : 0), 4);
                    out.p[out.n++] = q;
                }
            }
        gt = out;
    };
    auto flush_group = [&]() -> hipError_t {
        if (gt.n == 0) return hipSuccess;
        arrange_group();
        if (panel_w) {
            const int psp = (w.w_splits_panel > 1 && g_floats <= w.wslab_floats) ? w.w_splits_panel : 1;
            static LdsAttrT attr2, attr3;
            hipError_t e = hipSuccess;
            if (panel_w == 2) {
                e = ensure_lds_t((const void*)wgrad_panel_group_kernel<2>, wgrad_panel_lds(2), &attr2);
                if (e == hipSuccess)
                    hipLaunchKernelGGL(wgrad_panel_group_kernel<2>, dim3(g_tiles * psp), dim3(kGT), wgrad_panel_lds(2), s, gt, g_tiles, psp,
                                       F(w.wslab), w.wslab_floats);
            } else {
                e = ensure_lds_t((const void*)wgrad_panel_group_kernel<3>, wgrad_panel_lds(3), &attr3);
                if (e == hipSuccess)
                    hipLaunchKernelGGL(wgrad_panel_group_kernel<3>, dim3(g_tiles * psp), dim3(kGT), wgrad_panel_lds(3), s, gt, g_tiles, psp,
                                       F(w.wslab), w.wslab_floats);
            }
            if (e != hipSuccess) return e;
            if (psp > 1)
                hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((g_floats + 1023) / 1024), dim3(256), 0, s, gt, (const float*)F(w.wslab),
                                   w.wslab_floats, psp, g_floats);
            gt.n = 0; g_tiles = 0; g_floats = 0;
            return hipGetLastError();
        }
        const int sp = (w.w_splits > 1 && g_floats <= w.wslab_floats) ? w.w_splits : 1;
        // Long contractions (M >= ~18 k token rows) run as one launch per ROW WINDOW of ~12 k rows, window w > 0 adding to the
        // outputs of the windows before it (stream order: deterministic).  The tiles of a weight gradient share operand panels
        // through an XCD's L2, but the sharers drift apart over a long contraction -- FETCH_SIZE 24 GB for 6.2 GB of operands at
        // 8192 kitchen samples (90 k rows) against 1.9 GB for 0.78 GB at 1024 -- and a launch boundary lines them up again:
        // 14.6 -> 14.2 ms per 8192-sample step with 8 windows (4: 14.3, 16: 14.25, 32: 14.6); nothing to gain at 11 k rows.
        int n_win = (sp > 1 || (flags & BESO_TRAIN_PLAN_PER_OP)) ? 1 : (M + 6144) / 12288;
        n_win = n_win < 1 ? 1 : (n_win > 16 ? 16 : n_win);
        for (int win = 0; win < n_win; ++win)
        hipLaunchKernelGGL(tgemm_wgrad_group_kernel<E>, dim3(g_tiles * sp), dim3(kGT), 0, s, gt, g_tiles, sp, F(w.wslab),
                           w.wslab_floats, win, n_win);
        if (sp > 1)
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((g_floats + 1023) / 1024), dim3(256), 0, s, gt, (const float*)F(w.wslab),
                               w.wslab_floats, sp, g_floats);
        gt.n = 0; g_tiles = 0; g_floats = 0;
        return hipGetLastError();
    };
    // bias: the gradient of the bias that goes with this weight = the column sums of A.  It rides along as a column of ones
    // behind B's last real column (free whenever No is not a multiple of the 128-column tile: every shipped shape); otherwise
    // a colsum launch of its own.
    auto wgrad = [&](const E* A, int lda, int Mo, const E* B, int ldb, int No, int rows, float* out, float* bias = nullptr) -> hipError_t {
        if (gt.n == kMaxGroup) { hipError_t e = flush_group(); if (e != hipSuccess) return e; }
        if (bias && No % kTileMN == 0) {
            hipError_t e = colsum(A, lda, Mo, rows, bias);
            if (e != hipSuccess) return e;
            bias = nullptr;
        }
        if (panel_w) {
            // (the ones column rides in the tile that covers column No: n-wide tiles always hold it unless No == 128 W)
            if (bias && (No <= 128 * panel_w ? No == 128 * panel_w : No % kTileMN == 0)) {
                hipError_t e = colsum(A, lda, Mo, rows, bias);
                if (e != hipSuccess) return e;
                bias = nullptr;
            }
            int orient = 0;
            const int nt = wgrad_panel_tiles(Mo, No, panel_w, &orient);
            gt.p[gt.n++] = GProb{A, B, out, bias, lda, ldb, Mo, No, g_tiles, orient, rows, g_floats};
            g_tiles += nt;
            g_floats += (uint32_t)round_up(Mo * No + (bias ? Mo : 0), 4);
            return hipSuccess;
        }
        const int nt_n = (No + kTileMN - 1) / kTileMN, nt_m = (Mo + kTileMN - 1) / kTileMN;
        gt.p[gt.n++] = GProb{A, B, out, bias, lda, ldb, Mo, No, g_tiles, nt_n, rows, g_floats};
        g_tiles += nt_n * nt_m;
        g_floats += (uint32_t)round_up(Mo * No + (bias ? Mo : 0), 4);
        return hipSuccess;
    };
    // head (compact rows): dW = dpred^T xf (the act real rows of dpred's ap columns, straight into the gradient tensor),
    // db = its column sums, dxf = dpred W
    if (mlp_head) {
        TRY(colsum(P(w.dpred), ap, act, Ma, hb.g));
        // second layer: dW1 = dpred^T a, da = dpred W1 -> dz = da * SiLU'(z) (+ its column sums = db0); first layer:
        // dW0 = dz^T xf, dxf = dz W0.  Padded rows / columns are zeros all the way.
        TRY(wgrad(P(w.dpred), ap, ap, P(w.ha), Hp, Hp, Ma, F(w.dw_head)));
        TRY((tgemm<E, false, true>(P(w.dpred), ap, P(w.w_head), Hp, Ma, Hp, ap, 1, EpiSiluBwd<E>{P(w.hz), P(w.hdz), F(w.db_hid), Hp}, s)));
        TRY(wgrad(P(w.hdz), Hp, Hp, P(w.xf), D, D, Ma, F(w.dw_hid)));
        TRY((tgemm<E, false, true>(P(w.hdz), Hp, P(w.w_hid), D, Ma, D, Hp, 1, EpiStore<E>{F(w.dxn), nullptr, nullptr, D}, s)));
    } else {
        TRY(wgrad(P(w.dpred), ap, act, P(w.xf), D, D, Ma, hw.g, hb.g));
        TRY((tgemm<E, false, true>(P(w.dpred), ap, P(w.w_head), D, Ma, D, ap, 1, EpiStore<E>{F(w.dxn), nullptr, nullptr, D}, s)));
    }
    TRY(ln_bwd(x_last, w.stf, lnfw.p, nullptr, F(w.dxa), P(w.layer[L - 1].dyo), Ma, lnfw.g, lnfb.g, lp[L - 1].f2b.g, resid_p,
               (uint32_t)(4 * (L - 1) + 2)));
    // (Measured and rejected, round 2: the chain of data gradients between two attention backwards as ONE tile kernel, the mirror
    // image of the tail-block forward -- parity-green, 209 us per launch against 164 us + gaps for the eight per-op launches it
    // replaces: 3.60 vs 3.43 ms per 1024-sample step.  DESIGN.md section 7.)
    // (status of a fused.hip launch: BESO_OK or the error the caller reports)
#define FUSED(call) do { const int st_ = (call); if (st_ != BESO_OK) { *err = hipGetLastError(); *err_line = __LINE__; return st_; } } while (0)
    for (int l = L - 1; l >= 0; --l) {
        const TrainLayerWs& y = w.layer[l];
        const float* x_in = l == 0 ? F(w.x0) : F(w.layer[l - 1].x_out);
        const bool last = l == L - 1;
        const int rows = last ? Ma : M;
        float* dres = last ? F(w.dxa) : F(w.dx);          // residual gradient of this layer's second half
        // FC2: dW2 = dyo^T g, dh = (dyo W2) * GELU'(h)
        TRY(wgrad(P(y.dyo), D, D, P(y.g), D4, D4, rows, lp[l].f2w.g));
        E* dy_out = last ? P(w.dya) : P(w.dy);
        float* slab = F(w.b1slab) + (size_t)l * fused_train_dgrad_blocks(M) * D4;       // FC1 bias sums per workgroup (use_dgrad)
        if (use_dgrad) { b1_slabs[b1_n] = slab; b1_outs[b1_n] = lp[l].f1b.g; b1_blocks[b1_n] = fused_train_dgrad_blocks(rows); ++b1_n; }
        if (use_mlp_bwd) {
            // FC2 + GELU' -> FC1 -> LayerNorm-2 backward -> out-projection: one launch, dh and dym stay in LDS between the GEMMs
            float* part = F(w.ln_part) + (size_t)ln_calls * lnb_grid * 3 * D;
            lrt.nb[ln_calls] = fused_train_dgrad_blocks(rows);
            lrt.c[ln_calls++] = LnRedCall{lp[l].ln2w.g, lp[l].ln2b.g, lp[l].pb.g};
            const TrainLnBwd ln{F(y.x_mid), (const float*)F(y.st2), lp[l].ln2w.p, dres, dres, P(y.dym), part, resid_p, seed,
                                (uint32_t)(4 * l + 1), 0, x16};
            FUSED(fused_train_mlp_bwd(flay, ws + w.bimg, l, rows, P(y.dyo), P(y.h), P(y.dh), slab, dy_out, ln, s));
        } else if (use_dgrad) {
            // the same chain as three launches (shapes without the one-kernel form)
            FUSED(fused_train_dgrad(flay, ws + w.bimg, l, 3, rows, P(y.dyo), nullptr, nullptr, P(y.h), P(y.dh), slab, s));
            FUSED(dgrad_ln(l, 2, rows, P(y.dh), F(y.x_mid), y.st2, lp[l].ln2w.p, dres, P(y.dym), lp[l].ln2w.g, lp[l].ln2b.g, lp[l].pb.g,
                           resid_p, (uint32_t)(4 * l + 1), 0, x16));
            FUSED(fused_train_dgrad(flay, ws + w.bimg, l, 1, rows, P(y.dym), nullptr, dy_out, nullptr, nullptr, nullptr, s));
        } else {
            // per-op: dh = (dyo W2) * GELU'(h) (+ db1), dxn2 = dh W1, LayerNorm-2 backward, dy = dym Wp
            TRY((tgemm<E, false, true>(P(y.dyo), D, P(y.w_fc2), D4, rows, D4, D, 1, EpiGeluBwd<E>{P(y.h), P(y.dh), lp[l].f1b.g, D4}, s)));
            TRY((tgemm<E, false, true>(P(y.dh), D4, P(y.w_fc1), D, rows, D, D4, 1, EpiStore<E>{F(w.dxn), nullptr, nullptr, D}, s)));
            TRY(ln_bwd(F(y.x_mid), y.st2, lp[l].ln2w.p, dres, dres, P(y.dym), rows, lp[l].ln2w.g, lp[l].ln2b.g, lp[l].pb.g, resid_p,
                       (uint32_t)(4 * l + 1)));
            TRY((tgemm<E, false, true>(P(y.dym), D, P(y.w_proj), D, rows, D, D, 1, EpiStore<E>{nullptr, dy_out, nullptr, D}, s)));
        }
        // FC1: dW1 = dh^T xn2; proj: dWp = dym^T y
        TRY(wgrad(P(y.dh), D4, D4, P(y.xn2), D, D, rows, lp[l].f1w.g));
        TRY(wgrad(P(y.dym), D, D, last ? P(w.ya) : P(y.y), D, D, rows, lp[l].pw.g));
        if (last) {
            // back to all token rows: dy and the residual gradient are zero off the action rows
            const size_t n4 = (size_t)M * (D / 4);
            hipLaunchKernelGGL(scatter_rows2_kernel<E>, dim3(gs_grid(n4)), dim3(256), 0, s, (const E*)P(w.dya), P(w.dy),
                               (const float*)F(w.dxa), F(w.dx), M, t, T, G, D);
            TRY(hipGetLastError());
        }
        if (attn_small && sizeof(E) == 2 && !(flags & BESO_TRAIN_PLAN_PER_OP)) {
            const int n_pairs = batch * H;
            hipLaunchKernelGGL(attn_mfma_bwd_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, s, (const uint16_t*)P(y.qkv),
                               (const uint16_t*)P(w.dy), (uint16_t*)P(y.dqkv), n_pairs, T, D, H, hd, scale, attn_p, attn_ik, seed,
                               (uint32_t)(4 * l));
        } else if (attn_small) {
#define ATT(TT) hipLaunchKernelGGL((attn_small_kernel<E, true, TT>), dim3(batch * H), dim3(64), 0, s, (const E*)P(y.qkv), \
                                   (const E*)P(w.dy), P(y.dqkv), T, D, H, hd, scale, attn_p, attn_ik, seed, (uint32_t)(4 * l))
            if (T <= 8) ATT(8); else if (T <= 12) ATT(12); else ATT(16);
#undef ATT
        } else
            hipLaunchKernelGGL(attn_bwd_kernel<E>, dim3(batch * H), dim3(64), lds_b, s, (const E*)P(y.qkv), (const E*)P(w.dy),
                               P(y.dqkv), T, D, H, hd, scale, attn_p, attn_ik, seed, (uint32_t)(4 * l));
        TRY(hipGetLastError());
        // q/k/v: three weight gradients from the column blocks of dqkv, bias gradients, dxn1 = dqkv Wqkv
        TRY(wgrad(P(y.dqkv), D3, D, P(y.xn1), D, D, M, lp[l].qw.g, lp[l].qb.g));
        TRY(wgrad(P(y.dqkv) + D, D3, D, P(y.xn1), D, D, M, lp[l].kw.g, lp[l].kb.g));
        TRY(wgrad(P(y.dqkv) + 2 * D, D3, D, P(y.xn1), D, D, M, lp[l].vw.g, lp[l].vb.g));
        const bool first = l == 0;
        if (use_dgrad) {
            FUSED(dgrad_ln(l, 0, M, P(y.dqkv), x_in, y.st1, lp[l].ln1w.p, F(w.dx), first ? P(w.dx0b) : P(w.layer[l - 1].dyo),
                           lp[l].ln1w.g, lp[l].ln1b.g, first ? nullptr : lp[l - 1].f2b.g, first ? embed_p : resid_p,
                           first ? kEmbedSite : (uint32_t)(4 * (l - 1) + 2), first ? T : 0, first ? 0 : x16));
        } else {
            TRY((tgemm<E, false, true>(P(y.dqkv), D3, P(y.w_qkv), D, M, D, D3, 1, EpiStore<E>{F(w.dxn), nullptr, nullptr, D}, s)));
            TRY(ln_bwd(x_in, y.st1, lp[l].ln1w.p, F(w.dx), F(w.dx), first ? P(w.dx0b) : P(w.layer[l - 1].dyo), M, lp[l].ln1w.g,
                       lp[l].ln1b.g, first ? nullptr : lp[l - 1].f2b.g, first ? embed_p : resid_p,
                       first ? kEmbedSite : (uint32_t)(4 * (l - 1) + 2), first ? T : 0));
        }
        if (early_stream && l == train_early_layer(c) && l > 0) {
            // the gradients of layers l .. L-1 and ln_f are complete once their weight gradients and LayerNorm sums
            // have run: do those now and order `early_stream` behind this point (the C1 exchange of that range can
            // start under the backward of layers l-1 .. 0)
            TRY(flush_group());
            if (flush_b1() != BESO_OK) { *err = hipGetLastError(); *err_line = __LINE__; return BESO_ERR_HIP; }
            hipLaunchKernelGGL(ln_reduce_kernel, dim3((D + 63) / 64, 3, ln_calls), dim3(256), 0, s, (const float*)F(w.ln_part),
                               lrt, lnb_grid, D, 0);
            TRY(hipGetLastError());
            ln_reduced = ln_calls;
            hipEvent_t ev = nullptr;
            TRY(step_event(kEvEarly, &ev));
            TRY(hipEventRecord(ev, s));
            TRY(hipStreamWaitEvent(early_stream, ev, 0));
        }
    }
#undef FUSED
    // embeddings: dWcat[Ke][D] = Xemb^T dx0, routed to pos_emb / tok_emb / action_emb / sigma_emb after the launch
    TRY(wgrad(P(w.xemb), Ke, Ke, P(w.dx0b), D, D, M, F(w.dw_cat)));
    // The reductions of the FC1-bias slabs and of the LayerNorm / bias partial sums read what the data-gradient kernels wrote
    // and write gradient tensors of their own: nothing ties them to the grouped weight-gradient launch, which leaves 38 of the
    // 256 CUs idle (218 panel tiles) -- they run BESIDE it on the library's side stream (three launches of 12 + 12 + 24 us off the
    // step's chain; a dependent launch costs ~5 us whatever it does: tools/microbench/launch_chain).
    hipEvent_t ev_sf = nullptr, ev_sj = nullptr;
    if (panel_w) {
        TRY(step_side_stream(&rs));
        TRY(step_event(kEvSideFork, &ev_sf));
        TRY(step_event(kEvSideJoin, &ev_sj));
        TRY(hipEventRecord(ev_sf, s));
        TRY(hipStreamWaitEvent(rs, ev_sf, 0));
    } else TRY(flush_group());
    if (flush_b1() != BESO_OK) { *err = hipGetLastError(); *err_line = __LINE__; return BESO_ERR_HIP; }
    hipLaunchKernelGGL(ln_reduce_kernel, dim3((D + 63) / 64, 3, ln_calls - ln_reduced), dim3(256), 0, rs,
                       (const float*)F(w.ln_part), lrt, lnb_grid, D, ln_reduced);
    TRY(hipGetLastError());
    if (panel_w) {
        TRY(hipEventRecord(ev_sj, rs));
        TRY(flush_group());
        TRY(hipStreamWaitEvent(s, ev_sj, 0));
        rs = s;
    }
    if (mlp_head) {
        TRY(hipMemcpy2DAsync(hw.g, sizeof(float) * Hh, ws + w.dw_head, sizeof(float) * Hp, sizeof(float) * Hh, act,
                             hipMemcpyDeviceToDevice, s));                                    // [act][Hp] -> [act][100]
        TRY(hipMemcpyAsync(h0w.g, ws + w.dw_hid, sizeof(float) * (size_t)Hh * D, hipMemcpyDeviceToDevice, s));
        TRY(hipMemcpyAsync(h0b.g, ws + w.db_hid, sizeof(float) * Hh, hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(scatter_emb_kernel, dim3(64), dim3(256), 0, s, (const float*)F(w.dw_cat), pos.g, tokw.g, tokb.g, sigw.g,
                       sigb.g, actw.g, actb.g, D, obs, act, seq);
    TRY(hipGetLastError());
    return BESO_OK;
}

int train_loss_grad(const beso_config* c, const float* const* params, int n_params, float* grads_flat, int precision,
                    const float* state, const float* action, const float* goal, const float* noise, const float* sigma,
                    float* loss_out, int batch, int t, int flags, float embed_pdrop, float attn_pdrop, float resid_pdrop,
                    float goal_drop, uint32_t seed, float grad_scale,
                    void* workspace, size_t workspace_bytes, hipStream_t s, hipStream_t early_stream, hipStream_t loss_stream,
                    hipError_t* err, int* err_line) {
    int st = train_validate(c, batch, t);
    if (st != BESO_OK) return st;
    if (precision != BESO_PREC_BF16 && precision != BESO_PREC_FP32) return BESO_ERR_BAD_ARG;
    if (!params || !grads_flat || !state || !action || !noise || !sigma || !loss_out || !workspace) return BESO_ERR_BAD_ARG;
    if (c->goal_seq_len > 0 && !goal) return BESO_ERR_BAD_ARG;
    if (n_params != 3 + 16 * c->n_layers + 6 + (c->linear_output ? 2 : 4)) return BESO_ERR_BAD_ARG;
    for (int i = 0; i < n_params; ++i) if (!params[i]) return BESO_ERR_BAD_ARG;
    if (flags & ~(BESO_TRAIN_LAST_ACTION_ONLY | BESO_TRAIN_PLAN_PER_OP | BESO_TRAIN_PLAN_TILES)) return BESO_ERR_BAD_ARG;
    if (!(attn_pdrop >= 0.f && attn_pdrop < 1.f && resid_pdrop >= 0.f && resid_pdrop < 1.f && embed_pdrop >= 0.f &&
          embed_pdrop < 1.f && goal_drop >= 0.f && goal_drop <= 1.f))
        return BESO_ERR_BAD_ARG;
    TrainWs w;
    make_train_ws(c, batch, t, precision, &w);
    if (workspace_bytes < w.total) return BESO_ERR_WORKSPACE;
    if (precision == BESO_PREC_FP32)
        return loss_grad_e<float>(c, params, grads_flat, precision, state, action, goal, noise, sigma, loss_out, batch, t,
                                  flags, embed_pdrop, attn_pdrop, resid_pdrop, goal_drop, seed, grad_scale,
                                  (char*)workspace, w, s, early_stream, loss_stream, err, err_line);
    return loss_grad_e<uint16_t>(c, params, grads_flat, precision, state, action, goal, noise, sigma, loss_out, batch, t,
                                 flags, embed_pdrop, attn_pdrop, resid_pdrop, goal_drop, seed, grad_scale,
                                 (char*)workspace, w, s, early_stream, loss_stream, err, err_line);
}

int train_goal_mask(float* mask, size_t n, float goal_drop, uint32_t seed, hipStream_t s, hipError_t* err, int* err_line) {
    if (!mask || !(goal_drop >= 0.f && goal_drop <= 1.f)) return BESO_ERR_BAD_ARG;
    if (n == 0) return BESO_OK;
    int grid = (int)((n + 255) / 256); if (grid > 2048) grid = 2048;
    (void)hipGetLastError();
    hipLaunchKernelGGL(goal_mask_kernel, dim3(grid), dim3(256), 0, s, mask, n, goal_drop, seed);
    TRY(hipGetLastError());
    return BESO_OK;
}

#if BESO_DEV_API
// development aid: C[M][N] = op(A) op(B)^T through tgemm (fp32 output), for the layout tests
int train_debug_gemm(int precision, int a_kslow, int b_kslow, const void* A, int lda, const void* B, int ldb, float* C,
                     int ldc, int M, int N, int K, int splits, hipStream_t s, hipError_t* err, int* err_line) {
    if (!A || !B || !C) return BESO_ERR_BAD_ARG;
    if (precision == BESO_PREC_BF16 && ((a_kslow == 1 && (b_kslow == 2 || b_kslow == 3)) || (b_kslow == 1 && (a_kslow == 2 || a_kslow == 3)))) {
        // the panel-owning weight-gradient tiles: (1, W) = tiles of 128 rows x all N <= 128 W columns, (W, 1) = all M <= 128 W rows
        // x 128 columns; both operands k-slow, C contiguous (ldc == N); splits > 1: row ranges into slabs behind C's M x N floats
        const int W = a_kslow > 1 ? a_kslow : b_kslow;
        if (ldc != N || (a_kslow > 1 ? M : N) > 128 * W || M % 8 || N % 8 || lda % 8 || ldb % 8) return BESO_ERR_BAD_ARG;
        GTable t;
        int orient = a_kslow > 1 ? -1 : 0;
        const int tiles = orient ? (N + kTileMN - 1) / kTileMN : (M + kTileMN - 1) / kTileMN;
        t.n = 1;
        t.p[0] = GProb{A, B, C, nullptr, lda, ldb, M, N, 0, orient, K, 0u};
        if (splits < 1) splits = 1;
        float* slab = C + (size_t)round_up(M * N, 4);
        (void)hipGetLastError();
        if (W == 2) {
            TRY(hipFuncSetAttribute((const void*)wgrad_panel_group_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wgrad_panel_lds(2)));
            hipLaunchKernelGGL(wgrad_panel_group_kernel<2>, dim3(tiles * splits), dim3(kGT), wgrad_panel_lds(2), s, t, tiles, splits, slab,
                               (size_t)round_up(M * N, 4));
        } else {
            TRY(hipFuncSetAttribute((const void*)wgrad_panel_group_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wgrad_panel_lds(3)));
            hipLaunchKernelGGL(wgrad_panel_group_kernel<3>, dim3(tiles * splits), dim3(kGT), wgrad_panel_lds(3), s, t, tiles, splits, slab,
                               (size_t)round_up(M * N, 4));
        }
        TRY(hipGetLastError());
        if (splits > 1) {
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((M * N + 1023) / 1024), dim3(256), 0, s, t, (const float*)slab,
                               (size_t)round_up(M * N, 4), splits, (uint32_t)(M * N));
            TRY(hipGetLastError());
        }
        return BESO_OK;
    }
#define DBG(E, AK, BK)                                                                                             \
    do {                                                                                                           \
        if (splits > 1) TRY((tgemm<E, AK, BK>(A, lda, B, ldb, M, N, K, splits, epi_atomic(C, ldc), s)));          \
        else TRY((tgemm<E, AK, BK>(A, lda, B, ldb, M, N, K, 1, EpiStore<E>{C, nullptr, nullptr, ldc}, s)));      \
        return BESO_OK;                                                                                            \
    } while (0)
    if (precision == BESO_PREC_FP32) {
        if (!a_kslow && !b_kslow) DBG(float, false, false);
        if (!a_kslow && b_kslow) DBG(float, false, true);
        if (a_kslow && b_kslow) DBG(float, true, true);
    } else if (precision == BESO_PREC_BF16) {
        if (!a_kslow && !b_kslow) DBG(uint16_t, false, false);
        if (!a_kslow && b_kslow) DBG(uint16_t, false, true);
        if (a_kslow && b_kslow) DBG(uint16_t, true, true);
    }
#undef DBG
    return BESO_ERR_UNSUPPORTED;
}
#endif

}  // namespace beso
