// K3 (generic path): causal self-attention over the T <= 1+G+2W tokens of one sample
// (score_gpts.py:69-76): att = softmax(mask(q k^T / sqrt(hd))), y = att v, heads re-merged.
// The causal mask covers the WHOLE sequence including the sigma and goal tokens (:42-47,70).
//
// Work is tiny (0.5 % of the network's FLOPs) and the kernel is bound by moving qkv through HBM, so
// the layout is built for coalescing: a block owns a run of (sample, head) pairs, copies their
// q/k/v rows ([T][hd] each, hd contiguous in HBM) into LDS with 8-byte units, computes one query
// row per thread from LDS (keys/values are broadcast reads within a pair; online softmax; q and the
// output row in registers), parks the output row in the q slot and streams it back out coalesced.
// qkv row layout: [q(D) | k(D) | v(D)].
#include "common.h"

namespace beso {

template <typename E> struct Unit8;   // 8-byte unit -> floats
template <> struct Unit8<uint16_t> {
    static constexpr int kElems = 4;
    __device__ static __forceinline__ void unpack(uint2 u, float* f) {
        f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
        f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    }
    __device__ static __forceinline__ uint2 pack(const float* f) {
        uint2 u;
        u.x = (uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16);
        u.y = (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16);
        return u;
    }
};
template <> struct Unit8<float> {
    static constexpr int kElems = 2;
    __device__ static __forceinline__ void unpack(uint2 u, float* f) {
        f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
    }
    __device__ static __forceinline__ uint2 pack(const float* f) {
        return make_uint2(__float_as_uint(f[0]), __float_as_uint(f[1]));
    }
};

// HDP: compile-time bound of the head dim (registers); requires (hd * sizeof(E)) % 8 == 0.
template <typename E, int HDP>
__global__ void attention_kernel(const E* __restrict__ qkv, E* __restrict__ y, int n_pairs, int T, int D, int H,
                                 int hd, int ld_y, float scale, int pairs_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int EPU = Unit8<E>::kElems;            // elements per 8-byte unit
    constexpr int UMAX = HDP / EPU;                  // units per row (compile-time bound)
    const int upr = hd / EPU;                        // units per row (runtime)
    uint2* s = (uint2*)smem;                         // [pair][3][T][upr]
    const int pair0 = blockIdx.x * pairs_per_block;
    const int npair = min(pairs_per_block, n_pairs - pair0);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const size_t ldq = (size_t)3 * D;

    // ---- HBM -> LDS, 8-byte units, rows of hd contiguous elements
    const int units = npair * 3 * T * upr;
    for (int u = tid; u < units; u += nthr) {
        int k = u % upr, r = (u / upr) % T, c = (u / (upr * T)) % 3, p = u / (upr * T * 3);
        int pair = pair0 + p, vb = pair / H, h = pair % H;
        const E* src = qkv + ((size_t)vb * T + r) * ldq + (size_t)c * D + (size_t)h * hd;
        s[u] = ((const uint2*)src)[k];
    }
    __syncthreads();

    // ---- one query row per thread
    for (int item = tid; item < npair * T; item += nthr) {
        const int p = item / T, i = item % T;
        uint2* sq = s + ((size_t)(p * 3 + 0) * T + i) * upr;
        const uint2* sk = s + (size_t)(p * 3 + 1) * T * upr;
        const uint2* sv = s + (size_t)(p * 3 + 2) * T * upr;
        float q[HDP], o[HDP];
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            if (u < upr) Unit8<E>::unpack(sq[u], &q[u * EPU]);
#pragma unroll
            for (int e = 0; e < EPU; ++e) o[u * EPU + e] = 0.f;
        }
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j <= i; ++j) {
            float sc = 0.f;
#pragma unroll
            for (int u = 0; u < UMAX; ++u) {
                if (u < upr) {
                    float kf[EPU];
                    Unit8<E>::unpack(sk[j * upr + u], kf);
#pragma unroll
                    for (int e = 0; e < EPU; ++e) sc = fmaf(q[u * EPU + e], kf[e], sc);
                }
            }
            sc *= scale;                                   // (q k^T) * 1/sqrt(hd)  (score_gpts.py:69)
            const float mn = fmaxf(m, sc);
            const float alpha = expf(m - mn);              // exp(-inf) = 0 on the first key
            const float pj = expf(sc - mn);
            l = l * alpha + pj;
#pragma unroll
            for (int u = 0; u < UMAX; ++u) {
                if (u < upr) {
                    float vf[EPU];
                    Unit8<E>::unpack(sv[j * upr + u], vf);
#pragma unroll
                    for (int e = 0; e < EPU; ++e) o[u * EPU + e] = fmaf(pj, vf[e], o[u * EPU + e] * alpha);
                }
            }
            m = mn;
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            if (u < upr) {
                float of[EPU];
#pragma unroll
                for (int e = 0; e < EPU; ++e) of[e] = o[u * EPU + e] * inv;
                sq[u] = Unit8<E>::pack(of);                // park the output row in this thread's own q slot
            }
        }
    }
    __syncthreads();

    // ---- LDS -> HBM, coalesced; heads re-merged side by side (score_gpts.py:74-76)
    const int ounits = npair * T * upr;
    for (int u = tid; u < ounits; u += nthr) {
        int k = u % upr, r = (u / upr) % T, p = u / (upr * T);
        int pair = pair0 + p, vb = pair / H, h = pair % H;
        E* dst = y + ((size_t)vb * T + r) * ld_y + (size_t)h * hd;
        ((uint2*)dst)[k] = s[((size_t)(p * 3 + 0) * T + r) * upr + k];
    }
    // zero the K padding of the rows (done by the block that owns head 0 of the sample)
    const int padw = ld_y - D;
    if (padw > 0) {
        for (int p = 0; p < npair; ++p) {
            int pair = pair0 + p;
            if (pair % H != 0) continue;
            int vb = pair / H;
            for (int u = tid; u < T * padw; u += nthr)
                y[((size_t)vb * T + u / padw) * ld_y + D + u % padw] = Act<E>::from(0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16, hd <= 64, 16 < T <= 128: the same attention on the matrix pipe, one (sample, head) pair per
// wave, up to four pairs per workgroup.  LDS per pair: q and k rows ([Tp][72] bf16: 144-byte rows, conflict-free
// 16-byte row reads) and v TRANSPOSED ([64][Tp + 8]) so that the A operand of the P.V product (lane =
// head dim, four consecutive keys) is one 8-byte read.
//   S^T[j][i] = sum_d K[j][d] Q[i][d]      v_mfma_f32_16x16x32_bf16 per (key tile <= query tile) x 2 k-steps
//   D layout: lane (i = lane&15, g) holds keys 4g + r of the key tile -> softmax over keys = in-lane over
//   r and the key tiles + two xor-shuffles over g; exp2 with the scale folded in
//   Y^T[d][i] = sum_j V[j][d] P[i][j]      v_mfma_f32_16x16x16_bf16 per (d tile, key tile)
// The scalar kernel above spends 3.0 of the 7.1 ms of a long-horizon forward (T = 67, B = 512).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
constexpr int kAttRow = 72;          // bf16 per q/k row in LDS
constexpr int kAttMaxTiles = 8;      // T <= 128

__device__ __forceinline__ uint32_t att_pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    f2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
}

__global__ __launch_bounds__(256) void attention_mfma_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ y,
                                                             int n_pairs, int T, int D, int H, int hd, int ld_y,
                                                             float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int nt = (T + 15) >> 4, Tp = nt * 16, vrow = Tp + 8;
    const size_t pair_bytes = (size_t)(2 * Tp * kAttRow + 64 * vrow) * 2;
    uint16_t* sq = (uint16_t*)(smem + wv * pair_bytes);
    uint16_t* sk = sq + Tp * kAttRow;
    uint16_t* svt = sk + Tp * kAttRow;
    const int pair = blockIdx.x * (blockDim.x >> 6) + wv;
    if (pair >= n_pairs) return;                  // whole wave; no workgroup barrier below
    const int vb = pair / H, h = pair % H;
    const size_t ldq = (size_t)3 * D;
    const uint16_t* base = qkv + (size_t)vb * T * ldq + (size_t)h * hd;
    // ---- HBM -> LDS: 16-byte units (8 head dims of one token); dims >= hd and tokens >= T are zero
    for (int u = lane; u < Tp * 8; u += 64) {
        const int tok = u >> 3, c = u & 7;
        u32x4 q4 = {0, 0, 0, 0}, k4 = q4, v4 = q4;
        if (tok < T && 8 * c < hd) {
            const uint16_t* row = base + (size_t)tok * ldq + 8 * c;
            q4 = *(const u32x4*)row;
            k4 = *(const u32x4*)(row + D);
            v4 = *(const u32x4*)(row + 2 * D);
        }
        *(u32x4*)(sq + tok * kAttRow + 8 * c) = q4;
        *(u32x4*)(sk + tok * kAttRow + 8 * c) = k4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            svt[(8 * c + 2 * e) * vrow + tok] = (uint16_t)(v4[e] & 0xffffu);
            svt[(8 * c + 2 * e + 1) * vrow + tok] = (uint16_t)(v4[e] >> 16);
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): this wave's LDS writes (the data is wave-private)
    __builtin_amdgcn_wave_barrier();
    for (int qi = 0; qi < nt; ++qi) {
        u32x4 qf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[kk] = *(const u32x4*)(sq + (16 * qi + n) * kAttRow + 32 * kk + 8 * g);
        f32x4 sT[kAttMaxTiles];
        float m = -INFINITY;
        const int qtok = 16 * qi + n;
#pragma unroll
        for (int kj = 0; kj < kAttMaxTiles; ++kj) {
            if (kj <= qi) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const u32x4 kf = *(const u32x4*)(sk + (16 * kj + n) * kAttRow + 32 * kk + 8 * g);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf),
                                                                  __builtin_bit_cast(bf16x8, qf[kk]), acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kj + 4 * g + r;
                    acc[r] = (key <= qtok && key < T) ? acc[r] * scale_log2e : -INFINITY;   // causal over the whole sequence
                    m = fmaxf(m, acc[r]);
                }
                sT[kj] = acc;
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
        uint2 pb[kAttMaxTiles];
#pragma unroll
        for (int kj = 0; kj < kAttMaxTiles; ++kj) {
            if (kj <= qi) {
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(sT[kj][r] - m); sum += e[r]; }
                pb[kj] = make_uint2(att_pack2(e[0], e[1]), att_pack2(e[2], e[3]));
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;             // key 0 is never masked: sum >= 1 after the max shift
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            if (16 * dt >= hd) break;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kj = 0; kj < kAttMaxTiles; ++kj) {
                if (kj <= qi) {
                    const uint2 va = *(const uint2*)(svt + (16 * dt + n) * vrow + 16 * kj + 4 * g);
                    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, va),
                                                                    __builtin_bit_cast(s16x4_t, pb[kj]), acc, 0, 0, 0);
                }
            }
            // D layout: lane (query i = n, g) holds head dims 16dt + 4g .. +3: heads re-merged side by side
            const int d0 = 16 * dt + 4 * g;
            if (qtok < T && d0 < hd) {
                uint2 o = make_uint2(att_pack2(acc[0] * inv, acc[1] * inv), att_pack2(acc[2] * inv, acc[3] * inv));
                *(uint2*)(y + ((size_t)vb * T + qtok) * ld_y + (size_t)h * hd + d0) = o;
            }
        }
        if (h == 0 && ld_y > D && qtok < T)        // zero the K padding of the row (once per sample)
            for (int c = D + g; c < ld_y; c += 4) y[((size_t)vb * T + qtok) * ld_y + c] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// BF16X3 (round 6): the same matrix-pipe attention in SPLIT-bf16 arithmetic on fp32 rows -- the attention of the long-sequence
// shape's 1e-4 mode, which exchanges q | k | v and y as fp32 (fused_lin_x3) and ran attention_kernel<float> above: one query row
// per thread, a serial loop over the keys -- 591 us per layer at 256 samples of 67 tokens, 54 % of that mode's forward
// (profiles/r06_x3_stats.txt).  Every operand is a (hi, lo) pair of bf16 planes in LDS, hi = bf16(v), lo = bf16(v - hi); a product
// is three MFMAs, small terms first (lo hi + hi lo + hi hi, fp32 accumulate: the GEMMs' scheme, fused.hip gemm_x3); the
// probabilities are split the same way for P.V; softmax in fp32 (exp2 with the scale folded in).  One (sample, head) pair per
// WORKGROUP of four waves; q, k and v as row-major planes (V^T fragments through ds_read_b64_tr_b16): 69 KB at T = 67, two
// workgroups per CU.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void att_split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t h = att_pack2(v[2 * e], v[2 * e + 1]);
        hi[e] = h;
        lo[e] = att_pack2(v[2 * e] - __uint_as_float(h << 16), v[2 * e + 1] - __uint_as_float(h & 0xffff0000u));
    }
}

__global__ __launch_bounds__(256) void attention_mfma_x3_kernel(const float* __restrict__ qkv, float* __restrict__ y, int n_pairs,
                                                                int T, int D, int H, int hd, int ld_y, float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int nt = (T + 15) >> 4, Tp = nt * 16;
    const size_t plane = (size_t)3 * Tp * kAttRow;                        // bf16 elements of one (q | k | v) plane: row-major, 144-byte rows
    uint16_t* sq = (uint16_t*)smem;                                       // hi plane, the lo plane `plane` elements behind it
    uint16_t* sk = sq + Tp * kAttRow;
    uint16_t* sv = sk + Tp * kAttRow;                                     // (V stays row-major: its fragments come through the transpose read)
    // ONE pair per workgroup, its FOUR waves share the planes and deal the query tiles among themselves (a query tile qi costs
    // qi + 1 key tiles: dealt in snake order -- 4, 3, 2, 1 | 0 for the five tiles of 67 tokens).  One pair per WAVE (the bf16
    // kernel's form) left a CU with two waves at 137 KB of LDS: 70 us per layer at 256 samples for ~6 us of MFMA chain per pair.
    const int pair = blockIdx.x;
    const int vb = pair / H, h = pair % H;
    const size_t ldq = (size_t)3 * D;
    const float* base = qkv + (size_t)vb * T * ldq + (size_t)h * hd;
    // ---- HBM -> LDS: units of 8 head dims of one token, split into the two planes; dims >= hd and tokens >= T are zero.  Four
    // units (24 loads) in flight per lane: one unit at a time the loop was ten dependent memory round trips per pair -- most of
    // the kernel's time (clamped addresses, values selected afterwards: no load under a branch)
    constexpr int kU = 3;
    for (int u0 = threadIdx.x; u0 < Tp * 8; u0 += 256 * kU) {
        f32x4 r[kU][6];
#pragma unroll
        for (int j = 0; j < kU; ++j) {
            const int u = min(u0 + 256 * j, Tp * 8 - 1), tok = min(u >> 3, T - 1), c = u & 7;
            const float* row = base + (size_t)tok * ldq + min(8 * c, hd - 8);
            r[j][0] = *(const f32x4*)row; r[j][1] = *(const f32x4*)(row + 4);
            r[j][2] = *(const f32x4*)(row + D); r[j][3] = *(const f32x4*)(row + D + 4);
            r[j][4] = *(const f32x4*)(row + 2 * D); r[j][5] = *(const f32x4*)(row + 2 * D + 4);
        }
#pragma unroll
        for (int j = 0; j < kU; ++j) {
            const int u = u0 + 256 * j;
            if (u < Tp * 8) {
                const int tok = u >> 3, c = u & 7;
                const bool ok = tok < T && 8 * c < hd;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                u32x4 hi, lo;
                att_split8(ok ? r[j][0] : z, ok ? r[j][1] : z, hi, lo);
                *(u32x4*)(sq + tok * kAttRow + 8 * c) = hi; *(u32x4*)(sq + plane + tok * kAttRow + 8 * c) = lo;
                att_split8(ok ? r[j][2] : z, ok ? r[j][3] : z, hi, lo);
                *(u32x4*)(sk + tok * kAttRow + 8 * c) = hi; *(u32x4*)(sk + plane + tok * kAttRow + 8 * c) = lo;
                att_split8(ok ? r[j][4] : z, ok ? r[j][5] : z, hi, lo);
                *(u32x4*)(sv + tok * kAttRow + 8 * c) = hi; *(u32x4*)(sv + plane + tok * kAttRow + 8 * c) = lo;
            }
        }
    }
    __syncthreads();
    auto mma32 = [](f32x4 acc, const u32x4& a, const u32x4& b) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    };
    auto mma16 = [](f32x4 acc, const uint2& a, const uint2& b) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), acc, 0, 0, 0);
    };
    for (int it = 0; 4 * it < nt; ++it) {
        // snake: pass `it` deals tiles nt-1-4it .. downwards to waves 0..3 on even passes, upwards on odd ones
        const int qi = nt - 1 - 4 * it - ((it & 1) ? 3 - wv : wv);
        if (qi < 0) break;                        // (wave-uniform)
        u32x4 qh[2], ql[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qh[kk] = *(const u32x4*)(sq + (16 * qi + n) * kAttRow + 32 * kk + 8 * g);
            ql[kk] = *(const u32x4*)(sq + plane + (16 * qi + n) * kAttRow + 32 * kk + 8 * g);
        }
        f32x4 sT[kAttMaxTiles];
        float m = -INFINITY;
        const int qtok = 16 * qi + n;
#pragma unroll
        for (int kj = 0; kj < kAttMaxTiles; ++kj) {
            if (kj <= qi) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                u32x4 kh[2], kl[2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    kh[kk] = *(const u32x4*)(sk + (16 * kj + n) * kAttRow + 32 * kk + 8 * g);
                    kl[kk] = *(const u32x4*)(sk + plane + (16 * kj + n) * kAttRow + 32 * kk + 8 * g);
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) acc = mma32(acc, kl[kk], qh[kk]);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) acc = mma32(acc, kh[kk], ql[kk]);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) acc = mma32(acc, kh[kk], qh[kk]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kj + 4 * g + r;
                    acc[r] = (key <= qtok && key < T) ? acc[r] * scale_log2e : -INFINITY;   // causal over the whole sequence
                    m = fmaxf(m, acc[r]);
                }
                sT[kj] = acc;
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
        uint2 ph[kAttMaxTiles], pl[kAttMaxTiles];
#pragma unroll
        for (int kj = 0; kj < kAttMaxTiles; ++kj) {
            if (kj <= qi) {
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { e[r] = exp2f(sT[kj][r] - m); sum += e[r]; }
                ph[kj] = make_uint2(att_pack2(e[0], e[1]), att_pack2(e[2], e[3]));
                pl[kj] = make_uint2(att_pack2(e[0] - __uint_as_float(ph[kj].x << 16), e[1] - __uint_as_float(ph[kj].x & 0xffff0000u)),
                                    att_pack2(e[2] - __uint_as_float(ph[kj].y << 16), e[3] - __uint_as_float(ph[kj].y & 0xffff0000u)));
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;             // key 0 is never masked: sum >= 1 after the max shift
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            if (16 * dt >= hd) break;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kj = 0; kj < kAttMaxTiles; ++kj) {
                if (kj <= qi) {
                    // A operand of Y^T = V^T P^T: lane (d = 16 dt + n, key group g) holds V[16 kj + 4 g + r][d], r = 0 .. 3 -- gfx950's
                    // transpose read: the 16 lanes of a group address the [4 keys][16 dims] block row by row and each receives
                    // column n of it (fused.hip v_frag)
                    typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4;
                    const uint16_t* pv = sv + (16 * kj + 4 * g + (n >> 2)) * kAttRow + 16 * dt + 4 * (n & 3);
                    const uint2 vh = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pv)));
                    const uint2 vl = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pv + plane)));
                    acc = mma16(acc, vl, ph[kj]);
                    acc = mma16(acc, vh, pl[kj]);
                    acc = mma16(acc, vh, ph[kj]);
                }
            }
            // D layout: lane (query i = n, g) holds head dims 16dt + 4g .. +3: heads re-merged side by side
            const int d0 = 16 * dt + 4 * g;
            if (qtok < T && d0 < hd) *(f32x4*)(y + ((size_t)vb * T + qtok) * ld_y + (size_t)h * hd + d0) = acc * inv;
        }
        if (h == 0 && ld_y > D && qtok < T)        // zero the K padding of the row (once per sample)
            for (int c = D + g; c < ld_y; c += 4) y[((size_t)vb * T + qtok) * ld_y + c] = 0.f;
    }
}

// Fallback for head dims whose rows are not 8-byte multiples: one thread per query row straight
// from HBM (slow; no shipped configuration takes it).
template <typename E, int HDP>
__global__ void attention_rowwise_kernel(const E* __restrict__ qkv, E* __restrict__ y, int vbatch, int T, int D,
                                         int H, int hd, int ld_y, float scale) {
    int item = blockIdx.x * blockDim.x + threadIdx.x;     // ((vb*H + h)*T + i)
    if (item >= vbatch * H * T) return;
    int i = item % T, h = (item / T) % H, vb = item / (T * H);
    size_t ldq = (size_t)3 * D;
    const E* base = qkv + (size_t)vb * T * ldq + (size_t)h * hd;
    float q[HDP], o[HDP];
    const E* qr = base + (size_t)i * ldq;
#pragma unroll
    for (int d = 0; d < HDP; ++d) { q[d] = (d < hd) ? Act<E>::to(qr[d]) : 0.f; o[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j <= i; ++j) {
        const E* kr = base + (size_t)j * ldq + D;
        const E* vr = kr + D;
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < HDP; ++d) if (d < hd) sc = fmaf(q[d], Act<E>::to(kr[d]), sc);
        sc *= scale;
        float mn = fmaxf(m, sc), alpha = expf(m - mn), pj = expf(sc - mn);
        l = l * alpha + pj;
#pragma unroll
        for (int d = 0; d < HDP; ++d) if (d < hd) o[d] = fmaf(pj, Act<E>::to(vr[d]), o[d] * alpha);
        m = mn;
    }
    float inv = 1.0f / l;
    E* yr = y + ((size_t)vb * T + i) * ld_y + (size_t)h * hd;
#pragma unroll
    for (int d = 0; d < HDP; ++d) if (d < hd) yr[d] = Act<E>::from(o[d] * inv);
    if (h == 0) {
        E* pad = y + ((size_t)vb * T + i) * ld_y;
        for (int c = D; c < ld_y; ++c) pad[c] = Act<E>::from(0.f);
    }
}

template <typename E, int HDP>
static hipError_t launch_hdp(const void* qkv, void* y, int vbatch, int T, int D, int H, int hd, int ld_y,
                             hipStream_t s) {
    const float scale = 1.0f / sqrtf((float)hd);
    const int es = (int)sizeof(E);
    (void)hipGetLastError();
    if ((hd * es) % 8 == 0 && (D * es) % 8 == 0) {
        const int threads = 128;
        const size_t pair_bytes = (size_t)3 * T * hd * es;
        int ppb = threads / T;
        if (ppb < 1) ppb = 1;
        const size_t budget = 64 * 1024;
        if ((size_t)ppb * pair_bytes > budget) ppb = (int)(budget / pair_bytes);
        if (ppb >= 1) {
            const int n_pairs = vbatch * H;
            const int grid = (n_pairs + ppb - 1) / ppb;
            hipLaunchKernelGGL((attention_kernel<E, HDP>), dim3(grid), dim3(threads), (size_t)ppb * pair_bytes, s,
                               (const E*)qkv, (E*)y, n_pairs, T, D, H, hd, ld_y, scale, ppb);
            return hipGetLastError();
        }
    }
    const int total = vbatch * H * T;
    hipLaunchKernelGGL((attention_rowwise_kernel<E, HDP>), dim3((total + 127) / 128), dim3(128), 0, s,
                       (const E*)qkv, (E*)y, vbatch, T, D, H, hd, ld_y, scale);
    return hipGetLastError();
}

template <typename E>
static hipError_t launch_t(const void* qkv, void* y, int vbatch, int T, int D, int H, int ld_y, hipStream_t s) {
    const int hd = D / H;
    if (hd <= 32) return launch_hdp<E, 32>(qkv, y, vbatch, T, D, H, hd, ld_y, s);
    if (hd <= 64) return launch_hdp<E, 64>(qkv, y, vbatch, T, D, H, hd, ld_y, s);
    if (hd <= 128) return launch_hdp<E, 128>(qkv, y, vbatch, T, D, H, hd, ld_y, s);
    return hipErrorInvalidValue;
}

hipError_t launch_attention(const void* qkv, void* y, int vbatch, int T, int D, int H, int ld_y, int precision,
                            hipStream_t s) {
    const int hd = D / H;
    if (precision == BESO_PREC_BF16X3) {
        // fp32 rows in and out; split-bf16 products on the matrix pipe where the shape allows, the exact-fp32 kernel otherwise
        const int Tp = ((T + 15) / 16) * 16;
        const size_t pair_bytes = (size_t)3 * Tp * kAttRow * 4;
        if (hd <= 64 && hd % 8 == 0 && D % 8 == 0 && ld_y % 4 == 0 && T > 16 && T <= 16 * kAttMaxTiles && pair_bytes <= 160 * 1024) {
            if (hipFuncSetAttribute((const void*)attention_mfma_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024) != hipSuccess) return hipErrorInvalidValue;
            (void)hipGetLastError();
            const int n_pairs = vbatch * H;
            hipLaunchKernelGGL(attention_mfma_x3_kernel, dim3(n_pairs), dim3(256), pair_bytes, s, (const float*)qkv, (float*)y, n_pairs,
                               T, D, H, hd, ld_y, 1.4426950408889634f / sqrtf((float)hd));
            return hipGetLastError();
        }
        return launch_t<float>(qkv, y, vbatch, T, D, H, ld_y, s);
    }
    if (precision == BESO_PREC_FP32) return launch_t<float>(qkv, y, vbatch, T, D, H, ld_y, s);
    if (hd <= 64 && hd % 8 == 0 && D % 8 == 0 && T > 16 && T <= 16 * kAttMaxTiles) {
        const int Tp = ((T + 15) / 16) * 16;
        const size_t pair_bytes = (size_t)(2 * Tp * kAttRow + 64 * (Tp + 8)) * 2;
        int ppw = (int)((size_t)(160 * 1024) / pair_bytes);          // pairs (= waves) per workgroup
        ppw = ppw >= 4 ? 4 : (ppw >= 2 ? 2 : 1);
        const size_t lds = (size_t)ppw * pair_bytes;
        // (every call: the attribute belongs to the kernel on the CURRENT device, and this is not a hot launch site)
        if (hipFuncSetAttribute((const void*)attention_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) return hipErrorInvalidValue;
        (void)hipGetLastError();
        const int n_pairs = vbatch * H;
        hipLaunchKernelGGL(attention_mfma_kernel, dim3((n_pairs + ppw - 1) / ppw), dim3(64 * ppw), lds, s, (const uint16_t*)qkv,
                           (uint16_t*)y, n_pairs, T, D, H, hd, ld_y, 1.4426950408889634f / sqrtf((float)hd));
        return hipGetLastError();
    }
    return launch_t<uint16_t>(qkv, y, vbatch, T, D, H, ld_y, s);
}

}  // namespace beso
