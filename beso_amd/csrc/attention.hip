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
    if (precision == BESO_PREC_FP32) return launch_t<float>(qkv, y, vbatch, T, D, H, ld_y, s);
    return launch_t<uint16_t>(qkv, y, vbatch, T, D, H, ld_y, s);
}

}  // namespace beso
