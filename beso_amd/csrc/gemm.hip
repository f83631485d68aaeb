// Generic MFMA GEMM of the score network (K2/K4/K5/K6 of the unfused path):
//     C[M, N] = A[M, Kp] * Wt[Np, Kp]^T  (+ bias, + fused epilogue)
// A  : activations, row-major, K zero-padded to a multiple of 64      (tokens x features)
// Wt : torch Linear weight layout [out, in] = B^T, zero padded          (score_gpts.py:33-37,105-108)
// 128x128 block tile, 4 waves (2x2) of 64x64, 16x16 MFMA tiles, 128 bytes of K per LDS row per stage,
// XOR-swizzled 16-byte chunks, register-staged double buffering.
//   bf16: v_mfma_f32_16x16x32_bf16 (one per 16-byte chunk pair)
//   fp32: v_mfma_f32_16x16x4_f32   (four per chunk pair; exact fp32 -- the parity mode)
// Epilogues: bias -> store | bias + exact-erf GELU -> store | bias + residual add into fp32 x.
#include "common.h"

namespace beso {

template <typename E> struct Mma;
template <> struct Mma<uint16_t> {
    __device__ static __forceinline__ void mma(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                      acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // lane group g = lane>>4 holds k = 4g..4g+3 of a 16-wide k group; MFMA j consumes element j of
    // both operands, i.e. k-slot g of MFMA j is k = 4g + j for A and B alike.
    __device__ static __forceinline__ void mma(f32x4& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), acc, 0, 0, 0);
    }
};

__device__ __forceinline__ float gelu_erf(float v) {
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));   // nn.GELU() default (score_gpts.py:107)
}
template <typename E> __device__ __forceinline__ float gelu_for(float v);
template <> __device__ __forceinline__ float gelu_for<float>(float v) { return gelu_erf(v); }
template <> __device__ __forceinline__ float gelu_for<uint16_t>(float v) { return gelu_poly(v); }

template <typename E, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const E* __restrict__ A, int lda, const E* __restrict__ Wt,
                                                      int ldw, const float* __restrict__ bias, void* out_v, int ldo,
                                                      int n_store, int M, int Kp, int nt_n) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][kTileMN * kTileKBytes];   // 64 KiB
    constexpr int EPC = 16 / (int)sizeof(E);            // elements per 16-byte chunk
    constexpr int KSTAGE = kTileKBytes / (int)sizeof(E);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tile_n = blockIdx.x % nt_n, tile_m = blockIdx.x / nt_n;
    const int m0 = tile_m * kTileMN, n0 = tile_n * kTileMN;
    const int nk = Kp / KSTAGE;

    u32x4 ra[4], rb[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + 256 * i, row = c >> 3, kc = c & 7;
            int gm = m0 + row;
            gm = gm < M ? gm : M - 1;
            ra[i] = *(const u32x4*)(A + (size_t)gm * lda + (size_t)kt * KSTAGE + kc * EPC);
            rb[i] = *(const u32x4*)(Wt + (size_t)(n0 + row) * ldw + (size_t)kt * KSTAGE + kc * EPC);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + 256 * i, row = c >> 3, kc = c & 7;
            int off = row * kTileKBytes + ((kc ^ (row & 7)) << 4);
            *(u32x4*)(&lds[buf][0][off]) = ra[i];
            *(u32x4*)(&lds[buf][1][off]) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 af[4], bf[4];
            const int kc = s * 4 + (lane >> 4);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                int row = wm * 64 + mi * 16 + (lane & 15);
                af[mi] = *(const u32x4*)(&lds[cur][0][row * kTileKBytes + ((kc ^ (row & 7)) << 4)]);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                int row = wn * 64 + ni * 16 + (lane & 15);
                bf[ni] = *(const u32x4*)(&lds[cur][1][row * kTileKBytes + ((kc ^ (row & 7)) << 4)]);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) Mma<E>::mma(acc[mi][ni], af[mi], bf[ni]);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    // epilogue.  C/D layout of the 16x16 MFMA: col = lane & 15, row = 4*(lane >> 4) + reg.
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + (lane & 15);
        const float bn = bias[n];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + mi * 16 + (lane >> 4) * 4 + r;
                if (m < M && n < n_store) {
                    float v = acc[mi][ni][r] + bn;
                    if (EPI == EPI_BIAS_RESID) {
                        float* x = (float*)out_v + (size_t)m * ldo + n;
                        *x = *x + v;                                  // x + attn(..) / x + mlp(..)  (:113-114)
                    } else {
                        if (EPI == EPI_BIAS_GELU_STORE) v = gelu_for<E>(v);
                        ((E*)out_v)[(size_t)m * ldo + n] = Act<E>::from(v);
                    }
                }
            }
        }
    }
}

template <typename E>
static hipError_t launch_e(int epi, const void* A, int lda, const void* Wt, int ldw, const float* bias, void* out,
                           int ldo, int n_store, int M, int Np, int Kp, hipStream_t s) {
    (void)hipGetLastError();
    (void)hipGetLastError();   // clear any stale error left by other runtime users in this thread
    int nt_n = Np / kTileMN, nt_m = (M + kTileMN - 1) / kTileMN;
    dim3 grid(nt_n * nt_m), block(256);
    switch (epi) {
        case EPI_BIAS_STORE:
            hipLaunchKernelGGL((gemm_kernel<E, EPI_BIAS_STORE>), grid, block, 0, s, (const E*)A, lda, (const E*)Wt,
                               ldw, bias, out, ldo, n_store, M, Kp, nt_n);
            break;
        case EPI_BIAS_GELU_STORE:
            hipLaunchKernelGGL((gemm_kernel<E, EPI_BIAS_GELU_STORE>), grid, block, 0, s, (const E*)A, lda,
                               (const E*)Wt, ldw, bias, out, ldo, n_store, M, Kp, nt_n);
            break;
        case EPI_BIAS_RESID:
            hipLaunchKernelGGL((gemm_kernel<E, EPI_BIAS_RESID>), grid, block, 0, s, (const E*)A, lda, (const E*)Wt,
                               ldw, bias, out, ldo, n_store, M, Kp, nt_n);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_gemm(int precision, int epi, const void* A, int lda, const void* Wt, int ldw, const float* bias,
                       void* out, int ldo, int n_store, int M, int Np, int Kp, hipStream_t s) {
    (void)hipGetLastError();   // clear any stale error left by other runtime users in this thread
    if (precision == BESO_PREC_FP32) return launch_e<float>(epi, A, lda, Wt, ldw, bias, out, ldo, n_store, M, Np, Kp, s);
    return launch_e<uint16_t>(epi, A, lda, Wt, ldw, bias, out, ldo, n_store, M, Np, Kp, s);
}

}  // namespace beso
