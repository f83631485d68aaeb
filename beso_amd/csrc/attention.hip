// K3 (generic path): causal self-attention over the T <= 1+G+2W tokens of one sample
// (score_gpts.py:69-76): att = softmax(mask(q k^T / sqrt(hd))), y = att v, heads re-merged.
// The causal mask covers the WHOLE sequence including the sigma and goal tokens (:42-47,70).
// Work is tiny (0.5 % of the network's FLOPs): one thread per (sample, head, query row), online
// softmax, q and the output row in registers.  qkv row layout: [q(D) | k(D) | v(D)].
#include "common.h"

namespace beso {

template <typename E, int HDP>
__global__ void attention_kernel(const E* __restrict__ qkv, E* __restrict__ y, int vbatch, int T, int D, int H,
                                 int hd, int ld_y, float scale) {
    int item = blockIdx.x * blockDim.x + threadIdx.x;     // ((vb*H + h)*T + i)
    int total = vbatch * H * T;
    if (item >= total) return;
    int i = item % T;
    int h = (item / T) % H;
    int vb = item / (T * H);
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
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HDP; ++d) if (d < hd) s = fmaf(q[d], Act<E>::to(kr[d]), s);
        s *= scale;                           // (q k^T) * 1/sqrt(hd), as score_gpts.py:69
        float mn = fmaxf(m, s);
        float alpha = expf(m - mn);           // exp(-inf) = 0 on the first key
        float p = expf(s - mn);
        l = l * alpha + p;
#pragma unroll
        for (int d = 0; d < HDP; ++d) if (d < hd) o[d] = fmaf(p, Act<E>::to(vr[d]), o[d] * alpha);
        m = mn;
    }
    float inv = 1.0f / l;
    E* yr = y + ((size_t)vb * T + i) * ld_y + (size_t)h * hd;
#pragma unroll
    for (int d = 0; d < HDP; ++d) if (d < hd) yr[d] = Act<E>::from(o[d] * inv);
    // zero the K padding of the row once (head 0's thread)
    if (h == 0) {
        E* pad = y + ((size_t)vb * T + i) * ld_y;
        for (int c = D; c < ld_y; ++c) pad[c] = Act<E>::from(0.f);
    }
}

template <typename E>
static hipError_t launch_t(const void* qkv, void* y, int vbatch, int T, int D, int H, int ld_y, hipStream_t s) {
    int hd = D / H;
    int total = vbatch * H * T;
    int grid = (total + 127) / 128;
    float scale = 1.0f / sqrtf((float)hd);
    if (hd <= 32)
        hipLaunchKernelGGL((attention_kernel<E, 32>), dim3(grid), dim3(128), 0, s, (const E*)qkv, (E*)y, vbatch, T,
                           D, H, hd, ld_y, scale);
    else if (hd <= 64)
        hipLaunchKernelGGL((attention_kernel<E, 64>), dim3(grid), dim3(128), 0, s, (const E*)qkv, (E*)y, vbatch, T,
                           D, H, hd, ld_y, scale);
    else if (hd <= 128)
        hipLaunchKernelGGL((attention_kernel<E, 128>), dim3(grid), dim3(128), 0, s, (const E*)qkv, (E*)y, vbatch,
                           T, D, H, hd, ld_y, scale);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_attention(const void* qkv, void* y, int vbatch, int T, int D, int H, int ld_y, int precision,
                            hipStream_t s) {
    if (precision == BESO_PREC_FP32) return launch_t<float>(qkv, y, vbatch, T, D, H, ld_y, s);
    return launch_t<uint16_t>(qkv, y, vbatch, T, D, H, ld_y, s);
}

}  // namespace beso
