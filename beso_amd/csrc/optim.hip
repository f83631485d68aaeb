// K11 + K12: one optimizer step over ALL parameter tensors in one launch, optionally followed by the EMA
// update on the freshly written parameter (beso_agent.py:236-244 -> torch.optim.AdamW / Adam,
// configs/agents/beso_kitchen.yaml:9-12, beso_block_push.yaml:9-11; ema.py:45-53).
//
// HBM bound: per element it reads p, g, m, v (+ema) and writes p, m, v (+ema) = 28 (36) bytes; the
// 9.4 M-parameter kitchen model is 0.34 GB per step, ~70 us at 5 TB/s, against one launch per tensor and
// per operation (several hundred launches) for the eager optimizer.  The arithmetic follows torch's
// single-tensor Adam(W) (amsgrad = False, maximize = False) operation by operation in fp32:
//     AdamW:  p *= 1 - lr*wd                      Adam:  g += wd*p
//     m = m + (g - m)*(1 - beta1)                 (Tensor.lerp_)
//     v = v*beta2 + (1 - beta2)*g*g               (mul_ + addcmul_)
//     denom = sqrt(v)/sqrt(1 - beta2^t) + eps;  p -= (lr/(1 - beta1^t)) * (m/denom)
//     ema -= (1 - decay)*(ema - p)                (the reference's EMA, on the updated p)
#include "common.h"

namespace beso {

struct OptimChunk {          // == beso_optim_chunk (include/beso_hip.h)
    float* p;
    const float* g;
    unsigned long long off;  // element offset of the chunk in the flat state buffers (m, v, ema)
    unsigned int n;          // elements in the chunk
    unsigned int pad;
};
// chunks hold at most 4096 elements (beso_amd/optim.py); any count works

__global__ __launch_bounds__(256) void adam_ema_kernel(const OptimChunk* __restrict__ chunks, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ ema, float lr,
                                                       float beta1, float beta2, float eps, float wd, int decoupled,
                                                       float step_size, float rsqrt_bc2_inv, float ema_decay) {
    const OptimChunk c = chunks[blockIdx.x];
    for (unsigned i = threadIdx.x; i < c.n; i += 256) {
        float p = c.p[i], g = c.g[i];
        float mi = m[c.off + i], vi = v[c.off + i];
        if (decoupled) p = p * (1.0f - lr * wd);
        else g = fmaf(wd, p, g);                       // grad.add(param, alpha=wd); wd = 0 leaves g untouched
        mi = mi + (g - mi) * (1.0f - beta1);
        vi = vi * beta2;
        vi = fmaf((1.0f - beta2) * g, g, vi);          // addcmul_(g, g, value = 1 - beta2)
        const float denom = sqrtf(vi) / rsqrt_bc2_inv + eps;
        p = p - step_size * (mi / denom);
        c.p[i] = p;
        m[c.off + i] = mi;
        v[c.off + i] = vi;
        if (ema) {
            const float s = ema[c.off + i];
            ema[c.off + i] = s - (1.0f - ema_decay) * (s - p);
        }
    }
}

hipError_t launch_adam_ema(const void* chunks, int n_chunks, float* m, float* v, float* ema, float lr, float beta1,
                           float beta2, float eps, float wd, int decoupled, int step, float ema_decay, hipStream_t s) {
    (void)hipGetLastError();
    // bias corrections in double on the host, as Python floats in torch
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), sqrt_bc2 = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_ema_kernel, dim3(n_chunks), dim3(256), 0, s, (const OptimChunk*)chunks, m, v, ema, lr, beta1,
                       beta2, eps, wd, decoupled, step_size, sqrt_bc2, ema_decay);
    return hipGetLastError();
}

}  // namespace beso
