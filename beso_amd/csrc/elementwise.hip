// Memory-bound kernels of the score network: weight packing, token embedding (K1), LayerNorm,
// final-LN + action head + un-precondition (K7) and the sampler update (K8).
// Reference semantics: score_gpts.py:272-358, score_wrappers.py:31-96, gc_sampling.py.
#include <algorithm>
#include "common.h"

namespace beso {

// -----------------------------------------------------------------------------------------------
// K0: dst[rows_p][cols_p] <- src[rows][cols], converted to the GEMM operand type, zero padded.
// -----------------------------------------------------------------------------------------------
template <typename E>
__global__ void pack_matrix_kernel(const float* __restrict__ src, int rows, int cols, E* __restrict__ dst,
                                   int rows_p, int cols_p) {
    size_t n = (size_t)rows_p * cols_p;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int r = (int)(i / cols_p), c = (int)(i % cols_p);
        float v = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
        dst[i] = Act<E>::from(v);
    }
}

hipError_t launch_pack_matrix(const float* src, int rows, int cols, void* dst, int rows_p, int cols_p,
                              int precision, hipStream_t s) {
    (void)hipGetLastError();   // clear any stale error left by other runtime users in this thread
    size_t n = (size_t)rows_p * cols_p;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    if (precision == -1 || precision == BESO_PREC_FP32)   // -1: always-fp32 sections (biases, LN, embeddings)
        hipLaunchKernelGGL(pack_matrix_kernel<float>, dim3(grid), dim3(256), 0, s, src, rows, cols, (float*)dst,
                           rows_p, cols_p);
    else
        hipLaunchKernelGGL(pack_matrix_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, src, rows, cols,
                           (uint16_t*)dst, rows_p, cols_p);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// K1: precondition + embed + assemble  -> x[vbatch][T][D]  (fp32 residual stream)
//   token 0            : sigma_emb(log(sigma)/4)                         score_gpts.py:284-286
//   tokens 1..G        : tok_emb(goal_g) + pos[g]     (zeros in if uncond) :301-306,322
//   token G+1+2i       : tok_emb(state_i) + pos[G+i]                      :305,323
//   token G+2+2i       : action_emb(action_i * c_in) + pos[G+i]           :307,325; score_wrappers.py:96
// One block per token row, one thread per output feature.
// -----------------------------------------------------------------------------------------------
constexpr int kEmbFastLen = 32;      // inputs of a token the one-pass form of embed_kernel keeps a weight row in registers for
__global__ void embed_kernel(const float* __restrict__ state, const float* __restrict__ action,
                             const float* __restrict__ goal, const float* __restrict__ sigma,
                             const float* __restrict__ pos, const float* __restrict__ tok_w,
                             const float* __restrict__ tok_b, const float* __restrict__ sig_w,
                             const float* __restrict__ sig_b, const float* __restrict__ act_w,
                             const float* __restrict__ act_b, float* __restrict__ x, int B, int t, int T, int G,
                             int D, int obs, int act, int precondition, int uncond_from, float sigma_data) {
    extern __shared__ float in_vec[];   // the input vector of this token
    int row = blockIdx.x;               // vb*T + j
    int vb = row / T, j = row % T;
    int b = vb % B;
    float sg = sigma[b];
    int kind, len = 0, posrow = -1;     // kind 0 sigma, 1 goal/state (tok_emb), 2 action
    const float* src = nullptr;
    float scale = 1.f;
    if (j == 0) {
        kind = 0;
    } else if (j <= G) {
        kind = 1; len = obs; posrow = j - 1;
        src = (vb >= uncond_from) ? nullptr : goal + ((size_t)b * G + (j - 1)) * obs;
    } else {
        int idx = j - G - 1, i = idx >> 1;
        posrow = G + i;
        if ((idx & 1) == 0) { kind = 1; len = obs; src = state + ((size_t)b * t + i) * obs; }
        else {
            kind = 2; len = act; src = action + ((size_t)b * t + i) * act;
            if (precondition) scale = 1.0f / sqrtf(sg * sg + sigma_data * sigma_data);   // c_in
        }
    }
    // One feature per thread and a short input (the rollout's few token rows, launched with >= D threads): the thread's weight
    // row, bias and position entry are requested BEFORE the input is staged -- they do not depend on it -- and all at once (the
    // loop below waits for every 8 weights before it asks for the next: nine L2 round trips in a row at 30 inputs, most of this
    // launch's 6-9 us); the same left-to-right fma chain (bit-identical results).
    const bool fast = kind != 0 && len <= kEmbFastLen && D <= (int)blockDim.x;
    float wv[kEmbFastLen], bb = 0.f, pp = 0.f;
    if (fast && (int)threadIdx.x < D) {
        const int d = threadIdx.x;
        const float* w = (kind == 1 ? tok_w : act_w) + (size_t)d * len;
#pragma unroll
        for (int c = 0; c < kEmbFastLen; ++c) wv[c] = c < len ? w[c] : 0.f;
        bb = kind == 1 ? tok_b[d] : act_b[d];
        pp = pos[(size_t)posrow * D + d];
    }
    for (int c = threadIdx.x; c < (fast ? kEmbFastLen : len); c += blockDim.x) in_vec[c] = (src && c < len) ? src[c] * scale : 0.f;
    __syncthreads();
    if (fast) {
        if ((int)threadIdx.x < D) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < kEmbFastLen; ++c) acc = fmaf(in_vec[c], wv[c], acc);     // (past len: + 0 * 0)
            x[(size_t)row * D + threadIdx.x] = acc + bb + pp;
        }
        return;
    }
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float v;
        if (kind == 0) {
            v = sig_w[d] * (logf(sg) / 4.0f) + sig_b[d];
        } else {
            const float* w = (kind == 1 ? tok_w : act_w) + (size_t)d * len;
            float acc = 0.f;
            for (int c = 0; c < len; ++c) acc = fmaf(in_vec[c], w[c], acc);
            v = acc + (kind == 1 ? tok_b[d] : act_b[d]) + pos[(size_t)posrow * D + d];
        }
        x[(size_t)row * D + d] = v;
    }
}

// The same embedding with one block per (virtual) SAMPLE: a thread owns output feature d, keeps its rows of tok_emb /
// action_emb in registers and walks the sample's T tokens, whose inputs are staged in LDS once -- the weight rows are
// read once per sample instead of once per token (17,152 blocks re-reading 60 KB of weights took 113 us of the 1.45 ms
// long-horizon forward; this takes ~15).  obs <= 32, act <= 16 (the per-token kernel above serves anything else).
constexpr int kEmbObsMax = 32, kEmbActMax = 16;
__global__ void embed_sample_kernel(const float* __restrict__ state, const float* __restrict__ action,
                                    const float* __restrict__ goal, const float* __restrict__ sigma,
                                    const float* __restrict__ pos, const float* __restrict__ tok_w,
                                    const float* __restrict__ tok_b, const float* __restrict__ sig_w,
                                    const float* __restrict__ sig_b, const float* __restrict__ act_w,
                                    const float* __restrict__ act_b, float* __restrict__ x, int B, int t, int T, int G,
                                    int D, int obs, int act, int precondition, int uncond_from, float sigma_data) {
    extern __shared__ float in_all[];    // [G*obs goal | t*obs state | t*act action (pre-scaled)]
    const int vb = blockIdx.x, b = vb % B;
    const float sg = sigma[b];
    const float c_in = precondition ? 1.0f / sqrtf(sg * sg + sigma_data * sigma_data) : 1.f;
    float* gin = in_all;
    float* sin_ = in_all + G * obs;
    float* ain = sin_ + t * obs;
    const bool uncond = vb >= uncond_from;
    for (int i = threadIdx.x; i < G * obs; i += blockDim.x) gin[i] = uncond ? 0.f : goal[(size_t)b * G * obs + i];
    for (int i = threadIdx.x; i < t * obs; i += blockDim.x) sin_[i] = state[(size_t)b * t * obs + i];
    for (int i = threadIdx.x; i < t * act; i += blockDim.x) ain[i] = action[(size_t)b * t * act + i] * c_in;
    __syncthreads();
    const float lsg = logf(sg) / 4.0f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float wt[kEmbObsMax], wa[kEmbActMax];
#pragma unroll
        for (int c = 0; c < kEmbObsMax; ++c) wt[c] = c < obs ? tok_w[(size_t)d * obs + c] : 0.f;
#pragma unroll
        for (int c = 0; c < kEmbActMax; ++c) wa[c] = c < act ? act_w[(size_t)d * act + c] : 0.f;
        const float tb = tok_b[d], ab = act_b[d];
        float* xr = x + (size_t)vb * T * D + d;
        xr[0] = sig_w[d] * lsg + sig_b[d];
        for (int j = 1; j < T; ++j) {
            float acc = 0.f, add;
            if (j <= G || ((j - G - 1) & 1) == 0) {
                const float* in = j <= G ? gin + (j - 1) * obs : sin_ + ((j - G - 1) >> 1) * obs;
                // the same left-to-right fma chain as the per-token kernel (bit-identical results)
#pragma unroll
                for (int c = 0; c < kEmbObsMax; ++c) if (c < obs) acc = fmaf(in[c], wt[c], acc);
                add = tb;
            } else {
                const float* in = ain + ((j - G - 1) >> 1) * act;
#pragma unroll
                for (int c = 0; c < kEmbActMax; ++c) if (c < act) acc = fmaf(in[c], wa[c], acc);
                add = ab;
            }
            const int posrow = j <= G ? j - 1 : G + ((j - G - 1) >> 1);
            xr[(size_t)j * D] = acc + add + pos[(size_t)posrow * D + d];
        }
    }
}

hipError_t launch_embed(const Layout& lay, const char* packed, const FwdArgs& a, float* x, hipStream_t s, bool per_row) {
    (void)hipGetLastError();   // clear any stale error left by other runtime users in this thread
    // (per_row: a handful of samples -- one block per token row spreads them over more CUs than one block per sample, whose
    //  thread walks all T tokens of its feature one after the other: 17 us at B = 1 against 6; bit-identical results)
    if (lay.obs <= kEmbObsMax && lay.act <= kEmbActMax && a.T >= 8 && !per_row) {
        auto P = [&](size_t off) { return (const float*)(packed + off); };
        const int threads = lay.D >= 512 ? 512 : round_up(lay.D, 64);
        const size_t shmem = sizeof(float) * ((size_t)lay.G * lay.obs + (size_t)a.t * (lay.obs + lay.act));
        hipLaunchKernelGGL(embed_sample_kernel, dim3(a.vbatch), dim3(threads), shmem, s, a.state, a.action, a.goal, a.sigma,
                           P(lay.pos_emb), P(lay.tok_w), P(lay.tok_b), P(lay.sig_w), P(lay.sig_b), P(lay.act_w),
                           P(lay.act_b), x, a.batch, a.t, a.T, lay.G, lay.D, lay.obs, lay.act, a.precondition,
                           a.uncond_from, a.sigma_data);
        return hipGetLastError();
    }
    int rows = a.vbatch * a.T;
    // (per_row: a thread per feature where a block can hold them -- the kernel's one-pass form)
    int threads = per_row && lay.D <= 512 ? round_up(lay.D, 64) : lay.D >= 256 ? 256 : round_up(lay.D, 64);
    size_t shmem = sizeof(float) * (size_t)std::max(kEmbFastLen, lay.obs > lay.act ? lay.obs : lay.act);
    auto P = [&](size_t off) { return (const float*)(packed + off); };
    hipLaunchKernelGGL(embed_kernel, dim3(rows), dim3(threads), shmem, s, a.state, a.action, a.goal, a.sigma,
                       P(lay.pos_emb), P(lay.tok_w), P(lay.tok_b), P(lay.sig_w), P(lay.sig_b), P(lay.act_w),
                       P(lay.act_b), x, a.batch, a.t, a.T, lay.G, lay.D, lay.obs, lay.act, a.precondition,
                       a.uncond_from, a.sigma_data);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// LayerNorm (eps 1e-5, biased variance, affine): fp32 row -> GEMM operand row (zero padded to ld_out).
// One wave per row; the row lives in registers (D <= 64*kMaxPerLane).
// -----------------------------------------------------------------------------------------------
constexpr int kLnMaxPerLane = 16;

template <typename E>
__global__ void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                 const float* __restrict__ b, E* __restrict__ out, int rows, int D, int ld_out) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    if (wave >= rows) return;
    const float* xr = x + (size_t)wave * D;
    float v[kLnMaxPerLane];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        int c = lane + i * 64;
        v[i] = (c < D) ? xr[c] : 0.f;
        sum += v[i];
    }
    float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        int c = lane + i * 64;
        float d = (c < D) ? v[i] - mean : 0.f;
        sq = fmaf(d, d, sq);
    }
    float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + 1e-5f);
    E* o = out + (size_t)wave * ld_out;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        int c = lane + i * 64;
        if (c < D) o[c] = Act<E>::from((v[i] - mean) * rstd * w[c] + b[c]);
        else if (c < ld_out) o[c] = Act<E>::from(0.f);
    }
}

hipError_t launch_layernorm(const float* x, const float* w, const float* b, void* out, int rows, int D,
                            int ld_out, int precision, hipStream_t s) {
    (void)hipGetLastError();   // clear any stale error left by other runtime users in this thread
    int waves_per_block = 4;
    int grid = (rows + waves_per_block - 1) / waves_per_block;
    if (precision == BESO_PREC_FP32)
        hipLaunchKernelGGL(layernorm_kernel<float>, dim3(grid), dim3(256), 0, s, x, w, b, (float*)out, rows, D,
                           ld_out);
    else
        hipLaunchKernelGGL(layernorm_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, x, w, b, (uint16_t*)out,
                           rows, D, ld_out);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// K7: ln_f on the action-token rows only (score_gpts.py:341-353), action_pred (:354), then
// D_theta = F*c_out + action*c_skip (score_wrappers.py:96) and the classifier-free combination
// out_u + lambda*(out_c - out_u) (classifier_free_sampler.py:49).  One wave per (sample, step).
// -----------------------------------------------------------------------------------------------
// The same for the common small case -- Linear(D, act) head, D <= 384, act <= 12 (every shipped config) -- with EVERYTHING the
// row needs requested at once: the row, the LayerNorm parameters and all act weight rows (they do not depend on the LayerNorm:
// loading them behind it made the head four dependent memory round trips long, 11 of the small-batch forward's 172 us).  Same
// operations in the same order per output as head_row: bit-identical results.
__device__ __forceinline__ float head_row_small(const float* __restrict__ xr, const float* __restrict__ lnw,
                                               const float* __restrict__ lnb, const float* __restrict__ w0,
                                               const float* __restrict__ b0, int D, int act, int lane) {
    constexpr int NP = 6, NA = 12;
    float v[NP], gw[NP], gb[NP], wr[NA][NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int c = lane + i * 64;
        const bool ok = c < D;
        v[i] = ok ? xr[c] : 0.f;
        gw[i] = ok ? lnw[c] : 0.f;
        gb[i] = ok ? lnb[c] : 0.f;
#pragma unroll
        for (int o = 0; o < NA; ++o) wr[o][i] = (ok && o < act) ? w0[(size_t)o * D + c] : 0.f;
    }
    const float bo = lane < act ? b0[lane] : 0.f;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) sum += v[i];
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const float d = (lane + i * 64 < D) ? v[i] - mean : 0.f;
        sq = fmaf(d, d, sq);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < NP; ++i) v[i] = (lane + i * 64 < D) ? (v[i] - mean) * rstd * gw[i] + gb[i] : 0.f;
    float acc[NA];
#pragma unroll
    for (int o = 0; o < NA; ++o) {
        acc[o] = 0.f;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (lane + i * 64 < D) acc[o] = fmaf(v[i], wr[o][i], acc[o]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int o = 0; o < NA; ++o) acc[o] += __shfl_xor(acc[o], off, 64);
    }
    float mine = 0.f;
#pragma unroll
    for (int o = 0; o < NA; ++o)
        if (lane == o) mine = acc[o] + bo;
    return lane < act ? mine : 0.f;
}

__device__ __forceinline__ float head_row(const float* __restrict__ xr, const float* __restrict__ lnw,
                                         const float* __restrict__ lnb, const float* __restrict__ w0,
                                         const float* __restrict__ b0, const float* __restrict__ w1,
                                         const float* __restrict__ b1, int D, int act, int linear_output,
                                         int lane, float* hid /* LDS scratch [kHeadHidden] per wave */) {
    // returns pred[lane] (lanes >= act return 0)
    float mine = 0.f;
    float v[kLnMaxPerLane];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        int c = lane + i * 64;
        v[i] = (c < D) ? xr[c] : 0.f;
        sum += v[i];
    }
    float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        int c = lane + i * 64;
        float d = (c < D) ? v[i] - mean : 0.f;
        sq = fmaf(d, d, sq);
    }
    float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        int c = lane + i * 64;
        v[i] = (c < D) ? (v[i] - mean) * rstd * lnw[c] + lnb[c] : 0.f;
    }
    if (linear_output) {
        // four outputs at a time: their dot products and wave reductions are independent chains that overlap (one output
        // after the other was a chain of act x 6 dependent cross-lane steps: 3 us of the small-batch forward's head); the
        // same summation order per output as before
        for (int o0 = 0; o0 < act; o0 += 4) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < kLnMaxPerLane; ++i) {
                int c = lane + i * 64;
                if (c < D) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (o0 + u < act) acc[u] = fmaf(v[i], w0[(size_t)(o0 + u) * D + c], acc[u]);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] += __shfl_xor(acc[u], off, 64);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (o0 + u < act && lane == o0 + u) mine = acc[u] + b0[o0 + u];
        }
    } else {
        for (int o = 0; o < kHeadHidden; ++o) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < kLnMaxPerLane; ++i) {
                int c = lane + i * 64;
                if (c < D) acc = fmaf(v[i], w0[(size_t)o * D + c], acc);
            }
            float z = wave_sum(acc) + b0[o];
            if (lane == 0) hid[o] = z / (1.0f + expf(-z));   // SiLU
        }
        __builtin_amdgcn_wave_barrier();
        for (int o = 0; o < act; ++o) {
            float acc = 0.f;
            for (int c = lane; c < kHeadHidden; c += 64) acc = fmaf(hid[c], w1[(size_t)o * kHeadHidden + c], acc);
            float r = wave_sum(acc) + b1[o];
            if (lane == o) mine = r;
        }
        __builtin_amdgcn_wave_barrier();
    }
    return mine;
}

__global__ void head_kernel(const float* __restrict__ x, const float* __restrict__ action,
                            const float* __restrict__ sigma, const float* __restrict__ lnw,
                            const float* __restrict__ lnb, const float* __restrict__ w0,
                            const float* __restrict__ b0, const float* __restrict__ w1,
                            const float* __restrict__ b1, float* __restrict__ out, int B, int vbatch, int t, int T,
                            int G, int D, int act, int linear_output, int precondition, float cond_lambda,
                            float sigma_data) {
    __shared__ float hid_all[4][kHeadHidden];
    int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int item = blockIdx.x * 4 + wid;           // b*t + i
    if (item >= B * t) return;
    int b = item / t, i = item % t;
    int j = G + 2 + 2 * i;                     // action token of step i
    const bool small = linear_output && D <= 384 && act <= 12;          // (wave-uniform: a kernel argument)
    bool two = vbatch > B;
    float fc, fu = 0.f;
    if (small) {
        fc = head_row_small(x + ((size_t)b * T + j) * D, lnw, lnb, w0, b0, D, act, lane);
        if (two) fu = head_row_small(x + ((size_t)(b + B) * T + j) * D, lnw, lnb, w0, b0, D, act, lane);
    } else {
        fc = head_row(x + ((size_t)b * T + j) * D, lnw, lnb, w0, b0, w1, b1, D, act, linear_output, lane, hid_all[wid]);
        if (two)
            fu = head_row(x + ((size_t)(b + B) * T + j) * D, lnw, lnb, w0, b0, w1, b1, D, act, linear_output, lane,
                          hid_all[wid]);
    }
    if (lane < act) {
        float sg = sigma[b];
        float a = action[((size_t)b * t + i) * act + lane];
        float c_skip = 0.f, c_out = 1.f;
        if (precondition) {
            float sd2 = sigma_data * sigma_data;
            c_skip = sd2 / (sg * sg + sd2);
            c_out = sg * sigma_data / sqrtf(sg * sg + sd2);
        }
        float oc = fc * c_out + a * c_skip;
        float r = oc;
        if (two) {
            float ou = fu * c_out + a * c_skip;
            r = ou + cond_lambda * (oc - ou);
        }
        out[((size_t)b * t + i) * act + lane] = r;
    }
}

hipError_t launch_head(const Layout& lay, const char* packed, const FwdArgs& a, const float* x, hipStream_t s) {
    (void)hipGetLastError();   // clear any stale error left by other runtime users in this thread
    auto P = [&](size_t off) { return (const float*)(packed + off); };
    int items = a.batch * a.t;
    hipLaunchKernelGGL(head_kernel, dim3((items + 3) / 4), dim3(256), 0, s, x, a.action, a.sigma, P(lay.lnf_w),
                       P(lay.lnf_b), P(lay.head_w0), P(lay.head_b0), P(lay.head_w1), P(lay.head_b1), a.out,
                       a.batch, a.vbatch, a.t, a.T, lay.G, lay.D, lay.act, lay.linear_output,
                       a.precondition, a.cond_lambda, a.sigma_data);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------
// K8: sampler update, in the reference's operation order (gc_sampling.py:205-210,296-310,921-923)
// -----------------------------------------------------------------------------------------------
// sig_next != nullptr: the launch also writes the sigma vector of the NEXT network evaluation (sig_next[0 .. n_sig) = sigma_next;
// it would be a launch of its own between two dependent ones otherwise).
__global__ void sampler_step_kernel(int mode, float* __restrict__ out, float* __restrict__ aux,
                                    const float* __restrict__ x, const float* __restrict__ x2,
                                    const float* __restrict__ den, float c0, float c1, size_t n, float* __restrict__ sig_next,
                                    float sigma_next, int n_sig) {
    if (sig_next)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_sig; i += gridDim.x * blockDim.x) sig_next[i] = sigma_next;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float xv = x[i], dv = den[i];
        float r;
        if (mode == BESO_STEP_ADD_NOISE) {
            // action + randn * sigma_up (gc_sampling.py:246-247): a product and a sum, each rounded, as torch evaluates it (not
            // an fma) -- and as the head of the one-launch loop does, with which this form agrees bit for bit
#pragma clang fp contract(off)
            const float nz = x2[i] * c0;
            r = xv + nz;
        } else {
            float a = (mode == BESO_STEP_HEUN_CORRECT) ? aux[i] : 0.f;
            r = sampler_update(mode, xv, mode == BESO_STEP_HEUN_CORRECT ? x2[i] : 0.f, dv, a, c0, c1);
            if (mode == BESO_STEP_HEUN_PREDICT) aux[i] = a;
        }
        out[i] = r;
    }
}

hipError_t launch_sampler_step(int mode, float* out, float* aux, const float* x, const float* x2, const float* den,
                               float c0, float c1, size_t n, hipStream_t s, float* sig_next, float sigma_next, int n_sig) {
    (void)hipGetLastError();   // clear any stale error left by other runtime users in this thread
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(sampler_step_kernel, dim3(grid), dim3(256), 0, s, mode, out, aux, x, x2, den, c0, c1, n, sig_next, sigma_next,
                       n_sig);
    return hipGetLastError();
}

// rand_log_logistic (k_diffusion/utils.py:178-185): sigma = exp(logit(u (hi - lo) + lo) scale + loc) in float64, out fp32 -- the
// chain of seven elementwise launches behind torch.rand as one (the training step draws its sigmas this way every step).
__global__ void log_logistic_kernel(const double* __restrict__ u, float* __restrict__ out, size_t n, double loc, double scale,
                                    double lo, double hi) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)          // (torch evaluates every step as its own rounded operation)
        const double p = u[i] * (hi - lo);
        const double q = p + lo;
        const double lg = log(q / (1.0 - q));                // Tensor.logit(): log(x / (1 - x))
        const double m = lg * scale;
        const double a = m + loc;
        out[i] = (float)exp(a);
    }
}

// Scaler.scale_input / scale_output (networks/scaler/scaler_class.py:95-117): out = (x - mean) / den per feature, for up to
// kScaleMax tensors in ONE launch -- a training step scales state, goal and action: six elementwise launches of ~6 us each in
// front of the forward (round 6: the 125 us of small launches between the optimizer and the forward are 6 % of the 1024-sample
// step).  The reference's two rounded operations (a subtraction, a correctly rounded division): same bits.
struct ScaleTable { const float* src[kScaleMax]; float* dst[kScaleMax]; const float* mean[kScaleMax]; const float* den[kScaleMax];
                    unsigned long long first[kScaleMax + 1]; int cols[kScaleMax]; int n; };
__global__ void scale_rows_kernel(ScaleTable t) {
    const unsigned long long total = t.first[t.n];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        int k = 0;
#pragma unroll
        for (int j = 1; j < kScaleMax; ++j) k += (j < t.n && i >= t.first[j]) ? 1 : 0;
        const unsigned long long e = i - t.first[k];
        const int c = (int)(e % (unsigned long long)t.cols[k]);
        const float d = t.src[k][e] - t.mean[k][c];        // (a subtraction and a division: nothing to contract)
        t.dst[k][e] = d / t.den[k][c];
    }
}

hipError_t launch_scale_rows(const float* const* src, float* const* dst, const float* const* mean, const float* const* den,
                             const long long* rows, const int* cols, int n, hipStream_t s) {
    ScaleTable t;
    t.n = n;
    unsigned long long cur = 0;
    for (int k = 0; k < kScaleMax; ++k) {
        const bool on = k < n;
        t.src[k] = on ? src[k] : nullptr; t.dst[k] = on ? dst[k] : nullptr; t.mean[k] = on ? mean[k] : nullptr;
        t.den[k] = on ? den[k] : nullptr; t.cols[k] = on ? cols[k] : 1;
        t.first[k] = cur;
        if (on) cur += (unsigned long long)rows[k] * (unsigned long long)cols[k];
    }
    for (int k = n; k <= kScaleMax; ++k) t.first[k] = cur;
    t.first[n] = cur;
    (void)hipGetLastError();
    if (cur == 0) return hipSuccess;
    unsigned long long grid = (cur + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)grid), dim3(256), 0, s, t);
    return hipGetLastError();
}

hipError_t launch_log_logistic(const double* u, float* out, size_t n, double loc, double scale, double lo, double hi, hipStream_t s) {
    (void)hipGetLastError();
    int grid = (int)((n + 255) / 256);
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(log_logistic_kernel, dim3(grid), dim3(256), 0, s, u, out, n, loc, scale, lo, hi);
    return hipGetLastError();
}

}  // namespace beso
