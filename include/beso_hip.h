/*
 * beso_hip.h -- C ABI of libbeso_hip.so: the MI355X (gfx950) implementation of BESO's
 * score-denoising hot path.
 *
 * The reference (intuitive-robots/beso) is pure Python/PyTorch and has no FFI of its own; the
 * boundary below is what a binding for this path would bind.  Each entry point names the
 * reference interface it replaces (file:line relative to the reference checkout):
 *
 *   beso_pack_weights   <- GCDenoiser.state_dict() / load_state_dict   beso_agent.py:458-476
 *   beso_score_fwd      <- DiffusionGPT.forward                        k_diffusion/score_gpts.py:272-358
 *   beso_denoise_fwd    <- GCDenoiser.forward                          k_diffusion/score_wrappers.py:81-96
 *                          (+ ClassifierFreeSampleModel.forward        k_diffusion/classifier_free_sampler.py:35-49)
 *   beso_sampler_step   <- the per-step update of sample_ddim/_euler/_heun
 *                                                                      k_diffusion/gc_sampling.py:205-210,296-310,921-923
 *   beso_sample         <- sample_ddim / sample_euler / sample_heun    k_diffusion/gc_sampling.py:167-213,259-314,895-924
 *   beso_sample_ancestral <- sample_euler_ancestral                    k_diffusion/gc_sampling.py:216-256
 *   beso_loss_grad      <- GCDenoiser.loss + loss.backward()           k_diffusion/score_wrappers.py:45-79, beso_agent.py:228-233
 *                          (+ DiffusionGPT.mask_cond, training mode     k_diffusion/score_gpts.py:298-299, 360-371)
 *   beso_goal_mask      <- the Bernoulli mask of DiffusionGPT.mask_cond k_diffusion/score_gpts.py:365-368
 *   beso_log_logistic   <- rand_log_logistic (behind the uniform draw)   k_diffusion/utils.py:178-185 (beso_agent.py:227)
 *   beso_scale_rows     <- Scaler.scale_input / scale_output              networks/scaler/scaler_class.py:95-117 (base_agent.py:111-142)
 *   beso_loss_grad_overlap  (same, with the early gradient range for the overlapped all-reduce: SURVEY 8(e) C1)
 *   beso_loss_grad_streams  (same, plus a stream that is released as soon as the loss value is final)
 *   beso_adam_step      <- optimizer.step() + ema_helper.update()      beso_agent.py:236-244
 *   beso_gather_windows <- TrajectorySlicerDataset.__getitem__ x batch envs/dataloaders/trajectory_loader.py:160-197
 *
 * Conventions
 *   - plain C, plain pointers and sizes.  No torch types.  `stream` is a hipStream_t passed as void*.
 *   - every tensor is device memory, contiguous, fp32, owned by the caller.  The library never
 *     allocates or frees device memory and never synchronises the stream; work is enqueued on
 *     `stream` and the call returns.
 *   - return value: 0 = ok, negative = error (see beso_status_string).  Bad shapes and unsupported
 *     configurations are rejected before anything is enqueued.
 *   - thread-safety: the library keeps no mutable process-wide state; calls on distinct workspaces/streams are
 *     independent (a workspace must not be shared by concurrent calls).  What a call may vary -- which kernels run --
 *     travels in its own `flags` (BESO_PLAN_*); the launch-site timers (beso_profile_*) are per calling thread.
 *   - development aids (phase stamps, the GEMM layout probe) are not part of this library: they exist in the
 *     development build only (include/beso_hip_debug.h, libbeso_hip_dev.so).
 */
#ifndef BESO_HIP_H
#define BESO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: the entry points declared here are its whole dynamic symbol table */
#pragma GCC visibility push(default)

/* DiffusionGPT.__init__ kwargs that shape the computation (score_gpts.py:121-139) + GCDenoiser.sigma_data. */
typedef struct beso_config {
    int32_t obs_dim;        /* state_dim                                                      */
    int32_t act_dim;        /* action_dim                                                     */
    int32_t embed_dim;      /* D                                                              */
    int32_t n_layers;       /* L                                                              */
    int32_t n_heads;        /* H,  D % H == 0                                                 */
    int32_t goal_seq_len;   /* G actually used: 0 when goal_conditioned is False (:143-144)   */
    int32_t obs_seq_len;    /* W (window_size); block_size = G + 2W + 1 (:148)                */
    int32_t linear_output;  /* 1: action_pred = Linear(D, act); 0: Linear(D,100)-SiLU-Linear  */
    float   sigma_data;     /* GCDenoiser.sigma_data (score_wrappers.py:29)                   */
} beso_config;

/* arithmetic of the GEMMs (everything else -- residual stream, LayerNorm, softmax, GELU,
 * preconditioning, sampler update -- is fp32 in every mode) */
enum {
    BESO_PREC_BF16 = 0,   /* bf16 MFMA inputs, fp32 accumulate: throughput mode                 */
    BESO_PREC_FP32 = 1,   /* fp32-input MFMA (v_mfma_f32_16x16x4_f32), exact fp32: parity mode  */
    BESO_PREC_BF16X3 = 2, /* split-bf16 (hi*hi + hi*lo + lo*hi on the bf16 MFMA, fp32 accumulate): fp32-class
                             accuracy from the fused kernel: an instance of layers_kernel for kitchen and block-push,
                             split-bf16 block kernels for the long-horizon shape (D = 512); other shapes return
                             BESO_ERR_UNSUPPORTED; inference only */
    BESO_PREC_FP16 = 3    /* fp16 MFMA inputs (v_mfma_f32_16x16x32_f16: the bf16 rate, three more mantissa bits), fp32
                             accumulate: the one-launch kernel's shapes only (kitchen, block-push, long-horizon without
                             classifier-free pairs); operands must stay inside fp16's range (|v| < 65504: LayerNorm
                             outputs, GELU outputs, probabilities and N(0, 0.02)-scale weights do); inference only */
};

/* beso_loss_grad flags */
enum {
    BESO_TRAIN_LAST_ACTION_ONLY = 1, /* GCDenoiser.loss(pred_last_action_only=True): only the last step of every window
                                        is scored (score_wrappers.py:59-63,76-77; the caller zeroes the other steps' noise) */
    /* execution-plan hints (bf16): which kernels run the step, never what it computes -- all forms write the same kept
     * activations and evaluate the same dropout masks.  Default forward: ALL layers as one launch where the shape has that
     * kernel (kitchen, block-push; no dropout on the proj / MLP outputs); otherwise the per-op kernels below 16,000 token rows
     * and the tile kernel (a layer's out-projection .. the next layer's q/k/v as one launch) from there on.  Default backward:
     * the data gradients in the transposed formulation (GELU' / LayerNorm backward as GEMM epilogues, FC2 .. out-projection of a
     * layer as one launch), the attention backward on the matrix pipe, the weight gradients of >= 18 k token rows in row
     * windows. */
    BESO_TRAIN_PLAN_PER_OP = 2,      /* per-op kernels for every layer, forward and backward (the comparator of the above) */
    BESO_TRAIN_PLAN_TILES = 4        /* the tile kernel (one launch per layer) wherever the shape has it */
};

/* flags of the forward calls (beso_score_fwd, beso_denoise_fwd, beso_sample, beso_sample_ancestral) */
enum {
    BESO_FLAG_UNCOND = 1,     /* DiffusionGPT.forward(uncond=True): goals := 0 (score_gpts.py:301-302); forwards only */
    /* Execution-plan hints: WHICH kernels run, never what they compute.  They exist for parity tests (the per-op kernels are
     * the reference of the fused ones in the same arithmetic; the instances of the one-launch kernel agree bit for bit) and
     * for measurements; a hint the shape or precision cannot honour is ignored. */
    BESO_PLAN_PER_OP = 0x10,  /* per-op kernels only: LayerNorm, GEMMs, attention (bf16 / fp32; any shape) */
    BESO_PLAN_BLOCKS = 0x20,  /* at most the block kernels (LN2 + MLP block, tail block), not the one-launch kernel */
    BESO_PLAN_SMALL = 0x40,   /* the chip-wide small-batch path (bf16 / fp32, embed_dim <= 384: four short launches per layer --
                                 three up to 96 token rows in bf16 -- that spread every weight matrix over the CUs) at ANY batch
                                 size; without a hint the library
                                 takes it up to 448 token rows in bf16 (kitchen: 40 samples) and 4096 in fp32, where one
                                 workgroup per sample group would stream all the weights alone */
    BESO_PLAN_FUSED = 0x80,   /* the one-launch / block kernels at every batch size (never the small-batch path) */
    BESO_PLAN_SPW2 = 0x100,   /* samples per workgroup of the one-launch kernel: 2 (default up to 512 samples), */
    BESO_PLAN_SPW4 = 0x200,   /*   4 (up to 1024; the split-bf16 mode: above 512), */
    BESO_PLAN_SPW8 = 0x300,   /*   8 (larger batches) */
    BESO_PLAN_SPW_MASK = 0x300,
    BESO_PLAN_MASK = 0x3f0,
    BESO_SAMPLE_STEPWISE = 0x1000  /* beso_sample: enqueue evaluation by evaluation (see there) */
};

/* sampler ids for beso_sample / beso_sampler_step */
enum {
    BESO_SAMPLER_DDIM = 0,   /* gc_sampling.py:895-924 */
    BESO_SAMPLER_EULER = 1,  /* gc_sampling.py:167-213, s_churn = 0 */
    BESO_SAMPLER_HEUN = 2    /* gc_sampling.py:259-314, s_churn = 0 */
};

/* status codes */
enum {
    BESO_OK = 0,
    BESO_ERR_BAD_CONFIG = -1,      /* config fields out of range / D % H != 0                      */
    BESO_ERR_BAD_SHAPE = -2,       /* batch < 1, t < 1, t > obs_seq_len (score_gpts.py:282)         */
    BESO_ERR_BAD_ARG = -3,         /* null pointer, unknown precision / sampler / flag              */
    BESO_ERR_WORKSPACE = -4,       /* workspace or packed buffer too small                          */
    BESO_ERR_UNSUPPORTED = -5,     /* configuration this build has no kernel for                    */
    BESO_ERR_HIP = -6              /* a HIP runtime call failed (hipGetLastError has the detail)     */
};

const char* beso_version(void);
const char* beso_status_string(int status);
/* Detail of the last BESO_ERR_HIP on the calling thread (HIP error name and the failing call). */
const char* beso_last_error(void);

/* Number of parameter tensors, in the order of the reference module's named_parameters():
 * pos_emb, tok_emb.{weight,bias}, per block {ln1,ln2}.{weight,bias}, attn.{key,query,value,proj}.{weight,bias},
 * mlp.{0,2}.{weight,bias}; ln_f.{weight,bias}, sigma_emb.{weight,bias}, action_emb.{weight,bias},
 * action_pred[.0/.2].{weight,bias}.  Linear weights are torch layout [out, in], fp32.            */
int beso_num_params(const beso_config* cfg);

/* Size of the packed-weight image for `precision`, in bytes (0 on bad config). */
size_t beso_packed_bytes(const beso_config* cfg, int precision);

/* Re-lay the fp32 parameters (host array `params` of `n_params` DEVICE pointers, order above) into
 * the kernel-ready image `packed` (device, >= beso_packed_bytes): fused QKV rows, bf16 (or fp32)
 * GEMM operands zero-padded to the MFMA tile grid.  Call again whenever a parameter changes
 * (optimizer step, EMA swap, load_state_dict).                                                   */
int beso_pack_weights(const beso_config* cfg, const float* const* params, int n_params,
                      void* packed, size_t packed_bytes, int precision, void* stream);

/* Scratch needed by one forward over `batch` samples with `t` observations in the window
 * (T = 1 + G + 2t tokens each).  `cfg_guidance` != 0 doubles the token count (cond + uncond).    */
size_t beso_workspace_bytes(const beso_config* cfg, int batch, int t, int precision, int cfg_guidance);

/* DiffusionGPT.forward (eval mode): out[batch,t,act] = F(states[batch,t,obs], actions[batch,t,act],
 * goals[batch,G,obs], sigma[batch]).  No preconditioning.                                        */
int beso_score_fwd(const beso_config* cfg, const void* packed, int precision,
                   const float* state, const float* action, const float* goal, const float* sigma,
                   float* out, int batch, int t, int flags,
                   void* workspace, size_t workspace_bytes, void* stream);

/* GCDenoiser.forward: out = F(state, action*c_in, goal, sigma)*c_out + action*c_skip.
 * cond_lambda reproduces ClassifierFreeSampleModel: 1 -> conditional only, 0 -> unconditional only,
 * otherwise out_u + cond_lambda*(out_c - out_u) evaluated as ONE 2*batch pass.                    */
int beso_denoise_fwd(const beso_config* cfg, const void* packed, int precision,
                     const float* state, const float* action, const float* goal, const float* sigma,
                     float* out, int batch, int t, int flags, float cond_lambda,
                     void* workspace, size_t workspace_bytes, void* stream);

/* One sampler update, elementwise over n fp32 values, in the reference's operation order:
 *   BESO_STEP_DDIM         out = c0*x - c1*den              c0 = sigma_fn(t_next)/sigma_fn(t), c1 = expm1(-h)   gc_sampling.py:921-923
 *   BESO_STEP_EULER        d = (x - den)/c0; out = x + d*c1                 c0 = sigma_hat, c1 = dt              :205-210
 *   BESO_STEP_HEUN_PREDICT d = (x - den)/c0; aux = d; out = x + d*c1        (out = action_2)                     :296-305
 *   BESO_STEP_HEUN_CORRECT d2 = (x2 - den)/c0; out = x + ((aux + d2)/2)*c1  c0 = sigma_{i+1}                     :306-310
 * out may alias x.  x2 / aux may be NULL for the modes that do not use them.                     */
enum { BESO_STEP_DDIM = 0, BESO_STEP_EULER = 1, BESO_STEP_HEUN_PREDICT = 2, BESO_STEP_HEUN_CORRECT = 3,
       BESO_STEP_ADD_NOISE = 4 /* out = x + x2 * c0 (x2 = the randn of an ancestral step, c0 = sigma_up; den unused, may be x)  :246-247 */ };
int beso_sampler_step(int mode, float* out, float* aux, const float* x, const float* x2, const float* den,
                      float c0, float c1, size_t n, void* stream);

/* A whole sampling loop: x[batch,t,act] holds x_T on entry and the sample on return.
 * `sigmas` is a HOST array of n_sigmas values, the last one 0 (get_sigmas_*: gc_sampling.py:26-44).
 * No host synchronisation inside.  Where the shape has the one-launch kernel (bf16 / bf16x3: kitchen, block-push,
 * long-horizon without classifier-free pairs) the WHOLE loop is ONE launch: the workgroup that owns a sample from the
 * embedding to the head also applies the step's update and feeds itself the next evaluation (up to 128 evaluations per
 * launch; longer loops are cut at step boundaries).  Otherwise, and with BESO_SAMPLE_STEPWISE, every evaluation is
 * enqueued as the forward launch(es) + one update launch.  Both forms run the same arithmetic (bit-identical results). */
int beso_sample(const beso_config* cfg, const void* packed, int precision, int sampler,
                const float* state, const float* goal, float* x, int batch, int t,
                const float* sigmas, int n_sigmas, float cond_lambda, int flags,
                void* workspace, size_t workspace_bytes, void* stream);

/* sample_euler_ancestral (gc_sampling.py:216-256, scaler = None) as one enqueue: per step an Euler step to sigma_down and,
 * while sigma_down > 0, x += noise_i * sigma_up (get_ancestral_step: :107-114, fp32).  `noise` is a DEVICE array of
 * n_sigmas - 1 standard-normal tensors [batch,t,act] back to back -- the reference's `torch.randn_like(action)` of each
 * step, drawn by the caller (the library has no random number generator); entries of steps with sigma_down = 0 are not
 * read.  Where the shape has the one-launch kernel the WHOLE loop is ONE launch, as in beso_sample: the workgroup that owns
 * a sample applies the Euler update and adds its slice of the step's noise in the kernel's head.  Otherwise, and with
 * BESO_SAMPLE_STEPWISE, one forward launch + one or two update launches per step (bit-identical results).
 * `flags`: BESO_PLAN_* hints | BESO_SAMPLE_STEPWISE.  Everything else as beso_sample.                                  */
int beso_sample_ancestral(const beso_config* cfg, const void* packed, int precision, const float* state, const float* goal,
                          float* x, int batch, int t, const float* sigmas, int n_sigmas, float cond_lambda, float eta,
                          const float* noise, int flags, void* workspace, size_t workspace_bytes, void* stream);

/* One Adam / AdamW step over ALL parameter tensors in one launch, optionally followed by the EMA update
 * of the shadow copy on the updated parameters.  Replaces `self.optimizer.step()` + `self.ema_helper.update`
 * of the training step (reference beso_agent.py:236-244; torch.optim.AdamW for kitchen, torch.optim.Adam
 * for block-push: configs/agents/beso_kitchen.yaml:9-12, beso_block_push.yaml:9-11; ema.py:45-53); the
 * arithmetic is torch's single-tensor Adam(W) with amsgrad = False, maximize = False, in fp32.
 *   chunks      DEVICE array: each entry is at most 4096 consecutive elements of one parameter tensor
 *   exp_avg, exp_avg_sq, ema   flat fp32 state buffers indexed by chunk.off + i (ema may be NULL)
 *   decoupled_wd 1 = AdamW (p *= 1 - lr*wd), 0 = Adam (g += wd*p);  step = 1-based step count
 *   ema_decay   the decay actually applied this step, min(decay, (1+n)/(10+n)) (ema.py:45-48)       */
typedef struct beso_optim_chunk {
    float* p;
    const float* g;
    unsigned long long off;
    unsigned int n;
    unsigned int pad;
} beso_optim_chunk;
int beso_adam_step(const beso_optim_chunk* chunks, int n_chunks, float* exp_avg, float* exp_avg_sq, float* ema,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled_wd, int step,
                   float ema_decay, void* stream);

/* Training step, forward + backward: GCDenoiser.loss (score_wrappers.py:45-79; flags: BESO_TRAIN_LAST_ACTION_ONLY) of the
 * training-mode network (score_gpts.py:272-358 with the dropouts of :41,:79,:109) and the gradient of that loss with
 * respect to every parameter -- what `loss = model.loss(...); loss.backward()` leaves in `.grad`
 * (beso_agent.py:228-233).  Both action heads (linear_output 1 / 0).
 *   params      host array of n_params DEVICE pointers, order of beso_pack_weights (fp32, torch layouts)
 *   grads_flat  device fp32 buffer of beso_grad_floats(cfg) values: the gradients of all parameters back to back in
 *               the same order, each tensor contiguous.  OVERWRITTEN: zeroed first, then the weight gradients are
 *               plain stores of one grouped launch (no split-K, no atomics), while the bias / LayerNorm-affine /
 *               embedding gradients are accumulated from block partial sums (a few fp32 atomics per block: two runs
 *               agree to rounding in those tensors, not bit for bit).
 *   state [batch,t,obs], action [batch,t,act] (clean), goal [batch,G,obs] (UNMASKED: see goal_drop),
 *   noise [batch,t,act], sigma [batch];  loss_out: one device float.
 *   goal_drop   DiffusionGPT's goal_drop (`cond_mask_prob`, configs: cond_mask_prob): training-mode mask_cond
 *               (score_gpts.py:298-299, 360-371) -- every ELEMENT of goal is zeroed with this probability, kept ones are
 *               not rescaled -- applied inside the embedding kernel; the mask is the counter-based hash of (seed, element),
 *               the one beso_goal_mask writes out.  0 disables (eval mode, or goals masked by the caller).
 *   embed_pdrop / attn_pdrop / resid_pdrop: dropout probabilities of the token embeddings (not the sigma token), of
 *   the attention weights and of the proj / MLP outputs (DiffusionGPT's embed_pdrob, attn_pdrop, resid_pdrop); the
 *   masks are a counter-based hash of (seed, site, element), recomputed in the backward.  0 disables.
 *   grad_scale multiplies every gradient (1/world_size for data-parallel averaging); the loss is unscaled.
 *   precision   BESO_PREC_BF16: bf16 GEMM operands (weights, kept activations, gradient operands), fp32 accumulation,
 *               fp32 residual stream / LayerNorm / softmax / loss;  BESO_PREC_FP32: everything fp32 (parity mode).  */
size_t beso_train_workspace_bytes(const beso_config* cfg, int batch, int t, int precision);
size_t beso_grad_floats(const beso_config* cfg);
int beso_loss_grad(const beso_config* cfg, const float* const* params, int n_params, float* grads_flat, int precision,
                   const float* state, const float* action, const float* goal, const float* noise, const float* sigma,
                   float* loss_out, int batch, int t, int flags, float embed_pdrop, float attn_pdrop, float resid_pdrop,
                   float goal_drop, unsigned int seed, float grad_scale, void* workspace, size_t workspace_bytes, void* stream);
/* The keep-mask (1.0 / 0.0 per element of goal [batch,G,obs]) that beso_loss_grad applies for (goal_drop, seed):
 * `1 - torch.bernoulli(...)` of DiffusionGPT.mask_cond (score_gpts.py:365-368) with this library's generator.       */
int beso_goal_mask(float* mask, int batch, int goal_seq_len, int obs_dim, float goal_drop, unsigned int seed, void* stream);
/* The training feed on trajectories resident in HBM: one batch of TrajectorySlicerDataset.__getitem__
 * (envs/dataloaders/trajectory_loader.py:160-197; the collate of torch's DataLoader included) as one launch.
 *   observations [n_traj,t_max,obs_dim], actions [n_traj,t_max,act_dim]  padded trajectories (TensorDataset.tensors)
 *   seq_len [n_traj] int32        valid length of each trajectory (get_seq_length)
 *   slice_traj / slice_start [n_slices] int32   the slicer's table: window s = rows [start, start + window) (:128-135)
 *   batch_slices [batch] int64    which windows make up this batch (a chunk of a permutation)
 *   draws [batch] int64 >= 0      one random integer per sample; BESO_GOAL_RANDOM takes the future sequence at
 *                                 lo + draws % (hi - lo), lo = end + min_future_sep, hi = seq_len - goal_len (:169-182);
 *                                 may be NULL for the other modes or goal_len = 0
 *   goal_mode   BESO_GOAL_RANDOM | BESO_GOAL_TAIL (only_sample_tail, :175-176) | BESO_GOAL_SEQ_END (only_sample_seq_end, :177-178)
 *   goal_len    future_seq_len, 0 = not future conditional (goal_out may be NULL)
 *   obs_out [batch,window,obs_dim], act_out [batch,window,act_dim], goal_out [batch,goal_len,obs_dim]
 * Samples whose trajectory has no room for a future sequence get the reference's zeros placeholder (:185-186);
 * out-of-range slice ids produce zero rows instead of a fault.                                                     */
enum { BESO_GOAL_RANDOM = 0, BESO_GOAL_TAIL = 1, BESO_GOAL_SEQ_END = 2 };
int beso_gather_windows(const float* observations, const float* actions, const int* seq_len, int n_traj, int t_max,
                        int obs_dim, int act_dim, const int* slice_traj, const int* slice_start, long long n_slices,
                        const long long* batch_slices, const long long* draws, int batch, int window, int goal_len,
                        int goal_mode, int min_future_sep, float* obs_out, float* act_out, float* goal_out, void* stream);

/* rand_log_logistic (k_diffusion/utils.py:178-185), the sigma density of the shipped training configs, behind the caller's
 * uniform draw: out[i] = (float) exp(logit(u[i] * (cdf_hi - cdf_lo) + cdf_lo) * scale + loc), every operation in float64 as the
 * reference evaluates it -- one launch instead of seven elementwise ones per training step.  u: n float64 values in [0, 1)
 * (torch.rand(..., dtype=float64): the library has no random number generator); cdf_lo / cdf_hi: the logistic CDF of
 * log(min_value) / log(max_value).                                                                                      */
int beso_log_logistic(const double* u, float* out, size_t n, double loc, double scale, double cdf_lo, double cdf_hi, void* stream);

/* Scaler.scale_input / scale_output (networks/scaler/scaler_class.py:95-117, called three times per batch by
 * BaseAgent.process_batch, base_agent.py:111-142): dst_k[r][c] = (src_k[r][c] - mean_k[c]) / den_k[c] for n <= 4 tensors of
 * rows_k x cols_k fp32 values in ONE launch (den = std + 1e-12, the reference's denominator; a subtraction and a correctly
 * rounded division per element: the reference's bits).  dst_k may be src_k.                                              */
int beso_scale_rows(const float* const* src, float* const* dst, const float* const* mean, const float* const* den,
                    const long long* rows, const int* cols, int n, void* stream);

/* The same call for data-parallel training, where the exchange of the gradients (one all-reduce per range) should start
 * before the backward pass is over.  The gradients of the upper transformer layers l0 .. n_layers-1 and of ln_f are one
 * contiguous range of grads_flat -- [*begin, *end) floats, beso_grad_early_range -- and are completed FIRST: their weight
 * gradients and LayerNorm sums run as soon as the chain of data gradients has passed layer l0 (chosen so that those weight
 * gradients fill one round of workgroups: 4 of the 6 kitchen layers), and `early_stream`
 * (a second hipStream_t) is then made to wait for exactly that point (an event recorded on `stream`).  Work the caller
 * enqueues on early_stream after the call returns -- the all-reduce of that range -- runs under the backward of the
 * lower layers; everything else in grads_flat is final at the end of `stream` as before.  early_stream = NULL is
 * beso_loss_grad.  With fewer than two layers the range is empty and early_stream is left alone.                    */
int beso_grad_early_range(const beso_config* cfg, size_t* begin, size_t* end);
int beso_loss_grad_overlap(const beso_config* cfg, const float* const* params, int n_params, float* grads_flat, int precision,
                           const float* state, const float* action, const float* goal, const float* noise, const float* sigma,
                           float* loss_out, int batch, int t, int flags, float embed_pdrop, float attn_pdrop, float resid_pdrop,
                           float goal_drop, unsigned int seed, float grad_scale, void* workspace, size_t workspace_bytes,
                           void* stream, void* early_stream);
/* ... and with a third stream for the LOSS: `loss_stream` (NULL: none) is ordered behind the point where *loss_out is final --
 * the end of the forward half, a third of the way into the call's work.  The reference's train_step returns `loss.item()`
 * (beso_agent.py:248); reading the loss on loss_stream lets the host return with it while the backward pass and the optimizer
 * are still running on `stream`, and prepare the next step under them.  The call also uses loss_stream at its start, for the
 * step's copies of the weights (they depend on the parameters only and run beside the embedding of the batch on `stream`;
 * both streams are joined before the first layer): loss_stream must not carry unrelated work of the caller's that the step
 * should not wait for.                                                                                                   */
int beso_loss_grad_streams(const beso_config* cfg, const float* const* params, int n_params, float* grads_flat, int precision,
                           const float* state, const float* action, const float* goal, const float* noise, const float* sigma,
                           float* loss_out, int batch, int t, int flags, float embed_pdrop, float attn_pdrop, float resid_pdrop,
                           float goal_drop, unsigned int seed, float grad_scale, void* workspace, size_t workspace_bytes,
                           void* stream, void* early_stream, void* loss_stream);
/* Launch-site timers (bench.py's roofline, the launch-count assertions of the tests): while a site is selected ON THE
 * CALLING THREAD, HIP events are recorded on the launch stream around every launch that thread makes at that site (site 0 =
 * off).  beso_profile_read synchronises the events the calling thread recorded, returns their summed elapsed time and count,
 * and clears them.  Thread-local: other threads' calls are neither timed nor affected.                                      */
enum {
    BESO_SITE_OFF = 0, BESO_SITE_GEMM_QKV = 1, BESO_SITE_GEMM_PROJ = 2, BESO_SITE_GEMM_FC1 = 3,
    BESO_SITE_GEMM_FC2 = 4, BESO_SITE_ATTENTION = 5, BESO_SITE_LAYERNORM = 6, BESO_SITE_EMBED = 7,
    BESO_SITE_HEAD = 8, BESO_SITE_FORWARD = 9 /* one whole score-net forward */,
    BESO_SITE_FUSED_LAYER = 10 /* the fused kernels (one-launch kernel, block kernels) */,
    BESO_SITE_SMALL = 11 /* the layers of a forward on the chip-wide small-batch path (one pair of events per forward) */
};
void beso_profile_enable(int site);
int  beso_profile_read(double* total_ms, int* launches);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* BESO_HIP_H */
