// extern "C" entry points of libbeso_hip.so (include/beso_hip.h) and the host-side orchestration of
// one score-network forward / one sampling loop.  No allocation, no synchronisation: everything is
// enqueued on the caller's stream.
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <vector>
#include "common.h"
#include "fused.h"
#include "train.h"
#if BESO_DEV_API
#include "../../include/beso_hip_debug.h"
#endif

namespace beso {

// ---------------------------------------------------------------------------------------------
int validate_config(const beso_config* c) {
    if (!c) return BESO_ERR_BAD_ARG;
    if (c->obs_dim < 1 || c->act_dim < 1 || c->act_dim > 64 || c->embed_dim < 8 || c->n_layers < 1 ||
        c->n_layers > kMaxLayers || c->n_heads < 1 || c->goal_seq_len < 0 || c->obs_seq_len < 1)
        return BESO_ERR_BAD_CONFIG;
    if (c->embed_dim % c->n_heads != 0) return BESO_ERR_BAD_CONFIG;
    if (c->embed_dim > 1024) return BESO_ERR_UNSUPPORTED;              // LayerNorm keeps a row in registers
    if (c->embed_dim / c->n_heads > 128) return BESO_ERR_UNSUPPORTED;  // attention keeps q/o rows in registers
    if (!(c->sigma_data > 0.f)) return BESO_ERR_BAD_CONFIG;
    return BESO_OK;
}

static size_t carve(size_t& cur, size_t bytes) {
    size_t off = cur;
    cur = round_up_sz(cur + bytes, 256);
    return off;
}

bool make_layout(const beso_config* c, int precision, Layout* o) {
    if (validate_config(c) != BESO_OK) return false;
    if (precision != BESO_PREC_BF16 && precision != BESO_PREC_FP32 && precision != BESO_PREC_BF16X3 && precision != BESO_PREC_FP16)
        return false;
    memset(o, 0, sizeof(*o));
    o->D = c->embed_dim; o->H = c->n_heads; o->hd = o->D / o->H; o->L = c->n_layers;
    o->G = c->goal_seq_len; o->W = c->obs_seq_len; o->obs = c->obs_dim; o->act = c->act_dim;
    o->seq_size = o->G + o->W + 1;
    o->linear_output = c->linear_output ? 1 : 0;
    o->Kd = round_up(o->D, 64); o->Kh = round_up(4 * o->D, 64);
    o->Nqkv = round_up(3 * o->D, kTileMN); o->Nd = round_up(o->D, kTileMN); o->Nh = round_up(4 * o->D, kTileMN);
    o->elem_bytes = precision == BESO_PREC_FP32 ? 4 : 2;
    size_t cur = 0;
    const size_t f = sizeof(float), e = (size_t)o->elem_bytes;
    o->pos_emb = carve(cur, f * o->seq_size * o->D);
    o->tok_w = carve(cur, f * o->D * o->obs); o->tok_b = carve(cur, f * o->D);
    o->sig_w = carve(cur, f * o->D); o->sig_b = carve(cur, f * o->D);
    o->act_w = carve(cur, f * o->D * o->act); o->act_b = carve(cur, f * o->D);
    o->lnf_w = carve(cur, f * o->D); o->lnf_b = carve(cur, f * o->D);
    if (o->linear_output) {
        o->head_w0 = carve(cur, f * o->act * o->D); o->head_b0 = carve(cur, f * o->act);
        o->head_w1 = o->head_w0; o->head_b1 = o->head_b0;
    } else {
        o->head_w0 = carve(cur, f * kHeadHidden * o->D); o->head_b0 = carve(cur, f * kHeadHidden);
        o->head_w1 = carve(cur, f * o->act * kHeadHidden); o->head_b1 = carve(cur, f * o->act);
    }
    for (int l = 0; l < o->L; ++l) {
        LayerOff& y = o->layer[l];
        y.ln1_w = carve(cur, f * o->D); y.ln1_b = carve(cur, f * o->D);
        y.ln2_w = carve(cur, f * o->D); y.ln2_b = carve(cur, f * o->D);
        y.b_qkv = carve(cur, f * o->Nqkv); y.b_proj = carve(cur, f * o->Nd);
        y.b_fc1 = carve(cur, f * o->Nh); y.b_fc2 = carve(cur, f * o->Nd);
        y.w_qkv = carve(cur, e * o->Nqkv * o->Kd); y.w_proj = carve(cur, e * o->Nd * o->Kd);
        y.w_fc1 = carve(cur, e * o->Nh * o->Kd); y.w_fc2 = carve(cur, e * o->Nd * o->Kh);
    }
    o->fused = carve(cur, precision == BESO_PREC_FP16 ? fused_packed_bytes_f16(*o, BESO_PREC_BF16) : fused_packed_bytes(*o, precision));
    o->total = cur;
    return true;
}

bool make_workspace(const beso_config* c, const Layout& lay, int batch, int t, int precision, int cfg_guidance,
                    Workspace* w) {
    if (batch < 1 || t < 1 || t > c->obs_seq_len) return false;
    memset(w, 0, sizeof(*w));
    const size_t vb = (size_t)batch * (cfg_guidance ? 2 : 1);
    const size_t T = 1 + lay.G + 2 * (size_t)t;
    const size_t M = round_up_sz(vb * T, kTileMN);        // rows padded so tile loads never need a clamp
    // (the split-bf16 block kernels of the long-sequence shape exchange q/k/v and the attention output as fp32 rows)
    const size_t e = (precision == BESO_PREC_BF16X3 && fused_has_lin_blocks(lay, precision)) ? 4 : lay.elem_bytes, f = sizeof(float);
    size_t cur = 0;
    w->x = carve(cur, f * M * lay.D);
    w->xn = carve(cur, e * M * lay.Kd);
    w->qkv = carve(cur, e * M * 3 * lay.D);
    w->y = carve(cur, e * M * lay.Kd);
    w->h = carve(cur, e * M * lay.Kh);
    const size_t na = (size_t)batch * t * lay.act;
    w->den = carve(cur, f * na); w->x2 = carve(cur, f * na); w->d1 = carve(cur, f * na);
    w->sig = carve(cur, f * batch);
    w->small = carve(cur, f * (size_t)(lay.H + 1) * kSmallProjRows * lay.D);
    w->fused = carve(cur, 0);
    w->total = cur;
    return true;
}

// ---------------------------------------------------------------------------------------------
// profiling hooks
// ---------------------------------------------------------------------------------------------
// (state of the CALLING THREAD: a thread that times its launch sites does not see, and does not disturb, other callers)
static thread_local int g_prof_site = 0;
static thread_local std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;
static thread_local std::vector<hipEvent_t> g_prof_free;
static thread_local hipEvent_t g_prof_open = nullptr;

static hipEvent_t prof_get_event() {
    if (!g_prof_free.empty()) { hipEvent_t e = g_prof_free.back(); g_prof_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void profile_begin(int site, hipStream_t s) {
    if (g_prof_site != site) return;
    hipEvent_t e = prof_get_event();
    if (!e) return;
    (void)hipEventRecord(e, s);
    g_prof_open = e;
}

void profile_end(int site, hipStream_t s) {
    if (g_prof_site != site) return;
    if (!g_prof_open) return;
    hipEvent_t e = prof_get_event();
    if (!e) return;
    (void)hipEventRecord(e, s);
    g_prof_events.emplace_back(g_prof_open, e);
    g_prof_open = nullptr;
}

// ---------------------------------------------------------------------------------------------
// one forward of the score network (unfused generic path)
// ---------------------------------------------------------------------------------------------
static thread_local char g_last_error[256] = "";
static int record_hip_error(hipError_t e, const char* what, int line) {
    snprintf(g_last_error, sizeof(g_last_error), "%s (%d) from `%s` at api.hip:%d", hipGetErrorName(e), (int)e, what, line);
    return BESO_ERR_HIP;
}
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return record_hip_error(_e, #expr, __LINE__); } while (0)

static int forward_generic(const Layout& lay, const Workspace& ws, const char* packed, int precision,
                           const FwdArgs& a, char* wsp, hipStream_t s, int fused) {
    float* x = (float*)(wsp + ws.x);
    void* xn = wsp + ws.xn; void* qkv = wsp + ws.qkv; void* y = wsp + ws.y; void* h = wsp + ws.h;
    const int M = a.vbatch * a.T;
    auto F = [&](size_t off) { return (const float*)(packed + off); };
    int fused_edges = 0;
    if (fused == 2) {
        // embed -> all transformer layers -> head as ONE launch: the residual tile of 8 samples never leaves
        // the CU's registers (shapes whose head cannot be fused store x and run the head kernel)
        const bool f16 = precision == BESO_PREC_FP16;
        if (!((f16 ? fused_layer_edges_f16(lay) : fused_layer_edges(lay)) & 1)) {
            profile_begin(BESO_SITE_EMBED, s);
            HIP_TRY(launch_embed(lay, packed, a, x, s));
            profile_end(BESO_SITE_EMBED, s);
        }
        profile_begin(BESO_SITE_FUSED_LAYER, s);
        int st = f16 ? fused_layers_f16(lay, packed, a, x, &fused_edges, BESO_PREC_BF16, s)
                     : fused_layers(lay, packed, a, x, &fused_edges, precision, s);
        profile_end(BESO_SITE_FUSED_LAYER, s);
        if (st != BESO_OK) return st;
    } else {
        profile_begin(BESO_SITE_EMBED, s);
        HIP_TRY(launch_embed(lay, packed, a, x, s));
        profile_end(BESO_SITE_EMBED, s);
    }
    const bool lin_blocks = fused == 1 && fused_has_lin_blocks(lay, precision);
    for (int l = 0; l < (fused == 2 ? 0 : lay.L); ++l) {
        const LayerOff& o = lay.layer[l];
        if (lin_blocks && precision == BESO_PREC_BF16X3) {
            // the same two-launch form in split-bf16 arithmetic: lin_block_x3_kernel around the split-bf16 attention kernel on the
            // matrix pipe (round 6; windows of <= 16 tokens and odd head dims: the exact-fp32 kernel)
            if (l == 0) {
                profile_begin(BESO_SITE_FUSED_LAYER, s);
                int st0 = fused_lin_x3(lay, packed, -1, 0, x, nullptr, 0, (float*)qkv, M, s);
                profile_end(BESO_SITE_FUSED_LAYER, s);
                if (st0 != BESO_OK) return st0;
            }
            profile_begin(BESO_SITE_ATTENTION, s);
            HIP_TRY(launch_attention(qkv, y, a.vbatch, a.T, lay.D, lay.H, lay.Kd, BESO_PREC_BF16X3, s));
            profile_end(BESO_SITE_ATTENTION, s);
            profile_begin(BESO_SITE_FUSED_LAYER, s);
            int st = fused_lin_x3(lay, packed, l, l + 1 < lay.L ? l + 1 : -1, x, (const float*)y, lay.Kd, (float*)qkv, M, s);
            profile_end(BESO_SITE_FUSED_LAYER, s);
            if (st != BESO_OK) return st;
            continue;
        } else if (lin_blocks) {
            // long sequences (no fused attention phase): two launches per layer --
            //   attention(q/k/v of this layer)  ->  [proj + residual -> LN2 -> MLP -> LN1 + q/k/v of the NEXT layer]
            // with the residual tile in registers through the second one; layer 0's q/k/v come from their own block
            if (l == 0) {
                profile_begin(BESO_SITE_GEMM_QKV, s);
                int st0 = fused_lin_block(lay, packed, 0, 0, x, qkv, 3 * lay.D, M, s);
                profile_end(BESO_SITE_GEMM_QKV, s);
                if (st0 != BESO_OK) return st0;
            }
            profile_begin(BESO_SITE_ATTENTION, s);
            HIP_TRY(launch_attention(qkv, y, a.vbatch, a.T, lay.D, lay.H, lay.Kd, precision, s));
            profile_end(BESO_SITE_ATTENTION, s);
            profile_begin(BESO_SITE_FUSED_LAYER, s);
            int st = fused_lin_tail(lay, packed, l, x, y, lay.Kd, qkv, M, s);
            profile_end(BESO_SITE_FUSED_LAYER, s);
            if (st != BESO_OK) return st;
            continue;
        } else {
            profile_begin(BESO_SITE_LAYERNORM, s);
            HIP_TRY(launch_layernorm(x, F(o.ln1_w), F(o.ln1_b), xn, M, lay.D, lay.Kd, precision, s));
            profile_end(BESO_SITE_LAYERNORM, s);
            profile_begin(BESO_SITE_GEMM_QKV, s);
            HIP_TRY(launch_gemm(precision, EPI_BIAS_STORE, xn, lay.Kd, packed + o.w_qkv, lay.Kd, F(o.b_qkv), qkv,
                                3 * lay.D, 3 * lay.D, M, lay.Nqkv, lay.Kd, s));
            profile_end(BESO_SITE_GEMM_QKV, s);
            profile_begin(BESO_SITE_ATTENTION, s);
            HIP_TRY(launch_attention(qkv, y, a.vbatch, a.T, lay.D, lay.H, lay.Kd, precision, s));
            profile_end(BESO_SITE_ATTENTION, s);
            profile_begin(BESO_SITE_GEMM_PROJ, s);
            HIP_TRY(launch_gemm(precision, EPI_BIAS_RESID, y, lay.Kd, packed + o.w_proj, lay.Kd, F(o.b_proj), x, lay.D,
                                lay.D, M, lay.Nd, lay.Kd, s));
            profile_end(BESO_SITE_GEMM_PROJ, s);
        }
        if (fused == 1) {
            // LN2 + FC1 + GELU + FC2 + residual as one kernel, hidden activations never leave the CU
            profile_begin(BESO_SITE_FUSED_LAYER, s);
            int st = fused_mlp_block(lay, packed, l, x, M, s);
            profile_end(BESO_SITE_FUSED_LAYER, s);
            if (st != BESO_OK) return st;
            continue;
        }
        HIP_TRY(launch_layernorm(x, F(o.ln2_w), F(o.ln2_b), xn, M, lay.D, lay.Kd, precision, s));
        profile_begin(BESO_SITE_GEMM_FC1, s);
        HIP_TRY(launch_gemm(precision, EPI_BIAS_GELU_STORE, xn, lay.Kd, packed + o.w_fc1, lay.Kd, F(o.b_fc1), h,
                            lay.Kh, lay.Kh, M, lay.Nh, lay.Kd, s));
        profile_end(BESO_SITE_GEMM_FC1, s);
        profile_begin(BESO_SITE_GEMM_FC2, s);
        HIP_TRY(launch_gemm(precision, EPI_BIAS_RESID, h, lay.Kh, packed + o.w_fc2, lay.Kh, F(o.b_fc2), x, lay.D,
                            lay.D, M, lay.Nd, lay.Kh, s));
        profile_end(BESO_SITE_GEMM_FC2, s);
    }
    if (!(fused_edges & 2)) {
        profile_begin(BESO_SITE_HEAD, s);
        HIP_TRY(launch_head(lay, packed, a, x, s));
        profile_end(BESO_SITE_HEAD, s);
    }
    return BESO_OK;
}

static int forward(const beso_config* cfg, const void* packed, int precision, const float* state,
                   const float* action, const float* goal, const float* sigma, float* out, int batch, int t,
                   int flags, float cond_lambda, int precondition, void* workspace, size_t workspace_bytes,
                   hipStream_t s) {
    int st = validate_config(cfg);
    if (st != BESO_OK) return st;
    if (precision != BESO_PREC_BF16 && precision != BESO_PREC_FP32 && precision != BESO_PREC_BF16X3 && precision != BESO_PREC_FP16)
        return BESO_ERR_BAD_ARG;
    if (batch < 1 || t < 1 || t > cfg->obs_seq_len) return BESO_ERR_BAD_SHAPE;
    if (!packed || !state || !action || !sigma || !out || !workspace) return BESO_ERR_BAD_ARG;
    if (cfg->goal_seq_len > 0 && !goal) return BESO_ERR_BAD_ARG;
    if (flags & ~(BESO_FLAG_UNCOND | BESO_PLAN_MASK)) return BESO_ERR_BAD_ARG;
    Layout lay;
    if (!make_layout(cfg, precision, &lay)) return BESO_ERR_BAD_CONFIG;
    // ClassifierFreeSampleModel (classifier_free_sampler.py:35-49)
    bool uncond = (flags & BESO_FLAG_UNCOND) != 0;
    bool two = false;
    if (precondition && !uncond) {
        if (cond_lambda == 0.f) uncond = true;
        else if (cond_lambda != 1.f) two = true;
    }
    Workspace ws;
    if (!make_workspace(cfg, lay, batch, t, precision, two ? 1 : 0, &ws)) return BESO_ERR_BAD_SHAPE;
    if (workspace_bytes < ws.total) return BESO_ERR_WORKSPACE;
    FwdArgs a;
    a.state = state; a.action = action; a.goal = goal; a.sigma = sigma; a.out = out;
    a.batch = batch; a.vbatch = two ? 2 * batch : batch; a.t = t; a.T = 1 + lay.G + 2 * t;
    a.precondition = precondition;
    a.uncond_from = two ? batch : (uncond ? 0 : a.vbatch);
    a.cond_lambda = cond_lambda; a.sigma_data = cfg->sigma_data;
    a.plan = flags & BESO_PLAN_MASK;
    if (precision == BESO_PREC_FP16) a.plan &= ~(BESO_PLAN_PER_OP | BESO_PLAN_BLOCKS);      // (no per-op / block form: the hint is ignored, as the header says)
    if (small_wanted(lay, a, precision)) {
        // few samples: the weights, not the samples, are spread over the chip (small.hip)
        hipError_t e = hipSuccess;
        profile_begin(BESO_SITE_FORWARD, s);
        const int r = forward_small(lay, ws, (const char*)packed, precision, a, (char*)workspace, s, &e);
        profile_end(BESO_SITE_FORWARD, s);
        return r == BESO_ERR_HIP ? record_hip_error(e, "forward_small", __LINE__) : r;
    }
    const int level = precision == BESO_PREC_FP16 ? fused_level_f16(lay, a, BESO_PREC_BF16) : fused_level(lay, a, precision);
    // BF16X3 / FP16 are instances of the one-launch kernel (layers_kernel) -- BF16X3 also of its block-kernel form on the
    // long-sequence shape -- and have no per-op form
    if (precision == BESO_PREC_FP16 && level != 2) return BESO_ERR_UNSUPPORTED;
    if (precision == BESO_PREC_BF16X3 && level != 2 && !(level == 1 && fused_has_lin_blocks(lay, precision))) return BESO_ERR_UNSUPPORTED;
    profile_begin(BESO_SITE_FORWARD, s);
    int r = forward_generic(lay, ws, (const char*)packed, precision, a, (char*)workspace, s, level);
    profile_end(BESO_SITE_FORWARD, s);
    return r;
}

}  // namespace beso

using namespace beso;

extern "C" {

const char* beso_version(void) { return "beso_hip 0.1 (gfx950)"; }

const char* beso_last_error(void) { return g_last_error; }

const char* beso_status_string(int st) {
    switch (st) {
        case BESO_OK: return "ok";
        case BESO_ERR_BAD_CONFIG: return "bad model config";
        case BESO_ERR_BAD_SHAPE: return "bad shape (batch/t out of range; t must be <= obs_seq_len)";
        case BESO_ERR_BAD_ARG: return "bad argument (null pointer or unknown enum)";
        case BESO_ERR_WORKSPACE: return "workspace or packed buffer too small";
        case BESO_ERR_UNSUPPORTED: return "unsupported configuration";
        case BESO_ERR_HIP: return "HIP runtime error";
        default: return "unknown status";
    }
}

int beso_num_params(const beso_config* cfg) {
    if (validate_config(cfg) != BESO_OK) return 0;
    return 3 + 16 * cfg->n_layers + 6 + (cfg->linear_output ? 2 : 4);
}

size_t beso_packed_bytes(const beso_config* cfg, int precision) {
    Layout lay;
    if (!make_layout(cfg, precision, &lay)) return 0;
    return lay.total;
}

int beso_pack_weights(const beso_config* cfg, const float* const* p, int n_params, void* packed_v,
                      size_t packed_bytes, int precision, void* stream) {
    int st = validate_config(cfg);
    if (st != BESO_OK) return st;
    Layout lay;
    if (!make_layout(cfg, precision, &lay)) return BESO_ERR_BAD_ARG;
    if ((precision == BESO_PREC_BF16X3 || precision == BESO_PREC_FP16) && lay.fused == lay.total)
        return BESO_ERR_UNSUPPORTED;   // no fused instance for this shape
    if (!p || !packed_v) return BESO_ERR_BAD_ARG;
    if (n_params != beso_num_params(cfg)) return BESO_ERR_BAD_ARG;
    for (int i = 0; i < n_params; ++i) if (!p[i]) return BESO_ERR_BAD_ARG;
    if (packed_bytes < lay.total) return BESO_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* pk = (char*)packed_v;
    const int D = lay.D;
    int i = 0;
    // fp32 sections: precision -1
#define PACK32(off, rows, cols, rp, cp) HIP_TRY(launch_pack_matrix(p[i++], rows, cols, pk + (off), rp, cp, -1, s))
    // (the GEMM operands of the per-op path are not used by BF16X3 / FP16: their weights live in the fused image only)
#define PACKW(src, off, rows, cols, rp, cp) \
    do { if (precision != BESO_PREC_BF16X3 && precision != BESO_PREC_FP16) \
             HIP_TRY(launch_pack_matrix(src, rows, cols, pk + (off), rp, cp, precision, s)); } while (0)
    PACK32(lay.pos_emb, lay.seq_size, D, lay.seq_size, D);
    PACK32(lay.tok_w, D, lay.obs, D, lay.obs);
    PACK32(lay.tok_b, 1, D, 1, D);
    for (int l = 0; l < lay.L; ++l) {
        const LayerOff& o = lay.layer[l];
        PACK32(o.ln1_w, 1, D, 1, D); PACK32(o.ln1_b, 1, D, 1, D);
        PACK32(o.ln2_w, 1, D, 1, D); PACK32(o.ln2_b, 1, D, 1, D);
        // reference order: key, query, value, proj (score_gpts.py:33-39).  Fused rows: [q | k | v].
        const float *kw = p[i], *kb = p[i + 1], *qw = p[i + 2], *qb = p[i + 3], *vw = p[i + 4], *vb = p[i + 5];
        const float *pw = p[i + 6], *pb = p[i + 7];
        i += 8;
        const size_t e = lay.elem_bytes;
        // zero the whole padded operand first (rows 3D..Nqkv), then drop q,k,v in
        HIP_TRY(hipMemsetAsync(pk + o.w_qkv, 0, e * lay.Nqkv * lay.Kd, s));
        HIP_TRY(hipMemsetAsync(pk + o.b_qkv, 0, sizeof(float) * lay.Nqkv, s));
        PACKW(qw, o.w_qkv, D, D, D, lay.Kd);
        PACKW(kw, o.w_qkv + e * (size_t)D * lay.Kd, D, D, D, lay.Kd);
        PACKW(vw, o.w_qkv + e * (size_t)2 * D * lay.Kd, D, D, D, lay.Kd);
        HIP_TRY(launch_pack_matrix(qb, 1, D, pk + o.b_qkv, 1, D, -1, s));
        HIP_TRY(launch_pack_matrix(kb, 1, D, pk + o.b_qkv + sizeof(float) * D, 1, D, -1, s));
        HIP_TRY(launch_pack_matrix(vb, 1, D, pk + o.b_qkv + sizeof(float) * 2 * D, 1, D, -1, s));
        PACKW(pw, o.w_proj, D, D, lay.Nd, lay.Kd);
        HIP_TRY(launch_pack_matrix(pb, 1, D, pk + o.b_proj, 1, lay.Nd, -1, s));
        const float *f1w = p[i], *f1b = p[i + 1], *f2w = p[i + 2], *f2b = p[i + 3];
        i += 4;
        PACKW(f1w, o.w_fc1, 4 * D, D, lay.Nh, lay.Kd);
        HIP_TRY(launch_pack_matrix(f1b, 1, 4 * D, pk + o.b_fc1, 1, lay.Nh, -1, s));
        PACKW(f2w, o.w_fc2, D, 4 * D, lay.Nd, lay.Kh);
        HIP_TRY(launch_pack_matrix(f2b, 1, D, pk + o.b_fc2, 1, lay.Nd, -1, s));
    }
    PACK32(lay.lnf_w, 1, D, 1, D); PACK32(lay.lnf_b, 1, D, 1, D);
    PACK32(lay.sig_w, 1, D, 1, D); PACK32(lay.sig_b, 1, D, 1, D);          // sigma_emb.weight is [D,1]
    PACK32(lay.act_w, D, lay.act, D, lay.act); PACK32(lay.act_b, 1, D, 1, D);
    if (lay.linear_output) {
        PACK32(lay.head_w0, lay.act, D, lay.act, D); PACK32(lay.head_b0, 1, lay.act, 1, lay.act);
    } else {
        PACK32(lay.head_w0, kHeadHidden, D, kHeadHidden, D); PACK32(lay.head_b0, 1, kHeadHidden, 1, kHeadHidden);
        PACK32(lay.head_w1, lay.act, kHeadHidden, lay.act, kHeadHidden); PACK32(lay.head_b1, 1, lay.act, 1, lay.act);
    }
#undef PACK32
#undef PACKW
    if (i != n_params) return BESO_ERR_BAD_ARG;
    if (precision == BESO_PREC_FP16) return fused_pack_f16(lay, p, pk, BESO_PREC_BF16, s);
    return fused_pack(lay, p, pk, precision, s);
}

size_t beso_workspace_bytes(const beso_config* cfg, int batch, int t, int precision, int cfg_guidance) {
    Layout lay;
    Workspace ws;
    if (!make_layout(cfg, precision, &lay)) return 0;
    if (!make_workspace(cfg, lay, batch, t, precision, cfg_guidance, &ws)) return 0;
    return ws.total;
}

int beso_score_fwd(const beso_config* cfg, const void* packed, int precision, const float* state,
                   const float* action, const float* goal, const float* sigma, float* out, int batch, int t,
                   int flags, void* workspace, size_t workspace_bytes, void* stream) {
    return forward(cfg, packed, precision, state, action, goal, sigma, out, batch, t, flags, 1.0f, 0, workspace,
                   workspace_bytes, (hipStream_t)stream);
}

int beso_denoise_fwd(const beso_config* cfg, const void* packed, int precision, const float* state,
                     const float* action, const float* goal, const float* sigma, float* out, int batch, int t,
                     int flags, float cond_lambda, void* workspace, size_t workspace_bytes, void* stream) {
    return forward(cfg, packed, precision, state, action, goal, sigma, out, batch, t, flags, cond_lambda, 1,
                   workspace, workspace_bytes, (hipStream_t)stream);
}

int beso_sampler_step(int mode, float* out, float* aux, const float* x, const float* x2, const float* den,
                      float c0, float c1, size_t n, void* stream) {
    if (mode < BESO_STEP_DDIM || mode > BESO_STEP_ADD_NOISE || !out || !x || !den) return BESO_ERR_BAD_ARG;
    if (mode == BESO_STEP_ADD_NOISE && !x2) return BESO_ERR_BAD_ARG;
    if ((mode == BESO_STEP_HEUN_PREDICT || mode == BESO_STEP_HEUN_CORRECT) && !aux) return BESO_ERR_BAD_ARG;
    if (mode == BESO_STEP_HEUN_CORRECT && !x2) return BESO_ERR_BAD_ARG;
    if (n == 0) return BESO_OK;
    HIP_TRY(launch_sampler_step(mode, out, aux, x, x2, den, c0, c1, n, (hipStream_t)stream));
    return BESO_OK;
}

int beso_sample(const beso_config* cfg, const void* packed, int precision, int sampler, const float* state,
                const float* goal, float* x, int batch, int t, const float* sigmas, int n_sigmas,
                float cond_lambda, int flags, void* workspace, size_t workspace_bytes, void* stream) {
    int st = validate_config(cfg);
    if (st != BESO_OK) return st;
    if (sampler < BESO_SAMPLER_DDIM || sampler > BESO_SAMPLER_HEUN || (flags & ~(BESO_SAMPLE_STEPWISE | BESO_PLAN_MASK)))
        return BESO_ERR_BAD_ARG;
    const int plan = flags & BESO_PLAN_MASK;
    if (!sigmas || n_sigmas < 2 || !x || !workspace) return BESO_ERR_BAD_ARG;
    if (batch < 1 || t < 1 || t > cfg->obs_seq_len) return BESO_ERR_BAD_SHAPE;
    Layout lay;
    Workspace ws;
    if (!make_layout(cfg, precision, &lay)) return BESO_ERR_BAD_ARG;
    const int two = (cond_lambda != 0.f && cond_lambda != 1.f) ? 1 : 0;
    if (!make_workspace(cfg, lay, batch, t, precision, two, &ws)) return BESO_ERR_BAD_SHAPE;
    if (workspace_bytes < ws.total) return BESO_ERR_WORKSPACE;
    for (int i = 0; i + 1 < n_sigmas; ++i) if (!(sigmas[i] > 0.f)) return BESO_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    char* wsp = (char*)workspace;
    float* den = (float*)(wsp + ws.den);
    float* x2 = (float*)(wsp + ws.x2);
    float* d1 = (float*)(wsp + ws.d1);
    float* sig = (float*)(wsp + ws.sig);
    const size_t n = (size_t)batch * t * lay.act;
    // The evaluations of the loop and the update behind each (coefficients in fp32 on the host, as the reference's 0-d
    // tensors: gc_sampling.py:921-923 DDIM, :205-210 Euler, :296-310 Heun with plain Euler on the last step :301-303).
    std::vector<StepRec> recs;
    std::vector<int> step_first;                  // index of the first evaluation of every sampler step
    for (int i = 0; i + 1 < n_sigmas; ++i) {
        const float si = sigmas[i], sn = sigmas[i + 1];
        step_first.push_back((int)recs.size());
        if (sampler == BESO_SAMPLER_DDIM) {
            // t = -log(sigma); h = t_next - t; x = (sigma_fn(t_next)/sigma_fn(t))*x - expm1(-h)*den
            const float tt = -logf(si), tn = -logf(sn);      // sn == 0 -> tn = +inf -> x = den exactly
            const float h = tn - tt;
            recs.push_back(StepRec{si, expf(-tn) / expf(-tt), expm1f(-h), BESO_STEP_DDIM, 0.f});
        } else if (sampler == BESO_SAMPLER_EULER || sn == 0.f) {
            // gamma = 0: sigma_hat = sigma_i; d = (x - den)/sigma_hat; x += d*(sigma_next - sigma_hat)
            recs.push_back(StepRec{si, si, sn - si, BESO_STEP_EULER, 0.f});
        } else {
            // Heun: predictor, second evaluation at sigma_{i+1}, trapezoid corrector
            recs.push_back(StepRec{si, si, sn - si, BESO_STEP_HEUN_PREDICT, 0.f});
            recs.push_back(StepRec{sn, sn, sn - si, BESO_STEP_HEUN_CORRECT, 0.f});
        }
    }
    step_first.push_back((int)recs.size());
    {
        // ONE launch for the whole loop where the shape has the one-launch kernel (K8 fused into K7: the workgroup that owns a
        // sample from the embedding to the head also applies the update and feeds itself the next input); loops of more than
        // kMaxLoopEvals evaluations are cut at step boundaries (x travels through `x`, the Heun state lives inside a step)
        FwdArgs a;
        a.state = state; a.action = x; a.goal = goal; a.sigma = sig; a.out = x; a.aux = d1;
        a.batch = batch; a.vbatch = two ? 2 * batch : batch; a.t = t; a.T = 1 + lay.G + 2 * t;
        a.precondition = 1;
        a.uncond_from = two ? batch : (cond_lambda == 0.f ? 0 : a.vbatch);
        a.cond_lambda = cond_lambda; a.sigma_data = cfg->sigma_data;
        a.plan = precision == BESO_PREC_FP16 ? (plan & ~(BESO_PLAN_PER_OP | BESO_PLAN_BLOCKS)) : plan;      // (fp16 has no per-op / block form)
        if (!packed || !state || (cfg->goal_seq_len > 0 && !goal)) return BESO_ERR_BAD_ARG;
        const bool f16 = precision == BESO_PREC_FP16;
        // (few samples: every evaluation runs on the chip-wide small-batch path, step by step -- one workgroup carrying a sample
        //  group through the whole loop would stream all the weights alone, evaluation after evaluation)
        if (!(flags & BESO_SAMPLE_STEPWISE) && !small_wanted(lay, a, precision) &&
            (f16 ? fused_can_loop_f16(lay, a, BESO_PREC_BF16) : fused_can_loop(lay, a, precision))) {
            size_t i0 = 0;
            const size_t n_steps = step_first.size() - 1;
            while (i0 < n_steps) {
                size_t i1 = i0 + 1;
                while (i1 < n_steps && step_first[i1 + 1] - step_first[i0] <= kMaxLoopEvals) ++i1;
                SampleSteps S{};
                S.n = step_first[i1] - step_first[i0];
                for (int k = 0; k < S.n; ++k) S.rec[k] = recs[step_first[i0] + k];
                profile_begin(BESO_SITE_FUSED_LAYER, s);
                st = f16 ? fused_layers_f16(lay, (const char*)packed, a, (float*)(wsp + ws.x), nullptr, BESO_PREC_BF16, s, &S)
                         : fused_layers(lay, (const char*)packed, a, (float*)(wsp + ws.x), nullptr, precision, s, &S);
                profile_end(BESO_SITE_FUSED_LAYER, s);
                if (st != BESO_OK) return st;
                i0 = i1;
            }
            return BESO_OK;
        }
    }
    auto fill_sigma = [&](float v) -> hipError_t {
        uint32_t bits; memcpy(&bits, &v, 4);
        return hipMemsetD32Async((hipDeviceptr_t)sig, (int)bits, (size_t)batch, s);
    };
    // the sigma vector of an evaluation is written by the update launch in front of it (the first one by a fill): two
    // dependent launches per evaluation, not three
    HIP_TRY(fill_sigma(sigmas[0]));
    for (int i = 0; i + 1 < n_sigmas; ++i) {
        const float si = sigmas[i], sn = sigmas[i + 1];
        st = beso_denoise_fwd(cfg, packed, precision, state, x, goal, sig, den, batch, t, plan, cond_lambda, workspace,
                              workspace_bytes, stream);
        if (st != BESO_OK) return st;
        if (sampler == BESO_SAMPLER_DDIM) {
            // t = -log(sigma); h = t_next - t; x = (sigma_fn(t_next)/sigma_fn(t))*x - expm1(-h)*den  (gc_sampling.py:921-923)
            const float tt = -logf(si), tn = -logf(sn);      // sn == 0 -> tn = +inf -> x = den exactly
            const float h = tn - tt;
            const float c0 = expf(-tn) / expf(-tt), c1 = expm1f(-h);
            HIP_TRY(launch_sampler_step(BESO_STEP_DDIM, x, nullptr, x, nullptr, den, c0, c1, n, s, sig, sn, batch));
        } else if (sampler == BESO_SAMPLER_EULER || sn == 0.f) {
            // gamma = 0: sigma_hat = sigma_i; d = (x - den)/sigma_hat; x += d*(sigma_next - sigma_hat)  (:205-210, :301-303)
            HIP_TRY(launch_sampler_step(BESO_STEP_EULER, x, nullptr, x, nullptr, den, si, sn - si, n, s, sig, sn, batch));
        } else {
            // Heun: predictor, second evaluation at sigma_{i+1}, trapezoid corrector (:304-310)
            HIP_TRY(launch_sampler_step(BESO_STEP_HEUN_PREDICT, x2, d1, x, nullptr, den, si, sn - si, n, s, sig, sn, batch));
            st = beso_denoise_fwd(cfg, packed, precision, state, x2, goal, sig, den, batch, t, plan, cond_lambda,
                                  workspace, workspace_bytes, stream);
            if (st != BESO_OK) return st;
            HIP_TRY(launch_sampler_step(BESO_STEP_HEUN_CORRECT, x, d1, x, x2, den, sn, sn - si, n, s));     // (sig already holds sigma_{i+1})
        }
    }
    return BESO_OK;
}

int beso_sample_ancestral(const beso_config* cfg, const void* packed, int precision, const float* state, const float* goal,
                          float* x, int batch, int t, const float* sigmas, int n_sigmas, float cond_lambda, float eta,
                          const float* noise, int flags, void* workspace, size_t workspace_bytes, void* stream) {
    int st = validate_config(cfg);
    if (st != BESO_OK) return st;
    if (!sigmas || n_sigmas < 2 || !x || !workspace || !noise || !(eta >= 0.f) || (flags & ~(BESO_PLAN_MASK | BESO_SAMPLE_STEPWISE)))
        return BESO_ERR_BAD_ARG;
    const int plan = flags & BESO_PLAN_MASK;
    if (batch < 1 || t < 1 || t > cfg->obs_seq_len) return BESO_ERR_BAD_SHAPE;
    Layout lay;
    Workspace ws;
    if (!make_layout(cfg, precision, &lay)) return BESO_ERR_BAD_ARG;
    const int two = (cond_lambda != 0.f && cond_lambda != 1.f) ? 1 : 0;
    if (!make_workspace(cfg, lay, batch, t, precision, two, &ws)) return BESO_ERR_BAD_SHAPE;
    if (workspace_bytes < ws.total) return BESO_ERR_WORKSPACE;
    for (int i = 0; i + 1 < n_sigmas; ++i) if (!(sigmas[i] > 0.f)) return BESO_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    char* wsp = (char*)workspace;
    float* den = (float*)(wsp + ws.den);
    float* sig = (float*)(wsp + ws.sig);
    const size_t n = (size_t)batch * t * lay.act;
    // get_ancestral_step (:107-114) in fp32
    auto ancestral = [&](float sf, float sn, float& down, float& up) {
        down = sn; up = 0.f;
        if (eta != 0.f) {
            up = eta * sqrtf(sn * sn * (sf * sf - sn * sn) / (sf * sf));
            if (sn < up) up = sn;
            down = sqrtf(sn * sn - up * up);
        }
    };
    {
        // ONE launch for the whole loop (as beso_sample): every step is an Euler update to sigma_down followed by the caller's
        // noise of that step times sigma_up, both applied in the kernel's head by the workgroup that owns the sample
        FwdArgs a;
        a.state = state; a.action = x; a.goal = goal; a.sigma = sig; a.out = x; a.aux = nullptr;
        a.batch = batch; a.vbatch = two ? 2 * batch : batch; a.t = t; a.T = 1 + lay.G + 2 * t;
        a.precondition = 1;
        a.uncond_from = two ? batch : (cond_lambda == 0.f ? 0 : a.vbatch);
        a.cond_lambda = cond_lambda; a.sigma_data = cfg->sigma_data;
        a.plan = precision == BESO_PREC_FP16 ? (plan & ~(BESO_PLAN_PER_OP | BESO_PLAN_BLOCKS)) : plan;
        if (!packed || !state || (cfg->goal_seq_len > 0 && !goal)) return BESO_ERR_BAD_ARG;
        const bool f16 = precision == BESO_PREC_FP16;
        // (few samples: every evaluation runs on the chip-wide small-batch path, step by step -- one workgroup carrying a sample
        //  group through the whole loop would stream all the weights alone, evaluation after evaluation)
        if (!(flags & BESO_SAMPLE_STEPWISE) && !small_wanted(lay, a, precision) &&
            (f16 ? fused_can_loop_f16(lay, a, BESO_PREC_BF16) : fused_can_loop(lay, a, precision))) {
            const int n_steps = n_sigmas - 1;
            for (int i0 = 0; i0 < n_steps; i0 += kMaxLoopEvals) {
                SampleSteps S{};
                S.n = n_steps - i0 < kMaxLoopEvals ? n_steps - i0 : kMaxLoopEvals;
                for (int k = 0; k < S.n; ++k) {
                    const float sf = sigmas[i0 + k], sn = sigmas[i0 + k + 1];
                    float down, up;
                    ancestral(sf, sn, down, up);
                    S.rec[k] = StepRec{sf, sf, down - sf, BESO_STEP_EULER | (down > 0.f ? kStepAddNoise : 0), up};      // :240-247
                }
                a.noise = noise + (size_t)i0 * n;
                profile_begin(BESO_SITE_FUSED_LAYER, s);
                st = f16 ? fused_layers_f16(lay, (const char*)packed, a, (float*)(wsp + ws.x), nullptr, BESO_PREC_BF16, s, &S)
                         : fused_layers(lay, (const char*)packed, a, (float*)(wsp + ws.x), nullptr, precision, s, &S);
                profile_end(BESO_SITE_FUSED_LAYER, s);
                if (st != BESO_OK) return st;
            }
            return BESO_OK;
        }
    }
    for (int i = 0; i + 1 < n_sigmas; ++i) {
        const float sf = sigmas[i], sn = sigmas[i + 1];
        float down, up;
        ancestral(sf, sn, down, up);
        if (i == 0) {        // (the sigma vector of every later step is written by the previous step's update launch)
            uint32_t bits; memcpy(&bits, &sf, 4);
            HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)sig, (int)bits, (size_t)batch, s));
        }
        st = beso_denoise_fwd(cfg, packed, precision, state, x, goal, sig, den, batch, t, plan, cond_lambda, workspace,
                              workspace_bytes, stream);
        if (st != BESO_OK) return st;
        HIP_TRY(launch_sampler_step(BESO_STEP_EULER, x, nullptr, x, nullptr, den, sf, down - sf, n, s, sig, sn, batch));     // :240-245
        if (down > 0.f) HIP_TRY(launch_sampler_step(BESO_STEP_ADD_NOISE, x, nullptr, x, noise + (size_t)i * n, x, up, 0.f, n, s));
    }
    return BESO_OK;
}

#if BESO_DEV_API
// development builds only (include/beso_hip_debug.h; `python -m beso_amd.build --dev` -> libbeso_hip_dev.so)
void beso_debug_set_stamps(void* device_buf, int capacity_u64) { fused_set_stamps(device_buf, capacity_u64); }
#endif

int beso_adam_step(const beso_optim_chunk* chunks, int n_chunks, float* exp_avg, float* exp_avg_sq, float* ema,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled_wd, int step,
                   float ema_decay, void* stream) {
    if (!chunks || !exp_avg || !exp_avg_sq || n_chunks < 0 || step < 1) return BESO_ERR_BAD_ARG;
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && lr >= 0.f)) return BESO_ERR_BAD_ARG;
    if (ema && !(ema_decay >= 0.f && ema_decay <= 1.f)) return BESO_ERR_BAD_ARG;
    if (n_chunks == 0) return BESO_OK;
    hipError_t e = launch_adam_ema(chunks, n_chunks, exp_avg, exp_avg_sq, ema, lr, beta1, beta2, eps, weight_decay,
                                   decoupled_wd ? 1 : 0, step, ema_decay, (hipStream_t)stream);
    if (e != hipSuccess) return record_hip_error(e, "adam_ema_kernel", __LINE__);
    return BESO_OK;
}

int beso_gather_windows(const float* observations, const float* actions, const int* seq_len, int n_traj, int t_max,
                        int obs_dim, int act_dim, const int* slice_traj, const int* slice_start, long long n_slices,
                        const long long* batch_slices, const long long* draws, int batch, int window, int goal_len,
                        int goal_mode, int min_future_sep, float* obs_out, float* act_out, float* goal_out, void* stream) {
    if (!observations || !actions || !seq_len || !slice_traj || !slice_start || !batch_slices || !obs_out || !act_out)
        return BESO_ERR_BAD_ARG;
    if (n_traj < 1 || t_max < 1 || obs_dim < 1 || act_dim < 1 || n_slices < 1 || batch < 0 || window < 1 || window > t_max)
        return BESO_ERR_BAD_ARG;
    if (goal_len < 0 || goal_len > t_max || min_future_sep < 0) return BESO_ERR_BAD_ARG;
    if (goal_mode != BESO_GOAL_RANDOM && goal_mode != BESO_GOAL_TAIL && goal_mode != BESO_GOAL_SEQ_END) return BESO_ERR_BAD_ARG;
    if (goal_len > 0 && (!goal_out || (goal_mode == BESO_GOAL_RANDOM && !draws))) return BESO_ERR_BAD_ARG;
    if ((long long)window * (obs_dim > act_dim ? obs_dim : act_dim) > 0x7fffffffLL) return BESO_ERR_BAD_ARG;
    if (batch == 0) return BESO_OK;
    hipError_t e = launch_gather_windows(observations, actions, seq_len, n_traj, t_max, obs_dim, act_dim, slice_traj,
                                         slice_start, n_slices, batch_slices, draws, batch, window, goal_len, goal_mode,
                                         min_future_sep, obs_out, act_out, goal_out, (hipStream_t)stream);
    if (e != hipSuccess) return record_hip_error(e, "gather_windows_kernel", __LINE__);
    return BESO_OK;
}

size_t beso_train_workspace_bytes(const beso_config* cfg, int batch, int t, int precision) {
    return train_workspace_bytes(cfg, batch, t, precision);
}

size_t beso_grad_floats(const beso_config* cfg) { return train_grad_floats(cfg); }

int beso_loss_grad_streams(const beso_config* cfg, const float* const* params, int n_params, float* grads_flat, int precision,
                           const float* state, const float* action, const float* goal, const float* noise, const float* sigma,
                           float* loss_out, int batch, int t, int flags, float embed_pdrop, float attn_pdrop, float resid_pdrop,
                           float goal_drop, unsigned int seed, float grad_scale, void* workspace, size_t workspace_bytes,
                           void* stream, void* early_stream, void* loss_stream) {
    hipError_t e = hipSuccess;
    int line = 0;
    int st = train_loss_grad(cfg, params, n_params, grads_flat, precision, state, action, goal, noise, sigma, loss_out, batch,
                             t, flags, embed_pdrop, attn_pdrop, resid_pdrop, goal_drop, seed, grad_scale, workspace, workspace_bytes, (hipStream_t)stream,
                             (hipStream_t)early_stream, (hipStream_t)loss_stream, &e, &line);
    if (st == BESO_ERR_HIP) {
        snprintf(g_last_error, sizeof(g_last_error), "%s (%d) at train.hip:%d", hipGetErrorName(e), (int)e, line);
    }
    return st;
}

int beso_loss_grad_overlap(const beso_config* cfg, const float* const* params, int n_params, float* grads_flat, int precision,
                           const float* state, const float* action, const float* goal, const float* noise, const float* sigma,
                           float* loss_out, int batch, int t, int flags, float embed_pdrop, float attn_pdrop, float resid_pdrop,
                           float goal_drop, unsigned int seed, float grad_scale, void* workspace, size_t workspace_bytes,
                           void* stream, void* early_stream) {
    return beso_loss_grad_streams(cfg, params, n_params, grads_flat, precision, state, action, goal, noise, sigma, loss_out, batch, t,
                                  flags, embed_pdrop, attn_pdrop, resid_pdrop, goal_drop, seed, grad_scale, workspace,
                                  workspace_bytes, stream, early_stream, nullptr);
}

int beso_loss_grad(const beso_config* cfg, const float* const* params, int n_params, float* grads_flat, int precision,
                   const float* state, const float* action, const float* goal, const float* noise, const float* sigma,
                   float* loss_out, int batch, int t, int flags, float embed_pdrop, float attn_pdrop, float resid_pdrop,
                   float goal_drop, unsigned int seed, float grad_scale, void* workspace, size_t workspace_bytes, void* stream) {
    return beso_loss_grad_overlap(cfg, params, n_params, grads_flat, precision, state, action, goal, noise, sigma, loss_out, batch, t,
                                  flags, embed_pdrop, attn_pdrop, resid_pdrop, goal_drop, seed, grad_scale, workspace, workspace_bytes,
                                  stream, nullptr);
}

int beso_log_logistic(const double* u, float* out, size_t n, double loc, double scale, double cdf_lo, double cdf_hi, void* stream) {
    if (!u || !out || !(scale > 0.0) || !(cdf_lo >= 0.0 && cdf_hi <= 1.0 && cdf_lo <= cdf_hi)) return BESO_ERR_BAD_ARG;
    if (n == 0) return BESO_OK;
    hipError_t e = launch_log_logistic(u, out, n, loc, scale, cdf_lo, cdf_hi, (hipStream_t)stream);
    if (e != hipSuccess) return record_hip_error(e, "log_logistic_kernel", __LINE__);
    return BESO_OK;
}

int beso_scale_rows(const float* const* src, float* const* dst, const float* const* mean, const float* const* den,
                    const long long* rows, const int* cols, int n, void* stream) {
    if (n < 0 || n > kScaleMax || (n > 0 && (!src || !dst || !mean || !den || !rows || !cols))) return BESO_ERR_BAD_ARG;
    for (int k = 0; k < n; ++k)
        if (rows[k] < 0 || cols[k] < 1 || (rows[k] > 0 && (!src[k] || !dst[k] || !mean[k] || !den[k]))) return BESO_ERR_BAD_ARG;
    hipError_t e = launch_scale_rows(src, dst, mean, den, rows, cols, n, (hipStream_t)stream);
    if (e != hipSuccess) return record_hip_error(e, "scale_rows_kernel", __LINE__);
    return BESO_OK;
}

int beso_goal_mask(float* mask, int batch, int goal_seq_len, int obs_dim, float goal_drop, unsigned int seed, void* stream) {
    if (batch < 0 || goal_seq_len < 0 || obs_dim < 0) return BESO_ERR_BAD_ARG;
    hipError_t e = hipSuccess;
    int line = 0;
    int st = train_goal_mask(mask, (size_t)batch * goal_seq_len * obs_dim, goal_drop, seed, (hipStream_t)stream, &e, &line);
    if (st == BESO_ERR_HIP) snprintf(g_last_error, sizeof(g_last_error), "%s (%d) at train.hip:%d", hipGetErrorName(e), (int)e, line);
    return st;
}

int beso_grad_early_range(const beso_config* cfg, size_t* begin, size_t* end) {
    if (!cfg || !begin || !end) return BESO_ERR_BAD_ARG;
    int st = train_validate(cfg, 1, 1);
    if (st != BESO_OK) return st;
    train_early_range(cfg, begin, end);
    return BESO_OK;
}

#if BESO_DEV_API
int beso_debug_gemm(int precision, int a_kslow, int b_kslow, const void* A, int lda, const void* B, int ldb, float* C,
                    int ldc, int M, int N, int K, int splits, void* stream) {
    hipError_t e = hipSuccess;
    int line = 0;
    int st = train_debug_gemm(precision, a_kslow, b_kslow, A, lda, B, ldb, C, ldc, M, N, K, splits, (hipStream_t)stream, &e,
                              &line);
    if (st == BESO_ERR_HIP) {
        snprintf(g_last_error, sizeof(g_last_error), "%s (%d) at train.hip:%d", hipGetErrorName(e), (int)e, line);
    }
    return st;
}
#endif

void beso_profile_enable(int site) { g_prof_site = site; }

int beso_profile_read(double* total_ms, int* launches) {
    double tot = 0.0;
    int n = 0;
    for (auto& pr : g_prof_events) {
        if (hipEventSynchronize(pr.second) != hipSuccess) return BESO_ERR_HIP;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { tot += ms; ++n; }
        g_prof_free.push_back(pr.first);
        g_prof_free.push_back(pr.second);
    }
    g_prof_events.clear();
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return BESO_OK;
}

}  // extern "C"
