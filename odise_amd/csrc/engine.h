// engine.h — host-side graph executor pieces shared by the model stages (UNet, VAE, CLIP, heads):
// weight store (checkpoint-keyed host tensors -> packed device tensors), activation arena, layer structs and
// thin launch helpers over the op-level C ABI.  All device work goes through the exported ops, so what the
// op-level parity tests cover is exactly what the stages run.
#pragma once
#include <map>
#include <utility>
#include <string>
#include <vector>

#include "common.h"

namespace odise {

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

// Bump allocator over one hipMalloc'ed slab; mark()/release() give stack-like reuse inside a forward so the
// working set of a block stays small (L2 / Infinity-Cache resident) and pointers are identical call to call
// (required for hipGraph replay).
struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    void* alloc(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        if (off + bytes > cap) return nullptr;
        void* p = base + off;
        off += bytes;
        if (off > peak) peak = off;
        return p;
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
    void reset() { off = 0; }
};

struct Act {  // NHWC fp16 activation
    f16* p = nullptr;
    int n = 0, h = 0, w = 0, c = 0;
    // GroupNorm statistics fused into the conv that produces this tensor (gemm.hip GemmEpi::gn_stats): per-channel (sum, sumsq) of
    // every row block, [n][gn_blocks][c][2] fp32.  gn_part = buffer (Exec::alloc_gn_stats), gn_blocks = row blocks per image once a
    // conv has filled it (0 = not available: Exec::group_norm runs its own statistics pass)
    float* gn_part = nullptr;
    int gn_blocks = 0;
    int64_t pixels() const { return (int64_t)n * h * w; }
    int64_t elems() const { return pixels() * c; }
};

struct ConvW {
    f16* w = nullptr;   // [cout][k][k][cin_pad]
    float* b = nullptr; // [cout] or null
    int cin = 0, cin_pad = 0, cout = 0, k = 1;
};
struct LinW {
    f16* w = nullptr;   // [out][in]
    float* b = nullptr;
    int in = 0, out = 0;
};
struct NormW {
    float* g = nullptr;
    float* b = nullptr;
    int c = 0;
};

struct UNetModel;

// ---- encoder prefetch ------------------------------------------------------------------------------------------------------------------
// The reference's evaluation loop is a stream of batches with the loader prefetching (odise/evaluation/evaluator.py:87-126, odise/data/build.py:
// 138-151).  Here the NEXT batch's input side - normalise / pad, window extraction, VAE encoder, latent - can be enqueued while the current
// batch still runs: odise_hip_infer_prefetch registers the next batch (`pending`); the following odise_hip_infer enqueues that work on the
// context's lowest-priority stream once its own VAE lane is done (its second stream is then idle and its tail is a chain of small launches),
// into one of two side arenas; the odise_hip_infer of exactly that batch (`ready`) starts from the stored latent and encoder taps.  Same
// kernels on the same shapes: the results are bit-identical to the unpipelined call (tests/test_gpu_fullsize_batch.py).
struct EncoderOut {
    Act tap0, tap1;   // inputs of encoder blocks 5 / 7 (ldm.py:437-438)
    Act xt, zdec;     // q_sample(t = 0) of the scaled posterior mean; post_quant_conv input of the decoder
};
struct PrefetchKey {
    int B = 0, layout = 0;
    std::vector<const void*> images;
    std::vector<int> hw;
    bool operator==(const PrefetchKey& o) const { return B == o.B && layout == o.layout && images == o.images && hw == o.hw; }
};
struct Prefetch {
    bool has_pending = false, has_ready = false;
    PrefetchKey pending, ready;
    int ready_slot = 0;           // side arena that holds `ready`'s tensors (the slots alternate: the call in progress may still read the other one)
    EncoderOut out;               // of `ready`
    int crops = 0;                // crops (B x windows) the stored tensors cover
    bool use_now = false;         // set by odise_hip_infer for the call in progress: extractor_launch consumes `out`
    int n_enqueued = 0, n_hits = 0, n_dropped = 0, n_failed = 0;   // odise_hip_prefetch_stats: encoders enqueued ahead / consumed by the next call / prepared but not consumed / could not be enqueued
    Arena arena[2];
};

// MaskCLIP in two passes.  The attention mask the reference builds (clip.py:307-318) hides the mask tokens from EVERY query: the 577 image tokens
// of a picture run the plain tower whatever the masks are, and the Q mask tokens only read that stream's keys and values.  The image-token pass
// therefore needs nothing of the mask head: odise_hip_infer lets the pictures RIDE IN THE CROPS' TOWER (the implicit captioner's CLIP, the same
// frozen ViT-L/14@336 weights, clip.py:77-97 / 239-246: B pictures + B x K crops are one batch of token rows on the second lane, whose GEMM grids
// have room in their last round), which leaves q|k and V^T of the pictures' rows of every block here; the mask-token pass (B x Q rows) is what
// remains on the serial tail behind the mask head.
// Layout: the pictures are the FIRST images of the tower's batch, and block l writes its q|k rows at qk + l * qk_stride and its V^T columns at
// vt + l * vt_stride of a matrix with ldvt columns: the rows / columns of the other images (dead once the block's attention has run) lie where the
// next blocks' picture rows / columns go and are overwritten by them - one buffer of layers x pictures + 1 x others instead of layers x everything.
struct ClipKV {
    void* buf = nullptr;     // device, grows
    size_t cap = 0;
    f16* qk = nullptr;       // block l, picture b, token t: qk + l * qk_stride + (b * TP + t) * 2 * width  (q | k halves)
    f16* vt = nullptr;       // block l, channel c, picture b, token t: vt + l * vt_stride + c * ldvt + b * TP + t
    f16* cls = nullptr;      // [width] the class-token row after ln_pre: what every mask token starts from (clip.py:268-270)
    int B = 0, TP = 0, layers = 0, width = 0;
    size_t qk_stride = 0, vt_stride = 0;   // elements per block
    int64_t ldvt = 0;
    // the pass odise_hip_infer planned for its pictures, and whether it has been enqueued
    const float* plan_image = nullptr;
    int plan_B = 0, plan_H = 0, plan_W = 0;
    bool planned = false, ready = false, on_lane2 = false;   // on_lane2: a pass of its own on the second lane, published by ev_mclip
};

struct ModelStore {
    std::map<std::string, HostTensor> host;
    std::vector<void*> dev_allocs;              // weights that live as long as the context
    std::vector<void*>* alloc_sink = nullptr;   // the stage being built owns what is uploaded meanwhile (AllocScope); rebuilding a stage frees it
    void track(void* p) { (alloc_sink ? *alloc_sink : dev_allocs).push_back(p); }
    Arena arena;
    Arena arena2;   // activations of the second lane (Lane2)
    Prefetch pf;    // encoder prefetch of the next batch (odise_hip_infer_prefetch)
    ClipKV mclip;   // MaskCLIP image-token pass of the batch in progress
    UNetModel* unet = nullptr;
    struct ExtractorModel* extractor = nullptr;
    struct MaskGenModel* maskgen = nullptr;
    struct ClassifyModel* classify = nullptr;
    double macs = 0.0;  // analytic MACs of the ops launched since the last reset
};

ModelStore* store_of(odise_hip_ctx* ctx);
// Device weights of one stage: while an AllocScope lives, Packer::upload records its allocations in `owned`; free_allocs (device-synchronising)
// releases them when the stage is rebuilt or destroyed, so reloading a head / swapping a model does not grow the footprint.
struct AllocScope {
    ModelStore* ms;
    std::vector<void*>* prev;
    AllocScope(ModelStore* m, std::vector<void*>& owned) : ms(m), prev(m->alloc_sink) { m->alloc_sink = &owned; }
    ~AllocScope() { ms->alloc_sink = prev; }
};
void free_allocs(std::vector<void*>& owned);

// ---- weight packing (host) + upload -------------------------------------------------------------------------
struct Packer {
    odise_hip_ctx* ctx;
    ModelStore* ms;
    std::string prefix;
    std::string missing;  // first missing key (error reporting)

    const HostTensor* find(const std::string& key);
    int upload(const void* host, size_t bytes, void** dev);
    int conv(const std::string& key, ConvW& out, bool bias = true);     // key.weight [O,I,kh,kw] (+ key.bias)
    int linear(const std::string& key, LinW& out, bool bias = true);    // key.weight [O,I] or [O,I,1,1]
    int norm(const std::string& key, NormW& out);                       // key.weight/bias [C]
    int vec_f32(const std::string& key, float** out, int64_t expect);   // raw fp32 vector
};

// ---- launch helpers (count MACs, allocate outputs from the arena) ---------------------------------------------
struct Exec {
    odise_hip_ctx* ctx;
    ModelStore* ms;
    int alloc(Act& a, int n, int h, int w, int c);
    int alloc_gn_stats(Act& a);  // reserve the statistics buffer of an (already shaped) activation that a conv is about to write
    void* alloc_bytes(size_t bytes);

    int conv(const Act& x, const ConvW& w, Act& y, int stride = 1, int pad = -1, bool upsample = false, const Act* residual = nullptr,
             const float* per_image_add = nullptr, int64_t pia_ld = 0, int act = ODISE_ACT_NONE, int pad_t = -1, int pad_l = -1,
             int oh = -1, int ow = -1);
    // y[M, out] = act(x[M,in] W^T + b) (+ residual)
    int linear(const f16* x, int64_t M, const LinW& w, f16* y, int act = ODISE_ACT_NONE, const f16* residual = nullptr,
               bool geglu = false);
    int group_norm(const Act& x, const NormW& w, Act& y, float eps, int act);
    int layer_norm(const f16* x, f16* y, int64_t rows, const NormW& w, float eps);
    int gemm(const odise_gemm_desc& d);
    int attention(const odise_attn_desc& d);
};


// ---- second lane ---------------------------------------------------------------------------------------------------------------------
// While a Lane2 object lives, everything enqueued through the context goes to its second stream, allocates from the second arena and
// uses the second split-K workspace: two independent branches of a stage can then execute concurrently (small, latency-bound launches of
// one fill the CUs the other leaves idle) without sharing any scratch memory.  Ordering between the lanes is by events (lane2_*).
int ensure_lane2(odise_hip_ctx* ctx, ModelStore* ms, size_t arena_bytes);
struct Lane2 {
    odise_hip_ctx* ctx;
    ModelStore* ms;
    hipStream_t s0;
    void* w0;
    Lane2(odise_hip_ctx* c, ModelStore* m) : ctx(c), ms(m), s0(c->stream), w0(c->ws) {
        ctx->stream = ctx->stream2;
        ctx->ws = ctx->ws2;
        std::swap(ms->arena, ms->arena2);
    }
    ~Lane2() {
        ctx->stream = s0;
        ctx->ws = w0;
        std::swap(ms->arena, ms->arena2);
    }
};

// While a PrefetchLane lives, everything enqueued through the context goes to the prefetch stream (created on first use, lowest priority),
// allocates from side arena `slot` and uses the third split-K workspace.
int ensure_prefetch_lane(odise_hip_ctx* ctx, ModelStore* ms, int slot, size_t arena_bytes);
struct PrefetchLane {
    odise_hip_ctx* ctx;
    ModelStore* ms;
    int slot;
    hipStream_t s0;
    void* w0;
    void* stages0;
    PrefetchLane(odise_hip_ctx* c, ModelStore* m, int sl) : ctx(c), ms(m), slot(sl), s0(c->stream), w0(c->ws), stages0(c->stages) {
        ctx->stream = ctx->stream3;
        ctx->ws = ctx->ws3;
        ctx->stages = nullptr;     // the stage timeline describes the batch in progress only
        std::swap(ms->arena, ms->pf.arena[slot]);
    }
    ~PrefetchLane() {
        ctx->stream = s0;
        ctx->ws = w0;
        ctx->stages = stages0;
        std::swap(ms->arena, ms->pf.arena[slot]);
    }
};

// ---- stage entry points shared between translation units --------------------------------------------------------
int ensure_arena(odise_hip_ctx* ctx, ModelStore* ms, size_t bytes);
int unet_build(odise_hip_ctx* ctx, const char* prefix);
int unet_prepare_timestep(odise_hip_ctx* ctx, ModelStore* ms, UNetModel* u, int B, int t);
int unet_launch(odise_hip_ctx* ctx, ModelStore* ms, UNetModel* u, const float* x_t, const f16* x_nhwc, const float* context,
                const float* cond_emb, int B, int h, int w, bool standalone);
const Act* unet_taps(ModelStore* ms);
size_t extractor_arena_bytes(int B, int H, int W);
const Act* extractor_taps(ModelStore* ms);
bool extractor_ready(ModelStore* ms);
int extractor_launch(odise_hip_ctx* ctx, ModelStore* ms, const float* image, int B, int H, int W, bool standalone, bool join = true);
int extractor_join(odise_hip_ctx* ctx);
int extractor_encoder_only(odise_hip_ctx* ctx, ModelStore* ms, const float* image, int B, int H, int W, EncoderOut& out);   // the VAE encoder + latent of `image` on the current stream / arena
size_t encoder_arena_bytes(int B, int H, int W);

// misc.hip
struct LatentW {
    float wq[4][8];   // quant_conv rows 0..3 (the posterior mean)
    float bq[4];
    float wp[4][4];   // post_quant_conv
    float bp[4];
    float scale, qa, qb;  // scale_factor, sqrt(alpha_bar_t), sqrt(1 - alpha_bar_t)
};
int launch_latent_heads(odise_hip_ctx* ctx, const f16* h, const float* noise, f16* xt, f16* zdec, float* latent, int B, int P,
                        const LatentW& w);
int launch_image_to_nhwc(odise_hip_ctx* ctx, const float* x, f16* y, int N, int C, int HW, int Cpad, const float* scale3,
                         const float* shift3);
int launch_clip_preprocess(odise_hip_ctx* ctx, const float* x, f16* y, int N, int H, int W, int S);
int launch_softmax_rows(odise_hip_ctx* ctx, const f16* x, f16* y, int64_t rows, int cols, int64_t ld, float scale);
int launch_clip_assemble(odise_hip_ctx* ctx, const f16* patches, const float* cls, const float* pos, f16* tok, int B, int T, int extra,
                         int TP, int Cw);
int clip_tower(Exec& ex, const Act& img, int extra, const uint8_t* mask, int64_t ldm, f16* out, ClipKV* kv = nullptr, int kv_images = 0);
bool maskclip_kv_available(odise_hip_ctx* ctx, ModelStore* ms, int B);            // the store of pass 1 on its own can be reserved (else: run the one-pass form)
int maskclip_image_pass(Exec& ex, const float* image01, int B, int H, int W);   // -> ms->mclip (enqueued on the context's current stream)
int maskclip_mask_pass(Exec& ex, int Q, const uint8_t* mask, int64_t ldm, int64_t stride_mask, f16* out);
int maskclip_planned_pass(odise_hip_ctx* ctx, ModelStore* ms);                   // backbone stage, both lanes enqueued: ODISE_OPT_MASKCLIP_PASSES 3
int clip_dims(ModelStore* ms, int* image, int* patch, int* tokens, int* out_dim);
int launch_cond_inputs(odise_hip_ctx* ctx, const float* proj, const float* A1, const float* A2, float* out, int B, int T, int Cw);


// decoder_ops.hip
int launch_crop_extract(odise_hip_ctx* ctx, const float* img, float* crops, int B, int C, int H, int W, int S, int K, const int* boxes_dev);
int launch_crop_resize_bicubic(odise_hip_ctx* ctx, const float* img, float* crops, int B, int C, int H, int W, int s, int S, int K, const int* boxes_dev);
int launch_upsample_nearest(odise_hip_ctx* ctx, const f16* x, f16* y, int N, int H, int W, int OH, int OW, int C);
int launch_stitch(odise_hip_ctx* ctx, const f16* feat, f16* out, float* out_nchw, int B, int K, const int* boxes_dev, int ch, int cw, int OH,
                  int OW, int C);
int launch_broadcast_rows(odise_hip_ctx* ctx, const f16* x, f16* y, int64_t n, int B);
int layer_norm_add_table(odise_hip_ctx* ctx, const void* x, void* y, const float* gamma, const float* beta, int rows, int C, float eps, f16* y2,
                         const float* table, int P);   // y = LayerNorm(x), y2 (optional; C <= 256) = y + table[row % P]
int launch_add_vec_table(odise_hip_ctx* ctx, const f16* x, const float* vec, const float* table, f16* y, int64_t N, int P, int C);
bool msda_fused_ok(int M, int D, int L, int P);
int launch_msda_fused(odise_hip_ctx* ctx, const f16* value, const float* off, const float* aw, f16* out, const int* Hs, const int* Ws, const int* starts, int B,
                      int S, int M, int Lq, int ld_off = 0, int ld_aw = 0);   // msda.hip: softmax + sampling locations + gather in one kernel (L = 3, P = 4, D = 32)
int launch_msda_prepare(odise_hip_ctx* ctx, const float* off, const float* aw, float* loc, float* w, int B, int Lq, int M, int L, int P,
                        const int* Hs, const int* Ws, const int* starts);
int launch_bilinear_add(odise_hip_ctx* ctx, const f16* a, const f16* b, f16* y, int N, int H, int W, int OH, int OW, int C);
int launch_mask_binarize_f16(odise_hip_ctx* ctx, const f16* mask, f16* m01, float* inv, int64_t rows, int HW);
int launch_attn_mask(odise_hip_ctx* ctx, const f16* logits, uint8_t* out, int64_t rows, int H, int W, int oh, int ow, int64_t ldm);
int launch_attn_mask_f32(odise_hip_ctx* ctx, const float* logits, uint8_t* out, int64_t rows, int H, int W, int oh, int ow, int64_t ldm);


// classify_ops.hip
struct PostGeom {
    int h4, w4;      // mask logits resolution
    int ph, pw;      // padded network input size (first bilinear target, odise.py:326-331)
    int ih, iw;      // true image size (crop, sem_seg_postprocess)
    int oh, ow;      // requested output size
    int Q, Qpad;
};
int launch_resize_bilinear_norm(odise_hip_ctx* ctx, const float* x, f16* y, int B, int H, int W, int S);
int launch_maskclip_token_mask(odise_hip_ctx* ctx, const f16* logits, uint8_t* out, int B, int Q, int h, int w, int S, int patch, int T,
                               int64_t ldm);
int launch_l2_normalize_f16(odise_hip_ctx* ctx, const f16* x, f16* y, int64_t rows, int C);
int launch_l2_normalize_f32(odise_hip_ctx* ctx, const float* x, f16* y, int64_t rows, int C);
int launch_classify_rows(odise_hip_ctx* ctx, const float* L1, const float* L2, const int* seg, const int* ovl, const float* binary, float* out, int64_t rows, int K,
                         int Ktot, float ls1, float ls2, float alpha, float beta);
int launch_postprocess_pixels(odise_hip_ctx* ctx, const f16* logits, const float* kscore, f16* S, int* ids, int* counts, const PostGeom& g, const f16* PT = nullptr,
                              float* sem = nullptr, int K = 0, unsigned int* stats_partial = nullptr);
bool postprocess_pixels_tiled(const PostGeom& g);
int postprocess_pixels_stat_blocks(const PostGeom& g);
int launch_column_fold(odise_hip_ctx* ctx, const unsigned int* partial, float* out2, int nb, int Qpad);   // PT / sem: the semantic scores from the same pass (postprocess_pixels_fuses_semantic)
bool postprocess_pixels_fuses_semantic(const PostGeom& g);
int launch_column_stats(odise_hip_ctx* ctx, const f16* S, unsigned int* partial, float* out2, int npix, int Qpad);
int launch_panoptic_write(odise_hip_ctx* ctx, const int* ids, const int* map, int* seg, int npix);
int launch_instance_masks(odise_hip_ctx* ctx, const f16* logits, const int* idx, float* out, int n, const PostGeom& g, const int* n_dev = nullptr);
int launch_post_decide(odise_hip_ctx* ctx, const float* mask_cls, float* kscore, int* label, f16* semT, float* probs, int B, int Q, int Qpad, int K,
                       float object_mask_threshold);
int launch_panoptic_decide(odise_hip_ctx* ctx, const int* counts, const float* kscore, const int* label, const uint8_t* isthing, int* map, int* table,
                           int Q, int K, double overlap_threshold, int max_segments, int* stuff_scratch);
int launch_instance_topk(odise_hip_ctx* ctx, const float* probs, const float* inst_stats, const uint8_t* isthing, int* table, float* scores, int B,
                         int Q, int Qpad, int K, int topk, int things_only);
int launch_image_pad(odise_hip_ctx* ctx, const void* src, int layout, int h, int w, float* dst, int Hp, int Wp);
int launch_semantic_argmax(odise_hip_ctx* ctx, const f16* S, const f16* PT, int* out, int npix, int Qpad, int K);
// maskgen.cpp accessors used by classify.cpp
struct HeadOutputs {
    const f16* pred_masks;   // [B, Q, h4*w4] logits
    const f16* mask_embed;   // [B, Q, C]
    int B, Q, C, h4, w4;
    float logit_scale;
    const float* class_logits;  // [B, Q, 2] learned (object, no-object) logits of the caption variant's class_embed, or nullptr
};
int head_outputs(ModelStore* ms, HeadOutputs* out);
void maskgen_invalidate_outputs(ModelStore* ms);   // the arena was reallocated: outputs of earlier backbone / head calls are gone

// gemm.hip / norm.hip internals used by Exec
int conv_forced(odise_hip_ctx* ctx, const odise_conv_desc* d, int force_tile, int force_split, float* gn_stats, int* stats_blocks);
int group_norm_from_colpart(odise_hip_ctx* ctx, const void* x, void* y, const float* gamma, const float* beta, int N, int HW, int C, int groups,
                            float eps, int act, const float* colpart, int nblk);

}  // namespace odise
