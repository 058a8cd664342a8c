// amp_model: Mask R-CNN R50-FPN inference orchestration (what `DefaultPredictor.__call__` -> GeneralizedRCNN.inference
// does for AMPIS, notebook cells 24-28).  Host C++ only plans buffers and launches the HIP
// stages of this library on the context's stream; there is no CPU compute path.
//
// Parameters arrive under their detectron2 state_dict names (amp_model_load_tensor) in torch layouts and are re-laid for
// the kernels here: conv OIHW -> [Cout][KH][KW][Cin]; stem padded to Cin 4 / KW 8; FrozenBN -> per-channel (scale, shift);
// fc1 columns permuted from (c,ph,pw) to (ph,pw,c); cls_score+bbox_pred and objectness+anchor_deltas fused row-wise;
// ConvTranspose IOHW -> [(ky,kx,co)][Cin].
#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace {

struct ConvW {
    float* w = nullptr;      // device [Cout][KH][KW][Cin]
    float* scale = nullptr;  // device [Cout] or null
    float* shift = nullptr;  // device [Cout] or null
    int cout = 0, cin = 0, kh = 0, kw = 0;
    float* w_split = nullptr; // AMP_CONV_F16X3 operand copy of w (amp_split_weights), made at finalize; null: not eligible / out of range
    float* wt_split = nullptr; // training: split data-gradient form of w (refresh_dgrad_weights, once per step); null: not made
    float w_absmax = 0.f;    // max |w| seen at load (range check of the split copy)
    int groups = 1;          // > 1: w is the window layout [Cout][KH][KW][64] of amp_group_expand_weights (ResNeXt conv2)
};

struct NamedBuf {
    void* ptr;
    int dtype;   // 0 f32, 1 i32, 2 u64
    int ndim;
    long long shape[5];
};

struct Bump {
    char* base = nullptr;
    size_t cap = 0, off = 0, peak = 0;
    bool dry = false;
    void* alloc(size_t bytes) {
        const size_t a = (off + 255) & ~(size_t)255;
        off = a + bytes;
        if (off > peak) peak = off;
        if (dry) return reinterpret_cast<void*>(a + 256);   // non-null fake
        return (off <= cap) ? base + a : nullptr;
    }
    template <class T>
    T* get(size_t n) { return reinterpret_cast<T*>(alloc(n * sizeof(T))); }
};

const char* kStage[4] = {"res2", "res3", "res4", "res5"};
const int kOut[4] = {256, 512, 1024, 2048};
const int kStride[4] = {1, 2, 2, 2};

}  // namespace

struct amp_model {
    amp_ctx* ctx = nullptr;
    amp_model_cfg cfg;
    int nblk[4] = {3, 4, 6, 3};          // RESNETS.DEPTH 50 / 101
    int mid[4] = {64, 128, 256, 512};     // NUM_GROUPS * WIDTH_PER_GROUP * 2^stage
    // ---- parameters ----
    float* parena = nullptr;
    size_t parena_floats = 0, parena_used = 0;
    std::map<std::string, ConvW> conv;                     // keyed by detectron2 module prefix
    std::map<std::string, std::vector<float>> host_raw;    // raw host copies needed at finalize (BN stats, fused parts)
    std::map<std::string, bool> loaded;
    std::vector<std::string> expected;
    bool finalized = false;
    // ---- workspace ----
    Bump ws;
    std::map<std::string, NamedBuf> taps;
    int* d_batch_iota = nullptr;        // [max_batch * post_nms_topk] roi -> image
    int* d_flags = nullptr;             // [4] overflow flags
    // pinned host staging
    int* h_counts = nullptr;            // [max_batch] (pinned)
    int* h_small = nullptr;             // pinned scratch for small uploads
    char* h_res = nullptr;              // pinned: the per-detection results of one call (one D2H copy)
    size_t h_res_bytes = 0;
    std::vector<char> host_out;         // result storage
    // host results of the last call
    std::vector<int> r_n, r_classes, r_rle_len;
    std::vector<float> r_boxes, r_scores;
    std::vector<unsigned long long> r_rle_off;
    std::vector<uint32_t> r_pool;       // (fallback when the pinned pool below could not be allocated)
    uint32_t* h_pool = nullptr;         // pinned, host-cacheable: the run lengths of one call land here by DMA and are handed out in place
    size_t h_pool_counts = 0;
    const uint32_t* r_pool_ptr = nullptr;
    int rle_mode = 0;                   // amp_model_set_rle_output: 0 run lengths, 1 counts strings (encoded on the device), 2 both
    char* h_str = nullptr;              // pinned: the counts strings of one call
    size_t h_str_bytes = 0;
    hipEvent_t ev_counts = nullptr;     // marks the arrival of the detection counts on the host (run(): the wait that is not a stream sync)
    size_t str_bytes_last = 0;      // bytes of counts strings the previous inference produced (the size of the speculative read-back)
    std::vector<unsigned long long> r_str_off;
    std::vector<int> r_str_len;
    std::vector<int> r_out_h, r_out_w;
    float last_stage_ms[8];
    // ---- training ----
    float* garena = nullptr;            // gradients, same offsets as parena
    float* varena = nullptr;            // SGD momentum buffers, same offsets
    bool saving = false;                // run_trunk keeps every activation the backward pass needs
    bool acts_split = false;            // ... and kept them in the split row format (training on the native trunk, AMP_CONV_F16X3)
    int last_rpn_sparse = -1;           // 1: the last backward pass ran the RPN head's gradients over the sampled pixels only (rpn_sparse.hip)
    int last_chain = -1;                // the last backward pass's backbone chain: 0 fp32 storage, 1 split activations + fp32 gradients, 2 scaled split gradients (R50 / R101), 3 the same for ResNeXt blocks
    bool gs_chain_ok = false;           // ... and the backbone's backward chain can run on scaled split gradients (dense 3x3, stride in conv1: R50 / R101)
    bool mask_acts_split = false;       // the mask head's pooled input and fcn1..3 outputs of the last training forward likewise
    bool deconv_out_split = false;      // ... and the deconv's [N,28,28,256] output (round 4)
    bool mask_tail_split = false;       // ... and fcn4's output (the deconv's input): the deconv and its three gradient launches on pre-split operands
    struct BlockAct { std::string key; float *x_in, *t1, *t2, *sc, *out; int in_h, in_w, oh, ow, cin, mid, cout, stride, stage; bool has_sc; };
    std::vector<BlockAct> blocks;
    float* lat[4] = {nullptr, nullptr, nullptr, nullptr};
    float* rpn_t[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool rpn_train_fused = false;       // this training step's RPN head ran with the predictors in the conv's epilogue (no hidden tensor saved): the sparse backward recomputes its rows
    struct Trainable { float* p; size_t n; };
    std::vector<Trainable> trainable;
    unsigned long long* sgd_chunks = nullptr;   // device table for amp::sgd_chunks_run, built at the first amp_model_sgd_step
    int sgd_nchunks = 0;
    struct JobTable { amp::WeightJob* jobs = nullptr; void* chunks = nullptr; int nchunks = 0; };
    JobTable fwd_jobs, dgrad_jobs;              // refresh_split_weights / refresh_dgrad_weights: every layer's split copy in one launch
    bool fwd_jobs_dirty = true, dgrad_jobs_dirty = true;
    float* dgrad_arena = nullptr;               // split data-gradient weights (ConvW::wt_split), AMP_CONV_F16X3 training
    size_t dgrad_floats = 0;
    bool grads_valid = false;
    float* split_arena = nullptr;       // f16x3 operand copies of the weights (inference)
    size_t split_floats = 0;
    bool split_stale = false;
    int f32_reruns = 0;                 // batches re-run in AMP_CONV_F32 after the range flag was raised
    uint8_t* img_stage = nullptr;       // device copy of host images (amp_model_infer / forward_losses with imgs_on_host), grown on demand
    size_t img_stage_bytes = 0;
    std::vector<int> img_hw;            // optional per-image valid sizes for the next batches
    // ---- gradient exchange (comm.hip): merged arena ranges per bucket, in the order the backward pass completes them ----
    std::vector<int> gb_bucket;
    std::vector<size_t> gb_off, gb_n;
    int grad_overlap = -1;              // -1: on when the context has a communicator; 0 / 1: amp_model_set_grad_overlap
    unsigned int* bm_scratch = nullptr; // window bit maps of amp_mask_targets_bitmask (bitmask ground truth), grown on demand
    size_t bm_words = 0;
    int* bm_flag = nullptr;
    unsigned issued_mask = 0;           // buckets of the current gradients already handed to RCCL (bit b); all set = exchanged
};

static int g_split_chain = -1;   // -1: from the environment (AMP_NO_SPLIT_CHAIN), 0 / 1: set by amp_debug_set_split_chain (tests)
extern "C" void amp_debug_set_split_chain(int on) { g_split_chain = on; }
static int g_rpn_train_fuse = -1;    // -1: from the environment (AMP_NO_RPN_TRAIN_FUSE), 0 / 1: set by amp_debug_set_rpn_train_fuse (tests)
extern "C" void amp_debug_set_rpn_train_fuse(int on) { g_rpn_train_fuse = on; }
static int g_rpn_sparse = -1;        // -1: from the environment (AMP_NO_RPN_SPARSE), 0 / 1: set by amp_debug_set_rpn_sparse (tests)
extern "C" void amp_debug_set_rpn_sparse(int on) { g_rpn_sparse = on; }
static int g_mask_tail_split = -1;   // -1: from the environment (AMP_NO_MASK_TAIL_SPLIT), 0 / 1: set by amp_debug_set_mask_tail_split (tests)
extern "C" void amp_debug_set_mask_tail_split(int on) { g_mask_tail_split = on; }
static int g_gx = -1;                // -1: from the environment (AMP_NO_GX), 0 / 1: ResNeXt blocks on fp32 gradients / on the scaled split chain (tests)
extern "C" void amp_debug_set_gx(int on) { g_gx = on; }
extern "C" int amp_debug_last_backward_chain(amp_model* m) { return m ? m->last_chain : -1; }
extern "C" int amp_debug_last_rpn_sparse(amp_model* m) { return m ? m->last_rpn_sparse : -1; }

namespace {

float* palloc(amp_model* m, size_t n) {
    const size_t a = (m->parena_used + 63) & ~(size_t)63;
    if (a + n > m->parena_floats) return nullptr;
    m->parena_used = a + n;
    return m->parena + a;
}

int upload(amp_model* m, float* dst, const std::vector<float>& v) {
    AMP_HIP_CHECK(hipMemcpyAsync(dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice, m->ctx->stream));
    AMP_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    return AMP_OK;
}

void expect_conv_bn(amp_model* m, const std::string& p) {
    m->expected.push_back(p + ".weight");
    for (const char* f : {"weight", "bias", "running_mean", "running_var"}) m->expected.push_back(p + ".norm." + f);
}

void build_expected(amp_model* m) {
    expect_conv_bn(m, "backbone.bottom_up.stem.conv1");
    for (int s = 0; s < 4; ++s)
        for (int b = 0; b < m->nblk[s]; ++b) {
            const std::string p = std::string("backbone.bottom_up.") + kStage[s] + "." + std::to_string(b);
            if (b == 0) expect_conv_bn(m, p + ".shortcut");
            expect_conv_bn(m, p + ".conv1");
            expect_conv_bn(m, p + ".conv2");
            expect_conv_bn(m, p + ".conv3");
        }
    auto wb = [&](const std::string& p) { m->expected.push_back(p + ".weight"); m->expected.push_back(p + ".bias"); };
    for (int l = 2; l <= 5; ++l) { wb("backbone.fpn_lateral" + std::to_string(l)); wb("backbone.fpn_output" + std::to_string(l)); }
    wb("proposal_generator.rpn_head.conv");
    wb("proposal_generator.rpn_head.objectness_logits");
    wb("proposal_generator.rpn_head.anchor_deltas");
    wb("roi_heads.box_head.fc1"); wb("roi_heads.box_head.fc2");
    wb("roi_heads.box_predictor.cls_score"); wb("roi_heads.box_predictor.bbox_pred");
    for (int i = 1; i <= 4; ++i) wb("roi_heads.mask_head.mask_fcn" + std::to_string(i));
    wb("roi_heads.mask_head.deconv"); wb("roi_heads.mask_head.predictor");
}

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// OIHW -> [O][KH][KWp][Cp] with zero padding of the extra taps / channels
std::vector<float> oihw_to_ohwi(const float* w, int O, int I, int KH, int KW, int Cp, int KWp) {
    std::vector<float> out((size_t)O * KH * KWp * Cp, 0.f);
    for (int o = 0; o < O; ++o)
        for (int i = 0; i < I; ++i)
            for (int y = 0; y < KH; ++y)
                for (int x = 0; x < KW; ++x)
                    out[(((size_t)o * KH + y) * KWp + x) * Cp + i] = w[(((size_t)o * I + i) * KH + y) * KW + x];
    return out;
}

int launch_conv(amp_model* m, const ConvW& cw, const float* x, int B, int H, int W, int stride, int pad, bool relu,
                int res_mode, const float* res, int out_mode, float* y, int fmt = 0) {
    amp_conv_desc d;
    d.B = B; d.H = H; d.W = W; d.Cin = cw.cin; d.Cout = cw.cout; d.KH = cw.kh; d.KW = cw.kw;
    d.stride = stride; d.pad = pad; d.relu = relu ? 1 : 0; d.res_mode = res_mode; d.out_mode = out_mode;
    // pre-split weights when the layer has them; a layer whose weights exceed the fp16 range of the split copy stays on fp32 MFMA
    const bool eligible = cw.cin % 32 == 0 || (cw.cin == 4 && cw.kw == 8);   // dense and grouped layers, the padded stem
    // (after an SGD step the split copies are stale until the next inference refreshes them: split per call then)
    return amp::conv_run(m->ctx, &d, cw.groups, x, cw.w, m->split_stale ? nullptr : cw.w_split, (eligible && !cw.w_split) ? 1 : 0, cw.scale, cw.shift, res,
                         nullptr, y, 0, fmt);
}

// Inference in AMP_CONV_F16X3: tensors that only feed convolutions travel in the split operand format (written by the producer's
// epilogue / by RoIAlign, same bytes as fp32), so their consumers stage both operands by LDS-DMA and split nothing.
bool split_chain(amp_model* m, std::initializer_list<const char*> keys) {
    static const bool env_off = getenv("AMP_NO_SPLIT_CHAIN") != nullptr;
    const bool off = g_split_chain < 0 ? env_off : g_split_chain == 0;
    if (off || m->ws.dry || m->ctx->conv_mode != AMP_CONV_F16X3 || m->split_stale) return false;
    for (const char* k : keys) {
        auto it = m->conv.find(k);
        if (it == m->conv.end() || !it->second.w_split) return false;
    }
    return true;
}

void tap(amp_model* m, const char* name, void* p, int dtype, std::initializer_list<long long> shape) {
    NamedBuf nb;
    nb.ptr = p; nb.dtype = dtype; nb.ndim = (int)shape.size();
    int i = 0;
    for (long long s : shape) nb.shape[i++] = s;
    m->taps[name] = nb;
}

#define AMP_TRY(expr) do { int _s = (expr); if (_s != AMP_OK) return _s; } while (0)
#define AMP_ALLOC(var, T, n)                                                                  \
    T* var = ws.get<T>((size_t)(n));                                                          \
    if (!var) { amp::set_error("amp_model: workspace exhausted allocating %s (%zu bytes, cap %zu)", #var, \
                               (size_t)(n) * sizeof(T), ws.cap); return AMP_ERR_NOMEM; }

// Hand bucket b of the gradient arena to RCCL: everything the backward pass has launched so far is ordered before it, everything
// it launches from here on overlaps it.  No-op without a communicator or with the overlap switched off.
int issue_bucket(amp_model* m, int b) {
    amp_ctx* ctx = m->ctx;
    if (m->ws.dry || !ctx->comm || m->grad_overlap == 0) return AMP_OK;
    AMP_TRY(amp::wgrad_async_join(ctx));      // the bucket's last reductions may still be on the side stream
    // every range of the bucket, in groups of at most 64 per grouped RCCL call (a fragmented arena may hold more than 64)
    size_t off[64], n[64];
    int nr = 0;
    for (size_t i = 0; i < m->gb_bucket.size(); ++i) {
        if (m->gb_bucket[i] != b) continue;
        off[nr] = m->gb_off[i]; n[nr] = m->gb_n[i];
        if (++nr == 64) { AMP_TRY(amp::comm_allreduce_ranges(ctx, m->garena, off, n, nr, b)); nr = 0; }
    }
    AMP_TRY(amp::comm_allreduce_ranges(ctx, m->garena, off, n, nr, b));
    m->issued_mask |= 1u << b;
    return AMP_OK;
}

struct Trunk {
    float* feat[5];
    int fh[5], fw[5];
    amp_rpn_levels lv;
    amp_fpn_feats ff;
    int max_n;
    const int* img_hw;    // device [B][2] per-image sizes (amp_model_set_image_sizes) or null: clip / rescale with these, not the frame
    bool feat_split;      // p2..p6 are in the split row format (the trunk's native activation format in AMP_CONV_F16X3 inference)
};

// Shared by inference and training: preprocess -> ResNet-50 -> FPN -> RPN head. With ws.dry nothing is launched.
int run_trunk(amp_model* m, const uint8_t* imgs_d, int B, int H, int W, Trunk& T) {
    Bump& ws = m->ws;
    const bool dry = ws.dry;
    const amp_model_cfg& c = m->cfg;
    amp_ctx* ctx = m->ctx;
    ws.off = 0;
    if (!dry) m->taps.clear();
    m->blocks.clear();
    const int Hp = (H + 31) / 32 * 32, Wp = (W + 31) / 32 * 32;
    auto CONV = [&](const char* key) -> const ConvW& { return m->conv.at(key); };

    // ---------------- backbone ----------------
    AMP_ALLOC(x0, float, (size_t)B * Hp * Wp * 4);
    AMP_ALLOC(d_img_hw, int, (size_t)2 * B);
    bool x0_split = false, stem_from_u8 = false;
    if (!dry) {
        const bool sized = (int)m->img_hw.size() == 2 * B;
        if (sized) {
            AMP_HIP_CHECK(hipMemcpyAsync(d_img_hw, m->img_hw.data(), (size_t)2 * B * 4, hipMemcpyHostToDevice, ctx->stream));
            AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
        // the fused stem + pool kernel takes its pixels already split (hi | lo' halves): the same halves its own split would make
        const ConvW& sw0 = CONV("backbone.bottom_up.stem.conv1");
        x0_split = sw0.cin == 4 && sw0.kw == 8 && sw0.kh == 7 && sw0.cout == 64 && !m->split_stale && amp::stem_pool_applies(ctx, sw0.w_split) &&
                   (size_t)B * Hp * Wp * 16 < 0x80000000ull && (long long)B * Hp * Wp < (1ll << 27);
        // ... and the uint8 form of that kernel normalises and splits the pixels itself: no preprocessed tensor at all
        stem_from_u8 = x0_split && amp::stem_u8_applies(ctx, sw0.w_split);
        if (!stem_from_u8) AMP_TRY(amp::preprocess_run(ctx, imgs_d, B, H, W, Hp, Wp, c.pixel_mean, c.pixel_std, sized ? d_img_hw : nullptr, x0, x0_split ? 1 : 0));
        T.img_hw = sized ? d_img_hw : nullptr;
    } else {
        T.img_hw = nullptr;
    }
    // AMP_CONV_F16X3 inference: from the max-pool on, every activation of the trunk lives in the split hi|lo' row format (same bytes,
    // same row offsets as fp32): each conv epilogue writes it, the next conv stages it by LDS-DMA without splitting anything, and
    // residual / FPN top-down adds and RoIAlign decode it exactly (hi + lo' * 2^-11).  A conv that cannot read the format (weights
    // beyond the fp16 range) gets an fp32 input from its producer.  Training keeps fp32 activations: the
    // backward kernels read them.
    // Training (m->saving): the saved activations stay in the format too when EVERY trunk conv reads it (R50 / R101) -- the backward
    // kernels take them as they are (wgrad x_split, AMP_FMT_MASK_SPLIT, amp_relu_mask_split) -- otherwise the trunk keeps fp32.
    static const bool train_native = getenv("AMP_NO_TRAIN_NATIVE") == nullptr;      // EXPERIMENT switch
    bool native = !dry && split_chain(m, {}) && (!m->saving || train_native);
    auto reads_split = [&](const std::string& key) {
        if (!native) return false;
        auto it = m->conv.find(key);
        static const bool no_grouped_split = getenv("AMP_NO_GROUPED_SPLIT") != nullptr;      // EXPERIMENT switch: ResNeXt conv2 takes fp32 input
        // (round 4: a training step of a grouped backbone keeps the split format through conv2 too -- grouped_wgrad_kernel decodes split x)
        static const bool no_grouped_train_split = getenv("AMP_NO_GROUPED_TRAIN_SPLIT") != nullptr;      // EXPERIMENT switch: round 3's fp32 trunk for ResNeXt training
        const bool grouped_ok = m->saving ? !no_grouped_train_split : !no_grouped_split;
        return it != m->conv.end() && it->second.w_split != nullptr && (it->second.groups == 1 || grouped_ok) && it->second.cin % 32 == 0;
    };
    bool native_all = native;          // every dense conv of backbone / FPN / RPN can read the format
    if (native)
        for (auto& kv : m->conv) {
            const std::string& k = kv.first;
            const bool trunk = k.rfind("backbone.", 0) == 0 || k.rfind("proposal_generator.", 0) == 0;
            if (trunk && k != "backbone.bottom_up.stem.conv1" && !reads_split(k)) native_all = false;
        }
    if (m->saving && !native_all) native = false;     // (reads_split follows: everything fp32)
    m->acts_split = m->saving && native_all;
    // the split-GRADIENT chain of the backward pass assumes dense 3x3 layers with the block's stride in conv1; a ResNeXt trunk keeps its saved
    // activations split all the same (forward on the ring kernels, weight gradients with split x, FPN / RPN / heads as for R50) and runs
    // the blocks' backward on fp32 gradients
    m->gs_chain_ok = m->acts_split && c.stride_in_1x1;
    for (auto& kv : m->conv) if (kv.second.groups != 1) m->gs_chain_ok = false;
    const int SPL = 5;                 // tap dtype of a split tensor
    auto FMT = [](bool x_split, bool y_split, bool res_split) { return (x_split ? 1 : 0) | (y_split ? 2 : 0) | (res_split ? 4 : 0); };

    int h = Hp / 2, w = Wp / 2;
    AMP_ALLOC(stem, float, (size_t)B * h * w * 64);
    const int h4 = (h + 2 - 3) / 2 + 1, w4 = (w + 2 - 3) / 2 + 1;
    AMP_ALLOC(pool, float, (size_t)B * h4 * w4 * 64);
    if (!dry) {
        // stem + max-pool in one kernel where that exists (f16x3 arithmetic, fresh split weights); the stem's own output is then never written
        const ConvW& sw = CONV("backbone.bottom_up.stem.conv1");
        int fused = 1;
        if (stem_from_u8)
            fused = amp::stem_pool_u8_run(ctx, imgs_d, B, H, W, Hp, Wp, c.pixel_mean, c.pixel_std, T.img_hw, sw.w_split, sw.scale, sw.shift, pool, native_all ? 1 : 0);
        else if (sw.cin == 4 && sw.kw == 8 && sw.kh == 7 && sw.cout == 64 && !m->split_stale)
            fused = amp::stem_pool_run(ctx, B, Hp, Wp, x0, x0_split ? 1 : 0, sw.w_split, sw.scale, sw.shift, pool, native_all ? 1 : 0);
        if (fused < 0) return fused;
        if (fused == 1 && x0_split) { amp::set_error("amp_model: the input was split for the fused stem, which then did not launch"); return AMP_ERR_STATE; }
        if (fused == 1) {
            AMP_TRY(launch_conv(m, sw, x0, B, Hp, Wp, 2, 3, true, 0, nullptr, 0, stem));
            AMP_TRY(amp::maxpool_run(ctx, stem, B, h, w, 64, pool, native_all ? 1 : 0));
        }
    }
    if (!dry) tap(m, "stem_pool", pool, native_all ? SPL : 0, {B, h4, w4, 64});

    float* cur = pool;
    int ch = h4, cw_ = w4;
    float* res_out[4];
    int res_h[4], res_w[4];
    for (int s = 0; s < 4; ++s) {
        for (int b = 0; b < m->nblk[s]; ++b) {
            const std::string p = std::string("backbone.bottom_up.") + kStage[s] + "." + std::to_string(b);
            const int st = (b == 0) ? kStride[s] : 1;
            const int oh = (ch - 1) / st + 1, ow = (cw_ - 1) / st + 1;
            const size_t mark = ws.off;
            // block output first so it survives the release of the temporaries below it... allocate temporaries after
            AMP_ALLOC(out, float, (size_t)B * oh * ow * kOut[s]);
            const size_t keep = ws.off;
            const float* shortcut = cur;
            if (b == 0) {
                AMP_ALLOC(sc, float, (size_t)B * oh * ow * kOut[s]);
                if (!dry) AMP_TRY(launch_conv(m, CONV((p + ".shortcut").c_str()), cur, B, ch, cw_, st, 0, false, 0, nullptr, 0, sc, FMT(native_all, native_all, false)));
                shortcut = sc;
            }
            // RESNETS.STRIDE_IN_1X1: the block's stride sits in conv1 (MSRA R50) or in the 3x3 conv2 (ResNeXt)
            const int st1 = c.stride_in_1x1 ? st : 1, st2 = c.stride_in_1x1 ? 1 : st;
            const int h1 = (ch - 1) / st1 + 1, w1 = (cw_ - 1) / st1 + 1;
            AMP_ALLOC(t1, float, (size_t)B * h1 * w1 * m->mid[s]);
            AMP_ALLOC(t2, float, (size_t)B * oh * ow * m->mid[s]);
            if (!dry) {
                // formats: the block's input / output and the shortcut follow native_all; t1 / t2 are split when their one consumer reads it
                const bool t1s = reads_split(p + ".conv2"), t2s = reads_split(p + ".conv3");
                AMP_TRY(launch_conv(m, CONV((p + ".conv1").c_str()), cur, B, ch, cw_, st1, 0, true, 0, nullptr, 0, t1, FMT(native_all, t1s, false)));
                // res2 (64 mid channels, frozen: the backward pass never reads its t2): conv2 + conv3 in one launch, t2 stays in registers
                int fused23 = 1;
                const ConvW &c2w = CONV((p + ".conv2").c_str()), &c3w = CONV((p + ".conv3").c_str());
                if (s == 0 && t1s && t2s && native_all && st2 == 1 && c2w.groups == 1 && c2w.cin == 64 && c2w.cout == 64 && c2w.kh == 3 && c3w.cin == 64 && c3w.kh == 1 &&
                    !m->split_stale && c2w.scale && c3w.scale)
                    fused23 = amp::conv_c64_fused3_run(ctx, B, oh, ow, t1, c2w.w_split, c2w.scale, c2w.shift, c3w.w_split, c3w.scale, c3w.shift, c3w.cout, shortcut, out);
                if (fused23 < 0) return fused23;
                if (fused23 == 1) {
                    AMP_TRY(launch_conv(m, c2w, t1, B, h1, w1, st2, 1, true, 0, nullptr, 0, t2, FMT(t1s, t2s, false)));
                    AMP_TRY(launch_conv(m, c3w, t2, B, oh, ow, 1, 0, true, 1, shortcut, 0, out, FMT(t2s, native_all, native_all)));
                }
            }
            (void)mark;
            if (m->saving) {
                amp_model::BlockAct ba;
                ba.key = p; ba.x_in = cur; ba.t1 = t1; ba.t2 = t2; ba.sc = (b == 0) ? const_cast<float*>(shortcut) : nullptr; ba.out = out;
                ba.in_h = ch; ba.in_w = cw_; ba.oh = oh; ba.ow = ow; ba.cin = (b == 0) ? (s == 0 ? 64 : kOut[s - 1]) : kOut[s];
                ba.mid = m->mid[s]; ba.cout = kOut[s]; ba.stride = st; ba.stage = s; ba.has_sc = (b == 0);
                if (!dry) m->blocks.push_back(ba);
            } else {
                ws.off = keep;   // release sc/t1/t2 (stream order makes reuse safe)
            }
            cur = out;
            ch = oh; cw_ = ow;
        }
        res_out[s] = cur; res_h[s] = ch; res_w[s] = cw_;
        if (!dry) tap(m, kStage[s], cur, native_all ? SPL : 0, {B, ch, cw_, kOut[s]});
    }

    // ---------------- FPN ----------------
    float* feat[5];
    int fh[5], fw[5];
    const int fstride[5] = {4, 8, 16, 32, 64};
    float* prev_lat = nullptr;
    for (int l = 5; l >= 2; --l) {
        const int s = l - 2;
        const std::string ln = "backbone.fpn_lateral" + std::to_string(l), on = "backbone.fpn_output" + std::to_string(l);
        AMP_ALLOC(lat, float, (size_t)B * res_h[s] * res_w[s] * 256);
        AMP_ALLOC(outp, float, (size_t)B * res_h[s] * res_w[s] * 256);
        if (!dry) {
            // lateral sums and outputs in the native format: the coarser lateral is decoded by the finer level's top-down add
            AMP_TRY(launch_conv(m, CONV(ln.c_str()), res_out[s], B, res_h[s], res_w[s], 1, 0, false, prev_lat ? 2 : 0, prev_lat, 0, lat,
                                FMT(native_all, native_all, native_all && prev_lat != nullptr)));
            AMP_TRY(launch_conv(m, CONV(on.c_str()), lat, B, res_h[s], res_w[s], 1, 1, false, 0, nullptr, 0, outp, FMT(native_all, native_all, false)));
        }
        prev_lat = lat;
        m->lat[s] = lat;
        feat[s] = outp; fh[s] = res_h[s]; fw[s] = res_w[s];
    }
    fh[4] = (fh[3] - 1) / 2 + 1; fw[4] = (fw[3] - 1) / 2 + 1;
    AMP_ALLOC(p6, float, (size_t)B * fh[4] * fw[4] * 256);
    if (!dry) AMP_TRY(amp_subsample2(ctx, feat[3], B, fh[3], fw[3], 256, p6));
    feat[4] = p6;
    if (!dry) {
        const char* fn[5] = {"p2", "p3", "p4", "p5", "p6"};
        for (int l = 0; l < 5; ++l) tap(m, fn[l], feat[l], native_all ? SPL : 0, {B, fh[l], fw[l], 256});
    }

    // ---------------- RPN head ----------------
    const int ld_rpn = 16;   // 3 logits + 12 deltas + 1 zero pad column
    amp_rpn_levels lv;
    memset(&lv, 0, sizeof(lv));
    lv.nlevels = 5; lv.A = 3; lv.ld = ld_rpn;
    const int asz[5] = {32, 64, 128, 256, 512};
    int max_n = 0;
    for (int l = 0; l < 5; ++l) {
        const size_t mark = ws.off;
        AMP_ALLOC(pred, float, (size_t)B * fh[l] * fw[l] * ld_rpn);
        const size_t keep = ws.off;
        AMP_ALLOC(t, float, (size_t)B * fh[l] * fw[l] * 256);
        bool fused_rpn = false;
        if (!dry) {
            const ConvW& rc = CONV("proposal_generator.rpn_head.conv");
            const ConvW& rp = CONV("proposal_generator.rpn_head.pred");
            static const bool no_rpn_fuse = getenv("AMP_NO_RPN_FUSE") != nullptr;      // EXPERIMENT switch
            const long long Ml = (long long)B * fh[l] * fw[l];
            if (native_all && (!m->saving || m->rpn_train_fused) && !m->split_stale && !no_rpn_fuse && rc.cout == 256 && rc.w_split && rp.w_split && rp.cout == 16 && rp.cin == 256 && Ml >= 24576) {
                // inference on the native trunk -- and a training step whose backward pass takes the sparse route (it recomputes the hidden rows it needs):
                // the predictors run in the 3x3 conv's epilogue, the hidden tensor is never written
                fused_rpn = true;
                amp_conv_desc d;
                d.B = B; d.H = fh[l]; d.W = fw[l]; d.Cin = rc.cin; d.Cout = rc.cout; d.KH = rc.kh; d.KW = rc.kw; d.stride = 1; d.pad = 1; d.relu = 1; d.res_mode = 0; d.out_mode = 0;
                amp::RpnFuse rf{rp.w_split, rp.shift, pred};
                AMP_TRY(amp::conv_run(ctx, &d, 1, feat[l], rc.w, rc.w_split, 0, rc.scale, rc.shift, nullptr, nullptr, t, 0, 1, nullptr, &rf));
            } else {
                AMP_TRY(launch_conv(m, rc, feat[l], B, fh[l], fw[l], 1, 1, true, 0, nullptr, 0, t, FMT(native_all, native_all, false)));
                AMP_TRY(launch_conv(m, rp, t, B, fh[l], fw[l], 1, 0, false, 0, nullptr, 0, pred, FMT(native_all, false, false)));
            }
        }
        (void)mark;
        if (m->saving && !fused_rpn) m->rpn_t[l] = t;
        else { if (m->saving) m->rpn_t[l] = nullptr; if (!m->saving) ws.off = keep; }
        lv.pred[l] = pred; lv.h[l] = fh[l]; lv.w[l] = fw[l]; lv.stride[l] = fstride[l]; lv.anchor_size[l] = asz[l];
        max_n = std::max(max_n, fh[l] * fw[l] * 3);
        if (!dry) {
            const char* pn[5] = {"rpn_pred2", "rpn_pred3", "rpn_pred4", "rpn_pred5", "rpn_pred6"};
            tap(m, pn[l], pred, 0, {B, fh[l] * fw[l], ld_rpn});
        }
    }

    for (int l = 0; l < 5; ++l) { T.feat[l] = feat[l]; T.fh[l] = fh[l]; T.fw[l] = fw[l]; }
    T.lv = lv;
    T.max_n = max_n;
    T.feat_split = native_all;
    memset(&T.ff, 0, sizeof(T.ff));
    T.ff.C = 256;
    for (int l = 0; l < 4; ++l) { T.ff.feat[l] = feat[l]; T.ff.h[l] = fh[l]; T.ff.w[l] = fw[l]; T.ff.stride[l] = fstride[l]; }
    return AMP_OK;
}

struct Proposals {
    float* boxes;    // [B][Rcap][4]
    float* logits;   // [B][Rcap]
    int* count;      // [B]
    int* anchor;     // [B][Rcap] global anchor index each proposal was decoded from
    int Rcap;
};

// find_top_rpn_proposals: per-level top-k, decode, sort, NMS, first post_nms_topk.
int run_proposals(amp_model* m, const Trunk& T, int B, int H, int W, int k, int Rcap, Proposals& P) {
    Bump& ws = m->ws;
    const bool dry = ws.dry;
    const amp_model_cfg& c = m->cfg;
    amp_ctx* ctx = m->ctx;
    const amp_rpn_levels& lv = T.lv;
    const int max_n = T.max_n;
    const int cap = 5 * k;                 // candidates per image
    AMP_ALLOC(keys_scratch, uint32_t, (size_t)B * 5 * max_n);
    AMP_ALLOC(sel_idx, int, (size_t)B * 5 * k);
    AMP_ALLOC(sel_logit, float, (size_t)B * 5 * k);
    AMP_ALLOC(sel_count, int, (size_t)B * 5);
    AMP_ALLOC(cand_boxes, float, (size_t)B * cap * 4);
    AMP_ALLOC(cand_keys, unsigned long long, (size_t)B * cap);
    AMP_ALLOC(prop_count, int, (size_t)B);
    AMP_ALLOC(prop_boxes, float, (size_t)B * Rcap * 4);
    AMP_ALLOC(prop_logits, float, (size_t)B * Rcap);
    AMP_ALLOC(prop_lvl, int, (size_t)B * Rcap);
    AMP_ALLOC(cand_anchor, int, (size_t)B * cap);
    AMP_ALLOC(prop_anchor, int, (size_t)B * Rcap);
    // AMP_RPN_NMS=flat: the one-list path of rounds 1-3 (sort all levels' candidates, one NMS per image) for A/B runs; same proposals
    static const bool flat = [] { const char* e = getenv("AMP_RPN_NMS"); return e && !strcmp(e, "flat"); }();
    if (flat) {
        AMP_ALLOC(s_boxes, float, (size_t)B * cap * 4);
        AMP_ALLOC(s_scores, float, (size_t)B * cap);
        AMP_ALLOC(s_cats, int, (size_t)B * cap);
        AMP_ALLOC(s_count, int, (size_t)B);
        AMP_ALLOC(nms_mask, unsigned long long, (size_t)B * cap * ((cap + 63) / 64));
        AMP_ALLOC(keep_idx, int, (size_t)B * Rcap);
        AMP_ALLOC(s_anchor, int, (size_t)B * cap);
        if (!dry) {
            AMP_TRY(amp_rpn_topk(ctx, &lv, B, k, keys_scratch, max_n, sel_idx, sel_logit, sel_count));
            AMP_TRY(amp_rpn_decode_sized(ctx, &lv, B, k, sel_idx, sel_logit, sel_count, H, W, T.img_hw, cap, cand_boxes, cand_keys, cand_anchor));
            AMP_TRY(amp_sort_gather(ctx, B, cap, cap, cand_keys, cand_boxes, s_boxes, s_scores, s_cats, s_count, nullptr, cand_anchor, s_anchor));
            AMP_TRY(amp_nms(ctx, B, cap, s_boxes, s_cats, s_count, c.rpn_nms_thresh, Rcap, nms_mask, keep_idx, prop_count));
            AMP_TRY(amp_gather_dets(ctx, B, cap, Rcap, s_boxes, s_scores, s_cats, keep_idx, prop_count, prop_boxes, prop_logits, prop_lvl, s_anchor, prop_anchor));
        }
    } else {
        // each level's top-k list is its own NMS problem (the level is batched_nms's category); the survivors' lists are merged (nms.hip)
        AMP_ALLOC(lvl_scratch, unsigned long long, amp_rpn_nms_scratch_words(B, 5, k));
        if (!dry) {
            AMP_TRY(amp_rpn_topk(ctx, &lv, B, k, keys_scratch, max_n, sel_idx, sel_logit, sel_count));
            AMP_TRY(amp_rpn_decode_sized(ctx, &lv, B, k, sel_idx, sel_logit, sel_count, H, W, T.img_hw, cap, cand_boxes, cand_keys, cand_anchor));
            AMP_TRY(amp_rpn_nms_levels(ctx, B, 5, k, cap, cand_boxes, cand_keys, sel_count, c.rpn_nms_thresh, Rcap, lvl_scratch, prop_boxes,
                                       prop_logits, prop_lvl, prop_count, cand_anchor, prop_anchor));
        }
    }
    if (!dry) {
        tap(m, "rpn_sel_idx", sel_idx, 1, {B, 5, k});
        tap(m, "rpn_sel_logit", sel_logit, 0, {B, 5, k});
        tap(m, "rpn_cand_boxes", cand_boxes, 0, {B, cap, 4});
        tap(m, "prop_boxes", prop_boxes, 0, {B, Rcap, 4});
        tap(m, "prop_logits", prop_logits, 0, {B, Rcap});
        tap(m, "prop_count", prop_count, 1, {B});
    }
    P.boxes = prop_boxes; P.logits = prop_logits; P.count = prop_count; P.anchor = prop_anchor; P.Rcap = Rcap;
    return AMP_OK;
}

// The whole inference forward. With ws.dry == true nothing is launched: only the workspace peak is measured.
int run(amp_model* m, const uint8_t* imgs_d, int B, int H, int W, const int* out_h, const int* out_w) {
    Bump& ws = m->ws;
    const bool dry = ws.dry;
    const amp_model_cfg& c = m->cfg;
    amp_ctx* ctx = m->ctx;
    const int K = c.num_classes;
    auto CONV = [&](const char* key) -> const ConvW& { return m->conv.at(key); };
    Trunk T;
    AMP_TRY(run_trunk(m, imgs_d, B, H, W, T));
    Proposals PR;
    AMP_TRY(run_proposals(m, T, B, H, W, c.pre_nms_topk, c.post_nms_topk, PR));
    float* prop_boxes = PR.boxes;
    int* prop_count = PR.count;
    const int Rcap = PR.Rcap;
    const amp_fpn_feats& ff = T.ff;

    // ---------------- box head ----------------
    const int R = B * Rcap;
    const int ld_box = (5 * K + 1 + 3) / 4 * 4;
    AMP_ALLOC(pooled, float, (size_t)R * 49 * 256);
    AMP_ALLOC(fc1, float, (size_t)R * 1024);
    AMP_ALLOC(fc2, float, (size_t)R * 1024);
    AMP_ALLOC(box_pred, float, (size_t)R * ld_box);
    const int ccap = 8192;
    const int D = c.detections_per_image;
    AMP_ALLOC(dense_boxes, float, (size_t)R * K * 4);
    AMP_ALLOC(bkeys, unsigned long long, (size_t)B * ccap);
    AMP_ALLOC(bcount, int, (size_t)B);
    AMP_ALLOC(bs_boxes, float, (size_t)B * ccap * 4);
    AMP_ALLOC(bs_scores, float, (size_t)B * ccap);
    AMP_ALLOC(bs_cats, int, (size_t)B * ccap);
    AMP_ALLOC(bs_count, int, (size_t)B);
    AMP_ALLOC(bnms_mask, unsigned long long, (size_t)B * ccap * (ccap / 64));
    AMP_ALLOC(bkeep_idx, int, (size_t)B * D);
    AMP_ALLOC(det_count, int, (size_t)B);
    AMP_ALLOC(det_boxes, float, (size_t)B * D * 4);
    AMP_ALLOC(det_scores, float, (size_t)B * D);
    AMP_ALLOC(det_classes, int, (size_t)B * D);
    AMP_ALLOC(d_floor, float, (size_t)B);          // per-image score floors (only used when an image has more than ccap candidates)
    const int Ncap = std::max(B * D, 1);
    // compact detection list over the batch (allocated for B * D detections: its first kernels are queued before the host knows the count).  Everything the host reads back at the end sits in ONE contiguous block of the
    // workspace ([res0, res1): scores, classes, rescaled boxes, validity, run offsets / lengths, pool fill) and comes back with one
    // copy into pinned memory: eight pageable copies cost 0.3 ms of idle GPU per step.
    AMP_ALLOC(m_boxes, float, (size_t)Ncap * 4);
    AMP_ALLOC(m_batch, int, (size_t)Ncap);
    AMP_ALLOC(d_out_hw, int, (size_t)2 * B);
    AMP_ALLOC(mpooled, float, (size_t)Ncap * 196 * 256);
    AMP_ALLOC(mt_a, float, (size_t)Ncap * 196 * 256);
    AMP_ALLOC(mt_b, float, (size_t)Ncap * 784 * 256);
    const int Kp = (K + 3) / 4 * 4;
    AMP_ALLOC(mlogits, float, (size_t)Ncap * 784 * Kp);
    AMP_ALLOC(mprob, float, (size_t)Ncap * 784);
    const size_t res0 = (ws.off + 255) & ~(size_t)255;
    AMP_ALLOC(m_scores, float, (size_t)Ncap);
    AMP_ALLOC(m_classes, int, (size_t)Ncap);
    AMP_ALLOC(o_boxes, float, (size_t)Ncap * 4);
    AMP_ALLOC(o_valid, int, (size_t)Ncap);
    AMP_ALLOC(o_off, unsigned long long, (size_t)Ncap);
    AMP_ALLOC(o_len, int, (size_t)Ncap);
    AMP_ALLOC(pool_used, unsigned long long, 2);      // [0] run lengths (read back), [1] position scratch
    AMP_ALLOC(o_soff, unsigned long long, (size_t)Ncap);   // counts strings (rle_mode != 0): offset / length per detection, bytes in all
    AMP_ALLOC(o_slen, int, (size_t)Ncap);
    AMP_ALLOC(str_total, unsigned long long, 1);
    const size_t res1 = ws.off;
    AMP_ALLOC(rle_pool, unsigned int, (size_t)c.rle_pool_counts);
    AMP_ALLOC(pos_pool, unsigned int, (size_t)c.rle_pool_counts);
    const unsigned long long str_cap = (unsigned long long)c.rle_pool_counts * 4;      // bytes: a run needs 1-7 characters, 1.3 on average
    AMP_ALLOC(str_pool, char, (size_t)str_cap);
    AMP_ALLOC(d_ntotal, int, 1);
    bool mask_queued = false;     // compact list + mask RoIAlign already on the stream
    bool mchain = false;
    if (!dry) {
        const bool bchain = split_chain(m, {"roi_heads.box_head.fc1", "roi_heads.box_head.fc2"});
        AMP_TRY(amp::roi_align_run(ctx, &ff, prop_boxes, m->d_batch_iota, nullptr, R, 7, pooled, nullptr, bchain ? 1 : 0, T.feat_split ? 1 : 0));
        AMP_TRY(launch_conv(m, CONV("roi_heads.box_head.fc1"), pooled, 1, 1, R, 1, 0, true, 0, nullptr, 0, fc1, bchain ? 3 : 0));
        AMP_TRY(launch_conv(m, CONV("roi_heads.box_head.fc2"), fc1, 1, 1, R, 1, 0, true, 0, nullptr, 0, fc2, bchain ? 1 : 0));
        AMP_TRY(launch_conv(m, CONV("roi_heads.box_predictor"), fc2, 1, 1, R, 1, 0, false, 0, nullptr, 0, box_pred));
        auto detect = [&](const float* thresh_img) -> int {      // candidates above the score floor -> order -> per-class NMS -> first D
            AMP_TRY(amp::box_candidates_run(ctx, box_pred, ld_box, prop_boxes, prop_count, B, Rcap, K, c.bbox_reg_weights, c.score_thresh,
                                            H, W, T.img_hw, thresh_img, dense_boxes, bkeys, ccap, bcount, m->d_flags + 0));
            AMP_TRY(amp_sort_gather_n(ctx, B, ccap, Rcap * K, bkeys, dense_boxes, bs_boxes, bs_scores, bs_cats, bs_count, nullptr, nullptr, nullptr, bcount));
            AMP_TRY(amp_nms(ctx, B, ccap, bs_boxes, bs_cats, bs_count, c.nms_thresh, D, bnms_mask, bkeep_idx, det_count));
            return amp_gather_dets(ctx, B, ccap, D, bs_boxes, bs_scores, bs_cats, bkeep_idx, det_count, det_boxes, det_scores, det_classes, nullptr, nullptr);
        };
        AMP_TRY(detect(nullptr));
        tap(m, "box_pooled", pooled, bchain ? 5 : 0, {R, 7, 7, 256});
        tap(m, "box_pred", box_pred, 0, {R, ld_box});
        tap(m, "box_dense", dense_boxes, 0, {B, Rcap * K, 4});
        tap(m, "box_sorted", bs_boxes, 0, {B, ccap, 4});
        tap(m, "det_boxes", det_boxes, 0, {B, D, 4});
        tap(m, "det_scores", det_scores, 0, {B, D});
        tap(m, "det_classes", det_classes, 1, {B, D});
        tap(m, "det_count", det_count, 1, {B});
        // ---------------- sync point: detection counts decide the mask-branch GEMM sizes ----------------
        AMP_HIP_CHECK(hipMemcpyAsync(m->h_counts, det_count, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        AMP_HIP_CHECK(hipMemcpyAsync(m->h_counts + B, m->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        // The host needs the counts for the GEMM sizes of the mask head, the GPU does not need the host for what comes first: the compact
        // detection list and the mask pooler read the counts on the device and are queued BEFORE the wait, so the ~50 us round trip (copy
        // lands -> host wakes -> next launches arrive) passes under 0.3 ms of RoIAlign instead of an idle GPU.
        if (!m->ev_counts) AMP_HIP_CHECK(hipEventCreateWithFlags(&m->ev_counts, hipEventDisableTiming));
        AMP_HIP_CHECK(hipEventRecord(m->ev_counts, ctx->stream));
        AMP_REQUIRE(res1 - res0 <= m->h_res_bytes, "amp_model_infer: result staging too small");
        mchain = split_chain(m, {"roi_heads.mask_head.mask_fcn1", "roi_heads.mask_head.mask_fcn2", "roi_heads.mask_head.mask_fcn3",
                                 "roi_heads.mask_head.mask_fcn4", "roi_heads.mask_head.deconv"});
        auto queue_mask_inputs = [&]() -> int {
            AMP_TRY(amp::compact_dets_run(ctx, B, D, det_count, det_boxes, det_scores, det_classes, m_boxes, m_scores, m_classes, m_batch, d_ntotal));
            return amp::roi_align_run(ctx, &ff, m_boxes, m_batch, d_ntotal, Ncap, 14, mpooled, nullptr, mchain ? 1 : 0, T.feat_split ? 1 : 0);
        };
        AMP_TRY(queue_mask_inputs());
        mask_queued = true;
        AMP_HIP_CHECK(hipEventSynchronize(m->ev_counts));
        if (m->h_counts[B + 0]) {
            mask_queued = false;      // the detections are about to be recomputed: the compact list and the pooled features with them
            // An image has more than ccap candidates above SCORE_THRESH_TEST (many classes, a low threshold).  Greedy NMS takes its
            // decisions in score order, so the detections are decided by a PREFIX of the candidate list: raise that image's score floor
            // (bisection on the candidate count) until the prefix fits, run the chain again, and the result is the exact one as long as
            // the prefix still yields D detections.  Only if it does not -- more than ccap candidates and fewer than D of the best
            // ccap survive NMS -- is the batch refused.
            std::vector<float> lo(B, c.score_thresh), hi(B, 1.0f), cur(B, c.score_thresh);   // lo: overflows, hi: fits (a floor of 1 keeps nothing)
            std::vector<int> cnt(B);
            std::vector<char> raised(B, 0), done(B, 0);
            for (int it = 0; it < 30; ++it) {
                AMP_HIP_CHECK(hipMemcpyAsync(d_floor, cur.data(), (size_t)B * 4, hipMemcpyHostToDevice, ctx->stream));
                AMP_HIP_CHECK(hipMemsetAsync(m->d_flags, 0, sizeof(int), ctx->stream));
                AMP_TRY(amp::box_candidates_run(ctx, box_pred, ld_box, prop_boxes, prop_count, B, Rcap, K, c.bbox_reg_weights, c.score_thresh,
                                                H, W, T.img_hw, d_floor, dense_boxes, bkeys, ccap, bcount, m->d_flags + 0));
                AMP_HIP_CHECK(hipMemcpyAsync(cnt.data(), bcount, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
                AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                bool again = false;
                for (int b = 0; b < B; ++b) {
                    if (done[b]) continue;
                    if (cnt[b] > ccap) { lo[b] = cur[b]; raised[b] = 1; }
                    else if (!raised[b]) { done[b] = 1; continue; }                 // never overflowed: its floor stays SCORE_THRESH_TEST
                    else { hi[b] = cur[b]; if (cnt[b] >= ccap * 3 / 4 || hi[b] - lo[b] < 1e-7f) { done[b] = 1; continue; } }
                    cur[b] = 0.5f * (lo[b] + hi[b]);
                    again = true;
                }
                if (!again) break;
                if (it == 29) for (int b = 0; b < B; ++b) if (!done[b]) cur[b] = hi[b];     // the last floor known to fit
            }
            AMP_HIP_CHECK(hipMemcpyAsync(d_floor, cur.data(), (size_t)B * 4, hipMemcpyHostToDevice, ctx->stream));
            AMP_HIP_CHECK(hipMemsetAsync(m->d_flags, 0, sizeof(int), ctx->stream));
            AMP_TRY(detect(d_floor));
            AMP_HIP_CHECK(hipMemcpyAsync(m->h_counts, det_count, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            AMP_HIP_CHECK(hipMemcpyAsync(m->h_counts + B, m->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            for (int b = 0; b < B; ++b)
                if (raised[b] && m->h_counts[b] < D) {
                    amp::set_error("amp_model_infer: image %d has more than %d box candidates above SCORE_THRESH_TEST and fewer than %d of the best %d survive NMS: "
                                   "the result would depend on candidates that do not fit (raise SCORE_THRESH_TEST)", b, ccap, D, ccap);
                    return AMP_ERR_NOMEM;
                }
            if (m->h_counts[B + 0]) { amp::set_error("amp_model_infer: box candidates overflowed after the score floor was raised"); return AMP_ERR_STATE; }
            AMP_TRY(queue_mask_inputs());
            mask_queued = true;
        }
    }

    int N = B * D;   // dry run: worst case
    std::vector<int> off(B + 1, 0);
    if (!dry) {
        for (int b = 0; b < B; ++b) off[b + 1] = off[b] + std::min(m->h_counts[b], D);
        N = off[B];
    }

    if (dry) return AMP_OK;

    m->r_out_h.assign(out_h, out_h + B);
    m->r_out_w.assign(out_w, out_w + B);
    int max_hw = 1;
    for (int b = 0; b < B; ++b) max_hw = std::max(max_hw, std::max(out_h[b], out_w[b]));
    AMP_REQUIRE(max_hw <= c.max_out_hw, "amp_model_infer: output size %d exceeds cfg.max_out_hw=%d", max_hw, c.max_out_hw);
    for (int b = 0; b < B; ++b) { m->h_small[b] = out_h[b]; m->h_small[B + b] = out_w[b]; }
    AMP_HIP_CHECK(hipMemcpyAsync(d_out_hw, m->h_small, (size_t)2 * B * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    AMP_HIP_CHECK(hipMemsetAsync(pool_used, 0, 2 * sizeof(unsigned long long), ctx->stream));
    if (N > 0) {
        // ---------------- mask head (the compact list and the pooled features are on the stream already) ----------------
        AMP_REQUIRE(mask_queued, "amp_model_infer: internal: the mask branch's inputs were not queued");
        const int io = mchain ? 3 : 0;      // split in, split out
        AMP_TRY(launch_conv(m, CONV("roi_heads.mask_head.mask_fcn1"), mpooled, N, 14, 14, 1, 1, true, 0, nullptr, 0, mt_a, io));
        AMP_TRY(launch_conv(m, CONV("roi_heads.mask_head.mask_fcn2"), mt_a, N, 14, 14, 1, 1, true, 0, nullptr, 0, mpooled, io));
        AMP_TRY(launch_conv(m, CONV("roi_heads.mask_head.mask_fcn3"), mpooled, N, 14, 14, 1, 1, true, 0, nullptr, 0, mt_a, io));
        AMP_TRY(launch_conv(m, CONV("roi_heads.mask_head.mask_fcn4"), mt_a, N, 14, 14, 1, 1, true, 0, nullptr, 0, mpooled, io));
        static const bool no_fuse = getenv("AMP_NO_MASK_FUSE") != nullptr;
        if (mchain && !no_fuse) {
            // deconv + ReLU + the predictor row of each detection's class + sigmoid in ONE kernel: the [N,28,28,256] activation
            // (1.28 GB at 1600 detections) is never written
            const ConvW& cd = CONV("roi_heads.mask_head.deconv");
            const ConvW& cp = CONV("roi_heads.mask_head.predictor");
            amp_conv_desc d;
            d.B = N; d.H = 14; d.W = 14; d.Cin = 256; d.Cout = 1024; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0; d.relu = 1; d.res_mode = 0; d.out_mode = 1;
            amp::PredictFuse pf{cp.w, cp.shift, m_classes, K, mprob};
            AMP_TRY(amp::conv_run(ctx, &d, 1, mpooled, cd.w, cd.w_split, 0, nullptr, cd.shift, nullptr, nullptr, mprob, 0, 1, &pf));
        } else {
            AMP_TRY(launch_conv(m, CONV("roi_heads.mask_head.deconv"), mpooled, N, 14, 14, 1, 0, true, 0, nullptr, 1, mt_b, mchain ? 1 : 0));
            AMP_TRY(launch_conv(m, CONV("roi_heads.mask_head.predictor"), mt_b, N, 28, 28, 1, 0, false, 0, nullptr, 0, mlogits));
            AMP_TRY(amp_mask_prob(ctx, mlogits, m_classes, N, Kp, mprob));
        }
        AMP_TRY(amp_paste_rle_sized(ctx, mprob, m_boxes, m_batch, N, d_out_hw, d_out_hw + B, max_hw, H, W, T.img_hw, c.mask_threshold, o_boxes,
                              o_valid, rle_pool, (unsigned long long)c.rle_pool_counts, pool_used, o_off, o_len, m->d_flags + 1,
                              pos_pool, (unsigned long long)c.rle_pool_counts, pool_used + 1));
        if (m->rle_mode) AMP_TRY(amp::rle_strings_run(ctx, rle_pool, o_off, o_len, N, str_pool, str_cap, o_soff, o_slen, str_total));
        tap(m, "mask_prob", mprob, 0, {N, 28, 28});
        tap(m, "mask_rois", m_boxes, 0, {N, 4});
    }

    // ---------------- results to the host ----------------
    const char* hres = m->h_res;
    auto host_of = [&](const void* dev) { return hres + ((const char*)dev - (ws.base + res0)); };
    const float* hb = nullptr; const float* hs = nullptr; const int* hv = nullptr; const int* hc = nullptr; const int* hl = nullptr;
    const unsigned long long* ho = nullptr;
    const unsigned long long* hso = nullptr; const int* hsl = nullptr;
    bool str_tail_done = false;
    if (N > 0) {
        AMP_HIP_CHECK(hipMemcpyAsync(m->h_res, ws.base + res0, res1 - res0, hipMemcpyDeviceToHost, ctx->stream));
        AMP_HIP_CHECK(hipMemcpyAsync(m->h_counts + B, m->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        // The strings' size is only known after this block has arrived, and a second copy behind a second synchronisation is 30 us of
        // idle GPU: copy what the previous call needed plus a margin along with the block, and only a longer tail afterwards.
        size_t str_guess = 0;
        if (m->rle_mode && m->str_bytes_last) {
            str_guess = std::min<size_t>({(size_t)str_cap, m->h_str_bytes, m->str_bytes_last + m->str_bytes_last / 8 + 4096});
            AMP_HIP_CHECK(hipMemcpyAsync(m->h_str, str_pool, str_guess, hipMemcpyDeviceToHost, ctx->stream));
        }
        AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (m->h_counts[B + 1]) { amp::set_error("amp_model_infer: RLE pool (%zu counts) exhausted; raise cfg.rle_pool_counts", (size_t)c.rle_pool_counts); return AMP_ERR_NOMEM; }
        hb = (const float*)host_of(o_boxes); hs = (const float*)host_of(m_scores); hc = (const int*)host_of(m_classes);
        hv = (const int*)host_of(o_valid); hl = (const int*)host_of(o_len); ho = (const unsigned long long*)host_of(o_off);
        hso = (const unsigned long long*)host_of(o_soff); hsl = (const int*)host_of(o_slen);
        const unsigned long long used = *(const unsigned long long*)host_of(pool_used);
        const unsigned long long sbytes = m->rle_mode ? *(const unsigned long long*)host_of(str_total) : 0ull;
        if (m->rle_mode) {
            if (sbytes > str_cap || sbytes > m->h_str_bytes) {
                amp::set_error("amp_model_infer: the counts strings need %llu bytes, the string pool holds %llu; raise cfg.rle_pool_counts", sbytes,
                               std::min<unsigned long long>(str_cap, m->h_str_bytes));
                return AMP_ERR_NOMEM;
            }
            m->str_bytes_last = (size_t)sbytes;
            if (sbytes > str_guess) AMP_HIP_CHECK(hipMemcpyAsync(m->h_str + str_guess, str_pool + str_guess, (size_t)sbytes - str_guess, hipMemcpyDeviceToHost, ctx->stream));
            else str_tail_done = true;
        }
        // the pool holds run lengths only (positions live in their own scratch pool): the used prefix is exactly what the caller gets
        if (m->rle_mode == 1) {
            if (!str_tail_done) AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));      // strings only: the run lengths stay on the device
            m->r_pool_ptr = nullptr;
        } else if (m->h_pool && used <= m->h_pool_counts) {
            // DMA into host-cacheable pinned memory and hand that out: a pageable destination made the runtime stage the bytes and
            // the CPU copy them once more while the GPU sat idle (0.3 ms per step for 2.5 MB)
            if (used) AMP_HIP_CHECK(hipMemcpyAsync(m->h_pool, rle_pool, (size_t)used * 4, hipMemcpyDeviceToHost, ctx->stream));
            AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            m->r_pool_ptr = m->h_pool;
        } else {
        m->r_pool.resize((size_t)used);
        // (straight into the pageable vector: reading 2 MB back out of pinned, CPU-uncached staging cost 0.8 ms per step)
        if (used) AMP_HIP_CHECK(hipMemcpy(m->r_pool.data(), rle_pool, (size_t)used * 4, hipMemcpyDeviceToHost));
        m->r_pool_ptr = m->r_pool.data();
        }
    } else {
        m->r_pool.clear();
        m->r_pool_ptr = m->r_pool.data();
    }
    m->r_n.assign(B, 0);
    m->r_boxes.assign((size_t)B * D * 4, 0.f);
    m->r_scores.assign((size_t)B * D, 0.f);
    m->r_classes.assign((size_t)B * D, -1);
    m->r_rle_off.assign((size_t)B * D, 0ull);
    m->r_rle_len.assign((size_t)B * D, 0);
    m->r_str_off.assign((size_t)B * D, 0ull);
    m->r_str_len.assign((size_t)B * D, 0);
    for (int b = 0; b < B; ++b) {
        int n = 0;
        for (int i = off[b]; i < off[b + 1]; ++i) {
            if (!hv[i]) continue;   // detector_postprocess drops boxes that are empty after rescale + clip
            const size_t o = (size_t)b * D + n;
            memcpy(&m->r_boxes[o * 4], &hb[(size_t)i * 4], 16);
            m->r_scores[o] = hs[i];
            m->r_classes[o] = hc[i];
            m->r_rle_off[o] = ho[i];
            m->r_rle_len[o] = hl[i];
            if (m->rle_mode) { m->r_str_off[o] = hso[i]; m->r_str_len[o] = hsl[i]; }
            ++n;
        }
        m->r_n[b] = n;
    }
    return AMP_OK;
}

int refresh_split_weights(amp_model* m);
int refresh_dgrad_weights(amp_model* m);

// Training-mode forward + losses. With ws.dry nothing is launched (workspace sizing only).
int run_train(amp_model* m, const uint8_t* imgs_d, int B, int H, int W, const amp_gt* gt, unsigned int seed, float losses[5], bool backward) {
    Bump& ws = m->ws;
    const bool dry = ws.dry;
    const amp_model_cfg& c = m->cfg;
    amp_ctx* ctx = m->ctx;
    const int K = c.num_classes;
    auto CONV = [&](const char* key) -> const ConvW& { return m->conv.at(key); };
    Trunk T;
    struct ModeGuard { amp_ctx* c; int mode; ~ModeGuard() { c->conv_mode = mode; } } mode_guard{m->ctx, m->ctx->conv_mode};
    m->saving = backward;
    // the forward trunk of a training step runs on pre-split operands like inference: refresh the weights' split copies (stale since the
    // last SGD step) once here instead of splitting every layer's weights inside its own launch
    if (!dry && backward && m->split_stale && m->ctx->conv_mode == AMP_CONV_F16X3 && getenv("AMP_NO_TRAIN_NATIVE") == nullptr) AMP_TRY(refresh_split_weights(m));
    static const bool no_dgrad_batch = getenv("AMP_NO_DGRAD_BATCH") != nullptr;      // EXPERIMENT switch: transpose + split in front of every data-gradient conv
    const bool dgrad_batch = !dry && backward && !no_dgrad_batch && m->ctx->conv_mode == AMP_CONV_F16X3;
    if (dgrad_batch) AMP_TRY(refresh_dgrad_weights(m));
    // the RPN head's backward pass runs over the sampled anchors' pixels only (rpn_sparse.hip) -- and then the forward pass need not save the head's hidden
    // tensor: the predictors run in the conv's epilogue as in inference, the backward pass recomputes the rows it needs from the gathered patches
    bool sr_ok = false;
    if (!dry) {      // (the plan run comes before the parameters: no layer table yet)
        static const bool no_rpn_sparse = getenv("AMP_NO_RPN_SPARSE") != nullptr;          // EXPERIMENT switch: the dense backward of the RPN head
        static const bool no_rpn_train_fuse = getenv("AMP_NO_RPN_TRAIN_FUSE") != nullptr;  // EXPERIMENT switch: the head's hidden tensor saved by the forward pass
        const ConvW& cpred = CONV("proposal_generator.rpn_head.pred");
        const ConvW& cconv = CONV("proposal_generator.rpn_head.conv");
        sr_ok = backward && (g_rpn_sparse < 0 ? !no_rpn_sparse : g_rpn_sparse != 0) && cpred.cout <= 16 && cpred.cin == 256 && cpred.scale == nullptr && c.rpn_batch <= 512 &&
                cconv.cout == 256 && cconv.cin == 256 && cconv.kh == 3 && cconv.kw == 3;
        m->rpn_train_fused = !dry && sr_ok && (g_rpn_train_fuse < 0 ? !no_rpn_train_fuse : g_rpn_train_fuse != 0) && m->ctx->conv_mode == AMP_CONV_F16X3;
    }
    const int trunk_status = run_trunk(m, imgs_d, B, H, W, T);
    m->saving = false;
    const bool rpn_fused = m->rpn_train_fused;
    m->rpn_train_fused = false;
    AMP_TRY(trunk_status);
    const int total_gt = dry ? c.max_gt : gt->gt_off[B];
    const int npoly = dry ? c.max_poly_doubles : gt->poly_off[gt->inst_poly_off ? gt->inst_poly_off[total_gt] : total_gt];
    AMP_REQUIRE(total_gt <= c.max_gt && npoly <= c.max_poly_doubles, "amp_model_forward_losses: %d instances / %d polygon doubles exceed cfg.max_gt / max_poly_doubles", total_gt, npoly);
    int A = 0;
    for (int l = 0; l < 5; ++l) A += T.fh[l] * T.fw[l] * 3;

    AMP_ALLOC(d_gt_boxes, float, (size_t)std::max(total_gt, 1) * 4);
    AMP_ALLOC(d_gt_cls, int, (size_t)std::max(total_gt, 1));
    AMP_ALLOC(d_gt_off, int, (size_t)B + 1);
    // polygons: one per instance (gt->inst_poly_off == NULL) or a range of polygons per instance; bitmask instances carry run lengths
    const int n_polys = dry ? c.max_gt : (gt->inst_poly_off ? gt->inst_poly_off[total_gt] : total_gt);
    const size_t n_runs = (dry || !gt->rle_off) ? 0 : (size_t)gt->rle_off[total_gt];
    AMP_REQUIRE(dry || n_polys <= c.max_gt, "amp_model_forward_losses: %d polygons exceed cfg.max_gt", n_polys);
    AMP_ALLOC(d_poly_off, int, (size_t)std::max(n_polys, total_gt) + 1);
    AMP_ALLOC(d_inst_poly_off, int, (size_t)total_gt + 1);
    AMP_ALLOC(d_poly_xy, double, (size_t)std::max(npoly, 1));
    AMP_ALLOC(d_rle_off, unsigned long long, (size_t)total_gt + 1);
    AMP_ALLOC(d_rle_hw, int, (size_t)std::max(total_gt, 1) * 2);
    AMP_ALLOC(match_val, float, (size_t)B * A);
    AMP_ALLOC(match_idx, int, (size_t)B * A);
    AMP_ALLOC(gt_best, unsigned int, (size_t)std::max(total_gt, 1));
    AMP_ALLOC(label, signed char, (size_t)B * A);
    AMP_ALLOC(keys, uint32_t, (size_t)B * A);
    AMP_ALLOC(rpn_sampled, int, (size_t)B * c.rpn_batch);
    AMP_ALLOC(rpn_counts, int, (size_t)B * 2);
    AMP_ALLOC(rpn_partial, float, (size_t)B * 2);
    float* d_rpn_pred[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (backward) {
        for (int l = 0; l < 5; ++l) {
            AMP_ALLOC(dp, float, (size_t)B * T.fh[l] * T.fw[l] * T.lv.ld);
            d_rpn_pred[l] = dp;
            if (!dry) AMP_HIP_CHECK(hipMemsetAsync(dp, 0, (size_t)B * T.fh[l] * T.fw[l] * T.lv.ld * 4, ctx->stream));
        }
    }
    if (!dry) {
        AMP_HIP_CHECK(hipMemcpyAsync(d_gt_off, gt->gt_off, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        AMP_HIP_CHECK(hipMemcpyAsync(d_poly_off, gt->poly_off, (size_t)(n_polys + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        if (gt->inst_poly_off) AMP_HIP_CHECK(hipMemcpyAsync(d_inst_poly_off, gt->inst_poly_off, (size_t)(total_gt + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        if (gt->rle_off) {
            AMP_REQUIRE(gt->rle_counts && gt->rle_hw, "amp_model_forward_losses: amp_gt.rle_off without rle_counts / rle_hw");
            AMP_HIP_CHECK(hipMemcpyAsync(d_rle_off, gt->rle_off, (size_t)(total_gt + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
            if (total_gt) AMP_HIP_CHECK(hipMemcpyAsync(d_rle_hw, gt->rle_hw, (size_t)total_gt * 8, hipMemcpyHostToDevice, ctx->stream));
        }
        if (total_gt) {
            AMP_HIP_CHECK(hipMemcpyAsync(d_gt_boxes, gt->boxes, (size_t)total_gt * 16, hipMemcpyHostToDevice, ctx->stream));
            AMP_HIP_CHECK(hipMemcpyAsync(d_gt_cls, gt->classes, (size_t)total_gt * 4, hipMemcpyHostToDevice, ctx->stream));
        }
        if (npoly) AMP_HIP_CHECK(hipMemcpyAsync(d_poly_xy, gt->poly_xy, (size_t)npoly * 8, hipMemcpyHostToDevice, ctx->stream));
        AMP_TRY(amp_anchor_labels(ctx, &T.lv, B, d_gt_boxes, d_gt_off, total_gt, c.rpn_iou_lo, c.rpn_iou_hi, match_val, match_idx, gt_best, label));
        AMP_TRY(amp_rpn_sample_loss(ctx, &T.lv, backward ? d_rpn_pred : nullptr, B, d_gt_boxes, d_gt_off, label, match_idx, keys, c.rpn_batch, c.rpn_pos_frac, seed,
                                    rpn_sampled, rpn_counts, rpn_partial));
        tap(m, "rpn_label", label, 3, {B, A});
        tap(m, "rpn_match_idx", match_idx, 1, {B, A});
        tap(m, "rpn_sampled", rpn_sampled, 1, {B, c.rpn_batch});
        tap(m, "rpn_counts", rpn_counts, 1, {B, 2});
    }
    // proposals (training top-k), GT appended, matched and sampled
    Proposals PR;
    AMP_TRY(run_proposals(m, T, B, H, W, c.pre_nms_topk_train, c.post_nms_topk_train, PR));
    const int RB = c.roi_batch;
    const int ncap = c.post_nms_topk_train + c.max_gt;
    AMP_ALLOC(rkeys, uint32_t, (size_t)B * ncap);
    AMP_ALLOC(rcls_s, int, (size_t)B * ncap);
    AMP_ALLOC(rgti_s, int, (size_t)B * ncap);
    AMP_ALLOC(rois, float, (size_t)B * RB * 4);
    AMP_ALLOC(roi_cls, int, (size_t)B * RB);
    AMP_ALLOC(roi_gti, int, (size_t)B * RB);
    AMP_ALLOC(roi_counts, int, (size_t)B * 2);
    const int R = B * RB;
    const int ld_box = (5 * K + 1 + 3) / 4 * 4;
    AMP_ALLOC(roi_batch_idx, int, (size_t)R);
    AMP_ALLOC(pooled, float, (size_t)R * 49 * 256);
    AMP_ALLOC(fc1, float, (size_t)R * 1024);
    AMP_ALLOC(fc2, float, (size_t)R * 1024);
    AMP_ALLOC(box_pred, float, (size_t)R * ld_box);
    AMP_ALLOC(box_partial, float, (size_t)B * 2);
    float* d_box_pred = nullptr;
    if (backward) { AMP_ALLOC(dbp, float, (size_t)R * ld_box); d_box_pred = dbp; }
    std::vector<int> h_counts(2 * B, 0), h_cls, h_gti;
    if (!dry) {
        AMP_TRY(amp_roi_sample(ctx, B, PR.boxes, PR.count, PR.Rcap, d_gt_boxes, d_gt_cls, d_gt_off, K, RB, c.roi_fg_frac, c.roi_iou, seed, rkeys,
                               rcls_s, rgti_s, ncap, rois, roi_cls, roi_gti, roi_counts, PR.anchor, A));
        std::vector<int> iota(R);
        for (int i = 0; i < R; ++i) iota[i] = i / RB;
        AMP_HIP_CHECK(hipMemcpyAsync(roi_batch_idx, iota.data(), (size_t)R * 4, hipMemcpyHostToDevice, ctx->stream));
        AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // iota is a temporary
        // heads of a training step on the native trunk (HS): the pooled tensors and the mask head's hidden activations are stored in the
        // split row format as well (their convolutions stage pre-split operands; wgrad / the ReLU masks of the backward pass read the format)
        const bool HS = backward && m->acts_split && CONV("roi_heads.box_head.fc1").w_split != nullptr;
        AMP_TRY(amp::roi_align_run(ctx, &T.ff, rois, roi_batch_idx, nullptr, R, 7, pooled, nullptr, HS ? 1 : 0, T.feat_split ? 1 : 0));
        AMP_TRY(launch_conv(m, CONV("roi_heads.box_head.fc1"), pooled, 1, 1, R, 1, 0, true, 0, nullptr, 0, fc1, HS ? 1 : 0));
        AMP_TRY(launch_conv(m, CONV("roi_heads.box_head.fc2"), fc1, 1, 1, R, 1, 0, true, 0, nullptr, 0, fc2));
        AMP_TRY(launch_conv(m, CONV("roi_heads.box_predictor"), fc2, 1, 1, R, 1, 0, false, 0, nullptr, 0, box_pred));
        // counts decide the normalisers and the mask-branch sizes
        h_cls.resize(R); h_gti.resize(R);
        AMP_HIP_CHECK(hipMemcpyAsync(h_counts.data(), roi_counts, (size_t)2 * B * 4, hipMemcpyDeviceToHost, ctx->stream));
        AMP_HIP_CHECK(hipMemcpyAsync(h_cls.data(), roi_cls, (size_t)R * 4, hipMemcpyDeviceToHost, ctx->stream));
        AMP_HIP_CHECK(hipMemcpyAsync(h_gti.data(), roi_gti, (size_t)R * 4, hipMemcpyDeviceToHost, ctx->stream));
        AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        tap(m, "train_rois", rois, 0, {B, RB, 4});
        tap(m, "train_roi_cls", roi_cls, 1, {B, RB});
        tap(m, "train_roi_gti", roi_gti, 1, {B, RB});
        tap(m, "train_roi_counts", roi_counts, 1, {B, 2});
        tap(m, "train_box_pred", box_pred, 0, {R, ld_box});
    }
    int total_rois = 0, N = dry ? B * (int)(RB * c.roi_fg_frac) : 0;
    std::vector<int> fg_off(B + 1, 0);
    if (!dry) {
        for (int b = 0; b < B; ++b) { total_rois += h_counts[2 * b] + h_counts[2 * b + 1]; fg_off[b + 1] = fg_off[b] + h_counts[2 * b]; }
        N = fg_off[B];
        AMP_TRY(amp_box_loss(ctx, B, RB, K, box_pred, ld_box, d_box_pred, rois, roi_cls, roi_gti, d_gt_boxes, d_gt_off, c.bbox_reg_weights, total_rois,
                             box_partial));
    }
    // mask branch on the foreground RoIs (the first nfg of every image's sample)
    const int Nc = std::max(N, 1);
    AMP_ALLOC(m_rois, float, (size_t)Nc * 4);
    AMP_ALLOC(m_batch, int, (size_t)Nc);
    AMP_ALLOC(m_cls, int, (size_t)Nc);
    AMP_ALLOC(m_poly, int, (size_t)Nc);
    AMP_ALLOC(mpooled, float, (size_t)Nc * 196 * 256);
    AMP_ALLOC(mt_a, float, (size_t)Nc * 196 * 256);
    AMP_ALLOC(mt_b, float, (size_t)Nc * 784 * 256);
    const int Kp = (K + 3) / 4 * 4;
    AMP_ALLOC(mlogits, float, (size_t)Nc * 784 * Kp);
    AMP_ALLOC(m_partial, float, (size_t)Nc);
    AMP_ALLOC(m_targets, unsigned char, (size_t)Nc * 784);
    float *macts[5] = {mpooled, mt_a, mpooled, mt_a, mpooled};   // input, fcn1..fcn4 outputs (ping-pong in eval mode)
    float* d_mlogits = nullptr;
    if (backward) {
        for (int i = 1; i <= 4; ++i) { AMP_ALLOC(ma, float, (size_t)Nc * 196 * 256); macts[i] = ma; }
        AMP_ALLOC(dml, float, (size_t)Nc * 784 * Kp);
        d_mlogits = dml;
    }
    if (dry && !backward) return AMP_OK;
    std::vector<float> h_mpart(N);
    int bm_overflow = 0;
    if (N > 0 && !dry) {
        std::vector<int> hb(N), hc(N), hp(N);
        for (int b = 0; b < B; ++b) {
            const int nb = fg_off[b + 1] - fg_off[b];
            if (!nb) continue;
            AMP_HIP_CHECK(hipMemcpyAsync(m_rois + (size_t)fg_off[b] * 4, rois + (size_t)b * RB * 4, (size_t)nb * 16, hipMemcpyDeviceToDevice, ctx->stream));
            for (int i = 0; i < nb; ++i) {
                hb[fg_off[b] + i] = b;
                hc[fg_off[b] + i] = h_cls[b * RB + i];
                hp[fg_off[b] + i] = gt->gt_off[b] + h_gti[b * RB + i];
            }
        }
        AMP_HIP_CHECK(hipMemcpyAsync(m_batch, hb.data(), (size_t)N * 4, hipMemcpyHostToDevice, ctx->stream));
        AMP_HIP_CHECK(hipMemcpyAsync(m_cls, hc.data(), (size_t)N * 4, hipMemcpyHostToDevice, ctx->stream));
        AMP_HIP_CHECK(hipMemcpyAsync(m_poly, hp.data(), (size_t)N * 4, hipMemcpyHostToDevice, ctx->stream));
        AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // hb/hc/hp are temporaries
        const bool MS = backward && m->acts_split && split_chain(m, {"roi_heads.mask_head.mask_fcn1", "roi_heads.mask_head.mask_fcn2",
                                                                      "roi_heads.mask_head.mask_fcn3", "roi_heads.mask_head.mask_fcn4"});
        m->mask_acts_split = MS;          // macts[0..3] split
        // macts[4] (the deconv's input, the 'dy' operand of ITS weight gradient and the mask of its data gradient) split as well: the deconv runs on
        // the ring kernel, the predictor's data gradient leaves d(deconv out) * 2^16 as split rows with the deconv's bias sums on the side, and the
        // deconv's weight- and data-gradient launches stage both operands as they are (AMP_NO_MASK_TAIL_SPLIT: fp32 macts[4] and d_mtb as before)
        static const bool no_mt = getenv("AMP_NO_MASK_TAIL_SPLIT") != nullptr;      // EXPERIMENT switch
        const bool MT = MS && (g_mask_tail_split < 0 ? !no_mt : g_mask_tail_split != 0) && Kp <= 16 && (size_t)N * 784 * 256 * 4 < 0x80000000ull && ((long long)N * 196 + 127) / 128 >= 512 /* the ring kernel takes the deconv's data gradient */ && split_chain(m, {"roi_heads.mask_head.deconv"});
        m->mask_tail_split = MT;
        AMP_TRY(amp::roi_align_run(ctx, &T.ff, m_rois, m_batch, nullptr, N, 14, mpooled, nullptr, MS ? 1 : 0, T.feat_split ? 1 : 0));
        for (int i = 1; i <= 4; ++i) {
            const std::string key = "roi_heads.mask_head.mask_fcn" + std::to_string(i);
            AMP_TRY(launch_conv(m, CONV(key.c_str()), macts[i - 1], N, 14, 14, 1, 1, true, 0, nullptr, 0, macts[i], MS ? ((i < 4 || MT) ? 3 : 1) : 0));
        }
        // MT: the deconv's OUTPUT stays in the split row format too (round 4): the ring kernel scatters split rows straight from its accumulators
        // (conv_epilogue_direct, out_mode 1) instead of sending 1.6 GB of fp32 through the staged epilogue; the predictor reads it as a split
        // operand, its weight gradient takes x split, its data gradient the split activation as the ReLU mask (AMP_NO_DECONV_SPLIT: fp32 as before)
        static const bool no_ds = getenv("AMP_NO_DECONV_SPLIT") != nullptr;      // EXPERIMENT switch
        const bool DS = MT && !no_ds && split_chain(m, {"roi_heads.mask_head.predictor"});
        m->deconv_out_split = DS;
        AMP_TRY(launch_conv(m, CONV("roi_heads.mask_head.deconv"), macts[4], N, 14, 14, 1, 0, true, 0, nullptr, 1, mt_b, MT ? (DS ? 3 : 1) : 0));
        AMP_TRY(launch_conv(m, CONV("roi_heads.mask_head.predictor"), mt_b, N, 28, 28, 1, 0, false, 0, nullptr, 0, mlogits, DS ? 1 : 0));
        if (n_runs) {
            // bitmask ground truth: BitMasks.crop_and_resize from the run lengths.  The runs travel in a buffer of their own (their
            // size is data-dependent: not part of the workspace plan), the window bit maps in a scratch of <= 256 frame-sized slots.
            const int Hp = (H + 31) / 32 * 32, Wp = (W + 31) / 32 * 32;
            for (int i = 0; i < total_gt; ++i)
                AMP_REQUIRE(gt->rle_off[i + 1] == gt->rle_off[i] || (gt->rle_hw[2 * i] >= 1 && gt->rle_hw[2 * i] <= Hp && gt->rle_hw[2 * i + 1] >= 1 && gt->rle_hw[2 * i + 1] <= Wp),
                            "amp_model_forward_losses: the bitmask of instance %d is %dx%d, the frame %dx%d", i, gt->rle_hw[2 * i], gt->rle_hw[2 * i + 1], Hp, Wp);
            const size_t slot = (size_t)Wp * (Hp / 32 + 1);
            const int nslots = std::min(N, 256);
            const size_t need = slot * nslots + (n_runs + 63) / 64 * 64;
            if (m->bm_words < need) {
                AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                if (m->bm_scratch) AMP_HIP_CHECK(hipFree(m->bm_scratch));
                m->bm_scratch = nullptr; m->bm_words = 0;
                AMP_HIP_CHECK(hipMalloc(&m->bm_scratch, need * sizeof(unsigned int)));
                m->bm_words = need;
            }
            if (!m->bm_flag) { AMP_HIP_CHECK(hipMalloc(&m->bm_flag, sizeof(int))); AMP_HIP_CHECK(hipMemsetAsync(m->bm_flag, 0, sizeof(int), ctx->stream)); }
            unsigned int* d_runs = m->bm_scratch + slot * nslots;
            AMP_HIP_CHECK(hipMemcpyAsync(d_runs, gt->rle_counts, n_runs * sizeof(unsigned int), hipMemcpyHostToDevice, ctx->stream));
            AMP_TRY(amp_mask_targets_bitmask(ctx, N, m_rois, m_poly, d_rle_off, d_runs, d_rle_hw, m->bm_scratch, slot, nslots, m_targets, m->bm_flag));
        }
        AMP_TRY(amp_mask_target_loss_fmt(ctx, N, Kp, mlogits, d_mlogits, m_rois, m_cls, m_poly, d_poly_xy, d_poly_off, gt->inst_poly_off ? d_inst_poly_off : nullptr,
                                         n_runs ? d_rle_off : nullptr, m_targets, m_partial, m_targets));
        AMP_HIP_CHECK(hipMemcpyAsync(h_mpart.data(), m_partial, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (n_runs) AMP_HIP_CHECK(hipMemcpyAsync(&bm_overflow, m->bm_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        tap(m, "train_mask_targets", m_targets, 4, {N, 28, 28});
        tap(m, "train_mask_logits", mlogits, 0, {N, 28, 28, Kp});
    }
    if (!dry) {
        std::vector<float> h_rpn(2 * B), h_box(2 * B);
        AMP_HIP_CHECK(hipMemcpyAsync(h_rpn.data(), rpn_partial, (size_t)2 * B * 4, hipMemcpyDeviceToHost, ctx->stream));
        AMP_HIP_CHECK(hipMemcpyAsync(h_box.data(), box_partial, (size_t)2 * B * 4, hipMemcpyDeviceToHost, ctx->stream));
        AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        AMP_REQUIRE(!bm_overflow, "amp_model_forward_losses: a bitmask window did not fit its scratch slot (internal sizing error)");
        float s_bce = 0.f, s_loc = 0.f, s_ce = 0.f, s_l1 = 0.f, s_mask = 0.f;
        for (int b = 0; b < B; ++b) { s_bce += h_rpn[2 * b]; s_loc += h_rpn[2 * b + 1]; s_ce += h_box[2 * b]; s_l1 += h_box[2 * b + 1]; }
        for (int i = 0; i < N; ++i) s_mask += h_mpart[i];
        const float rpn_norm = (float)(c.rpn_batch * B);
        losses[0] = total_rois ? s_ce / (float)total_rois : 0.f;
        losses[1] = s_l1 / (float)std::max(total_rois, 1);
        losses[2] = N ? s_mask / ((float)N * 784.f) : 0.f;
        losses[3] = s_bce / rpn_norm;
        losses[4] = s_loc / rpn_norm;
    }
    if (!backward) return AMP_OK;

    // =============================================== backward ===============================================
    // Gradients of the five losses (each with weight 1) w.r.t. every trainable parameter, into m->garena (same offsets as the
    // parameters).  Stem and res2 are frozen (FREEZE_AT = 2), FrozenBN has no parameters; its scale is folded into the weight
    // transforms (data gradients) and the wgrad reduce (weight gradients).
    if (!dry) AMP_TRY(amp::comm_wait_done(ctx));    // a previous exchange of this arena (a step without sgd_step, a re-run) must have finished
    if (!dry) { m->issued_mask = 0; m->grads_valid = false; }
    const size_t WG_SCRATCH = (size_t)64 << 20;     // floats: split-K slabs of the largest layer
    const size_t WT_SCRATCH = (size_t)13 << 20;     // floats: transformed weights of the largest layer (fc1: 12.85 M)
    AMP_ALLOC(wg_scratch, float, WG_SCRATCH);
    AMP_ALLOC(wt_scratch, float, WT_SCRATCH);
    // The slab reductions (69 launches of ~20 us between MFMA kernels) run on a second stream behind an event (amp::wgrad_async_*), with
    // a second scratch so that the next layer's kernel does not wait for the previous layer's reduction.  AMP_ASYNC_REDUCE=0: on the main stream.
    static const bool async_on = getenv("AMP_ASYNC_REDUCE") ? atoi(getenv("AMP_ASYNC_REDUCE")) != 0 : true;
    AMP_ALLOC(wg_scratch1, float, async_on ? WG_SCRATCH : 64);
    struct AsyncScope {          // every exit from the backward pass joins the side stream and leaves the context in the plain mode
        amp_ctx* c; bool on;
        ~AsyncScope() { if (on) (void)amp::wgrad_async_end(c); }
    } async_scope{ctx, false};
    if (!dry && async_on) { AMP_TRY(amp::wgrad_async_begin(ctx, wg_scratch1)); async_scope.on = true; }
    AMP_ALLOC(cs_scratch, float, (size_t)4 << 20);   // ceil(M/512) * N floats of the largest bias-gradient reduction
    const size_t DYS_SCRATCH = (size_t)B * T.fh[0] * T.fw[0] * 256;      // floats: the largest dy converted to a scaled split operand (p2 level)
    AMP_ALLOC(dys_scratch, float, DYS_SCRATCH);
    const bool GSW = m->acts_split && getenv("AMP_NO_SPLIT_GRADS") == nullptr && ctx->conv_mode == AMP_CONV_F16X3;
    const float* dys_of = nullptr;       // the dy tensor whose scaled split copy dys_scratch currently holds
    long long dys_rows = 0;
    auto GW = [&](const ConvW& cw) { return m->garena + (cw.w - m->parena); };
    auto GB = [&](const ConvW& cw) { return m->garena + (cw.shift - m->parena); };
    auto bgrad = [&](const ConvW& cw, const float* dy, long long M_, bool acc) -> int {
        return amp_colsum(ctx, dy, (int)M_, cw.cout, cs_scratch, GB(cw), acc ? 1 : 0);
    };
    // weight gradient; bias = true: the bias gradient (column sums of dy) as well -- summed on the side by the AMP_CONV_F16X3 kernel from the
    // dy tiles it stages anyway, by a separate amp_colsum pass over dy on the fp32 MFMA
    const bool AS = m->acts_split;      // the trunk's saved activations are in the split row format
    auto wgrad = [&](const ConvW& cw, const float* x, int B_, int H_, int W_, int stride, int pad, const float* dy, bool acc, bool bias = false, int xfmt = 0) -> int {   // xfmt: 1 = x split, 2 = dy split and scaled by 2^16
        amp_conv_desc d;
        d.B = B_; d.H = H_; d.W = W_; d.Cin = cw.cin; d.Cout = cw.cout; d.KH = cw.kh; d.KW = cw.kw; d.stride = stride; d.pad = pad;
        d.relu = 0; d.res_mode = 0; d.out_mode = 0;
        AMP_REQUIRE(amp_conv_wgrad_scratch_floats(&d) <= WG_SCRATCH, "backward: wgrad scratch too small");
        static const bool no_fused_bias = getenv("AMP_NO_FUSED_BIAS") != nullptr;      // EXPERIMENT switch
        dys_of = nullptr;                          // a scaled split copy is only trusted right after the weight gradient that made it
        const bool fused = bias && ctx->conv_mode == AMP_CONV_F16X3 && !no_fused_bias && !(xfmt & 2);
        // a big biased layer whose activation is already split (FPN output / RPN conv at p2, p3; the mask head): one pass over dy gives the bias
        // gradient AND dy * 2^16 in the split format, and the weight gradient runs on wgrad_split_kernel (+35-45 % on these shapes)
        static const bool no_conv = getenv("AMP_NO_DY_CONVERT") != nullptr;      // EXPERIMENT switch
        static const long long dyc_min_rows = getenv("AMP_DY_CONVERT_MIN_ROWS") ? atoll(getenv("AMP_DY_CONVERT_MIN_ROWS")) : 200000;      // EXPERIMENT switch
        const long long Mo = (long long)B_ * ((H_ + 2 * pad - cw.kh) / stride + 1) * ((W_ + 2 * pad - cw.kw) / stride + 1);
        if (bias && xfmt == 1 && GSW && !no_conv && cw.cout % 128 == 0 && cw.cin % 128 == 0 && cw.kh * cw.kw * cw.cin >= 256 && (Mo >= dyc_min_rows || (cw.kh * cw.kw * cw.cin >= 4096 && Mo >= 4096)) &&
            (size_t)Mo * cw.cout <= DYS_SCRATCH) {
            AMP_TRY(amp_colsum_split(ctx, dy, (int)Mo, cw.cout, cs_scratch, GB(cw), acc ? 1 : 0, dys_scratch, 16));
            dys_of = dy; dys_rows = Mo;                 // the data-gradient convolution of the same dy can stage this copy (dgrad below)
            return amp_conv2d_wgrad_fmt(ctx, &d, x, dys_scratch, cw.scale, wg_scratch, GW(cw), acc ? 1 : 0, 16, 0, 3, nullptr, 0);
        }
        // AMP_CONV_F16X3 splits dy * 2^16 like the data gradients below (ignored on the fp32 MFMA)
        AMP_TRY(amp_conv2d_wgrad_fmt(ctx, &d, x, dy, cw.scale, wg_scratch, GW(cw), acc ? 1 : 0, 16, 0, xfmt, fused ? GB(cw) : nullptr, acc ? 1 : 0));
        if (bias && !fused) {
            const int Ho_ = (H_ + 2 * pad - cw.kh) / stride + 1, Wo_ = (W_ + 2 * pad - cw.kw) / stride + 1;
            return bgrad(cw, dy, (long long)B_ * Ho_ * Wo_, acc);
        }
        return AMP_OK;
    };
    // the split data-gradient weights of this step (refresh_dgrad_weights) where conv_run would take a split copy: its LDS-DMA kernels,
    // i.e. operands below 2 GiB (cout % 32 == 0 by construction); null: transpose and split in front of the launch as before
    auto dgrad_wsplit = [&](const ConvW& cw, int B_, int Hy, int Wy) -> const float* {
        if (!dgrad_batch || !cw.wt_split) return nullptr;
        const size_t xb = (size_t)B_ * Hy * Wy * cw.cout * 4, wb = (size_t)cw.cout * cw.kh * cw.kw * cw.cin * 4;
        return (xb < 0x80000000ull && wb < 0x80000000ull) ? cw.wt_split : nullptr;
    };
    // dx = conv(dy, flipped/transposed/scaled w) (+ res) (* mask>0); dy is [B_,Hy,Wy,cw.cout]
    auto dgrad = [&](const ConvW& cw, const float* dy, int B_, int Hy, int Wy, int fwd_pad, const float* res, const float* mask, float* dx, bool mask_split = false) -> int {
        AMP_REQUIRE((size_t)cw.cout * cw.kh * cw.kw * cw.cin <= WT_SCRATCH, "backward: weight-transform scratch too small");
        const float* wsp = dgrad_wsplit(cw, B_, Hy, Wy);
        const float* wt = wsp ? cw.w : wt_scratch;        // with a split copy the fp32 pointer is not read
        if (!wsp) AMP_TRY(amp_dgrad_weights(ctx, cw.w, cw.scale, cw.cout, cw.kh, cw.kw, cw.cin, wt_scratch));
        amp_conv_desc d;
        d.B = B_; d.H = Hy; d.W = Wy; d.Cin = cw.cout; d.Cout = cw.cin; d.KH = cw.kh; d.KW = cw.kw; d.stride = 1; d.pad = cw.kh - 1 - fwd_pad;
        d.relu = 0; d.res_mode = res ? 1 : 0; d.out_mode = 0;
        // data gradient on the context's arithmetic; AMP_CONV_F16X3 splits dy * 2^16 (gradients of 1e-9..1e-4 would sit in the f16
        // subnormals) -- unless the weight gradient of the same dy has just left that very tensor in dys_scratch: then the ring kernel stages
        // it as it is and undoes the 2^16 in its fold
        static const bool no_reuse = getenv("AMP_NO_DY_REUSE") != nullptr;      // EXPERIMENT switch
        const long long nblk256 = (((long long)B_ * Hy * Wy + 127) / 128) * (cw.cin / 256);      // conv_run takes the 128 x 256 ring kernel for this
        const bool ring_fits = cw.cin % 256 == 0 && (nblk256 >= 512 || (nblk256 >= 192 && cw.kh * cw.kw * cw.cout / 32 >= 64));
        if (dy == dys_of && !no_reuse && dys_rows == (long long)B_ * Hy * Wy && ring_fits && ctx->conv_mode == AMP_CONV_F16X3)
            return amp::conv_run(ctx, &d, 1, dys_scratch, wt, wsp, 0, nullptr, nullptr, res, mask, dx, 16, 1 | ((mask && mask_split) ? 8 : 0));
        return amp::conv_run(ctx, &d, 1, dy, wt, wsp, 0, nullptr, nullptr, res, mask, dx, 16, (mask && mask_split) ? 8 : 0);
    };
    // The backbone's chain on SCALED SPLIT gradients (GS): a gradient tensor is kept as the split rows of d * 2^16, so its data-gradient
    // convolution stages both operands by LDS-DMA on the ring kernel (no split, no scaling: the scale rides through the linear chain),
    // residual and ReLU mask come in the split format, and the weight-gradient kernel passes the halves through (xfmt 3).
    static const bool no_gs = getenv("AMP_NO_SPLIT_GRADS") != nullptr;      // EXPERIMENT switch
    const bool GS = AS && !no_gs && ctx->conv_mode == AMP_CONV_F16X3 && m->gs_chain_ok;
    static const bool no_gx = getenv("AMP_NO_GX") != nullptr;      // EXPERIMENT switch: ResNeXt blocks on fp32 gradients (split activations only)
    const bool GX = AS && !GS && !no_gs && (g_gx < 0 ? !no_gx : g_gx != 0) && ctx->conv_mode == AMP_CONV_F16X3 && c.num_groups > 1 && !c.stride_in_1x1;
    if (!dry) m->last_chain = !AS ? 0 : (GS ? 2 : (GX ? 3 : 1));      // which backbone chain ran (amp_debug_last_backward_chain)
    auto dgrad_s = [&](const ConvW& cw, const float* dy_s, int B_, int Hy, int Wy, int fwd_pad, const float* res_s, const float* mask_s, float* dx_s) -> int {
        AMP_REQUIRE((size_t)cw.cout * cw.kh * cw.kw * cw.cin <= WT_SCRATCH, "backward: weight-transform scratch too small");
        const float* wsp = dgrad_wsplit(cw, B_, Hy, Wy);
        if (!wsp) AMP_TRY(amp_dgrad_weights(ctx, cw.w, cw.scale, cw.cout, cw.kh, cw.kw, cw.cin, wt_scratch));
        amp_conv_desc d;
        d.B = B_; d.H = Hy; d.W = Wy; d.Cin = cw.cout; d.Cout = cw.cin; d.KH = cw.kh; d.KW = cw.kw; d.stride = 1; d.pad = cw.kh - 1 - fwd_pad;
        d.relu = 0; d.res_mode = res_s ? 1 : 0; d.out_mode = 0;
        return amp::conv_run(ctx, &d, 1, dy_s, wsp ? cw.w : wt_scratch, wsp, 0, nullptr, nullptr, res_s, mask_s, dx_s, 0, 1 | 2 | (res_s ? 4 : 0) | (mask_s ? 8 : 0));
    };

    // ---- gradient buffers of the FPN outputs p2..p6 ----
    float* d_feat[5];
    for (int l = 0; l < 5; ++l) {
        AMP_ALLOC(df, float, (size_t)B * T.fh[l] * T.fw[l] * 256);
        d_feat[l] = df;
        // p2..p5: the first RoIAlign backward below writes every cell (amp::roi_align_bwd_run, init): no zero fill of 1.3 GB at B = 16; p6 has no pooler
        if (!dry && l == 4) AMP_HIP_CHECK(hipMemsetAsync(df, 0, (size_t)B * T.fh[l] * T.fw[l] * 256 * 4, ctx->stream));
    }
    int dfeat_init = 1;      // handed to the first RoIAlign backward, 0 afterwards
    int fstr[4] = {4, 8, 16, 32};

    // ---- mask head ----
    AMP_ALLOC(d_mtb, float, (size_t)Nc * 784 * 256);
    AMP_ALLOC(d_ma, float, (size_t)Nc * 196 * 256);
    AMP_ALLOC(d_mb, float, (size_t)Nc * 196 * 256);
    AMP_ALLOC(dwd_t, float, (size_t)256 * 4 * 256);
    AMP_ALLOC(dbd_t, float, 256);
    if (!dry && N > 0) {
        const ConvW& cp = CONV("roi_heads.mask_head.predictor");
        const bool MT = m->mask_tail_split && ctx->conv_mode == AMP_CONV_F16X3;
        const bool DS = MT && m->deconv_out_split;
        AMP_TRY(wgrad(cp, mt_b, N, 28, 28, 1, 0, d_mlogits, false, true, DS ? 1 : 0));
        if (DS) AMP_TRY(amp_small_k_dgrad_split_ld(ctx, d_mlogits, Kp, K, cp.w, 256, mt_b, d_mtb, N * 784, 16, cs_scratch, dbd_t, 0));    // + the deconv's bias sums
        else if (MT) AMP_TRY(amp_small_k_dgrad_split_f32act(ctx, d_mlogits, Kp, K, cp.w, 256, mt_b, d_mtb, N * 784, 16, cs_scratch, dbd_t, 0));
        else AMP_TRY(amp_small_k_dgrad(ctx, d_mlogits, Kp, K, cp.w, 256, mt_b, d_mtb, (size_t)N * 784));
        const ConvW& cd = CONV("roi_heads.mask_head.deconv");
        {   // weight gradient in [ci][tap][co] form, then transposed into the forward layout [(tap,co)][ci]
            amp_conv_desc d;
            d.B = N; d.H = 28; d.W = 28; d.Cin = 256; d.Cout = 256; d.KH = 2; d.KW = 2; d.stride = 2; d.pad = 0; d.relu = 0; d.res_mode = 0; d.out_mode = 0;
            AMP_REQUIRE(amp_conv_wgrad_scratch_floats(&d) <= WG_SCRATCH, "backward: wgrad scratch too small");
            // here the gradient is the 'x' operand (MT: both operands split rows, the 2^16 sits in x and is undone like a dy shift)
            if (MT) AMP_TRY(amp_conv2d_wgrad_fmt(ctx, &d, d_mtb, macts[4], nullptr, wg_scratch, dwd_t, 0, 16, 0, 3, nullptr, 0));
            else AMP_TRY(amp_conv2d_wgrad_scaled(ctx, &d, d_mtb, macts[4], nullptr, wg_scratch, dwd_t, 0, 0, 16));
            AMP_TRY(amp::wgrad_async_join(ctx));      // dwd_t is read right away
            AMP_TRY(amp_deconv_grad_transpose(ctx, dwd_t, GW(cd), 256, 4, 256, 0));
            if (!MT) AMP_TRY(amp_colsum(ctx, d_mtb, N * 784, 256, cs_scratch, dbd_t, 0));
            for (int q = 0; q < 4; ++q) AMP_HIP_CHECK(hipMemcpyAsync(GB(cd) + q * 256, dbd_t, 256 * 4, hipMemcpyDeviceToDevice, ctx->stream));
            // data gradient: a 2x2 stride-2 convolution of d_mtb with w[ci][ky][kx][co], masked by fcn4's ReLU
            const float* wsp = (cd.scale == nullptr && cd.cout == 1024 && cd.cin == 256) ? dgrad_wsplit(cd, N, 28, 28) : nullptr;   // [ci][(tap, co)] = the 2x2 window rows
            if (!wsp) AMP_TRY(amp_dgrad_weights(ctx, cd.w, nullptr, 1024, 1, 1, 256, wt_scratch));
            amp_conv_desc g;
            g.B = N; g.H = 28; g.W = 28; g.Cin = 256; g.Cout = 256; g.KH = 2; g.KW = 2; g.stride = 2; g.pad = 0; g.relu = 0; g.res_mode = 0; g.out_mode = 0;
            // MT: the data gradient leaves the mask head's chain in the scaled split domain (the 2^16 rides on): fcn4..fcn1 below take d * 2^16 as
            // split rows straight from the producing convolution's epilogue -- no fp32 tensor, no conversion pass
            if (MT) AMP_TRY(amp::conv_run(ctx, &g, 1, d_mtb, wsp ? cd.w : wt_scratch, wsp, 0, nullptr, nullptr, nullptr, macts[4], d_ma, 0, 1 | 2 | 8));
            else AMP_TRY(amp::conv_run(ctx, &g, 1, d_mtb, wsp ? cd.w : wt_scratch, wsp, 0, nullptr, nullptr, nullptr, macts[4], d_ma, 16, 0));
        }
        float* dcur = d_ma;
        float* dnext = d_mb;
        for (int i = 4; i >= 1; --i) {
            const std::string key = "roi_heads.mask_head.mask_fcn" + std::to_string(i);
            const ConvW& cf = CONV(key.c_str());
            if (MT) {
                // bias gradient: column sums of the split rows (read-only: half the bytes of the converting pass); weight gradient on both split
                // operands; data gradient split -> split with fcn(i-1)'s split rows as the ReLU mask, the last one (no mask) back to fp32 for RoIAlign
                AMP_TRY(amp_colsum_of_split(ctx, dcur, N * 196, 256, cs_scratch, GB(cf), 0, 16));
                AMP_TRY(wgrad(cf, macts[i - 1], N, 14, 14, 1, 1, dcur, false, false, 3));
                if (i > 1) {
                    AMP_TRY(dgrad_s(cf, dcur, N, 14, 14, 1, nullptr, macts[i - 1], dnext));
                } else {
                    AMP_REQUIRE((size_t)cf.cout * cf.kh * cf.kw * cf.cin <= WT_SCRATCH, "backward: weight-transform scratch too small");
                    const float* wsp1 = dgrad_wsplit(cf, N, 14, 14);
                    if (!wsp1) AMP_TRY(amp_dgrad_weights(ctx, cf.w, cf.scale, cf.cout, cf.kh, cf.kw, cf.cin, wt_scratch));
                    amp_conv_desc d1;
                    d1.B = N; d1.H = 14; d1.W = 14; d1.Cin = cf.cout; d1.Cout = cf.cin; d1.KH = cf.kh; d1.KW = cf.kw; d1.stride = 1; d1.pad = cf.kh - 1 - 1;
                    d1.relu = 0; d1.res_mode = 0; d1.out_mode = 0;
                    AMP_TRY(amp::conv_run(ctx, &d1, 1, dcur, wsp1 ? cf.w : wt_scratch, wsp1, 0, nullptr, nullptr, nullptr, nullptr, dnext, 16, 1));
                }
            } else {
                AMP_TRY(wgrad(cf, macts[i - 1], N, 14, 14, 1, 1, dcur, false, true, m->mask_acts_split));
                AMP_TRY(dgrad(cf, dcur, N, 14, 14, 1, nullptr, i > 1 ? macts[i - 1] : nullptr, dnext, m->mask_acts_split));
            }
            std::swap(dcur, dnext);
        }
        AMP_TRY(amp::roi_align_bwd_run(ctx, d_feat, T.fh, T.fw, fstr, 256, m_rois, m_batch, N, 14, dcur, B, dfeat_init));
        dfeat_init = 0;
    } else if (!dry) {
        for (const char* key : {"roi_heads.mask_head.predictor", "roi_heads.mask_head.deconv", "roi_heads.mask_head.mask_fcn1", "roi_heads.mask_head.mask_fcn2",
                                "roi_heads.mask_head.mask_fcn3", "roi_heads.mask_head.mask_fcn4"}) {
            const ConvW& cw = CONV(key);
            AMP_HIP_CHECK(hipMemsetAsync(GW(cw), 0, (size_t)cw.cout * cw.kh * cw.kw * cw.cin * 4, ctx->stream));
            AMP_HIP_CHECK(hipMemsetAsync(GB(cw), 0, (size_t)cw.cout * 4, ctx->stream));
        }
    }
    AMP_TRY(issue_bucket(m, 0));

    // ---- box head ----
    AMP_ALLOC(d_fc2, float, (size_t)R * 1024);
    AMP_ALLOC(d_fc1, float, (size_t)R * 1024);
    AMP_ALLOC(d_pooled, float, (size_t)R * 49 * 256);
    if (!dry) {
        const ConvW& cb = CONV("roi_heads.box_predictor");
        AMP_TRY(wgrad(cb, fc2, 1, 1, R, 1, 0, d_box_pred, false, true));
        AMP_TRY(dgrad(cb, d_box_pred, 1, 1, R, 0, nullptr, fc2, d_fc2));
        const ConvW& c2 = CONV("roi_heads.box_head.fc2");
        AMP_TRY(wgrad(c2, fc1, 1, 1, R, 1, 0, d_fc2, false, true));
        AMP_TRY(dgrad(c2, d_fc2, 1, 1, R, 0, nullptr, fc1, d_fc1));
        const ConvW& c1 = CONV("roi_heads.box_head.fc1");
        AMP_TRY(wgrad(c1, pooled, 1, 1, R, 1, 0, d_fc1, false, true, AS && c1.w_split != nullptr));
        AMP_TRY(dgrad(c1, d_fc1, 1, 1, R, 0, nullptr, nullptr, d_pooled));
        AMP_TRY(amp::roi_align_bwd_run(ctx, d_feat, T.fh, T.fw, fstr, 256, rois, roi_batch_idx, R, 7, d_pooled, B, dfeat_init));
        dfeat_init = 0;
    }
    AMP_TRY(issue_bucket(m, 1));

    // ---- RPN head (shared weights: gradients accumulate over the 5 levels) ----
    // The losses touch <= RPN.BATCH_SIZE_PER_IMAGE anchors per image: the head's gradients over those pixels only (rpn_sparse.hip), all levels in one list
    const int SRR = B * c.rpn_batch;
    AMP_ALLOC(sr_rows, unsigned int, (size_t)SRR);
    AMP_ALLOC(sr_nrows, int, (size_t)B);
    AMP_ALLOC(sr_dpred, float, (size_t)SRR * 16);
    AMP_ALLOC(sr_act, float, (size_t)SRR * 256);
    AMP_ALLOC(sr_dt, float, (size_t)SRR * 256);
    AMP_ALLOC(sr_xg, float, (size_t)SRR * 2304);
    AMP_ALLOC(sr_G, float, (size_t)SRR * 2304);
    AMP_ALLOC(sr_wt, float, (size_t)2304 * 256);
    bool SR = false;
    if (!dry) {
        const ConvW& cpred = CONV("proposal_generator.rpn_head.pred");
        const ConvW& cconv = CONV("proposal_generator.rpn_head.conv");
        SR = sr_ok && T.lv.ld == 16;
        AMP_REQUIRE(SR || !rpn_fused, "%s", "backward: the RPN head's hidden tensor was not saved and the sparse backward pass does not apply");
        if (SR) {
            amp::RpnSparseArgs sa;
            sa.B = B; sa.batch = c.rpn_batch; sa.ld = T.lv.ld; sa.K = cpred.cout; sa.C = 256;
            for (int l = 0; l < 5; ++l) {
                sa.fh[l] = T.fh[l]; sa.fw[l] = T.fw[l];
                sa.dpred[l] = d_rpn_pred[l]; sa.t[l] = m->rpn_t[l]; sa.feat[l] = T.feat[l]; sa.dfeat[l] = d_feat[l];
            }
            sa.sampled = rpn_sampled; sa.counts = rpn_counts;
            sa.t_split = AS ? 1 : 0; sa.feat_split = AS ? 1 : 0;
            sa.recompute_t = rpn_fused ? 1 : 0; sa.w_conv_split = cconv.w_split; sa.conv_shift = cconv.shift;
            sa.w_pred = cpred.w; sa.w_conv = cconv.w; sa.conv_scale = cconv.scale;
            sa.gw_pred = GW(cpred); sa.gb_pred = GB(cpred); sa.gw_conv = GW(cconv); sa.gb_conv = GB(cconv);
            sa.rows = sr_rows; sa.nrows = sr_nrows; sa.dpred_rows = sr_dpred; sa.act_rows = sr_act; sa.dt_rows = sr_dt; sa.xg = sr_xg; sa.G = sr_G; sa.wt = sr_wt;
            sa.wg_scratch = wg_scratch; sa.wg_scratch_floats = WG_SCRATCH;
            AMP_TRY(amp::rpn_sparse_backward(ctx, sa));
            dys_of = nullptr;
        }
        m->last_rpn_sparse = SR ? 1 : 0;
    }
    for (int l = 0; l < 5; ++l) {
        AMP_ALLOC(d_t, float, (size_t)B * T.fh[l] * T.fw[l] * 256);
        if (dry || SR) continue;
        const ConvW& cpred = CONV("proposal_generator.rpn_head.pred");
        const ConvW& cconv = CONV("proposal_generator.rpn_head.conv");
        AMP_TRY(wgrad(cpred, m->rpn_t[l], B, T.fh[l], T.fw[l], 1, 0, d_rpn_pred[l], l > 0, true, AS));
        const long long Ml = (long long)B * T.fh[l] * T.fw[l];
        static const bool no_rpn_split = getenv("AMP_NO_RPN_SPLIT") != nullptr;      // EXPERIMENT switch
        if (GSW && !no_rpn_split && Ml >= 200000 && (size_t)Ml * 256 <= DYS_SCRATCH && cpred.scale == nullptr && T.lv.ld == 16 && cpred.cout <= 16 && cconv.cout == 256) {
            // big levels on the native trunk: the predictor's data gradient (K = 15) is one bandwidth-bound pass that writes d_t * 2^16 as split
            // rows and sums the conv's bias gradient on the side; both gradients of the conv then stage that copy on the ring kernels
            AMP_TRY(amp_small_k_dgrad_split(ctx, d_rpn_pred[l], cpred.cout, cpred.w, 256, m->rpn_t[l], dys_scratch, (int)Ml, 16, cs_scratch, GB(cconv), l > 0 ? 1 : 0));
            amp_conv_desc dd;
            dd.B = B; dd.H = T.fh[l]; dd.W = T.fw[l]; dd.Cin = cconv.cin; dd.Cout = cconv.cout; dd.KH = cconv.kh; dd.KW = cconv.kw; dd.stride = 1; dd.pad = 1;
            dd.relu = 0; dd.res_mode = 0; dd.out_mode = 0;
            AMP_REQUIRE(amp_conv_wgrad_scratch_floats(&dd) <= WG_SCRATCH, "backward: wgrad scratch too small");
            AMP_TRY(amp_conv2d_wgrad_fmt(ctx, &dd, T.feat[l], dys_scratch, cconv.scale, wg_scratch, GW(cconv), l > 0 ? 1 : 0, 16, 0, 3, nullptr, 0));
            dys_of = d_t; dys_rows = Ml;             // (d_t itself stays unwritten: its only reader below takes the split copy)
        } else {
            AMP_TRY(dgrad(cpred, d_rpn_pred[l], B, T.fh[l], T.fw[l], 0, nullptr, m->rpn_t[l], d_t, AS));
            AMP_TRY(wgrad(cconv, T.feat[l], B, T.fh[l], T.fw[l], 1, 1, d_t, l > 0, true, AS));
        }
        AMP_TRY(dgrad(cconv, d_t, B, T.fh[l], T.fw[l], 1, d_feat[l], nullptr, d_feat[l]));   // accumulate in place
    }
    AMP_TRY(issue_bucket(m, 2));

    // ---- FPN ----
    if (!dry) AMP_TRY(amp_subsample2_bwd(ctx, d_feat[4], d_feat[3], B, T.fh[3], T.fw[3], 256));    // p6 = p5[::2, ::2]
    float* d_res[4] = {nullptr, nullptr, nullptr, nullptr};   // gradients w.r.t. res3..res5 outputs (index = stage)
    float* d_lat_prev = nullptr;                              // 2x2 sums of the finer level's lateral gradient
    for (int l = 2; l <= 5; ++l) {
        const int s_ = l - 2;
        const int fh_ = T.fh[s_], fw_ = T.fw[s_];
        AMP_ALLOC(d_lat, float, (size_t)B * fh_ * fw_ * 256);
        float* d_lat_next = nullptr;
        if (l < 5) { AMP_ALLOC(dln, float, (size_t)B * T.fh[s_ + 1] * T.fw[s_ + 1] * 256); d_lat_next = dln; }
        if (l >= 3) { AMP_ALLOC(dr, float, (size_t)B * fh_ * fw_ * kOut[s_]); d_res[s_] = dr; }
        if (dry) { d_lat_prev = d_lat_next; continue; }
        const std::string ln = "backbone.fpn_lateral" + std::to_string(l), on = "backbone.fpn_output" + std::to_string(l);
        const ConvW& co = CONV(on.c_str());
        const ConvW& cl = CONV(ln.c_str());
        AMP_TRY(wgrad(co, m->lat[s_], B, fh_, fw_, 1, 1, d_feat[s_], false, true, AS));
        AMP_TRY(dgrad(co, d_feat[s_], B, fh_, fw_, 1, d_lat_prev, nullptr, d_lat));      // + top-down share from the finer level
        if (l < 5) {
            AMP_TRY(amp::upsample2_bwd_run(ctx, d_lat, d_lat_next, B, T.fh[s_ + 1], T.fw[s_ + 1], 256, 1));      // first writer: no zero fill
        }
        const float* res_in = nullptr;   // input of the lateral conv = output of stage s_
        for (auto& ba : m->blocks) if (ba.stage == s_) res_in = ba.out;
        AMP_TRY(wgrad(cl, res_in, B, fh_, fw_, 1, 0, d_lat, false, true, AS));
        if (l >= 3) AMP_TRY(dgrad(cl, d_lat, B, fh_, fw_, 0, nullptr, nullptr, d_res[s_]));
        d_lat_prev = d_lat_next;
    }
    AMP_TRY(issue_bucket(m, 3));

    // ---- ResNet res5 .. res3 (reverse block order) ----
    float* dcur = nullptr;
    bool dcur_masked = false;
    const int nblk = dry ? 0 : (int)m->blocks.size();
    // dry run: reserve the worst-case gradient buffers of every trainable block
    if (dry) {
        int hh = ((H + 31) / 32 * 32) / 4, ww = ((W + 31) / 32 * 32) / 4;
        for (int s_ = 1; s_ < 4; ++s_) {
            hh = (hh - 1) / 2 + 1; ww = (ww - 1) / 2 + 1;
            { AMP_ALLOC(r0, float, (size_t)B * hh * ww * kOut[s_]); (void)r0; }     // the stage's entry gradient as a scaled split tensor
            for (int b_ = 0; b_ < m->nblk[s_]; ++b_) {
                AMP_ALLOC(r1, float, (size_t)B * hh * ww * m->mid[s_]);
                AMP_ALLOC(r2, float, (size_t)B * hh * ww * m->mid[s_]);
                AMP_ALLOC(r3, float, (size_t)B * hh * ww * kOut[s_]);
                AMP_ALLOC(r4, float, (size_t)B * hh * ww * kOut[s_]);
                (void)r1; (void)r2; (void)r3; (void)r4;
                if (m->cfg.num_groups > 1) {     // ResNeXt on the split trunk: conv2's input decoded to fp32 for the grouped weight gradient (at the block's INPUT resolution in its first block)
                    const int up = (b_ == 0 && !c.stride_in_1x1) ? 2 : 1;
                    AMP_ALLOC(r5, float, (size_t)B * (hh * up) * (ww * up) * m->mid[s_]);
                    AMP_ALLOC(r6, float, (size_t)B * hh * ww * m->mid[s_]);          // ... and its output gradient (the scaled split chain)
                    (void)r5; (void)r6;
                }
            }
        }
    }
    for (int bi = nblk - 1; bi >= 0; --bi) {
        const amp_model::BlockAct& ba = m->blocks[bi];
        if (ba.stage == 0) break;                               // res2 is frozen
        const bool last_of_stage = (bi + 1 == nblk) || m->blocks[bi + 1].stage != ba.stage;
        if (last_of_stage) {
            // gradient from the FPN lateral; the next stage's first block (if any) has already added its share into d_res
            dcur = d_res[ba.stage];
            dcur_masked = false;
        }
        const size_t out_elems = (size_t)B * ba.oh * ba.ow * ba.cout;
        const ConvW& c3 = CONV((ba.key + ".conv3").c_str());
        const ConvW& c2 = CONV((ba.key + ".conv2").c_str());
        const ConvW& c1 = CONV((ba.key + ".conv1").c_str());
        // RESNETS.STRIDE_IN_1X1: the block's stride sits in conv1 (MSRA R50 / R101) or in the 3x3 conv2 (ResNeXt); t1 is conv1's output
        const int st1 = c.stride_in_1x1 ? ba.stride : 1, st2 = c.stride_in_1x1 ? 1 : ba.stride;
        const int h1 = (ba.in_h - 1) / st1 + 1, w1 = (ba.in_w - 1) / st1 + 1;
        AMP_ALLOC(d_t2, float, (size_t)B * ba.oh * ba.ow * ba.mid);
        AMP_ALLOC(d_t1, float, (size_t)B * h1 * w1 * ba.mid);
        float* d_t2_up = nullptr;            // stride-2 conv2: dy spread over the even positions of a zeroed map, then a stride-1 data gradient
        if (st2 == 2) { AMP_ALLOC(up, float, (size_t)B * h1 * w1 * ba.mid); d_t2_up = up; }
        // conv2's two gradients: dense, or grouped (ResNeXt: window layout, grouped_bwd.hip)
        auto c2_backward = [&]() -> int {
            const float* dy2 = d_t2;
            if (st2 == 2) {
                AMP_HIP_CHECK(hipMemsetAsync(d_t2_up, 0, (size_t)B * h1 * w1 * ba.mid * 4, ctx->stream));
                AMP_TRY(amp_subsample2_bwd(ctx, d_t2, d_t2_up, B, h1, w1, ba.mid));
                dy2 = d_t2_up;
            } else {
                AMP_REQUIRE(st2 == 1, "backward: conv2 with stride %d", st2);
            }
            if (c2.groups > 1) {
                amp_conv_desc dw;
                dw.B = B; dw.H = h1; dw.W = w1; dw.Cin = ba.mid; dw.Cout = ba.mid; dw.KH = 3; dw.KW = 3; dw.stride = st2; dw.pad = 1; dw.relu = 0; dw.res_mode = 0; dw.out_mode = 0;
                AMP_REQUIRE(amp_grouped_wgrad_scratch_floats(&dw) <= WG_SCRATCH && (size_t)ba.mid * 9 * 64 <= WT_SCRATCH, "backward: grouped scratch too small");
                AMP_TRY(amp::wgrad_async_join(ctx));      // the grouped kernels use wg_scratch on the main stream: no reduction may still be reading it
                // t1 is split on the native trunk.  grouped_wgrad_kernel can decode it on the load (amp_conv2d_grouped_wgrad_fmt, fmt 1), but its
                // loop is bound by load issue and the two 2-byte loads + converts per value DOUBLE its time (6.6 -> 13.0 ms per X-101 step at
                // B = 4 / 1024^2, measured); one decoding pass into a scratch (read 4 B, write 4 B per value: ~0.05 ms per layer) and the fp32 form is cheaper
                const float* t1f = ba.t1;
                if (AS) {
                    AMP_ALLOC(t1_dec, float, (size_t)B * h1 * w1 * ba.mid);
                    AMP_TRY(amp_unsplit_rows(ctx, ba.t1, (long long)B * h1 * w1, ba.mid, t1_dec));
                    t1f = t1_dec;
                }
                AMP_TRY(amp_conv2d_grouped_wgrad(ctx, &dw, c2.groups, t1f, d_t2, c2.scale, wg_scratch, GW(c2)));
                AMP_TRY(amp_group_dgrad_weights(ctx, c2.w, c2.scale, ba.mid, 3, 3, wt_scratch));
                amp_conv_desc dd = dw;
                dd.stride = 1;
                dys_of = nullptr;
                return amp::conv_run(ctx, &dd, c2.groups, dy2, wt_scratch, nullptr, 0, nullptr, nullptr, nullptr, ba.t1, d_t1, 16, AS ? 8 : 0);      // (the ReLU mask t1 likewise)
            }
            AMP_TRY(wgrad(c2, ba.t1, B, h1, w1, st2, 1, d_t2, false, false, AS));
            return dgrad(c2, dy2, B, h1, w1, 1, nullptr, ba.t1, d_t1, AS);
        };
        const bool need_dx = !(ba.stage == 1 && ba.has_sc);   // the input of res3.0 is the frozen res2 output
        if (GS) {
            // every gradient of the chain below is a scaled split tensor (see dgrad_s)
            if (last_of_stage) {
                AMP_ALLOC(dcur_s, float, out_elems);
                AMP_TRY(amp_relu_mask_to_split(ctx, dcur, ba.out, dcur_s, out_elems, ba.cout, 16));
                dcur = dcur_s;
            } else if (!dcur_masked) {
                amp::set_error("backward: a block whose input is not the previous block's output is not expected inside a stage");
                return AMP_ERR_STATE;
            }
            dcur_masked = false;
            AMP_TRY(wgrad(c3, ba.t2, B, ba.oh, ba.ow, 1, 0, dcur, false, false, 3));
            AMP_TRY(dgrad_s(c3, dcur, B, ba.oh, ba.ow, 0, nullptr, ba.t2, d_t2));
            AMP_TRY(wgrad(c2, ba.t1, B, ba.oh, ba.ow, 1, 1, d_t2, false, false, 3));
            AMP_TRY(dgrad_s(c2, d_t2, B, ba.oh, ba.ow, 1, nullptr, ba.t1, d_t1));
            AMP_TRY(wgrad(c1, ba.x_in, B, ba.in_h, ba.in_w, ba.stride, 0, d_t1, false, false, 3));
            if (ba.has_sc) {
                const ConvW& cs = CONV((ba.key + ".shortcut").c_str());
                AMP_TRY(wgrad(cs, ba.x_in, B, ba.in_h, ba.in_w, ba.stride, 0, dcur, false, false, 3));
                if (need_dx) {
                    AMP_ALLOC(tmp_sc, float, (size_t)B * ba.oh * ba.ow * ba.cin);
                    AMP_ALLOC(tmp_in, float, (size_t)B * ba.oh * ba.ow * ba.cin);
                    AMP_TRY(dgrad_s(cs, dcur, B, ba.oh, ba.ow, 0, nullptr, nullptr, tmp_sc));
                    AMP_TRY(dgrad_s(c1, d_t1, B, ba.oh, ba.ow, 0, tmp_sc, nullptr, tmp_in));
                    float* dprev = d_res[ba.stage - 1];             // fp32: already holds the FPN lateral's share
                    if (ba.stride == 2) AMP_TRY(amp_subsample2_bwd_split(ctx, tmp_in, dprev, B, ba.in_h, ba.in_w, ba.cin, 16));
                    else { amp::set_error("backward: stride-1 projection block is not expected here"); return AMP_ERR_STATE; }
                }
            } else {
                AMP_ALLOC(d_in, float, out_elems);
                const bool fuse = bi > 0 && m->blocks[bi - 1].out == ba.x_in;
                AMP_TRY(dgrad_s(c1, d_t1, B, ba.oh, ba.ow, 0, dcur, fuse ? ba.x_in : nullptr, d_in));
                dcur = d_in;
                dcur_masked = fuse;
            }
        } else if (GX) {
            // ResNeXt (grouped conv2, the block's stride in it) on the scaled split gradient chain: conv3 / conv1 / shortcut exactly as above;
            // conv2's weight gradient on decoded operands (the 2^16 undone by its reduce pass), its data gradient = the grouped FORWARD kernel on
            // transposed windows, split in / split out with t1 as the ReLU mask (a stride-2 layer: dy's rows scattered into a zeroed map first);
            // in a stage's first block conv1's data gradient lives at the INPUT resolution and joins the fp32 stage-exit map there
            if (last_of_stage) {
                AMP_ALLOC(dcur_s, float, out_elems);
                AMP_TRY(amp_relu_mask_to_split(ctx, dcur, ba.out, dcur_s, out_elems, ba.cout, 16));
                dcur = dcur_s;
            } else if (!dcur_masked) {
                amp::set_error("backward: a block whose input is not the previous block's output is not expected inside a stage");
                return AMP_ERR_STATE;
            }
            dcur_masked = false;
            AMP_TRY(wgrad(c3, ba.t2, B, ba.oh, ba.ow, 1, 0, dcur, false, false, 3));
            AMP_TRY(dgrad_s(c3, dcur, B, ba.oh, ba.ow, 0, nullptr, ba.t2, d_t2));
            {
                amp_conv_desc dw;
                dw.B = B; dw.H = h1; dw.W = w1; dw.Cin = ba.mid; dw.Cout = ba.mid; dw.KH = 3; dw.KW = 3; dw.stride = st2; dw.pad = 1; dw.relu = 0; dw.res_mode = 0; dw.out_mode = 0;
                AMP_REQUIRE(amp_grouped_wgrad_scratch_floats(&dw) <= WG_SCRATCH && (size_t)ba.mid * 9 * 64 <= WT_SCRATCH, "backward: grouped scratch too small");
                AMP_REQUIRE(st2 == 1 || st2 == 2, "backward: conv2 with stride %d", st2);
                AMP_ALLOC(t1_dec, float, (size_t)B * h1 * w1 * ba.mid);
                AMP_TRY(amp_unsplit_rows(ctx, ba.t1, (long long)B * h1 * w1, ba.mid, t1_dec));      // x is read nine times per pixel: decoded once; dy once: decoded on the load
                AMP_TRY(amp::wgrad_async_join(ctx));      // the grouped kernels use wg_scratch on the main stream
                // dy too is decoded by a pass of its own: decoding it on the load (fmt 2) costs the kernel more than the pass (55.1 against 52.8 ms
                // per X-101 step, A/B in one call; AMP_GX_DY_INKERNEL=1 for the other)
                static const bool dy_pass = getenv("AMP_GX_DY_INKERNEL") == nullptr;
                if (dy_pass) {
                    AMP_ALLOC(dt2_dec, float, (size_t)B * ba.oh * ba.ow * ba.mid);
                    AMP_TRY(amp_unsplit_rows(ctx, d_t2, (long long)B * ba.oh * ba.ow, ba.mid, dt2_dec));
                    AMP_TRY(amp_conv2d_grouped_wgrad_fmt(ctx, &dw, c2.groups, t1_dec, dt2_dec, c2.scale, wg_scratch, GW(c2), 0, 16));
                } else
                AMP_TRY(amp_conv2d_grouped_wgrad_fmt(ctx, &dw, c2.groups, t1_dec, d_t2, c2.scale, wg_scratch, GW(c2), 2, 16));
                AMP_TRY(amp_group_dgrad_weights(ctx, c2.w, c2.scale, ba.mid, 3, 3, wt_scratch));
                const float* dy2 = d_t2;
                if (st2 == 2) { AMP_TRY(amp_scatter2_rows(ctx, d_t2, d_t2_up, B, h1, w1, ba.mid)); dy2 = d_t2_up; }
                amp_conv_desc dd = dw;
                dd.stride = 1;
                dys_of = nullptr;
                AMP_TRY(amp::conv_run(ctx, &dd, c2.groups, dy2, wt_scratch, nullptr, 0, nullptr, nullptr, nullptr, ba.t1, d_t1, 0, 1 | 2 | 8));
            }
            AMP_TRY(wgrad(c1, ba.x_in, B, ba.in_h, ba.in_w, st1, 0, d_t1, false, false, 3));
            if (ba.has_sc) {
                const ConvW& cs = CONV((ba.key + ".shortcut").c_str());
                AMP_TRY(wgrad(cs, ba.x_in, B, ba.in_h, ba.in_w, ba.stride, 0, dcur, false, false, 3));
                if (need_dx) {
                    if (ba.stride != 2 || st1 != 1) { amp::set_error("backward: a ResNeXt projection block with stride %d / %d is not expected here", ba.stride, st1); return AMP_ERR_STATE; }
                    float* dprev = d_res[ba.stage - 1];             // fp32, input resolution: already holds the FPN lateral's share
                    AMP_ALLOC(tmp_sc, float, (size_t)B * ba.oh * ba.ow * ba.cin);
                    AMP_ALLOC(tmp_full, float, (size_t)B * ba.in_h * ba.in_w * ba.cin);
                    AMP_TRY(dgrad_s(cs, dcur, B, ba.oh, ba.ow, 0, nullptr, nullptr, tmp_sc));
                    AMP_TRY(dgrad_s(c1, d_t1, B, h1, w1, 0, nullptr, nullptr, tmp_full));
                    AMP_TRY(amp_accumulate_split(ctx, tmp_full, dprev, (long long)B * ba.in_h * ba.in_w, ba.cin, 16));
                    AMP_TRY(amp_subsample2_bwd_split(ctx, tmp_sc, dprev, B, ba.in_h, ba.in_w, ba.cin, 16));
                }
            } else {
                AMP_ALLOC(d_in, float, out_elems);
                const bool fuse = bi > 0 && m->blocks[bi - 1].out == ba.x_in;
                AMP_TRY(dgrad_s(c1, d_t1, B, ba.oh, ba.ow, 0, dcur, fuse ? ba.x_in : nullptr, d_in));
                dcur = d_in;
                dcur_masked = fuse;
            }
        } else {
        // d(pre-activation) = d(out) * (out > 0); inside a stage the previous iteration's input-gradient conv has applied it already
        if (!dcur_masked) AMP_TRY(AS ? amp_relu_mask_split(ctx, dcur, ba.out, out_elems, ba.cout) : amp_relu_mask(ctx, dcur, ba.out, out_elems));
        dcur_masked = false;
        AMP_TRY(wgrad(c3, ba.t2, B, ba.oh, ba.ow, 1, 0, dcur, false, false, AS));
        AMP_TRY(dgrad(c3, dcur, B, ba.oh, ba.ow, 0, nullptr, ba.t2, d_t2, AS));
        AMP_TRY(c2_backward());
        AMP_TRY(wgrad(c1, ba.x_in, B, ba.in_h, ba.in_w, st1, 0, d_t1, false, false, AS));
        if (ba.has_sc) {
            const ConvW& cs = CONV((ba.key + ".shortcut").c_str());
            AMP_TRY(wgrad(cs, ba.x_in, B, ba.in_h, ba.in_w, ba.stride, 0, dcur, false, false, AS));
            if (need_dx) {
                float* dprev = d_res[ba.stage - 1];             // already holds the FPN lateral's share
                AMP_ALLOC(tmp_sc, float, (size_t)B * ba.oh * ba.ow * ba.cin);
                AMP_TRY(dgrad(cs, dcur, B, ba.oh, ba.ow, 0, nullptr, nullptr, tmp_sc));
                if (ba.stride != 2) { amp::set_error("backward: stride-1 projection block is not expected here"); return AMP_ERR_STATE; }
                if (st1 == 2) {     // both branches at the block's output resolution: add, then spread over the input map
                    AMP_ALLOC(tmp_in, float, (size_t)B * ba.oh * ba.ow * ba.cin);
                    AMP_TRY(dgrad(c1, d_t1, B, ba.oh, ba.ow, 0, tmp_sc, nullptr, tmp_in));
                    AMP_TRY(amp_subsample2_bwd(ctx, tmp_in, dprev, B, ba.in_h, ba.in_w, ba.cin));
                } else {            // conv1 ran at the input resolution: its data gradient joins dprev there; the strided shortcut's is spread
                    AMP_ALLOC(tmp_full, float, (size_t)B * ba.in_h * ba.in_w * ba.cin);
                    AMP_TRY(dgrad(c1, d_t1, B, h1, w1, 0, dprev, nullptr, tmp_full));
                    AMP_HIP_CHECK(hipMemcpyAsync(dprev, tmp_full, (size_t)B * ba.in_h * ba.in_w * ba.cin * 4, hipMemcpyDeviceToDevice, ctx->stream));
                    AMP_TRY(amp_subsample2_bwd(ctx, tmp_sc, dprev, B, ba.in_h, ba.in_w, ba.cin));
                }
            }
        } else {
            AMP_ALLOC(d_in, float, out_elems);
            // + identity shortcut, times the ReLU mask of the block below (its output IS this block's input)
            const bool fuse = bi > 0 && m->blocks[bi - 1].out == ba.x_in;
            AMP_TRY(dgrad(c1, d_t1, B, ba.oh, ba.ow, 0, dcur, fuse ? ba.x_in : nullptr, d_in, AS));
            dcur = d_in;
            dcur_masked = fuse;
        }
        }
        if (bi == 0 || m->blocks[bi - 1].stage != ba.stage) AMP_TRY(issue_bucket(m, 7 - ba.stage));   // res5 -> 4, res4 -> 5, res3 -> 6
    }
    if (async_scope.on) { async_scope.on = false; AMP_TRY(amp::wgrad_async_end(ctx)); }
    if (!dry) {
        if (m->grad_overlap != 0) AMP_TRY(amp::comm_mark_producer_end(ctx));
        m->grads_valid = true;
    }
    return AMP_OK;
}

}  // namespace

extern "C" {

int amp_model_cfg_default(amp_model_cfg* c) {
    AMP_REQUIRE(c, "amp_model_cfg_default: null");
    memset(c, 0, sizeof(*c));
    c->num_classes = 80;
    c->pixel_mean[0] = 103.530f; c->pixel_mean[1] = 116.280f; c->pixel_mean[2] = 123.675f;
    c->pixel_std[0] = c->pixel_std[1] = c->pixel_std[2] = 1.0f;
    c->pre_nms_topk = 1000; c->post_nms_topk = 1000; c->rpn_nms_thresh = 0.7f;
    c->score_thresh = 0.05f; c->nms_thresh = 0.5f; c->detections_per_image = 100;
    c->bbox_reg_weights[0] = c->bbox_reg_weights[1] = 10.f; c->bbox_reg_weights[2] = c->bbox_reg_weights[3] = 5.f;
    c->mask_threshold = 0.5f;
    c->max_batch = 1; c->max_h = 1344; c->max_w = 1344; c->max_out_hw = 4096;
    c->rle_pool_counts = 0;
    c->train_enable = 0;
    c->pre_nms_topk_train = 2000; c->post_nms_topk_train = 1000;
    c->rpn_batch = 256; c->rpn_pos_frac = 0.5f; c->rpn_iou_lo = 0.3f; c->rpn_iou_hi = 0.7f;
    c->roi_batch = 512; c->roi_fg_frac = 0.25f; c->roi_iou = 0.5f;
    c->max_gt = 16384; c->max_poly_doubles = 16384 * 80;
    c->resnet_depth = 50; c->num_groups = 1; c->width_per_group = 64; c->stride_in_1x1 = 1;
    return AMP_OK;
}

int amp_model_create(amp_ctx* ctx, const amp_model_cfg* cfg, amp_model** out) {
    AMP_REQUIRE(ctx && cfg && out, "amp_model_create: null argument");
    AMP_REQUIRE(cfg->num_classes >= 1 && cfg->num_classes <= 255, "amp_model_create: num_classes out of range");
    AMP_REQUIRE(cfg->pre_nms_topk >= 1 && cfg->pre_nms_topk <= 2048, "amp_model_create: pre_nms_topk must be in [1,2048]");
    AMP_REQUIRE(cfg->post_nms_topk >= 1 && cfg->detections_per_image >= 1, "amp_model_create: bad topk");
    AMP_REQUIRE(cfg->max_batch >= 1 && cfg->max_h >= 32 && cfg->max_w >= 32, "amp_model_create: bad capacity");
    amp_model* m = new amp_model();
    m->ctx = ctx;
    m->cfg = *cfg;
    if (m->cfg.rle_pool_counts == 0)
        m->cfg.rle_pool_counts = (size_t)m->cfg.max_batch * m->cfg.detections_per_image * 16384;
    const int K = cfg->num_classes;
    {   // backbone variant: R50 (3,4,6,3) or R101 / X101 (3,4,23,3); ResNeXt = grouped 3x3 with the stride in it
        amp_model_cfg& c = m->cfg;
        if (c.resnet_depth == 0) { c.resnet_depth = 50; c.num_groups = 1; c.width_per_group = 64; c.stride_in_1x1 = 1; }
        const bool ok_depth = c.resnet_depth == 50 || c.resnet_depth == 101;
        const int width = c.num_groups * c.width_per_group;
        if (!ok_depth || c.num_groups < 1 || width < 64 || width % 64 != 0 ||
            (c.num_groups > 1 && !(c.width_per_group == 8 || c.width_per_group == 16 || c.width_per_group == 32 || c.width_per_group == 64))) {
            amp::set_error("amp_model_create: unsupported backbone (depth %d, groups %d x width %d)", c.resnet_depth, c.num_groups, c.width_per_group);
            delete m;
            return AMP_ERR_ARG;
        }
        m->nblk[2] = (c.resnet_depth == 101) ? 23 : 6;
        for (int s_ = 0; s_ < 4; ++s_) m->mid[s_] = width << s_;
    }
    build_expected(m);
    // parameter arena: every tensor (R50-FPN ~44.5 M floats) + the 64-wide window copies of grouped 3x3 weights + padding
    {
        size_t n = (size_t)27 * 1000 * 1000 + (size_t)K * 6 * 1024 + 65536;   // stem, FPN, RPN, heads
        int cin = 64;
        for (int s_ = 0; s_ < 4; ++s_)
            for (int b_ = 0; b_ < m->nblk[s_]; ++b_) {
                const size_t mid = m->mid[s_], cout = kOut[s_];
                n += (size_t)cin * mid + mid * 9 * (m->cfg.num_groups > 1 ? 64 : mid) + mid * cout + (b_ == 0 ? (size_t)cin * cout : 0) + 8 * (mid + cout) + 1024;
                cin = (int)cout;
            }
        m->parena_floats = n;
    }
    if (hipMalloc(&m->parena, m->parena_floats * sizeof(float)) != hipSuccess) {
        amp::set_error("amp_model_create: hipMalloc of the parameter arena failed");
        delete m;
        return AMP_ERR_HIP;
    }
    // workspace: dry-run the plan at the maximum shape
    m->ws.dry = true;
    std::vector<int> oh(cfg->max_batch, cfg->max_out_hw), ow(cfg->max_batch, cfg->max_out_hw);
    int st = run(m, nullptr, cfg->max_batch, cfg->max_h, cfg->max_w, oh.data(), ow.data());
    if (st == AMP_OK && cfg->train_enable) {
        float dummy[5];
        st = run_train(m, nullptr, cfg->max_batch, cfg->max_h, cfg->max_w, nullptr, 0, dummy, true);
    }
    if (st != AMP_OK) { (void)hipFree(m->parena); delete m; return st; }
    m->ws.dry = false;
    m->ws.cap = m->ws.peak + (1 << 20);
    if (hipMalloc(&m->ws.base, m->ws.cap) != hipSuccess) {
        amp::set_error("amp_model_create: hipMalloc of the %zu-byte workspace failed", m->ws.cap);
        (void)hipFree(m->parena);
        delete m;
        return AMP_ERR_HIP;
    }
    if (cfg->train_enable) {
        if (hipMalloc(&m->garena, m->parena_floats * sizeof(float)) != hipSuccess || hipMalloc(&m->varena, m->parena_floats * sizeof(float)) != hipSuccess) {
            amp::set_error("amp_model_create: hipMalloc of the gradient / momentum arenas failed");
            (void)hipFree(m->parena); (void)hipFree(m->ws.base);
            delete m;
            return AMP_ERR_HIP;
        }
        (void)hipMemset(m->garena, 0, m->parena_floats * sizeof(float));
        (void)hipMemset(m->varena, 0, m->parena_floats * sizeof(float));
    }
    const int R = cfg->max_batch * cfg->post_nms_topk;
    std::vector<int> iota(R);
    for (int i = 0; i < R; ++i) iota[i] = i / cfg->post_nms_topk;
    m->h_res_bytes = (size_t)cfg->max_batch * cfg->detections_per_image * 64 + 20 * 256;
    // every small device / pinned allocation is checked: under memory pressure (two contexts per GPU, a re-created training net) a
    // failed one must surface as AMP_ERR_NOMEM here, not as a fault in the first call that touches the null pointer
    bool ok = hipMalloc(&m->d_batch_iota, (size_t)R * sizeof(int)) == hipSuccess &&
              hipMemcpy(m->d_batch_iota, iota.data(), (size_t)R * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
              hipMalloc(&m->d_flags, 4 * sizeof(int)) == hipSuccess && hipMemset(m->d_flags, 0, 4 * sizeof(int)) == hipSuccess &&
              hipHostMalloc(&m->h_counts, (size_t)(cfg->max_batch + 8) * sizeof(int)) == hipSuccess &&
              hipHostMalloc(&m->h_res, m->h_res_bytes) == hipSuccess &&
              hipHostMalloc(&m->h_small, (size_t)(2 * cfg->max_batch + cfg->max_batch * cfg->detections_per_image + 8) * sizeof(int)) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        amp::set_error("amp_model_create: out of memory allocating the index / flag / pinned staging buffers");
        amp_model_destroy(m);
        return AMP_ERR_NOMEM;
    }
    // run-length staging: pinned for the DMA, non-coherent = cacheable for the host that reads it after the stream sync (optional:
    // without it the results go through a pageable vector)
    m->h_pool_counts = (size_t)std::min<unsigned long long>(m->cfg.rle_pool_counts, (unsigned long long)64 << 20);
    if (hipHostMalloc(reinterpret_cast<void**>(&m->h_pool), m->h_pool_counts * 4, hipHostMallocNonCoherent) != hipSuccess) { m->h_pool = nullptr; m->h_pool_counts = 0; (void)hipGetLastError(); }
    *out = m;
    return AMP_OK;
}

void amp_model_destroy(amp_model* m) {
    if (!m) return;
    (void)hipFree(m->parena);
    (void)hipFree(m->garena);
    (void)hipFree(m->varena);
    (void)hipFree(m->sgd_chunks);
    (void)hipFree(m->fwd_jobs.jobs); (void)hipFree(m->fwd_jobs.chunks); (void)hipFree(m->dgrad_jobs.jobs); (void)hipFree(m->dgrad_jobs.chunks);
    (void)hipFree(m->dgrad_arena);
    (void)hipFree(m->split_arena);
    (void)hipFree(m->img_stage);
    if (m->bm_scratch) (void)hipFree(m->bm_scratch);
    if (m->bm_flag) (void)hipFree(m->bm_flag);
    (void)hipFree(m->ws.base);
    (void)hipFree(m->d_batch_iota);
    (void)hipFree(m->d_flags);
    (void)hipHostFree(m->h_counts);
    (void)hipHostFree(m->h_small);
    (void)hipHostFree(m->h_res);
    if (m->h_pool) (void)hipHostFree(m->h_pool);
    if (m->h_str) (void)hipHostFree(m->h_str);
    if (m->ev_counts) (void)hipEventDestroy(m->ev_counts);
    delete m;
}

size_t amp_model_workspace_bytes(amp_model* m) { return m ? m->ws.cap : 0; }

int amp_model_set_rle_output(amp_model* m, int mode) {
    AMP_REQUIRE(m && mode >= 0 && mode <= 2, "amp_model_set_rle_output: mode is AMP_RLE_COUNTS (0), AMP_RLE_STRINGS (1) or AMP_RLE_BOTH (2)");
    if (mode != 0 && !m->h_str) {
        m->h_str_bytes = (size_t)std::min<unsigned long long>((unsigned long long)m->cfg.rle_pool_counts * 4, (unsigned long long)256 << 20);
        if (hipHostMalloc(reinterpret_cast<void**>(&m->h_str), m->h_str_bytes, hipHostMallocNonCoherent) != hipSuccess) {
            m->h_str = nullptr; m->h_str_bytes = 0; (void)hipGetLastError();
            amp::set_error("amp_model_set_rle_output: cannot pin %zu bytes for the counts strings", m->h_str_bytes);
            return AMP_ERR_NOMEM;
        }
    }
    m->rle_mode = mode;
    return AMP_OK;
}

int amp_model_num_tensors(amp_model* m) { return m ? (int)m->expected.size() : 0; }
const char* amp_model_tensor_name(amp_model* m, int i) {
    return (m && i >= 0 && i < (int)m->expected.size()) ? m->expected[i].c_str() : nullptr;
}

int amp_model_load_tensor(amp_model* m, const char* name_c, const float* data, const long long* shape, int ndim) {
    AMP_REQUIRE(m && name_c && data && shape && ndim >= 1 && ndim <= 4, "amp_model_load_tensor: bad argument");
    const std::string name(name_c);
    bool known = false;
    for (auto& e : m->expected) if (e == name) { known = true; break; }
    AMP_REQUIRE(known, "amp_model_load_tensor: unknown tensor '%s'", name_c);
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    const int K = m->cfg.num_classes;
    const std::string prefix = name.substr(0, name.rfind('.'));
    const bool is_w = ends_with(name, ".weight");

    auto put_conv = [&](const std::string& key, std::vector<float>&& v, int cout, int cin, int kh, int kw) -> int {
        ConvW& cw = m->conv[key];
        if (!cw.w) {
            cw.w = palloc(m, v.size());
            AMP_REQUIRE(cw.w, "amp_model_load_tensor: parameter arena exhausted at %s", name_c);
        }
        cw.cout = cout; cw.cin = cin; cw.kh = kh; cw.kw = kw;
        float mx = 0.f;
        for (float f : v) mx = std::max(mx, std::fabs(f));
        cw.w_absmax = mx;
        return upload(m, cw.w, v);
    };
    auto put_shift = [&](const std::string& key, const std::vector<float>& v) -> int {
        ConvW& cw = m->conv[key];
        if (!cw.shift) {
            cw.shift = palloc(m, v.size());
            AMP_REQUIRE(cw.shift, "amp_model_load_tensor: parameter arena exhausted at %s", name_c);
        }
        return upload(m, cw.shift, v);
    };

    if (name.find(".norm.") != std::string::npos) {
        m->host_raw[name].assign(data, data + numel);           // folded at finalize
    } else if (name == "backbone.bottom_up.stem.conv1.weight") {
        AMP_REQUIRE(ndim == 4 && shape[0] == 64 && shape[1] == 3 && shape[2] == 7 && shape[3] == 7, "%s: expected [64,3,7,7]", name_c);
        AMP_TRY(put_conv(prefix, oihw_to_ohwi(data, 64, 3, 7, 7, 4, 8), 64, 4, 7, 8));
    } else if (prefix == "roi_heads.box_head.fc1") {
        if (is_w) {
            AMP_REQUIRE(ndim == 2 && shape[0] == 1024 && shape[1] == 12544, "%s: expected [1024,12544]", name_c);
            std::vector<float> v(numel);
            for (int o = 0; o < 1024; ++o)
                for (int ch = 0; ch < 256; ++ch)
                    for (int p = 0; p < 49; ++p) v[(size_t)o * 12544 + p * 256 + ch] = data[(size_t)o * 12544 + ch * 49 + p];
            AMP_TRY(put_conv(prefix, std::move(v), 1024, 12544, 1, 1));
        } else {
            AMP_REQUIRE(numel == 1024, "%s: expected [1024]", name_c);
            AMP_TRY(put_shift(prefix, std::vector<float>(data, data + numel)));
        }
    } else if (prefix == "roi_heads.box_head.fc2") {
        if (is_w) {
            AMP_REQUIRE(ndim == 2 && shape[0] == 1024 && shape[1] == 1024, "%s: expected [1024,1024]", name_c);
            AMP_TRY(put_conv(prefix, std::vector<float>(data, data + numel), 1024, 1024, 1, 1));
        } else {
            AMP_TRY(put_shift(prefix, std::vector<float>(data, data + numel)));
        }
    } else if (prefix == "roi_heads.box_predictor.cls_score" || prefix == "roi_heads.box_predictor.bbox_pred" ||
               prefix == "proposal_generator.rpn_head.objectness_logits" || prefix == "proposal_generator.rpn_head.anchor_deltas") {
        const bool first = prefix.find("cls_score") != std::string::npos || prefix.find("objectness") != std::string::npos;
        const bool box = prefix.find("box_predictor") != std::string::npos;
        const size_t rows = box ? (first ? K + 1 : 4 * K) : (first ? 3 : 12);
        const size_t cols = box ? 1024 : 256;
        AMP_REQUIRE(numel == rows * (is_w ? cols : 1), "%s: unexpected size %zu", name_c, numel);
        m->host_raw[name].assign(data, data + numel);           // fused at finalize
    } else if (prefix == "roi_heads.mask_head.deconv") {
        if (is_w) {
            AMP_REQUIRE(ndim == 4 && shape[0] == 256 && shape[1] == 256 && shape[2] == 2 && shape[3] == 2, "%s: expected [256,256,2,2]", name_c);
            std::vector<float> v(numel);
            for (int ci = 0; ci < 256; ++ci)
                for (int co = 0; co < 256; ++co)
                    for (int ky = 0; ky < 2; ++ky)
                        for (int kx = 0; kx < 2; ++kx)
                            v[((size_t)(ky * 2 + kx) * 256 + co) * 256 + ci] = data[(((size_t)ci * 256 + co) * 2 + ky) * 2 + kx];
            AMP_TRY(put_conv(prefix, std::move(v), 1024, 256, 1, 1));
        } else {
            AMP_REQUIRE(numel == 256, "%s: expected [256]", name_c);
            std::vector<float> v(1024);
            for (int q = 0; q < 4; ++q) for (int co = 0; co < 256; ++co) v[q * 256 + co] = data[co];
            AMP_TRY(put_shift(prefix, v));
        }
    } else if (prefix == "roi_heads.mask_head.predictor") {
        const int Kp = (K + 3) / 4 * 4;
        if (is_w) {
            AMP_REQUIRE(numel == (size_t)K * 256, "%s: expected [%d,256,1,1]", name_c, K);
            std::vector<float> v((size_t)Kp * 256, 0.f);
            memcpy(v.data(), data, numel * 4);
            AMP_TRY(put_conv(prefix, std::move(v), Kp, 256, 1, 1));
        } else {
            AMP_REQUIRE(numel == (size_t)K, "%s: expected [%d]", name_c, K);
            std::vector<float> v(Kp, 0.f);
            memcpy(v.data(), data, numel * 4);
            AMP_TRY(put_shift(prefix, v));
        }
    } else if (is_w) {
        AMP_REQUIRE(ndim == 4, "%s: expected a 4-d conv weight", name_c);
        const int O = (int)shape[0], I = (int)shape[1], KH = (int)shape[2], KW = (int)shape[3];
        AMP_REQUIRE(I % 4 == 0, "%s: Cin %% 4 != 0", name_c);
        const int G = m->cfg.num_groups;
        if (G > 1 && ends_with(prefix, ".conv2") && prefix.rfind("backbone.bottom_up.res", 0) == 0) {
            // grouped 3x3 [O, O/G, 3, 3] -> block-diagonal window layout [O][KH][KW][64] (see amp_group_expand_weights)
            const int cpg = I;
            AMP_REQUIRE(O == cpg * G && O % 64 == 0 && 64 % cpg == 0, "%s: expected [%d,%d,3,3] with %d groups", name_c, O, O / G, G);
            std::vector<float> v((size_t)O * KH * KW * 64, 0.f);
            for (int o = 0; o < O; ++o) {
                const int j0 = (o / cpg) * cpg - (o & ~63);          // window slot of the group's first channel
                for (int i = 0; i < cpg; ++i)
                    for (int y = 0; y < KH; ++y)
                        for (int x = 0; x < KW; ++x)
                            v[(((size_t)o * KH + y) * KW + x) * 64 + j0 + i] = data[(((size_t)o * I + i) * KH + y) * KW + x];
            }
            AMP_TRY(put_conv(prefix, std::move(v), O, O, KH, KW));
            m->conv[prefix].groups = G;
        } else {
            AMP_TRY(put_conv(prefix, oihw_to_ohwi(data, O, I, KH, KW, I, KW), O, I, KH, KW));
        }
    } else {
        AMP_TRY(put_shift(prefix, std::vector<float>(data, data + numel)));
    }
    m->loaded[name] = true;
    m->finalized = false;
    return AMP_OK;
}

namespace {
// AMP_CONV_F16X3 operand copies (hi|lo f16 halves, x 2^8): every dense layer with Cin % 32 == 0 whose weights fit the fp16 range.
// Re-made after the weights change (amp_model_sgd_step marks them stale).
// device tables of a weight_jobs_kernel launch (common.h WeightJob): built once per set of pointers, one launch per refresh
int upload_weight_jobs(amp_model::JobTable& t, const std::vector<amp::WeightJob>& jobs) {
    std::vector<unsigned int> ch;      // uint2 {job, first pair}
    for (size_t j = 0; j < jobs.size(); ++j) {
        if (jobs[j].transpose == 2) {     // one chunk per 64 x 64 tile of a tap
            const size_t ntiles = (size_t)jobs[j].KH * jobs[j].KW * (jobs[j].N / 64) * (jobs[j].C / 64);
            for (size_t t = 0; t < ntiles; ++t) { ch.push_back((unsigned int)j); ch.push_back((unsigned int)t); }
            continue;
        }
        const size_t npairs = (size_t)jobs[j].N * jobs[j].KH * jobs[j].KW * jobs[j].C / 2;
        for (size_t p0 = 0; p0 < npairs; p0 += 8192) { ch.push_back((unsigned int)j); ch.push_back((unsigned int)p0); }
    }
    if (t.jobs) (void)hipFree(t.jobs);
    if (t.chunks) (void)hipFree(t.chunks);
    t.jobs = nullptr; t.chunks = nullptr; t.nchunks = 0;
    if (jobs.empty()) return AMP_OK;
    AMP_HIP_CHECK(hipMalloc(&t.jobs, jobs.size() * sizeof(amp::WeightJob)));
    AMP_HIP_CHECK(hipMalloc(&t.chunks, ch.size() * sizeof(unsigned int)));
    AMP_HIP_CHECK(hipMemcpy(t.jobs, jobs.data(), jobs.size() * sizeof(amp::WeightJob), hipMemcpyHostToDevice));
    AMP_HIP_CHECK(hipMemcpy(t.chunks, ch.data(), ch.size() * sizeof(unsigned int), hipMemcpyHostToDevice));
    t.nchunks = (int)(ch.size() / 2);
    return AMP_OK;
}

int refresh_split_weights(amp_model* m) {
    size_t need = 0;
    for (auto& kv : m->conv) {
        const ConvW& cw = kv.second;
        if ((cw.cin % 32 == 0 || (cw.cin == 4 && cw.kw == 8)) && cw.w_absmax < 60000.f) need += ((size_t)cw.cout * cw.kh * cw.kw * (cw.groups > 1 ? 64 : cw.cin) + 63) & ~(size_t)63;
    }
    if (m->split_floats < need) {
        AMP_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
        if (m->split_arena) AMP_HIP_CHECK(hipFree(m->split_arena));
        m->split_arena = nullptr; m->split_floats = 0;
        AMP_HIP_CHECK(hipMalloc(&m->split_arena, need * sizeof(float)));
        m->split_floats = need;
        m->fwd_jobs_dirty = true;
    }
    if (m->fwd_jobs_dirty) {
        std::vector<amp::WeightJob> jobs;
        size_t off = 0;
        for (auto& kv : m->conv) {
            ConvW& cw = kv.second;
            cw.w_split = nullptr;
            if (!((cw.cin % 32 == 0 || (cw.cin == 4 && cw.kw == 8)) && cw.w_absmax < 60000.f)) continue;
            const int kin = cw.groups > 1 ? 64 : cw.cin;     // grouped 3x3: window layout [Cout][KH][KW][64]
            const size_t n = (size_t)cw.cout * cw.kh * cw.kw * kin;
            cw.w_split = m->split_arena + off;
            off += (n + 63) & ~(size_t)63;
            jobs.push_back(amp::WeightJob{cw.w, nullptr, reinterpret_cast<unsigned int*>(cw.w_split), cw.cout, cw.kh, cw.kw, kin, 0});
        }
        AMP_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));     // the old tables may still be read by a launch in flight
        AMP_TRY(upload_weight_jobs(m->fwd_jobs, jobs));
        m->fwd_jobs_dirty = false;
    }
    AMP_TRY(amp::weight_jobs_run(m->ctx, m->fwd_jobs.jobs, m->fwd_jobs.chunks, m->fwd_jobs.nchunks));
    m->split_stale = false;
    return AMP_OK;
}

// AMP_CONV_F16X3 training: the data-gradient form of every weight the backward pass convolves with (flipped, transposed, FrozenBN scale
// folded in, split) -- once per step in one launch instead of a transpose and a split in front of each of its 63 data-gradient convs.
int refresh_dgrad_weights(amp_model* m) {
    auto takes = [](const std::string& key, const ConvW& cw) {
        if (key.rfind("backbone.bottom_up.stem", 0) == 0 || key.rfind("backbone.bottom_up.res2", 0) == 0) return false;
        return cw.groups == 1 && cw.cout % 32 == 0 && cw.cin % 4 == 0;
    };
    size_t need = 0;
    for (auto& kv : m->conv)
        if (takes(kv.first, kv.second)) need += ((size_t)kv.second.cout * kv.second.kh * kv.second.kw * kv.second.cin + 63) & ~(size_t)63;
    if (m->dgrad_floats < need) {
        AMP_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
        if (m->dgrad_arena) AMP_HIP_CHECK(hipFree(m->dgrad_arena));
        m->dgrad_arena = nullptr; m->dgrad_floats = 0;
        AMP_HIP_CHECK(hipMalloc(&m->dgrad_arena, need * sizeof(float)));
        m->dgrad_floats = need;
        m->dgrad_jobs_dirty = true;
    }
    if (m->dgrad_jobs_dirty) {
        std::vector<amp::WeightJob> jobs;
        size_t off = 0;
        for (auto& kv : m->conv) {
            ConvW& cw = kv.second;
            cw.wt_split = nullptr;
            if (!takes(kv.first, cw)) continue;
            const size_t n = (size_t)cw.cout * cw.kh * cw.kw * cw.cin;
            cw.wt_split = m->dgrad_arena + off;
            off += (n + 63) & ~(size_t)63;
            // the mask head's deconv is stored as a 1x1 layer with 4 x 256 outputs: same transform
            jobs.push_back(amp::WeightJob{cw.w, cw.scale, reinterpret_cast<unsigned int*>(cw.wt_split), cw.cout, cw.kh, cw.kw, cw.cin,
                                          (cw.cout % 64 == 0 && cw.cin % 64 == 0) ? 2 : 1});
        }
        AMP_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
        AMP_TRY(upload_weight_jobs(m->dgrad_jobs, jobs));
        m->dgrad_jobs_dirty = false;
    }
    return amp::weight_jobs_run(m->ctx, m->dgrad_jobs.jobs, m->dgrad_jobs.chunks, m->dgrad_jobs.nchunks);
}

}  // namespace

int amp_model_finalize(amp_model* m) {
    AMP_REQUIRE(m, "amp_model_finalize: null");
    for (auto& e : m->expected)
        AMP_REQUIRE(m->loaded.count(e), "amp_model_finalize: tensor '%s' was never loaded", e.c_str());
    const int K = m->cfg.num_classes;
    // FrozenBN -> scale = w * rsqrt(var + eps), shift = b - mean * scale (detectron2 FrozenBatchNorm2d, fp32)
    for (auto& kv : m->conv) {
        const std::string& p = kv.first;
        auto it = m->host_raw.find(p + ".norm.weight");
        if (it == m->host_raw.end()) continue;
        const auto& g = it->second;
        const auto& be = m->host_raw.at(p + ".norm.bias");
        const auto& mu = m->host_raw.at(p + ".norm.running_mean");
        const auto& var = m->host_raw.at(p + ".norm.running_var");
        const size_t n = g.size();
        AMP_REQUIRE((int)n == kv.second.cout && be.size() == n && mu.size() == n && var.size() == n, "amp_model_finalize: BN size mismatch at %s", p.c_str());
        std::vector<float> sc(n), sh(n);
        for (size_t i = 0; i < n; ++i) {
            sc[i] = g[i] * (1.0f / sqrtf(var[i] + 1e-5f));
            sh[i] = be[i] - mu[i] * sc[i];
        }
        ConvW& cw = kv.second;
        if (!cw.scale) { cw.scale = palloc(m, n); AMP_REQUIRE(cw.scale, "amp_model_finalize: parameter arena exhausted"); }
        if (!cw.shift) { cw.shift = palloc(m, n); AMP_REQUIRE(cw.shift, "amp_model_finalize: parameter arena exhausted"); }
        AMP_TRY(upload(m, cw.scale, sc));
        AMP_TRY(upload(m, cw.shift, sh));
    }
    auto fuse = [&](const char* key, const char* a, const char* b, int ra, int rb, int cols) -> int {
        const int rp = (ra + rb + 3) / 4 * 4;   // rows padded to a multiple of 4 (zero rows): 16-byte output rows
        std::vector<float> w((size_t)rp * cols, 0.f), bias(rp, 0.f);
        const auto& wa = m->host_raw.at(std::string(a) + ".weight");
        const auto& wb = m->host_raw.at(std::string(b) + ".weight");
        const auto& ba = m->host_raw.at(std::string(a) + ".bias");
        const auto& bb = m->host_raw.at(std::string(b) + ".bias");
        memcpy(w.data(), wa.data(), wa.size() * 4);
        memcpy(w.data() + wa.size(), wb.data(), wb.size() * 4);
        memcpy(bias.data(), ba.data(), ba.size() * 4);
        memcpy(bias.data() + ba.size(), bb.data(), bb.size() * 4);
        ConvW& cw = m->conv[key];
        if (!cw.w) { cw.w = palloc(m, w.size()); cw.shift = palloc(m, bias.size()); }
        AMP_REQUIRE(cw.w && cw.shift, "amp_model_finalize: parameter arena exhausted");
        cw.cout = rp; cw.cin = cols; cw.kh = cw.kw = 1;
        cw.w_absmax = 0.f;
        for (float f : w) cw.w_absmax = std::max(cw.w_absmax, std::fabs(f));
        AMP_TRY(upload(m, cw.w, w));
        return upload(m, cw.shift, bias);
    };
    AMP_TRY(fuse("proposal_generator.rpn_head.pred", "proposal_generator.rpn_head.objectness_logits",
                 "proposal_generator.rpn_head.anchor_deltas", 3, 12, 256));
    AMP_TRY(fuse("roi_heads.box_predictor", "roi_heads.box_predictor.cls_score", "roi_heads.box_predictor.bbox_pred", K + 1, 4 * K, 1024));
    m->fwd_jobs_dirty = m->dgrad_jobs_dirty = true;      // eligibility (w_absmax) and the FrozenBN scale pointers are settled here
    AMP_TRY(refresh_split_weights(m));
    // trainable tensors: every conv / fc weight and true bias outside the frozen stem + res2 (FREEZE_AT = 2); FrozenBN has none
    m->trainable.clear();
    if (m->sgd_chunks) { (void)hipFree(m->sgd_chunks); m->sgd_chunks = nullptr; m->sgd_nchunks = 0; }   // rebuilt by the next amp_model_sgd_step
    for (auto& kv : m->conv) {
        const std::string& key = kv.first;
        if (key.rfind("backbone.bottom_up.stem", 0) == 0 || key.rfind("backbone.bottom_up.res2", 0) == 0) continue;
        const ConvW& cw = kv.second;
        m->trainable.push_back({cw.w, (size_t)cw.cout * cw.kh * cw.kw * (cw.groups > 1 ? 64 : cw.cin)});
        const bool has_bn = m->host_raw.count(key + ".norm.weight") != 0;
        if (!has_bn && cw.shift) m->trainable.push_back({cw.shift, (size_t)cw.cout});
    }
    {   // gradient-exchange plan: one entry per trainable tensor -> merged ranges per bucket
        std::vector<int> tb;
        std::vector<size_t> to, tn;
        for (auto& kv : m->conv) {
            const int b = amp_grad_bucket_of(kv.first.c_str());
            if (b < 0) continue;
            const ConvW& cw = kv.second;
            tb.push_back(b); to.push_back((size_t)(cw.w - m->parena)); tn.push_back((size_t)cw.cout * cw.kh * cw.kw * (cw.groups > 1 ? 64 : cw.cin));
            if (!m->host_raw.count(kv.first + ".norm.weight") && cw.shift) { tb.push_back(b); to.push_back((size_t)(cw.shift - m->parena)); tn.push_back((size_t)cw.cout); }
        }
        const int cap = 64 * AMP_GRAD_BUCKETS;
        m->gb_bucket.assign(cap, 0); m->gb_off.assign(cap, 0); m->gb_n.assign(cap, 0);
        int cnt = 0;
        // gaps bridged: the 64-float alignment padding and the FrozenBN scale / shift vectors between two weights (gradient 0)
        AMP_TRY(amp_plan_grad_buckets((int)tb.size(), tb.data(), to.data(), tn.data(), 64, cap, m->gb_bucket.data(), m->gb_off.data(), m->gb_n.data(), &cnt));
        m->gb_bucket.resize(cnt); m->gb_off.resize(cnt); m->gb_n.resize(cnt);
    }
    m->finalized = true;
    return AMP_OK;
}

int amp_model_grad_buckets(amp_model* m, int cap, int* out_bucket, size_t* out_off, size_t* out_n, int* out_count) {
    AMP_REQUIRE(m && out_bucket && out_off && out_n && out_count, "amp_model_grad_buckets: null argument");
    AMP_REQUIRE(m->finalized, "amp_model_grad_buckets: call amp_model_finalize first");
    AMP_REQUIRE((int)m->gb_bucket.size() <= cap, "amp_model_grad_buckets: %zu ranges, capacity %d", m->gb_bucket.size(), cap);
    for (size_t i = 0; i < m->gb_bucket.size(); ++i) { out_bucket[i] = m->gb_bucket[i]; out_off[i] = m->gb_off[i]; out_n[i] = m->gb_n[i]; }
    *out_count = (int)m->gb_bucket.size();
    return AMP_OK;
}

int amp_model_set_grad_overlap(amp_model* m, int mode) {
    AMP_REQUIRE(m && (mode == 0 || mode == 1), "amp_model_set_grad_overlap: bad argument");
    m->grad_overlap = mode;
    return AMP_OK;
}

int amp_model_grads_exchanged(amp_model* m, int* exchanged) {
    AMP_REQUIRE(m && exchanged, "amp_model_grads_exchanged: null argument");
    *exchanged = m->grads_valid && m->issued_mask == (1u << AMP_GRAD_BUCKETS) - 1u;
    return AMP_OK;
}

int amp_model_broadcast_params(amp_model* m, int root) {
    AMP_REQUIRE(m && m->finalized && m->ctx->comm, "amp_model_broadcast_params: needs a finalized model and a communicator on the context");
    // parameters (conv weights, FrozenBN scale / shift, biases: the whole arena) and, when training, the SGD momentum: what DDP's
    // constructor does with rank 0's state_dict.  The f16x3 operand copies are derived data: refreshed before their next use.
    AMP_TRY(amp_comm_broadcast(m->ctx, m->parena, m->parena_used * sizeof(float), root));
    if (m->varena) AMP_TRY(amp_comm_broadcast(m->ctx, m->varena, m->parena_used * sizeof(float), root));
    m->split_stale = true;
    return AMP_OK;
}

int amp_model_allreduce_grads(amp_model* m) {
    AMP_REQUIRE(m && m->garena && m->ctx->comm, "amp_model_allreduce_grads: needs cfg.train_enable and a communicator on the context");
    AMP_REQUIRE(m->grads_valid, "amp_model_allreduce_grads: no gradients (call amp_model_forward_backward first)");
    AMP_REQUIRE(m->issued_mask == 0, "amp_model_allreduce_grads: these gradients were already exchanged (buckets 0x%x); a second SUM would count them twice", m->issued_mask);
    const int keep = m->grad_overlap;
    m->grad_overlap = 1;
    int st = AMP_OK;
    for (int b = 0; b < AMP_GRAD_BUCKETS && st == AMP_OK; ++b) st = issue_bucket(m, b);
    if (st == AMP_OK) st = amp::comm_mark_producer_end(m->ctx);
    m->grad_overlap = keep;
    return st;
}

static int stage_images(amp_model* m, const uint8_t* host, size_t bytes) {
    if (m->img_stage_bytes < bytes) {
        AMP_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
        if (m->img_stage) AMP_HIP_CHECK(hipFree(m->img_stage));
        m->img_stage = nullptr; m->img_stage_bytes = 0;
        AMP_HIP_CHECK(hipMalloc(&m->img_stage, bytes));
        m->img_stage_bytes = bytes;
    }
    AMP_HIP_CHECK(hipMemcpyAsync(m->img_stage, host, bytes, hipMemcpyHostToDevice, m->ctx->stream));
    return AMP_OK;
}

int amp_model_infer(amp_model* m, const uint8_t* imgs_bgr, int imgs_on_host, int B, int H, int W, const int* out_h_h,
                    const int* out_w_h, amp_dets* out) {
    AMP_REQUIRE(m && imgs_bgr && out, "amp_model_infer: null argument");
    AMP_REQUIRE(m->finalized, "amp_model_infer: call amp_model_finalize after loading every tensor");
    const int Hp = (H + 31) / 32 * 32, Wp = (W + 31) / 32 * 32;
    AMP_REQUIRE(B >= 1 && B <= m->cfg.max_batch && H >= 1 && W >= 1 && Hp <= m->cfg.max_h && Wp <= m->cfg.max_w &&
                (size_t)Hp * Wp <= (size_t)m->cfg.max_h * m->cfg.max_w,
                "amp_model_infer: batch %dx%dx%d exceeds the capacity the model was created with (%dx%dx%d)", B, H, W,
                m->cfg.max_batch, m->cfg.max_h, m->cfg.max_w);
    std::vector<int> oh(B, H), ow(B, W);
    if ((int)m->img_hw.size() == 2 * B)      // differently sized images in one frame: the default output size is each image's own
        for (int b = 0; b < B; ++b) { oh[b] = m->img_hw[2 * b]; ow[b] = m->img_hw[2 * b + 1]; }
    if (out_h_h && out_w_h) { oh.assign(out_h_h, out_h_h + B); ow.assign(out_w_h, out_w_h + B); }
    AMP_HIP_CHECK(hipSetDevice(m->ctx->device));
    AMP_HIP_CHECK(hipMemsetAsync(m->d_flags, 0, 4 * sizeof(int), m->ctx->stream));
    const uint8_t* imgs_d = imgs_bgr;
    if (imgs_on_host) {
        AMP_TRY(stage_images(m, imgs_bgr, (size_t)B * H * W * 3));
        imgs_d = m->img_stage;
    }
    if (m->split_stale && m->ctx->conv_mode == AMP_CONV_F16X3) AMP_TRY(refresh_split_weights(m));
    int st = run(m, imgs_d, B, H, W, oh.data(), ow.data());
    if (m->ctx->conv_mode == AMP_CONV_F16X3) {   // an operand left the fp16 range of the split arithmetic: the whole batch again on fp32 MFMA
        int flag = 0;
        if (amp_conv_range_flag(m->ctx, 1, &flag) == AMP_OK && flag) {
            if (m->f32_reruns++ == 0)
                fprintf(stderr, "[ampis_hip] an activation exceeded the fp16 range of AMP_CONV_F16X3; re-running the batch in AMP_CONV_F32\n");
            m->ctx->conv_mode = AMP_CONV_F32;
            AMP_HIP_CHECK(hipMemsetAsync(m->d_flags, 0, 4 * sizeof(int), m->ctx->stream));
            st = run(m, imgs_d, B, H, W, oh.data(), ow.data());
            m->ctx->conv_mode = AMP_CONV_F16X3;
        }
    }
    if (st != AMP_OK) return st;
    out->B = B;
    out->D = m->cfg.detections_per_image;
    out->n = m->r_n.data();
    out->boxes = m->r_boxes.data();
    out->scores = m->r_scores.data();
    out->classes = m->r_classes.data();
    out->rle_off = m->r_rle_off.data();
    out->rle_len = m->r_rle_len.data();
    out->rle_counts = m->r_pool_ptr;
    out->out_h = m->r_out_h.data();
    out->out_w = m->r_out_w.data();
    out->rle_str = m->rle_mode ? m->h_str : nullptr;
    out->rle_str_off = m->rle_mode ? m->r_str_off.data() : nullptr;
    out->rle_str_len = m->rle_mode ? m->r_str_len.data() : nullptr;
    return AMP_OK;
}

static int train_entry(amp_model* m, const uint8_t* imgs_bgr, int imgs_on_host, int B, int H, int W, const amp_gt* gt,
                       unsigned int seed, float losses_h[5], int backward);

int amp_model_forward_losses(amp_model* m, const uint8_t* imgs_bgr, int imgs_on_host, int B, int H, int W, const amp_gt* gt,
                             unsigned int seed, float losses_h[5]) {
    return train_entry(m, imgs_bgr, imgs_on_host, B, H, W, gt, seed, losses_h, 0);
}

int amp_model_forward_backward(amp_model* m, const uint8_t* imgs_bgr, int imgs_on_host, int B, int H, int W, const amp_gt* gt,
                               unsigned int seed, float losses_h[5]) {
    return train_entry(m, imgs_bgr, imgs_on_host, B, H, W, gt, seed, losses_h, 1);
}

int amp_model_sgd_step(amp_model* m, float lr, float momentum, float weight_decay, float grad_scale) {
    AMP_REQUIRE(m && m->garena && m->varena, "amp_model_sgd_step: the model was created without cfg.train_enable");
    AMP_REQUIRE(m->grads_valid, "amp_model_sgd_step: no gradients (call amp_model_forward_backward first)");
    AMP_TRY(amp::comm_wait_done(m->ctx));    // the gradient exchange (if any) completes before the first update kernel, on the device
    if (!m->sgd_chunks) {     // one launch for every trainable tensor: chunks of <= 16384 floats of the arena (tensors start 64-float aligned)
        std::vector<unsigned long long> ch;
        for (auto& t : m->trainable)
            for (size_t o = 0; o < t.n; o += 16384) {
                const size_t off = (size_t)(t.p - m->parena) + o, n = std::min<size_t>(16384, t.n - o);
                AMP_REQUIRE(off % 4 == 0 && off < (1ull << 32), "amp_model_sgd_step: arena offset out of range");
                ch.push_back((unsigned long long)off | ((unsigned long long)n << 32));
            }
        m->sgd_nchunks = (int)ch.size();
        if (!ch.empty()) {
            AMP_HIP_CHECK(hipMalloc(&m->sgd_chunks, ch.size() * sizeof(unsigned long long)));
            AMP_HIP_CHECK(hipMemcpy(m->sgd_chunks, ch.data(), ch.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        }
    }
    AMP_TRY(amp::sgd_chunks_run(m->ctx, m->sgd_chunks, m->sgd_nchunks, m->parena, m->garena, m->varena, lr, momentum, weight_decay, grad_scale));
    m->grads_valid = false;
    m->split_stale = true;               // the f16x3 operand copies no longer match the weights
    return AMP_OK;
}

int amp_model_grad_arena(amp_model* m, float** grads, size_t* nfloats) {
    AMP_REQUIRE(m && grads && nfloats && m->garena, "amp_model_grad_arena: the model was created without cfg.train_enable");
    *grads = m->garena;
    *nfloats = m->parena_used;
    return AMP_OK;
}

int amp_model_momentum_arena(amp_model* m, float** vel_dev, size_t* nfloats) {
    AMP_REQUIRE(m && vel_dev && nfloats && m->varena, "amp_model_momentum_arena: the model was created without cfg.train_enable");
    *vel_dev = m->varena;
    *nfloats = m->parena_used;
    return AMP_OK;
}

static int train_entry(amp_model* m, const uint8_t* imgs_bgr, int imgs_on_host, int B, int H, int W, const amp_gt* gt,
                       unsigned int seed, float losses_h[5], int backward) {
    AMP_REQUIRE(m && losses_h, "amp_model_forward_losses: null argument");
    AMP_REQUIRE(m->finalized, "amp_model_forward_losses: call amp_model_finalize after loading every tensor");
    AMP_REQUIRE(m->cfg.train_enable, "amp_model_forward_losses: the model was created with cfg.train_enable = 0");
    AMP_HIP_CHECK(hipSetDevice(m->ctx->device));
    const uint8_t* imgs_d = imgs_bgr;
    // everything that can fail from here on -- bad arguments included -- fails INSIDE the collective protocol below
    auto prepare = [&]() -> int {
        AMP_REQUIRE(imgs_bgr && gt, "amp_model_forward_losses: null argument");
        AMP_REQUIRE(gt->B == B && gt->gt_off && gt->boxes && gt->classes && gt->poly_off && gt->poly_xy, "amp_model_forward_losses: incomplete ground truth");
        const int Hp = (H + 31) / 32 * 32, Wp = (W + 31) / 32 * 32;
        AMP_REQUIRE(B >= 1 && B <= m->cfg.max_batch && Hp <= m->cfg.max_h && Wp <= m->cfg.max_w, "amp_model_forward_losses: batch exceeds the model capacity");
        if (imgs_on_host) {
            AMP_TRY(stage_images(m, imgs_bgr, (size_t)B * H * W * 3));
            imgs_d = m->img_stage;
        }
        return AMP_OK;
    };
    // training: forward and data-gradient convolutions follow the context's mode (AMP_CONV_F16X3: weights are split per call, they
    // change every step); weight gradients are fp32 MFMA
    const int mode = m->ctx->conv_mode;
    // With a communicator of more than one rank the backward entry is a COLLECTIVE protocol: every rank issues the same sequence of
    // RCCL calls whatever happens locally.  A rank whose step failed (e.g. AMP_ERR_NOMEM) still hands the buckets it had not reached
    // to RCCL (the step is thrown away by everyone) and then takes part in the MAX over {0 fine, 1 range flag, 2 failed}, so the other
    // ranks neither block in a collective nobody joins nor apply an update one rank does not have.
    int world = 0;
    (void)amp_comm_info(m->ctx, nullptr, &world, nullptr);
    const bool collective = backward && world > 1;
    if (backward) { m->issued_mask = 0; m->grads_valid = false; }
    auto finish_collectives = [&](int status) -> int {      // after a failed run_train: the buckets this rank did not reach
        if (!collective || status == AMP_OK || m->grad_overlap == 0) return AMP_OK;
        AMP_TRY(amp::comm_wait_done(m->ctx));
        for (int b = 0; b < AMP_GRAD_BUCKETS; ++b)
            if (!(m->issued_mask >> b & 1)) AMP_TRY(issue_bucket(m, b));
        return amp::comm_mark_producer_end(m->ctx);
    };
    auto agree = [&](int status, int* verdict) -> int {     // verdict = MAX over the ranks of {range flag, 2 if the rank failed}
        *verdict = 0;
        if (status != AMP_OK) {
            const int two = 2;
            (void)hipMemcpyAsync(m->ctx->d_conv_flag, &two, sizeof(int), hipMemcpyHostToDevice, m->ctx->stream);
            (void)hipStreamSynchronize(m->ctx->stream);
        }
        if (collective) AMP_TRY(amp::comm_agree_flag(m->ctx, m->ctx->d_conv_flag));
        int flag = 0;
        if (collective || mode == AMP_CONV_F16X3) AMP_TRY(amp_conv_range_flag(m->ctx, 1, &flag));
        *verdict = flag;
        return AMP_OK;
    };
    std::string first_error;
    int st = prepare();
    if (st == AMP_OK) st = run_train(m, imgs_d, B, H, W, gt, seed, losses_h, backward != 0);
    m->ctx->conv_mode = mode;
    if (st != AMP_OK) first_error = amp_last_error();
    if (collective || (st == AMP_OK && mode == AMP_CONV_F16X3)) {
        AMP_TRY(finish_collectives(st));
        int verdict = 0;
        AMP_TRY(agree(st, &verdict));
        if (verdict >= 2) {
            m->grads_valid = false;
            if (st != AMP_OK) { amp::set_error("%s", first_error.c_str()); return st; }
            amp::set_error("amp_model_forward_backward: another rank's step failed; this rank's gradients were discarded");
            return AMP_ERR_STATE;
        }
        if (verdict == 1 && mode == AMP_CONV_F16X3) {   // an activation beyond the fp16 range (on any rank): the step again, entirely on fp32 MFMA
            if (m->f32_reruns++ == 0)
                fprintf(stderr, "[ampis_hip] an activation exceeded the fp16 range of AMP_CONV_F16X3; re-running the step in AMP_CONV_F32\n");
            m->ctx->conv_mode = AMP_CONV_F32;
            if (backward) { m->issued_mask = 0; m->grads_valid = false; }
            st = run_train(m, imgs_d, B, H, W, gt, seed, losses_h, backward != 0);
            m->ctx->conv_mode = mode;
            if (collective) {                            // the re-run issued collectives too: agree on ITS outcome
                if (st != AMP_OK) first_error = amp_last_error();
                AMP_TRY(finish_collectives(st));
                AMP_TRY(agree(st, &verdict));
                if (verdict >= 2) {
                    m->grads_valid = false;
                    if (st != AMP_OK) { amp::set_error("%s", first_error.c_str()); return st; }
                    amp::set_error("amp_model_forward_backward: another rank's step failed; this rank's gradients were discarded");
                    return AMP_ERR_STATE;
                }
            }
        }
    }
    return st;
}

int amp_model_set_image_sizes(amp_model* m, const int* hw_h, int B) {
    AMP_REQUIRE(m && B >= 0, "amp_model_set_image_sizes: bad argument");
    if (hw_h) m->img_hw.assign(hw_h, hw_h + 2 * B); else m->img_hw.clear();
    return AMP_OK;
}

// One tensor between the host (detectron2 / torch layout) and an arena (stored layout).  kind: 0 parameters, 1 gradients, 2 SGD momentum.
// put = false: arena -> out; put = true: out -> arena (momentum only: parameters go through amp_model_load_tensor + finalize).  The
// layout loops are written once; `mv` copies in the direction asked for.
static int xfer_tensor(amp_model* m, const char* name_c, int kind, float* out, size_t cap, bool put) {
    const char* fn = put ? "amp_model_set_momentum_tensor" : "amp_model_get_tensor";
    AMP_REQUIRE(m && name_c && out, "%s: null argument", fn);
    AMP_REQUIRE(m->finalized, "%s: model not finalized", fn);
    AMP_REQUIRE(kind >= 0 && kind <= 2, "%s: kind %d (0 parameter, 1 gradient, 2 momentum)", fn, kind);
    AMP_REQUIRE(kind != 1 || m->garena, "%s: no gradient arena (cfg.train_enable = 0)", fn);
    AMP_REQUIRE(kind != 2 || m->varena, "%s: no momentum arena (cfg.train_enable = 0)", fn);
    const std::string name(name_c);
    const std::string prefix = name.substr(0, name.rfind('.'));
    const bool is_w = ends_with(name, ".weight");
    const int K = m->cfg.num_classes;
    AMP_REQUIRE(name.find(".norm.") == std::string::npos, "%s: FrozenBN statistics have no gradient / are kept on the host", fn);
    AMP_TRY(amp::comm_wait_done(m->ctx));      // a gradient read after an exchange sees the reduced values
    AMP_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    auto arena_ptr = [&](const float* dev) -> float* {
        return kind == 1 ? m->garena + (dev - m->parena) : kind == 2 ? m->varena + (dev - m->parena) : const_cast<float*>(dev);
    };
    auto fetch = [&](const float* dev, size_t n, std::vector<float>& h) -> int {
        h.resize(n);
        AMP_HIP_CHECK(hipMemcpy(h.data(), arena_ptr(dev), n * 4, hipMemcpyDeviceToHost));
        return AMP_OK;
    };
    auto store = [&](const float* dev, const std::vector<float>& h) -> int {
        if (!put) return AMP_OK;
        AMP_HIP_CHECK(hipMemcpy(arena_ptr(dev), h.data(), h.size() * 4, hipMemcpyHostToDevice));
        return AMP_OK;
    };
    auto mv = [put](float& host_layout, float& stored) { if (put) stored = host_layout; else host_layout = stored; };
    std::vector<float> h;
    std::string key = prefix;
    int row0 = 0, rows = -1;   // row slice of a fused tensor
    if (prefix == "roi_heads.box_predictor.cls_score") { key = "roi_heads.box_predictor"; row0 = 0; rows = K + 1; }
    else if (prefix == "roi_heads.box_predictor.bbox_pred") { key = "roi_heads.box_predictor"; row0 = K + 1; rows = 4 * K; }
    else if (prefix == "proposal_generator.rpn_head.objectness_logits") { key = "proposal_generator.rpn_head.pred"; row0 = 0; rows = 3; }
    else if (prefix == "proposal_generator.rpn_head.anchor_deltas") { key = "proposal_generator.rpn_head.pred"; row0 = 3; rows = 12; }
    else if (prefix == "roi_heads.mask_head.predictor") { row0 = 0; rows = K; }
    auto it = m->conv.find(key);
    AMP_REQUIRE(it != m->conv.end(), "%s: unknown tensor '%s'", fn, name_c);
    const ConvW& cw = it->second;
    if (!is_w) {
        AMP_REQUIRE(cw.shift, "%s: '%s' has no bias", fn, name_c);
        AMP_TRY(fetch(cw.shift, (size_t)cw.cout, h));
        const bool deconv = prefix == "roi_heads.mask_head.deconv";
        const int n = deconv ? 256 : (rows >= 0 ? rows : cw.cout);
        AMP_REQUIRE((size_t)n <= cap, "%s: buffer too small", fn);
        for (int j = 0; j < n; ++j) mv(out[j], h[row0 + j]);
        if (put && deconv) for (int q = 1; q < 4; ++q) for (int co = 0; co < 256; ++co) h[q * 256 + co] = out[co];   // the bias is stored once per tap
        return store(cw.shift, h);
    }
    if (cw.groups > 1) {   // window layout [O][KH][KW][64] -> grouped OIHW [O][O/G][KH][KW]
        const int cpg = cw.cout / cw.groups;
        AMP_TRY(fetch(cw.w, (size_t)cw.cout * cw.kh * cw.kw * 64, h));
        AMP_REQUIRE(cap >= (size_t)cw.cout * cpg * cw.kh * cw.kw, "%s: buffer too small", fn);
        for (int o = 0; o < cw.cout; ++o) {
            const int j0 = (o / cpg) * cpg - (o & ~63);
            for (int i = 0; i < cpg; ++i) for (int y = 0; y < cw.kh; ++y) for (int x = 0; x < cw.kw; ++x)
                mv(out[(((size_t)o * cpg + i) * cw.kh + y) * cw.kw + x], h[(((size_t)o * cw.kh + y) * cw.kw + x) * 64 + j0 + i]);
        }
        return store(cw.w, h);
    }
    const size_t kk = (size_t)cw.kh * cw.kw * cw.cin;
    AMP_TRY(fetch(cw.w, (size_t)cw.cout * kk, h));
    if (prefix == "backbone.bottom_up.stem.conv1") {
        AMP_REQUIRE(cap >= (size_t)64 * 3 * 49, "%s: buffer too small", fn);
        for (int o = 0; o < 64; ++o) for (int i = 0; i < 3; ++i) for (int y = 0; y < 7; ++y) for (int x = 0; x < 7; ++x)
            mv(out[((o * 3 + i) * 7 + y) * 7 + x], h[(((size_t)o * 7 + y) * 8 + x) * 4 + i]);
    } else if (prefix == "roi_heads.box_head.fc1") {
        AMP_REQUIRE(cap >= (size_t)1024 * 12544, "%s: buffer too small", fn);
        for (int o = 0; o < 1024; ++o) for (int ch = 0; ch < 256; ++ch) for (int p = 0; p < 49; ++p)
            mv(out[(size_t)o * 12544 + ch * 49 + p], h[(size_t)o * 12544 + p * 256 + ch]);
    } else if (prefix == "roi_heads.mask_head.deconv") {
        AMP_REQUIRE(cap >= (size_t)256 * 256 * 4, "%s: buffer too small", fn);
        for (int ci = 0; ci < 256; ++ci) for (int co = 0; co < 256; ++co) for (int ky = 0; ky < 2; ++ky) for (int kx = 0; kx < 2; ++kx)
            mv(out[(((size_t)ci * 256 + co) * 2 + ky) * 2 + kx], h[((size_t)(ky * 2 + kx) * 256 + co) * 256 + ci]);
    } else if (rows >= 0) {
        AMP_REQUIRE(cap >= (size_t)rows * kk, "%s: buffer too small", fn);
        for (size_t j = 0; j < (size_t)rows * kk; ++j) mv(out[j], h[(size_t)row0 * kk + j]);
    } else {   // [O][KH][KW][I] -> OIHW
        AMP_REQUIRE(cap >= (size_t)cw.cout * kk, "%s: buffer too small", fn);
        for (int o = 0; o < cw.cout; ++o) for (int i = 0; i < cw.cin; ++i) for (int y = 0; y < cw.kh; ++y) for (int x = 0; x < cw.kw; ++x)
            mv(out[(((size_t)o * cw.cin + i) * cw.kh + y) * cw.kw + x], h[(((size_t)o * cw.kh + y) * cw.kw + x) * cw.cin + i]);
    }
    return store(cw.w, h);
}

int amp_model_get_tensor(amp_model* m, const char* name, int kind, float* out, size_t cap) {
    return xfer_tensor(m, name, kind, out, cap, false);
}

int amp_model_set_momentum_tensor(amp_model* m, const char* name, const float* data, size_t n) {
    return xfer_tensor(m, name, 2, const_cast<float*>(data), n, true);
}

int amp_model_get_tap(amp_model* m, const char* name, void** ptr, int* dtype, int* ndim, long long shape[5]) {
    AMP_REQUIRE(m && name && ptr && dtype && ndim && shape, "amp_model_get_tap: null argument");
    auto it = m->taps.find(name);
    AMP_REQUIRE(it != m->taps.end(), "amp_model_get_tap: no buffer named '%s' (run amp_model_infer first)", name);
    *ptr = it->second.ptr; *dtype = it->second.dtype; *ndim = it->second.ndim;
    for (int i = 0; i < 5; ++i) shape[i] = i < it->second.ndim ? it->second.shape[i] : 1;
    return AMP_OK;
}

}  // extern "C"
