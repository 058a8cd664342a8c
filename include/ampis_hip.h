/*
 * ampis_hip.h — C ABI of libampis_hip.so: the MI355X (gfx950) Mask R-CNN R50-FPN hot path that
 * rccohn/AMPIS reaches through detectron2.
 *
 * The reference has no FFI of its own; its boundary to this path is the detectron2 Python API
 * (SURVEY.md §8b). Each entry point below names the reference call site it replaces:
 *   - amp_infer*                 <- `predictor(img)` at colab/AMPIS Tutorial.ipynb cell 26/28 (DefaultPredictor.__call__),
 *                                   whose output is consumed by ampis/data_utils.py:275-278 (compress_pred)
 *   - amp_dets (RLE counts)      <- `RLE.encode(np.asfortranarray(x))` at ampis/data_utils.py:275
 *   - amp_rle_*                  <- pycocotools.mask calls at ampis/data_utils.py:275, ampis/analyze.py:108,158,315-321,
 *                                   ampis/structures.py:465-468,568,752
 *   - amp_model_* / amp_init     <- `DefaultPredictor(cfg)` (notebook cell 24) / `DefaultTrainer(cfg)` (cell 22,
 *                                   ampis/data_utils.py:135-160)
 * The per-kernel entry points (amp_conv2d_nhwc, amp_roi_align, amp_nms, ...) are the stages of that path
 *  and exist so the parity tests can check each stage against the CPU oracle.
 *
 * Conventions: plain C types only; every pointer is a DEVICE pointer unless its name ends in _h (host);
 * activations are fp32 NHWC, weights fp32 [Cout][KH][KW][Cin]; every function returns AMP_OK (0) or a
 * negative amp_status and leaves a message readable through amp_last_error(); handles are opaque, one per
 * HIP stream, thread-compatible but not thread-safe.
 */
#ifndef AMPIS_HIP_H
#define AMPIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum amp_status {
    AMP_OK = 0,
    AMP_ERR_ARG = -1,      /* bad argument / shape the kernels do not support */
    AMP_ERR_HIP = -2,      /* HIP runtime error (message has the call and hipGetErrorString) */
    AMP_ERR_NOMEM = -3,    /* workspace too small */
    AMP_ERR_STATE = -4     /* call out of order (e.g. infer before weights are loaded) */
} amp_status;

typedef struct amp_ctx amp_ctx;

/* Library / context -------------------------------------------------------------------------- */
const char* amp_last_error(void);
int  amp_version(void);
/* flags & AMP_STREAM_BORROW: launch on `hip_stream` (a hipStream_t; NULL = the legacy default stream) and never
 * destroy it; otherwise hip_stream is ignored and the context creates (and owns) a non-blocking HIP stream. */
#define AMP_STREAM_BORROW 1
int  amp_init(int device, void* hip_stream, int flags, amp_ctx** out);
void amp_destroy(amp_ctx* ctx);
int  amp_sync(amp_ctx* ctx);
void* amp_stream(amp_ctx* ctx);
/* HIP-event stopwatch on the context's stream: start records an event, stop records a second one, waits for it
 * and returns the elapsed milliseconds between the two (bench.py times kernels with this, not with torch events). */
int  amp_timer_start(amp_ctx* ctx);
int  amp_timer_stop(amp_ctx* ctx, float* ms_h);

/* Live profile of the dominant kernel: between begin and end every amp_conv2d_nhwc launch on this context is bracketed
 * by a HIP-event pair on the context's stream; end waits for the stream and sums duration and algorithmic FLOPs
 * (2*M*Cout*KH*KW*Cin) per slot: [0] = wide conv tiles (128x128 / 128x256), [1] = 128x64 conv tiles, [2] = the weight-gradient
 * MFMA kernel of amp_conv2d_wgrad (wgrad_f16x3_kernel / wgrad_mfma_kernel, without its row-table and reduce passes). */
typedef struct amp_prof_summary {
    long long launches[3];
    double ms[3];
    double flops[3];
    int truncated;            /* 1 when more launches happened than max_launches */
} amp_prof_summary;
int  amp_prof_begin(amp_ctx* ctx, int max_launches);
int  amp_prof_end(amp_ctx* ctx, amp_prof_summary* out);
/* between begin and end: stop / resume recording (the event pairs themselves cost ~5 us of idle GPU per launch: sample some steps) */
int  amp_prof_pause(amp_ctx* ctx, int paused);
/* After amp_prof_end: the recorded launches one by one, in launch order -- duration, algorithmic FLOPs and algorithmic BYTES (every
 * operand the launch must read and every value it must write, once: sampled input rows + weights + output + residual / mask), and the
 * GEMM shape M x N x K.  What a per-layer roofline needs: a short-K 1x1 layer is bounded by its bytes, not by the matrix pipe
 * (tools/layer_roofline.py).  *n_out = number of recorded launches (may exceed cap; the first cap are written). */
typedef struct amp_prof_launch {
    double ms, flops, bytes;
    int M, N, K;
    int slot;                 /* index into amp_prof_summary's arrays */
} amp_prof_launch;
int  amp_prof_launches(amp_ctx* ctx, amp_prof_launch* out, int cap, int* n_out);

/* Device memory helpers (so that hosts without torch can drive the library) ----------------- */
int amp_malloc(amp_ctx* ctx, size_t bytes, void** out);
int amp_free(amp_ctx* ctx, void* p);
int amp_memcpy_h2d(amp_ctx* ctx, void* dst, const void* src_h, size_t bytes);
int amp_memcpy_d2h(amp_ctx* ctx, void* dst_h, const void* src, size_t bytes);
int amp_memset(amp_ctx* ctx, void* dst, int value, size_t bytes);

/* Convolution arithmetic of this context (every conv / fc of the inference path and, in training, the forward and data-gradient
 * convolutions; weight gradients always run on the fp32 MFMA).
 * AMP_CONV_F32:   v_mfma_f32_32x32x2_f32, exact fp32 products, fp32 accumulation.
 * AMP_CONV_F16X3: fp32 in / fp32 out on the f16 matrix pipe: each operand is split x = hi + lo (two f16, 22 significant bits for
 *                 2^-25 < |x| < 65504), a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with exact products and fp32 accumulation;
 *                 error against an fp64 reference measured at or below the fp32-MFMA kernel's (tests/test_conv_modes_gpu.py).
 *                 An operand of magnitude >= 65504 leaves a non-finite accumulator, raises the range flag below, and
 *                 amp_model_infer re-runs the batch in AMP_CONV_F32.  Default; override with amp_set_conv_mode or AMP_CONV_MODE=f32|f16x3. */
enum { AMP_CONV_F32 = 0, AMP_CONV_F16X3 = 1 };
int amp_set_conv_mode(amp_ctx* ctx, int mode);
int amp_get_conv_mode(amp_ctx* ctx);
/* *flag_h = 1 when an AMP_CONV_F16X3 convolution since the last clear left a non-finite accumulator; synchronises the stream */
int amp_conv_range_flag(amp_ctx* ctx, int clear, int* flag_h);

/* w [rows][K] fp32 (K % 32 == 0) -> the AMP_CONV_F16X3 operand layout (same byte size); amp_conv2d_nhwc does this per call,
 * amp_model_finalize once per layer. */
int amp_split_weights(amp_ctx* ctx, const float* w, long long rows, int K, float* w_split);

/* Stage a7 (DefaultPredictor.__call__: ResizeShortestEdge): uint8 bilinear resize, bit-exact with PIL's Image.resize(BILINEAR)
 * (antialiased triangle filter, 22-bit fixed-point coefficients, horizontal then vertical pass).  src [H,W,3], dst [h,w,3] and
 * tmp (amp_resize_scratch_bytes) are device pointers. */
size_t amp_resize_scratch_bytes(int H, int W, int h, int w);
int amp_resize_bilinear_u8(amp_ctx* ctx, const unsigned char* src, int H, int W, unsigned char* dst, int h, int w, void* tmp);
/* Train-time input pipeline on the device (detectron2 DatasetMapper under ampis/data_utils.py:171-175: ResizeShortestEdge + RandomFlip, then
 * ImageList.from_tensors): the same resize written into the top-left h x w pixels of a frame slot whose rows are dst_pitch pixels apart,
 * mirrored left-right when flip != 0 (== numpy out[:, ::-1] of the resized image).  h == H and w == W: a (mirrored) copy, tmp may be null. */
int amp_resize_flip_u8(amp_ctx* ctx, const unsigned char* src, int H, int W, unsigned char* dst, int dst_pitch, int h, int w, int flip, void* tmp);

/* Stage a9/a10/a11/a14/a16: implicit-GEMM convolution on fp32 MFMA ------------------------- */
typedef struct amp_conv_desc {
    int B, H, W, Cin;         /* input  [B,H,W,Cin]  (Cin % 4 == 0) */
    int Cout;                 /* weight [Cout][KH][KW][Cin] */
    int KH, KW, stride, pad;
    int relu;                 /* 1: y = max(y, 0) after the affine and the residual */
    int res_mode;             /* 0 none; 1 res[B,Ho,Wo,Cout] added; 2 res[B,Ho/2,Wo/2,Cout] nearest-upsampled x2 then added (FPN top-down) */
    int out_mode;             /* 0 NHWC [B,Ho,Wo,Cout]; 1 ConvTranspose 2x2 s2 scatter: Cout = 4*C2 ordered (ky,kx,co) -> y[B,2Ho,2Wo,C2];
                                 2 stride-2 scatter: row (b,oy,ox) -> y[B,2Ho,2Wo,Cout] at (2oy,2ox), y pre-zeroed (dgrad of a strided 1x1) */
} amp_conv_desc;
/* y = act( conv(x, w) * scale[c] + shift[c] (+ res) ); scale may be NULL (= 1), shift may be NULL (= 0). */
int amp_conv2d_nhwc(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* w,
                    const float* scale, const float* shift, const float* res, float* y);
/* Grouped convolution (detectron2 ResNeXt `conv2`, groups = RESNETS.NUM_GROUPS): Cin == Cout, Cin / groups in {8,16,32,64}.
 * w_win is the window layout [Cout][KH][KW][64] made by amp_group_expand_weights from grouped weights [Cout][KH][KW][Cin/groups]. */
int amp_group_expand_weights(amp_ctx* ctx, const float* w, int Cout, int KH, int KW, int cpg, float* w_win);
int amp_conv2d_grouped_nhwc(amp_ctx* ctx, const amp_conv_desc* d, int groups, const float* x, const float* w_win,
                            const float* scale, const float* shift, const float* res, float* y);
/* the same on split-format tensors (AMP_CONV_F16X3; fmt: AMP_FMT_* bits as in amp_conv2d_nhwc_fmt): the 64-channel window of an N tile
 * is 256 B of a split row as it is of an fp32 row, so a ResNeXt trunk stays in the format through its grouped layers */
int amp_conv2d_grouped_nhwc_fmt(amp_ctx* ctx, const amp_conv_desc* d, int groups, const float* x, const float* w_win,
                                const float* scale, const float* shift, const float* res, float* y, int fmt);
/* Backward of the grouped convolution (training a ResNeXt backbone).  Weight gradient in the window layout (zero outside a channel's
 * group), times scale[co] when given (FrozenBN); fp32 MFMA, partial sums over row slices added in slice order.  scratch:
 * amp_grouped_wgrad_scratch_floats(d) floats.  Data gradient: the grouped forward convolution of dy with amp_group_dgrad_weights'
 * windows (transposed inside each 64-channel tile, taps flipped, times scale[co]); stride 1 (a stride-2 layer spreads dy over the even
 * positions of a zeroed map first). */
size_t amp_grouped_wgrad_scratch_floats(const amp_conv_desc* d);
int amp_conv2d_grouped_wgrad(amp_ctx* ctx, const amp_conv_desc* d, int groups, const float* x, const float* dy, const float* scale,
                             float* scratch, float* grad_win);
/* The same with operands in the split hi|lo' row format (same bytes as fp32): fmt bit 0: x, bit 1: dy; decoded on the load (exact).  dy -- split
 * or fp32 -- may hold dy * 2^dy_shift (the scaled gradient chain of a training step): the reduce pass multiplies the sums by 2^-dy_shift. */
int amp_conv2d_grouped_wgrad_fmt(amp_ctx* ctx, const amp_conv_desc* d, int groups, const float* x, const float* dy, const float* scale,
                                 float* scratch, float* grad_win, int fmt, int dy_shift);
int amp_group_dgrad_weights(amp_ctx* ctx, const float* w_win, const float* scale, int C, int KH, int KW, float* wt_win);
/* same, with an optional mask tensor indexed like y: y = mask > 0 ? y : 0 (applied last; the ReLU backward of a data gradient) */
int amp_conv2d_nhwc_ex(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* w, const float* scale, const float* shift,
                       const float* res, const float* mask, float* y);

/* The split operand format as a tensor format (AMP_CONV_F16X3 inference: the trunk's native activation format).  A [rows][C] fp32
 * tensor and its split form have the same byte size and the same row offsets; inside a row every 32 channels are 32 f16 `hi` halves
 * (64 B) followed by 32 f16 halves of lo' = (x - hi) * 2^11 (64 B).  amp_split_weights makes it from fp32 rows, amp_unsplit_rows
 * reads it back (hi + lo' * 2^-11, exact in fp32); a convolution can take its input and its residual in it and write its output in
 * it (fmt bits below), which changes the data path (both operands staged by LDS-DMA, no split in the kernel), not the arithmetic. */
enum { AMP_FMT_X_SPLIT = 1, AMP_FMT_Y_SPLIT = 2, AMP_FMT_RES_SPLIT = 4, AMP_FMT_MASK_SPLIT = 8 };
int amp_conv2d_nhwc_fmt(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* w, const float* scale, const float* shift,
                        const float* res, float* y, int fmt);
int amp_unsplit_rows(amp_ctx* ctx, const float* x_split, long long rows, int C, float* out);
/* The tail of a res2 bottleneck (detectron2 BottleneckBlock.forward, modeling/backbone/resnet.py: conv2 3x3 64 -> 64 + FrozenBN + ReLU, conv3 1x1
 * 64 -> C3 + FrozenBN, + shortcut, ReLU) in ONE launch, every tensor in the split row format and the weights pre-split (amp_split_weights): conv2's
 * output never reaches memory.  Bit-identical to the two amp_conv2d_nhwc_fmt calls it replaces.  AMP_ERR_STATE when it does not apply
 * (not AMP_CONV_F16X3, C3 % 64 != 0 or > 256, fewer than 512 tiles of 8 x 16 pixels). */
int amp_bottleneck64_tail(amp_ctx* ctx, int B, int H, int W, const float* x_split, const float* w2_split, const float* scale2, const float* shift2,
                          const float* w3_split, const float* scale3, const float* shift3, int C3, const float* res_split, float* y_split);

/* Stage a19: backward building blocks -------------------------------------------------------------------------------- */
/* dW[Cout][KH][KW][Cin] (= or +=) scale[n] * conv-wgrad(dy [B*Ho*Wo, Cout], x [B,H,W,Cin]); `d` describes the FORWARD conv.
 * Cin % 128 == 0, Cout % 4 == 0. scratch: amp_conv_wgrad_scratch_floats(d) floats. Deterministic (split-K slabs, fixed-order sum). */
size_t amp_conv_wgrad_scratch_floats(const amp_conv_desc* d);
int amp_conv2d_wgrad(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* dy, const float* scale, float* scratch,
                     float* grad, int accumulate);
/* same; in AMP_CONV_F16X3 mode dy (or x) is multiplied by 2^dy_shift (2^x_shift) before the f16 split and the result by the inverse
 * (exact): loss gradients of 1e-9..1e-4 need it to keep fp32-equivalent accuracy. At most one shift non-zero; ignored in AMP_CONV_F32.
 * |operand * 2^shift| >= 65504 raises amp_conv_range_flag(). amp_conv2d_wgrad == shifts 0. */
int amp_conv2d_wgrad_scaled(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* dy, const float* scale, float* scratch,
                            float* grad, int accumulate, int dy_shift, int x_shift);
/* The same with x in the split hi|lo' row format (AMP_FMT_X_SPLIT; written by amp_conv2d_nhwc_fmt / the native trunk): x_split = 1;
 * x_split & 2: dy is in that format as well and already multiplied by 2^dy_shift (what a data-gradient convolution with AMP_FMT_Y_SPLIT
 * on scaled split gradients writes);
 * bias_grad != NULL (AMP_CONV_F16X3 only): bias_grad[n] (= or +=) sum over the pixels of dy[.][n], summed on the side by the MFMA kernel
 * from the dy tiles it stages anyway (replaces a separate amp_colsum pass over dy). */
int amp_conv2d_wgrad_fmt(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* dy, const float* scale, float* scratch,
                            float* grad, int accumulate, int dy_shift, int x_shift, int x_split, float* bias_grad, int bias_accumulate);
/* out[n] (= or +=) sum_m dy[m][n]; N % 4 == 0; scratch >= ceil(M/512)*N floats */
int amp_colsum(amp_ctx* ctx, const float* dy, int M, int N, float* scratch, float* out, int accumulate);
/* the same pass also writing dy * 2^shift in the split row format (N % 32 == 0): the dy operand of amp_conv2d_wgrad_fmt(x_split & 2) */
int amp_colsum_split(amp_ctx* ctx, const float* dy, int M, int N, float* scratch, float* out, int accumulate, float* dy_split, int shift);
/* the column sums alone of a dy that is already split rows of dy * 2^shift (read-only; the values summed are the 22-bit ones the split holds) */
int amp_colsum_of_split(amp_ctx* ctx, const float* dy_split, int M, int N, float* scratch, float* out, int accumulate, int shift);
/* wt[Cin][KH][KW][Cout] = flipped / transposed / scaled copy of w[Cout][KH][KW][Cin]: conv(dy, wt) is the data gradient */
int amp_dgrad_weights(amp_ctx* ctx, const float* w, const float* scale, int Cout, int KH, int KW, int Cin, float* wt);
/* the same tensor already in the AMP_CONV_F16X3 split row format (= amp_split_weights of amp_dgrad_weights' result, bit for bit, in one
 * pass; Cout % 32 == 0): what a training step of amp_model makes for all its layers in one launch.  Allocates two small tables per call. */
int amp_dgrad_weights_split(amp_ctx* ctx, const float* w, const float* scale, int Cout, int KH, int KW, int Cin, float* wt_split);
/* dfeat[level] += RoIAlign-backward(dout [R,P,P,C]).  C == 256: owner-computes, no atomics -- every 4x4 tile of a gradient map is
 * summed by one wave over the RoIs in index order: bitwise reproducible.  Other C: float atomics (reproducible to fp32 rounding).
 * _batched: B = number of images (maps are [B,h,w,C]); amp_roi_align_bwd derives it from batch_idx (one small read-back). */
int amp_roi_align_bwd(amp_ctx* ctx, float* const dfeat[4], const int fh[4], const int fw[4], const int stride[4], int C, const float* rois,
                      const int* batch_idx, int R, int P, const float* dout);
int amp_roi_align_bwd_batched(amp_ctx* ctx, float* const dfeat[4], const int fh[4], const int fw[4], const int stride[4], int C, const float* rois,
                              const int* batch_idx, int R, int P, const float* dout, int B);
int amp_upsample2_bwd(amp_ctx* ctx, const float* dfine, float* dcoarse, int B, int Hc, int Wc, int C);   /* dcoarse += 2x2 sums */
int amp_subsample2_bwd(amp_ctx* ctx, const float* dy, float* dx, int B, int H, int W, int C);           /* dx[::2, ::2] += dy */
int amp_relu_mask(amp_ctx* ctx, float* g, const float* act, size_t n);
/* The same with the activation in the split hi|lo' row format ([.., C] rows, C % 32 == 0). */
int amp_relu_mask_split(amp_ctx* ctx, float* g, const float* act_split, size_t n, int C);
/* Scaled split gradients (the backbone's backward pass on the ring kernel): out = split(2^shift * (act > 0 ? g : 0)), and the
 * stride-2 scatter-add back into an fp32 gradient: dx[b,2y,2x,:] += 2^-shift * decode(dy_split[b,y,x,:]). */
int amp_relu_mask_to_split(amp_ctx* ctx, const float* g, const float* act_split, float* out_split, size_t n, int C, int shift);
int amp_subsample2_bwd_split(amp_ctx* ctx, const float* dy_split, float* dx, int B, int H, int W, int C, int shift);
/* dx (fp32, same resolution) += decode(dy_split) * 2^-shift: a scaled split gradient joins an fp32 gradient map (ResNeXt first blocks, where
 * conv1's data gradient lives at the block's INPUT resolution). */
int amp_accumulate_split(amp_ctx* ctx, const float* dy_split, float* dx, long long rows, int C, int shift);
/* up[b, 2y, 2x, :] = src[b, y, x, :] as raw 16-byte chunks (split rows stay split rows), every other pixel of `up` zero: the input of the stride-1
 * data gradient of a stride-2 convolution whose gradient arrives in the split format.  up: [B, H, W, C], src: [B, ceil(H/2), ceil(W/2), C]. */
int amp_scatter2_rows(amp_ctx* ctx, const float* src, float* up, int B, int H, int W, int C);                                  /* g *= (act > 0) */
int amp_small_k_dgrad(amp_ctx* ctx, const float* dl, int ld, int K, const float* w, int C, const float* act, float* dx, size_t npix);
/* The same product for dl rows of 16 floats (K <= 16), C % 32 == 0 and act in the split row format, leaving dx * 2^shift as split rows (the
 * dy operand of the ring kernels) and colsum_out[c] (= or +=) the sum over the pixels of dx[.][c] (the bias gradient of the layer that
 * produced act); scratch: ceil(npix / 512) * C floats. */
int amp_small_k_dgrad_split(amp_ctx* ctx, const float* dl, int K, const float* w, int C, const float* act_split, float* dx_split, int npix,
                            int shift, float* scratch, float* colsum_out, int accumulate);
/* The same with dl rows of ld = 4, 8, 12 or 16 floats (K <= ld) and act as plain fp32 rows (the mask predictor behind the deconv, whose output
 * is fp32): dx is amp_small_k_dgrad's value, * 2^shift, as split rows; colsum_out = the deconv's bias gradient. */
int amp_small_k_dgrad_split_f32act(amp_ctx* ctx, const float* dl, int ld, int K, const float* w, int C, const float* act, float* dx_split, int npix,
                                   int shift, float* scratch, float* colsum_out, int accumulate);
/* ... and with act in the split row format again (a deconv output kept split: amp_conv2d_nhwc_fmt with out_mode 1 and AMP_FMT y split) */
int amp_small_k_dgrad_split_ld(amp_ctx* ctx, const float* dl, int ld, int K, const float* w, int C, const float* act_split, float* dx_split, int npix,
                               int shift, float* scratch, float* colsum_out, int accumulate);
int amp_colsum_finish(amp_ctx* ctx, const float* partial, int parts, int N, float* out, int accumulate);
int amp_deconv_grad_transpose(amp_ctx* ctx, const float* in, float* out, int Cin, int T, int C2, int accumulate);
/* torch.optim.SGD: g' = grad_scale*g + wd*p; v = mu*v + g'; p -= lr*v */
int amp_sgd_update(amp_ctx* ctx, float* p, const float* g, float* v, size_t n, float lr, float momentum, float weight_decay, float grad_scale);


/* Stage a8 / a9 / a10: memory-bound NHWC helpers --------------------------------------------- */
/* uint8 BGR [B,H,W,3] -> fp32 [B,Hp,Wp,4] = (x - mean) / std, zero padded (4th channel 0). img_hw: optional device [B][2]
 * valid (h,w) of each image inside the common HxW frame; pixels beyond it are 0 after normalisation (ImageList.from_tensors). */
int amp_preprocess(amp_ctx* ctx, const uint8_t* img_bgr, int B, int H, int W, int Hp, int Wp, const float mean[3],
                   const float std[3], const int* img_hw, float* out);
int amp_maxpool3x3s2(amp_ctx* ctx, const float* x, int B, int H, int W, int C, float* y);   /* k3 s2 p1 */
int amp_subsample2(amp_ctx* ctx, const float* x, int B, int H, int W, int C, float* y);     /* x[:, ::2, ::2] */

/* Stage a12: RPN candidate selection ----------------------------------------------------------- */
typedef struct amp_rpn_levels {
    int nlevels;               /* <= 5 */
    int A;                     /* anchors per location (3) */
    int ld;                    /* row length of pred: A logits then A*4 deltas (15) */
    const float* pred[5];      /* per level [B, h*w, ld] */
    int h[5], w[5], stride[5], anchor_size[5];
} amp_rpn_levels;
/* per (image, level) top-k by logit, descending, ties by ascending anchor index. keys_scratch: [B*nlevels*max_n] u32. */
int amp_rpn_topk(amp_ctx* ctx, const amp_rpn_levels* lv, int B, int k, uint32_t* keys_scratch, int max_n, int* sel_idx,
                 float* sel_logit, int* sel_count);
/* decode + clip the selected anchors; sortkey = 0 for non-finite / empty boxes. boxes [B,cap,4], sortkey [B,cap]. */
int amp_rpn_decode(amp_ctx* ctx, const amp_rpn_levels* lv, int B, int k, const int* sel_idx, const float* sel_logit,
                   const int* sel_count, int img_h, int img_w, int cap, float* boxes, unsigned long long* sortkey,
                   int* anchor_id /* [B,cap] global anchor index of each candidate, or NULL */);
/* same with per-image sizes: img_hw = device int [B][2] (h, w) of each image inside the common frame (a batch of differently sized
 * images, detectron2 ImageList): every image's proposals are clipped to ITS size (find_top_rpn_proposals); NULL = img_h / img_w. */
int amp_rpn_decode_sized(amp_ctx* ctx, const amp_rpn_levels* lv, int B, int k, const int* sel_idx, const float* sel_logit,
                         const int* sel_count, int img_h, int img_w, const int* img_hw, int cap, float* boxes,
                         unsigned long long* sortkey, int* anchor_id);
/* per image: order `cap` (<= 16384) 64-bit sort words descending; gather boxes_in[b][pos] (box_stride entries per image);
 * outputs sorted boxes/scores/categories [B,cap], the number of valid entries [B], optionally the source positions. */
int amp_sort_gather(amp_ctx* ctx, int B, int cap, int box_stride, const unsigned long long* sortkey, const float* boxes_in,
                    float* boxes_out, float* score_out, int* cat_out, int* count_out, int* pos_out,
                    const int* payload_in /* [B,box_stride] or NULL */, int* payload_out /* [B,cap] or NULL */);
/* the same with a hint: n_used[b] (device, may exceed cap) = the sort words of image b are its first n_used[b] slots and the rest are 0
 * (the compacted candidate list of amp_box_candidates): only the smallest power of two that holds them is sorted. NULL: all cap slots. */
int amp_sort_gather_n(amp_ctx* ctx, int B, int cap, int box_stride, const unsigned long long* sortkey, const float* boxes_in,
                      float* boxes_out, float* score_out, int* cat_out, int* count_out, int* pos_out,
                      const int* payload_in, int* payload_out, const int* n_used);

/* Stages a12 / a15: batched greedy NMS on score-sorted boxes (cap <= 16384 per image) ---------- */
/* mask_scratch: [B*cap*ceil(cap/64)] u64. keep_idx [B,max_keep] (positions, ascending), keep_count [B]. */
int amp_nms(amp_ctx* ctx, int B, int cap, const float* boxes, const int* cats, const int* counts, float thresh,
            int max_keep, unsigned long long* mask_scratch, int* keep_idx, int* keep_count);

/* Stage a12, second half of find_top_rpn_proposals (detectron2 proposal_utils.py: batched_nms(boxes, scores, lvl, nms_thresh)[:post_nms_topk]):
 * the decoded candidates of amp_rpn_decode (level l of image b at [off, off + sel_count[b][l]) of its `cap` slots, each level in
 * descending order; sort word 0 = invalid) are suppressed PER LEVEL and the survivors of the L levels merged by sort word: the first
 * max_keep per image with boxes / logits / levels (/ payload_in[b][slot] -> payload_out), the remaining rows zero / -1 as
 * amp_gather_dets leaves them.  Identical to amp_sort_gather + amp_nms(cats = level) + amp_gather_dets on the same candidates.
 * scratch: amp_rpn_nms_scratch_words(B, L, k) u64. */
size_t amp_rpn_nms_scratch_words(int B, int L, int k);
int amp_rpn_nms_levels(amp_ctx* ctx, int B, int L, int k, int cap, const float* cand_boxes, const unsigned long long* cand_keys,
                       const int* sel_count, float thresh, int max_keep, unsigned long long* scratch, float* prop_boxes,
                       float* prop_scores, int* prop_lvl, int* prop_count, const int* payload_in /* [B,cap] or NULL */,
                       int* payload_out /* [B,max_keep] or NULL */);

/* Stages a13 / a16: RoIAlign (aligned, sampling_ratio 0) with FPN level assignment ------------- */
typedef struct amp_fpn_feats {
    const float* feat[4];      /* p2..p5, NHWC [B,h,w,C] */
    int h[4], w[4], stride[4];
    int C;
} amp_fpn_feats;
/* rois [R,4] (x1,y1,x2,y2 in image coordinates), batch_idx [R] or NULL, roi_count device int or NULL -> out [R,P,P,C]. */
int amp_roi_align(amp_ctx* ctx, const amp_fpn_feats* f, const float* rois, const int* batch_idx, const int* roi_count,
                  int R, int P, float* out, int* level_out);

/* same with the feature maps (AMP_FMT_X_SPLIT) and / or the pooled output (AMP_FMT_Y_SPLIT) in the split row format; C % 32 == 0 */
int amp_roi_align_fmt(amp_ctx* ctx, const amp_fpn_feats* f, const float* rois, const int* batch_idx, const int* roi_count,
                      int R, int P, float* out, int* level_out, int fmt);

/* Stage a15: box-head inference ------------------------------------------------------------------ */
int amp_box_candidates(amp_ctx* ctx, const float* pred, int ld, const float* proposals, const int* prop_count, int B,
                       int Rcap, int K, const float reg_weights[4], float score_thresh, int img_h, int img_w,
                       float* dense_boxes, unsigned long long* keys, int ccap, int* cand_count, int* overflow);
int amp_box_candidates_sized(amp_ctx* ctx, const float* pred, int ld, const float* proposals, const int* prop_count, int B,
                             int Rcap, int K, const float reg_weights[4], float score_thresh, int img_h, int img_w,
                             const int* img_hw /* device [B][2] or NULL, as in amp_rpn_decode_sized */,
                             float* dense_boxes, unsigned long long* keys, int ccap, int* cand_count, int* overflow);
int amp_gather_dets(amp_ctx* ctx, int B, int cap, int D, const float* sboxes, const float* sscores, const int* scats,
                    const int* keep_idx, const int* keep_count, float* det_boxes, float* det_scores, int* det_classes,
                    const int* payload_in /* [B,cap] or NULL */, int* payload_out /* [B,D] or NULL */);
/* [B][D] detections with device-side counts -> compact rows in image order (+ the image index of every row) for the mask branch */
int amp_compact_dets(amp_ctx* ctx, int B, int D, const int* det_count, const float* det_boxes, const float* det_scores, const int* det_classes,
                     float* boxes, float* scores, int* classes, int* batch);

/* Stages a16 / a17 / a3: mask probability, paste + threshold + RLE counts ----------------------- */
int amp_mask_prob(amp_ctx* ctx, const float* logits, const int* classes, int N, int K, float* prob);
/* the RPN head of one pyramid level in one kernel (AMP_CONV_F16X3; stage a11): x_split [B,H,W,256] (split rows) -> 3x3 conv + bias + ReLU ->
 * the 16 predictor rows (3 objectness logits, 12 anchor deltas, 1 zero row: w_pred [16][256], b_pred [16]) as a second product in the
 * epilogue -> pred [B*H*W][16]; the 256-channel hidden tensor is never written. B*H*W >= 24576. */
int amp_rpn_head_fused(amp_ctx* ctx, const float* x_split, int B, int H, int W, const float* w_conv, const float* b_conv, const float* w_pred,
                       const float* b_pred, float* pred);
/* the tail of the mask head in one kernel (AMP_CONV_F16X3): x_split [N,14,14,256] (split rows) -> ConvTranspose2d 2x2 s2 (w_deconv
 * [(ky,kx,co)][256], bias [1024] = the 256 biases once per tap) -> ReLU -> predictor row of classes[n] (pred_w [K][256], pred_b [K]) ->
 * sigmoid -> prob [N,28,28]; the [N,28,28,256] activation is never written */
int amp_mask_deconv_predict(amp_ctx* ctx, const float* x_split, int N, const float* w_deconv, const float* bias, const float* pred_w,
                            const float* pred_b, const int* classes, int K, float* prob);
int amp_paste_rle(amp_ctx* ctx, const float* prob, const float* det_boxes, const int* det_batch, int N, const int* out_h,
                  const int* out_w, int max_out_hw, int in_h, int in_w, float threshold, float* out_boxes, int* valid,
                  unsigned int* pool, unsigned long long pool_cap, unsigned long long* pool_used, unsigned long long* rle_off,
                  int* rle_len, int* overflow);
/* same; in_hw = device int [B][2]: the network-input size of each image (detector_postprocess scales by output / image size) or NULL;
 * pos_scratch (+ capacity, device counter) = a second pool for the transition positions the run lengths are made from: `pool` then
 * holds the run lengths only, densely (half the bytes to read back); NULL = positions behind each mask's run lengths in `pool`. */
int amp_paste_rle_sized(amp_ctx* ctx, const float* prob, const float* det_boxes, const int* det_batch, int N, const int* out_h,
                        const int* out_w, int max_out_hw, int in_h, int in_w, const int* in_hw, float threshold, float* out_boxes,
                        int* valid, unsigned int* pool, unsigned long long pool_cap, unsigned long long* pool_used,
                        unsigned long long* rle_off, int* rle_len, int* overflow, unsigned int* pos_scratch,
                        unsigned long long pos_cap, unsigned long long* pos_used);

/* Stage a18: training-mode label assignment, seeded sampling and losses (+ their gradients w.r.t. the network outputs) ---- */
int amp_anchor_labels(amp_ctx* ctx, const amp_rpn_levels* lv, int B, const float* gt_boxes, const int* gt_off, int total_gt,
                      float iou_lo, float iou_hi, float* match_val, int* match_idx, unsigned int* gt_best, signed char* label);
int amp_rpn_sample_loss(amp_ctx* ctx, const amp_rpn_levels* lv, float* const dpred[5], int B, const float* gt_boxes,
                        const int* gt_off, const signed char* label, const int* match_idx, uint32_t* keys_scratch, int batch,
                        float pos_frac, unsigned int seed, int* sampled, int* counts, float* partial);
int amp_roi_sample(amp_ctx* ctx, int B, const float* prop_boxes, const int* prop_count, int Pcap, const float* gt_boxes,
                   const int* gt_classes, const int* gt_off, int K, int batch, float fg_frac, float iou_thresh, unsigned int seed,
                   uint32_t* keys_scratch, int* cls_scratch, int* gti_scratch, int ncap, float* rois, int* roi_cls, int* roi_gti,
                   int* counts, const int* prop_anchor /* [B,Pcap] stable ids for the sampling hash */, int num_anchors);
int amp_box_loss(amp_ctx* ctx, int B, int batch, int K, const float* pred, int ld, float* dpred, const float* rois, const int* roi_cls,
                 const int* roi_gti, const float* gt_boxes, const int* gt_off, const float reg_weights[4], int total_rois, float* partial);
/* Sigmoid focal loss (fvcore sigmoid_focal_loss, reduction 'sum', times `scale`) over logits [N,K] with integer labels (label K =
 * background, negative = the row is ignored), forward and gradient in one pass: *loss_d = scale * sum, dlogits (optional) = scale *
 * d sum / d logits.  alpha < 0 switches the class weighting off.  partial: device scratch of partial_cap floats (<= 2048 used); sums run in a
 * fixed order (bitwise reproducible).  Not on the reference's Mask R-CNN path; the north star names it. */
int amp_sigmoid_focal_loss(amp_ctx* ctx, long long N, int K, const float* logits, const int* labels, float alpha, float gamma, float scale,
                           float* dlogits, float* partial, int partial_cap, float* loss_d);
/* Mask targets + mask BCE loss.  Polygon ground truth: detectron2 PolygonMasks.crop_and_resize = rasterize_polygons_within_box (every
 * polygon of the instance by pycocotools' rleFrPoly at 28 x 28, merged).  Bitmask ground truth (what `get_ddicts('binary'|'label'|'rle')`
 * emits, ampis/data_utils.py:394-433,482-525): BitMasks.crop_and_resize = roi_align(mask, scale 1, sampling_ratio 0, aligned) >= 0.5,
 * evaluated straight from the instance's COCO run lengths (amp_mask_targets_bitmask; scratch = nslots x slot_words words, a slot must
 * hold width x ceil(height / 32) words of the largest mask).  amp_mask_target_loss = one polygon per instance, no bitmasks. */
int amp_mask_target_loss(amp_ctx* ctx, int N, int K, const float* logits, float* dlogits, const float* rois, const int* cls,
                         const int* poly_id, const double* poly_xy, const int* poly_off, float* partial, unsigned char* target_out);
int amp_mask_targets_bitmask(amp_ctx* ctx, int N, const float* rois, const int* inst, const unsigned long long* rle_off,
                             const uint32_t* rle_counts, const int* rle_hw, unsigned int* scratch, size_t slot_words, int nslots,
                             unsigned char* target /* [N,28,28] */, int* overflow_flag);
int amp_mask_target_loss_fmt(amp_ctx* ctx, int N, int K, const float* logits, float* dlogits, const float* rois, const int* cls,
                             const int* inst, const double* poly_xy, const int* poly_off /* per polygon */,
                             const int* inst_poly_off /* [instances+1] or NULL: one polygon per instance */,
                             const unsigned long long* rle_off /* [instances+1] or NULL */, const unsigned char* target_in,
                             float* partial, unsigned char* target_out);

/* Host-side COCO RLE codec (all pointers HOST) --------------------------------------------------- */
int amp_rle_to_string(const uint32_t* cnts, int m, char* out, size_t cap, size_t* len);
/* n lists out of one pool (list i = pool[off[i] .. +len[i])) -> n strings back to back; string i = out[str_off[i] .. str_off[i+1]) */
int amp_rle_to_strings(const uint32_t* pool, const unsigned long long* off, const int* len, int n, char* out, size_t cap,
                       size_t* str_off /* [n+1] */);
/* Device op behind AMP_RLE_STRINGS (all pointers are device pointers): mask i = pool[off[i] .. off[i] + len[i]) -> its counts string at
 * str[str_off[i] .. str_off[i] + str_len[i]); *total = bytes needed (> cap: the tail was not written). maskApi.c rleToString, restated. */
int amp_rle_strings_device(amp_ctx* ctx, const uint32_t* pool, const unsigned long long* off, const int* len, int n, char* str,
                           unsigned long long cap, unsigned long long* str_off, int* str_len, unsigned long long* total);
int amp_rle_from_string(const char* s, size_t len, uint32_t* cnts, int cap, int* m_out);
int amp_rle_encode(const uint8_t* mask_colmajor, int h, int w, uint32_t* cnts, int cap, int* m_out);
int amp_rle_decode(const uint32_t* cnts, int m, int h, int w, uint8_t* mask_colmajor);
int amp_rle_area(const uint32_t* cnts, int m, unsigned long long* area);
int amp_rle_iou(const uint32_t* dt, int md, const uint32_t* gt, int mg, int iscrowd, double* iou);
/* all-pairs IoU of two pools of run lists (pycocotools.mask.iou): out[d * ng + g]; bounding boxes are compared first (h = mask height) */
int amp_rle_iou_matrix(const uint32_t* dpool, const unsigned long long* doff, const int* dlen, int nd, const uint32_t* gpool,
                       const unsigned long long* goff, const int* glen, int ng, const unsigned char* iscrowd /* [ng] or NULL */, int h,
                       double* out);
/* Pixel classes of `npairs` mask pairs (a = apool run list pair_a[p], b = bpool run list pair_b[p]) in one call: |a AND b|, |a \ b|,
 * |b \ a|.  Replaces the per-match pycocotools.mask.merge(intersect=True) + area calls of det_seg_scores (ampis/analyze.py:315-321). */
int amp_rle_pair_overlap(const uint32_t* apool, const unsigned long long* aoff, const int* alen, const uint32_t* bpool,
                         const unsigned long long* boff, const int* blen, const int* pair_a, const int* pair_b, int npairs,
                         unsigned long long* inter, unsigned long long* only_a, unsigned long long* only_b);
/* Nearest-neighbour resize (+ horizontal mirror when flip) of a mask in the run-length domain: the runs of
 * flip(PIL.Image.resize(decode(cnts), (nw, nh), NEAREST)) -- what detectron2's ResizeTransform.apply_segmentation + HFlipTransform do to a bitmask
 * annotation -- without decoding (Pillow's ImagingScaleAffine pixel correspondence, restated).  cap >= nh * nw + 1 is always enough. */
int amp_rle_resize_nearest(const uint32_t* cnts, int m, int h, int w, int nh, int nw, int flip, uint32_t* out, int cap, int* m_out);
int amp_rle_merge2(const uint32_t* A, int ka, const uint32_t* B, int kb, int intersect, uint32_t* out, int cap, int* m_out);
/* polygon (k vertices, flat xy) -> runs of an h x w mask (pycocotools rleFrPoly / frPyObjects) */
int amp_rle_from_polygon(const double* xy, int k, int h, int w, uint32_t* cnts, int cap, int* m_out);

/* The model: DefaultPredictor(cfg) / predictor(img) ---------------------------------------------- */
typedef struct amp_model amp_model;
typedef struct amp_model_cfg {
    int num_classes;                 /* MODEL.ROI_HEADS.NUM_CLASSES */
    float pixel_mean[3], pixel_std[3];
    int pre_nms_topk, post_nms_topk; /* MODEL.RPN.{PRE,POST}_NMS_TOPK_TEST */
    float rpn_nms_thresh;            /* MODEL.RPN.NMS_THRESH */
    float score_thresh, nms_thresh;  /* MODEL.ROI_HEADS.{SCORE,NMS}_THRESH_TEST */
    int detections_per_image;        /* TEST.DETECTIONS_PER_IMAGE */
    float bbox_reg_weights[4];
    float mask_threshold;
    int max_batch, max_h, max_w;     /* capacity: largest (padded) network input */
    int max_out_hw;                  /* capacity: largest side of an output (original) image */
    size_t rle_pool_counts;          /* capacity of the RLE run pool in uint32 (0 = default) */
    /* training-mode forward (amp_model_forward_losses); train_enable = 0 skips its workspace */
    int train_enable;
    int pre_nms_topk_train, post_nms_topk_train;   /* MODEL.RPN.{PRE,POST}_NMS_TOPK_TRAIN (2000 / 1000) */
    int rpn_batch;                   /* MODEL.RPN.BATCH_SIZE_PER_IMAGE (256) */
    float rpn_pos_frac;              /* MODEL.RPN.POSITIVE_FRACTION (0.5) */
    float rpn_iou_lo, rpn_iou_hi;    /* MODEL.RPN.IOU_THRESHOLDS (0.3, 0.7) */
    int roi_batch;                   /* MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE (512) */
    float roi_fg_frac;               /* MODEL.ROI_HEADS.POSITIVE_FRACTION (0.25) */
    float roi_iou;                   /* MODEL.ROI_HEADS.IOU_THRESHOLDS (0.5) */
    int max_gt;                      /* capacity: ground-truth instances per batch */
    int max_poly_doubles;            /* capacity: polygon coordinates (doubles) per batch */
    /* backbone variant (inference; training a grouped backbone is not built): MODEL.RESNETS.{DEPTH, NUM_GROUPS, WIDTH_PER_GROUP,
     * STRIDE_IN_1X1}.  R50-FPN = 50/1/64/1 (default, also when resnet_depth == 0); X101-32x8d-FPN = 101/32/8/0. */
    int resnet_depth, num_groups, width_per_group, stride_in_1x1;
} amp_model_cfg;
/* Ground truth of one batch (all pointers HOST): instances of image b are [gt_off[b], gt_off[b+1]).  The mask of an instance is
 * EITHER polygons (flat x0,y0,x1,y1,... in input-image pixels; polygon q = poly_xy[poly_off[q] .. poly_off[q+1]); the polygons of
 * instance i are [inst_poly_off[i], inst_poly_off[i+1]), or polygon i alone when inst_poly_off is NULL) -- detectron2's
 * INPUT.MASK_FORMAT = 'polygon' -- OR a bitmask given as COCO run lengths of the instance's full-image mask at network-input
 * resolution (column-major, first run counts zeros; runs of instance i = rle_counts[rle_off[i] .. rle_off[i+1]), size rle_hw[i]) --
 * MASK_FORMAT = 'bitmask'.  An instance with runs uses them; rle_off == NULL: polygons only. */
typedef struct amp_gt {
    int B;
    const int* gt_off;
    const float* boxes;              /* [total,4] XYXY */
    const int* classes;              /* [total] in [0, num_classes) */
    const int* poly_off;             /* [polygons+1], in doubles */
    const double* poly_xy;
    const int* inst_poly_off;        /* [total+1] or NULL */
    const unsigned long long* rle_off; /* [total+1] or NULL */
    const uint32_t* rle_counts;
    const int* rle_hw;               /* [total,2] */
} amp_gt;
/* Host view of the detections of the last amp_model_infer call; pointers stay valid until the next call. */
typedef struct amp_dets {
    int B, D;                        /* images, capacity per image (detections_per_image) */
    const int* n;                    /* [B] detections per image, sorted by descending score */
    const float* boxes;              /* [B,D,4] XYXY in output-image pixels */
    const float* scores;             /* [B,D] */
    const int* classes;              /* [B,D] */
    const unsigned long long* rle_off; /* [B,D] offset of the mask's run lengths in rle_counts */
    const int* rle_len;              /* [B,D] number of runs (column-major, first run counts zeros) */
    const uint32_t* rle_counts;      /* NULL under AMP_RLE_STRINGS */
    const int* out_h; const int* out_w; /* [B] mask size */
    /* amp_model_set_rle_output(m, AMP_RLE_STRINGS | AMP_RLE_BOTH): the masks as COCO compressed-RLE "counts" strings, encoded on the
     * device -- the bytes pycocotools' mask.encode()["counts"] holds, what compress_pred stores (ampis/data_utils.py:275). NULL otherwise. */
    const char* rle_str;             /* string of detection (b, i) = rle_str[rle_str_off[b*D+i] .. + rle_str_len[b*D+i]) (no terminator) */
    const unsigned long long* rle_str_off; /* [B,D] */
    const int* rle_str_len;          /* [B,D] */
} amp_dets;
enum { AMP_RLE_COUNTS = 0, AMP_RLE_STRINGS = 1, AMP_RLE_BOTH = 2 };
/* What amp_model_infer hands back for the masks (default AMP_RLE_COUNTS: uint32 run lengths). With AMP_RLE_STRINGS the run lengths
 * never leave the device: 3-4x fewer bytes over PCIe and no host-side encoding (replaces the pycocotools call of data_utils.py:275). */
int  amp_model_set_rle_output(amp_model* m, int mode);
int  amp_model_cfg_default(amp_model_cfg* cfg);
int  amp_model_create(amp_ctx* ctx, const amp_model_cfg* cfg, amp_model** out);
void amp_model_destroy(amp_model* m);
size_t amp_model_workspace_bytes(amp_model* m);
int  amp_model_num_tensors(amp_model* m);
const char* amp_model_tensor_name(amp_model* m, int i);
/* data_h: host fp32 in the torch layout of detectron2's state_dict entry `name` (conv OIHW, linear [out,in]). */
int  amp_model_load_tensor(amp_model* m, const char* name, const float* data_h, const long long* shape, int ndim);
int  amp_model_finalize(amp_model* m);
/* imgs_bgr: uint8 [B,H,W,3], device pointer (imgs_on_host = 0) or host pointer (1). out_h_h/out_w_h: host [B] original
 * image sizes the boxes/masks are rescaled to (NULL = network input size). */
int  amp_model_infer(amp_model* m, const uint8_t* imgs_bgr, int imgs_on_host, int B, int H, int W, const int* out_h_h,
                     const int* out_w_h, amp_dets* out);
/* Training-mode forward (what `model(data)` returns to LossEvalHook, ampis/data_utils.py:111-122):
 * losses_h[5] = loss_cls, loss_box_reg, loss_mask, loss_rpn_cls, loss_rpn_loc. Sub-sampling is seeded (oracle/train.py). */
int  amp_model_forward_losses(amp_model* m, const uint8_t* imgs_bgr, int imgs_on_host, int B, int H, int W, const amp_gt* gt,
                              unsigned int seed, float losses_h[5]);
/* Same forward, then the backward pass: gradients of the summed losses w.r.t. every trainable tensor land in
 * the gradient arena (same offsets as the parameters; amp_model_grad_arena exposes it for the RCCL all-reduce of row a20). */
int  amp_model_forward_backward(amp_model* m, const uint8_t* imgs_bgr, int imgs_on_host, int B, int H, int W, const amp_gt* gt,
                                unsigned int seed, float losses_h[5]);
int  amp_model_grad_arena(amp_model* m, float** grads_dev, size_t* nfloats);
/* the SGD momentum buffers: one arena with the parameters' offsets (like amp_model_grad_arena).  The layout depends on the model
 * configuration only (classes, backbone), not on the capacity: a host can carry it over to a re-created model or into a checkpoint. */
int  amp_model_momentum_arena(amp_model* m, float** vel_dev, size_t* nfloats);
/* torch.optim.SGD step on every trainable tensor: g' = grad_scale*g + wd*p; v = mu*v + g'; p -= lr*v */
int  amp_model_sgd_step(amp_model* m, float lr, float momentum, float weight_decay, float grad_scale);
/* Current value (kind 0) / gradient (1) / SGD momentum (2) of one tensor, converted back to the torch layout of detectron2's
 * state_dict entry `name` (host). */
int  amp_model_get_tensor(amp_model* m, const char* name, int kind, float* out_h, size_t capacity_floats);
/* The inverse for the momentum: the buffer of `name` in torch layout -> the momentum arena (resuming a checkpoint that stores the
 * velocity under the parameters' names, independent of this build's arena layout). */
int  amp_model_set_momentum_tensor(amp_model* m, const char* name, const float* data_h, size_t nfloats);
/* Valid (h,w) of each image of the NEXT batches inside the common frame (host [B][2]); NULL = every image fills the frame. */
int  amp_model_set_image_sizes(amp_model* m, const int* hw_h, int B);
/* Device buffer of an intermediate stage of the last infer call (parity tests): dtype 0 f32, 1 i32, 2 u64. */
int  amp_model_get_tap(amp_model* m, const char* name, void** ptr, int* dtype, int* ndim, long long shape[5]);

/* Several batches in flight on one GPU (serving throughput) -------------------------------------------------------------
 * An amp_pipeline runs `depth` lanes -- models the caller created on contexts of their own (own stream, workspace and result buffers),
 * normally loaded with the same weights -- from one worker thread each, so that ONE calling thread keeps `depth` batches in flight:
 * the next batch's convolutions fill the chip while the previous batch sits in its latency-bound selection / NMS / paste kernels and
 * host read-backs (+9 % images/s at depth 2 on BASELINE configs[1]; splitting one call into micro-batches does not).
 * submit() returns at once with a ticket (lanes round-robin; at most `depth` uncollected tickets), wait(ticket) blocks until that
 * batch is done and hands out its amp_dets, valid until the lane's next submit.  Host images must stay alive until wait() returns.
 * Every batch runs the same kernels in the same order as amp_model_infer: results are bit-identical to the plain call. */
typedef struct amp_pipeline amp_pipeline;
int amp_pipeline_create(amp_model* const* models, int depth, amp_pipeline** out);
int amp_pipeline_submit(amp_pipeline* p, const uint8_t* imgs_bgr, int imgs_on_host, int B, int H, int W, const int* out_h_h,
                        const int* out_w_h, long long* ticket);
int amp_pipeline_wait(amp_pipeline* p, long long ticket, amp_dets* out);
int amp_pipeline_destroy(amp_pipeline* p);       /* waits for batches in flight; the models stay the caller's */

/* Multi-GPU exchange: RCCL over xGMI, one process and one communicator per GPU ---------------------------------------
 * Replaces what the reference gets from detectron2's launch()/DDP + NCCL: the gradient all-reduce under
 * `DefaultTrainer(cfg).train()` (ampis/data_utils.py:135, notebook cell 22) and `comm.synchronize()`
 * (ampis/data_utils.py:107).  librccl is dlopen'ed by the first amp_comm_* call; nothing here needs torch.
 * Rank 0 makes the id, the host hands its AMP_COMM_ID_BYTES bytes to every rank by any side channel (a file, a TCP store,
 * torch.distributed on gloo), every rank calls amp_comm_init.  Collectives run on a stream owned by the communicator and are
 * ordered against the context's stream by HIP events. */
#define AMP_COMM_ID_BYTES 128
int amp_comm_unique_id(unsigned char* id_h /* [AMP_COMM_ID_BYTES] */);
int amp_comm_init(amp_ctx* ctx, int rank, int world, const unsigned char* id_h);
int amp_comm_destroy(amp_ctx* ctx);                                   /* also done by amp_destroy */
int amp_comm_info(amp_ctx* ctx, int* rank, int* world /* 0: no communicator */, int* rccl_version);
/* every rank's stream work so far has completed on return (host-blocking) */
int amp_barrier(amp_ctx* ctx);
/* in-place all-reduce of a device buffer, ordered after the context's stream and complete before its later work */
enum { AMP_F32 = 0, AMP_F64 = 1, AMP_I32 = 2 };
enum { AMP_SUM = 0, AMP_MAX = 1 };
int amp_allreduce(amp_ctx* ctx, void* buf, size_t count, int dtype, int op);
/* timing of the last gradient exchange: exposed_ms = from the end of the backward pass (context stream) to the end of the last
 * bucket (communication stream), 0 when the exchange finished first; span_ms = first bucket ready -> last bucket reduced */
int amp_comm_stats(amp_ctx* ctx, float* exposed_ms, float* span_ms);
/* the context's stream waits (on the device) for every collective issued so far: call it before reading the gradient arena through
 * amp_model_grad_arena after an exchange (amp_model_sgd_step and amp_model_get_tensor do it themselves); no-op without a communicator */
int amp_comm_wait(amp_ctx* ctx);
/* per bucket of the last gradient exchange: microseconds on the communication stream from "inputs ready and stream free" to
 * "reduced" (with peers: includes waiting for the slowest rank to arrive); -1 for a bucket that was not exchanged */
int amp_comm_bucket_stats(amp_ctx* ctx, float us[/* AMP_GRAD_BUCKETS */ 7]);
/* in-place broadcast of `bytes` of device memory from rank `root`, ordered after the context's stream and complete before its
 * later work (ncclBroadcast on the communication stream) */
int amp_comm_broadcast(amp_ctx* ctx, void* buf, size_t bytes, int root);

/* The gradient arena is exchanged in AMP_GRAD_BUCKETS buckets, in the order the backward pass completes them:
 * 0 mask head, 1 box head, 2 RPN head, 3 FPN, 4 res5, 5 res4, 6 res3 (stem and res2 are frozen: -1).  Host-only helpers (no
 * device call), so that the plan can be checked -- and driven over gloo -- without a GPU. */
#define AMP_GRAD_BUCKETS 7
int amp_grad_bucket_of(const char* state_dict_name);
/* tensors (bucket[t], off[t], n[t]) -> per bucket, in issue order, the merged float ranges of the arena (ranges of one bucket
 * closer than max_gap floats are joined; the gap holds padding / frozen entries whose gradient is 0) */
int amp_plan_grad_buckets(int ntensors, const int* bucket, const size_t* off, const size_t* n, size_t max_gap, int cap,
                          int* out_bucket, size_t* out_off, size_t* out_n, int* out_count);
/* the plan of this model's arena: ranges in issue order (cap >= 64 is always enough) */
int amp_model_grad_buckets(amp_model* m, int cap, int* out_bucket, size_t* out_off, size_t* out_n, int* out_count);
/* mode 1 (default once the context has a communicator): amp_model_forward_backward hands every bucket to RCCL as soon as the
 * backward pass has completed it, so that the exchange overlaps the remaining weight- and data-gradient kernels, and
 * amp_model_sgd_step waits for the last bucket on the device.  mode 0: nothing is exchanged inside forward_backward; the host
 * calls amp_model_allreduce_grads (all buckets at once) or reduces amp_model_grad_arena itself. */
int amp_model_set_grad_overlap(amp_model* m, int mode);
int amp_model_allreduce_grads(amp_model* m);            /* refused when the current gradients were already exchanged */
/* 1 when every bucket of the current gradients has been handed to RCCL (by forward_backward or by amp_model_allreduce_grads) */
int amp_model_grads_exchanged(amp_model* m, int* exchanged);
/* rank `root`'s parameters (the whole arena: weights, biases, FrozenBN scale / shift) and SGD momentum to every rank: what torch
 * DDP's constructor does under DefaultTrainer (ampis/data_utils.py:135).  Call once after loading weights / resuming.
 *
 * amp_model_forward_backward on a communicator of more than one rank is a collective: every rank must call it for every step; a
 * rank whose step fails still completes the step's RCCL sequence, and then EVERY rank returns an error for that step (the failing
 * rank its own status, the others AMP_ERR_STATE) with the gradients discarded.
 * What that covers: failures that leave the HIP runtime usable -- bad arguments, capacity / out-of-memory refusals, the f16 range flag.
 * A HIP error is sticky: the failing rank can then no longer issue its remaining buckets or join the agreement, and its peers would wait
 * in a collective nobody joins.  RCCL has no timeout of its own, so on AMP_ERR_HIP the caller must end the job (the trainer does: it
 * raises, the launcher tears the group down); the library does not try to continue on a dead device. */
int amp_model_broadcast_params(amp_model* m, int root);

#ifdef __cplusplus
}
#endif
#endif /* AMPIS_HIP_H */
