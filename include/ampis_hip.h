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
/* stream may be NULL: the context then creates (and owns) a non-blocking HIP stream. */
int  amp_init(int device, void* hip_stream, amp_ctx** out);
void amp_destroy(amp_ctx* ctx);
int  amp_sync(amp_ctx* ctx);
void* amp_stream(amp_ctx* ctx);

/* Device memory helpers (so that hosts without torch can drive the library) ----------------- */
int amp_malloc(amp_ctx* ctx, size_t bytes, void** out);
int amp_free(amp_ctx* ctx, void* p);
int amp_memcpy_h2d(amp_ctx* ctx, void* dst, const void* src_h, size_t bytes);
int amp_memcpy_d2h(amp_ctx* ctx, void* dst_h, const void* src, size_t bytes);
int amp_memset(amp_ctx* ctx, void* dst, int value, size_t bytes);

/* Stage a9/a10/a11/a14/a16: implicit-GEMM convolution on fp32 MFMA ------------------------- */
typedef struct amp_conv_desc {
    int B, H, W, Cin;         /* input  [B,H,W,Cin]  (Cin % 4 == 0) */
    int Cout;                 /* weight [Cout][KH][KW][Cin] */
    int KH, KW, stride, pad;
    int relu;                 /* 1: y = max(y, 0) after the affine and the residual */
    int res_mode;             /* 0 none; 1 res[B,Ho,Wo,Cout] added; 2 res[B,Ho/2,Wo/2,Cout] nearest-upsampled x2 then added (FPN top-down) */
    int out_mode;             /* 0 NHWC [B,Ho,Wo,Cout]; 1 ConvTranspose 2x2 s2 scatter: Cout = 4*C2 ordered (ky,kx,co) -> y[B,2Ho,2Wo,C2] */
} amp_conv_desc;
/* y = act( conv(x, w) * scale[c] + shift[c] (+ res) ); scale may be NULL (= 1), shift may be NULL (= 0). */
int amp_conv2d_nhwc(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* w,
                    const float* scale, const float* shift, const float* res, float* y);

#ifdef __cplusplus
}
#endif
#endif /* AMPIS_HIP_H */
