"""ctypes binding of libampis_hip.so (the C ABI declared in include/ampis_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libampis_hip.so")


class AmpError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("B", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad", "relu", "res_mode", "out_mode")]


class RpnLevels(C.Structure):
    _fields_ = [("nlevels", C.c_int), ("A", C.c_int), ("ld", C.c_int), ("pred", C.c_void_p * 5),
                ("h", C.c_int * 5), ("w", C.c_int * 5), ("stride", C.c_int * 5), ("anchor_size", C.c_int * 5)]


class FpnFeats(C.Structure):
    _fields_ = [("feat", C.c_void_p * 4), ("h", C.c_int * 4), ("w", C.c_int * 4), ("stride", C.c_int * 4),
                ("C", C.c_int)]


class ModelCfg(C.Structure):
    _fields_ = [("num_classes", C.c_int), ("pixel_mean", C.c_float * 3), ("pixel_std", C.c_float * 3),
                ("pre_nms_topk", C.c_int), ("post_nms_topk", C.c_int), ("rpn_nms_thresh", C.c_float),
                ("score_thresh", C.c_float), ("nms_thresh", C.c_float), ("detections_per_image", C.c_int),
                ("bbox_reg_weights", C.c_float * 4), ("mask_threshold", C.c_float),
                ("max_batch", C.c_int), ("max_h", C.c_int), ("max_w", C.c_int), ("max_out_hw", C.c_int),
                ("rle_pool_counts", C.c_size_t),
                ("train_enable", C.c_int), ("pre_nms_topk_train", C.c_int), ("post_nms_topk_train", C.c_int),
                ("rpn_batch", C.c_int), ("rpn_pos_frac", C.c_float), ("rpn_iou_lo", C.c_float), ("rpn_iou_hi", C.c_float),
                ("roi_batch", C.c_int), ("roi_fg_frac", C.c_float), ("roi_iou", C.c_float),
                ("max_gt", C.c_int), ("max_poly_doubles", C.c_int),
                ("resnet_depth", C.c_int), ("num_groups", C.c_int), ("width_per_group", C.c_int), ("stride_in_1x1", C.c_int)]


class Gt(C.Structure):
    _fields_ = [("B", C.c_int), ("gt_off", C.POINTER(C.c_int)), ("boxes", C.POINTER(C.c_float)),
                ("classes", C.POINTER(C.c_int)), ("poly_off", C.POINTER(C.c_int)), ("poly_xy", C.POINTER(C.c_double)),
                ("inst_poly_off", C.POINTER(C.c_int)), ("rle_off", C.POINTER(C.c_ulonglong)), ("rle_counts", C.POINTER(C.c_uint32)),
                ("rle_hw", C.POINTER(C.c_int))]


class ProfSummary(C.Structure):
    _fields_ = [("launches", C.c_longlong * 3), ("ms", C.c_double * 3), ("flops", C.c_double * 3), ("truncated", C.c_int)]


class ProfLaunch(C.Structure):
    _fields_ = [("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("slot", C.c_int)]


class Dets(C.Structure):
    _fields_ = [("B", C.c_int), ("D", C.c_int), ("n", C.POINTER(C.c_int)), ("boxes", C.POINTER(C.c_float)),
                ("scores", C.POINTER(C.c_float)), ("classes", C.POINTER(C.c_int)),
                ("rle_off", C.POINTER(C.c_ulonglong)), ("rle_len", C.POINTER(C.c_int)),
                ("rle_counts", C.POINTER(C.c_uint32)), ("out_h", C.POINTER(C.c_int)), ("out_w", C.POINTER(C.c_int)),
                ("rle_str", C.c_void_p), ("rle_str_off", C.POINTER(C.c_ulonglong)), ("rle_str_len", C.POINTER(C.c_int))]


_lib = None


def lib():
    """Load (once) and return the shared library; raises AmpError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AmpError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ampis_amd/csrc`). There is no CPU fallback for the hot path.")
    L = C.CDLL(LIB_PATH)
    L.amp_last_error.restype = C.c_char_p
    L.amp_stream.restype = C.c_void_p
    _declare(L)
    _lib = L
    return L


def _declare(L):
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    sig = {
        "amp_version": ([], i),
        "amp_init": ([i, vp, i, C.POINTER(vp)], i),
        "amp_timer_start": ([vp], i),
        "amp_prof_begin": ([vp, i], i),
        "amp_prof_end": ([vp, C.POINTER(ProfSummary)], i),
        "amp_prof_pause": ([vp, i], i),
        "amp_prof_launches": ([vp, C.POINTER(ProfLaunch), i, C.POINTER(i)], i),
        "amp_timer_stop": ([vp, C.POINTER(f)], i),
        "amp_destroy": ([vp], None),
        "amp_sync": ([vp], i),
        "amp_malloc": ([vp, C.c_size_t, C.POINTER(vp)], i),
        "amp_free": ([vp, vp], i),
        "amp_memcpy_h2d": ([vp, vp, vp, C.c_size_t], i),
        "amp_memcpy_d2h": ([vp, vp, vp, C.c_size_t], i),
        "amp_memset": ([vp, vp, i, C.c_size_t], i),
        "amp_conv2d_nhwc": ([vp, C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp], i),
        "amp_conv2d_nhwc_ex": ([vp, C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp], i),
        "amp_conv2d_nhwc_fmt": ([vp, C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, i], i),
        "amp_unsplit_rows": ([vp, vp, C.c_longlong, i, vp], i),
        "amp_group_expand_weights": ([vp, vp, i, i, i, i, vp], i),
        "amp_resize_scratch_bytes": ([i, i, i, i], C.c_size_t),
        "amp_resize_bilinear_u8": ([vp, vp, i, i, vp, i, i, vp], i),
        "amp_resize_flip_u8": ([vp, vp, i, i, vp, i, i, i, i, vp], i),
        "amp_compact_dets": ([vp, i, i, vp, vp, vp, vp, vp, vp, vp, vp], i),
        "amp_set_conv_mode": ([vp, i], i),
        "amp_get_conv_mode": ([vp], i),
        "amp_conv_range_flag": ([vp, i, C.POINTER(C.c_int)], i),
        "amp_split_weights": ([vp, vp, C.c_longlong, i, vp], i),
        "amp_bottleneck64_tail": ([vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, i, vp, vp], i),
        "amp_conv2d_grouped_nhwc": ([vp, C.POINTER(ConvDesc), i, vp, vp, vp, vp, vp, vp], i),
        "amp_conv2d_grouped_nhwc_fmt": ([vp, C.POINTER(ConvDesc), i, vp, vp, vp, vp, vp, vp, i], i),
        "amp_conv_wgrad_scratch_floats": ([C.POINTER(ConvDesc)], C.c_size_t),
        "amp_conv2d_wgrad": ([vp, C.POINTER(ConvDesc), vp, vp, vp, vp, vp, i], i),
        "amp_conv2d_wgrad_scaled": ([vp, C.POINTER(ConvDesc), vp, vp, vp, vp, vp, i, i, i], i),
        "amp_colsum": ([vp, vp, i, i, vp, vp, i], i),
        "amp_dgrad_weights": ([vp, vp, vp, i, i, i, i, vp], i),
        "amp_dgrad_weights_split": ([vp, vp, vp, i, i, i, i, vp], i),
        "amp_preprocess": ([vp, vp, i, i, i, i, i, C.POINTER(f), C.POINTER(f), vp, vp], i),
        "amp_model_set_image_sizes": ([vp, vp, i], i),
        "amp_maxpool3x3s2": ([vp, vp, i, i, i, i, vp], i),
        "amp_subsample2": ([vp, vp, i, i, i, i, vp], i),
        "amp_rpn_topk": ([vp, C.POINTER(RpnLevels), i, i, vp, i, vp, vp, vp], i),
        "amp_rpn_decode": ([vp, C.POINTER(RpnLevels), i, i, vp, vp, vp, i, i, i, vp, vp, vp], i),
        "amp_rpn_decode_sized": ([vp, C.POINTER(RpnLevels), i, i, vp, vp, vp, i, i, vp, i, vp, vp, vp], i),
        "amp_sort_gather": ([vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp], i),
        "amp_sort_gather_n": ([vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp], i),
        "amp_nms": ([vp, i, i, vp, vp, vp, f, i, vp, vp, vp], i),
        "amp_rpn_nms_scratch_words": ([i, i, i], C.c_size_t),
        "amp_rpn_nms_levels": ([vp, i, i, i, i, vp, vp, vp, f, i, vp, vp, vp, vp, vp, vp, vp], i),
        "amp_roi_align": ([vp, C.POINTER(FpnFeats), vp, vp, vp, i, i, vp, vp], i),
        "amp_roi_align_fmt": ([vp, C.POINTER(FpnFeats), vp, vp, vp, i, i, vp, vp, i], i),
        "amp_box_candidates": ([vp, vp, i, vp, vp, i, i, i, C.POINTER(f), f, i, i, vp, vp, i, vp, vp], i),
        "amp_box_candidates_sized": ([vp, vp, i, vp, vp, i, i, i, C.POINTER(f), f, i, i, vp, vp, vp, i, vp, vp], i),
        "amp_gather_dets": ([vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp], i),
        "amp_mask_prob": ([vp, vp, vp, i, i, vp], i),
        "amp_mask_deconv_predict": ([vp, vp, i, vp, vp, vp, vp, vp, i, vp], i),
        "amp_paste_rle": ([vp, vp, vp, vp, i, vp, vp, i, i, i, f, vp, vp, vp, C.c_ulonglong, vp, vp, vp, vp], i),
        "amp_paste_rle_sized": ([vp, vp, vp, vp, i, vp, vp, i, i, i, vp, f, vp, vp, vp, C.c_ulonglong, vp, vp, vp, vp, vp, C.c_ulonglong, vp], i),
        "amp_rle_to_string": ([vp, i, vp, C.c_size_t, C.POINTER(C.c_size_t)], i),
        "amp_rle_to_strings": ([vp, vp, vp, i, vp, C.c_size_t, vp], i),
        "amp_rle_from_string": ([vp, C.c_size_t, vp, i, C.POINTER(i)], i),
        "amp_rle_encode": ([vp, i, i, vp, i, C.POINTER(i)], i),
        "amp_rle_decode": ([vp, i, i, i, vp], i),
        "amp_rle_area": ([vp, i, C.POINTER(C.c_ulonglong)], i),
        "amp_rle_iou": ([vp, i, vp, i, i, C.POINTER(C.c_double)], i),
        "amp_rle_iou_matrix": ([vp, vp, vp, i, vp, vp, vp, i, vp, i, vp], i),
        "amp_rle_merge2": ([vp, i, vp, i, i, vp, i, C.POINTER(i)], i),
        "amp_rle_resize_nearest": ([vp, i, i, i, i, i, i, vp, i, C.POINTER(i)], i),
        "amp_rle_pair_overlap": ([vp, vp, vp, vp, vp, vp, vp, vp, i, vp, vp, vp], i),
        "amp_rle_from_polygon": ([vp, i, i, i, vp, i, C.POINTER(i)], i),
        "amp_model_cfg_default": ([C.POINTER(ModelCfg)], i),
        "amp_model_create": ([vp, C.POINTER(ModelCfg), C.POINTER(vp)], i),
        "amp_model_destroy": ([vp], None),
        "amp_model_workspace_bytes": ([vp], C.c_size_t),
        "amp_model_num_tensors": ([vp], i),
        "amp_model_tensor_name": ([vp, i], C.c_char_p),
        "amp_model_load_tensor": ([vp, C.c_char_p, vp, C.POINTER(C.c_longlong), i], i),
        "amp_model_finalize": ([vp], i),
        "amp_model_infer": ([vp, vp, i, i, i, i, vp, vp, C.POINTER(Dets)], i),
        "amp_model_forward_losses": ([vp, vp, i, i, i, i, C.POINTER(Gt), C.c_uint, C.POINTER(f)], i),
        "amp_anchor_labels": ([vp, C.POINTER(RpnLevels), i, vp, vp, i, f, f, vp, vp, vp, vp], i),
        "amp_rpn_sample_loss": ([vp, C.POINTER(RpnLevels), vp, i, vp, vp, vp, vp, vp, i, f, C.c_uint, vp, vp, vp], i),
        "amp_roi_sample": ([vp, i, vp, vp, i, vp, vp, vp, i, i, f, f, C.c_uint, vp, vp, vp, i, vp, vp, vp, vp, vp, i], i),
        "amp_box_loss": ([vp, i, i, i, vp, i, vp, vp, vp, vp, vp, vp, C.POINTER(f), i, vp], i),
        "amp_mask_target_loss": ([vp, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp], i),
        "amp_grouped_wgrad_scratch_floats": ([C.POINTER(ConvDesc)], C.c_size_t),
        "amp_conv2d_grouped_wgrad": ([vp, C.POINTER(ConvDesc), i, vp, vp, vp, vp, vp], i),
        "amp_conv2d_grouped_wgrad_fmt": ([vp, C.POINTER(ConvDesc), i, vp, vp, vp, vp, vp, i, i], i),
        "amp_group_dgrad_weights": ([vp, vp, vp, i, i, i, vp], i),
        "amp_sigmoid_focal_loss": ([vp, C.c_longlong, i, vp, vp, f, f, f, vp, vp, i, vp], i),
        "amp_mask_targets_bitmask": ([vp, i, vp, vp, vp, vp, vp, vp, C.c_size_t, i, vp, vp], i),
        "amp_mask_target_loss_fmt": ([vp, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp], i),
        "amp_model_forward_backward": ([vp, vp, i, i, i, i, C.POINTER(Gt), C.c_uint, C.POINTER(f)], i),
        "amp_model_grad_arena": ([vp, C.POINTER(vp), C.POINTER(C.c_size_t)], i),
        "amp_model_momentum_arena": ([vp, C.POINTER(vp), C.POINTER(C.c_size_t)], i),
        "amp_model_sgd_step": ([vp, f, f, f, f], i),
        "amp_model_get_tensor": ([vp, C.c_char_p, i, vp, C.c_size_t], i),
        "amp_model_set_momentum_tensor": ([vp, C.c_char_p, vp, C.c_size_t], i),
        "amp_roi_align_bwd": ([vp, vp, vp, vp, vp, i, vp, vp, i, i, vp], i),
        "amp_roi_align_bwd_batched": ([vp, vp, vp, vp, vp, i, vp, vp, i, i, vp, i], i),
        "amp_upsample2_bwd": ([vp, vp, vp, i, i, i, i], i),
        "amp_subsample2_bwd": ([vp, vp, vp, i, i, i, i], i),
        "amp_relu_mask": ([vp, vp, vp, C.c_size_t], i),
        "amp_small_k_dgrad": ([vp, vp, i, i, vp, i, vp, vp, C.c_size_t], i),
        "amp_deconv_grad_transpose": ([vp, vp, vp, i, i, i, i], i),
        "amp_sgd_update": ([vp, vp, vp, vp, C.c_size_t, f, f, f, f], i),
        "amp_pipeline_create": ([vp, i, C.POINTER(vp)], i),
        "amp_pipeline_submit": ([vp, vp, i, i, i, i, vp, vp, C.POINTER(C.c_longlong)], i),
        "amp_pipeline_wait": ([vp, C.c_longlong, C.POINTER(Dets)], i),
        "amp_pipeline_destroy": ([vp], i),
        "amp_comm_unique_id": ([vp], i),
        "amp_comm_init": ([vp, i, i, vp], i),
        "amp_comm_destroy": ([vp], i),
        "amp_comm_info": ([vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)], i),
        "amp_barrier": ([vp], i),
        "amp_allreduce": ([vp, vp, C.c_size_t, i, i], i),
        "amp_comm_stats": ([vp, C.POINTER(f), C.POINTER(f)], i),
        "amp_comm_bucket_stats": ([vp, C.POINTER(f)], i),
        "amp_comm_wait": ([vp], i),
        "amp_comm_broadcast": ([vp, vp, C.c_size_t, i], i),
        "amp_grad_bucket_of": ([C.c_char_p], i),
        "amp_plan_grad_buckets": ([i, vp, vp, vp, C.c_size_t, i, vp, vp, vp, C.POINTER(i)], i),
        "amp_model_grad_buckets": ([vp, i, vp, vp, vp, C.POINTER(i)], i),
        "amp_model_set_grad_overlap": ([vp, i], i),
        "amp_model_allreduce_grads": ([vp], i),
        "amp_model_grads_exchanged": ([vp, C.POINTER(i)], i),
        "amp_model_broadcast_params": ([vp, i], i),
        "amp_model_get_tap": ([vp, C.c_char_p, C.POINTER(vp), C.POINTER(i), C.POINTER(i), C.POINTER(C.c_longlong)], i),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = res
    L._amp_sig = sig


def check(status, what=""):
    if status != 0:
        msg = lib().amp_last_error().decode(errors="replace")
        raise AmpError(f"{what or 'libampis_hip'} failed with status {status}: {msg}")


class Context:
    """One amp_ctx (one HIP stream). borrow_stream=<hipStream_t int> (0 = legacy default stream) launches on
    that stream (e.g. torch's current stream); borrow_stream=None lets the library own a non-blocking stream."""

    def __init__(self, device=0, borrow_stream=None):
        self._h = C.c_void_p()
        if borrow_stream is None:
            st = lib().amp_init(int(device), None, 0, C.byref(self._h))
        else:
            st = lib().amp_init(int(device), C.c_void_p(int(borrow_stream)), 1, C.byref(self._h))
        check(st, "amp_init")
        self.device = int(device)

    CONV_F32, CONV_F16X3 = 0, 1

    @property
    def conv_mode(self):
        """Convolution arithmetic: CONV_F32 (fp32 MFMA) or CONV_F16X3 (split-operand f16 MFMA, fp32-equivalent; the default)."""
        return lib().amp_get_conv_mode(self._h)

    @conv_mode.setter
    def conv_mode(self, mode):
        mode = {"f32": 0, "f16x3": 1}.get(mode, mode)
        check(lib().amp_set_conv_mode(self._h, int(mode)), "amp_set_conv_mode")

    def conv_range_flag(self, clear=True):
        f = C.c_int()
        check(lib().amp_conv_range_flag(self._h, int(clear), C.byref(f)), "amp_conv_range_flag")
        return bool(f.value)

    def timer_start(self):
        check(lib().amp_timer_start(self._h), "amp_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        check(lib().amp_timer_stop(self._h, C.byref(ms)), "amp_timer_stop")
        return ms.value

    def prof_begin(self, max_launches=8192):
        check(lib().amp_prof_begin(self._h, int(max_launches)), "amp_prof_begin")

    def prof_pause(self, paused=True):
        check(lib().amp_prof_pause(self._h, int(bool(paused))), "amp_prof_pause")

    def prof_end(self):
        s = ProfSummary()
        check(lib().amp_prof_end(self._h, C.byref(s)), "amp_prof_end")
        return {"launches": list(s.launches), "ms": list(s.ms), "flops": list(s.flops), "truncated": bool(s.truncated)}

    def prof_launches(self):
        """After prof_end: the recorded launches in launch order, dicts of ms / flops / bytes (algorithmic) / M / N / K / slot."""
        n = C.c_int(0)
        check(lib().amp_prof_launches(self._h, None, 0, C.byref(n)), "amp_prof_launches")
        buf = (ProfLaunch * max(n.value, 1))()
        check(lib().amp_prof_launches(self._h, buf, n.value, C.byref(n)), "amp_prof_launches")
        return [{"ms": r.ms, "flops": r.flops, "bytes": r.bytes, "M": r.M, "N": r.N, "K": r.K, "slot": r.slot} for r in buf[:n.value]]

    # ---- RCCL communicator of this context (include/ampis_hip.h "Multi-GPU exchange") ----
    COMM_ID_BYTES = 128
    F32, F64, I32 = 0, 1, 2
    SUM, MAX = 0, 1

    @staticmethod
    def comm_unique_id():
        """Rank 0: the 128 bytes every rank passes to comm_init (hand them over by any side channel)."""
        buf = (C.c_ubyte * Context.COMM_ID_BYTES)()
        check(lib().amp_comm_unique_id(buf), "amp_comm_unique_id")
        return bytes(buf)

    def comm_init(self, rank, world, unique_id):
        assert len(unique_id) == self.COMM_ID_BYTES
        buf = (C.c_ubyte * self.COMM_ID_BYTES).from_buffer_copy(unique_id)
        check(lib().amp_comm_init(self._h, int(rank), int(world), buf), "amp_comm_init")

    def comm_info(self):
        """(rank, world, rccl_version); world == 0 when the context has no communicator."""
        r, w, v = C.c_int(), C.c_int(), C.c_int()
        check(lib().amp_comm_info(self._h, C.byref(r), C.byref(w), C.byref(v)), "amp_comm_info")
        return r.value, w.value, v.value

    def comm_destroy(self):
        check(lib().amp_comm_destroy(self._h), "amp_comm_destroy")

    def barrier(self):
        check(lib().amp_barrier(self._h), "amp_barrier")

    def allreduce(self, dptr, count, dtype=0, op=0):
        check(lib().amp_allreduce(self._h, C.c_void_p(int(dptr)), int(count), int(dtype), int(op)), "amp_allreduce")

    def comm_stats(self):
        e, s = C.c_float(), C.c_float()
        check(lib().amp_comm_stats(self._h, C.byref(e), C.byref(s)), "amp_comm_stats")
        return {"exposed_ms": e.value, "span_ms": s.value}

    def rowtab_stats(self):
        """Weight-gradient row-table cache of this context (wgrad.hip): hits, misses, generation switches, resident bytes."""
        v = (C.c_ulonglong * 4)()
        check(lib().amp_debug_rowtab_stats(self._h, v), "amp_debug_rowtab_stats")
        return {"hits": int(v[0]), "misses": int(v[1]), "generation_switches": int(v[2]), "resident_bytes": int(v[3])}

    def comm_wait(self):
        """The context's stream waits for every collective issued so far (before reading the gradient arena after an exchange)."""
        check(lib().amp_comm_wait(self._h), "amp_comm_wait")

    def comm_bucket_stats(self):
        """Microseconds of each of the 7 gradient buckets of the last exchange on the communication stream (-1: not exchanged)."""
        us = (C.c_float * 7)()
        check(lib().amp_comm_bucket_stats(self._h, us), "amp_comm_bucket_stats")
        return [round(float(x), 1) for x in us]

    def broadcast(self, dptr, nbytes, root=0):
        check(lib().amp_comm_broadcast(self._h, C.c_void_p(int(dptr)), int(nbytes), int(root)), "amp_comm_broadcast")

    def d2h(self, arr, src):
        check(lib().amp_memcpy_d2h(self._h, arr.ctypes.data_as(C.c_void_p), C.c_void_p(int(src)), arr.nbytes), "amp_memcpy_d2h")

    def malloc(self, nbytes):
        p = C.c_void_p()
        check(lib().amp_malloc(self._h, int(nbytes), C.byref(p)), "amp_malloc")
        return p.value

    def free(self, p):
        check(lib().amp_free(self._h, C.c_void_p(p)), "amp_free")

    def h2d(self, dst, arr):
        check(lib().amp_memcpy_h2d(self._h, C.c_void_p(dst), arr.ctypes.data_as(C.c_void_p), arr.nbytes), "amp_memcpy_h2d")

    @property
    def handle(self):
        return self._h

    @property
    def stream(self):
        return lib().amp_stream(self._h)

    def sync(self):
        check(lib().amp_sync(self._h), "amp_sync")

    def close(self):
        if self._h:
            lib().amp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ptr(t):
    """Device (or host) address of a torch tensor / numpy array / None as c_void_p."""
    if t is None:
        return C.c_void_p()
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    if hasattr(t, "ctypes"):
        return C.c_void_p(t.ctypes.data)
    return C.c_void_p(int(t))
