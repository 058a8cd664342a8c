"""Python handle on amp_model (the C-ABI Mask R-CNN R50-FPN inference engine of libampis_hip.so).

Mirrors what `DefaultPredictor(cfg)` builds for AMPIS (notebook cell 24): a model with loaded weights that maps
BGR uint8 images to boxes / scores / classes / RLE masks.  No torch needed; numpy only marshals host buffers.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Dets, Gt, ModelCfg, check, lib
from . import rle as _rle

_TAP_DTYPES = {0: np.float32, 1: np.int32, 2: np.uint64, 3: np.int8, 4: np.uint8}


class PackedGt:
    """The amp_gt struct of a batch, flattened once.  Per image: dict(boxes [G,4], classes [G]) plus the instances' masks as
      * polygons: list[G]; an entry is one flat xy array (one polygon) or a list of flat xy arrays (several polygons: their union,
        detectron2 polygons_to_bitmask), or `poly_flat` / `poly_len` (the mapper's pre-flattened single polygons); and / or
      * masks_rle: list[G] of COCO RLE dicts of the instance's full-image mask at network-input resolution (None for an instance that
        has polygons): detectron2's INPUT.MASK_FORMAT = 'bitmask' (BitMasks.crop_and_resize).
    Keeps the arrays alive; reusable across steps (a data loader packs in its worker, not in the training loop)."""

    def __init__(self, gt):
        B = len(gt)
        off = np.zeros(B + 1, dtype=np.int32)
        for b, g in enumerate(gt):
            off[b + 1] = off[b] + len(g["boxes"])
        total = int(off[B])
        boxes = np.ascontiguousarray(np.concatenate([np.asarray(g["boxes"], np.float32).reshape(-1, 4) for g in gt]) if total else np.zeros((0, 4), np.float32))
        classes = np.ascontiguousarray(np.concatenate([np.asarray(g["classes"], np.int32).reshape(-1) for g in gt]) if total else np.zeros(0, np.int32), dtype=np.int32)
        ipoff = None
        if total and all("poly_flat" in g and len(g["poly_len"]) == len(g["boxes"]) for g in gt):
            # the mapper's transform left every image's polygons as one flat array already (data.transform_parsed)
            poff = np.zeros(total + 1, dtype=np.int32)
            poff[1:] = np.cumsum(np.concatenate([np.asarray(g["poly_len"], np.int64) for g in gt]))
            pxy = np.ascontiguousarray(np.concatenate([np.asarray(g["poly_flat"], np.float64).reshape(-1) for g in gt]))
            assert int(poff[-1]) == len(pxy), "poly_len does not add up to poly_flat"
        else:
            per_inst = []          # per instance: list of flat polygons (possibly empty when the instance has a bitmask)
            for g in gt:
                ps = g.get("polygons")
                n = len(g["boxes"])
                if ps is None:
                    ps = [[] for _ in range(n)]
                assert len(ps) == n, "one polygons entry per instance"
                for p in ps:
                    if p is None:
                        per_inst.append([])
                    elif isinstance(p, (list, tuple)) and (len(p) == 0 or not np.isscalar(p[0])):
                        per_inst.append([np.asarray(q, np.float64).reshape(-1) for q in p])
                    else:
                        per_inst.append([np.asarray(p, np.float64).reshape(-1)])
            flat = [q for inst in per_inst for q in inst]
            if all(len(inst) == 1 for inst in per_inst):
                pass                                                      # polygon index == instance index
            else:
                ipoff = np.zeros(total + 1, dtype=np.int32)
                if total:
                    ipoff[1:] = np.cumsum([len(inst) for inst in per_inst])
            poff = np.zeros(max(len(flat), total) + 1, dtype=np.int32)
            if flat:
                poff[1:len(flat) + 1] = np.cumsum([len(q) for q in flat])
                poff[len(flat) + 1:] = poff[len(flat)]
            pxy = np.ascontiguousarray(np.concatenate(flat) if flat else np.zeros(1, np.float64))
        # bitmask instances
        roff = rcnt = rhw = None
        if any(g.get("masks_rle") is not None for g in gt):
            runs, roff, rhw = [], np.zeros(total + 1, dtype=np.uint64), np.zeros((max(total, 1), 2), dtype=np.int32)
            i = 0
            for g in gt:
                mr = g.get("masks_rle") or [None] * len(g["boxes"])
                assert len(mr) == len(g["boxes"]), "one masks_rle entry per instance"
                for r in mr:
                    n = 0
                    if r is not None:
                        c = _rle._counts(r)
                        assert int(c.sum()) == int(r["size"][0]) * int(r["size"][1]), "RLE does not cover its size"
                        runs.append(c); n = len(c)
                        rhw[i] = (int(r["size"][0]), int(r["size"][1]))
                    roff[i + 1] = roff[i] + n
                    i += 1
            rcnt = np.ascontiguousarray(np.concatenate(runs) if runs else np.zeros(1, np.uint32), dtype=np.uint32)
        self.B, self.total = B, total
        self._keep = (off, boxes, classes, poff, pxy, ipoff, roff, rcnt, rhw)
        P = C.POINTER
        self.struct = Gt(B, off.ctypes.data_as(P(C.c_int)), boxes.ctypes.data_as(P(C.c_float)),
                         classes.ctypes.data_as(P(C.c_int)), poff.ctypes.data_as(P(C.c_int)), pxy.ctypes.data_as(P(C.c_double)),
                         ipoff.ctypes.data_as(P(C.c_int)) if ipoff is not None else None,
                         roff.ctypes.data_as(P(C.c_ulonglong)) if roff is not None else None,
                         rcnt.ctypes.data_as(P(C.c_uint32)) if rcnt is not None else None,
                         rhw.ctypes.data_as(P(C.c_int)) if rhw is not None else None)


RLE_COUNTS, RLE_STRINGS, RLE_BOTH = 0, 1, 2


class InferPipeline:
    """`depth` batches in flight on one GPU from one calling thread (include/ampis_hip.h amp_pipeline): `depth` models with the same
    weights on contexts of their own, driven by worker threads inside the library.

        pipe = InferPipeline(0, num_classes, depth=2, max_batch=8, max_h=1024, max_w=1024, detections_per_image=200)
        pipe.load_params(params)
        tickets = [pipe.submit(batch) for batch in first_two]
        for batch in rest: out = pipe.wait(tickets.pop(0)); tickets.append(pipe.submit(batch))      # results in submission order

    or simply `for out in pipe.map(batches)`.  Every batch's result is bit-identical to MaskRCNN.infer of that batch."""

    def __init__(self, device, num_classes, depth=2, rle="bytes", **model_kwargs):
        self.rle = rle
        self.ctxs = [_lib.Context(device) for _ in range(depth)]
        self.models = [MaskRCNN(c, num_classes, **model_kwargs) for c in self.ctxs]
        self._h = C.c_void_p()
        self._keep = {}                 # ticket -> host arrays that must outlive the batch
        self._open = False

    def load_params(self, params):
        for m in self.models:
            m.load_params(params)
            m.set_rle_output(RLE_COUNTS if self.rle == "counts" else RLE_STRINGS)
        handles = (C.c_void_p * len(self.models))(*[m._h for m in self.models])
        check(lib().amp_pipeline_create(handles, len(self.models), C.byref(self._h)), "amp_pipeline_create")
        self._open = True

    @property
    def depth(self):
        return len(self.models)

    def submit(self, images=None, out_sizes=None, device_ptr=None, shape=None):
        """Hand the next batch over and return its ticket at once.  images: uint8 [B,H,W,3] on the host (kept alive here until wait),
        or device_ptr + shape=(B,H,W) of a frame already in HBM (complete before this call; not overwritten before wait)."""
        if device_ptr is not None:
            B, H, W = shape
            p, on_host, keep = C.c_void_p(int(device_ptr)), 0, None
        else:
            keep = np.ascontiguousarray(images, dtype=np.uint8)
            assert keep.ndim == 4 and keep.shape[3] == 3, "images must be [B,H,W,3] uint8 BGR"
            B, H, W, _ = keep.shape
            p, on_host = keep.ctypes.data_as(C.c_void_p), 1
        oh = ow = None
        if out_sizes is not None:
            oh = (C.c_int * B)(*[int(s[0]) for s in out_sizes])
            ow = (C.c_int * B)(*[int(s[1]) for s in out_sizes])
        t = C.c_longlong()
        check(lib().amp_pipeline_submit(self._h, p, on_host, B, H, W, oh, ow, C.byref(t)), "amp_pipeline_submit")
        self._keep[t.value] = keep
        return t.value

    def wait_raw(self, ticket):
        d = Dets()
        try:
            check(lib().amp_pipeline_wait(self._h, int(ticket), C.byref(d)), "amp_pipeline_wait")
        finally:
            self._keep.pop(int(ticket), None)
        return d

    def wait(self, ticket):
        return MaskRCNN.unpack(self.wait_raw(ticket), self.rle)

    def map(self, batches):
        """Results of an iterable of host batches, in order, with `depth` of them in flight."""
        pending = []
        for b in batches:
            if len(pending) == self.depth:
                yield self.wait(pending.pop(0))
            pending.append(self.submit(b))
        while pending:
            yield self.wait(pending.pop(0))

    def close(self):
        if self._open:
            check(lib().amp_pipeline_destroy(self._h), "amp_pipeline_destroy")
            self._open = False
        for m in self.models:
            m.close()
        for c in self.ctxs:
            c.close()
        self.models, self.ctxs = [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MaskRCNN:
    def __init__(self, ctx, num_classes, max_batch=1, max_h=1344, max_w=1344, max_out_hw=4096,
                 detections_per_image=100, pre_nms_topk=1000, post_nms_topk=1000, rpn_nms_thresh=0.7,
                 score_thresh=0.05, nms_thresh=0.5, mask_threshold=0.5, pixel_mean=(103.530, 116.280, 123.675),
                 pixel_std=(1.0, 1.0, 1.0), rle_pool_counts=0, train=False, max_gt=16384, max_poly_doubles=16384 * 80,
                 pre_nms_topk_train=2000, post_nms_topk_train=1000, rpn_batch=256, roi_batch=512, arch="R50"):
        self.ctx = ctx
        cfg = ModelCfg()
        check(lib().amp_model_cfg_default(C.byref(cfg)), "amp_model_cfg_default")
        cfg.num_classes = int(num_classes)
        for i in range(3):
            cfg.pixel_mean[i] = float(pixel_mean[i])
            cfg.pixel_std[i] = float(pixel_std[i])
        cfg.pre_nms_topk, cfg.post_nms_topk = int(pre_nms_topk), int(post_nms_topk)
        cfg.rpn_nms_thresh, cfg.score_thresh, cfg.nms_thresh = float(rpn_nms_thresh), float(score_thresh), float(nms_thresh)
        cfg.detections_per_image = int(detections_per_image)
        cfg.mask_threshold = float(mask_threshold)
        pad = lambda v: (int(v) + 31) // 32 * 32
        cfg.max_batch, cfg.max_h, cfg.max_w, cfg.max_out_hw = int(max_batch), pad(max_h), pad(max_w), int(max_out_hw)
        cfg.rle_pool_counts = int(rle_pool_counts)
        cfg.train_enable = int(bool(train))
        cfg.max_gt, cfg.max_poly_doubles = int(max_gt), int(max_poly_doubles)
        cfg.pre_nms_topk_train, cfg.post_nms_topk_train = int(pre_nms_topk_train), int(post_nms_topk_train)
        cfg.rpn_batch, cfg.roi_batch = int(rpn_batch), int(roi_batch)
        from .params import ARCHS
        a = ARCHS[arch] if isinstance(arch, str) else arch      # MODEL.RESNETS.*: R50-FPN (default), R101, X101-32x8d
        cfg.resnet_depth, cfg.num_groups = int(a["depth"]), int(a["groups"])
        cfg.width_per_group, cfg.stride_in_1x1 = int(a["width_per_group"]), int(bool(a["stride_in_1x1"]))
        self.arch = a
        self.cfg = cfg
        self.num_classes = int(num_classes)
        self._h = C.c_void_p()
        check(lib().amp_model_create(ctx.handle, C.byref(cfg), C.byref(self._h)), "amp_model_create")
        self._finalized = False

    # ---- parameters ----
    def tensor_names(self):
        n = lib().amp_model_num_tensors(self._h)
        return [lib().amp_model_tensor_name(self._h, i).decode() for i in range(n)]

    def load_params(self, params):
        """params: mapping detectron2 state_dict name -> array-like (torch layout). Unknown keys are rejected."""
        for name in self.tensor_names():
            if name not in params:
                raise _lib.AmpError(f"missing parameter tensor {name!r}")
            a = np.ascontiguousarray(np.asarray(params[name]), dtype=np.float32)
            shape = (C.c_longlong * a.ndim)(*a.shape)
            check(lib().amp_model_load_tensor(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim),
                  f"amp_model_load_tensor({name})")
        check(lib().amp_model_finalize(self._h), "amp_model_finalize")
        self._finalized = True

    @property
    def workspace_bytes(self):
        return lib().amp_model_workspace_bytes(self._h)

    # ---- inference ----
    def infer_raw(self, images, out_sizes=None, device_ptr=None, shape=None):
        """Run the hot path. `images`: uint8 ndarray [B,H,W,3] (host) or None with device_ptr + shape=(B,H,W).
        Returns the ctypes Dets view (valid until the next call)."""
        if device_ptr is not None:
            B, H, W = shape
            p, on_host = C.c_void_p(int(device_ptr)), 0
        else:
            images = np.ascontiguousarray(images, dtype=np.uint8)
            assert images.ndim == 4 and images.shape[3] == 3, "images must be [B,H,W,3] uint8 BGR"
            B, H, W, _ = images.shape
            p, on_host = images.ctypes.data_as(C.c_void_p), 1
        oh = ow = None
        if out_sizes is not None:
            oh = (C.c_int * B)(*[int(s[0]) for s in out_sizes])
            ow = (C.c_int * B)(*[int(s[1]) for s in out_sizes])
        d = Dets()
        check(lib().amp_model_infer(self._h, p, on_host, B, H, W, oh, ow, C.byref(d)), "amp_model_infer")
        return d

    def infer(self, images=None, out_sizes=None, device_ptr=None, shape=None, rle="bytes"):
        """Returns a list (one per image) of dict(boxes f32[N,4], scores f32[N], classes i64[N],
        masks = list of COCO RLE dicts {'size':[h,w], 'counts': bytes}) -- the content compress_pred produces
        (ampis/data_utils.py:275-278). rle='counts' keeps the uncompressed uint32 run lengths instead."""
        self.set_rle_output(RLE_COUNTS if rle == "counts" else RLE_STRINGS)
        return self.unpack(self.infer_raw(images, out_sizes, device_ptr, shape), rle)

    @staticmethod
    def unpack(d, rle="bytes"):
        """amp_dets (ctypes view into the model's result buffers) -> the per-image dicts of infer(); copies everything it returns."""
        B, D = d.B, d.D
        n = np.ctypeslib.as_array(d.n, (B,)).copy()
        boxes = np.ctypeslib.as_array(d.boxes, (B, D, 4))
        scores = np.ctypeslib.as_array(d.scores, (B, D))
        classes = np.ctypeslib.as_array(d.classes, (B, D))
        if rle == "counts":
            off = np.ctypeslib.as_array(d.rle_off, (B, D))
            ln = np.ctypeslib.as_array(d.rle_len, (B, D))
            total = int((off[ln > 0] + ln[ln > 0]).max()) if (ln > 0).any() else 0
            pool = np.ctypeslib.as_array(d.rle_counts, (total,)) if total > 0 else np.zeros(1, dtype=np.uint32)
        else:       # the counts strings were encoded on the device (amp_model_set_rle_output): one bytes object, sliced per mask
            soff = np.ctypeslib.as_array(d.rle_str_off, (B, D))
            slen = np.ctypeslib.as_array(d.rle_str_len, (B, D))
            total = int((soff + slen.astype(np.uint64)).max()) if B * D else 0
            raw = C.string_at(d.rle_str, total) if total > 0 else b""
        out = []
        for b in range(B):
            k = int(n[b])
            h, w = int(d.out_h[b]), int(d.out_w[b])
            if rle == "counts":
                masks = [{"size": [h, w], "counts": pool[int(off[b, i]): int(off[b, i]) + int(ln[b, i])].copy()} for i in range(k)]
            else:
                so, sl = soff[b, :k].tolist(), slen[b, :k].tolist()
                masks = [{"size": [h, w], "counts": raw[so[i]: so[i] + sl[i]]} for i in range(k)]
            out.append(dict(boxes=boxes[b, :k].copy(), scores=scores[b, :k].copy(),
                            classes=classes[b, :k].astype(np.int64), masks=masks, image_size=(h, w)))
        return out

    def set_rle_output(self, mode):
        """What infer_raw's Dets carries for the masks: RLE_COUNTS (uint32 run lengths, default), RLE_STRINGS (COCO counts strings encoded
        on the device; the run lengths stay there) or RLE_BOTH."""
        if getattr(self, "_rle_mode", RLE_COUNTS) != mode:
            check(lib().amp_model_set_rle_output(self._h, int(mode)), "amp_model_set_rle_output")
            self._rle_mode = mode

    LOSS_NAMES = ("loss_cls", "loss_box_reg", "loss_mask", "loss_rpn_cls", "loss_rpn_loc")

    def forward_losses(self, images, gt, seed=0, backward=False, device_ptr=None, shape=None):
        """Training-mode forward: the loss dict `model(data)` returns to LossEvalHook (ampis/data_utils.py:111-122).
        images uint8 [B,H,W,3]; gt: per image dict(boxes [G,4], classes [G], polygons list of flat xy arrays).
        backward=True also runs the backward pass (gradients stay on the device until sgd_step / get_tensor).
        `gt` may be a PackedGt (the flat arrays of the C ABI, built once by a data-loader worker); images may already be on the
        device (images=None, device_ptr + shape=(B,H,W))."""
        if device_ptr is not None:
            B, H, W = shape
            img_p, on_host = C.c_void_p(int(device_ptr)), 0
        else:
            images = np.ascontiguousarray(images, dtype=np.uint8)
            B, H, W, _ = images.shape
            img_p, on_host = images.ctypes.data_as(C.c_void_p), 1
        packed = gt if isinstance(gt, PackedGt) else PackedGt(gt)
        assert packed.B == B
        out = (C.c_float * 5)()
        fn = lib().amp_model_forward_backward if backward else lib().amp_model_forward_losses
        check(fn(self._h, img_p, on_host, B, H, W, C.byref(packed.struct), int(seed) & 0xFFFFFFFF, out),
              "amp_model_forward_backward" if backward else "amp_model_forward_losses")
        return {n: float(out[i]) for i, n in enumerate(self.LOSS_NAMES)}

    def set_image_sizes(self, sizes):
        """sizes: list of (h, w) valid extents inside the common frame for the next batches, or None."""
        if sizes is None:
            check(lib().amp_model_set_image_sizes(self._h, None, 0), "amp_model_set_image_sizes")
        else:
            a = np.ascontiguousarray(np.asarray(sizes, dtype=np.int32).reshape(-1, 2))
            check(lib().amp_model_set_image_sizes(self._h, a.ctypes.data_as(C.c_void_p), len(a)), "amp_model_set_image_sizes")

    def sgd_step(self, lr, momentum=0.9, weight_decay=1e-4, grad_scale=1.0):
        check(lib().amp_model_sgd_step(self._h, float(lr), float(momentum), float(weight_decay), float(grad_scale)), "amp_model_sgd_step")

    def grad_arena(self):
        """(device pointer, number of floats) of the flat gradient arena (for the RCCL all-reduce)."""
        p, n = C.c_void_p(), C.c_size_t()
        check(lib().amp_model_grad_arena(self._h, C.byref(p), C.byref(n)), "amp_model_grad_arena")
        return p.value, n.value

    def momentum(self, value=None):
        """Read (value=None -> float32 ndarray) or write the SGD momentum arena; the layout depends only on (classes, backbone)."""
        p, n = C.c_void_p(), C.c_size_t()
        check(lib().amp_model_momentum_arena(self._h, C.byref(p), C.byref(n)), "amp_model_momentum_arena")
        if value is None:
            out = np.empty(n.value, dtype=np.float32)
            self.ctx.sync()
            self.ctx.d2h(out, p.value)
            return out
        value = np.ascontiguousarray(value, dtype=np.float32)
        if value.size != n.value:
            raise _lib.AmpError(f"momentum arena has {n.value} floats, got {value.size} (different NUM_CLASSES / backbone?)")
        self.ctx.h2d(p.value, value)

    def grad_buckets(self):
        """[(bucket, offset, n)] float ranges of the gradient arena in the order the backward pass completes (and exchanges) them."""
        cap = 512
        b, o, n, cnt = (C.c_int * cap)(), (C.c_size_t * cap)(), (C.c_size_t * cap)(), C.c_int()
        check(lib().amp_model_grad_buckets(self._h, cap, b, o, n, C.byref(cnt)), "amp_model_grad_buckets")
        return [(b[i], o[i], n[i]) for i in range(cnt.value)]

    def set_grad_overlap(self, on):
        check(lib().amp_model_set_grad_overlap(self._h, int(bool(on))), "amp_model_set_grad_overlap")

    def allreduce_grads(self):
        check(lib().amp_model_allreduce_grads(self._h), "amp_model_allreduce_grads")

    def grads_exchanged(self):
        """True when every bucket of the current gradients has been handed to RCCL (inside forward_backward, or by allreduce_grads)."""
        x = C.c_int()
        check(lib().amp_model_grads_exchanged(self._h, C.byref(x)), "amp_model_grads_exchanged")
        return bool(x.value)

    def broadcast_params(self, root=0):
        """Rank `root`'s parameters and SGD momentum to every rank of the context's communicator (DDP's constructor broadcast)."""
        check(lib().amp_model_broadcast_params(self._h, int(root)), "amp_model_broadcast_params")

    def get_tensor(self, name, grad=False, momentum=False):
        """Current value (or gradient, or SGD momentum buffer) of a parameter in detectron2 / torch layout."""
        from . import params as P
        shape = P.param_shapes(self.num_classes, self.arch)[name]
        out = np.empty(shape, dtype=np.float32)
        kind = 2 if momentum else int(bool(grad))
        check(lib().amp_model_get_tensor(self._h, name.encode(), kind, out.ctypes.data_as(C.c_void_p), out.size), "amp_model_get_tensor")
        return out

    def trainable_names(self):
        """Tensors the SGD step updates: everything but FrozenBN statistics and the frozen stem / res2 (FREEZE_AT = 2)."""
        from . import params as P
        return [k for k in P.param_shapes(self.num_classes, self.arch)
                if ".norm." not in k and not k.startswith(("backbone.bottom_up.stem", "backbone.bottom_up.res2"))]

    def momentum_dict(self):
        """SGD momentum buffers under the parameters' names, torch layout: what a checkpoint stores (independent of the arena layout)."""
        return {k: self.get_tensor(k, momentum=True) for k in self.trainable_names()}

    def load_momentum_dict(self, bufs):
        """Inverse of momentum_dict.  Sizes are validated; names the model does not train are reported, not loaded."""
        from . import params as P
        shapes = P.param_shapes(self.num_classes, self.arch)
        train = set(self.trainable_names())
        unknown = [k for k in bufs if k not in train]
        for k, v in bufs.items():
            if k not in train:
                continue
            a = np.ascontiguousarray(v, dtype=np.float32)
            if tuple(a.shape) != tuple(shapes[k]):
                raise _lib.AmpError(f"momentum buffer {k!r}: checkpoint {tuple(a.shape)} vs model {tuple(shapes[k])}")
            check(lib().amp_model_set_momentum_tensor(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), a.size), f"amp_model_set_momentum_tensor({k})")
        return unknown

    def state_dict(self):
        """All trainable tensors + the FrozenBN statistics they were loaded with are not tracked here; returns the trainable part."""
        from . import params as P
        return {k: self.get_tensor(k) for k in P.param_shapes(self.num_classes, self.arch) if ".norm." not in k}

    def tap(self, name):
        """Copy an intermediate device buffer of the last infer call to the host (parity tests)."""
        ptr, dt, nd = C.c_void_p(), C.c_int(), C.c_int()
        shape = (C.c_longlong * 5)()
        check(lib().amp_model_get_tap(self._h, name.encode(), C.byref(ptr), C.byref(dt), C.byref(nd), shape), "amp_model_get_tap")
        shp = tuple(int(shape[i]) for i in range(nd.value))
        if dt.value == 5:    # split operand format of AMP_CONV_F16X3: per 32 channels 32 f16 hi halves, then 32 f16 halves of lo * 2^11
            raw = np.empty(shp[:-1] + (shp[-1] // 32, 2, 32), dtype=np.float16)
            if raw.size:
                check(lib().amp_memcpy_d2h(self.ctx.handle, raw.ctypes.data_as(C.c_void_p), ptr, raw.nbytes), "amp_memcpy_d2h")
            r = raw.astype(np.float64)
            return (r[..., 0, :] + r[..., 1, :] / 2048.0).reshape(shp).astype(np.float32)
        a = np.empty(shp, dtype=_TAP_DTYPES[dt.value])
        if a.size:
            check(lib().amp_memcpy_d2h(self.ctx.handle, a.ctypes.data_as(C.c_void_p), ptr, a.nbytes), "amp_memcpy_d2h")
        return a

    def close(self):
        if self._h:
            lib().amp_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
