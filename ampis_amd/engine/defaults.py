"""DefaultPredictor / DefaultTrainer façade over the C-ABI model (notebook cells 22-28).

DefaultPredictor(cfg)(img) follows detectron2's DefaultPredictor.__call__: BGR uint8 HxWx3 -> ResizeShortestEdge
(INPUT.MIN_SIZE_TEST / MAX_SIZE_TEST, PIL bilinear on uint8, exactly detectron2's CPU-side transform) -> the HIP hot path
(amp_model_infer) -> {'instances': Instances(image_size=(H, W), pred_boxes, scores, pred_classes, pred_masks)} rescaled to
the original image.  There is no CPU fallback: cfg.MODEL.DEVICE must name a HIP device.
"""
import logging
import os

import numpy as np
import torch

from .. import _lib, checkpoint, params as P
from ..model import MaskRCNN
from ..structures import Boxes, Instances, RLEBitMasks

logger = logging.getLogger("ampis_amd")


def read_image_bgr(path):
    """cv2.imread stand-in (notebook cells 26/28): 8-bit image -> HxWx3 BGR uint8 (grayscale replicated)."""
    from PIL import Image
    im = Image.open(str(path))
    im = im.convert("RGB")
    return np.ascontiguousarray(np.asarray(im)[:, :, ::-1])


def resize_shortest_edge(img, min_size, max_size):
    """detectron2 ResizeShortestEdge.get_transform + ResizeTransform.apply_image (PIL bilinear on uint8)."""
    h, w = img.shape[:2]
    if min_size == 0:
        return img
    scale = min_size * 1.0 / min(h, w)
    newh, neww = (min_size, scale * w) if h < w else (scale * h, min_size)
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    neww, newh = int(neww + 0.5), int(newh + 0.5)
    if (newh, neww) == (h, w):
        return img
    from PIL import Image
    return np.asarray(Image.fromarray(img).resize((neww, newh), Image.BILINEAR))


def shortest_edge_size(h, w, min_size, max_size):
    """Output size of detectron2 ResizeShortestEdge.get_transform (same rounding)."""
    if min_size == 0:
        return h, w
    scale = min_size * 1.0 / min(h, w)
    newh, neww = (min_size, scale * w) if h < w else (scale * h, min_size)
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


def device_resize_shortest_edge(ctx, img, min_size, max_size, bufs):
    """Upload `img` (uint8 HxWx3) and resize it on the device with the PIL-exact kernel.  Returns (device pointer of the network
    input, (h, w), bufs); `bufs` caches the device buffers between calls."""
    import ctypes as C
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape[:2]
    h, w = shortest_edge_size(H, W, min_size, max_size)
    need = {"src": img.nbytes, "dst": h * w * 3, "tmp": int(_lib.lib().amp_resize_scratch_bytes(H, W, h, w))}
    for k, n in need.items():
        if bufs.get(k + "_n", 0) < n:
            if bufs.get(k):
                _lib.check(_lib.lib().amp_free(ctx.handle, C.c_void_p(bufs[k])), "amp_free")
            bufs[k] = ctx.malloc(n)
            bufs[k + "_n"] = n
    ctx.h2d(bufs["src"], img)
    if (h, w) == (H, W):
        return bufs["src"], (h, w), bufs
    _lib.check(_lib.lib().amp_resize_bilinear_u8(ctx.handle, C.c_void_p(bufs["src"]), H, W, C.c_void_p(bufs["dst"]), h, w, C.c_void_p(bufs["tmp"])),
               "amp_resize_bilinear_u8")
    return bufs["dst"], (h, w), bufs


def _device_index(dev):
    dev = str(dev)
    if dev.startswith("cpu"):
        raise _lib.AmpError("cfg.MODEL.DEVICE='cpu': ampis_amd has no CPU path for the Mask R-CNN hot path "
                            "(the CPU restatement under oracle/ is test infrastructure only)")
    return int(dev.split(":")[1]) if ":" in dev else 0


class DefaultPredictor:
    def __init__(self, cfg):
        self.cfg = cfg.clone()
        if len(cfg.DATASETS.TEST) == 0:
            logger.warning("cfg.DATASETS.TEST is empty")
        assert cfg.INPUT.FORMAT in ("RGB", "BGR"), cfg.INPUT.FORMAT
        self.input_format = cfg.INPUT.FORMAT
        self.num_classes = int(cfg.MODEL.ROI_HEADS.NUM_CLASSES)
        self.ctx = _lib.Context(_device_index(cfg.MODEL.DEVICE))
        w = str(cfg.MODEL.WEIGHTS)
        self.arch = P.arch_from_cfg(cfg)
        if w:
            self.params = checkpoint.load_checkpoint(w, self.num_classes, self.arch)
        else:
            logger.warning("cfg.MODEL.WEIGHTS is empty: using seeded random initialisation (like an un-loaded detectron2 model)")
            self.params = P.init_params(self.num_classes, seed=0, style="d2", arch=self.arch)
        self._model = None
        self._cap = (0, 0, 0)

    def _ensure(self, h, w, out_hw):
        hp, wp = (h + 31) // 32 * 32, (w + 31) // 32 * 32
        if self._model is None or hp > self._cap[0] or wp > self._cap[1] or out_hw > self._cap[2]:
            if self._model is not None:
                self._model.close()
            c = self.cfg
            cap = (max(hp, self._cap[0]), max(wp, self._cap[1]), max(out_hw, self._cap[2], 64))
            self._model = MaskRCNN(self.ctx, self.num_classes, max_batch=1, max_h=cap[0], max_w=cap[1], max_out_hw=cap[2],
                                   detections_per_image=int(c.TEST.DETECTIONS_PER_IMAGE),
                                   pre_nms_topk=int(c.MODEL.RPN.PRE_NMS_TOPK_TEST), post_nms_topk=int(c.MODEL.RPN.POST_NMS_TOPK_TEST),
                                   rpn_nms_thresh=float(c.MODEL.RPN.NMS_THRESH), score_thresh=float(c.MODEL.ROI_HEADS.SCORE_THRESH_TEST),
                                   nms_thresh=float(c.MODEL.ROI_HEADS.NMS_THRESH_TEST), pixel_mean=tuple(c.MODEL.PIXEL_MEAN),
                                   pixel_std=tuple(c.MODEL.PIXEL_STD), arch=self.arch)
            self._model.load_params(self.params)
            self._cap = cap
        return self._model

    def __call__(self, original_image):
        original_image = np.asarray(original_image)
        assert original_image.ndim == 3 and original_image.shape[2] == 3 and original_image.dtype == np.uint8, \
            "DefaultPredictor expects an HxWx3 uint8 image"
        if self.input_format == "RGB":
            original_image = original_image[:, :, ::-1]
        height, width = original_image.shape[:2]
        # ResizeShortestEdge on the device (PIL-exact bilinear, amp_resize_bilinear_u8): the image goes up once, as uint8
        dptr, (h, w), self._bufs = device_resize_shortest_edge(self.ctx, original_image, int(self.cfg.INPUT.MIN_SIZE_TEST),
                                                               int(self.cfg.INPUT.MAX_SIZE_TEST), getattr(self, "_bufs", {}))
        model = self._ensure(h, w, max(height, width))
        r = model.infer(device_ptr=dptr, shape=(1, h, w), out_sizes=[(height, width)])[0]
        inst = Instances((height, width))
        inst.pred_boxes = Boxes(torch.from_numpy(r["boxes"]))
        inst.scores = torch.from_numpy(r["scores"])
        inst.pred_classes = torch.from_numpy(r["classes"])
        inst.pred_masks = RLEBitMasks(r["masks"], (height, width))
        return {"instances": inst}


    def close(self):
        """Free the device side of the predictor (model, the stream() lanes, context)."""
        if getattr(self, "_pipe", None) is not None:
            self._pipe.close()
            self._pipe = None
            self._pipe_key = None
        if self._model is not None:
            self._model.close()
            self._model = None
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None

    def _model_kwargs(self):
        c = self.cfg
        return dict(detections_per_image=int(c.TEST.DETECTIONS_PER_IMAGE), pre_nms_topk=int(c.MODEL.RPN.PRE_NMS_TOPK_TEST),
                    post_nms_topk=int(c.MODEL.RPN.POST_NMS_TOPK_TEST), rpn_nms_thresh=float(c.MODEL.RPN.NMS_THRESH),
                    score_thresh=float(c.MODEL.ROI_HEADS.SCORE_THRESH_TEST), nms_thresh=float(c.MODEL.ROI_HEADS.NMS_THRESH_TEST),
                    pixel_mean=tuple(c.MODEL.PIXEL_MEAN), pixel_std=tuple(c.MODEL.PIXEL_STD), arch=self.arch)

    def stream(self, images, depth=2):
        """`predictor(img)` for every image of an iterable, with `depth` images in flight on the GPU (amp_pipeline: the next image's
        convolutions run while the previous one is in its selection / NMS / paste tail and its results travel to the host).  Yields
        the same dicts `__call__` returns, in order, bit for bit (ResizeShortestEdge here is the PIL resize the device kernel reproduces
        exactly).  The lanes are sized for the largest frame cfg.INPUT.{MIN,MAX}_SIZE_TEST allows and kept for later calls."""
        from ..model import InferPipeline
        c = self.cfg
        mx = (int(c.INPUT.MAX_SIZE_TEST) + 31) // 32 * 32
        key = (int(depth), mx)
        if getattr(self, "_pipe_key", None) != key:
            if getattr(self, "_pipe", None) is not None:
                self._pipe.close()
            self._pipe = InferPipeline(self.ctx.device if hasattr(self.ctx, "device") else _device_index(c.MODEL.DEVICE), self.num_classes, depth=int(depth),
                                       max_batch=1, max_h=mx, max_w=mx, max_out_hw=4096, **self._model_kwargs())
            self._pipe.load_params(self.params)
            self._pipe_key = key
        pipe = self._pipe
        pending = []

        def finish(entry):
            ticket, (height, width) = entry
            r = pipe.wait(ticket)[0]
            inst = Instances((height, width))
            inst.pred_boxes = Boxes(torch.from_numpy(r["boxes"]))
            inst.scores = torch.from_numpy(r["scores"])
            inst.pred_classes = torch.from_numpy(r["classes"])
            inst.pred_masks = RLEBitMasks(r["masks"], (height, width))
            return {"instances": inst}

        for original_image in images:
            original_image = np.asarray(original_image)
            assert original_image.ndim == 3 and original_image.shape[2] == 3 and original_image.dtype == np.uint8, \
                "DefaultPredictor expects HxWx3 uint8 images"
            if self.input_format == "RGB":
                original_image = original_image[:, :, ::-1]
            height, width = original_image.shape[:2]
            small = resize_shortest_edge(np.ascontiguousarray(original_image), int(c.INPUT.MIN_SIZE_TEST), int(c.INPUT.MAX_SIZE_TEST))
            if len(pending) == pipe.depth:
                yield finish(pending.pop(0))
            pending.append((pipe.submit(small[None], out_sizes=[(height, width)]), (height, width)))
        while pending:
            yield finish(pending.pop(0))


class TrainModel:
    """What `trainer.model` is: callable on a batch (list of mapped dicts).  In training mode (the default, and what
    LossEvalHook relies on: ampis/data_utils.py:116) it returns the dict of the five losses as floats."""

    def __init__(self, net, ctx, ensure=None):
        self.net, self.ctx = net, ctx
        self.training = True
        self._seed = 0
        self._ensure = ensure          # callable (h, w): makes sure `net` exists and can take a batch of that frame (DefaultTrainer._ensure_net)

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    @staticmethod
    def stack(batch):
        from ..data import mapped_hw
        if all("device_plan" in d for d in batch):      # deferred: the uploader builds the frame on the device (no host copy of 3 MB per image)
            hw = [mapped_hw(d) for d in batch]
            H, W = max(h for h, _ in hw), max(w for _, w in hw)
            return None, (None if all(h == H and w == W for h, w in hw) else hw)
        assert not any("device_plan" in d for d in batch), "a batch mixes deferred and host-resized images"
        hs = [d["image_bgr"].shape[0] for d in batch]
        ws = [d["image_bgr"].shape[1] for d in batch]
        H, W = max(hs), max(ws)
        imgs = np.zeros((len(batch), H, W, 3), dtype=np.uint8)
        for i, d in enumerate(batch):
            imgs[i, : hs[i], : ws[i]] = d["image_bgr"]
        sizes = None if (min(hs) == H and min(ws) == W) else list(zip(hs, ws))
        return imgs, sizes

    @staticmethod
    def collate(batch):
        """Common frame + per-image sizes + flattened annotations of a batch (what the C ABI takes).  The train loader does this in
        its worker threads and leaves the result on the batch (`CollatedBatch.collated`); `__call__` does it itself otherwise."""
        from ..model import PackedGt
        imgs, sizes = TrainModel.stack(batch)
        return imgs, sizes, PackedGt([d["gt"] for d in batch])

    def __call__(self, batch, backward=False, seed=None):
        assert self.training, "TrainModel is the training-mode surface; use DefaultPredictor for inference"
        imgs, sizes, gt = getattr(batch, "collated", None) or self.collate(batch)
        dev = getattr(batch, "device", None)          # the loader's collating thread has put the frame into HBM already (_Uploader)
        assert imgs is not None or dev is not None, "a deferred batch (device_plan) needs the loader's uploader"
        if self._ensure is not None:   # a validation batch (LossEvalHook, ampis/data_utils.py:116) may be larger than every training batch so far
            self._ensure(*(dev[1][1:] if imgs is None else imgs.shape[1:3]))
        self.net.set_image_sizes(sizes)
        if seed is None:
            self._seed += 1
            seed = 0x5EED0000 + self._seed
        if dev is not None:
            return self.net.forward_losses(None, gt, seed=seed, backward=backward, device_ptr=dev[0], shape=dev[1])
        return self.net.forward_losses(imgs, gt, seed=seed, backward=backward)


class _Uploader:
    """H2D of the next batches off the training thread: a context of its own (its own HIP stream, non-blocking) and a ring of device
    buffers; the train loader's collating thread calls it, the training thread passes the pointer to amp_model_forward_backward.
    The copy is complete (stream-synchronised) when __call__ returns, so the compute stream needs no event.  A buffer is re-used after
    `nbuf` batches: the loader holds at most PREFETCH_DEPTH finished batches plus the one being collated, the trainer one."""

    device_resize = True      # the train loader defers ResizeShortestEdge + flip + stacking to `frames` (AMP_HOST_TRAIN_INPUT=1: the host path)

    def __init__(self, device, nbuf):
        import threading
        self.ctx = _lib.Context(device)
        self.bufs = [(0, 0)] * nbuf       # (pointer, bytes)
        self.k = 0
        self._lock = threading.Lock()     # close() may come from the training thread while the collating thread is uploading
        self._stage = {}                  # "src" / "tmp": device staging of one decoded image and the resize scratch
        if os.environ.get("AMP_HOST_TRAIN_INPUT"):
            self.device_resize = False

    def _scratch(self, key, nbytes):
        ptr, cap = self._stage.get(key, (0, 0))
        if cap < nbytes:
            if ptr:
                self.ctx.free(ptr)
            ptr, cap = self.ctx.malloc(nbytes), nbytes
            self._stage[key] = (ptr, cap)
        return ptr

    def frames(self, batch):
        """The stacked frame of a deferred batch, built in HBM: every decoded image goes up once as it is, amp_resize_flip_u8 (PIL-exact
        resize + mirror) writes it into its slot of the zeroed frame -- what DatasetMapper + ImageList.from_tensors do on the host, byte for
        byte (tests/test_train_input_gpu.py).  An image that needs neither resize nor flip and fills the frame is copied straight into it."""
        import ctypes as C
        from ..data import mapped_hw
        hw = [mapped_hw(d) for d in batch]
        B, H, W = len(batch), max(h for h, _ in hw), max(w for _, w in hw)
        nbytes = B * H * W * 3
        with self._lock:
            if self.ctx is None:
                return None
            ptr, cap = self.bufs[self.k]
            if cap < nbytes:
                if ptr:
                    self.ctx.free(ptr)
                ptr, cap = self.ctx.malloc(nbytes), nbytes
                self.bufs[self.k] = (ptr, cap)
            L = _lib.lib()
            if any((h, w) != (H, W) for h, w in hw):
                _lib.check(L.amp_memset(self.ctx.handle, C.c_void_p(ptr), 0, nbytes), "amp_memset")       # the padding of the smaller images
            for b, d in enumerate(batch):
                img = np.ascontiguousarray(d["image_bgr"], dtype=np.uint8)
                h0, w0 = img.shape[:2]
                (h, w), flip = hw[b], bool(d["device_plan"][2])
                slot = ptr + b * H * W * 3
                if (h, w) == (h0, w0) and not flip and w == W:
                    self.ctx.h2d(slot, img)                                   # rows contiguous in the slot: no kernel
                    continue
                src = self._scratch("src", img.nbytes)
                self.ctx.h2d(src, img)
                tmp = self._scratch("tmp", int(L.amp_resize_scratch_bytes(h0, w0, h, w))) if (h, w) != (h0, w0) else 0
                _lib.check(L.amp_resize_flip_u8(self.ctx.handle, C.c_void_p(src), h0, w0, C.c_void_p(slot), W, h, w, int(flip), C.c_void_p(tmp) if tmp else None),
                           "amp_resize_flip_u8")
                self.ctx.sync()                                               # `src` is re-used by the next image
            self.ctx.sync()
            self.k = (self.k + 1) % len(self.bufs)
            return ptr, (B, H, W)

    def __call__(self, imgs):
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        with self._lock:
            if self.ctx is None:          # closed: the batch stays on the host (nobody will consume it)
                return None
            ptr, cap = self.bufs[self.k]
            if cap < imgs.nbytes:
                if ptr:
                    self.ctx.free(ptr)
                ptr, cap = self.ctx.malloc(imgs.nbytes), imgs.nbytes
                self.bufs[self.k] = (ptr, cap)
            self.ctx.h2d(ptr, imgs)           # returns when the bytes are in HBM
            self.k = (self.k + 1) % len(self.bufs)
            return ptr, tuple(int(v) for v in imgs.shape[:3])

    def close(self):
        with self._lock:
            if self.ctx is None:
                return
            for ptr, _ in list(self.bufs) + list(self._stage.values()):
                if ptr:
                    self.ctx.free(ptr)
            self.bufs, self._stage = [], {}
            self.ctx.close()
            self.ctx = None


def comm_backend_is_rccl():
    from ..utils import comm
    return comm.backend() == "rccl"


class DefaultTrainer:
    """detectron2's DefaultTrainer surface (notebook cell 22; subclassed by AmpisTrainer, ampis/data_utils.py:135-177):
    DefaultTrainer(cfg) -> .resume_or_load(resume=False) -> .train(); .model, .cfg, .iter, .max_iter, .storage, build_hooks().
    One process per GPU (engine.launch starts them); with torch.distributed initialised each rank trains on its shard of the
    global batch, and the gradient arena is all-reduced bucket by bucket over RCCL while the backward pass is still running
    (utils/comm.py; torch.distributed only hands the RCCL id to the ranks)."""

    def __init__(self, cfg):
        from ..data import build_detection_train_loader
        from ..utils import comm
        self.cfg = cfg
        self.iter = self.start_iter = 0
        self.max_iter = int(cfg.SOLVER.MAX_ITER)
        self.storage = None
        self.is_main_process = comm.is_main_process()
        self.world_size = comm.get_world_size()
        self.num_classes = int(cfg.MODEL.ROI_HEADS.NUM_CLASSES)
        dev = _device_index(cfg.MODEL.DEVICE) if ":" in str(cfg.MODEL.DEVICE) else (int(os.environ.get("LOCAL_RANK", "0")) if not str(cfg.MODEL.DEVICE).startswith("cpu") else _device_index(cfg.MODEL.DEVICE))
        self.ctx = _lib.Context(dev)
        if self.world_size > 1 and comm.backend() == "rccl":
            comm.attach_rccl(self.ctx)       # the gradient exchange and synchronize() run on RCCL inside the library from here on
        self.arch = P.arch_from_cfg(cfg)     # a grouped (ResNeXt) backbone is inference-only: amp_model_create refuses to train it
        self.params = P.init_params(self.num_classes, seed=max(int(cfg.get("SEED", -1)), 0), style="d2", arch=self.arch)
        self._net = None
        self._momentum = None                              # SGD velocity carried over a re-created net / read from a checkpoint
        self.model = TrainModel(None, self.ctx, ensure=self._ensure_net)
        self._cap = self._capacity_from_cfg()              # one allocation for everything the loaders can produce
        self._per_rank = int(cfg.SOLVER.IMS_PER_BATCH) // self.world_size
        from ..data import PREFETCH_DEPTH
        workers = int(cfg.DATALOADER.get("NUM_WORKERS", 0)) if "DATALOADER" in cfg else 0
        self._dev, self._workers, self._loader_epoch = dev, workers, 0
        self._uploader = self.data_loader = None
        self._build_loader()
        self._hooks = []
        self._need_broadcast = self.world_size > 1 and comm.backend() == "rccl"     # rank 0's weights and momentum win (DDP's constructor)
        self.register_hooks(self.build_hooks())

    def _build_loader(self):
        from ..data import build_detection_train_loader, PREFETCH_DEPTH
        from ..utils import comm
        # batches arrive in HBM (48 MB / step otherwise copied by the training thread)
        self._uploader = _Uploader(self._dev, PREFETCH_DEPTH + 3) if self._workers > 0 else None
        self.data_loader = build_detection_train_loader(self.cfg, rank=comm.get_rank(), world_size=self.world_size,
                                                        seed=max(int(self.cfg.get("SEED", -1)), 0) + 7919 * self._loader_epoch, upload=self._uploader)
        self._loader_epoch += 1

    def close(self):
        """Stop the loader threads and free what the trainer holds on the device (uploader ring, net, context).  train() releases the
        loader and the uploader itself; the net stays until close() so that `trainer.model` can still be evaluated after training."""
        self._release_loader()
        if self._net is not None:
            self._sync_params()
            self._net.close()
            self._net = None
            self.model.net = None
        from ..utils import comm
        if comm._rccl_ctx is self.ctx:
            comm.detach_rccl()
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None

    def _release_loader(self):
        dl, self.data_loader = self.data_loader, None
        if dl is not None and hasattr(dl, "close"):
            dl.close()                       # generator: runs the loader's `finally` (stops the producer, shuts the pools down)
        if self._uploader is not None:
            self._uploader.close()           # serialised against an upload still running in the collating thread
            self._uploader = None

    def _broadcast_if_needed(self):
        if self._need_broadcast and self._net is not None:
            self._net.broadcast_params(0)
            self._need_broadcast = False

    # ---- model / weights ----
    def _capacity_from_cfg(self):
        """Upper bound (padded h, w) of every frame the train mapper can produce from cfg.DATASETS.TRAIN + TEST (the validation-loss
        hook maps TEST images with the same augmentation): ResizeShortestEdge of each image's (height, width) for every
        MIN_SIZE_TRAIN choice, both orientations kept apart.  Falls back to MAX_SIZE_TRAIN x MAX_SIZE_TRAIN when a dataset does not
        say how large its images are.  Sized once, the net is never re-created mid-run (which would reset nothing any more -- the
        momentum arena is carried over -- but costs seconds)."""
        from ..data import DatasetCatalog
        c = self.cfg
        mins = c.INPUT.MIN_SIZE_TRAIN
        mins = [int(mins)] if isinstance(mins, (int, float)) else [int(v) for v in mins]
        mx = int(c.INPUT.MAX_SIZE_TRAIN)
        hmax = wmax = 0
        try:
            for name in tuple(c.DATASETS.TRAIN) + tuple(c.DATASETS.TEST):
                for d in DatasetCatalog.get(name):
                    if "height" in d and "width" in d:
                        h0, w0 = int(d["height"]), int(d["width"])
                    elif "image_bgr" in d:
                        h0, w0 = d["image_bgr"].shape[:2]
                    else:
                        raise KeyError("height")
                    for m_ in mins:
                        h, w = shortest_edge_size(h0, w0, m_, mx)
                        hmax, wmax = max(hmax, h), max(wmax, w)
        except Exception:          # unknown dataset / no sizes: the square bound
            hmax = wmax = mx
        pad = lambda v: (int(v) + 31) // 32 * 32
        return (pad(max(hmax, 32)), pad(max(wmax, 32)))

    def _ensure_net(self, h, w):
        hp, wp = (h + 31) // 32 * 32, (w + 31) // 32 * 32
        cap = self._cap
        if self._net is not None and hp <= cap[0] and wp <= cap[1]:
            return
        if self._net is not None:
            logger.warning(f"a {h}x{w} batch exceeds the capacity {cap} derived from cfg: re-creating the net (weights and SGD momentum are carried over)")
            self._sync_params()
            self._momentum = self._net.momentum()
            self._net.close()
        c = self.cfg
        cap = (max(hp, cap[0]), max(wp, cap[1]))
        self._net = MaskRCNN(self.ctx, self.num_classes, max_batch=self._per_rank, max_h=cap[0], max_w=cap[1], max_out_hw=max(cap),
                             train=True, max_gt=self._per_rank * 2048, max_poly_doubles=self._per_rank * 2048 * 128,
                             pre_nms_topk_train=int(c.MODEL.RPN.PRE_NMS_TOPK_TRAIN), post_nms_topk_train=int(c.MODEL.RPN.POST_NMS_TOPK_TRAIN),
                             rpn_batch=int(c.MODEL.RPN.BATCH_SIZE_PER_IMAGE), roi_batch=int(c.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE),
                             pixel_mean=tuple(c.MODEL.PIXEL_MEAN), pixel_std=tuple(c.MODEL.PIXEL_STD), arch=self.arch)
        self._net.load_params(self.params)
        self._restore_momentum()
        self._broadcast_if_needed()
        self._cap = cap
        self.model.net = self._net

    def _restore_momentum(self):
        """self._momentum: a flat arena (carried over a re-created net of the same configuration) or a dict of per-parameter buffers
        in torch layout (a checkpoint)."""
        mom, self._momentum = self._momentum, None
        if mom is None:
            return
        if isinstance(mom, dict) and set(mom) == {"arena"}:       # files written before the per-name format: this build's flat arena
            mom = mom["arena"]
        if isinstance(mom, dict):
            unknown = self._net.load_momentum_dict(mom)
            if unknown:
                logger.warning(f"momentum buffers of {len(unknown)} tensors the model does not train were ignored: {', '.join(unknown[:6])}")
        else:
            self._net.momentum(mom)

    def _sync_params(self):
        if self._net is not None:
            self.params.update(self._net.state_dict())

    def resume_or_load(self, resume=True):
        """detectron2 DefaultTrainer.resume_or_load: resume=True continues from OUTPUT_DIR/last_checkpoint when there is one (weights,
        iteration, SGD momentum); otherwise (and with resume=False) cfg.MODEL.WEIGHTS is loaded with DetectionCheckpointer semantics
        -- tensors missing from the file or of another shape (the COCO heads under NUM_CLASSES = 1, everything but the backbone for an
        ImageNet R-50.pkl) keep their detectron2-style initialisation, with a warning naming them -- and training starts at iteration 0.
        A detectron2:// or http(s) path cannot be fetched here (no network): training then starts from the initialisation, with a warning."""
        w = str(self.cfg.MODEL.WEIGHTS)
        resumed = False
        if resume:
            last = os.path.join(self.cfg.OUTPUT_DIR, "last_checkpoint")
            if os.path.isfile(last):
                w = os.path.join(self.cfg.OUTPUT_DIR, open(last).read().strip())
                resumed = True
        if w and not w.startswith(("detectron2://", "http://", "https://")):
            self.params, self.load_report = checkpoint.load_checkpoint(w, self.num_classes, self.arch, init=self.params, strict=resumed, with_report=True)
            if resumed:
                it = checkpoint.checkpoint_iteration(w)
                self.start_iter = self.iter = (it + 1) if it is not None else 0
                mom = checkpoint.checkpoint_momentum(w)
                if mom:
                    self._momentum = mom            # per-parameter buffers, torch layout
        elif w:
            logger.warning(f"cfg.MODEL.WEIGHTS={w!r} needs a download; no network: training from the seeded random initialisation")
        self._need_broadcast = self.world_size > 1 and comm_backend_is_rccl()
        if self._net is not None:
            self._net.load_params(self.params)
            self._restore_momentum()
            self._broadcast_if_needed()

    def save_checkpoint(self, path):
        self._sync_params()
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        mom = self._net.momentum_dict() if self._net is not None else None      # under the parameters' names, torch layout
        checkpoint.save_checkpoint(path, self.params, iteration=self.iter, optimizer=mom)
        with open(os.path.join(os.path.dirname(path) or ".", "last_checkpoint"), "w") as f:
            f.write(os.path.basename(path))

    # ---- hooks ----
    def build_hooks(self):
        """[PeriodicCheckpointer, PeriodicWriter]: the writer is last (AmpisTrainer does hooks.insert(-1, LossEvalHook(...)))."""
        from .train_loop import PeriodicCheckpointer, PeriodicWriter
        out = str(self.cfg.OUTPUT_DIR)
        return [PeriodicCheckpointer(int(self.cfg.SOLVER.CHECKPOINT_PERIOD), out), PeriodicWriter(out, period=20)]

    def register_hooks(self, hooks):
        for h in hooks:
            if h is not None:
                h.trainer = self
        self._hooks.extend([h for h in hooks if h is not None])

    # ---- loop ----
    def lr_at(self, it):
        from .train_loop import warmup_multistep_lr
        s = self.cfg.SOLVER
        return warmup_multistep_lr(it, float(s.BASE_LR), tuple(s.STEPS), float(s.GAMMA), int(s.WARMUP_ITERS), float(s.WARMUP_FACTOR))

    def run_step(self):
        from ..utils import comm
        if self.data_loader is None:
            self._build_loader()
        from ..data import mapped_hw
        batch = next(self.data_loader)
        h = max(mapped_hw(d)[0] for d in batch)
        w = max(mapped_hw(d)[1] for d in batch)
        self._ensure_net(h, w)
        losses = self.model(batch, backward=True, seed=(self.iter * 7919 + comm.get_rank()) & 0x7FFFFFFF)
        scale = comm.all_reduce_gradients(self._net, self.ctx)
        lr = self.lr_at(self.iter)
        s = self.cfg.SOLVER
        self._net.sgd_step(lr, float(s.MOMENTUM), float(s.WEIGHT_DECAY), grad_scale=scale)
        total = sum(losses.values())
        if not np.isfinite(total):
            raise FloatingPointError(f"Loss became infinite or NaN at iteration={self.iter}!\nloss_dict = {losses}")
        self.storage.put_scalars(total_loss=total, lr=lr, **losses)

    def train(self):
        from .train_loop import EventStorage
        os.makedirs(str(self.cfg.OUTPUT_DIR), exist_ok=True)
        logger.info(f"Starting training from iteration {self.start_iter}")
        self.storage = EventStorage(self.start_iter)
        self.iter = self.start_iter
        for h in self._hooks:
            h.before_train()
        # The loader threads run Python between their numpy / PIL calls; a training thread that comes back from the library waits for the
        # interpreter lock for up to one switch interval (5 ms by default) per call -- several calls per step.  0.2 ms while training.
        import sys
        switch = sys.getswitchinterval()
        if self._uploader is not None:
            sys.setswitchinterval(min(switch, 2e-4))
        try:
            for self.iter in range(self.start_iter, self.max_iter):
                for h in self._hooks:
                    h.before_step()
                self.run_step()
                for h in self._hooks:
                    h.after_step()
                self.storage.step()
            # detectron2 TrainerBase.train: "self.iter == max_iter can be used by `after_train` to tell whether the training successfully
            # finished or failed due to exceptions" -- after a complete run trainer.iter is max_iter, not max_iter - 1
            if self.max_iter > self.start_iter:
                self.iter += 1
        finally:
            sys.setswitchinterval(switch)
            try:
                for h in self._hooks:
                    h.after_train()
            finally:
                self._release_loader()          # loader threads, pools and the uploader's device ring live for one train() only
        self._sync_params()
