"""detectron2.data façade: DatasetCatalog / MetadataCatalog (notebook cells 13, 16, 18, 28; ampis/visualize.py:152), the
DatasetMapper and the loaders AmpisTrainer names (ampis/data_utils.py:24,171-175).

A mapped sample is a dict: image_bgr uint8 [h,w,3] (after ResizeShortestEdge / RandomFlip), height / width of the ORIGINAL image,
and in training mode gt = dict(boxes f32 [G,4] XYXY, classes i64 [G], polygons list[G] of flat xy float64) in resized coordinates.
Ground truth: INPUT.MASK_FORMAT='polygon' (what the tutorial uses, notebook cell 20; one or several polygons per instance) and
'bitmask' (RLE / polygon segmentations turned into per-instance bitmasks at network-input resolution, kept as COCO run lengths:
what get_ddicts('binary' | 'label' | 'rle') emits, ampis/data_utils.py:394-433,482-525)."""
import types

import numpy as np


class _DatasetCatalog(dict):
    def register(self, name, func):
        assert callable(func), "You must register a function with `DatasetCatalog.register`!"
        assert name not in self, f"Dataset '{name}' is already registered!"
        self[name] = func

    def get(self, name):
        try:
            f = self[name]
        except KeyError as e:
            raise KeyError(f"Dataset '{name}' is not registered! Available datasets are: {', '.join(self.keys())}") from e
        return f()

    def list(self):
        return list(self.keys())

    def remove(self, name):
        self.pop(name)

    @property
    def data(self):   # notebook cell 13 prints list(DatasetCatalog.data.keys())
        return self


class Metadata(types.SimpleNamespace):
    name = "N/A"

    def as_dict(self):
        return dict(self.__dict__)

    def set(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
        return self

    def get(self, key, default=None):
        return getattr(self, key, default)


class _MetadataCatalog(dict):
    def get(self, name):
        assert len(name)
        if name not in self:
            self[name] = Metadata(name=name)
        return self[name]

    def list(self):
        return list(self.keys())

    def remove(self, name):
        self.pop(name)


DatasetCatalog = _DatasetCatalog()
MetadataCatalog = _MetadataCatalog()


def _transform_annotations_loop(annos, scale_x, scale_y, flip, new_w, new_h):
    """transform_annotations one instance at a time (the statement of the semantics; the tests hold the vectorised form against it)."""
    boxes, classes, polys = [], [], []
    for a in annos:
        if a.get("iscrowd", 0):
            continue
        seg = _polygon_segmentation(a)
        b = np.asarray(a["bbox"], dtype=np.float64).copy()
        if int(a.get("bbox_mode", 0)) == 1:   # XYWH_ABS -> XYXY_ABS
            b[2:] += b[:2]
        ps = [np.asarray(q, dtype=np.float64).reshape(-1).copy() for q in seg]
        b[0::2] *= scale_x; b[1::2] *= scale_y
        for p in ps:
            p[0::2] *= scale_x; p[1::2] *= scale_y
        if flip:
            b[0], b[2] = new_w - b[2], new_w - b[0]
            for p in ps:
                p[0::2] = new_w - p[0::2]
        b = np.clip(b, 0, [new_w, new_h, new_w, new_h])
        if b[2] - b[0] <= 1e-5 or b[3] - b[1] <= 1e-5:
            continue
        boxes.append(b.astype(np.float32))
        classes.append(int(a["category_id"]))
        polys.append(ps[0] if len(ps) == 1 else ps)       # several polygons: their union is the instance (polygons_to_bitmask)
    return dict(boxes=np.asarray(boxes, np.float32).reshape(-1, 4), classes=np.asarray(classes, np.int64), polygons=polys)


def _polygon_segmentation(a):
    """The polygon list of one annotation under INPUT.MASK_FORMAT='polygon'; detectron2's message when it is something else."""
    seg = a.get("segmentation")
    if isinstance(seg, dict):
        raise ValueError("Cannot convert segmentation of type 'dict' (RLE) to PolygonMasks: the dataset was registered with "
                         "mask_format='bitmask' annotations -- set cfg.INPUT.MASK_FORMAT = 'bitmask'")
    if not isinstance(seg, (list, tuple)) or len(seg) == 0:
        raise ValueError(f"Cannot convert segmentation of type '{type(seg).__name__}' to PolygonMasks: expected a list of polygons")
    return seg


def transform_annotations_bitmask(annos, old_h, old_w, new_h, new_w, flip):
    """INPUT.MASK_FORMAT='bitmask' (detectron2 detection_utils: transform_instance_annotations + annotations_to_instances(mask_format=
    'bitmask') + filter_empty_instances): every instance ends as ONE full-image mask at network-input resolution --
      * an RLE segmentation is decoded, resized like the image's geometry with NEAREST interpolation (ResizeTransform.apply_segmentation:
        PIL) and mirrored with it;
      * a polygon segmentation has its vertices scaled / mirrored and is rasterised at the new size (polygons_to_bitmask);
    instances whose box or mask is empty are dropped.  The mask is handed on as COCO run lengths (`masks_rle`), never as a dense
    N x H x W tensor: the device builds the 28 x 28 targets from the runs (amp_mask_targets_bitmask)."""
    from . import rle
    boxes, classes, masks = [], [], []
    sx, sy = new_w / old_w, new_h / old_h
    for a in annos:
        if a.get("iscrowd", 0):
            continue
        b = np.asarray(a["bbox"], dtype=np.float64).copy()
        if int(a.get("bbox_mode", 0)) == 1:
            b[2:] += b[:2]
        b[0::2] *= sx; b[1::2] *= sy
        if flip:
            b[0], b[2] = new_w - b[2], new_w - b[0]
        b = np.clip(b, 0, [new_w, new_h, new_w, new_h])
        seg = a.get("segmentation")
        if isinstance(seg, dict):
            assert tuple(int(v) for v in seg["size"]) == (old_h, old_w), f"segmentation of size {seg['size']} on an image of {(old_h, old_w)}"
            # PIL NEAREST resize + mirror on the run lengths themselves (amp_rle_resize_nearest: the same pixels as decode -> Image.resize ->
            # [:, ::-1] -> encode, 0.2 ms instead of 4 ms per instance of a 1024 x 1536 micrograph)
            r = rle.resize_nearest(seg, new_h, new_w, flip) if ((new_h, new_w) != (old_h, old_w) or flip) else {"size": [old_h, old_w], "counts": rle.counts_to_string(rle._counts(seg))}
        elif isinstance(seg, (list, tuple)) and len(seg):
            ps = []
            for q in seg:
                p = np.asarray(q, dtype=np.float64).reshape(-1).copy()
                p[0::2] *= sx; p[1::2] *= sy
                if flip:
                    p[0::2] = new_w - p[0::2]
                ps.append(p.tolist())
            r = rle.merge(rle.frPyObjects(ps, new_h, new_w))
        else:
            raise ValueError(f"Cannot convert segmentation of type '{type(seg).__name__}' to BitMasks: expected polygons or an RLE dict")
        if b[2] - b[0] <= 1e-5 or b[3] - b[1] <= 1e-5 or rle.area(r) == 0:
            continue
        boxes.append(b.astype(np.float32)); classes.append(int(a["category_id"])); masks.append(r)
    return dict(boxes=np.asarray(boxes, np.float32).reshape(-1, 4), classes=np.asarray(classes, np.int64), polygons=[None] * len(masks), masks_rle=masks)


def parse_annotations(annos):
    """The annotation list of one image as arrays: XYXY boxes [n,4] float64, classes [n], every polygon's xy values in one flat float64
    array + their lengths.  None when a polygon has a dangling coordinate (the per-instance form decides what happens then).
    Independent of scale and flip, so a mapper parses a dataset dict once and re-uses the arrays in every epoch."""
    annos = [a for a in annos if not a.get("iscrowd", 0)]
    n = len(annos)
    segs = []
    for a in annos:
        seg = _polygon_segmentation(a)
        if len(seg) != 1:
            return None            # an instance made of several polygons: the per-instance form handles it
        segs.append(seg[0])
    if n == 0:
        return dict(n=0)
    lens = np.fromiter((len(s) for s in segs), dtype=np.int64, count=n)
    if np.any(lens % 2):
        return None
    b = np.array([a["bbox"] for a in annos], dtype=np.float64).reshape(n, 4)
    xywh = np.fromiter((int(a.get("bbox_mode", 0)) == 1 for a in annos), dtype=bool, count=n)   # XYWH_ABS -> XYXY_ABS
    b[xywh, 2:] += b[xywh, :2]
    flat = np.concatenate([np.asarray(s, dtype=np.float64).reshape(-1) for s in segs])           # every polygon starts at an even index
    classes = np.fromiter((int(a["category_id"]) for a in annos), dtype=np.int64, count=n)
    return dict(n=n, boxes=b, classes=classes, flat=flat, cuts=np.cumsum(lens)[:-1])


def transform_parsed(parsed, scale_x, scale_y, flip, new_w, new_h):
    """transform_annotations on the arrays of parse_annotations (which it leaves untouched)."""
    if parsed["n"] == 0:
        return dict(boxes=np.zeros((0, 4), np.float32), classes=np.zeros(0, np.int64), polygons=[], poly_flat=np.zeros(0, np.float64), poly_len=np.zeros(0, np.int64))
    b, flat = parsed["boxes"].copy(), parsed["flat"].copy()
    b[:, 0::2] *= scale_x; b[:, 1::2] *= scale_y
    flat[0::2] *= scale_x; flat[1::2] *= scale_y
    if flip:
        b[:, 0], b[:, 2] = new_w - b[:, 2], new_w - b[:, 0].copy()
        flat[0::2] = new_w - flat[0::2]
    b = np.clip(b, 0, [new_w, new_h, new_w, new_h])
    keep = ~((b[:, 2] - b[:, 0] <= 1e-5) | (b[:, 3] - b[:, 1] <= 1e-5))
    polys = np.split(flat, parsed["cuts"])
    lens = np.diff(np.concatenate([[0], parsed["cuts"], [len(flat)]]))
    if not keep.all():
        polys = [p for p, k in zip(polys, keep) if k]
        flat, lens = (np.concatenate(polys) if polys else np.zeros(0, np.float64)), lens[keep]
    # poly_flat / poly_len: the polygons once more as one array (what PackedGt concatenates per batch; `polygons` are views into it)
    return dict(boxes=b[keep].astype(np.float32), classes=parsed["classes"][keep], polygons=polys, poly_flat=flat, poly_len=lens)


def transform_annotations(annos, scale_x, scale_y, flip, new_w, new_h):
    """detectron2 detection_utils.transform_instance_annotations + annotations_to_instances + filter_empty_instances for
    polygon masks: boxes and polygon vertices are scaled (and mirrored), boxes clipped, empty boxes dropped.
    All instances of an image at once (a powder micrograph has hundreds: one numpy call per instance and step kept the loader threads
    on the interpreter lock, 6-15 ms per image); same float64 operations per value as the per-instance form."""
    parsed = parse_annotations(annos)
    if parsed is None:
        return _transform_annotations_loop(annos, scale_x, scale_y, flip, new_w, new_h)
    return transform_parsed(parsed, scale_x, scale_y, flip, new_w, new_h)


class DatasetMapper:
    """DatasetMapper(cfg, is_train): image read + ResizeShortestEdge (+ RandomFlip and ground truth in training mode)."""

    def __init__(self, cfg, is_train=True, seed=0):
        self.cfg = cfg
        self.is_train = is_train
        self._rng = np.random.default_rng(seed)
        self.force_size = None      # a fixed scale instead of the MIN_SIZE_TRAIN draw (tests)
        self._parsed = {}           # id(annotation list) -> (the list, parse_annotations of it): a dataset dict is parsed once, like
                                    # detectron2's loader serialises its dataset once (edits to a registered dict after that are not seen)

    def draw(self):
        """The random choices of the next image (scale, flip), drawn in call order from the mapper's generator.  The train loader
        draws them in its producer thread and hands the deterministic rest (`apply`) to worker threads."""
        c = self.cfg
        if not self.is_train:
            return int(c.INPUT.MIN_SIZE_TEST), False
        sizes = c.INPUT.MIN_SIZE_TRAIN
        sizes = (sizes,) if isinstance(sizes, int) else tuple(sizes)
        min_size = self.force_size if self.force_size is not None else int(sizes[self._rng.integers(len(sizes))])
        flip = str(c.INPUT.get("RANDOM_FLIP", "horizontal")) == "horizontal" and bool(self._rng.random() < 0.5)
        return min_size, flip

    def apply(self, dataset_dict, min_size, flip, defer=False):
        """Image read + ResizeShortestEdge (+ flip, ground truth): no random state, safe to run in a worker thread.
        defer=True (the train loader with a device uploader, round 4): the pixels are NOT resized or mirrored here -- d["image_bgr"] stays the
        decoded image and d["device_plan"] = (new_h, new_w, flip) tells the uploader what amp_resize_flip_u8 has to do; the annotations are
        transformed for the new size as always.  The frame the network sees is byte for byte the one the host path stacks
        (tests/test_train_input_gpu.py)."""
        from .engine.defaults import read_image_bgr, resize_shortest_edge, shortest_edge_size
        c = self.cfg
        img = dataset_dict["image_bgr"] if "image_bgr" in dataset_dict else read_image_bgr(dataset_dict["file_name"])
        h, w = img.shape[:2]
        max_size = int(c.INPUT.MAX_SIZE_TRAIN) if self.is_train else int(c.INPUT.MAX_SIZE_TEST)
        defer = bool(defer) and self.is_train
        if defer:
            out = img
            nh, nw = shortest_edge_size(h, w, min_size, max_size)
        else:
            out = resize_shortest_edge(np.ascontiguousarray(img), min_size, max_size)
            nh, nw = out.shape[:2]
        d = {k: v for k, v in dataset_dict.items() if k not in ("annotations", "image_bgr")}
        d["height"], d["width"] = h, w
        if defer:
            d["device_plan"] = (int(nh), int(nw), bool(flip))
        if self.is_train:
            if flip and not defer:
                out = out[:, ::-1]
            annos = dataset_dict.get("annotations", [])
            if str(c.INPUT.get("MASK_FORMAT", "polygon")) == "bitmask":
                d["gt"] = transform_annotations_bitmask(annos, h, w, nh, nw, flip)
                d["image_bgr"] = out if defer else np.ascontiguousarray(out)
                return d
            ent = self._parsed.get(id(annos))
            if ent is None or ent[0] is not annos:
                ent = (annos, parse_annotations(annos))
                self._parsed[id(annos)] = ent
            d["gt"] = (transform_parsed(ent[1], nw / w, nh / h, flip, nw, nh) if ent[1] is not None
                       else _transform_annotations_loop(annos, nw / w, nh / h, flip, nw, nh))
        d["image_bgr"] = out if defer else np.ascontiguousarray(out)
        return d

    def __call__(self, dataset_dict):
        return self.apply(dataset_dict, *self.draw())


def build_detection_test_loader(cfg, dataset_name, mapper=None):
    """Iterable with len(); yields one-image batches (list of one dict), like detectron2's test loader."""
    dicts = DatasetCatalog.get(dataset_name)
    mapper = mapper or DatasetMapper(cfg, False)

    class _Loader:
        def __len__(self):
            return len(dicts)

        def __iter__(self):
            for d in dicts:
                yield [mapper(d)]

    return _Loader()


def build_detection_train_loader(cfg, mapper=None, rank=0, world_size=1, seed=0, upload=None):
    """Infinite iterator of per-rank batches (lists of mapped dicts). TrainingSampler semantics: an infinite stream of seeded
    shuffles of the dataset indices, rank r takes elements r, r+world, ...; IMS_PER_BATCH is the GLOBAL batch.  MIN_SIZE_TRAIN is
    drawn per image, as detectron2's ResizeShortestEdge does; a batch of differently sized images is stacked top-left into a common
    frame with its per-image sizes (TrainModel.stack, amp_model_set_image_sizes).
    upload (optional, used with NUM_WORKERS > 0): callable(stacked uint8 frame) -> (device pointer, (B, H, W)); the collating thread
    calls it, so a batch reaches the training thread already in HBM (`CollatedBatch.device`)."""
    names = cfg.DATASETS.TRAIN
    dicts = [d for n in names for d in DatasetCatalog.get(n)]
    dicts = [d for d in dicts if len(d.get("annotations", [])) > 0]       # FILTER_EMPTY_ANNOTATIONS
    assert len(dicts) > 0, "empty training set"
    total = int(cfg.SOLVER.IMS_PER_BATCH)
    assert total % world_size == 0, "SOLVER.IMS_PER_BATCH must be divisible by the number of GPUs"
    per_rank = total // world_size
    mapper = mapper or DatasetMapper(cfg, True, seed=seed + 1000 * rank)
    def stream():
        g = np.random.default_rng(seed)
        while True:
            yield from g.permutation(len(dicts)).tolist()

    def plans():
        """Per batch: the (dataset dict, scale, flip) of every image, all random choices made here, in order."""
        s = stream()
        k = 0
        while True:
            plan = []
            while len(plan) < per_rank:
                idx = next(s)
                if k % world_size == rank:
                    plan.append((dicts[idx],) + tuple(mapper.draw()))
                k += 1
            yield plan

    workers = int(cfg.DATALOADER.get("NUM_WORKERS", 0)) if "DATALOADER" in cfg else 0
    if workers <= 0 or not hasattr(mapper, "apply"):
        return ([mapper.apply(*t) if hasattr(mapper, "apply") else mapper(t[0]) for t in plan] for plan in plans())
    return _prefetched(plans(), mapper, workers, upload=upload)


class CollatedBatch(list):
    """A batch (list of mapped dicts, what `model(batch)` takes) that also carries its stacked frame, per-image sizes and flattened
    annotations, prepared off the training thread."""
    collated = None
    device = None      # (device pointer, (B, H, W)) of the stacked frame when the loader was given an `upload` callable


def mapped_hw(d):
    """(h, w) of a mapped image as the network will see it: the deferred plan's size, else the pixels' own."""
    plan = d.get("device_plan")
    return (plan[0], plan[1]) if plan else tuple(d["image_bgr"].shape[:2])


def _collate(futures, upload=None):
    batch = CollatedBatch(f.result() for f in futures)
    if all("gt" in d for d in batch):
        from .engine.defaults import TrainModel
        batch.collated = TrainModel.collate(batch)
        if upload is not None:
            # deferred batches (device_plan): the uploader resizes / mirrors / stacks on the device; else it copies the host-stacked frame
            batch.device = upload.frames(batch) if batch.collated[0] is None else upload(batch.collated[0])
    return batch


PREFETCH_DEPTH = 3


def _prefetched(plans, mapper, workers, depth=PREFETCH_DEPTH, upload=None):
    """DATALOADER.NUM_WORKERS threads decode / resize / transform the images of the next `depth` batches while the GPU trains on the
    current one (detectron2 uses worker processes; PIL and numpy release the GIL for the heavy parts).  Order and content are exactly
    those of the sequential loader: every random choice was drawn by the producer before the work was handed out."""
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor
    q = queue.Queue(maxsize=depth)
    pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="amp-loader")
    collator = ThreadPoolExecutor(max_workers=1, thread_name_prefix="amp-collate")   # its own thread: it waits on the pool's futures
    stop = threading.Event()

    def produce():
        try:
            for plan in plans:
                defer = upload is not None and getattr(upload, "device_resize", False)
                futs = collator.submit(_collate, [pool.submit(mapper.apply, *t, defer) if defer else pool.submit(mapper.apply, *t) for t in plan], upload)
                while not stop.is_set():
                    try:
                        q.put(futs, timeout=0.2)
                        break
                    except queue.Full:
                        continue
                if stop.is_set():
                    return
        except BaseException as e:   # noqa: BLE001 -- surfaces in the consumer
            q.put(e)

    t = threading.Thread(target=produce, name="amp-loader-producer", daemon=True)
    t.start()

    def consume():
        try:
            while True:
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
                yield item.result()
        finally:
            stop.set()
            for ex in (pool, collator):
                try:
                    ex.shutdown(wait=False, cancel_futures=True)
                except Exception:   # noqa: BLE001 -- a generator finalised while the interpreter shuts down finds queue / threading torn down already
                    pass

    return consume()
