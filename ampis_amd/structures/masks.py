"""Mask containers.  BitMasks / PolygonMasks: the constructor + attributes AMPIS reads (ampis/structures.py:280,284,
427-429,564,675-677,743-745).  RLEBitMasks: what our predictor returns as `pred_masks` -- the N x H x W bool tensor of
detectron2 is never materialised on the device (210 MB / image); masks live as COCO RLE and decode lazily, so
the reference's `[RLE.encode(np.asfortranarray(x.to('cpu').numpy())) for x in pred.pred_masks]` (data_utils.py:275) still
works unmodified, while ampis_amd.data_utils.compress_pred takes the RLE directly."""
import numpy as np
import torch

from .. import rle as _rle


class BitMasks:
    def __init__(self, tensor):
        self.tensor = torch.as_tensor(np.asarray(tensor) if not isinstance(tensor, torch.Tensor) else tensor).to(torch.bool)
        assert self.tensor.dim() == 3, self.tensor.size()
        self.image_size = self.tensor.shape[1:]

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        if isinstance(item, int):
            return BitMasks(self.tensor[item].unsqueeze(0))
        return BitMasks(self.tensor[item])

    def __iter__(self):
        yield from self.tensor

    def to(self, *a, **k):
        return BitMasks(self.tensor.to(*a, **k))


class PolygonMasks:
    def __init__(self, polygons):
        assert isinstance(polygons, list)
        self.polygons = [[np.asarray(p, dtype=np.float64) for p in inst] for inst in polygons]

    def __len__(self):
        return len(self.polygons)

    def __getitem__(self, item):
        if isinstance(item, int):
            return PolygonMasks([self.polygons[item]])
        if isinstance(item, slice):
            return PolygonMasks(self.polygons[item])
        idx = np.asarray(item.cpu() if isinstance(item, torch.Tensor) else item)
        if idx.dtype == bool:
            idx = np.flatnonzero(idx)
        return PolygonMasks([self.polygons[int(i)] for i in idx])

    def __iter__(self):
        return iter(self.polygons)

    def to(self, *a, **k):
        return self


class _LazyMask:
    """One H x W mask that decodes from RLE on demand; quacks like the torch bool tensor row AMPIS iterates over."""

    def __init__(self, rle):
        self.rle = rle

    def to(self, *a, **k):
        return self

    def cpu(self):
        return self

    def numpy(self):
        return _rle.decode(self.rle).astype(bool)

    def __array__(self, dtype=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    @property
    def shape(self):
        return tuple(self.rle["size"])


class RLEBitMasks:
    def __init__(self, rles, image_size):
        self.rle = list(rles)
        self.image_size = tuple(image_size)

    def __len__(self):
        return len(self.rle)

    @property
    def shape(self):
        return (len(self.rle),) + self.image_size

    def __iter__(self):
        for r in self.rle:
            yield _LazyMask(r)

    def __getitem__(self, item):
        if isinstance(item, (int, np.integer)):
            return _LazyMask(self.rle[int(item)])
        if isinstance(item, slice):
            return RLEBitMasks(self.rle[item], self.image_size)
        idx = np.asarray(item.cpu() if isinstance(item, torch.Tensor) else item)
        if idx.dtype == bool:
            idx = np.flatnonzero(idx)
        return RLEBitMasks([self.rle[int(i)] for i in idx], self.image_size)

    def to(self, *a, **k):
        return self

    def cpu(self):
        return self

    def numpy(self):
        if not self.rle:
            return np.zeros((0,) + self.image_size, dtype=bool)
        return np.stack([_rle.decode(r).astype(bool) for r in self.rle])

    @property
    def tensor(self):
        return torch.from_numpy(self.numpy())
