"""detectron2.structures.{Boxes, BoxMode}: the members AMPIS touches (`.tensor` at
ampis/data_utils.py:276, BoxMode.XYXY_ABS at data_utils.py:29,426,472,518)."""
from enum import IntEnum, unique

import torch


@unique
class BoxMode(IntEnum):
    XYXY_ABS = 0
    XYWH_ABS = 1
    XYXY_REL = 2
    XYWH_REL = 3
    XYWHA_ABS = 4

    @staticmethod
    def convert(box, from_mode, to_mode):
        """XYXY_ABS <-> XYWH_ABS of one box (list / tuple) or an [N, 4] array; the other modes need the image size and are not used by
        AMPIS (ampis/data_utils.py:426,472,518 write XYXY_ABS only)."""
        import numpy as np
        from_mode, to_mode = BoxMode(from_mode), BoxMode(to_mode)
        if from_mode == to_mode:
            return box
        assert {from_mode, to_mode} == {BoxMode.XYXY_ABS, BoxMode.XYWH_ABS}, "only XYXY_ABS <-> XYWH_ABS"
        single = isinstance(box, (list, tuple))
        a = np.array(box, dtype=np.float64).reshape(-1, 4)
        if to_mode == BoxMode.XYWH_ABS:
            a[:, 2:] -= a[:, :2]
        else:
            a[:, 2:] += a[:, :2]
        return type(box)(a[0].tolist()) if single else a


class Boxes:
    def __init__(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4))
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, device):
        return Boxes(self.tensor.to(device=device))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size):
        h, w = box_size
        self.tensor[:, 0].clamp_(min=0, max=w)
        self.tensor[:, 1].clamp_(min=0, max=h)
        self.tensor[:, 2].clamp_(min=0, max=w)
        self.tensor[:, 3].clamp_(min=0, max=h)

    def nonempty(self, threshold=0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def scale(self, sx, sy):
        self.tensor[:, 0::2] *= sx
        self.tensor[:, 1::2] *= sy

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    def __iter__(self):
        yield from self.tensor

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    @property
    def device(self):
        return self.tensor.device
