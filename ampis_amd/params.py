"""Parameter table of Mask R-CNN R50-FPN (detectron2 `mask_rcnn_R_50_FPN_3x`, the model AMPIS names at
GETTING_STARTED.md:30 / notebook cell 20) and its seeded random initialisation.

Names and shapes are detectron2's state_dict names/shapes ([D2-KNOWLEDGE]) so that a
`model_final_*.pkl` / `*.pth` can be mapped 1:1 when one is supplied.  `pack_arena` converts
that dict into the flat fp32 device arena the HIP path reads (layouts in DESIGN.md §3).
"""
from collections import OrderedDict

import numpy as np

RES_STAGES = (("res2", 3, 64, 256, 1), ("res3", 4, 128, 512, 2), ("res4", 6, 256, 1024, 2), ("res5", 3, 512, 2048, 2))
# MODEL.RESNETS.{DEPTH, NUM_GROUPS, WIDTH_PER_GROUP, STRIDE_IN_1X1} of the detectron2 model-zoo backbones
ARCHS = {
    "R50": dict(depth=50, groups=1, width_per_group=64, stride_in_1x1=True),
    "R101": dict(depth=101, groups=1, width_per_group=64, stride_in_1x1=True),
    "X101": dict(depth=101, groups=32, width_per_group=8, stride_in_1x1=False),    # X-101-32x8d (BASELINE configs[4])
}


def arch_from_cfg(cfg):
    """MODEL.RESNETS.* of a (façade) cfg -> the arch dict MaskRCNN / param_shapes take."""
    r = cfg.MODEL.RESNETS
    return dict(depth=int(r.DEPTH), groups=int(r.NUM_GROUPS), width_per_group=int(r.WIDTH_PER_GROUP), stride_in_1x1=bool(r.STRIDE_IN_1X1))


def res_stages(arch="R50"):
    """(name, blocks, bottleneck width, out channels, stride, groups) per stage."""
    a = ARCHS[arch] if isinstance(arch, str) else arch
    blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[a["depth"]]
    width = a["groups"] * a["width_per_group"]
    return tuple((f"res{i + 2}", blocks[i], width << i, 256 << i, 1 if i == 0 else 2, a["groups"]) for i in range(4))
FPN_LEVELS = (2, 3, 4, 5)
FPN_IN = {2: 256, 3: 512, 4: 1024, 5: 2048}
FPN_CH = 256
NUM_ANCHORS = 3
POOL_BOX = 7
POOL_MASK = 14
FC_DIM = 1024
BN_EPS = 1e-5


def param_shapes(num_classes, arch="R50"):
    """OrderedDict name -> shape in detectron2/torch layout (conv OIHW, ConvTranspose IOHW, linear [out,in])."""
    K = int(num_classes)
    s = OrderedDict()

    def conv_bn(prefix, cout, cin, k):
        s[prefix + ".weight"] = (cout, cin, k, k)
        for f in ("weight", "bias", "running_mean", "running_var"):
            s[f"{prefix}.norm.{f}"] = (cout,)

    conv_bn("backbone.bottom_up.stem.conv1", 64, 3, 7)
    cin = 64
    for name, nblk, mid, cout, _stride, groups in res_stages(arch):
        for b in range(nblk):
            p = f"backbone.bottom_up.{name}.{b}"
            if b == 0:
                conv_bn(p + ".shortcut", cout, cin, 1)
            conv_bn(p + ".conv1", mid, cin, 1)
            conv_bn(p + ".conv2", mid, mid // groups, 3)
            conv_bn(p + ".conv3", cout, mid, 1)
            cin = cout
    for l in FPN_LEVELS:
        s[f"backbone.fpn_lateral{l}.weight"] = (FPN_CH, FPN_IN[l], 1, 1)
        s[f"backbone.fpn_lateral{l}.bias"] = (FPN_CH,)
        s[f"backbone.fpn_output{l}.weight"] = (FPN_CH, FPN_CH, 3, 3)
        s[f"backbone.fpn_output{l}.bias"] = (FPN_CH,)
    r = "proposal_generator.rpn_head."
    s[r + "conv.weight"] = (FPN_CH, FPN_CH, 3, 3)
    s[r + "conv.bias"] = (FPN_CH,)
    s[r + "objectness_logits.weight"] = (NUM_ANCHORS, FPN_CH, 1, 1)
    s[r + "objectness_logits.bias"] = (NUM_ANCHORS,)
    s[r + "anchor_deltas.weight"] = (NUM_ANCHORS * 4, FPN_CH, 1, 1)
    s[r + "anchor_deltas.bias"] = (NUM_ANCHORS * 4,)
    s["roi_heads.box_head.fc1.weight"] = (FC_DIM, FPN_CH * POOL_BOX * POOL_BOX)
    s["roi_heads.box_head.fc1.bias"] = (FC_DIM,)
    s["roi_heads.box_head.fc2.weight"] = (FC_DIM, FC_DIM)
    s["roi_heads.box_head.fc2.bias"] = (FC_DIM,)
    s["roi_heads.box_predictor.cls_score.weight"] = (K + 1, FC_DIM)
    s["roi_heads.box_predictor.cls_score.bias"] = (K + 1,)
    s["roi_heads.box_predictor.bbox_pred.weight"] = (4 * K, FC_DIM)
    s["roi_heads.box_predictor.bbox_pred.bias"] = (4 * K,)
    for i in range(1, 5):
        s[f"roi_heads.mask_head.mask_fcn{i}.weight"] = (FPN_CH, FPN_CH, 3, 3)
        s[f"roi_heads.mask_head.mask_fcn{i}.bias"] = (FPN_CH,)
    s["roi_heads.mask_head.deconv.weight"] = (FPN_CH, FPN_CH, 2, 2)  # ConvTranspose2d: [Cin, Cout, kH, kW]
    s["roi_heads.mask_head.deconv.bias"] = (FPN_CH,)
    s["roi_heads.mask_head.predictor.weight"] = (K, FPN_CH, 1, 1)
    s["roi_heads.mask_head.predictor.bias"] = (K,)
    return s


def count_params(num_classes, arch="R50"):
    return int(sum(int(np.prod(v)) for v in param_shapes(num_classes, arch).values()))


def init_params(num_classes, seed=0, style="d2", dtype=np.float32, arch="R50"):
    """Seeded random initialisation -> OrderedDict name -> np.ndarray (torch layout).

    style="d2":     detectron2's initialisers: c2_msra_fill (Kaiming normal, fan_out) for the
                    backbone and mask-head convs, c2_xavier_fill for FPN and the FCs, N(0,.01) RPN / cls_score,
                    N(0,.001) bbox_pred / mask predictor, zero biases, identity FrozenBN statistics.
    style="spread": same layer shapes, but non-trivial FrozenBN statistics, non-zero biases and wider predictor
                    weights, so that scores / deltas / mask logits are well spread and every epilogue term
                    (scale, shift, bias) is exercised by the parity tests.  Bit-parity with torch's RNG is not a
                    goal: oracle and HIP path read the same arrays.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = param_shapes(num_classes, arch)
    out = OrderedDict()
    spread = style == "spread"
    assert style in ("d2", "spread")
    for name, shp in shapes.items():
        leaf = name.rsplit(".", 1)[-1]
        if ".norm." in name:
            if leaf == "weight":
                v = rng.uniform(0.6, 1.0, shp) if spread else np.ones(shp)
                if spread and ".conv3.norm." in name:
                    v = v * 0.5  # damp the residual branch so activations stay O(1) through 16 blocks
            elif leaf == "bias":
                v = rng.normal(0, 0.1, shp) if spread else np.zeros(shp)
            elif leaf == "running_mean":
                v = rng.normal(0, 0.1, shp) if spread else np.zeros(shp)
            else:
                v = rng.uniform(0.8, 1.25, shp) if spread else np.ones(shp)
        elif leaf == "bias":
            v = rng.normal(0, 0.05, shp) if spread else np.zeros(shp)
        elif "rpn_head" in name:
            std = 0.01
            if spread and "conv.weight" not in name:
                std = 0.03
            v = rng.normal(0, std, shp)
        elif "cls_score" in name:
            v = rng.normal(0, 0.004 if spread else 0.01, shp)
        elif "bbox_pred" in name:
            v = rng.normal(0, 0.004 if spread else 0.001, shp)
        elif "mask_head.predictor" in name:
            v = rng.normal(0, 0.05 if spread else 0.001, shp)
        elif "fpn_" in name or "box_head.fc" in name:
            # c2_xavier_fill = kaiming_uniform_(a=1): U(-b, b), b = sqrt(3 / fan_in)
            fan_in = int(np.prod(shp[1:]))
            b = np.sqrt(3.0 / fan_in)
            v = rng.uniform(-b, b, shp)
        elif "deconv" in name:
            fan_out = shp[0] * shp[2] * shp[3]  # torch fan_out of an IOHW tensor: size(0) * receptive field
            v = rng.normal(0, np.sqrt(2.0 / fan_out), shp)
        else:
            # c2_msra_fill = kaiming_normal_(mode="fan_out", nonlinearity="relu")
            fan_out = shp[0] * int(np.prod(shp[2:]))
            v = rng.normal(0, np.sqrt(2.0 / fan_out), shp)
        out[name] = np.ascontiguousarray(v, dtype=dtype)
    return out
