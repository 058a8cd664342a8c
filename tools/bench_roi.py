"""RoIAlign micro-benchmark at the bench's shape (B=8, 1024^2 -> p2..p5, 256 channels): realistic proposals vs degenerate ones that
all hit the same few cells (everything from L2): tells memory time from instruction / latency time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ampis_amd import ops

ctx = ops.torch_context(0)
if os.environ.get('AMP_ROI_LANES'):
    from ampis_amd import _lib
    _lib.lib().amp_debug_set_roi_lanes(int(os.environ['AMP_ROI_LANES']))
d = "cuda:0"
B, S, C = 8, 1024, 256
g = torch.Generator().manual_seed(0)
feats = [torch.randn(B, S // s, S // s, C, device=d) for s in (4, 8, 16, 32)]
fs = [ops.split_rows(ctx, f) for f in feats]


def rois_like_proposals(n_per_img):
    ctr = torch.rand(B * n_per_img, 2, generator=g) * S
    size = torch.exp(torch.randn(B * n_per_img, 2, generator=g) * 0.6 + 4.3).clamp(8, 800)      # median ~74 px
    r = torch.cat([ctr - size / 2, ctr + size / 2], 1).clamp(0, S).float()
    return r.to(d), (torch.arange(B * n_per_img) // n_per_img).int().to(d)


def run(name, rois, bidx, P, fmt, F):
    for _ in range(2):
        ops.roi_align(ctx, F, rois, bidx, P, fmt=fmt)
    torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(5):
        ops.roi_align(ctx, F, rois, bidx, P, fmt=fmt)
    ms = ctx.timer_stop() / 5
    out_gb = rois.shape[0] * P * P * C * 4 / 1e9
    print(f"{name:44s} P={P:2d} R={rois.shape[0]:5d}  {ms * 1e3:8.1f} us   (output {out_gb:.2f} GB -> {out_gb / ms:.2f} TB/s of writes)", flush=True)


for P, n in ((7, 1000), (14, 200)):
    rois, bidx = rois_like_proposals(n)
    same = rois.clone(); same[:] = torch.tensor([100.0, 100.0, 174.0, 174.0])
    for nm, F, fmt in (("fp32 maps", feats, 0), ("split maps -> split out", fs, 3)):
        run(f"{nm}, proposal-like boxes", rois, bidx, P, fmt, F)
        run(f"{nm}, every box the same 74 px box", same, bidx * 0, P, fmt, F)

# XCD-major RoI order (AMP_ROI_XCD / amp_debug_set_roi_xcd): same outputs, fewer L2 misses?
from ampis_amd import _lib
L = _lib.lib()
for P, n in ((7, 1000), (14, 200)):
    rois, bidx = rois_like_proposals(n)
    # proposals of a real image cluster on its particles: half of the boxes are jittered copies of 150 "objects" per image
    obj = rois.view(B, n, 4)[:, :150]
    pick = torch.randint(0, 150, (B, n // 2), generator=g).to(d)
    jit = (torch.randn(B, n // 2, 4, generator=g) * 4.0).to(d)
    clustered = rois.view(B, n, 4).clone()
    clustered[:, : n // 2] = (torch.gather(obj, 1, pick[..., None].expand(-1, -1, 4)) + jit).clamp(0, S)
    clustered = clustered.view(-1, 4).contiguous()
    for nm, R_ in (("uniform centres", rois), ("clustered on 150 objects / image", clustered)):
        outs = []
        for mode in (0, 2, 0, 2):
            L.amp_debug_set_roi_xcd(mode)
            run(f"split maps, {nm}, XCD-major order = {mode}", R_, bidx, P, 3, fs)
            outs.append(ops.roi_align(ctx, fs, R_, bidx, P, fmt=3)[0].clone())
        torch.cuda.synchronize()
        assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), "XCD-major order changed the result"
L.amp_debug_set_roi_xcd(1)

# sample tables in LDS + fma_mix decode (roi_align_split_tab_kernel, AMP_ROI_TAB / amp_debug_set_roi_tab) against roi_align_split_kernel
for P, n in ((7, 1000), (14, 200)):
    rois, bidx = rois_like_proposals(n)
    same = rois.clone(); same[:] = torch.tensor([100.0, 100.0, 174.0, 174.0])
    for nm, R_, bi in (("proposal-like boxes", rois, bidx), ("every box the same 74 px box", same, bidx * 0)):
        outs = []
        for mode in (0, 1, 0, 1):
            L.amp_debug_set_roi_tab(mode)
            run(f"split maps, {nm}, sample tables = {mode}", R_, bi, P, 3, fs)
            outs.append(ops.roi_align(ctx, fs, R_, bi, P, fmt=3)[0].clone())
        torch.cuda.synchronize()
        assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), "the table kernel changed the result"
L.amp_debug_set_roi_tab(1)
