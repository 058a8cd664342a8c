"""The resize in front of DefaultPredictor's forward (detectron2 ResizeShortestEdge -> PIL bilinear): oracle pinned against PIL on
the CPU; the HIP kernel (amp_resize_bilinear_u8) bit-exact against both on the GPU."""
import numpy as np
import pytest

SIZES = [(1024, 1536, 800, 1200), (240, 300, 160, 200), (97, 131, 200, 251), (64, 64, 64, 48), (50, 70, 20, 70), (33, 45, 100, 17)]


@pytest.mark.parametrize("size", SIZES[1:])
def test_oracle_resize_equals_pil(size):
    from PIL import Image
    from oracle import resize as R
    H, W, h, w = size
    img = np.random.default_rng(H + w).integers(0, 256, (H, W, 3), dtype=np.uint8)
    assert np.array_equal(R.resize_bilinear_u8(img, h, w), np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR)))


@pytest.mark.gpu
@pytest.mark.parametrize("size", SIZES)
def test_hip_resize_equals_pil(gpu_ctx, size):
    from PIL import Image
    from ampis_amd import ops
    H, W, h, w = size
    rng = np.random.default_rng(H * 7 + w)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    if H == 1024:   # a micrograph-like image as well as noise
        from ampis_amd import synth
        img = synth.micrograph(1, H, W)[0]
    got = ops.resize_bilinear_u8(gpu_ctx, img, h, w)
    assert got.shape == (h, w, 3) and np.array_equal(got, np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR)))


@pytest.mark.gpu
def test_predictor_device_resize_matches_host_resize(gpu_ctx):
    """DefaultPredictor resizes on the device; the network must see exactly the bytes the host (PIL) resize would produce."""
    from ampis_amd import synth
    from ampis_amd.engine.defaults import device_resize_shortest_edge, resize_shortest_edge
    img, _ = synth.micrograph(5, 240, 300)
    want = resize_shortest_edge(img, 160, 256)
    dptr, (h, w), keep = device_resize_shortest_edge(gpu_ctx, img, 160, 256, {})
    got = np.empty((h, w, 3), np.uint8)
    from ampis_amd._lib import check, lib
    import ctypes as C
    check(lib().amp_memcpy_d2h(gpu_ctx.handle, got.ctypes.data_as(C.c_void_p), C.c_void_p(dptr), got.nbytes))
    assert (h, w) == want.shape[:2] and np.array_equal(got, want)
