"""Seeded synthetic powder micrographs: the benchmark / parity input generator.

uint8 grayscale replicated to 3 identical BGR channels (the powder PNGs are mode 'L'; cv2.imread yields 3 channels,
notebook cell 26).  Background N(60, 12^2); ~Poisson(200) particles in [150, 280] (reference pickles: 193-277 per image)
with log-normal areas fitted to the particle-results statistics (median 3 952 px^2 at 1024x1536, sigma_lnA 1.45) and
0-3 satellites on each particle's rim.  Returns the image and the ground-truth instances (boxes, classes, radii).
"""
import numpy as np


def micrograph(index, h=1024, w=1024, seed=1234, mean_particles=200):
    rng = np.random.Generator(np.random.PCG64(seed + int(index)))
    img = rng.normal(60.0, 12.0, (h, w)).astype(np.float32)
    n = int(np.clip(rng.poisson(mean_particles), 0.75 * mean_particles, 1.4 * mean_particles))
    scale = (h * w) / (1024.0 * 1536.0)
    area = np.exp(rng.normal(np.log(3952.0 * scale), 1.45, 4 * n))
    area = np.clip(area, 60.0, 29360.0 * scale)
    boxes, classes, radii, polygons = [], [], [], []
    ang32 = np.linspace(0.0, 2.0 * np.pi, 33)[:-1]
    placed = []
    for a in area:
        if len(placed) >= n:
            break
        r = float(np.sqrt(a / np.pi))
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        if any((cx - px) ** 2 + (cy - py) ** 2 < (0.8 * (r + pr)) ** 2 for px, py, pr in placed[-60:]):
            continue   # rejection: limit overlap
        placed.append((cx, cy, r))
    sats = []
    for cx, cy, r in placed:
        for _ in range(int(rng.integers(0, 4))):
            sr = float(np.sqrt(np.clip(np.exp(rng.normal(np.log(130.0 * scale), 1.2)), 12, 1500) / np.pi))
            th = rng.uniform(0, 2 * np.pi)
            sats.append((cx + (r + 0.3 * sr) * np.cos(th), cy + (r + 0.3 * sr) * np.sin(th), sr))
    for cls, items in ((0, placed), (1, sats)):
        for cx, cy, r in items:
            x0, x1 = int(max(0, np.floor(cx - r - 1))), int(min(w, np.ceil(cx + r + 2)))
            y0, y1 = int(max(0, np.floor(cy - r - 1))), int(min(h, np.ceil(cy + r + 2)))
            if x1 <= x0 or y1 <= y0:
                continue
            yy, xx = np.mgrid[y0:y1, x0:x1]
            d2 = (xx - cx) ** 2 + (yy - cy) ** 2
            inside = d2 <= r * r
            if not inside.any():
                continue
            val = rng.normal(190.0, 15.0) - 45.0 * d2 / (r * r)   # radial shading
            img[y0:y1, x0:x1] = np.where(inside, val, img[y0:y1, x0:x1])
            poly = np.stack([np.clip(cx + r * np.cos(ang32), 0, w), np.clip(cy + r * np.sin(ang32), 0, h)], axis=1)   # 32-gon
            boxes.append([poly[:, 0].min(), poly[:, 1].min(), poly[:, 0].max(), poly[:, 1].max()])   # bbox of the polygon,
            classes.append(cls)                                                                   # as get_ddicts does (data_utils.py:471)
            radii.append(r)
            polygons.append(poly.reshape(-1).astype(np.float64))
    # separable 5-tap Gaussian blur (sigma 1) + sensor noise
    k = np.exp(-0.5 * (np.arange(-2, 3) ** 2)).astype(np.float32)
    k /= k.sum()
    pad = np.pad(img, 2, mode="edge")
    img = sum(k[i] * pad[:, i:i + w] for i in range(5))
    img = sum(k[i] * img[i:i + h, :] for i in range(5))
    img = img + rng.normal(0.0, 4.0, (h, w)).astype(np.float32)
    u8 = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    bgr = np.repeat(u8[:, :, None], 3, axis=2)
    gt = dict(boxes=np.asarray(boxes, np.float32).reshape(-1, 4), classes=np.asarray(classes, np.int64),
              radii=np.asarray(radii, np.float32), polygons=polygons)
    return bgr, gt


def batch(n, h=1024, w=1024, seed=1234, first_index=0):
    out = [micrograph(first_index + i, h, w, seed) for i in range(n)]
    return np.stack([o[0] for o in out]), [o[1] for o in out]
