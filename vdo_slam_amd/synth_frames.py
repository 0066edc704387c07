"""Synthetic KITTI-shaped RGB-D frames (SURVEY.md §8d "Synthetic inputs"): 1242x375 gray image
(value noise + random rectangles giving FAST corners), disparity-coded depth, dense optical flow
with exactly-zero holes, and an int32 semantic mask with box-shaped objects."""
from __future__ import annotations

import numpy as np

from .synth import KITTI_H, KITTI_K, KITTI_W

BF = 387.5744            # example/kitti-0000-0013.yaml
DEPTH_MAP_FACTOR = 256.0
TH_DEPTH_BG = 40.0
TH_DEPTH_OBJ = 25.0


def _value_noise(rng, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.random((gh, gw))
    ys = np.arange(h) / cell; xs = np.arange(w) / cell
    y0 = ys.astype(int); x0 = xs.astype(int)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_gray(seed=20260924, w=KITTI_W, h=KITTI_H, n_rect=400):
    rng = np.random.default_rng(seed)
    img = 0.5 * _value_noise(rng, h, w, 64) + 0.3 * _value_noise(rng, h, w, 16) + 0.2 * _value_noise(rng, h, w, 4)
    img = 40 + 150 * img
    for _ in range(n_rect):
        x0 = rng.integers(0, w - 8); y0 = rng.integers(0, h - 8)
        ww = rng.integers(6, 60); hh = rng.integers(6, 40)
        img[y0:y0 + hh, x0:x0 + ww] += rng.choice([-1.0, 1.0]) * rng.uniform(25, 90)
    img += rng.normal(0, 2.0, img.shape)
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))


def make_frame(seed=1, w=KITTI_W, h=KITTI_H, n_objects=5):
    """Returns dict(gray u8[h,w], depth_raw f32[h,w] (disparity*256), flow f32[h,w,2], mask i32[h,w])."""
    rng = np.random.default_rng(seed)
    gray = make_gray(seed + 20260924, w, h)
    fx, fy, cx, cy = KITTI_K
    # depth: ground plane + far wall
    vv, uu = np.mgrid[0:h, 0:w].astype(np.float64)
    z = np.full((h, w), 60.0)
    below = vv > cy + 5
    z[below] = np.minimum(60.0, 1.65 * fy / (vv[below] - cy))
    mask = np.zeros((h, w), np.int32)
    for k in range(n_objects):
        bw = int(rng.integers(60, 200)); bh = int(rng.integers(40, 120))
        x0 = int(rng.integers(0, w - bw)); y0 = int(rng.integers(h // 3, h - bh))
        zz = rng.uniform(6, 22)
        mask[y0:y0 + bh, x0:x0 + bw] = k + 1
        z[y0:y0 + bh, x0:x0 + bw] = zz
    z = np.clip(z, 4, 80)
    disp = np.rint(256.0 * BF / z)
    invalid = rng.random((h, w)) < 0.02
    disp[invalid] = 0
    depth_raw = disp.astype(np.float32)
    # flow: forward camera motion 0.8 m + per-object offsets + noise; exact zeros on 1 % of the pixels
    tz = 0.8
    zn = np.maximum(z - tz, 0.5)
    un = (uu - cx) * z / zn + cx; vn = (vv - cy) * z / zn + cy
    flow = np.stack([un - uu, vn - vv], -1)
    for k in range(n_objects):
        m = mask == k + 1
        flow[m] += rng.normal(0, 3.0, 2)
    flow += rng.normal(0, 0.3, flow.shape)
    zero = rng.random((h, w)) < 0.01
    flow[zero] = 0
    return dict(gray=gray, depth_raw=np.ascontiguousarray(depth_raw), flow=np.ascontiguousarray(flow.astype(np.float32)),
                mask=np.ascontiguousarray(mask))
