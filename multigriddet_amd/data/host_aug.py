"""Per-image host augmentations of the reference's tf.data pipeline, restated in numpy / PIL.

The reference applies these per image, on [0,255] float images, before batching (multigriddet/data/generators.py:
1918-1949): tf_random_resize_crop_pad (:347-462, aspect jitter 0.3, scale jitter 0.5, 128-grey canvas) -> horizontal flip
p=0.5 (:227-257) -> brightness 0.2 / contrast [0.8,1.2] / saturation [0.8,1.2] / hue 0.1 / grayscale p=0.1
(:260-344, each computed on [0,1] and clipped) -> rot90 with p=0.05 (:465-558; the drawn Gaussian angle is unused by the
reference, only k*90 degrees is applied).  GridMask (p=0.1) runs on the device (csrc/aug.hip).

These are host I/O stages outside the accelerated path: same operations, same parameter ranges and box
arithmetic; the random stream is numpy's, not TensorFlow's, and PIL's bicubic filter stands in for tf.image.resize, so the
pixels are "parity unpinned" (no TensorFlow here to generate fixtures from).  Every function takes and returns
(image float32 [H,W,3] in [0,255], boxes float32 [n,5] = x1,y1,x2,y2,cls).
"""
import numpy as np


def random_resize_crop_pad(rng, image, boxes, target_hw, aspect_ratio_jitter=0.3, scale_jitter=0.5):
    from PIL import Image
    th, tw = target_hw
    sh, sw = image.shape[:2]
    ar = (tw / th) * (rng.uniform(1 - aspect_ratio_jitter, 1 + aspect_ratio_jitter) /
                      rng.uniform(1 - aspect_ratio_jitter, 1 + aspect_ratio_jitter))
    scale = rng.uniform(scale_jitter, 1.0 / scale_jitter)
    if ar < 1.0:
        ph = int(scale * th)
        pw = int(ph * ar)
    else:
        pw = int(scale * tw)
        ph = int(pw / ar)
    pw, ph = max(pw, 1), max(ph, 1)
    im = Image.fromarray(np.clip(image, 0, 255).astype(np.uint8)).resize((pw, ph), Image.BICUBIC)
    res = np.asarray(im, np.float32)
    dx = int(rng.integers(0, max(1, tw - pw)))
    dy = int(rng.integers(0, max(1, th - ph)))
    out = np.full((th, tw, 3), 128.0, np.float32)
    ch, cw = min(ph, th - dy), min(pw, tw - dx)
    out[dy:dy + ch, dx:dx + cw] = res[:ch, :cw]
    b = boxes.copy()
    if len(b):
        b[:, [0, 2]] = b[:, [0, 2]] * (pw / sw) + dx
        b[:, [1, 3]] = b[:, [1, 3]] * (ph / sh) + dy
        b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, tw)
        b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, th)
    return out, b


def random_horizontal_flip(rng, image, boxes):
    if not rng.uniform() > 0.5:
        return image, boxes
    w = image.shape[1]
    b = boxes.copy()
    if len(b):
        b[:, 0], b[:, 2] = w - boxes[:, 2], w - boxes[:, 0]
    return image[:, ::-1].copy(), b


def _rgb_to_hsv(x):
    mx, mn = x.max(-1), x.min(-1)
    d = mx - mn
    s = np.where(mx > 0, d / np.maximum(mx, 1e-12), 0.0)
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    dd = np.maximum(d, 1e-12)
    h = np.where(mx == r, (g - b) / dd, np.where(mx == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd))
    h = np.where(d > 0, (h / 6.0) % 1.0, 0.0)
    return h, s, mx


def _hsv_to_rgb(h, s, v):
    i = np.floor(h * 6.0)
    f = h * 6.0 - i
    p, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    i = i.astype(np.int32) % 6
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.stack([r, g, b], -1)


def color_jitter(rng, image, brightness=0.2, contrast=(0.8, 1.2), saturation=(0.8, 1.2), hue=0.1, gray_prob=0.1):
    """tf.image.random_brightness / contrast / saturation / hue / rgb_to_grayscale on the image / 255, each clipped."""
    x = image.astype(np.float32) / 255.0
    x = np.clip(x + rng.uniform(-brightness, brightness), 0, 1)
    m = x.mean((0, 1), keepdims=True)
    x = np.clip((x - m) * rng.uniform(*contrast) + m, 0, 1)
    h, s, v = _rgb_to_hsv(x)
    x = np.clip(_hsv_to_rgb(h, np.clip(s * rng.uniform(*saturation), 0, 1), v), 0, 1)
    h, s, v = _rgb_to_hsv(x)
    x = np.clip(_hsv_to_rgb((h + rng.uniform(-hue, hue)) % 1.0, s, v), 0, 1)
    if rng.uniform() < gray_prob:
        g = x @ np.array([0.2989, 0.5870, 0.1140], np.float32)
        x = np.repeat(g[..., None], 3, -1)
    return (x * 255.0).astype(np.float32)


def random_rot90(rng, image, boxes, prob=0.05):
    """k in {1,2,3} counter-clockwise quarter turns with probability `prob` (square inputs keep their shape)."""
    if not rng.uniform() < prob:
        return image, boxes
    k = int(rng.integers(1, 4))
    h, w = image.shape[:2]
    out = np.rot90(image, k).copy()
    b = boxes.copy()
    if len(b):
        x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
        if k == 1:
            b[:, 0], b[:, 1], b[:, 2], b[:, 3] = y1, w - x2, y2, w - x1
        elif k == 2:
            b[:, 0], b[:, 1], b[:, 2], b[:, 3] = w - x2, h - y2, w - x1, h - y1
        else:
            b[:, 0], b[:, 1], b[:, 2], b[:, 3] = h - y2, x1, h - y1, x2
        nh, nw = out.shape[:2]
        b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, nw)
        b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, nh)
    return out, b


def drop_degenerate(boxes, min_size=1.0):
    if not len(boxes):
        return boxes
    keep = ((boxes[:, 2] - boxes[:, 0]) >= min_size) & ((boxes[:, 3] - boxes[:, 1]) >= min_size)
    return boxes[keep]


def augment_image(rng, image, boxes, target_hw):
    """The reference's per-image chain in its order (generators.py:1918-1943), GridMask excluded (device)."""
    image, boxes = random_resize_crop_pad(rng, image, boxes, target_hw)
    image, boxes = random_horizontal_flip(rng, image, boxes)
    image = color_jitter(rng, image)
    if target_hw[0] == target_hw[1]:
        image, boxes = random_rot90(rng, image, boxes)
    return image, drop_degenerate(boxes)
