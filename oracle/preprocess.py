"""CPU restatement of PIL's two-pass 8-bit resampler driven by the product's coefficient tables.  TEST INFRASTRUCTURE
ONLY (used by tests/ to pin multigriddet_amd.utils.preprocessing.resample_tables against PIL itself and against the
reference's letterbox outputs in tests/golden/letterbox.npz; csrc/preprocess.hip repeats these two loops on the GPU).
Follows Pillow's ImagingResample (horizontal pass to uint8, then vertical), which is what the reference's
letterbox_resize runs through Image.resize(BICUBIC) (multigriddet/utils/preprocessing.py:44)."""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _pass(img, k, bounds, axis):
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((k.shape[0],) + img.shape[1:], np.uint8)
    for i in range(k.shape[0]):
        lo, n = int(bounds[i, 0]), int(bounds[i, 1])
        s = (1 << (PRECISION_BITS - 1)) + np.tensordot(k[i, :n].astype(np.int64), img[lo:lo + n], axes=(0, 0))
        out[i] = np.clip(s >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resize_u8(img, out_hw, tables):
    """img uint8 [H,W,3] -> uint8 [nh,nw,3]; tables(in, out) -> (k, bounds, ksize)."""
    nh, nw = out_hw
    kx, bx, _ = tables(img.shape[1], nw)
    ky, by, _ = tables(img.shape[0], nh)
    return _pass(_pass(img, kx, bx, 1), ky, by, 0)


def letterbox(img, model_hw, tables, geometry, fill=128):
    nh, nw, dy, dx = geometry(img.shape[:2], model_hw)
    canvas = np.full((model_hw[0], model_hw[1], 3), fill, np.uint8)
    canvas[dy:dy + nh, dx:dx + nw] = resize_u8(img, (nh, nw), tables)
    return canvas
