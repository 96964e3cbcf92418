"""Inference-side letterbox (reference multigriddet/utils/preprocessing.py:12-90): bicubic resize,
centred paste on a (128,128,128) canvas, /255, batch dim."""
import numpy as np


def letterbox_resize(image, target_size, return_padding_info=False):
    from PIL import Image
    tw, th = target_size
    w, h = image.size
    scale = min(tw / w, th / h)
    nw, nh = int(w * scale), int(h * scale)
    dx, dy = (tw - nw) // 2, (th - nh) // 2
    canvas = Image.new("RGB", (tw, th), (128, 128, 128))
    canvas.paste(image.resize((nw, nh), Image.BICUBIC), (dx, dy))
    if return_padding_info:
        return canvas, (nw, nh), (dx, dy)
    return canvas


def preprocess_image(image, model_image_size):
    """model_image_size = (h, w).  Returns float32 [1, h, w, 3] in [0,1]."""
    boxed = letterbox_resize(image, tuple(reversed(model_image_size)))
    return np.expand_dims(np.asarray(boxed, np.float32) / 255.0, 0)


# ---------------------------------------------------------------------------------------------------------------
# Device letterbox (csrc/preprocess.hip, mgd_letterbox_u8): the host only builds PIL's resampling tables.
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0,
                    np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))


def resample_tables(in_size, out_size):
    """PIL's bicubic coefficient tables for resizing an axis of `in_size` pixels to `out_size`
    (Pillow ImagingResample: precompute_coeffs + normalize_coeffs_8bpc, filter support 2, a = -0.5).
    Returns (k int32 [out][ksize], bounds int32 [out][2] = (first source index, count), ksize)."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(np.ceil(support)) * 2 + 1
    k = np.zeros((out_size, ksize), np.float64)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = _bicubic((np.arange(xmax) + xmin - center + 0.5) * ss)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        k[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    kk = np.where(k < 0, -0.5 + k * (1 << _PRECISION_BITS), 0.5 + k * (1 << _PRECISION_BITS))
    return np.trunc(kk).astype(np.int32), bounds, ksize


def letterbox_geometry(src_hw, model_hw):
    """(nh, nw, dy, dx) of letterbox_resize (reference utils/preprocessing.py:33-41)."""
    h, w = src_hw
    th, tw = model_hw
    scale = min(tw / w, th / h)
    nw, nh = int(w * scale), int(h * scale)
    return nh, nw, (th - nh) // 2, (tw - nw) // 2


class DeviceLetterbox:
    """uint8 frames -> the model's fp32 NHWC input on the GPU (bit-identical to preprocess_image); coefficient tables
    are cached per (frame size, model size) - a video stream builds them once."""

    def __init__(self, model_hw, device="cuda:0"):
        import torch
        self.model_hw = (int(model_hw[0]), int(model_hw[1]))
        self.device = torch.device(device)
        self._tables = {}
        self._ws = None

    def _get(self, h, w):
        import torch
        key = (h, w)
        t = self._tables.get(key)
        if t is None:
            nh, nw, dy, dx = letterbox_geometry((h, w), self.model_hw)
            if nh < 1 or nw < 1:
                raise ValueError(f"image {h}x{w} collapses to {nh}x{nw} in a {self.model_hw} letterbox")
            kx, bx, ksx = resample_tables(w, nw)
            ky, by, ksy = resample_tables(h, nh)
            dev = [torch.from_numpy(np.ascontiguousarray(a)).to(self.device) for a in (kx, bx, ky, by)]
            t = self._tables[key] = (nh, nw, dy, dx, ksx, ksy, dev)
        return t

    def __call__(self, frames, out=None):
        """frames: list of uint8 HWC arrays / tensors (any sizes).  Returns fp32 [B, Hm, Wm, 3] on the device."""
        import ctypes as C
        import torch
        from .. import _lib as L
        L.require_gpu()
        lib = L.load()
        hm, wm = self.model_hw
        B = len(frames)
        if out is None:
            out = torch.empty(B, hm, wm, 3, dtype=torch.float32, device=self.device)
        for i, f in enumerate(frames):
            fr = f if isinstance(f, torch.Tensor) else torch.from_numpy(np.array(f, dtype=np.uint8, order="C"))
            fr = fr.to(self.device, torch.uint8).contiguous()
            h, w = int(fr.shape[0]), int(fr.shape[1])
            nh, nw, dy, dx, ksx, ksy, (kx, bx, ky, by) = self._get(h, w)
            need = lib.mgd_letterbox_workspace_size(h, nw)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
            L.check(lib.mgd_letterbox_u8(L.ptr(fr), h, w, L.ptr(out[i]), hm, wm, nh, nw, dy, dx, L.ptr(kx), L.ptr(bx), ksx,
                                         L.ptr(ky), L.ptr(by), ksy, C.c_float(128.0), L.ptr(self._ws),
                                         C.c_size_t(self._ws.numel()), L.stream_ptr()), "letterbox_u8")
        return out
