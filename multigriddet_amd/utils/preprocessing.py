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
