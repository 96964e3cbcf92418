#!/usr/bin/env python3
"""Golden vectors for G4 (inference-side letterbox + normalise): outputs of the reference's own
`letterbox_resize` / `preprocess_image` (multigriddet/utils/preprocessing.py:12-90, pure PIL/numpy, loaded by file
path) on seeded synthetic images.  Fixture tooling only: run here, where /root/reference is mounted; the .npz is
committed, the reference never travels."""
import importlib.util
import os

import numpy as np
from PIL import Image

REF = "/root/reference/multigriddet/utils/preprocessing.py"
OUT = os.path.dirname(os.path.abspath(__file__))

spec = importlib.util.spec_from_file_location("refmg_preprocessing", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

rng = np.random.default_rng(123)
cases = [((37, 53), (64, 64)), ((80, 60), (64, 96)), ((100, 30), (96, 96)), ((64, 64), (64, 64)), ((120, 160), (160, 160))]
out = {}
for i, ((h, w), (mh, mw)) in enumerate(cases):
    # smooth + noise content so that the bicubic filter matters
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / 7.0 + i), 127 + 100 * np.cos(yy / 5.0), (xx + yy) * 255.0 / (h + w)], -1)
    img = np.clip(base + rng.normal(0, 20, base.shape), 0, 255).astype(np.uint8)
    pil = Image.fromarray(img)
    boxed, size, off = mod.letterbox_resize(pil, (mw, mh), return_padding_info=True)
    data = mod.preprocess_image(pil, (mh, mw))
    out[f"img{i}"] = img
    out[f"model_hw{i}"] = np.array([mh, mw], np.int32)
    out[f"boxed{i}"] = np.asarray(boxed, np.uint8)
    out[f"pad{i}"] = np.array([size[0], size[1], off[0], off[1]], np.int32)
    # preprocess_image is exactly boxed / 255 in float32 with a batch dimension: assert it here instead of storing it
    assert data.dtype == np.float32 and data.shape == (1, mh, mw, 3)
    assert np.array_equal(data[0], np.asarray(boxed, np.float32) / np.float32(255.0))
out["n"] = np.int32(len(cases))
np.savez_compressed(os.path.join(OUT, "letterbox.npz"), **out)
print("wrote letterbox.npz", {k: v.shape for k, v in out.items() if k.startswith("boxed")})
