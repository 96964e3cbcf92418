"""On-device batch augmentation: host draws + mgd_mosaic / mgd_mixup / mgd_gridmask.

Mirrors tf_random_mosaic / tf_random_mixup / tf_random_gridmask (reference
multigriddet/data/generators.py:561-1009, 1012-1161, 1164-1282).  The reference draws inside the TF
graph; here the draws come from a numpy Generator on the host so that the CPU oracle (oracle/aug.py)
can replay them - the distributions are the reference's: per output slot 4 source indices with
replacement, crop_x/crop_y ~ U{[0.2 S, 0.8 S)}; MixUp lambda ~ U(0,1) clipped to [0.2, 0.8], partner
!= self; GridMask d ~ U{[S/7, S/3)}, l = d/2, start ~ U{[0, d)}.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib as L


def draw_mosaic(rng, B, S, min_offset=0.2):
    src = rng.integers(0, B, size=(B, 4)).astype(np.int32)
    lo, hi = int(S * min_offset), int(S * (1 - min_offset))
    crop = np.stack([rng.integers(lo, hi, size=B), rng.integers(lo, hi, size=B)], 1).astype(np.int32)
    return src, crop


def mosaic(images, boxes, src, crop, min_box_ratio=0.03, M_out=None):
    """images fp32 CUDA [B,S,S,3]; boxes fp32 CUDA [B,M,5].  Raises RuntimeError on capacity overflow
    (the reference asserts, generators.py:954-976)."""
    B, S = images.shape[0], images.shape[1]
    M_in = boxes.shape[1]
    M_out = M_out or M_in
    dev = images.device
    out_i = torch.empty_like(images)
    out_b = torch.empty(B, M_out, 5, dtype=torch.float32, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    srcd, cropd = torch.from_numpy(src).to(dev), torch.from_numpy(crop).to(dev)
    min_wh = max(10.0, S * min_box_ratio)
    L.check(L.load().mgd_mosaic(L.ptr(images), L.ptr(boxes), B, S, M_in, L.ptr(srcd), L.ptr(cropd), C.c_float(min_wh),
                                L.ptr(out_i), L.ptr(out_b), M_out, L.ptr(ovf), L.stream_ptr()), "mosaic")
    if int(ovf.item()):
        raise RuntimeError("Mosaic augmentation (batch): Box capacity overflow. Merged boxes exceed configured capacity.")
    return out_i, out_b


def draw_mixup(rng, B):
    lam = float(np.clip(rng.uniform(0.0, 1.0), 0.2, 0.8))
    partner = rng.integers(0, B, size=B)
    partner = np.where(partner == np.arange(B), (partner + 1) % B, partner).astype(np.int32)
    return partner, np.full(B, lam, np.float32)


def mixup(images, boxes, partner, lam, M_out=None):
    B, S = images.shape[0], images.shape[1]
    M_in = boxes.shape[1]
    M_out = M_out or M_in
    dev = images.device
    out_i = torch.empty_like(images)
    out_b = torch.empty(B, M_out, 5, dtype=torch.float32, device=dev)
    pd, ld = torch.from_numpy(partner).to(dev), torch.from_numpy(lam).to(dev)   # keep alive across the call
    L.check(L.load().mgd_mixup(L.ptr(images), L.ptr(boxes), B, S, M_in, L.ptr(pd), L.ptr(ld), L.ptr(out_i),
                               L.ptr(out_b), M_out, L.stream_ptr()), "mixup")
    return out_i, out_b


def draw_gridmask(rng, B, S, prob=0.1, d1_ratio=1.0 / 7.0, d2_ratio=1.0 / 3.0, grid_ratio=0.5):
    apply = (rng.uniform(size=B) < prob).astype(np.int32)
    d = rng.integers(int(S * d1_ratio), int(S * d2_ratio), size=B)
    l = (d.astype(np.float32) * grid_ratio).astype(np.int64)
    st = np.array([rng.integers(0, int(x)) for x in d])
    return apply, np.stack([d, l, st], 1).astype(np.int32)


def gridmask(images, boxes, apply, params, keep_frac=0.3):
    """In place on images in the [0,255] range and on the box list (kept boxes compacted to the front)."""
    B, S = images.shape[0], images.shape[1]
    dev = images.device
    ad, pd = torch.from_numpy(apply).to(dev), torch.from_numpy(params).to(dev)   # keep alive across the call
    L.check(L.load().mgd_gridmask(L.ptr(images), L.ptr(boxes), B, S, boxes.shape[1], L.ptr(ad), L.ptr(pd),
                                  C.c_float(keep_frac), L.stream_ptr()), "gridmask")
    return images, boxes
