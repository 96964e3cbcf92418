"""CPU oracle for the batch augmentations (Mosaic, MixUp, GridMask).  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
PARITY UNPINNED: the reference implements these as TensorFlow graph code that cannot run here and its
tests hold no numeric expectations; this is a numpy restatement, taking the random draws
as arguments (the product makes the same draws on the host, multigriddet_amd/data/augment.py):
  mosaic    multigriddet/data/generators.py:794-998  (quadrants TL=img0, BL=img1, BR=img2, TR=img3 pasted
            without shift; boxes clipped to their quadrant, kept if they overlap it and w,h >= max(10, 0.03 S);
            concatenated in quadrant order, zero padded; overflow -> error :954-976)
  mixup     :1012-1161  (lambda*img1 + (1-lambda)*img2; valid boxes of both, in order)
  gridmask  :1164-1282  (horizontal stripes of a hh x hh mask, hh = ceil(sqrt(H^2+W^2)), cropped at the
            centre; image * (1 - mask): only the stripes survive; a box is kept if the un-inverted mask
            covers > 0.3 of its integer-truncated area)
"""
import numpy as np


def mosaic(images, boxes, src, crop, min_box_ratio=0.03, M_out=None):
    B, S = images.shape[0], images.shape[1]
    M_out = M_out or boxes.shape[1]
    out_i = np.empty_like(images)
    out_b = np.zeros((B, M_out, 5), np.float32)
    min_size = np.float32(max(10.0, S * min_box_ratio))
    for b in range(B):
        cx, cy = int(crop[b, 0]), int(crop[b, 1])
        i0, i1, i2, i3 = (images[src[b, q]] for q in range(4))
        left = np.concatenate([i0[:cy, :cx], i1[cy:, :cx]], 0)
        right = np.concatenate([i3[:cy, cx:], i2[cy:, cx:]], 0)
        out_i[b] = np.concatenate([left, right], 1)
        kept = []
        cxf, cyf, W, H = np.float32(cx), np.float32(cy), np.float32(S), np.float32(S)
        for q in range(4):
            lx, hx = (np.float32(0), cxf) if q in (0, 1) else (cxf, W)
            ly, hy = (np.float32(0), cyf) if q in (0, 3) else (cyf, H)
            for x1, y1, x2, y2, c in boxes[src[b, q]]:
                if not ((x2 - x1) * (y2 - y1) > 0):
                    continue
                if not (y2 > ly and y1 < hy and x2 > lx and x1 < hx):
                    continue
                nx1, ny1, nx2, ny2 = max(x1, lx), max(y1, ly), min(x2, hx), min(y2, hy)
                if (nx2 - nx1) >= min_size and (ny2 - ny1) >= min_size:
                    kept.append([nx1, ny1, nx2, ny2, c])
        if len(kept) > M_out:
            raise RuntimeError("Mosaic augmentation (batch): Box capacity overflow.")
        if kept:
            out_b[b, :len(kept)] = np.array(kept, np.float32)
    return out_i, out_b


def mixup(images, boxes, partner, lam, M_out=None):
    B = images.shape[0]
    M_out = M_out or boxes.shape[1]
    out_i = np.empty_like(images)
    out_b = np.zeros((B, M_out, 5), np.float32)
    for b in range(B):
        la = np.float32(lam[b])
        out_i[b] = la * images[b] + (np.float32(1.0) - la) * images[partner[b]]
        kept = [bx for k in (b, partner[b]) for bx in boxes[k] if (bx[2] - bx[0]) * (bx[3] - bx[1]) > 0]
        out_b[b, :min(len(kept), M_out)] = np.array(kept[:M_out], np.float32).reshape(-1, 5)
    return out_i, out_b


def gridmask(images, boxes, apply, params, keep_frac=0.3):
    B, S = images.shape[0], images.shape[1]
    out_i, out_b = images.copy(), boxes.copy()
    hh = int(np.ceil(np.sqrt(np.float32(2.0) * np.float32(S) * np.float32(S))))
    off = (hh - S) // 2
    for b in range(B):
        if not apply[b]:
            continue
        d, l, st = (int(v) for v in params[b])
        mask = np.ones(hh, np.float32)
        for i in range(-1, hh // d + 2):
            s, t = np.clip(d * i + st, 0, hh), np.clip(d * i + st + l, 0, hh)
            mask[s:t] = 0.0
        m = mask[off:off + S]                                  # rows of the cropped mask
        out_i[b] = np.clip(images[b] * (1.0 - m)[:, None, None], 0.0, 255.0)
        kept = []
        for bx in boxes[b]:
            x1, y1, x2, y2 = (int(v) for v in bx[:4])
            area = float((x2 - x1) * (y2 - y1))
            ya, yb, xa, xb = max(y1, 0), min(y2, S), max(x1, 0), min(x2, S)
            valid = float(m[ya:yb].sum()) * max(xb - xa, 0) if yb > ya else 0.0
            if valid > area * keep_frac:
                kept.append(bx)
        out_b[b] = 0
        if kept:
            out_b[b, :len(kept)] = np.array(kept, np.float32)
    return out_i, out_b
