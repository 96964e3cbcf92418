"""Anchor / class file parsing (reference multigriddet/utils/anchors.py:282-405)."""
from typing import List

import numpy as np


def load_anchors(anchors_path: str) -> List[np.ndarray]:
    """One line per scale, 'w,h, w,h, w,h' (trailing commas tolerated); first line = stride 32."""
    anchors = []
    with open(anchors_path) as f:
        for line in f:
            pairs = []
            for tok in line.strip().split():
                tok = tok.rstrip(",")
                if "," not in tok:
                    continue
                try:
                    w, h = map(float, tok.split(","))
                except ValueError:
                    continue
                pairs.append([w, h])
            if pairs:
                anchors.append(np.array(pairs))
    return anchors


def load_classes(classes_path: str) -> List[str]:
    with open(classes_path) as f:
        return [l.strip() for l in f if l.strip()]


def compute_class_weights(annotation_file: str, num_classes: int, method: str = "balanced") -> np.ndarray:
    """Class weights from annotation-line class frequencies (reference anchors.py:333-405):
    'balanced' = N / (C * (n_c + 1e-8)), 'inverse' = 1/(f_c + 1e-8), 'sqrt_inverse' = 1/(sqrt(f_c) + 1e-8),
    each divided by its mean (+1e-8) and clipped to [0.1, 10]; no boxes -> ones."""
    counts = np.zeros(num_classes, dtype=np.float32)
    total = 0.0
    with open(annotation_file) as f:
        for line in f:
            parts = line.strip().split()
            for tok in parts[1:]:
                try:
                    c = tok.split(",")
                    if len(c) >= 5:
                        cid = int(float(c[4]))
                        if 0 <= cid < num_classes:
                            counts[cid] += 1.0
                            total += 1.0
                except (ValueError, IndexError):
                    continue
    if total == 0:
        return np.ones(num_classes, dtype=np.float32)
    freq = counts / (total + 1e-8)
    if method == "balanced":
        w = total / (num_classes * (counts + 1e-8))
    elif method == "inverse":
        w = 1.0 / (freq + 1e-8)
    elif method == "sqrt_inverse":
        w = 1.0 / (np.sqrt(freq) + 1e-8)
    else:
        return np.ones(num_classes, dtype=np.float32)
    w = w / (np.mean(w) + 1e-8)
    return np.clip(w, 0.1, 10.0).astype(np.float32)
