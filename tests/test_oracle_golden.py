"""CPU: the oracle restatements against the golden vectors produced by the reference's own numpy
code (tests/golden/make_golden.py) and against the reference's two known-answer target cases."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, coco_anchors
from oracle import decode as odec
from oracle import targets as otgt


def _g(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.mark.parametrize("name,size", [("targets_np_608_1.npz", 608), ("targets_np_416_2.npz", 416),
                                       ("targets_np_608_3.npz", 608)])
def test_numpy_builder_restatement_bit_exact(name, size):
    g = _g(name)
    grids = [(size // s, size // s) for s in (32, 16, 8)]
    yt = otgt.preprocess_true_boxes(g["boxes"], (size, size), coco_anchors(), 80, False, grid_shapes=grids)
    for l in range(3):
        assert np.array_equal(yt[l], g[f"y{l}"]), f"layer {l}"


@pytest.mark.parametrize("tag", ["consistency", "9cell"])
def test_known_answer_cases_both_builders(tag):
    """tests/test_target_consistency.py:29-51 and tests/test_9cell_alignment.py:21-50 of the
    reference: one box with cx == cy -> T1 and T2 agree to 1e-5; xy in [-1, 2)."""
    g = _g(f"targets_kat_{tag}.npz")
    anchors = [g["a0"], g["a1"], g["a2"]]
    grids = [(19, 19), (38, 38), (76, 76)]
    y_np = otgt.preprocess_true_boxes(g["boxes"], (608, 608), anchors, 1, False, grid_shapes=grids)
    y_tf = otgt.tf_preprocess_true_boxes(g["boxes"], (608, 608), anchors, 1, False, grid_shapes=grids)
    npos = 0
    for l in range(3):
        assert np.array_equal(y_np[l], g[f"y{l}"])
        if tag == "consistency":        # integer centre -> floor(//2) == /2, builders agree fully
            np.testing.assert_allclose(y_tf[l], y_np[l], atol=1e-5)
        else:                           # 311.999: T2 floors the centre, T1 does not; same cells
            assert np.array_equal(y_tf[l][..., 4:], y_np[l][..., 4:])
            np.testing.assert_allclose(y_tf[l][..., 2:4], y_np[l][..., 2:4], atol=1e-5)
        m = y_tf[l][0, ..., 4] > 0.5
        npos += int(m.sum())
        xy = y_tf[l][0][m][:, 0:2]
        assert np.all(xy >= -1.0) and np.all(xy < 2.0)
    assert npos == 9


def test_tf_builder_matches_numpy_builder_modulo_swap():
    """Isolated, integer-centred boxes: T1 == T2 after undoing T1's x/y fraction swap
    (generators.py:3337-3338).  This is what ties the T1 restatement to executable reference code."""
    rng = np.random.default_rng(7)
    anchors = coco_anchors()
    for _ in range(40):
        w, h = rng.integers(6, 300, 2) * 2
        cx, cy = rng.integers(160, 440, 2)
        tb = np.array([[[cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, rng.integers(0, 80)]]], np.float32)
        a = otgt.tf_preprocess_true_boxes(tb, (608, 608), anchors, 80)
        b = otgt.preprocess_true_boxes(tb, (608, 608), anchors, 80, False,
                                       grid_shapes=[(19, 19), (38, 38), (76, 76)])
        for l in range(3):
            assert np.array_equal(a[l][..., 4:], b[l][..., 4:])
            np.testing.assert_allclose(a[l][..., 2:4], b[l][..., 2:4], atol=1e-6)
            m = b[l][0, ..., 4] > 0.5
            if m.any():
                # T2 cell value = (-ki + fx, -kj + fy) with ki the column offset; T1 writes
                # (-kj' + fy, -ki' + fx) with kj' the column offset => fractions swapped.
                ys, xs = np.nonzero(m)
                fx_b = b[l][0, ys, xs, 0] - np.floor(b[l][0, ys, xs, 0])
                fy_b = b[l][0, ys, xs, 1] - np.floor(b[l][0, ys, xs, 1])
                fx_a = a[l][0, ys, xs, 0] - np.floor(a[l][0, ys, xs, 0])
                fy_a = a[l][0, ys, xs, 1] - np.floor(a[l][0, ys, xs, 1])
                np.testing.assert_allclose(fx_a, fy_b, atol=1e-5)
                np.testing.assert_allclose(fy_a, fx_b, atol=1e-5)


def test_bestfit_restatement():
    g = _g("bestfit.npz")
    anchors = coco_anchors()
    for wh, (l, k) in zip(g["wh"], g["layer_anchor"]):
        ll, kk, _ = otgt.best_fit_and_layer(wh, anchors)
        assert (ll, kk) == (l, k)


@pytest.mark.parametrize("method,key", [("iou", "standard"), ("diou", "diou"), ("iou", "cluster")])
@pytest.mark.parametrize("thr", [0.3, 0.45, 0.5])
def test_greedy_nms_restatement(method, key, thr):
    g = _g("nms.npz")
    keep = odec.greedy_nms(g["boxes"], g["scores"], thr, method)
    tag = f"{key}_{int(thr * 100)}"
    assert np.array_equal(g["boxes"][keep], g[f"{tag}_boxes"])
    assert np.array_equal(g["scores"][keep], g[f"{tag}_scores"])
    assert np.array_equal(g["classes"][keep], g[f"{tag}_classes"])


def test_soft_nms_restatement():
    g = _g("nms.npz")
    mask, soft = odec.soft_nms(g["boxes"], g["scores"])
    assert np.array_equal(g["boxes"][mask], g["soft_45_boxes"])
    np.testing.assert_allclose(soft[mask], g["soft_45_scores"], rtol=1e-6)


def regen_heads(g):
    rng = np.random.default_rng(int(g["seed"]))
    size = int(g["size"])
    heads = [(2.0 * rng.standard_normal((1, s, s, 88))).astype(np.float32)
             for s in (size // 32, size // 16, size // 8)]
    np.testing.assert_allclose([h.astype(np.float64).sum() for h in heads], g["head_sums"], rtol=0, atol=1e-9)
    return heads, size


@pytest.mark.parametrize("name", ["decode_416_20.npz", "decode_608_21.npz", "decode_608_22.npz"])
def test_decode_restatement(name):
    g = _g(name)
    heads, size = regen_heads(g)
    step = int(g["row_step"])
    dec = odec.decode_predictions(heads, coco_anchors(), 80, (size, size))
    np.testing.assert_allclose(dec[:, ::step], g["decoded"], rtol=2e-5, atol=1e-6)
    cor = odec.correct_boxes(dec, tuple(g["image_shape"]), (size, size))
    np.testing.assert_allclose(cor[:, ::step], g["corrected"], rtol=2e-5, atol=1e-4)
    for method, thr, conf in (("diou", 0.45, 0.1), ("diou", 0.5, 0.3), ("cluster", 0.45, 0.1), ("soft", 0.45, 0.3)):
        b, c, s = odec.postprocess(heads, coco_anchors(), 80, (size, size), tuple(g["image_shape"]), (size, size),
                                   max_boxes=100, confidence=conf, nms_threshold=thr, nms_method=method)
        tag = f"{method}_{int(thr * 100)}_{int(conf * 100)}"
        assert np.array_equal(np.asarray(b), g[f"{tag}_boxes"]), tag
        assert np.array_equal(np.asarray(c), g[f"{tag}_classes"]), tag
        np.testing.assert_allclose(np.asarray(s), g[f"{tag}_scores"], rtol=1e-5)
