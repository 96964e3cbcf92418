"""CPU: properties of the augmentation oracle (no reference fixture exists for these TF-only functions)."""
import numpy as np

from oracle import aug as oa


def test_mosaic_quadrant_layout_and_box_clipping():
    S = 64
    img = np.stack([np.full((S, S, 3), v, np.float32) for v in (10, 20, 30, 40)])
    boxes = np.zeros((4, 8, 5), np.float32)
    boxes[0, 0] = [5, 5, 60, 60, 1]
    src = np.array([[0, 1, 2, 3]] * 4, np.int32)
    crop = np.array([[40, 24]] * 4, np.int32)
    oi, ob = oa.mosaic(img, boxes, src, crop)
    assert oi[0, 0, 0, 0] == 10 and oi[0, 63, 0, 0] == 20 and oi[0, 63, 63, 0] == 30 and oi[0, 0, 63, 0] == 40
    # only image 0's box, clipped to the top-left quadrant (x < 40, y < 24)
    assert np.array_equal(ob[0, 0], [5, 5, 40, 24, 1]) and ob[0, 1:].sum() == 0


def test_gridmask_keeps_only_stripes():
    S = 70
    img = np.full((1, S, S, 3), 100.0, np.float32)
    boxes = np.zeros((1, 4, 5), np.float32)
    oi, _ = oa.gridmask(img, boxes, np.array([1]), np.array([[14, 7, 3]], np.int32))
    rows = oi[0, :, 0, 0]
    assert set(np.unique(rows)) == {0.0, 100.0}
    assert abs((rows > 0).mean() - 0.5) < 0.15          # l = d/2 -> about half of the rows survive
