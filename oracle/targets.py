"""CPU oracle for the 3x3 multi-grid y_true target builder.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker.  The product path (`multigriddet_amd`) never imports `oracle`.

Two builders exist in the reference and they disagree; both are restated:

* `tf_preprocess_true_boxes`  (T1)  follows  multigriddet/data/generators.py:2696-3390
  -- the default trainer path.  TensorFlow cannot run in this container, so this restatement is
  pinned by (a) the reference's two known-answer cases, whose inputs have cx == cy and on which
  T1 and T2 must agree to 1e-5 (tests/test_target_consistency.py:29-51,
  tests/test_9cell_alignment.py:21-50; fixtures tests/golden/targets_kat_*.npz made by running
  the reference's *numpy* builder), and (b) cross-checks against T2 on isolated integer-centred
  boxes with the documented x/y fraction swap undone.  Behaviour under colliding boxes
  (TF-CPU `tensor_scatter_nd_update` = last update wins) is "parity unpinned".
* `preprocess_true_boxes`     (T2)  follows  multigriddet/data/generators.py:3393-3473
  (+ best_fit_and_layer :2514-2544, iol_common_center :2486-2494) -- pinned bit-for-bit by
  tests/golden/targets_np_*.npz, produced by the reference's own numpy function.
"""
import numpy as np


def _f32(x):
    return np.float32(x)


def tf_preprocess_true_boxes(true_boxes, input_shape, anchors, num_classes, multi_anchor_assign=False,
                             grid_shapes=None, return_assignment=False):
    """T1 restatement, float32 arithmetic in the reference's operation order.

    generators.py:2730-2731 centre/wh; :2756 validity; :2816-2825 IoL (+1e-7); :2881-2931
    layer = argmax of per-layer max IoL, anchor = argmax within that layer; :2962-2975 grid
    position (trunc); :2993-3024 nine candidates, ki-major; :3245-3277 occupancy read from the
    still-zero tensor => every in-bounds candidate is assigned; :3316-3356 the written row (note
    x <- -kj + ty, y <- -ki + tx: the reference's swapped fractions, reproduced); :3370 scatter,
    duplicates resolved last-writer-wins in (batch, box, candidate) order (TF-CPU behaviour).
    """
    tb = np.asarray(true_boxes, dtype=np.float32)
    B, M = tb.shape[0], tb.shape[1]
    H, W = int(input_shape[0]), int(input_shape[1])
    L = len(anchors)
    anchors = [np.asarray(a, dtype=np.float32) for a in anchors]
    if grid_shapes is None:
        grid_shapes = [(H // s, W // s) for s in (32, 16, 8)][:L]
    nA = [len(a) for a in anchors]
    y_true = [np.zeros((B, int(grid_shapes[l][0]), int(grid_shapes[l][1]), 5 + nA[l] + num_classes), np.float32)
              for l in range(L)]
    all_a = np.concatenate(anchors, 0)
    a_area = all_a[:, 0] * all_a[:, 1]
    starts = np.concatenate([[0], np.cumsum(nA)[:-1]]).astype(int)
    assign = np.full((B, M, 4), -1, np.int32)      # layer, anchor, row, col  (for integer parity)
    eps = _f32(1e-7)
    for b in range(B):
        for t in range(M):
            x1, y1, x2, y2, c = tb[b, t]
            bx = (x1 + x2) / _f32(2.0)
            by = (y1 + y2) / _f32(2.0)
            bw = x2 - x1
            bh = y2 - y1
            if not (bw * bh > 0.0):
                continue
            inter = np.minimum(bw, all_a[:, 0]) * np.minimum(bh, all_a[:, 1])
            iol = inter / (np.maximum(bw * bh, a_area) + eps)
            per_layer = np.array([iol[starts[l]:starts[l] + nA[l]].max() for l in range(L)], np.float32)
            layer = int(np.argmax(per_layer))
            k = int(np.argmax(iol[starts[layer]:starts[layer] + nA[layer]]))
            gh, gw = int(grid_shapes[layer][0]), int(grid_shapes[layer][1])
            cx = bx * (_f32(gw) / _f32(W))
            cy = by * (_f32(gh) / _f32(H))
            col = int(cx)                       # tf.cast(float -> int32) truncates
            row = int(cy)
            tx = cx - _f32(col)
            ty = cy - _f32(row)
            aw, ah = anchors[layer][k]
            tw = np.log(np.maximum(bw / aw, _f32(1e-3)))
            th = np.log(np.maximum(bh / ah, _f32(1e-3)))
            cls = int(c)
            assign[b, t] = (layer, k, row, col)
            for cand in range(9):
                ki, kj = cand // 3 - 1, cand % 3 - 1
                r, q = row + ki, col + kj
                if r < 0 or r >= gh or q < 0 or q >= gw:
                    continue
                cell = np.zeros(5 + nA[layer] + num_classes, np.float32)
                cell[0] = _f32(-kj) + ty
                cell[1] = _f32(-ki) + tx
                cell[2], cell[3], cell[4] = tw, th, 1.0
                cell[5 + k] = 1.0
                if 0 <= cls < num_classes:      # tf.one_hot: out-of-range index -> all zeros
                    cell[5 + nA[layer] + cls] = 1.0
                y_true[layer][b, r, q] = cell
    if return_assignment:
        return y_true, assign
    return y_true


def iol_common_center(anchors, wh):
    """generators.py:2486-2494 (no epsilon)."""
    inter = np.minimum(np.expand_dims(wh, -2), anchors)
    return (inter[..., 0] * inter[..., 1]) / np.maximum(
        np.expand_dims(wh[..., 0] * wh[..., 1], -1), anchors[:, 0] * anchors[:, 1])


def best_fit_and_layer(box_wh, anchors):
    """generators.py:2514-2544 with multi_anchor_assign=False: IoL rounded to 3 decimals, then the
    first element of argsort(-iol) (insertion sort for n<=16 => first maximum wins)."""
    all_a = np.concatenate(anchors, 0)
    iols = np.round(iol_common_center(all_a, box_wh), 3)
    idx = int(np.argsort(-iols, kind="stable")[0])
    acc = 0
    for l, a in enumerate(anchors):
        if idx < acc + len(a):
            return l, idx - acc, iols
        acc += len(a)
    raise AssertionError


def preprocess_true_boxes(true_boxes, input_shape, anchors, num_classes, multi_anchor_assign=False,
                          grid_shapes=None, iou_thresh=0.2):
    """T2 restatement (generators.py:3393-3473), sequential, numpy dtypes as the reference has them
    under numpy 2 (cx/cy become float64 because `grid / np.int32` is a strong float64 scalar)."""
    assert (np.asarray(true_boxes)[..., 4] < num_classes).all(), "class id must be less than num_classes"
    L = len(anchors)
    tb = np.array(true_boxes, dtype="float32")
    ishape = np.array(input_shape, dtype="int32")
    anchors = [np.asarray(a, dtype=np.float32) for a in anchors]
    bxy = (tb[..., 0:2] + tb[..., 2:4]) // 2          # floor division (:3415)
    bwh = tb[..., 2:4] - tb[..., 0:2]
    tb[..., 0:2] = bxy
    tb[..., 2:4] = bwh
    B = tb.shape[0]
    if grid_shapes is None:
        grid_shapes = [ishape // {0: 32, 1: 16, 2: 8, 3: 4, 4: 2}[l] for l in range(L)]
    y_true = [np.zeros((B, grid_shapes[l][0], grid_shapes[l][1], 5 + len(anchors[l]) + num_classes), "float32")
              for l in range(L)]
    for b in range(B):
        for t in range(tb.shape[1]):
            bw, bh = bwh[b, t]
            if bw * bh <= 0.0:
                continue
            l, k, _ = best_fit_and_layer(bwh[b, t], anchors)
            c = int(tb[b, t, 4])
            cx = np.float64(tb[b, t, 0]) * (grid_shapes[l][0] / np.float64(ishape[0]))   # height ratio (:3438)
            cy = np.float64(tb[b, t, 1]) * (grid_shapes[l][1] / np.float64(ishape[1]))
            i, j = int(cx), int(cy)                   # i = column, j = row
            tx, ty = float(cx - i), float(cy - j)
            rw = bw / anchors[l][k][0]
            rh = bh / anchors[l][k][1]
            tw = np.log(rw if rw >= 1e-3 else 1e-3)
            th = np.log(rh if rh >= 1e-3 else 1e-3)
            count = 0
            for ki in range(-1, 2):
                kii = i + ki
                for kj in range(-1, 2):
                    kjj = j + kj
                    if kii < 0 or kii >= grid_shapes[l][0]:
                        continue
                    if kjj < 0 or kjj >= grid_shapes[l][1]:
                        continue
                    if y_true[l][b, kjj, kii, 4] == 1 and count >= 3:      # skip rule (:3463)
                        continue
                    y_true[l][b, kjj, kii] *= 0
                    y_true[l][b, kjj, kii, 0:4] = [-ki + tx, -kj + ty, tw, th]
                    y_true[l][b, kjj, kii, 4] = 1.0
                    y_true[l][b, kjj, kii, 5 + k] = 1.0
                    y_true[l][b, kjj, kii, 5 + len(anchors[l]) + c] = 1.0
                    count += 1
    return y_true
